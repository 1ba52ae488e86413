"""Builds the shared libraries in-tree (solstrale-rust_amd/_build/):

  libsolstrale_hip.so   hand-written HIP for gfx950 + the C ABI of include/solstrale_hip.h   (hipcc)
  libsolstrale_host.so  C++ host mirror of the reference's Scene / ray_trace surface          (g++)

hipcc cross-compiles gfx950 code objects without a GPU. -ffp-contract=off keeps the device arithmetic the plain
IEEE sequence the parity tests pin (DESIGN.md "fp32 arithmetic contract").
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SRC = ["csrc/sol_render.hip", "csrc/sol_wavefront.hip", "csrc/sol_aux.hip", "csrc/sol_api.cpp"]
HIP_DEPS = HIP_SRC + ["csrc/sol_types.h", "csrc/sol_math.h", "csrc/sol_trace.h", "csrc/sol_shade.h", "csrc/sol_path.h", "csrc/sol_launch.h", "csrc/sol_tree.h",
                      "../include/solstrale_hip.h"]
HOST_SRC = ["host/solstrale_host.cpp", "host/solstrale_obj.cpp", "host/solstrale_host_c.cpp"]
HOST_DEPS = HOST_SRC + ["host/solstrale.hpp", "../include/solstrale_hip.h", "../include/solstrale_host.h"]

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-value"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(HERE, d)) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=HERE)


def build(force=False, extra_hip_flags=()):
    os.makedirs(BUILD, exist_ok=True)
    hip_lib = os.path.join(BUILD, "libsolstrale_hip.so")
    host_lib = os.path.join(BUILD, "libsolstrale_host.so")
    if force or _stale(hip_lib, HIP_DEPS + ["build.py"]):
        _run([HIPCC] + HIP_FLAGS + list(extra_hip_flags) + HIP_SRC + ["-o", hip_lib])
    if force or _stale(host_lib, HOST_DEPS + ["build.py"]) or os.path.getmtime(hip_lib) > os.path.getmtime(host_lib):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wextra", "-pthread"] + HOST_SRC +
             ["-o", host_lib, "-L" + BUILD, "-lsolstrale_hip", "-Wl,-rpath,$ORIGIN"])
    return hip_lib, host_lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
