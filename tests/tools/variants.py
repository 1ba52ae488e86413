"""Kernel A/B harness (not a pytest). Builds variants of the device library with extra compiler flags into _var/<name>/ (here,
cross-compiled, no GPU needed) and - on the GPU box - times each one with perf_quick.py in its own process
(SOLSTRALE_BUILD_DIR selects the library). Every frame CRC must stay the same: images do not depend on scheduling.
  python tests/tools/variants.py build name1="-DSOL_X=1 -DSOL_Y=2" name2="..."     (CPU container)
  python tests/tools/variants.py run [--spp N] [--scenes "c2 c3"] name1 name2 ...   (GPU box; `default` = _build/)
"""
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
VAR = os.path.join(ROOT, "_var")


def build(specs):
    sys.path.insert(0, os.path.join(ROOT, "solstrale-rust_amd"))
    import build as b
    for spec in specs:
        name, _, flags = spec.partition("=")
        b.build(extra_hip_flags=shlex.split(flags), out_dir=os.path.join(VAR, name))


def run(names, spp, scenes):
    """Exit code != 0 when a requested variant is missing or its run failed: a sweep must not silently time nothing."""
    failed = []
    for name in names:
        env = dict(os.environ)
        if name != "default":
            env["SOLSTRALE_BUILD_DIR"] = os.path.join(VAR, name)
            if not os.path.exists(os.path.join(VAR, name, "libsolstrale_hip.so")):
                print(f"== {name}\nMISSING: {os.path.join(VAR, name)}/libsolstrale_hip.so (build it first: variants.py build {name}=..)", flush=True)
                failed.append(name)
                continue
        print(f"== {name}", flush=True)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "perf_quick.py")] + scenes + ["--spp", str(spp)],
                           env=env, capture_output=True, text=True, timeout=900)
        print(r.stdout.strip(), flush=True)
        if r.returncode != 0:
            print(f"FAILED (rc {r.returncode}): {r.stderr[-800:]}", flush=True)
            failed.append(name)
    if failed:
        print(f"variants that did not run: {' '.join(failed)}", flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        a = sys.argv[2:]
        spp, scenes = 64, ["c2", "c3"]
        if "--spp" in a:
            i = a.index("--spp"); spp = int(a[i + 1]); del a[i:i + 2]
        if "--scenes" in a:
            i = a.index("--scenes"); scenes = a[i + 1].split(); del a[i:i + 2]
        sys.exit(run(a, spp, scenes))
