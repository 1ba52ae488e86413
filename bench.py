#!/usr/bin/env python3
"""bench.py -- the headline measurement of the accelerated path (BASELINE.json: "Msamples/s + Mrays/s, Sponza 262k tri
1080p/512spp, 1/2/4/8 GPU").

A "step" is one pass of the hot path over one frame of synthetic input: the C3 workload (Sponza-class procedural atrium,
262 267 triangles, 1920x1080, 512 spp), scene and BVH already resident in HBM when the timed region starts:
    sol_clear -> sol_render (persistent path-tracing kernel + chunk resolve) -> sol_gather (N > 1: the per-rank tile
    accumulators travel to rank 0 over RCCL/xGMI inside the C ABI; then the un-permute into the row-major image).
N ranks (one process per GPU) shard the 8x8-pixel tiles of the frame round-robin. `python bench.py --gpus N` starts its own
N rank processes (before any GPU call; a failed rank makes the run exit non-zero); under torch.distributed.run
(RANK/WORLD_SIZE in the environment) it is one of the ranks. The ONLY communicator is the one the C ABI owns (sol_comm_init);
the ranks' rendezvous - the 128-byte id, the barriers around the timed region, the max-over-ranks time - goes through a
key-value store (the parent's socket, or the launcher's TCPStore), not through a second RCCL process group. Two scaling modes:
  --scaling strong (default: the metric's own definition - the SAME 1080p x 512 spp job on 1/2/4/8 GPUs): total work fixed;
  --scaling weak: per-GPU work fixed (512*N spp on the shared frame).
`value` is the whole-job aggregate Msamples/s = W*H*spp_total / step time (max over ranks).

The JSON line also carries
  roofline      : algorithmic bytes of the dominant kernel (sol_render_kernel) per launch / its HIP-event duration, against
                  the 8 TB/s HBM peak (DESIGN.md "Measurement"; bytes per sample come from a counter-enabled run). NOTE: these
                  bytes are served almost entirely by L1/L2/Infinity Cache - `frac` is the contract's algorithmic figure, not
                  HBM utilisation; `traffic` (measured L2<->fabric bytes per launch, from rocprofv3 --pmc passes of this same
                  run, N=1) and `traffic_frac` say what reaches the memory side;
  roofline_valu : what actually bounds the kernel - vector-instruction issue x lane utilisation, from a PMC pass;
  cpu_baseline  : the f64 CPU restatement of the reference algorithm (oracle/, "port") timed on this box's host cores on a
                  bounded sample of the same frame (rank 0, N=1 only).
"""
import argparse
import csv
import glob
import json
import os
import pickle
import resource
import shutil
import socket
import subprocess
import sys
import tempfile
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "solstrale-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0x5017A1E
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
N_SIMD = 256 * 4
KERNEL = os.environ.get("SOL_BENCH_KERNEL", "sol_render_kernel<false")  # the uncounted product kernel (probes at scene creation run the counted variant); env: A/B runs of another kernel


def algorithmic_bytes(st, sizes):
    """Bytes the kernel must touch, from exact counters (DESIGN.md "Measurement"): every BVH node fetched, every primitive
    record tested, shading + material records per scatter, texels, and the accumulator traffic."""
    return (st["node_visits"] * sizes["node"] + st["triangle_tests"] * sizes["triangle"] + st["quad_tests"] * sizes["quad"] +
            st["sphere_tests"] * sizes["sphere"] + st["shades"] * (sizes["triangle_shade"] + sizes["material"]) +
            st["texel_fetches"] * 3 + st["samples"] * 12.0 / 16.0 * 3.0)  # 12 B chunk sum written, read, and added per 16 samples


def host_cores():
    """Cores this process may actually use: affinity mask, capped by a cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return n


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c4", "c5", "profiling"],
                    help="c1..c5: the BASELINE.json configs (default c3, the one the metric is quoted on); profiling: the one workload the reference "
                         "itself defines (src/bin/profiling.rs:14-37: its test scene, 800x400, 1000 spp)")
    ap.add_argument("--camera-preset", default="default", choices=["default", "interior", "closeup"],
                    help="STRESS variants of the stand-ins, reported beside the headline, never instead of it: c3/c4 'interior' (under the gallery, "
                         "looking along the colonnade: every camera ray hits), c5 'closeup' (the statue fills the frame, glass in front of metal)")
    ap.add_argument("--mesh-preset", default="regular", choices=["regular", "heterogeneous"],
                    help="c3/c4: 'heterogeneous' = the same atrium with the triangle statistics of a hand-modelled asset (large wall triangles, "
                         "long thin rails and rods, rotated drapes and arches, ornament clusters 1000x smaller): the STRESS mesh for the tree "
                         "builder, reported beside the headline, never instead of it")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N>1: strong = the workload's total spp split by tiles over the ranks (default); weak = that spp per GPU")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel of the job (strong) / per GPU (weak); default: the workload's")
    ap.add_argument("--obj", default="", help="render this OBJ (+MTL/textures beside it; e.g. the real sponza.obj) instead of the "
                                              "procedural stand-in; config.workload then names the file")
    ap.add_argument("--camera", default="", help="with --obj: fx,fy,fz,tx,ty,tz[,vfov] (look_from, look_at)")
    ap.add_argument("--light", default="", help="with --obj: qx,qy,qz,ux,uy,uz,vx,vy,vz[,r,g,b] Quad light (corner, two edges)")
    ap.add_argument("--hdri", action="store_true", help="c5 only: add the procedural HDR environment map (EXTENSION: the reference has no "
                                                        "environment lights; reported separately from the plain c5 line)")
    ap.add_argument("--partition", default="modulo", choices=["balanced", "modulo"],
                    help="N>1: which rank owns an 8x8 block - 'modulo' (default): block b -> rank b mod N; 'balanced': dealt out by the "
                         "blocks' cost in the creation probe (one-GPU estimate at 8 ranks: 7.78x instead of 7.74x). The table path has run "
                         "through sol_gather over the test transport only; it becomes the default once a real multi-GPU run has passed its "
                         "frame check (ADVICE r03)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc passes (roofline.traffic = null, no roofline_valu)")
    ap.add_argument("--no-all-traced", action="store_true", help="skip the extra steps that time the job with every sample traced (background_blocks."
                    "value_with_every_sample_traced = null); the counter passes use it: they must see the timed launch and nothing else")
    ap.add_argument("--pmc-spp", type=int, default=64)
    ap.add_argument("--no-build", action="store_true", help="do not run the build step (use under rocprofv3: no child processes)")
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dry run on a box with ONE GPU: every rank uses cuda:0 and goes through the same sol_comm_init / "
                         "sol_gather calls as a real run, over the test-only transport stub tests/stub_rccl (RCCL refuses two "
                         "ranks on one device); its throughput means nothing and the JSON line says so")
    return ap.parse_args()


# ---- rendezvous of the ranks: a tiny key-value store (set / blocking get / add) ------------------------------------------------
class _StoreServer(threading.Thread):
    """The parent of self-launched ranks serves the store on a socket it binds BEFORE the ranks start and keeps open."""

    def __init__(self):
        super().__init__(daemon=True)
        self.sock = socket.socket()
        self.sock.bind(("127.0.0.1", 0))
        self.sock.listen(64)
        self.port = self.sock.getsockname()[1]
        self.data, self.cv = {}, threading.Condition()

    def run(self):
        while True:
            try:
                conn, _ = self.sock.accept()
            except OSError:
                return
            threading.Thread(target=self.serve, args=(conn,), daemon=True).start()

    def serve(self, conn):
        f = conn.makefile("rwb")
        try:
            while True:
                try:
                    op, key, val = pickle.load(f)
                except EOFError:
                    return
                with self.cv:
                    if op == "set":
                        self.data[key] = val
                        self.cv.notify_all()
                        out = None
                    elif op == "add":
                        out = self.data[key] = self.data.get(key, 0) + val
                        self.cv.notify_all()
                    else:  # get: blocks until the key exists
                        self.cv.wait_for(lambda: key in self.data)
                        out = self.data[key]
                pickle.dump(out, f)
                f.flush()
        except (OSError, pickle.PickleError):
            return


class _SockStore:
    def __init__(self, host, port):
        self.f = socket.create_connection((host, port), timeout=600).makefile("rwb")

    def _call(self, op, key, val=None):
        pickle.dump((op, key, val), self.f)
        self.f.flush()
        return pickle.load(self.f)

    def set(self, key, val):
        self._call("set", key, val)

    def get(self, key):
        return self._call("get", key)

    def add(self, key, n):
        return self._call("add", key, n)


class _TorchStore:
    """Under torch.distributed.run: the launcher's TCPStore (c10d key-value store; no process group, no RCCL)."""

    def __init__(self, rank, world):
        from datetime import timedelta
        from torch.distributed import TCPStore
        host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
        agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "") == "True"  # the agent already serves MASTER_PORT
        self.s = TCPStore(host, port, world, is_master=(rank == 0 and not agent), timeout=timedelta(seconds=600), wait_for_workers=False)

    def set(self, key, val):
        self.s.set("solbench/" + key, pickle.dumps(val))

    def get(self, key):
        self.s.wait(["solbench/" + key])
        return pickle.loads(self.s.get("solbench/" + key))

    def add(self, key, n):
        return int(self.s.add("solbench/" + key, n))


class Coord:
    """What the ranks need from each other outside the data path: one broadcast of bytes, barriers, the maximum of a float."""

    def __init__(self, rank, world):
        self.rank, self.world, self.n = rank, world, 0
        self.store = None
        if world > 1:
            self.store = _SockStore(*os.environ["SOLBENCH_STORE"].rsplit(":", 1)) if os.environ.get("SOLBENCH_STORE") else _TorchStore(rank, world)

    def _next(self, what):
        self.n += 1
        return f"{what}{self.n}"

    def bcast(self, data=None):
        if self.world == 1:
            return data
        key = self._next("bcast")
        if self.rank == 0:
            self.store.set(key, data)
        return self.store.get(key)

    def barrier(self):
        if self.world == 1:
            return
        key = self._next("barrier")
        if self.store.add(key, 1) == self.world:
            self.store.set(key + "/open", True)
        self.store.get(key + "/open")

    def max(self, x):
        if self.world == 1:
            return x
        key = self._next("max")
        self.store.set(f"{key}/{self.rank}", float(x))
        return max(self.store.get(f"{key}/{r}") for r in range(self.world))

    def all(self, x):
        """Every rank's value, by rank."""
        if self.world == 1:
            return [x]
        key = self._next("all")
        self.store.set(f"{key}/{self.rank}", x)
        return [self.store.get(f"{key}/{r}") for r in range(self.world)]


# ---- N > 1 without a launcher: this process becomes the parent of N rank processes ----------------------------------------
def launch_ranks(args):
    """Starts N copies of this script, one per GPU, BEFORE anything here touches the GPU. Rank 0 prints the JSON line (its
    stdout is ours); any rank failing fails the run (the others are terminated: a lost rank would hang their collective)."""
    if not args.no_build:
        import __graft_entry__
        __graft_entry__.build_product()  # once, here: the ranks must not race on the build (compiles and dlopens only, no GPU call)
    server = _StoreServer()  # (the listening socket exists from here on: no port to lose between choosing and using it)
    server.start()
    extra = {}
    if args.rehearse:  # the one-GPU transport stub in front of the real librccl.so.1, for the rank processes only
        stub = os.path.join(ROOT, "tests", "stub_rccl")
        subprocess.check_call(["make", "-s", "-C", stub])
        extra["LD_LIBRARY_PATH"] = os.path.join(stub, "_build") + os.pathsep + os.environ.get("LD_LIBRARY_PATH", "")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   SOLBENCH_STORE=f"127.0.0.1:{server.port}", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
        cmd = [sys.executable, os.path.abspath(__file__)] + [a for a in sys.argv[1:] if a != "--no-build"] + ["--no-build"]
        procs.append(subprocess.Popen(cmd, env=env, stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        live = list(procs)
        while live:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in live:
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ---- rocprofv3 --pmc passes of this same workload (N = 1) -------------------------------------------------------------------
PMC_GROUPS = [  # FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: separate passes (MI355X_MICROARCH.md, rocprofv3 PMC slots)
    ["FETCH_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES"],
    ["WRITE_SIZE", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"],
]


def pmc_passes(args, spp):
    """Runs this script under `rocprofv3 --pmc` (child processes; one pass per counter group, one timed launch of `spp` samples
    per pixel each) and returns ({counter: sum over the product kernel's dispatches}, mean kernel ns) or (None, reason)."""
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    out = tempfile.mkdtemp(prefix="solbench_pmc_")
    counters, dur = {}, []
    try:
        for grp in PMC_GROUPS:
            d = os.path.join(out, grp[0])
            cmd = [rocprof, "--pmc"] + grp + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable,
                                              os.path.abspath(__file__), "--workload", args.workload, "--spp", str(spp), "--steps", "1",
                                              "--warmup", "0", "--no-cpu-baseline", "--no-pmc", "--no-build", "--no-all-traced"]
            if args.hdri:
                cmd += ["--hdri"]
            if args.camera_preset != "default":
                cmd += ["--camera-preset", args.camera_preset]
            if args.mesh_preset != "regular":
                cmd += ["--mesh-preset", args.mesh_preset]
            if args.obj:
                cmd += ["--obj", args.obj] + (["--camera", args.camera] if args.camera else []) + (["--light", args.light] if args.light else [])
            try:
                r = subprocess.run(cmd, cwd=tempfile.gettempdir(), env=dict(os.environ, TMPDIR=tempfile.gettempdir()),
                                   capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                return None, f"rocprofv3 pass {grp[0]} timed out"
            if r.returncode != 0:
                return None, f"rocprofv3 pass {grp[0]} failed (rc {r.returncode}): {(r.stderr or '')[-300:]}"
            seen = set()
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if KERNEL not in row["Kernel_Name"]:
                        continue
                    counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                    if (grp[0], row["Dispatch_Id"]) not in seen:
                        seen.add((grp[0], row["Dispatch_Id"]))
                        dur.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
            if not seen:
                return None, f"rocprofv3 pass {grp[0]}: no dispatch of {KERNEL} in the output"
    finally:
        shutil.rmtree(out, ignore_errors=True)
    return counters, sum(dur) / len(dur)


def make_scene(args, spp_total):
    from solstrale_amd import RenderConfig, scenes
    wl = args.workload
    w, h, spp0 = {"c1": (400, 400, 50), "c2": (1920, 1080, 256), "c3": (1920, 1080, 512), "c4": (3840, 2160, 1024),
                  "c5": (1920, 1080, 2048), "profiling": (800, 400, 1000)}[wl]
    preset = args.camera_preset
    if preset != "default" and not ((wl in ("c3", "c4") and preset == "interior") or (wl == "c5" and preset == "closeup")) or (preset != "default" and args.obj):
        raise SystemExit(f"--camera-preset {preset} does not exist for workload {wl}")
    if args.mesh_preset != "regular" and (wl not in ("c3", "c4") or args.obj):
        raise SystemExit(f"--mesh-preset {args.mesh_preset} does not exist for workload {wl}")
    if args.obj:
        cam = light = None
        if args.camera:
            c = [float(x) for x in args.camera.split(",")]
            cam = (c[0:3], c[3:6], c[6] if len(c) > 6 else 55.)
        if args.light:
            l = [float(x) for x in args.light.split(",")]
            light = (l[0:3], l[3:6], l[6:9], tuple(l[9:12]) if len(l) >= 12 else (18., 17., 15.))
        make = lambda rc: scenes.obj_file_scene(args.obj, rc, cam, light)
        name = f"{wl.upper()} shape ({w}x{h}) on the supplied OBJ file {os.path.basename(args.obj)} (host OBJ+MTL loader), 1 quad light + sky"
    elif wl in ("c3", "c4"):
        name = (f"{wl.upper()} Sponza-class procedural atrium (stand-in: the real sponza.obj is not available offline; --obj takes one), "
                f"{scenes.SPONZA_TRIANGLES} triangles, 24 Lambertian materials (8 image-textured), 1 quad light + sky" +
                (" - STRESS camera 'interior' (under the gallery along the colonnade; not the headline view)" if preset == "interior" else "") +
                (" - STRESS mesh 'heterogeneous' (large wall triangles, long thin rails and rods, rotated drapes and arches, ornament clusters "
                 "> 1000x smaller; not the headline mesh)" if args.mesh_preset == "heterogeneous" else ""))
        make = lambda rc: scenes.sponza_like(rc, camera=preset, mesh=args.mesh_preset)
    elif wl == "c5":
        name = (f"C5 statue-class displaced mesh (stand-in), ~{scenes.STATUE_TRIANGLES} triangles, Metal(0.1) + Dielectric(1.5), 1 quad light" +
                (" + procedural 2048x1024 HDR environment map (EXTENSION: not in the reference)" if args.hdri else "") +
                (" - STRESS camera 'closeup' (the statue fills the frame, seen from above; not the default view)" if preset == "closeup" else ""))
        make = lambda rc: scenes.statue_like(rc, environment=args.hdri, camera=preset)
    elif wl == "profiling":
        name = ("the reference's own profiling workload (src/bin/profiling.rs:14-37): create_test_scene (tests/scenes.rs:17-122: image texture, glass, "
                "ConstantMedium, nested BVH, sphere + quad + triangle lights, aperture 0.1), 800x400, 1000 spp")
        make = lambda rc: scenes.create_test_scene(rc)
    elif wl == "c2":
        name = "C2 Cornell box + 10000 Lambertian spheres"
        make = lambda rc: scenes.cornell_spheres(rc)
    else:
        name = "C1 Cornell box (18 quads)"
        make = lambda rc: scenes.cornell_box(rc)
    return w, h, spp0, name, make


def worker(args):
    import __graft_entry__
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.rehearse and world > 1 and not os.environ.get("SOLBENCH_STORE"):
        raise SystemExit("--rehearse is started by bench.py itself (python bench.py --gpus N --rehearse): its ranks need the transport stub on "
                         "LD_LIBRARY_PATH and must not load torch's RCCL")
    # torch is here for device memory, the stream and the synchronisation of the timed region; the rehearsal's ranks do without
    # it (importing torch loads torch's own RCCL, which the product's dlopen would then hand back instead of the stub)
    torch = None
    if not args.rehearse:
        import torch
    if rank == 0 and not args.no_build:
        __graft_entry__.build_product()  # (the product libraries; the A/B library and the oracle are built where they are used)
    coord = Coord(rank, world)
    coord.barrier()
    # every other rank only LOADS what rank 0 built (or what --no-build promises is there): no second writer in the build directory
    if args.rehearse:
        local_rank = 0
    from solstrale_amd import DeviceScene, RenderConfig, _abi, comm_unique_id, device_count, record_sizes
    if device_count() < 1:
        raise SystemExit("bench.py: no HIP device; the hot path has no CPU fallback")
    dev = None
    if torch is not None:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    w, h, spp0, name, make = make_scene(args, 0)
    spp_arg = args.spp or spp0
    spp = spp_arg * world if args.scaling == "weak" else spp_arg  # samples per pixel of the whole job
    t0 = time.time()
    scene = make(RenderConfig(w, h, spp))
    t_build = time.time() - t0
    t0 = time.time()
    ds = DeviceScene(scene, local_rank)
    t_upload = time.time() - t0
    bt = ds.build_times()
    tree_info = ds.info()
    rccl_ranks = None
    if world > 1:
        # the ONE communicator of the run lives behind the C ABI (sol_comm_init: ncclCommInitRank); the store only ships the id
        uid = coord.bcast(bytes(comm_unique_id()) if rank == 0 else None)
        if args.partition == "balanced":  # blocks dealt out by their cost in the creation probe (every rank derives the same table)
            ds.set_option(_abi.OPT_BALANCED_PARTITION, 1)
        ds.comm_init(rank, world, uid)
        rccl_ranks = world
        # every rank derives the partition table from its own creation probe: they must agree, or blocks are rendered twice / not at
        # all and the gather assembles a wrong frame without an error
        part = ds.info()
        parts = coord.all((part["partition_table"], part["partition_crc"]))
        if len(set(parts)) != 1:
            raise SystemExit(f"bench.py: the ranks disagree on the tile partition (table, checksum per rank: {parts})")
    acc = image = None
    image_ptr = 0  # (0: the scene's own image buffer)
    if torch is not None:
        ds.set_stream(torch.cuda.current_stream(dev).cuda_stream)
        n_acc = ds.accum_floats()
        acc = torch.zeros(n_acc, dtype=torch.float32, device=dev)
        ds.bind_accum(acc.data_ptr(), n_acc)
        if rank == 0:
            image = torch.empty(h * w * 3, dtype=torch.float32, device=dev)
            image_ptr = image.data_ptr()
    ds.kernel_timing(True)
    max_spp_call = max(16, ds.max_samples_per_call() // 16 * 16)  # a sol_render call handles < 2^32 work items

    def step():
        ds.clear()
        f = 0
        while f < spp:
            n = min(spp - f, max_spp_call)
            ds.render(f, n, SEED)
            f += n
        ds.gather(image_ptr)  # N > 1: collective (grouped ncclSend / ncclRecv into rank 0); always the un-permute on rank 0

    def fence():
        coord.barrier()
        if torch is not None:
            torch.cuda.synchronize(dev)
        else:
            ds.sync()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = coord.max(time.perf_counter() - t0)
    k_ms, grid = ds.last_kernel_ms()  # duration of the last step's (last) render kernel (HIP events on its stream)
    # what a scaling curve needs to explain itself: every rank's kernel time, and the gather timed on its own (one more, untimed,
    # gather of the accumulators the last step left: host clock around sol_gather + synchronise, maximum over the ranks)
    per_rank_kernel_ms = coord.all(round(k_ms, 3))
    per_rank_setup_s = coord.all((round(t_build, 3), round(t_upload, 3)))  # (every rank builds and creates its own copy of the scene)
    gather_ms = None
    if world > 1:
        fence()
        tg = time.perf_counter()
        ds.gather(image_ptr)
        fence()
        gather_ms = round(coord.max(time.perf_counter() - tg) * 1e3, 3)
    last_call_spp = spp - (spp - 1) // max_spp_call * max_spp_call
    # The same job with EVERY sample generated and traced (SOL_OPT_BACKGROUND_BLOCKS 0), timed the same way after the timed region of
    # `value` (one warm-up, then as many steps as `value` had; every N): so that the line carries both figures and nobody has to take the
    # skipped samples on trust - a scaling curve can be drawn from either.
    value_all_traced = None
    all_traced_steps = max(1, args.steps)  # the same number of steps as `value` (round-4 advisor: both figures with the same step count)
    if tree_info["background_blocks"] > 0 and not args.no_all_traced:  # (every rank of a job finds the same blocks: the scene and the proof are deterministic)
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 0)
        step()
        fence()
        ta = time.perf_counter()
        for _ in range(all_traced_steps):
            step()
        fence()
        value_all_traced = float(w) * h * spp / (coord.max(time.perf_counter() - ta) / all_traced_steps) / 1e6  # (N > 1: maximum over the ranks, like `value`)
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 1)

    # ---- counters: exact per-sample algorithmic bytes and rays from a counter-enabled run of the same kernels ----
    frame = ds.read_image() if (rank == 0 and args.rehearse) else None  # the gathered frame of the last step
    if torch is not None:
        ds.bind_accum(0, 0)
    ds.kernel_timing(False)
    c_spp = min(16, spp)
    ds.clear()
    ds.render(0, c_spp, SEED, counted=True)
    st = ds.stats()
    sizes = record_sizes()
    rays_per_sample = st["rays"] / st["samples"]  # of the whole algorithm: a counted render traces every sample
    launch_samples = st["samples"] // c_spp * last_call_spp  # samples of the launch k_ms belongs to (this rank)
    pst = ds.path_stats()
    # Background blocks (SolSceneInfo::background_blocks: blocks proved at scene creation to see only the constant background are summed,
    # not traced). What the timed launches touched is counted by a second counted render that skips them as a plain render does: the
    # roofline's algorithmic bytes and the Mrays/s below are those of the rays actually traced, never of rays that were not.
    st_traced = st
    if tree_info["background_blocks"] > 0 and not os.environ.get("SOL_BACKGROUND_BLOCKS") == "0":
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 2)
        ds.clear()
        ds.render(0, c_spp, SEED, counted=True)
        st_traced = ds.stats()
        ds.set_option(_abi.OPT_BACKGROUND_BLOCKS, 1)
    background_samples = st["samples"] - st_traced["samples"]
    bytes_per_sample = (algorithmic_bytes(st_traced, sizes) + background_samples * 12.0 / 16.0 * 3.0) / st["samples"]
    traced_rays_per_sample = st_traced["rays"] / st["samples"]

    out = None
    if rank == 0:
        total_samples = float(w) * h * spp
        ms_per_step = dt / args.steps * 1e3
        value = total_samples / (dt / args.steps) / 1e6
        achieved = bytes_per_sample * launch_samples / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": args.scaling,  # (at N = 1 both modes are the same job)
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if not args.obj else "user-supplied OBJ",
            "config": {"workload": name, "width": w, "height": h, "spp_total": spp, "pixel_samples_per_gpu": int(w * h * spp // world),
                       "seed": SEED, "max_depth": 50, "scaling_mode": args.scaling,
                       "sharding": f"8x8 tiles " + ("dealt out by probe cost" if (world > 1 and args.partition == "balanced") else "round-robin") + f" over {world} rank(s); every rank renders its tiles with all {spp} spp; "
                                   f"gather to rank 0 " + ("inside the C ABI (sol_gather: grouped ncclSend/ncclRecv)" + (" over the one-GPU test transport stub" if args.rehearse else "")
                                                           if world > 1 else "(single rank: un-permute only)")},
            "mrays_per_s": round(value * traced_rays_per_sample, 2),
            "rays_per_sample": round(rays_per_sample, 4),
            "traced_rays_per_sample": round(traced_rays_per_sample, 4),
            "background_blocks": {"blocks": tree_info["background_blocks"], "sample_fraction": round(background_samples / st["samples"], 4),
                                  "value_with_every_sample_traced": None if value_all_traced is None else round(value_all_traced, 2),
                                  "value_with_every_sample_traced_steps": all_traced_steps if value_all_traced is not None else None,
                                  "note": "8x8 pixel blocks of which sol_scene_create proved that no camera ray of theirs, whatever the jitter, comes near a "
                                          "primitive's box: every sample is the background colour, summed in the reference's order without being traced "
                                          "(frames bit-identical with SOL_OPT_BACKGROUND_BLOCKS 0; SOL_BACKGROUND_BLOCKS=0 in the environment switches the proof off). "
                                          "mrays_per_s, traced_rays_per_sample and the roofline's algorithmic bytes count only what was traced; rays_per_sample, the "
                                          "histogram and primary_hit_fraction describe the whole algorithm (one camera ray per background sample)"},
            "node_visits_per_ray": round(st_traced["node_visits"] / max(1, st_traced["rays"]), 3),
            "primitive_tests_per_ray": round((st_traced["triangle_tests"] + st_traced["quad_tests"] + st_traced["sphere_tests"]) / max(1, st_traced["rays"]), 3),
            "primary_hit_fraction": round(pst["primary_hit_fraction"], 4),
            "rays_per_path_histogram": {k: round(v, 4) for k, v in pst["rays_per_path_histogram"].items()},
            "rays_per_sample_note": "the open-roofed atrium ends most paths on the sky after ~3 rays; a closed interior costs several "
                                    "times more rays per sample - Mrays/s is the figure that transfers between scenes",
            # `bound`: the contract's figure is the ALGORITHMIC-bytes rate against the HBM peak and is named so - the kernel is not
            # HBM-bound (its records are served by L1 / L2 / Infinity Cache); what limits it is `binding` (round-4 review: a reader of one field
            # must not take "hbm" for the limit)
            "roofline": {"bound": "algorithmic_hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": None, "traffic_source": None,
                         "binding": "valu_issue", "binding_achieved": None, "binding_peak": None, "binding_unit": None, "binding_frac": None,
                         "binding_note": "what actually limits the kernel: vector-instruction issue x lane utilisation (roofline_valu, mirrored here when the "
                                         "PMC passes ran); the hbm figures above are the contract's algorithmic-bytes bound, not the limit",
                         "frac_note": "ALGORITHMIC bytes (every node / primitive / texel record the search touches) per second; they are "
                                      "served by L1/L2/Infinity Cache, so this is NOT HBM utilisation (it exceeds what an HBM copy reaches, 6.3 TB/s, "
                                      "and may pass 1.0) - see traffic_frac for what reaches the memory side and binding / roofline_valu for the limit",
                         "kernel": "sol_render_kernel", "kernel_ms": round(k_ms, 3), "grid_blocks": grid,
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                         "samples_per_launch": int(launch_samples),
                         "counters_per_sample": {k: round(v / st["samples"], 4) for k, v in st_traced.items() if k not in ("samples", "max_stack")}},
            "setup_s": {"scene_and_bvh_build_host": round(t_build, 2), "sol_scene_create": round(t_upload, 2),
                        "scene_source": ("OBJ + MTL file through the host loader (parse, material table, textures, reference BVH, flatten)" if args.obj else
                                         "procedural stand-in generated in Python, reference BVH built by the C++ host"),
                        "peak_host_rss_mb": round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0, 1),
                        **{"create_" + k: round(v, 3) for k, v in bt.items()}},
            "world_tree": {"builder": tree_info["tree_name"], "presplit_extra_references": tree_info["split_references"],
                           "presplit_triangles": tree_info["split_triangles"], "presplit_box_area_ratio": round(tree_info["split_area_ratio"], 4),
                           "reinsertion_moves": tree_info["reinsertion_moves"], "reinsertion_area_ratio": round(tree_info["reinsertion_area_ratio"], 4),
                           "strict_triangles": tree_info["strict_triangles"],
                           "note": "pre-splitting is kept only when it shrinks the summed box area of the primitives below 0.85 (regular meshes stay unsplit); "
                                   "strict_triangles: the scene has needle triangles (aspect >= 32:1) - fatter box pad and the triangle consistency rule of the fp32 contract"},
        }
        if world > 1:
            out["rccl_ranks"] = rccl_ranks
            out["per_rank_kernel_ms"] = per_rank_kernel_ms
            out["per_rank_setup_s"] = {"scene_and_bvh_build_host": [a for a, _ in per_rank_setup_s], "sol_scene_create": [b for _, b in per_rank_setup_s],
                                       "note": "every rank loads / generates the scene and calls sol_scene_create itself (deterministic: same tree, same partition - checked); "
                                               "outside the timed region, as the metric defines (SURVEY.md 8d)"}
            out["gather_ms"] = gather_ms
            out["gather_note"] = "one extra sol_gather (grouped ncclSend / ncclRecv into rank 0 + un-permute) after the timed region, host clock incl. the barrier; the timed steps contain their own"
            out["partition"] = {"table": bool(part["partition_table"]), "crc": part["partition_crc"], "agreed_by_all_ranks": True}
        if args.rehearse:
            out["rehearsal"] = "all ranks on cuda:0, sol_gather over the test-only transport stub (tests/stub_rccl): not a measurement"
    if args.rehearse and world > 1:
        # the assembled frame must equal what one rank renders alone (the image is a pure function of scene and seed)
        coord.barrier()
        ds.comm_destroy()
        if rank == 0:
            import numpy as np
            ds.set_partition(0, 1)
            ds.clear()
            ds.render(0, spp, SEED)
            out["rehearsal_frame_check"] = bool(np.array_equal(ds.read(), frame))
    ds.close()
    del ds
    if rank == 0 and world == 1 and not args.no_pmc:
        p_spp = min(args.pmc_spp, spp)
        counters, info = pmc_passes(args, p_spp)
        rf = out["roofline"]
        if counters is None:
            rf["traffic_source"] = f"not measured: {info}"
        else:
            p_samples = launch_samples / last_call_spp * p_spp
            g = counters.get
            # FETCH_SIZE / WRITE_SIZE are KiB at the L2<->fabric boundary (Infinity-Cache hits included); gfx950 tallies a
            # 128-B read request as 64 B: reads x2 (guide, "HBM")
            fabric_per_sample = (g("FETCH_SIZE", 0.0) * 2048.0 + g("WRITE_SIZE", 0.0) * 1024.0) / p_samples
            rf["traffic"] = int(fabric_per_sample * launch_samples)
            rf["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE (x2 gfx950 correction) + WRITE_SIZE, separate passes of this run's workload at "
                                    f"{p_spp} spp, scaled per sample to this launch; L2<->fabric bytes, Infinity-Cache hits included")
            rf["traffic_frac"] = round(rf["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5)
            # the x2 is the guide's correction for wide coalesced reads; a divergent 16-byte gather is counted as it is (one 64-byte
            # request per miss: profiles/r04_counter_questions.txt), and this kernel's reads are such gathers: the truth lies between
            rf["traffic_uncorrected"] = int((g("FETCH_SIZE", 0.0) * 1024.0 + g("WRITE_SIZE", 0.0) * 1024.0) / p_samples * launch_samples)
            rf["traffic_frac_range"] = [round(rf["traffic_uncorrected"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5), rf["traffic_frac"]]
            clock = g("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (info * 1e-9) if g("GRBM_GUI_ACTIVE") else 2.4e9
            k_cycles = info * 1e-9 * clock
            lane_util = g("SQ_THREAD_CYCLES_VALU", 0.0) / max(1.0, g("SQ_ACTIVE_INST_VALU", 0.0) * 64.0)
            # SQ_* cycle counters count quad-cycles: ACTIVE_INST_VALU x 4 = cycles in which some wave executes a vector instruction;
            # one SIMD accepts at most one wave64 vector instruction per 4 cycles from a single wave and 2 cycles overall (SIMD-32)
            busy4 = g("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (N_SIMD * k_cycles)
            inst_per_s = g("SQ_INSTS_VALU", 0.0) / (info * 1e-9)
            peak_inst = N_SIMD * clock / 2.0
            rf["binding_achieved"] = round(inst_per_s * lane_util / 1e9, 2)
            rf["binding_peak"] = round(peak_inst / 1e9, 2)
            rf["binding_unit"] = "G wave64-instr/s x lane utilisation"
            rf["binding_frac"] = round(inst_per_s * lane_util / peak_inst, 5)
            out["roofline_valu"] = {
                "bound": "valu_issue", "achieved": round(inst_per_s * lane_util / 1e9, 2), "peak": round(peak_inst / 1e9, 2),
                "unit": "G wave64-instr/s x lane utilisation", "frac": round(inst_per_s * lane_util / peak_inst, 5),
                "valu_instr_per_sample": round(g("SQ_INSTS_VALU", 0.0) / p_samples, 1), "lane_utilisation": round(lane_util, 4),
                "issue_busy_at_4_cycles_per_instr": round(busy4, 4), "issue_frac_of_2_cycle_peak": round(inst_per_s / peak_inst, 4),
                "issue_note": "SQ_ACTIVE_INST_VALU x 4 cycles / SIMD cycles: 1.0 = every vector instruction took one 4-cycle issue slot and every slot was "
                              "taken; above 1.0 (since the build without packed fp32 instructions) the plain fp32 multiplies / adds issue at their faster rate",
                "wave_cycles_waiting_frac": round(g("SQ_WAIT_ANY", 0.0) / max(1.0, g("SQ_WAVE_CYCLES", 0.0)), 4),
                "clock_ghz": round(clock / 1e9, 3), "kernel_ms_under_pmc": round(info * 1e-6, 3), "pmc_spp": p_spp}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            import orc  # TEST INFRASTRUCTURE, used here only as the timed CPU baseline
            t0 = time.time()
            cores = host_cores()
            _, ost = orc.render(scene, 0, 1, SEED, real=orc.ORC_F64, threads=cores)
            t1 = time.time() - t0
            n_spp = int(max(1, min(64, args.cpu_seconds / max(t1, 1e-3))))
            t0 = time.time()
            _, ost = orc.render(scene, 1, n_spp, SEED, real=orc.ORC_F64, threads=cores)
            tc = time.time() - t0
            cpu_v = w * h * n_spp / tc / 1e6
            out["cpu_baseline"] = {"value": round(cpu_v, 4), "unit": "Msamples/s", "cores": int(ost["threads"]), "kind": "port",
                                   "sample": f"full {w}x{h} frame, {n_spp} spp ({w * h * n_spp} samples, {tc:.1f} s), f64 restatement of the "
                                             f"reference algorithm (reference-order BVH search, no culling), row-parallel std::thread",
                                   "mrays_per_s": round(ost["rays"] / tc / 1e6, 3), "gpu_over_cpu": round(value / cpu_v, 1)}
        print(json.dumps(out), flush=True)
    coord.barrier()


def main():
    # RCCL / device-memory sharing between processes needs dmabuf IPC on this driver; the same setting for launcher-started ranks as
    # for self-launched ones (it must be in the environment before the HIP runtime starts in this process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))
    worker(args)


if __name__ == "__main__":
    main()
