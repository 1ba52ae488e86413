#!/usr/bin/env python3
"""bench.py -- the headline measurement of the accelerated path (BASELINE.json: "Msamples/s + Mrays/s, Sponza 262k tri
1080p/512spp, 1/2/4/8 GPU").

A "step" is one pass of the hot path over one frame of synthetic input: the C3 workload (Sponza-class procedural atrium,
262 267 triangles, 1920x1080, 512 spp per GPU), scene and BVH already resident in HBM when the timed region starts:
    sol_clear -> sol_render (persistent path-tracing kernel + chunk resolve) -> gather of the per-rank tile accumulators to
    rank 0 (RCCL over xGMI through torch.distributed when N > 1) -> sol_unpermute on rank 0.
N ranks (one process per GPU, launched by torch.distributed.run) shard the 8x8-pixel tiles of the frame round-robin; the
per-GPU work is fixed (each rank renders its tiles with 512*N spp), so scaling is "weak" and `value` is the whole-job
aggregate Msamples/s = N * 1920*1080*512 / step time.

The JSON line also carries
  roofline     : algorithmic bytes of the dominant kernel (sol_render_kernel) per launch / its HIP-event duration, against
                 the 8 TB/s HBM peak (DESIGN.md "Measurement"; bytes per sample come from a counter-enabled run);
  cpu_baseline : the f64 CPU restatement of the reference algorithm (oracle/, "port") timed on this box's host cores on a
                 bounded sample of the same frame (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "solstrale-rust_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

SEED = 0x5017A1E
HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=["c1", "c2", "c3", "c4", "c5"])
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel per GPU (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--rehearse", action="store_true",
                    help="N>1 dry run on a box with ONE GPU: every rank uses cuda:0, the gather goes through gloo on host "
                         "copies (RCCL refuses two ranks on one device); exercises partition, gather protocol and un-permute, "
                         "its throughput means nothing and the JSON line says so")
    args = ap.parse_args()

    import torch
    import __graft_entry__
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if rank == 0:
        __graft_entry__.build()
    import torch.distributed as dist
    if args.rehearse:
        local_rank = 0
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist.barrier()
    if rank != 0:
        __graft_entry__.build()  # no-op when up to date; loads the libraries
    from solstrale_amd import DeviceScene, RenderConfig, device_count, record_sizes, scenes, tiles
    if device_count() < 1:
        raise SystemExit("bench.py: no HIP device; the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    if args.workload == "c3":
        w, h, spp0 = 1920, 1080, 512
        name = f"C3 Sponza-class procedural atrium, {scenes.SPONZA_TRIANGLES} triangles, 24 Lambertian materials (8 image-textured), 1 quad light"
        make = lambda rc: scenes.sponza_like(rc)
    elif args.workload == "c4":  # configs[3]: C3's scene at 4K, 1024 spp over 8 GPUs = 128 spp of per-GPU work
        w, h, spp0 = 3840, 2160, 128
        name = f"C4 Sponza-class procedural atrium, {scenes.SPONZA_TRIANGLES} triangles, 4K (BASELINE: 1024 spp over 8 GPUs = 128 spp per GPU)"
        make = lambda rc: scenes.sponza_like(rc)
    elif args.workload == "c5":  # configs[4]: 2048 spp over 8 GPUs = 256 spp of per-GPU work
        w, h, spp0 = 1920, 1080, 256
        name = f"C5 statue-class displaced mesh, ~{scenes.STATUE_TRIANGLES} triangles, Metal(0.1) + Dielectric(1.5), 1 quad light (BASELINE: 2048 spp over 8 GPUs = 256 spp per GPU)"
        make = lambda rc: scenes.statue_like(rc)
    elif args.workload == "c2":
        w, h, spp0 = 1920, 1080, 256
        name = "C2 Cornell box + 10000 Lambertian spheres"
        make = lambda rc: scenes.cornell_spheres(rc)
    else:
        w, h, spp0 = 400, 400, 50
        name = "C1 Cornell box (18 quads)"
        make = lambda rc: scenes.cornell_box(rc)
    spp_gpu = args.spp or spp0
    spp = spp_gpu * world  # weak scaling: per-GPU work fixed
    t0 = time.time()
    scene = make(RenderConfig(w, h, spp))
    t_build = time.time() - t0
    t0 = time.time()
    ds = DeviceScene(scene, local_rank)
    t_upload = time.time() - t0
    ds.set_partition(rank, world)
    stream = torch.cuda.current_stream(dev)
    ds.set_stream(stream.cuda_stream)
    n_acc = ds.accum_floats()
    acc = torch.zeros(n_acc, dtype=torch.float32, device=dev)
    ds.bind_accum(acc.data_ptr(), n_acc)
    image = torch.empty(h * w * 3, dtype=torch.float32, device=dev) if rank == 0 else None
    ds.kernel_timing(True)
    # a sol_render call handles at most ~4e9 work items; split the sample range if needed
    max_spp_call = max(16, (0xFFFF0000 // max(1, n_acc // 3)) * 16)

    def step():
        ds.clear()
        f = 0
        while f < spp:
            n = min(spp - f, max_spp_call)
            ds.render(f, n, SEED)
            f += n
        if args.rehearse and world > 1:
            torch.cuda.synchronize(dev)
            g = tiles.gather_to_rank0(acc.cpu(), world, rank)
            gathered = g.to(dev) if rank == 0 else None
        else:
            gathered = tiles.gather_to_rank0(acc, world, rank)
        if rank == 0:
            ds.unpermute(gathered.data_ptr(), world, image.data_ptr())

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.rehearse else dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    k_ms, grid = ds.last_kernel_ms()  # duration of the last step's render kernel (HIP events on its stream)

    # ---- counters: exact per-sample algorithmic bytes and rays from a counter-enabled run of the same kernels ----
    ds.bind_accum(0, 0)
    ds.kernel_timing(False)
    c_spp = min(16, spp)
    ds.clear()
    ds.render(0, c_spp, SEED, counted=True)
    st = ds.stats()
    sizes = record_sizes()
    bytes_per_sample = algorithmic_bytes(st, sizes) / st["samples"]
    rays_per_sample = st["rays"] / st["samples"]
    local_samples = st["samples"] // c_spp * spp  # samples this rank renders per step

    out = None
    if rank == 0:
        total_samples = float(w) * h * spp
        ms_per_step = dt / args.steps * 1e3
        value = total_samples / (dt / args.steps) / 1e6
        achieved = bytes_per_sample * local_samples / (k_ms * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                # measured L2<->fabric bytes per sample (profiles/, separate rocprofv3 --pmc passes) x this launch's samples
                if args.workload == "c3" and "_bytes_per_sample" in tj:
                    traffic = int(tj["_bytes_per_sample"] * local_samples)
            except Exception:
                traffic = None
        out = {
            "metric": "Msamples/s", "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": name, "width": w, "height": h, "spp_per_gpu": spp_gpu, "spp_total": spp, "seed": SEED,
                       "max_depth": 50, "sharding": f"8x8 tiles round-robin over {world} rank(s), gather to rank 0"},
            "mrays_per_s": round(value * rays_per_sample, 2),
            "rays_per_sample": round(rays_per_sample, 4),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "kernel": "sol_render_kernel", "kernel_ms": round(k_ms, 3), "grid_blocks": grid,
                         "algorithmic_bytes_per_sample": round(bytes_per_sample, 1),
                         "samples_per_launch": int(local_samples),
                         "counters_per_sample": {k: round(v / st["samples"], 4) for k, v in st.items() if k not in ("samples", "max_stack")}},
            "setup_s": {"scene_and_bvh_build": round(t_build, 2), "upload": round(t_upload, 2)},
        }
        if args.rehearse:
            out["rehearsal"] = "all ranks on cuda:0, gloo gather through host copies: not a measurement"
            # the assembled frame must equal what one rank renders alone (the image is a pure function of scene and seed)
            ds.set_partition(0, 1)
            ds.bind_accum(0, 0)
            ds.clear()
            ds.render(0, min(spp, 32), SEED)
            single = torch.from_numpy(ds.read()).to(dev).reshape(-1)
            ds.set_partition(rank, world)
            out["rehearsal_frame_check"] = "skipped (spp > 32)" if spp > 32 else bool(torch.equal(single, image))
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
    ds.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
