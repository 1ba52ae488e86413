"""bench.py end to end on the GPU box: the default single-GPU contract line; two- and three-rank rehearsals of the N>1 path started
by bench.py itself (no launcher; all ranks on cuda:0, going through sol_comm_init / sol_gather over the test-only transport stub
tests/stub_rccl - RCCL refuses two ranks on one device) whose assembled frame must equal the single-rank frame bit for bit; and the
real RCCL communicator behind the C ABI exercised with the one rank a one-GPU box can hold. (The ranks' rendezvous under
torch.distributed.run - the launcher's TCPStore - is covered on the CPU: tests/test_distributed_cpu.py.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline"}


def _line(out):
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, "bench.py", "--workload", "c1", "--spp", "32", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1",
                        "--pmc-spp", "32"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert REQUIRED <= set(d) and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["value"] > 0 and d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["dtype"] == "f32"
    rf = d["roofline"]
    # (`bound` names the contract's figure for what it is - algorithmic bytes against the HBM peak -, `binding` what limits the kernel)
    assert rf["bound"] == "algorithmic_hbm" and rf["binding"] == "valu_issue" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    # traffic is either measured in this run (rocprofv3 --pmc passes) or null with the reason
    assert (rf["traffic"] is None and rf["traffic_source"].startswith("not measured")) or rf["traffic"] > 0
    if rf["traffic"]:
        rv = d["roofline_valu"]
        assert rv["bound"] == "valu_issue" and 0 < rv["frac"] < 1 and 0 < rv["lane_utilisation"] <= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "spp" in cb["sample"]


@pytest.mark.parametrize("ranks,scaling,spp_total", [(2, "strong", 16), (2, "weak", 32), (3, "strong", 16)])
def test_rehearsal_without_a_launcher_goes_through_sol_gather(ranks, scaling, spp_total):
    """`python bench.py --gpus N --rehearse` starts its own ranks; the data path is the real one (sol_comm_init, sol_render, sol_gather
    with world > 1) over the stub transport; the frame rank 0 assembles equals the single-rank frame."""
    cmd = [sys.executable, "bench.py", "--gpus", str(ranks), "--workload", "c1", "--spp", "16", "--steps", "1", "--warmup", "1", "--rehearse",
           "--scaling", scaling, "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == ranks and d["scaling"] == scaling and d["config"]["spp_total"] == spp_total
    assert d["rehearsal_frame_check"] is True and d["rccl_ranks"] == ranks and "sol_gather" in d["config"]["sharding"]


def test_both_figures_in_a_multi_rank_line():
    """A scene with background blocks (the C3 atrium's sky) through two ranks: the N > 1 line carries the figure with every sample traced
    as well, from as many steps as `value` (round-4 review: a scaling curve must not quote only the flattering figure)."""
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--workload", "c3", "--spp", "16", "--steps", "2", "--warmup", "1", "--rehearse", "--no-cpu-baseline"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    bb = d["background_blocks"]
    assert d["n_gpus"] == 2 and bb["blocks"] > 0 and bb["value_with_every_sample_traced"] > 0 and bb["value_with_every_sample_traced_steps"] == 2
    assert len(d["per_rank_setup_s"]["sol_scene_create"]) == 2 and all(t > 0 for t in d["per_rank_setup_s"]["sol_scene_create"])
    assert d["rehearsal_frame_check"] is True


def test_a_supplied_obj_replaces_the_stand_in():
    """`--obj PATH`: a user-supplied OBJ (+MTL, textures) goes through the host OBJ+MTL loader (src/loader/obj.rs:38-136 restated in
    host/solstrale_obj.cpp) and is rendered in place of the procedural stand-in; the line says which file it was (SURVEY.md 8d:
    "if a real sponza.obj is supplied at run time, use it and say so")."""
    obj = os.path.join(ROOT, "tests", "golden", "resources", "spider", "spider.obj")
    r = subprocess.run([sys.executable, "bench.py", "--obj", obj, "--workload", "c1", "--spp", "16", "--steps", "1", "--warmup", "1",
                        "--no-cpu-baseline", "--no-pmc"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert "spider.obj" in d["config"]["workload"] and d["data"] == "user-supplied OBJ" and d["value"] > 0
    assert d["roofline"]["counters_per_sample"]["triangle_tests"] > 0 and d["rays_per_sample"] >= 1.0


def test_a_failing_rank_fails_the_run():
    """A rank that dies must not leave `bench.py --gpus N` hanging or exiting 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--workload", "c1", "--spp", "16", "--steps", "1", "--rehearse",
                        "--no-cpu-baseline", "--obj", "/nonexistent/file.obj"], cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rccl_communicator_behind_the_abi_single_rank():
    """sol_comm_unique_id / sol_comm_init / sol_gather / sol_comm_self_check with world = 1 (all a one-GPU box can hold): RCCL
    is loaded, a communicator is created, the accumulator makes a round trip through ncclSend / ncclRecv unchanged, and the
    gathered image equals sol_read's."""
    from solstrale_amd import DeviceScene, RenderConfig, comm_unique_id, scenes
    sc = scenes.cornell_box(RenderConfig(96, 64, 16))
    with DeviceScene(sc) as ds:
        ds.comm_init(0, 1, comm_unique_id())
        ds.render(0, 16, 7)
        ds.comm_self_check()
        ds.gather(0)
        img = ds.read_image()
        assert (img == ds.read()).all() and np.isfinite(img).all() and img.mean() > 0
        ds.comm_destroy()
