"""Creation and scheduling options through the C ABI (sol_scene_create_ex, sol_scene_set_option; no reference analogue: the
reference has one BVH builder and no scheduler to tune). None of them may change a frame."""
import numpy as np
import pytest

import parity_util as pu
from solstrale_amd import DeviceError, DeviceScene, RenderConfig, _abi, scenes

pytestmark = pytest.mark.gpu


def _frame(ds, spp=16):
    ds.clear()
    ds.render(0, spp, pu.SEED)
    return ds.read()


@pytest.mark.parametrize("make", [scenes.cornell_spheres, scenes.sponza_like, scenes.create_test_scene], ids=["c2", "c3", "test"])
def test_every_world_tree_choice_renders_the_same_frame(make):
    sc = make(RenderConfig(240, 136, 16))
    frames = {}
    for tree in (_abi.TREE_AUTO, _abi.TREE_REF, _abi.TREE_SAH8, _abi.TREE_SAH16, _abi.TREE_SAH64, _abi.TREE_DEVICE, _abi.TREE_HOST_PROBE):
        with DeviceScene(sc, world_tree=tree) as ds:
            frames[tree] = _frame(ds)
            bt = ds.build_times()
            assert (bt["device_tree"] > 0) == (tree in (_abi.TREE_AUTO, _abi.TREE_DEVICE)), (tree, bt)
    for tree, img in frames.items():
        assert (img == frames[_abi.TREE_AUTO]).all(), tree


def test_scheduler_options_do_not_change_the_frame():
    sc = scenes.sponza_like(RenderConfig(240, 136, 16))
    with DeviceScene(sc) as ds:
        want = _frame(ds)
        for opt, values in ((_abi.OPT_SWITCH_BELOW, (0, 8, 40, 64)), (_abi.OPT_MAX_BLOCKS_PER_CU, (1, 2, 0)), (_abi.OPT_WORK_ORDER, (0, 1)),
                            (_abi.OPT_FINE_TAIL, (0, 4, 64, -1)), (_abi.OPT_KERNEL, (1, 0))):
            for v in values:
                ds.set_option(opt, v)
                assert (_frame(ds) == want).all(), (opt, v)
        for opt, v in ((_abi.OPT_SWITCH_BELOW, 65), (_abi.OPT_SWITCH_BELOW, -1), (_abi.OPT_KERNEL, 4), (_abi.OPT_KERNEL, 2), (_abi.OPT_FINE_TAIL, -2), (_abi.OPT_FINE_TAIL, 65), (99, 0)):
            with pytest.raises(DeviceError) as e:
                ds.set_option(opt, v)
            assert e.value.code == _abi.SOL_EINVAL
        assert (_frame(ds) == want).all()


@pytest.mark.parametrize("make", [scenes.sponza_like, scenes.create_test_scene], ids=["c3", "test_scene_with_medium"])
@pytest.mark.parametrize("size", [(240, 136), (250, 131)], ids=["whole_blocks", "edge_blocks"])
def test_fine_tail_keeps_every_frame(size, make):
    """SOL_OPT_FINE_TAIL: the last items of a launch are handed out one sample at a time and their colours added up in sample
    order afterwards - the same sums, bit for bit, for every tail length, sample count (whole and ragged last chunks, one
    chunk and many), sample offset, partition and accumulation over calls."""
    sc = make(RenderConfig(size[0], size[1], 16))
    with DeviceScene(sc) as ds:
        def frames():
            out = []
            for spp in (5, 16, 37, 64):
                out.append(_frame(ds, spp))
            ds.clear()
            ds.render(0, 20, pu.SEED)
            ds.render(20, 17, pu.SEED)  # (accumulates on the first call's sums; chunks are counted from each call's first sample)
            out.append(ds.read())
            ds.set_partition(3, 8)
            ds.clear()
            ds.render(0, 37, pu.SEED)
            out.append(ds.read())  # (this rank's tiles; the others stay zero)
            ds.set_partition(0, 1)
            return out
        ds.set_option(_abi.OPT_FINE_TAIL, 0)
        want = frames()
        for tail in (1, 4, 16, 64):
            ds.set_option(_abi.OPT_FINE_TAIL, tail)
            for k, (a, b) in enumerate(zip(frames(), want)):
                assert a.shape == b.shape and (a == b).all(), (tail, k)


def test_creation_without_the_work_order_probe():
    """`no_work_order_probe` (an EverySample preview that must start at once): no counted 4-spp probe of the frame at creation."""
    sc = scenes.statue_like(RenderConfig(480, 270, 16), n_triangles=60000)
    # (what the probe leaves behind: per-block costs, which the balanced partition needs - without them the option falls back to b % world.
    # Its time, a few ms of the creation, is not asserted: a first launch in a fresh process costs more than the probe.)
    with DeviceScene(sc) as ds:
        want = _frame(ds)
        ds.set_option(_abi.OPT_BALANCED_PARTITION, 1)
        ds.set_partition(0, 2)
        assert ds.info()["partition_table"] == 1
    with DeviceScene(sc, no_work_order_probe=True) as ds:
        assert (_frame(ds) == want).all()
        ds.set_option(_abi.OPT_BALANCED_PARTITION, 1)
        ds.set_partition(0, 2)
        assert ds.info()["partition_table"] == 0


def test_bad_creation_options_are_rejected():
    sc = scenes.cornell_box(RenderConfig(32, 32, 1))
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc, world_tree=17)
    assert e.value.code == _abi.SOL_EINVAL


def test_a_change_of_rank_clears_the_sums_of_the_old_layout():
    """sol_scene_set_partition: sums already in the accumulators lie in the OLD block -> slot layout; with the modulo partition the
    partition checksum depends on (world, blocks) only, so set_partition(3, 8) -> render -> set_partition(5, 8) -> render -> read used to
    hand back rank 3's sums in rank 5's pixels (round-4 advisor finding). The guard now compares rank and world as well."""
    sc = scenes.cornell_box(RenderConfig(96, 64, 8))
    with DeviceScene(sc) as ds:
        ds.set_partition(5, 8)
        ds.render(0, 8, pu.SEED)
        want = ds.read()
        ds.set_partition(3, 8)
        ds.clear()
        ds.render(0, 8, pu.SEED)
        ds.set_partition(5, 8)   # no clear by the caller
        ds.render(0, 8, pu.SEED)
        got = ds.read()
    assert want.any() and (got == want).all()


def test_rebinding_rules_of_a_caller_bound_accumulator():
    """sol_scene_set_partition refuses to drop a caller-bound accumulator silently (it would keep gathering from a buffer that
    is no longer written)."""
    import torch
    sc = scenes.cornell_box(RenderConfig(64, 64, 4))
    with DeviceScene(sc) as ds:
        n = ds.accum_floats()
        buf = torch.zeros(n, dtype=torch.float32, device="cuda")
        ds.bind_accum(buf.data_ptr(), n)
        ds.set_partition(0, 1)  # same size: allowed, the binding stays
        with pytest.raises(DeviceError) as e:
            ds.set_partition(1, 2)
        assert e.value.code == _abi.SOL_EINVAL and "unbind" in e.value.msg
        ds.bind_accum(0, 0)
        ds.set_partition(1, 2)
        ds.render(0, 4, pu.SEED)
        assert np.isfinite(ds.read()).all()


def _scene_with_an_unbounded_triangle():
    """One triangle has a vertex at 1e39: finite in f64, infinite as fp32 - its device box cannot be collected by the GPU builder
    (nor by the SAH rebuilds); the collapsed reference tree holds it as a box no finite ray parameter reaches."""
    import orc  # noqa: F401  (the checker of the frames below)
    from solstrale_amd import CameraConfig, SceneBuilder
    b = SceneBuilder()
    white = b.Lambertian(b.SolidColor(.7, .7, .7))
    world = [b.Quad((-3., 0., -3.), (6., 0, 0), (0, 0, 6.), white), b.Sphere((0., 1., 0.), 1., white),
             b.Triangle((1e39, 0., 0.), (1., 0., 0.), (0., 1., 0.), white), b.Triangle((-1., 0., 1.), (1., 0., 1.), (0., 2., 1.), white),
             b.Sphere((0., 8., 0.), 2., b.DiffuseLight(5, 5, 5))]
    return b.finish(b.Bvh(world), CameraConfig(40., 0., (0., 2., 7.), (0., 1., 0.), (0, 1, 0)), (.2, .3, .4), RenderConfig(64, 48, 4))


def test_auto_tree_falls_back_when_the_device_build_cannot_make_a_tree():
    """SOL_TREE_AUTO is the GPU build; a scene it cannot handle (here: a primitive whose fp32 box is not finite) must still be
    accepted by the default path - as the host path accepts it - through the host candidates, and SolSceneInfo says so. An
    explicit SOL_TREE_DEVICE keeps the error."""
    import orc
    sc = _scene_with_an_unbounded_triangle()
    with DeviceScene(sc) as ds:  # defaults
        info = ds.info()
        assert info["tree_fallback"] and info["tree_name"] == "ref" and "device build failed" in info["tree_note"], info
        img = _frame(ds, 4)
    ref, _ = orc.render(sc, 0, 4, pu.SEED, real=orc.ORC_F32)
    assert pu.compare(img, ref, 4)["bad_pixels"] == 0
    with DeviceScene(sc, world_tree=_abi.TREE_REF) as ds:
        assert not ds.info()["tree_fallback"] and (_frame(ds, 4) == img).all()
    with pytest.raises(DeviceError) as e:
        DeviceScene(sc, world_tree=_abi.TREE_DEVICE)
    assert "collected" in str(e.value)
    with DeviceScene(scenes.cornell_box(RenderConfig(32, 32, 1))) as ds:  # an ordinary scene: no fallback
        info = ds.info()
        assert not info["tree_fallback"] and info["tree_name"].startswith("device") and info["tree_note"] == ""


def test_stack_use_stays_within_the_bound_the_kernel_choice_relies_on():
    """The render kernel without a spill path is chosen from the host's bound on the stack use (SolSceneInfo.stack_bound <=
    lds_stack): in that variant an overflow would silently overwrite other lanes' stacks. A counted render measures the deepest
    use; it must stay within the bound - on the BASELINE scenes, on a scene with a constant medium, and on a deep chain."""
    import test_gpu_parity as tgp
    from solstrale_amd import CameraConfig, SceneBuilder
    cases = [scenes.cornell_box(RenderConfig(96, 96, 4)), scenes.cornell_spheres(RenderConfig(160, 90, 4)), scenes.sponza_like(RenderConfig(160, 90, 4)),
             scenes.statue_like(RenderConfig(160, 90, 4)), scenes.create_test_scene(RenderConfig(160, 80, 4))]
    b = SceneBuilder()
    chain = tgp._sphere_chain(b)
    light = b.Sphere((0., 1e4, 0.), 3e3, b.DiffuseLight(3, 3, 3))
    cases.append(b.finish(b.Bvh([chain, light]), CameraConfig(12., 0., (-30., 0.4, 0.3), (50., 0.3, 0.), (0, 1, 0)), (.1, .1, .1), RenderConfig(64, 64, 4)))
    for sc in cases:
        for tree in (_abi.TREE_AUTO, _abi.TREE_REF):
            with DeviceScene(sc, world_tree=tree) as ds:
                info = ds.info()
                ds.render(0, 4, pu.SEED, counted=True)
                st = ds.stats()
            assert st["max_stack"] <= info["stack_bound"], (tree, st["max_stack"], info)  # (0: a one-node tree never pushes)
            assert info["stack_bound"] <= info["lds_stack"] + info["spill_stack"]
