"""rt_scene_build_tree_device (BvhNode::new, bvh.rs:15-66, done on the device): a tree over the same objects returns the
same closest hits whatever its shape, so the device-built tree is held to the oracle's hits and to the hits and image
sums of the host-built tree, bit for bit (exact ties aside). Needs an MI355X."""
import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def _rays(ctx, info, osc, rng, w=160, h=120, n_secondary=20000):
    p = api.params_from_info(info, width=w, height=h, seed=util.RENDER_SEED)
    cam = ctx.generate_rays(info.camera, p, 0)
    first = osc.trace_first_hit(cam, rng.random((len(cam), osc.n_media)) if osc.n_media else None)
    rays = util.ray_suite(cam, first, rng, n_secondary=n_secondary)
    return rays, (rng.random((len(rays), osc.n_media)) if osc.n_media else None)


@pytest.mark.parametrize("name,n", [("spheres_grid", 50000), ("final_scene", None), ("random_scene", None), ("my_scene", None),
                                    ("stress_volume", None), ("two_spheres", None)])
def test_device_built_tree_returns_the_same_hits(gpu_ctx, name, n):
    _, info, blob, desc = util.build_scene(name, n=n, flags=api.FLAG_DEVICE_TREE)
    _, _, blob_sah, _ = util.build_scene(name, n=n)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    scene, scene_sah = gpu_ctx.upload(blob), gpu_ctx.upload(blob_sah)
    rng = np.random.default_rng(41)
    rays, xi = _rays(gpu_ctx, info, osc, rng)
    want = osc.trace_first_hit(rays, xi)
    host = gpu_ctx.trace_first_hit(scene_sah, rays, xi)
    standin = gpu_ctx.trace_first_hit(scene, rays, xi)           # what the compiler left is a tree already
    st = gpu_ctx.build_tree(scene)
    built = gpu_ctx.trace_first_hit(scene, rays, xi)
    print("\n%s: %d leaves, height %d, %d key bits, stack %d; collect %.3f sort %.3f hierarchy %.3f refit %.3f commit %.3f ms" %
          (name, st.leaves, st.height, st.key_bits, st.stack_need, st.ms_collect, st.ms_sort, st.ms_hierarchy, st.ms_refit, st.ms_commit))
    assert st.leaves >= 2 and 1 <= st.height <= 37 and st.stack_need <= 39
    for got in (standin, built):
        rep = util.first_hit_report(got, host)                   # same arithmetic on both sides: nothing but ties may differ
        assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, rep
        rep = util.first_hit_report(got, want)
        assert rep["prim_match"] >= 0.9999 and rep["t_rel_max"] <= 1e-5 and rep["t_bit_equal"] >= 0.98, rep
    # the render walks the tree through the other traversal kernel (LDS stacks and node cache)
    p = api.params_from_info(info, width=96, height=64, seed=util.RENDER_SEED, spp_per_pass=4)
    a = gpu_ctx.render(scene, info.camera, p, 8)
    b = gpu_ctx.render(scene_sah, info.camera, p, 8)
    assert float((util.rel_err(a, b) > 1e-9).mean()) <= 0.002   # paths are identical unless a tie resolves differently
    scene.close()
    scene_sah.close()


@pytest.mark.parametrize("count", [2, 3, 5, 64, 65, 1000])
def test_small_and_ragged_leaf_counts(gpu_ctx, count):
    """Leaf counts around the edges of the build: two leaves (one node), odd counts, one wave and one more, and
    coincident centres (equal Morton codes: the hierarchy falls back to splitting by position)."""
    rng = np.random.default_rng(count)
    b = api.Builder(seed=util.SCENE_SEED, flags=api.FLAG_DEVICE_TREE)
    mat = b.lambertian_color(0.5, 0.5, 0.5)
    centres = rng.uniform(-10.0, 10.0, (count, 3))
    centres[count // 2:] = centres[count // 2]                   # the second half sits on one point, radii differ
    world = b.list([b.sphere(c, 0.2 + 0.01 * i, mat) for i, c in enumerate(centres)])
    blob, desc = b.compile(world), b.describe(world)
    osc = ck.OracleScene(desc)
    scene = gpu_ctx.upload(blob)
    st = gpu_ctx.build_tree(scene)
    assert st.leaves == count and st.height >= 1
    o = rng.uniform(-30.0, 30.0, (5000, 3))
    rays = util.rays_from(o, centres[rng.integers(0, count, 5000)] + rng.normal(0.0, 0.3, (5000, 3)) - o)
    rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays), osc.trace_first_hit(rays))
    assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] >= 0.98, rep
    with pytest.raises(api.RtError) as e:                        # the stand-in is gone: a second build is refused ...
        gpu_ctx.build_tree(scene)
    assert e.value.code == -1
    rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays), osc.trace_first_hit(rays))
    assert rep["prim_match"] == 1.0, rep                         # ... and leaves the tree as it was
    scene.close()


def test_scenes_without_a_tree_for_the_device(gpu_ctx):
    _, info, blob, _ = util.build_scene("final_scene")          # compiled without the flag
    scene = gpu_ctx.upload(blob)
    with pytest.raises(api.RtError) as e:
        gpu_ctx.build_tree(scene)
    assert e.value.code == -1
    scene.close()
    _, info, blob, desc = util.build_scene("cornell_box", flags=api.FLAG_DEVICE_TREE)   # flag ignored: replay trees
    scene = gpu_ctx.upload(blob)
    with pytest.raises(api.RtError) as e:
        gpu_ctx.build_tree(scene)
    assert e.value.code == -1
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    rays, xi = _rays(gpu_ctx, info, osc, np.random.default_rng(3), 80, 60, 2000)
    rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays, xi), osc.trace_first_hit(rays, xi))
    assert rep["prim_match"] >= 0.9999, rep                      # and the scene is as usable as ever
    scene.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_graphs_with_device_built_trees(gpu_ctx, seed):
    """Random nestings of every hittable kind: the leaves handed to the device are then primitives, media, instanced
    leaves and whole sub-trees alike. Worlds that must keep a reference-order tree leave nothing to build (-1)."""
    for flags in (api.FLAG_DEVICE_TREE, api.FLAG_DEVICE_TREE | api.FLAG_ANY_TIME):
        try:
            b, world = util.random_scene_graph(seed, flags)
            blob, desc = b.compile(world), b.describe(world)
        except api.RtError as e:
            assert e.code == -2
            continue
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        scene = gpu_ctx.upload(blob)
        try:
            gpu_ctx.build_tree(scene)
        except api.RtError as e:
            assert e.code == -1
        rng = np.random.default_rng(1000 + seed)
        rays = util.random_rays(rng, 4000, any_time=bool(flags & api.FLAG_ANY_TIME))
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays, xi), osc.trace_first_hit(rays, xi))
        assert rep["prim_match"] >= 0.9995 and rep["t_rel_max"] <= 1e-5 and rep["t_bit_equal"] >= 0.95, (seed, flags, rep)
        scene.close()


def _chain_scene(extra_at_origin):
    """Centres at 1000 * 2^-k along each axis: every Morton bit separates one of them from the rest, so the hierarchy
    over full-resolution codes is a chain some sixty nodes tall."""
    b = api.Builder(seed=util.SCENE_SEED, flags=api.FLAG_DEVICE_TREE)
    mat = b.lambertian_color(0.5, 0.5, 0.5)
    centres = [(0.0, 0.0, 0.0)]
    for k in range(1, 22):
        x = 1000.0 * 2.0 ** -k
        centres += [(x, 0.0, 0.0), (0.0, x, 0.0), (0.0, 0.0, x)]
    centres += [(1e-5 * (i % 7), 1e-5 * (i % 11), 1e-5 * (i % 13)) for i in range(extra_at_origin)]
    world = b.list([b.sphere(c, 1e-4 * (1 + i % 5), mat) for i, c in enumerate(centres)])
    return b.compile(world), b.describe(world), np.array(centres)


def test_a_tree_too_tall_for_the_stack_is_rebuilt_from_coarser_codes(gpu_ctx, monkeypatch):
    monkeypatch.setenv("RT_TREE_NO_ROTATIONS", "1")   # (rotations shorten such a chain; this test is about the fallback)
    blob, desc, centres = _chain_scene(0)
    scene = gpu_ctx.upload(blob)
    st = gpu_ctx.build_tree(scene)
    assert st.key_bits == 30 and st.height <= 38 and st.stack_need <= 39, (st.key_bits, st.height)
    osc = ck.OracleScene(desc)
    rng = np.random.default_rng(9)
    o = rng.uniform(-50.0, 1100.0, (4000, 3))
    rays = util.rays_from(o, centres[rng.integers(0, len(centres), 4000)] + rng.normal(0.0, 1e-4, (4000, 3)) - o)
    rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays), osc.trace_first_hit(rays))
    assert rep["prim_match"] == 1.0 and rep["both_hit"].sum() > 500, rep
    scene.close()


def test_a_tree_too_tall_even_then_leaves_the_standin_in_place(gpu_ctx, monkeypatch):
    monkeypatch.setenv("RT_TREE_NO_ROTATIONS", "1")
    blob, desc, centres = _chain_scene(3000)     # thirty chain levels above a dozen levels of coincident centres
    scene = gpu_ctx.upload(blob)
    with pytest.raises(api.RtError) as e:
        gpu_ctx.build_tree(scene)
    assert e.value.code == -2 and "tall" in str(e.value)
    osc = ck.OracleScene(desc)
    rng = np.random.default_rng(10)
    o = rng.uniform(-50.0, 1100.0, (4000, 3))
    rays = util.rays_from(o, centres[rng.integers(0, len(centres), 4000)] + rng.normal(0.0, 1e-4, (4000, 3)) - o)
    rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays), osc.trace_first_hit(rays))
    assert rep["prim_match"] == 1.0 and rep["both_hit"].sum() > 500, rep
    scene.close()


def test_the_chain_scenes_with_rotations(gpu_ctx):
    """The same degenerate scenes with the build as it runs by default: whatever height it ends at, the hits are the oracle's."""
    for extra in (0, 3000):
        blob, desc, centres = _chain_scene(extra)
        scene = gpu_ctx.upload(blob)
        try:
            st = gpu_ctx.build_tree(scene)
            assert st.stack_need <= 39
        except api.RtError as e:
            assert e.code == -2
        osc = ck.OracleScene(desc)
        rng = np.random.default_rng(11)
        o = rng.uniform(-50.0, 1100.0, (4000, 3))
        rays = util.rays_from(o, centres[rng.integers(0, len(centres), 4000)] + rng.normal(0.0, 1e-4, (4000, 3)) - o)
        rep = util.first_hit_report(gpu_ctx.trace_first_hit(scene, rays), osc.trace_first_hit(rays))
        assert rep["prim_match"] == 1.0 and rep["both_hit"].sum() > 500, rep
        scene.close()
