"""Scenes no one should build and the reference accepts anyway: what it computes for them is a matter of which comparison
lets NaN through, which box has no volume and what a list tested first - and the product has to compute the same.

The named cases are the ones that took a change in the product when util.hostile_scene_graph first found them; the random
scenes follow. The oracle (a restatement of the reference's code, statement by statement) is the judge throughout."""
import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

MISS = 0xFFFFFFFF


def first_hits(b, world, rays, xi_value=0.5):
    """(oracle, product's device code on the CPU) for the same rays, with every medium's uniform = xi_value."""
    blob, desc = b.compile(world), b.describe(world)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    xi = np.full((len(rays), osc.n_media), xi_value) if osc.n_media else None
    o = osc.trace_first_hit(rays, xi)
    h, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
    assert util.same_first_hits(h, o) == 1.0, (h[:4], o[:4])
    return o


def grey(b):
    return b.lambertian_color(0.5, 0.5, 0.5)


AXIS = util.rays_from([(0, 0, -10), (0.2, 0.1, -10)], [(0, 0, 1), (0, 0.01, 1)], time=0.3)


def nan_sphere(b, m):
    """time0 == time1: centre = c0 + (c1 - c0) * (t - t0) / 0 (moving_sphere.rs:37-40), the discriminant is NaN, and
    `discriminant < 0`, `root < t_min || root > t_max` all let it pass (moving_sphere.rs:51-60): every ray hits, at t = NaN."""
    return b.moving_sphere((0, 0, 0), (1, 0, 0), 0.5, 0.5, 1.0, m)


def test_after_a_nan_hit_the_order_of_a_list_decides():
    # `t < NaN` is false: a sphere is not hit any more (sphere.rs:43). A rectangle's `t < t_min || t > t_max` lets NaN
    # through (aarect.rs:26-28) and brings t back to a number, after which the sphere in front of it is found
    for members, want_t in (("rect", 10.0), ("sphere", None), ("rect sphere", 6.0), ("sphere rect", 10.0)):
        b = api.Builder()
        m = grey(b)
        objs = {"rect": lambda: b.xy_rect(-1, 1, -1, 1, 0.0, m), "sphere": lambda: b.sphere((0, 0, -3), 1.0, m)}
        w = b.list([nan_sphere(b, m)] + [objs[k]() for k in members.split()])
        o = first_hits(b, w, AXIS)
        assert (np.isnan(o["t"][0]) if want_t is None else o["t"][0] == want_t), (members, o["t"])


def test_a_nan_hit_reaches_into_trees_that_come_later():
    # with t_max = NaN every box passes (aabb.rs:43-47 keep t_max), and inside the node the order left, right decides
    b = api.Builder()
    m = grey(b)
    tree = b.bvh(b.list([b.xy_rect(-1, 1, -1, 1, 0.0, m), b.sphere((0, 0, 3), 1.0, m), b.sphere((0, 0, -3), 1.0, m), b.box((-1, -1, 5), (1, 1, 6), m)]), 0.0, 1.0)
    first_hits(b, b.list([nan_sphere(b, m), tree]), AXIS)
    first_hits(b, b.list([nan_sphere(b, m), b.translate(tree, (0.1, 0, 0)), b.sphere((0, 0, -6), 1.0, m)]), AXIS)
    # a BvhNode's box made of NaN (the sphere's box at any time is) passes every ray too; one made of infinities none
    o = first_hits(b, b.list([b.bvh(b.list([b.rotate_y(nan_sphere(b, m), 360.0)]), 0.0, 1.0)]), AXIS)
    assert np.isnan(o["t"]).all()
    o = first_hits(b, b.list([b.bvh(b.list([nan_sphere(b, m)]), 5.0, 6.0)]), AXIS)
    assert (o["prim_id"] == MISS).all()
    # flag bit 0 (RotateY's box really rotated): inf * sin(360 deg) - inf: a box of NaN beside a medium's honest box
    b = api.Builder(flags=api.FLAG_BOOK_BBOX)
    m = grey(b)
    nan = b.moving_sphere((-1.6, -5.0, 4.9), (-1.5, -5.5, 2.9), 0.5, 0.5, 0.0, m)
    w = b.list([b.list([b.translate(b.bvh(b.list([b.rotate_y(nan, 360.0)]), 5.0, 6.0), (1e7, 0, 0)),
                        b.constant_medium(b.sphere((-0.6, 0.6, -2.0), 3.0, m), 1e-12, 1, 1, 1)])])
    o = first_hits(b, w, util.rays_from([(-11.4, -1.5, -4.5)], [(2.3, -0.5, -1.1)], time=0.25))
    assert np.isnan(o["t"]).all()


def test_negative_density_scatters_in_front_of_the_boundary_and_depends_on_what_came_before():
    # hit_distance = -1/density * ln(xi) < 0 (constant_medium.rs:56): t = entry + hit_distance / length lies before the entry
    # point, unclipped. Tested after a nearer surface the medium is cut off by t_max instead (constant_medium.rs:40-47).
    def world(b, order):
        m = grey(b)
        fog = b.constant_medium(b.sphere((0, 0, 0), 1.0, m), -0.5, 1, 1, 1)
        wall = b.xy_rect(-5, 5, -5, 5, -1.2, m)                              # in front of the fog's boundary (z = -1)
        return b.list([fog, wall] if order == "fog wall" else [wall, fog])
    b = api.Builder()
    o = first_hits(b, world(b, "fog wall"), AXIS[:1])
    assert o["t"][0] == 9.0 + 2.0 * np.log(0.5)                              # 7.61: the fog, although the wall at 8.8 hides it
    b = api.Builder()
    o = first_hits(b, world(b, "wall fog"), AXIS[:1])
    assert o["t"][0] == 8.8
    # the same under trees, where the product would otherwise walk nearest first
    for order in ("fog wall", "wall fog"):
        b = api.Builder()
        w = world(b, order)
        first_hits(b, b.list([b.bvh(w, 0.0, 1.0), b.sphere((0, 0, 4), 1.0, grey(b))]), AXIS)
    # below zero the closest distance stops everything that compares (all but NaN): t stays negative
    b = api.Builder()
    m = grey(b)
    w = b.list([b.constant_medium(b.sphere((0, 0, -9), 2.0, m), -0.01, 1, 1, 1), b.sphere((0, 0, 0), 1.0, m), b.xy_rect(-1, 1, -1, 1, 3.0, m)])
    o = first_hits(b, w, AXIS)
    assert (o["t"] < 0).all()
    b = api.Builder()
    m = grey(b)
    w = b.list([b.translate(b.bvh(b.list([b.constant_medium(b.sphere((0, 0, -9), 2.0, m), -0.01, 1, 1, 1)]), 0.0, 1.0), (0, 0, 0)), nan_sphere(b, m)])
    o = first_hits(b, w, AXIS)
    assert np.isnan(o["t"]).all()                                            # ... and NaN passes: an entry of a list is asked whatever t_max is


def test_a_node_holding_one_object_asks_it_twice():
    # bvh.rs:33-35: left = right = the object; the second call has t_max = the first one's t (bvh.rs:118-121). For a surface
    # that changes nothing. A tree below, holding a medium of negative density, finds the second time what the first culled.
    b = api.Builder()
    m = grey(b)
    inner = b.bvh(b.list([b.rotate_y(b.sphere((0, 0, -3), 1.0, m), 0.0), b.constant_medium(b.translate(b.sphere((0, 0, 0), 1.0, m), (0, 0, 1)), -0.3, 1, 1, 1)]), 5.0, 6.0)
    w = b.list([nan_sphere(b, m), b.bvh(b.list([inner]), 0.0, 1.0)])
    first_hits(b, w, AXIS)
    first_hits(b, b.list([b.bvh(b.list([inner]), 0.0, 1.0)]), AXIS, 0.9)


def test_boxes_without_volume_and_inside_out():
    # AABB::hit never passes a box without volume (t_max <= t_min, aabb.rs:47), so under a BvhNode a flat Box is gone, while
    # a list still hits its two coinciding faces; p0 > p1 in one axis leaves the two faces across that axis hittable
    b = api.Builder()
    m = grey(b)
    flat = b.box((-1, -1, 0), (1, 1, 0), m)
    o = first_hits(b, b.list([flat]), AXIS)
    assert o["t"].tolist() == [10.0, 10.0]
    o = first_hits(b, b.list([b.bvh(b.list([flat]), 0.0, 1.0), b.sphere((0, 0, 5), 1.0, m)]), AXIS)
    assert o["t"][0] == 14.0
    b = api.Builder()
    m = grey(b)
    inverted = b.box((-1, -1, 1), (1, 1, -1), m)                            # z0 > z1: the xy faces at z = 1 and z = -1 remain
    o = first_hits(b, b.list([inverted, b.sphere((0, 0, 5), 1.0, m)]), AXIS)
    assert o["t"][0] == 9.0
    side = util.rays_from([(-10, 0, 0)], [(1, 0, 0)])                       # the yz faces have an empty z range
    o = first_hits(b, b.list([inverted]), side)
    assert o["prim_id"][0] == MISS
    o = first_hits(b, b.list([b.bvh(b.list([inverted, b.sphere((0, 3, 0), 1.0, m)]), 0.0, 1.0)]), AXIS)
    # rectangles without extent, with and without a tree above
    b = api.Builder()
    m = grey(b)
    line = b.xy_rect(0.0, 0.0, -1, 1, 0.0, m)
    first_hits(b, b.list([line, b.bvh(b.list([line, b.xy_rect(1, -1, -1, 1, 1.0, m)]), 0.0, 1.0)]), AXIS)


FLAG_SETS = (0, api.FLAG_ANY_TIME, api.FLAG_BOOK_BBOX, api.FLAG_REFERENCE_TREE, api.FLAG_DEVICE_TREE)


def compile_or_skip(seed, flags):
    try:
        b, world = util.hostile_scene_graph(seed, flags)
        return b, b.compile(world), b.describe(world)
    except api.RtError as e:
        assert e.code in (-2, -5), e                                         # declined (too deep a chain), or the reference's own panic
        return None


@pytest.mark.parametrize("seed", range(60))
def test_hostile_scene_graphs(seed):
    ran = 0
    for flags in FLAG_SETS:
        made = compile_or_skip(seed, flags)
        if made is None:
            continue
        _, blob, desc = made
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        rng = np.random.default_rng(seed)
        rays = util.random_rays(rng, 2000, any_time=bool(flags & api.FLAG_ANY_TIME))
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        try:
            got, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
        except RuntimeError:                                                 # ray times outside what the blob was compiled for
            assert not flags & api.FLAG_ANY_TIME
            continue
        assert util.same_first_hits(got, osc.trace_first_hit(rays, xi)) == 1.0, (seed, flags)
        ran += 1
    if ran == 0:
        pytest.skip("the reference panics on this one")


@pytest.mark.parametrize("seed", range(24))
def test_hostile_scene_graphs_render(seed):
    made = compile_or_skip(seed, 0)
    if made is None:
        pytest.skip("the reference panics on this one")
    _, blob, desc = made
    cam = api.make_camera((0, 2, -26), (0, 0, 0), (0, 1, 0), 40.0, 4 / 3, 0.1, 20.0, 0.25, 0.75)
    p = api.make_params(16, 12, max_depth=8, background=(0.7, 0.8, 1.0), seed=77 + seed)
    try:
        got, segments = ck.hostsim_render(blob, cam, p, 2)
    except RuntimeError:
        pytest.skip("the shutter is outside what the blob was compiled for")
    want, st = ck.OracleScene(desc, share_bvh_axes=True).render(cam, p, 2, threads=2)
    assert segments == st.ray_segments                                       # the same paths, bounce for bounce
    with np.errstate(invalid="ignore"):
        # (the product multiplies a path's attenuations from the camera outwards, the reference from the far end back)
        close = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= 1e-12 * np.abs(want))
    assert close.all()


MATERIAL_CAMERAS = ((0, 0.0), (1, 0.5))                                    # (background mode, aperture)


def material_render_setup(seed, mode, aperture, width=16, height=12):
    cam = api.make_camera((0, 2, -20), (0, 0, 0), (0, 1, 0), 45.0, 4 / 3, aperture, 18.0, 0.0, 1.0)
    return cam, api.make_params(width, height, max_depth=12, background=(0.7, 0.8, 1.0), background_mode=mode, seed=77 + seed)


@pytest.mark.parametrize("seed", range(40))
def test_hostile_materials_render(seed):
    """Shading under parameters outside every range (see util.hostile_materials_scene): the same paths, the same sums."""
    b, world = util.hostile_materials_scene(seed)
    blob, desc = b.compile(world), b.describe(world)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    for mode, aperture in MATERIAL_CAMERAS:
        cam, p = material_render_setup(seed, mode, aperture)
        got, segments = ck.hostsim_render(blob, cam, p, 3)
        want, st = osc.render(cam, p, 3, threads=2)
        assert segments == st.ray_segments
        with np.errstate(invalid="ignore"):
            close = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= 1e-11 * np.abs(want))
        assert close.all(), (seed, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 40, 4))
def test_hostile_materials_render_on_the_gpu(gpu_ctx, seed):
    b, world = util.hostile_materials_scene(seed)
    blob, desc = b.compile(world), b.describe(world)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    scene = gpu_ctx.upload(blob)
    for mode, aperture in MATERIAL_CAMERAS:
        cam, p = material_render_setup(seed, mode, aperture, 48, 36)
        st = api.rt_stats()
        got = gpu_ctx.render(scene, cam, p, 4, stats=st)
        want, ost = osc.render(cam, p, 4)
        with np.errstate(invalid="ignore", divide="ignore"):
            close = (got == want) | (np.isnan(got) & np.isnan(want)) | (np.abs(got - want) <= 1e-9 * np.abs(want))
        # (sin, cos, log and pow of the device library differ from the host's in the last bit now and then; a path that meets
        # such a difference at a checker's edge or a refraction's threshold goes elsewhere)
        assert float((~close).mean()) <= 0.01, (seed, mode, float((~close).mean()))
        assert abs(st.ray_segments / ost.ray_segments - 1.0) <= 2e-3
    scene.close()


def sane_scene(seed, flags):
    try:
        b, world = util.random_scene_graph(seed, flags)
        return b.compile(world), b.describe(world)
    except api.RtError as e:
        assert e.code == -2
        return None


@pytest.mark.parametrize("seed", range(40))
def test_hostile_rays(seed):
    """Well-formed scenes, ill-formed rays (see util.hostile_rays)."""
    for flags in (0, api.FLAG_ANY_TIME):
        made = sane_scene(seed, flags)
        if made is None:
            continue
        blob, desc = made
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        rng = np.random.default_rng(300 + seed)
        rays = util.hostile_rays(rng, 3000)
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        got, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
        assert util.same_first_hits(got, osc.trace_first_hit(rays, xi)) == 1.0, (seed, flags)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 40, 4))
def test_hostile_rays_on_the_gpu(gpu_ctx, seed):
    made = sane_scene(seed, api.FLAG_ANY_TIME)
    if made is None:
        pytest.skip("deeper instance chains than the product supports")
    blob, desc = made
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    rng = np.random.default_rng(300 + seed)
    rays = util.hostile_rays(rng, 30000)
    xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
    scene = gpu_ctx.upload(blob)
    lane = gpu_ctx.trace_first_hit(scene, rays, xi)
    want = osc.trace_first_hit(rays, xi)
    assert util.same_first_hits(lane, want, t_rel=1e-12) >= 0.9995, seed
    assert util.same_first_hits(lane, want) >= 0.95, seed
    for variant in (0, 3):
        assert util.same_first_hits(gpu_ctx.trace_first_hit(scene, rays, xi, wavefront=variant), lane) == 1.0, (seed, variant)
    scene.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(0, 60, 3))
def test_hostile_scene_graphs_on_the_gpu(gpu_ctx, seed):
    ran = 0
    for flags in (0, api.FLAG_ANY_TIME, api.FLAG_DEVICE_TREE):
        made = compile_or_skip(seed, flags)
        if made is None:
            continue
        _, blob, desc = made
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        rng = np.random.default_rng(seed)
        rays = util.random_rays(rng, 20000, any_time=bool(flags & api.FLAG_ANY_TIME))
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        scene = gpu_ctx.upload(blob)
        if flags & api.FLAG_DEVICE_TREE:
            try:
                gpu_ctx.build_tree(scene)
            except api.RtError as e:                                         # nothing was left to the device: the world keeps an order
                assert e.code == -1
        try:
            lane = gpu_ctx.trace_first_hit(scene, rays, xi)
        except api.RtError as e:
            assert e.code == -2 and not flags & api.FLAG_ANY_TIME
            scene.close()
            continue
        want = osc.trace_first_hit(rays, xi)
        # (log() of the two maths libraries differs in the last bit for a few per cent of the medium hits)
        assert util.same_first_hits(lane, want, t_rel=1e-12) >= 0.9995, (seed, flags)
        assert util.same_first_hits(lane, want) >= 0.95, (seed, flags)
        for variant in (0, 3):
            wave = gpu_ctx.trace_first_hit(scene, rays, xi, wavefront=variant)
            assert util.same_first_hits(wave, lane) == 1.0, (seed, flags, variant)
        scene.close()
        ran += 1
    if ran == 0:
        pytest.skip("the reference panics on this one")
