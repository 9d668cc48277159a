"""The product's per-lane device code (traversal, hit reconstruction, materials, textures, camera), compiled for the
CPU by the host simulator, against the oracle on every scene: identical input rays -> identical first hits,
identical seeds -> identical images. The GPU tests repeat the same comparisons through the C ABI."""
import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

MISS = 0xFFFFFFFF


@pytest.mark.parametrize("name", list(util.SCENES))
def test_first_hit_matches_oracle(name):
    _, info, blob, desc = util.build_scene(name)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    p = api.params_from_info(info, width=40, height=30, seed=util.RENDER_SEED)
    cam_o = ck.oracle_generate_rays(info.camera, p, 0)
    cam_h = ck.hostsim_generate_rays(info.camera, p, 0)
    assert cam_o.tobytes() == cam_h.tobytes()                      # camera.rs:56-64 + main.rs:162-163, bit for bit
    rng = np.random.default_rng(5)
    first = osc.trace_first_hit(cam_o, rng.random((len(cam_o), osc.n_media)) if osc.n_media else None)
    rays = util.ray_suite(cam_o, first, rng)
    xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
    want = osc.trace_first_hit(rays, xi)
    got, stack_high = ck.hostsim_trace_first_hit(blob, rays, xi)
    rep = util.first_hit_report(got, want)
    import struct
    assert stack_high <= struct.unpack_from("<I", blob, 32)[0]     # the stack the header asks for (no spare entry) is enough
    assert rep["prim_match"] == 1.0, rep
    assert rep["t_bit_equal"] == 1.0, rep                          # same libm on both sides here: exact
    assert rep["ties"] < 0.02
    same = got["prim_id"] == want["prim_id"]
    for f in ("px", "py", "pz", "nx", "ny", "nz"):
        assert np.array_equal(got[f][same].view(np.uint64), want[f][same].view(np.uint64)), f     # bit compare: NaN == NaN
    assert np.array_equal(got["flags"][same], want["flags"][same]) and np.array_equal(got["material_id"][same], want["material_id"][same])
    # u, v: equal wherever the reference writes them; MovingSphere and ConstantMedium leave stale values behind
    # (moving_sphere.rs:63-68, constant_medium.rs:59-65) where the product reports 0
    same_uv = (got["u"].view(np.uint64) == want["u"].view(np.uint64)) & (got["v"].view(np.uint64) == want["v"].view(np.uint64))
    differs = ~same_uv & (want["prim_id"] != MISS) & same
    assert np.all(np.isin(want["prim_id"][differs], util.ids_without_uv(desc)))
    assert np.all((got["u"][differs] == 0.0) & (got["v"][differs] == 0.0))
    assert (want["prim_id"] != MISS).mean() > 0.3
    # An oracle that draws its own BVH split axes (a different tree) must still find the same closest hit, except
    # (a) exact ties - coincident faces of neighbouring boxes reached at the same t, where the reference's own winner
    # depends on its random tree - and (b) a RotateY instance under a BVH, whose unrotated box (hittable.rs:115)
    # makes what gets clipped depend on the tree (cornell_box).
    if name != "cornell_box":
        other = ck.OracleScene(desc, bvh_seed=77, share_bvh_axes=False).trace_first_hit(rays, xi)
        differs = other["prim_id"] != got["prim_id"]
        assert np.array_equal(other["t"][differs].view(np.uint64), got["t"][differs].view(np.uint64))
        assert differs.mean() < 0.03


@pytest.mark.parametrize("name", list(util.SCENES))
def test_render_matches_oracle(name):
    _, info, blob, desc = util.build_scene(name)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    p = api.params_from_info(info, width=24, height=16, seed=util.RENDER_SEED)
    want, st = osc.render(info.camera, p, 6, threads=4)
    got, segments = ck.hostsim_render(blob, info.camera, p, 6)
    assert segments == st.ray_segments                             # the same paths, segment for segment
    # the product multiplies attenuations front to back, the reference's recursion back to front (main.rs:70):
    # same factors, different association -> a few ulp
    assert util.rel_err(got, want).max() < 1e-13
    assert want.mean() > 0


def test_book_compat_switches_match_oracle():
    _, info, blob, desc = util.build_scene("cornell_box")
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    for compat in (0, 1, 2, 3):
        p = api.params_from_info(info, width=20, height=20, seed=9, compat=compat)
        cam = api.make_camera((278, 278, -800), (278, 278, 0), (0, 1, 0), 40.0, 1.0, 8.0, 800.0)  # wide aperture: lens matters
        want, _ = osc.render(cam, p, 3, threads=2)
        got, _ = ck.hostsim_render(blob, cam, p, 3)
        assert util.rel_err(got, want).max() < 1e-13
    a, _ = ck.hostsim_render(blob, cam, api.params_from_info(info, width=20, height=20, seed=9, compat=0), 3)
    c, _ = ck.hostsim_render(blob, cam, api.params_from_info(info, width=20, height=20, seed=9, compat=2), 3)
    assert not np.array_equal(a, c)                                # the XZ-vs-XY lens disk changes the image


def test_sample_order_and_offsets_compose():
    # summing samples [0,6) at once == [0,2) then [2,6) into the same buffer, on both implementations
    _, info, blob, desc = util.build_scene("simple_light")
    p = api.params_from_info(info, width=12, height=8, seed=4)
    whole, _ = ck.hostsim_render(blob, info.camera, p, 6)
    part, _ = ck.hostsim_render(blob, info.camera, p, 2)
    part, _ = ck.hostsim_render(blob, info.camera, p, 4, sample_offset=2, rgb_sum=part)
    assert np.array_equal(whole, part)
    osc = ck.OracleScene(desc)
    o_whole, _ = osc.render(info.camera, p, 6, threads=1)
    o_part, _ = osc.render(info.camera, p, 2, threads=3)
    o_part, _ = osc.render(info.camera, p, 4, sample_offset=2, threads=2, rgb_sum=o_part)
    assert np.array_equal(o_whole, o_part)                         # and the oracle does not depend on its thread count


@pytest.mark.parametrize("name", ["final_scene", "random_scene", "cornell_smoke", "my_scene"])
def test_reference_topology_trees_find_the_same_hits(name):
    """Builder flag bit 1 keeps every BvhNode as BvhNode::new builds it (bvh.rs:15-66) instead of dissolving it into a
    surface-area-heuristic tree: the closest hits must not depend on which tree is walked (exact ties aside)."""
    _, info, blob, desc = util.build_scene(name)
    _, _, blob_ref, desc_ref = util.build_scene(name, flags=2)
    assert blob != blob_ref
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    p = api.params_from_info(info, width=40, height=30, seed=util.RENDER_SEED)
    cam = ck.oracle_generate_rays(info.camera, p, 0)
    rng = np.random.default_rng(11)
    first = osc.trace_first_hit(cam, rng.random((len(cam), osc.n_media)) if osc.n_media else None)
    rays = util.ray_suite(cam, first, rng)
    xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
    a, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
    b, _ = ck.hostsim_trace_first_hit(blob_ref, rays, xi)
    rep = util.first_hit_report(a, b)
    assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0 and rep["ties"] < 0.02, rep
    want = osc.trace_first_hit(rays, xi)
    rep = util.first_hit_report(b, want)
    assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, rep


def test_moving_sphere_outside_the_interval_its_tree_bound_covers():
    """A moving sphere at world level sits in the tree under the bound of its own [time0, time1]; rays of other times
    (center() extrapolates, moving_sphere.rs:37-40) must still find it, through the lock-step list."""
    import copy
    _, info, blob, desc = util.build_scene("final_scene")
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    # center(t) = (400, 400, 200) + (30, 0, 0) * t (main.rs:603-612): aim at where it is at t = 3, 5, -2 and at t = 0.5
    times = np.array([3.0, 5.0, -2.0, 0.5])
    origin = np.array([278.0, 278.0, -800.0])
    targets = np.stack([400.0 + 30.0 * times, np.full(4, 400.0), np.full(4, 200.0)], axis=1)
    rays = util.rays_from(np.broadcast_to(origin, (4, 3)), targets - origin, time=times)
    xi = np.full((4, osc.n_media), 1e-12)             # free flight longer than the scene: no scatter on the way
    want = osc.trace_first_hit(rays, xi)
    got, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
    moving = [int(l.split()[1]) for l in desc.split(b"\n") if l.startswith(b"moving_sphere ")][0]
    assert np.all(want["prim_id"] == moving) and np.array_equal(got["prim_id"], want["prim_id"])
    assert np.array_equal(got["t"].view(np.uint64), want["t"].view(np.uint64))
    # a render whose camera shutter leaves [0, 1]: every sample agrees with the oracle's
    cam = copy.copy(info.camera)
    cam.time0, cam.time1 = -1.0, 4.0
    p = api.params_from_info(info, width=24, height=18, seed=util.RENDER_SEED)
    a, _ = ck.hostsim_render(blob, cam, p, 2)
    b, _ = osc.render(cam, p, 2, threads=4)
    assert float((util.rel_err(a, b) > 1e-9).mean()) <= 0.002


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_graphs(seed):
    """Random nestings of every hittable kind: lists inside BvhNodes inside instances, media over rotated boxes, moving
    spheres under Translate, several levels deep. Builder flags: 0 = the compiler's own trees, 2 = the reference's
    topology, 4 = valid for rays of any time (then some rays carry a time outside every [time0, time1])."""
    for flags in (0, 2, 4):
        try:
            b, world = util.random_scene_graph(seed, flags)
            blob, desc = b.compile(world), b.describe(world)
        except api.RtError as e:          # deeper than the supported instance nesting: a documented limit
            assert e.code == -2
            continue
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        rng = np.random.default_rng(1000 + seed)
        rays = util.random_rays(rng, 1500, any_time=flags == 4)
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        want = osc.trace_first_hit(rays, xi)
        got, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
        rep = util.first_hit_report(got, want)
        assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, (seed, flags, rep)
        assert (want["prim_id"] != MISS).mean() > 0.01


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_instanced_leaves_in_every_encoding(variant):
    """A primitive under Translate / RotateY objects is named by chain and primitive number in the reference itself where
    both fit, by linked instance records where they do not: first hits, hit records and renders agree with the oracle
    in both forms, in lists and in trees, with the compiler's trees, the reference's, and the device stand-in."""
    for flags in (0, 2, 8):
        b, world = util.many_instances_scene(variant, flags)
        blob, desc = b.compile(world), b.describe(world)
        refs = np.frombuffer(blob, dtype="<u4")
        h = __import__("test_abi_and_compiler").header(blob)
        nodes = np.frombuffer(blob, dtype="<u4", count=h["sections"]["nodes"][1] * 16, offset=h["sections"]["nodes"][0]).reshape(-1, 16)
        entries = np.frombuffer(blob, dtype="<u4", count=h["sections"]["list_entries"][1], offset=h["sections"]["list_entries"][0])
        kinds = set((np.concatenate([nodes[:, 12], nodes[:, 13], entries]) >> 28).tolist())
        assert 10 in kinds and h["sections"]["chains"][1] >= 1
        if variant == 0:
            assert 7 in kinds and h["sections"]["chains"][1] == 64   # the table is full: the rest is linked records
        osc = ck.OracleScene(desc, share_bvh_axes=True)
        rng = np.random.default_rng(500 + variant)
        rays = util.random_rays(rng, 12000 if variant == 2 else 4000, any_time=False)
        xi = rng.random((len(rays), osc.n_media)) if osc.n_media else None
        want = osc.trace_first_hit(rays, xi)
        got, _ = ck.hostsim_trace_first_hit(blob, rays, xi)
        rep = util.first_hit_report(got, want)
        assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, (variant, flags, rep)
        hit = rep["both_hit"]
        if variant == 2 and flags == 0:   # the object whose chain the table cut in two is among those the rays hit
            INST = np.dtype([("a", "<f8"), ("b", "<f8"), ("c", "<f8"), ("op", "<u4"), ("child", "<u4"), ("parent", "<i4"), ("src", "<u4")])
            insts = np.frombuffer(blob, dtype=INST, count=h["sections"]["insts"][1], offset=h["sections"]["insts"][0])
            cut = [r for r in insts if int(r["child"]) >> 28 == 7 and int(insts[int(r["child"]) & 0x03FFFFFF]["child"]) >> 28 == 10]
            assert len(cut) == 1 and cut[0]["op"] & 0x100 == 0 and insts[int(cut[0]["child"]) & 0x03FFFFFF]["op"] & 0x100 == 0
            named = int(insts[int(cut[0]["child"]) & 0x03FFFFFF]["child"])
            table = ("sphere_meta", "msphere_meta", "rect_meta", "box_meta")[(named >> 26) & 3]
            meta = np.frombuffer(blob, dtype="<u4", count=h["sections"][table][1] * 4, offset=h["sections"][table][0]).reshape(-1, 4)
            assert int(meta[named & 0xFFFFF][2]) in set(want["prim_id"][hit].tolist())
        for f in ("px", "py", "pz", "nx", "ny", "nz"):   # the hit record is rebuilt through the member's instance records
            assert util.abs_err(got[f][hit], want[f][hit]).max() < 1e-12, (variant, flags, f)
        assert np.array_equal(got["flags"][hit], want["flags"][hit]) and np.array_equal(got["material_id"][hit], want["material_id"][hit])
        assert (want["prim_id"] != MISS).mean() > (0.03 if variant == 2 else 0.1)
    cam = api.make_camera((0, 3, -24), (0, 0, 0), (0, 1, 0), 40.0, 4 / 3, 0.1, 20.0, 0.0, 1.0)
    p = api.make_params(20, 15, max_depth=10, background=(0.7, 0.8, 1.0), seed=91 + variant)
    a, _ = ck.hostsim_render(blob, cam, p, 2)
    w, _ = osc.render(cam, p, 2, threads=4)
    assert float((util.rel_err(a, w) > 1e-9).mean()) == 0.0 and a.mean() > 0.0


def test_ray_times_outside_the_interval_a_bvh_over_moving_spheres_was_built_for():
    """random_scene keeps ~390 moving spheres in BvhNode::new_(.., 0.0, 1.0): rebuilt by the compiler, that tree is right
    for ray times in [0, 1] only and says so; compiled with flag 4 it is replayed in the reference's form for any time."""
    import copy
    _, info, blob, desc = util.build_scene("random_scene")
    cam = copy.copy(info.camera)
    cam.time0, cam.time1 = -1.0, 3.0
    p = api.params_from_info(info, width=24, height=16, seed=util.RENDER_SEED)
    with pytest.raises(RuntimeError):
        ck.hostsim_render(blob, cam, p, 1)
    rays = ck.oracle_generate_rays(cam, p, 0)
    assert rays["time"].min() < 0.0 and rays["time"].max() > 1.0
    with pytest.raises(RuntimeError):
        ck.hostsim_trace_first_hit(blob, rays, None)
    _, info, blob, desc = util.build_scene("random_scene", flags=4)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    got, _ = ck.hostsim_trace_first_hit(blob, rays, None)
    rep = util.first_hit_report(got, osc.trace_first_hit(rays, None))
    assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, rep
    a, _ = ck.hostsim_render(blob, cam, p, 2)
    b, _ = osc.render(cam, p, 2, threads=4)
    assert float((util.rel_err(a, b) > 1e-9).mean()) <= 0.002


@pytest.mark.parametrize("seed", range(8))
def test_random_scene_graphs_render_like_the_oracle(seed):
    try:
        b, world = util.random_scene_graph(seed, 0)
        blob, desc = b.compile(world), b.describe(world)
    except api.RtError as e:
        assert e.code == -2
        return
    cam = api.make_camera((0, 2, -26), (0, 0, 0), (0, 1, 0), 40.0, 4 / 3, 0.1, 20.0, 0.0, 1.0)
    p = api.make_params(20, 15, max_depth=12, background=(0.7, 0.8, 1.0), seed=77 + seed)
    a, _ = ck.hostsim_render(blob, cam, p, 2)
    w, _ = ck.OracleScene(desc, share_bvh_axes=True).render(cam, p, 2, threads=4)
    assert float((util.rel_err(a, w) > 1e-9).mean()) == 0.0 and a.mean() > 0.0


def test_replayed_boxes_are_tested_on_their_f64_bounds():
    """A BvhNode over a RotateY keeps the child's UNROTATED box (hittable.rs:115-143), so whether a ray gets at the rotated
    geometry depends on the verdict of that box test (aabb.rs:35-75) - on the f64 bounds, which f32 does not hold when they
    are 0.1, 1.7, 2.9 ...: rays that pass within a fraction of an f32 ulp outside a face of the box must be turned away as
    the reference turns them away, although the node's own (outward-rounded f32) copy of the box would let them in."""
    b = api.Builder(seed=11)
    grey = b.lambertian_color(0.5, 0.5, 0.5)
    p0, p1 = np.array([0.1, 0.2, 0.3]), np.array([1.7, 2.9, 3.1])
    box = b.rotate_y(b.box(tuple(p0), tuple(p1), grey), 20.0)
    other = b.rotate_y(b.box((4.1, 0.3, 0.7), (5.3, 1.9, 2.3), grey), -30.0)
    world = b.list([b.bvh(b.list([box, other]), 0.0, 1.0)])
    blob, desc = b.compile(world), b.describe(world)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    rng = np.random.default_rng(3)
    rays = []
    for axis in range(3):
        for bound, sign in ((p0[axis], -1.0), (p1[axis], 1.0)):
            gap = abs(float(np.float32(bound)) - bound) + np.spacing(np.float32(bound)).astype(np.float64)   # about one f32 ulp
            n = 4000
            o = rng.uniform(p0 - 0.5, p1 + 0.5, (n, 3))
            d = rng.normal(size=(n, 3))
            d[:, axis] = 0.0                                           # in a plane parallel to the face ...
            o[:, axis] = bound + sign * rng.uniform(0.0, 1.0, n) * gap  # ... a hair outside it
            rays.append(util.rays_from(o, d))
    rays = np.concatenate(rays)
    want = osc.trace_first_hit(rays)
    got, _ = ck.hostsim_trace_first_hit(blob, rays)
    assert np.array_equal(got["prim_id"], want["prim_id"])
    assert np.array_equal(got["t"].view(np.uint64), want["t"].view(np.uint64))
    assert (want["prim_id"] == MISS).mean() > 0.5                   # (most of them are turned away; with the f32 copy of the bounds some of those got in)


def empty_worlds():
    """-> [(blob, description)]: a world with nothing in it, one that holds only empty lists (one of them under a RotateY),
    and one with a sphere between two empty lists. HittableList::hit finds nothing in an empty list (hittable_list.rs:40-53)."""
    out = []
    for kind in range(3):
        b = api.Builder(seed=1)
        grey = b.lambertian_color(0.5, 0.5, 0.5)
        if kind == 0:
            w = b.list()
        elif kind == 1:
            w = b.list([b.list(), b.rotate_y(b.list(), 10.0)])
        else:
            w = b.list([b.list(), b.sphere((0, 0, 0), 1.0, grey), b.translate(b.list(), (1, 2, 3))])
        out.append((b.compile(w), b.describe(w)))
    return out


def test_empty_worlds():
    cam = api.make_camera((0, 0, 5), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 5.0, 0.0, 1.0)
    p = api.make_params(16, 12, max_depth=5, background=(0.25, 0.5, 1.0), seed=3)
    rays = util.rays_from([(0, 0, 5), (3, 3, 5)], [(0, 0, -1), (0, 0, -1)])
    for kind, (blob, desc) in enumerate(empty_worlds()):
        osc = ck.OracleScene(desc)
        got, segments = ck.hostsim_render(blob, cam, p, 3)
        want, st = osc.render(cam, p, 3)
        assert np.array_equal(got, want) and segments == st.ray_segments
        if kind < 2:
            assert segments == 16 * 12 * 3 and np.array_equal(got[0, 0], 3 * np.array([0.25, 0.5, 1.0]))   # every ray: the background
        hit, _ = ck.hostsim_trace_first_hit(blob, rays)
        assert np.array_equal(hit["prim_id"], osc.trace_first_hit(rays)["prim_id"]) and hit["prim_id"][1] == MISS


def test_the_references_vec3_assertions_hold_on_the_products_vector_type():
    """vec3.rs:259-399 - twenty known answers in a commented-out test module, the only ones the reference holds - replayed
    on V3 of raytracer_amd/csrc/rt_math.h as the host compiler builds it (the GPU suite runs the same function on the device)."""
    import ctypes as C
    h = ck.hostsim()
    h.hs_vec3_kat.restype = C.c_uint32
    h.hs_vec3_kat.argtypes = [C.c_double]
    assert h.hs_vec3_kat(1.0) == 0
    assert h.hs_vec3_kat(1.0000001) != 0     # (the check does look at the values: a scale factor breaks the equalities)
