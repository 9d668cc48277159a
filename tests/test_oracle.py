"""The CPU oracle against everything that pins it: the reference's own (commented-out) Vec3 assertions
(raytracer/src/vec3.rs:259-399), the published Philox-4x32-10 known-answer vectors, and hand-derived analytic
cases for every primitive, instance, medium, texture and the pixel conversion. The reference holds no other
test, image or vector, so beyond the Vec3 block parity is pinned by derivation from the source only.

Every analytic case is also run through the product's per-lane device code (host simulator), which must agree."""
import ctypes as C
import math

import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

MISS = 0xFFFFFFFF


def both(builder, world, rays, xi=None, compat=0):
    """first hits from the oracle and from the product's device code on the CPU; asserts they agree."""
    blob, desc = builder.compile(world), builder.describe(world)
    o = ck.OracleScene(desc, share_bvh_axes=True).trace_first_hit(rays, xi, compat)
    h, _ = ck.hostsim_trace_first_hit(blob, rays, xi, compat)
    assert np.array_equal(o["prim_id"], h["prim_id"])
    for f in ("t", "px", "py", "pz", "nx", "ny", "nz", "flags", "material_id"):
        assert np.array_equal(o[f], h[f]), f
    return o


def test_reference_vec3_assertions():
    assert ck.oracle.orc_vec3_kat() == 0


@pytest.mark.parametrize("ctr,key,want", [
    ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
    ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
    ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
])
def test_philox_known_answers(ctr, key, want):
    # Random123 kat_vectors for philox4x32_10; checked for the oracle's and the product's implementation
    for fn in (ck.oracle.orc_philox, ck.hostsim().hs_philox):
        out = (C.c_uint32 * 4)()
        fn((C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), out)
        assert list(out) == want


def test_sphere_hit_and_uv():
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    s = b.sphere((0, 0, 0), 1.0, m)
    w = b.list([s])
    rays = util.rays_from([(0, 0, -5), (0, 0, 0), (0, 5, 0), (5, 0, 0), (0, 0, -5), (0, 0, -5)],
                          [(0, 0, 1), (0, 0, 1), (0, -1, 0), (-1, 0, 0), (0, 1, 0), (0, 0, -1)])
    h = both(b, w, rays)
    # outside, along +z: t = 4, p = (0,0,-1), normal = outward, front face (sphere.rs:34-51)
    assert h["t"][0] == 4.0 and (h["px"][0], h["py"][0], h["pz"][0]) == (0.0, 0.0, -1.0)
    assert (h["nx"][0], h["ny"][0], h["nz"][0]) == (0.0, 0.0, -1.0) and h["flags"][0] & 2
    # u = (atan2(-z, x) + pi) / 2pi = (atan2(1, 0) + pi) / 2pi = 0.75 ; v = acos(-y)/pi = 0.5 (sphere.rs:24-30)
    assert h["u"][0] == pytest.approx(0.75, abs=1e-15) and h["v"][0] == pytest.approx(0.5, abs=1e-15)
    # from the centre: second root, t = 1, normal flipped against the ray, back face (sphere.rs:52-60)
    assert h["t"][1] == 1.0 and h["nz"][1] == -1.0 and not (h["flags"][1] & 2)
    # north pole: v = acos(-1)/pi = 1
    assert h["t"][2] == 4.0 and h["v"][2] == pytest.approx(1.0, abs=1e-15)
    # +x: u = (atan2(0, 1) + pi)/2pi = 0.5
    assert h["u"][3] == pytest.approx(0.5, abs=1e-15)
    # misses: parallel offset ray, ray pointing away (both roots negative -> below t_min)
    assert h["prim_id"][4] == MISS and h["prim_id"][5] == MISS
    assert h["prim_id"][0] == s


def test_sphere_tangent_ray_misses_and_moving_sphere_tangent_hits():
    # Sphere needs delta > 0 (sphere.rs:40); MovingSphere only rejects discriminant < 0 (moving_sphere.rs:51)
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    w = b.list([b.sphere((0, 0, 0), 1.0, m)])
    tangent = util.rays_from([(1, 0, -5)], [(0, 0, 1)])
    assert both(b, w, tangent)["prim_id"][0] == MISS
    b2 = api.Builder()
    m2 = b2.lambertian_color(0.5, 0.5, 0.5)
    ms = b2.moving_sphere((0, 0, 0), (0, 0, 0), 0.0, 1.0, 1.0, m2)
    h = both(b2, b2.list([ms]), tangent)
    assert h["prim_id"][0] == ms and h["t"][0] == 5.0


def test_moving_sphere_centre_and_inclusive_range():
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    ms = b.moving_sphere((0, 0, 0), (2, 0, 0), 0.0, 1.0, 1.0, m)
    w = b.list([ms])
    # at time 0.5 the centre is (1,0,0) (moving_sphere.rs:37-40): a ray down x = 1 hits at z = -1
    r = util.rays_from([(1, 0, -5), (1, 0, -5)], [(0, 0, 1)], time=[0.5, 0.5])
    r["tmax"][1] = 4.0  # root == t_max is accepted (moving_sphere.rs:56)
    h = both(b, w, r)
    assert h["t"][0] == 4.0 and h["t"][1] == 4.0 and h["prim_id"][1] == ms
    assert h["u"][0] == 0.0 and h["v"][0] == 0.0  # moving spheres never write u, v


def test_rects_inclusive_bounds_uv_and_normals():
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    xy = b.xy_rect(0, 2, 0, 4, 3.0, m)
    xz = b.xz_rect(0, 2, 0, 4, -3.0, m)
    yz = b.yz_rect(0, 2, 0, 4, 7.0, m)
    w = b.list([xy, xz, yz])
    rays = util.rays_from([(0.5, 1.0, 0.0), (2.0, 4.0, 10.0), (0.5, 0.0, 1.0), (0.0, 0.5, 1.0), (2.0000001, 1, 0)],
                          [(0, 0, 1), (0, 0, -1), (0, -1, 0), (1, 0, 0), (0, 0, 1)])
    h = both(b, w, rays)
    assert h["prim_id"].tolist() == [xy, xy, xz, yz, MISS]
    assert h["t"][0] == 3.0 and h["u"][0] == 0.25 and h["v"][0] == 0.25            # arrect.rs:41-42
    assert (h["nx"][0], h["ny"][0], h["nz"][0]) == (0.0, 0.0, -1.0) and not (h["flags"][0] & 2)  # ray along +z sees the back
    assert h["t"][1] == 7.0 and h["u"][1] == 1.0 and h["v"][1] == 1.0 and h["flags"][1] & 2      # corner is inside (arrect.rs:38)
    assert h["t"][2] == 3.0 and (h["nx"][2], h["ny"][2], h["nz"][2]) == (0.0, 1.0, 0.0)
    assert h["t"][3] == 7.0 and (h["nx"][3], h["ny"][3], h["nz"][3]) == (-1.0, 0.0, 0.0)


def test_box_faces_and_equal_t_tie():
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    bx = b.box((0, 0, 0), (1, 2, 3), m)
    w = b.list([bx])
    rays = util.rays_from([(0.5, 1, -1), (0.5, 1, 4), (0.5, 5, 1), (0.5, -5, 1), (5, 1, 1), (-5, 1, 1), (0.5, 1, 1), (-1, -1, 1.5)],
                          [(0, 0, 1), (0, 0, -1), (0, -1, 0), (0, 1, 0), (-1, 0, 0), (1, 0, 0), (0, 0, 1), (1, 1, 0)])
    h = both(b, w, rays)
    faces = (h["flags"] >> 4) & 7
    # side order of _box.rs:22-71: 0 XY@z1, 1 XY@z0, 2 XZ@y1, 3 XZ@y0, 4 YZ@x1, 5 YZ@x0
    assert faces[:6].tolist() == [1, 0, 2, 3, 4, 5]
    assert h["t"][:6].tolist() == [1.0, 1.0, 3.0, 5.0, 4.0, 5.0]
    assert h["t"][6] == 2.0 and faces[6] == 0                     # from inside
    # the ray through the edge x = 0, y = 0 reaches XZ@y0 and YZ@x0 at the same t = 1: the later side (5) wins
    # because rects accept t == t_max (arrect.rs:33, hittable_list.rs:46-49)
    assert h["t"][7] == 1.0 and faces[7] == 5
    assert (h["prim_id"] == bx).all()


def test_translate_and_rotate_y_book_and_reference_behaviour():
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    bx = b.box((0, 0, 0), (2, 2, 2), m)
    inst = b.translate(b.rotate_y(bx, 90.0), (10, 0, 0))
    w = b.list([inst])
    # RotateY(90) maps local (x, z) to world (z, -x): the box occupies x in [0,2], z in [-2,0]; +10 in x.
    rays = util.rays_from([(11, 1, -10)], [(0, 0, 1)])
    ref = both(b, w, rays, compat=0)
    book = both(b, w, rays, compat=1)
    assert ref["prim_id"][0] == bx and ref["t"][0] == pytest.approx(8.0, abs=1e-12) and ref["t"][0] == book["t"][0]
    s, c = math.sin(math.radians(90.0)), math.cos(math.radians(90.0))
    # local hit: rotated ray origin (ox', oz') = (c*1 - s*(-10), s*1 + c*(-10)) -> hits YZ@x1 (x = 2) ... local p
    lp = np.array([c * 1 + s * 10, 1.0, s * 1 - c * 10]) + 8.0 * np.array([-s * 1.0, 0.0, c * 1.0])
    assert book["px"][0] == pytest.approx(c * lp[0] + s * lp[2] + 10, abs=1e-9)
    assert book["pz"][0] == pytest.approx(-s * lp[0] + c * lp[2], abs=1e-9)
    assert (book["px"][0], book["pz"][0]) == (pytest.approx(11.0, abs=1e-9), pytest.approx(-2.0, abs=1e-9))
    # reference (hittable.rs:167-168): p.x = c*x + s*z ; p.z = -s*(new p.x) + c*z
    px_ref = c * lp[0] + s * lp[2]
    pz_ref = -s * px_ref + c * lp[2]
    assert ref["px"][0] == pytest.approx(px_ref + 10, abs=1e-9) and ref["pz"][0] == pytest.approx(pz_ref, abs=1e-9)
    assert abs(ref["pz"][0] - book["pz"][0]) > 0.5                # the quirk is visible
    # Translate re-runs set_face_normal with the already flipped normal (hittable.rs:76-77): front_face ends up true
    assert ref["flags"][0] & 2


def test_rotate_y_keeps_unrotated_bbox_unless_book_flag():
    # hittable.rs:115-143: the rotated extents are computed and dropped. Under a BVH the instance is then culled
    # against the child's unrotated box.
    def scene(flags):
        b = api.Builder(flags=flags)
        m = b.lambertian_color(0.5, 0.5, 0.5)
        bx = b.box((0, 0, 0), (4, 1, 1), m)
        far = b.sphere((0, 0, 100), 0.5, m)
        rot = b.rotate_y(bx, 90.0)
        lst = b.list([rot, far])
        return b, b.list([b.bvh(lst, 0.0, 1.0)]), bx
    # rotated box occupies x in [0,1], z in [-4,0]; its stored (unrotated) box is x in [0,4], z in [0,1]
    b0, w0, bx0 = scene(0)
    b1, w1, bx1 = scene(1)
    rays2 = util.rays_from([(0.5, 0.5, -3.0)], [(1e-9, 0, -1)])   # travels away from z >= 0: never inside the stored box
    assert both(b0, w0, rays2)["prim_id"][0] == MISS              # reference: culled although the box is there
    assert both(b1, w1, rays2)["prim_id"][0] == bx1               # book bbox: found


def test_constant_medium_free_flight():
    b = api.Builder()
    m = b.dielectric(1.5)
    s = b.sphere((0, 0, 0), 2.0, m)
    med = b.constant_medium(s, 0.5, 1, 1, 1)
    w = b.list([med])
    xi = np.array([[0.5], [1e-30], [0.5], [0.5]])
    rays = util.rays_from([(0, 0, -5), (0, 0, -5), (0, 0, 0), (0, 0, -5)], [(0, 0, 2), (0, 0, 2), (0, 0, 2), (0, 0, 2)])
    rays["tmax"][3] = 1.6
    h = both(b, w, rays, xi)
    # enters at t1 = 1.5, leaves at 3.5; |d| = 2; hit_distance = -(1/0.5) ln 0.5 = 1.386 -> t = 1.5 + 1.386/2
    assert h["t"][0] == pytest.approx(1.5 + (-2.0 * math.log(0.5)) / 2.0, rel=1e-15) and h["prim_id"][0] == med
    assert (h["nx"][0], h["ny"][0], h["nz"][0]) == (1.0, 0.0, 0.0) and h["flags"][0] & 2   # constant_medium.rs:62-63
    assert h["prim_id"][1] == MISS                       # free flight longer than the 4 units inside
    # from inside: rec1.t < t_min is clamped to t_min = 0.001 (constant_medium.rs:36-38)
    assert h["t"][2] == pytest.approx(0.001 + (-2.0 * math.log(0.5)) / 2.0, rel=1e-15)
    # a closer t_max clips the exit: distance inside = (1.6 - 1.5) * 2 = 0.2 < 1.386 -> miss
    assert h["prim_id"][3] == MISS
    # without injected uniforms media never hit
    assert both(b, w, rays, None)["prim_id"].tolist() == [MISS] * 4


def test_list_order_and_bvh_agree():
    rng = np.random.default_rng(7)
    b = api.Builder()
    m = b.lambertian_color(0.5, 0.5, 0.5)
    objs = [b.sphere(tuple(rng.uniform(-10, 10, 3)), float(rng.uniform(0.3, 1.5)), m) for _ in range(200)]
    lst = b.list(objs)
    flat = b.list([lst])
    origins = rng.uniform(-30, 30, (4000, 3))
    rays = util.rays_from(origins, rng.uniform(-10, 10, (4000, 3)) - origins)
    h_list = both(b, flat, rays)
    tree = b.list([b.bvh(lst, 0.0, 1.0)])
    h_tree = both(b, tree, rays)
    assert np.array_equal(h_list["prim_id"], h_tree["prim_id"]) and np.array_equal(h_list["t"], h_tree["t"])
    assert (h_list["prim_id"] != MISS).mean() > 0.2


def test_textures_and_materials_through_sample_radiance():
    # a diffuse light seen directly: radiance = emit * k, both faces (material.rs:201-203)
    b = api.Builder()
    chk = b.checker(b.solid(1, 0, 0), b.solid(0, 0, 1))
    light = b.diffuse_light(chk, 2.0)
    w = b.list([b.xy_rect(-100, 100, -100, 100, 5.0, light)])
    blob, desc = b.compile(w), b.describe(w)
    cam = api.make_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 40.0, 1.0, 0.0, 1.0)
    p = api.make_params(8, 8, max_depth=5, seed=3)
    osc = ck.OracleScene(desc)
    for x, y in [(1, 1), (6, 2), (3, 7)]:
        got_o = osc.sample_radiance(cam, p, x, y, 0)
        got_h = ck.hostsim_sample_radiance(blob, cam, p, x, y, 0)
        assert np.array_equal(got_o, got_h)
        assert got_o.tolist() in ([2.0, 0.0, 0.0], [0.0, 0.0, 2.0])   # checker picks even (red) or odd (blue), times k


def test_furnace_white_sphere_in_uniform_light():
    # a Lambertian sphere of albedo 1 under a constant background of radiance 1 reflects exactly 1 along every path
    b = api.Builder()
    w = b.list([b.sphere((0, 0, 0), 1.0, b.lambertian_color(1, 1, 1))])
    blob, desc = b.compile(w), b.describe(w)
    cam = api.make_camera((0, 0, -4), (0, 0, 0), (0, 1, 0), 40.0, 1.0, 0.0, 4.0)
    p = api.make_params(16, 16, max_depth=50, background=(1, 1, 1), seed=11)
    img_o, st = ck.OracleScene(desc).render(cam, p, 8, threads=2)
    img_h, seg = ck.hostsim_render(blob, cam, p, 8)
    assert st.ray_segments == seg
    assert np.array_equal(img_o, img_h)
    # paths end on the background with throughput exactly 1 unless they exhaust 50 bounces (vanishingly rare here)
    assert np.all(img_o == 8.0)


def test_pixel_conversion():
    # main.rs:172-189: sqrt(sum/spp), clamp to [0, 0.999], * 255.999, truncate
    sums = np.array([[0.0, 4.0, 1.0], [16.0, 0.25 * 4, -1.0], [np.nan, 1e300, 2.0]])
    want = []
    for v in sums.ravel():
        c = math.sqrt(v / 4.0) if v >= 0 else float("nan")
        c = c if math.isnan(c) else min(max(c, 0.0), 0.999)
        want.append(0 if math.isnan(c) else int(c * 255.999))
    assert ck.oracle_tonemap(sums, 4).ravel().tolist() == want
    assert api.tonemap(sums, 4).ravel().tolist() == want
    assert want[:3] == [0, 255, 127]


# ---- textures seen through a light: emitted = texture.value(u, v, p) * k (material.rs:201-203), nothing random after the camera ray

def lit_rect_samples(b, tex, k_light, rect, cam, width, height, n_samples, seed=5):
    """Radiance of (pixel, sample) pairs looking at an xy_rect light with texture `tex`, from the oracle and from the
    product's device code, which must agree to the bit; with them the camera rays, so that the caller can redo the
    lookup by hand."""
    x0, x1, y0, y1, z = rect
    w = b.list([b.xy_rect(x0, x1, y0, y1, z, b.diffuse_light(tex, k_light))])
    blob, desc = b.compile(w), b.describe(w)
    p = api.make_params(width, height, max_depth=4, seed=seed)
    osc = ck.OracleScene(desc)
    out = []
    for s in range(n_samples):
        rays = ck.hostsim_generate_rays(cam, p, s)
        assert np.array_equal(rays, ck.oracle_generate_rays(cam, p, s))
        for y in range(height):
            for x in range(width):
                r = rays[y * width + x]
                got = osc.sample_radiance(cam, p, x, y, s)
                assert np.array_equal(got, ck.hostsim_sample_radiance(blob, cam, p, x, y, s))
                t = (z - r["oz"]) / r["dz"]                                   # aarect.rs:35-52
                px, py = r["ox"] + t * r["dx"], r["oy"] + t * r["dy"]
                inside = x0 <= px <= x1 and y0 <= py <= y1
                out.append((got, inside, (px, py, z), ((px - x0) / (x1 - x0), (py - y0) / (y1 - y0))))
    return out, desc


def test_image_texture_lookup_by_hand():
    # texture.rs:105-127: clamp u, v; v flipped; i = trunc(u*w), j = trunc(v*h), both capped; row j of the file from the top
    rng = np.random.default_rng(3)
    texels = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)               # 7 wide, 5 high, every texel different
    b = api.Builder()
    cam = api.make_camera((0, 0, 0), (0, 0, 1), (0, 1, 0), 90.0, 1.0, 0.0, 1.0)
    seen, _ = lit_rect_samples(b, b.image(texels), 1.5, (-0.9, 0.9, -0.8, 0.8, 1.0), cam, 12, 12, 3)
    hits = 0
    rows, cols = set(), set()
    for got, inside, _, (u, v) in seen:
        if not inside:
            assert got.tolist() == [0.0, 0.0, 0.0]                         # past the rect: the black background
            continue
        i = min(int(min(max(u, 0.0), 1.0) * 7), 6)
        j = min(int((1.0 - min(max(v, 0.0), 1.0)) * 5), 4)
        want = (1.0 / 255.0) * texels[j, i].astype(np.float64) * 1.5
        assert np.array_equal(got, want), (u, v, i, j)
        hits += 1
        rows.add(j)
        cols.add(i)
    assert hits > 200 and rows == set(range(5)) and cols == set(range(7))


def test_image_texture_corners_clamp():
    # u = 1 or v = 0 exactly gives index w or h, which the lookup caps to the last texel (texture.rs:113-118)
    texels = np.arange(2 * 2 * 3, dtype=np.uint8).reshape(2, 2, 3) * 20
    b = api.Builder()
    tex = b.image(texels)
    light = b.diffuse_light(tex, 1.0)
    # a sphere's (u, v): v = 0 at the south pole, 1 at the north pole (sphere.rs get_sphere_uv); look at both poles
    w = b.list([b.sphere((0, 0, 0), 1.0, light)])
    rays = util.rays_from([(0, 5, 0), (0, -5, 0)], [(0, -1, 0), (0, 1, 0)])
    h = both(b, w, rays)
    assert h["v"].tolist() == [1.0, 0.0]
    blob, desc = b.compile(w), b.describe(w)
    osc = ck.OracleScene(desc)
    for lookfrom, row in (((0, 5, 0), 0), ((0, -5, 0), 1)):                 # north pole: v=1 -> flipped 0 -> top row of the file
        cam = api.make_camera(lookfrom, (0, 0, 0), (0, 0, 1), 1e-9, 1.0, 0.0, 5.0)
        p = api.make_params(2, 2, max_depth=3, seed=2)
        got = osc.sample_radiance(cam, p, 0, 0, 0)
        assert np.array_equal(got, ck.hostsim_sample_radiance(blob, cam, p, 0, 0, 0))
        assert any(np.array_equal(got, (1.0 / 255.0) * texels[row, i].astype(np.float64)) for i in range(2)), (row, got)


def test_checker_texture_by_hand():
    # texture.rs:46-54: the sign of sin(10x) sin(10y) sin(10z) at the hit point; zero counts as even
    b = api.Builder()
    cam = api.make_camera((0.1, 0.2, 0), (0.1, 0.2, 1), (0, 1, 0), 100.0, 1.0, 0.0, 1.0)
    tex = b.checker(b.solid(0.25, 0.5, 0.75), b.checker(b.solid(1, 0, 0), b.solid(0, 1, 0)))   # nested: odd is a checker itself
    seen, _ = lit_rect_samples(b, tex, 2.0, (-5, 5, -5, 5, 1.3), cam, 16, 16, 2)
    kinds = {"even": 0, "odd": 0}
    for got, inside, (px, py, pz), _ in seen:
        assert inside
        sines = math.sin(10.0 * px) * math.sin(10.0 * py) * math.sin(10.0 * pz)
        if abs(sines) < 1e-12:
            continue
        if sines < 0.0:           # the inner checker sees the same point, so the same negative sign: always its odd colour
            assert got.tolist() == [0.0, 2.0, 0.0]
            kinds["odd"] += 1
        else:
            assert got.tolist() == [0.5, 1.0, 1.5]
            kinds["even"] += 1
    assert kinds["even"] > 100 and kinds["odd"] > 100


class PerlinByHand:
    """perlin.rs:56-125 written again in Python over the tables the scene description lists."""

    def __init__(self, desc, tex_id=0):
        line = [l for l in desc.split(b"\n") if l.startswith(b"tex_noise %d " % tex_id)][0].split()
        self.scale = float.fromhex(line[2].decode())
        self.vec = np.array([float.fromhex(v.decode()) for v in line[3:3 + 768]]).reshape(256, 3)
        perm = np.array([int(v) for v in line[3 + 768:3 + 768 + 768]]).reshape(3, 256)     # (behind them: the build stream's position)
        assert all(sorted(perm[a].tolist()) == list(range(256)) for a in range(3))   # three permutations
        assert np.all(np.abs(self.vec) <= 1.0)
        self.perm = perm

    def noise(self, p):
        fl = [math.floor(c) for c in p]
        u, v, w = (p[a] - fl[a] for a in range(3))
        uu, vv, ww = u * u * (3.0 - 2.0 * u), v * v * (3.0 - 2.0 * v), w * w * (3.0 - 2.0 * w)
        accum = 0.0
        for i in (0, 1):
            for j in (0, 1):
                for k in (0, 1):
                    g = self.vec[self.perm[0][255 & int(fl[0] + i)] ^ self.perm[1][255 & int(fl[1] + j)] ^ self.perm[2][255 & int(fl[2] + k)]]
                    dot = g[0] * (u - i) + g[1] * (v - j) + g[2] * (w - k)
                    accum += (i * uu + (1.0 - i) * (1.0 - uu)) * (j * vv + (1.0 - j) * (1.0 - vv)) * (k * ww + (1.0 - k) * (1.0 - ww)) * dot
        return accum

    def turb(self, p, depth=7):
        accum, weight, q = 0.0, 1.0, list(p)
        for _ in range(depth):
            accum += weight * self.noise(q)
            weight *= 0.5
            q = [2.0 * c for c in q]
        return abs(accum)

    def value(self, p):                                                     # texture.rs:73-79
        return 0.5 * (1.0 + math.sin(self.scale * p[2] + 10.0 * self.turb(p)))


def test_noise_texture_by_hand():
    b = api.Builder(seed=77)
    cam = api.make_camera((0.3, -0.2, -1), (0.3, -0.2, 1), (0, 1, 0), 120.0, 1.0, 0.0, 1.0)
    seen, desc = lit_rect_samples(b, b.noise(4.0), 1.0, (-40, 40, -40, 40, 2.5), cam, 14, 14, 2)   # points on both sides of 0: floor() of negatives
    perlin = PerlinByHand(desc)
    # on the integer lattice every corner's offset vector or its weight is zero: noise 0, turbulence 0 at every octave
    for q in [(0, 0, 0), (3, -7, 250), (-1, 255, 256), (1000, -1000, 5)]:
        assert perlin.noise([float(c) for c in q]) == 0.0 and perlin.turb([float(c) for c in q]) == 0.0
    # the lattice has period 256 in every axis (the `255 &` of perlin.rs:73-75)
    assert perlin.noise([1.25, -2.5, 3.75]) == perlin.noise([257.25, -258.5, 3.75 + 512.0])
    negative = 0
    for got, inside, p, _ in seen:
        assert inside
        want = perlin.value(p)
        assert got[0] == got[1] == got[2] and 0.0 <= got[0] <= 1.0
        assert abs(got[0] - want) <= 1e-12, (p, got[0], want)               # the sums are taken in the same order; sin() is libm's both times
        negative += p[0] < 0 and p[1] < 0
    assert negative > 20


# ---- materials: closed forms for what a scatter rule does to many samples of one (almost) fixed ray

def narrow_beam(lookfrom, lookat, background, spp, seed=9, depth=50):
    """A 2x2 image (the smallest: main.rs:159-160 divides by width - 1) under a field of view of 1e-7 degrees: every camera
    ray is the axis to within 1e-8 rad, so the samples are that many draws of whatever the materials do with that one ray.
    Returns (camera, params)."""
    up = (0, 1, 0) if abs(lookat[1] - lookfrom[1]) < 0.9 * math.dist(lookfrom, lookat) else (0, 0, 1)
    cam = api.make_camera(lookfrom, lookat, up, 1e-7, 1.0, 0.0, 1.0)
    return cam, api.make_params(2, 2, max_depth=depth, background=background, seed=seed)


def beam_sum(b, world, cam, p, samples):
    """Sum over `samples` draws (a quarter of them in each of the four pixels), from the oracle and the product's device code."""
    blob, desc = b.compile(world), b.describe(world)
    assert samples % 4 == 0
    img_o, st = ck.OracleScene(desc).render(cam, p, samples // 4, threads=2)
    img_h, seg = ck.hostsim_render(blob, cam, p, samples // 4)
    assert np.array_equal(img_o, img_h) and st.ray_segments == seg
    return img_o.reshape(4, 3).sum(axis=0), st


def test_mirror_reflects_albedo_exactly_and_fuzz_absorbs_the_cap():
    # material.rs:80-97: fuzz 0 -> one bounce, attenuation = albedo, then the white background
    b = api.Builder()
    w = b.list([b.xz_rect(-50, 50, -50, 50, 0.0, b.metal(0.75, 0.5, 0.25, 0.0))])
    cam, p = narrow_beam((0, 3, -3), (0, 0, 0), (1, 1, 1), 64)
    total, st = beam_sum(b, w, cam, p, 64)
    assert total.tolist() == [48.0, 32.0, 16.0] and st.ray_segments == 128
    # fuzz 1 at 45 degrees: reflected + s, s uniform in the unit ball; absorbed when its y falls to <= 0, i.e. s_y <= -cos 45:
    # a ball cap of height h = 1 - cos 45, volume fraction h^2 (3 - h) / 4 = 0.05806; survivors go up and leave the plane
    b = api.Builder()
    w = b.list([b.xz_rect(-1e6, 1e6, -1e6, 1e6, 0.0, b.metal(1.0, 0.5, 0.25, 1.0))])
    n = 40000
    total, _ = beam_sum(b, w, cam, p, n)
    h = 1.0 - math.sqrt(0.5)
    keep = 1.0 - h * h * (3.0 - h) / 4.0
    sigma = math.sqrt(keep * (1 - keep) / n)
    assert abs(total[0] / n - keep) < 4 * sigma, (total[0] / n, keep, sigma)
    assert total[1] == 0.5 * total[0] and total[2] == 0.25 * total[0]      # every survivor carries exactly the albedo
    # a fuzz above 1 is 1 (material.rs:68-72): the same draws, the same sum
    b2 = api.Builder()
    w2 = b2.list([b2.xz_rect(-1e6, 1e6, -1e6, 1e6, 0.0, b2.metal(1.0, 0.5, 0.25, 7.0))])
    total2, _ = beam_sum(b2, w2, cam, p, n)
    assert np.array_equal(total, total2)


def schlick(cosine, ref_idx):                                               # material.rs:109-113
    r0 = ((1.0 - ref_idx) / (1.0 + ref_idx)) ** 2
    return r0 + (1.0 - r0) * (1.0 - cosine) ** 5


def test_dielectric_interface_reflects_by_schlick_and_totally_from_inside():
    # one glass face (an XyRect: outward normal +z, aarect.rs:54). From the +z side at 60 degrees the ray is on the front:
    # ratio 1/1.5, refracts (towards the red light behind) unless reflectance(cos 60) > xi (material.rs:138-147): back to the blue sky
    n = 40000
    b = api.Builder()
    glass = b.dielectric(1.5)
    red = b.diffuse_light(b.solid(1, 0, 0), 1.0)
    w = b.list([b.xy_rect(-1e3, 1e3, -1e3, 1e3, 0.0, glass), b.xy_rect(-1e6, 1e6, -1e6, 1e6, -10.0, red)])
    s60, c60 = math.sin(math.radians(60)), math.cos(math.radians(60))
    cam, p = narrow_beam((5 * s60, 0, 5 * c60), (0, 0, 0), (0, 0, 1), n)
    total, _ = beam_sum(b, w, cam, p, n)
    assert total[1] == 0.0 and total[0] + total[2] == n                     # attenuation is ones: every sample is wholly red or wholly blue
    refl = schlick(c60, 1.0 / 1.5)
    assert abs(refl - 0.07) < 1e-12
    sigma = math.sqrt(refl * (1 - refl) / n)
    assert abs(total[2] / n - refl) < 4 * sigma, (total[2] / n, refl)
    # from the -z side the ray is inside: ratio 1.5, 1.5 sin 60 > 1, so it cannot refract and every sample reflects (back down to the red light)
    cam, p = narrow_beam((5 * s60, 0, -5 * c60), (0, 0, 0), (0, 0, 1), 256)
    total, st = beam_sum(b, w, cam, p, 256)
    assert total.tolist() == [256.0, 0.0, 0.0] and st.ray_segments == 512
    # just below the critical angle (asin(1/1.5) = 41.81 degrees) some get out
    s40, c40 = math.sin(math.radians(40)), math.cos(math.radians(40))
    cam, p = narrow_beam((5 * s40, 0, -5 * c40), (0, 0, 0), (0, 0, 1), 4000)
    total, _ = beam_sum(b, w, cam, p, 4000)
    refl = schlick(c40, 1.5)
    assert abs(total[0] / 4000 - refl) < 4 * math.sqrt(refl * (1 - refl) / 4000) and total[2] > 0


def test_glass_sphere_on_axis_transmits_one_minus_r_over_one_plus_r():
    # normal incidence on both faces: each meeting reflects with r = 0.04 whatever the side; a path leaves forwards after an odd
    # number of face meetings... summed over the internal bounces, forwards (1-r)^2 / (1-r^2) = (1-r)/(1+r)
    n = 40000
    b = api.Builder()
    red = b.diffuse_light(b.solid(1, 0, 0), 1.0)
    w = b.list([b.sphere((0, 0, 0), 1.0, b.dielectric(1.5)), b.xy_rect(-100, 100, -100, 100, 20.0, red)])
    cam, p = narrow_beam((0, 0, -6), (0, 0, 0), (0, 0, 1), n)
    total, _ = beam_sum(b, w, cam, p, n)
    assert total[1] == 0.0 and total[0] + total[2] == n
    want = (1 - 0.04) / (1 + 0.04)
    assert abs(total[0] / n - want) < 4 * math.sqrt(want * (1 - want) / n), (total[0] / n, want)


def test_hollow_glass_sphere_keeps_the_negative_radius_normal():
    # a sphere of radius -r has its outward normal pointing inwards ((p - centre) / radius, sphere.rs:50): the air in the middle
    # is its outside, the glass shell its inside. In a plain list no box is asked for, so its inside-out box does not matter.
    b = api.Builder()
    g = b.dielectric(1.5)
    w = b.list([b.sphere((0, 0, 0), 1.0, g), b.sphere((0, 0, 0), -0.9, g)])
    rays = util.rays_from([(0, 0, -5), (0, 0, -0.95), (0, 0, 0)], [(0, 0, 1), (0, 0, 1), (0, 0, 1)])
    h = both(b, w, rays)
    assert h["prim_id"].tolist() == [0, 1, 1]
    assert h["t"].tolist() == [4.0, pytest.approx(0.05, abs=1e-12), 0.9]
    assert (h["flags"] >> 1 & 1).tolist() == [1, 0, 1]                          # outer from outside: front; inner from the shell: back; inner from the middle: front
    assert h["nz"].tolist() == [-1.0, -1.0, -1.0]                           # always against the ray (hittable.rs set_face_normal)


def test_bvh_over_negative_radius_spheres_culls_as_the_reference_does():
    # sphere.rs:66-72 gives a sphere of radius r < 0 the box [c - r, c + r] = [c + |r|, c - |r|], inside out. A BvhNode tests no
    # box for an object it holds directly, only its own box = surrounding_box of the two (bvh.rs:55-60, aabb.rs:77-90), and that
    # union can cut parts of such a sphere off or, for a node holding it twice (bvh.rs:33-35), be inside out itself: never hit.
    b = api.Builder()
    g = b.dielectric(1.5)
    alone = b.list([b.bvh(b.list([b.sphere((0, 0, 0), -0.9, g)]), 0.0, 1.0)])
    rays = util.rays_from([(0, 0, -5), (0, 0, 0), (0.3, 0.1, -5)], [(0, 0, 1), (0, 1, 0), (0, 0, 1)])
    h = both(b, alone, rays)
    assert h["prim_id"].tolist() == [MISS, MISS, MISS]
    # with the outer sphere beside it under the node, the union is the outer sphere's box: a hollow ball as the book draws it
    b = api.Builder()
    g = b.dielectric(1.5)
    hollow = b.list([b.bvh(b.list([b.sphere((0, 0, 0), 1.0, g), b.sphere((0, 0, 0), -0.9, g)]), 0.0, 1.0)])
    h = both(b, hollow, util.rays_from([(0, 0, -5), (0, 0, -0.95), (0, 0, 0)], [(0, 0, 1), (0, 0, 1), (0, 0, 1)]))
    assert h["t"].tolist() == [4.0, pytest.approx(0.05, abs=1e-12), 0.9]
    # many of both signs, some of radius 0: whatever the unions cut off, the product cuts off too
    for trial in range(6):
        b, w, in_list, rays = util.inside_out_spheres(trial)
        h = both(b, w, rays)
        flat = both(b, in_list, rays)                                        # without the node: nothing is cut off
        assert (h["prim_id"] != MISS).sum() < (flat["prim_id"] != MISS).sum()
        assert (h["prim_id"] != MISS).mean() > 0.1
