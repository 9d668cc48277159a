"""What the product's scene functions (raytracer_amd/csrc/rt_scenes.cpp) put into a scene, checked against literals read
off the reference's own scene functions - main.rs:576-688 (final_scene), :409-486 (cornell_box), :690-835 (my_scene),
:382-407 (simple_light) - and the construction-time arithmetic the oracle would otherwise take on trust: the Perlin tables
(perlin.rs:15-54), which the oracle makes itself from the logged position of the build stream and which must equal the
product's. The parser below reads the constructor log (rt_builder_describe); it shares nothing with product or oracle."""
import ctypes as C
import math
from collections import Counter

import numpy as np
import pytest

import checkers as ck
import util


def parse(desc):
    """constructor log -> {id: (kind, [numbers...])} per handle space, plus list membership in order"""
    tex, mat, obj, members = {}, {}, {}, {}
    data = desc
    pos = 0
    while pos < len(data):
        end = data.find(b"\n", pos)
        if end < 0:
            end = len(data)
        tok = data[pos:end].split()
        pos = end + 1
        if not tok:
            continue
        op = tok[0].decode()

        def num(t):
            t = t.decode()
            return float.fromhex(t) if ("x" in t or "inf" in t or "nan" in t) else int(t)
        if op == "tex_image":          # header line, then raw texel bytes
            tex[int(tok[1])] = ("image", [int(tok[2]), int(tok[3])])
            pos += int(tok[4]) + 1
        elif op == "tex_noise":
            tex[int(tok[1])] = ("noise", [float.fromhex(tok[2].decode())])
        elif op.startswith("tex_"):
            tex[int(tok[1])] = (op[4:], [num(t) for t in tok[2:]])
        elif op.startswith("mat_"):
            mat[int(tok[1])] = (op[4:], [num(t) for t in tok[2:]])
        elif op == "list":
            obj[int(tok[1])] = ("list", [])
            members[int(tok[1])] = []
        elif op == "list_add":
            members[int(tok[1])].append(int(tok[2]))
        elif op == "bvh":
            obj[int(tok[1])] = ("bvh", [int(tok[2]), num(tok[3]), num(tok[4])])
        elif op in ("sphere", "moving_sphere", "xy_rect", "xz_rect", "yz_rect", "box", "translate", "rotate_y", "constant_medium"):
            obj[int(tok[1])] = (op, [num(t) for t in tok[2:]])
    return tex, mat, obj, members


def solid_of(tex, mat, m):
    kind, a = mat[m]
    assert kind in ("lambertian", "diffuse_light", "isotropic")
    assert tex[a[0]][0] == "solid"
    return tex[a[0]][1]


def world_of(name, **kw):
    b, info, blob, desc = util.build_scene(name, **kw)
    tex, mat, obj, members = parse(desc)
    return info, tex, mat, obj, members, desc


def test_final_scene_holds_what_main_rs_576_688_builds():
    info, tex, mat, obj, members, _ = world_of("final_scene")
    top = members[info.world]
    kinds = [obj[o][0] for o in top]
    assert kinds == ["bvh", "xz_rect", "moving_sphere", "sphere", "sphere", "sphere", "constant_medium", "constant_medium", "sphere", "sphere",
                     "translate"]                                                     # the 11 entries, in the order of main.rs:599-685
    # 20 x 20 boxes, 100 wide, from -1000, y1 in [1, 101), one green Lambertian (main.rs:577-597)
    boxes_list, t0, t1 = obj[top[0]][1]
    assert (t0, t1) == (0.0, 1.0) and len(members[boxes_list]) == 400
    seen = set()
    for o in members[boxes_list]:
        kind, (x0, y0, z0, x1, y1, z1, m) = obj[o]
        assert kind == "box" and y0 == 0.0 and x1 == x0 + 100.0 and z1 == z0 + 100.0 and 1.0 <= y1 < 101.0
        assert solid_of(tex, mat, m) == [0.48, 0.83, 0.53]
        seen.add((x0, z0))
    assert seen == {(-1000.0 + 100.0 * i, -1000.0 + 100.0 * j) for i in range(20) for j in range(20)}
    # the light (main.rs:603-606)
    *r, m = obj[top[1]][1]
    assert r == [123.0, 423.0, 147.0, 412.0, 554.0] and mat[m][0] == "diffuse_light" and solid_of(tex, mat, m) == [7.0, 7.0, 7.0] and mat[m][1][1] == 1.0
    # the moving sphere (main.rs:608-618)
    *r, m = obj[top[2]][1]
    assert r == [400.0, 400.0, 200.0, 430.0, 400.0, 200.0, 0.0, 1.0, 50.0] and solid_of(tex, mat, m) == [0.7, 0.3, 0.1]
    # glass, metal, the glass ball that is also a medium's boundary (main.rs:620-641)
    assert obj[top[3]][1][:4] == [260.0, 150.0, 45.0, 50.0] and mat[obj[top[3]][1][4]] == ("dielectric", [1.5])
    assert obj[top[4]][1][:4] == [0.0, 150.0, 145.0, 50.0] and mat[obj[top[4]][1][4]] == ("metal", [0.8, 0.8, 0.9, 1.0])
    assert obj[top[5]][1][:4] == [360.0, 150.0, 145.0, 70.0] and mat[obj[top[5]][1][4]] == ("dielectric", [1.5])
    boundary, density, *colour = obj[top[6]][1][:5]
    assert boundary == top[5] and density == 0.2 and colour == [0.2, 0.4, 0.9]
    # the fog around everything (main.rs:642-651): its boundary is not a member of the world
    boundary, density, *colour = obj[top[7]][1][:5]
    assert obj[boundary][1][:4] == [0.0, 0.0, 0.0, 5000.0] and boundary not in top and density == 0.0001 and colour == [1.0, 1.0, 1.0]
    # earth and marble (main.rs:653-666)
    assert obj[top[8]][1][:4] == [400.0, 200.0, 400.0, 100.0]
    e = mat[obj[top[8]][1][4]]
    assert e[0] == "lambertian" and tex[e[1][0]] == ("image", [1024, 512])                # earthmap.jpg is 1024 x 512
    assert obj[top[9]][1][:4] == [220.0, 280.0, 300.0, 80.0]
    n = mat[obj[top[9]][1][4]]
    assert n[0] == "lambertian" and tex[n[1][0]] == ("noise", [0.1])
    # Translate(RotateY(BvhNode(1000 spheres r = 10 in [0, 165)^3), 15), (-100, 270, 395)) (main.rs:668-685)
    child, *off = obj[top[10]][1]
    assert off == [-100.0, 270.0, 395.0] and obj[child][0] == "rotate_y" and obj[child][1][1] == 15.0
    cluster = obj[obj[child][1][0]]
    assert cluster[0] == "bvh" and cluster[1][1:] == [0.0, 1.0]
    spheres = members[cluster[1][0]]
    assert len(spheres) == 1000
    for o in spheres:
        kind, (x, y, z, r, m) = obj[o]
        assert kind == "sphere" and r == 10.0 and all(0.0 <= c < 165.0 for c in (x, y, z)) and solid_of(tex, mat, m) == [0.73, 0.73, 0.73]
    assert Counter(k for k, _ in obj.values()) == Counter({"box": 400, "sphere": 1006, "list": 3, "bvh": 2, "constant_medium": 2, "xz_rect": 1,
                                                          "moving_sphere": 1, "rotate_y": 1, "translate": 1})


def test_cornell_box_holds_what_main_rs_409_486_builds():
    info, tex, mat, obj, members, _ = world_of("cornell_box")
    # the reference returns the BvhNode itself; a world must be a list, so it is wrapped in a one-entry list
    assert [obj[o][0] for o in members[info.world]] == ["bvh"]
    lst, t0, t1 = obj[members[info.world][0]][1]
    assert (t0, t1) == (0.0, 0.0)
    # rt_bvh sorts its list in place as BvhNode::new does (bvh.rs:24-31), so compare as a set of constructor calls
    got = sorted((obj[o][0], tuple(obj[o][1][:-1]), tuple(solid_of(tex, mat, obj[o][1][-1]))) for o in members[lst] if obj[o][0].endswith("rect"))
    red, white, green, light = (0.65, 0.05, 0.05), (0.73, 0.73, 0.73), (0.12, 0.45, 0.15), (15.0, 15.0, 15.0)
    assert got == sorted([("yz_rect", (0.0, 555.0, 0.0, 555.0, 555.0), green), ("yz_rect", (0.0, 555.0, 0.0, 555.0, 0.0), red),
                          ("xz_rect", (213.0, 343.0, 227.0, 332.0, 554.0), light), ("xz_rect", (0.0, 555.0, 0.0, 555.0, 0.0), white),
                          ("xz_rect", (0.0, 555.0, 0.0, 555.0, 555.0), white), ("xy_rect", (0.0, 555.0, 0.0, 555.0, 555.0), white)])
    inst = sorted((tuple(obj[o][1][1:]), obj[obj[o][1][0]][1][1], tuple(obj[obj[obj[o][1][0]][1][0]][1][:6])) for o in members[lst]
                  if obj[o][0] == "translate")
    assert inst == sorted([((265.0, 0.0, 295.0), 15.0, (0.0, 0.0, 0.0, 165.0, 330.0, 165.0)),
                           ((130.0, 0.0, 65.0), -18.0, (0.0, 0.0, 0.0, 165.0, 165.0, 165.0))])
    assert len(members[lst]) == 8


def test_my_scene_holds_what_main_rs_690_835_builds():
    info, tex, mat, obj, members, _ = world_of("my_scene")
    top = members[info.world]
    assert [obj[o][0] for o in top] == ["box", "box"] + ["sphere"] * 7 + ["xz_rect", "xy_rect", "sphere", "bvh"]
    k = {}
    for o in top[:12]:
        m = mat[obj[o][1][-1]]
        k[o] = m
    assert obj[top[0]][1][:6] == [-4.0, -7.0, -7.0, 10.0, 7.0, 7.0] and k[top[0]][0] == "diffuse_light" and k[top[0]][1][1] == 1.5   # star1, k = 1.5 (material.rs:165-170)
    assert tex[k[top[0]][1][0]] == ("image", [857, 800])
    assert obj[top[1]][1][:6] == [-6.0, -9.0, -9.0, 12.0, 9.0, 9.0] and k[top[1]] == ("dielectric", [3.0])
    stars = [(-14.0, 8.0, 10.0, 3.0), (15.0, 7.0, -7.0, 2.0), (18.0, 11.0, 18.0, 4.0), (22.0, -13.0, 15.0, 1.5), (-10.0, -20.0, 17.0, 1.5),
             (17.0, -17.0, 17.0, 0.5), (-10.0, -9.0, -13.0, 2.0)]
    for o, want in zip(top[2:9], stars):
        assert tuple(obj[o][1][:4]) == want and k[o][0] == "diffuse_light" and k[o][1][1] == 1.5 and tex[k[o][1][0]] == ("image", [426, 375])   # star3
    assert obj[top[9]][1][:5] == [-100.0, 100.0, -100.0, 100.0, 10.0] and k[top[9]] == ("metal", [0.9, 0.9, 0.9, 0.0])
    assert obj[top[10]][1][:5] == [-100.0, 100.0, -100.0, 100.0, -20.0] and k[top[10]] == ("metal", [0.9, 0.9, 0.9, 0.0])
    assert obj[top[11]][1][:4] == [16.0, -10.0, 15.0, 5.0] and k[top[11]][1][1] == 0.5 and tex[k[top[11]][1][0]] == ("image", [646, 676])      # star5, k = 0.5
    # 3 x 10000 spheres of radius 0.1 on three rings in the plane y = -10 around (16, -10, 15) (main.rs:807-833)
    lst, t0, t1 = obj[top[12]][1]
    assert (t0, t1) == (0.0, 0.0) and len(members[lst]) == 30000
    radii = Counter()
    for o in members[lst]:
        x, y, z, r, m = obj[o][1]
        assert r == 0.1 and y == -10.0
        d = math.hypot(x - 16.0, z - 15.0)
        assert 8.0 * 0.8 - 1e-9 <= d < 10.0 + 1e-9
        radii[tex[mat[m][1][0]][1][0]] += 1
    assert radii == Counter({610: 20000, 646: 10000})                                  # star6 twice per round, star5 once


def test_simple_light_holds_what_main_rs_382_407_builds():
    info, tex, mat, obj, members, _ = world_of("simple_light")
    assert [obj[o][0] for o in members[info.world]] == ["bvh"]        # the reference returns the BvhNode; wrapped in a one-entry world
    lst, t0, t1 = obj[members[info.world][0]][1]
    assert (t0, t1) == (0.0, 0.0)
    top = sorted(members[lst], key=lambda o: (obj[o][0], obj[o][1][3]))   # (rt_bvh sorts the list in place, bvh.rs:24-31)
    assert [obj[o][0] for o in top] == ["sphere", "sphere", "xy_rect"]
    assert obj[top[0]][1][:4] == [0.0, 2.0, 0.0, 2.0] and obj[top[1]][1][:4] == [0.0, -1000.0, 0.0, 1000.0]
    assert tex[mat[obj[top[0]][1][4]][1][0]] == ("noise", [4.0]) and mat[obj[top[0]][1][4]] == mat[obj[top[1]][1][4]]
    assert obj[top[2]][1][:5] == [3.0, 5.0, 1.0, 3.0, -2.0] and solid_of(tex, mat, obj[top[2]][1][5]) == [4.0, 4.0, 4.0]


@pytest.mark.parametrize("name", ["final_scene", "two_perlin_spheres", "simple_light", "stress_volume"])
def test_the_oracle_makes_its_own_perlin_tables_and_they_equal_the_products(name):
    """Perlin::init (perlin.rs:46-54) and its shuffle (:28-34: `swap(p[i], p[random_int(0, n - i)])`, not Fisher-Yates) exist
    twice: in the scene builder, whose tables go into the blob, and in the oracle, which draws from its own restatement of the
    build stream at the position the log records. Equal tables: each is checked by something other than itself."""
    b, info, blob, desc = util.build_scene(name)
    osc = ck.OracleScene(desc)
    ck.oracle.orc_perlin_tables.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
    ids = [int(l.split()[1]) for l in desc.split(b"\n") if l.startswith(b"tex_noise ")]
    assert ids
    hdr = util.blob_header(blob)
    blob_vec = np.frombuffer(blob, dtype=np.float64, count=hdr["perlin_vec"][1], offset=hdr["perlin_vec"][0]).reshape(-1, 768)
    blob_perm = np.frombuffer(blob, dtype=np.int32, count=hdr["perlin_perm"][1], offset=hdr["perlin_perm"][0]).reshape(-1, 768)
    assert len(blob_vec) == len(ids)
    for k, t in enumerate(ids):
        own_v, own_p, log_v, log_p = np.zeros(768), np.zeros(768, np.int32), np.zeros(768), np.zeros(768, np.int32)
        assert ck.oracle.orc_perlin_tables(osc._s, t, own_v.ctypes.data, own_p.ctypes.data, log_v.ctypes.data, log_p.ctypes.data) == 0
        assert np.array_equal(own_v.view(np.uint64), blob_vec[k].view(np.uint64)) and np.array_equal(own_p, blob_perm[k])
        assert np.array_equal(log_v.view(np.uint64), blob_vec[k].view(np.uint64)) and np.array_equal(log_p, blob_perm[k])
        # and they are what perlin.rs asks for: gradients in [-1, 1)^3, three permutations of 0..255, none of them the identity
        assert np.all(own_v >= -1.0) and np.all(own_v < 1.0) and len(np.unique(own_v)) == 768
        for axis in range(3):
            p = own_p[256 * axis:256 * (axis + 1)]
            assert sorted(p.tolist()) == list(range(256)) and (p != np.arange(256)).sum() > 200
