"""The C-ABI library: it loads, exports every symbol include/rt.h declares, reports the reference's panic sites as
error codes, and its scene compiler emits the flat layout the kernels expect. No device compute here."""
import ctypes as C
import os
import re
import struct

import numpy as np
import pytest

import util
from raytracer_amd import _lib, api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "rt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    names = declared_functions()
    assert len(names) >= 40
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "librt_hip.so does not export " + n
    assert sorted(_lib.PROTOTYPES) == names, "ctypes prototypes and include/rt.h disagree"


def test_struct_sizes_match_the_header_layout():
    assert C.sizeof(_lib.rt_ray_d) == 72 and C.sizeof(_lib.rt_hit) == 88
    assert C.sizeof(_lib.rt_camera) == 15 * 8
    assert C.sizeof(_lib.rt_params) == 16 + 24 + 8 + 7 * 4 + 4


def test_init_without_a_device_is_an_error_code_not_a_crash():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert _lib.lib.rt_init(C.byref(h), 0) == _lib.RT_ERR_NO_DEVICE and not h.value


def test_reference_panic_sites_become_error_codes():
    b = api.Builder()
    with pytest.raises(_lib.RtError) as e:      # image::open(..).unwrap() (texture.rs:91)
        b.named_scene("earth")
    assert e.value.code == _lib.RT_ERR_PANIC
    with pytest.raises(_lib.RtError) as e:      # BvhNode over an empty list indexes out of bounds (bvh.rs:34)
        b.bvh(b.list(), 0.0, 1.0)
    assert e.value.code == _lib.RT_ERR_PANIC
    m = b.lambertian_color(1, 1, 1)
    empty = b.list()
    with pytest.raises(_lib.RtError) as e:      # "No bounding box" (bvh.rs:58,77): an empty list has none
        b.bvh(b.list([b.sphere((0, 0, 0), 1, m), empty]), 0.0, 1.0)
    assert e.value.code == _lib.RT_ERR_PANIC
    s = b.sphere((0, 0, 0), 1, m)
    for bad in (lambda: b.sphere((0, 0, 0), 1.0, 99), lambda: b.lambertian(99), lambda: b.translate(1234, (0, 0, 0)),
                lambda: b.list_add(s, empty), lambda: b.named_scene("no_such_scene")):
        with pytest.raises(_lib.RtError) as e:
            bad()
        assert e.value.code == _lib.RT_ERR_INVALID


def test_unsupported_shapes_are_refused_at_compile_time():
    b = api.Builder()
    m = b.lambertian_color(1, 1, 1)
    s = b.sphere((0, 0, 0), 1, m)
    deep = s
    for _ in range(5):                           # five nested transforms > RT_MAX_CHAIN
        deep = b.translate(deep, (1, 0, 0))
    with pytest.raises(_lib.RtError) as e:
        b.compile(b.list([deep]))
    assert e.value.code == _lib.RT_ERR_UNSUPPORTED
    lst = b.list([b.sphere((3, 0, 0), 1, m), b.sphere((6, 0, 0), 1, m)])
    med = b.constant_medium(b.bvh(lst, 0, 1), 0.1, 1, 1, 1)   # boundary that is not a primitive
    with pytest.raises(_lib.RtError) as e:
        b.compile(b.list([med]))
    assert e.value.code == _lib.RT_ERR_UNSUPPORTED


HEADER = struct.Struct("<IIQIIIIIIIIdddddQ")  # magic, version, total, world_flat_any_time, world_flat, world_tree, n_media, stack_need, max_chain, flags, n_objects, shutter0, shutter1, valid0, valid1


def header(blob):
    magic, version, total, world, world_flat, world_tree, n_media, stack_need, max_chain, flags, n_objects, shutter0, shutter1, valid0, valid1, node_extent, n_exact_nodes = HEADER.unpack_from(blob, 0)
    secs = np.frombuffer(blob, dtype="<u8", count=19 * 2, offset=HEADER.size).reshape(19, 2)
    names = ["nodes", "spheres", "sphere_meta", "mspheres", "msphere_meta", "rects", "rect_meta", "boxes", "box_meta", "media",
             "insts", "list_entries", "materials", "textures", "perlin_vec", "perlin_perm", "texels", "exact_boxes", "chains"]
    return dict(magic=magic, version=version, total=total, world_flat_any_time=world, world_flat=world_flat, world_tree=world_tree, n_media=n_media, stack_need=stack_need,
                shutter=(shutter0, shutter1), valid=(valid0, valid1),
                max_chain=max_chain, flags=flags, n_objects=n_objects, sections={n: tuple(map(int, s)) for n, s in zip(names, secs)})


NODE = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3), ("c0", "<u4"), ("c1", "<u4"), ("pad", "<u4", 2)])
REFERENCE_TREE = 2


def nodes_of(blob, h):
    assert NODE.itemsize == 64
    off, cnt = h["sections"]["nodes"]
    return np.frombuffer(blob, dtype=NODE, count=cnt, offset=off)


def is_empty(lo, hi):
    return np.all(np.isposinf(lo)) and np.all(np.isneginf(hi))


def is_infinite(lo, hi):
    return np.all(np.isneginf(lo)) and np.all(np.isposinf(hi))


def test_blob_layout_of_the_book2_final_scene():
    _, info, blob, _ = util.build_scene("final_scene")
    h = header(blob)
    assert h["magic"] == 0x31425452 and h["total"] == len(blob) and len(blob) % 256 == 0
    s = h["sections"]
    # The world list holds 11 entries (main.rs:576-688). One is tested by every lane in lock step: the r=5000 medium that
    # contains the whole scene. The BvhNode of 400 boxes is dissolved, and so is the instanced cluster - Translate over
    # RotateY over a BvhNode of 1000 spheres becomes 1000 spheres that share one chain record (the two steps by value)
    # and one pair of instance records (for the way back out), each named together with the chain by its reference and
    # bounded in the world's frame: one tree of 400 + 8 + 1000 leaves = 1407 nodes, no frame change while walking it.
    # The moving sphere sits in the tree under the bound of its own time interval [0, 1] and, for rays of other times,
    # on the second lock-step list as well.
    assert s["nodes"][1] == 1407
    assert s["boxes"][1] == 400 and s["mspheres"][1] == 1 and s["rects"][1] == 1
    assert s["spheres"][1] == 1000 + 5 + 2            # + the two medium boundaries (one is the shared r=70 sphere again)
    assert s["media"][1] == 2 == h["n_media"] and s["insts"][1] == 2 and s["chains"][1] == 1 and h["max_chain"] == 2
    assert s["texels"][1] == 1024 * 512 * 3 and s["perlin_vec"][1] == 768 and s["perlin_perm"][1] == 768
    assert all(off % 256 == 0 for off, _ in s.values())
    assert 2 <= h["stack_need"] <= 40
    entries = np.frombuffer(blob, dtype="<u4", count=s["list_entries"][1], offset=s["list_entries"][0])
    flat_first = h["world_flat"] & 0x03FFFFFF
    flat = entries[flat_first:flat_first + 2]
    assert (h["world_flat"] >> 28) == 8 and flat[1] == 0 and flat[0] >> 28 == 6
    any_first = h["world_flat_any_time"] & 0x03FFFFFF
    anyt = entries[any_first:any_first + 3]
    assert (h["world_flat_any_time"] >> 28) == 8 and anyt[2] == 0 and sorted(int(e) >> 28 for e in anyt[:2]) == [3, 6]
    assert h["shutter"] == (0.0, 1.0)
    # nodes are numbered breadth-first from the world's root
    nodes = nodes_of(blob, h)
    assert h["world_tree"] == (1 << 28) | 0
    top = nodes[0]
    lo = np.minimum(top["lo0"], top["lo1"])
    hi = np.maximum(top["hi0"], top["hi1"])
    # everything in the tree: ground boxes x, z in [-1000, 1000), the light at y = 554, the cluster up to z < 620
    assert np.all(lo <= [-1000, 0, -1000]) and np.all(lo > [-1001, -1, -1001])
    assert np.all(hi >= [1000 - 1, 554, 1000 - 1]) and np.all(hi < [1001, 600, 1001])
    # the cluster's spheres hang under the tree as references of type 10 (aux 0 = sphere, chain 0, sphere number), each
    # with the world-space bound of its sphere: r=10, centres in [0,165)^3 rotated 15 degrees about y and moved by (-100,270,395)
    found = [(n["lo0"], n["hi0"], n["c0"]) for n in nodes if n["c0"] >> 28 == 10] + [(n["lo1"], n["hi1"], n["c1"]) for n in nodes if n["c1"] >> 28 == 10]
    assert len(found) == 1000 and not any(n["c0"] >> 28 == 7 or n["c1"] >> 28 == 7 for n in nodes)
    assert all((int(f[2]) >> 20) & 0xFF == 0 for f in found) and len({int(f[2]) & 0xFFFFF for f in found}) == 1000
    wlo = np.min([f[0] for f in found], axis=0)
    whi = np.max([f[1] for f in found], axis=0)
    assert np.all(wlo < [-100, 265, 350]) and np.all(whi > [60, 430, 540])
    assert np.all(wlo > [-170, 250, 330]) and np.all(whi < [130, 450, 620])
    assert all(np.all(np.abs((f[1] - f[0]) - 20.0) < 0.01) for f in found)
    INST = np.dtype([("a", "<f8"), ("b", "<f8"), ("c", "<f8"), ("op", "<u4"), ("child", "<u4"), ("parent", "<i4"), ("src", "<u4")])
    insts = np.frombuffer(blob, dtype=INST, count=s["insts"][1], offset=s["insts"][0])
    assert insts["op"].tolist() == [0x100, 0x101] and insts["parent"].tolist() == [-1, 0]
    meta = np.frombuffer(blob, dtype=np.dtype([("material", "<u4"), ("parent", "<i4"), ("src", "<u4"), ("pad", "<u4")]), count=s["sphere_meta"][1], offset=s["sphere_meta"][0])
    assert sorted(int(meta["parent"][int(f[2]) & 0xFFFFF]) for f in found) == [1] * 1000
    CHAIN = np.dtype([("n", "<u4"), ("pad", "<u4"), ("op", np.dtype([("a", "<f8"), ("b", "<f8"), ("c", "<f8"), ("op", "<u4"), ("pad", "<u4")]), 4)])
    assert CHAIN.itemsize == 136
    chain = np.frombuffer(blob, dtype=CHAIN, count=1, offset=s["chains"][0])[0]
    assert chain["n"] == 2 and chain["op"]["op"].tolist() == [0, 1, 0, 0]
    assert chain["op"][0][["a", "b", "c"]].tolist() == (-100.0, 270.0, 395.0) == insts[0][["a", "b", "c"]].tolist()
    assert chain["op"][1]["a"] == insts[1]["a"] == np.sin(np.radians(15.0)) or abs(chain["op"][1]["a"] - np.sin(np.radians(15.0))) < 1e-15


def test_a_small_reference_order_tree_becomes_lock_step_steps():
    """The Cornell box is eight objects under a BvhNode whose boxes do not all bound: its depth-first walk (bvh.rs:112-122)
    goes onto the lock-step list as steps - a box test per BvhNode, in front of the root's too, and one in front of the
    world's entry - and the scene has no tree and no node."""
    b, info, blob, _ = util.build_scene("cornell_box")
    h = header(blob)
    s = h["sections"]
    assert s["nodes"][1] == 0 and h["world_tree"] == 0 and (h["world_flat"] >> 28) == 8 and h["stack_need"] == 1
    entries = np.frombuffer(blob, dtype="<u4", count=s["list_entries"][1], offset=s["list_entries"][0])
    at = h["world_flat"] & 0x03FFFFFF
    assert entries[at] >> 28 == 11
    n = int(entries[at] & 0x03FFFFFF)
    ctrl, operand = entries[at + 1:at + 1 + 2 * n:2], entries[at + 2:at + 2 + 2 * n:2]
    assert entries[at + 1 + 2 * n] == 0                       # the list ends behind the steps
    gates = (ctrl & 0x100) != 0
    depth = ctrl & 63
    assert gates.sum() == 7 + 1 and (~gates).sum() == 8       # 7 BvhNodes (bvh.rs:33-50) + the world's entry; 8 objects
    assert sorted((operand[~gates] >> 28).tolist()) == [4] * 6 + [10] * 2   # six rects, two boxes named with their chains
    assert sorted(operand[gates].tolist()) == list(range(1, 9)) and s["exact_boxes"][1] == 8
    # depth-first: a step is at most one level below the step before it, the walk starts at depth 1, and what follows a
    # box test one level down is inside it
    assert depth[0] == 1 and gates[0] and np.all(np.diff(depth.astype(int)) <= 1) and depth.max() <= 31
    assert all(depth[i + 1] == depth[i] + 1 for i in range(n - 1) if gates[i])


def test_a_world_of_a_few_plain_objects_goes_onto_the_list_whole():
    """cornell_smoke: eight objects in a plain list (five walls, the light, two media on instanced boxes). No tree: every
    object a step at depth 1, in the world's own order; a moving sphere or builder flag 8 (a tree for the device to build)
    keeps the tree."""
    b, info, blob, _ = util.build_scene("cornell_smoke")
    h = header(blob)
    s = h["sections"]
    assert s["nodes"][1] == 0 and h["world_tree"] == 0 and (h["world_flat"] >> 28) == 8 and h["world_flat"] == h["world_flat_any_time"]
    entries = np.frombuffer(blob, dtype="<u4", count=s["list_entries"][1], offset=s["list_entries"][0])
    at = h["world_flat"] & 0x03FFFFFF
    n = int(entries[at] & 0x03FFFFFF)
    assert entries[at] >> 28 == 11 and n == 8 and entries[at + 1 + 2 * n] == 0
    ctrl, operand = entries[at + 1:at + 1 + 2 * n:2], entries[at + 2:at + 2 + 2 * n:2]
    assert np.all(ctrl == 1)                                   # no box test, everything at depth 1
    assert (operand >> 28).tolist() == [4, 4, 4, 4, 4, 4, 6, 6]   # main.rs:488-574: the walls and the light, then the two media
    _, _, blob8, _ = util.build_scene("cornell_smoke", flags=api.FLAG_DEVICE_TREE)
    h8 = header(blob8)
    assert h8["sections"]["nodes"][1] >= 7 and (h8["world_tree"] >> 28) == 1
    # the Book-1 scene has 480-odd objects: a tree, and its huge ground sphere on the list as a plain entry
    _, _, big, _ = util.build_scene("random_scene")
    hb = header(big)
    eb = np.frombuffer(big, dtype="<u4", count=hb["sections"]["list_entries"][1], offset=hb["sections"]["list_entries"][0])
    assert (hb["world_tree"] >> 28) == 1 and not np.any(eb >> 28 == 11)


def test_cornell_replays_the_reference_order(monkeypatch):
    monkeypatch.setenv("RT_NO_LOCKSTEP_TREE", "1")            # the form a larger tree of that kind takes
    b, info, blob, _ = util.build_scene("cornell_box")
    h = header(blob)
    # 8 objects -> 7 BvhNodes (bvh.rs:33-50); each node one record, + one record in front of the root and of every
    # right child that is a node (their own boxes, tested when the reference tests them) + the world's bounding node
    nodes = nodes_of(blob, h)
    assert len(nodes) == 7 + 1 + 3 + 1 and h["world_flat"] == 0 and (h["world_tree"] >> 28) == 1
    assert h["sections"]["rects"][1] == 6 and h["sections"]["boxes"][1] == 2
    # its RotateY instances keep unrotated boxes: no record below the world's may leave the order to the ray
    # (child 1 is absent, or is entered at t_min like nothing else can be)
    assert h["world_tree"] == (1 << 28) | 0
    for i, n in enumerate(nodes[1:]):
        assert is_empty(n["lo1"], n["hi1"]) if n["c1"] == 0 else is_infinite(n["lo1"], n["hi1"]), i
    top = nodes[0]
    assert top["c0"] == (1 << 28) | 1 and top["c1"] == 0
    root_gate = nodes[1]
    assert root_gate["c0"] == (1 << 28) | 2 and root_gate["c1"] == 0
    assert np.all(root_gate["lo0"] <= [0, 0, 0]) and np.all(root_gate["hi0"] >= [555, 555, 555]) and np.all(root_gate["hi0"] < 556)


@pytest.mark.parametrize("flags", [0, REFERENCE_TREE])
def test_tree_boxes_contain_and_hug_their_subtrees(flags):
    rng = np.random.default_rng(3)
    b = api.Builder(flags=flags)
    m = b.lambertian_color(1, 1, 1)
    c = rng.uniform(-1000, 1000, (257, 3)) + rng.uniform(0, 1e-3, (257, 3))
    r = rng.uniform(0.1, 3.0, 257)
    lst = b.list([b.sphere(tuple(ci), float(ri), m) for ci, ri in zip(c, r)])
    blob = b.compile(b.list([b.bvh(lst, 0, 1)]))
    h = header(blob)
    nodes = nodes_of(blob, h)
    so, sn = h["sections"]["spheres"]
    sph = np.frombuffer(blob, dtype="<f8", count=sn * 4, offset=so).reshape(sn, 4)
    if flags == 0:
        assert len(nodes) == 256         # the BvhNode is dissolved into the world's tree: one object per leaf
    seen = []

    def bounds(ref):
        ty, idx = ref >> 28, ref & 0x03FFFFFF
        assert (ref >> 26) & 3 == 0
        if ty == 2:
            seen.append(idx)
            return sph[idx, :3] - sph[idx, 3], sph[idx, :3] + sph[idx, 3]
        assert ty == 1
        n = nodes[idx]
        out = None
        for cref, lo32, hi32 in ((int(n["c0"]), n["lo0"], n["hi0"]), (int(n["c1"]), n["lo1"], n["hi1"])):
            if cref == 0:
                assert is_empty(lo32, hi32)
                continue
            lo, hi = bounds(cref)
            assert np.all(lo32.astype(np.float64) <= lo) and np.all(hi32.astype(np.float64) >= hi)
            # and tight: within one f32 ulp
            assert np.all(lo - lo32.astype(np.float64) <= np.spacing(np.abs(lo32)).astype(np.float64))
            assert np.all(hi32.astype(np.float64) - hi <= np.spacing(np.abs(hi32)).astype(np.float64))
            out = (lo, hi) if out is None else (np.minimum(out[0], lo), np.maximum(out[1], hi))
        return out
    if flags == 0:
        bounds(h["world_tree"])
    else:
        top = nodes[h["world_tree"] & 0x03FFFFFF]   # a BvhNode kept whole is a leaf of the world's tree: its bound has a margin
        lo, hi = bounds(int(top["c0"]))
        assert top["c1"] == 0 and np.all(top["lo0"] < lo) and np.all(top["hi0"] > hi) and np.all(top["hi0"] - hi < 0.01)
    assert sorted(seen) == list(range(257))          # every sphere under exactly one leaf
    # breadth-first numbering from the world's root: children always come after their parent, level by level
    assert h["world_tree"] == (1 << 28) | 0
    expect = 1
    for i, n in enumerate(nodes):
        for cref in (int(n["c0"]), int(n["c1"])):
            if cref >> 28 == 1:
                assert (cref & 0x03FFFFFF) == expect
                expect += 1
    assert expect == len(nodes)


def test_description_log_replays_to_the_same_scene_twice():
    _, _, blob1, desc1 = util.build_scene("random_scene")
    _, _, blob2, desc2 = util.build_scene("random_scene")
    assert blob1 == blob2 and desc1 == desc2          # scene construction is deterministic in its seeds


def test_one_million_sphere_scene_compiles():
    b, info, blob, _ = util.build_scene("spheres_grid", n=1_000_000)
    h = header(blob)
    # the ground sphere is tested in lock step; 1,000,000 spheres, one per leaf -> 999,999 nodes
    assert h["sections"]["nodes"][1] == 999_999 and h["world_flat"] >> 28 == 8
    assert h["sections"]["spheres"][1] + h["sections"]["mspheres"][1] == 1_000_001
    assert h["stack_need"] <= 40 and info.width == 3840 and info.height == 2160


def test_ppm_writer_and_checkpoint_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    api.write_ppm(str(tmp_path / "a.ppm"), img)
    data = open(str(tmp_path / "a.ppm"), "rb").read()
    assert data == b"P6\n7 5\n255\n" + img.tobytes()
    s = rng.random((5, 7, 3))
    api.save_checkpoint(str(tmp_path / "c.npz"), s, 12, 0x5EED0002)
    got, done, seed = api.load_checkpoint(str(tmp_path / "c.npz"))
    assert got.tobytes() == s.tobytes() and done == 12 and seed == 0x5EED0002
    # the pixel conversion of main.rs:172-189: sqrt(sum / spp) clamped to 0.999, times 255.999, truncated
    assert api.tonemap(np.array([[[0.0, 4.0, 1.0]]]), 4).tolist() == [[[0, 255, 127]]]


def test_png_writer_round_trip(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    api.write_png(str(tmp_path / "a.png"), img)
    with Image.open(str(tmp_path / "a.png")) as im:
        assert im.mode == "RGB" and im.size == (13, 9)
        assert np.array_equal(np.asarray(im), img)


def test_jpeg_writer(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    y, x = np.mgrid[0:48, 0:64]
    img = np.stack([x * 4, y * 5, (x + y) * 2], axis=2).astype(np.uint8)      # smooth: JPEG keeps it within a few levels
    api.write_jpeg(str(tmp_path / "a.jpg"), img)
    with Image.open(str(tmp_path / "a.jpg")) as im:
        assert im.format == "JPEG" and im.size == (64, 48)
        assert np.abs(np.asarray(im.convert("RGB")).astype(int) - img.astype(int)).mean() < 3.0
