"""Builder flag bit 3: the world's tree is left to the device. What the compiler emits then - a balanced stand-in tree
over the same leaves - must already be a correct tree (same closest hits as the surface-area-heuristic tree), and the
header must say where it is. The device build itself is covered by tests/test_gpu_tree_build.py."""
import struct

import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

TAIL = struct.Struct("<IIIII")   # device_tree_root, _leaves, _leaf_need, _need_other, _need_add
TAIL_OFFSET = 400                # behind the 19 section records of RtBlobHeader
NODE = 1


def tail(blob):
    return TAIL.unpack_from(blob, TAIL_OFFSET)


def walk(blob, root_index, leaves):
    """Leaves of the stand-in below node `root_index`, found by halving [0, leaves) as the device build does."""
    nodes_off, nodes_cnt = struct.unpack_from("<QQ", blob, 96)
    out = []

    def go(node, s, e):
        assert node < nodes_cnt
        rec = struct.unpack_from("<12f4I", blob, nodes_off + 64 * node)
        m = s + (e - s) // 2
        for c, (a, b) in ((rec[12], (s, m)), (rec[13], (m, e))):
            if b - a == 1:
                assert c >> 28 != 0
                out.append(c)
            else:
                assert c >> 28 == NODE
                go(c & 0x03FFFFFF, a, b)
    go(root_index, 0, leaves)
    return out


@pytest.mark.parametrize("name,n", [("final_scene", None), ("spheres_grid", 3000), ("random_scene", None), ("my_scene", None)])
def test_standin_tree_is_a_correct_tree(name, n):
    _, info, blob_sah, _ = util.build_scene(name, n=n)
    _, _, blob, desc = util.build_scene(name, n=n, flags=api.FLAG_DEVICE_TREE)
    root, leaves, leaf_need, need_other, need_add = tail(blob)
    assert tail(blob_sah) == (0, 0, 0, 0, 0)
    assert root >> 28 == NODE and leaves >= 2
    found = walk(blob, root & 0x03FFFFFF, leaves)
    assert len(found) == leaves and len(set(found)) == leaves
    p = api.params_from_info(info, width=96, height=64, seed=util.RENDER_SEED)
    cam = ck.hostsim_generate_rays(info.camera, p, 0)
    rng = np.random.default_rng(5)
    n_media = desc.count(b"\nconstant_medium ")
    first, _ = ck.hostsim_trace_first_hit(blob_sah, cam, rng.random((len(cam), n_media)) if n_media else None)
    rays = util.ray_suite(cam, first, rng, n_secondary=3000)
    xi = rng.random((len(rays), n_media)) if n_media else None
    a, high = ck.hostsim_trace_first_hit(blob, rays, xi)
    b, _ = ck.hostsim_trace_first_hit(blob_sah, rays, xi)
    assert high <= struct.unpack_from("<I", blob, 32)[0]     # the stack the header asks for is enough
    rep = util.first_hit_report(a, b)
    assert rep["prim_match"] == 1.0 and rep["t_bit_equal"] == 1.0, rep


def test_flag_is_ignored_where_the_tree_must_keep_the_reference_order():
    _, _, blob, _ = util.build_scene("cornell_box", flags=api.FLAG_DEVICE_TREE)     # RotateY under a BvhNode: replay trees
    assert tail(blob)[:2] == (0, 0)
    _, _, blob, _ = util.build_scene("final_scene", flags=api.FLAG_DEVICE_TREE | api.FLAG_REFERENCE_TREE)
    assert tail(blob)[:2] == (0, 0)


@pytest.mark.parametrize("name", ["final_scene", "random_scene", "stress_volume"])
def test_the_flag_does_not_change_the_scene(name):
    """rt_bvh skips the reference's build under the flag but still makes its axis draws (bvh.rs:22): Perlin tables and
    whatever else the scene code draws afterwards come out the same, and so does the constructor log."""
    _, _, _, d0 = util.build_scene(name)
    _, _, _, d1 = util.build_scene(name, flags=api.FLAG_DEVICE_TREE)
    assert d0.replace(b"flags 0\n", b"") == d1.replace(b"flags 8\n", b"")
