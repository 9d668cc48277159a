"""The oracle against the kernel and the sizes that are benchmarked. Needs an MI355X.

test_gpu_parity.py puts gate 1 on rt_trace_first_hit (one lane, one ray, nodes from global memory) and compares renders
at sizes the oracle can render whole. What the benchmark times is something else: k_extend_tree - persistent waves, one
1024-lane workgroup on each of the 256 CUs, the top of the tree in LDS, the queue dealt in pieces of up to 4096 rays,
leaf rounds - on passes of millions of paths. Here
  * the first-hit gate runs through that kernel (rt_trace_first_hit_wavefront), on a million rays per scene so that every
    CU has a workgroup and the pieces are larger than the minimum, in the instantiation the scene renders with and in
    the one with exact box tests and the general medium code;
  * every configuration of the benchmark set is rendered at its full image size in one pass, and the per-pixel sums of a
    few thousand random pixels are compared with the oracle's ray_color of the same (pixel, sample) seeds;
  * the 1 M-sphere scene - nodes out of cache, the deepest stacks - gets both, on the host-built and the device-built tree."""
import numpy as np
import pytest

import checkers as ck
import util
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def big_ray_suite(ctx, osc, info, rng, width=640, height=480, n_secondary=150000):
    """>= 1 M rays: two camera samples per pixel, rays leaving the surfaces those hit, axis-parallel and short-range rays."""
    p = api.params_from_info(info, width=width, height=height, seed=util.RENDER_SEED)
    cam = np.concatenate([ctx.generate_rays(info.camera, p, s) for s in range(2)])
    xi0 = rng.random((len(cam), osc.n_media)) if osc.n_media else None
    rays = util.ray_suite(cam, osc.trace_first_hit(cam, xi0), rng, n_secondary=n_secondary)
    return rays, (rng.random((len(rays), osc.n_media)) if osc.n_media else None)


def same_hits_except_ties(a, b):
    """Two GPU answers to the same rays: every byte equal, except where two primitives lie at the same t to the bit."""
    differ = a["prim_id"] != b["prim_id"]
    assert np.array_equal(a["t"][differ].view(np.uint64), b["t"][differ].view(np.uint64))
    assert differ.mean() <= 0.02
    same = ~differ
    return a[same].tobytes() == b[same].tobytes()


@pytest.mark.parametrize("name", list(util.SCENES))
def test_gate1_through_the_benchmarked_kernel(gpu_ctx, name):
    _, info, blob, desc = util.build_scene(name)
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    scene = gpu_ctx.upload(blob)
    rng = np.random.default_rng(29)
    rays, xi = big_ray_suite(gpu_ctx, osc, info, rng)
    assert len(rays) >= 1 << 20                                    # 256 workgroups, pieces of 4096 rays
    want = osc.trace_first_hit(rays, xi)
    lane = gpu_ctx.trace_first_hit(scene, rays, xi)                 # the one-lane-one-ray kernel
    for variant in (0, 3):
        got = gpu_ctx.trace_first_hit(scene, rays, xi, wavefront=variant)
        rep = util.first_hit_report(got, want)
        print("\n%s variant %d: %d rays, prim match %.6f (ties %.5f), t bit-equal %.6f, t max rel err %.3e" %
              (name, variant, len(rays), rep["prim_match"], rep["ties"], rep["t_bit_equal"], rep["t_rel_max"]))
        assert rep["prim_match"] >= 0.9999 and rep["t_rel_max"] <= 1e-5 and rep["t_bit_equal"] >= 0.98, rep
        hit = rep["both_hit"]
        for f in ("px", "py", "pz", "nx", "ny", "nz"):
            assert util.rel_err(got[f][hit], want[f][hit]).max() <= 1e-9, f
        assert np.array_equal(got["flags"][hit], want["flags"][hit])
        assert np.array_equal(got["material_id"][hit], want["material_id"][hit])
        assert same_hits_except_ties(got, lane)
    if osc.n_media:                                                 # media switched off: NULL uniforms
        assert same_hits_except_ties(gpu_ctx.trace_first_hit(scene, rays[:300000], None, wavefront=0),
                                     gpu_ctx.trace_first_hit(scene, rays[:300000], None))
    scene.close()


def test_wavefront_first_hit_edge_cases(gpu_ctx):
    _, info, blob, desc = util.build_scene("cornell_smoke")
    scene = gpu_ctx.upload(blob)
    assert len(gpu_ctx.trace_first_hit(scene, np.zeros(0, dtype=api.RAY_DTYPE), wavefront=0)) == 0
    rays = util.rays_from([(278, 278, -800), (278, 278, 100), (278, 278, 100), (278, 278, 100)],
                          [(0, 0, 1), (0, 0, 0), (np.nan, 0, 1), (0, 1e-300, 0)])   # zero and NaN directions: no hang, no fault
    got = gpu_ctx.trace_first_hit(scene, rays, None, wavefront=0)
    want = gpu_ctx.trace_first_hit(scene, rays, None)
    assert np.array_equal(got["prim_id"][[0, 3]], want["prim_id"][[0, 3]])
    for n in (1, 63, 64, 65, 1025):                                                 # ragged waves and workgroups
        r = np.tile(util.rays_from([(278, 278, -800)], [(0.1, 0.05, 1)]), n)
        xi = np.tile(np.array([[0.3, 0.7]]), (n, 1))
        assert gpu_ctx.trace_first_hit(scene, r, xi, wavefront=3).tobytes() == gpu_ctx.trace_first_hit(scene, r, xi).tobytes()
    with pytest.raises(api.RtError):
        gpu_ctx.trace_first_hit(scene, rays, None, wavefront=64)
    scene.close()


def sampled_sums(osc, camera, params, xs, ys, spp):
    """The oracle's pixel_color after `spp` samples (main.rs:161-167) for the chosen pixels only."""
    out = np.zeros((len(xs), 3))
    for k, (x, y) in enumerate(zip(xs, ys)):
        acc = np.zeros(3)
        for s in range(spp):
            acc += osc.sample_radiance(camera, params, int(x), int(y), s)
        out[k] = acc
    return out


def check_sampled_render(ctx, scene, osc, info, spp, n_pixels, label, rng):
    p = api.params_from_info(info, seed=util.RENDER_SEED, spp_per_pass=spp)
    st = api.rt_stats()
    img = ctx.render(scene, info.camera, p, spp, stats=st)
    W, H = info.width, info.height
    assert st.passes == 1 and st.camera_samples == W * H * spp and W * H * spp >= 256 * 1024 * 8   # pieces above the minimum
    assert np.isfinite(img).all() and (img >= 0).all()
    xs, ys = rng.integers(0, W, n_pixels), rng.integers(0, H, n_pixels)
    want = sampled_sums(osc, info.camera, p, xs, ys, spp)
    got = img[ys, xs]
    err = util.rel_err(got, want)
    off = float((err > 1e-9).mean())
    print("\n%s %dx%d, %d spp in one pass (%.1f M paths, %.2f rays per sample): %d pixels against the oracle, bit-equal "
          "entries %.4f, off by > 1e-9: %.5f, mean ratio %.9f" % (label, W, H, spp, W * H * spp / 1e6, st.ray_segments / st.camera_samples,
                                                                n_pixels, float((got == want).mean()), off, got.mean() / want.mean()))
    assert off <= 0.003                                             # identical seeds -> identical paths (a medium's log() may differ by an ulp)
    assert abs(got.mean() / want.mean() - 1.0) <= 2e-3
    return img


@pytest.mark.parametrize("name,spp", [("random_scene", 4), ("cornell_box", 8), ("final_scene", 6), ("stress_volume", 2), ("my_scene", 4)])
def test_full_size_render_against_sampled_oracle(gpu_ctx, name, spp):
    """C1 1200x800, C2 600x600, C3 800x800, C5 1920x1080 and the reference's live scene, each at its own image size."""
    _, info, blob, desc = util.build_scene(name)
    assert (info.width, info.height) == {"random_scene": (1200, 800), "cornell_box": (600, 600), "final_scene": (800, 800),
                                         "stress_volume": (1920, 1080), "my_scene": (800, 800)}[name]
    osc = ck.OracleScene(desc, share_bvh_axes=True)
    scene = gpu_ctx.upload(blob)
    check_sampled_render(gpu_ctx, scene, osc, info, spp, 3000, name, np.random.default_rng(31))
    scene.close()


def test_one_million_spheres_at_full_size(gpu_ctx):
    """C4: 1 000 000 spheres (64 MB of nodes: out of every cache, the tallest trees, the deepest stacks), 3840x2160,
    on the host-built tree and on the device-built one: first hits on > 3e5 rays through both kernels, a full-size pass
    against the oracle's samples, and the two trees against each other bit for bit."""
    n = 1000000
    _, info, blob, desc = util.build_scene("spheres_grid", n=n)
    assert (info.width, info.height) == (3840, 2160)
    osc = ck.OracleScene(desc)                                      # its own tree, the reference's build (bvh.rs:15-66)
    rng = np.random.default_rng(37)
    rays, _ = big_ray_suite(gpu_ctx, osc, info, rng, width=480, height=270, n_secondary=60000)
    assert len(rays) >= 300000
    want = osc.trace_first_hit(rays)
    images = []
    for flags in (0, api.FLAG_DEVICE_TREE):
        if flags:
            _, info, blob, _ = util.build_scene("spheres_grid", n=n, flags=flags)
        scene = gpu_ctx.upload(blob)
        if flags:
            ts = gpu_ctx.build_tree(scene)
            assert ts.leaves >= n and ts.height >= 20
        for got in (gpu_ctx.trace_first_hit(scene, rays), gpu_ctx.trace_first_hit(scene, rays, wavefront=0),
                    gpu_ctx.trace_first_hit(scene, rays, wavefront=3)):
            rep = util.first_hit_report(got, want)
            assert rep["prim_match"] >= 0.9999 and rep["t_rel_max"] <= 1e-5 and rep["t_bit_equal"] >= 0.98, (flags, rep)
        images.append(check_sampled_render(gpu_ctx, scene, osc, info, 2, 2500, "spheres_grid 1M, builder flags %d," % flags,
                                           np.random.default_rng(41)))
        scene.close()
    # two different trees over the same spheres: the same image, except where a path met an exact tie
    assert float((images[0] != images[1]).mean()) <= 1e-4
