"""rt_trace_first_hit_noalloc (the entry point for callers that cannot allocate device memory) against
rt_trace_first_hit, through the C ABI: same rays, same uniforms, every byte of every hit record equal. Two calls in a
row share the library's fixed work area, the second with a smaller batch, so leftovers of the first would show."""
import numpy as np
import pytest

import util
from raytracer_amd import api

pytestmark = pytest.mark.gpu


def test_noalloc_first_hit_equals_first_hit(gpu_ctx):
    _, info, blob, _ = util.build_scene("final_scene")
    scene = gpu_ctx.upload(blob)
    n_media = 2
    p = api.params_from_info(info, width=160, height=120, seed=util.RENDER_SEED)
    rng = np.random.default_rng(5)
    cam = gpu_ctx.generate_rays(info.camera, p, 0)
    first = gpu_ctx.trace_first_hit(scene, cam, rng.random((len(cam), n_media)))
    rays = util.ray_suite(cam, first, rng, n_secondary=10000)
    assert 40000 <= len(rays) <= 80000
    xi = rng.random((len(rays), n_media))
    want = gpu_ctx.trace_first_hit(scene, rays, xi)
    assert (want["prim_id"] != 0xFFFFFFFF).mean() > 0.5
    got = gpu_ctx.trace_first_hit(scene, rays, xi, noalloc=True)
    assert got.tobytes() == want.tobytes()
    # a smaller batch right behind it, other rays, in the same work area
    sub = rng.choice(len(rays), 777, replace=False)
    got2 = gpu_ctx.trace_first_hit(scene, rays[sub], xi[sub], noalloc=True)
    assert got2.tobytes() == want[sub].tobytes()
    # without uniforms (media never hit), one ray, and nothing at all
    want3 = gpu_ctx.trace_first_hit(scene, rays[:3001], None)
    assert gpu_ctx.trace_first_hit(scene, rays[:3001], None, noalloc=True).tobytes() == want3.tobytes()
    assert gpu_ctx.trace_first_hit(scene, rays[:1], xi[:1], noalloc=True).tobytes() == want[:1].tobytes()
    assert len(gpu_ctx.trace_first_hit(scene, rays[:0], None, noalloc=True)) == 0
    # beyond the work area: refused with the limit in the message, and the next call still works
    big = np.tile(rays, 3)[:120000]
    with pytest.raises(api.RtError) as e:
        gpu_ctx.trace_first_hit(scene, big, None, noalloc=True)
    assert e.value.code == -1 and "exceed" in str(e.value)
    assert gpu_ctx.trace_first_hit(scene, rays[sub], xi[sub], noalloc=True).tobytes() == want[sub].tobytes()
    scene.close()
