"""rt_render_multi: the partition + gather of the reference's main() (main.rs:135-152, 192, 242-253) behind the C ABI.
On the one GPU a test box has, several contexts share device 0: the scene is copied device to device, every context
renders its tiles on a host thread of its own, the buffers are gathered on the first context's device - and the image must
be the one-context image bit for bit, whatever the number of contexts. Needs an MI355X."""
import hashlib

import numpy as np
import pytest

from raytracer_amd import api, _lib

import checkers as ck

pytestmark = pytest.mark.gpu


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize("name,width,height,spp", [("final_scene", 96, 80, 6), ("cornell_box", 70, 70, 5), ("my_scene", 64, 64, 4)])
def test_contexts_sharing_one_gpu_give_the_one_context_image(gpu_ctx, name, width, height, spp):
    assert _lib.lib.rt_device_count() >= 1
    b = api.Builder()
    b.register_reference_textures(tuple(api.TEXTURE_FILES))
    info = b.named_scene(name)
    blob = b.compile(info.world)
    p = api.params_from_info(info, width=width, height=height, spp_per_pass=3)
    scene = gpu_ctx.upload(blob)
    one, one8 = gpu_ctx.render(scene, info.camera, p, spp, want_rgb8=True)
    scene.close()
    others = [api.Context(0) for _ in range(2)]
    try:
        for n in (1, 2, 3):
            stats = []
            got, got8 = api.Context.render_multi([gpu_ctx] + others[:n - 1], blob, info.camera, p, spp, want_rgb8=True, stats=stats)
            assert _sha(got) == _sha(one), (name, n)
            assert np.array_equal(got8, one8)
            assert len(stats) == n and sum(s.camera_samples for s in stats) == width * height * spp
            assert all(s.camera_samples > 0 for s in stats)   # every context rendered a share
    finally:
        for c in others:
            c.close()
    # and the sums are what the oracle computes for the same seeds
    want, _ = ck.OracleScene(b.describe(info.world), share_bvh_axes=True).render(info.camera, p, spp, threads=4)
    rel = np.abs(one - want) / np.maximum(np.abs(want), 1e-30)
    rel[one == want] = 0
    assert float((rel > 1e-9).mean()) <= 0.01


def test_render_multi_rejects_bad_arguments(gpu_ctx):
    b = api.Builder()
    b.register_reference_textures()
    info = b.named_scene("simple_light")
    blob = b.compile(info.world)
    p = api.params_from_info(info, width=16, height=16)
    with pytest.raises(api.RtError) as e:
        api.Context.render_multi([gpu_ctx, gpu_ctx], blob, info.camera, p, 1)   # one context twice
    assert e.value.code == -1
    with pytest.raises(api.RtError) as e:
        api.Context.render_multi([gpu_ctx], blob[:100], info.camera, p, 1)       # not a blob
    assert e.value.code == -1
