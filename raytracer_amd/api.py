"""Thin Python face of the C ABI: scene building, upload, first-hit queries and rendering.

Python here is plumbing (tests, bench, multi-GPU launch); everything that computes lives in librt_hip.so.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import RtError, lib, rt_camera, rt_params, rt_scene_info, rt_stats, rt_tree_stats

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
RAY_DTYPE = np.dtype([(n, "<f8") for n in ("ox", "oy", "oz", "dx", "dy", "dz", "time", "tmin", "tmax")])
HIT_DTYPE = np.dtype([("t", "<f8"), ("prim_id", "<u4"), ("flags", "<u4"), ("px", "<f8"), ("py", "<f8"), ("pz", "<f8"),
                      ("nx", "<f8"), ("ny", "<f8"), ("nz", "<f8"), ("u", "<f8"), ("v", "<f8"), ("material_id", "<u4"),
                      ("pad", "<u4")])
assert RAY_DTYPE.itemsize == C.sizeof(_lib.rt_ray_d) and HIT_DTYPE.itemsize == C.sizeof(_lib.rt_hit)

# the names the reference's scene code opens (main.rs:653, material.rs:165-188) -> decoded fixtures
TEXTURE_FILES = {
    "earthmap.jpg": "earthmap_rgb8.npz",
    "raytracer/star1.jpg": "star1_rgb8.npz",
    "raytracer/star3.jpeg": "star3_rgb8.npz",
    "raytracer/star5.jpeg": "star5_rgb8.npz",
    "raytracer/star6.jpeg": "star6_rgb8.npz",
}


# rt_builder_set_flags bits (include/rt.h)
FLAG_BOOK_BBOX, FLAG_REFERENCE_TREE, FLAG_ANY_TIME, FLAG_DEVICE_TREE = 1, 2, 4, 8


def load_texture(filename):
    """RGB8 array (H, W, 3) for one of the reference's image files."""
    return np.ascontiguousarray(np.load(os.path.join(DATA_DIR, TEXTURE_FILES[filename]))["rgb"], dtype=np.uint8)


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


class Builder:
    """rt_builder: the reference's constructors as methods returning integer handles."""

    def __init__(self, seed=0x5EED0001, flags=0):
        self._b = lib.rt_builder_new(seed)
        if not self._b:
            raise MemoryError("rt_builder_new")
        if flags:
            self._chk(lib.rt_builder_set_flags(self._b, flags))

    def close(self):
        if self._b:
            lib.rt_builder_free(self._b)
            self._b = None

    __del__ = close

    def _chk(self, v):
        if v < 0:
            raise RtError(v, lib.rt_builder_last_error(self._b).decode())
        return v

    # textures
    def solid(self, r, g, b): return self._chk(lib.rt_tex_solid(self._b, r, g, b))
    def checker(self, even, odd): return self._chk(lib.rt_tex_checker(self._b, even, odd))
    def noise(self, scale): return self._chk(lib.rt_tex_noise(self._b, scale))

    def image(self, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        return self._chk(lib.rt_tex_image(self._b, rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))

    # materials
    def lambertian(self, tex): return self._chk(lib.rt_mat_lambertian(self._b, tex))
    def lambertian_color(self, r, g, b): return self.lambertian(self.solid(r, g, b))
    def metal(self, r, g, b, fuzz): return self._chk(lib.rt_mat_metal(self._b, r, g, b, fuzz))
    def dielectric(self, ir): return self._chk(lib.rt_mat_dielectric(self._b, ir))
    def diffuse_light(self, tex, k=1.0): return self._chk(lib.rt_mat_diffuse_light(self._b, tex, k))
    def isotropic(self, tex): return self._chk(lib.rt_mat_isotropic(self._b, tex))

    # hittables
    def sphere(self, c, r, mat): return self._chk(lib.rt_sphere(self._b, _d3(c), r, mat))
    def moving_sphere(self, c0, c1, t0, t1, r, mat): return self._chk(lib.rt_moving_sphere(self._b, _d3(c0), _d3(c1), t0, t1, r, mat))
    def xy_rect(self, x0, x1, y0, y1, k, mat): return self._chk(lib.rt_xy_rect(self._b, x0, x1, y0, y1, k, mat))
    def xz_rect(self, x0, x1, z0, z1, k, mat): return self._chk(lib.rt_xz_rect(self._b, x0, x1, z0, z1, k, mat))
    def yz_rect(self, y0, y1, z0, z1, k, mat): return self._chk(lib.rt_yz_rect(self._b, y0, y1, z0, z1, k, mat))
    def box(self, p0, p1, mat): return self._chk(lib.rt_box(self._b, _d3(p0), _d3(p1), mat))
    def translate(self, child, off): return self._chk(lib.rt_translate(self._b, child, _d3(off)))
    def rotate_y(self, child, deg): return self._chk(lib.rt_rotate_y(self._b, child, deg))
    def constant_medium(self, boundary, density, r, g, b): return self._chk(lib.rt_constant_medium(self._b, boundary, density, r, g, b))

    def list(self, members=()):
        h = self._chk(lib.rt_list_new(self._b))
        for m in members:
            self.list_add(h, m)
        return h

    def list_add(self, lst, child): self._chk(lib.rt_list_add(self._b, lst, child))
    def bvh(self, lst, t0, t1): return self._chk(lib.rt_bvh(self._b, lst, t0, t1))

    def register_image(self, filename, rgb):
        rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
        self._chk(lib.rt_builder_register_image(self._b, filename.encode(), rgb.ctypes.data, rgb.shape[1], rgb.shape[0]))

    def register_reference_textures(self, names=("earthmap.jpg",)):
        for n in names:
            self.register_image(n, load_texture(n))

    def named_scene(self, name, scene_seed=0x5EED0001, n=0):
        info = rt_scene_info()
        self._chk(lib.rt_build_named_scene(self._b, name.encode(), scene_seed, n, C.byref(info)))
        return info

    def compile(self, world):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(lib.rt_builder_compile(self._b, world, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            lib.rt_free(p)

    def describe(self, world):
        p, n = C.c_void_p(), C.c_size_t()
        self._chk(lib.rt_builder_describe(self._b, world, C.byref(p), C.byref(n)))
        try:
            return C.string_at(p, n.value)
        finally:
            lib.rt_free(p)


def make_camera(lookfrom, lookat, vup, vfov, aspect, aperture, focus, time0=0.0, time1=1.0):
    c = rt_camera()
    c.from_[:] = lookfrom
    c.at[:] = lookat
    c.vup[:] = vup
    c.vfov, c.aspect, c.aperture, c.focus, c.time0, c.time1 = vfov, aspect, aperture, focus, time0, time1
    return c


def make_params(width, height, max_depth=50, background=(0.0, 0.0, 0.0), background_mode=0, seed=0x5EED0002, compat=0,
                tile=0, tile_rank=0, tile_world=1, spp_per_pass=0, profile=0):
    p = rt_params()
    p.width, p.height, p.max_depth, p.background_mode = width, height, max_depth, background_mode
    p.background[:] = background
    p.seed, p.compat, p.tile, p.tile_rank, p.tile_world = seed, compat, tile, tile_rank, tile_world
    p.spp_per_pass, p.profile = spp_per_pass, profile
    return p


def params_from_info(info, **kw):
    base = dict(width=info.width, height=info.height, max_depth=info.max_depth, background=tuple(info.background),
                background_mode=info.background_mode)
    base.update(kw)
    return make_params(**base)


class Scene:
    def __init__(self, ctx, handle, nbytes):
        self.ctx, self._h, self.nbytes = ctx, handle, nbytes

    def close(self):
        if self._h and self.ctx._c:
            lib.rt_scene_free(self.ctx._c, self._h)
        self._h = None


class Context:
    """rt_context on one HIP device."""

    def __init__(self, device=0):
        h = C.c_void_p()
        rc = lib.rt_init(C.byref(h), device)
        if rc != 0:
            raise RtError(rc, "rt_init failed (no usable HIP device?)")
        self._c = h

    def close(self):
        if self._c:
            lib.rt_shutdown(self._c)
            self._c = None

    def _chk(self, rc):
        if rc != 0:
            raise RtError(rc, lib.rt_last_error(self._c).decode())

    def upload(self, blob):
        h = C.c_void_p()
        self._chk(lib.rt_scene_upload(self._c, blob, len(blob), C.byref(h)))
        return Scene(self, h, len(blob))

    def upload_from_device(self, d_ptr, nbytes):
        h = C.c_void_p()
        self._chk(lib.rt_scene_upload_from_device(self._c, d_ptr, nbytes, C.byref(h)))
        return Scene(self, h, nbytes)

    def build_tree(self, scene):
        """rt_scene_build_tree_device: arranges the world's tree of a scene compiled with FLAG_DEVICE_TREE on the GPU."""
        st = rt_tree_stats()
        self._chk(lib.rt_scene_build_tree_device(self._c, scene._h, C.byref(st)))
        return st

    def trace_first_hit(self, scene, rays, medium_xi=None, noalloc=False, wavefront=None):
        """noalloc=True: through rt_trace_first_hit_noalloc (fixed work area, limited batch, no device allocation).
        wavefront=v (0..63): through rt_trace_first_hit_wavefront, i.e. the renderer's traversal kernel, variant v."""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        xi = None
        if medium_xi is not None:
            xi = np.ascontiguousarray(medium_xi, dtype=np.float64)
        args = (self._c, scene._h, rays.ctypes.data, len(rays), xi.ctypes.data if xi is not None else None, out.ctypes.data)
        if wavefront is not None:
            self._chk(lib.rt_trace_first_hit_wavefront(*args, int(wavefront)))
        else:
            self._chk((lib.rt_trace_first_hit_noalloc if noalloc else lib.rt_trace_first_hit)(*args))
        return out

    def generate_rays(self, camera, params, sample):
        out = np.zeros(params.width * params.height, dtype=RAY_DTYPE)
        self._chk(lib.rt_generate_rays(self._c, C.byref(camera), C.byref(params), sample, out.ctypes.data))
        return out

    def render(self, scene, camera, params, spp, want_rgb8=False, stats=None):
        n = params.width * params.height
        rgb_sum = np.zeros((params.height, params.width, 3), dtype=np.float64)
        rgb8 = np.zeros((params.height, params.width, 3), dtype=np.uint8) if want_rgb8 else None
        self._chk(lib.rt_render(self._c, scene._h, C.byref(camera), C.byref(params), spp, rgb_sum.ctypes.data,
                                rgb8.ctypes.data if want_rgb8 else None, C.byref(stats) if stats is not None else None))
        assert rgb_sum.size == n * 3
        return (rgb_sum, rgb8) if want_rgb8 else rgb_sum

    @staticmethod
    def render_multi(contexts, blob, camera, params, spp, want_rgb8=False, stats=None):
        """rt_render_multi: the compiled scene `blob` rendered by several contexts at once, tiles dealt round-robin, scene
        copied and image gathered GPU to GPU inside the library. `stats`: a list that receives one rt_stats per context."""
        rgb_sum = np.zeros((params.height, params.width, 3), dtype=np.float64)
        rgb8 = np.zeros((params.height, params.width, 3), dtype=np.uint8) if want_rgb8 else None
        handles = (C.c_void_p * len(contexts))(*[c._c for c in contexts])
        st = (rt_stats * len(contexts))()
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        contexts[0]._chk(lib.rt_render_multi(handles, len(contexts), buf, len(blob), C.byref(camera), C.byref(params), spp,
                                             rgb_sum.ctypes.data, rgb8.ctypes.data if want_rgb8 else None, st))
        if stats is not None:
            stats.extend(st)
        return (rgb_sum, rgb8) if want_rgb8 else rgb_sum

    def render_more(self, scene, camera, params, rgb_sum, samples_done, spp, stats=None):
        """Continues a render: adds samples [samples_done, samples_done + spp) of every pixel to the host array
        rgb_sum (height, width, 3) and returns it. Random numbers are keyed on (pixel, sample, bounce), so the result is
        bit-identical to having rendered samples_done + spp in one call: a saved sum is a checkpoint."""
        rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float64)
        assert rgb_sum.shape == (params.height, params.width, 3)
        d = C.c_void_p()
        self._chk(lib.rt_device_alloc(self._c, rgb_sum.nbytes, C.byref(d)))
        try:
            self._chk(lib.rt_device_upload(self._c, d, rgb_sum.ctypes.data, rgb_sum.nbytes))
            self.render_accumulate(scene, camera, params, samples_done, spp, d.value, stats)
            out = np.empty_like(rgb_sum)
            self._chk(lib.rt_device_download(self._c, out.ctypes.data, d, out.nbytes))
        finally:
            lib.rt_device_free(self._c, d)
        return out

    def render_accumulate(self, scene, camera, params, sample_offset, spp, d_rgb_sum, stats=None):
        """d_rgb_sum: integer device address of width*height*3 doubles (e.g. torch_tensor.data_ptr())."""
        self._chk(lib.rt_render_accumulate(self._c, scene._h, C.byref(camera), C.byref(params), sample_offset, spp,
                                           C.c_void_p(d_rgb_sum), C.byref(stats) if stats is not None else None))


def tonemap(rgb_sum, spp):
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    out = np.zeros(rgb_sum.shape, dtype=np.uint8)
    lib.rt_tonemap(rgb_sum.ctypes.data, rgb_sum.size // 3, float(spp), out.ctypes.data)
    return out


def write_ppm(path, rgb8):
    """Binary PPM of an (height, width, 3) u8 image, rows in the order the reference stores them (main.rs:170,244-251:
    image row y = the v = y / (height - 1) scanline, no flip)."""
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    assert rgb8.ndim == 3 and rgb8.shape[2] == 3
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb8.shape[1], rgb8.shape[0]))
        f.write(rgb8.tobytes())


def write_png(path, rgb8):
    """8-bit RGB PNG (lossless stand-in for the reference's `img.save("output/test.jpg")`, main.rs:255), same row order as
    write_ppm. One zlib stream, filter type 0 on every row."""
    import struct
    import zlib
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    assert rgb8.ndim == 3 and rgb8.shape[2] == 3
    h, w = rgb8.shape[:2]

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    rows = np.concatenate([np.zeros((h, 1), dtype=np.uint8), rgb8.reshape(h, w * 3)], axis=1)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(rows.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def write_jpeg(path, rgb8, quality=75):
    """`img.save("output/test.jpg")` (main.rs:255) through Pillow's baseline encoder at the `image` crate's default
    quality. The bytes are not those of the crate's own encoder (different Huffman tables and rounding; nothing in the
    reference pins them); the decoded picture agrees to JPEG precision. Pillow is the one optional dependency here."""
    from PIL import Image
    rgb8 = np.ascontiguousarray(rgb8, dtype=np.uint8)
    assert rgb8.ndim == 3 and rgb8.shape[2] == 3
    Image.fromarray(rgb8, "RGB").save(path, format="JPEG", quality=int(quality))


def save_checkpoint(path, rgb_sum, samples_done, seed):
    """The linear radiance sums and how many samples per pixel they hold: enough to resume with Context.render_more."""
    np.savez(path, rgb_sum=np.asarray(rgb_sum, dtype=np.float64), samples_done=np.uint32(samples_done), seed=np.uint64(seed))


def load_checkpoint(path):
    with np.load(path, allow_pickle=False) as z:
        return z["rgb_sum"].copy(), int(z["samples_done"]), int(z["seed"])
