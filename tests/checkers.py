"""ctypes bindings of the two CPU checkers used by the tests: the oracle (oracle/liboracle.so, an independent
restatement of the reference) and the host simulator (tests/hostsim/libhostsim.so, the product's per-lane
device code compiled for the CPU). Test infrastructure only."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import _lib
from raytracer_amd.api import HIT_DTYPE, RAY_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (bench.py times a build of the same source made for the host it runs on, and names it here)
ORACLE_SO = os.environ.get("RT_ORACLE_SO") or os.path.join(ROOT, "oracle", "liboracle.so")
HOSTSIM_SO = os.environ.get("RT_HOSTSIM_SO") or os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
_P = C.c_void_p


class orc_stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("camera_samples", "ray_segments", "node_tests", "sphere_tests", "moving_sphere_tests",
                                          "rect_tests", "medium_tests")] + [("seconds", C.c_double), ("threads", C.c_uint32),
                                                                            ("jobs", C.c_uint32)]


def _load_oracle():
    o = C.CDLL(ORACLE_SO)
    o.orc_load.restype = _P
    o.orc_load.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64, C.c_int, C.c_char_p, C.c_size_t]
    o.orc_free.argtypes = [_P]
    o.orc_n_media.restype = C.c_uint32
    o.orc_n_media.argtypes = [_P]
    o.orc_trace_first_hit.argtypes = [_P, _P, C.c_size_t, _P, C.c_uint32, _P, _P]
    o.orc_generate_rays.argtypes = [C.POINTER(_lib.rt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _P]
    o.orc_render.argtypes = [_P, C.POINTER(_lib.rt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double), C.c_uint32,
                             C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _P, C.POINTER(orc_stats)]
    o.orc_sample_radiance.argtypes = [_P, C.POINTER(_lib.rt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double),
                                      C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    o.orc_tonemap.argtypes = [_P, C.c_size_t, C.c_double, _P]
    o.orc_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    o.orc_vec3_kat.restype = C.c_int
    return o


def _load_hostsim():
    h = C.CDLL(HOSTSIM_SO)
    h.hs_trace_first_hit.argtypes = [_P, C.c_size_t, _P, C.c_size_t, _P, C.c_uint32, _P, C.POINTER(C.c_uint32)]
    h.hs_generate_rays.argtypes = [C.POINTER(_lib.rt_camera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, _P]
    h.hs_render.argtypes = [_P, C.c_size_t, C.POINTER(_lib.rt_camera), C.POINTER(_lib.rt_params), C.c_uint32, C.c_uint32, _P,
                            C.POINTER(C.c_uint64)]
    h.hs_sample_radiance.argtypes = [_P, C.c_size_t, C.POINTER(_lib.rt_camera), C.POINTER(_lib.rt_params), C.c_uint32, C.c_uint32,
                                     C.c_uint32, C.POINTER(C.c_double)]
    h.hs_philox.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return h


oracle = _load_oracle()
_hostsim = None


def hostsim():
    global _hostsim
    if _hostsim is None:
        _hostsim = _load_hostsim()
    return _hostsim


class OracleScene:
    """The oracle's own object graph, rebuilt from the product's constructor log."""

    def __init__(self, description, bvh_seed=0xB0B, share_bvh_axes=False):
        err = C.create_string_buffer(512)
        self._s = oracle.orc_load(description, len(description), bvh_seed, int(share_bvh_axes), err, 512)
        if not self._s:
            raise RuntimeError("oracle load failed: " + err.value.decode())
        self.n_media = oracle.orc_n_media(self._s)

    def close(self):
        if self._s:
            oracle.orc_free(self._s)
            self._s = None

    __del__ = close

    def trace_first_hit(self, rays, medium_xi=None, compat=0, counts=False):
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        out = np.zeros(len(rays), dtype=HIT_DTYPE)
        xi = np.ascontiguousarray(medium_xi, dtype=np.float64) if medium_xi is not None else None
        cnt = np.zeros(8, dtype=np.uint64)
        oracle.orc_trace_first_hit(self._s, rays.ctypes.data, len(rays), xi.ctypes.data if xi is not None else None, compat,
                                   out.ctypes.data, cnt.ctypes.data)
        return (out, cnt) if counts else out

    def render(self, camera, params, spp, sample_offset=0, threads=0, jobs=0, rgb_sum=None):
        if rgb_sum is None:
            rgb_sum = np.zeros((params.height, params.width, 3), dtype=np.float64)
        st = orc_stats()
        bg = (C.c_double * 3)(*params.background)
        oracle.orc_render(self._s, C.byref(camera), params.width, params.height, params.max_depth, bg, params.background_mode,
                          params.seed, params.compat, sample_offset, spp, threads, jobs, rgb_sum.ctypes.data, C.byref(st))
        return rgb_sum, st

    def sample_radiance(self, camera, params, x, y, sample):
        rgb = (C.c_double * 3)()
        bg = (C.c_double * 3)(*params.background)
        oracle.orc_sample_radiance(self._s, C.byref(camera), params.width, params.height, params.max_depth, bg,
                                   params.background_mode, params.seed, params.compat, x, y, sample, rgb)
        return np.array(rgb[:])


def oracle_generate_rays(camera, params, sample):
    out = np.zeros(params.width * params.height, dtype=RAY_DTYPE)
    oracle.orc_generate_rays(C.byref(camera), params.width, params.height, sample, params.seed, params.compat, out.ctypes.data)
    return out


def oracle_tonemap(rgb_sum, spp):
    rgb_sum = np.ascontiguousarray(rgb_sum, dtype=np.float64)
    out = np.zeros(rgb_sum.shape, dtype=np.uint8)
    oracle.orc_tonemap(rgb_sum.ctypes.data, rgb_sum.size // 3, float(spp), out.ctypes.data)
    return out


def hostsim_trace_first_hit(blob, rays, medium_xi=None, compat=0):
    rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
    out = np.zeros(len(rays), dtype=HIT_DTYPE)
    xi = np.ascontiguousarray(medium_xi, dtype=np.float64) if medium_xi is not None else None
    high = C.c_uint32()
    rc = hostsim().hs_trace_first_hit(blob, len(blob), rays.ctypes.data, len(rays), xi.ctypes.data if xi is not None else None,
                                      compat, out.ctypes.data, C.byref(high))
    if rc != 0:
        raise RuntimeError("hostsim trace failed: %d" % rc)
    import struct
    need = struct.unpack_from("<I", blob, 32)[0]
    assert high.value <= need, "traversal used %d stack entries, the blob's header asks for %d" % (high.value, need)
    return out, high.value


def hostsim_generate_rays(camera, params, sample):
    out = np.zeros(params.width * params.height, dtype=RAY_DTYPE)
    rc = hostsim().hs_generate_rays(C.byref(camera), params.width, params.height, sample, params.seed, params.compat, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("hostsim generate_rays failed: %d" % rc)
    return out


def hostsim_render(blob, camera, params, spp, sample_offset=0, rgb_sum=None):
    if rgb_sum is None:
        rgb_sum = np.zeros((params.height, params.width, 3), dtype=np.float64)
    seg = C.c_uint64()
    rc = hostsim().hs_render(blob, len(blob), C.byref(camera), C.byref(params), sample_offset, spp, rgb_sum.ctypes.data, C.byref(seg))
    if rc != 0:
        raise RuntimeError("hostsim render failed: %d" % rc)
    return rgb_sum, seg.value


def hostsim_sample_radiance(blob, camera, params, x, y, sample):
    rgb = (C.c_double * 3)()
    rc = hostsim().hs_sample_radiance(blob, len(blob), C.byref(camera), C.byref(params), x, y, sample, rgb)
    if rc != 0:
        raise RuntimeError("hostsim sample failed: %d" % rc)
    return np.array(rgb[:])
