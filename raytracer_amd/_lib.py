"""ctypes binding of the C ABI declared in include/rt.h (librt_hip.so).

There is no fallback: if the HIP library has not been built, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RT_HIP_LIB: another build of the same library, e.g. one with the debug counters of -DRT_PHASE_STATS)
LIB_PATH = os.environ.get("RT_HIP_LIB") or os.path.join(_HERE, "librt_hip.so")

RT_OK, RT_ERR_INVALID, RT_ERR_UNSUPPORTED, RT_ERR_NO_DEVICE, RT_ERR_OOM, RT_ERR_PANIC = 0, -1, -2, -3, -4, -5
RT_MISS = 0xFFFFFFFF
RT_HIT_FLAG_HIT, RT_HIT_FLAG_FRONT = 1, 2


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code


class rt_camera(C.Structure):
    _fields_ = [("from_", C.c_double * 3), ("at", C.c_double * 3), ("vup", C.c_double * 3), ("vfov", C.c_double),
                ("aspect", C.c_double), ("aperture", C.c_double), ("focus", C.c_double), ("time0", C.c_double),
                ("time1", C.c_double)]


class rt_scene_info(C.Structure):
    _fields_ = [("world", C.c_int), ("camera", rt_camera), ("background", C.c_double * 3), ("background_mode", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("spp", C.c_uint32), ("max_depth", C.c_uint32)]


class rt_ray_d(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("ox", "oy", "oz", "dx", "dy", "dz", "time", "tmin", "tmax")]


class rt_hit(C.Structure):
    _fields_ = [("t", C.c_double), ("prim_id", C.c_uint32), ("flags", C.c_uint32)] + \
               [(n, C.c_double) for n in ("px", "py", "pz", "nx", "ny", "nz", "u", "v")] + \
               [("material_id", C.c_uint32), ("pad", C.c_uint32)]


class rt_params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("max_depth", C.c_uint32), ("background_mode", C.c_uint32),
                ("background", C.c_double * 3), ("seed", C.c_uint64), ("compat", C.c_uint32), ("tile", C.c_uint32),
                ("tile_rank", C.c_uint32), ("tile_world", C.c_uint32), ("spp_per_pass", C.c_uint32), ("profile", C.c_uint32),
                ("pad", C.c_uint32)]


class rt_stats(C.Structure):
    _fields_ = [("camera_samples", C.c_uint64), ("ray_segments", C.c_uint64), ("medium_tests", C.c_uint64), ("passes", C.c_uint64),
                ("seconds_render", C.c_double), ("raygen_ms", C.c_double), ("extend_ms", C.c_double), ("shade_main_ms", C.c_double),
                ("shade_ms", C.c_double * 5), ("accumulate_ms", C.c_double),
                ("raygen_launches", C.c_uint64), ("extend_launches", C.c_uint64), ("shade_main_launches", C.c_uint64),
                ("shade_launches", C.c_uint64), ("accumulate_launches", C.c_uint64),
                ("extend_segments", C.c_uint64)]


class rt_tree_stats(C.Structure):
    _fields_ = [("leaves", C.c_uint32), ("height", C.c_uint32), ("key_bits", C.c_uint32), ("stack_need", C.c_uint32),
                ("ms_collect", C.c_double), ("ms_sort", C.c_double), ("ms_hierarchy", C.c_double), ("ms_refit", C.c_double),
                ("ms_commit", C.c_double), ("seconds", C.c_double)]


# every symbol include/rt.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_D3 = C.POINTER(C.c_double)
PROTOTYPES = {
    "rt_builder_new": (_P, [C.c_uint64]),
    "rt_builder_free": (None, [_P]),
    "rt_builder_last_error": (C.c_char_p, [_P]),
    "rt_builder_set_flags": (C.c_int, [_P, C.c_uint32]),
    "rt_tex_solid": (C.c_int, [_P, C.c_double, C.c_double, C.c_double]),
    "rt_tex_checker": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_tex_noise": (C.c_int, [_P, C.c_double]),
    "rt_tex_image": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    "rt_mat_lambertian": (C.c_int, [_P, C.c_int]),
    "rt_mat_metal": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double]),
    "rt_mat_dielectric": (C.c_int, [_P, C.c_double]),
    "rt_mat_diffuse_light": (C.c_int, [_P, C.c_int, C.c_double]),
    "rt_mat_isotropic": (C.c_int, [_P, C.c_int]),
    "rt_sphere": (C.c_int, [_P, _D3, C.c_double, C.c_int]),
    "rt_moving_sphere": (C.c_int, [_P, _D3, _D3, C.c_double, C.c_double, C.c_double, C.c_int]),
    "rt_xy_rect": (C.c_int, [_P] + [C.c_double] * 5 + [C.c_int]),
    "rt_xz_rect": (C.c_int, [_P] + [C.c_double] * 5 + [C.c_int]),
    "rt_yz_rect": (C.c_int, [_P] + [C.c_double] * 5 + [C.c_int]),
    "rt_box": (C.c_int, [_P, _D3, _D3, C.c_int]),
    "rt_translate": (C.c_int, [_P, C.c_int, _D3]),
    "rt_rotate_y": (C.c_int, [_P, C.c_int, C.c_double]),
    "rt_constant_medium": (C.c_int, [_P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "rt_list_new": (C.c_int, [_P]),
    "rt_list_add": (C.c_int, [_P, C.c_int, C.c_int]),
    "rt_bvh": (C.c_int, [_P, C.c_int, C.c_double, C.c_double]),
    "rt_builder_compile": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "rt_builder_describe": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "rt_free": (None, [_P]),
    "rt_builder_register_image": (C.c_int, [_P, C.c_char_p, _P, C.c_uint32, C.c_uint32]),
    "rt_build_named_scene": (C.c_int, [_P, C.c_char_p, C.c_uint64, C.c_uint64, C.POINTER(rt_scene_info)]),
    "rt_init": (C.c_int, [C.POINTER(_P), C.c_int]),
    "rt_shutdown": (None, [_P]),
    "rt_last_error": (C.c_char_p, [_P]),
    "rt_scene_upload": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(_P)]),
    "rt_scene_upload_from_device": (C.c_int, [_P, _P, C.c_size_t, C.POINTER(_P)]),
    "rt_scene_free": (None, [_P, _P]),
    "rt_scene_build_tree_device": (C.c_int, [_P, _P, C.POINTER(rt_tree_stats)]),
    "rt_trace_first_hit": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P]),
    "rt_trace_first_hit_wavefront": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P, C.c_uint32]),
    "rt_trace_first_hit_noalloc": (C.c_int, [_P, _P, _P, C.c_size_t, _P, _P]),
    "rt_generate_rays": (C.c_int, [_P, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_uint32, _P]),
    "rt_render_accumulate": (C.c_int, [_P, _P, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_uint32, C.c_uint32, _P,
                                       C.POINTER(rt_stats)]),
    "rt_render": (C.c_int, [_P, _P, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_uint32, _P, _P, C.POINTER(rt_stats)]),
    "rt_device_count": (C.c_int, []),
    "rt_render_multi": (C.c_int, [C.POINTER(_P), C.c_uint32, _P, C.c_size_t, C.POINTER(rt_camera), C.POINTER(rt_params), C.c_uint32, _P, _P,
                                  C.POINTER(rt_stats)]),
    "rt_tonemap": (None, [_P, C.c_size_t, C.c_double, _P]),
    "rt_selftest_vec3": (C.c_int, [_P, C.POINTER(C.c_uint32)]),
    "rt_device_alloc": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "rt_device_free": (C.c_int, [_P, _P]),
    "rt_device_memset": (C.c_int, [_P, _P, C.c_int, C.c_size_t]),
    "rt_device_download": (C.c_int, [_P, _P, _P, C.c_size_t]),
    "rt_device_upload": (C.c_int, [_P, _P, _P, C.c_size_t]),
}


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing: build it with `python -m raytracer_amd.build product` "
                          "(there is no CPU fallback for the render path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()
