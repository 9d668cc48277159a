//! 1:1 with include/rt.h: every struct `#[repr(C)]`, every entry point, same order as the header.
//! Checked against the header by tests/test_rust_shim.py (names, arity, argument and return types, struct sizes).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_UNSUPPORTED: c_int = -2;
pub const RT_ERR_NO_DEVICE: c_int = -3;
pub const RT_ERR_OOM: c_int = -4;
pub const RT_ERR_PANIC: c_int = -5;
pub const RT_HIT_FLAG_HIT: u32 = 1;
pub const RT_HIT_FLAG_FRONT: u32 = 2;
pub const RT_MISS: u32 = 0xFFFF_FFFF;

pub enum rt_builder {}
pub enum rt_context {}
pub enum rt_scene {}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub from: [f64; 3],
    pub at: [f64; 3],
    pub vup: [f64; 3],
    pub vfov: f64,
    pub aspect: f64,
    pub aperture: f64,
    pub focus: f64,
    pub time0: f64,
    pub time1: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_scene_info {
    pub world: c_int,
    pub camera: rt_camera,
    pub background: [f64; 3],
    pub background_mode: u32,
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub max_depth: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_ray_d {
    pub ox: f64,
    pub oy: f64,
    pub oz: f64,
    pub dx: f64,
    pub dy: f64,
    pub dz: f64,
    pub time: f64,
    pub tmin: f64,
    pub tmax: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_hit {
    pub t: f64,
    pub prim_id: u32,
    pub flags: u32,
    pub px: f64,
    pub py: f64,
    pub pz: f64,
    pub nx: f64,
    pub ny: f64,
    pub nz: f64,
    pub u: f64,
    pub v: f64,
    pub material_id: u32,
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_params {
    pub width: u32,
    pub height: u32,
    pub max_depth: u32,
    pub background_mode: u32,
    pub background: [f64; 3],
    pub seed: u64,
    pub compat: u32,
    pub tile: u32,
    pub tile_rank: u32,
    pub tile_world: u32,
    pub spp_per_pass: u32,
    pub profile: u32,
    pub pad: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_stats {
    pub camera_samples: u64,
    pub ray_segments: u64,
    pub medium_tests: u64,
    pub passes: u64,
    pub seconds_render: f64,
    pub raygen_ms: f64,
    pub extend_ms: f64,
    pub shade_main_ms: f64,
    pub shade_ms: [f64; 5],
    pub accumulate_ms: f64,
    pub raygen_launches: u64,
    pub extend_launches: u64,
    pub shade_main_launches: u64,
    pub shade_launches: u64,
    pub accumulate_launches: u64,
    pub extend_segments: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_tree_stats {
    pub leaves: u32,
    pub height: u32,
    pub key_bits: u32,
    pub stack_need: u32,
    pub ms_collect: f64,
    pub ms_sort: f64,
    pub ms_hierarchy: f64,
    pub ms_refit: f64,
    pub ms_commit: f64,
    pub seconds: f64,
}

extern "C" {
    // ---- scene building: the reference's constructors, flattened to handles ----
    pub fn rt_builder_new(seed: u64) -> *mut rt_builder;
    pub fn rt_builder_free(b: *mut rt_builder);
    pub fn rt_builder_last_error(b: *const rt_builder) -> *const c_char;
    pub fn rt_builder_set_flags(b: *mut rt_builder, flags: u32) -> c_int;
    pub fn rt_tex_solid(b: *mut rt_builder, r: f64, g: f64, bl: f64) -> c_int;
    pub fn rt_tex_checker(b: *mut rt_builder, even_tex: c_int, odd_tex: c_int) -> c_int;
    pub fn rt_tex_noise(b: *mut rt_builder, scale: f64) -> c_int;
    pub fn rt_tex_image(b: *mut rt_builder, rgb8: *const u8, width: u32, height: u32) -> c_int;
    pub fn rt_mat_lambertian(b: *mut rt_builder, albedo_tex: c_int) -> c_int;
    pub fn rt_mat_metal(b: *mut rt_builder, r: f64, g: f64, bl: f64, fuzz: f64) -> c_int;
    pub fn rt_mat_dielectric(b: *mut rt_builder, ir: f64) -> c_int;
    pub fn rt_mat_diffuse_light(b: *mut rt_builder, emit_tex: c_int, k: f64) -> c_int;
    pub fn rt_mat_isotropic(b: *mut rt_builder, albedo_tex: c_int) -> c_int;
    pub fn rt_sphere(b: *mut rt_builder, center: *const f64, radius: f64, mat: c_int) -> c_int;
    pub fn rt_moving_sphere(b: *mut rt_builder, c0: *const f64, c1: *const f64, time0: f64, time1: f64, radius: f64, mat: c_int) -> c_int;
    pub fn rt_xy_rect(b: *mut rt_builder, x0: f64, x1: f64, y0: f64, y1: f64, k: f64, mat: c_int) -> c_int;
    pub fn rt_xz_rect(b: *mut rt_builder, x0: f64, x1: f64, z0: f64, z1: f64, k: f64, mat: c_int) -> c_int;
    pub fn rt_yz_rect(b: *mut rt_builder, y0: f64, y1: f64, z0: f64, z1: f64, k: f64, mat: c_int) -> c_int;
    pub fn rt_box(b: *mut rt_builder, p0: *const f64, p1: *const f64, mat: c_int) -> c_int;
    pub fn rt_translate(b: *mut rt_builder, child: c_int, offset: *const f64) -> c_int;
    pub fn rt_rotate_y(b: *mut rt_builder, child: c_int, angle_degrees: f64) -> c_int;
    pub fn rt_constant_medium(b: *mut rt_builder, boundary: c_int, density: f64, r: f64, g: f64, bl: f64) -> c_int;
    pub fn rt_list_new(b: *mut rt_builder) -> c_int;
    pub fn rt_list_add(b: *mut rt_builder, list: c_int, child: c_int) -> c_int;
    pub fn rt_bvh(b: *mut rt_builder, list: c_int, time0: f64, time1: f64) -> c_int;
    pub fn rt_builder_compile(b: *mut rt_builder, world_list: c_int, blob: *mut *mut c_void, nbytes: *mut usize) -> c_int;
    pub fn rt_builder_describe(b: *mut rt_builder, world_list: c_int, text: *mut *mut c_char, nbytes: *mut usize) -> c_int;
    pub fn rt_free(p: *mut c_void);
    pub fn rt_builder_register_image(b: *mut rt_builder, filename: *const c_char, rgb8: *const u8, width: u32, height: u32) -> c_int;
    pub fn rt_build_named_scene(b: *mut rt_builder, name: *const c_char, scene_seed: u64, n: u64, info: *mut rt_scene_info) -> c_int;

    // ---- device side ----
    pub fn rt_init(out: *mut *mut rt_context, device: c_int) -> c_int;
    pub fn rt_shutdown(ctx: *mut rt_context);
    pub fn rt_last_error(ctx: *const rt_context) -> *const c_char;
    pub fn rt_scene_upload(ctx: *mut rt_context, blob: *const c_void, nbytes: usize, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_upload_from_device(ctx: *mut rt_context, d_blob: *const c_void, nbytes: usize, out: *mut *mut rt_scene) -> c_int;
    pub fn rt_scene_free(ctx: *mut rt_context, scene: *mut rt_scene);
    pub fn rt_scene_build_tree_device(ctx: *mut rt_context, scene: *mut rt_scene, stats: *mut rt_tree_stats) -> c_int;
    pub fn rt_trace_first_hit(ctx: *mut rt_context, scene: *const rt_scene, rays: *const rt_ray_d, n: usize, medium_xi: *const f64, out: *mut rt_hit) -> c_int;
    pub fn rt_trace_first_hit_wavefront(ctx: *mut rt_context, scene: *const rt_scene, rays: *const rt_ray_d, n: usize, medium_xi: *const f64, out: *mut rt_hit, variant: u32) -> c_int;
    pub fn rt_trace_first_hit_noalloc(ctx: *mut rt_context, scene: *const rt_scene, rays: *const rt_ray_d, n: usize, medium_xi: *const f64, out: *mut rt_hit) -> c_int;
    pub fn rt_generate_rays(ctx: *mut rt_context, camera: *const rt_camera, params: *const rt_params, sample: u32, out: *mut rt_ray_d) -> c_int;
    pub fn rt_render_accumulate(ctx: *mut rt_context, scene: *const rt_scene, camera: *const rt_camera, params: *const rt_params, sample_offset: u32, spp: u32, d_rgb_sum: *mut f64, stats: *mut rt_stats) -> c_int;
    pub fn rt_render(ctx: *mut rt_context, scene: *const rt_scene, camera: *const rt_camera, params: *const rt_params, spp: u32, rgb_sum_out: *mut f64, rgb8_out: *mut u8, stats: *mut rt_stats) -> c_int;
    pub fn rt_device_count() -> c_int;
    pub fn rt_render_multi(ctxs: *const *mut rt_context, n_ctx: u32, blob: *const c_void, nbytes: usize, camera: *const rt_camera, params: *const rt_params, spp: u32, rgb_sum_out: *mut f64, rgb8_out: *mut u8, stats: *mut rt_stats) -> c_int;
    pub fn rt_tonemap(rgb_sum: *const f64, n_pixels: usize, spp: f64, rgb8_out: *mut u8);
    pub fn rt_selftest_vec3(ctx: *mut rt_context, failed: *mut u32) -> c_int;
    pub fn rt_device_alloc(ctx: *mut rt_context, nbytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn rt_device_free(ctx: *mut rt_context, d_ptr: *mut c_void) -> c_int;
    pub fn rt_device_memset(ctx: *mut rt_context, d_ptr: *mut c_void, value: c_int, nbytes: usize) -> c_int;
    pub fn rt_device_download(ctx: *mut rt_context, host_dst: *mut c_void, d_src: *const c_void, nbytes: usize) -> c_int;
    pub fn rt_device_upload(ctx: *mut rt_context, d_dst: *mut c_void, host_src: *const c_void, nbytes: usize) -> c_int;
}
