/* C ABI of the MI355X path-tracing hot path (librt_hip.so).
 *
 * The reference (Winter-Is-Coming-Stark/Raytracer, a Rust crate) has no FFI of its own: the hot path
 * sits behind three traits and one function. Each entry point below names the reference interface it
 * stands in for. A Rust host binds these with `extern "C"`; the Python package binds them with ctypes.
 *
 * Conventions: every pointer passed in or out is owned by the caller unless stated; the library owns
 * all device memory; calls block until their result is ready; nothing unwinds or aborts across this
 * boundary — every reference `panic!` site is an error code here. Return value 0 = ok, < 0 = error
 * (text via rt_last_error / rt_builder_last_error). One rt_context (and one rt_builder) may be used
 * from one host thread at a time. All geometry is IEEE f64 like the reference (vec3.rs:5-9).
 */
#ifndef RT_H
#define RT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum {
    RT_OK = 0,
    RT_ERR_INVALID = -1,     /* bad argument / bad handle */
    RT_ERR_UNSUPPORTED = -2, /* a scene shape the device path does not implement */
    RT_ERR_NO_DEVICE = -3,   /* no HIP device, or HIP runtime error */
    RT_ERR_OOM = -4,
    RT_ERR_PANIC = -5        /* a condition on which the reference would panic */
};

/* ------------------------------------------------------------------------------------------------
 * Scene building: the reference's constructors, flattened to handles.
 * A handle is a non-negative int; negative = error code. Textures, materials and hittables have
 * separate handle spaces. Hittable handles are issued 0,1,2,... in creation order and are the
 * `prim_id` reported by rt_trace_first_hit.
 * ------------------------------------------------------------------------------------------------ */
typedef struct rt_builder rt_builder;

/* `seed` drives the randomness the reference takes from thread_rng() while BUILDING a scene:
 * BVH split axes (bvh.rs:22) and Perlin tables (perlin.rs:28-54). */
rt_builder* rt_builder_new(uint64_t seed);
void rt_builder_free(rt_builder*);
const char* rt_builder_last_error(const rt_builder*);
/* flags: bit 0 = RotateY keeps its rotated bounding box (book behaviour) instead of discarding it
 * (hittable.rs:117-143); set before creating RotateY objects.
 * bit 1 = every BvhNode keeps the topology BvhNode::new builds (random axis, median split, bvh.rs:15-66). Without
 * it, a BvhNode whose boxes all bound their contents is compiled into a surface-area-heuristic tree over the same
 * objects: closest hits are the same, fewer boxes are tested. Trees with a box that does not bound its contents
 * (anything over the reference's RotateY) always keep the reference's topology and test order.
 * bit 2 = rays may carry any time. A BvhNode over moving spheres has boxes for its own [time0, time1] only (bvh.rs:15,
 * moving_sphere.rs:71-84); by default such a node is rebuilt like any other and the compiled scene is valid for ray
 * times inside that interval - a render with a wider camera shutter, or a first-hit query outside it, returns
 * RT_ERR_UNSUPPORTED. With the bit set such nodes are replayed in the reference's form (same culling, slower) and
 * any time is accepted. Set before creating moving spheres.
 * bit 3 = the world's tree is left to the device: rt_builder_compile puts the leaves under a balanced stand-in tree
 * (correct, but slow to walk) instead of searching for good splits, and rt_scene_build_tree_device arranges them after the
 * upload. Ignored together with bit 1 and for worlds that hold a BvhNode whose boxes do not bound their contents.
 * With bit 3 rt_bvh keeps only a node's box and forgets its members' order, so a scene that turns out to need the
 * reference's order of tests everywhere - one holding a moving sphere with time0 == time1 (every ray "hits" it at
 * t = NaN), or a medium of density < 0 - must create that object BEFORE the rt_bvh calls it affects: created after
 * them, rt_builder_compile returns RT_ERR_UNSUPPORTED (the reference-shaped trees can no longer be made). Without bit 3
 * the order of creation does not matter.
 * Default 0. */
int rt_builder_set_flags(rt_builder*, uint32_t flags);

/* Texture (texture.rs:8-10) */
int rt_tex_solid(rt_builder*, double r, double g, double b);          /* SolidColor::new, texture.rs:18 */
int rt_tex_checker(rt_builder*, int even_tex, int odd_tex);           /* CheckerTexture::new, /root/reference/texture.rs:40-48 */
int rt_tex_noise(rt_builder*, double scale);                          /* NoiseTexture::new, texture.rs:63-70 */
/* ImageTexture::new (texture.rs:90-103) with the decode already done: tightly packed RGB8, row 0 = top. */
int rt_tex_image(rt_builder*, const uint8_t* rgb8, uint32_t width, uint32_t height);

/* Material (material.rs:10-22) */
int rt_mat_lambertian(rt_builder*, int albedo_tex);                   /* Lambertian::new_by_pointer, material.rs:36 */
int rt_mat_metal(rt_builder*, double r, double g, double b, double fuzz); /* Metal::new, material.rs:68-77 (fuzz clamped to <= 1) */
int rt_mat_dielectric(rt_builder*, double ir);                        /* Dielectric::new, material.rs:105 */
int rt_mat_diffuse_light(rt_builder*, int emit_tex, double k);        /* DiffuseLight{emit,k}, material.rs:153-189 */
int rt_mat_isotropic(rt_builder*, int albedo_tex);                    /* Isotropic, material.rs:207-217 */

/* Hittable (hittable.rs:48-51).
 * A radius <= 0 is taken as the reference takes it: the surface of radius |r| with the normal turned inwards
 * (sphere.rs:46), hit like any other sphere in a list; under rt_bvh its inside-out box (sphere.rs:66-72) culls what
 * the reference's BvhNode would cull (such a tree is kept in the reference's shape and walked in its order).
 * Other parameters without sense are not refused either, since the reference computes something definite for them:
 * a rect or box without extent in an axis (hit in a list, never under rt_bvh: AABB::hit passes no box without
 * volume, aabb.rs:47), a box with p0 > p1 (keeps the faces across that axis), a medium of density < 0 (scatters in
 * front of its boundary; what it reports depends on what was tested before it, constant_medium.rs:40-61), a moving
 * sphere with time0 == time1 (every ray hits it at t = NaN, moving_sphere.rs:37-60). Scenes holding the last two are
 * walked in the reference's order of tests throughout, which costs speed, not results. */
int rt_sphere(rt_builder*, const double center[3], double radius, int mat);                 /* sphere.rs:16 */
int rt_moving_sphere(rt_builder*, const double c0[3], const double c1[3], double time0, double time1,
                     double radius, int mat);                                               /* moving_sphere.rs:19 */
int rt_xy_rect(rt_builder*, double x0, double x1, double y0, double y1, double k, int mat); /* arrect.rs:18 */
int rt_xz_rect(rt_builder*, double x0, double x1, double z0, double z1, double k, int mat); /* arrect.rs:70 */
int rt_yz_rect(rt_builder*, double y0, double y1, double z0, double z1, double k, int mat); /* arrect.rs:122 */
int rt_box(rt_builder*, const double p0[3], const double p1[3], int mat);                   /* _box.rs:15 */
int rt_translate(rt_builder*, int child, const double offset[3]);                           /* hittable.rs:61 */
int rt_rotate_y(rt_builder*, int child, double angle_degrees);                              /* hittable.rs:104 */
int rt_constant_medium(rt_builder*, int boundary, double density, double r, double g, double b); /* constant_medium.rs:15 */
int rt_list_new(rt_builder*);                                                               /* hittable_list.rs:27 */
int rt_list_add(rt_builder*, int list, int child);                                          /* hittable_list.rs:15 */
/* BvhNode::new_(list, time0, time1) (bvh.rs:68-71). Sorts `list` in place like the reference - except with builder flag
 * bit 3 where all boxes bound their contents: the node then only keeps its box and the list keeps its order (nothing reads
 * the order of a list whose tree the device will build). RT_ERR_PANIC if a member has no bounding box (bvh.rs:58,77). */
int rt_bvh(rt_builder*, int list, double time0, double time1);

/* Flatten the object graph below `world_list` into a relocatable blob (malloc'ed; free with rt_free).
 * RT_ERR_UNSUPPORTED for shapes the device path does not take: a ConstantMedium whose boundary is not a
 * primitive/box under at most RT_MAX_CHAIN Translate/RotateY, or instance nesting deeper than that. */
int rt_builder_compile(rt_builder*, int world_list, void** blob, size_t* nbytes);
/* The same scene as a replayable text log of the constructor calls above (exact hex floats), for an
 * independent implementation to rebuild the object graph from. malloc'ed; free with rt_free. */
int rt_builder_describe(rt_builder*, int world_list, char** text, size_t* nbytes);
void rt_free(void*);

/* Ready-made scenes: the reference's scene functions (main.rs:260-835) and two scaled-up variants.
 * Names: "final_scene", "cornell_box", "cornell_smoke", "random_scene", "two_spheres",
 * "two_perlin_spheres", "earth", "simple_light", "spheres_grid" (n = spheres), "stress_volume".
 * Scene randomness (box heights, sphere centres, materials) comes from `scene_seed`.
 * "earthmap.jpg" must be registered first for scenes that open it (texture.rs:91 would panic). */
typedef struct {
    double from[3], at[3], vup[3], vfov, aspect, aperture, focus, time0, time1; /* Camera::new inputs, camera.rs:21-29 */
} rt_camera;
typedef struct {
    int world;                 /* hittable handle of the world list */
    rt_camera camera;          /* the camera that goes with the scene */
    double background[3];
    uint32_t background_mode;  /* 0 constant, 1 sky gradient */
    uint32_t width, height, spp, max_depth; /* the configuration's image */
} rt_scene_info;
int rt_builder_register_image(rt_builder*, const char* filename, const uint8_t* rgb8, uint32_t width, uint32_t height);
int rt_build_named_scene(rt_builder*, const char* name, uint64_t scene_seed, uint64_t n, rt_scene_info* info);

/* ------------------------------------------------------------------------------------------------
 * Device side
 * ------------------------------------------------------------------------------------------------ */
typedef struct rt_context rt_context; /* one per process: a HIP device, a stream, work buffers */
typedef struct rt_scene rt_scene;     /* device-resident copy of a compiled blob */

/* Ray + the t_min/t_max arguments of Hittable::hit (ray.rs:5-9, hittable.rs:49) */
typedef struct { double ox, oy, oz, dx, dy, dz, time, tmin, tmax; } rt_ray_d;
/* HitRecord (hittable.rs:9-18) plus the identity of what was hit */
typedef struct {
    double t;
    uint32_t prim_id;     /* hittable handle of the Sphere/MovingSphere/rect/_Box/ConstantMedium hit; 0xFFFFFFFF = miss */
    uint32_t flags;       /* bit 0 hit, bit 1 front_face, bits 4-6 box face (order of _box.rs:22-71) */
    double px, py, pz, nx, ny, nz, u, v;
    uint32_t material_id; /* material handle */
    uint32_t pad;
} rt_hit;
#define RT_HIT_FLAG_HIT 1u
#define RT_HIT_FLAG_FRONT 2u
#define RT_MISS 0xFFFFFFFFu

typedef struct {
    uint32_t width, height;
    uint32_t max_depth;        /* main.rs:79 */
    uint32_t background_mode;  /* 0 = constant `background` (main.rs:56-58); 1 = sky gradient of the older snapshot (/root/reference/main.rs:59-61) */
    double background[3];
    uint64_t seed;             /* render seed: Philox key material */
    uint32_t compat;           /* bit 0: book-correct RotateY back-rotation (else hittable.rs:167-171 as written);
                                  bit 1: book-correct XY lens disk (else vec3.rs:105 / camera.rs:58 as written) */
    uint32_t tile;             /* tile edge in pixels for multi-GPU sharding; 0 = default 32 */
    uint32_t tile_rank, tile_world; /* this context renders tiles k with k % tile_world == tile_rank; world 0/1 = everything */
    uint32_t spp_per_pass;     /* camera samples per pixel traced together. A pass holds 232 bytes of device memory per path (this
                                  context's pixels x spp_per_pass) and runs faster the larger it is: its bounces are dependent rounds
                                  of launches. 0 = as many as the job has, up to seven eighths of the device memory that is free
                                  (halved while the allocation fails); a given value is taken as is: RT_ERR_OOM if it does not fit */
    uint32_t profile;          /* 1 = bracket every kernel launch with HIP events (fills the *_ms fields) */
    uint32_t pad;
} rt_params;

typedef struct {
    uint64_t camera_samples;   /* iterations of main.rs:161-167 */
    uint64_t ray_segments;     /* world.hit calls issued by ray_color (main.rs:56), primary included */
    uint64_t medium_tests;     /* ConstantMedium::hit calls (constant_medium.rs:25) the reference makes for these segments with its media in the
                                  world list: ray_segments x n_media. Not counted as rays; the kernels skip many of them */
    uint64_t passes;
    double seconds_render;     /* host wall clock around the device work */
    /* HIP-event sums per kernel when profile = 1. extend_ms = closest-hit search (k_extend_tree); shade_main_ms =
     * misses, shading of plain-colour materials and queueing of the rest (k_shade_main); shade_ms[m] = shading of hits
     * on computed textures of material type m (Lambertian, Metal, Dielectric, DiffuseLight, Isotropic). */
    double raygen_ms, extend_ms, shade_main_ms, shade_ms[5], accumulate_ms;
    uint64_t raygen_launches, extend_launches, shade_main_launches, shade_launches, accumulate_launches;
    uint64_t extend_segments;  /* segments traced by the timed extend launches */
} rt_stats;

int rt_init(rt_context** out, int device);  /* RT_ERR_NO_DEVICE when HIP reports no usable device */
void rt_shutdown(rt_context*);
const char* rt_last_error(const rt_context*);

/* A blob is checked before it is used: version, section bounds, table indices, the steps of a lock-step tree; the world's
 * tree, where its leaves are plain objects, is walked once - a node that is its own descendant is RT_ERR_INVALID, and a
 * header whose stack_need the tree exceeds only costs the scene the faster traversal kernels (which push without asking for
 * room), not the render: the general kernel reports an overflow if one happens (RT_ERR_UNSUPPORTED from the render call). */
int rt_scene_upload(rt_context*, const void* blob, size_t nbytes, rt_scene** out);             /* host blob */
int rt_scene_upload_from_device(rt_context*, const void* d_blob, size_t nbytes, rt_scene** out); /* blob already in HBM (after an RCCL broadcast) */
void rt_scene_free(rt_context*, rt_scene*);

/* BvhNode::new (bvh.rs:15-66) for the world's tree, on the device: for a scene compiled with builder flag bit 3, arranges
 * the leaves of the stand-in tree by a linear BVH build (Morton codes of the leaf centroids, radix sort, hierarchy and
 * boxes in parallel, two passes of tree rotations) and writes the nodes over the stand-in's. Closest hits are those of any other tree over the same
 * objects. RT_ERR_INVALID for a scene compiled without the flag or built already; RT_ERR_UNSUPPORTED when the tree would be taller than
 * the traversal stack (the stand-in then stays, and the scene stays usable). No render may be in flight on the scene. */
typedef struct {
    uint32_t leaves, height;  /* objects under the tree; nodes on its longest root-to-leaf chain */
    uint32_t key_bits;        /* Morton code bits used: 63, or 30 when 63 gave too tall a tree */
    uint32_t stack_need;      /* traversal stack entries the scene needs with the new tree */
    double ms_collect, ms_sort, ms_hierarchy, ms_refit, ms_commit; /* device time of the stages */
    double seconds;           /* host wall clock of the call */
} rt_tree_stats;
int rt_scene_build_tree_device(rt_context*, rt_scene*, rt_tree_stats* stats /* nullable */);

/* Hittable::hit of the world list for a batch of rays: out[i] = what world.hit(rays[i], tmin, tmax, rec)
 * (hittable_list.rs:40-53) leaves in rec. medium_xi: n * n_media uniforms, row i = the value each
 * ConstantMedium would draw at constant_medium.rs:53 for ray i (medium order = creation order);
 * NULL = ConstantMedium objects never hit. */
int rt_trace_first_hit(rt_context*, const rt_scene*, const rt_ray_d* rays, size_t n, const double* medium_xi,
                       rt_hit* out);

/* The same query answered by the kernel the renderer spends its time in (k_extend_tree: persistent waves, node cache
 * in LDS, leaf rounds, queue dealt in pieces; for a scene whose world is all on the lock-step list - a few objects, or a
 * small reference-order tree as steps - k_extend_list: one ray per lane, the list front to back) instead of the
 * one-lane-one-ray kernel of rt_trace_first_hit: same arguments, same results bit for bit except the winner of an exact
 * tie. Exists so that the parity gate of hittable_list.rs:40-53 / bvh.rs:112-122 can be put on the kernel that is
 * benchmarked; also the faster of the two for large batches.
 * variant: bit 0 = use the instantiation with exact f64 box tests (aabb.rs:35-75) even if no node of the scene needs them,
 * bit 1 = the one with the general ConstantMedium boundary code even if every medium is bounded by a sphere,
 * bit 2 = k_extend_tree even for a scene without a tree, bit 3 = not the instantiation that leaves out the code for lists,
 * frame changes and linked instance records (which a scene whose tree has only plain leaves gets otherwise), bit 4 = the
 * node records in LDS as 64-byte records, not as arrays read by the ray's signs (which a small plain tree gets otherwise),
 * bit 5 = moving spheres keep a leaf test of their own (a plain tree over a scene where at least one primitive in eight is a
 * moving sphere otherwise runs them through the spheres' test, one square root and division sequence for both kinds). */
int rt_trace_first_hit_wavefront(rt_context*, const rt_scene*, const rt_ray_d* rays, size_t n, const double* medium_xi,
                                 rt_hit* out, uint32_t variant);

/* rt_trace_first_hit without any device allocation. No counterpart in the reference: it is for callers that cannot
 * allocate device memory while they run - tight serving loops, harnesses that forbid hipMalloc. (It allocates nothing;
 * it still copies from the caller's pageable host memory and waits for the result, so a HIP graph capture cannot record it.)
 * Same arguments and same results, bit for bit; the staging copies of rays, uniforms and hit records and the kernel's
 * overflow flag live in a fixed 16 MiB work area that the library holds on every GPU it runs on, so the batch is
 * limited: n * (72 + 96 + 8 * n_media [0 when medium_xi is NULL]) <= 16 MiB - 768 bytes (about 99 000 rays without
 * uniforms); RT_ERR_INVALID with the limit in rt_last_error beyond that. The work area is shared by every context on
 * the same GPU: calls that overlap on one GPU race. */
int rt_trace_first_hit_noalloc(rt_context*, const rt_scene*, const rt_ray_d* rays, size_t n, const double* medium_xi,
                               rt_hit* out);

/* Camera::get_ray behind the pixel jitter (main.rs:162-164, camera.rs:56-64) for sample `sample` of each
 * pixel: out[(y*width + x)] for the whole image. For tests. */
int rt_generate_rays(rt_context*, const rt_camera*, const rt_params*, uint32_t sample, rt_ray_d* out);

/* The sample loop (main.rs:155-168) for samples [sample_offset, sample_offset+spp) of this context's
 * tiles: adds each sample's ray_color, in sample order, to d_rgb_sum — a DEVICE buffer of
 * width*height*3 doubles, row y <-> v = y/(height-1) (row 0 is the bottom of the view, main.rs:163,170).
 * Pixels of other ranks' tiles are left untouched. stats (nullable) is accumulated into. */
int rt_render_accumulate(rt_context*, const rt_scene*, const rt_camera*, const rt_params*,
                         uint32_t sample_offset, uint32_t spp, double* d_rgb_sum, rt_stats* stats);

/* Whole job with host buffers: rgb_sum_out = width*height*3 linear radiance sums over `spp` samples;
 * rgb8_out (nullable) = the reference's pixel conversion (main.rs:172-189). */
int rt_render(rt_context*, const rt_scene*, const rt_camera*, const rt_params*, uint32_t spp,
              double* rgb_sum_out, uint8_t* rgb8_out, rt_stats* stats);

/* The partition and the gather that main() wraps round the sample loop (main.rs:135-152: row bands dealt to the workers
 * of a thread pool, each sharing the scene through Arc::clone; :192 every job sends its band; :242-253 the bands are
 * copied into the image), over the GPUs of one node. ctxs[0 .. n_ctx) are contexts of this process - normally one per
 * device (rt_init(&c, i) for i < rt_device_count()), several on one device work too. The compiled scene `blob` is
 * uploaded to the first context's device and copied from there to every other device directly (GPU to GPU: xGMI on a
 * fully connected node); context i, on a host thread of its own, renders every sample of the 32x32 tiles k with
 * k % n_ctx == i (tile_rank / tile_world of `params` are overwritten); the per-context buffers travel to the first
 * device, where each context's tiles are taken from its buffer, and the image comes back as rt_render returns it.
 * Random numbers are keyed on (pixel, sample, bounce), so the image is the one-context image bit for bit for any n_ctx.
 * stats (nullable): n_ctx records, one per context. Errors are reported on ctxs[0]. No context may be in use elsewhere
 * during the call. */
int rt_device_count(void); /* HIP devices visible to the process (0 when there is none) */
int rt_render_multi(rt_context* const* ctxs, uint32_t n_ctx, const void* blob, size_t nbytes, const rt_camera*, const rt_params*,
                    uint32_t spp, double* rgb_sum_out, uint8_t* rgb8_out, rt_stats* stats);

/* main.rs:172-189 on the host for an already gathered sum buffer */
void rt_tonemap(const double* rgb_sum, size_t n_pixels, double spp, uint8_t* rgb8_out);

/* The reference's own Vec3 assertions (vec3.rs:259-399: twenty known answers in a commented-out test module - add, sub,
 * scalar forms, mul, div, cross, neg, squared_length, length, unit, and the zero-length unit() that panics there and is
 * the zero vector here), evaluated by one lane of the device on the vector type the kernels use. *failed = one bit per
 * assertion that did not hold, in the module's order; 0 = all hold. */
int rt_selftest_vec3(rt_context*, uint32_t* failed);

/* Device memory for callers without their own allocator (freed by rt_device_free). */
int rt_device_alloc(rt_context*, size_t nbytes, void** d_ptr);
int rt_device_free(rt_context*, void* d_ptr);
int rt_device_memset(rt_context*, void* d_ptr, int value, size_t nbytes);
int rt_device_download(rt_context*, void* host_dst, const void* d_src, size_t nbytes);
int rt_device_upload(rt_context*, void* d_dst, const void* host_src, size_t nbytes);

#ifdef __cplusplus
}
#endif
#endif /* RT_H */
