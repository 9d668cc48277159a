// TEST TOOL — runs the per-lane device code (raytracer_amd/csrc/rt_device.h) on the CPU, one path at a time,
// straight from a compiled scene blob. It exists so that the blob layout, the traversal, the hit
// reconstruction and the materials can be checked against the oracle in a container without a GPU, and under
// host sanitizers. It is never loaded by the product and is not a fallback: the product has no CPU path.
#include <stdint.h>
// work counters of the traversal, per bounce: [bounce][0] node visits, [bounce][1 + RtRefType] object tests
static thread_local uint64_t (*g_visits)[12] = nullptr;
static thread_local uint32_t g_bounce = 0;
#define RT_VISIT(kind) do { if (g_visits) ++g_visits[g_bounce][(kind)]; } while (0)
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rt.h"
#include "../../raytracer_amd/csrc/rt_builder.h"
#include "../../raytracer_amd/csrc/rt_device.h"

namespace {

struct VecStack {
    uint32_t v[RT_STACK_CAP];
    uint32_t sp = 0, high = 0;
    bool overflow = false;
    void push(uint32_t x) {
        if (sp < RT_STACK_CAP) { v[sp++] = x; if (sp > high) high = sp; }
        else overflow = true;
    }
    uint32_t pop() { return v[--sp]; }
    bool empty() const { return sp == 0; }
};
struct HeldRay {
    Ray r;
    Ray operator()() const { return r; }
};
struct PhiloxXi {
    Sampler s;
    double operator()(uint32_t m) const { return sampler_draw1(s, RT_STREAM_MEDIUM0 + m, 0u); }
};
struct ArrayXi {
    const double* row;
    double operator()(uint32_t m) const { return row[m]; }
};

bool view_of(const void* blob, size_t nbytes, DScene& v, RtBlobHeader& h) {
    if (nbytes < sizeof h) return false;
    memcpy(&h, blob, sizeof h);
    if (h.magic != RT_BLOB_MAGIC || h.version != RT_BLOB_VERSION || h.total_bytes != nbytes) return false;
    const uint8_t* b = (const uint8_t*)blob;
    v.nodes = (const RtNode*)(b + h.nodes.offset);
    v.spheres = (const RtSphere*)(b + h.spheres.offset);
    v.sphere_meta = (const RtMeta*)(b + h.sphere_meta.offset);
    v.mspheres = (const RtMSphere*)(b + h.mspheres.offset);
    v.msphere_meta = (const RtMeta*)(b + h.msphere_meta.offset);
    v.rects = (const RtRect*)(b + h.rects.offset);
    v.rect_meta = (const RtMeta*)(b + h.rect_meta.offset);
    v.boxes = (const RtBox*)(b + h.boxes.offset);
    v.box_meta = (const RtMeta*)(b + h.box_meta.offset);
    v.media = (const RtMedium*)(b + h.media.offset);
    v.insts = (const RtInst*)(b + h.insts.offset);
    v.list_entries = (const uint32_t*)(b + h.list_entries.offset);
    v.materials = (const RtMaterial*)(b + h.materials.offset);
    v.textures = (const RtTexture*)(b + h.textures.offset);
    v.perlin_vec = (const double*)(b + h.perlin_vec.offset);
    v.perlin_perm = (const int32_t*)(b + h.perlin_perm.offset);
    v.texels = (const uint8_t*)(b + h.texels.offset);
    v.exact_boxes = (const RtExactBox*)(b + h.exact_boxes.offset);
    v.chains = (const RtChain*)(b + h.chains.offset);
    v.flat_one_medium = 0u; v.flat_pad = 0u;
#ifdef RT_BOUNDS_CHECK
    v.counts = RtCounts{(uint32_t)h.nodes.count, (uint32_t)h.spheres.count, (uint32_t)h.mspheres.count, (uint32_t)h.rects.count, (uint32_t)h.boxes.count,
                        (uint32_t)h.media.count, (uint32_t)h.insts.count, (uint32_t)h.list_entries.count, (uint32_t)h.materials.count,
                        (uint32_t)h.textures.count, (uint32_t)h.exact_boxes.count, (uint32_t)h.chains.count};
#endif
    memset(&v.flat_medium, 0, sizeof v.flat_medium); memset(&v.flat_sphere, 0, sizeof v.flat_sphere);
    v.world = RT_REF_NONE;
    v.world_flat = h.world_flat_any_time;
    v.world_tree = h.world_tree;
    v.n_media = h.n_media;
    v.compat = 0;
    v.node_extent = h.node_extent;
    return true;
}

uint32_t material_of(const DScene& sc, const RtHitLite& h) {
    const uint32_t ty = RT_REF_TYPE(h.ref), idx = RT_REF_INDEX(h.ref);
    if (ty == RT_REF_SPHERE) return sc.sphere_meta[idx].material;
    if (ty == RT_REF_MSPHERE) return sc.msphere_meta[idx].material;
    if (ty == RT_REF_RECT) return sc.rect_meta[idx].material;
    if (ty == RT_REF_BOX) return sc.box_meta[idx].material;
    return sc.media[idx].material;
}

} // namespace

extern "C" {

// what k_first_hit does, serially. max_stack (nullable) receives the deepest stack use seen.
int hs_trace_first_hit(const void* blob, size_t nbytes, const rt_ray_d* rays, size_t n, const double* xi_rows, uint32_t compat,
                       rt_hit* out, uint32_t* max_stack) {
    DScene sc;
    RtBlobHeader h;
    if (!view_of(blob, nbytes, sc, h)) return RT_ERR_INVALID;
    sc.compat = compat;
    uint32_t high = 0;
    for (size_t i = 0; i < n; ++i)
        if (rays[i].time < h.valid0 || rays[i].time > h.valid1) return RT_ERR_UNSUPPORTED; // as rt_trace_first_hit
    for (size_t i = 0; i < n; ++i) {
        Ray r;
        r.o = v3(rays[i].ox, rays[i].oy, rays[i].oz); r.d = v3(rays[i].dx, rays[i].dy, rays[i].dz); r.tm = rays[i].time;
        VecStack st;
        ArrayXi xi;
        xi.row = xi_rows ? xi_rows + i * sc.n_media : nullptr;
        HeldRay held;
        held.r = r;
        const RtHitLite hl = rt_extend(sc, held, rays[i].tmin, rays[i].tmax, st, xi, xi_rows == nullptr);
        if (st.overflow) return RT_ERR_UNSUPPORTED;
        if (st.high > high) high = st.high;
        rt_hit o;
        memset(&o, 0, sizeof o);
        if (hl.ref == RT_REF_NONE) { o.prim_id = RT_MISS; o.t = rays[i].tmax; }
        else {
            const HitRec rec = rt_reconstruct(sc, r, hl, true);
            o.t = rec.t; o.prim_id = rec.src_id;
            o.flags = RT_HIT_FLAG_HIT | (rec.front_face ? RT_HIT_FLAG_FRONT : 0u) | (hl.aux << 4);
            o.px = rec.p.x; o.py = rec.p.y; o.pz = rec.p.z; o.nx = rec.normal.x; o.ny = rec.normal.y; o.nz = rec.normal.z;
            o.u = rec.u; o.v = rec.v; o.material_id = rec.material;
        }
        out[i] = o;
    }
    if (max_stack) *max_stack = high;
    return RT_OK;
}

int hs_generate_rays(const rt_camera* cam, uint32_t width, uint32_t height, uint32_t sample, uint64_t seed, uint32_t compat,
                     rt_ray_d* out) {
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return rc;
    for (uint32_t pixel = 0; pixel < width * height; ++pixel) {
        const uint32_t y = pixel / width, x = pixel - y * width;
        const Sampler smp = sampler_make(pixel, sample, 0u, seed);
        const Ray r = rt_primary_ray(cd, x, y, width, height, smp, compat);
        rt_ray_d o{r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.tm, 0.001, RT_INF};
        out[pixel] = o;
    }
    return RT_OK;
}

// One sample exactly as the kernels compute it: raygen, then per bounce extend -> (miss | shade), with the
// forward throughput product and radiance sum the path state holds on the device.
static V3 sample_radiance(const DScene& sc, const RtCameraDev& cd, const rt_params* p, uint32_t x, uint32_t y, uint32_t sample,
                          uint64_t* segments) {
    const uint32_t pixel = y * p->width + x;
    Ray r = rt_primary_ray(cd, x, y, p->width, p->height, sampler_make(pixel, sample, 0u, p->seed), p->compat);
    V3 thr = v3(1.0, 1.0, 1.0), rad = v3(0.0, 0.0, 0.0);
    for (uint32_t it = 0; it < p->max_depth; ++it) {
        const Sampler smp = sampler_make(pixel, sample, it, p->seed);
        VecStack st;
        PhiloxXi xi;
        xi.s = smp;
        if (segments) ++*segments;
        g_bounce = it;
        if (g_visits) ++g_visits[it][11];
        HeldRay held;
        held.r = r;
        const RtHitLite h = rt_extend(sc, held, 0.001, RT_INF, st, xi, false);
        if (h.ref == RT_REF_NONE) {
            const V3 bg = rt_background(p->background, p->background_mode, r);
            rad.x += thr.x * bg.x; rad.y += thr.y * bg.y; rad.z += thr.z * bg.z;
            break;
        }
        const RtMaterial m = sc.materials[material_of(sc, h)];
        const HitRec rec = rt_reconstruct(sc, r, h, rt_material_reads_uv(sc, m));
        const ShadeOut o = rt_shade(sc, m, r, rec, smp);
        if (m.type == RT_MAT_DIFFUSE_LIGHT) { rad.x += thr.x * o.emitted.x; rad.y += thr.y * o.emitted.y; rad.z += thr.z * o.emitted.z; }
        if (!(o.did_scatter && it + 1u < p->max_depth)) break;
        r.o = o.scattered.o; r.d = o.scattered.d;
        thr.x *= o.attenuation.x; thr.y *= o.attenuation.y; thr.z *= o.attenuation.z;
    }
    return rad;
}

int hs_render(const void* blob, size_t nbytes, const rt_camera* cam, const rt_params* p, uint32_t sample_offset, uint32_t spp,
              double* rgb_sum, uint64_t* segments) {
    DScene sc;
    RtBlobHeader h;
    if (!view_of(blob, nbytes, sc, h)) return RT_ERR_INVALID;
    sc.compat = p->compat;
    if (cam->time0 < h.valid0 || cam->time1 > h.valid1) return RT_ERR_UNSUPPORTED;
    if (cam->time0 >= h.shutter0 && cam->time1 <= h.shutter1 && cam->time0 <= cam->time1) sc.world_flat = h.world_flat; // as rt_render_accumulate
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return rc;
    for (uint32_t y = 0; y < p->height; ++y)
        for (uint32_t x = 0; x < p->width; ++x) {
            double* px = rgb_sum + ((size_t)y * p->width + x) * 3;
            // the device sums a pass's samples first and then adds to the buffer only through k_accumulate's
            // running sum, which starts from the buffer value: same order as here
            double r = px[0], g = px[1], b = px[2];
            for (uint32_t s = 0; s < spp; ++s) {
                const V3 c = sample_radiance(sc, cd, p, x, y, sample_offset + s, segments);
                r += c.x; g += c.y; b += c.z;
            }
            px[0] = r; px[1] = g; px[2] = b;
        }
    return RT_OK;
}

int hs_sample_radiance(const void* blob, size_t nbytes, const rt_camera* cam, const rt_params* p, uint32_t x, uint32_t y,
                       uint32_t sample, double rgb[3]) {
    DScene sc;
    RtBlobHeader h;
    if (!view_of(blob, nbytes, sc, h)) return RT_ERR_INVALID;
    sc.compat = p->compat;
    if (cam->time0 < h.valid0 || cam->time1 > h.valid1) return RT_ERR_UNSUPPORTED;
    if (cam->time0 >= h.shutter0 && cam->time1 <= h.shutter1 && cam->time0 <= cam->time1) sc.world_flat = h.world_flat; // as rt_render_accumulate
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return rc;
    const V3 c = sample_radiance(sc, cd, p, x, y, sample, nullptr);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
    return RT_OK;
}

// Work per bounce over a whole render: visits[bounce][0] node visits, [1 + RtRefType] object tests, [11] segments.
int hs_visit_stats(const void* blob, size_t nbytes, const rt_camera* cam, const rt_params* p, uint32_t spp, uint64_t* visits /* max_depth x 12 */) {
    DScene sc;
    RtBlobHeader h;
    if (!view_of(blob, nbytes, sc, h)) return RT_ERR_INVALID;
    sc.compat = p->compat;
    if (cam->time0 < h.valid0 || cam->time1 > h.valid1) return RT_ERR_UNSUPPORTED;
    if (cam->time0 >= h.shutter0 && cam->time1 <= h.shutter1 && cam->time0 <= cam->time1) sc.world_flat = h.world_flat; // as rt_render_accumulate
    RtCameraDev cd;
    int rc = rt_camera_derive(cam, &cd);
    if (rc != RT_OK) return rc;
    memset(visits, 0, sizeof(uint64_t) * 12 * p->max_depth);
    g_visits = (uint64_t (*)[12])visits;
    for (uint32_t y = 0; y < p->height; ++y)
        for (uint32_t x = 0; x < p->width; ++x)
            for (uint32_t s = 0; s < spp; ++s) sample_radiance(sc, cd, p, x, y, s, nullptr);
    g_visits = nullptr;
    return RT_OK;
}

// Work of the traversal for a caller's rays: visits[0] node visits, visits[1 + RtRefType] object tests, summed over the rays.
int hs_trace_visits(const void* blob, size_t nbytes, const rt_ray_d* rays, size_t n, const double* xi_rows, uint64_t* visits /* 12 */) {
    DScene sc;
    RtBlobHeader h;
    if (!view_of(blob, nbytes, sc, h)) return RT_ERR_INVALID;
    memset(visits, 0, sizeof(uint64_t) * 12);
    g_visits = (uint64_t (*)[12])visits;
    g_bounce = 0;
    for (size_t i = 0; i < n; ++i) {
        Ray r;
        r.o = v3(rays[i].ox, rays[i].oy, rays[i].oz); r.d = v3(rays[i].dx, rays[i].dy, rays[i].dz); r.tm = rays[i].time;
        VecStack st;
        ArrayXi xi;
        xi.row = xi_rows ? xi_rows + i * sc.n_media : nullptr;
        HeldRay held;
        held.r = r;
        (void)rt_extend(sc, held, rays[i].tmin, rays[i].tmax, st, xi, xi_rows == nullptr);
    }
    g_visits = nullptr;
    return RT_OK;
}

void hs_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    Philox4 o = philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1]);
    for (int i = 0; i < 4; ++i) out[i] = o.v[i];
}


// The reference's Vec3 assertions (vec3.rs:259-399) on the product's own vector type, evaluated here on the CPU.
uint32_t hs_vec3_kat(double one) { return rt_vec3_kat(one); }
} // extern "C"
