// Per-lane hot path: everything one path segment does, written once as host+device inline functions.
// The HIP kernels (rt_kernels.hip) call these with an LDS-backed stack; nothing here touches wave or
// workgroup state, so the same code can be single-stepped on a CPU by a test harness.
//
// Each function names the reference code it reproduces. Arithmetic is kept in the reference's
// operation order (no reassociation, no reciprocal-multiply in place of a divide) so that results
// match an f64 CPU evaluation bit for bit wherever only + - * / sqrt are involved.
#pragma once
#include "rt_math.h"
#include "rt_types.h"

#ifndef RT_INF
#define RT_INF (__builtin_inf())
#endif

// Debug builds (-DRT_BOUNDS_CHECK: `python -m raytracer_amd.build --sanitize`, tools/build_checked_lib.sh) carry the
// length of every section beside its pointer and check each reference before it is followed: the type it names and its
// index. A violation is printed (host and device) and the access is skipped; the host build aborts, so that a sanitizer run
// stops at the first one. Release builds compile none of it.
#ifdef RT_BOUNDS_CHECK
#include <stdio.h>
#include <stdlib.h>
struct RtCounts { uint32_t nodes, spheres, mspheres, rects, boxes, media, insts, list_entries, materials, textures, exact_boxes, chains; };
RT_HD bool rt_in_range(const char* what, uint32_t idx, uint32_t n, uint32_t ref) {
    if (idx < n) return true;
    printf("rt bounds: %s index %u is not below %u (reference %08x)\n", what, idx, n, ref);
#if !defined(__HIP_DEVICE_COMPILE__)
    abort();
#endif
    return false;
}
#define RT_IN_RANGE(sc, section, idx, ref) rt_in_range(#section, (uint32_t)(idx), (sc).counts.section, (uint32_t)(ref))
#else
#define RT_IN_RANGE(sc, section, idx, ref) true
#endif

// Device-side view of an uploaded blob: typed pointers to its sections.
struct DScene {
    const RtNode* nodes;
    const RtSphere* spheres;
    const RtMeta* sphere_meta;
    const RtMSphere* mspheres;
    const RtMeta* msphere_meta;
    const RtRect* rects;
    const RtMeta* rect_meta;
    const RtBox* boxes;
    const RtMeta* box_meta;
    const RtMedium* media;
    const RtInst* insts;
    const uint32_t* list_entries;
    const RtMaterial* materials;
    const RtTexture* textures;
    const double* perlin_vec;
    const int32_t* perlin_perm;
    const uint8_t* texels;
    const RtExactBox* exact_boxes;
    const RtChain* chains;
    uint32_t world, world_flat, world_tree;
    uint32_t n_media;
    uint32_t compat; // bit 0: book-correct RotateY back-rotation; bit 1: book-correct XY lens disk
    double node_extent; // largest finite |coordinate| of any node box
    // When the lock-step list is exactly one ConstantMedium bounded by a plain sphere (a fog around the whole scene), its
    // records ride here, in the kernel's arguments: scalar reads that do not queue behind the rays a refill has just
    // asked for, instead of three dependent vector reads (list entry, medium, sphere) with the whole wave waiting.
    uint32_t flat_one_medium; // 0: walk the list; else the medium's reference
    uint32_t flat_pad;
    RtMedium flat_medium;
    RtSphere flat_sphere;
#ifdef RT_BOUNDS_CHECK
    RtCounts counts;
#endif
};
#define RT_COMPAT_BOOK_ROTATE 1u
#define RT_COMPAT_BOOK_LENS 2u

// ---- instances ------------------------------------------------------------------------------------
// Ray as the child of `inst` sees it. Translate: hittable.rs:71; RotateY: hittable.rs:149-158.
RT_HD Ray rt_apply_inst(const RtInst& in, const Ray& r) {
    Ray o = r;
    if ((in.op & RT_OP_MASK) == RT_OP_TRANSLATE) {
        o.o = r.o - v3(in.a, in.b, in.c);
    } else {
        const double s = in.a, c = in.b;
        o.o.x = c * r.o.x - s * r.o.z;
        o.o.z = s * r.o.x + c * r.o.z;
        o.d.x = c * r.d.x - s * r.d.z;
        o.d.z = s * r.d.x + c * r.d.z;
    }
    return o;
}

// The direction part of rt_apply_inst.
RT_HD V3 rt_apply_inst_dir(const RtInst& in, V3 d) {
    if ((in.op & RT_OP_MASK) == RT_OP_TRANSLATE) return d;
    const double s = in.a, c = in.b;
    return v3(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
}

// The ray as the primitive of an RT_REF_ILEAF reference sees it: the chain's Translate / RotateY steps, outermost first
// (the same arithmetic as rt_apply_inst, read from one shared record instead of a linked RtInst each).
RT_HD Ray rt_apply_chain_op(const RtChainOp& in, const Ray& r) {
    Ray o = r;
    if ((in.op & RT_OP_MASK) == RT_OP_TRANSLATE) {
        o.o = r.o - v3(in.a, in.b, in.c);
    } else {
        const double s = in.a, c = in.b;
        o.o.x = c * r.o.x - s * r.o.z;
        o.o.z = s * r.o.x + c * r.o.z;
        o.d.x = c * r.d.x - s * r.d.z;
        o.d.z = s * r.d.x + c * r.d.z;
    }
    return o;
}
RT_HD V3 rt_apply_chain_op_dir(const RtChainOp& in, V3 d) { // the direction part
    if ((in.op & RT_OP_MASK) == RT_OP_TRANSLATE) return d;
    const double s = in.a, c = in.b;
    return v3(c * d.x - s * d.z, d.y, s * d.x + c * d.z);
}
RT_HD Ray rt_apply_chain(const DScene& sc, uint32_t chain, Ray r) {
    const RtChain* ch = sc.chains + chain;
    const uint32_t n = ch->n;
#pragma unroll 1
    for (uint32_t k = 0u; k < n && k < (uint32_t)RT_MAX_CHAIN; ++k) r = rt_apply_chain_op(ch->op[k], r);
    return r;
}

// The world ray carried down to the frame below instance `inst` (RT_NO_PARENT = world frame).
RT_HD Ray rt_ray_to_level(const DScene& sc, const Ray& wray, int32_t inst) {
    int32_t chain[RT_MAX_CHAIN];
    int32_t p = inst;
#pragma unroll
    for (int k = 0; k < RT_MAX_CHAIN; ++k) {
        chain[k] = p;
        if (p >= 0) p = sc.insts[p].parent_inst;
    }
    Ray r = wray;
#pragma unroll
    for (int k = RT_MAX_CHAIN - 1; k >= 0; --k) {
        if (chain[k] >= 0) r = rt_apply_inst(sc.insts[chain[k]], r);
    }
    return r;
}

// ---- primitive tests: return t only; the shade stage re-derives p / normal / uv from (ray, t) -------

// AABB::hit (aabb.rs:35-75), same arithmetic as the reference -
// (bound - origin) * (1/d) per slab - arranged for fewer instructions:
//   * the near / far bound of each axis is picked by the sign of 1/d instead of swapping t0 and t1 afterwards
//     (the same two products, selected before instead of after);
//   * the reference returns after any axis that empties the interval; the interval only ever shrinks, so one
//     test after the third axis gives the same verdict;
//   * the reference's `if t0 > t_min {t0} else {t_min}` ignores a NaN t0 (a zero direction component against a
//     coincident bound); so does fmax, and t_min itself is never NaN. The far side is reduced the same way and
//     joined with t_max by a select, because t_max (the closest hit so far) CAN be NaN - an axis-parallel ray
//     in the plane of a rect (arrect.rs:32-40) - and the reference then accepts every box.
// Returns whether the box is hit and the distance at which the ray enters it (>= t_min).
struct RtSlab {
    V3 inv; // 1 / d (aabb.rs:36,49,62)
    bool nx, ny, nz; // inv < 0 per axis (aabb.rs:39,52,65)
    // Constants of the conservative f32 test (rt_node_boxes_f32): slab distance = bound * i - c, with c pushed
    // outwards by the error bound (cn* for the near side, cf* for the far side).
    float ix, iy, iz, cnx, cny, cnz, cfx, cfy, cfz;
};
// `extent`: largest |coordinate| of any box the ray will be tested against.
// EXACT = false: for a traversal that never runs the f64 test (no RT_NODE_EXACT node in the scene): everything is then only
// wanted to f32 and is worked out in f32 - the reciprocal as the hardware's estimate (one instruction; off by 2^-23 |1/d|
// at most, and as much as 2^-24 more for d's own rounding), c as the product of the rounded origin with it. Against the
// bound below that doubles the share of i and adds 2^-24 |c| twice over: still under a quarter of the margin. `inv` is
// not set.
RT_HD float rt_rcp_estimate(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(x);
#else
    return 1.0f / x;
#endif
}
template <bool EXACT = true>
RT_HD RtSlab rt_slab_of(const Ray& r, double extent) {
    RtSlab s;
    if (!EXACT) {
        s.ix = rt_rcp_estimate((float)r.d.x); s.iy = rt_rcp_estimate((float)r.d.y); s.iz = rt_rcp_estimate((float)r.d.z);
        s.inv = v3(0.0, 0.0, 0.0);
        // (the sign of 1/d, not of d: a component of -0.0 has the reciprocal -inf, and the near side is then the upper bound)
        s.nx = s.ix < 0.0f; s.ny = s.iy < 0.0f; s.nz = s.iz < 0.0f;
        const float e = (float)extent;
        const float cx = (float)r.o.x * s.ix, cy = (float)r.o.y * s.iy, cz = (float)r.o.z * s.iz;
        const float mx = 0x1p-20f * (e * __builtin_fabsf(s.ix) + __builtin_fabsf(cx));
        const float my = 0x1p-20f * (e * __builtin_fabsf(s.iy) + __builtin_fabsf(cy));
        const float mz = 0x1p-20f * (e * __builtin_fabsf(s.iz) + __builtin_fabsf(cz));
        s.cnx = cx + mx; s.cfx = cx - mx;
        s.cny = cy + my; s.cfy = cy - my;
        s.cnz = cz + mz; s.cfz = cz - mz;
        return s;
    }
    s.inv = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    s.ix = (float)s.inv.x; s.iy = (float)s.inv.y; s.iz = (float)s.inv.z;
    // (the sign of 1/d, not of d: a component of -0.0 has the reciprocal -inf, and the near side is then the upper bound)
    s.nx = s.inv.x < 0.0; s.ny = s.inv.y < 0.0; s.nz = s.inv.z < 0.0;
    // f32 evaluation of (b - o) / d as fma(b, i, -c) with i = f32(1/d), c = f32(o/d): the result is off by at most
    // 2^-24 (|b i| + |c|) for the rounding of i, as much again for the rounding of c and of the result - under
    // 2^-22 (extent |1/d| + |o/d|). The margin is four times that.
    const double mx = 0x1p-20 * (extent * fabs(s.inv.x) + fabs(r.o.x * s.inv.x));
    const double my = 0x1p-20 * (extent * fabs(s.inv.y) + fabs(r.o.y * s.inv.y));
    const double mz = 0x1p-20 * (extent * fabs(s.inv.z) + fabs(r.o.z * s.inv.z));
    s.cnx = (float)(r.o.x * s.inv.x + mx); s.cfx = (float)(r.o.x * s.inv.x - mx);
    s.cny = (float)(r.o.y * s.inv.y + my); s.cfy = (float)(r.o.y * s.inv.y - my);
    s.cnz = (float)(r.o.z * s.inv.z + mz); s.cfz = (float)(r.o.z * s.inv.z - mz);
    return s;
}
template <class Bound> // float: the node's own copy of the bounds; double: the bounds the reference holds
RT_HD bool rt_aabb_hit(const Bound (&lo)[3], const Bound (&hi)[3], const Ray& r, const RtSlab& k, double t_min, double t_max,
                       double& t_enter) {
    const double t0x = ((double)(k.nx ? hi[0] : lo[0]) - r.o.x) * k.inv.x;
    const double t1x = ((double)(k.nx ? lo[0] : hi[0]) - r.o.x) * k.inv.x;
    const double t0y = ((double)(k.ny ? hi[1] : lo[1]) - r.o.y) * k.inv.y;
    const double t1y = ((double)(k.ny ? lo[1] : hi[1]) - r.o.y) * k.inv.y;
    const double t0z = ((double)(k.nz ? hi[2] : lo[2]) - r.o.z) * k.inv.z;
    const double t1z = ((double)(k.nz ? lo[2] : hi[2]) - r.o.z) * k.inv.z;
    const double t_near = fmax(fmax(fmax(t0x, t_min), t0y), t0z);
    const double t_far = fmin(fmin(t1x, t1y), t1z);
    t_max = t_far < t_max ? t_far : t_max;
    t_enter = t_near;
    return !(t_max <= t_near);
}
// Both boxes of a node whose boxes only cull (every box bounds what is below it, so entering a box the exact test
// would have rejected changes nothing but the work): f32 throughout, every rounding taken outwards. A NaN slab
// distance (0 * inf, inf - inf: a ray parallel to the slab) is ignored by fmaxf / fminf, which errs towards entering.
// t_min rounded down, t_max rounded up; e0, e1: entry distances, good enough to choose the nearer child.
// (the arithmetic, given each box's bound on the side the ray enters (n) and leaves (f) by, per axis)
RT_HD void rt_node_boxes_f32_picked(float n0x, float n0y, float n0z, float f0x, float f0y, float f0z, float n1x, float n1y, float n1z,
                                    float f1x, float f1y, float f1z, const RtSlab& k, float t_min, float t_max, bool& h0, bool& h1,
                                    float& e0, float& e1) {
    const float a0x = __builtin_fmaf(n0x, k.ix, -k.cnx), a0y = __builtin_fmaf(n0y, k.iy, -k.cny), a0z = __builtin_fmaf(n0z, k.iz, -k.cnz);
    const float b0x = __builtin_fmaf(f0x, k.ix, -k.cfx), b0y = __builtin_fmaf(f0y, k.iy, -k.cfy), b0z = __builtin_fmaf(f0z, k.iz, -k.cfz);
    const float a1x = __builtin_fmaf(n1x, k.ix, -k.cnx), a1y = __builtin_fmaf(n1y, k.iy, -k.cny), a1z = __builtin_fmaf(n1z, k.iz, -k.cnz);
    const float b1x = __builtin_fmaf(f1x, k.ix, -k.cfx), b1y = __builtin_fmaf(f1y, k.iy, -k.cfy), b1z = __builtin_fmaf(f1z, k.iz, -k.cfz);
    e0 = __builtin_fmaxf(__builtin_fmaxf(a0x, a0y), __builtin_fmaxf(a0z, t_min));
    e1 = __builtin_fmaxf(__builtin_fmaxf(a1x, a1y), __builtin_fmaxf(a1z, t_min));
    const float x0 = __builtin_fminf(__builtin_fminf(b0x, b0y), __builtin_fminf(b0z, t_max));
    const float x1 = __builtin_fminf(__builtin_fminf(b1x, b1y), __builtin_fminf(b1z, t_max));
    h0 = !(x0 < e0);
    h1 = !(x1 < e1);
}
RT_HD void rt_node_boxes_f32(const RtNode& n, const RtSlab& k, float t_min, float t_max, bool& h0, bool& h1, float& e0, float& e1) {
    const float n0x = k.nx ? n.hi0[0] : n.lo0[0], f0x = k.nx ? n.lo0[0] : n.hi0[0];
    const float n0y = k.ny ? n.hi0[1] : n.lo0[1], f0y = k.ny ? n.lo0[1] : n.hi0[1];
    const float n0z = k.nz ? n.hi0[2] : n.lo0[2], f0z = k.nz ? n.lo0[2] : n.hi0[2];
    const float n1x = k.nx ? n.hi1[0] : n.lo1[0], f1x = k.nx ? n.lo1[0] : n.hi1[0];
    const float n1y = k.ny ? n.hi1[1] : n.lo1[1], f1y = k.ny ? n.lo1[1] : n.hi1[1];
    const float n1z = k.nz ? n.hi1[2] : n.lo1[2], f1z = k.nz ? n.lo1[2] : n.hi1[2];
    rt_node_boxes_f32_picked(n0x, n0y, n0z, f0x, f0y, f0z, n1x, n1y, n1z, f1x, f1y, f1z, k, t_min, t_max, h0, h1, e0, e1);
}
// One node record as four 16-byte loads (records are 64-byte aligned).
RT_HD RtNode rt_load_node(const RtNode* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint4 a = *(const uint4*)p, b = *((const uint4*)p + 1), c = *((const uint4*)p + 2), d = *((const uint4*)p + 3);
    RtNode n;
    n.lo0[0] = __uint_as_float(a.x); n.lo0[1] = __uint_as_float(a.y); n.lo0[2] = __uint_as_float(a.z);
    n.hi0[0] = __uint_as_float(a.w); n.hi0[1] = __uint_as_float(b.x); n.hi0[2] = __uint_as_float(b.y);
    n.lo1[0] = __uint_as_float(b.z); n.lo1[1] = __uint_as_float(b.w); n.lo1[2] = __uint_as_float(c.x);
    n.hi1[0] = __uint_as_float(c.y); n.hi1[1] = __uint_as_float(c.z); n.hi1[2] = __uint_as_float(c.w);
    n.c0 = d.x; n.c1 = d.y;
    n.pad[0] = d.z; n.pad[1] = d.w;
    return n;
#else
    return *p;
#endif
}

// Sphere::hit (sphere.rs:34-64): strict delta > 0 and strict t range.
RT_HD bool rt_sphere_t(V3 center, double radius, const Ray& r, double t_min, double t_max, double& t_out) {
    V3 oc = r.o - center;
    double a = v3_sqlen(r.d);
    double half_b = v3_dot(r.d, oc);
    double c = v3_sqlen(oc) - radius * radius;
    double delta = half_b * half_b - a * c;
    if (delta > 0.0) {
        double root = sqrt(delta);
        double t = (-half_b - root) / a;
        if (t > t_min && t < t_max) { t_out = t; return true; }
        t = (-half_b + root) / a;
        if (t > t_min && t < t_max) { t_out = t; return true; }
    }
    return false;
}

// Both of the above behind one flag per lane, so that a wave whose lanes hold spheres of both kinds runs the square root and
// the divisions once: `inclusive` chooses MovingSphere::hit's comparisons (a zero discriminant hits, the ends of the t range
// count), else Sphere::hit's. The arithmetic of the two is the same, term for term.
RT_HD bool rt_either_sphere_t(V3 center, double radius, bool inclusive, const Ray& r, double t_min, double t_max, double& t_out) {
    V3 oc = r.o - center;
    double a = v3_sqlen(r.d);
    double half_b = v3_dot(r.d, oc);
    double c = v3_sqlen(oc) - radius * radius;
    double delta = half_b * half_b - a * c;
    if (inclusive ? delta < 0.0 : !(delta > 0.0)) return false;
    double root = sqrt(delta);
    double t = (-half_b - root) / a;
    if (inclusive ? (t < t_min || t > t_max) : !(t > t_min && t < t_max)) {
        t = (-half_b + root) / a;
        if (inclusive ? (t < t_min || t > t_max) : !(t > t_min && t < t_max)) return false;
    }
    t_out = t;
    return true;
}

// MovingSphere::center (moving_sphere.rs:37-40)
RT_HD V3 rt_msphere_center(const RtMSphere& s, double time) {
    V3 c0 = v3(s.c0[0], s.c0[1], s.c0[2]), c1 = v3(s.c1[0], s.c1[1], s.c1[2]);
    return c0 + (c1 - c0) * (time - s.t0) / (s.t1 - s.t0);
}
// MovingSphere::hit (moving_sphere.rs:44-69): discriminant == 0 hits, inclusive t range.
RT_HD bool rt_msphere_t(const RtMSphere& s, const Ray& r, double t_min, double t_max, double& t_out) {
    V3 oc = r.o - rt_msphere_center(s, r.tm);
    double a = v3_sqlen(r.d);
    double half_b = v3_dot(oc, r.d);
    double c = v3_sqlen(oc) - s.r * s.r;
    double discriminant = half_b * half_b - a * c;
    if (discriminant < 0.0) return false;
    double sqrtd = sqrt(discriminant);
    double root = (-half_b - sqrtd) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sqrtd) / a;
        if (root < t_min || root > t_max) return false;
    }
    t_out = root;
    return true;
}

// One axis-aligned rectangle (arrect.rs:31-40, 83-92, 135-144). AXIS as in RtRect, a compile-time constant so that
// picking the ray components never turns into indexed access to a copy of the ray in scratch memory.
template <int AXIS>
RT_HD bool rt_rect_axis_t(double a0, double a1, double b0, double b1, double k, const Ray& r, double t_min, double t_max,
                          double& t_out) {
    const double ok = AXIS == 0 ? r.o.z : (AXIS == 1 ? r.o.y : r.o.x);
    const double dk = AXIS == 0 ? r.d.z : (AXIS == 1 ? r.d.y : r.d.x);
    const double oa = AXIS == 2 ? r.o.y : r.o.x;
    const double da = AXIS == 2 ? r.d.y : r.d.x;
    const double ob = AXIS == 0 ? r.o.y : r.o.z;
    const double db = AXIS == 0 ? r.d.y : r.d.z;
    const double t = (k - ok) / dk;
    if (t < t_min || t > t_max) return false;
    const double a = oa + t * da;
    const double b = ob + t * db;
    if (a < a0 || a > a1 || b < b0 || b > b1) return false;
    t_out = t;
    return true;
}
RT_HD bool rt_rect_t(uint32_t axis, double a0, double a1, double b0, double b1, double k, const Ray& r, double t_min,
                     double t_max, double& t_out) {
    if (axis == 0u) return rt_rect_axis_t<0>(a0, a1, b0, b1, k, r, t_min, t_max, t_out);
    if (axis == 1u) return rt_rect_axis_t<1>(a0, a1, b0, b1, k, r, t_min, t_max, t_out);
    return rt_rect_axis_t<2>(a0, a1, b0, b1, k, r, t_min, t_max, t_out);
}

// _Box::hit = HittableList::hit over its six sides in construction order (_box.rs:22-71,
// hittable_list.rs:40-53): XY@z1, XY@z0, XZ@y1, XZ@y0, YZ@x1, YZ@x0, each against the shrinking
// closest_so_far with the rects' inclusive range, so a later equal-t side wins.
RT_HD bool rt_box_t(const RtBox& b, const Ray& r, double t_min, double t_max, double& t_out, uint32_t& face) {
    bool hit = false;
    double closest = t_max, t;
    if (rt_rect_axis_t<0>(b.p0[0], b.p1[0], b.p0[1], b.p1[1], b.p1[2], r, t_min, closest, t)) { hit = true; closest = t; face = 0u; }
    if (rt_rect_axis_t<0>(b.p0[0], b.p1[0], b.p0[1], b.p1[1], b.p0[2], r, t_min, closest, t)) { hit = true; closest = t; face = 1u; }
    if (rt_rect_axis_t<1>(b.p0[0], b.p1[0], b.p0[2], b.p1[2], b.p1[1], r, t_min, closest, t)) { hit = true; closest = t; face = 2u; }
    if (rt_rect_axis_t<1>(b.p0[0], b.p1[0], b.p0[2], b.p1[2], b.p0[1], r, t_min, closest, t)) { hit = true; closest = t; face = 3u; }
    if (rt_rect_axis_t<2>(b.p0[1], b.p1[1], b.p0[2], b.p1[2], b.p1[0], r, t_min, closest, t)) { hit = true; closest = t; face = 4u; }
    if (rt_rect_axis_t<2>(b.p0[1], b.p1[1], b.p0[2], b.p1[2], b.p0[0], r, t_min, closest, t)) { hit = true; closest = t; face = 5u; }
    if (hit) t_out = closest;
    return hit;
}
// The rectangle a box face stands for (same order as above).
RT_HD RtRect rt_box_face(const RtBox& b, uint32_t face) {
    RtRect q;
    q.pad = 0u;
    switch (face) {
    case 0u: q.axis = 0u; q.a0 = b.p0[0]; q.a1 = b.p1[0]; q.b0 = b.p0[1]; q.b1 = b.p1[1]; q.k = b.p1[2]; break;
    case 1u: q.axis = 0u; q.a0 = b.p0[0]; q.a1 = b.p1[0]; q.b0 = b.p0[1]; q.b1 = b.p1[1]; q.k = b.p0[2]; break;
    case 2u: q.axis = 1u; q.a0 = b.p0[0]; q.a1 = b.p1[0]; q.b0 = b.p0[2]; q.b1 = b.p1[2]; q.k = b.p1[1]; break;
    case 3u: q.axis = 1u; q.a0 = b.p0[0]; q.a1 = b.p1[0]; q.b0 = b.p0[2]; q.b1 = b.p1[2]; q.k = b.p0[1]; break;
    case 4u: q.axis = 2u; q.a0 = b.p0[1]; q.a1 = b.p1[1]; q.b0 = b.p0[2]; q.b1 = b.p1[2]; q.k = b.p1[0]; break;
    default: q.axis = 2u; q.a0 = b.p0[1]; q.a1 = b.p1[1]; q.b0 = b.p0[2]; q.b1 = b.p1[2]; q.k = b.p0[0]; break;
    }
    return q;
}

// A medium's boundary: a primitive or box, possibly under Translate/RotateY records.
RT_HD bool rt_boundary_t(const DScene& sc, uint32_t ref, Ray r, double t_min, double t_max, double& t_out) {
#pragma unroll 1
    for (int guard = 0; guard <= RT_MAX_CHAIN && RT_REF_TYPE(ref) == RT_REF_INST; ++guard) {
        const RtInst in = sc.insts[RT_REF_INDEX(ref)];
        r = rt_apply_inst(in, r);
        ref = in.child;
    }
    const uint32_t idx = RT_REF_INDEX(ref);
    switch (RT_REF_TYPE(ref)) {
    case RT_REF_SPHERE: {
        const RtSphere s = sc.spheres[idx];
        return rt_sphere_t(v3(s.cx, s.cy, s.cz), s.r, r, t_min, t_max, t_out);
    }
    case RT_REF_MSPHERE: return rt_msphere_t(sc.mspheres[idx], r, t_min, t_max, t_out);
    case RT_REF_RECT: {
        const RtRect q = sc.rects[idx];
        return rt_rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, r, t_min, t_max, t_out);
    }
    case RT_REF_BOX: {
        uint32_t face;
        return rt_box_t(sc.boxes[idx], r, t_min, t_max, t_out, face);
    }
    default: return false;
    }
}

// A source of medium uniforms may promise that every medium of the scene is bounded by a plain sphere
// (static constexpr bool only_sphere_media = true): the general boundary test is then not compiled in. It costs
// the traversal kernel 35 registers it does not have.
template <class T, class = void> struct RtOnlySphereMedia { static constexpr bool value = false; };
template <class T> struct RtOnlySphereMedia<T, decltype((void)T::only_sphere_media)> { static constexpr bool value = T::only_sphere_media; };

// ConstantMedium::hit (constant_medium.rs:25-66). `xi(medium_index)` is the uniform the reference draws at :53; it is
// only asked for once the ray is known to cross the boundary.
template <class XiFn>
RT_HD bool rt_medium_t(const DScene& sc, const RtMedium& m, const Ray& r, double t_min, double t_max, const XiFn& xi,
                       double& t_out, const RtSphere* boundary_sphere = nullptr) {
    double t1 = 0.0, t2 = 0.0;
    if (boundary_sphere != nullptr || RtOnlySphereMedia<XiFn>::value || RT_REF_TYPE(m.boundary) == RT_REF_SPHERE) {
        // The two boundary tests on a plain sphere are Sphere::hit (sphere.rs:34-64) twice on the same ray: same
        // discriminant, same two quotients, only the range differs - computed once, compared twice.
        const RtSphere s = boundary_sphere != nullptr ? *boundary_sphere : sc.spheres[RT_REF_INDEX(m.boundary)]; // (the caller may hold the record already)
        const V3 oc = r.o - v3(s.cx, s.cy, s.cz);
        const double a = v3_sqlen(r.d);
        const double half_b = v3_dot(r.d, oc);
        const double c = v3_sqlen(oc) - s.r * s.r;
        const double delta = half_b * half_b - a * c;
        if (!(delta > 0.0)) return false;
        const double root = sqrt(delta);
        const double q0 = (-half_b - root) / a, q1 = (-half_b + root) / a;
        if (q0 > -RT_INF && q0 < RT_INF) t1 = q0;
        else if (q1 > -RT_INF && q1 < RT_INF) t1 = q1;
        else return false;
        const double lo = t1 + 0.0001;
        if (q0 > lo && q0 < RT_INF) t2 = q0;
        else if (q1 > lo && q1 < RT_INF) t2 = q1;
        else return false;
    } else {   // boundary.hit(r, -inf, inf) then boundary.hit(r, rec1.t + 0.0001, inf) (constant_medium.rs:29-34); written as a
        // loop so that the boundary test is instantiated once
        double lo = -RT_INF, t = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            if (!rt_boundary_t(sc, m.boundary, r, lo, RT_INF, t)) return false;
            if (pass == 0) { t1 = t; lo = t + 0.0001; } else t2 = t;
        }
    }
    if (t1 < t_min) t1 = t_min;
    if (t2 > t_max) t2 = t_max;
    if (t1 >= t2) return false;
    if (t1 < 0.0) t1 = 0.0;
    const double ray_length = v3_len(r.d);
    const double distance_inside_boundary = (t2 - t1) * ray_length;
    const double hit_distance = m.neg_inv_density * log(xi(m.medium_index));
    if (hit_distance > distance_inside_boundary) return false;
    t_out = t1 + hit_distance / ray_length;
    return true;
}

// ---- closest-hit traversal ("extend") ---------------------------------------------------------------
// Reproduces world.hit(r, t_min, t_max, rec) (main.rs:56): HittableList::hit scans its entries in
// order against a shrinking closest_so_far (hittable_list.rs:40-53); BvhNode::hit tests its box with
// the current closest distance, then always left before right (bvh.rs:112-122). Because the closest
// distance only ever shrinks along that depth-first order, the recursion equals a loop with one
// running `best.t`.
//
// Shape of the loop (chosen for 64-wide waves). The closest hit does not depend on the order in which
// the world's entries are tested - only the winner of an exact tie does - so:
//  * the entries that are a primitive or medium are tested first, by every lane in lock step: entry
//    and parameters are wave-uniform (scalar loads, no divergence, no stack);
//  * the remaining entries (BvhNode, nested lists) are walked per lane with an explicit stack, in the
//    reference's order, as two alternating steps: descend through nodes until the lane holds a
//    non-node reference, then test that reference.
//
// Stack: push(uint32_t), pop(), empty().  XiFn: double operator()(uint32_t medium_index).
// RayFn: Ray operator()() returning the world-space ray (the kernels re-read it from the path buffers
// rather than holding a second copy in registers across an instance).
struct RtBest {
    double t;
    uint32_t ref, aux;
};
// A host build that wants to know how much work a ray causes defines RT_VISIT(kind) before including this header
// (kinds: 0 = a node's two box tests, 1 + RtRefType = a test of that kind of object).
#ifndef RT_VISIT
#define RT_VISIT(kind) ((void)0)
#endif

// One non-node reference against the ray of its frame. Returns true if it became the closest hit.
template <class XiFn>
RT_HD void rt_test_leaf(const DScene& sc, uint32_t ref, const Ray& r, double t_min, RtBest& best, const XiFn& xi,
                        bool skip_media) {
    const uint32_t idx = RT_REF_INDEX(ref);
    double t;
    RT_VISIT(1u + RT_REF_TYPE(ref));
#ifdef RT_BOUNDS_CHECK
    {
        const uint32_t ty = RT_REF_TYPE(ref);
        const bool known = ty == RT_REF_SPHERE ? RT_IN_RANGE(sc, spheres, idx, ref) : ty == RT_REF_MSPHERE ? RT_IN_RANGE(sc, mspheres, idx, ref)
                         : ty == RT_REF_RECT ? RT_IN_RANGE(sc, rects, idx, ref) : ty == RT_REF_BOX ? RT_IN_RANGE(sc, boxes, idx, ref)
                         : ty == RT_REF_MEDIUM ? RT_IN_RANGE(sc, media, idx, ref) : rt_in_range("leaf reference type", ty, 0u, ref);
        if (!known) return;
    }
#endif
    switch (RT_REF_TYPE(ref)) {
    case RT_REF_SPHERE: {
        const RtSphere s = sc.spheres[idx];
        if (rt_sphere_t(v3(s.cx, s.cy, s.cz), s.r, r, t_min, best.t, t)) { best.t = t; best.ref = ref; best.aux = 0u; }
        break;
    }
    case RT_REF_MSPHERE: {
        if (rt_msphere_t(sc.mspheres[idx], r, t_min, best.t, t)) { best.t = t; best.ref = ref; best.aux = 0u; }
        break;
    }
    case RT_REF_RECT: {
        const RtRect q = sc.rects[idx];
        if (rt_rect_t(q.axis, q.a0, q.a1, q.b0, q.b1, q.k, r, t_min, best.t, t)) { best.t = t; best.ref = ref; best.aux = 0u; }
        break;
    }
    case RT_REF_BOX: {
        uint32_t face = 0u;
        if (rt_box_t(sc.boxes[idx], r, t_min, best.t, t, face)) { best.t = t; best.ref = ref; best.aux = face; }
        break;
    }
    case RT_REF_MEDIUM: {
        if (!skip_media) {
            const RtMedium m = sc.media[idx];
            if (rt_medium_t(sc, m, r, t_min, best.t, xi, t)) { best.t = t; best.ref = ref; best.aux = 0u; }
        }
        break;
    }
    default: break;
    }
}

// Per-lane traversal state and its two step functions. A caller alternates rt_trav_descend (box tests only,
// until the lane holds something that is not a node) and rt_trav_visit (one non-node reference) until
// s.cur == RT_REF_NONE. Written as steps so that a kernel can interleave them with its own wave-level work
// (handing finished lanes a new ray) while the straight-line rt_traverse below just loops.
struct RtTrav {
    Ray r;        // ray of the current frame
    RtSlab slab;  // its box-test constants
    RtBest best;  // closest hit so far; best.t doubles as the reference's shrinking t_max
    uint32_t cur; // next reference to process; RT_REF_NONE = finished
};

// One node: s.cur must be a node. Both child boxes are tested against the current closest distance; the lane goes
// on with the child the ray enters first and keeps the other for later.
//
// BvhNode::hit (bvh.rs:112-122) tests its own box, then the left child, then the right one with the tightened
// t_max. Where every box really bounds its contents the closest hit does not depend on which boxes were tested,
// when, or in which order, so such trees are stored with real boxes on both children and walked nearest first.
// Where a box does not bound its contents (RotateY keeps its child's unrotated box, hittable.rs:115-143) what
// the reference finds depends on exactly which boxes it tests and with which t_max, so those trees are encoded
// to replay it: the node is flagged ordered (child 0 first, always), child 1 carries the infinite box and leads to a
// one-child node holding the real right box, which is therefore tested only after the left subtree is done.
// Nodes: RtNode load(uint32_t index), RtExactBox exact_box(uint32_t k) - where node records and the f64 bounds of exact
// nodes come from (global memory, or a kernel's cached copy).
struct GlobalNodes {
    const RtNode* nodes;
    const RtExactBox* exact; // f64 bounds of the boxes whose verdict is behaviour (RtNode::pad[1])
#ifdef RT_BOUNDS_CHECK
    uint32_t n_nodes;
#endif
    RT_HD RtNode load(uint32_t index) const { return rt_load_node(nodes + index); }
    RT_HD RtExactBox exact_box(uint32_t k) const { return exact[k]; }
};
// A stack may offer prefetch() - the entry that a pop would uncover, asked for before it is known whether this step
// pops - and node_step(), which ends the step with it: where a stack read has a latency worth hiding behind the node fetch.
template <class Stack> RT_HD auto rt_stack_prefetch(const Stack& st, int) -> decltype(st.prefetch()) { return st.prefetch(); }
template <class Stack> RT_HD uint32_t rt_stack_prefetch(const Stack&, long) { return 0u; }
// (ROOM_CHECKED = false: the caller knows the stack cannot overflow; a stack with a node_step of its own may then leave its check out)
template <bool ROOM_CHECKED, class Stack>
RT_HD auto rt_stack_node_step(Stack& st, bool both, bool any, uint32_t go, uint32_t keep, uint32_t under, int)
    -> decltype(st.template node_step<ROOM_CHECKED>(both, any, go, keep, under)) { return st.template node_step<ROOM_CHECKED>(both, any, go, keep, under); }
template <bool ROOM_CHECKED, class Stack>
RT_HD uint32_t rt_stack_node_step(Stack& st, bool both, bool any, uint32_t go, uint32_t keep, uint32_t, long) {
    if (both) st.push(keep);
    return any ? go : (st.empty() ? (uint32_t)RT_REF_NONE : st.pop());
}
// ... and pop_or_none(), where the stack can do "the top entry, or RT_REF_NONE from an empty stack" without asking which.
template <class Stack> RT_HD auto rt_stack_pop_or_none(Stack& st, int) -> decltype(st.pop_or_none()) { return st.pop_or_none(); }
template <class Stack> RT_HD uint32_t rt_stack_pop_or_none(Stack& st, long) { return st.empty() ? (uint32_t)RT_REF_NONE : st.pop(); }
// EXACT_NODES = false: for scenes without a node flagged RT_NODE_EXACT (the flag is then not looked at).
template <bool EXACT_NODES = true, bool ROOM_CHECKED = true, class Stack, class Nodes>
RT_HD void rt_trav_node_step(const Nodes& nodes, RtTrav& s, Stack& st, double t_min) {
    const uint32_t under = rt_stack_prefetch(st, 0);
#ifdef RT_BOUNDS_CHECK
    if (!rt_in_range("nodes", RT_REF_INDEX(s.cur), nodes.n_nodes, s.cur)) { s.cur = RT_REF_NONE; return; }
#endif
    const RtNode n = nodes.load(RT_REF_INDEX(s.cur));
    RT_VISIT(0u);
    bool h0, h1, swap;
    if (EXACT_NODES && n.pad[0] != 0u) {
        // Such a node has one real box at most, child 0's, held in f64 (the f32 copy in the record is for code that only
        // culls); its other boxes stand in front of objects that the reference tests without asking a box (a list's
        // entries, hittable_list.rs:45-51; a BvhNode's object children, bvh.rs:116-121): those are entered whatever the
        // closest distance has become - below t_min or NaN included - and child 0 always goes first.
        h0 = true;
        if (n.pad[1] != 0u) { // AABB::hit (aabb.rs:35-75) on the f64 bounds the reference's node holds
            double e0;
            const RtExactBox b = nodes.exact_box(n.pad[1] - 1u);
            h0 = rt_aabb_hit(b.lo, b.hi, s.r, s.slab, t_min, s.best.t, e0);
        }
        h1 = true;
        swap = false;
    } else {
        const float lo = (float)t_min, hi = (float)s.best.t; // widened by an ulp each: the conversions round to nearest
        float e0, e1;
        rt_node_boxes_f32(n, s.slab, lo - __builtin_fabsf(lo) * 0x1p-23f, hi + __builtin_fabsf(hi) * 0x1p-23f, h0, h1, e0, e1);
        swap = e1 < e0;
    }
    h1 = h1 && n.c1 != RT_REF_NONE; // (an absent child has the empty box, which a ray without direction would still "hit")
    // both boxes hit: on with the nearer child, the other is kept for later; one: on with it; none: the next kept one
    const bool both = h0 && h1;
    const bool first0 = h0 & (!h1 | !swap); // on with child 0: the nearer of two, or the only one (and the other one is what is kept of two)
    s.cur = rt_stack_node_step<ROOM_CHECKED>(st, both, h0 || h1, first0 ? n.c0 : n.c1, first0 ? n.c1 : n.c0, under, 0);
}
template <class Stack, class Nodes>
RT_HD void rt_trav_descend(const Nodes& nodes, RtTrav& s, Stack& st, double t_min) {
#pragma unroll 1
    while (RT_REF_TYPE(s.cur) == RT_REF_NODE) rt_trav_node_step(nodes, s, st, t_min);
}

// AHEAD = true: for a tree whose leaves are all plain - a primitive, a medium, or a primitive named together with a chain of
// at most two steps (the caller has checked: no list entry, no instance record, no frame to leave ever arrives here). The
// branches for those are left out, and a leaf's record and the steps of its chain are all asked for before any of them is
// used (one wait instead of up to four in a row).
// MERGE_MOVING (with AHEAD): a moving sphere's centre at the ray's time is worked out first and its lane then takes the
// spheres' test along with theirs (rt_either_sphere_t) - for scenes with many of them, where a wave at its leaves holds
// both kinds and would run the square root and the divisions twice.
template <bool AHEAD = false, bool MERGE_MOVING = false, class Stack, class XiFn, class RayFn>
RT_HD void rt_trav_visit(const DScene& sc, RtTrav& s, Stack& st, double t_min, const XiFn& xi, bool skip_media,
                         const RayFn& world_ray) {
    const uint32_t idx = RT_REF_INDEX(s.cur);
    const uint32_t type = RT_REF_TYPE(s.cur);
#ifdef RT_BOUNDS_CHECK
    {
        bool ok = true;
        if (type == RT_REF_LIST) ok = RT_IN_RANGE(sc, list_entries, idx, s.cur);
        else if (type == RT_REF_POP || type == RT_REF_INST) ok = RT_IN_RANGE(sc, insts, idx, s.cur);
        else if (type == RT_REF_ILEAF) ok = RT_IN_RANGE(sc, chains, RT_ILEAF_CHAIN(s.cur), s.cur);
        else if (type > RT_REF_ILEAF || type == RT_REF_NODE || type == RT_REF_NONE) ok = rt_in_range("reference type at a leaf", type, 0u, s.cur);
        if (!ok) { s.cur = rt_stack_pop_or_none(st, 0); return; }
    }
#endif
    if (!AHEAD && type == RT_REF_LIST) { // HittableList: entries in order (hittable_list.rs:45-51)
        const uint32_t e = sc.list_entries[idx];
        if (e != RT_REF_NONE) {
            st.push(RT_MAKE_REF(RT_REF_LIST, idx + 1u));
            s.cur = e;
            return;
        }
    } else if (!AHEAD && type == RT_REF_POP) { // leave instance idx: back to the ray of its own frame
        if (!st.empty()) { // (else nothing is left to test in the frame the lane would return to)
            const RtInst in = sc.insts[idx];
            s.r = rt_ray_to_level(sc, world_ray(), in.parent_inst);
            s.slab = rt_slab_of(s.r, sc.node_extent);
        }
    } else {
        uint32_t leaf = s.cur;
        Ray moved = s.r;
        if (type == RT_REF_ILEAF) leaf = RT_ILEAF_PRIM(s.cur);
        constexpr bool ahead = AHEAD;
        const bool moving_ahead = ahead && MERGE_MOVING && RT_REF_TYPE(leaf) == RT_REF_MSPHERE;
        const bool sphere_ahead = (ahead && RT_REF_TYPE(leaf) == RT_REF_SPHERE) || moving_ahead;
        const bool box_ahead = ahead && RT_REF_TYPE(leaf) == RT_REF_BOX;
        const bool medium_ahead = ahead && RT_REF_TYPE(leaf) == RT_REF_MEDIUM && !skip_media;
        // (a sphere's four doubles or a box's six, asked for beside the chain's record and before the wave works through
        // its kinds of leaves one after the other)
        RtBox rec;
        rec.p0[0] = 0.0; rec.p0[1] = 0.0; rec.p0[2] = 0.0; rec.p1[0] = 0.0; rec.p1[1] = 0.0; rec.p1[2] = 0.0;
#ifdef RT_BOUNDS_CHECK
        if ((moving_ahead && !RT_IN_RANGE(sc, mspheres, RT_REF_INDEX(leaf), s.cur)) || (sphere_ahead && !moving_ahead && !RT_IN_RANGE(sc, spheres, RT_REF_INDEX(leaf), s.cur)) || (box_ahead && !RT_IN_RANGE(sc, boxes, RT_REF_INDEX(leaf), s.cur)) ||
            (medium_ahead && !RT_IN_RANGE(sc, media, RT_REF_INDEX(leaf), s.cur))) { s.cur = rt_stack_pop_or_none(st, 0); return; }
#endif
        if (moving_ahead) { // (its centre at the ray's time - no step of a chain changes that - and the lane is one of the spheres')
            const RtMSphere ms = sc.mspheres[RT_REF_INDEX(leaf)];
            const V3 c = rt_msphere_center(ms, s.r.tm);
            rec.p0[0] = c.x; rec.p0[1] = c.y; rec.p0[2] = c.z; rec.p1[0] = ms.r;
        } else if (sphere_ahead) {
            const RtSphere sp = sc.spheres[RT_REF_INDEX(leaf)];
            rec.p0[0] = sp.cx; rec.p0[1] = sp.cy; rec.p0[2] = sp.cz; rec.p1[0] = sp.r;
        } else if (box_ahead) {
            rec = sc.boxes[RT_REF_INDEX(leaf)];
        } else if (medium_ahead) { // (32 bytes, kept as raw doubles until the medium's turn)
            const double* raw = (const double*)(sc.media + RT_REF_INDEX(leaf));
            rec.p0[0] = raw[0]; rec.p0[1] = raw[1]; rec.p0[2] = raw[2]; rec.p1[0] = raw[3];
        }
        if (type == RT_REF_ILEAF) {
            if (ahead) {
                const RtChain* ch = sc.chains + RT_ILEAF_CHAIN(s.cur);
                const uint32_t n = ch->n;
                const RtChainOp op0 = ch->op[0], op1 = ch->op[1]; // (unused steps of a record are zero-filled)
                if (n > 0u) moved = rt_apply_chain_op(op0, moved);
                if (n > 1u) moved = rt_apply_chain_op(op1, moved);
            } else {
                moved = rt_apply_chain(sc, RT_ILEAF_CHAIN(s.cur), moved);
            }
        } else if (!AHEAD && type == RT_REF_INST) {
            RtInst in = sc.insts[idx];
            if ((in.op & RT_INST_LEAF) == 0u) { // a subtree in its own frame: enter it, leave through a POP marker
                st.push(RT_MAKE_REF(RT_REF_POP, idx));
                s.r = rt_apply_inst(in, s.r);
                s.slab = rt_slab_of(s.r, sc.node_extent);
                s.cur = in.child;
                return;
            }
            // one primitive seen through instances: a moved copy of the ray is tested, the frame's ray and its box-test
            // constants stay as they are
#pragma unroll 1
            for (int guard = 0;; ++guard) {
                moved = rt_apply_inst(in, moved);
                leaf = in.child;
                if (RT_REF_TYPE(leaf) != RT_REF_INST || guard >= RT_MAX_CHAIN) break;
                in = sc.insts[RT_REF_INDEX(leaf)];
            }
        }
        if (sphere_ahead) {
            double t;
            RT_VISIT(1u + RT_REF_TYPE(leaf));
            const bool got = MERGE_MOVING ? rt_either_sphere_t(v3(rec.p0[0], rec.p0[1], rec.p0[2]), rec.p1[0], moving_ahead, moved, t_min, s.best.t, t)
                                          : rt_sphere_t(v3(rec.p0[0], rec.p0[1], rec.p0[2]), rec.p1[0], moved, t_min, s.best.t, t);
            if (got) { s.best.t = t; s.best.ref = leaf; s.best.aux = 0u; }
        } else if (box_ahead) {
            double t;
            uint32_t face = 0u;
            RT_VISIT(1u + RT_REF_BOX);
            if (rt_box_t(rec, moved, t_min, s.best.t, t, face)) { s.best.t = t; s.best.ref = leaf; s.best.aux = face; }
        } else if (medium_ahead) {
            double t;
            RtMedium m;
            static_assert(sizeof(RtMedium) == 32, "RtMedium is fetched ahead as four doubles");
            const double raw[4] = {rec.p0[0], rec.p0[1], rec.p0[2], rec.p1[0]};
            __builtin_memcpy(&m, raw, sizeof m);
            RT_VISIT(1u + RT_REF_MEDIUM);
            if (rt_medium_t(sc, m, moved, t_min, s.best.t, xi, t)) { s.best.t = t; s.best.ref = leaf; s.best.aux = 0u; }
        } else {
            rt_test_leaf(sc, leaf, moved, t_min, s.best, xi, skip_media);
        }
    }
    s.cur = rt_stack_pop_or_none(st, 0);
}

// Everything below `root`, per lane, starting from an empty stack. `r` is the ray of root's frame.
template <class Stack, class XiFn, class RayFn>
RT_HD void rt_traverse(const DScene& sc, uint32_t root, const Ray& r, double t_min, RtBest& best, Stack& st, const XiFn& xi,
                       bool skip_media, const RayFn& world_ray) {
    RtTrav s;
    s.r = r;
    s.slab = rt_slab_of(r, sc.node_extent);
    s.best = best;
    s.cur = root;
    GlobalNodes nodes;
    nodes.nodes = sc.nodes;
    nodes.exact = sc.exact_boxes;
#ifdef RT_BOUNDS_CHECK
    nodes.n_nodes = sc.counts.nodes;
#endif
#pragma unroll 1
    for (;;) {
        rt_trav_descend(nodes, s, st, t_min);
        if (s.cur == RT_REF_NONE) break;
        rt_trav_visit(sc, s, st, t_min, xi, skip_media, world_ray);
    }
    best = s.best;
}

// The world entries that are a primitive or a medium, possibly wrapped in Translate / RotateY: every lane
// tests the same entry at the same time, so entry and parameters are wave-uniform (scalar loads, no stack).
// ONE_MEDIUM: compiled with the short cut for a list that is one sphere-bounded medium (DScene::flat_one_medium). The
// traversal kernel's instantiations are short of registers, each in its own way; only the one that scenes like the
// Book-2 scene get carries it (elsewhere the same few lines cost 1 - 1.5 % of the kernel's time and won nothing).
// One entry of a lock-step list - a primitive or medium, alone or below instances: named by its reference, RT_REF_ILEAF,
// or below linked records, RT_REF_INST (the compiler never puts a named leaf under linked records on such a list).
// `r` is the world ray on entry and on return.
template <class XiFn, class RayFn>
RT_HD void rt_flat_entry(const DScene& sc, uint32_t e, Ray& r, const RayFn& world_ray, double t_min, RtBest& best, const XiFn& xi,
                         bool skip_media) {
    bool moved = false;
    if (RT_REF_TYPE(e) == RT_REF_ILEAF) {
        r = rt_apply_chain(sc, RT_ILEAF_CHAIN(e), r);
        e = RT_ILEAF_PRIM(e);
        moved = true;
    }
#pragma unroll 1
    while (RT_REF_TYPE(e) == RT_REF_INST) {
        const RtInst in = sc.insts[RT_REF_INDEX(e)];
        r = rt_apply_inst(in, r);
        e = in.child;
        moved = true;
    }
    rt_test_leaf(sc, e, r, t_min, best, xi, skip_media);
    if (moved) r = world_ray();
}

// A small tree in the reference's order (RT_REF_STEPS), walked by all lanes together: BvhNode::hit tests its box against
// the closest distance so far, then left, then right (bvh.rs:112-122), so the walk is the same sequence of steps for every
// ray and only WHICH steps a ray takes part in differs - one bit per depth says whether the lane is inside. Every step's
// operands are wave-uniform (scalar loads), no lane keeps a stack, and no lane waits for another to reach a leaf. The box
// tests are AABB::hit on the reference's f64 bounds (aabb.rs:35-75), with 1 / d as the reference computes it.
// `at` = index of the RT_REF_STEPS entry; returns the number of list words the steps occupy.
template <class XiFn, class RayFn>
RT_HD uint32_t rt_flat_steps(const DScene& sc, uint32_t at, Ray& r, const RayFn& world_ray, double t_min, RtBest& best,
                             const XiFn& xi, bool skip_media) {
    const uint32_t n_steps = RT_REF_INDEX(sc.list_entries[at]);
    RtSlab k;
    k.inv = v3(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
    k.nx = k.inv.x < 0.0; k.ny = k.inv.y < 0.0; k.nz = k.inv.z < 0.0;
    uint32_t inside = 1u; // bit d: the lane is inside the subtree at depth d
#pragma unroll 1
    for (uint32_t j = 0u; j < n_steps; ++j) {
        const uint32_t ctrl = sc.list_entries[at + 1u + 2u * j], operand = sc.list_entries[at + 2u + 2u * j];
        const uint32_t d = ctrl & RT_STEP_DEPTH_MASK;
        const bool here = ((inside >> (d - 1u)) & 1u) != 0u;
        if ((ctrl & RT_STEP_GATE) != 0u) {
            bool h = false;
            if (here) {
                RT_VISIT(0u);
                bool known = true;
#ifdef RT_BOUNDS_CHECK
                known = RT_IN_RANGE(sc, exact_boxes, operand - 1u, ctrl);
#endif
                if (known) {
                    const RtExactBox b = sc.exact_boxes[operand - 1u];
                    double entered;
                    h = rt_aabb_hit(b.lo, b.hi, r, k, t_min, best.t, entered);
                }
            }
            inside = (inside & ~(1u << d)) | ((h ? 1u : 0u) << d);
        } else if (here) {
            rt_flat_entry(sc, operand, r, world_ray, t_min, best, xi, skip_media);
        }
    }
    return 2u * n_steps;
}

// The world entries that are a primitive or a medium, possibly wrapped in Translate / RotateY: every lane
// tests the same entry at the same time, so entry and parameters are wave-uniform (scalar loads, no stack).
// ONE_MEDIUM: compiled with the short cut for a list that is one sphere-bounded medium (DScene::flat_one_medium), and
// without the walk of a small tree (RT_REF_STEPS: a scene that has one has no other tree, and its kernels are the ones
// without the short cut). The traversal kernel's instantiations are short of registers, each in its own way; only the one
// that scenes like the Book-2 scene get carries the short cut (elsewhere the same few lines cost 1 - 1.5 % of the kernel's
// time and won nothing).
template <bool ONE_MEDIUM = false, class XiFn, class RayFn>
RT_HD void rt_extend_flat(const DScene& sc, const RayFn& world_ray, double t_min, RtBest& best, const XiFn& xi, bool skip_media) {
    if (sc.world_flat == RT_REF_NONE) return;
    if (ONE_MEDIUM && sc.flat_one_medium != 0u) {
        if (!skip_media) {
            double t;
            RT_VISIT(1u + RT_REF_MEDIUM);
            if (rt_medium_t(sc, sc.flat_medium, world_ray(), t_min, best.t, xi, t, &sc.flat_sphere)) { best.t = t; best.ref = sc.flat_one_medium; best.aux = 0u; }
        }
        return;
    }
    Ray r = world_ray();
#pragma unroll 1
    for (uint32_t li = RT_REF_INDEX(sc.world_flat);; ++li) {
        const uint32_t e = sc.list_entries[li];
        if (e == RT_REF_NONE) break;
        if (!ONE_MEDIUM && RT_REF_TYPE(e) == RT_REF_STEPS) { li += rt_flat_steps(sc, li, r, world_ray, t_min, best, xi, skip_media); continue; }
        rt_flat_entry(sc, e, r, world_ray, t_min, best, xi, skip_media);
    }
}

template <class Stack, class XiFn, class RayFn>
RT_HD RtHitLite rt_extend(const DScene& sc, const RayFn& world_ray, double t_min, double t_max, Stack& st, const XiFn& xi,
                          bool skip_media) {
    RtBest best;
    best.t = t_max;
    best.ref = RT_REF_NONE;
    best.aux = 0u;
    rt_extend_flat(sc, world_ray, t_min, best, xi, skip_media);
    if (sc.world_tree != RT_REF_NONE)
        rt_traverse(sc, sc.world_tree, world_ray(), t_min, best, st, xi, skip_media, world_ray);
    RtHitLite h;
    h.t = best.t;
    h.ref = best.ref;
    h.aux = best.aux;
    return h;
}

// ---- hit reconstruction ------------------------------------------------------------------------------
struct HitRec {              // hittable.rs:9-18
    V3 p, normal;
    double t, u, v;
    uint32_t material, src_id;
    bool front_face;
};
// HitRecord::set_face_normal (hittable.rs:32-39)
RT_HD void rt_set_face_normal(HitRec& rec, V3 dir, V3 outward) {
    rec.front_face = v3_dot(dir, outward) < 0.0;
    rec.normal = rec.front_face ? outward : -outward;
}
// Sphere::get_sphere_uv (sphere.rs:24-30)
RT_HD void rt_sphere_uv(V3 p, double& u, double& v) {
    double theta = acos(-p.y);
    double phi = atan2(-p.z, p.x) + RT_PI;
    u = phi / (2.0 * RT_PI);
    v = theta / RT_PI;
}

// What the tables say about a hit primitive - material, handle, the instance directly above it - and, for a sphere, its
// record, asked for in the same breath.
struct RtHitMeta {
    uint32_t material, src_id;
    int32_t parent;
    RtSphere sphere;
};
RT_HD RtHitMeta rt_hit_meta(const DScene& sc, const RtHitLite& h) {
    const uint32_t type = RT_REF_TYPE(h.ref), idx = RT_REF_INDEX(h.ref);
    RtHitMeta g;
    g.sphere.cx = 0.0; g.sphere.cy = 0.0; g.sphere.cz = 0.0; g.sphere.r = 1.0;
#ifdef RT_BOUNDS_CHECK
    {   // a closest hit names a primitive or a medium by its own table: nothing else may arrive here
        const bool known = type == RT_REF_SPHERE ? RT_IN_RANGE(sc, spheres, idx, h.ref) : type == RT_REF_MSPHERE ? RT_IN_RANGE(sc, mspheres, idx, h.ref)
                         : type == RT_REF_RECT ? RT_IN_RANGE(sc, rects, idx, h.ref) : type == RT_REF_BOX ? RT_IN_RANGE(sc, boxes, idx, h.ref)
                         : type == RT_REF_MEDIUM ? RT_IN_RANGE(sc, media, idx, h.ref) : rt_in_range("hit reference type", type, 0u, h.ref);
        if (!known || (type == RT_REF_BOX && !rt_in_range("box face", h.aux, 6u, h.ref))) { g.material = 0u; g.src_id = 0xFFFFFFFFu; g.parent = RT_NO_PARENT; return g; }
    }
#endif
    if (type == RT_REF_SPHERE) g.sphere = sc.spheres[idx];
    if (type == RT_REF_SPHERE) { const RtMeta m = sc.sphere_meta[idx]; g.material = m.material; g.src_id = m.src_id; g.parent = m.parent_inst; }
    else if (type == RT_REF_MSPHERE) { const RtMeta m = sc.msphere_meta[idx]; g.material = m.material; g.src_id = m.src_id; g.parent = m.parent_inst; }
    else if (type == RT_REF_RECT) { const RtMeta m = sc.rect_meta[idx]; g.material = m.material; g.src_id = m.src_id; g.parent = m.parent_inst; }
    else if (type == RT_REF_BOX) { const RtMeta m = sc.box_meta[idx]; g.material = m.material; g.src_id = m.src_id; g.parent = m.parent_inst; }
    else { const RtMedium m = sc.media[idx]; g.material = m.material; g.src_id = m.src_id; g.parent = m.parent_inst; }
#ifdef RT_BOUNDS_CHECK
    if (!RT_IN_RANGE(sc, materials, g.material, h.ref)) g.material = 0u;
    if (g.parent >= 0 && !RT_IN_RANGE(sc, insts, g.parent, h.ref)) g.parent = RT_NO_PARENT;
#endif
    return g;
}
// The instances above a hit primitive, level 0 directly above it. Each level's record is read ONCE, on the way up the
// parent links, and the first RT_KEPT_LEVELS of them stay in registers: the way in, the way out and the directions
// in between use every level several times, and a read each time is a round trip to the cache each time, one after
// the other - that, not arithmetic, is what a wave with an instanced hit used to spend its time on.
#define RT_KEPT_LEVELS 2
struct RtHitLevels {
    int32_t chain[RT_MAX_CHAIN];
    RtChainOp kept[RT_KEPT_LEVELS];
};
RT_HD RtHitLevels rt_hit_levels(const DScene& sc, int32_t parent) {
    RtHitLevels lv;
    int32_t p = parent;
#pragma unroll
    for (int k = 0; k < RT_MAX_CHAIN; ++k) {
        lv.chain[k] = p;
        if (k < RT_KEPT_LEVELS) {
            lv.kept[k].a = 0.0; lv.kept[k].b = 0.0; lv.kept[k].c = 0.0; lv.kept[k].op = RT_OP_TRANSLATE; lv.kept[k].pad = 0u;
            if (p >= 0 && RT_IN_RANGE(sc, insts, p, 0u)) {
                const RtInst in = sc.insts[p];
                lv.kept[k].a = in.a; lv.kept[k].b = in.b; lv.kept[k].c = in.c; lv.kept[k].op = in.op;
                p = in.parent_inst;
            }
        } else if (p >= 0 && RT_IN_RANGE(sc, insts, p, 0u)) {
            p = sc.insts[p].parent_inst;
        }
    }
    return lv;
}

// Rebuilds the HitRecord the reference's nested hit() calls would have produced for the winning
// primitive: local hit in the primitive's frame, then each enclosing Translate (hittable.rs:75-78)
// and RotateY (hittable.rs:164-174, including its use of the already-overwritten x) outward.
// MovingSphere and ConstantMedium never write u,v (moving_sphere.rs:63-68, constant_medium.rs:59-65);
// they are 0 here. `want_uv` = false skips the acos/atan2 when no texture will read them.
// `g`, `lv`: rt_hit_meta and rt_hit_levels of the hit - a caller with reads of its own to make (the material) puts them
// between the two, so that they travel together with the first instance record instead of in front of it.
RT_HD HitRec rt_reconstruct(const DScene& sc, const Ray& wray, const RtHitLite& h, bool want_uv, const RtHitMeta& g, const RtHitLevels& lv) {
    HitRec rec;
    rec.t = h.t;
    rec.u = 0.0;
    rec.v = 0.0;
    rec.front_face = false;
    const uint32_t type = RT_REF_TYPE(h.ref), idx = RT_REF_INDEX(h.ref);
    rec.material = g.material; rec.src_id = g.src_id;
#ifdef RT_BOUNDS_CHECK
    if (g.src_id == 0xFFFFFFFFu) { rec.p = v3(0.0, 0.0, 0.0); rec.normal = v3(0.0, 0.0, 0.0); return rec; } // rt_hit_meta has reported it
#endif
    const RtSphere sphere = g.sphere;
    const int32_t (&chain)[RT_MAX_CHAIN] = lv.chain;
    const auto level = [&](int k) {
        if (k < RT_KEPT_LEVELS) return lv.kept[k];
        const RtInst in = sc.insts[chain[k]];
        RtChainOp o;
        o.a = in.a; o.b = in.b; o.c = in.c; o.op = in.op; o.pad = 0u;
        return o;
    };
    Ray r = wray;
#pragma unroll
    for (int k = RT_MAX_CHAIN - 1; k >= 0; --k)
        if (chain[k] >= 0) r = rt_apply_chain_op(level(k), r);

    rec.p = ray_at(r, h.t);
    if (type == RT_REF_SPHERE || type == RT_REF_MSPHERE) { // sphere.rs:57-61, moving_sphere.rs:63-66: (p - centre) / radius
        V3 centre;
        double radius;
        if (type == RT_REF_SPHERE) { centre = v3(sphere.cx, sphere.cy, sphere.cz); radius = sphere.r; }
        else { const RtMSphere s = sc.mspheres[idx]; centre = rt_msphere_center(s, r.tm); radius = s.r; }
        V3 outward = (rec.p - centre) / radius;
        rt_set_face_normal(rec, r.d, outward);
        if (want_uv && type == RT_REF_SPHERE) rt_sphere_uv(outward, rec.u, rec.v);
    } else if (type == RT_REF_RECT || type == RT_REF_BOX) {
        RtRect q;
        if (type == RT_REF_RECT) q = sc.rects[idx]; else q = rt_box_face(sc.boxes[idx], h.aux);
        double oa, da, ob, db;
        V3 outward;
        if (q.axis == 0u) { oa = r.o.x; da = r.d.x; ob = r.o.y; db = r.d.y; outward = v3(0.0, 0.0, 1.0); }
        else if (q.axis == 1u) { oa = r.o.x; da = r.d.x; ob = r.o.z; db = r.d.z; outward = v3(0.0, 1.0, 0.0); }
        else { oa = r.o.y; da = r.d.y; ob = r.o.z; db = r.d.z; outward = v3(1.0, 0.0, 0.0); }
        double a = oa + h.t * da, b = ob + h.t * db;
        rec.u = (a - q.a0) / (q.a1 - q.a0);   // arrect.rs:41-42
        rec.v = (b - q.b0) / (q.b1 - q.b0);
        rt_set_face_normal(rec, r.d, outward);
    } else { // ConstantMedium (constant_medium.rs:62-63)
        rec.normal = v3(1.0, 0.0, 0.0);
        rec.front_face = true;
    }

#pragma unroll
    for (int k = 0; k < RT_MAX_CHAIN; ++k) {
        if (chain[k] >= 0) {
            const RtChainOp in = level(k);
            // the direction of the ray below instance chain[k] (what hittable.rs:80, 176 pass to set_face_normal), worked
            // out again from the world ray rather than kept from the way in: four kept directions are 24 registers
            V3 dir = wray.d;
#pragma unroll
            for (int j = RT_MAX_CHAIN - 1; j >= k; --j)
                if (chain[j] >= 0) dir = rt_apply_chain_op_dir(level(j), dir);
            if ((in.op & RT_OP_MASK) == RT_OP_TRANSLATE) {
                rec.p = rec.p + v3(in.a, in.b, in.c);
                rt_set_face_normal(rec, dir, rec.normal);
            } else {
                const double s = in.a, c = in.b;
                V3 p_ = rec.p, n_ = rec.normal;
                if (sc.compat & RT_COMPAT_BOOK_ROTATE) {
                    p_.x = c * rec.p.x + s * rec.p.z;
                    p_.z = -s * rec.p.x + c * rec.p.z;
                    n_.x = c * rec.normal.x + s * rec.normal.z;
                    n_.z = -s * rec.normal.x + c * rec.normal.z;
                } else { // hittable.rs:167-171: the z lines read the x just written
                    p_.x = c * p_.x + s * p_.z;
                    p_.z = -s * p_.x + c * p_.z;
                    n_.x = c * n_.x + s * n_.z;
                    n_.z = -s * n_.x + c * n_.z;
                }
                rec.p = p_;
                rt_set_face_normal(rec, dir, n_);
            }
        }
    }
    return rec;
}
RT_HD HitRec rt_reconstruct(const DScene& sc, const Ray& wray, const RtHitLite& h, bool want_uv) {
    const RtHitMeta g = rt_hit_meta(sc, h);
    return rt_reconstruct(sc, wray, h, want_uv, g, rt_hit_levels(sc, g.parent));
}

// ---- textures ------------------------------------------------------------------------------------------
// Perlin::noise + trilinear_interp (perlin.rs:56-111). `tab` selects one table set.
RT_HD double rt_perlin_noise(const DScene& sc, uint32_t tab, V3 p) {
    const double* ranvec = sc.perlin_vec + (size_t)tab * RT_PERLIN_POINTS * 3;
    const int32_t* perm = sc.perlin_perm + (size_t)tab * RT_PERLIN_POINTS * 3;
    const double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    const double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    const double uu = u * u * (3.0 - 2.0 * u);
    const double vv = v * v * (3.0 - 2.0 * v);
    const double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ix = 255 & rt_sat_i32(fx + (double)i);
                const int iy = 255 & rt_sat_i32(fy + (double)j);
                const int iz = 255 & rt_sat_i32(fz + (double)k);
                const int g = perm[ix] ^ perm[RT_PERLIN_POINTS + iy] ^ perm[2 * RT_PERLIN_POINTS + iz];
                const V3 c = v3(ranvec[3 * g], ranvec[3 * g + 1], ranvec[3 * g + 2]);
                const double di = (double)i, dj = (double)j, dk = (double)k;
                const V3 weight_v = v3(u - di, v - dj, w - dk);
                accum += (di * uu + (1.0 - di) * (1.0 - uu)) * (dj * vv + (1.0 - dj) * (1.0 - vv)) *
                         (dk * ww + (1.0 - dk) * (1.0 - ww)) * v3_dot(c, weight_v);
            }
    return accum;
}
// Perlin::turb (perlin.rs:113-125)
RT_HD double rt_perlin_turb(const DScene& sc, uint32_t tab, V3 p, int depth) {
    double accum = 0.0, weight = 1.0;
    V3 temp_p = p;
#pragma unroll 1
    for (int i = 0; i < depth; ++i) {
        accum += weight * rt_perlin_noise(sc, tab, temp_p);
        weight *= 0.5;
        temp_p = temp_p * 2.0;
    }
    return fabs(accum);
}
// Rust `f64 as usize` (saturating, NaN -> 0), texture.rs:110-111
RT_HD uint64_t rt_sat_usize(double v) {
    if (!(v == v) || v <= 0.0) return 0u;
    if (v >= 18446744073709551615.0) return UINT64_MAX;
    return (uint64_t)v;
}
// NoiseTexture::value (texture.rs:73-79)
RT_HD V3 rt_tex_noise_value(const DScene& sc, const RtTexture& t, V3 p) {
    double s = 0.5 * (1.0 + sin(t.p[0] * p.z + 10.0 * rt_perlin_turb(sc, t.a, p, 7)));
    return v3(s, s, s);
}
// Texture::value for SolidColor / CheckerTexture / NoiseTexture / ImageTexture
// (texture.rs:23-27, 46-54, 73-79, 105-127).
// KINDS: which kinds of computed texture the scene holds at all, when the caller knows (bit RT_TEX_CHECKER, RT_TEX_NOISE,
// RT_TEX_IMAGE): the code for the others is left out - a kernel that shades a checker of two colours then carries neither
// the Perlin noise nor the sphere's (u, v) for an image, and needs half the registers.
#define RT_TEX_KINDS_ALL ((1u << RT_TEX_CHECKER) | (1u << RT_TEX_NOISE) | (1u << RT_TEX_IMAGE))
template <uint32_t KINDS = RT_TEX_KINDS_ALL>
RT_HD V3 rt_tex_value(const DScene& sc, uint32_t tex, double u, double v, V3 p) {
    RtTexture t = sc.textures[tex];
    if (KINDS & (1u << RT_TEX_CHECKER)) {
#pragma unroll 1
        for (int guard = 0; guard < 8 && t.type == RT_TEX_CHECKER; ++guard) {
            double sines = sin(10.0 * p.x) * sin(10.0 * p.y) * sin(10.0 * p.z);
            t = sc.textures[sines < 0.0 ? t.b : t.a];
        }
    }
    if (t.type == RT_TEX_SOLID) return v3(t.p[0], t.p[1], t.p[2]);
    if ((KINDS & (1u << RT_TEX_NOISE)) && t.type == RT_TEX_NOISE) return rt_tex_noise_value(sc, t, p);
    if ((KINDS & (1u << RT_TEX_IMAGE)) && t.type == RT_TEX_IMAGE) {
        const uint64_t width = t.b, height = t.c;
        double u_ = rt_clamp(u, 0.0, 1.0);
        double v_ = 1.0 - rt_clamp(v, 0.0, 1.0);
        uint64_t i = rt_sat_usize(u_ * (double)width);
        uint64_t j = rt_sat_usize(v_ * (double)height);
        if (i >= width) i = width - 1u;
        if (j >= height) j = height - 1u;
        const double color_scale = 1.0 / 255.0;
        const uint8_t* px = sc.texels + t.a + j * (3u * width) + i * 3u;
        return v3(color_scale * (double)px[0], color_scale * (double)px[1], color_scale * (double)px[2]);
    }
    return v3(0.0, 0.0, 0.0);
}
RT_HD bool rt_tex_reads_uv(const DScene& sc, uint32_t tex) {
    // an image anywhere below a checker still needs uv; checkers are rare, so be conservative
    const uint32_t ty = sc.textures[tex].type;
    return ty == RT_TEX_IMAGE || ty == RT_TEX_CHECKER;
}
RT_HD bool rt_material_reads_uv(const DScene& sc, const RtMaterial& m) {
    if (m.type == RT_MAT_LAMBERTIAN || m.type == RT_MAT_ISOTROPIC || m.type == RT_MAT_DIFFUSE_LIGHT)
        return rt_tex_reads_uv(sc, m.tex);
    return false;
}

// ---- materials -----------------------------------------------------------------------------------------
RT_HD uint32_t rt_bucket_of(uint32_t mat_type, uint32_t tex_type) {
    const uint32_t tex = tex_type == RT_TEX_SOLID ? 0u : 1u;
    switch (mat_type) {
    case RT_MAT_LAMBERTIAN: return tex_type == RT_TEX_NOISE ? (uint32_t)RT_BUCKET_LAMBERTIAN_NOISE : RT_BUCKET_LAMBERTIAN_SOLID + tex;
    case RT_MAT_METAL: return RT_BUCKET_METAL;
    case RT_MAT_DIELECTRIC: return RT_BUCKET_DIELECTRIC;
    case RT_MAT_DIFFUSE_LIGHT: return RT_BUCKET_LIGHT_SOLID + tex;
    default: return RT_BUCKET_ISOTROPIC_SOLID + tex;
    }
}
constexpr uint32_t rt_bucket_material(uint32_t bucket) {
    return bucket <= RT_BUCKET_LAMBERTIAN_TEX || bucket == RT_BUCKET_LAMBERTIAN_NOISE ? RT_MAT_LAMBERTIAN : bucket == RT_BUCKET_METAL ? RT_MAT_METAL
         : bucket == RT_BUCKET_DIELECTRIC ? RT_MAT_DIELECTRIC : bucket <= RT_BUCKET_LIGHT_TEX ? RT_MAT_DIFFUSE_LIGHT : RT_MAT_ISOTROPIC;
}
constexpr bool rt_bucket_solid(uint32_t bucket) {
    return bucket == RT_BUCKET_LAMBERTIAN_SOLID || bucket == RT_BUCKET_LIGHT_SOLID || bucket == RT_BUCKET_ISOTROPIC_SOLID;
}
// SolidColor::value (texture.rs:23-27) when the caller knows the texture is one
RT_HD V3 rt_tex_solid_value(const DScene& sc, uint32_t tex) {
    const RtTexture t = sc.textures[tex];
    return v3(t.p[0], t.p[1], t.p[2]);
}
struct ShadeOut {
    V3 emitted;      // Material::emitted (material.rs:19-21, 201-203)
    V3 attenuation;
    Ray scattered;
    bool did_scatter;
};
// Dielectric::reflectance (material.rs:109-113)
RT_HD double rt_reflectance(double cosine, double ref_idx) {
    double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
    r0 = r0 * r0;
    return r0 + (1.0 - r0) * rt_powi5(1.0 - cosine);
}
// emitted + scatter for the material of `rec`, the two calls ray_color makes at main.rs:62-66.
// MAT = the material type when the caller knows it at compile time (the per-material shade kernels), which prunes
// the other four bodies; RT_MAT_COUNT = decide from m.type at run time. SOLID / NOISE = the caller knows the texture is a
// SolidColor / a NoiseTexture.
template <uint32_t MAT, bool SOLID = false, bool NOISE = false, uint32_t KINDS = RT_TEX_KINDS_ALL>
RT_HD ShadeOut rt_shade_as(const DScene& sc, const RtMaterial& m, const Ray& r_in, const HitRec& rec, const Sampler& smp) {
    ShadeOut o;
    o.emitted = v3(0.0, 0.0, 0.0);
    o.attenuation = v3(0.0, 0.0, 0.0);
    o.scattered.o = v3(0.0, 0.0, 0.0);
    o.scattered.d = v3(0.0, 0.0, 0.0);
    o.scattered.tm = 0.0;
    o.did_scatter = false;
    const uint32_t type = MAT < RT_MAT_COUNT ? MAT : m.type;
    // What several materials need is computed once, in front of the switch: lanes of a wave that hold different
    // materials then run one rejection loop together instead of one per material in turn.
    V3 in_sphere = v3(0.0, 0.0, 0.0), unit_in = v3(0.0, 0.0, 0.0);
    if (type == RT_MAT_LAMBERTIAN || type == RT_MAT_METAL || type == RT_MAT_ISOTROPIC) in_sphere = rt_random_in_unit_sphere(smp);
    if (type == RT_MAT_METAL || type == RT_MAT_DIELECTRIC) unit_in = v3_unit(r_in.d);
    V3 colour = v3(0.0, 0.0, 0.0); // Texture::value of the material's texture (Lambertian / Isotropic albedo, DiffuseLight emit)
    if (type == RT_MAT_LAMBERTIAN || type == RT_MAT_DIFFUSE_LIGHT || type == RT_MAT_ISOTROPIC)
        colour = SOLID ? v3(m.p[1], m.p[2], m.p[3]) : NOISE ? rt_tex_noise_value(sc, sc.textures[m.tex], rec.p) : rt_tex_value<KINDS>(sc, m.tex, rec.u, rec.v, rec.p);
    switch (type) {
    case RT_MAT_LAMBERTIAN: { // material.rs:41-59; random_unit_vector = unit_vector(random_in_unit_sphere), vec3.rs:82-84
        V3 scatter_direction = rec.normal + v3_unit(in_sphere);
        if (v3_near_zero(scatter_direction)) scatter_direction = rec.normal;
        o.scattered.o = rec.p; o.scattered.d = scatter_direction; o.scattered.tm = r_in.tm;
        o.attenuation = colour;
        o.did_scatter = true;
        break;
    }
    case RT_MAT_METAL: { // material.rs:80-97
        V3 reflected = v3_reflect(unit_in, rec.normal);
        o.scattered.o = rec.p;
        o.scattered.d = reflected + in_sphere * m.p[3];
        o.scattered.tm = r_in.tm;
        o.attenuation = v3(m.p[0], m.p[1], m.p[2]);
        o.did_scatter = v3_dot(o.scattered.d, rec.normal) > 0.0;
        break;
    }
    case RT_MAT_DIELECTRIC: { // material.rs:116-151
        o.attenuation = v3(1.0, 1.0, 1.0);
        const double ir = m.p[0];
        double refraction_ratio = rec.front_face ? 1.0 / ir : ir;
        V3 unit_direction = unit_in;
        double cos_theta = v3_dot(-unit_direction, rec.normal);
        if (cos_theta > 1.0) cos_theta = 1.0;
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        V3 direction;
        if (cannot_refract || rt_reflectance(cos_theta, refraction_ratio) > sampler_draw1(smp, RT_STREAM_DIELECTRIC, 0u))
            direction = v3_reflect(unit_direction, rec.normal);
        else
            direction = v3_refract(unit_direction, rec.normal, refraction_ratio);
        o.scattered.o = rec.p; o.scattered.d = direction; o.scattered.tm = r_in.tm;
        o.did_scatter = true;
        break;
    }
    case RT_MAT_DIFFUSE_LIGHT: { // material.rs:191-204: both faces emit
        o.emitted = colour * m.p[0];
        break;
    }
    case RT_MAT_ISOTROPIC: { // material.rs:219-231: direction is not normalised
        o.scattered.o = rec.p; o.scattered.d = in_sphere; o.scattered.tm = r_in.tm;
        o.attenuation = colour;
        o.did_scatter = true;
        break;
    }
    default: break;
    }
    return o;
}

RT_HD ShadeOut rt_shade(const DScene& sc, const RtMaterial& m, const Ray& r_in, const HitRec& rec, const Sampler& smp) {
    return rt_shade_as<RT_MAT_COUNT>(sc, m, r_in, rec, smp);
}

// ---- camera --------------------------------------------------------------------------------------------
// Vec3::random_in_unit_disk (vec3.rs:102-110) returns (x, 0, z); Camera::get_ray reads .x and .y
// (camera.rs:57-58), so only the horizontal lens offset survives. Book mode samples (x, y, 0).
RT_HD Ray rt_camera_get_ray(const RtCameraDev& cam, double s, double t, const Sampler& smp, uint32_t compat) {
    V3 p = v3(0.0, 0.0, 0.0);
#pragma unroll 1
    for (uint32_t k = 0; k < RT_REJECTION_CAP; ++k) {
        double a, b;
        sampler_draw2(smp, RT_STREAM_LENS, k, a, b);
        if (compat & RT_COMPAT_BOOK_LENS) p = v3(rt_range(a, -1.0, 1.0), rt_range(b, -1.0, 1.0), 0.0);
        else p = v3(rt_range(a, -1.0, 1.0), 0.0, rt_range(b, -1.0, 1.0));
        if (v3_sqlen(p) < 1.0) break;
    }
    const V3 rd = p * cam.lens_radius;
    const V3 cu = v3(cam.u[0], cam.u[1], cam.u[2]), cv = v3(cam.v[0], cam.v[1], cam.v[2]);
    const V3 origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    const V3 llc = v3(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    const V3 hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    const V3 ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    const V3 offset = cu * rd.x + cv * rd.y;
    Ray r;
    r.o = origin + offset;
    r.d = llc + hor * s + ver * t - origin - offset;
    r.tm = rt_range(sampler_draw1(smp, RT_STREAM_TIME, 0u), cam.time0, cam.time1);
    return r;
}
// The sample loop body up to the ray_color call (main.rs:162-164) for sample `smp.sample` of pixel (x, y).
RT_HD Ray rt_primary_ray(const RtCameraDev& cam, uint32_t x, uint32_t y, uint32_t width, uint32_t height,
                         const Sampler& smp, uint32_t compat) {
    double jx, jy;
    sampler_draw2(smp, RT_STREAM_JITTER, 0u, jx, jy);
    const double u_ = ((double)x + jx) / ((double)width - 1.0);
    const double v_ = ((double)y + jy) / ((double)height - 1.0);
    return rt_camera_get_ray(cam, u_, v_, smp, compat);
}
// Miss colour: the constant `background` of main.rs:56-58, or (mode 1) the sky gradient the older
// snapshot of the renderer uses (/root/reference/main.rs:59-61).
RT_HD V3 rt_background(const double bg[3], uint32_t mode, const Ray& r) {
    if (mode == 1u) {
        V3 unit_direction = v3_unit(r.d);
        double t = (unit_direction.y + 1.0) * 0.5;
        return v3(1.0, 1.0, 1.0) * (1.0 - t) + v3(0.5, 0.7, 1.0) * t;
    }
    return v3(bg[0], bg[1], bg[2]);
}
// Final pixel conversion (main.rs:172-189)
RT_HD uint8_t rt_tonemap_channel(double sum, double spp) {
    double scale = 1.0 / spp;
    double c = sqrt(scale * sum);
    c = rt_clamp(c, 0.0, 0.999);
    c = c * 255.999;
    // `as i64` then `as u8`: saturating, NaN -> 0; the clamp bounds it to 0..255 already
    int64_t q = (c == c) ? (int64_t)c : 0;
    return (uint8_t)q;
}
