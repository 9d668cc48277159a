// f64 vector math and the counter-based RNG shared by host code and HIP kernels.
//
// Arithmetic follows the reference's Vec3 (raytracer/src/vec3.rs:13-257) operation by operation and
// in the same association order, and this translation unit is always compiled with
// -ffp-contract=off, so +,-,*,/ and sqrt give the same bits on the host and on gfx950 (both are
// IEEE correctly rounded); only sin/cos/acos/atan2/log may differ in the last ulp between libms.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ inline
#else
#define RT_HD inline
#endif

#define RT_PI 3.14159265358979323846 /* std::f64::consts::PI, rtweekend.rs:2 */

struct V3 {
    double x, y, z;
};
RT_HD V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   // vec3.rs:113-123
RT_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   // vec3.rs:157-167
RT_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   // vec3.rs:201-211
RT_HD V3 operator*(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }     // vec3.rs:213-223
RT_HD V3 operator/(V3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }     // vec3.rs:235-245
RT_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }                        // vec3.rs:247-257
RT_HD double v3_sqlen(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }        // vec3.rs:30-32
RT_HD double v3_len(V3 a) { return sqrt(v3_sqlen(a)); }                          // vec3.rs:34-36
RT_HD double v3_dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }    // vec3.rs:38-40
RT_HD V3 v3_cross(V3 u, V3 v) {                                                  // vec3.rs:42-48
    return v3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
// vec3.rs:50-55 panics on a zero vector; here the defined result is the zero vector (callers that can
// reach it — Lambertian — then take their near_zero branch).
RT_HD V3 v3_unit(V3 a) {
    double l = v3_len(a);
    if (l == 0.0) return v3(0.0, 0.0, 0.0);
    return a / l;
}
RT_HD bool v3_near_zero(V3 a) {                                                  // vec3.rs:86-89
    const double s = 1e-8;
    return fabs(a.x) < s && fabs(a.y) < s && fabs(a.z) < s;
}
RT_HD V3 v3_reflect(V3 v, V3 n) { return v - n * v3_dot(v, n) * 2.0; }           // vec3.rs:91-93
RT_HD V3 v3_refract(V3 uv, V3 n, double etai_over_etat) {                        // vec3.rs:95-100
    double cos_theta = v3_dot(-uv, n);
    V3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    V3 r_out_parallel = (-n) * sqrt(fabs(1.0 - v3_sqlen(r_out_perp)));
    return r_out_perp + r_out_parallel;
}
RT_HD double rt_clamp(double x, double lo, double hi) {                          // rtweekend.rs:9-17
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
RT_HD double rt_fmin(double a, double b) { return a <= b ? a : b; }              // rtweekend.rs:29-34
RT_HD double rt_fmax(double a, double b) { return a >= b ? a : b; }              // rtweekend.rs:36-41
// x.powi(5) as LLVM expands it: x * ((x*x)*(x*x))  (material.rs:112)
RT_HD double rt_powi5(double x) { double x2 = x * x; double x4 = x2 * x2; return x4 * x; }
// Rust `f64 as i32`: saturating, NaN -> 0 (perlin.rs:73-75)
RT_HD int32_t rt_sat_i32(double v) {
    if (!(v == v)) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}

struct Ray {
    V3 o, d;
    double tm;
};
RT_HD V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }                // ray.rs:36-38

// ---- Philox-4x32-10 (Salmon et al., SC'11) ----------------------------------------------------------
// The reference draws from rand::thread_rng() (rtweekend.rs:19-27, vec3.rs:57-110), an OS-seeded
// stream nobody can replay; only its distribution contract (uniform [0,1), [a,b)) carries over. Here
// every draw site owns a counter slot, so a sample's path does not depend on scheduling or GPU count:
//   key     = (pixel index y*W+x, seed low word)
//   counter = (sample index, bounce<<16 | stream id, attempt, seed high word)
struct Philox4 {
    uint32_t v[4];
};
RT_HD void rt_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    hi = (uint32_t)(p >> 32);
    lo = (uint32_t)p;
}
RT_HD Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        rt_mulhilo(0xD2511F53u, c0, hi0, lo0);
        rt_mulhilo(0xCD9E8D57u, c2, hi1, lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    Philox4 o;
    o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
    return o;
}
// 53-bit uniform in [0,1) from two 32-bit words (same resolution as rand's gen::<f64>(), main.rs:162)
RT_HD double rt_u01(uint32_t hi, uint32_t lo) {
    uint64_t w = ((uint64_t)hi << 32) | (uint64_t)lo;
    return (double)(w >> 11) * 0x1.0p-53;
}

// Identifies one camera sample; `bounce` advances along the path.
struct Sampler {
    uint32_t pixel, sample, bounce, seed_lo, seed_hi;
};
RT_HD Sampler sampler_make(uint32_t pixel, uint32_t sample, uint32_t bounce, uint64_t seed) {
    Sampler s;
    s.pixel = pixel; s.sample = sample; s.bounce = bounce;
    s.seed_lo = (uint32_t)seed; s.seed_hi = (uint32_t)(seed >> 32);
    return s;
}
// two independent uniforms for (stream, attempt)
RT_HD void sampler_draw2(const Sampler& s, uint32_t stream, uint32_t attempt, double& a, double& b) {
    Philox4 p = philox4x32_10(s.sample, (s.bounce << 16) | stream, attempt, s.seed_hi, s.pixel, s.seed_lo);
    a = rt_u01(p.v[0], p.v[1]);
    b = rt_u01(p.v[2], p.v[3]);
}
RT_HD double sampler_draw1(const Sampler& s, uint32_t stream, uint32_t attempt) {
    double a, b;
    sampler_draw2(s, stream, attempt, a, b);
    return a;
}
// random_double(low, high) (rtweekend.rs:19-22): uniform on [low, high)
RT_HD double rt_range(double u, double lo, double hi) { return lo + (hi - lo) * u; }

// The draws of a rejection loop whose length is not known in advance. A wave runs such a loop until its slowest lane
// accepts (about six rounds for a 52 % acceptance rate), so the cost per round matters more than anything else in the
// shading kernels. One Philox block (stream, slot 0) seeds a xoshiro128++ generator (Blackman & Vigna, 32-bit
// state words: add, rotate, shift, xor only), which then supplies the 53-bit uniforms: a round costs a third of what
// two Philox blocks cost, and the sequence is still a pure function of (pixel, sample, bounce, stream, seed).
struct SubStream {
    uint32_t s0, s1, s2, s3;
};
RT_HD SubStream substream_make(const Sampler& s, uint32_t stream) {
    const Philox4 p = philox4x32_10(s.sample, (s.bounce << 16) | stream, 0u, s.seed_hi, s.pixel, s.seed_lo);
    SubStream g;
    g.s0 = p.v[0]; g.s1 = p.v[1]; g.s2 = p.v[2]; g.s3 = p.v[3];
    if ((g.s0 | g.s1 | g.s2 | g.s3) == 0u) g.s0 = 1u; // the one state xoshiro cannot leave
    return g;
}
RT_HD uint32_t substream_next(SubStream& g) {
    const uint32_t a = g.s0 + g.s3;
    const uint32_t result = ((a << 7) | (a >> 25)) + g.s0;
    const uint32_t t = g.s1 << 9;
    g.s2 ^= g.s0;
    g.s3 ^= g.s1;
    g.s1 ^= g.s2;
    g.s0 ^= g.s3;
    g.s2 ^= t;
    g.s3 = (g.s3 << 11) | (g.s3 >> 21);
    return result;
}
RT_HD double substream_u01(SubStream& g) {
    const uint32_t hi = substream_next(g);
    const uint32_t lo = substream_next(g);
    return rt_u01(hi, lo);
}
// Vec3::random_in_unit_sphere (vec3.rs:73-80): rejection from [-1,1)^3; x, y, z of each attempt are the next three
// uniforms of the scatter sub-stream.
RT_HD V3 rt_random_in_unit_sphere(const Sampler& s) {
    V3 p = v3(0.0, 0.0, 0.0);
    SubStream g = substream_make(s, 3u /*RT_STREAM_SCATTER*/);
    for (uint32_t k = 0; k < 4096u; ++k) {
        const double ux = substream_u01(g), uy = substream_u01(g), uz = substream_u01(g);
        p = v3(rt_range(ux, -1.0, 1.0), rt_range(uy, -1.0, 1.0), rt_range(uz, -1.0, 1.0));
        if (v3_sqlen(p) < 1.0) return p;
    }
    return p;
}
RT_HD V3 rt_random_unit_vector(const Sampler& s) { return v3_unit(rt_random_in_unit_sphere(s)); } // vec3.rs:82-84

// The reference's own Vec3 assertions (raytracer/src/vec3.rs:259-399, a commented-out test module: the only known answers
// the reference holds), replayed on V3. `one` must be 1.0 and come from where the compiler cannot see it (a kernel
// argument, a caller's variable), so that the arithmetic runs on the machine under test instead of being folded away.
// Returns a bit per failed assertion, in the order of the module; 0 = all twenty hold.
RT_HD uint32_t rt_vec3_kat(double one) {
    uint32_t failed = 0u, k = 0u;
    const auto same = [](V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; };
    const auto expect = [&](bool ok) { if (!ok) failed |= 1u << k; ++k; };
    const auto mk = [&](double x, double y, double z) { return v3(x * one, y * one, z * one); };
    const auto splat = [&](double s) { return v3(s * one, s * one, s * one); };       // Vec3 (+|-) f64: vec3.rs:125-155, 169-199
    expect(same(mk(1.0, 2.0, 3.0), v3(1.0, 2.0, 3.0)));                                                     // test_new
    expect(same(mk(1.0, 0.0, -1.0) + mk(2.0, 4.0, 6.0), v3(3.0, 4.0, 5.0)));                                // test_add
    { V3 x = mk(1.0, 0.0, -1.0); x = x + mk(2.0, 4.0, 6.0); expect(same(x, v3(3.0, 4.0, 5.0))); }           // test_add_assign
    expect(same(mk(1.0, 0.0, -1.0) + splat(233.0), v3(234.0, 233.0, 232.0)));                               // test_add_f64
    { V3 x = mk(1.0, 0.0, -1.0); x = x + splat(233.0); expect(same(x, v3(234.0, 233.0, 232.0))); }          // test_add_assign_f64
    expect(same(mk(1.0, 0.0, -1.0) - mk(2.0, 4.0, 6.0), v3(-1.0, -4.0, -7.0)));                             // test_sub
    { V3 x = mk(1.0, 0.0, -1.0); x = x - mk(2.0, 4.0, 6.0); expect(same(x, v3(-1.0, -4.0, -7.0))); }        // test_sub_assign
    expect(same(mk(1.0, 0.0, -1.0) - splat(1.0), v3(0.0, -1.0, -2.0)));                                     // test_sub_f64
    { V3 x = mk(1.0, 0.0, -1.0); x = x - splat(1.0); expect(same(x, v3(0.0, -1.0, -2.0))); }                // test_sub_assign_f64
    { V3 x = mk(1.0, 0.0, -1.0); x = x * (2.0 * one); expect(same(x, v3(2.0, 0.0, -2.0))); }                // test_mul_assign
    expect(same(mk(1.0, 0.0, -1.0) * (1.0 * one), v3(1.0, 0.0, -1.0)));                                     // test_mul_f64
    expect(same(mk(1.0, -2.0, 0.0) / (2.0 * one), v3(0.5, -1.0, 0.0)));                                     // test_div
    expect(same(v3_cross(mk(1.0, 2.0, 3.0), mk(2.0, 3.0, 4.0)), v3(8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0)));      // test_cross
    expect(same(-mk(1.0, -2.0, 3.0), v3(-1.0, 2.0, -3.0)));                                                 // test_neg
    expect(v3_sqlen(mk(1.0, 2.0, 3.0)) == 14.0);                                                            // test_squared_length
    expect(v3_len(mk(3.0, 4.0, 5.0)) == sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0));                           // test_length
    expect(same(v3_unit(mk(233.0, 0.0, 0.0)), v3(1.0, 0.0, 0.0)));                                          // test_unit, first
    expect(same(v3_unit(mk(-233.0, 0.0, 0.0)), v3(-1.0, 0.0, 0.0)));                                        // test_unit, second
    // test_unit_panic: the reference panics on a zero vector ("div 0!", vec3.rs:52); nothing may trap on the device, so the
    // defined result here is the zero vector - and the condition the panic tests is still what decides it
    expect(v3_len(mk(0.0, 0.0, 0.0)) == 0.0);
    expect(same(v3_unit(mk(0.0, 0.0, 0.0)), v3(0.0, 0.0, 0.0)));
    return failed;
}
