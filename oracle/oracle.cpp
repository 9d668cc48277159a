// TEST INFRASTRUCTURE — CPU oracle. Not part of the product: only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py may load this library.
//
// A C++ f64 restatement of the hot path of Winter-Is-Coming-Stark/Raytracer, written by following the
// reference's Rust sources function by function and keeping their shape: an object graph of
// shared_ptr<dyn> nodes with virtual hit()/scatter()/value(), a recursive ray_color, a linear
// HittableList, an always-left-then-right BvhNode, and every quirk of the originals (cited inline).
// It shares no code with the product; its only input is the replayable constructor log the product's
// scene builder emits, from which it rebuilds its own graph and its own BVHs.
//
// PARITY PINNING: the reference has no live tests, no golden images and cannot be built here (no Rust
// toolchain). The only known answers it holds are 20 commented-out Vec3 assertions
// (raytracer/src/vec3.rs:259-399); orc_vec3_kat() below replays them. Everything else is "parity
// unpinned": derived from reading the source, cross-checked by analytic cases in tests/.
//
// Randomness: the reference draws from rand::thread_rng() (OS seeded, not replayable). The oracle draws
// from Philox-4x32-10 with one counter slot per draw site:
//     key = (pixel, seed_lo); counter = (sample, bounce << 16 | stream, attempt, seed_hi)
//     streams: 0 pixel jitter, 1 lens disk, 2 shutter time, 3 scatter direction (attempt k -> slots 2k,
//     2k+1), 4 dielectric reflect/refract, 16+m free-flight draw of ConstantMedium number m.
// Each call yields two 53-bit uniforms (words 0,1 and words 2,3).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

#include "../include/rt.h"

namespace orc {

const double INFINITY_ = std::numeric_limits<double>::infinity(); // rtweekend.rs:1
const double PI = 3.14159265358979323846;                        // rtweekend.rs:2

// ---------------------------------------------------------------------------------------------------------
// Philox-4x32-10, written from the published algorithm (Salmon, Moraes, Dror, Shaw, SC'11).
// ---------------------------------------------------------------------------------------------------------
struct U4 { uint32_t a, b, c, d; };
static U4 philox(U4 ctr, uint32_t k0, uint32_t k1) {
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * ctr.a;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * ctr.c;
        U4 n;
        n.a = (uint32_t)(p1 >> 32) ^ ctr.b ^ k0;
        n.b = (uint32_t)p1;
        n.c = (uint32_t)(p0 >> 32) ^ ctr.d ^ k1;
        n.d = (uint32_t)p0;
        ctr = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return ctr;
}
static double to_unit(uint32_t hi, uint32_t lo) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

enum Stream : uint32_t { S_JITTER = 0, S_LENS = 1, S_TIME = 2, S_SCATTER = 3, S_DIELECTRIC = 4, S_MEDIUM0 = 16 };

struct Counters { // per-thread visit counts (what the reference's traversal touches)
    uint64_t nodes = 0, spheres = 0, msphere = 0, rects = 0, media = 0, segments = 0, samples = 0;
};

// The draw context of one camera sample: what thread_rng() is to the reference.
struct Rng {
    uint32_t pixel = 0, sample = 0, bounce = 0;
    uint64_t seed = 0;
    const double* medium_xi = nullptr; // injected free-flight uniforms (first-hit queries)
    bool skip_media = false;
    uint32_t compat = 0;
    Counters* cnt = nullptr;
    void draw2(uint32_t stream, uint32_t attempt, double& x, double& y) const {
        U4 c{sample, (bounce << 16) | stream, attempt, (uint32_t)(seed >> 32)};
        U4 o = philox(c, pixel, (uint32_t)seed);
        x = to_unit(o.a, o.b);
        y = to_unit(o.c, o.d);
    }
    double draw1(uint32_t stream, uint32_t attempt) const { double x, y; draw2(stream, attempt, x, y); return x; }
};
static double in_range(double u, double low, double high) { return low + (high - low) * u; } // gen_range(low..high)

// ---------------------------------------------------------------------------------------------------------
// vec3.rs
// ---------------------------------------------------------------------------------------------------------
struct Vec3 {
    double x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
    static Vec3 ones() { return Vec3(1.0, 1.0, 1.0); }
    static Vec3 zero() { return Vec3(0.0, 0.0, 0.0); }
    double squared_length() const { return x * x + y * y + z * z; }
    double length() const { return std::sqrt(squared_length()); }
    static double dot(Vec3 u, Vec3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
    static Vec3 cross(Vec3 u, Vec3 v) { return Vec3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x); }
    Vec3 unit() const; // vec3.rs:50-55 panics on zero length; see definition
    bool near_zero() const { const double s = 1e-8; return std::fabs(x) < s && std::fabs(y) < s && std::fabs(z) < s; }
    bool operator==(const Vec3& o) const { return x == o.x && y == o.y && z == o.z; }
};
typedef Vec3 Color;
typedef Vec3 Point3;
static Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static Vec3 operator+(Vec3 a, double b) { return Vec3(a.x + b, a.y + b, a.z + b); }
static Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static Vec3 operator-(Vec3 a, double b) { return Vec3(a.x - b, a.y - b, a.z - b); }
static Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static Vec3 operator*(Vec3 a, double b) { return Vec3(a.x * b, a.y * b, a.z * b); }
static Vec3 operator/(Vec3 a, double b) { return Vec3(a.x / b, a.y / b, a.z / b); }
static Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
struct Panic { std::string what; };
Vec3 Vec3::unit() const {
    // The reference panics ("div 0!"). A render cannot unwind per sample, so the oracle defines the result
    // as the zero vector (the product does the same); orc_vec3_kat checks the panic condition itself.
    if (length() == 0.0) return Vec3::zero();
    return *this / length();
}
static Vec3 reflect(Vec3 v, Vec3 n) { return v - n * Vec3::dot(v, n) * 2.0; } // vec3.rs:91-93
static Vec3 refract(Vec3 uv, Vec3 n, double etai_over_etat) {                  // vec3.rs:95-100
    double cos_theta = Vec3::dot(-uv, n);
    Vec3 r_out_perp = (uv + n * cos_theta) * etai_over_etat;
    Vec3 r_out_parallel = -n * std::sqrt(std::fabs(1.0 - r_out_perp.squared_length()));
    return r_out_perp + r_out_parallel;
}
// Draws of the scatter rejection loop: xoshiro128++ seeded by the Philox block (S_SCATTER, slot 0) of the sample's
// bounce; every uniform is 53 bits from two successive outputs (first = high word).
struct Xoshiro128pp {
    uint32_t s[4];
    static uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
    uint32_t next() {
        const uint32_t result = rotl(s[0] + s[3], 7) + s[0];
        const uint32_t t = s[1] << 9;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 11);
        return result;
    }
    double unit() { uint32_t hi = next(); uint32_t lo = next(); return to_unit(hi, lo); }
};
static Vec3 random_in_unit_sphere(const Rng& rng) { // vec3.rs:73-80
    U4 c{rng.sample, (rng.bounce << 16) | S_SCATTER, 0u, (uint32_t)(rng.seed >> 32)};
    U4 o = philox(c, rng.pixel, (uint32_t)rng.seed);
    Xoshiro128pp g{{o.a, o.b, o.c, o.d}};
    if ((g.s[0] | g.s[1] | g.s[2] | g.s[3]) == 0u) g.s[0] = 1u;
    Vec3 p;
    for (uint32_t k = 0; k < 4096; ++k) {
        double a = g.unit(), b = g.unit(), c2 = g.unit();
        p = Vec3(in_range(a, -1.0, 1.0), in_range(b, -1.0, 1.0), in_range(c2, -1.0, 1.0));
        if (p.squared_length() < 1.0) return p;
    }
    return p;
}
static Vec3 random_unit_vector(const Rng& rng) { return random_in_unit_sphere(rng).unit(); } // vec3.rs:82-84
static Vec3 random_in_unit_disk(const Rng& rng) {                                            // vec3.rs:102-110
    Vec3 p;
    for (uint32_t k = 0; k < 4096; ++k) {
        double a, b;
        rng.draw2(S_LENS, k, a, b);
        if (rng.compat & 2u) p = Vec3(in_range(a, -1.0, 1.0), in_range(b, -1.0, 1.0), 0.0); // book: XY disk
        else p = Vec3(in_range(a, -1.0, 1.0), 0.0, in_range(b, -1.0, 1.0));                  // reference: (x, 0, z)
        if (p.squared_length() < 1.0) return p;
    }
    return p;
}

// rtweekend.rs
static double degrees_to_radians(double degrees) { return degrees * PI / 180.0; }
static double clamp(double x, double min, double max) { if (x < min) return min; if (x > max) return max; return x; }
static double fmin_(double a, double b) { if (a <= b) return a; return b; }
static double fmax_(double a, double b) { if (a >= b) return a; return b; }

// ray.rs
struct Ray {
    Point3 orig; Vec3 dir; double tm;
    Ray() : tm(0.0) {}
    Ray(Point3 o, Vec3 d, double t) : orig(o), dir(d), tm(t) {}
    Point3 origin() const { return orig; }
    Vec3 direction() const { return dir; }
    Point3 at(double t) const { return orig + dir * t; }
    double time() const { return tm; }
};

// aabb.rs
struct AABB {
    Point3 minimum, maximum;
    AABB() {}
    AABB(Point3 a, Point3 b) : minimum(a), maximum(b) {}
    Point3 min() const { return minimum; }
    Point3 max() const { return maximum; }
    bool hit(const Ray& r, double t_min, double t_max) const { // aabb.rs:35-75
        {
            double inv_d = 1.0 / r.direction().x;
            double t0 = (min().x - r.origin().x) * inv_d;
            double t1 = (max().x - r.origin().x) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        {
            double inv_d = 1.0 / r.direction().y;
            double t0 = (min().y - r.origin().y) * inv_d;
            double t1 = (max().y - r.origin().y) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        {
            double inv_d = 1.0 / r.direction().z;
            double t0 = (min().z - r.origin().z) * inv_d;
            double t1 = (max().z - r.origin().z) * inv_d;
            if (inv_d < 0.0) std::swap(t0, t1);
            t_min = t0 > t_min ? t0 : t_min;
            t_max = t1 < t_max ? t1 : t_max;
            if (t_max <= t_min) return false;
        }
        return true;
    }
    static AABB surrounding_box(AABB box0, AABB box1) { // aabb.rs:77-91
        Point3 small(fmin_(box0.min().x, box1.min().x), fmin_(box0.min().y, box1.min().y), fmin_(box0.min().z, box1.min().z));
        Point3 big(fmax_(box0.max().x, box1.max().x), fmax_(box0.max().y, box1.max().y), fmax_(box0.max().z, box1.max().z));
        return AABB(small, big);
    }
};

// ---------------------------------------------------------------------------------------------------------
// texture.rs, perlin.rs
// ---------------------------------------------------------------------------------------------------------
struct Texture {
    virtual ~Texture() {}
    virtual Color value(double u, double v, Point3 p) const = 0;
};
struct SolidColor : Texture {
    Color color_value;
    explicit SolidColor(Color c) : color_value(c) {}
    Color value(double, double, Point3) const override { return color_value; }
};
struct CheckerTexture : Texture {
    std::shared_ptr<Texture> odd, even;
    CheckerTexture(std::shared_ptr<Texture> even_, std::shared_ptr<Texture> odd_) : odd(odd_), even(even_) {}
    Color value(double u, double v, Point3 p) const override { // texture.rs:46-54
        double sines = std::sin(10.0 * p.x) * std::sin(10.0 * p.y) * std::sin(10.0 * p.z);
        if (sines < 0.0) return odd->value(u, v, p);
        return even->value(u, v, p);
    }
};
static int32_t f64_as_i32(double v) { // Rust `as i32`: saturating, NaN -> 0
    if (std::isnan(v)) return 0;
    if (v <= -2147483648.0) return INT32_MIN;
    if (v >= 2147483647.0) return INT32_MAX;
    return (int32_t)v;
}
static size_t f64_as_usize(double v) { // Rust `as usize`
    if (std::isnan(v) || v <= 0.0) return 0;
    if (v >= 18446744073709551615.0) return SIZE_MAX;
    return (size_t)v;
}
// The uniforms Perlin::init draws while a scene is built. The reference takes them from thread_rng(); here they come from
// the scene builder's sequential stream, whose definition is restated (not shared): draw number 2c and 2c+1 are words (0,1)
// and (2,3) of Philox(counter = (c lo, c hi, 0x5EED, stream), key = (seed lo, seed hi)). The constructor log records the
// stream's position in front of each NoiseTexture, so that the tables below are made here, by this file's own reading of
// perlin.rs:15-54, and only compared with the ones the product made.
struct BuildStream {
    uint64_t seed = 0, counter = 0;
    uint32_t stream = 0;
    bool have_spare = false;
    double spare = 0.0;
    double next() {
        if (have_spare) { have_spare = false; return spare; }
        const U4 o = philox(U4{(uint32_t)counter, (uint32_t)(counter >> 32), 0x5EEDu, stream}, (uint32_t)seed, (uint32_t)(seed >> 32));
        ++counter;
        spare = to_unit(o.c, o.d);
        have_spare = true;
        return to_unit(o.a, o.b);
    }
    double random_double(double low, double high) { return low + (high - low) * next(); }       // rtweekend.rs:19-22
    int random_int(int low, int high) {                                                          // rtweekend.rs:24-27
        const int v = low + (int)std::floor(next() * (double)(high - low));
        return v >= high ? high - 1 : v;
    }
};
struct Perlin {
    std::vector<Vec3> ranvec;
    std::vector<int32_t> perm_x, perm_y, perm_z;
    static void permut(std::vector<int32_t>& p, int n, BuildStream& rng) { // perlin.rs:28-34: swaps p[i] with an entry of p[0 .. n-i), not of p[0 .. i]
        for (int i = 1; i < n; ++i) {
            const size_t target = (size_t)rng.random_int(0, n - i);
            std::swap(p[(size_t)i], p[target]);
        }
    }
    void perlin_generate_perm(int axis, BuildStream& rng) {                // perlin.rs:15-26
        std::vector<int32_t>& tmp = axis == 0 ? perm_x : axis == 1 ? perm_y : perm_z;
        const int point_count = 256;
        for (int i = 0; i < point_count; ++i) tmp.push_back(i);
        permut(tmp, point_count, rng);
    }
    void init(BuildStream& rng) {                                          // perlin.rs:46-54; Vec3::random_range, vec3.rs:65-71
        for (int i = 0; i < 256; ++i) {
            const double x = rng.random_double(-1.0, 1.0), y = rng.random_double(-1.0, 1.0), z = rng.random_double(-1.0, 1.0);
            ranvec.push_back(Vec3(x, y, z));
        }
        perlin_generate_perm(0, rng);
        perlin_generate_perm(1, rng);
        perlin_generate_perm(2, rng);
    }
    static double trilinear_interp(Vec3 c[2][2][2], double u, double v, double w) { // perlin.rs:87-111
        double accum = 0.0;
        double uu = u * u * (3.0 - 2.0 * u);
        double vv = v * v * (3.0 - 2.0 * v);
        double ww = w * w * (3.0 - 2.0 * w);
        for (double i = 0.0; i < 2.0; i += 1.0)
            for (double j = 0.0; j < 2.0; j += 1.0)
                for (double k = 0.0; k < 2.0; k += 1.0) {
                    Vec3 weight_v(u - i, v - j, w - k);
                    accum += (i * uu + (1.0 - i) * (1.0 - uu)) * (j * vv + (1.0 - j) * (1.0 - vv)) *
                             (k * ww + (1.0 - k) * (1.0 - ww)) * Vec3::dot(c[(int)i][(int)j][(int)k], weight_v);
                }
        return accum;
    }
    double noise(Point3 p) const { // perlin.rs:56-85
        double u = p.x - std::floor(p.x), v = p.y - std::floor(p.y), w = p.z - std::floor(p.z);
        double i_ = std::floor(p.x), j_ = std::floor(p.y), k_ = std::floor(p.z);
        Vec3 c[2][2][2];
        for (double di = 0.0; di < 2.0; di += 1.0)
            for (double dj = 0.0; dj < 2.0; dj += 1.0)
                for (double dk = 0.0; dk < 2.0; dk += 1.0) {
                    size_t id_x = (size_t)(255 & f64_as_i32(i_ + di));
                    size_t id_y = (size_t)(255 & f64_as_i32(j_ + dj));
                    size_t id_z = (size_t)(255 & f64_as_i32(k_ + dk));
                    size_t tmp = (size_t)(perm_x[id_x] ^ perm_y[id_y] ^ perm_z[id_z]);
                    c[(int)di][(int)dj][(int)dk] = ranvec[tmp];
                }
        return trilinear_interp(c, u, v, w);
    }
    double turb(Point3 p, int depth) const { // perlin.rs:113-125
        double accum = 0.0;
        Point3 temp_p = p;
        double weight = 1.0;
        for (int i = 0; i < depth; ++i) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p = temp_p * 2.0;
        }
        return std::fabs(accum);
    }
};
struct NoiseTexture : Texture {
    Perlin noise;                 // made by Perlin::init above
    std::vector<double> listed_vec;   // what the constructor log lists (the product's tables): never used to render,
    std::vector<int32_t> listed_perm; // only handed out for comparison (orc_perlin_tables)
    double scale;
    Color value(double, double, Point3 p) const override { // texture.rs:73-79
        return Color(1.0, 1.0, 1.0) * 0.5 * (1.0 + std::sin(scale * p.z + 10.0 * noise.turb(p, 7)));
    }
};
struct ImageTexture : Texture {
    std::vector<uint8_t> data;
    uint32_t width = 0, height = 0, bytes_per_scanline = 0;
    Color value(double u_, double v_, Point3) const override { // texture.rs:105-127
        u_ = clamp(u_, 0.0, 1.0);
        v_ = 1.0 - clamp(v_, 0.0, 1.0);
        size_t i = f64_as_usize(u_ * (double)width);
        size_t j = f64_as_usize(v_ * (double)height);
        if (i >= (size_t)width) i = (size_t)(width - 1);
        if (j >= (size_t)height) j = (size_t)(height - 1);
        double color_scale = 1.0 / 255.0;
        return Color(color_scale * (double)data[j * bytes_per_scanline + i * 3],
                     color_scale * (double)data[j * bytes_per_scanline + i * 3 + 1],
                     color_scale * (double)data[j * bytes_per_scanline + i * 3 + 2]);
    }
};

// ---------------------------------------------------------------------------------------------------------
// hittable.rs: HitRecord, Hittable; material.rs
// ---------------------------------------------------------------------------------------------------------
struct Material;
struct HitRecord {
    Point3 p;
    Vec3 normal;
    bool front_face;
    double t;
    std::shared_ptr<Material> mat_ptr;
    double u, v;
    // identity of what was hit (not in the reference; lets tests compare primitive ids)
    uint32_t prim_id = 0xFFFFFFFFu, face = 0, material_id = 0;
    HitRecord();
    void set_face_normal(const Ray& r, Vec3& outward_normal) { // hittable.rs:32-39
        front_face = Vec3::dot(r.direction(), outward_normal) < 0.0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};
struct Material {
    int id = -1;
    virtual ~Material() {}
    virtual bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered, const Rng& rng) const = 0;
    virtual Color emitted(double, double, const Vec3&) const { return Color(0.0, 0.0, 0.0); } // material.rs:19-21
};
struct Lambertian : Material {
    std::shared_ptr<Texture> albedo;
    explicit Lambertian(Color a) : albedo(std::make_shared<SolidColor>(a)) {}
    explicit Lambertian(std::shared_ptr<Texture> a) : albedo(a) {}
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered, const Rng& rng) const override {
        Vec3 scatter_direction = rec.normal + random_unit_vector(rng); // material.rs:49
        if (scatter_direction.near_zero()) scatter_direction = rec.normal;
        scattered = Ray(rec.p, scatter_direction, r_in.time());
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
};
// HitRecord::new (hittable.rs:21-31) allocates a fresh Lambertian (and its SolidColor) every time.
HitRecord::HitRecord() : front_face(false), t(0.0), mat_ptr(std::make_shared<Lambertian>(Color(0.0, 0.0, 0.0))), u(0.0), v(0.0) {}

struct Metal : Material {
    Color albedo;
    double fuzz;
    Metal(Color a, double f) : albedo(a) { double f1 = f; if (f1 > 1.0) f1 = 1.0; fuzz = f1; } // material.rs:68-77
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered, const Rng& rng) const override {
        Vec3 reflected = reflect(r_in.direction().unit(), rec.normal);
        scattered = Ray(rec.p, reflected + random_in_unit_sphere(rng) * fuzz, r_in.time());
        attenuation = albedo;
        return Vec3::dot(scattered.direction(), rec.normal) > 0.0;
    }
};
struct Dielectric : Material {
    double ir;
    explicit Dielectric(double i) : ir(i) {}
    static double reflectance(double cosine, double ref_idx) { // material.rs:109-113
        double r0 = (1.0 - ref_idx) / (1.0 + ref_idx);
        r0 = r0 * r0;
        double x = 1.0 - cosine;
        double x2 = x * x;             // powi(5) as LLVM expands it: x * (x^2)^2
        double x4 = x2 * x2;
        return r0 + (1.0 - r0) * (x4 * x);
    }
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered, const Rng& rng) const override {
        attenuation = Color::ones();
        double refraction_ratio = rec.front_face ? 1.0 / ir : ir;
        Vec3 unit_direction = r_in.direction().unit();
        double cos_theta = Vec3::dot(-unit_direction, rec.normal);
        if (cos_theta > 1.0) cos_theta = 1.0;
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = refraction_ratio * sin_theta > 1.0;
        Vec3 direction;
        if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng.draw1(S_DIELECTRIC, 0))
            direction = reflect(unit_direction, rec.normal);
        else
            direction = refract(unit_direction, rec.normal, refraction_ratio);
        scattered = Ray(rec.p, direction, r_in.time());
        return true;
    }
};
struct DiffuseLight : Material {
    std::shared_ptr<Texture> emit;
    double k;
    DiffuseLight(std::shared_ptr<Texture> e, double k_) : emit(e), k(k_) {}
    bool scatter(const Ray&, const HitRecord&, Color&, Ray&, const Rng&) const override { return false; }
    Color emitted(double u_, double v_, const Vec3& p_) const override { return emit->value(u_, v_, p_) * k; } // material.rs:201-203
};
struct Isotropic : Material {
    std::shared_ptr<Texture> albedo;
    explicit Isotropic(std::shared_ptr<Texture> a) : albedo(a) {}
    bool scatter(const Ray& r_in, const HitRecord& rec, Color& attenuation, Ray& scattered, const Rng& rng) const override {
        scattered = Ray(rec.p, random_in_unit_sphere(rng), r_in.time()); // material.rs:227
        attenuation = albedo->value(rec.u, rec.v, rec.p);
        return true;
    }
};

struct Hittable {
    uint32_t id = 0xFFFFFFFFu;
    virtual ~Hittable() {}
    virtual bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const = 0;
    virtual bool bounding_box(double time0, double time1, AABB& output_box) const = 0;
};
typedef std::shared_ptr<Hittable> HittablePtr;

// sphere.rs
struct Sphere : Hittable {
    Point3 center; double radius; std::shared_ptr<Material> mat_ptr;
    Sphere(Point3 c, double r, std::shared_ptr<Material> m) : center(c), radius(r), mat_ptr(m) {}
    static void get_sphere_uv(Point3 p, double& u, double& v) { // sphere.rs:24-30
        double theta = std::acos(-p.y);
        double phi = std::atan2(-p.z, p.x) + PI;
        u = phi / (2.0 * PI);
        v = theta / PI;
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // sphere.rs:34-64
        if (rng.cnt) rng.cnt->spheres++;
        Vec3 oc = r.origin() - center;
        double a = r.direction().squared_length();
        double half_b = Vec3::dot(r.direction(), oc);
        double c = oc.squared_length() - radius * radius;
        double delta = half_b * half_b - a * c;
        if (delta > 0.0) {
            double root = std::sqrt(delta);
            double t = (-half_b - root) / a;
            for (int pass = 0; pass < 2; ++pass) {
                if (t > t_min && t < t_max) {
                    rec.t = t;
                    rec.p = r.at(t);
                    Vec3 outward_normal = (rec.p - center) / radius;
                    rec.set_face_normal(r, outward_normal);
                    get_sphere_uv(outward_normal, rec.u, rec.v);
                    rec.mat_ptr = mat_ptr;
                    rec.prim_id = id; rec.face = 0; rec.material_id = (uint32_t)mat_ptr->id;
                    return true;
                }
                t = (-half_b + root) / a;
            }
        }
        return false;
    }
    bool bounding_box(double, double, AABB& output_box) const override { // sphere.rs:66-72
        output_box = AABB(center - Vec3(radius, radius, radius), center + Vec3(radius, radius, radius));
        return true;
    }
};
// moving_sphere.rs
struct MovingSphere : Hittable {
    Point3 center0, center1; double time0, time1, radius; std::shared_ptr<Material> mat_ptr;
    Point3 center(double time) const { return center0 + (center1 - center0) * (time - time0) / (time1 - time0); } // :37-40
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :44-69
        if (rng.cnt) rng.cnt->msphere++;
        Vec3 oc = r.origin() - center(r.time());
        double a = r.direction().squared_length();
        double half_b = Vec3::dot(oc, r.direction());
        double c = oc.squared_length() - radius * radius;
        double discriminant = half_b * half_b - a * c;
        if (discriminant < 0.0) return false;
        double sqrtd = std::sqrt(discriminant);
        double root = (-half_b - sqrtd) / a;
        if (root < t_min || root > t_max) {
            root = (-half_b + sqrtd) / a;
            if (root < t_min || root > t_max) return false;
        }
        rec.t = root;
        rec.p = r.at(rec.t);
        Vec3 outward_normal = (rec.p - center(r.time())) / radius;
        rec.set_face_normal(r, outward_normal);
        rec.mat_ptr = mat_ptr; // u, v are not written: whatever the record held stays (moving_sphere.rs:63-68)
        rec.prim_id = id; rec.face = 0; rec.material_id = (uint32_t)mat_ptr->id;
        return true;
    }
    bool bounding_box(double t0, double t1, AABB& output_box) const override { // :71-84
        AABB box0(center(t0) - Vec3(radius, radius, radius), center(t0) + Vec3(radius, radius, radius));
        AABB box1(center(t1) - Vec3(radius, radius, radius), center(t1) + Vec3(radius, radius, radius));
        output_box = AABB::surrounding_box(box0, box1);
        return true;
    }
};
// arrect.rs
struct XYRect : Hittable {
    std::shared_ptr<Material> mp; double x0, x1, y0, y1, k; uint32_t face = 0;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :31-49
        if (rng.cnt) rng.cnt->rects++;
        double t = (k - r.origin().z) / r.direction().z;
        if (t < t_min || t > t_max) return false;
        double x = r.origin().x + t * r.direction().x;
        double y = r.origin().y + t * r.direction().y;
        if (x < x0 || x > x1 || y < y0 || y > y1) return false;
        rec.u = (x - x0) / (x1 - x0);
        rec.v = (y - y0) / (y1 - y0);
        rec.t = t;
        Vec3 outward_normal(0.0, 0.0, 1.0);
        rec.set_face_normal(r, outward_normal);
        rec.mat_ptr = mp;
        rec.p = r.at(t);
        rec.prim_id = id; rec.face = face; rec.material_id = (uint32_t)mp->id;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {
        output_box = AABB(Point3(x0, y0, k - 0.0001), Point3(x1, y1, k + 0.0001));
        return true;
    }
};
struct XZRect : Hittable {
    std::shared_ptr<Material> mp; double x0, x1, z0, z1, k; uint32_t face = 0;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :83-101
        if (rng.cnt) rng.cnt->rects++;
        double t = (k - r.origin().y) / r.direction().y;
        if (t < t_min || t > t_max) return false;
        double x = r.origin().x + t * r.direction().x;
        double z = r.origin().z + t * r.direction().z;
        if (x < x0 || x > x1 || z < z0 || z > z1) return false;
        rec.u = (x - x0) / (x1 - x0);
        rec.v = (z - z0) / (z1 - z0);
        rec.t = t;
        Vec3 outward_normal(0.0, 1.0, 0.0);
        rec.set_face_normal(r, outward_normal);
        rec.mat_ptr = mp;
        rec.p = r.at(t);
        rec.prim_id = id; rec.face = face; rec.material_id = (uint32_t)mp->id;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {
        output_box = AABB(Point3(x0, k - 0.0001, z0), Point3(x1, k + 0.0001, z1));
        return true;
    }
};
struct YZRect : Hittable {
    std::shared_ptr<Material> mp; double y0, y1, z0, z1, k; uint32_t face = 0;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :135-153
        if (rng.cnt) rng.cnt->rects++;
        double t = (k - r.origin().x) / r.direction().x;
        if (t < t_min || t > t_max) return false;
        double y = r.origin().y + t * r.direction().y;
        double z = r.origin().z + t * r.direction().z;
        if (y < y0 || y > y1 || z < z0 || z > z1) return false;
        rec.u = (y - y0) / (y1 - y0);
        rec.v = (z - z0) / (z1 - z0);
        rec.t = t;
        Vec3 outward_normal(1.0, 0.0, 0.0);
        rec.set_face_normal(r, outward_normal);
        rec.mat_ptr = mp;
        rec.p = r.at(t);
        rec.prim_id = id; rec.face = face; rec.material_id = (uint32_t)mp->id;
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override {
        output_box = AABB(Point3(k - 0.0001, y0, z0), Point3(k + 0.0001, y1, z1));
        return true;
    }
};
// hittable_list.rs
struct HittableList : Hittable {
    std::vector<HittablePtr> objects;
    void add(HittablePtr o) { objects.push_back(o); }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :40-53
        HitRecord tmp_rec;
        bool hit_anything = false;
        double closest_so_far = t_max;
        for (const HittablePtr& object : objects) {
            if (object->hit(r, t_min, closest_so_far, tmp_rec, rng)) {
                hit_anything = true;
                closest_so_far = tmp_rec.t;
                rec = tmp_rec;
            }
        }
        return hit_anything;
    }
    bool bounding_box(double time0, double time1, AABB& output_box) const override { // :55-75
        if (objects.empty()) return false;
        AABB temp_box;
        bool first_box = true;
        for (const HittablePtr& object : objects) {
            if (!object->bounding_box(time0, time1, temp_box)) return false;
            output_box = first_box ? temp_box : AABB::surrounding_box(output_box, temp_box);
            first_box = true; // sic (hittable_list.rs:72)
        }
        return true;
    }
};
// _box.rs
struct Box_ : Hittable {
    Point3 box_min, box_max;
    HittableList sides;
    Box_(Point3 p0, Point3 p1, std::shared_ptr<Material> ptr, uint32_t id_) : box_min(p0), box_max(p1) {
        id = id_;
        auto xy = [&](double k, uint32_t f) { auto q = std::make_shared<XYRect>(); q->mp = ptr; q->x0 = p0.x; q->x1 = p1.x; q->y0 = p0.y; q->y1 = p1.y; q->k = k; q->face = f; q->id = id_; sides.add(q); };
        auto xz = [&](double k, uint32_t f) { auto q = std::make_shared<XZRect>(); q->mp = ptr; q->x0 = p0.x; q->x1 = p1.x; q->z0 = p0.z; q->z1 = p1.z; q->k = k; q->face = f; q->id = id_; sides.add(q); };
        auto yz = [&](double k, uint32_t f) { auto q = std::make_shared<YZRect>(); q->mp = ptr; q->y0 = p0.y; q->y1 = p1.y; q->z0 = p0.z; q->z1 = p1.z; q->k = k; q->face = f; q->id = id_; sides.add(q); };
        xy(p1.z, 0); xy(p0.z, 1); xz(p1.y, 2); xz(p0.y, 3); yz(p1.x, 4); yz(p0.x, 5); // _box.rs:22-71
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { return sides.hit(r, t_min, t_max, rec, rng); }
    bool bounding_box(double, double, AABB& output_box) const override { output_box = AABB(box_min, box_max); return true; }
};
// hittable.rs: Translate, RotateY
struct Translate : Hittable {
    HittablePtr ptr; Vec3 offset;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :70-79
        Ray moved_r(r.origin() - offset, r.direction(), r.time());
        if (!ptr->hit(moved_r, t_min, t_max, rec, rng)) return false;
        rec.p = rec.p + offset;
        Vec3 outward_normal = rec.normal;
        rec.set_face_normal(moved_r, outward_normal);
        return true;
    }
    bool bounding_box(double time0, double time1, AABB& output_box) const override { // :81-90
        if (!ptr->bounding_box(time0, time1, output_box)) return false;
        output_box = AABB(output_box.min() + offset, output_box.max() + offset);
        return true;
    }
};
struct RotateY : Hittable {
    HittablePtr ptr; double sin_theta, cos_theta; bool hasbox; AABB bbox; bool book_rotate_bbox = false;
    RotateY(HittablePtr p, double angle, bool book_bbox) : ptr(p) { // :104-144
        double radians = degrees_to_radians(angle);
        // glibc's sin()/cos() and sincos() may differ in the last bit; compilers merge the two calls or not as they
        // please, so the pair is asked for explicitly (the product's builder does the same)
        ::sincos(radians, &sin_theta, &cos_theta);
        hasbox = ptr->bounding_box(0.0, 1.0, bbox);
        Point3 min(INFINITY_, INFINITY_, INFINITY_), max(-INFINITY_, -INFINITY_, -INFINITY_);
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double x = i * bbox.max().x + (1.0 - i) * bbox.min().x;
                    double y = j * bbox.max().y + (1.0 - j) * bbox.min().y;
                    double z = k * bbox.max().z + (1.0 - k) * bbox.min().z;
                    double newx = cos_theta * x + sin_theta * z;
                    double newz = cos_theta * z - sin_theta * x;
                    Vec3 tester(newx, y, newz);
                    min.x = fmin_(min.x, tester.x); min.y = fmin_(min.y, tester.y); min.z = fmin_(min.z, tester.z);
                    max.x = fmax_(max.x, tester.x); max.y = fmax_(max.y, tester.y); max.z = fmax_(max.z, tester.z);
                }
        // the reference returns `tmp` here without storing min/max (hittable.rs:143)
        if (book_bbox) bbox = AABB(min, max);
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :148-177
        Point3 origin = r.origin();
        Vec3 direction = r.direction();
        origin.x = cos_theta * r.origin().x - sin_theta * r.origin().z;
        origin.z = sin_theta * r.origin().x + cos_theta * r.origin().z;
        direction.x = cos_theta * r.direction().x - sin_theta * r.direction().z;
        direction.z = sin_theta * r.direction().x + cos_theta * r.direction().z;
        Ray rotated_r(origin, direction, r.time());
        if (!ptr->hit(rotated_r, t_min, t_max, rec, rng)) return false;
        Point3 p_ = rec.p;
        Vec3 normal = rec.normal;
        if (rng.compat & 1u) { // book-correct back rotation
            p_.x = cos_theta * rec.p.x + sin_theta * rec.p.z;
            p_.z = -sin_theta * rec.p.x + cos_theta * rec.p.z;
            normal.x = cos_theta * rec.normal.x + sin_theta * rec.normal.z;
            normal.z = -sin_theta * rec.normal.x + cos_theta * rec.normal.z;
        } else { // as written: the second line of each pair reads the x just overwritten
            p_.x = cos_theta * p_.x + sin_theta * p_.z;
            p_.z = -sin_theta * p_.x + cos_theta * p_.z;
            normal.x = cos_theta * normal.x + sin_theta * normal.z;
            normal.z = -sin_theta * normal.x + cos_theta * normal.z;
        }
        rec.p = p_;
        rec.set_face_normal(rotated_r, normal);
        return true;
    }
    bool bounding_box(double, double, AABB& output_box) const override { output_box = bbox; return hasbox; }
};
// constant_medium.rs
struct ConstantMedium : Hittable {
    HittablePtr boundary; std::shared_ptr<Material> phase_function; double neg_inv_density; uint32_t medium_index = 0;
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :25-66
        if (rng.skip_media) return false;
        if (rng.cnt) rng.cnt->media++;
        HitRecord rec1, rec2;
        if (!boundary->hit(r, -INFINITY_, INFINITY_, rec1, rng)) return false;
        if (!boundary->hit(r, rec1.t + 0.0001, INFINITY_, rec2, rng)) return false;
        if (rec1.t < t_min) rec1.t = t_min;
        if (rec2.t > t_max) rec2.t = t_max;
        if (rec1.t >= rec2.t) return false;
        if (rec1.t < 0.0) rec1.t = 0.0;
        double ray_length = r.direction().length();
        double distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
        double xi = rng.medium_xi ? rng.medium_xi[medium_index] : rng.draw1(S_MEDIUM0 + medium_index, 0);
        double hit_distance = neg_inv_density * std::log(xi);
        if (hit_distance > distance_inside_boundary) return false;
        rec.t = rec1.t + hit_distance / ray_length;
        rec.p = r.at(rec.t);
        rec.normal = Vec3(1.0, 0.0, 0.0);
        rec.front_face = true;
        rec.mat_ptr = phase_function;
        rec.prim_id = id; rec.face = 0; rec.material_id = (uint32_t)phase_function->id;
        return true;
    }
    bool bounding_box(double time0, double time1, AABB& output_box) const override { return boundary->bounding_box(time0, time1, output_box); }
};
// bvh.rs
struct SeqRng { // the thread_rng() of scene construction
    uint64_t seed = 0, counter = 0;
    const std::string* replay = nullptr; // when set: the split axes to use, one digit per BvhNode::new call
    size_t replay_pos = 0;
    double next() {
        U4 o = philox(U4{(uint32_t)counter, (uint32_t)(counter >> 32), 0x0BC1E5u, 7u}, (uint32_t)seed, (uint32_t)(seed >> 32));
        ++counter;
        return to_unit(o.a, o.b);
    }
    int random_int(int low, int high) {
        if (replay && replay_pos < replay->size()) return (*replay)[replay_pos++] - '0';
        return random_int_own(low, high);
    }
    int random_int_own(int low, int high) { int v = low + (int)std::floor(next() * (double)(high - low)); return v >= high ? high - 1 : v; }
};
struct BvhNode : Hittable {
    HittablePtr left, right;
    AABB box_;
    static bool box_compare(const HittablePtr& a, const HittablePtr& b, int axis) { // :73-87
        AABB boxa, boxb;
        if (!a->bounding_box(0.0, 0.0, boxa) || !b->bounding_box(0.0, 0.0, boxb)) throw Panic{"No bounding box"};
        if (axis == 0) return boxa.min().x < boxb.min().x;
        if (axis == 1) return boxa.min().y < boxb.min().y;
        return boxa.min().z < boxb.min().z;
    }
    BvhNode(std::vector<HittablePtr>& src_objects, size_t start, size_t end, double time0, double time1, SeqRng& rng) { // :15-66
        int axis = rng.random_int(0, 3);
        auto comparator = [axis](const HittablePtr& a, const HittablePtr& b) { return box_compare(a, b, axis); };
        size_t object_span = end - start;
        if (object_span == 1) {
            left = src_objects[start];
            right = src_objects[start];
        } else if (object_span == 2) {
            if (comparator(src_objects[start], src_objects[start + 1])) { left = src_objects[start]; right = src_objects[start + 1]; }
            else { left = src_objects[start + 1]; right = src_objects[start]; }
        } else {
            std::stable_sort(src_objects.begin() + (long)start, src_objects.begin() + (long)end, comparator);
            size_t mid = start + object_span / 2;
            left = std::make_shared<BvhNode>(src_objects, start, mid, time0, time1, rng);
            right = std::make_shared<BvhNode>(src_objects, mid, end, time0, time1, rng);
        }
        AABB box_left, box_right;
        if (!left->bounding_box(time0, time1, box_left) || !right->bounding_box(time0, time1, box_right))
            throw Panic{"No bounding box in bvh_node constructor"};
        box_ = AABB::surrounding_box(box_left, box_right);
    }
    bool hit(const Ray& r, double t_min, double t_max, HitRecord& rec, const Rng& rng) const override { // :112-122
        if (rng.cnt) rng.cnt->nodes++;
        if (!box_.hit(r, t_min, t_max)) return false;
        bool hit_left = left->hit(r, t_min, t_max, rec, rng);
        double maxi = hit_left ? rec.t : t_max;
        bool hit_right = right->hit(r, t_min, maxi, rec, rng);
        return hit_left || hit_right;
    }
    bool bounding_box(double, double, AABB& output_box) const override { output_box = box_; return true; }
};

// camera.rs
struct Camera {
    Point3 origin, lower_left_corner; Vec3 horizontal, vertical, u, v; double lens_radius, time0, time1;
    Camera(Point3 lookfrom, Point3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture, double focus_dist,
           double t0, double t1) { // :21-54
        double theta = degrees_to_radians(vfov);
        double h = std::tan(theta / 2.0);
        double viewport_height = 2.0 * h;
        double viewport_width = aspect_ratio * viewport_height;
        Vec3 w1 = (lookfrom - lookat).unit();
        Vec3 u1 = Vec3::cross(vup, w1).unit();
        Vec3 v1 = Vec3::cross(w1, u1);
        Vec3 hor = u1 * viewport_width * focus_dist;
        Vec3 ver = v1 * viewport_height * focus_dist;
        Vec3 llc = lookfrom - hor * 0.5 - ver * 0.5 - w1 * focus_dist;
        origin = lookfrom; horizontal = hor; vertical = ver; lower_left_corner = llc; u = u1; v = v1;
        lens_radius = aperture / 2.0; time0 = t0; time1 = t1;
    }
    Ray get_ray(double s, double t, const Rng& rng) const { // :56-64
        Vec3 rd = random_in_unit_disk(rng) * lens_radius;
        Vec3 offset = u * rd.x + v * rd.y;
        return Ray(origin + offset, lower_left_corner + horizontal * s + vertical * t - origin - offset,
                   in_range(rng.draw1(S_TIME, 0), time0, time1));
    }
};

// main.rs:49-71. background_mode 1 = the sky gradient of the older snapshot (/root/reference/main.rs:59-61).
static Color ray_color(const Ray& r, Color background, uint32_t background_mode, const HittableList& world, int depth, int max_depth,
                       Rng& rng) {
    HitRecord rec;
    if (depth <= 0) return Color(0.0, 0.0, 0.0);
    rng.bounce = (uint32_t)(max_depth - depth);
    if (rng.cnt) rng.cnt->segments++;
    if (!world.hit(r, 0.001, INFINITY_, rec, rng)) {
        if (background_mode == 1) {
            Vec3 unit_direction = r.direction().unit();
            double t = (unit_direction.y + 1.0) * 0.5;
            return Color(1.0, 1.0, 1.0) * (1.0 - t) + Color(0.5, 0.7, 1.0) * t;
        }
        return background;
    }
    Ray scattered;
    Color attenuation = Color::zero();
    Color emitted = rec.mat_ptr->emitted(rec.u, rec.v, rec.p);
    HitRecord tmp_rec = rec;
    if (!rec.mat_ptr->scatter(r, tmp_rec, attenuation, scattered, rng)) return emitted;
    return emitted + ray_color(scattered, background, background_mode, world, depth - 1, max_depth, rng) * attenuation;
}

// ---------------------------------------------------------------------------------------------------------
// scene loading: replay of the product's constructor log
// ---------------------------------------------------------------------------------------------------------
struct Scene {
    std::map<int, std::shared_ptr<Texture>> tex;
    std::map<int, std::shared_ptr<Material>> mat;
    std::map<int, HittablePtr> obj;
    std::shared_ptr<HittableList> world;
    uint32_t flags = 0, n_media = 0;
    SeqRng build_rng;
};

struct Reader {
    const char* p; const char* end;
    bool eof() const { return p >= end; }
    std::string line() {
        const char* s = p;
        while (p < end && *p != '\n') ++p;
        std::string l(s, p);
        if (p < end) ++p;
        return l;
    }
};

static Scene* load(const char* text, size_t n, uint64_t bvh_seed, bool share_axes, std::string& err) {
    std::unique_ptr<Scene> sc(new Scene());
    sc->build_rng.seed = bvh_seed;
    Reader rd{text, text + n};
    try {
        while (!rd.eof()) {
            std::string l = rd.line();
            if (l.empty()) continue;
            std::istringstream in(l);
            std::string op;
            in >> op;
            auto D = [&]() { std::string tok; in >> tok; if (tok.empty()) throw Panic{"short line: " + l.substr(0, 60)}; return std::strtod(tok.c_str(), nullptr); };
            auto I = [&]() { long v; if (!(in >> v)) throw Panic{"short line: " + l.substr(0, 60)}; return (int)v; };
            auto M = [&](int id) { auto it = sc->mat.find(id); if (it == sc->mat.end()) throw Panic{"unknown material"}; return it->second; };
            auto T = [&](int id) { auto it = sc->tex.find(id); if (it == sc->tex.end()) throw Panic{"unknown texture"}; return it->second; };
            auto O = [&](int id) { auto it = sc->obj.find(id); if (it == sc->obj.end()) throw Panic{"unknown object"}; return it->second; };
            if (op == "rtdesc") continue;
            else if (op == "flags") sc->flags = (uint32_t)I();
            else if (op == "tex_solid") { int id = I(); double r = D(), g = D(), b = D(); sc->tex[id] = std::make_shared<SolidColor>(Color(r, g, b)); }
            else if (op == "tex_checker") { int id = I(); int e = I(), o = I(); sc->tex[id] = std::make_shared<CheckerTexture>(T(e), T(o)); }
            else if (op == "tex_noise") {
                int id = I();
                auto t = std::make_shared<NoiseTexture>();
                t->scale = D();
                for (int i = 0; i < 768; ++i) t->listed_vec.push_back(D());
                for (int i = 0; i < 768; ++i) t->listed_perm.push_back(I());
                std::string tag;
                in >> tag;
                if (tag != "rng") throw Panic{"tex_noise: the position of the build stream is missing"};
                BuildStream rng;
                auto U = [&]() { unsigned long long v; if (!(in >> v)) throw Panic{"short line: " + l.substr(0, 60)}; return (uint64_t)v; };
                { const uint64_t lo = U(), hi = U(); rng.seed = lo | (hi << 32); }
                { const uint64_t lo = U(), hi = U(); rng.counter = lo | (hi << 32); }
                rng.stream = (uint32_t)U();
                rng.have_spare = I() != 0;
                rng.spare = D();
                t->noise.init(rng);
                sc->tex[id] = t;
            } else if (op == "tex_image") {
                int id = I(); int w = I(), h = I(); long nb; in >> nb;
                if (nb != (long)w * h * 3 || rd.p + nb > rd.end) throw Panic{"bad image block"};
                auto t = std::make_shared<ImageTexture>();
                t->width = (uint32_t)w; t->height = (uint32_t)h; t->bytes_per_scanline = 3u * (uint32_t)w;
                t->data.assign((const uint8_t*)rd.p, (const uint8_t*)rd.p + nb);
                rd.p += nb;
                if (rd.p < rd.end && *rd.p == '\n') ++rd.p;
                sc->tex[id] = t;
            }
            else if (op == "mat_lambertian") { int id = I(); auto m = std::make_shared<Lambertian>(T(I())); m->id = id; sc->mat[id] = m; }
            else if (op == "mat_metal") { int id = I(); double r = D(), g = D(), b = D(), f = D(); auto m = std::make_shared<Metal>(Color(r, g, b), f); m->id = id; sc->mat[id] = m; }
            else if (op == "mat_dielectric") { int id = I(); auto m = std::make_shared<Dielectric>(D()); m->id = id; sc->mat[id] = m; }
            else if (op == "mat_diffuse_light") { int id = I(); int t = I(); double k = D(); auto m = std::make_shared<DiffuseLight>(T(t), k); m->id = id; sc->mat[id] = m; }
            else if (op == "mat_isotropic") { int id = I(); auto m = std::make_shared<Isotropic>(T(I())); m->id = id; sc->mat[id] = m; }
            else if (op == "sphere") {
                int id = I(); double x = D(), y = D(), z = D(), r = D(); int m = I();
                auto s = std::make_shared<Sphere>(Point3(x, y, z), r, M(m)); s->id = (uint32_t)id; sc->obj[id] = s;
            } else if (op == "moving_sphere") {
                int id = I();
                auto s = std::make_shared<MovingSphere>();
                double a = D(), b = D(), c = D(); s->center0 = Point3(a, b, c);
                a = D(); b = D(); c = D(); s->center1 = Point3(a, b, c);
                s->time0 = D(); s->time1 = D(); s->radius = D(); s->mat_ptr = M(I()); s->id = (uint32_t)id; sc->obj[id] = s;
            } else if (op == "xy_rect") {
                int id = I(); auto q = std::make_shared<XYRect>(); q->x0 = D(); q->x1 = D(); q->y0 = D(); q->y1 = D(); q->k = D(); q->mp = M(I()); q->id = (uint32_t)id; sc->obj[id] = q;
            } else if (op == "xz_rect") {
                int id = I(); auto q = std::make_shared<XZRect>(); q->x0 = D(); q->x1 = D(); q->z0 = D(); q->z1 = D(); q->k = D(); q->mp = M(I()); q->id = (uint32_t)id; sc->obj[id] = q;
            } else if (op == "yz_rect") {
                int id = I(); auto q = std::make_shared<YZRect>(); q->y0 = D(); q->y1 = D(); q->z0 = D(); q->z1 = D(); q->k = D(); q->mp = M(I()); q->id = (uint32_t)id; sc->obj[id] = q;
            } else if (op == "box") {
                int id = I(); double a = D(), b = D(), c = D(), d = D(), e = D(), f = D(); int m = I();
                sc->obj[id] = std::make_shared<Box_>(Point3(a, b, c), Point3(d, e, f), M(m), (uint32_t)id);
            } else if (op == "translate") {
                int id = I(); auto t = std::make_shared<Translate>(); t->ptr = O(I()); double a = D(), b = D(), c = D(); t->offset = Vec3(a, b, c); t->id = (uint32_t)id; sc->obj[id] = t;
            } else if (op == "rotate_y") {
                int id = I(); int child = I(); double angle = D();
                auto r = std::make_shared<RotateY>(O(child), angle, (sc->flags & 1u) != 0); r->id = (uint32_t)id; sc->obj[id] = r;
            } else if (op == "constant_medium") {
                int id = I(); int boundary = I(); double d = D(), r = D(), g = D(), b = D(); int mat_id = I(); int tex_id = I();
                auto m = std::make_shared<ConstantMedium>();
                m->boundary = O(boundary);
                m->neg_inv_density = -1.0 / d; // constant_medium.rs:18
                auto tex = std::make_shared<SolidColor>(Color(r, g, b));
                sc->tex[tex_id] = tex;
                auto iso = std::make_shared<Isotropic>(tex); iso->id = mat_id; sc->mat[mat_id] = iso;
                m->phase_function = iso;
                m->medium_index = sc->n_media++;
                m->id = (uint32_t)id; sc->obj[id] = m;
            } else if (op == "list") { int id = I(); auto lst = std::make_shared<HittableList>(); lst->id = (uint32_t)id; sc->obj[id] = lst; }
            else if (op == "list_add") {
                int lid = I(); int child = I();
                auto lst = std::dynamic_pointer_cast<HittableList>(O(lid));
                if (!lst) throw Panic{"list_add: not a list"};
                lst->add(O(child));
            } else if (op == "bvh") {
                int id = I(); int lid = I(); double t0 = D(), t1 = D();
                std::string axes;
                in >> axes;
                // Which objects a rotated instance's (unrotated, hittable.rs:115) box is unioned with decides what it
                // clips, so the image depends on the split axes drawn. With share_bvh_axes the product's draws are
                // replayed; otherwise the oracle draws its own.
                sc->build_rng.replay = (share_axes && !axes.empty()) ? &axes : nullptr;
                sc->build_rng.replay_pos = 0;
                auto lst = std::dynamic_pointer_cast<HittableList>(O(lid));
                if (!lst || lst->objects.empty()) throw Panic{"bvh: empty or not a list"};
                auto node = std::make_shared<BvhNode>(lst->objects, 0, lst->objects.size(), t0, t1, sc->build_rng); // bvh.rs:68-71
                sc->build_rng.replay = nullptr;
                node->id = (uint32_t)id; sc->obj[id] = node;
            } else if (op == "world") {
                int id = I();
                auto lst = std::dynamic_pointer_cast<HittableList>(O(id));
                if (!lst) { lst = std::make_shared<HittableList>(); lst->add(O(id)); }
                sc->world = lst;
            } else throw Panic{"unknown op: " + op};
        }
    } catch (const Panic& p) {
        err = p.what;
        return nullptr;
    }
    if (!sc->world) { err = "no world line"; return nullptr; }
    return sc.release();
}

static Rng make_rng(uint32_t pixel, uint32_t sample, uint64_t seed, uint32_t compat, Counters* cnt) {
    Rng r; r.pixel = pixel; r.sample = sample; r.bounce = 0; r.seed = seed; r.compat = compat; r.cnt = cnt;
    return r;
}
static Camera make_camera(const rt_camera* c) {
    return Camera(Point3(c->from[0], c->from[1], c->from[2]), Point3(c->at[0], c->at[1], c->at[2]), Vec3(c->vup[0], c->vup[1], c->vup[2]),
                  c->vfov, c->aspect, c->aperture, c->focus, c->time0, c->time1);
}

} // namespace orc

extern "C" {

struct orc_stats {
    uint64_t camera_samples, ray_segments, node_tests, sphere_tests, moving_sphere_tests, rect_tests, medium_tests;
    double seconds;
    uint32_t threads, jobs;
};

void* orc_load(const char* text, size_t n, uint64_t bvh_seed, int share_bvh_axes, char* err, size_t errlen) {
    std::string e;
    orc::Scene* s = orc::load(text, n, bvh_seed, share_bvh_axes != 0, e);
    if (!s && err && errlen) snprintf(err, errlen, "%s", e.c_str());
    return s;
}
void orc_free(void* s) { delete (orc::Scene*)s; }
uint32_t orc_n_media(void* s) { return ((orc::Scene*)s)->n_media; }
// The Perlin tables of NoiseTexture `tex_id`: the ones this file's Perlin::init made (vec: 768 doubles, perm: perm_x, perm_y,
// perm_z, 768 ints) and the ones the constructor log listed. 0 = ok, -1 = no such noise texture.
int orc_perlin_tables(void* s, int tex_id, double* own_vec, int32_t* own_perm, double* listed_vec, int32_t* listed_perm) {
    orc::Scene* sc = (orc::Scene*)s;
    auto it = sc->tex.find(tex_id);
    if (it == sc->tex.end()) return -1;
    const orc::NoiseTexture* t = dynamic_cast<const orc::NoiseTexture*>(it->second.get());
    if (!t) return -1;
    for (int i = 0; i < 256; ++i) { own_vec[3 * i] = t->noise.ranvec[(size_t)i].x; own_vec[3 * i + 1] = t->noise.ranvec[(size_t)i].y; own_vec[3 * i + 2] = t->noise.ranvec[(size_t)i].z; }
    for (int i = 0; i < 256; ++i) { own_perm[i] = t->noise.perm_x[(size_t)i]; own_perm[256 + i] = t->noise.perm_y[(size_t)i]; own_perm[512 + i] = t->noise.perm_z[(size_t)i]; }
    for (int i = 0; i < 768; ++i) { listed_vec[i] = t->listed_vec[(size_t)i]; listed_perm[i] = t->listed_perm[(size_t)i]; }
    return 0;
}

// world.hit for a batch of rays (hittable_list.rs:40-53). counts (nullable): 7 uint64 as in orc_stats order
// from node_tests on, summed over the batch.
int orc_trace_first_hit(void* scene, const rt_ray_d* rays, size_t n, const double* medium_xi, uint32_t compat, rt_hit* out,
                        uint64_t* counts) {
    orc::Scene* sc = (orc::Scene*)scene;
    orc::Counters cnt;
    for (size_t i = 0; i < n; ++i) {
        orc::Rng rng = orc::make_rng(0, 0, 0, compat, &cnt);
        rng.medium_xi = medium_xi ? medium_xi + i * sc->n_media : nullptr;
        rng.skip_media = medium_xi == nullptr;
        orc::Ray r(orc::Point3(rays[i].ox, rays[i].oy, rays[i].oz), orc::Vec3(rays[i].dx, rays[i].dy, rays[i].dz), rays[i].time);
        orc::HitRecord rec;
        rt_hit o;
        memset(&o, 0, sizeof o);
        if (sc->world->hit(r, rays[i].tmin, rays[i].tmax, rec, rng)) {
            o.t = rec.t; o.prim_id = rec.prim_id;
            o.flags = RT_HIT_FLAG_HIT | (rec.front_face ? RT_HIT_FLAG_FRONT : 0u) | (rec.face << 4);
            o.px = rec.p.x; o.py = rec.p.y; o.pz = rec.p.z; o.nx = rec.normal.x; o.ny = rec.normal.y; o.nz = rec.normal.z;
            o.u = rec.u; o.v = rec.v; o.material_id = rec.material_id;
        } else {
            o.prim_id = RT_MISS; o.t = rays[i].tmax;
        }
        out[i] = o;
    }
    if (counts) { counts[0] = cnt.nodes; counts[1] = cnt.spheres; counts[2] = cnt.msphere; counts[3] = cnt.rects; counts[4] = cnt.media; }
    return 0;
}

// The body of the sample loop up to the ray_color call (main.rs:162-164) for one sample of every pixel.
int orc_generate_rays(const rt_camera* cam, uint32_t width, uint32_t height, uint32_t sample, uint64_t seed, uint32_t compat, rt_ray_d* out) {
    orc::Camera c = orc::make_camera(cam);
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            orc::Rng rng = orc::make_rng(y * width + x, sample, seed, compat, nullptr);
            double jx, jy;
            rng.draw2(orc::S_JITTER, 0, jx, jy);
            double u_ = ((double)x + jx) / ((double)width - 1.0);
            double v_ = ((double)y + jy) / ((double)height - 1.0);
            orc::Ray r = c.get_ray(u_, v_, rng);
            rt_ray_d o{r.orig.x, r.orig.y, r.orig.z, r.dir.x, r.dir.y, r.dir.z, r.tm, 0.001, orc::INFINITY_};
            out[(size_t)y * width + x] = o;
        }
    return 0;
}

// main.rs:143-195: n_jobs row bands handed to n_threads workers; inside a band x is the outer loop and y the
// inner one; each pixel sums samples [sample_offset, sample_offset + spp) in order into rgb_sum (+=).
// x range [x_begin, x_end) lets a caller bound the work.
int orc_render(void* scene, const rt_camera* cam, uint32_t width, uint32_t height, uint32_t max_depth, const double background[3],
               uint32_t background_mode, uint64_t seed, uint32_t compat, uint32_t sample_offset, uint32_t spp, uint32_t n_threads,
               uint32_t n_jobs, double* rgb_sum, orc_stats* stats) {
    orc::Scene* sc = (orc::Scene*)scene;
    orc::Camera c = orc::make_camera(cam);
    if (n_threads == 0) n_threads = std::max(1u, std::thread::hardware_concurrency());
    if (n_jobs == 0) n_jobs = 8 * n_threads;
    n_jobs = std::min(n_jobs, height);
    std::atomic<uint32_t> next_job(0);
    std::vector<orc::Counters> per_thread(n_threads);
    const orc::Color bg(background[0], background[1], background[2]);
    const auto t0 = std::chrono::steady_clock::now();
    auto worker = [&](uint32_t tid) {
        orc::Counters& cnt = per_thread[tid];
        for (;;) {
            uint32_t i = next_job.fetch_add(1);
            if (i >= n_jobs) break;
            uint32_t row_begin = (uint32_t)((uint64_t)height * i / n_jobs), row_end = (uint32_t)((uint64_t)height * (i + 1) / n_jobs);
            for (uint32_t x = 0; x < width; ++x)
                for (uint32_t y = row_begin; y < row_end; ++y) {
                    orc::Color pixel_color(rgb_sum[((size_t)y * width + x) * 3], rgb_sum[((size_t)y * width + x) * 3 + 1],
                                           rgb_sum[((size_t)y * width + x) * 3 + 2]);
                    for (uint32_t s = 0; s < spp; ++s) {
                        orc::Rng rng = orc::make_rng(y * width + x, sample_offset + s, seed, compat, &cnt);
                        double jx, jy;
                        rng.draw2(orc::S_JITTER, 0, jx, jy);
                        double u_ = ((double)x + jx) / ((double)width - 1.0);
                        double v_ = ((double)y + jy) / ((double)height - 1.0);
                        orc::Ray r_ = c.get_ray(u_, v_, rng);
                        pixel_color = pixel_color + orc::ray_color(r_, bg, background_mode, *sc->world, (int)max_depth, (int)max_depth, rng);
                        cnt.samples++;
                    }
                    rgb_sum[((size_t)y * width + x) * 3] = pixel_color.x;
                    rgb_sum[((size_t)y * width + x) * 3 + 1] = pixel_color.y;
                    rgb_sum[((size_t)y * width + x) * 3 + 2] = pixel_color.z;
                }
        }
    };
    std::vector<std::thread> pool;
    for (uint32_t t = 1; t < n_threads; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto& t : pool) t.join();
    if (stats) {
        memset(stats, 0, sizeof *stats);
        for (const orc::Counters& k : per_thread) {
            stats->camera_samples += k.samples; stats->ray_segments += k.segments; stats->node_tests += k.nodes;
            stats->sphere_tests += k.spheres; stats->moving_sphere_tests += k.msphere; stats->rect_tests += k.rects;
            stats->medium_tests += k.media;
        }
        stats->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        stats->threads = n_threads; stats->jobs = n_jobs;
    }
    return 0;
}

// One sample's ray_color (main.rs:165), for per-sample comparisons.
int orc_sample_radiance(void* scene, const rt_camera* cam, uint32_t width, uint32_t height, uint32_t max_depth,
                        const double background[3], uint32_t background_mode, uint64_t seed, uint32_t compat, uint32_t x, uint32_t y,
                        uint32_t sample, double rgb[3]) {
    orc::Scene* sc = (orc::Scene*)scene;
    orc::Camera c = orc::make_camera(cam);
    orc::Rng rng = orc::make_rng(y * width + x, sample, seed, compat, nullptr);
    double jx, jy;
    rng.draw2(orc::S_JITTER, 0, jx, jy);
    orc::Ray r_ = c.get_ray(((double)x + jx) / ((double)width - 1.0), ((double)y + jy) / ((double)height - 1.0), rng);
    orc::Color col = orc::ray_color(r_, orc::Color(background[0], background[1], background[2]), background_mode, *sc->world,
                                    (int)max_depth, (int)max_depth, rng);
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    return 0;
}

// main.rs:172-189
void orc_tonemap(const double* rgb_sum, size_t n_pixels, double samples_per_pixel, uint8_t* out) {
    for (size_t i = 0; i < n_pixels * 3; ++i) {
        double c = std::sqrt((1.0 / samples_per_pixel) * rgb_sum[i]);
        c = orc::clamp(c, 0.0, 0.999);
        c = c * 255.999;
        long long q = std::isnan(c) ? 0 : (long long)c;
        out[i] = (uint8_t)q;
    }
}

void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    orc::U4 o = orc::philox(orc::U4{ctr[0], ctr[1], ctr[2], ctr[3]}, key[0], key[1]);
    out[0] = o.a; out[1] = o.b; out[2] = o.c; out[3] = o.d;
}

// The reference's own (commented-out) Vec3 assertions, raytracer/src/vec3.rs:263-397. Returns the number
// of failed checks (0 = all pass).
int orc_vec3_kat(void) {
    using orc::Vec3;
    int bad = 0;
#define CHECK(c) do { if (!(c)) ++bad; } while (0)
    CHECK(Vec3(1.0, 2.0, 3.0) == Vec3(1.0, 2.0, 3.0));                                         // test_new
    CHECK(Vec3(1.0, 0.0, -1.0) + Vec3(2.0, 4.0, 6.0) == Vec3(3.0, 4.0, 5.0));                  // test_add
    { Vec3 x(1.0, 0.0, -1.0); x = x + Vec3(2.0, 4.0, 6.0); CHECK(x == Vec3(3.0, 4.0, 5.0)); } // test_add_assign
    CHECK(Vec3(1.0, 0.0, -1.0) + 233.0 == Vec3(234.0, 233.0, 232.0));                          // test_add_f64
    { Vec3 x(1.0, 0.0, -1.0); x = x + 233.0; CHECK(x == Vec3(234.0, 233.0, 232.0)); }         // test_add_assign_f64
    CHECK(Vec3(1.0, 0.0, -1.0) - Vec3(2.0, 4.0, 6.0) == Vec3(-1.0, -4.0, -7.0));               // test_sub
    { Vec3 x(1.0, 0.0, -1.0); x = x - Vec3(2.0, 4.0, 6.0); CHECK(x == Vec3(-1.0, -4.0, -7.0)); } // test_sub_assign
    CHECK(Vec3(1.0, 0.0, -1.0) - 1.0 == Vec3(0.0, -1.0, -2.0));                                // test_sub_f64
    { Vec3 x(1.0, 0.0, -1.0); x = x - 1.0; CHECK(x == Vec3(0.0, -1.0, -2.0)); }               // test_sub_assign_f64
    { Vec3 x(1.0, 0.0, -1.0); x = x * 2.0; CHECK(x == Vec3(2.0, 0.0, -2.0)); }                // test_mul_assign
    CHECK(Vec3(1.0, 0.0, -1.0) * 1.0 == Vec3(1.0, 0.0, -1.0));                                 // test_mul_f64
    CHECK(Vec3(1.0, -2.0, 0.0) / 2.0 == Vec3(0.5, -1.0, 0.0));                                 // test_div
    CHECK(Vec3::cross(Vec3(1.0, 2.0, 3.0), Vec3(2.0, 3.0, 4.0)) == Vec3(8.0 - 9.0, 6.0 - 4.0, 3.0 - 4.0)); // test_cross
    CHECK(-Vec3(1.0, -2.0, 3.0) == Vec3(-1.0, 2.0, -3.0));                                     // test_neg
    CHECK(Vec3(1.0, 2.0, 3.0).squared_length() == 14.0);                                       // test_squared_length
    CHECK(Vec3(3.0, 4.0, 5.0).length() == std::sqrt(3.0 * 3.0 + 4.0 * 4.0 + 5.0 * 5.0));       // test_length
    CHECK(Vec3(233.0, 0.0, 0.0).unit() == Vec3(1.0, 0.0, 0.0));                                // test_unit
    CHECK(Vec3(-233.0, 0.0, 0.0).unit() == Vec3(-1.0, 0.0, 0.0));
    CHECK(Vec3(0.0, 0.0, 0.0).length() == 0.0);                                                // test_unit_panic: the panic condition
#undef CHECK
    return bad;
}

} // extern "C"
