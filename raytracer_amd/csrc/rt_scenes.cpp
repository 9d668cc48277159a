// The C++ mirror of the reference's constructors (rt_scene.hpp) and the ready-made scenes:
// the reference's scene functions (raytracer/src/main.rs:260-835) plus two scaled-up variants.
#include <cmath>
#include <cstring>
#include <vector>

#include "rt_builder.h"
#include "rt_scene.hpp"

namespace rt {

static thread_local rt_builder* g_builder = nullptr;

SceneScope::SceneScope(rt_builder* b) : prev_(g_builder) { g_builder = b; }
SceneScope::~SceneScope() { g_builder = prev_; }
rt_builder* SceneScope::current() {
    if (!g_builder) throw SceneError(RT_ERR_INVALID, "no rt::SceneScope is active on this thread");
    return g_builder;
}
int check(int v) {
    if (v < 0) throw SceneError(v, rt_builder_last_error(g_builder));
    return v;
}
static rt_builder* B() { return SceneScope::current(); }

double Vec3::length() const { return std::sqrt(squared_length()); }
Vec3 Vec3::unit() const { // vec3.rs:50-55
    if (length() == 0.0) throw SceneError(RT_ERR_PANIC, "div 0!");
    return *this / length();
}

Texture SolidColor::new_(Color c) { return Texture{check(rt_tex_solid(B(), c.x, c.y, c.z))}; }
Texture CheckerTexture::new_(Texture even, Texture odd) { return Texture{check(rt_tex_checker(B(), even.id, odd.id))}; }
Texture CheckerTexture::new_by_color(Color c1, Color c2) { return new_(SolidColor::new_(c1), SolidColor::new_(c2)); }
Texture NoiseTexture::new_(double sc) { return Texture{check(rt_tex_noise(B(), sc))}; }
Texture ImageTexture::new_(const char* filename) {
    rt_builder* b = B();
    auto it = b->images.find(filename);
    if (it == b->images.end()) // image::open(filename).unwrap() panics, texture.rs:91
        throw SceneError(RT_ERR_PANIC, std::string("image not registered: ") + filename);
    return Texture{check(rt_tex_image(b, it->second.rgb.data(), it->second.width, it->second.height))};
}

Material Lambertian::new_(Color a) { return new_by_pointer(SolidColor::new_(a)); }
Material Lambertian::new_by_pointer(Texture a) { return Material{check(rt_mat_lambertian(B(), a.id))}; }
Material Metal::new_(Color a, double f) { return Material{check(rt_mat_metal(B(), a.x, a.y, a.z, f))}; }
Material Dielectric::new_(double i) { return Material{check(rt_mat_dielectric(B(), i))}; }
Material DiffuseLight::new_by_color(Color c) { return new_by_texture(SolidColor::new_(c), 1.0); }
Material DiffuseLight::new_by_texture(Texture emit, double k) { return Material{check(rt_mat_diffuse_light(B(), emit.id, k))}; }

Hittable Sphere::new_(Point3 c, double r, Material m) {
    const double cc[3] = {c.x, c.y, c.z};
    return Hittable{check(rt_sphere(B(), cc, r, m.id))};
}
Hittable MovingSphere::new_(Point3 c0, Point3 c1, double t0, double t1, double r, Material m) {
    const double a[3] = {c0.x, c0.y, c0.z}, b[3] = {c1.x, c1.y, c1.z};
    return Hittable{check(rt_moving_sphere(B(), a, b, t0, t1, r, m.id))};
}
Hittable XYRect::new_(double x0, double x1, double y0, double y1, double k, Material m) { return Hittable{check(rt_xy_rect(B(), x0, x1, y0, y1, k, m.id))}; }
Hittable XZRect::new_(double x0, double x1, double z0, double z1, double k, Material m) { return Hittable{check(rt_xz_rect(B(), x0, x1, z0, z1, k, m.id))}; }
Hittable YZRect::new_(double y0, double y1, double z0, double z1, double k, Material m) { return Hittable{check(rt_yz_rect(B(), y0, y1, z0, z1, k, m.id))}; }
Hittable _Box::new_(Point3 p0, Point3 p1, Material m) {
    const double a[3] = {p0.x, p0.y, p0.z}, b[3] = {p1.x, p1.y, p1.z};
    return Hittable{check(rt_box(B(), a, b, m.id))};
}
Hittable Translate::new_(Hittable p, Vec3 d) {
    const double o[3] = {d.x, d.y, d.z};
    return Hittable{check(rt_translate(B(), p.id, o))};
}
Hittable RotateY::new_(Hittable p, double angle) { return Hittable{check(rt_rotate_y(B(), p.id, angle))}; }
Hittable ConstantMedium::new_by_color(Hittable b, double d, Color c) {
    return Hittable{check(rt_constant_medium(B(), b.id, d, c.x, c.y, c.z))};
}
HittableList HittableList::new_default() { return HittableList{check(rt_list_new(B()))}; }
HittableList HittableList::new_(Hittable object) {
    HittableList l = new_default();
    l.add(object);
    return l;
}
void HittableList::add(Hittable object) { check(rt_list_add(B(), id, object.id)); }
Hittable BvhNode::new_(HittableList& list, double t0, double t1) { return Hittable{check(rt_bvh(B(), list.id, t0, t1))}; }

rt_camera Camera::new_(Point3 from, Point3 at, Vec3 vup, double vfov, double aspect, double aperture, double focus,
                       double t0, double t1) {
    rt_camera c;
    c.from[0] = from.x; c.from[1] = from.y; c.from[2] = from.z;
    c.at[0] = at.x; c.at[1] = at.y; c.at[2] = at.z;
    c.vup[0] = vup.x; c.vup[1] = vup.y; c.vup[2] = vup.z;
    c.vfov = vfov; c.aspect = aspect; c.aperture = aperture; c.focus = focus; c.time0 = t0; c.time1 = t1;
    return c;
}

} // namespace rt

// Camera::new (camera.rs:30-53): the derived frame the reference's Camera struct stores.
int rt_camera_derive(const rt_camera* c, RtCameraDev* out) {
    if (!c || !out) return RT_ERR_INVALID;
    if (!(c->time0 < c->time1)) return RT_ERR_PANIC; // gen_range on an empty range panics (camera.rs:62)
    const V3 lookfrom = v3(c->from[0], c->from[1], c->from[2]), lookat = v3(c->at[0], c->at[1], c->at[2]);
    const V3 vup = v3(c->vup[0], c->vup[1], c->vup[2]);
    const double theta = c->vfov * RT_PI / 180.0;
    const double h = std::tan(theta / 2.0);
    const double viewport_height = 2.0 * h;
    const double viewport_width = c->aspect * viewport_height;
    const V3 d = lookfrom - lookat;
    if (v3_len(d) == 0.0) return RT_ERR_PANIC;
    const V3 w1 = d / v3_len(d);
    const V3 cr = v3_cross(vup, w1);
    if (v3_len(cr) == 0.0) return RT_ERR_PANIC;
    const V3 u1 = cr / v3_len(cr);
    const V3 v1 = v3_cross(w1, u1);
    const V3 hor = u1 * viewport_width * c->focus;
    const V3 ver = v1 * viewport_height * c->focus;
    const V3 llc = lookfrom - hor * 0.5 - ver * 0.5 - w1 * c->focus;
    const V3* src[6] = {&lookfrom, &llc, &hor, &ver, &u1, &v1};
    double* dst[6] = {out->origin, out->lower_left_corner, out->horizontal, out->vertical, out->u, out->v};
    for (int i = 0; i < 6; ++i) { dst[i][0] = src[i]->x; dst[i][1] = src[i]->y; dst[i][2] = src[i]->z; }
    out->lens_radius = c->aperture / 2.0;
    out->time0 = c->time0;
    out->time1 = c->time1;
    return RT_OK;
}

// ---- scenes -------------------------------------------------------------------------------------------------
namespace {
using namespace rt;

struct SceneRng { // the draws the reference's scene functions take from thread_rng()
    SeqRng r;
    double gen() { return r.next(); }                                  // rng.gen::<f64>()
    double random_double(double lo, double hi) { return r.range(lo, hi); }
    Vec3 random() { double x = gen(), y = gen(), z = gen(); return Vec3{x, y, z}; }            // vec3.rs:57-63
    Vec3 random_range(double lo, double hi) {                                                   // vec3.rs:65-71
        double x = r.range(lo, hi), y = r.range(lo, hi), z = r.range(lo, hi);
        return Vec3{x, y, z};
    }
};

void info_image(rt_scene_info* info, uint32_t w, uint32_t h, uint32_t spp) {
    info->width = w; info->height = h; info->spp = spp; info->max_depth = 50;
    info->background[0] = info->background[1] = info->background[2] = 0.0;
    info->background_mode = 0;
}
// The Cornell / final-scene camera of "Ray Tracing: The Next Week" (the reference defines none for them).
rt_camera book2_camera() {
    return Camera::new_(Point3{278, 278, -800}, Point3{278, 278, 0}, Vec3{0, 1, 0}, 40.0, 1.0, 0.0, 10.0, 0.0, 1.0);
}

// main.rs:576-688
HittableList final_scene(SceneRng& rng) {
    HittableList boxes1 = HittableList::new_default();
    Material ground = Lambertian::new_(Color{0.48, 0.83, 0.53});
    const int boxes_per_side = 20;
    for (int i = 0; i < boxes_per_side; ++i)
        for (int j = 0; j < boxes_per_side; ++j) {
            double w = 100.0;
            double x0 = -1000.0 + i * w, z0 = -1000.0 + j * w, y0 = 0.0;
            double x1 = x0 + w, y1 = rng.random_double(1.0, 101.0), z1 = z0 + w;
            boxes1.add(_Box::new_(Point3{x0, y0, z0}, Point3{x1, y1, z1}, ground));
        }
    HittableList objects = HittableList::new_default();
    objects.add(BvhNode::new_(boxes1, 0.0, 1.0));

    Material light = DiffuseLight::new_by_color(Color{7.0, 7.0, 7.0});
    objects.add(XZRect::new_(123.0, 423.0, 147.0, 412.0, 554.0, light));

    Point3 center1{400.0, 400.0, 200.0};
    Point3 center2 = center1 + Vec3{30.0, 0.0, 0.0};
    Material moving_sphere_material = Lambertian::new_(Color{0.7, 0.3, 0.1});
    objects.add(MovingSphere::new_(center1, center2, 0.0, 1.0, 50.0, moving_sphere_material));

    objects.add(Sphere::new_(Point3{260.0, 150.0, 45.0}, 50.0, Dielectric::new_(1.5)));
    objects.add(Sphere::new_(Point3{0.0, 150.0, 145.0}, 50.0, Metal::new_(Color{0.8, 0.8, 0.9}, 1.0)));

    Hittable boundary = Sphere::new_(Point3{360.0, 150.0, 145.0}, 70.0, Dielectric::new_(1.5));
    objects.add(boundary);
    objects.add(ConstantMedium::new_by_color(boundary, 0.2, Color{0.2, 0.4, 0.9}));
    boundary = Sphere::new_(Point3{0.0, 0.0, 0.0}, 5000.0, Dielectric::new_(1.5));
    objects.add(ConstantMedium::new_by_color(boundary, 0.0001, Color{1.0, 1.0, 1.0}));

    Material emat = Lambertian::new_by_pointer(ImageTexture::new_("earthmap.jpg"));
    objects.add(Sphere::new_(Point3{400.0, 200.0, 400.0}, 100.0, emat));
    Texture pertext = NoiseTexture::new_(0.1);
    objects.add(Sphere::new_(Point3{220.0, 280.0, 300.0}, 80.0, Lambertian::new_by_pointer(pertext)));

    HittableList boxes2 = HittableList::new_default();
    Material white = Lambertian::new_(Color{0.73, 0.73, 0.73});
    const int ns = 1000;
    for (int j = 0; j < ns; ++j) boxes2.add(Sphere::new_(rng.random_range(0.0, 165.0), 10.0, white));
    objects.add(Translate::new_(RotateY::new_(BvhNode::new_(boxes2, 0.0, 1.0), 15.0), Vec3{-100.0, 270.0, 395.0}));
    return objects;
}

// The eight objects shared by cornell_box (main.rs:409-486) and cornell_smoke (main.rs:488-574).
void cornell_walls(HittableList& objects, double light_x0, double light_x1, double light_z0, double light_z1,
                   double emit, Material& white_out) {
    Material red = Lambertian::new_(Color{0.65, 0.05, 0.05});
    Material white = Lambertian::new_(Color{0.73, 0.73, 0.73});
    Material green = Lambertian::new_(Color{0.12, 0.45, 0.15});
    Material light = DiffuseLight::new_by_color(Color{emit, emit, emit});
    objects.add(YZRect::new_(0.0, 555.0, 0.0, 555.0, 555.0, green));
    objects.add(YZRect::new_(0.0, 555.0, 0.0, 555.0, 0.0, red));
    objects.add(XZRect::new_(light_x0, light_x1, light_z0, light_z1, 554.0, light));
    objects.add(XZRect::new_(0.0, 555.0, 0.0, 555.0, 0.0, white));
    objects.add(XZRect::new_(0.0, 555.0, 0.0, 555.0, 555.0, white));
    objects.add(XYRect::new_(0.0, 555.0, 0.0, 555.0, 555.0, white));
    white_out = white;
}
Hittable cornell_box() { // returns the BvhNode the reference returns; the caller wraps it in a list
    HittableList objects = HittableList::new_default();
    Material white;
    cornell_walls(objects, 213.0, 343.0, 227.0, 332.0, 15.0, white);
    Hittable box1 = _Box::new_(Point3{0, 0, 0}, Point3{165.0, 330.0, 165.0}, white);
    box1 = RotateY::new_(box1, 15.0);
    box1 = Translate::new_(box1, Vec3{265.0, 0.0, 295.0});
    objects.add(box1);
    Hittable box2 = _Box::new_(Point3{0, 0, 0}, Point3{165.0, 165.0, 165.0}, white);
    box2 = RotateY::new_(box2, -18.0);
    box2 = Translate::new_(box2, Vec3{130.0, 0.0, 65.0});
    objects.add(box2);
    return BvhNode::new_(objects, 0.0, 0.0);
}
HittableList cornell_smoke() {
    HittableList objects = HittableList::new_default();
    Material white;
    cornell_walls(objects, 113.0, 443.0, 127.0, 432.0, 7.0, white);
    Hittable box1 = _Box::new_(Point3{0, 0, 0}, Point3{165.0, 330.0, 165.0}, white);
    box1 = RotateY::new_(box1, 15.0);
    box1 = Translate::new_(box1, Vec3{265.0, 0.0, 295.0});
    Hittable box2 = _Box::new_(Point3{0, 0, 0}, Point3{165.0, 165.0, 165.0}, white);
    box2 = RotateY::new_(box2, -18.0);
    box2 = Translate::new_(box2, Vec3{130.0, 0.0, 65.0});
    objects.add(ConstantMedium::new_by_color(box1, 0.01, Color{0.0, 0.0, 0.0}));
    objects.add(ConstantMedium::new_by_color(box2, 0.01, Color{1.0, 1.0, 1.0}));
    return objects;
}

// One cell of the random-spheres grid (main.rs:279-304).
void random_cell(SceneRng& rng, HittableList& world, double a, double b, double y_of_centre, bool skip_near_big) {
    double choose_mat = rng.gen();
    double cx = a + 0.9 * rng.gen();
    double cz = b + 0.9 * rng.gen();
    Point3 center{cx, y_of_centre, cz};
    if (!skip_near_big || (center - Point3{4.0, 0.2, 0.0}).length() > 0.9) {
        if (choose_mat < 0.8) {
            Color albedo = rng.random() * rng.random();
            Material m = Lambertian::new_(albedo);
            Point3 center2 = center + Vec3{0.0, rng.random_double(0.0, 0.5), 0.0};
            world.add(MovingSphere::new_(center, center2, 0.0, 1.0, 0.2, m));
        } else if (choose_mat < 0.95) {
            Color albedo = rng.random_range(0.5, 1.0);
            double fuzz = rng.random_double(0.0, 0.5);
            world.add(Sphere::new_(center, 0.2, Metal::new_(albedo, fuzz)));
        } else {
            world.add(Sphere::new_(center, 0.2, Dielectric::new_(1.5)));
        }
    }
}
// main.rs:260-331 (commented out there; live in /root/reference/main.rs:143-216)
Hittable random_scene(SceneRng& rng) {
    HittableList world = HittableList::new_default();
    Texture checker = CheckerTexture::new_by_color(Color{0.2, 0.3, 0.1}, Color{0.9, 0.9, 0.9});
    world.add(Sphere::new_(Point3{0.0, -1000.0, 0.0}, 1000.0, Lambertian::new_by_pointer(checker)));
    for (double a = -11.0; a < 11.0; a += 1.0)
        for (double b = -11.0; b < 11.0; b += 1.0) random_cell(rng, world, a, b, 0.2, true);
    world.add(Sphere::new_(Point3{0.0, 1.0, 0.0}, 1.0, Dielectric::new_(1.5)));
    world.add(Sphere::new_(Point3{-4.0, 1.0, 0.0}, 1.0, Lambertian::new_(Color{0.4, 0.2, 0.1})));
    world.add(Sphere::new_(Point3{4.0, 1.0, 0.0}, 1.0, Metal::new_(Color{0.7, 0.6, 0.5}, 0.0)));
    return BvhNode::new_(world, 0.0, 1.0);
}
Hittable two_spheres() { // main.rs:333-351
    HittableList objects = HittableList::new_default();
    Texture checker = CheckerTexture::new_by_color(Color{0.2, 0.3, 0.1}, Color{0.9, 0.9, 0.9});
    objects.add(Sphere::new_(Vec3{0.0, -10.0, 0.0}, 10.0, Lambertian::new_by_pointer(checker)));
    objects.add(Sphere::new_(Vec3{0.0, 10.0, 0.0}, 10.0, Lambertian::new_by_pointer(checker)));
    return BvhNode::new_(objects, 0.0, 0.0);
}
Hittable two_perlin_spheres() { // main.rs:353-368
    HittableList objects = HittableList::new_default();
    Texture pertext = NoiseTexture::new_(4.0);
    objects.add(Sphere::new_(Vec3{0.0, -1000.0, 0.0}, 1000.0, Lambertian::new_by_pointer(pertext)));
    objects.add(Sphere::new_(Vec3{0.0, 2.0, 0.0}, 2.0, Lambertian::new_by_pointer(pertext)));
    return BvhNode::new_(objects, 0.0, 0.0);
}
HittableList earth() { // main.rs:370-380
    Material earth_surface = Lambertian::new_by_pointer(ImageTexture::new_("earthmap.jpg"));
    HittableList objects = HittableList::new_default();
    objects.add(Sphere::new_(Vec3{0.0, 0.0, 0.0}, 2.0, earth_surface));
    return objects;
}
Hittable simple_light() { // main.rs:382-407
    HittableList objects = HittableList::new_default();
    Texture pertext = NoiseTexture::new_(4.0);
    objects.add(Sphere::new_(Vec3{0.0, -1000.0, 0.0}, 1000.0, Lambertian::new_by_pointer(pertext)));
    objects.add(Sphere::new_(Vec3{0.0, 2.0, 0.0}, 2.0, Lambertian::new_by_pointer(pertext)));
    Material difflight = DiffuseLight::new_by_color(Color{4.0, 4.0, 4.0});
    objects.add(XYRect::new_(3.0, 5.0, 1.0, 3.0, -2.0, difflight));
    return BvhNode::new_(objects, 0.0, 0.0);
}
// main.rs:690-835, the scene the reference's main() actually renders. Needs the four star images
// registered under the names material.rs:165-188 opens.
HittableList my_scene(SceneRng& rng) {
    HittableList objects = HittableList::new_default();
    Material star1 = DiffuseLight::new_by_texture(ImageTexture::new_("raytracer/star1.jpg"), 1.5);
    objects.add(_Box::new_(Point3{-4.0, -7.0, -7.0}, Point3{10.0, 7.0, 7.0}, star1));
    objects.add(_Box::new_(Point3{-6.0, -9.0, -9.0}, Point3{12.0, 9.0, 9.0}, Dielectric::new_(3.0)));
    Material star3 = DiffuseLight::new_by_texture(ImageTexture::new_("raytracer/star3.jpeg"), 1.5);
    objects.add(Sphere::new_(Point3{-14.0, 8.0, 10.0}, 3.0, star3));
    objects.add(Sphere::new_(Point3{15.0, 7.0, -7.0}, 2.0, star3));
    objects.add(Sphere::new_(Point3{18.0, 11.0, 18.0}, 4.0, star3));
    objects.add(Sphere::new_(Point3{22.0, -13.0, 15.0}, 1.5, star3));
    objects.add(Sphere::new_(Point3{-10.0, -20.0, 17.0}, 1.5, star3));
    objects.add(Sphere::new_(Point3{17.0, -17.0, 17.0}, 0.5, star3));
    objects.add(Sphere::new_(Point3{-10.0, -9.0, -13.0}, 2.0, star3));
    objects.add(XZRect::new_(-100.0, 100.0, -100.0, 100.0, 10.0, Metal::new_(Color{0.9, 0.9, 0.9}, 0.0)));
    objects.add(XYRect::new_(-100.0, 100.0, -100.0, 100.0, -20.0, Metal::new_(Color{0.9, 0.9, 0.9}, 0.0)));
    Material star5 = DiffuseLight::new_by_texture(ImageTexture::new_("raytracer/star5.jpeg"), 0.5);
    Material star6 = DiffuseLight::new_by_texture(ImageTexture::new_("raytracer/star6.jpeg"), 0.5);
    objects.add(Sphere::new_(Point3{16.0, -10.0, 15.0}, 5.0, star5));
    HittableList fogs = HittableList::new_default();
    auto disk_unit = [&]() { // Vec3::random_in_unit_disk().unit(), vec3.rs:102-110
        for (;;) {
            Vec3 p{rng.random_double(-1.0, 1.0), 0.0, rng.random_double(-1.0, 1.0)};
            if (p.squared_length() < 1.0) return p.unit();
        }
    };
    const int ns = 10000;
    for (int j = 0; j < ns; ++j) {
        Vec3 d = disk_unit();
        fogs.add(Sphere::new_(Point3{16.0, -10.0, 15.0} + d * 10.0 * rng.random_double(0.8, 1.0), 0.1, star6));
        d = disk_unit();
        fogs.add(Sphere::new_(Point3{16.0, -10.0, 15.0} + d * 9.5 * rng.random_double(0.7, 1.0), 0.1, star5));
        d = disk_unit();
        fogs.add(Sphere::new_(Point3{16.0, -10.0, 15.0} + d * 8.0 * rng.random_double(0.8, 1.0), 0.1, star6));
    }
    objects.add(BvhNode::new_(fogs, 0.0, 0.0));
    return objects;
}

// Scaled random-spheres: the random_scene material mix on a side x side jittered grid laid on the
// r = 1000 ground sphere, one BvhNode over everything.
Hittable spheres_grid(SceneRng& rng, uint64_t n) {
    HittableList world = HittableList::new_default();
    Texture checker = CheckerTexture::new_by_color(Color{0.2, 0.3, 0.1}, Color{0.9, 0.9, 0.9});
    world.add(Sphere::new_(Point3{0.0, -1000.0, 0.0}, 1000.0, Lambertian::new_by_pointer(checker)));
    long side = (long)std::llround(std::sqrt((double)n));
    if (side < 1) side = 1;
    const double half = (double)side / 2.0;
    for (long i = 0; i < side; ++i)
        for (long j = 0; j < side; ++j) {
            double a = (double)i - half, b = (double)j - half;
            // rest the sphere on the curved ground below (a + 0.45, b + 0.45)
            double gx = a + 0.45, gz = b + 0.45;
            double y = std::sqrt(std::max(0.0, 1000.0 * 1000.0 - gx * gx - gz * gz)) - 1000.0 + 0.2;
            random_cell(rng, world, a, b, y, false);
        }
    return BvhNode::new_(world, 0.0, 1.0);
}
// Texture/volume-heavy stress scene: Perlin-marble ground, 64 image-textured and Perlin spheres,
// one ceiling light, all inside a thin global ConstantMedium.
HittableList stress_volume(SceneRng& rng) {
    HittableList objects = HittableList::new_default();
    Texture marble = NoiseTexture::new_(4.0);
    objects.add(Sphere::new_(Point3{0.0, -1000.0, 0.0}, 1000.0, Lambertian::new_by_pointer(marble)));
    Material earth_m = Lambertian::new_by_pointer(ImageTexture::new_("earthmap.jpg"));
    Material perlin_m = Lambertian::new_by_pointer(NoiseTexture::new_(2.0));
    HittableList balls = HittableList::new_default();
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 8; ++j) {
            double r = rng.random_double(0.5, 2.0);
            double x = -14.0 + 4.0 * i + rng.random_double(0.0, 1.5), z = -14.0 + 4.0 * j + rng.random_double(0.0, 1.5);
            balls.add(Sphere::new_(Point3{x, r, z}, r, ((i + j) & 1) ? earth_m : perlin_m));
        }
    objects.add(BvhNode::new_(balls, 0.0, 1.0));
    objects.add(XZRect::new_(-10.0, 10.0, -10.0, 10.0, 25.0, DiffuseLight::new_by_color(Color{7.0, 7.0, 7.0})));
    Hittable boundary = Sphere::new_(Point3{0.0, 0.0, 0.0}, 5000.0, Dielectric::new_(1.5));
    objects.add(ConstantMedium::new_by_color(boundary, 0.0001, Color{1.0, 1.0, 1.0}));
    return objects;
}

} // namespace

extern "C" int rt_build_named_scene(rt_builder* b, const char* name, uint64_t scene_seed, uint64_t n, rt_scene_info* info) {
    if (!b) return RT_ERR_INVALID;
    if (!name || !info) return b->fail(RT_ERR_INVALID, "named scene: null argument");
    SceneRng rng;
    rng.r.seed = scene_seed;
    rng.r.stream = 2;
    memset(info, 0, sizeof *info);
    try {
        rt::SceneScope scope(b);
        const std::string s = name;
        if (s == "final_scene") {
            info->world = final_scene(rng).id;
            info->camera = book2_camera();
            info_image(info, 800, 800, 10000);
        } else if (s == "cornell_box") {
            info->world = HittableList::new_(cornell_box()).id;
            info->camera = book2_camera();
            info_image(info, 600, 600, 1000);
        } else if (s == "cornell_smoke") {
            info->world = cornell_smoke().id;
            info->camera = book2_camera();
            info_image(info, 600, 600, 1000);
        } else if (s == "random_scene") {
            info->world = HittableList::new_(random_scene(rng)).id;
            // /root/reference/main.rs:76-91 (lookfrom x = 12, not the book's 13)
            info->camera = Camera::new_(Point3{12.0, 2.0, 3.0}, Point3{0, 0, 0}, Vec3{0, 1, 0}, 20.0, 1200.0 / 800.0, 0.1, 10.0, 0.0, 1.0);
            info_image(info, 1200, 800, 100);
            info->background_mode = 1;
        } else if (s == "two_spheres" || s == "two_perlin_spheres" || s == "earth" || s == "simple_light") {
            if (s == "two_spheres") info->world = HittableList::new_(two_spheres()).id;
            else if (s == "two_perlin_spheres") info->world = HittableList::new_(two_perlin_spheres()).id;
            else if (s == "earth") info->world = earth().id;
            else info->world = HittableList::new_(simple_light()).id;
            const bool lit = s == "simple_light";
            info->camera = lit ? Camera::new_(Point3{26.0, 3.0, 6.0}, Point3{0.0, 2.0, 0.0}, Vec3{0, 1, 0}, 20.0, 1.5, 0.0, 10.0, 0.0, 1.0)
                               : Camera::new_(Point3{13.0, 2.0, 3.0}, Point3{0, 0, 0}, Vec3{0, 1, 0}, 20.0, 1.5, 0.0, 10.0, 0.0, 1.0);
            info_image(info, 600, 400, 400);
            info->background_mode = lit ? 0 : 1;
        } else if (s == "my_scene") {
            info->world = my_scene(rng).id;
            info->camera = Camera::new_(Point3{30.0, 0.0, 50.0}, Point3{0, 0, 0}, Vec3{0, 1, 0}, 55.0, 1.0, 0.0, 10.0, 0.0, 1.0); // main.rs:107-120
            info_image(info, 800, 800, 5000);
        } else if (s == "spheres_grid") {
            info->world = HittableList::new_(spheres_grid(rng, n ? n : 1000000)).id;
            const double side = std::sqrt((double)(n ? n : 1000000));
            info->camera = Camera::new_(Point3{0.6 * side, 0.25 * side, 0.6 * side}, Point3{0, 0, 0}, Vec3{0, 1, 0}, 40.0,
                                        3840.0 / 2160.0, 0.1, 0.88 * side, 0.0, 1.0);
            info_image(info, 3840, 2160, 256);
            info->background_mode = 1;
        } else if (s == "stress_volume") {
            info->world = stress_volume(rng).id;
            info->camera = Camera::new_(Point3{30.0, 8.0, 12.0}, Point3{0.0, 1.0, 0.0}, Vec3{0, 1, 0}, 40.0, 1920.0 / 1080.0, 0.0, 10.0, 0.0, 1.0);
            info_image(info, 1920, 1080, 1024);
        } else {
            return b->fail(RT_ERR_INVALID, "unknown scene '%s'", name);
        }
    } catch (const rt::SceneError& e) {
        b->error = e.what();
        return e.code;
    } catch (const std::exception& e) {
        b->error = e.what();
        return RT_ERR_OOM;
    }
    return RT_OK;
}
