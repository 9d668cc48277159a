// C++ host API with the reference's names and constructor arguments, over the C builder in rt.h.
//
// The reference builds its scene as Arc<dyn Hittable/Material/Texture> values (raytracer/src/main.rs:260-835).
// Here each constructor returns a small handle value instead of an Arc; which builder receives the
// calls is set by an rt::SceneScope on the current thread, so signatures stay those of the reference:
//
//     rt::SceneScope scope(builder);
//     auto red   = rt::Lambertian::new_(rt::Color{0.65, 0.05, 0.05});
//     auto wall  = rt::YZRect::new_(0, 555, 0, 555, 555, red);
//     rt::HittableList world = rt::HittableList::new_default();  world.add(wall);
//
// Where the reference would panic (missing image file, no bounding box), rt::SceneError is thrown.
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/rt.h"

namespace rt {

struct SceneError : std::runtime_error {
    int code;
    SceneError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};

struct Vec3 { // vec3.rs:5-9
    double x, y, z;
    static Vec3 new_(double x, double y, double z) { return Vec3{x, y, z}; }
    static Vec3 zero() { return Vec3{0, 0, 0}; }
    static Vec3 ones() { return Vec3{1, 1, 1}; }
    double squared_length() const { return x * x + y * y + z * z; }
    double length() const;
    Vec3 unit() const;
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline Vec3 operator*(Vec3 a, double s) { return Vec3{a.x * s, a.y * s, a.z * s}; }
inline Vec3 operator/(Vec3 a, double s) { return Vec3{a.x / s, a.y / s, a.z / s}; }
using Point3 = Vec3;
using Color = Vec3;

struct Texture { int id; };
struct Material { int id; };
struct Hittable { int id; };

class SceneScope {
public:
    explicit SceneScope(rt_builder* b);
    ~SceneScope();
    static rt_builder* current();
private:
    rt_builder* prev_;
};
int check(int handle_or_code); // throws SceneError on a negative value

struct SolidColor { static Texture new_(Color c); };                                   // texture.rs:18
struct CheckerTexture {                                                                // /root/reference/texture.rs:40-58
    static Texture new_(Texture even, Texture odd);
    static Texture new_by_color(Color c1, Color c2);
};
struct NoiseTexture { static Texture new_(double sc); };                               // texture.rs:63
struct ImageTexture { static Texture new_(const char* filename); };                    // texture.rs:90

struct Lambertian {                                                                    // material.rs:30-38
    static Material new_(Color a);
    static Material new_by_pointer(Texture a);
};
struct Metal { static Material new_(Color a, double f); };                             // material.rs:68
struct Dielectric { static Material new_(double i); };                                 // material.rs:105
struct DiffuseLight {                                                                  // material.rs:158-189
    static Material new_by_color(Color c);
    static Material new_by_texture(Texture emit, double k);
};

struct Sphere { static Hittable new_(Point3 cen, double r, Material m); };             // sphere.rs:16
struct MovingSphere {                                                                  // moving_sphere.rs:19
    static Hittable new_(Point3 cen0, Point3 cen1, double time0, double time1, double r, Material m);
};
struct XYRect { static Hittable new_(double x0, double x1, double y0, double y1, double k, Material mat); }; // arrect.rs:18
struct XZRect { static Hittable new_(double x0, double x1, double z0, double z1, double k, Material mat); }; // arrect.rs:70
struct YZRect { static Hittable new_(double y0, double y1, double z0, double z1, double k, Material mat); }; // arrect.rs:122
struct _Box { static Hittable new_(Point3 p0, Point3 p1, Material ptr); };             // _box.rs:15
struct Translate { static Hittable new_(Hittable p, Vec3 displacement); };             // hittable.rs:61
struct RotateY { static Hittable new_(Hittable p, double angle); };                    // hittable.rs:104
struct ConstantMedium { static Hittable new_by_color(Hittable b, double d, Color c); }; // constant_medium.rs:15

struct HittableList {                                                                  // hittable_list.rs:8-37
    int id;
    static HittableList new_default();
    static HittableList new_(Hittable object);
    void add(Hittable object);
    operator Hittable() const { return Hittable{id}; }
};
struct BvhNode { static Hittable new_(HittableList& list, double time0, double time1); }; // bvh.rs:68

// Camera::new((lookfrom, lookat), vup, vfov, aspect_ratio, aperture, focus_dist, (time0, time1)), camera.rs:21-29
struct Camera {
    static rt_camera new_(Point3 lookfrom, Point3 lookat, Vec3 vup, double vfov, double aspect_ratio, double aperture,
                          double focus_dist, double time0, double time1);
};

} // namespace rt
