//! The reference's scene-building API with the same type names and constructor signatures
//! (raytracer/src/{sphere,moving_sphere,arrect,_box,hittable,hittable_list,bvh,constant_medium,material,texture,
//! camera}.rs), each object registering itself with one process-wide `rt_builder` and keeping the integer handle it
//! got. Scene functions written against the reference's types (main.rs:260-835) compile against these unchanged,
//! apart from `Arc<dyn Hittable>` arguments, which only need `handle()` here.
//! UNVERIFIED: no Rust toolchain exists where this tree is developed.
#![allow(dead_code)]
use crate::ffi;
use std::ffi::CStr;
use std::os::raw::c_int;
use std::sync::atomic::{AtomicPtr, Ordering};
use std::sync::Arc;

#[derive(Clone, Copy, Debug, PartialEq)]
pub struct Vec3 {
    pub x: f64,
    pub y: f64,
    pub z: f64,
}
pub use Vec3 as Color;
pub use Vec3 as Point3;
impl Vec3 {
    pub fn new(x: f64, y: f64, z: f64) -> Self {
        Self { x, y, z }
    }
    fn arr(&self) -> [f64; 3] {
        [self.x, self.y, self.z]
    }
    /// vec3.rs:30-36, 50-55
    pub fn length(&self) -> f64 {
        (self.x * self.x + self.y * self.y + self.z * self.z).sqrt()
    }
    pub fn unit(&self) -> Self {
        let l = self.length();
        assert!(l != 0.0, "div 0!");
        Self::new(self.x / l, self.y / l, self.z / l)
    }
    /// vec3.rs:102-110: a point of the unit disk in the XZ plane (as the reference has it)
    pub fn random_in_unit_disk() -> Self {
        loop {
            let p = Self::new(random_double(-1.0, 1.0), 0.0, random_double(-1.0, 1.0));
            if p.x * p.x + p.z * p.z < 1.0 {
                return p;
            }
        }
    }
}
impl std::ops::Add for Vec3 {
    type Output = Vec3;
    fn add(self, o: Vec3) -> Vec3 {
        Vec3::new(self.x + o.x, self.y + o.y, self.z + o.z)
    }
}
impl std::ops::Mul<f64> for Vec3 {
    type Output = Vec3;
    fn mul(self, s: f64) -> Vec3 {
        Vec3::new(self.x * s, self.y * s, self.z * s)
    }
}

/// rtweekend.rs:19-22. Scene functions draw their own random numbers on the host, as in the reference (`rand` is its
/// dependency, not this crate's: a 64-bit LCG seeded once stands in, so that a build is repeatable).
pub fn random_double(low: f64, high: f64) -> f64 {
    use std::sync::atomic::AtomicU64;
    static STATE: AtomicU64 = AtomicU64::new(0x5EED_0001_2545_F491);
    let mut s = STATE.load(Ordering::Relaxed);
    s = s.wrapping_mul(6364136223846793005).wrapping_add(1442695040888963407);
    STATE.store(s, Ordering::Relaxed);
    low + (high - low) * ((s >> 11) as f64 * (1.0 / 9007199254740992.0))
}

static BUILDER: AtomicPtr<ffi::rt_builder> = AtomicPtr::new(std::ptr::null_mut());

/// The process-wide builder. `seed` stands in for the `thread_rng()` the reference uses while it builds a scene
/// (BVH split axes, bvh.rs:22; Perlin tables, perlin.rs:28-54).
pub fn builder_init(seed: u64, flags: u32) {
    let b = unsafe { ffi::rt_builder_new(seed) };
    assert!(!b.is_null(), "rt_builder_new failed");
    if flags != 0 {
        check(unsafe { ffi::rt_builder_set_flags(b, flags) });
    }
    BUILDER.store(b, Ordering::SeqCst);
}
pub fn builder() -> *mut ffi::rt_builder {
    let b = BUILDER.load(Ordering::SeqCst);
    assert!(!b.is_null(), "scene::builder_init has not been called");
    b
}
/// Every place the reference panics comes back as a negative code; keep the behaviour.
pub fn check(code: c_int) -> c_int {
    if code < 0 {
        let b = BUILDER.load(Ordering::SeqCst);
        let msg = if b.is_null() {
            String::from("?")
        } else {
            unsafe { CStr::from_ptr(ffi::rt_builder_last_error(b)) }.to_string_lossy().into_owned()
        };
        panic!("rt error {}: {}", code, msg);
    }
    code
}

pub trait Texture: Send + Sync {
    fn handle(&self) -> c_int;
}
pub trait Material: Send + Sync {
    fn handle(&self) -> c_int;
}
pub trait Hittable: Send + Sync {
    fn handle(&self) -> c_int;
}

// ---- textures (texture.rs) ----
pub struct SolidColor(c_int);
impl SolidColor {
    pub fn new(c: Color) -> Self {
        SolidColor(check(unsafe { ffi::rt_tex_solid(builder(), c.x, c.y, c.z) }))
    }
}
pub struct CheckerTexture(c_int);
impl CheckerTexture {
    pub fn new_by_color(c1: Color, c2: Color) -> Self {
        let (even, odd) = (SolidColor::new(c1), SolidColor::new(c2));
        CheckerTexture(check(unsafe { ffi::rt_tex_checker(builder(), even.0, odd.0) }))
    }
    pub fn new(even: Arc<dyn Texture>, odd: Arc<dyn Texture>) -> Self {
        CheckerTexture(check(unsafe { ffi::rt_tex_checker(builder(), even.handle(), odd.handle()) }))
    }
}
pub struct NoiseTexture(c_int);
impl NoiseTexture {
    pub fn new(sc: f64) -> Self {
        NoiseTexture(check(unsafe { ffi::rt_tex_noise(builder(), sc) }))
    }
}
pub struct ImageTexture(c_int);
impl ImageTexture {
    /// Decodes with the `image` crate exactly as texture.rs:90-103 does and hands over the RGB8 bytes.
    pub fn new(filename: &str) -> Self {
        let img = image::open(filename).unwrap();
        let (w, h) = (img.width(), img.height());
        let buf = img.into_rgb8().into_raw();
        ImageTexture(check(unsafe { ffi::rt_tex_image(builder(), buf.as_ptr(), w, h) }))
    }
}
macro_rules! impl_handle {
    ($tr:ident: $($t:ident),*) => { $(impl $tr for $t { fn handle(&self) -> c_int { self.0 } })* };
}
impl_handle!(Texture: SolidColor, CheckerTexture, NoiseTexture, ImageTexture);

// ---- materials (material.rs) ----
pub struct Lambertian(c_int);
impl Lambertian {
    pub fn new(a: Color) -> Self {
        Self::new_by_pointer(Arc::new(SolidColor::new(a)))
    }
    pub fn new_by_pointer(a: Arc<dyn Texture>) -> Self {
        Lambertian(check(unsafe { ffi::rt_mat_lambertian(builder(), a.handle()) }))
    }
}
pub struct Metal(c_int);
impl Metal {
    pub fn new(a: Color, f: f64) -> Self {
        Metal(check(unsafe { ffi::rt_mat_metal(builder(), a.x, a.y, a.z, f) }))
    }
}
pub struct Dielectric(c_int);
impl Dielectric {
    pub fn new(i: f64) -> Self {
        Dielectric(check(unsafe { ffi::rt_mat_dielectric(builder(), i) }))
    }
}
pub struct DiffuseLight(c_int);
impl DiffuseLight {
    fn with(emit: &dyn Texture, k: f64) -> Self {
        DiffuseLight(check(unsafe { ffi::rt_mat_diffuse_light(builder(), emit.handle(), k) }))
    }
    pub fn new_by_color(c: Color) -> Self {
        Self::with(&SolidColor::new(c), 1.0)
    }
    pub fn new_by_color1() -> Self {
        Self::with(&ImageTexture::new("raytracer/star1.jpg"), 1.5)
    }
    pub fn new_by_color3() -> Self {
        Self::with(&ImageTexture::new("raytracer/star3.jpeg"), 1.5)
    }
    pub fn new_by_color5() -> Self {
        Self::with(&ImageTexture::new("raytracer/star5.jpeg"), 0.5)
    }
    pub fn new_by_color6() -> Self {
        Self::with(&ImageTexture::new("raytracer/star6.jpeg"), 0.5)
    }
}
pub struct Isotropic(c_int);
impl Isotropic {
    pub fn new_by_color(c: Color) -> Self {
        Isotropic(check(unsafe { ffi::rt_mat_isotropic(builder(), SolidColor::new(c).0) }))
    }
}
impl_handle!(Material: Lambertian, Metal, Dielectric, DiffuseLight, Isotropic);

// ---- hittables ----
pub struct Sphere(c_int);
impl Sphere {
    pub fn new(cen: Point3, r: f64, m: Arc<dyn Material>) -> Self {
        Sphere(check(unsafe { ffi::rt_sphere(builder(), cen.arr().as_ptr(), r, m.handle()) }))
    }
}
pub struct MovingSphere(c_int);
impl MovingSphere {
    pub fn new(cen0: Point3, cen1: Point3, time0: f64, time1: f64, r: f64, m: Arc<dyn Material>) -> Self {
        MovingSphere(check(unsafe {
            ffi::rt_moving_sphere(builder(), cen0.arr().as_ptr(), cen1.arr().as_ptr(), time0, time1, r, m.handle())
        }))
    }
}
pub struct XYRect(c_int);
impl XYRect {
    pub fn new(x0: f64, x1: f64, y0: f64, y1: f64, k: f64, mat: Arc<dyn Material>) -> Self {
        XYRect(check(unsafe { ffi::rt_xy_rect(builder(), x0, x1, y0, y1, k, mat.handle()) }))
    }
}
pub struct XZRect(c_int);
impl XZRect {
    pub fn new(x0: f64, x1: f64, z0: f64, z1: f64, k: f64, mat: Arc<dyn Material>) -> Self {
        XZRect(check(unsafe { ffi::rt_xz_rect(builder(), x0, x1, z0, z1, k, mat.handle()) }))
    }
}
pub struct YZRect(c_int);
impl YZRect {
    pub fn new(y0: f64, y1: f64, z0: f64, z1: f64, k: f64, mat: Arc<dyn Material>) -> Self {
        YZRect(check(unsafe { ffi::rt_yz_rect(builder(), y0, y1, z0, z1, k, mat.handle()) }))
    }
}
pub struct _Box(c_int);
impl _Box {
    pub fn new(p0: Point3, p1: Point3, ptr: Arc<dyn Material>) -> Self {
        _Box(check(unsafe { ffi::rt_box(builder(), p0.arr().as_ptr(), p1.arr().as_ptr(), ptr.handle()) }))
    }
}
pub struct Translate(c_int);
impl Translate {
    pub fn new(p: Arc<dyn Hittable>, displacement: Vec3) -> Self {
        Translate(check(unsafe { ffi::rt_translate(builder(), p.handle(), displacement.arr().as_ptr()) }))
    }
}
pub struct RotateY(c_int);
impl RotateY {
    pub fn new(p: Arc<dyn Hittable>, angle: f64) -> Self {
        RotateY(check(unsafe { ffi::rt_rotate_y(builder(), p.handle(), angle) }))
    }
}
pub struct ConstantMedium(c_int);
impl ConstantMedium {
    pub fn new_by_color(b: Arc<dyn Hittable>, d: f64, c: Color) -> Self {
        ConstantMedium(check(unsafe { ffi::rt_constant_medium(builder(), b.handle(), d, c.x, c.y, c.z) }))
    }
}
pub struct HittableList(c_int);
impl HittableList {
    pub fn new_default() -> Self {
        HittableList(check(unsafe { ffi::rt_list_new(builder()) }))
    }
    pub fn new(object: Arc<dyn Hittable>) -> Self {
        let mut l = Self::new_default();
        l.add(object);
        l
    }
    pub fn add(&mut self, object: Arc<dyn Hittable>) {
        check(unsafe { ffi::rt_list_add(builder(), self.0, object.handle()) });
    }
}
pub struct BvhNode(c_int);
impl BvhNode {
    /// bvh.rs:68-71. Sorts `list` in place like the reference (unless the tree is left to the device, builder flag 8).
    pub fn new_(list: &mut HittableList, time0: f64, time1: f64) -> Self {
        BvhNode(check(unsafe { ffi::rt_bvh(builder(), list.0, time0, time1) }))
    }
}
impl_handle!(Hittable: Sphere, MovingSphere, XYRect, XZRect, YZRect, _Box, Translate, RotateY, ConstantMedium, HittableList, BvhNode);

// ---- camera (camera.rs:20-54): only the arguments travel; the library derives the rest the same way ----
#[derive(Clone, Copy)]
pub struct Camera(pub ffi::rt_camera);
impl Camera {
    pub fn new(
        (lookfrom, lookat): (Point3, Point3),
        vup: Vec3,
        vfov: f64,
        aspect_ratio: f64,
        aperture: f64,
        focus_dist: f64,
        (time0, time1): (f64, f64),
    ) -> Camera {
        Camera(ffi::rt_camera {
            from: lookfrom.arr(),
            at: lookat.arr(),
            vup: vup.arr(),
            vfov,
            aspect: aspect_ratio,
            aperture,
            focus: focus_dist,
            time0,
            time1,
        })
    }
}
