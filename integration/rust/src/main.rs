//! `main()` of the reference (raytracer/src/main.rs:73-257) with its thread pool, channel and per-pixel loop
//! (:135-253) replaced by one compile, one upload and one render call. Scene and camera are built exactly as there.
//! UNVERIFIED: no Rust toolchain exists where this tree is developed.
mod ffi;
mod scene;

use scene::*;
use std::ffi::CStr;
use std::sync::Arc;

fn check_ctx(ctx: *mut ffi::rt_context, code: i32) {
    if code != ffi::RT_OK {
        let msg = unsafe { CStr::from_ptr(ffi::rt_last_error(ctx)) }.to_string_lossy().into_owned();
        panic!("rt error {}: {}", code, msg);
    }
}

/// A scene function in the reference's own words (cf. simple_light, main.rs:382-407).
fn simple_light() -> HittableList {
    let mut objects = HittableList::new_default();
    let pertext = Arc::new(NoiseTexture::new(4.0));
    objects.add(Arc::new(Sphere::new(Point3::new(0.0, -1000.0, 0.0), 1000.0, Arc::new(Lambertian::new_by_pointer(pertext.clone())))));
    objects.add(Arc::new(Sphere::new(Point3::new(0.0, 2.0, 0.0), 2.0, Arc::new(Lambertian::new_by_pointer(pertext)))));
    let difflight = Arc::new(DiffuseLight::new_by_color(Color::new(4.0, 4.0, 4.0)));
    objects.add(Arc::new(XYRect::new(3.0, 5.0, 1.0, 3.0, -2.0, difflight)));
    objects
}

fn main() {
    // image (main.rs:75-79)
    let aspect_ratio = 1.0;
    let image_width: u32 = 800;
    let image_height: u32 = (image_width as f64 / aspect_ratio) as u32;
    let samples_per_pixel: u32 = 5000;
    let max_depth: u32 = 50;

    // world + camera (main.rs:103-120), built through the same constructors
    scene::builder_init(0x5EED_0001, 0);
    let world = simple_light();
    let background = Vec3::new(0.0, 0.0, 0.0);
    let cam = Camera::new(
        (Point3::new(26.0, 3.0, 6.0), Point3::new(0.0, 2.0, 0.0)),
        Vec3::new(0.0, 1.0, 0.0),
        20.0,
        aspect_ratio,
        0.0,
        10.0,
        (0.0, 1.0),
    );

    // render: what main.rs:135-253 did with a thread pool
    let (mut blob, mut nbytes) = (std::ptr::null_mut(), 0usize);
    scene::check(unsafe { ffi::rt_builder_compile(scene::builder(), world.handle(), &mut blob, &mut nbytes) });
    let mut ctx = std::ptr::null_mut();
    let rc = unsafe { ffi::rt_init(&mut ctx, 0) };
    assert!(rc == ffi::RT_OK, "rt_init failed: {}", rc);
    let mut sc = std::ptr::null_mut();
    check_ctx(ctx, unsafe { ffi::rt_scene_upload(ctx, blob, nbytes, &mut sc) });
    unsafe { ffi::rt_free(blob) };

    let params = ffi::rt_params {
        width: image_width,
        height: image_height,
        max_depth,
        background_mode: 0,
        background: [background.x, background.y, background.z],
        seed: 0x5EED_0002,
        ..Default::default()
    };
    let n = (image_width * image_height * 3) as usize;
    let mut sum = vec![0f64; n];
    let mut rgb8 = vec![0u8; n];
    let mut stats = ffi::rt_stats::default();
    check_ctx(ctx, unsafe {
        ffi::rt_render(ctx, sc, &cam.0, &params, samples_per_pixel, sum.as_mut_ptr(), rgb8.as_mut_ptr(), &mut stats)
    });
    println!(
        "{} samples, {} ray segments, {:.3} s on the GPU",
        stats.camera_samples, stats.ray_segments, stats.seconds_render
    );

    // rgb8 row y is the reference's image row y (v = y / (h - 1), main.rs:163,170): no flip
    let img = image::RgbImage::from_raw(image_width, image_height, rgb8).unwrap();
    img.save("output/test.jpg").unwrap();

    unsafe {
        ffi::rt_scene_free(ctx, sc);
        ffi::rt_shutdown(ctx);
        ffi::rt_builder_free(scene::builder());
    }
}
