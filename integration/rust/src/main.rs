//! `main()` of the reference (raytracer/src/main.rs:73-257) with its thread pool, channel and per-pixel loop
//! (:135-253) replaced by one compile and one call that renders on every GPU of the node (`rt_render_multi`: scene
//! copied to each device, image tiles dealt round-robin, buffers gathered on the first device). Scene (`my_scene`, the
//! reference's live one) and camera are built exactly as there.
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

/// The reference's live scene function, word for word (main.rs:690-835; the commented-out parts left out).
fn my_scene() -> HittableList {
    let mut objects = HittableList::new_default();

    let star1 = Arc::new(DiffuseLight::new_by_color1());
    objects.add(Arc::new(_Box::new(Point3::new(-4.0, -7.0, -7.0), Point3::new(10.0, 7.0, 7.0), star1.clone())));
    objects.add(Arc::new(_Box::new(Point3::new(-6.0, -9.0, -9.0), Point3::new(12.0, 9.0, 9.0), Arc::new(Dielectric::new(3.0)))));

    let star3 = Arc::new(DiffuseLight::new_by_color3());
    objects.add(Arc::new(Sphere::new(Point3::new(-14.0, 8.0, 10.0), 3.0, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(15.0, 7.0, -7.0), 2.0, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(18.0, 11.0, 18.0), 4.0, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(22.0, -13.0, 15.0), 1.5, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(-10.0, -20.0, 17.0), 1.5, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(17.0, -17.0, 17.0), 0.5, star3.clone())));
    objects.add(Arc::new(Sphere::new(Point3::new(-10.0, -9.0, -13.0), 2.0, star3.clone())));

    objects.add(Arc::new(XZRect::new(-100.0, 100.0, -100.0, 100.0, 10.0, Arc::new(Metal::new(Color::new(0.9, 0.9, 0.9), 0.0)))));
    objects.add(Arc::new(XYRect::new(-100.0, 100.0, -100.0, 100.0, -20.0, Arc::new(Metal::new(Color::new(0.9, 0.9, 0.9), 0.0)))));

    let star5 = Arc::new(DiffuseLight::new_by_color5());
    let star6 = Arc::new(DiffuseLight::new_by_color6());
    objects.add(Arc::new(Sphere::new(Point3::new(16.0, -10.0, 15.0), 5.0, star5.clone())));

    let mut fogs = HittableList::new_default();
    let ns = 10000;
    for _j in 0..ns {
        let fog_sphere = Arc::new(Sphere::new(
            Point3::new(16.0, -10.0, 15.0) + Vec3::random_in_unit_disk().unit() * 10.0 * random_double(0.8, 1.0),
            0.1,
            star6.clone(),
        ));
        fogs.add(fog_sphere);
        let fog_sphere = Arc::new(Sphere::new(
            Point3::new(16.0, -10.0, 15.0) + Vec3::random_in_unit_disk().unit() * 9.5 * random_double(0.7, 1.0),
            0.1,
            star5.clone(),
        ));
        fogs.add(fog_sphere);
        let fog_sphere = Arc::new(Sphere::new(
            Point3::new(16.0, -10.0, 15.0) + Vec3::random_in_unit_disk().unit() * 8.0 * random_double(0.8, 1.0),
            0.1,
            star6.clone(),
        ));
        fogs.add(fog_sphere);
    }
    objects.add(Arc::new(BvhNode::new_(&mut fogs, 0.0, 0.0)));
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
    let world = my_scene();
    let background = Vec3::new(0.0, 0.0, 0.0);
    let lookfrom = Point3::new(30.0, 0.0, 50.0);
    let lookat = Point3::new(0.0, 0.0, 0.0);
    let vup = Vec3::new(0.0, 1.0, 0.0);
    let dist_to_focus = 10.0;
    let aperture = 0.0;
    let cam = Camera::new((lookfrom, lookat), vup, 55.0, aspect_ratio, aperture, dist_to_focus, (0.0, 1.0));

    // render: what main.rs:135-253 did with a thread pool
    let (mut blob, mut nbytes) = (std::ptr::null_mut(), 0usize);
    scene::check(unsafe { ffi::rt_builder_compile(scene::builder(), world.handle(), &mut blob, &mut nbytes) });
    // one context per GPU of the node (the reference: n_workers threads, main.rs:128,137)
    let n_gpus = unsafe { ffi::rt_device_count() };
    assert!(n_gpus > 0, "no HIP device");
    let mut ctxs: Vec<*mut ffi::rt_context> = Vec::new();
    for device in 0..n_gpus {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { ffi::rt_init(&mut ctx, device) };
        assert!(rc == ffi::RT_OK, "rt_init({}) failed: {}", device, rc);
        ctxs.push(ctx);
    }
    println!("using {} GPUs", n_gpus);

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
    let mut stats = vec![ffi::rt_stats::default(); ctxs.len()];
    check_ctx(ctxs[0], unsafe {
        ffi::rt_render_multi(
            ctxs.as_ptr(),
            ctxs.len() as u32,
            blob,
            nbytes,
            &cam.0,
            &params,
            samples_per_pixel,
            sum.as_mut_ptr(),
            rgb8.as_mut_ptr(),
            stats.as_mut_ptr(),
        )
    });
    unsafe { ffi::rt_free(blob) };
    for (i, st) in stats.iter().enumerate() {
        println!("GPU {}: {} samples, {} ray segments, {:.3} s", i, st.camera_samples, st.ray_segments, st.seconds_render);
    }

    // rgb8 row y is the reference's image row y (v = y / (h - 1), main.rs:163,170): no flip
    let img = image::RgbImage::from_raw(image_width, image_height, rgb8).unwrap();
    img.save("output/test.jpg").unwrap();

    unsafe {
        for ctx in ctxs {
            ffi::rt_shutdown(ctx);
        }
        ffi::rt_builder_free(scene::builder());
    }
}
