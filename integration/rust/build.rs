// Links librt_hip.so (built in-tree by `python -m raytracer_amd.build product`). RT_HIP_LIB_DIR overrides the
// directory; by default it is ../../raytracer_amd relative to this crate.
fn main() {
    let dir = std::env::var("RT_HIP_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{}/../../raytracer_amd", here)
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
}
