// Links the prebuilt librays_hip.so (built with `make -C raytracing_amd/csrc`, hipcc, gfx950).
// RAYS_HIP_LIB_DIR points at the directory that holds it.
fn main() {
    let dir = std::env::var("RAYS_HIP_LIB_DIR").expect("set RAYS_HIP_LIB_DIR to the directory of librays_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rays_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RAYS_HIP_LIB_DIR");
}
