// Links libsprsolve_hip.so (built with `make -C sprsolve_amd/csrc`).
// SPRSOLVE_HIP_LIB_DIR points at the directory that holds it.
fn main() {
    if let Ok(dir) = std::env::var("SPRSOLVE_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=sprsolve_hip");
    println!("cargo:rerun-if-env-changed=SPRSOLVE_HIP_LIB_DIR");
}
