// Points rustc at the two shared libraries built by `make -C gswt_renderer_amd/csrc` (GSWT_LIB_DIR overrides).
fn main() {
    let dir = std::env::var("GSWT_LIB_DIR").unwrap_or_else(|_| {
        format!("{}/../gswt_renderer_amd/lib", std::env::var("CARGO_MANIFEST_DIR").unwrap())
    });
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=gswt_hip");
    println!("cargo:rustc-link-lib=dylib=gswt_host");
    println!("cargo:rerun-if-env-changed=GSWT_LIB_DIR");
}
