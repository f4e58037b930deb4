// Point the linker at the directory that holds librtiow_gpu.so (built by `make -C rtiow-rust_amd/csrc`).
fn main() {
    let dir = std::env::var("RTIOW_GPU_LIB_DIR").unwrap_or_else(|_| "../../../csrc".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtiow_gpu");
    println!("cargo:rerun-if-env-changed=RTIOW_GPU_LIB_DIR");
}
