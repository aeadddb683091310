// ZKGPU_LIB_DIR must point at the directory holding libzkgpu.so (zkvm_amd/ in this repository).
fn main() {
    if let Ok(dir) = std::env::var("ZKGPU_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=zkgpu");
    println!("cargo:rerun-if-env-changed=ZKGPU_LIB_DIR");
}
