// Links librofl_zk.so (built by `python -m rofl_project_code_amd.build`; hipcc --offload-arch=gfx950).
fn main() {
    let dir = std::env::var("ROFL_ZK_LIB_DIR").expect("set ROFL_ZK_LIB_DIR to the directory holding librofl_zk.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rofl_zk");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=ROFL_ZK_LIB_DIR");
}
