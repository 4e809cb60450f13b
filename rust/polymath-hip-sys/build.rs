// Where libpolymath_hip.so lives: POLYMATH_HIP_LIB_DIR (the `polymath_amd/` directory of this repository after
// `python __graft_entry__.py`).  `cargo check` never links, so the variable is only needed for build / test / run.
fn main() {
    println!("cargo:rerun-if-env-changed=POLYMATH_HIP_LIB_DIR");
    if let Ok(dir) = std::env::var("POLYMATH_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=polymath_hip");
}
