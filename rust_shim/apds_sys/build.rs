// Links libapds_hip.so. Set APDS_LIB_DIR to the directory holding it (cubesat-apds_amd/ in this repository).
fn main() {
    let dir = std::env::var("APDS_LIB_DIR").expect("set APDS_LIB_DIR to the directory of libapds_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=apds_hip");
    println!("cargo:rerun-if-env-changed=APDS_LIB_DIR");
}
