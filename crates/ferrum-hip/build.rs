// Link against libferrum_hip.so.  FERRUM_HIP_LIB_DIR points at the directory that holds it
// (ferrum-infer-rs_amd/lib after `make -C ferrum-infer-rs_amd/csrc`, or the native-operator artifact directory the
// resolver unpacked, ferrum-native-ops/src/resolver.rs:136-318).
fn main() {
    let dir = std::env::var("FERRUM_HIP_LIB_DIR").unwrap_or_else(|_| "/opt/ferrum-hip/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=ferrum_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FERRUM_HIP_LIB_DIR");
}
