// Links libmicroflow_amd.so (built by `python microflow_rs_amd/build.py`).
fn main() {
    let dir = std::env::var("MICROFLOW_AMD_LIB_DIR")
        .unwrap_or_else(|_| "../../microflow_rs_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=microflow_amd");
    println!("cargo:rerun-if-env-changed=MICROFLOW_AMD_LIB_DIR");
}
