// Links libkryst_hip.so (built by `make -C kryst_amd/csrc`: hipcc --offload-arch=gfx950 -ffp-contract=off).
// KRYST_HIP_LIB_DIR names the directory that holds it (default: ../../kryst_amd/lib relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("KRYST_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../kryst_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kryst_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=KRYST_HIP_LIB_DIR");
}
