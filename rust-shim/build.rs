// Replaces the reference's core/build.rs (WGSL `// #include` preprocessing, core/build.rs:1-17): there are no
// shaders to expand; the crate only has to find libkmeans_hip.so.
//
//   KMEANS_HIP_LIB_DIR  directory that holds libkmeans_hip.so
//                       (default: ../kmeans-gpu_amd/lib relative to this crate, i.e. `make -C kmeans-gpu_amd`)
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var_os("KMEANS_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|| {
        PathBuf::from(env::var_os("CARGO_MANIFEST_DIR").unwrap()).join("..").join("kmeans-gpu_amd").join("lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=kmeans_hip");
    // let binaries built against this crate find the library at run time without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=KMEANS_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
