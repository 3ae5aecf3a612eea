//! Links `libsp1hip.so`. The library is built by this repository's `python -c "import __graft_entry__ as g; g.build()"`
//! (hipcc --offload-arch=gfx950) into `sp1_amd/lib/`; point `SP1HIP_LIB_DIR` elsewhere for an installed copy.
//! Mirrors what sp1-gpu-sys' build script does for its CUDA archive (/root/reference/sp1-gpu/crates/sys/build.rs), minus
//! the compilation: this crate never invokes hipcc.
use std::{env, path::PathBuf};

fn main() {
    let dir = env::var("SP1HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../sp1_amd/lib")
    });
    let dir = dir.canonicalize().unwrap_or(dir);
    assert!(
        dir.join("libsp1hip.so").exists(),
        "libsp1hip.so not found in {} (build it with __graft_entry__.build(), or set SP1HIP_LIB_DIR)",
        dir.display()
    );
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=sp1hip");
    // the loader must find it at run time too
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=SP1HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/sp1hip.h");
}
