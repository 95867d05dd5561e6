// Tells rustc where libomrdeskew.so lives.  OMRDESKEW_LIB_DIR defaults to the in-tree build output
// (omr-img-corrector_amd/lib, produced by `make -C omr-img-corrector_amd/csrc`).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("OMRDESKEW_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../omr-img-corrector_amd/lib")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=omrdeskew");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=OMRDESKEW_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/omrdeskew.h");
}
