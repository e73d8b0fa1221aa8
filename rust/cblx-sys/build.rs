// cblx-sys/build.rs — links libcblx.so (built by `python -c "import __graft_entry__ as g; g.build()"` -> cbl_amd/libcblx.so).
// CBLX_DIR names the directory that holds it; default: ../../cbl_amd relative to this crate.
// NOT compiled in the image this repository was built in (no Rust toolchain there): shipped as source.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("CBLX_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../cbl_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=cblx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=CBLX_DIR");
    println!("cargo:rerun-if-changed=../../include/cblx.h");
}
