// build.rs of the reference crate when it links the MI355X hot path.
// PSEUDOALIGNER_AMD_LIB_DIR = directory that holds libpseudoaligner_amd.so (rust-pseudoaligner_amd/ of this repository).
fn main() {
    let dir = std::env::var("PSEUDOALIGNER_AMD_LIB_DIR").expect("set PSEUDOALIGNER_AMD_LIB_DIR to the directory of libpseudoaligner_amd.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=pseudoaligner_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=PSEUDOALIGNER_AMD_LIB_DIR");
}
