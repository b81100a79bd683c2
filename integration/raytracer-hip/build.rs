// NOT COMPILED: the image this file was written in has no Rust toolchain.
//
// librt_hip.so is built outside cargo, by `make -C rust-raytracer_amd/csrc`
// (hipcc --offload-arch=gfx950); RT_HIP_LIB_DIR names the directory it lies in.
fn main() {
    println!("cargo:rerun-if-env-changed=RT_HIP_LIB_DIR");
    let dir = std::env::var("RT_HIP_LIB_DIR").expect("set RT_HIP_LIB_DIR to the directory that holds librt_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rt_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
