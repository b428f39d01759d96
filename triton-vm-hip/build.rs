// Link against libtriton_hip.so (the C ABI) and libtriton_host.so (the C++ mirror of Prover::prove above it, stage 2).
// TRITON_HIP_LIB_DIR = the directory that holds both (…/triton_vm_amd after `python -m triton_vm_amd.build`, which runs
// hipcc --offload-arch=gfx950 over csrc/*.hip and g++ over host/triton_host.cpp).
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=TRITON_HIP_LIB_DIR");
    let dir = env::var("TRITON_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("triton_vm_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=triton_hip");
    println!("cargo:rustc-link-lib=dylib=triton_host");
    // the RCCL communicator of the multi-GPU prover (host/rccl_comm.cpp -> libtriton_rccl.so, which pulls in librccl): only with
    // the `rccl` feature, so that a single-GPU box without RCCL builds and links the product
    if env::var_os("CARGO_FEATURE_RCCL").is_some() {
        println!("cargo:rustc-link-lib=dylib=triton_rccl");
    }
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
}
