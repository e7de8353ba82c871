// Builds libfpgpu.so with hipcc for gfx950 (the Makefile under crypto_primitives_amd/ compiles one
// translation unit per limb count) and links it.  FPGPU_SRC points at that directory; ROCM_PATH
// at the ROCm installation (default /opt/rocm).
use std::{env, path::PathBuf, process::Command};

fn main() {
    let src = env::var("FPGPU_SRC").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../../crypto_primitives_amd")
    });
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    let jobs = env::var("NUM_JOBS").unwrap_or_else(|_| "8".into());
    let status = Command::new("make")
        .arg("-C")
        .arg(&src)
        .args(["-j", &jobs, "libfpgpu.so", "ARCH=gfx950"])
        .env("HIPCC", format!("{rocm}/bin/hipcc"))
        .status()
        .expect("make / hipcc not found");
    assert!(status.success(), "building libfpgpu.so failed");
    println!("cargo:rustc-link-search=native={}", src.display());
    println!("cargo:rustc-link-lib=dylib=fpgpu");
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", src.display());
    println!("cargo:rerun-if-changed={}", src.join("csrc").display());
    println!("cargo:rerun-if-changed={}", src.join("../include/fpgpu.h").display());
    println!("cargo:rerun-if-env-changed=FPGPU_SRC");
}
