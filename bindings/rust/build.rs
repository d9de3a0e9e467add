// Link against this repository's library.  LDPC_TOOLBOX_HIP_LIB_DIR = the directory that holds
// libldpc_toolbox.so (ldpc_toolbox_amd/lib after `make -C ldpc_toolbox_amd/csrc`).
fn main() {
    let dir = std::env::var("LDPC_TOOLBOX_HIP_LIB_DIR")
        .unwrap_or_else(|_| format!("{}/../../ldpc_toolbox_amd/lib", env!("CARGO_MANIFEST_DIR")));
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=LDPC_TOOLBOX_HIP_LIB_DIR");
}
