// Link against the in-tree libvrt.so (voxelraytracing_amd/libvrt.so); VRT_LIB_DIR overrides the directory.
fn main() {
    let dir = std::env::var("VRT_LIB_DIR").unwrap_or_else(|_| {
        let here = std::path::PathBuf::from(std::env::var("CARGO_MANIFEST_DIR").unwrap());
        here.join("../../../voxelraytracing_amd").to_string_lossy().into_owned()
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=vrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=VRT_LIB_DIR");
}
