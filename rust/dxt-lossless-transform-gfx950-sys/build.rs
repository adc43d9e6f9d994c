// UNCOMPILED (see ../README.md).
// The shared library is built outside cargo by hipcc: `make -C dxt-lossless-transform_amd/csrc` (no Python needed; the package's
// own _build.py does the same: one object per source with `hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -c`, then
// `hipcc -shared`).  Point DXTLT_GFX950_LIB_DIR at the directory that holds libdxtlt_gfx950.so.
fn main() {
    println!("cargo:rerun-if-env-changed=DXTLT_GFX950_LIB_DIR");
    if let Ok(dir) = std::env::var("DXTLT_GFX950_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=dxtlt_gfx950");
}
