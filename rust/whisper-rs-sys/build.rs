// Links the prebuilt engine instead of compiling whisper.cpp.
// WHISPER_MI355_LIB_DIR must point at the directory that holds libwhisper_mi355.so
// (handy_amd/ in this repository after `python -c 'import __graft_entry__ as g; g.build()'`).
fn main() {
    let dir = std::env::var("WHISPER_MI355_LIB_DIR").expect("set WHISPER_MI355_LIB_DIR to the directory containing libwhisper_mi355.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=whisper_mi355");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=WHISPER_MI355_LIB_DIR");
}
