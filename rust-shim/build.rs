fn main() {
    // point BN254_PAIRING_LIB_DIR at the directory holding libbn254_pairing_hip.so
    if let Ok(dir) = std::env::var("BN254_PAIRING_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=bn254_pairing_hip");
}
