#!/bin/bash
# build_from_header.sh <tag> <pairing_asm_gen.h> -- builds build/variants/lib_<tag>.so from an existing kernel header (e.g. one
# taken from an earlier commit: `git show <rev>:plonky2-bn254-pairing_amd/csrc/pairing_asm_gen.h > /tmp/h.h`) with the CURRENT host code.
set -e
cd "$(dirname "$0")/../.."
tag=$1; hdr=$2
mkdir -p build/variants/$tag
cp plonky2-bn254-pairing_amd/csrc/*.h plonky2-bn254-pairing_amd/csrc/bn254_kernels.hip build/variants/$tag/
cp "$hdr" build/variants/$tag/pairing_asm_gen.h
sed -i 's#"../../include/bn254_pairing.h"#"'$PWD'/include/bn254_pairing.h"#' build/variants/$tag/bn254_kernels.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared build/variants/$tag/bn254_kernels.hip -o build/variants/lib_$tag.so 2>&1 | grep -E "error" || true
rm -rf build/variants/$tag
ls -la build/variants/lib_$tag.so
