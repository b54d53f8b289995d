#!/bin/bash
# build_from_header.sh <tag> <pairing_asm_gen.h> -- builds build/variants/lib_<tag>.so from an existing kernel header (e.g. one
# taken from an earlier commit: `git show <rev>:plonky2-bn254-pairing_amd/csrc/pairing_asm_gen.h > /tmp/h.h`) with the CURRENT host code.
set -eo pipefail
cd "$(dirname "$0")/../.."
tag=$1; hdr=$2
mkdir -p build/variants/$tag
cp plonky2-bn254-pairing_amd/csrc/*.h plonky2-bn254-pairing_amd/csrc/bn254_kernels.hip build/variants/$tag/
cp "$hdr" build/variants/$tag/pairing_asm_gen.h
sed -i 's#"../../include/bn254_pairing.h"#"'$PWD'/include/bn254_pairing.h"#' build/variants/$tag/bn254_kernels.hip
rm -f build/variants/lib_$tag.so                  # never leave an older build behind: a failed compile must not be measured
if ! hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -lz build/variants/$tag/bn254_kernels.hip -o build/variants/lib_$tag.so.tmp > build/variants/$tag.log 2>&1; then
    grep -E "error" build/variants/$tag.log | head -20 >&2 || true
    echo "build of variant $tag FAILED (log: build/variants/$tag.log)" >&2
    rm -f build/variants/lib_$tag.so.tmp
    exit 1
fi
mv build/variants/lib_$tag.so.tmp build/variants/lib_$tag.so
rm -rf build/variants/$tag build/variants/$tag.log
ls -la build/variants/lib_$tag.so
