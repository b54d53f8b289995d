#!/bin/bash
# build_variant.sh <tag> [ENV=VAL ...] -- regenerates the kernel header under the given generator switches and builds
# build/variants/lib_<tag>.so (the committed header / library are left untouched).
set -e
cd "$(dirname "$0")/../.."
tag=$1; shift
mkdir -p build/variants/$tag
cp plonky2-bn254-pairing_amd/csrc/*.h plonky2-bn254-pairing_amd/csrc/bn254_kernels.hip build/variants/$tag/
env "$@" python - <<PY
import sys
sys.path.insert(0, "tools")
import gen_kernels
text, stats = gen_kernels.render(verbose=False)
open("build/variants/$tag/pairing_asm_gen.h", "w").write(text)
print("$tag", stats)
PY
sed -i 's#"../../include/bn254_pairing.h"#"'$PWD'/include/bn254_pairing.h"#' build/variants/$tag/bn254_kernels.hip
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared build/variants/$tag/bn254_kernels.hip -o build/variants/lib_$tag.so 2>&1 | grep -E "error" || true
rm -rf build/variants/$tag
ls -la build/variants/lib_$tag.so
