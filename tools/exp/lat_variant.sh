#!/bin/bash
# lat_variant.sh <tag> [ENV=VAL ...] -- like build_variant.sh for the lane-cooperative kernel: regenerates csrc/cvm_asm_gen.h under the
# given switches (CVM_* in tools/cvm_kernel.py / tools/cvm.py, CVM_SYNTH=kind,count for the per-round cost programs) and builds
# build/variants/lib_<tag>.so; the committed header and library are left untouched.
set -eo pipefail
cd "$(dirname "$0")/../.."
tag=$1; shift
mkdir -p build/variants/$tag
cp plonky2-bn254-pairing_amd/csrc/*.h plonky2-bn254-pairing_amd/csrc/bn254_kernels.hip build/variants/$tag/
env "$@" python - <<PY
import sys
sys.path.insert(0, "tools")
import gen_kernels
text, stats = gen_kernels.render_cvm()
open("build/variants/$tag/cvm_asm_gen.h", "w").write(text)
print("$tag", {k: v for k, v in stats.items() if k != "by_kind"})
PY
sed -i 's#"../../include/bn254_pairing.h"#"'$PWD'/include/bn254_pairing.h"#' build/variants/$tag/bn254_kernels.hip
rm -f build/variants/lib_$tag.so
if ! hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -lz ${HIPCC_EXTRA:-} build/variants/$tag/bn254_kernels.hip -o build/variants/lib_$tag.so.tmp > build/variants/$tag.log 2>&1; then
    grep -E "error" build/variants/$tag.log | head -20 >&2 || true
    echo "build of variant $tag FAILED (log: build/variants/$tag.log)" >&2
    rm -f build/variants/lib_$tag.so.tmp
    exit 1
fi
mv build/variants/lib_$tag.so.tmp build/variants/lib_$tag.so
rm -rf build/variants/$tag build/variants/$tag.log
ls -la build/variants/lib_$tag.so
