#!/bin/bash
# build_pad.sh <pad bytes> -- the committed kernel header with another code offset behind the 64 KB boundary (BN254_KERNEL_PAD)
set -e
cd "$(dirname "$0")/../.."
pad=$1
mkdir -p build/variants
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -lz -DBN254_KERNEL_PAD=$pad plonky2-bn254-pairing_amd/csrc/bn254_kernels.hip -o build/variants/lib_pad$pad.so
ls build/variants/lib_pad$pad.so
