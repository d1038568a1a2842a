#!/bin/bash
# usage: exp_build.sh <file-stem> "<extra hipcc flags>"   -- rebuild one kernel object with extra flags and relink the library (GPU box or here)
set -e
cd "$(dirname "$0")/../../norma_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-value -Wno-unused-result $2 -c $1.hip -o build/$1.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libnorma_hip.so build/k_gemm.o build/k_mel.o build/k_elem.o build/k_attn_enc.o build/k_decode.o build/nh_api.o build/norma_host.o
