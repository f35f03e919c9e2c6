#!/bin/bash
# build libhriemo.so with extra attention.hip flags ("$@"), for A/B experiments on the GPU box
cd /root/repo/hri-emo_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c attention.hip -o build/attention.o 2>&1 | grep -v "not a recognized feature" 
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libhriemo.so build/gemm.o build/gemm_mx8.o build/attention.o build/rowops.o build/runtime.o
