#!/bin/bash
# Lab: link libavt_<tag>.so = the lab library with gemm_persist.hip taken from a git revision (or the working tree: rev = WT), for same-box A/B
# of kernel versions.  usage: tools/lab/build_persist_variant.sh <tag> <rev|WT> [extra -D flags]
set -e
TAG=$1; REV=$2; shift 2
cd "$(dirname "$0")/../../avt_amd/csrc"
make -j8 lab > /dev/null
mkdir -p build/pv
if [ "$REV" = WT ]; then cp gemm_persist.hip build/pv/gemm_persist_$TAG.hip; else git show $REV:avt_amd/csrc/gemm_persist.hip > build/pv/gemm_persist_$TAG.hip; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -I../../include -I. -DAVT_LAB -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" -c build/pv/gemm_persist_$TAG.hip -o build/pv/gemm_persist_$TAG.o
OBJS="gemm.lab.o layernorm.lab.o vit_attention.lab.o cls_attention.lab.o head_attention.lab.o elementwise.lab.o preproc.lab.o xent.lab.o optim.lab.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS api.o build/pv/gemm_persist_$TAG.o -o ../libavt_$TAG.so
ls -la ../libavt_$TAG.so
