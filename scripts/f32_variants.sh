#!/bin/bash
# Timing-only variants of the fp32 strip kernel (F32_DBG bit mask, kernels_f32.hip): builds one shared library per
# mask HERE (cross-compile), to be timed on the GPU box with scripts/f32_variants_time.sh.
set -e
cd "$(dirname "$0")/../cbo_with_oop_amd/csrc"
mkdir -p ../../gpurun_out
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I."
for m in "$@"; do
  /opt/rocm/bin/hipcc $FL -DF32_DBG=$m -c kernels_f32.hip -o /tmp/kernels_f32_v$m.o 2>/dev/null
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcbo_hip_v$m.so cbo_api.o kernels_kmat.o kernels_trsm.o kernels_chol.o kernels_acq.o kernels_sem.o /tmp/kernels_f32_v$m.o
  echo built v$m
done
