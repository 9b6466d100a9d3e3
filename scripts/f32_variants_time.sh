#!/bin/bash
# On the GPU box: scripts/f32_check.py for every libcbo_hip_v*.so variant (timing-only builds of kernels_f32.hip).
cd "$GRAFT_REPO_ROOT"
for lib in cbo_with_oop_amd/libcbo_hip_v*.so; do
  echo "== $lib"
  CBO_HIP_ALLOW_DIAG=1 CBO_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 scripts/f32_check.py "$@" 2>&1 | grep -E "f32:|^n=" || echo failed
done
