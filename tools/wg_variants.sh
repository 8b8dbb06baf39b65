#!/bin/bash
# Builds variants of the workgroup backward kernel into tools/variants/libgcpnet_hip_wgb_<tag>.so (git-ignored; they travel to the GPU box).
# Every argument is tag=flags, e.g.  tools/wgb_variants.sh base= x1=-DGCP_WG_X=1 early=-DGCP_CB_STORE_EARLY
# (GCP_WG_X bits: measurement builds whose results are WRONG -- only the clock is read; see gcp_wg_bwd.hip.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/gcpnet_amd/csrc; V=$R/tools/variants; mkdir -p $V
OBJS=$(ls $C/*.o | grep -v gcp_wg_bwd.o)
for A in "$@"; do
  TAG=${A%%=*}; FL=${A#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DGCP_WG_ONLY_SHIPPED $FL \
      -c $C/gcp_wg_bwd.hip -o $V/wgb_$TAG.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libgcpnet_hip_wgb_$TAG.so $OBJS $V/wgb_$TAG.o
  rm -f $V/wgb_$TAG.o
  echo built wgb_$TAG "($FL)"
done
