#!/bin/bash
# Builds variants of the chain forward kernel into tools/variants/libgcpnet_hip_cf_<tag>.so (git-ignored; they travel to the GPU box).
# Every argument is tag=flags, e.g.  tools/cf_variants.sh base= x1=-DGCP_CF_X=1 early=-DGCP_CB_STORE_EARLY
# (GCP_CF_X bits: measurement builds whose results are WRONG -- only the clock is read; see gcp2_chain_fwd.hip.)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/gcpnet_amd/csrc; V=$R/tools/variants; mkdir -p $V
OBJS=$(ls $C/*.o | grep -v gcp2_chain_fwd.o)
for A in "$@"; do
  TAG=${A%%=*}; FL=${A#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DGCP_CF_ONLY_SHIPPED $FL \
      -c $C/gcp2_chain_fwd.hip -o $V/cf_$TAG.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libgcpnet_hip_cf_$TAG.so $OBJS $V/cf_$TAG.o
  rm -f $V/cf_$TAG.o
  echo built cf_$TAG "($FL)"
done
