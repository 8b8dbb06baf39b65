#!/bin/bash
# Builds measurement variants of the chain backward kernel (GCP_CB_X bits, results WRONG -- only the clock is read) into
# tools/variants/libgcpnet_hip_cbx<bits>.so (git-ignored; they travel to the GPU box).  usage: tools/cb_variants.sh 0 1 2 4 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/gcpnet_amd/csrc; V=$R/tools/variants; mkdir -p $V
OBJS=$(ls $C/*.o | grep -v gcp2_chain_bwd.o)
for X in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -DGCP_CB_ONLY_SHIPPED -DGCP_CB_X=$X ${CB_EXTRA} \
      -c $C/gcp2_chain_bwd.hip -o $V/cbx$X.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/libgcpnet_hip_cbx$X.so $OBJS $V/cbx$X.o
  rm -f $V/cbx$X.o
  echo built cbx$X
done
