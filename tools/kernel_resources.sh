#!/bin/bash
# Registers, scratch and LDS of every kernel, as the compiler reports them (no GPU needed):
#   tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
# Each translation unit with the flags the Makefile builds it with.
cd "$(dirname "$0")/../objective-slam_amd/csrc"
for f in oslam_kernels oslam_vote_wide oslam_sort oslam_posegpu oslam_voxel oslam_depth; do
  fl=""; [ $f = oslam_vote_wide ] && fl="-mllvm -disable-machine-licm"
  echo "== $f.hip $fl"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -ffp-contract=off -fno-fast-math -I../../include -I. $fl \
      -Rpass-analysis=kernel-resource-usage -c $f.hip -o /tmp/kr_$$.o 2>&1 \
    | grep -E "Function Name|VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs Spill|VGPRs Spill|LDS Size" | sed 's/.*remark: *//; s/ \[-Rpass.*//'
done
rm -f /tmp/kr_$$.o
