#!/bin/bash
# Per-kernel resource usage (VGPRs, SGPRs, LDS, scratch) of a .hip source as compiled for the product library.
#   tools/kernel_meta.sh mm3dgs_slam_amd/csrc/fused.hip
set -e
src=$1; out=/tmp/kmeta_$(basename $src .hip).co
/opt/rocm/bin/hipcc --offload-arch=gfx950 --cuda-device-only -O3 -std=c++17 -munsafe-fp-atomics -fno-slp-vectorize -c $src -o $out
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$out --output=$out.elf --unbundle
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $out.elf | awk '
/^ +\.name: / {name=$2}
/\.group_segment_fixed_size:/ {lds=$2}
/\.private_segment_fixed_size:/ {scr=$2}
/\.sgpr_count:/ {sg=$2}
/\.vgpr_count:/ {vg=$2}
/\.agpr_count:/ {ag=$2}
/\.vgpr_spill_count:/ {sp=$2; printf "%-90s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %5d spill %d\n", substr(name,1,90), vg, ag, sg, lds, scr, sp}'
