#!/bin/bash
# VGPR / AGPR / SGPR / scratch / LDS of every kernel in a HIP source (gfx950), from the code object's metadata.
# usage: tools/kernel_regs.sh owl-vit-object-detection_amd/csrc/attention_fwd.hip [extra hipcc flags]
set -e
src=$1; shift
tmp=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only --no-gpu-bundle-output -c "$src" -o "$tmp/k.co" "$@"
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$tmp/k.co" | awk '
  /\.name:/ && !/\.args/ {name=$2}
  /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {a=$2} /\.sgpr_count:/ {s=$2}
  /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {l=$2}
  /\.vgpr_spill_count:/ {sp=$2}
  /\.wavefront_size:/ {printf "%-110s vgpr %3s agpr %3s sgpr %3s scratch %5s spill %3s lds %6s\n", name, v, a, s, p, sp, l}'
rm -rf "$tmp"
