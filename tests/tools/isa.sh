#!/bin/bash
# Device ISA of one HIP source under the product's flags (CPU container; hipcc cross-compiles): isa.sh <out.s> [source] [extra flags]
# Used to check that a source clean-up leaves the product kernels' instructions unchanged, and to read register counts.
set -e
here="$(cd "$(dirname "$0")/../../solstrale-rust_amd" && pwd)"
out="$1"; src="${2:-csrc/sol_render.hip}"; shift; shift || true
cd "$here"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Xclang -target-feature -Xclang -packed-fp32-ops \
  -Wall -Wno-unused-value --cuda-device-only -S "$src" -o "$out" "$@" 2>/dev/null
grep -v -E '^\s*(;|\.loc|\.file|\.ident|//)' "$out" | sed 's/;.*//' > "${out%.s}_clean.s"
grep -E "^\s*\.(vgpr_count|sgpr_spill_count|private_segment_fixed_size|vgpr_spill_count):|\.name:" "$out" | paste - - - - - | sed 's/\s\+/ /g'
