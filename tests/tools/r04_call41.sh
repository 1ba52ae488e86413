# C5 (and C2) on the wavefront variants of the A/B library: where path lengths diverge most, does a compacted search pay?
export SOLSTRALE_BUILD_DIR=$PWD/solstrale-rust_amd/_build_ab
for k in v1 v2 v3; do
  echo "== SOL_KERNEL=$k"
  SOL_KERNEL=$k python tests/tools/perf_quick.py c5 c2 --spp 64 2>&1 | cut -c1-200
done
