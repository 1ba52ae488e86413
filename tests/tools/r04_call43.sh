# the reference's profiling workload (test scene, 800 x 400) against the search / service switch threshold
for s in 4 8 16 24 32 48; do
  echo "== SOL_SWITCH=$s"
  SOL_SWITCH=$s python tests/tools/perf_quick.py test --spp 256 --phases 2>&1 | cut -c1-170
done
