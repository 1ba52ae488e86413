for b in "" sah8 sah16 sah64 host; do
  echo "== SOL_BVH=$b"
  SOL_BVH=$b python tests/tools/perf_quick.py c3 c5 c2 --spp 64 --phases 2>&1 | cut -c1-200
done
