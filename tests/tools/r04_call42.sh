# C5 (and C3, C2) against the number of reinsertion rounds, twice each: do 12 rounds give what 16 do?
for rep in 1 2; do
for r in 8 12 16; do
  echo "== SOL_REINSERT=$r (run $rep)"
  SOL_REINSERT=$r python tests/tools/perf_quick.py c5 c3 c2 --spp 64 --phases 2>&1 | cut -c1-160
done
done
