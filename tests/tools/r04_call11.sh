set -o pipefail
mkdir -p gpurun_out
{ echo "== base (one radius-16 tree, commit bef2398)"; SOLSTRALE_BUILD_DIR=$PWD/_var/base timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 c5 c3h test --spp 64
echo "== new (radius candidates 16 / 8 / 32 + probe)"; SOL_VERBOSE=1 timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 c5 c3h test --spp 64 2>&1 | grep -v "work order\|device tree (radius"
echo "== base"; SOLSTRALE_BUILD_DIR=$PWD/_var/base timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 c5 c3h test --spp 64
echo "== new"; timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 c5 c3h test --spp 64; } > gpurun_out/r04m_radius_candidates_ab.txt 2>&1
cat gpurun_out/r04m_radius_candidates_ab.txt
