set -o pipefail
mkdir -p gpurun_out
for r in 8 16 32 64; do echo "== SOL_PLOC_R=$r"; SOL_PLOC_R=$r timeout -k 10 300 python tests/tools/split_sweep.py c3 c5 c2 c3h --budgets -1 --slacks 3 --reinsert 8; done > gpurun_out/r04k_ploc_radius.txt 2>&1
cat gpurun_out/r04k_ploc_radius.txt
