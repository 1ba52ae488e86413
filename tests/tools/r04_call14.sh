set -o pipefail
mkdir -p gpurun_out
for nc in 1.5 2.0 2.5 3.5 5.0; do echo "== SOL_NODE_COST=$nc (SOL_PLOC_R=8)"; SOL_NODE_COST=$nc SOL_PLOC_R=8 timeout -k 10 300 python tests/tools/split_sweep.py c3 c5 c2 c3h --budgets -1 --slacks 3 --reinsert 8; done > gpurun_out/r04q_node_cost.txt 2>&1
cat gpurun_out/r04q_node_cost.txt
