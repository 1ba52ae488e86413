set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python tests/tools/split_sweep.py c3 c3h c3hi c5 c2 --budgets -1 --slacks 3 --reinsert 8,16,32 > gpurun_out/r04e_reinsert_rounds.txt 2>&1; echo "sweep rc $?"; cat gpurun_out/r04e_reinsert_rounds.txt
timeout -k 10 300 python tests/tools/split_sweep.py c3h c3hi --budgets 0 --slacks 3 --reinsert 0,8 > gpurun_out/r04e_hetero_unsplit.txt 2>&1; cat gpurun_out/r04e_hetero_unsplit.txt
