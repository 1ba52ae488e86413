set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 bash tests/tools/counter_questions.sh r04 > /dev/null 2>&1; echo "counter questions rc $?"; cat gpurun_out/r04_counter_questions.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "full_size_c5" > gpurun_out/r04i_c5_full.log 2>&1; echo "c5 full-size test rc $?"; tail -5 gpurun_out/r04i_c5_full.log
timeout -k 10 300 python tests/tools/split_diff.py c3h 512 0 -1 3 > gpurun_out/r04i_split_diff_512.txt 2>&1; head -3 gpurun_out/r04i_split_diff_512.txt
