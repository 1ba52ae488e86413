set -o pipefail
timeout -k 10 300 python -m pytest tests/test_environment.py -x -q -m gpu 2>&1 | tail -4
bash tests/tools/profile_round.sh r02 stats
bash tests/tools/profile_round.sh r02 pmc > gpurun_out/r02_pmc.log 2>&1; tail -3 gpurun_out/r02_pmc.log
for w in c1 c2 c4 c5; do timeout -k 10 300 python bench.py --workload $w --no-pmc --cpu-seconds 5 > gpurun_out/r02_${w}_bench.json 2> gpurun_out/r02_${w}_bench.err; echo "$w rc $?"; done
timeout -k 10 300 python bench.py --workload c5 --hdri --no-pmc --no-cpu-baseline > gpurun_out/r02_c5_hdri_bench.json 2> gpurun_out/r02_c5_hdri_bench.err; echo "c5 hdri rc $?"
for f in gpurun_out/r02_c*_bench.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], d['roofline']['frac'])"; done
