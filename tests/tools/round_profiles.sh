set -o pipefail
bash tests/tools/profile_round.sh r02 stats
bash tests/tools/profile_round.sh r02 pmc > gpurun_out/r02_pmc.log 2>&1; tail -2 gpurun_out/r02_pmc.log
for w in c1 c2 c4 c5; do timeout -k 10 300 python bench.py --workload $w --no-pmc --cpu-seconds 5 > gpurun_out/r02_${w}_bench.json 2> gpurun_out/r02_${w}_bench.err; echo "$w rc $?"; done
timeout -k 10 300 python bench.py --workload c5 --hdri --no-pmc --no-cpu-baseline > gpurun_out/r02_c5_hdri_bench.json 2> gpurun_out/r02_c5_hdri_bench.err; echo "c5 hdri rc $?"
for f in gpurun_out/r02_c*_bench.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f', d['value'], d['mrays_per_s'], d['ms_per_step'], d['roofline']['frac'])"; done
python tests/tools/strong_scaling_estimate.py c3 > gpurun_out/r02_scaling_estimate.log 2>&1; cat gpurun_out/r02_scaling_estimate.log
python tests/tools/tree_probe.py c2 c3 c5 > gpurun_out/r02_tree_probe.log 2>&1; cut -c1-60,150-270 gpurun_out/r02_tree_probe.log
