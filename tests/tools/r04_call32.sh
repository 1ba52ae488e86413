#!/bin/bash
# round 4, call 32: the bench line carries the value with every sample traced beside the headline (C3, C5)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python bench.py --no-build > gpurun_out/call32_bench.json 2> gpurun_out/call32_bench.err; echo rc $?
timeout -k 10 400 python bench.py --workload c5 --steps 2 --no-cpu-baseline --no-pmc --no-build > gpurun_out/call32_c5_bench.json 2> gpurun_out/call32_c5_bench.err; echo rc $?
for f in gpurun_out/call32_bench.json gpurun_out/call32_c5_bench.json; do python -c "
import json; d=json.load(open('$f')); print(d['value'], d['background_blocks']['value_with_every_sample_traced'], d['background_blocks']['sample_fraction'], d['roofline']['kernel_ms'])"; done
