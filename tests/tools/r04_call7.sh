set -o pipefail
mkdir -p gpurun_out
run() { echo "== $1 SOL_SWITCH_FIRST=$2"; SOLSTRALE_BUILD_DIR=$3 SOL_SWITCH_FIRST=$2 timeout -k 10 300 python tests/tools/perf_quick.py c1 c2 c3 c5 c3h --spp 64; }
{
run base 16 $PWD/_var/base
run new 16 ""
run new 32 ""
run new 40 ""
run new 48 ""
run new 56 ""
run base 16 $PWD/_var/base
run new 16 ""
} > gpurun_out/r04h_switch_first.txt 2>&1
cat gpurun_out/r04h_switch_first.txt
