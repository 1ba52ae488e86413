set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python tests/tools/variants.py run --spp 64 --scenes "c2 c3 c5 c3h" default pm6 pm12 pm16 fastinv default fastinv > gpurun_out/r04n_variants.txt 2>&1; echo "rc $?"; cat gpurun_out/r04n_variants.txt
