set -o pipefail
mkdir -p gpurun_out
cd tests/tools
{ echo "End-of-round-4 build (GPU tree: reinsertion, three radii + probe, pre-splitting where kept; resumable medium) against the fp32 oracle, FULL frames (tests/tools/gpu_full_oracle.py): C3 1080p x 32 spp, C2 1080p x 32 spp, the reference test scene 800x400 x 64 spp, C1 400x400 x 50, the HETEROGENEOUS atrium 1080p x 16 spp (pre-split tree), C5 960x540 x 8 spp. bad = pixels over 1e-5 relative."
timeout -k 10 300 python gpu_full_oracle.py c3 32 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c2 32 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py test 64 800 400 default
timeout -k 10 300 python gpu_full_oracle.py c1 50 400 400 default
timeout -k 10 300 python gpu_full_oracle.py c3h 16 1920 1080 default
timeout -k 10 300 python gpu_full_oracle.py c5 8 960 540 default
timeout -k 10 500 python random_parity_sweep.py 9000 2000
} > ../../gpurun_out/r04o_full_frame_oracle.txt 2>&1
cat ../../gpurun_out/r04o_full_frame_oracle.txt
