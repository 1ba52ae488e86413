#!/bin/bash
# round 5: OBJ ingest at BASELINE size through bench.py --obj (exports the stand-ins as OBJ + MTL first; tests/tools/export_obj.py)
cd "$(dirname "$0")"
python export_obj.py atrium /tmp/r05_obj 262267 1024 || exit 1
python export_obj.py statue /tmp/r05_obj || exit 1
cd ../..
python bench.py --obj /tmp/r05_obj/atrium.obj --workload c3 --spp 64 --steps 2 --warmup 1 --camera=-13.0,2.2,0.6,6.0,4.5,-0.4,55 --light=-6,13.5,-2,12,0,0,0,0,4,18,17,15 --no-cpu-baseline --no-pmc > gpurun_out/r05_obj_atrium_bench.json 2> gpurun_out/r05_obj_atrium_bench.err || exit 1
python bench.py --obj /tmp/r05_obj/statue.obj --workload c5 --spp 64 --steps 2 --warmup 1 --camera=4.5,3.6,8.0,0.2,2.6,0.0,38 --light=-2.5,8.5,-1,5,0,0,0,0,4,20,19,17 --no-cpu-baseline --no-pmc > gpurun_out/r05_obj_statue_bench.json 2> gpurun_out/r05_obj_statue_bench.err || exit 1
ls -la /tmp/r05_obj/*.obj
