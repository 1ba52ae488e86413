# The two counter questions of VERDICT r03 #5 (run on the GPU box from the repo root): FETCH_SIZE for divergent 16-byte gathers, and what
# TD_TD_BUSY means when the vector ALU is idle. Output: gpurun_out/<tag>_counter_questions.txt
set -o pipefail
tag=${1:-r04}
out=gpurun_out/${tag}_counter_questions.txt
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w tests/tools/micro/fetch_gather.hip -o gpurun_out/fetch_gather || exit 1
export TMPDIR=/tmp
{
echo "# plain run"; gpurun_out/fetch_gather 4
for grp in "FETCH_SIZE" "TD_TD_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum"; do
  d=gpurun_out/${tag}_fg_$(echo $grp | tr ' ' '+')
  rm -rf "$d"
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$d" -- gpurun_out/fetch_gather 4 > "$d.log" 2>&1 || { echo "pass $grp failed"; tail -5 "$d.log"; continue; }
  python3 - "$d" <<'PY'
import csv, glob, os, sys
root = sys.argv[1]
rows = {}
for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0])
        rows.setdefault(k, {"ns": float(r["End_Timestamp"]) - float(r["Start_Timestamp"])})[r["Counter_Name"]] = float(r["Counter_Value"])
for (disp, name), c in sorted(rows.items(), key=lambda kv: int(kv[0][0])):
    ns = c.pop("ns")
    print(f"dispatch {disp:>3s} {name:14s} {ns / 1e6:9.3f} ms  " + "  ".join(f"{k} {v:.6g}" for k, v in sorted(c.items())))
PY
done
} > "$out" 2>&1
cat "$out"
