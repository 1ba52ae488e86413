"""Folds the rocprofv3 --pmc passes of tests/tools/profile_round.sh into one JSON summary (not a pytest).
Usage: python tests/tools/pmc_summary.py gpurun_out/<tag>_pmc <kernel-name-substring> <samples-in-the-launch> > profiles/<tag>_pmc_summary.json

Counters are summed over every dispatch whose kernel name contains the substring (bench.py --steps 1 --warmup 0 launches
the uncounted render kernel exactly once). Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM):
FETCH_SIZE / WRITE_SIZE are KiB of L2<->fabric requests (Infinity-Cache hits included); on gfx950 FETCH_SIZE tallies a
128-B request as 64 B, so reads are doubled for the "corrected" figure - an UPPER bound here: a divergent 16-byte gather is one
64-byte request per miss and is counted as it is (profiles/r04_counter_questions.txt), so the truth lies between the two figures."""
import csv
import glob
import json
import os
import sys


def main():
    root, kname, samples = sys.argv[1], sys.argv[2], float(sys.argv[3])
    counters, dur_ns = {}, []
    # one file per counter group: the newest (gpurun merges every call's output into the same local directory, so a group
    # directory can hold the passes of earlier calls too)
    newest = {}
    for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        grp = os.path.relpath(f, root).split(os.sep)[0]
        if grp not in newest or os.path.getmtime(f) > os.path.getmtime(newest[grp]):
            newest[grp] = f
    for f in sorted(newest.values()):
        seen = set()
        for row in csv.DictReader(open(f)):
            if kname not in row["Kernel_Name"]:
                continue
            counters[row["Counter_Name"]] = counters.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if row["Dispatch_Id"] not in seen:
                seen.add(row["Dispatch_Id"])
                dur_ns.append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
    c = c0 = counters
    g = lambda k: c.get(k, float("nan"))
    n_simd = 256 * 4
    kernel_ms = sum(dur_ns) / max(1, len(dur_ns)) / 1e6
    # shader clock during the profiled launch: GRBM_GUI_ACTIVE sums the 8 XCDs (guide, "DVFS give-back"); else SQ_BUSY_CYCLES
    # (summed over the 32 shader engines); else the 2.4 GHz maximum
    if "GRBM_GUI_ACTIVE" in c0:
        clock = c0["GRBM_GUI_ACTIVE"] / 8.0 / (kernel_ms * 1e-3)
    elif "SQ_BUSY_CYCLES" in c0:
        clock = c0["SQ_BUSY_CYCLES"] / 32.0 / (kernel_ms * 1e-3)
    else:
        clock = 2.4e9
    k_cycles = kernel_ms * 1e-3 * clock
    d = {
        "kernel_ms_under_pmc (mean of the passes)": round(kernel_ms, 2),
        "valu_insts_per_sample": g("SQ_INSTS_VALU") / samples,
        "salu_insts_per_sample": g("SQ_INSTS_SALU") / samples,
        "vmem_read_insts_per_sample": g("SQ_INSTS_VMEM_RD") / samples,
        "lds_insts_per_sample": g("SQ_INSTS_LDS") / samples,
        # SQ_* cycle counters count quad-cycles (guide: 'tick vs SQ PMC units')
        "valu_lane_utilization (THREAD_CYCLES_VALU / (ACTIVE_INST_VALU*64))": g("SQ_THREAD_CYCLES_VALU") / (g("SQ_ACTIVE_INST_VALU") * 64.0),
        "shader_clock_ghz": clock / 1e9,
        "simd_valu_busy_frac_at_4_cycles_per_instr (ACTIVE_INST_VALU*4 / (SIMDs * kernel cycles))": g("SQ_ACTIVE_INST_VALU") * 4.0 / (n_simd * k_cycles),
        "valu_issue_frac_of_2_cycle_peak (INSTS_VALU*2 / (SIMDs * kernel cycles))": g("SQ_INSTS_VALU") * 2.0 / (n_simd * k_cycles),
        "wave_cycles_waiting_frac (WAIT_ANY / WAVE_CYCLES)": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"),
        "wave_cycles_issue_stall_frac (WAIT_INST_ANY / WAVE_CYCLES)": g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES"),
        "mean_waves_per_simd (WAVE_CYCLES*4 / (SIMDs * kernel cycles))": g("SQ_WAVE_CYCLES") * 4.0 / (n_simd * k_cycles),
        "l1_hit_rate (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)": 1.0 - g("TCP_TCC_READ_REQ_sum") / g("TCP_TOTAL_CACHE_ACCESSES_sum"),
        "l2_hit_rate (TCC_HIT / (TCC_HIT + TCC_MISS))": g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")),
        "ta_busy_frac (TA_BUSY_avr cycles / kernel cycles)": g("TA_BUSY_avr") / k_cycles,
        # (TD_TD_BUSY is not reported: it reads 0.9 - 0.99 for ANY kernel with vector-memory requests in flight - a VALU-idle gather from
        # HBM, from L2, a streaming read alike - profiles/r04_counter_questions.txt)
        "fabric_read_bytes_uncorrected (FETCH_SIZE KiB * 1024)": g("FETCH_SIZE") * 1024.0,
        "fabric_read_bytes_corrected_x2": g("FETCH_SIZE") * 2048.0,
        "fabric_write_bytes (WRITE_SIZE KiB * 1024)": g("WRITE_SIZE") * 1024.0,
        "fabric_bytes_per_sample_corrected": (g("FETCH_SIZE") * 2048.0 + g("WRITE_SIZE") * 1024.0) / samples,
    }
    d = {k: v for k, v in d.items() if v == v}  # drop rows whose counters were not collected (NaN)
    counters = {k: v for k, v in counters.items() if not (k == "SQ_ACTIVE_INST_VMEM" and v == 0.0)}  # reads 0 on gfx950: not a measurement
    json.dump({"kernel": kname, "samples_in_launch": samples, "counters": counters, "derived": d}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
