"""Condense the rocprofv3 output of scripts/profile_passes.sh into <tag>_kernel_stats.csv and <tag>_pmc_summary.json (written to
gpurun_out/; copy them into profiles/).  Per-launch means over the launches of the solve kernel."""
import csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(out)
stats = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True))
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
summary = {"batch": batch, "command": "scripts/profile_passes.sh: rocprofv3 --kernel-trace --stats | --pmc <group> (one pass per group) -- python3 bench.py --no-cpu-baseline --no-extra",
           "kernel": None, "kernel_stats": None, "counters": {}}
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w") as f:
        f.write(open(stats[0]).read())
    k = max((r for r in rows if "rti_solve_kernel" in r["Name"] or "rti_split_kernel" in r["Name"]), key=lambda r: float(r["TotalDurationNs"]), default=None)
    if k:
        summary["kernel"] = k["Name"]
        summary["kernel_stats"] = {"calls": int(k["Calls"]), "avg_ns": float(k["AverageNs"]), "percentage": float(k["Percentage"])}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    acc = {}
    for r in csv.DictReader(open(f)):
        if "rti_solve_kernel" not in r.get("Kernel_Name", "") and "rti_split_kernel" not in r.get("Kernel_Name", ""):
            continue
        if summary["kernel"] and r["Kernel_Name"] != summary["kernel"]:
            continue
        a = acc.setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
    for name, (tot, n) in acc.items():
        summary["counters"][name] = {"mean_per_launch": tot / n, "launches": n}
c = summary["counters"]
d = {}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    summary["hbm_traffic_bytes_per_launch"] = {
        "fetch_raw_kb": c["FETCH_SIZE"]["mean_per_launch"], "write_raw_kb": c["WRITE_SIZE"]["mean_per_launch"],
        "fetch_bytes_corrected_x2": 2 * 1024 * c["FETCH_SIZE"]["mean_per_launch"], "write_bytes": 1024 * c["WRITE_SIZE"]["mean_per_launch"],
        "note": "FETCH_SIZE/WRITE_SIZE are in KB; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream (x2 correction "
                "shown), other access widths (ours: 8 B per lane) are uncalibrated; WRITE_SIZE is exact for streaming stores"}
if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
    d["valu_active_fraction_of_wave_cycles"] = c["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"]
if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
    d["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"]["mean_per_launch"] / c["SQ_WAVES"]["mean_per_launch"]
if "SQ_THREAD_CYCLES_VALU" in c and "SQ_INSTS_VALU" in c:
    # lanes enabled (EXEC) per VALU instruction: SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU reads exactly K on a kernel that runs with K of 64 lanes enabled
    # (scripts/bin_src/lanes_counter_test.hip -> profiles/r04_lanes_counter_calibration.json: 64.0, 32.0, 16.0, 8.0, 1.0).  The sweeps of these kernels keep EXEC
    # full on purpose (unconditional arithmetic, dead stores), so this is an upper bound on the lanes that carry data: bench.py's lanes_useful models those.
    d["valu_lanes_active"] = c["SQ_THREAD_CYCLES_VALU"]["mean_per_launch"] / c["SQ_INSTS_VALU"]["mean_per_launch"]
if "SQ_INSTS_VALU" in c and summary.get("kernel_stats"):
    # fraction of the chip's VALU issue slots: one wave64 VALU instruction holds a SIMD for 4 cycles; 1024 SIMDs; 2.4 GHz nominal
    d["valu_issue_fraction"] = c["SQ_INSTS_VALU"]["mean_per_launch"] * 4.0 / (1024 * summary["kernel_stats"]["avg_ns"] * 2.4)
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and summary.get("kernel_stats"):
    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (MI355X_MICROARCH.md: 64 per v_mfma_f64_16x16x4, 16 per v_mfma_f64_4x4x4_4b -- checked against SQ_INSTS_MFMA),
    # summed over the SIMDs the launch ran on; utilisation = that / (SIMDs x kernel duration in cycles at the 2.4 GHz nominal clock)
    simds = 1024
    d["mfma_busy_cycles_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / simds
    d["mfma_busy_fraction_of_kernel_time"] = d["mfma_busy_cycles_per_simd"] / (summary["kernel_stats"]["avg_ns"] * 2.4)
    if "SQ_INSTS_MFMA" in c and c["SQ_INSTS_MFMA"]["mean_per_launch"]:
        d["mfma_busy_cycles_per_mfma_instruction"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["mean_per_launch"] / c["SQ_INSTS_MFMA"]["mean_per_launch"]
if "SQ_INSTS_VALU_MFMA_MOPS_F64" in c and summary.get("kernel_stats"):
    # one MOPS count = 512 floating-point operations (rocprofv3 counter definition); rate against the FP64 matrix peak of 78.6 TFLOP/s
    flops = 512.0 * c["SQ_INSTS_VALU_MFMA_MOPS_F64"]["mean_per_launch"]
    d["mfma_f64_flops_per_launch"] = flops
    d["mfma_f64_tflops"] = flops / summary["kernel_stats"]["avg_ns"] * 1e-3
    d["mfma_f64_fraction_of_peak_78.6"] = d["mfma_f64_tflops"] / 78.6
summary["derived"] = d
json.dump(summary, open(os.path.join(root, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps({"kernel": summary["kernel"], "stats": summary["kernel_stats"], "derived": d, "counters": sorted(c)}, indent=1))
