#!/bin/bash
# Calibrates the lane-activity counters on kernels of known EXEC popcount (scripts/bin_src/lanes_counter_test.hip) -> gpurun_out/lanes_counter_calibration.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lanes_cal
rm -rf "$OUT"; mkdir -p "$OUT" "$ROOT/scripts/bin"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o "$ROOT/scripts/bin/lanes_counter_test" "$ROOT/scripts/bin_src/lanes_counter_test.hip" || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1 || true
grep -o "SQ_[A-Z_]*THREAD[A-Z_]*\|SQ_ACTIVE_INST_VALU[A-Z_0-9]*\|SQ_INST_CYCLES_VALU[A-Z_]*\|SQ_VALU_[A-Z_]*" "$OUT/counters_available.txt" | sort -u > "$OUT/candidates.txt"
cat "$OUT/candidates.txt"
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES" "SQ_INSTS_VALU SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    i=$((i + 1))
    rocprofv3 --pmc $grp --output-format csv -d "$OUT/pmc$i" -o pmc -- "$ROOT/scripts/bin/lanes_counter_test" > "$OUT/pmc$i.log" 2>&1 || echo "pass $i failed: $(tail -3 $OUT/pmc$i.log)"
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]; acc = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
res = {}
for k, c in sorted(acc.items()):
    m = {n: v[0] / v[1] for n, v in c.items()}
    d = dict(m)
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_ACTIVE_INST_VALU" in m and m["SQ_ACTIVE_INST_VALU"]:
        d["thread_cycles_over_active_inst"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_ACTIVE_INST_VALU"]
    if "SQ_THREAD_CYCLES_VALU" in m and m.get("SQ_INSTS_VALU"):
        d["thread_cycles_over_insts"] = m["SQ_THREAD_CYCLES_VALU"] / m["SQ_INSTS_VALU"]
    if "SQ_ACTIVE_INST_VALU" in m and m.get("SQ_INSTS_VALU"):
        d["active_inst_over_insts"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_INSTS_VALU"]
    res[k] = d
json.dump(res, open(os.path.join(os.path.dirname(out), "lanes_counter_calibration.json"), "w"), indent=1)
print(json.dumps(res, indent=1))
PY
