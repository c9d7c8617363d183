#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/ubench_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d $OUT/a -o a -- $R/build/ubench_scan 96 > $OUT/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/b -o b -- $R/build/ubench_scan 96 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
dur = collections.defaultdict(list)
for row in csv.DictReader(open(glob.glob("$OUT/a/*kernel_trace.csv")[0])):
    dur[row["Kernel_Name"][:60]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for g in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(g)):
        agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", g.split("/")[-1])
    for k, d in agg.items():
        if "V7" in k or "V9" in k or "valu_probe" in k or "V4" in k:
            # last dispatch of each kernel (the 'miss' set for scans)
            print("  ", k[:50], {c: v[-1] for c, v in d.items()}, "dur_ns(last)", dur[k][-1] if k in dur else None)
PY
