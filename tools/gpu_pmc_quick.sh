#!/bin/bash
# quick PMC: clock (GRBM_GUI_ACTIVE / duration) and VALU utilisation of the trace kernel
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/pmcq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT -o q -- python $R/bench.py --steps 1 --warmup 0 --spp ${1:-200} --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
dur = {}
for row in csv.DictReader(open(glob.glob("$OUT/*kernel_trace.csv")[0])):
    if "trace_kernel" in row["Kernel_Name"]: dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
c = collections.defaultdict(float)
for row in csv.DictReader(open(glob.glob("$OUT/*counter_collection.csv")[0])):
    if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] += float(row["Counter_Value"])
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("dur_ms %.2f clock_GHz %.3f valu_insts %.4g valu_util(2cyc) %.3f  waves %d  wait_any %.3f wait_inst %.3f active %.3f salu/valu %.3f" % (
    dur / 1e6, cyc / dur, c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] * 2 / (1024 * cyc), c["SQ_WAVES"],
    c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_SALU"]/c["SQ_INSTS_VALU"]))
PY
