#!/bin/bash
# quick PMC of the trace kernel: clock, VALU instructions per wave-segment, VALU busy, wait split.
# usage: gpu_pmc2.sh <tag> <dtype> <width> <spp> [cull]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}; OUT=$R/gpurun_out/pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/gpu_quick.py ${2:-f32} ${3:-1920} ${4:-100} 50 ${5:-plain} 1"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/a -o q -- $CMD > $OUT/a.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/b -o q -- $CMD > $OUT/b.txt 2>&1
grep -h "Msamples" $OUT/a.txt | tail -1
python3 - <<PY
import csv, glob, collections, re
c = collections.defaultdict(float); dur = 0
for d in ("a", "b"):
    for f in glob.glob("$OUT/%s/*kernel_trace.csv" % d):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"] and d == "a": dur = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    for f in glob.glob("$OUT/%s/*counter_collection.csv" % d):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] += float(row["Counter_Value"])
line = open("$OUT/a.txt").read()
m = re.search(r"tests ([0-9.e+]+)", line); segs = None
m2 = re.search(r"segs/sample ([0-9.]+)", line); m3 = re.search(r"(\d+)x(\d+) spp (\d+)", line)
if m2 and m3: segs = float(m2.group(1)) * int(m3.group(1)) * int(m3.group(2)) * int(m3.group(3))
cyc = c["GRBM_GUI_ACTIVE"] / 8
print("dur_ms %.2f clock_GHz %.3f valu_insts %.4g valu_busy(2cyc) %.3f wait_any %.3f wait_inst %.3f active %.3f salu/valu %.3f" % (
    dur / 1e6, cyc / max(dur, 1), c["SQ_INSTS_VALU"], c["SQ_INSTS_VALU"] * 2 / (1024 * max(cyc, 1)),
    c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_SALU"] / c["SQ_INSTS_VALU"]))
if segs: print("VALU per wave-segment %.0f   (segments %.4g)" % (c["SQ_INSTS_VALU"] / (segs / 64), segs))
print("lane utilisation %.3f  lds_insts/valu %.4f vmem/valu %.5f smem/valu %.4f flat %.4g" % (
    c["SQ_THREAD_CYCLES_VALU"] / max(c["SQ_ACTIVE_INST_VALU"], 1) / 64, c["SQ_INSTS_LDS"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_VMEM"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_SMEM"] / c["SQ_INSTS_VALU"], c["SQ_INSTS_FLAT"]))
PY
