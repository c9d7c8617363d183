#!/bin/bash
# Quick PMC look at one trace kernel launch (1080p, SPP spp): per wave-segment instruction counts and busy fractions.
#   usage: tools/gpu_pmc_quick.sh [spp=200] [gpu_quick.py mode: plain|pool|cull]      env: RTW_HIP_LIB
R=${GRAFT_REPO_ROOT:-/root/repo}; SPP=${1:-200}; MODE=${2:-plain}; O=$R/gpurun_out/pmcq_$MODE; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/gpu_quick.py f32 1920 $SPP 50 $MODE 1"
pmc() { tag=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$tag -o p -- $CMD > $O/$tag.log 2>&1; }
pmc A GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc B SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pmc C SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT
grep -h "kernel\|Msamples" $O/A.log | tail -2
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); dur = {}
for tag in "ABC":
    for f in glob.glob("$O/%s/*counter_collection.csv" % tag):
        for row in csv.DictReader(open(f)):
            if "trace" in row["Kernel_Name"]: tot[row["Counter_Name"]] += float(row["Counter_Value"])
    for f in glob.glob("$O/%s/*kernel_trace.csv" % tag):
        for row in csv.DictReader(open(f)):
            if "trace" in row["Kernel_Name"]: dur[tag] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
import re
sps = float(re.search(r"segs/sample ([0-9.]+)", open("$O/A.log").read()).group(1))      # counted by the kernel (depends on the numerics mode)
segs = 1920 * 1080 * $SPP * sps; ws = segs / 64
cyc = tot["GRBM_GUI_ACTIVE"] / 8; simd = 1024 * cyc
print("kernel ms per pass", dur)
print("per wave-segment: VALU %.0f  SALU %.0f  MFMA %.1f  LDS %.1f  VMEM %.1f  SMEM %.2f" % (tot["SQ_INSTS_VALU"] / ws, tot["SQ_INSTS_SALU"] / ws, tot["SQ_INSTS_MFMA"] / ws, tot["SQ_INSTS_LDS"] / ws, tot["SQ_INSTS_VMEM"] / ws, tot["SQ_INSTS_SMEM"] / ws))
print("SIMD cycles per wave-segment %.0f; MFMA busy %.1f%%; VALU x2 %.1f%%; clock %.2f GHz" % (simd / ws, 100 * tot["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 100 * 2 * tot["SQ_INSTS_VALU"] / simd, cyc / (dur.get("A", 1) * 1e6)))
wc = tot["SQ_WAVE_CYCLES"]
print("of wave-cycles: WAIT_ANY %.1f%%  WAIT_INST_ANY %.1f%%  ACTIVE_INST_ANY %.1f%%; waves/SIMD avg %.2f" % (100 * tot["SQ_WAIT_ANY"] / wc, 100 * tot["SQ_WAIT_INST_ANY"] / wc, 100 * tot["SQ_ACTIVE_INST_ANY"] / wc, wc / 4 / simd if simd else 0))
print("lane utilisation of VALU: %.1f%%" % (100 * tot["SQ_THREAD_CYCLES_VALU"] / (64 * tot["SQ_ACTIVE_INST_VALU"]) if tot["SQ_ACTIVE_INST_VALU"] else 0))
print("LDS: idx_active %.3g  bank conflict %.3g (%.1f%%)  ACTIVE_INST_LDS %.3g  WAIT_INST_LDS %.3g" % (tot["SQ_LDS_IDX_ACTIVE"], tot["SQ_LDS_BANK_CONFLICT"], 100 * tot["SQ_LDS_BANK_CONFLICT"] / max(tot["SQ_LDS_IDX_ACTIVE"], 1), tot["SQ_ACTIVE_INST_LDS"], tot["SQ_WAIT_INST_LDS"]))
print({k: "%.4g" % v for k, v in sorted(tot.items())})
PY
