#!/bin/bash
# VALU / MFMA instructions per wave-segment of ONE launch of the trace kernel (rocprofv3 --pmc, own pass), next to its warm kernel time.
#   usage: tools/gpu_pmc_quick.sh [dtype=f32] [spp=200] [mode=plain|cull] [depth=50] [width=1920]      env: RTW_HIP_LIB (a library variant)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; DT=${1:-f32}; SPP=${2:-200}; MODE=${3:-plain}; DEPTH=${4:-50}; W=${5:-1920}
O=$R/gpurun_out/pmcq_$$; rm -rf $O; mkdir -p $O
(cd /tmp && TMPDIR=/tmp rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py $DT $W $SPP $DEPTH $MODE 1 > $O/log.txt 2>&1)
python tools/gpu_quick.py $DT $W $SPP $DEPTH $MODE 3 2>/dev/null | grep kernel | tail -1 > $O/warm.txt
python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$O/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
log = open("$O/log.txt").read()
m = re.search(r"(\d+)x(\d+) spp (\d+).*kernel ([0-9.]+) ms.*segs/sample ([0-9.]+)", log)
segs = float(m.group(5)) * int(m.group(1)) * int(m.group(2)) * int(m.group(3)); ws = segs / 64
m2 = re.search(r"kernel ([0-9.]+) ms", open("$O/warm.txt").read()); warm = float(m2.group(1)) if m2 else float("nan")
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
print("$DT $W x $SPP spp d$DEPTH $MODE: per wave-segment VALU %.0f  SALU %.0f  MFMA %.1f  LDS %.0f | warm kernel %.2f ms = %.3f ns per wave-segment | MFMA-busy %.1f %% of SIMD cycles"
      % (c["SQ_INSTS_VALU"] / ws, c.get("SQ_INSTS_SALU", 0) / ws, c.get("SQ_INSTS_MFMA", 0) / ws, c.get("SQ_INSTS_LDS", 0) / ws, warm, warm * 1e6 / ws,
         100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0))
PY
rm -rf $O
