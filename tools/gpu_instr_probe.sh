#!/bin/bash
# VALU instructions per wave-segment of library variants (instruction-count probes: a phase executed twice)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for lib in "$@"; do
  O=$R/gpurun_out/probe_$(basename $lib .so); rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp RTW_HIP_LIB=$R/$lib rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py ${DT:-f32} 1920 ${SPP:-200} 50 plain 1 > $O/log.txt 2>&1)
  python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$O/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
log = open("$O/log.txt").read()
m = re.search(r"(\d+)x(\d+) spp (\d+).*kernel ([0-9.]+) ms.*segs/sample ([0-9.]+)", log)
segs = float(m.group(5)) * int(m.group(1)) * int(m.group(2)) * int(m.group(3))
print("%-28s VALU per wave-segment %7.0f   kernel %s ms" % ("$lib".split("/")[-1], c["SQ_INSTS_VALU"] / (segs / 64), m.group(4)))
PY
done
