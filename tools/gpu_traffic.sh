#!/bin/bash
# HBM traffic of the trace kernel: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (guide: they do not fit one pass)
# usage: gpu_traffic.sh <tag> <bench args...>
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; shift; O=$R/gpurun_out/traffic_$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $@"
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $CMD > $O/f.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $CMD > $O/w.log 2>&1
python3 - <<PY
import csv, glob
c = {}
for f in glob.glob("$O/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
print("$TAG", {k: round(v, 1) for k, v in c.items()}, "=> HBM bytes/launch (FETCH x2 + WRITE, KiB units): %.1f MB" % ((2 * c.get("FETCH_SIZE", 0) + c.get("WRITE_SIZE", 0)) * 1024 / 1e6))
PY
