#!/bin/bash
# What does each phase of the trace kernel cost?  Library variants that execute ONE phase twice (same results), built on the
# GPU box; kernel time and VALU instructions per wave-segment of each (1080p, SPP samples).  rejcap2/3: the rejection loop stops after 2 / 3
# trials (WRONG image: the upper bound of what parking its stragglers could gain); cmp: sign collection by v_cmp -> SGPR masks (same image); fastdiv: approximate reciprocals / reciprocal square roots instead of the
# IEEE divisions and square roots of the shading (WRONG image: the ceiling of exact-but-cheaper sequences); operands: the ray-operand build twice;
# noaccum: nothing is added to the pixels (WRONG image: what the miss path costs).  usage: tools/gpu_probe_phases.sh [names...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/probe
declare -A FL=( [base]="" [mfma]="-DRTW_DUP_MFMA" [eval]="-DRTW_DUP_EVAL" [extract]="-DRTW_DUP_EXTRACT" [resolve]="-DRTW_DUP_RESOLVE_PAIRS" [reject]="-DRTW_DUP_REJECT" [noskip]="-DRTW_SCAN_SKIP=0" [rejcap3]="-DRTW_PROBE_REJ_CAP=3" [rejcap2]="-DRTW_PROBE_REJ_CAP=2" [cmp]="-DRTW_SCAN_CMP=1" [fastdiv]="-DRTW_PROBE_FASTDIV" [operands]="-DRTW_DUP_OPERANDS" [noaccum]="-DRTW_PROBE_NO_ACCUM" )
NAMES=${@:-base mfma eval extract resolve reject noskip}
for n in $NAMES; do make -s -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_p_$n.so EXTRA="${FL[$n]}" 2>&1 | grep -E "error"; done
for n in $NAMES; do
  O=$R/gpurun_out/probe/$n; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp RTW_HIP_LIB=/tmp/librtw_p_$n.so rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py ${DT:-f32} 1920 ${SPP:-200} 50 plain 1 > $O/log.txt 2>&1)
  RTW_HIP_LIB=/tmp/librtw_p_$n.so python tools/gpu_quick.py ${DT:-f32} 1920 ${SPP:-200} 50 plain 3 2>/dev/null | grep kernel | tail -1 > $O/plain.txt
  python3 - <<PY
import csv, glob, re
c = {}
for f in glob.glob("$O/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] = c.get(row["Counter_Name"], 0) + float(row["Counter_Value"])
log = open("$O/log.txt").read()
m = re.search(r"(\d+)x(\d+) spp (\d+).*kernel ([0-9.]+) ms.*segs/sample ([0-9.]+)", log)
segs = float(m.group(5)) * int(m.group(1)) * int(m.group(2)) * int(m.group(3))
m2 = re.search(r"kernel ([0-9.]+) ms", open("$O/plain.txt").read())
print("%-10s VALU per wave-segment %7.0f   kernel %s ms (unprofiled, warm: %s ms)" % ("$n", c["SQ_INSTS_VALU"] / (segs / 64), m.group(4), m2.group(1) if m2 else "?"))
PY
done
