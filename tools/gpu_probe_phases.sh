#!/bin/bash
# What does each phase of the trace kernel cost?  Library variants that execute ONE phase twice (same results), built on the
# GPU box; kernel time and VALU instructions per wave-segment of each (1080p, SPP samples).  rejcap2/3: the rejection loop stops after 2 / 3
# trials (WRONG image: the upper bound of what parking its stragglers could gain); cmp: sign collection by v_cmp -> SGPR masks (same image); fastdiv: approximate reciprocals / reciprocal square roots instead of the
# IEEE divisions and square roots of the shading (WRONG image: the ceiling of exact-but-cheaper sequences); operands: the ray-operand build twice;
# noaccum: nothing is added to the pixels (WRONG image: what the miss path costs).
# Round 5 -- the matrix pipe's real share: mfma / mfma2 execute every MFMA pair 2 x / 3 x (same image); nomfma replaces the pairs by a constant "no candidate" (WRONG
# image: only the in-lane huge spheres are hit; the work per wave-segment WITHOUT the matrix pipe and without candidates).  Every build's pass also reads
# SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES and the wave-cycle split.   usage: tools/gpu_probe_phases.sh [names...]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/probe
declare -A FL=( [base]="" [mfma]="-DRTW_DUP_MFMA" [eval]="-DRTW_DUP_EVAL" [extract]="-DRTW_DUP_EXTRACT" [resolve]="-DRTW_DUP_RESOLVE_PAIRS" [reject]="-DRTW_DUP_REJECT" [noskip]="-DRTW_SCAN_SKIP=0" [rejcap3]="-DRTW_PROBE_REJ_CAP=3" [rejcap2]="-DRTW_PROBE_REJ_CAP=2" [cmp]="-DRTW_SCAN_CMP=1" [fastdiv]="-DRTW_PROBE_FASTDIV" [operands]="-DRTW_DUP_OPERANDS" [noaccum]="-DRTW_PROBE_NO_ACCUM" [mfma2]="-DRTW_DUP_MFMA=2" [nomfma]="-DRTW_PROBE_NO_MFMA" )
NAMES=${@:-base mfma eval extract resolve reject noskip}
echo "== ${DT:-f32} ${WIDTH:-1920} x ${SPP:-200} spp, depth ${DEPTH:-50}"
for n in $NAMES; do [ -f /tmp/librtw_p_$n.so ] || make -s -j8 -C raytracingweekend.jl_amd/csrc -B OUT=/tmp/librtw_p_$n.so OBJDIR=/tmp/obj_p_$n EXTRA="${FL[$n]}" 2>&1 | grep -E "error"; done
for n in $NAMES; do
  O=$R/gpurun_out/probe/$n; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp RTW_HIP_LIB=/tmp/librtw_p_$n.so rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py ${DT:-f32} ${WIDTH:-1920} ${SPP:-200} ${DEPTH:-50} plain 1 > $O/log.txt 2>&1)
  RTW_HIP_LIB=/tmp/librtw_p_$n.so python tools/gpu_quick.py ${DT:-f32} ${WIDTH:-1920} ${SPP:-200} ${DEPTH:-50} plain 3 2>/dev/null | grep kernel | tail -1 > $O/plain.txt
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
ws = segs / 64
cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8                       # shader cycles of the launch (8 XCDs count)
warm = float(m2.group(1)) if m2 else float("nan")
print("%-9s segs/sample %.4f  VALU/wave-seg %6.0f  MFMA/wave-seg %5.1f  kernel %8s ms (warm %8.2f ms = %6.2f ns per wave-segment)  MFMA-busy %4.1f %% of SIMD cycles  SQ_BUSY_CYCLES/cyc %.2f  wave-cycles: issuing %4.1f %% issue-stalled %4.1f %%"
      % ("$n", float(m.group(5)), c["SQ_INSTS_VALU"] / ws, c.get("SQ_INSTS_MFMA", 0) / ws, m.group(4), warm, warm * 1e6 / ws,
         100 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * cyc) if cyc else 0, c.get("SQ_BUSY_CYCLES", 0) / cyc if cyc else 0,
         100 * c.get("SQ_ACTIVE_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1), 100 * c.get("SQ_WAIT_INST_ANY", 0) / max(c.get("SQ_WAVE_CYCLES", 1), 1)))
PY
done
