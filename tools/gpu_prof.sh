#!/bin/bash
# rocprofv3 passes over a short bench.py run: kernel-trace stats + PMC groups (each its own run)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01}
SPP=${2:-50}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
run_pmc() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run_pmc pmcA SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM
run_pmc pmcB SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU
run_pmc pmcC GRBM_GUI_ACTIVE FETCH_SIZE
run_pmc pmcD GRBM_GUI_ACTIVE WRITE_SIZE
run_pmc pmcE SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT32
find $OUT -name "*.csv" | head -40
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
python3 - <<PY
import csv, glob, collections
for g in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(g)):
        k = row["Kernel_Name"][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print("==", g.split("/")[-1])
    for k, d in agg.items():
        if "trace" in k:
            print("  ", k, {c: v for c, v in d.items()})
PY
tail -3 $OUT/trace.log
