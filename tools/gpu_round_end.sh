#!/bin/bash
# what the driver does at round end, plus the rocprof summary for profiles/
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/round
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python -m pytest tests -m gpu -q 2>&1 | tail -2
python bench.py > gpurun_out/round/bench_n1.json 2> gpurun_out/round/bench_n1.err; cat gpurun_out/round/bench_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 1 --warmup 0 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/round/trace -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/round/trace.log 2>&1
cat $R/gpurun_out/round/trace/*kernel_stats.csv
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/round/pmcF -o f -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/round/pmcF.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/round/pmcW -o w -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/round/pmcW.log 2>&1
python3 - <<PY
import csv, glob
for g in sorted(glob.glob("$R/gpurun_out/round/pmc*/*counter_collection.csv")):
    for row in csv.DictReader(open(g)):
        print(row["Kernel_Name"][:40], row["Counter_Name"], row["Counter_Value"])
PY
# the opt-in accelerated mode, same procedure
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/round/trace_cull -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --group-cull > $R/gpurun_out/round/trace_cull.log 2>&1
echo "== group-cull kernel stats"; cat $R/gpurun_out/round/trace_cull/*kernel_stats.csv
