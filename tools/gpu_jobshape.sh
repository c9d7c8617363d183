#!/bin/bash
# job shapes (column strips): kernel time and HBM write traffic per job size.  usage: tools/gpu_jobshape.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python -m pytest tests/test_gpu_round2.py tests/test_gpu_render.py -m gpu -x -q 2>&1 | tail -2
for dt in f32 f64; do for jp in 4 8 16; do
  O=$R/gpurun_out/jobshape/${dt}_$jp; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp RTW_JOB_PIXELS=$jp rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o p -- python $R/tools/gpu_quick.py $dt 1920 1000 50 plain 1 > $O/log.txt 2>&1)
  RTW_JOB_PIXELS=$jp python tools/gpu_quick.py $dt 1920 1000 50 plain 3 2>/dev/null | grep kernel | tail -1 > $O/plain.txt
  python3 - <<PY
import csv, glob, re
w = 0
for f in glob.glob("$O/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "trace_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "WRITE_SIZE": w += float(row["Counter_Value"])
m2 = re.search(r"kernel ([0-9.]+) ms", open("$O/plain.txt").read())
alg = 1920 * 1080 * 3 * (8 if "$dt" == "f64" else 4)
print("$dt job_pixels $jp: WRITE_SIZE %.1f MB = %.2f x the frame; kernel %s ms" % (w * 1024 / 1e6, w * 1024 / alg, m2.group(1) if m2 else "?"))
PY
done; done
