#!/bin/bash
# What to run on the first box with more than one MI355X: the multi-GPU tests (they skip on one GPU), then the headline bench at
# N = 1, 2, 4, 8 (as many as are visible) with both collectives, one JSON line per run, and the in-library device list (peer copies
# and RTW_FLAG_RCCL_REDUCE) through the host-buffer entry point -- all of it by bench.py (`--gpus N`, `--in-library-devices N`).  usage: tools/gpu_multi.sh [outdir=gpurun_out/multi]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=${1:-$R/gpurun_out/multi}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible devices: $NG"
timeout 1200 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round6.py -q -k "multi_gpu or multi_device or gather or rccl or in_library or shards" -rs 2>&1 | tail -8
for N in 1 2 4 8; do
  [ $N -le $NG ] || continue
  for C in reduce gather; do
    [ $N -eq 1 ] && [ $C = gather ] && continue
    timeout 900 python bench.py --gpus $N --steps 5 --warmup 2 --collective $C --no-cpu-baseline --no-extras > $O/bench_n${N}_$C.json 2> $O/bench_n${N}_$C.err
    python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_n${N}_$C.json"))
    print("N=$N $C: %.1f Msamples/s  %.2f ms/step  render max/min %.2f / %.2f ms  collective %.3f ms  backend %s  sha %s" % (
        d["value"], d["ms_per_step"], d["render_ms_max"], d["render_ms_min"], d["collective_ms"], d["backend"], d["frame_sha256"][:12]))
except Exception as e:
    print("N=$N $C: FAILED", e); print(open("$O/bench_n${N}_$C.err").read()[-1500:])
PY
  done
done
# the in-library device list (one process; what a Julia caller gets with devices=...): bench.py times it -- peer copies and the RCCL reduce
for N in 1 2 4 8; do
  [ $N -le $NG ] || continue
  timeout 900 python bench.py --in-library-devices $N --steps 3 > $O/bench_inlib_n${N}.json 2> $O/bench_inlib_n${N}.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/bench_inlib_n${N}.json"))["in_library_devices"]
    for k in ("peer", "rccl_reduce"):
        v = d[k]
        print("in-library N=$N %-11s: %s" % (k, v.get("error") or "%.2f ms  %.1f Msamples/s  kernel max %.2f ms  per device %s  gather_path %s  frame equal %s" % (
            v["ms"], v["value"], v["kernel_ms_max"], [x["kernel_ms"] for x in v["per_device_kernel_ms"]], v["gather_path_names"], v.get("frame_sha256_equal"))))
except Exception as e:
    print("in-library N=$N: FAILED", e); print(open("$O/bench_inlib_n${N}.err").read()[-1500:])
PY
done
