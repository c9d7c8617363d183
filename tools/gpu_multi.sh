#!/bin/bash
# What to run on the first box with more than one MI355X: the multi-GPU tests (they skip on one GPU), then the headline bench at
# N = 1, 2, 4, 8 (as many as are visible) with both collectives, one JSON line per run, and the in-library device list (peer copies
# and RTW_FLAG_RCCL_REDUCE) timed through the host-buffer entry point.  usage: tools/gpu_multi.sh [outdir=gpurun_out/multi]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=${1:-$R/gpurun_out/multi}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible devices: $NG"
timeout 1200 python -m pytest tests/test_gpu_round4.py -q -k "multi_gpu or gather or rccl" -rs 2>&1 | tail -8
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
python3 - <<PY
import time, numpy as np, torch
torch.cuda.init()
import rtw_amd as R
T = np.float32; R.reseed(); scene = R.scene_random_spheres(elem_type=T); cam = R.t_cam1(elem_type=T)
ng = torch.cuda.device_count()
for n in (1, 2, 4, 8):
    if n > ng: continue
    for rccl in (False, True):
        kw = dict(devices=list(range(n)), rccl_reduce=rccl) if (n > 1 or rccl) else {}
        R.render(scene, cam, 1920, 1000, depth=50, **kw)                  # first call: uploads, communicator
        t = time.perf_counter(); img = R.render(scene, cam, 1920, 1000, depth=50, **kw); dt = time.perf_counter() - t
        st = R.last_stats()
        print("in-library N=%d %s: %.2f ms  %.1f Msamples/s  kernel max %.2f ms  gather_path %d" % (n, "rccl" if rccl else "peer", dt * 1e3, 1920 * 1080 * 1000 / dt / 1e6, st["kernel_ms"], st["gather_path"]))
PY
