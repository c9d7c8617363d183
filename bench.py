#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path render -> ray_color -> hit/scatter on MI355X.

  python bench.py --gpus N --steps K --warmup W [--dtype f32|f64] [--width 1920|3840]

One "step" = one full render of the workload.  Default = BASELINE.json's headline, configs[2]:
scene_random_spheres (reseed!(); 485 spheres), camera t_cam1, 1920x1080, 1000 spp, depth 50,
Float32.  `--dtype f64 --width 3840` is the single-GPU share of configs[4] (3840x2160, Float64); the
default run also times it (2 steps) and reports it as `f64_4k`.
The scene is already resident in HBM when the timed region starts and the image is left in HBM
on rank 0.  With N > 1 ranks the SAME image is tile-sharded over the ranks (strong scaling) and
the zero-padded shard framebuffers are summed onto rank 0 with one RCCL reduce over xGMI inside
the timed region (`--collective gather`: each rank sends only its compact shard).

N > 1: `python bench.py --gpus N ...` launched plainly (no WORLD_SIZE in the environment) starts its N ranks
itself (torch.distributed.run, rendezvous on 127.0.0.1) and REFUSES to run when fewer than N devices are visible;
launched by torch.distributed.run it checks WORLD_SIZE == N.  The line reports the world size the process group
observed, the backend, every rank's device and mean kernel time, and the SHA-256 of the assembled frame
(identical for every N: the image does not depend on the sharding).
RTW_BENCH_ONE_DEVICE=1 (test aid for one-GPU boxes): every rank uses cuda:0 and the collective runs over
gloo -- exercises the N > 1 control flow (sharding, collective, max-over-ranks timing), not RCCL itself.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- roofline of the only kernel (rtw::trace_kernel), against FIXED hardware yard-sticks (round 4: the definitions no
                  longer follow the kernel's formulation; DESIGN.md section 7 re-expresses rounds 1-3 in them):
                    bound "mfma": `achieved` = EXECUTED matrix-pipe flops (counted ray-sphere tests x 64 flop: two chained
                      v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays) / the kernel's HIP-event time, `peak` = 2500 TFLOP/s (the
                      guide's dense f16 MFMA peak), `frac` = achieved / peak;          (all-VALU scan: bound "valu_fp32", peak 157.3)
                    `algorithmic`: counted tests x 17 flop (/root/reference/src/hit.jl:13-19) / the same time, and that
                      figure over 157.3 TF (`frac_fp32_vector`, SURVEY 8(d)'s definition; > 1 because pass 1 does not execute the
                      17 flop as vector flops) and over 2500 TF;
                    `issue_model` (a MODEL, labelled so): a SIMD issues either a 32-cycle MFMA or a VALU instruction; 4 MFMA + 1
                      v_alignbit_b32 per 64 tests, the alignbit charged 2 cycles (the guide's FMA class) -> 445.6 algorithmic TF, or
                      4.3 cycles (measured slow class, tools/ubench_rates.hip) -> 322.3 TF;
                    `issue_busy`: (MFMA-busy + 2 x VALU instructions) / SIMD-cycles, and `traffic`: HBM bytes per launch (FETCH_SIZE x 2 +
                      WRITE_SIZE) -- MEASURED IN THIS RUN by three `rocprofv3 --pmc` passes of this script that the default run spawns (one
                      launch each, ~12 s each; `traffic_static` false); when rocprofv3 is missing or a pass fails, the committed figures of
                      profiles/ (`traffic_static` true).  Plus the algorithmic HBM figure.
  value        -- device-resident rate (scene in HBM, image left in HBM), as the task's bench contract prescribes;
                  `value_end_to_end` is SURVEY 8(d)'s metric: the host-buffer entry point rtw_render_* timed the same way (barrier +
                  synchronize around K calls; render + D2H of the image into the caller's buffer; scene upload cached by the library).
  ray_pool     -- null since round 6: the ray-pool kernel (rtw_pool.hpp; same image, 16 - 19 % slower) is a `make POOL=1` build option.
  scan_valu    -- the same workload with RTW_FLAG_SCAN_VALU (the contract discriminant for every
                  sphere on the vector ALUs, the round-1/2 scan): same image, for comparison.
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores on a
                  bounded sample of the same workload (same image, fewer spp), as a CURVE over thread
                  counts (16 / 32 / 64 / all; each count in its own process with the threads pinned:
                  OMP_PLACES=cores OMP_PROC_BIND=close): `cpu_baseline` is the best leg and carries the
                  curve (`thread_curve_Msamples_per_s`) and what the host grants (`host`: affinity, cgroup
                  cpu.max -- the GPU boxes of this pool grant 16 CPUs of time, so 16 threads win);
                  `cpu_baseline_16t` is the north star's comparison point, `cpu_baseline_all_threads` the last leg.
  small_frames -- the launch- / tail-bound regime: BASELINE configs[0] and configs[1] and the small renders the reference publishes
                  (src/proto/proto.jl:64-66, :195-200), per CALL of the C entry points with prebuilt structs: median / min / p90 us of 200
                  host-buffer calls, kernel us, overhead, Msamples/s, fraction of the headline's rate, the device-resident variant,
                  `vs_published` (other hardware: context).  Its rates are repeated inside `config` (`small_frames_Msamples_per_s`),
                  `accelerated.value` and `value_end_to_end` inside `roofline`: the driver's record keeps the VALUES of those two objects.
                  `--small-frames N` runs only this leg.
  accelerated  -- the opt-in RTW_FLAG_GROUP_CULL mode (same image bit for bit), reported separately.
  end_to_end   -- one call of the host-buffer entry point rtw_render_* (scene upload, render,
                  D2H of the image: the PCIe-inclusive rate; never `value`).
  depth16      -- the same workload at the reference's own depth (src/ray_color.jl:14).
  f64_4k       -- configs[4]'s single-GPU share (3840x2160, Float64), with its own roofline and cpu_baseline.
  f64_1080p_d16 -- the reference's PUBLISHED configuration (Float64, 1920x1080, 1000 spp, depth 16: README.md:86,118-119,
                  1.617 Msamples/s on a Ryzen 3700 -- different hardware), with its own roofline, cpu_baseline and `vs_published`.
  N > 1        -- `render_ms_max` / `render_ms_min` (per-rank kernel time, mean over the steps) and `collective_ms` (HIP events on rank 0's
                  stream from the end of ITS render to the end of the one collective, mean over the steps: the collective itself PLUS the wait
                  for the slowest rank's render, i.e. about render_ms_max - render_ms(rank 0) more than the transfer): a bad scaling point can be attributed.
  numerics_legs -- the same workload in the other numerics modes of the ray-sphere test (`--numerics`, include/rtw_hip.h RTW_FLAG_NUMERICS_*):
                  DIFFERENT images by design (Float32: the contract form traces ~4 % fewer segments per sample); each with its segments per sample.
  in_library_devices -- rtw_render_* on host buffers with a device LIST (what a Julia caller gets with `devices=...`): gather by peer copies and
                  by the in-library RCCL reduce, per-device kernel ms, end-to-end ms, gather_path, frame hash.  On a one-GPU box the list repeats
                  ordinal 0 (peer) / holds it once (RCCL, one rank) and the leg says `emulation: true`; `--in-library-devices N` runs only this leg.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 vector 157.3 TFLOP/s (= 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz);
# the guide lists no FP64 vector figure: AMD's MI355X datasheet gives 78.6 TFLOP/s (half the FP32 rate).
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}
MFMA_F16_PEAK_TFLOPS = 2500.0   # same guide: ~2.5 PF dense BF16/FP16 MFMA (2495 TF measured with 32x32x16)
MFMA_FLOP_PER_TEST = 64         # two chained v_mfma_f32_32x32x16_f16 per 32 x 32 tests: 2 x K=16 x 2 flop per test
HBM_PEAK_GBS = 8000.0
FLOP_PER_TEST = 17              # 3 sub + 5 + 5 (dots) + mul/sub + mul/sub, src/hit.jl:13-19
N_SIMD, CLOCK_HZ = 1024, 2.4e9
MFMA_CYCLES_PER_64_TESTS = 4.0  # 4 MFMAs x 32 cycles per block of 32 spheres x 64 rays (2048 tests)
ALIGNBIT_CYCLES = {"alignbit_2_cycles": 2.0,      # one v_alignbit_b32 per test: the guide's FMA-class issue cost ...
                   "alignbit_4p3_cycles": 4.3}    # ... and the slow-class cost measured on this chip (tools/ubench_rates.hip)
TRAFFIC_FILE = os.path.join("profiles", "r06_hbm_traffic.json")
PMC_FILE = os.path.join("profiles", "r06_pmc_summary.json")
PUBLISHED_MSAMPLES = 1.617      # /root/reference/README.md:86,118-119: 1282.44 s for 1920x1080x1000 spp, Float64, depth 16, 16 threads, Ryzen 3700


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the accelerated / scan_valu / ray_pool / end_to_end (value_end_to_end) / depth16 / f64_4k / f64_1080p_d16 legs: ONE kind of launch, for profiling passes")
    ap.add_argument("--group-cull", action="store_true", help="time the opt-in accelerated scan instead of the plain one")
    ap.add_argument("--scan-valu", action="store_true", help="time the all-VALU plain scan (RTW_FLAG_SCAN_VALU) instead of the matrix-pipe filter")
    ap.add_argument("--ray-pool", action="store_true", help="time the opt-in ray-pool kernel (RTW_FLAG_RAY_POOL) instead of the lane-loop kernel")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc passes that measure roofline.traffic / issue_busy in this run")
    ap.add_argument("--chunks", type=int, default=0, help="sample chunks per pixel (0 = library default rule)")
    ap.add_argument("--numerics", choices=["reference", "contract", "reference_fma2"], default="reference",
                    help="the deciding arithmetic of the ray-sphere test (include/rtw_hip.h RTW_FLAG_NUMERICS_*): reference = src/hit.jl:16-18 as the reference "
                         "evaluates it (the default of the library); the default run also times the other two as `numerics_legs`")
    ap.add_argument("--in-library-devices", type=int, default=0, metavar="N",
                    help="time ONLY the in-library device list (what a Julia caller gets with devices=...): rtw_render_* on host buffers with N devices, "
                         "gathered by peer copies and by the in-library RCCL reduce; N > the visible devices: ordinals repeat (emulation, labelled so)")
    ap.add_argument("--small-frames", type=int, default=0, metavar="CALLS",
                    help="time ONLY the small-frame workloads (BASELINE configs[0], configs[1] and the reference's published small renders), CALLS calls each")
    ap.add_argument("--emulate-shard-of", type=int, default=0,
                    help="analysis only: on ONE GPU render shard 0 of N (what each rank of an N-GPU run does)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU-baseline sample duration per leg (a leg per thread count: 16, 32, 64, all)")
    ap.add_argument("--collective", choices=["reduce", "gather"], default="reduce",
                    help="N > 1: reduce = sum of zero-padded full frames onto rank 0 (BASELINE configs[3]); "
                         "gather = each rank sends only its compact tile-major shard (1/N of a frame)")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.emulate_shard_of > 1 and (args.collective == "gather" or args.gpus > 1):
        ap.error("--emulate-shard-of is a one-rank, reduce-layout analysis mode")
    return args


def self_spawn(args):
    """`python bench.py --gpus N` from a plain shell: start the N ranks under torch.distributed.run."""
    import torch
    one_device = os.environ.get("RTW_BENCH_ONE_DEVICE") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (no CPU fallback)")
    if have < args.gpus and not one_device:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible -- refusing to time fewer GPUs than asked for "
                         f"(RTW_BENCH_ONE_DEVICE=1 emulates the {args.gpus}-rank control flow on one device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, RTW_BENCH_SELF_SPAWNED="1")
    raise SystemExit(subprocess.call(cmd, env=env))


class Ctx:
    pass


def setup(args):
    import numpy as np
    import torch
    import torch.distributed as dist
    import rtw_amd as R

    c = Ctx()
    c.np, c.torch, c.dist, c.R, c.args = np, torch, dist, R, args
    c.rank = int(os.environ.get("RANK", "0"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    if c.world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={c.world}: the launcher must start exactly --gpus ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    c.one_device = os.environ.get("RTW_BENCH_ONE_DEVICE") == "1"
    if c.one_device:
        c.local_rank = 0
    elif c.local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {c.rank}: local rank {c.local_rank} has no device ({torch.cuda.device_count()} visible); "
                         f"--gpus {args.gpus} needs {args.gpus} devices")
    torch.cuda.set_device(c.local_rank)
    c.dev = torch.device("cuda", c.local_rank)
    c.backend = None
    if c.world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        c.backend = "gloo" if c.one_device else "nccl"              # "nccl" is RCCL on ROCm
        if c.one_device:
            dist.init_process_group("gloo", rank=c.rank, world_size=c.world)
        else:
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world, device_id=c.dev)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    c.stream = torch.cuda.current_stream(c.dev)
    return c


def fence(c):
    if c.world > 1:
        c.dist.barrier()
    c.torch.cuda.synchronize(c.dev)


class Workload:
    """scene_random_spheres / t_cam1 at one precision and size, resident on this rank's device."""

    def __init__(self, c, dtype, W, spp, depth):
        np, torch, R = c.np, c.torch, c.R
        self.c, self.dtype, self.W, self.spp, self.depth = c, dtype, W, spp, depth
        self.T = np.float64 if dtype == "f64" else np.float32
        self.tT = torch.float64 if dtype == "f64" else torch.float32
        self.jl = "Float64" if dtype == "f64" else "Float32"
        self.H = R.image_height(W)
        R.reseed()                                              # src/proto/proto.jl:198-199
        self.scene = R.scene_random_spheres(elem_type=self.T)
        self.cam = R.t_cam1(elem_type=self.T)
        self.n_spheres = len(self.scene)
        self.renderer = R.DeviceRenderer(self.scene, self.cam, device=c.local_rank)
        n = self.H * W * 3
        if c.args.collective == "gather":                       # compact shards of ragged frames can exceed H*W*3 (whole tiles)
            n = max(n, R.compact_elems(W, 0, c.world))
        self.fb = torch.empty(n, dtype=self.tT, device=c.dev)
        self.frame = None

    def close(self):
        self.renderer.close()
        self.fb = None

    def timed(self, n_steps, n_warm, *, cull, depth, record=None, valu=False, coll_ms=None, pool=None, numerics=None):
        """n_warm untimed + n_steps timed steps bracketed by barrier + synchronize; returns max-over-ranks seconds.
        `coll_ms` (a list): per timed step, the HIP-event time between the end of this rank's render and the end of the collective."""
        c, a = self.c, self.c.args
        torch = c.torch
        evs = []
        pool = a.ray_pool if pool is None else pool

        def step(rec, timed_step):
            def shard(idx, cnt):
                if a.emulate_shard_of > 1:
                    idx, cnt = 0, a.emulate_shard_of
                self.renderer.render_into(self.fb.data_ptr(), self.W, self.spp, depth=depth, seed=1, n_chunks=a.chunks, shard_index=idx,
                                          shard_count=cnt, stream=c.stream.cuda_stream, group_cull=cull,
                                          compact=a.collective == "gather", scan_valu=valu and not cull, n_elems=self.fb.numel(),
                                          ray_pool=pool and not cull and not valu, numerics=numerics or a.numerics)
                return self.fb
            hook = None
            if timed_step and coll_ms is not None and c.world > 1:
                e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                hook = lambda: e1.record(c.stream)
            self.frame = c.R.render_sharded(shard, self.W, mode=a.collective, after_render=hook)    # this rank's tiles, ONE collective onto rank 0
            if hook is not None:
                e2.record(c.stream)
                evs.append((e1, e2))
            if rec is not None:
                rec.append(self.renderer.stats())               # waits for this rank's kernel (HIP events on `stream`)
        for _ in range(n_warm):
            step(None, False)
        fence(c)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(record, True)
        fence(c)
        dt = time.perf_counter() - t0
        if c.world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=c.dev)
            c.dist.all_reduce(tmax, op=c.dist.ReduceOp.MAX)
            dt = float(tmax.item())
        if coll_ms is not None:
            coll_ms.extend(e1.elapsed_time(e2) for e1, e2 in evs)
        return dt

    def timed_host(self, n_steps, n_warm, *, cull, depth, valu=False):
        """SURVEY 8(d)'s metric: K calls of the host-buffer entry point (what the Julia ccall binds: render + D2H of the image into
        the caller's buffer, blocking; the scene upload is cached by the library between calls), barrier + synchronize around them."""
        c, a = self.c, self.c.args
        k_ms = []

        def call():
            c.R.render(self.scene, self.cam, self.W, self.spp, depth=depth, seed=1, n_chunks=a.chunks, device=c.local_rank,
                       group_cull=cull, scan_valu=valu and not cull, ray_pool=a.ray_pool and not cull and not valu, numerics=a.numerics)
            k_ms.append(c.R.last_stats()["kernel_ms"])
        t = time.perf_counter()
        call()                                                  # the first call of a scene builds the library's per-device context
        first_ms = (time.perf_counter() - t) * 1e3
        for _ in range(max(0, n_warm - 1)):
            call()
        fence(c)
        t0 = time.perf_counter()
        for _ in range(n_steps):
            call()
        fence(c)
        return time.perf_counter() - t0, first_ms, k_ms[-n_steps:]

    def frame_sha256(self):
        """SHA-256 of the assembled H x W x 3 frame of the last step (rank 0; outside every timed region)."""
        n = self.H * self.W * 3
        host = self.frame.reshape(-1)[:n].cpu().numpy()
        return hashlib.sha256(host.tobytes()).hexdigest()


def issue_busy_from_pmc(key):
    """(MFMA-busy cycles + 2 x VALU instructions) / SIMD-cycles of the committed PMC pass of this workload (static)."""
    try:
        pm = json.load(open(os.path.join(ROOT, PMC_FILE)))
        a, m = pm["pmc_%s_sqA" % key]["counters"], pm["pmc_%s_mfma" % key]["counters"]
        simd = N_SIMD * a["GRBM_GUI_ACTIVE"] / 8
        return {"mfma_busy": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / simd, 4), "valu_x2": round(2 * a["SQ_INSTS_VALU"] / simd, 4),
                "sum": round((m["SQ_VALU_MFMA_BUSY_CYCLES"] + 2 * a["SQ_INSTS_VALU"]) / simd, 4),
                "source": PMC_FILE + " (rocprofv3 --pmc passes of this command, one launch each; static: not measured in this run)"}
    except Exception:
        return None


def roofline_of(wl, k_s, tests_per_launch, seg_per_sample, *, cull, valu, world, shard_div):
    """Fixed yard-sticks (round 4): the guide's dense f16 MFMA peak for the executed matrix-pipe flops, the FP32 vector peak of SURVEY 8(d)
    for the algorithmic 17 flop per test; the issue model is reported as a model, with both costs of the alignbit."""
    W, H, spp, depth = wl.W, wl.H, wl.spp, wl.depth
    algorithmic = tests_per_launch * FLOP_PER_TEST / k_s / 1e12
    mfma_tflops = tests_per_launch * MFMA_FLOP_PER_TEST / k_s / 1e12
    esize = 8 if wl.dtype == "f64" else 4
    alg_bytes = W * H * 3 * esize / world / shard_div + wl.n_spheres * 12 * esize   # framebuffer write + one scene read
    # HBM bytes per launch from the PMC passes kept under profiles/ (separate --pmc runs of this same command): static,
    # valid for the exact workload they were taken on -- rocprofv3 counters cannot be read from inside this process
    traffic = traffic_src = None
    mode = "cull" if cull else ("valu" if valu else "plain")
    try:
        tr = json.load(open(os.path.join(ROOT, TRAFFIC_FILE)))
        key = f"{wl.dtype}_{W}x{H}_{spp}spp_d{depth}_{mode}"
        if world == 1 and shard_div == 1 and key in tr:
            traffic, traffic_src = tr[key]["hbm_bytes_per_launch"], tr[key].get("source")
    except Exception:
        pass
    matrix = not (cull or valu)
    vpeak = VALU_PEAK_TFLOPS[wl.dtype]
    issue = None
    if matrix:
        issue = {"model": True, "what": "a SIMD issues EITHER a 32-cycle v_mfma_f32_32x32x16_f16 or a VALU instruction (no overlap on gfx950: tools/ubench_mfma_pipe.hip, and in the kernel itself "
                                        "profiles/r05_probe_phases.txt: every MFMA executed twice / three times costs +38 % / +80 %, the kernel without them runs 19 % faster per wave-segment); "
                                        "per 64 tests 4 MFMA cycles + one v_alignbit_b32"}
        for name, cyc in ALIGNBIT_CYCLES.items():
            pk = N_SIMD * CLOCK_HZ / (MFMA_CYCLES_PER_64_TESTS + cyc) * 64 * FLOP_PER_TEST / 1e12
            issue[name] = {"peak_algorithmic_TFLOPs": round(pk, 1), "frac": round(algorithmic / pk, 4)}
    headline = (wl.dtype, W, spp, depth) == ("f32", 1920, 1000, 50) and world == 1 and shard_div == 1
    if cull:
        bound, achieved, peak = "mfma", None, MFMA_F16_PEAK_TFLOPS
    elif valu:
        bound, achieved, peak = "valu_" + ("fp64" if wl.dtype == "f64" else "fp32"), algorithmic, vpeak
    else:
        bound, achieved, peak = "mfma", mfma_tflops, MFMA_F16_PEAK_TFLOPS
    return {
        "bound": bound,
        "kernel": f"rtw::{'trace_pool_kernel' if (wl.c.args.ray_pool and matrix and wl.dtype == 'f32') else 'trace_kernel'}<{'double' if wl.dtype == 'f64' else 'float'}>",
        "achieved": None if achieved is None else round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
        "frac": None if achieved is None else round(achieved / peak, 4),
        "definition": None if not matrix else "achieved = EXECUTED f16 matrix-pipe flops (counted ray-sphere tests x 64: two chained v_mfma_f32_32x32x16_f16 per 32 spheres x 32 rays) / "
                                              "kernel time; peak = dense f16 MFMA peak of MI355X_MICROARCH.md (fixed since round 4)",
        "algorithmic": None if cull else {"flop_per_test": FLOP_PER_TEST, "achieved": round(algorithmic, 3), "unit": "TFLOP/s",
                                          "frac_fp32_vector": round(algorithmic / VALU_PEAK_TFLOPS["f32"], 4), "fp32_vector_peak": VALU_PEAK_TFLOPS["f32"],
                                          "frac_mfma_f16_peak": round(algorithmic / MFMA_F16_PEAK_TFLOPS, 4),
                                          "note": "SURVEY 8(d): counted tests x 17 flop / kernel time; > 1 of the FP32 vector peak because pass 1 (a conservative "
                                                  "bilinear filter on the matrix pipe) does not execute the 17 flop as vector flops -- the exact 17-flop test runs on its candidates only"},
        "issue_model": issue,
        "issue_busy": issue_busy_from_pmc("f32") if (matrix and headline) else None,
        "traffic": traffic, "traffic_static": True,
        "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes of this command; read from " + TRAFFIC_FILE + ")",
        "traffic_source": traffic_src,
        "kernel_ms": round(k_s * 1e3, 3), "tests_per_launch": int(tests_per_launch), "segments_per_sample": round(seg_per_sample, 4),
        "note": "Every sphere is tested against every ray segment, but pass 1 of the scan is a conservative FILTER: D = (d.c)^2 + 2 p.c + (q^2 - |o|^2) + "
                "(r^2 - |c|^2) is bilinear in (ray features) x (sphere features), so one K = 32 contraction over f16-split features evaluates 32 spheres x 32 "
                "rays and the VALU adds one alignbit per test (rigorous margin, DESIGN.md 6.1); the exact contract arithmetic (FP32 or FP64) runs only on "
                "the filter's candidates.  `scan_valu` is the same workload with every test on the vector ALUs."
                if matrix else
                ("all-VALU plain scan (RTW_FLAG_SCAN_VALU): 11 instructions per test (Float32) / 13 binary32 filter instructions (Float64); "
                 "achieved = counted tests x 17 flop / kernel time"
                 if valu else "group-cull mode: tests are skipped, no roofline fraction"),
        "hbm": {"algorithmic_bytes": int(alg_bytes), "achieved_GBs": round(alg_bytes / k_s / 1e9, 4),
                "peak_GBs": HBM_PEAK_GBS, "frac": round(alg_bytes / k_s / 1e9 / HBM_PEAK_GBS, 8)},
    }


def live_pmc(dtype, width, spp, depth, numerics="reference", timeout_s=90):
    """HBM traffic and issue-busy of ONE launch of this workload, measured in this run: three `rocprofv3 --pmc` passes of this script
    itself (`--steps 1 --warmup 0 --no-extras`: exactly one trace-kernel launch each; FETCH_SIZE and WRITE_SIZE in separate passes, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled: its gfx950 correction; KiB units).  Returns None when rocprofv3 is not there, a
    pass fails or times out -- the line then carries the committed figures of profiles/ (marked static)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    base = tempfile.mkdtemp(prefix="rtw_pmc_", dir="/tmp")
    cmd = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras", "--dtype", dtype,
           "--width", str(width), "--spp", str(spp), "--depth", str(depth), "--numerics", numerics]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = {}
    try:
        for tag, counters in (("fetch", ["GRBM_GUI_ACTIVE", "FETCH_SIZE"]), ("write", ["GRBM_GUI_ACTIVE", "WRITE_SIZE"]),
                              ("issue", ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_MFMA"])):
            d = os.path.join(base, tag)
            r = subprocess.run([exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd,
                               cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
            if r.returncode != 0:
                return None
            rows = [row for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True) for row in csv.DictReader(open(f))
                    if "trace_" in row["Kernel_Name"]]
            if not rows:
                return None
            first = min(int(row["Dispatch_Id"]) for row in rows)
            if len(set(row["Dispatch_Id"] for row in rows)) != 1:
                return None                                     # (must be ONE launch: per-launch figures)
            for row in rows:
                if int(row["Dispatch_Id"]) == first:
                    out[(tag, row["Counter_Name"])] = out.get((tag, row["Counter_Name"]), 0.0) + float(row["Counter_Value"])
        fetch, write = out[("fetch", "FETCH_SIZE")] * 1024, out[("write", "WRITE_SIZE")] * 1024
        simd = N_SIMD * out[("issue", "GRBM_GUI_ACTIVE")] / 8
        return {"traffic": int(2 * fetch + write), "fetch_size_bytes_raw": int(fetch), "write_size_bytes": int(write),
                "issue_busy": {"mfma_busy": round(out[("issue", "SQ_VALU_MFMA_BUSY_CYCLES")] / simd, 4),
                               "valu_x2": round(2 * out[("issue", "SQ_INSTS_VALU")] / simd, 4),
                               "sum": round((out[("issue", "SQ_VALU_MFMA_BUSY_CYCLES")] + 2 * out[("issue", "SQ_INSTS_VALU")]) / simd, 4),
                               "valu_instructions": int(out[("issue", "SQ_INSTS_VALU")]), "mfma_instructions": int(out[("issue", "SQ_INSTS_MFMA")]),
                               "source": "rocprofv3 --pmc pass of this command spawned by this run (one launch)"}}
    except Exception:
        return None
    finally:
        shutil.rmtree(base, ignore_errors=True)


def host_cpu_info():
    """what this process may use of the host: affinity mask, cgroup CPU quota, logical CPUs"""
    info = {"logical_cpus": os.cpu_count(), "affinity_cpus": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_cpu_max": None}
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            info["cgroup_cpu_max"] = open(path).read().strip()
            break
        except OSError:
            pass
    try:
        info["model"] = next(ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name"))
    except Exception:
        info["model"] = None
    info["cgroup_cpus_granted"] = None                            # CFS quota / period: the CPU TIME this container may use, in CPUs
    try:
        q, per = (info["cgroup_cpu_max"] or "").split()[:2]
        if q != "max":
            info["cgroup_cpus_granted"] = round(float(q) / float(per), 2)
    except Exception:
        pass
    return info


def cpu_leg_child(argv):
    """`bench.py --cpu-leg dtype width depth numerics threads seconds`: ONE timing of the CPU oracle in a fresh process whose OpenMP runtime was
    started with the binding of the environment (OMP_PLACES / OMP_PROC_BIND are read when libgomp initialises: not changeable in the parent,
    where torch has loaded it long ago).  No torch, no GPU.  Prints one JSON line."""
    import numpy as np
    import rtw_amd as R
    import rtw_oracle as O
    dtype, W, depth, numerics, nthr, seconds = argv[0], int(argv[1]), int(argv[2]), argv[3], int(argv[4]), float(argv[5])
    O.build()
    T = np.float64 if dtype == "f64" else np.float32
    H = R.image_height(W)
    R.reseed()
    scene, cam = R.scene_random_spheres(elem_type=T), R.t_cam1(elem_type=T)
    flat = R.flatten_scene(scene, T)
    t = time.perf_counter()
    O.render(flat, cam, W, H, 1, T=T, max_depth=depth, seed=1, n_chunks=1, omp_threads=nthr, numerics=numerics)
    t1 = time.perf_counter() - t
    s_spp = int(max(1, min(64, round(seconds / max(t1, 1e-3)))))
    t = time.perf_counter()
    O.render(flat, cam, W, H, s_spp, T=T, max_depth=depth, seed=1, omp_threads=nthr, numerics=numerics)
    tc = time.perf_counter() - t
    print(json.dumps({"value": round(W * H * s_spp / tc / 1e6, 4), "spp": s_spp, "seconds": round(tc, 2), "threads": nthr}), flush=True)


def cpu_legs(wl, seconds, thread_counts=None):
    """The CPU oracle (kind "port"; the Julia reference cannot run here) on this box's host cores, bounded sample, as a CURVE over thread
    counts -- 16 (the north star's comparison point), 32, 64 and every CPU this process may use -- each leg in its own process with the
    threads pinned (OMP_PLACES=cores OMP_PROC_BIND=close, passive waiting).  Returns (legs, the 16-thread leg or None)."""
    host = host_cpu_info()
    avail = host["affinity_cpus"] or host["logical_cpus"] or 1
    counts = thread_counts or sorted({n for n in (16, 32, 64, avail) if n <= avail} or {avail})
    legs = []
    for nthr in counts:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        env.update(OMP_NUM_THREADS=str(nthr), OMP_PLACES="cores", OMP_PROC_BIND="close", OMP_WAIT_POLICY="passive", OMP_DYNAMIC="false")
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", wl.dtype, str(wl.W), str(wl.depth), wl.c.args.numerics, str(nthr), str(seconds)]
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=max(120.0, 12 * seconds))
            d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        except Exception as e:
            legs.append({"value": 0.0, "unit": "Msamples/s", "cores": nthr, "kind": "port", "sample": f"FAILED: {str(e)[:200]}"})
            continue
        legs.append({"value": d["value"], "unit": "Msamples/s", "cores": nthr, "kind": "port",
                     "sample": f"same scene/camera/{wl.W}x{wl.H}/depth {wl.depth}/{wl.jl}, {d['spp']} spp ({d['seconds']:.1f} s), oracle/ C port, OpenMP schedule(dynamic, 64) over "
                               f"pixels, {nthr} threads pinned (OMP_PLACES=cores OMP_PROC_BIND=close, own process); the Julia reference cannot run here (no julia in the image)"})
    curve = {str(l["cores"]): l["value"] for l in legs}
    best = max(legs, key=lambda l: l["value"])
    for l in legs:
        l["thread_curve_Msamples_per_s"] = curve
        l["host"] = host
    if len(legs) > 1 and legs[-1]["value"] < 0.9 * best["value"]:
        granted = host.get("cgroup_cpus_granted")
        best["sample"] += (f"; more threads are SLOWER here ({curve}): the process sees {avail} logical CPUs of a shared host, but its cgroup grants it "
                           + (f"the CPU time of {granted:g} CPUs (cpu.max = {host['cgroup_cpu_max']}): threads beyond that are throttled by the CFS quota and only add scheduling overhead"
                              if granted else f"a limited share (cpu.max: {host['cgroup_cpu_max']}): more threads share cores / are throttled"))
    return legs, next((l for l in legs if l["cores"] == 16), None)


SMALL_FRAMES = [
    # (key, scene builder, camera, width, spp, depth, dtype, what, published ms on the reference's Ryzen 3700 / 16 threads or None)
    ("cfg1_random_320x180_64spp_d16_f32", "scene_random_spheres", "t_cam1", 320, 64, 16, "f32",
     "BASELINE.json configs[1]: scene_random_spheres, 320x180, 64 spp, depth 16, Float32", None),
    ("proto_random_200x112_32spp_d16_f64", "scene_random_spheres", "t_cam1", 200, 32, 16, "f64",
     "render(scene_random_spheres, t_cam1, 200, 32), Float64, depth 16 (/root/reference/src/proto/proto.jl:195-200: 296.824 ms)", 296.824),
    ("proto_2spheres_96x54_16spp_d16_f64", "scene_2_spheres", "t_default_cam", 96, 16, 16, "f64",
     "render(scene_2_spheres, t_default_cam, 96, 16), Float64, depth 16 (/root/reference/src/proto/proto.jl:64-66: 951.447 us)", 0.951447),
    ("cfg0_2spheres_96x54_16spp_d4_f32", "scene_2_spheres", "t_default_cam", 96, 16, 4, "f32",
     "BASELINE.json configs[0]: scene_2_spheres, 96x54, 16 spp, depth 4, Float32 (the reference's CPU-runnable case)", None),
]


def small_frames_leg(c, headline_rate, calls=200, cull_too=True):
    """The launch- / tail-bound regime: BASELINE configs[1] and the small workloads the reference itself publishes.
    Per workload, the C entry points are called DIRECTLY (prebuilt ctypes structs: what a `ccall` costs, no Python scene flattening in
    the timed region):
      host   rtw_render_* on host buffers (scene cached by the library; render + D2H into the caller's buffer; blocking), `calls` calls,
             each timed by itself with perf_counter_ns -> median / min / p90 in us, kernel us (HIP events of the same calls, median),
             overhead = median call - median kernel;
      device rtw_render_device_* into a device buffer + stream synchronize (the device-resident rate, like `value`)."""
    import ctypes as C
    import statistics
    np, torch, R = c.np, c.torch, c.R
    from rtw_amd import _capi
    L = _capi.lib()
    out = {"calls": calls, "headline_Msamples_per_s": round(headline_rate, 2),
           "how": "direct C-ABI calls with prebuilt structs; per-call perf_counter_ns; kernel = HIP events around the launch (rtw_stats)"}
    for key, scene_fn, cam_fn, W, spp, depth, dtype, what, pub_ms in SMALL_FRAMES:
        T = np.float64 if dtype == "f64" else np.float32
        R.reseed()
        scene = getattr(R, scene_fn)(elem_type=T)
        cam = getattr(R, cam_fn)(elem_type=T)
        H = R.image_height(W)
        flat = R.flatten_scene(scene, T)
        S, keep = _capi.make_scene(flat, T)
        Cm = _capi.make_camera(cam, T)
        host = np.empty(H * W * 3, dtype=T)
        fn = L.rtw_render_f64 if dtype == "f64" else L.rtw_render_f32
        fnd = L.rtw_render_device_f64 if dtype == "f64" else L.rtw_render_device_f32
        st = _capi.Stats()
        samples = W * H * spp
        entry = {"workload": what, "samples": samples}
        modes = [("plain", 0)] + ([("group_cull", _capi.FLAG_GROUP_CULL)] if (cull_too and scene_fn == "scene_random_spheres") else [])
        for mode, flags in modes:
            P = _capi.make_params(W, H, spp, depth, 1, 0, 0, 1, c.local_rank, 1, flags, numerics=c.args.numerics)
            outp = host.ctypes.data_as(C.c_void_p)
            for _ in range(5):
                _capi.check(fn(C.byref(S), C.byref(Cm), C.byref(P), outp))
            t_us, k_us = [], []
            for _ in range(calls):
                t0 = time.perf_counter_ns()
                rc = fn(C.byref(S), C.byref(Cm), C.byref(P), outp)
                t1 = time.perf_counter_ns()
                _capi.check(rc)
                _capi.check(L.rtw_stats(C.byref(st)))
                t_us.append((t1 - t0) / 1e3)
                k_us.append(st.kernel_ms * 1e3)
            t_us.sort()
            med, kmed = statistics.median(t_us), statistics.median(k_us)
            # device-resident: the scene handle is the caller's, the image stays in HBM
            handle = C.c_void_p()
            up = L.rtw_scene_upload_f64 if dtype == "f64" else L.rtw_scene_upload_f32
            _capi.check(up(C.byref(S), c.local_rank, C.byref(handle)))
            fb = torch.empty(H * W * 3, dtype=torch.float64 if dtype == "f64" else torch.float32, device=c.dev)
            Pd = _capi.make_params(W, H, spp, depth, 1, 0, 0, 1, -1, 1, flags, numerics=c.args.numerics)
            d_us = []
            for k in range(calls + 5):
                t0 = time.perf_counter_ns()
                rc = fnd(handle, C.byref(Cm), C.byref(Pd), C.c_void_p(fb.data_ptr()), C.c_void_p(c.stream.cuda_stream))
                c.stream.synchronize()
                t1 = time.perf_counter_ns()
                _capi.check(rc)
                if k >= 5:
                    d_us.append((t1 - t0) / 1e3)
            L.rtw_scene_free(handle)
            dmed = statistics.median(d_us)
            e = {"call_us_median": round(med, 1), "call_us_min": round(t_us[0], 1), "call_us_p90": round(t_us[int(0.9 * (len(t_us) - 1))], 1),
                 "kernel_us_median": round(kmed, 1), "overhead_us": round(med - kmed, 1),
                 "Msamples_per_s": round(samples / med, 2), "kernel_only_Msamples_per_s": round(samples / max(kmed, 1e-3), 2),
                 "device_resident_us_median": round(dmed, 1), "device_resident_Msamples_per_s": round(samples / dmed, 2),
                 "frac_of_headline": round(samples / med / headline_rate, 4), "grid_blocks": st.grid_blocks, "n_chunks": st.n_chunks,
                 "segments_per_sample": round(st.segments / max(1, st.samples), 4)}
            if pub_ms is not None and mode == "plain":
                e["vs_published"] = {"published_ms": pub_ms, "ratio": round(pub_ms * 1e3 / med, 1),
                                     "published_on": "Ryzen 3700 (8C/16T), julia -t 16 -- DIFFERENT HARDWARE, context only"}
            entry[mode] = e
            fb = None
        del keep
        out[key] = entry
    return out


def in_library_leg(c, wl, n_devices, numerics, steps=2):
    """The in-library device list (rtw_params.n_devices / device_ids: `render(...; devices=...)` of the Julia shim): K host-buffer calls per
    gather path, timed like `value_end_to_end`.  `n_devices` beyond the visible devices: ordinals repeat -- the shards then share a device
    (same code path up to the copy: rtw_stats_t.gather_path says SAME_DEVICE) and the RCCL variant runs on the distinct ordinals only."""
    R, torch = c.R, c.torch
    have = torch.cuda.device_count()
    ids = [k % have for k in range(n_devices)]
    emulation = n_devices > have
    out = {"n_devices": n_devices, "visible_devices": have, "emulation": emulation,
           "test_aids_env": {k: os.environ[k] for k in ("RTW_ENABLE_TEST_AIDS", "RTW_DEBUG_REMOTE_SHARDS", "RTW_DEBUG_NO_PEER") if k in os.environ} or None,
           "workload": f"{wl.W}x{wl.H}, {wl.spp} spp, depth {wl.depth}, {wl.jl}, numerics {numerics}"}
    names = {1: "peer", 2: "host_staged", 4: "rccl_reduce", 8: "same_device"}
    sha_of = lambda im: hashlib.sha256(c.np.ascontiguousarray(im.transpose(1, 0, 2)).tobytes()).hexdigest()     # the column-major Matrix{RGB{T}} bytes, like Workload.frame_sha256
    # (librccl prints a version banner on stdout when its first communicator is made: this process's stdout carries ONE JSON line, so
    #  file descriptor 1 points at stderr while the leg runs)
    sys.stdout.flush()
    saved_fd = os.dup(1)
    os.dup2(2, 1)
    for key, kw in (("peer", dict(devices=ids)), ("rccl_reduce", dict(devices=sorted(set(ids)), rccl_reduce=True))):
        try:
            def call():
                img = R.render(wl.scene, wl.cam, wl.W, wl.spp, depth=wl.depth, seed=1, n_chunks=c.args.chunks, numerics=numerics, **kw)
                return img, R.last_stats()
            t = time.perf_counter()
            img, st = call()                                     # first call: contexts, scene uploads, peer access / communicators
            first_ms = (time.perf_counter() - t) * 1e3
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                img, st = call()
            dt = (time.perf_counter() - t0) / steps
            out[key] = {"devices": kw["devices"], "value": round(wl.W * wl.H * wl.spp / dt / 1e6, 2), "unit": "Msamples/s", "ms": round(dt * 1e3, 3),
                        "first_call_ms": round(first_ms, 3), "steps": steps, "kernel_ms_max": round(st["kernel_ms"], 3),
                        "per_device_kernel_ms": [{"device": d, "kernel_ms": round(ms, 3)} for d, ms in st["per_device"]],
                        "gather_path": st["gather_path"], "gather_path_names": [v for k, v in names.items() if st["gather_path"] & k],
                        "segments": st["segments"], "frame_sha256": sha_of(img)}
        except Exception as e:                                   # (e.g. librccl missing: the leg says so instead of killing the line)
            out[key] = {"error": str(e)[:300]}
    out["sha_of"] = sha_of
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)                           # (the banner sits in C stdio's buffer: out with it while fd 1 is still stderr)
    except Exception:
        pass
    os.dup2(saved_fd, 1)
    os.close(saved_fd)
    return out


def cfg_name(dtype, W, spp, depth, world):
    return {("f32", 1920, 1000, 50): "BASELINE.json configs[2]" if world == 1 else "BASELINE.json configs[3]",
            ("f64", 3840, 1000, 50): "BASELINE.json configs[4]" + (", one GPU" if world == 1 else "")}.get((dtype, W, spp, depth), "not a BASELINE config")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-leg":
        return cpu_leg_child(sys.argv[2:])
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_spawn(args)                                        # does not return
    c = setup(args)
    np, torch, dist, R = c.np, c.torch, c.dist, c.R
    rank, world = c.rank, c.world

    if args.small_frames > 0:                                   # this leg only (tools/gpu_small_frames.sh; profiling passes)
        if world != 1:
            raise SystemExit("--small-frames is a one-process mode")
        print(json.dumps({"small_frames": small_frames_leg(c, 0.0 + float(os.environ.get("RTW_HEADLINE_RATE", "5600")), calls=args.small_frames)}), flush=True)
        return
    wl = Workload(c, args.dtype, args.width, args.spp, args.depth)
    W, H, spp, depth = wl.W, wl.H, wl.spp, wl.depth
    if args.in_library_devices > 0:                             # this leg only (tools/gpu_multi.sh; one process, the library drives the devices)
        if world != 1:
            raise SystemExit("--in-library-devices is a one-process mode (the library itself uses the devices)")
        leg = in_library_leg(c, wl, args.in_library_devices, args.numerics, steps=max(1, args.steps))
        one = c.R.render(wl.scene, wl.cam, W, spp, depth=depth, seed=1, n_chunks=args.chunks, numerics=args.numerics, device=0)
        leg["one_device_frame_sha256"] = leg.pop("sha_of")(one)
        leg["one_device_kernel_ms"] = round(c.R.last_stats()["kernel_ms"], 3)
        for k in ("peer", "rccl_reduce"):
            if "frame_sha256" in leg.get(k, {}):
                leg[k]["frame_sha256_equal"] = leg[k]["frame_sha256"] == leg["one_device_frame_sha256"]
        print(json.dumps({"in_library_devices": leg}), flush=True)
        wl.close()
        return
    stats, coll = [], []
    dt = wl.timed(args.steps, args.warmup, cull=args.group_cull, depth=depth, record=stats, valu=args.scan_valu, coll_ms=coll)
    sha = wl.frame_sha256() if rank == 0 else None
    kernel_ms = [s["kernel_ms"] for s in stats]
    tests = [s["sphere_tests"] for s in stats]
    segments = [s["segments"] for s in stats]
    per_rank = None
    if world > 1:
        agg = torch.tensor([sum(tests), sum(segments)], dtype=torch.float64, device=c.dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        all_segments = float(agg[1])
        mine = torch.tensor([sum(kernel_ms) / len(kernel_ms), float(c.local_rank), sum(coll) / max(1, len(coll))], dtype=torch.float64, device=c.dev)
        allk = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allk, mine)
        per_rank = [{"rank": r, "device": int(t[1].item()), "kernel_ms": round(float(t[0].item()), 3), "collective_ms": round(float(t[2].item()), 3)}
                    for r, t in enumerate(allk)]
    else:
        all_segments = float(sum(segments))

    shard_div = args.emulate_shard_of if args.emulate_shard_of > 1 else 1
    samples_per_step = W * H * spp / shard_div
    value = samples_per_step * args.steps / dt / 1e6

    extras = not args.no_extras and args.emulate_shard_of <= 1
    accel = depth16 = scan_valu = end_to_end = f64_4k = f64_pub = None
    if extras and not args.group_cull:
        # the opt-in accelerated scan (RTW_FLAG_GROUP_CULL, bit-identical image), timed the same way, reported
        # separately: `value` stays the reference's plain linear scan so that the roofline figure means what it says
        dta = wl.timed(args.steps, 1, cull=True, depth=depth)
        accel = {"mode": "RTW_FLAG_GROUP_CULL (kd-sorted blocks of 32 spheres skipped when no ray of the wave can touch the block's grown box, in front of the matrix-pipe filter; same image bit for bit)",
                 "value": round(samples_per_step * args.steps / dta / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(dta / args.steps * 1e3, 3), "frame_sha256_equal": (wl.frame_sha256() == sha) if rank == 0 else None}
    if extras and not args.group_cull and not args.scan_valu:
        nv = max(1, min(args.steps, 2))
        dtv = wl.timed(nv, 1, cull=False, depth=depth, valu=True) / nv
        scan_valu = {"mode": "RTW_FLAG_SCAN_VALU (the discriminant of the numerics mode for every sphere on the vector ALUs: no filter, no margin constants; same image bit for bit)",
                     "value": round(samples_per_step / dtv / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dtv * 1e3, 3), "steps": nv, "warmup": 1,
                     "frame_sha256_equal": (wl.frame_sha256() == sha) if rank == 0 else None}
    ray_pool = None      # (round 6: the ray-pool kernel is a `make POOL=1` build option -- tools/gpu_pool_check.sh times it; `--ray-pool` needs such a build)
    if extras and depth != 16:
        st16 = []
        dt16 = wl.timed(1, 0, cull=args.group_cull, depth=16, record=st16, valu=args.scan_valu)
        depth16 = {"value": round(samples_per_step / dt16 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt16 * 1e3, 3),
                   "segments_per_sample": round(st16[0]["segments"] * world / samples_per_step, 4) if world == 1 else None,
                   "note": "same workload at depth 16, the reference's only depth (src/ray_color.jl:14)"}
    numerics_legs = in_lib = None
    if extras and not (args.group_cull or args.scan_valu or args.ray_pool):
        numerics_legs = {}
        for mode in ("reference", "contract", "reference_fma2"):
            if mode == args.numerics:
                continue
            stn = []
            nn = max(1, min(args.steps, 2))
            dtn = wl.timed(nn, 1, cull=False, depth=depth, record=stn, numerics=mode) / nn
            numerics_legs[mode] = {"value": round(samples_per_step / dtn / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dtn * 1e3, 3), "steps": nn, "warmup": 1,
                                   "segments_per_sample": round(stn[0]["segments"] * world / samples_per_step, 4) if world == 1 else None,
                                   "frame_sha256": wl.frame_sha256() if rank == 0 else None,
                                   "note": "a DIFFERENT image by design: another evaluation order of src/hit.jl:16-18 (DESIGN.md section 4)"}
    if extras and world == 1:
        # the in-library device list in a CHILD process (`bench.py --in-library-devices N`): on a multi-GPU box this is the first time the
        # peer copies / the N-rank RCCL reduce run at all -- whatever happens there (an error, a hang) must not cost the headline line
        have = torch.cuda.device_count()
        cmd = [sys.executable, os.path.abspath(__file__), "--in-library-devices", str(have if have > 1 else 2), "--steps", str(max(1, min(args.steps, 2))),
               "--dtype", args.dtype, "--width", str(W), "--spp", str(spp), "--depth", str(depth), "--numerics", args.numerics, "--chunks", str(args.chunks)]
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            in_lib = json.loads(lines[-1])["in_library_devices"] if (r.returncode == 0 and lines) else {"error": f"exit code {r.returncode}: {r.stderr[-400:]}"}
        except subprocess.TimeoutExpired:
            in_lib = {"error": "timed out after 240 s"}
        except Exception as e:
            in_lib = {"error": str(e)[:300]}
        for k in ("peer", "rccl_reduce"):
            if "frame_sha256" in in_lib.get(k, {}):
                in_lib[k]["frame_sha256_equal"] = in_lib[k]["frame_sha256"] == sha
    if extras and world == 1:
        # SURVEY 8(d)'s metric -- the host-buffer entry point (what the Julia ccall binds): render + D2H of the image into the caller's
        # buffer, timed like `value` (the scene upload is cached by the library: the first call, reported separately, pays it)
        ne = args.steps
        dte, first_ms, k_ms = wl.timed_host(ne, 1, cull=args.group_cull, depth=depth, valu=args.scan_valu)
        end_to_end = {"value": round(W * H * spp * ne / dte / 1e6, 2), "unit": "Msamples/s", "steps": ne, "ms": round(dte / ne * 1e3, 3),
                      "kernel_ms": round(sum(k_ms) / len(k_ms), 3), "overhead_ms": round(dte / ne * 1e3 - sum(k_ms) / len(k_ms), 3),
                      "first_call_ms": round(first_ms, 3),
                      "note": "rtw_render_* on host buffers (SURVEY 8(d)'s definition of the metric): render + D2H of the image, PCIe-inclusive, blocking; "
                              "the scene upload is cached between calls (first_call_ms includes it)"}

    small = None
    if extras and world == 1 and not (args.group_cull or args.scan_valu or args.ray_pool):
        try:
            small = small_frames_leg(c, value)
        except Exception as e:                                   # (never costs the headline line)
            small = {"error": str(e)[:300]}

    line = None
    cpu_counts_short = None
    if rank == 0:
        # the only kernel = trace_kernel.  Per launch (this rank's shard): algorithmic flops =
        # sphere tests x 17; duration = mean HIP-event time on the launch stream.
        k_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        roofline = roofline_of(wl, k_s, sum(tests) / len(tests), all_segments / (samples_per_step * args.steps),
                               cull=args.group_cull, valu=args.scan_valu, world=world, shard_div=shard_div)
        if extras and world == 1 and not (args.group_cull or args.scan_valu or args.ray_pool) and not args.no_live_pmc:
            pm = live_pmc(args.dtype, W, spp, depth, args.numerics)            # ~3 x 12 s; None on any failure (the static figures stay)
            if pm:
                roofline["traffic"], roofline["traffic_static"] = pm["traffic"], False
                roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes spawned by THIS run (one launch each; FETCH_SIZE x 2 + WRITE_SIZE)"
                roofline["traffic_detail"] = {"fetch_size_bytes_raw": pm["fetch_size_bytes_raw"], "write_size_bytes": pm["write_size_bytes"]}
                roofline["issue_busy"] = pm["issue_busy"]
        cpu = cpu16 = None
        legs = []
        if world == 1 and not args.no_cpu_baseline:
            legs, cpu16 = cpu_legs(wl, args.cpu_seconds)
            cpu = max(legs, key=lambda d: d["value"])
            cpu_counts_short = sorted({cpu["cores"]} | ({16} if cpu16 else set()))       # the Float64 legs: 16 threads and the best count of this curve
        line = {
            "metric": f"Msamples/s (pixels x spp) on scene_random_spheres {W}x{H}",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "value_definition": "device-resident: scene in HBM before, image left in HBM (rank 0) after the timed region -- the bench contract of the task; "
                                "SURVEY 8(d)'s end-to-end metric (render + D2H into the caller's buffer) is `value_end_to_end`",
            "value_end_to_end": end_to_end["value"] if end_to_end else None,
            "kernel_only": round(samples_per_step / (sum(kernel_ms) / len(kernel_ms)) / 1e3, 2) if world == 1 else None,
            "render_ms_max": max(r["kernel_ms"] for r in per_rank) if per_rank else round(sum(kernel_ms) / len(kernel_ms), 3),
            "render_ms_min": min(r["kernel_ms"] for r in per_rank) if per_rank else round(sum(kernel_ms) / len(kernel_ms), 3),
            "collective_ms": round(per_rank[0]["collective_ms"], 3) if per_rank else 0.0,
            "collective_ms_max": max(r["collective_ms"] for r in per_rank) if per_rank else 0.0,
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"scene_random_spheres ({wl.n_spheres} spheres, reseed!() seed 1), t_cam1, {W}x{H}, {spp} spp, "
                                   f"depth {depth}, {wl.jl} ({cfg_name(args.dtype, W, spp, depth, world)})",
                       "scan": "group_cull (opt-in)" if args.group_cull else
                               ("plain linear scan over all spheres, all on the VALU (RTW_FLAG_SCAN_VALU)" if args.scan_valu else
                                "plain linear scan over all spheres (reference algorithm): matrix-pipe filter + exact test of its candidates" +
                                (", ray-pool kernel (RTW_FLAG_RAY_POOL)" if args.ray_pool else "")),
                       "parallelism": f"tile-sharded x{world}" + (f" + 1 RCCL {args.collective}" if world > 1 else ""),
                       "rng": f"Xoroshiro128+ per (pixel, chunk), {stats[0]['n_chunks']} chunks/pixel; exact fixed-point pixel accumulation",
                       "numerics": args.numerics + " (RTW_FLAG_NUMERICS_*: the evaluation order of the ray-sphere discriminant, src/hit.jl:16-18; "
                                                   "reference = StaticArrays' un-fused dot, one rounding per written operation)"},
            "world_size_observed": dist.get_world_size() if world > 1 else 1, "backend": c.backend,
            "launched_by": "bench.py (self-spawned torch.distributed.run)" if os.environ.get("RTW_BENCH_SELF_SPAWNED") == "1" else
                           ("torch.distributed.run" if "WORLD_SIZE" in os.environ else "python"),
            "one_device_emulation": c.one_device if world > 1 else None,
            "per_rank": per_rank, "frame_sha256": sha,
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_16t": cpu16,
            "cpu_baseline_all_threads": (legs[-1] if legs else None), "accelerated": accel,
            "scan_valu": scan_valu, "ray_pool": ray_pool, "end_to_end": end_to_end, "depth16": depth16,
            "segments_per_sample": round(all_segments / (samples_per_step * args.steps), 4),
            "numerics": args.numerics, "numerics_legs": numerics_legs, "in_library_devices": in_lib,
            "small_frames": small,
        }
        # (the driver's record keeps the VALUES of `config` and `roofline`, of other objects only the key names: the rates a reader of
        #  BENCH_rNN.json needs are repeated there)
        if small and "error" not in small:
            line["config"]["small_frames_Msamples_per_s"] = {k: {m: v[m]["Msamples_per_s"] for m in ("plain", "group_cull") if m in v}
                                                             for k, v in small.items() if isinstance(v, dict) and "plain" in v}
            line["config"]["small_frames_call_us_median"] = {k: v["plain"]["call_us_median"] for k, v in small.items() if isinstance(v, dict) and "plain" in v}
        if accel:
            line["roofline"]["accelerated_value"] = accel["value"]
            line["roofline"]["accelerated_ms_per_step"] = accel["ms_per_step"]
        if end_to_end:
            line["roofline"]["value_end_to_end"] = end_to_end["value"]
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        if cpu16:
            line["gpu_over_cpu_16t"] = round(value / cpu16["value"], 1)
    wl.close()

    # configs[4]'s single-GPU share, in the default run only (the headline stays configs[2])
    if extras and world == 1 and (args.dtype, W, spp, depth) == ("f32", 1920, 1000, 50) and not (args.group_cull or args.scan_valu):
        w4 = Workload(c, "f64", 3840, 1000, 50)
        st4 = []
        n4 = 2
        dt4 = w4.timed(n4, 1, cull=False, depth=50, record=st4)
        k4 = sum(s["kernel_ms"] for s in st4) / len(st4) / 1e3
        samples4 = w4.W * w4.H * 1000
        f64_4k = {"config": {"workload": f"scene_random_spheres ({w4.n_spheres} spheres), t_cam1, 3840x2160, 1000 spp, depth 50, Float64 "
                                         f"({cfg_name('f64', 3840, 1000, 50, 1)})"},
                  "value": round(samples4 * n4 / dt4 / 1e6, 2), "unit": "Msamples/s", "steps": n4, "warmup": 1, "ms_per_step": round(dt4 / n4 * 1e3, 3),
                  "dtype": "f64", "frame_sha256": w4.frame_sha256(),
                  "roofline": roofline_of(w4, k4, sum(s["sphere_tests"] for s in st4) / len(st4),
                                          sum(s["segments"] for s in st4) / (samples4 * n4), cull=False, valu=False, world=1, shard_div=1)}
        f64_4k["roofline"]["note_f64"] = ("pass 1 is the same binary32/f16 matrix-pipe filter in both precisions; FP64 arithmetic (two issue slots per "
                                          "instruction) only in pass 2 and shading, so the same issue bound applies")
        if not args.no_cpu_baseline:
            legs4, _ = cpu_legs(w4, min(args.cpu_seconds, 6.0), thread_counts=cpu_counts_short)
            f64_4k["cpu_baseline"] = max(legs4, key=lambda d: d["value"])
            f64_4k["gpu_over_cpu"] = round(f64_4k["value"] / f64_4k["cpu_baseline"]["value"], 1)
        w4.close()
        # the reference's PUBLISHED configuration (README.md:86,118-119; src/proto/proto.jl:229-230): Float64, 1920x1080, 1000 spp, depth 16
        wp = Workload(c, "f64", 1920, 1000, 16)
        stp = []
        n_p = 3
        dtp = wp.timed(n_p, 1, cull=False, depth=16, record=stp)
        kp = sum(s["kernel_ms"] for s in stp) / len(stp) / 1e3
        samples_p = wp.W * wp.H * 1000
        f64_pub = {"config": {"workload": f"scene_random_spheres ({wp.n_spheres} spheres), t_cam1, 1920x1080, 1000 spp, depth 16, Float64 "
                                          "(the configuration of the reference's only published render time, /root/reference/README.md:86,118-119)"},
                   "value": round(samples_p * n_p / dtp / 1e6, 2), "unit": "Msamples/s", "steps": n_p, "warmup": 1, "ms_per_step": round(dtp / n_p * 1e3, 3),
                   "dtype": "f64", "frame_sha256": wp.frame_sha256(),
                   "roofline": roofline_of(wp, kp, sum(s["sphere_tests"] for s in stp) / len(stp),
                                           sum(s["segments"] for s in stp) / (samples_p * n_p), cull=False, valu=False, world=1, shard_div=1)}
        f64_pub["vs_published"] = {"published": PUBLISHED_MSAMPLES, "unit": "Msamples/s", "ratio": round(f64_pub["value"] / PUBLISHED_MSAMPLES, 1),
                                   "published_on": "Ryzen 3700 (8C/16T), julia -t 16, 1282.44 s -- DIFFERENT HARDWARE, context only (not `vs_baseline`)"}
        if not args.no_cpu_baseline:
            legsp, _ = cpu_legs(wp, min(args.cpu_seconds, 6.0), thread_counts=cpu_counts_short)
            f64_pub["cpu_baseline"] = max(legsp, key=lambda d: d["value"])
            f64_pub["gpu_over_cpu"] = round(f64_pub["value"] / f64_pub["cpu_baseline"]["value"], 1)
        wp.close()
    if rank == 0:
        line["f64_4k"] = f64_4k
        line["f64_1080p_d16"] = f64_pub
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
