#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path render -> ray_color -> hit/scatter on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full render of BASELINE.json's headline workload (configs[2]):
scene_random_spheres (reseed!(); 485 spheres), camera t_cam1, 1920x1080, 1000 spp, depth 50,
Float32, scene already resident in HBM, image left in HBM on rank 0.  With N > 1 ranks the SAME
image is tile-sharded over the ranks (strong scaling) and the zero-padded shard framebuffers are
summed onto rank 0 with one RCCL reduce over xGMI inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- FP32 VALU roofline of the dominant kernel (trace_kernel): algorithmic flops =
                  counted ray-sphere tests x 17 flop (/root/reference/src/hit.jl:13-19),
                  divided by the kernel's HIP-event time on its launch stream; plus the HBM figure
                  the north star asks for (algorithmic bytes / time vs 8 TB/s).
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores on a
                  bounded sample of the same workload (same image, fewer spp).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

FP32_VALU_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 vector (= FP32 matrix)
HBM_PEAK_GBS = 8000.0
FLOP_PER_TEST = 17              # 3 sub + 5 + 5 (dots) + mul/sub + mul/sub, src/hit.jl:13-19


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--group-cull", action="store_true", help="time the opt-in accelerated scan instead of the plain one")
    ap.add_argument("--chunks", type=int, default=0, help="sample chunks per pixel (0 = library default rule)")
    ap.add_argument("--emulate-shard-of", type=int, default=0,
                    help="analysis only: on ONE GPU render shard 0 of N (what each rank of an N-GPU run does)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline sample duration")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import rtw_amd as R

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    T = np.float32
    W, spp, depth = args.width, args.spp, args.depth
    H = R.image_height(W)
    R.reseed()                                              # src/proto/proto.jl:198-199
    scene = R.scene_random_spheres(elem_type=T)
    cam = R.t_cam1(elem_type=T)
    n_spheres = len(scene)
    renderer = R.DeviceRenderer(scene, cam, device=local_rank)
    fb = torch.empty(H * W * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)

    kernel_ms, total_ms, tests, segments = [], [], [], []
    stats_chunks = [0]

    def step(record):
        def shard(idx, cnt):
            if args.emulate_shard_of > 1:
                idx, cnt = 0, args.emulate_shard_of
            renderer.render_into(fb.data_ptr(), W, spp, depth=depth, seed=1, n_chunks=args.chunks, shard_index=idx,
                                 shard_count=cnt, stream=stream.cuda_stream, group_cull=args.group_cull)
            return fb
        R.render_sharded(shard, W)                          # renders this rank's tiles, one reduce onto rank 0
        if record:
            st = renderer.stats()                           # waits for this rank's kernels (HIP events on `stream`)
            kernel_ms.append(st["kernel_ms"]); total_ms.append(st["total_ms"]); stats_chunks[0] = st["n_chunks"]
            tests.append(st["sphere_tests"]); segments.append(st["segments"])

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        agg = torch.tensor([sum(tests), sum(segments), max(kernel_ms) if kernel_ms else 0.0], dtype=torch.float64, device=dev)
        tot = agg.clone(); dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        kmax = agg[2:3].clone(); dist.all_reduce(kmax, op=dist.ReduceOp.MAX)
        all_tests, all_segments = float(tot[0]), float(tot[1])
    else:
        all_tests, all_segments = float(sum(tests)), float(sum(segments))

    samples_per_step = W * H * spp
    value = samples_per_step * args.steps / dt / 1e6

    # the opt-in accelerated scan (RTW_FLAG_GROUP_CULL, bit-identical image), timed the same way, reported
    # separately: `value` stays the reference's plain linear scan so that the roofline figure means what it says
    accel = None
    if not args.group_cull and args.emulate_shard_of <= 1:
        def step_accel():
            def shard(idx, cnt):
                renderer.render_into(fb.data_ptr(), W, spp, depth=depth, seed=1, n_chunks=args.chunks, shard_index=idx,
                                     shard_count=cnt, stream=stream.cuda_stream, group_cull=True)
                return fb
            R.render_sharded(shard, W)
        step_accel()
        fence()
        ta = time.perf_counter()
        for _ in range(args.steps):
            step_accel()
        fence()
        dta = time.perf_counter() - ta
        if world > 1:
            tmax = torch.tensor([dta], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dta = float(tmax.item())
        accel = {"mode": "RTW_FLAG_GROUP_CULL (kd clusters of 16 + conservative per-ray grown AABB slab test; same image bit for bit)",
                 "value": round(samples_per_step * args.steps / dta / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(dta / args.steps * 1e3, 3)}

    if rank == 0:
        # dominant kernel = trace_kernel.  Per launch (this rank's shard): algorithmic flops =
        # sphere tests x 17; duration = mean HIP-event time on the launch stream.
        k_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        tests_per_launch = sum(tests) / len(tests)
        achieved_tflops = tests_per_launch * FLOP_PER_TEST / k_s / 1e12
        alg_bytes = W * H * 3 * 4 / world + n_spheres * 48        # framebuffer write + one scene read
        # HBM bytes per launch from the committed PMC passes (profiles/), valid for the default workload only
        traffic = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic_1080p_1000spp.json")))
            if (W, spp, depth, world) == (1920, 1000, 50, 1):
                traffic = tr["hbm_bytes_per_launch"]
        except Exception:
            pass
        roofline = {
            "bound": "valu_fp32", "kernel": "rtw::trace_kernel<float>",
            "achieved": round(achieved_tflops, 3), "peak": FP32_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved_tflops / FP32_VALU_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/)",
            "kernel_ms": round(k_s * 1e3, 3), "tests_per_launch": int(tests_per_launch),
            "flop_per_test": FLOP_PER_TEST, "segments_per_sample": round(all_segments / (samples_per_step * args.steps), 4),
            "note": "peak = MI355X FP32 vector peak (= dense FP32 MFMA peak); the path has no dense contraction, so no MFMA",
            "hbm": {"algorithmic_bytes": int(alg_bytes), "achieved_GBs": round(alg_bytes / k_s / 1e9, 4),
                    "peak_GBs": HBM_PEAK_GBS, "frac": round(alg_bytes / k_s / 1e9 / HBM_PEAK_GBS, 8)},
        }
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            import rtw_oracle as O
            O.build()
            flat = R.flatten_scene(scene, T)
            threads = O.max_threads()
            t = time.perf_counter()
            O.render(flat, cam, W, H, 1, T=T, max_depth=depth, seed=1, n_chunks=1)
            t1 = time.perf_counter() - t
            s_spp = int(max(1, min(64, round(args.cpu_seconds / max(t1, 1e-3)))))
            t = time.perf_counter()
            O.render(flat, cam, W, H, s_spp, T=T, max_depth=depth, seed=1)
            tc = time.perf_counter() - t
            cpu = {"value": round(W * H * s_spp / tc / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                   "sample": f"same scene/camera/{W}x{H}/depth {depth}, {s_spp} spp ({tc:.1f} s), oracle/ C port with OpenMP; "
                             f"the Julia reference cannot run here (no julia in the image)"}
        if args.emulate_shard_of > 1:
            samples_per_step = samples_per_step / args.emulate_shard_of
            value = samples_per_step * args.steps / dt / 1e6
        line = {
            "metric": "Msamples/s (pixels x spp) on scene_random_spheres 1920x1080",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"scene_random_spheres ({n_spheres} spheres, reseed!() seed 1), t_cam1, {W}x{H}, {spp} spp, "
                                   f"depth {depth}, Float32 (BASELINE.json configs[2])",
                       "scan": "group_cull (opt-in)" if args.group_cull else "plain linear scan over all spheres (reference algorithm)",
                       "parallelism": f"tile-sharded x{world}" + (" + 1 RCCL reduce" if world > 1 else ""),
                       "rng": f"Xoroshiro128+ per (pixel, chunk), {stats_chunks[0]} chunks/pixel"},
            "roofline": roofline, "cpu_baseline": cpu, "accelerated": accel,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
