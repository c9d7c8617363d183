#!/usr/bin/env python3
"""bench.py -- Msamples/s of the hot path render -> ray_color -> hit/scatter on MI355X.

  python bench.py --gpus N --steps K --warmup W [--dtype f32|f64] [--width 1920|3840]
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one full render of the workload.  Default = BASELINE.json's headline, configs[2]:
scene_random_spheres (reseed!(); 485 spheres), camera t_cam1, 1920x1080, 1000 spp, depth 50,
Float32.  `--dtype f64 --width 3840` is the single-GPU share of configs[4] (3840x2160, Float64).
The scene is already resident in HBM when the timed region starts and the image is left in HBM
on rank 0.  With N > 1 ranks the SAME image is tile-sharded over the ranks (strong scaling) and
the zero-padded shard framebuffers are summed onto rank 0 with one RCCL reduce over xGMI inside
the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline     -- roofline of the only kernel (rtw::trace_kernel): achieved = algorithmic flops =
                  counted ray-sphere tests x 17 flop (/root/reference/src/hit.jl:13-19)
                  divided by the kernel's HIP-event time on its launch stream.  The every-ray-
                  every-sphere part that this count measures runs as a conservative f16-split filter
                  on the matrix pipe plus one FP32 fma + one alignbit per test, the
                  exact contract arithmetic (FP32 / FP64) only on its candidates; a SIMD issues EITHER
                  an MFMA or a VALU instruction (measured), so `peak` is the issue bound of that
                  formulation: 4 MFMA cycles + 2 VALU x 2 cycles per 64 tests -> 334.2 algorithmic
                  TFLOP/s.  `vs_fp32_vector_peak` relates the same achieved figure to the 157.3 TF
                  vector peak that bounded the all-VALU scan of rounds 1-2 (it can exceed 1 now),
                  `mfma_f16` gives the executed matrix-pipe flops against the 2.5 PF dense peak.
                  `traffic` = HBM bytes per launch from the PMC passes (profiles/), plus the
                  algorithmic HBM figure vs 8 TB/s.
  scan_valu    -- the same workload with RTW_FLAG_SCAN_VALU (the contract discriminant for every
                  sphere on the vector ALUs, the round-1/2 scan): same image, for comparison.
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores on a
                  bounded sample of the same workload (same image, fewer spp): the faster of a
                  16-thread leg (`cpu_baseline_16t`, the north star's comparison point) and an
                  all-threads leg (`cpu_baseline_all_threads`).
  accelerated  -- the opt-in RTW_FLAG_GROUP_CULL mode (same image bit for bit), reported separately.
  end_to_end   -- one call of the host-buffer entry point rtw_render_* (scene upload, render,
                  D2H of the image: the PCIe-inclusive rate; never `value`).
  depth16      -- the same workload at the reference's own depth (src/ray_color.jl:14).
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

# /opt/skills/guides/MI355X_MICROARCH.md: peak FP32 vector 157.3 TFLOP/s (= 256 CU x 4 SIMD x 32 lanes x 2 flop x 2.4 GHz);
# the guide lists no FP64 vector figure: AMD's MI355X datasheet gives 78.6 TFLOP/s (half the FP32 rate).
VALU_PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}
MFMA_F16_PEAK_TFLOPS = 2500.0   # same guide: ~2.5 PF dense BF16/FP16 MFMA (2495 TF measured with 32x32x16)
MFMA_FLOP_PER_TEST = 64         # two v_mfma_f32_32x32x16_f16 products per (ray, sphere): 2 x K=16 x 2 flop
HBM_PEAK_GBS = 8000.0
FLOP_PER_TEST = 17              # 3 sub + 5 + 5 (dots) + mul/sub + mul/sub, src/hit.jl:13-19


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--spp", type=int, default=1000)
    ap.add_argument("--depth", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the accelerated / end_to_end / depth16 legs")
    ap.add_argument("--group-cull", action="store_true", help="time the opt-in accelerated scan instead of the plain one")
    ap.add_argument("--scan-valu", action="store_true", help="time the all-VALU plain scan (RTW_FLAG_SCAN_VALU) instead of the matrix-pipe filter")
    ap.add_argument("--chunks", type=int, default=0, help="sample chunks per pixel (0 = library default rule)")
    ap.add_argument("--emulate-shard-of", type=int, default=0,
                    help="analysis only: on ONE GPU render shard 0 of N (what each rank of an N-GPU run does)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target CPU-baseline sample duration per leg")
    ap.add_argument("--collective", choices=["reduce", "gather"], default="reduce",
                    help="N > 1: reduce = sum of zero-padded full frames onto rank 0 (BASELINE configs[3]); "
                         "gather = each rank sends only its compact tile-major shard (1/N of a frame)")
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
    # RTW_BENCH_ONE_DEVICE=1 (test aid for one-GPU boxes): every rank uses cuda:0 and the collective runs over gloo --
    # exercises the N > 1 control flow of this script (sharding, collective, max-over-ranks timing), not RCCL itself
    one_device = os.environ.get("RTW_BENCH_ONE_DEVICE") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    T = np.float64 if args.dtype == "f64" else np.float32
    tT = torch.float64 if args.dtype == "f64" else torch.float32
    jl = "Float64" if args.dtype == "f64" else "Float32"
    W, spp, depth = args.width, args.spp, args.depth
    H = R.image_height(W)
    R.reseed()                                              # src/proto/proto.jl:198-199
    scene = R.scene_random_spheres(elem_type=T)
    cam = R.t_cam1(elem_type=T)
    n_spheres = len(scene)
    renderer = R.DeviceRenderer(scene, cam, device=local_rank)
    fb = torch.empty(H * W * 3, dtype=tT, device=dev)
    stream = torch.cuda.current_stream(dev)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def timed(n_steps, n_warm, *, cull, depth_, record=None, valu=args.scan_valu):
        """W untimed + K timed steps bracketed by barrier + synchronize; returns max-over-ranks seconds."""
        def step(rec):
            def shard(idx, cnt):
                if args.emulate_shard_of > 1:
                    idx, cnt = 0, args.emulate_shard_of
                renderer.render_into(fb.data_ptr(), W, spp, depth=depth_, seed=1, n_chunks=args.chunks, shard_index=idx,
                                     shard_count=cnt, stream=stream.cuda_stream, group_cull=cull,
                                     compact=args.collective == "gather", scan_valu=valu and not cull)
                return fb
            R.render_sharded(shard, W, mode=args.collective)    # renders this rank's tiles, ONE collective onto rank 0
            if rec is not None:
                rec.append(renderer.stats())                # waits for this rank's kernel (HIP events on `stream`)
        for _ in range(n_warm):
            step(None)
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step(record)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    stats = []
    dt = timed(args.steps, args.warmup, cull=args.group_cull, depth_=depth, record=stats)
    kernel_ms = [s["kernel_ms"] for s in stats]
    tests = [s["sphere_tests"] for s in stats]
    segments = [s["segments"] for s in stats]
    if world > 1:
        agg = torch.tensor([sum(tests), sum(segments)], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        all_segments = float(agg[1])
    else:
        all_segments = float(sum(segments))

    shard_div = args.emulate_shard_of if args.emulate_shard_of > 1 else 1
    samples_per_step = W * H * spp / shard_div
    value = samples_per_step * args.steps / dt / 1e6

    extras = not args.no_extras and args.emulate_shard_of <= 1
    accel = depth16 = None
    if extras and not args.group_cull:
        # the opt-in accelerated scan (RTW_FLAG_GROUP_CULL, bit-identical image), timed the same way, reported
        # separately: `value` stays the reference's plain linear scan so that the roofline figure means what it says
        dta = timed(args.steps, 1, cull=True, depth_=depth)
        accel = {"mode": "RTW_FLAG_GROUP_CULL (kd-sorted blocks of 32 spheres skipped when no ray of the wave can touch the block's grown box, in front of the matrix-pipe filter; same image bit for bit)",
                 "value": round(samples_per_step * args.steps / dta / 1e6, 2), "unit": "Msamples/s",
                 "ms_per_step": round(dta / args.steps * 1e3, 3)}
    scan_valu = None
    if extras and not args.group_cull and not args.scan_valu:
        dtv = timed(1, 0, cull=False, depth_=depth, valu=True)
        scan_valu = {"mode": "RTW_FLAG_SCAN_VALU (contract discriminant for every sphere on the vector ALUs; same image bit for bit)",
                     "value": round(samples_per_step / dtv / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dtv * 1e3, 3)}
    if extras and depth != 16:
        st16 = []
        dt16 = timed(1, 0, cull=args.group_cull, depth_=16, record=st16)
        depth16 = {"value": round(samples_per_step / dt16 / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(dt16 * 1e3, 3),
                   "segments_per_sample": round(st16[0]["segments"] * world / samples_per_step, 4) if world == 1 else None,
                   "note": "same workload at depth 16, the reference's only depth (src/ray_color.jl:14)"}

    end_to_end = None
    if extras and world == 1 and rank == 0:
        # host-buffer entry point, what the Julia ccall binds: scene upload + render + image D2H, blocking
        t = time.perf_counter()
        R.render(scene, cam, W, spp, depth=depth, seed=1, n_chunks=args.chunks, device=local_rank, group_cull=args.group_cull,
                 scan_valu=args.scan_valu)
        te = time.perf_counter() - t
        end_to_end = {"value": round(W * H * spp / te / 1e6, 2), "unit": "Msamples/s", "ms": round(te * 1e3, 3),
                      "kernel_ms": round(R.last_stats()["kernel_ms"], 3),
                      "note": "rtw_render_* on host buffers: H2D scene, render, D2H image (PCIe-inclusive); never `value`"}

    if rank == 0:
        # the only kernel = trace_kernel.  Per launch (this rank's shard): algorithmic flops =
        # sphere tests x 17; duration = mean HIP-event time on the launch stream.
        k_s = (sum(kernel_ms) / len(kernel_ms)) / 1e3
        tests_per_launch = sum(tests) / len(tests)
        # issue bound of the matrix-pipe formulation: per 64 tests (one sphere x one wave) 4 v_mfma_f32_32x32x16_f16 per 32 spheres x
        # 32 cycles (MI355X_MICROARCH.md: 32 cyc/SIMD) = 4 cycles, + 2 VALU x 2 cycles (v_fma_f32: 2 cyc/SIMD) = 4 cycles; MFMA and
        # VALU issue do not overlap on a SIMD (tools/ubench_mfma_overlap.hip, profiles/r02_ubench_mfma.txt)
        issue_peak = 1024 * 2.4e9 / 8 * 64 * FLOP_PER_TEST / 1e12
        peak = issue_peak if not args.scan_valu else VALU_PEAK_TFLOPS[args.dtype]
        achieved_tflops = tests_per_launch * FLOP_PER_TEST / k_s / 1e12
        mfma_tflops = tests_per_launch * MFMA_FLOP_PER_TEST / k_s / 1e12
        esize = 8 if args.dtype == "f64" else 4
        alg_bytes = W * H * 3 * esize / world / shard_div + n_spheres * 12 * esize   # framebuffer write + one scene read
        # HBM bytes per launch from the PMC passes (profiles/), valid for the exact workload they were taken on
        traffic = traffic_src = None
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_traffic.json")))
            key = f"{args.dtype}_{W}x{H}_{spp}spp_d{depth}_{'cull' if args.group_cull else ('valu' if args.scan_valu else 'plain')}"
            if world == 1 and shard_div == 1 and key in tr:
                traffic, traffic_src = tr[key]["hbm_bytes_per_launch"], tr[key].get("source")
        except Exception:
            pass
        if args.group_cull:
            achieved_tflops = float("nan")          # the cull mode skips tests: a fraction of never-executed tests would be meaningless
        matrix = not (args.group_cull or args.scan_valu)
        roofline = {
            "bound": ("valu_" + ("fp64" if args.dtype == "f64" else "fp32")) if args.scan_valu else "mfma",
            "bound_detail": None if not matrix else "SIMD issue time shared by v_mfma_f32_32x32x16_f16 (32 cycles each) and FP32 VALU (2 cycles each)",
            "kernel": f"rtw::trace_kernel<{'double' if args.dtype == 'f64' else 'float'}>",
            "achieved": None if args.group_cull else round(achieved_tflops, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": None if args.group_cull else round(achieved_tflops / peak, 4),
            "peak_derivation": None if not matrix else
                "1024 SIMDs x 2.4 GHz / (4 MFMA + 4 VALU cycles per 64 tests) x 64 tests x 17 algorithmic flop (cycle counts: MI355X_MICROARCH.md)",
            "traffic": traffic, "traffic_unit": "HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes)",
            "traffic_source": traffic_src,
            "kernel_ms": round(k_s * 1e3, 3), "tests_per_launch": int(tests_per_launch),
            "flop_per_test": FLOP_PER_TEST, "segments_per_sample": round(all_segments / (samples_per_step * args.steps), 4),
            "vs_fp32_vector_peak": None if not matrix else {"peak": VALU_PEAK_TFLOPS["f32"], "frac": round(achieved_tflops / VALU_PEAK_TFLOPS["f32"], 4),
                                                             "note": "the bound of the all-VALU scan (rounds 1-2: 0.486 / 0.50); the algorithmic flops are "
                                                                     "no longer executed as vector flops, so this can exceed 1"},
            "mfma_f16": None if not matrix else {"executed_flop_per_test": MFMA_FLOP_PER_TEST, "achieved": round(mfma_tflops, 1),
                                                  "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mfma_tflops / MFMA_F16_PEAK_TFLOPS, 4)},
            "valu_fp32": None if not matrix else {"instructions_per_test": 2, "what": "v_fma_f32 (hb^2 + m) + v_alignbit_b32 (sign bit into the candidate mask)"},
            "note": "achieved = counted ray-sphere tests x 17 algorithmic flop / kernel time.  Every sphere is tested against every ray segment, but "
                    "pass 1 of the scan is a conservative FILTER: the discriminant is bilinear in (ray features) x (sphere features), so two "
                    "v_mfma_f32_32x32x16_f16 over f16-split features evaluate 32 spheres x 32 rays and the VALU adds one fma + one alignbit per test "
                    "(rigorous margin, DESIGN.md 6.1); the exact contract arithmetic (17 flop, FP32 or FP64) runs only on the filter's candidates.  "
                    "`scan_valu` is the same workload with every test on the vector ALUs."
                    if matrix else
                    ("all-VALU plain scan (RTW_FLAG_SCAN_VALU): 11 instructions per test (Float32) / 13 binary32 filter instructions (Float64)"
                     if args.scan_valu else "group-cull mode: tests are skipped, no roofline fraction"),
            "hbm": {"algorithmic_bytes": int(alg_bytes), "achieved_GBs": round(alg_bytes / k_s / 1e9, 4),
                    "peak_GBs": HBM_PEAK_GBS, "frac": round(alg_bytes / k_s / 1e9 / HBM_PEAK_GBS, 8)},
        }
        cpu = cpu16 = None
        legs = []
        if world == 1 and not args.no_cpu_baseline:
            import rtw_oracle as O
            O.build()
            flat = R.flatten_scene(scene, T)
            threads = O.max_threads()

            def cpu_leg(nthr):
                t = time.perf_counter()
                O.render(flat, cam, W, H, 1, T=T, max_depth=depth, seed=1, n_chunks=1, omp_threads=nthr)
                t1 = time.perf_counter() - t
                s_spp = int(max(1, min(64, round(args.cpu_seconds / max(t1, 1e-3)))))
                t = time.perf_counter()
                O.render(flat, cam, W, H, s_spp, T=T, max_depth=depth, seed=1, omp_threads=nthr)
                tc = time.perf_counter() - t
                return {"value": round(W * H * s_spp / tc / 1e6, 4), "unit": "Msamples/s", "cores": nthr, "kind": "port",
                        "sample": f"same scene/camera/{W}x{H}/depth {depth}/{jl}, {s_spp} spp ({tc:.1f} s), oracle/ C port with OpenMP; "
                                  f"the Julia reference cannot run here (no julia in the image)"}
            # the GPU boxes are shared hosts (2 x EPYC 9575F, cgroup-limited): all-threads runs are often SLOWER than
            # 16 threads there.  Both legs are reported; `cpu_baseline` is the faster one (the fairer baseline).
            legs = [cpu_leg(16)] if threads >= 16 else []
            legs.append(cpu_leg(threads))
            cpu16 = legs[0] if threads >= 16 else None
            cpu = max(legs, key=lambda d: d["value"])
        cfg_name = {("f32", 1920, 1000, 50): "BASELINE.json configs[2]" if world == 1 else "BASELINE.json configs[3]",
                    ("f64", 3840, 1000, 50): "BASELINE.json configs[4]" + (", one GPU" if world == 1 else "")}.get(
            (args.dtype, W, spp, depth), "not a BASELINE config")
        line = {
            "metric": f"Msamples/s (pixels x spp) on scene_random_spheres {W}x{H}",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"scene_random_spheres ({n_spheres} spheres, reseed!() seed 1), t_cam1, {W}x{H}, {spp} spp, "
                                   f"depth {depth}, {jl} ({cfg_name})",
                       "scan": "group_cull (opt-in)" if args.group_cull else
                               ("plain linear scan over all spheres, all on the VALU (RTW_FLAG_SCAN_VALU)" if args.scan_valu else
                                "plain linear scan over all spheres (reference algorithm): matrix-pipe filter + exact test of its candidates"),
                       "parallelism": f"tile-sharded x{world}" + (f" + 1 RCCL {args.collective}" if world > 1 else ""),
                       "rng": f"Xoroshiro128+ per (pixel, chunk), {stats[0]['n_chunks']} chunks/pixel; exact fixed-point pixel accumulation"},
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_16t": cpu16,
            "cpu_baseline_all_threads": (legs[-1] if world == 1 and not args.no_cpu_baseline else None), "accelerated": accel,
            "scan_valu": scan_valu, "end_to_end": end_to_end, "depth16": depth16,
        }
        if cpu:
            line["gpu_over_cpu"] = round(value / cpu["value"], 1)
        if cpu16:
            line["gpu_over_cpu_16t"] = round(value / cpu16["value"], 1)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
