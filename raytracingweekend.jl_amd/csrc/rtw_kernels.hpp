// rtw_kernels.hpp -- the trace kernel: render -> ray_color -> hit/scatter fused, including the
// pixel accumulation and the gamma / store of src/render.jl:40.  gfx950 only; wave = 64 lanes.
//
// Work decomposition
//   job   = one 4x4 pixel block of an 8x8 tile (2x2 or 1 pixel for small shards) for ALL its sample chunks.  Jobs come
//           from one global queue (one atomic per job); a job is owned by ONE workgroup, whose
//           LDS holds the block's 16 x 3 pixel accumulators while the job is in flight.
//   item  = (pixel, chunk): `chunk_spp` consecutive samples of one pixel drawn from the item's own
//           Xoroshiro128+ stream.  64 consecutive items = job pixels x (64 / job pixels) chunks = one wave batch,
//           taken from the workgroup's ticket counter in LDS.
//   lane  = persistent worker.  It owns one item at a time, starts its next sample the moment
//           its path ends (no lock-step on path length) and takes the next item when its chunk
//           is finished.  The only convergent part is the sphere scan, which every lane with a
//           ray executes every iteration.
//   accum = a finished chunk sum (3 doubles) is added to the job's accumulators EXACTLY (64.64
//           fixed point, LDS integer atomics) -- addition order does not matter, so there is no
//           per-chunk workspace in HBM and no second kernel: the lane that retires the job's last
//           item triggers the store of the 16 pixels (sum -> / spp -> sqrt -> RGB{T}).
//   Results do not depend on scheduling, grid size, shard count or job-slot availability.
#pragma once
#include "rtw_device.hpp"

namespace rtw {

#define RTW_JOB_PX 16        // slot capacity: pixels per job are 16 (4x4 block), 4 (2x2) or 1 -- KParams::job_shift
#define RTW_SLOT_BYTES 4608  // LDS for job slots per workgroup: 24 slots of 1 pixel, 12 of 4 or 4 of 16
#define RTW_REF_BITS 9       // item reference = slot (5 bits) << 4 | pixel (4 bits)
#define RTW_REF_MASK 511u
#define RTW_SLOT_FREE 0xffffffffu
#define RTW_SLOT_OPENING 0xfffffffeu
#define RTW_JOB_EOF 0xffffffffu

struct KParams {
    int width, height, spp, max_depth;
    uint64_t seed;
    int n_chunks;      // effective (non-empty) chunks per pixel
    int chunk_spp;
    int shard_index, shard_count;
    int tiles_i, tiles_j;  // 8x8 tiles along rows (i) and columns (j)
    unsigned total_jobs;   // 4 * (tiles owned by this shard)
    unsigned bpj;          // batches per job = ceil(n_chunks / (64 >> job_shift))
    unsigned n_slots, slot_stride, div_slots_m, div_slots_s;   // job slots per workgroup (by job size), bytes per slot, n / n_slots
    unsigned job_shift;    // log2(pixels per job): 4, 2 or 0.  A batch = (1 << job_shift) pixels x (64 >> job_shift) chunks.
                           // Smaller jobs = finer load balance at the end of the queue (small shards); same image.
    // exact unsigned division by loop-invariant divisors (host: make_udiv):
    // n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> s   for every 32-bit n
    unsigned div_bpj_m, div_bpj_s, div_tiles_m, div_tiles_s;
    int gamma;
    int out_layout;        // 0: Matrix{RGB{T}} column-major full frame; 1: compact, tile-major, this shard only
};

struct DevCounters {
    unsigned next_job;
    unsigned pad;
    unsigned long long segments;
    unsigned long long samples;
    unsigned long long phase[8];   // RTW_PHASE_PROFILE=1 only: wave-cycles per phase (s_memtime)
    unsigned long long t_first, t_last, t_end_sum, n_waves;   // wall clock (100 MHz) of the first wave start, the last wave
                                                              // end and the sum of all wave ends: the end-of-queue drain
    unsigned end_hist[4096];                                  // waves by end time since t_first, 0.25 ms bins (RTW_DRAIN_PROFILE)
};

// One job in flight: the bookkeeping of the open/retire protocol and the block header, followed in LDS by
// the job's pixel accumulators (8 x u64 per pixel: r.lo r.hi g.lo g.hi b.lo b.hi poison pad).
struct JobSlot {
    unsigned ready_seq;                      // RTW_SLOT_FREE | RTW_SLOT_OPENING | the job sequence number it holds
    unsigned job;                            // global job id, or RTW_JOB_EOF (queue exhausted; never freed)
    int remaining;                           // items of the job not yet finished
    unsigned valid;                          // bit px: pixel px of the block lies inside the image
    int i_base, j_base;                      // 0-based row / column of the block's first pixel
    unsigned k_tile;                         // local tile index (compact output layout)
    unsigned pad;
    double uv[8];                            // j / W for the block's columns, (H - i) / H for its rows (src/render.jl:26-27), as binary64
    __device__ __forceinline__ unsigned long long *acc(unsigned px) { return reinterpret_cast<unsigned long long *>(this + 1) + 8u * px; }
    __device__ __forceinline__ const unsigned long long *acc(unsigned px) const { return reinterpret_cast<const unsigned long long *>(this + 1) + 8u * px; }
};
static_assert(sizeof(JobSlot) == 96, "JobSlot header is 96 bytes (16-byte aligned accumulators follow)");
template <typename T> struct WgShared {
    unsigned char slots[RTW_SLOT_BYTES];      // n_slots x (96-byte JobSlot + 64 bytes per job pixel)
    __device__ __forceinline__ JobSlot *slot(unsigned i, unsigned stride) { return reinterpret_cast<JobSlot *>(slots + i * stride); }
    unsigned ticket;                         // next batch of this workgroup
    unsigned pad[3];
    Camera<T> cam;                           // read per new sample (keeps 22 SGPRs out of the scan loop)
    KParams P;                               // read where needed (item pull, store): not held in SGPRs across the scan
};

// Phase profiler (opt-in instantiation, never used for timed runs): s_memtime stamps around
// the phases of the lane loop, summed per wave.
template <bool ON> struct PhaseClock {
    unsigned long long t0 = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void start() { if (ON) t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void lap(int k) {
        if (ON) { unsigned long long t = __builtin_readcyclecounter(); acc[k] += t - t0; t0 = t; }
    }
    __device__ __forceinline__ void count(int k, unsigned n) { if (ON) acc[k] += n; }      // event counters in the spare cells 6, 7
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) {
    if (s & 0x80000000u) return n;                      // divisor 1
    const unsigned t = __umulhi(n, m);
    return (t + ((n - t) >> 1)) >> s;
}

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ unsigned uniform(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }

// waves per SIMD the trace kernel is compiled for (second __launch_bounds__ argument):
// Float32 -> 7 (VGPR cap 72), Float64 -> 5 (cap 96; measured 1.5 % faster than 4 waves at 126 VGPRs)
#ifndef RTW_TRACE_WAVES_F32
#define RTW_TRACE_WAVES_F32 7
#endif
#ifndef RTW_TRACE_WAVES_F64
#define RTW_TRACE_WAVES_F64 5
#endif
template <typename T> struct TraceWaves { static constexpr int value = RTW_TRACE_WAVES_F32; };
template <> struct TraceWaves<double> { static constexpr int value = RTW_TRACE_WAVES_F64; };
// the group-cull variant keeps the slab-test constants live across the scan: fewer waves
#ifndef RTW_TRACE_WAVES_CULL_F32
#define RTW_TRACE_WAVES_CULL_F32 5
#endif
template <typename T, bool CULL, bool MFMA = false> struct TraceWavesOf { static constexpr int value = TraceWaves<T>::value; };
template <> struct TraceWavesOf<float, true, false> { static constexpr int value = RTW_TRACE_WAVES_CULL_F32; };
template <> struct TraceWavesOf<double, true, false> { static constexpr int value = 3; };
#ifndef RTW_TRACE_WAVES_MFMA_F32
#define RTW_TRACE_WAVES_MFMA_F32 5
#endif
#ifndef RTW_TRACE_WAVES_MFMA_F64
#define RTW_TRACE_WAVES_MFMA_F64 4
#endif
template <> struct TraceWavesOf<float, false, true> { static constexpr int value = RTW_TRACE_WAVES_MFMA_F32; };   // 32 result registers per product
template <> struct TraceWavesOf<double, false, true> { static constexpr int value = RTW_TRACE_WAVES_MFMA_F64; };
template <> struct TraceWavesOf<float, true, true> { static constexpr int value = RTW_TRACE_WAVES_MFMA_F32; };    // group cull on the matrix pipe
template <> struct TraceWavesOf<double, true, true> { static constexpr int value = RTW_TRACE_WAVES_MFMA_F64; };
// LDS of the matrix-pipe scan per workgroup (4 waves): the result cells (the pair lists take the place of the per-lane lists)
// + per wave the generator states of the 64 items of its current batch (below: "pool"), 16 B each
template <typename T> __host__ __device__ constexpr size_t mfma_cell_bytes() { return 4 * (64 * sizeof(unsigned long long) + (sizeof(T) == 8 ? 64 * sizeof(unsigned) : 0)) + 4 * 64 * 16; }

// store_job / open_job run once per job.  Inlined: as real calls (noinline) they keep the lane loop's register
// pressure lower, but every call saves ~20 live VGPRs to scratch -- 2.7 GB of HBM writes per frame (PMC).
#ifndef RTW_RARE_ATTR
#define RTW_RARE_ATTR __forceinline__
#endif

// Exact accumulation of one sample's radiance into pixel `a` of a job slot.
__device__ __forceinline__ void fx_accumulate(unsigned long long *a, double r, double g, double b) {
    const double cs[3] = {r, g, b};
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {        // not unrolled: one channel's temporaries at a time

        unsigned long long lo, hi;
        if (fx_from_double(cs[c], lo, hi)) {
            const unsigned long long old = __hip_atomic_fetch_add(&a[2 * c], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            hi += (old + lo < old) ? 1ull : 0ull;
            if (hi) __hip_atomic_fetch_add(&a[2 * c + 1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(&a[6], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

// One channel of one sample: `acc` = the pixel's accumulators (a[2c], a[2c+1] per channel, a[6] = poison count).
__device__ __forceinline__ void fx_accumulate_channel(unsigned long long *acc, unsigned c, double v) {
    unsigned long long lo, hi;
    if (fx_from_double(v, lo, hi)) {
        const unsigned long long old = __hip_atomic_fetch_add(&acc[2 * c], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        hi += (old + lo < old) ? 1ull : 0ull;
        if (hi) __hip_atomic_fetch_add(&acc[2 * c + 1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else {
        __hip_atomic_fetch_add(&acc[6], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

// The store of one finished job (src/render.jl:40, src/vec.jl:22): lane = (pixel, channel).
template <typename T>
__device__ RTW_RARE_ATTR void store_job(const KParams &P, const JobSlot *S, unsigned lane, T *__restrict__ out) {
    const unsigned px = lane & ((1u << P.job_shift) - 1u), ch = lane >> P.job_shift, side = P.job_shift >> 1;
    if (ch < 3u) {
        if ((S->valid >> px) & 1u) {
            const int i0 = S->i_base + (int)(px & ((1u << side) - 1u)), j0 = S->j_base + (int)(px >> side);
            const unsigned long long *a = S->acc(px);
            double v = fx_to_double(a[2 * ch], a[2 * ch + 1]);
            if (a[6] != 0ull) v = __builtin_nan("");
            v = v / (double)P.spp;
            if (P.gamma) v = __builtin_sqrt(v);
            const size_t pix = P.out_layout == 0 ? (size_t)j0 * (size_t)P.height + (size_t)i0
                                                 : (size_t)S->k_tile * 64u + (size_t)((i0 & 7) + 8 * (j0 & 7));
            out[pix * 3 + ch] = (T)v;
        }
    }
}

// Open a job slot (whole wave): take job ids from the global queue until one has a pixel inside the
// image (or the queue is exhausted), zero its accumulators and fill in the block's header.
__device__ RTW_RARE_ATTR void open_job(const KParams &P, JobSlot *S, unsigned lane, DevCounters *ctr) {
    unsigned g, valid = 0, k = 0;
    int i_base = 0, j_base = 0;
    for (;;) {
        g = 0;
        if (lane == 0) g = atomicAdd(&ctr->next_job, 1u);
        g = uniform(g);
        if (g >= P.total_jobs) { g = RTW_JOB_EOF; break; }
        const unsigned side = P.job_shift >> 1, sub_shift = 6u - P.job_shift, bps_shift = 3u - side;
        k = g >> sub_shift;                                  // job = (local tile) * (blocks per tile) + block
        const unsigned q = g & ((1u << sub_shift) - 1u);
        const unsigned t = k * (unsigned)P.shard_count + (unsigned)P.shard_index;
        const unsigned tj = udiv_magic(t, P.div_tiles_m, P.div_tiles_s), ti = t - tj * (unsigned)P.tiles_i;
        i_base = (int)(ti * 8u + ((q & ((1u << bps_shift) - 1u)) << side));
        j_base = (int)(tj * 8u + ((q >> bps_shift) << side));
        const int i0 = i_base + (int)(lane & ((1u << side) - 1u)), j0 = j_base + (int)((lane >> side) & ((1u << side) - 1u));
        valid = (unsigned)__ballot(lane < (1u << P.job_shift) && i0 < P.height && j0 < P.width);
        if (valid) break;                                    // (blocks entirely outside the image are skipped)
    }
    if (g != RTW_JOB_EOF) {
        if (lane < (4u << P.job_shift)) {
            unsigned zero = 0u;
            __asm__ volatile("" : "+v"(zero));       // (made here: a hoisted zero quad costs 4 loop-long registers)
            reinterpret_cast<uint4 *>(S->acc(0))[lane] = uint4{zero, zero, zero, zero};   // 64 B per pixel
        }
        if (lane < 4) S->uv[lane] = (double)(j_base + (int)lane + 1) / (double)P.width;                        // j / W
        else if (lane < 8) S->uv[lane] = (double)(P.height - (i_base + (int)lane - 4 + 1)) / (double)P.height;   // (H - i) / H
        if (lane == 0) {
            S->remaining = (int)((unsigned)__popc(valid) * (unsigned)P.n_chunks);
            S->valid = valid; S->i_base = i_base; S->j_base = j_base; S->k_tile = k;
        }
    }
    if (lane == 0) S->job = g;
}

template <typename T, bool PROFILE, bool LDS_SCENE, bool CULL, bool MFMA = false>
__global__ __launch_bounds__(256, (TraceWavesOf<T, CULL, MFMA>::value)) void trace_kernel(KParams P_arg, Camera<T> cam_arg, DevScene<T> scene,
                                                   CullScene<T> cull, T *__restrict__ out, DevCounters *ctr) {
    using V4 = typename Vec4<T>::type;
    const unsigned lane = lane_id();
    // LDS: [per-lane candidate lists, stride 256][job slots, ticket, camera][scene geom copy (LDS_SCENE only)]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *my_list = reinterpret_cast<unsigned short *>(smem) + threadIdx.x;
    constexpr size_t list_bytes = RTW_LIST_CAP * 256 * sizeof(unsigned short);
    constexpr size_t shared_bytes = (sizeof(WgShared<T>) + 15) / 16 * 16;
    WgShared<T> *sh = reinterpret_cast<WgShared<T> *>(smem + list_bytes);
    constexpr size_t cell_bytes = MFMA ? mfma_cell_bytes<T>() : 0;
    V4 *lds_geom = reinterpret_cast<V4 *>(smem + list_bytes + shared_bytes + cell_bytes);
    WaveScratch ws = {nullptr, nullptr, nullptr};
    [[maybe_unused]] ulonglong2 *pool_rng = nullptr;      // MFMA variants: generator states of the wave's current batch, made by the whole wave
    [[maybe_unused]] unsigned long long pool_valid = 0;   // ... and which of its 64 items are real (chunk < n_chunks, pixel inside the image)
    if constexpr (MFMA) {
        static_assert(list_bytes == 4 * RTW_PAIR_CAP * sizeof(unsigned), "the pair lists reuse the per-lane list area");
        const unsigned wv = threadIdx.x >> 6;
        unsigned char *cells = smem + list_bytes + shared_bytes;
        ws.pairs = reinterpret_cast<unsigned *>(smem) + wv * RTW_PAIR_CAP;
        ws.keys = reinterpret_cast<unsigned long long *>(cells) + wv * 64;
        ws.kidx = reinterpret_cast<unsigned *>(cells + 4 * 64 * sizeof(unsigned long long)) + wv * 64;
        pool_rng = reinterpret_cast<ulonglong2 *>(cells + mfma_cell_bytes<T>() - 4 * 64 * 16) + wv * 64;
    }
    unsigned short *lds_orig = reinterpret_cast<unsigned short *>(lds_geom + (CULL ? cull_exact_count(cull) : 0));
    if (threadIdx.x < P_arg.n_slots) { JobSlot *S0 = sh->slot(threadIdx.x, P_arg.slot_stride); S0->ready_seq = RTW_SLOT_FREE; S0->job = 0u; }
    if (threadIdx.x == 0) { sh->ticket = 0u; sh->cam = cam_arg; sh->P = P_arg; }
    const KParams &P = sh->P;
    if (LDS_SCENE) {
        if (CULL) stage_cull_scene<T>(cull, lds_geom, lds_orig);
        else stage_scene<T>(scene, lds_geom);
    }
    __syncthreads();

    const unsigned long long t_wave_start = wall_clock64();
    // ---- wave-uniform state ----
    unsigned pool_next = 0, pool_end = 0;   // unassigned items [pool_next, pool_end) of the wave's current batch
    unsigned pool_slot = 0, pool_b = 0;
    bool have_ticket = false;               // a batch ticket drawn but not yet usable (its job slot is not open)
    unsigned tk_seq = 0, tk_b = 0;
    unsigned long long n_segments = 0, n_samples = 0;

    // ---- per-lane state that lives across iterations (and so across the scan) ----
    bool alive = true;        // still pulling work
    bool have_item = false;   // owns an item (its job's `remaining` is decremented when the chunk is done)
    bool has_ray = false;     // a ray is ready for the scan
    unsigned ref_depth = 0;   // (bounces left << 9) | item_ref, item_ref = slot * 16 + pixel of the owned item
    int samples_left = 0;
    bool jitter = false;      // false only for sample 1 of the pixel (src/render.jl:30-31)
    Rng rng = {1, 2};
    V3<T> ro = {0, 0, 0}, rd = {0, 0, 1};
    double thr_r = 1, thr_g = 1, thr_b = 1;

    const T w_div = (T)(float)P.width;    // f32_image_width  (src/render.jl:16)
    const T h_div = (T)(float)P.height;   // f32_image_height (src/render.jl:17)

    PhaseClock<PROFILE> clk;
    for (;;) {
        clk.start();
        // ---- (S) closest hit over the whole sphere list (src/hit.jl:38-50) ----
        T t_hit = 0;
        int idx = -1;
        if constexpr (MFMA && CULL) {
            if (__any(has_ray)) {
                const MfmaCull mc = mfma_cull_of(cull);
                if (LDS_SCENE) idx = hit_world_mfma<T>(scene, (const V4 *)lds_geom, ro, rd, has_ray, (T)1e-4, t_hit, ws, lane, clk, &mc, (const unsigned short *)lds_orig);
                else idx = hit_world_mfma<T>(scene, cull.exact, ro, rd, has_ray, (T)1e-4, t_hit, ws, lane, clk, &mc, cull.orig);
            }
        } else if constexpr (MFMA) {
            if (__any(has_ray)) {
                if (LDS_SCENE) idx = hit_world_mfma<T>(scene, (const V4 *)lds_geom, ro, rd, has_ray, (T)1e-4, t_hit, ws, lane, clk);
                else idx = hit_world_mfma<T>(scene, scene.geom, ro, rd, has_ray, (T)1e-4, t_hit, ws, lane, clk);
            }
        } else if (has_ray) {
            if (CULL && LDS_SCENE)
                idx = hit_world_cull<T, 256>(cull, (const V4 *)lds_geom, (const unsigned short *)lds_orig, ro, rd, (T)1e-4,
                                             (T)__builtin_huge_val(), t_hit, my_list, clk);
            else if (CULL)
                idx = hit_world_cull<T, 256>(cull, cull.exact, cull.orig, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
            else if (LDS_SCENE)
                idx = hit_world<T, 256>(scene, (const V4 *)lds_geom, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
            else
                idx = hit_world<T, 256>(scene, scene.geom, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
        }
        n_segments += (unsigned long long)__popcll(__ballot(has_ray));
        clk.lap(2);

        // ---- (H1) a miss ends the sample: its radiance thr * sky (src/ray_color.jl:36) is added EXACTLY
        //      to the pixel's accumulators in LDS (a path that runs out of depth adds 0: nothing to do) ----
        const bool hit = has_ray && idx >= 0;
        if constexpr (MFMA) {
            // About a quarter of the lanes end a path per iteration, and each has three channels to convert and add: instead
            // of three rounds at ~25 % lane utilisation the (lane, channel) tasks are dealt to ALL lanes through the wave's
            // candidate list area in LDS (free between two scans): one round for up to 21 paths.
            const bool miss = has_ray && idx < 0;
            const unsigned long long miss_mask = __ballot(miss);
            if (miss_mask) {
                unsigned char *scr = reinterpret_cast<unsigned char *>(ws.pairs);
                double *task_val = reinterpret_cast<double *>(scr);                           // 192 x 8 B
                unsigned short *task_acc = reinterpret_cast<unsigned short *>(scr + 1536);    // 192 x 2 B: LDS offset of the pixel's accumulators | channel
                if (miss) {
                    const C3 sky = skycolor(rd);
                    const unsigned k3 = 3u * __builtin_amdgcn_mbcnt_hi((unsigned)(miss_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)miss_mask, 0u));
                    const unsigned item_ref = ref_depth & RTW_REF_MASK;
                    const unsigned a0 = (unsigned)(reinterpret_cast<unsigned char *>(sh->slot(item_ref >> 4, P.slot_stride)->acc(item_ref & 15u)) - smem);
                    task_val[k3] = thr_r * sky.r; task_val[k3 + 1u] = thr_g * sky.g; task_val[k3 + 2u] = thr_b * sky.b;
                    task_acc[k3] = (unsigned short)a0; task_acc[k3 + 1u] = (unsigned short)(a0 | 1u); task_acc[k3 + 2u] = (unsigned short)(a0 | 2u);
                }
                __builtin_amdgcn_wave_barrier();
                const unsigned n3 = 3u * (unsigned)__popcll(miss_mask);
                for (unsigned t0 = 0; t0 < n3; t0 += 64u) {
                    const unsigned t = t0 + lane;
                    if (t < n3) {
                        const unsigned ac = task_acc[t];
                        fx_accumulate_channel(reinterpret_cast<unsigned long long *>(smem + (ac & 0xfff8u)), ac & 7u, task_val[t]);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        } else if (has_ray && idx < 0) {
            const C3 sky = skycolor(rd);
            const unsigned item_ref = ref_depth & RTW_REF_MASK;
            fx_accumulate(sh->slot(item_ref >> 4, P.slot_stride)->acc(item_ref & 15u), thr_r * sky.r, thr_g * sky.g, thr_b * sky.b);
        }
        has_ray = false;

        // ---- (A) lanes whose chunk is done retire it, finished jobs are stored, new items are taken ----
        const bool need = alive && !hit && samples_left == 0;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask) {
            bool last = false;
            if (need && have_item) {
                last = __hip_atomic_fetch_add(&sh->slot((ref_depth & RTW_REF_MASK) >> 4, P.slot_stride)->remaining, -1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 1;
                have_item = false;
            }
            // jobs whose last item just finished: the wave stores their 16 pixels and frees the slot
            unsigned long long fin = __ballot(last);
            while (fin) {
                const int L = __builtin_ctzll(fin);
                fin &= fin - 1ull;
                JobSlot *S = sh->slot(uniform((unsigned)__shfl((int)((ref_depth & RTW_REF_MASK) >> 4), L)), P.slot_stride);
                store_job<T>(P, S, lane, out);
                __hip_atomic_store(&S->ready_seq, RTW_SLOT_FREE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // a wave without unassigned items draws a batch ticket and tries to make it usable
            if (pool_next >= pool_end) {
                if (!have_ticket) {
                    unsigned t = 0;
                    if (lane == 0) t = __hip_atomic_fetch_add(&sh->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    t = uniform(t);
                    tk_seq = udiv_magic(t, P.div_bpj_m, P.div_bpj_s);
                    tk_b = t - tk_seq * P.bpj;
                    have_ticket = true;
                }
                const unsigned sl = tk_seq - udiv_magic(tk_seq, P.div_slots_m, P.div_slots_s) * P.n_slots;   // tk_seq mod n_slots
                JobSlot *S = sh->slot(sl, P.slot_stride);
                unsigned rs = uniform(__hip_atomic_load(&S->ready_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                if (rs == RTW_SLOT_FREE && tk_b == 0u) {
                    // this wave opens the job: claim the slot, take a job from the global queue, zero the accumulators
                    unsigned won = 0;
                    if (lane == 0) {
                        unsigned expect = RTW_SLOT_FREE;
                        won = __hip_atomic_compare_exchange_strong(&S->ready_seq, &expect, RTW_SLOT_OPENING, __ATOMIC_ACQUIRE,
                                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                    }
                    if (uniform(won)) {
                        open_job(P, S, lane, ctr);
                        __hip_atomic_store(&S->ready_seq, tk_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        rs = tk_seq;
                    }
                }
                if (rs < RTW_SLOT_OPENING) {
                    if (uniform(S->job) == RTW_JOB_EOF) {
                        // the global queue is exhausted (this slot stays marked for good): these lanes are done
                        if (need) alive = false;
                        have_ticket = false;
                    } else if (rs == tk_seq) {
                        pool_slot = sl; pool_b = tk_b;
                        pool_next = 0; pool_end = 64;
                        have_ticket = false;
                        if constexpr (MFMA) {
                            // Lane l prepares item l of the batch: setting up a stream is 2 splitmix64 + 1 step (~60 VALU), and a
                            // wave takes items a few lanes at a time -- almost every iteration for ~6 % of its lanes.
                            const unsigned px = lane & ((1u << P.job_shift) - 1u), chunk = tk_b * (64u >> P.job_shift) + (lane >> P.job_shift);
                            const unsigned side = P.job_shift >> 1;
                            const int i0 = S->i_base + (int)(px & ((1u << side) - 1u)), j0 = S->j_base + (int)(px >> side);
                            Rng r0;
                            rng_stream(P.seed, (unsigned long long)j0 * (unsigned)P.height + (unsigned)i0, chunk, r0);
                            pool_rng[lane] = ulonglong2{r0.x, r0.y};
                            pool_valid = __ballot((int)chunk < P.n_chunks && ((S->valid >> px) & 1u));
                            __builtin_amdgcn_wave_barrier();
                        }
                    }
                }
            }
            // hand out items of the wave's batch: item p = (pixel p mod job_px, chunk (64 / job_px) b + p / job_px)
            const unsigned long long take_mask = __ballot(need && alive);
            if (take_mask && pool_next < pool_end) {
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(take_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)take_mask, 0u));   // takers below this lane
                const unsigned p = pool_next + rank;
                if (need && alive && p < pool_end) {
                    const JobSlot *S = sh->slot(pool_slot, P.slot_stride);
                    const unsigned px = p & ((1u << P.job_shift) - 1u), chunk = pool_b * (64u >> P.job_shift) + (p >> P.job_shift);
                    bool real;
                    if constexpr (MFMA) real = ((pool_valid >> p) & 1ull) != 0ull;
                    else real = (int)chunk < P.n_chunks && ((S->valid >> px) & 1u);
                    if (real) {
                        if constexpr (MFMA) {
                            const ulonglong2 st = pool_rng[p];
                            rng.x = st.x; rng.y = st.y;
                        } else {
                            const unsigned side = P.job_shift >> 1;
                            const int i0 = S->i_base + (int)(px & ((1u << side) - 1u)), j0 = S->j_base + (int)(px >> side);
                            const unsigned long long pix = (unsigned long long)j0 * (unsigned)P.height + (unsigned)i0;
                            rng_stream(P.seed, pix, chunk, rng);
                        }
                        const int s0 = (int)chunk * P.chunk_spp;
                        samples_left = min(P.spp, s0 + P.chunk_spp) - s0;
                        jitter = s0 != 0;                                             // sample 1 of the pixel is centred
                        ref_depth = pool_slot * 16u + px;
                        have_item = true;
                    }
                    // padding item (chunk beyond n_chunks, pixel outside the image): nothing to do, pull again
                }
                pool_next = min(pool_end, pool_next + (unsigned)__popcll(take_mask));
            }
        }
        if (!__any(alive)) break;
        clk.lap(0);

        // ---- (H2) a hit starts the scatter (src/ray_color.jl:20-33) ----
        int todo = PATH_READY;    // PATH_BALL: scatter waits for a unit-ball sample; PATH_NORM: direction to normalise
        V3<T> vec = {0, 0, 0};    // PATH_BALL: n (Lambertian) / reflect(d, n) (Metal); PATH_NORM: the raw direction
        T vscale_ = 1;            // PATH_BALL: 1 (Lambertian) / fuzz (Metal)
        int kind = 0;
        if (hit) {
            const V4 g = CULL ? cull.exact[idx] : scene.geom[idx];    // CULL: device order
            const V4 m0 = CULL ? cull.mat0[idx] : scene.mat0[idx];
            const V4 m1 = CULL ? cull.mat1[idx] : scene.mat1[idx];
            HitRec<T> rec;
            make_hitrec<T>({g.x, g.y, g.z}, m0.x, ro, rd, t_hit, rec);
            kind = (int)m0.z;
            todo = scatter_begin<T>(rng, kind, m0.y, rd, rec, vec, vscale_);
            const V3<T> att = attenuation_of<T>(kind, {m1.x, m1.y, m1.z});
            thr_r = thr_r * (double)att.x; thr_g = thr_g * (double)att.y; thr_b = thr_b * (double)att.z;
            ro = rec.p;
            ref_depth -= 1u << RTW_REF_BITS;                                    // one bounce used
            if (todo == PATH_READY) { rd = vec; has_ray = ref_depth > RTW_REF_MASK; }   // depth 0: ray_color returns 0
        }
        clk.lap(3);

        // ---- (B) start the next sample (src/render.jl:29-37): jitter, then the lens disk in (R) ----
        bool new_sample = false;  // camera ray under construction (waits for a unit-disk sample)
        T su = 0, sv = 0;         // (u + du, v + dv)
        if (alive && !hit && samples_left > 0) {
            T du = 0, dv = 0;
            if (jitter) {
                T r1, r2;
                trand(rng, r1); du = r1 / w_div;
                trand(rng, r2); dv = r2 / h_div;
            }
            const JobSlot *S = sh->slot((ref_depth & RTW_REF_MASK) >> 4, P.slot_stride);
            const unsigned px = ref_depth & 15u;
            const unsigned side = P.job_shift >> 1;
            su = (T)S->uv[px >> side] + du;                            // T(j / W) + du,       src/render.jl:26,37
            sv = (T)S->uv[4 + (px & ((1u << side) - 1u))] + dv;        // T((H - i) / H) + dv, src/render.jl:27,37
            new_sample = true;
            jitter = true;
            samples_left -= 1;
        }
        n_samples += (unsigned long long)__popcll(__ballot(new_sample));

        // ---- (R) ONE rejection loop for every lane that needs a random point: unit ball for
        //      Lambertian / Metal scatter (src/rand.jl:15-22), unit disk for the lens (:31-38) ----
        const bool ball = todo == PATH_BALL;
        V3<T> rp = {0, 0, 0};
        T len2 = 0;
        {
            bool pending = ball || new_sample;
#ifdef RTW_DUP_REJECT    // instruction-count probe: the rejection loop twice (the first run on a copy of the generator)
            { Rng r2 = rng; V3<T> q2 = {0, 0, 0}; T l2 = 0; bool p2 = pending;
              while (__any(p2)) { if (p2) { l2 = reject_trial<T>(r2, ball, q2); p2 = !(l2 <= T(1)); } }
              __asm__ volatile("" :: "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(l2), "v"((unsigned)r2.x), "v"((unsigned)r2.y)); }
#endif
            while (pending) {
                len2 = reject_trial<T>(rng, ball, rp);
                pending = !(len2 <= T(1));
            }
        }
        // ---- (F) finish the scatter / the camera ray; ONE normalize for all of them ----
        if (ball) {
            todo = scatter_finish<T>(kind, vec, vscale_, rp, len2, vec);
            if (todo == PATH_READY) rd = vec;                     // degenerate Lambertian direction: n, as is
        }
        if (new_sample) {
            __asm__ volatile("" ::: "memory");                    // (keeps the camera loads inside this branch)
            const Camera<T> cam = sh->cam;
            camera_ray_raw<T>(cam, su, sv, rp.x, rp.y, ro, vec);   // src/camera.jl:43-48
            todo = PATH_NORM;
            thr_r = thr_g = thr_b = 1.0;
            ref_depth = (ref_depth & RTW_REF_MASK) | ((unsigned)P.max_depth << RTW_REF_BITS);
        }
        if (todo == PATH_NORM) rd = normalize(vec);
        if (ball || new_sample || todo == PATH_NORM) has_ray = ref_depth > RTW_REF_MASK;   // depth <= 0: ray_color returns 0 (src/ray_color.jl:15)
        clk.lap(1);
        if (!__any(has_ray)) __builtin_amdgcn_s_sleep(2);    // every lane waits for a job slot: do not hammer LDS
    }

    if (PROFILE && lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&ctr->phase[k], clk.acc[k]);
    }
    if (lane == 0) {
        atomicAdd(&ctr->segments, n_segments);
        atomicAdd(&ctr->samples, n_samples);
        const unsigned long long t_wave_end = wall_clock64();
        atomicMin(&ctr->t_first, t_wave_start);
        atomicMax(&ctr->t_last, t_wave_end);
        atomicAdd(&ctr->t_end_sum, t_wave_end);
        atomicAdd(&ctr->n_waves, 1ull);
        const unsigned long long t0 = __hip_atomic_load(&ctr->t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long bin = (t_wave_end - t0) / 25000ull;
        atomicAdd(&ctr->end_hist[bin < 4095ull ? (unsigned)bin : 4095u], 1u);
    }
}

}  // namespace rtw
