// rtw_pool.hpp -- the ray-pool form of the trace kernel (round 4): the same path as rtw_kernels.hpp's trace_kernel
// (render -> ray_color -> hit/scatter, src/render.jl:29-38 -> src/ray_color.jl:14-38), same streams, same exact pixel sums, same
// image bit for bit -- but a ray no longer belongs to a lane.
//
// Why.  In trace_kernel a lane drags its ray through scan -> shade -> rejection loop -> finish in one loop, and everything that is
// not the scan runs with whatever lanes happen to need it: the rejection loop at 23 % lane utilisation (a wave waits ~9 trials for
// its slowest lane), the dielectric branch at 5 %, the camera branch at 25 %, the item pull at 6 % -- 71 % of the kernel's VALU
// instructions at ~35 % utilisation (profiles/r03_probe_phases.txt).  Re-batching inside a wave cannot fix that (a lane that sits
// out a scan wastes as much as it saves).  Here the rays of a WORKGROUP live in LDS:
//   slot   = one virtual lane: a ray + its generator + its throughput + the item it works on, 80 B (Float32) in LDS.  R slots
//            per workgroup (R > 64 W: the slots that no wave holds are parked in the stage queues below).
//   queue  = ring of slot numbers per stage (LDS, multi-producer / multi-consumer, generation-stamped entries):
//              SCAN  rays that need the closest hit            LM    hits on a Lambertian / Metal sphere
//              END   paths that ended (miss: add thr * sky)    DIEL  hits on a Dielectric sphere
//                    and slots that need a sample / an item    REJ   unit-ball samples still pending after the inline trials
//              WAIT  slots that need an item while every job slot of the workgroup is taken (small renders): served only when
//                    nothing else is queued -- the slots that hold the items of those jobs are in the other queues
//   wave   = picks the fullest queue, takes up to 64 slots of ONE kind and runs that stage for them on all its lanes:
//            the scan always sees 64 rays, the Lambertian/Metal stage shades 64 hits and runs its first rejection trials at
//            100 % / 48 % utilisation (stragglers go to REJ, where they meet the stragglers of other batches), dielectrics are
//            shaded 64 at a time, path ends are accumulated 64 at a time and start their next sample together.
// Results do not depend on which wave runs what: every slot carries its own stream, the pixel sums are exact integers.
#pragma once
#include "rtw_kernels.hpp"

namespace rtw {

enum { PQ_SCAN = 0, PQ_LM = 1, PQ_END = 2, PQ_DIEL = 3, PQ_REJ = 4, PQ_WAIT = 5, PQ_COUNT = 6 };
enum { PC_DEAD = 12, PC_BUSY = 13, PC_FREE = 14, PC_EOF = 15 };
#define RTW_POOL_RING 2048u          // entries per queue ring (u16: slot | generation << 11); any queue can hold every slot
#define RTW_POOL_RING_SHIFT 11
#define RTW_POOL_PAIR_CAP 256u       // candidate list entries per wave (trace_kernel: 512)
#ifndef RTW_POOL_LM_ROUNDS
#define RTW_POOL_LM_ROUNDS 3         // unit-ball trials run inline by the Lambertian/Metal stage (100 %, 48 %, 23 % utilisation; 2: +2.5 %, 6: +1 %, 12: +5 % time)
#endif
#ifndef RTW_POOL_REJ_ROUNDS
#define RTW_POOL_REJ_ROUNDS 3        // trials per visit of the straggler stage
#endif
// the `misc` word of a slot
#define RTW_PM_SAMPLES 0x0fffffffu   // samples of the item still to start
#define RTW_PM_JITTER (1u << 28)     // false only for sample 1 of the pixel (src/render.jl:30-31)
#define RTW_PM_ITEM (1u << 29)       // owns an item (its job's `remaining` is decremented when the chunk is done)
#define RTW_PM_METAL (1u << 30)      // REJ: the pending scatter is a Metal's (else Lambertian)
#define RTW_POOL_MAX_CHUNK_SPP 0x0fffffff
#ifndef RTW_POOL_W
#define RTW_POOL_W 16                // waves per workgroup: one workgroup per CU, 4 waves per SIMD
#endif
#ifndef RTW_POOL_R
#define RTW_POOL_R 1216              // slots per workgroup: 1024 in the waves' hands + 192 parked
#endif
#ifndef RTW_POOL_THR_PACK                     // ready thresholds, one byte per queue: SCAN | LM << 8 | END << 16 | DIEL << 24 | REJ << 32
#define RTW_POOL_THR_PACK (64ull | 64ull << 8 | 48ull << 16 | 16ull << 24 | 32ull << 32 | 1ull << 40)
#endif
#ifndef RTW_POOL_POLL
#define RTW_POOL_POLL 4                       // s_sleep argument of a wave that waits for a queue to become ready (x 64 cycles)
#endif
#ifndef RTW_POOL_MIN_BUSY
#define RTW_POOL_MIN_BUSY (RTW_POOL_W / 4)    // a partial batch is taken only while fewer waves than this hold one
#endif

// Slot records.  Float32 (80 B): [0] o.xyz t  [16] d.xyz w  [32] rng  [48] thr_r thr_g  [64] thr_b ref_depth misc
//               Float64 (112 B): [0] o.xy  [16] o.z t  [32] d.xy  [48] d.z w  [64] rng  [80] thr_r thr_g  [96] thr_b ref_depth misc
// w = the sphere hit (SCAN -> LM / DIEL), -1 = miss (SCAN -> END: add thr * sky), -2 = nothing to add (END), or the Metal's fuzz (REJ;
// d then holds the scatter's base vector).  ref_depth = bounces left << 9 | job slot << 4 | pixel of the job (as in trace_kernel).
template <typename T> struct PoolRec;
template <> struct PoolRec<float> {
    static constexpr unsigned BYTES = 80, O = 0, D = 16, RNG = 32, THR = 48, TAIL = 64, T_OFF = 12, W_OFF = 28;
    static __device__ __forceinline__ void ld_o(const unsigned char *r, V3<float> &o, float &t) { const float4 v = *reinterpret_cast<const float4 *>(r + O); o = {v.x, v.y, v.z}; t = v.w; }
    static __device__ __forceinline__ void ld_d(const unsigned char *r, V3<float> &d, float &w) { const float4 v = *reinterpret_cast<const float4 *>(r + D); d = {v.x, v.y, v.z}; w = v.w; }
    static __device__ __forceinline__ void st_o(unsigned char *r, V3<float> o, float t) { *reinterpret_cast<float4 *>(r + O) = float4{o.x, o.y, o.z, t}; }
    static __device__ __forceinline__ void st_d(unsigned char *r, V3<float> d, float w) { *reinterpret_cast<float4 *>(r + D) = float4{d.x, d.y, d.z, w}; }
    static __device__ __forceinline__ float w_of(int i) { return __int_as_float(i); }
    static __device__ __forceinline__ int i_of(float w) { return __float_as_int(w); }
};
template <> struct PoolRec<double> {
    static constexpr unsigned BYTES = 112, O = 0, D = 32, RNG = 64, THR = 80, TAIL = 96, T_OFF = 24, W_OFF = 56;
    static __device__ __forceinline__ void ld_o(const unsigned char *r, V3<double> &o, double &t) { const double2 a = *reinterpret_cast<const double2 *>(r + O), b = *reinterpret_cast<const double2 *>(r + O + 16); o = {a.x, a.y, b.x}; t = b.y; }
    static __device__ __forceinline__ void ld_d(const unsigned char *r, V3<double> &d, double &w) { const double2 a = *reinterpret_cast<const double2 *>(r + D), b = *reinterpret_cast<const double2 *>(r + D + 16); d = {a.x, a.y, b.x}; w = b.y; }
    static __device__ __forceinline__ void st_o(unsigned char *r, V3<double> o, double t) { *reinterpret_cast<double2 *>(r + O) = double2{o.x, o.y}; *reinterpret_cast<double2 *>(r + O + 16) = double2{o.z, t}; }
    static __device__ __forceinline__ void st_d(unsigned char *r, V3<double> d, double w) { *reinterpret_cast<double2 *>(r + D) = double2{d.x, d.y}; *reinterpret_cast<double2 *>(r + D + 16) = double2{d.z, w}; }
    static __device__ __forceinline__ double w_of(int i) { return __longlong_as_double((long long)i); }
    static __device__ __forceinline__ int i_of(double w) { return (int)__double_as_longlong(w); }
};

// The item dispenser: trace_kernel's per-wave batch (job slot -> 64 items at a time) as ONE structure of the workgroup, used under
// `lock` by the wave whose END batch needs items (a wave that held a private batch could starve the job it belongs to: it hands
// items out only when it happens to run the END stage).  There are no tickets: one job is dispensed at a time, batch after batch,
// and the next job opens in ANY free job slot -- with trace_kernel's slot = sequence mod n_slots one slow job (a 50-bounce path
// inside the glass sphere) blocks the dispenser as soon as the sequence wraps, and 1216 slots go through 12 jobs quickly.
struct PoolDisp {
    unsigned lock, next, end, slot, b, have_job;
    unsigned long long valid;            // which of the batch's 64 items are real (chunk < n_chunks, pixel inside the image)
};
template <typename T, int W> struct PoolShared {
    unsigned char slots[RTW_SLOT_BYTES];                 // job slots, as in WgShared
    __device__ __forceinline__ JobSlot *slot(unsigned i, unsigned stride) { return reinterpret_cast<JobSlot *>(slots + i * stride); }
    unsigned ctl[16];                                    // what a wave looks at to choose its next batch, ONE 64-byte read:
                                                         //   [2q], [2q + 1] head and tail of queue q (free-running positions)
                                                         //   [12] dead: slots that found the job queues exhausted -- the kernel ends at R
                                                         //   [13] busy: waves that hold a batch right now
                                                         //   [14] free job slots   [15] eof: the job queues are exhausted
    unsigned fin_waves;
    unsigned long long fin_segments, fin_samples;
    PoolDisp disp;
    unsigned long long prof[32];                         // PROFILE instantiation: per stage (batches, slots, wave-cycles); [24] idle, [25] lost pops, [26] wave-cycles
    JobCache jobs;
    Camera<T> cam;
    KParams P;
};
template <typename T, int W, int R> __host__ __device__ constexpr size_t pool_fixed_lds_bytes() {
    return (size_t)W * RTW_POOL_PAIR_CAP * 4 + (size_t)W * 64 * 8 + (sizeof(T) == 8 ? (size_t)W * 64 * 4 : 0) +
           (sizeof(PoolShared<T, W>) + 15) / 16 * 16 + (size_t)PQ_COUNT * RTW_POOL_RING * 2 + (size_t)R * PoolRec<T>::BYTES;
}
// + the scene copy: geom (n_alloc x V4) and one material-kind byte per sphere
template <typename T> __host__ __device__ inline size_t pool_scene_lds_bytes(int n, int n_pad) {
    const size_t n_alloc = (size_t)scene_geom_alloc(n, n_pad);
    return n_alloc * sizeof(typename Vec4<T>::type) + (n_alloc + 15) / 16 * 16;
}

// fx_accumulate with the three channels in flight together (the END stage adds 64 samples at a time: three dependent
// LDS round trips per batch instead of six)
__device__ __forceinline__ void fx_accumulate3(unsigned long long *a, double r, double g, double b) {
    unsigned long long lo[3], hi[3], old[3] = {0ull, 0ull, 0ull};
    bool ok[3];
    ok[0] = fx_from_double(r, lo[0], hi[0]); ok[1] = fx_from_double(g, lo[1], hi[1]); ok[2] = fx_from_double(b, lo[2], hi[2]);
#pragma unroll
    for (int c = 0; c < 3; ++c) if (ok[c]) old[c] = __hip_atomic_fetch_add(&a[2 * c], lo[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (ok[c]) {
            const unsigned long long h = hi[c] + ((old[c] + lo[c] < old[c]) ? 1ull : 0ull);
            if (h) __hip_atomic_fetch_add(&a[2 * c + 1], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(&a[6], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
}

template <typename T, int W, int R, bool PROFILE>
__global__ __launch_bounds__(W * 64, (W + 3) / 4) void trace_pool_kernel(KParams P_arg, Camera<T> cam_arg, DevScene<T> scene, T *__restrict__ out, DevCounters *ctr) {
    using V4 = typename Vec4<T>::type;
    using L = PoolRec<T>;
    static_assert(R > 64 * W && R < 2048, "the pool holds more slots than the waves can take, and a slot number has 11 bits");
    const unsigned lane = lane_id();
    const unsigned wv = threadIdx.x >> 6;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr size_t pairs_bytes = (size_t)W * RTW_POOL_PAIR_CAP * 4;
    constexpr size_t keys_bytes = (size_t)W * 64 * 8 + (sizeof(T) == 8 ? (size_t)W * 64 * 4 : 0);
    constexpr size_t shared_bytes = (sizeof(PoolShared<T, W>) + 15) / 16 * 16;
    constexpr size_t ring_bytes = (size_t)PQ_COUNT * RTW_POOL_RING * 2;
    PoolShared<T, W> *sh = reinterpret_cast<PoolShared<T, W> *>(smem + pairs_bytes + keys_bytes);
    unsigned short *rings = reinterpret_cast<unsigned short *>(smem + pairs_bytes + keys_bytes + shared_bytes);
    unsigned char *recs = smem + pairs_bytes + keys_bytes + shared_bytes + ring_bytes;
    V4 *lds_geom = reinterpret_cast<V4 *>(recs + (size_t)R * L::BYTES);
    const int n_alloc = scene_geom_alloc(scene.n, scene.n_pad);
    unsigned char *lds_kind = reinterpret_cast<unsigned char *>(lds_geom + n_alloc);
    WaveScratch ws;
    ws.pairs = reinterpret_cast<unsigned *>(smem) + wv * RTW_POOL_PAIR_CAP;
    ws.keys = reinterpret_cast<unsigned long long *>(smem + pairs_bytes) + wv * 64;
    ws.kidx = reinterpret_cast<unsigned *>(smem + pairs_bytes + (size_t)W * 64 * 8) + wv * 64;
    ws.cap = 80; ws.cap2 = RTW_POOL_PAIR_CAP - 160;          // (entries of two words + single candidates)

    // ---- set-up: every slot starts in END, "nothing to add, needs an item" ----
    for (unsigned i = threadIdx.x; i < PQ_COUNT * RTW_POOL_RING; i += W * 64) {
        const unsigned k = i - PQ_END * RTW_POOL_RING;
        rings[i] = (unsigned short)(k < (unsigned)R ? k : 0xffffu);      // (generation 0 | slot k), else generation 31: stale
    }
    for (unsigned i = threadIdx.x; i < (unsigned)R; i += W * 64) {
        unsigned char *r = recs + i * L::BYTES;
        L::st_o(r, V3<T>{0, 0, 0}, T(0));
        L::st_d(r, V3<T>{0, 0, 1}, L::w_of(-2));
        *reinterpret_cast<ulonglong2 *>(r + L::RNG) = ulonglong2{1ull, 2ull};
        *reinterpret_cast<double2 *>(r + L::THR) = double2{1.0, 1.0};
        *reinterpret_cast<double *>(r + L::TAIL) = 1.0;
        *reinterpret_cast<uint2 *>(r + L::TAIL + 8) = uint2{0u, 0u};
    }
    if (threadIdx.x < P_arg.n_slots) { JobSlot *S0 = sh->slot(threadIdx.x, P_arg.slot_stride); S0->ready_seq = RTW_SLOT_FREE; S0->job = 0u; }
    if (threadIdx.x < 16) sh->ctl[threadIdx.x] = threadIdx.x == 2 * PQ_END + 1 ? (unsigned)R : threadIdx.x == PC_FREE ? P_arg.n_slots : 0u;
    if (threadIdx.x == 0) {
        sh->fin_waves = 0u; sh->fin_segments = 0ull; sh->fin_samples = 0ull;
        sh->disp.lock = 0u; sh->disp.next = 0u; sh->disp.end = 0u; sh->disp.slot = 0u; sh->disp.b = 0u;
        sh->disp.have_job = 0u; sh->disp.valid = 0ull;
        sh->jobs.jc = 0ull; sh->jobs.jc_lock = 0u; sh->jobs.queue_off = 0u; sh->jobs.last_g = 0u; sh->jobs.static_used = 0u;
        sh->cam = cam_arg; sh->P = P_arg;
    }
    if (PROFILE && threadIdx.x < 32) sh->prof[threadIdx.x] = 0ull;
    const KParams &P = sh->P;
    stage_scene<T>(scene, lds_geom);
    for (int i = threadIdx.x; i < n_alloc; i += W * 64) lds_kind[i] = (unsigned char)(i < scene.n ? (int)scene.mat0[i].z : 0);
    __syncthreads();

    unsigned long long n_segments = 0, n_samples = 0;
    const T w_div = (T)(float)P.width;    // f32_image_width  (src/render.jl:16)
    const T h_div = (T)(float)P.height;   // f32_image_height (src/render.jl:17)
    PhaseClock<PROFILE> clk;
    unsigned long long t_kernel = 0;
    if (PROFILE) t_kernel = __builtin_readcyclecounter();

#ifdef RTW_POOL_WATCHDOG
    unsigned wd_iter = 0, wd_q = 99, wd_n = 0;
#endif
    for (;;) {
#ifdef RTW_POOL_WATCHDOG
        if (++wd_iter > (unsigned)RTW_POOL_WATCHDOG) {      // debug builds: dump the pool's state once and let the kernel end
            if (lane == 0 && atomicCAS(&ctr->end_hist[0], 0u, 0xdeadbeefu) == 0u) {
                unsigned *d = ctr->end_hist + 1;
                d[0] = blockIdx.x; d[1] = wv; d[2] = wd_q; d[3] = wd_n; d[4] = sh->ctl[PC_DEAD];
                for (int k = 0; k < 16; ++k) d[5 + k] = sh->ctl[k];
                const unsigned *pd = reinterpret_cast<const unsigned *>(&sh->disp);
                for (int k = 0; k < 8; ++k) d[24 + k] = pd[k];
                for (unsigned k = 0; k < P.n_slots && k < 24u; ++k) { const JobSlot *S = sh->slot(k, P.slot_stride); d[36 + 3 * k] = S->ready_seq; d[37 + 3 * k] = S->job; d[38 + 3 * k] = (unsigned)S->remaining; }
                d[110] = sh->jobs.queue_off; d[111] = (unsigned)n_segments; d[112] = (unsigned)n_samples;
            }
            __hip_atomic_store(&sh->ctl[PC_DEAD], (unsigned)R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
#endif
        unsigned long long t_top = 0;
        if (PROFILE) t_top = __builtin_readcyclecounter();
        // ---- choose a stage: a queue that fills a wave (the shading stages first: the scan queue is where rays park), else the fullest ----
        // lane l < 16 reads ctl[l]; lane 2q + 1 then holds queue q's tail and, one lane down, its head
        unsigned cv = 0;
        if (lane < 16) cv = __hip_atomic_load(&sh->ctl[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned cnt = cv - (unsigned)__builtin_amdgcn_update_dpp(0, (int)cv, 0x111 /* row_shr:1 */, 0xf, 0xf, false);
        if ((unsigned)__builtin_amdgcn_readlane((int)cv, PC_DEAD) >= (unsigned)R) break;
        // waiting slots can be served once a job slot is free (or the job queues are exhausted: they are done)
        const bool wait_open = ((unsigned)__builtin_amdgcn_readlane((int)cv, PC_FREE) | (unsigned)__builtin_amdgcn_readlane((int)cv, PC_EOF)) != 0u;
        // a queue is READY when it holds its threshold: 64 for the scan (a partial scan costs a whole one), fewer for the cheap stages
        // with a thin flow (dielectrics are 6 % of the hits: waiting for 64 of them parks slots the scan queue needs)
        const unsigned thr = (unsigned)((RTW_POOL_THR_PACK >> (4u * (lane & 14u))) & 0xffull);        // (8 bits per queue: lane >> 1)
        const unsigned readym = (unsigned)__ballot((lane & 1u) && lane < 2u * PQ_COUNT && cnt >= thr && (lane != 2u * PQ_WAIT + 1u || wait_open));
        unsigned q;
        if (readym) {
            // the shading queues first (the scan queue is where rays park); odd waves look at them from the other end, so that
            // waves that look at the same moment do not all reach for the same batch
            const unsigned shading = readym & ~2u;
            q = !shading ? 0u : (wv & 1u) ? (31u - (unsigned)__builtin_clz(shading)) >> 1 : (unsigned)__builtin_ctz(shading) >> 1;
        } else {
            // No queue is ready.  While enough other waves hold batches, WAIT for them to push: a partial batch costs a stage its
            // full instruction count and, taken greedily, keeps every queue short for good (measured: scans at 42 - 51 of 64 lanes).
            // Only when few waves hold work -- the end of the frame, small renders -- take the fullest queue.
            const unsigned anym = (unsigned)__ballot((lane & 1u) && lane < 2u * PQ_COUNT && cnt != 0u && (lane != 2u * PQ_WAIT + 1u || wait_open));
            if (!anym || (unsigned)__builtin_amdgcn_readlane((int)cv, PC_BUSY) >= RTW_POOL_MIN_BUSY) {
                __builtin_amdgcn_s_sleep(RTW_POOL_POLL);
                if (PROFILE && lane == 0) __hip_atomic_fetch_add(&sh->prof[24], __builtin_readcyclecounter() - t_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
            q = 0u; unsigned best = 0u;
#pragma unroll
            for (int k = 0; k < PQ_COUNT; ++k) { const unsigned c = (anym >> (2 * k + 1)) & 1u ? (unsigned)__builtin_amdgcn_readlane((int)cnt, 2 * k + 1) : 0u; if (c > best) { best = c; q = (unsigned)k; } }
        }
        unsigned n = (unsigned)__builtin_amdgcn_readlane((int)cnt, (int)(2u * q + 1u));
        n = n > 64u ? 64u : n;
        const unsigned h0 = (unsigned)__builtin_amdgcn_readlane((int)cv, (int)(2u * q));
        unsigned ok = 0;
        if (lane == 0) {
            unsigned expect = h0;
            ok = __hip_atomic_compare_exchange_strong(&sh->ctl[2u * q], &expect, h0 + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
        }
        if (!uniform(ok)) {                                                     // another wave took them
            if (PROFILE && lane == 0) __hip_atomic_fetch_add(&sh->prof[25], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (lane == 0) __hip_atomic_fetch_add(&sh->ctl[PC_BUSY], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef RTW_POOL_WATCHDOG
        wd_q = q; wd_n = n;
#endif
        const bool valid = lane < n;
        unsigned id = 0;
        {   // positions h0 .. h0 + n - 1 are reserved; an entry may still be on its way (its producer is between its claim and its store)
            const unsigned pos = h0 + lane, want = (pos >> RTW_POOL_RING_SHIFT) & 31u;
            const unsigned short *cell = rings + q * RTW_POOL_RING + (pos & (RTW_POOL_RING - 1u));
            bool stale = valid;
            unsigned e = 0;
            for (;;) {
                if (stale) { e = __hip_atomic_load(cell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); stale = (e >> RTW_POOL_RING_SHIFT) != want; }
                if (!__any(stale)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            id = valid ? (e & (RTW_POOL_RING - 1u)) : 0u;
        }
        unsigned char *r = recs + id * L::BYTES;
        // Hand the batch's slots on: up to three destination queues, ONE LDS atomic instruction for their tails (lane j claims for
        // destination j), then every slot's number goes to its ring cell, stamped with the cell's generation.  The store is a release:
        // the slot's record is written before a consumer can see its number.
        auto push3 = [&](unsigned qa, bool pa, unsigned qb, bool pb, unsigned qc, bool pc) {
            const unsigned long long ma = __ballot(pa), mb = __ballot(pb), mc = __ballot(pc);
            const unsigned na = (unsigned)__popcll(ma), nb = (unsigned)__popcll(mb), nc = (unsigned)__popcll(mc);
            unsigned pos = 0;
            if (lane < 3) {
                const unsigned add = lane == 0 ? na : lane == 1 ? nb : nc, qq = lane == 0 ? qa : lane == 1 ? qb : qc;
                if (add) pos = __hip_atomic_fetch_add(&sh->ctl[2u * qq + 1u], add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            const unsigned p0 = (unsigned)__builtin_amdgcn_readlane((int)pos, 0), p1 = (unsigned)__builtin_amdgcn_readlane((int)pos, 1), p2 = (unsigned)__builtin_amdgcn_readlane((int)pos, 2);
            if (pa || pb || pc) {
                const unsigned long long m = pa ? ma : pb ? mb : mc;
                const unsigned p = (pa ? p0 : pb ? p1 : p2) + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const unsigned qq = pa ? qa : pb ? qb : qc;
                __hip_atomic_store(rings + qq * RTW_POOL_RING + (p & (RTW_POOL_RING - 1u)), (unsigned short)(id | (((p >> RTW_POOL_RING_SHIFT) & 31u) << RTW_POOL_RING_SHIFT)),
                                   __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        };

        if (q == PQ_SCAN) {
            // ---- closest hit over the whole sphere list (src/hit.jl:38-50) for 64 rays ----
            V3<T> ro, rd; T t_hit, w_;
            L::ld_o(r, ro, t_hit); L::ld_d(r, rd, w_);
            t_hit = 0;
            const int idx = hit_world_mfma<T>(scene, (const V4 *)lds_geom, ro, rd, valid, (T)1e-4, t_hit, ws, lane, clk);
            n_segments += (unsigned long long)n;
            unsigned kind = 0;
            if (valid && idx >= 0) kind = lds_kind[idx];
            if (valid) {
                *reinterpret_cast<T *>(r + L::T_OFF) = t_hit;
                *reinterpret_cast<T *>(r + L::W_OFF) = L::w_of(idx);
            }
            push3(PQ_LM, valid && idx >= 0 && kind != DIELECTRIC, PQ_END, valid && idx < 0, PQ_DIEL, valid && idx >= 0 && kind == DIELECTRIC);
            clk.lap(2);
        } else if (q == PQ_LM || q == PQ_REJ) {
            // ---- Lambertian / Metal scatter (src/material.jl:13-34).  LM: shade the hit, then the first trials of the unit-ball
            //      rejection (src/rand.jl:15-22); REJ: more trials for the samples still pending ----
            V3<T> ro = {0, 0, 0}, rd, vec; T t_hit = 0, w_, vs;
            Rng rng; unsigned ref_depth, misc; bool metal;
            {
                const ulonglong2 g = *reinterpret_cast<const ulonglong2 *>(r + L::RNG); rng.x = g.x; rng.y = g.y;
                const uint2 tl = *reinterpret_cast<const uint2 *>(r + L::TAIL + 8); ref_depth = tl.x; misc = tl.y;
            }
            if (q == PQ_LM) {
                L::ld_o(r, ro, t_hit); L::ld_d(r, rd, w_);
                const int idx = valid ? L::i_of(w_) : 0;
                const V4 g = lds_geom[idx];
                const V4 m0 = scene.mat0[idx];
                const V4 m1 = scene.mat1[idx];
                double2 t01 = *reinterpret_cast<const double2 *>(r + L::THR);
                double t2 = *reinterpret_cast<const double *>(r + L::TAIL);
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, ro, rd, t_hit, rec);
                metal = (int)m0.z == METAL;
                const V3<T> refl = reflect(rd, rec.n);
                vec = metal ? refl : rec.n;                                  // Lambertian: n + u;  Metal: reflect(d, n) + fuzz u
                vs = metal ? m0.y : T(1);
                t01.x = t01.x * (double)m1.x; t01.y = t01.y * (double)m1.y; t2 = t2 * (double)m1.z;      // attenuation = albedo
                ro = rec.p;
                ref_depth -= 1u << RTW_REF_BITS;                             // one bounce used
                if (valid) {
                    *reinterpret_cast<double2 *>(r + L::THR) = t01;
                    *reinterpret_cast<double *>(r + L::TAIL) = t2;
                }
                misc = metal ? (misc | RTW_PM_METAL) : (misc & ~RTW_PM_METAL);
            } else {
                L::ld_d(r, vec, vs);
                metal = (misc & RTW_PM_METAL) != 0u;
            }
            bool pending = valid;
            V3<T> rp = {0, 0, 0}; T len2 = 0;
            constexpr int rounds = RTW_POOL_LM_ROUNDS;      // (REJ uses RTW_POOL_REJ_ROUNDS; same loop)
            const int n_rounds = q == PQ_LM ? rounds : RTW_POOL_REJ_ROUNDS;
            for (int k = 0; k < n_rounds && __any(pending); ++k) {
                if (pending) { len2 = reject_trial<T>(rng, true, rp); pending = !(len2 <= T(1)); }
            }
            const bool done = valid && !pending;
            bool has_ray = false;
            if (done) {
                V3<T> dir;
                const int todo = scatter_finish<T>(metal ? METAL : LAMBERTIAN, vec, vs, rp, len2, dir);
                rd = todo == PATH_NORM ? normalize(dir) : dir;
                has_ray = ref_depth > RTW_REF_MASK;          // depth <= 0: ray_color returns 0 (src/ray_color.jl:15)
            }
            if (valid) {
                if (q == PQ_LM) L::st_o(r, ro, T(0));
                if (done) L::st_d(r, rd, L::w_of(-2)); else L::st_d(r, vec, vs);
                *reinterpret_cast<ulonglong2 *>(r + L::RNG) = ulonglong2{rng.x, rng.y};
                if (q == PQ_LM) *reinterpret_cast<uint2 *>(r + L::TAIL + 8) = uint2{ref_depth, misc};
            }
            push3(PQ_SCAN, done && has_ray, PQ_REJ, valid && !done, PQ_END, done && !has_ray);
            clk.lap(3);
        } else if (q == PQ_DIEL) {
            // ---- Dielectric scatter (src/material.jl:41-53): attenuation 1, at most one draw ----
            V3<T> ro, rd; T t_hit, w_;
            L::ld_o(r, ro, t_hit); L::ld_d(r, rd, w_);
            Rng rng;
            { const ulonglong2 g = *reinterpret_cast<const ulonglong2 *>(r + L::RNG); rng.x = g.x; rng.y = g.y; }
            unsigned ref_depth = *reinterpret_cast<const unsigned *>(r + L::TAIL + 8);
            const int idx = valid ? L::i_of(w_) : 0;
            const V4 g = lds_geom[idx];
            const V4 m0 = scene.mat0[idx];
            const V4 m1 = scene.mat1[idx];
            HitRec<T> rec;
            make_hitrec<T>({g.x, g.y, g.z}, m0.x, ro, rd, t_hit, rec);
            const DielConst<T> dc = {m0.w, m1.x, m1.y};
            V3<T> vec; T vs;
            const int todo = scatter_begin<T>(rng, DIELECTRIC, m0.y, rd, rec, vec, vs, &dc);
            rd = todo == PATH_NORM ? normalize(vec) : vec;                   // a reflection is NOT renormalised (src/material.jl:48)
            ref_depth -= 1u << RTW_REF_BITS;
            const bool has_ray = ref_depth > RTW_REF_MASK;
            if (valid) {
                L::st_o(r, rec.p, T(0));
                L::st_d(r, rd, L::w_of(-2));
                *reinterpret_cast<ulonglong2 *>(r + L::RNG) = ulonglong2{rng.x, rng.y};
                *reinterpret_cast<unsigned *>(r + L::TAIL + 8) = ref_depth;
            }
            push3(PQ_SCAN, valid && has_ray, PQ_END, valid && !has_ray, PQ_END, false);
            clk.lap(3);
        } else {
            // ---- END: a miss adds thr * sky to its pixel EXACTLY (src/ray_color.jl:36; a path out of depth adds nothing); finished
            //      chunks retire, finished jobs are stored, new items are taken; then the next sample starts (src/render.jl:29-37) ----
            V3<T> ro = {0, 0, 0}, rd; T w_;
            L::ld_d(r, rd, w_);
            Rng rng;
            { const ulonglong2 g = *reinterpret_cast<const ulonglong2 *>(r + L::RNG); rng.x = g.x; rng.y = g.y; }
            double2 t01 = *reinterpret_cast<const double2 *>(r + L::THR);
            double t2 = *reinterpret_cast<const double *>(r + L::TAIL);
            unsigned ref_depth, misc;
            { const uint2 tl = *reinterpret_cast<const uint2 *>(r + L::TAIL + 8); ref_depth = tl.x; misc = tl.y; }
            if (valid && L::i_of(w_) == -1) {
                const C3 sky = skycolor(rd);
                const unsigned item_ref = ref_depth & RTW_REF_MASK;
                fx_accumulate3(sh->slot(item_ref >> 4, P.slot_stride)->acc(item_ref & 15u), t01.x * sky.r, t01.y * sky.g, t2 * sky.b);
            }
            unsigned samples_left = misc & RTW_PM_SAMPLES;
            bool jitter = (misc & RTW_PM_JITTER) != 0u, have_item = (misc & RTW_PM_ITEM) != 0u;
            const bool need = valid && samples_left == 0u;
            bool dying = false, got = false, blocked = false;
            unsigned long long pix = 0; unsigned chunk = 0;
            if (__any(need)) {
                bool last = false;
                if (need && have_item) {
                    last = __hip_atomic_fetch_add(&sh->slot((ref_depth & RTW_REF_MASK) >> 4, P.slot_stride)->remaining, -1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == 1;
                    have_item = false;
                }
                // jobs whose last item just finished: the wave stores their pixels and frees the slot
                unsigned long long fin = __ballot(last);
                while (fin) {
                    const int Lf = __builtin_ctzll(fin);
                    fin &= fin - 1ull;
                    JobSlot *S = sh->slot(uniform((unsigned)__shfl((int)((ref_depth & RTW_REF_MASK) >> 4), Lf)), P.slot_stride);
                    store_job<T>(P, S, lane, out);
                    __hip_atomic_store(&S->ready_seq, RTW_SLOT_FREE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (lane == 0) __hip_atomic_fetch_add(&sh->ctl[PC_FREE], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // items from the workgroup's dispenser (trace_kernel's ticket / slot protocol, under a lock)
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    const bool want = need && !got && !dying;
                    const unsigned long long take_mask = __ballot(want);
                    if (!take_mask) break;
                    PoolDisp *D = &sh->disp;
                    if (lane == 0) {
                        for (;;) {
                            unsigned expect = 0u;
                            if (__hip_atomic_compare_exchange_strong(&D->lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    unsigned d_next = uniform(__hip_atomic_load(&D->next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    unsigned d_end = uniform(__hip_atomic_load(&D->end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    unsigned d_slot = uniform(__hip_atomic_load(&D->slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    unsigned d_b = uniform(__hip_atomic_load(&D->b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    unsigned long long d_valid = __hip_atomic_load(&D->valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    bool stop = false;
                    if (d_next >= d_end) {
                        // the batch is used up: the next batch of the job being dispensed, or a new job in any free job slot
                        unsigned have_job = uniform(__hip_atomic_load(&D->have_job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        const unsigned eof = uniform(__hip_atomic_load(&sh->ctl[PC_EOF], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        bool have_batch = false;
                        stop = true;
                        if (eof) {
                            if (want) dying = true;                 // the job queues are exhausted: these slots are done
                        } else if (have_job && d_b + 1u < P.bpj) {
                            d_b += 1u; have_batch = true;
                        } else {
                            have_job = 0u;
                            unsigned rs = 0u;
                            if (lane < P.n_slots) rs = __hip_atomic_load(&sh->slot(lane, P.slot_stride)->ready_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            const unsigned long long free_m = __ballot(lane < P.n_slots && rs == RTW_SLOT_FREE);
                            if (free_m) {
                                const unsigned sl = (unsigned)__builtin_ctzll(free_m);
                                JobSlot *S = sh->slot(sl, P.slot_stride);
                                if (lane == 0) __hip_atomic_store(&S->ready_seq, RTW_SLOT_OPENING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                open_job(P, S, lane, ctr, &sh->jobs);
                                if (uniform(__hip_atomic_load(&S->job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == RTW_JOB_EOF) {
                                    if (lane == 0) {
                                        __hip_atomic_store(&S->ready_seq, RTW_SLOT_FREE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                        __hip_atomic_store(&sh->ctl[PC_EOF], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                    if (want) dying = true;
                                } else {
                                    if (lane == 0) {
                                        __hip_atomic_store(&S->ready_seq, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);      // in flight
                                        __hip_atomic_fetch_sub(&sh->ctl[PC_FREE], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                                    }
                                    d_slot = sl; d_b = 0u; have_job = 1u; have_batch = true;
                                }
                            } else {
                                blocked = want;                     // every job slot holds a job in flight: wait for one to retire
                            }
                        }
                        if (have_batch) {
                            const JobSlot *S = sh->slot(d_slot, P.slot_stride);
                            const unsigned px = lane & ((1u << P.job_shift) - 1u), ch = d_b * (64u >> P.job_shift) + (lane >> P.job_shift);
                            d_valid = __ballot((int)ch < P.n_chunks && ((S->valid >> px) & 1u));
                            d_next = 0u; d_end = 64u;
                            stop = false;
                        }
                        if (lane == 0) __hip_atomic_store(&D->have_job, have_job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    if (d_next < d_end) {
                        // item p of the batch = (pixel p mod job_px, chunk (64 / job_px) b + p / job_px)
                        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(take_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)take_mask, 0u));
                        const unsigned p = d_next + rank;
                        if (want && p < d_end) {
                            const JobSlot *S = sh->slot(d_slot, P.slot_stride);
                            const unsigned px = p & ((1u << P.job_shift) - 1u), ch = d_b * (64u >> P.job_shift) + (p >> P.job_shift);
                            if ((d_valid >> p) & 1ull) {
                                const unsigned rs_ = P.rows_shift;
                                const int i0 = S->i_base + (int)(px & ((1u << rs_) - 1u)), j0 = S->j_base + (int)(px >> rs_);
                                pix = (unsigned long long)j0 * (unsigned)P.height + (unsigned)i0;
                                chunk = ch; got = true;
                                const int s0 = (int)ch * P.chunk_spp;
                                samples_left = (unsigned)(min(P.spp, s0 + P.chunk_spp) - s0);
                                jitter = s0 != 0;                                             // sample 1 of the pixel is centred
                                ref_depth = d_slot * 16u + px;
                                have_item = true;
                            }
                            // padding item (chunk beyond n_chunks, pixel outside the image): nothing to do, pull again
                        }
                        d_next = min(d_end, d_next + (unsigned)__popcll(take_mask));
                    }
                    if (lane == 0) {
                        __hip_atomic_store(&D->next, d_next, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&D->end, d_end, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&D->slot, d_slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&D->b, d_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&D->valid, d_valid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(&D->lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                    __builtin_amdgcn_wave_barrier();
                    if (stop) break;
                }
                if (got) rng_stream(P.seed, pix, chunk, rng);
                const unsigned nd = (unsigned)__popcll(__ballot(dying));
                if (nd && lane == 0) __hip_atomic_fetch_add(&sh->ctl[PC_DEAD], nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // the next sample: jitter, lens disk (src/rand.jl:31-38), camera ray (src/camera.jl:43-48)
            const bool ns = valid && !dying && samples_left > 0u;
            bool has_ray = false;
            if (__any(ns)) {
                T su = 0, sv = 0;
                if (ns) {
                    T du = 0, dv = 0;
                    if (jitter) {
                        T r1, r2;
                        trand(rng, r1); du = r1 / w_div;
                        trand(rng, r2); dv = r2 / h_div;
                    }
                    const JobSlot *S = sh->slot((ref_depth & RTW_REF_MASK) >> 4, P.slot_stride);
                    const unsigned px = ref_depth & 15u, rs_ = P.rows_shift;
                    su = (T)S->uv[px >> rs_] + du;                              // T(j / W) + du,       src/render.jl:26,37
                    sv = (T)S->uv[4 + (px & ((1u << rs_) - 1u))] + dv;          // T((H - i) / H) + dv, src/render.jl:27,37
                    jitter = true;
                    samples_left -= 1u;
                }
                n_samples += (unsigned long long)__popcll(__ballot(ns));
                V3<T> rp = {0, 0, 0};
                bool pending = ns;
                while (pending) {
                    const T len2 = reject_trial<T>(rng, false, rp);
                    pending = !(len2 <= T(1));
                }
                if (ns) {
                    const Camera<T> cam = sh->cam;
                    V3<T> raw;
                    camera_ray_raw<T>(cam, su, sv, rp.x, rp.y, ro, raw);
                    rd = normalize(raw);
                    t01 = double2{1.0, 1.0}; t2 = 1.0;
                    ref_depth = (ref_depth & RTW_REF_MASK) | ((unsigned)P.max_depth << RTW_REF_BITS);
                    has_ray = ref_depth > RTW_REF_MASK;
                }
            }
            if (valid && !dying) {
                misc = samples_left | (jitter ? RTW_PM_JITTER : 0u) | (have_item ? RTW_PM_ITEM : 0u);
                if (ns) {
                    L::st_o(r, ro, T(0));
                    *reinterpret_cast<double2 *>(r + L::THR) = t01;
                }
                L::st_d(r, rd, L::w_of(-2));
                *reinterpret_cast<ulonglong2 *>(r + L::RNG) = ulonglong2{rng.x, rng.y};
                *reinterpret_cast<double *>(r + L::TAIL) = t2;
                *reinterpret_cast<uint2 *>(r + L::TAIL + 8) = uint2{ref_depth, misc};
            }
            push3(PQ_SCAN, ns && has_ray, PQ_END, valid && !dying && !(ns && has_ray) && !(blocked && !ns), PQ_WAIT, valid && !dying && blocked && !ns);
            clk.lap(0);
        }
        if (lane == 0) __hip_atomic_fetch_sub(&sh->ctl[PC_BUSY], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (PROFILE && lane == 0) {
            __hip_atomic_fetch_add(&sh->prof[4u * q], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sh->prof[4u * q + 1u], (unsigned long long)n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&sh->prof[4u * q + 2u], __builtin_readcyclecounter() - t_top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }

    if (PROFILE && lane == 0) {
        __hip_atomic_fetch_add(&sh->prof[26], __builtin_readcyclecounter() - t_kernel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh->prof[27], clk.acc[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // (wave, block) evaluations without any candidate
        __hip_atomic_fetch_add(&sh->prof[28], clk.acc[7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // ... of all
    }
    if (lane == 0) {
        __hip_atomic_fetch_add(&sh->fin_segments, n_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh->fin_samples, n_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_fetch_add(&sh->fin_waves, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(W - 1)) {
            atomicAdd(&ctr->segments, __hip_atomic_load(&sh->fin_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ctr->samples, __hip_atomic_load(&sh->fin_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ctr->n_waves, (unsigned long long)W);
            if (PROFILE) {
                unsigned long long *gp = reinterpret_cast<unsigned long long *>(ctr->end_hist + 256);
                for (int k = 0; k < 32; ++k) atomicAdd(&gp[k], __hip_atomic_load(&sh->prof[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            }
        }
    }
}

}  // namespace rtw
