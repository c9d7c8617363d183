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
#define RTW_POOL_RING 2048u          // entries per queue ring (u16: slot | generation << 11); any queue can hold every slot
#define RTW_POOL_RING_SHIFT 11
#define RTW_POOL_PAIR_CAP 256u       // candidate list entries per wave (trace_kernel: 512)
#ifndef RTW_POOL_LM_ROUNDS
#define RTW_POOL_LM_ROUNDS 2         // unit-ball trials run inline by the Lambertian/Metal stage (100 %, 48 % utilisation)
#endif
#ifndef RTW_POOL_REJ_ROUNDS
#define RTW_POOL_REJ_ROUNDS 2        // trials per visit of the straggler stage
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

// The item dispenser: trace_kernel's per-wave batch (ticket -> job slot -> 64 items) as ONE structure of the workgroup, used under
// `lock` by the wave whose END batch needs items (a wave that held a private batch could starve the job it belongs to: it hands
// items out only when it happens to run the END stage).
struct PoolDisp {
    unsigned lock, ticket, next, end, slot, b, have_ticket, tk_seq, tk_b, pad;
    unsigned long long valid;            // which of the batch's 64 items are real (chunk < n_chunks, pixel inside the image)
};
template <typename T, int W> struct PoolShared {
    unsigned char slots[RTW_SLOT_BYTES];                 // job slots, as in WgShared
    __device__ __forceinline__ JobSlot *slot(unsigned i, unsigned stride) { return reinterpret_cast<JobSlot *>(slots + i * stride); }
    uint2 q[8];                                          // (head, tail) of each queue: free-running positions
    unsigned dead;                                       // slots that found the job queues exhausted: the kernel ends at R
    unsigned fin_waves;
    unsigned long long fin_segments, fin_samples;
    PoolDisp disp;
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
    ws.cap = RTW_POOL_PAIR_CAP;

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
    if (threadIdx.x < 8) sh->q[threadIdx.x] = uint2{0u, threadIdx.x == PQ_END ? (unsigned)R : 0u};
    if (threadIdx.x == 0) {
        sh->dead = 0u; sh->fin_waves = 0u; sh->fin_segments = 0ull; sh->fin_samples = 0ull;
        sh->disp.lock = 0u; sh->disp.ticket = 0u; sh->disp.next = 0u; sh->disp.end = 0u; sh->disp.slot = 0u; sh->disp.b = 0u;
        sh->disp.have_ticket = 0u; sh->disp.tk_seq = 0u; sh->disp.tk_b = 0u; sh->disp.valid = 0ull;
        sh->jobs.jc = 0ull; sh->jobs.jc_lock = 0u; sh->jobs.queue_off = 0u; sh->jobs.last_g = 0u;
        sh->cam = cam_arg; sh->P = P_arg;
    }
    const KParams &P = sh->P;
    stage_scene<T>(scene, lds_geom);
    for (int i = threadIdx.x; i < n_alloc; i += W * 64) lds_kind[i] = (unsigned char)(i < scene.n ? (int)scene.mat0[i].z : 0);
    __syncthreads();

    unsigned long long n_segments = 0, n_samples = 0;
    const T w_div = (T)(float)P.width;    // f32_image_width  (src/render.jl:16)
    const T h_div = (T)(float)P.height;   // f32_image_height (src/render.jl:17)
    PhaseClock<PROFILE> clk;
    unsigned idle = 0;

#ifdef RTW_POOL_WATCHDOG
    unsigned wd_iter = 0, wd_q = 99, wd_n = 0;
#endif
    for (;;) {
#ifdef RTW_POOL_WATCHDOG
        if (++wd_iter > (unsigned)RTW_POOL_WATCHDOG) {      // debug builds: dump the pool's state once and let the kernel end
            if (lane == 0 && atomicCAS(&ctr->end_hist[0], 0u, 0xdeadbeefu) == 0u) {
                unsigned *d = ctr->end_hist + 1;
                d[0] = blockIdx.x; d[1] = wv; d[2] = wd_q; d[3] = wd_n; d[4] = sh->dead;
                for (int k = 0; k < PQ_COUNT; ++k) { d[5 + 2 * k] = sh->q[k].x; d[6 + 2 * k] = sh->q[k].y; }
                const unsigned *pd = reinterpret_cast<const unsigned *>(&sh->disp);
                for (int k = 0; k < 12; ++k) d[20 + k] = pd[k];
                for (unsigned k = 0; k < P.n_slots && k < 24u; ++k) { const JobSlot *S = sh->slot(k, P.slot_stride); d[36 + 3 * k] = S->ready_seq; d[37 + 3 * k] = S->job; d[38 + 3 * k] = (unsigned)S->remaining; }
                d[110] = sh->jobs.queue_off; d[111] = (unsigned)n_segments; d[112] = (unsigned)n_samples;
            }
            __hip_atomic_store(&sh->dead, (unsigned)R, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
#endif
        // ---- choose a stage: a queue that fills a wave (the shading stages first: the scan queue is where rays park), else the fullest ----
        unsigned hd = 0, cnt = 0;
        if (lane < PQ_COUNT) {
            hd = __hip_atomic_load(&sh->q[lane].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            cnt = __hip_atomic_load(&sh->q[lane].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - hd;
        }
        if (uniform(__hip_atomic_load(&sh->dead, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= (unsigned)R) break;
        const unsigned fullm = (unsigned)__ballot(cnt >= 64u) & ((1u << PQ_WAIT) - 1u);
        unsigned q, n;
        if (fullm) {
            const unsigned shading = fullm & ~1u;
            q = shading ? (unsigned)__builtin_ctz(shading) : 0u;
            n = 64u;
        } else {
            q = 0u; n = (unsigned)__builtin_amdgcn_readlane((int)cnt, 0);
#pragma unroll
            for (int k = 1; k < PQ_WAIT; ++k) { const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)cnt, k); if (c > n) { n = c; q = (unsigned)k; } }
            if (n == 0u) {
                // nothing but (perhaps) waiting slots: give the waves that hold the items a moment, then look whether a job slot is free
                if (idle < 8u) __builtin_amdgcn_s_sleep(1); else __builtin_amdgcn_s_sleep(8);
                idle += 1u;
                n = (unsigned)__builtin_amdgcn_readlane((int)cnt, PQ_WAIT);
                if (n == 0u || (idle & 3u) != 0u) continue;
                q = PQ_WAIT; n = n > 64u ? 64u : n;
            }
        }
        unsigned ok = 0;
        if (lane == q) {
            unsigned expect = hd;
            ok = __hip_atomic_compare_exchange_strong(&sh->q[lane].x, &expect, hd + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
        }
        if (!__builtin_amdgcn_readlane((int)ok, (int)q)) continue;              // another wave took them
        if (q != PQ_WAIT) idle = 0u;
#ifdef RTW_POOL_WATCHDOG
        wd_q = q; wd_n = n;
#endif
        const unsigned h0 = (unsigned)__builtin_amdgcn_readlane((int)hd, (int)q);
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
        auto push = [&](unsigned q2, bool pred) {
            const unsigned long long m = __ballot(pred);
            if (!m) return;
            unsigned pos = 0;
            if (lane == 0) pos = __hip_atomic_fetch_add(&sh->q[q2].y, (unsigned)__popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            pos = uniform(pos);
            if (pred) {
                const unsigned p = pos + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                __hip_atomic_store(rings + q2 * RTW_POOL_RING + (p & (RTW_POOL_RING - 1u)), (unsigned short)(id | (((p >> RTW_POOL_RING_SHIFT) & 31u) << RTW_POOL_RING_SHIFT)),
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
            push(PQ_LM, valid && idx >= 0 && kind != DIELECTRIC);
            push(PQ_END, valid && idx < 0);
            push(PQ_DIEL, valid && idx >= 0 && kind == DIELECTRIC);
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
            push(PQ_SCAN, done && has_ray);
            push(PQ_REJ, valid && !done);
            push(PQ_END, done && !has_ray);
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
            push(PQ_SCAN, valid && has_ray);
            push(PQ_END, valid && !has_ray);
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
                fx_accumulate(sh->slot(item_ref >> 4, P.slot_stride)->acc(item_ref & 15u), t01.x * sky.r, t01.y * sky.g, t2 * sky.b);
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
                        unsigned have_ticket = uniform(__hip_atomic_load(&D->have_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        unsigned tk_seq = uniform(__hip_atomic_load(&D->tk_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        unsigned tk_b = uniform(__hip_atomic_load(&D->tk_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (!have_ticket) {
                            const unsigned t = uniform(__hip_atomic_load(&D->ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                            if (lane == 0) __hip_atomic_store(&D->ticket, t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            tk_seq = udiv_magic(t, P.div_bpj_m, P.div_bpj_s);
                            tk_b = t - tk_seq * P.bpj;
                            have_ticket = 1u;
                        }
                        const unsigned sl = tk_seq - udiv_magic(tk_seq, P.div_slots_m, P.div_slots_s) * P.n_slots;   // tk_seq mod n_slots
                        JobSlot *S = sh->slot(sl, P.slot_stride);
                        unsigned rs = uniform(__hip_atomic_load(&S->ready_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (rs == RTW_SLOT_FREE && tk_b == 0u) {
                            // (only the lock holder opens jobs: the slot is ours)
                            if (lane == 0) __hip_atomic_store(&S->ready_seq, RTW_SLOT_OPENING, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            open_job(P, S, lane, ctr, &sh->jobs);
                            __hip_atomic_store(&S->ready_seq, tk_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            rs = tk_seq;
                        }
                        stop = true;                            // (unless a batch becomes available below)
                        if (rs < RTW_SLOT_OPENING) {
                            if (uniform(__hip_atomic_load(&S->job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == RTW_JOB_EOF) {
                                // the job queues are exhausted (this slot stays marked for good): these slots are done
                                if (want) dying = true;
                                have_ticket = 0u;
                            } else if (rs == tk_seq) {
                                d_slot = sl; d_b = tk_b; d_next = 0u; d_end = 64u;
                                have_ticket = 0u;
                                const unsigned px = lane & ((1u << P.job_shift) - 1u), ch = tk_b * (64u >> P.job_shift) + (lane >> P.job_shift);
                                d_valid = __ballot((int)ch < P.n_chunks && ((S->valid >> px) & 1u));
                                stop = false;
                            }
                            else blocked = want;     // the slot still holds an older job in flight -- try again later
                        } else blocked = want;
                        if (lane == 0) {
                            __hip_atomic_store(&D->have_ticket, have_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(&D->tk_seq, tk_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_store(&D->tk_b, tk_b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
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
                if (nd && lane == 0) __hip_atomic_fetch_add(&sh->dead, nd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
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
            push(PQ_SCAN, ns && has_ray);
            push(PQ_END, valid && !dying && !(ns && has_ray) && !(blocked && !ns));
            push(PQ_WAIT, valid && !dying && blocked && !ns);
            clk.lap(0);
        }
    }

    if (lane == 0) {
        __hip_atomic_fetch_add(&sh->fin_segments, n_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh->fin_samples, n_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__hip_atomic_fetch_add(&sh->fin_waves, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == (unsigned)(W - 1)) {
            atomicAdd(&ctr->segments, __hip_atomic_load(&sh->fin_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ctr->samples, __hip_atomic_load(&sh->fin_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ctr->n_waves, (unsigned long long)W);
        }
    }
}

}  // namespace rtw
