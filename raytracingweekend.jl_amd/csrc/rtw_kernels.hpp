// rtw_kernels.hpp -- the trace kernel: render -> ray_color -> hit/scatter fused, including the
// pixel accumulation and the gamma / store of src/render.jl:40.  gfx950 only; wave = 64 lanes.
//
// Work decomposition
//   job   = a block of 16, 8, 4 or 1 pixels of an 8x8 tile (rtw_launch.hip picks the size: 4 normally, 1 for small shards of long
//           renders, 16 for frames of a handful of jobs per workgroup) for ALL its sample chunks.  Jobs come from one queue per XCD
//           (claim_job: static first claims, then guided claims through an LDS cache); a job is owned by ONE workgroup, in one of
//           its job slots in LDS -- header + the pixels' accumulators -- while its items are in flight.  Slots are handed out OUT OF
//           ORDER: a straggler holds its own slot only (phase A of the lane loop).
//   item  = (pixel, chunk): `chunk_spp` consecutive samples of one pixel drawn from the item's own
//           Xoroshiro128+ stream.  64 consecutive items = job pixels x (64 / job pixels) chunks = one wave batch,
//           taken from the workgroup's ticket counter in LDS.
//   lane  = persistent worker.  It owns one item at a time, starts its next sample the moment
//           its path ends (no lock-step on path length) and takes the next item when its chunk
//           is finished.  The only convergent part is the sphere scan, which every lane with a
//           ray executes every iteration.
//   accum = every sample's radiance (3 doubles) is added to the job's accumulators EXACTLY (64.64
//           fixed point, LDS integer atomics) -- addition order does not matter, so there is no
//           per-chunk workspace in HBM and no second kernel: the lane that retires the job's last
//           item triggers the store of the job's pixels (sum -> / spp -> sqrt -> RGB{T}).
//   Results do not depend on scheduling, grid size, shard count or job-slot availability.
#pragma once
#include "rtw_device.hpp"

namespace rtw {

#define RTW_JOB_PX 16        // slot capacity: pixels per job are 16 (8 rows x 2 columns), 8 (8 x 1), 4 (4 x 1) or 1 -- KParams::job_shift
#ifndef RTW_SLOT_BYTES
#define RTW_SLOT_BYTES 4608  // LDS for job slots per workgroup: 24 slots of 1 pixel, 12 of 4, 7 of 8 or 4 of 16
#endif
#define RTW_REF_BITS 9       // item reference = slot (5 bits) << 4 | pixel (4 bits)
#define RTW_REF_MASK 511u
#define RTW_SLOT_FREE 0xffffffffu
#define RTW_SLOT_OPENING 0xfffffffeu       // (the ray-pool kernel's marker)
#define RTW_SLOT_OPENING_BIT 0x80000000u   // lane loop: ready_seq = this bit + the job's sequence number while its opener claims the job
#define RTW_JOB_EOF 0xffffffffu
#define RTW_JOB_RETRY 0xfffffffeu          // open_job: no job YET (another wave is refilling the workgroup's job cache) -- the lane loop asks again

struct KParams {
    int width, height, spp, max_depth;
    uint64_t seed;
    int n_chunks;      // effective (non-empty) chunks per pixel
    int chunk_spp;
    int shard_index, shard_count;
    int tiles_i, tiles_j;  // 8x8 tiles along rows (i) and columns (j)
    unsigned total_jobs;   // (64 >> job_shift) * (tiles owned by this shard)
    unsigned local_tiles;  // tiles owned by this shard
    unsigned bpj;          // batches per job = ceil(n_chunks / (64 >> job_shift))
    unsigned n_slots, slot_stride;          // job slots per workgroup (by job size), bytes per slot
    unsigned job_shift;    // log2(pixels per job): 4, 3, 2 or 0.  A batch = (1 << job_shift) pixels x (64 >> job_shift) chunks.
                           // Smaller jobs = finer load balance at the end of the queue (small shards); same image.
    unsigned rows_shift;   // a job is (1 << rows_shift) rows x (1 << (job_shift - rows_shift)) columns: rows first, because rows are
                           // contiguous in the column-major Matrix{RGB{T}} -- a finished job is stored as whole column strips
                           // (pixel px of the job = row px & (rows - 1), column px >> rows_shift)
    // exact unsigned division by loop-invariant divisors (host: make_udiv):
    // n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> s   for every 32-bit n
    unsigned div_bpj_m, div_bpj_s, div_tiles_m, div_tiles_s;
    int gamma;
    int out_layout;        // 0: Matrix{RGB{T}} column-major full frame; 1: compact, tile-major, this shard only
    int drain_profile;     // RTW_DRAIN_PROFILE (test aid): the workgroups report their waves' end clocks (DevCounters::t_first ...)
};

struct DevCounters {
    unsigned next_job[8];          // one job queue per XCD (claim_job)
    unsigned long long segments;
    unsigned long long samples;
    unsigned long long phase[32];  // RTW_PHASE_PROFILE=1 only: wave-cycles per phase (s_memtime) [0..5], event counters [6..31] (tools/valu_budget.py names them)
    unsigned long long t_first, t_last, t_end_sum, n_waves;   // wall clock (100 MHz) of the first wave start, the last wave
                                                              // end and the sum of all wave ends: the end-of-queue drain
    unsigned end_hist[4096];                                  // waves by end time since t_first, 0.25 ms bins (RTW_DRAIN_PROFILE)
};

// One job in flight: the bookkeeping of the open/retire protocol and the block header, followed in LDS by
// the job's pixel accumulators (8 x u64 per pixel: r.lo r.hi g.lo g.hi b.lo b.hi poison pad).
struct JobSlot {
    unsigned ready_seq;                      // RTW_SLOT_FREE | RTW_SLOT_OPENING_BIT + seq (being opened) | the job sequence number it holds
    unsigned job;                            // the job's queue position; RTW_JOB_EOF / RTW_JOB_RETRY while open_job reports that there is none (yet)
    int remaining;                           // items of the job not yet finished
    unsigned valid;                          // bit px: pixel px of the block lies inside the image
    int i_base, j_base;                      // 0-based row / column of the block's first pixel
    unsigned k_tile;                         // local tile index (compact output layout)
    unsigned pad;
    double uv[12];                           // [0..3] j / W for the block's columns, [4..11] (H - i) / H for its rows (src/render.jl:26-27), as binary64
    __device__ __forceinline__ unsigned long long *acc(unsigned px) { return reinterpret_cast<unsigned long long *>(this + 1) + 8u * px; }
    __device__ __forceinline__ const unsigned long long *acc(unsigned px) const { return reinterpret_cast<const unsigned long long *>(this + 1) + 8u * px; }
};
static_assert(sizeof(JobSlot) == 128, "JobSlot header is 128 bytes (16-byte aligned accumulators follow)");
struct JobCache { unsigned long long jc; unsigned jc_lock; unsigned queue_off; unsigned last_g; unsigned static_used; };       // see claim_job
template <typename T> struct WgShared {
    unsigned char slots[RTW_SLOT_BYTES];      // n_slots x (128-byte JobSlot + 64 bytes per job pixel)
    __device__ __forceinline__ JobSlot *slot(unsigned i, unsigned stride) { return reinterpret_cast<JobSlot *>(slots + i * stride); }
    unsigned ticket;                         // next batch of this workgroup
    unsigned eof;                            // an opener found every job queue exhausted (lane loop; see the slot protocol there)
    unsigned fin_waves;                      // waves of this workgroup that have finished, and what they counted: the last one adds
    unsigned long long fin_segments, fin_samples;   // it to DevCounters (one pair of global atomics per workgroup, not per wave)
    unsigned long long t_wave[8];            // wall clock at the start [0..3] and the end [4..7] of each wave (RTW_DRAIN_PROFILE)
    JobCache jobs;                           // queue positions claimed but not started yet (claim_job)
    Camera<T> cam;                           // read per new sample (keeps 22 SGPRs out of the scan loop)
    KParams P;                               // read where needed (item pull, store): not held in SGPRs across the scan
};

// Phase profiler (opt-in instantiation, never used for timed runs): s_memtime stamps around
// the phases of the lane loop, summed per wave.
template <bool ON> struct PhaseClock {
    unsigned long long t0 = 0, acc[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void start() { if (ON) t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void lap(int k) {
        if (ON) { unsigned long long t = __builtin_readcyclecounter(); acc[k] += t - t0; t0 = t; }
    }
    static constexpr bool on() { return ON; }
    __device__ __forceinline__ void count(int k, unsigned n) { if (ON) acc[k] += n; }      // event counters in the cells 6 .. 31
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
    // lane = 3 * pixel + channel: consecutive lanes write consecutive elements of a column strip
    const unsigned px = lane / 3u, ch = lane - 3u * px, rs = P.rows_shift;
    if (px < (1u << P.job_shift)) {
        if ((S->valid >> px) & 1u) {
            const int i0 = S->i_base + (int)(px & ((1u << rs) - 1u)), j0 = S->j_base + (int)(px >> rs);
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

// The XCD (one of the chip's 8 dies, each with its own L2) this wave runs on: HW_REG_XCC_ID[3:0].  Used for AFFINITY only --
// any value in 0 .. 7 gives the same image.
__device__ __forceinline__ unsigned xcd_id() { return (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u; }

// Job queues.  One queue per XCD (one of the chip's 8 dies, each with its own L2), a workgroup's home queue being its own die's:
// full frame (shard_count == 1): tile-column tj (8 pixel columns = one contiguous 8 H RGB{T} strip of the column-major
// Matrix{RGB{T}}) belongs to queue tj mod 8; sharded / compact: local tile k belongs to queue k mod 8.  A queue's positions are
// tile-major (position = tile within the queue x sub-blocks per tile + sub-block), so a die works its way through a tile column
// from the top, and the jobs in flight on the chip at any time are neighbours in the frame: measured, that is worth 1 % of
// the frame time (rays of neighbouring pixels share the blocks of spheres the wave-level early-out of the scan can skip:
// with the queue cut into 8 far-apart streams, taken in turn, the same kernel ran 364.3 ms instead of 361.5), and what the
// frame's write traffic needs: neighbouring 48-byte strips are stored at about the same time and leave the L2 as whole sectors.
// A workgroup takes from its home queue and, when that is exhausted, from the next ones (the tail of the frame).
__device__ __forceinline__ unsigned queue_tiles(const KParams &P, unsigned xq) {
    if (P.shard_count == 1) { const unsigned tj = (unsigned)P.tiles_j; return tj > xq ? ((tj - xq + 7u) >> 3) * (unsigned)P.tiles_i : 0u; }
    return P.local_tiles > xq ? (P.local_tiles - xq + 7u) >> 3 : 0u;
}
__device__ __forceinline__ unsigned queue_positions(const KParams &P, unsigned xq, unsigned sub_shift) { return queue_tiles(P, xq) << sub_shift; }

// One claim from the job queues (lane 0 of the wave that opens a slot).  A global atomic is a 32-byte write (and read) at
// the memory side -- whatever its scope: this memory is write-through in the L2 and "workgroup"-scope atomics on a line that only
// one die touches went to HBM just the same (measured: a two-level queue with die-local second-level counters TRIPLED
// WRITE_SIZE) -- and one per 4-pixel job was 16.6 MB per 1080p frame, two thirds of the frame itself.  So a workgroup claims up
// to RTW_JOB_CLAIM consecutive queue positions -- neighbouring strips of one tile -- with ONE atomic and hands the others out
// from LDS (`jc`: queue << 56 | end << 28 | next; `jc_lock` serialises refills, a wave that does not get the lock claims a
// single position).  What a workgroup holds nobody else can take, and neighbours cost alike (the glass sphere of the headline
// scene spans 76 tiles: a workgroup there starts a job every 4 ms, not every 0.9): a fixed 8 per claim left 17 ms of idle wave
// slots at the end of the frame.  Hence guided self-scheduling: a claim takes left / (RTW_CLAIM_TAIL x workgroups of the
// die) positions, at most RTW_JOB_CLAIM -- 8 for the first two thirds of the queue, then 7, 6 ... 1 -- where `left` is
// what this workgroup's previous claim saw.  Measured at 1080p x 1000 spp: as fast as single claims (366.9 vs 367.5 ms, the
// same 2.3 ms of drain) with 30.1 instead of 43.0 MB of WRITE_SIZE; RTW_CLAIM_TAIL 8: 11 - 13 ms of drain; 24: +0.6 MB.
// (Rejected: claims of 8 positions FAR APART in the frame -- no drain, but 1 % slower for the lost coherence, see above; a
// static share of the queue per workgroup, no atomics at all -- the workgroups' shares differ by +-35 % in cost, 438 ms.)
// Another die's queue is only ever visited at its end: single claims there.  Returns 1 with a job, 0 when every queue is exhausted FOR GOOD,
// 2 (try again) when this caller found the queues exhausted WITHOUT holding the refill lock: the lock's holder may be about to publish
// positions it has claimed (what it publishes are jobs; "exhausted" must not be concluded over its head -- since round 6 one such conclusion
// ends the whole workgroup, see sh->eof in the lane loop).  Under the lock "exhausted" is final: the cache is empty, only lock holders fill
// it, and the queue counters never come back.
#ifndef RTW_JOB_CLAIM
#define RTW_JOB_CLAIM 8u
#endif
#ifndef RTW_CLAIM_TAIL
#define RTW_CLAIM_TAIL 16u
#endif
// Static first claims.  The first RTW_STATIC_CLAIMS claims of a workgroup take no atomic at all: workgroup b owns positions
// (b >> 3) K .. (b >> 3) K + K - 1 of queue b & 7 (its own die's, with the hardware's round-robin placement), and what the atomic counters
// hand out starts behind the static part of each queue (static_base).  At the start of a render every wave of the chip wants a job at
// once: 5 120 returning atomics on 8 addresses are executed one after the other by the memory side -- measured on a 96 x 54 frame:
// 166 us of kernel time with them, 64 us with jobs so large that a quarter as many were claimed; on 320 x 180 x 64 spp the start-up is
// 10 % of the frame.  (Which workgroup renders which job does not matter to the image.)
#ifndef RTW_OPEN_RETRY
#define RTW_OPEN_RETRY 1      // (0: debugging aid -- an opener whose claim must be repeated spins in open_job instead of giving its slot back)
#endif
#ifndef RTW_STATIC_CLAIMS
#define RTW_STATIC_CLAIMS 4u
#endif
// (a queue shorter than RTW_STATIC_CLAIMS x its workgroups deals fewer -- possibly no -- static positions per workgroup: every workgroup
//  of the queue gets the same number, the rest goes through the counter)
__device__ __forceinline__ unsigned static_claims(unsigned xq, unsigned q_pos) {
    const unsigned wgs = (gridDim.x + 7u - xq) >> 3;                  // workgroups b with b & 7 == xq
    const unsigned k = wgs ? q_pos / wgs : 0u;
    return k < RTW_STATIC_CLAIMS ? k : RTW_STATIC_CLAIMS;
}
__device__ __forceinline__ unsigned static_base(unsigned xq, unsigned q_pos) { return static_claims(xq, q_pos) * ((gridDim.x + 7u - xq) >> 3); }
__device__ __forceinline__ int claim_job(const KParams &P, DevCounters *ctr, JobCache *C, unsigned xcd, unsigned sub_shift, unsigned &xq_out, unsigned &gq_out) {
    constexpr unsigned long long M28 = (1ull << 28) - 1ull;
    if (__hip_atomic_load(&C->static_used, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < RTW_STATIC_CLAIMS) {
        const unsigned k = __hip_atomic_fetch_add(&C->static_used, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned xq = blockIdx.x & 7u, ks = static_claims(xq, queue_positions(P, xq, sub_shift));
        if (k < ks) { xq_out = xq; gq_out = (blockIdx.x >> 3) * ks + k; return 1; }
    }
    for (;;) {
        unsigned long long w = __hip_atomic_load(&C->jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while ((w & M28) < ((w >> 28) & M28)) {                    // cached positions: take one
            if (__hip_atomic_compare_exchange_strong(&C->jc, &w, w + 1ull, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                xq_out = (unsigned)(w >> 56); gq_out = (unsigned)(w & M28);
                return 1;
            }
        }
        unsigned expect = 0u;
        const bool locked = __hip_atomic_compare_exchange_strong(&C->jc_lock, &expect, 1u, __ATOMIC_ACQUIRE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (locked) {
            w = __hip_atomic_load(&C->jc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if ((w & M28) < ((w >> 28) & M28)) {                   // refilled by the previous lock holder meanwhile: use that
                __hip_atomic_store(&C->jc_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                continue;
            }
        }
        unsigned off = __hip_atomic_load(&C->queue_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        bool got = false;
        while (off < 8u) {
            const unsigned xq = (xcd + off) & 7u;
            const unsigned q_pos = queue_positions(P, xq, sub_shift);
            // guided self-scheduling (what this workgroup's previous claim returned tells how far the queue is)
            const unsigned seen = __hip_atomic_load(&C->last_g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), left = q_pos > seen ? q_pos - seen : 0u;
            unsigned n = 1u;
            if (locked && off == 0u) { n = left / (RTW_CLAIM_TAIL * (gridDim.x / 8u + 1u)); n = n > RTW_JOB_CLAIM ? RTW_JOB_CLAIM : n < 1u ? 1u : n; }
            const unsigned g0 = atomicAdd(&ctr->next_job[xq], n) + static_base(xq, q_pos);         // (behind the queue's static part)
            if (off == 0u) __hip_atomic_store(&C->last_g, g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (any wave's: a recent position)
            if (g0 >= q_pos) { off += 1u; continue; }            // this queue is exhausted (for good): the next die's
            xq_out = xq; gq_out = g0; got = true;
            const unsigned end = g0 + n < q_pos ? g0 + n : q_pos;
            if (locked && end > g0 + 1u)
                __hip_atomic_store(&C->jc, ((unsigned long long)xq << 56) | ((unsigned long long)end << 28) | (unsigned long long)(g0 + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        __hip_atomic_store(&C->queue_off, off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (locked) __hip_atomic_store(&C->jc_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        return got ? 1 : (locked ? 0 : 2);
    }
}

// Open a job slot (whole wave): take job ids from the queues until one has a pixel inside the
// image (or every queue is exhausted), zero its accumulators and fill in the block's header.
// S->job: the job's queue position, RTW_JOB_EOF (every queue is exhausted for good) or -- WITH_RETRY only -- RTW_JOB_RETRY (claim_job's "try again";
// without WITH_RETRY the claim is repeated here until it is decided).
template <bool WITH_RETRY = false>
__device__ RTW_RARE_ATTR void open_job(const KParams &P, JobSlot *S, unsigned lane, DevCounters *ctr, JobCache *jcache) {
    unsigned g = RTW_JOB_EOF, valid = 0, k = 0;
    int i_base = 0, j_base = 0;
    const unsigned rs = P.rows_shift, cs = P.job_shift - rs, sub_shift = 6u - P.job_shift, bps_shift = 3u - rs;
    const unsigned xcd = xcd_id();
    for (;;) {
        unsigned xq = 0, gq = 0, ok = 0;
        if (lane == 0) ok = (unsigned)claim_job(P, ctr, jcache, xcd, sub_shift, xq, gq);
        ok = uniform(ok);
        if (ok == 2u) { if (WITH_RETRY) { g = RTW_JOB_RETRY; break; } __builtin_amdgcn_s_sleep(1); continue; }
        if (!ok) break;
        xq = uniform(xq); gq = uniform(gq);
        const unsigned qt = gq >> sub_shift, q = gq & ((1u << sub_shift) - 1u);      // tile within the queue, block within the tile
        unsigned tj, ti;
        if (P.shard_count == 1) {
            const unsigned tjq = udiv_magic(qt, P.div_tiles_m, P.div_tiles_s);
            ti = qt - tjq * (unsigned)P.tiles_i; tj = xq + 8u * tjq;
            k = tj * (unsigned)P.tiles_i + ti;
        } else {
            k = xq + 8u * qt;
            const unsigned t = k * (unsigned)P.shard_count + (unsigned)P.shard_index;
            tj = udiv_magic(t, P.div_tiles_m, P.div_tiles_s); ti = t - tj * (unsigned)P.tiles_i;
        }
        i_base = (int)(ti * 8u + ((q & ((1u << bps_shift) - 1u)) << rs));
        j_base = (int)(tj * 8u + ((q >> bps_shift) << cs));
        const int i0 = i_base + (int)(lane & ((1u << rs) - 1u)), j0 = j_base + (int)((lane >> rs) & ((1u << cs) - 1u));
        valid = (unsigned)__ballot(lane < (1u << P.job_shift) && i0 < P.height && j0 < P.width);
        if (valid) { g = gq; break; }                            // (blocks entirely outside the image are skipped)
    }
    if (g < RTW_JOB_RETRY) {
        if (lane < (4u << P.job_shift)) {
            unsigned zero = 0u;
            __asm__ volatile("" : "+v"(zero));       // (made here: a hoisted zero quad costs 4 loop-long registers)
            reinterpret_cast<uint4 *>(S->acc(0))[lane] = uint4{zero, zero, zero, zero};   // 64 B per pixel
        }
        if (lane < 4) S->uv[lane] = (double)(j_base + (int)lane + 1) / (double)P.width;                        // j / W
        else if (lane < 12) S->uv[lane] = (double)(P.height - (i_base + (int)lane - 4 + 1)) / (double)P.height;   // (H - i) / H
        if (lane == 0) {
            S->remaining = (int)((unsigned)__popc(valid) * (unsigned)P.n_chunks);
            S->valid = valid; S->i_base = i_base; S->j_base = j_base; S->k_tile = k;
        }
    }
    if (lane == 0) S->job = g;
}

// NUMK >= 0: the numerics mode is fixed at compile time (the launcher picks such an instance for the default mode of the headline
// variants: the other modes' code and their scalar state are then not in the kernel at all); NUMK < 0: the mode of the arguments.
template <typename T, bool PROFILE, bool LDS_SCENE, bool CULL, bool MFMA = false, int NUMK = -1>
__global__ __launch_bounds__(256, (TraceWavesOf<T, CULL, MFMA>::value)) void trace_kernel(KParams P_arg, Camera<T> cam_arg, DevScene<T> scene,
                                                   CullScene<T> cull, T *__restrict__ out, DevCounters *ctr) {
    using V4 = typename Vec4<T>::type;
    if constexpr (NUMK >= 0) { scene.numerics = NUMK; cull.numerics = NUMK; }
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
    // (CULL on the matrix pipe: the tables of the block vote behind the index list)
    [[maybe_unused]] unsigned *lds_tab = reinterpret_cast<unsigned *>(reinterpret_cast<unsigned char *>(lds_orig) + (CULL ? ((size_t)cull_exact_count(cull) * sizeof(unsigned short) + 15) / 16 * 16 : 0));
    if (threadIdx.x < P_arg.n_slots) { JobSlot *S0 = sh->slot(threadIdx.x, P_arg.slot_stride); S0->ready_seq = RTW_SLOT_FREE; S0->job = 0u; }
    if (threadIdx.x == 0) { sh->ticket = 0u; sh->eof = 0u; sh->fin_waves = 0u; sh->fin_segments = 0ull; sh->fin_samples = 0ull; sh->jobs.jc = 0ull; sh->jobs.jc_lock = 0u; sh->jobs.queue_off = 0u; sh->jobs.last_g = 0u; sh->jobs.static_used = 0u; sh->cam = cam_arg; sh->P = P_arg; }
    const KParams &P = sh->P;
    if (LDS_SCENE) {
        if (CULL) stage_cull_scene<T>(cull, lds_geom, lds_orig);
        else stage_scene<T>(scene, lds_geom);
        if constexpr (CULL && MFMA) {
            const unsigned *gt = reinterpret_cast<const unsigned *>(cull.mf_box + 8 * (cull.mf_blocks + 1));
            for (int i = threadIdx.x; i < cull_tab_words(cull.mf_blocks); i += blockDim.x) lds_tab[i] = gt[i];
        }
    }
    __syncthreads();

    if (lane == 0 && (threadIdx.x >> 6) < 4u) sh->t_wave[threadIdx.x >> 6] = wall_clock64();     // (in LDS: not a register pair held across the lane loop)
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
                MfmaCull mc = mfma_cull_of(cull);
                if (LDS_SCENE) mc.tab = lds_tab;
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
        if (PROFILE) { clk.count(8, 1u); clk.count(9, (unsigned)__popcll(__ballot(has_ray))); if (!__any(has_ray)) clk.count(10, 1u); }      // lane loop iterations, lanes with a ray, iterations without any

        // ---- (H1) a miss ends the sample: its radiance thr * sky (src/ray_color.jl:36) is added EXACTLY
        //      to the pixel's accumulators in LDS (a path that runs out of depth adds 0: nothing to do) ----
        const bool hit = has_ray && idx >= 0;
        if constexpr (MFMA) {
            // About a quarter of the lanes end a path per iteration, and each has three channels to convert and add: instead
            // of three rounds at ~25 % lane utilisation the (lane, channel) tasks are dealt to ALL lanes through the wave's
            // candidate list area in LDS (free between two scans): one round for up to 21 paths.
            const bool miss = RTW_PROBE_MISS(has_ray && idx < 0);
            const unsigned long long miss_mask = __ballot(miss);
            if (miss_mask) {
                clk.count(18, 1u); clk.count(19, (3u * (unsigned)__popcll(miss_mask) + 63u) / 64u);         // H1 executed; its rounds of 64 (lane, channel) tasks
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
            clk.count(20, 1u);                                                                            // A executed
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
                clk.count(21, 1u);                                                                        // jobs stored
                __hip_atomic_store(&S->ready_seq, RTW_SLOT_FREE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            // Items for the lanes that need one: from the wave's current batch; when that runs out (or there is none) the wave draws a
            // batch ticket, makes it usable (finds or opens the job's slot) and serves the remaining takers in the same iteration -- a
            // lane that finds the pool short would otherwise sit out a whole scan (measured: 1 - 9 % of all lane-iterations).
            // Job slots are allocated OUT OF ORDER: job `seq` lives in whichever slot its opener found free, and every ticket of the
            // job finds it by its sequence number (lane l looks at slot l).  A slot is held until the job's last item is done -- a
            // straggler (a 50-bounce path inside the glass sphere) holds ITS slot only; with the slots used as a ring it blocked every
            // later job of the workgroup (measured at 1080p: 11 % of the lane-iterations lost at 200 spp, 69 % at 64 spp).
            //   ready_seq:  RTW_SLOT_FREE | RTW_SLOT_OPENING_BIT + seq (its opener is claiming a job) | seq (open)
            //   sh->eof:    set, before the slot is given back, by an opener whose claim found every queue exhausted FOR GOOD (claim_job's 0,
            //               decided under the refill lock: from then on every claim of this workgroup fails).  An opener advertises its
            //               slot BEFORE it claims: a wave that reads eof first and then finds its job neither open nor opening knows
            //               that the job does not exist.
#pragma unroll 1
            for (int round = 0; round < 2; ++round) {
                const bool taker = need && alive && !have_item;
                const unsigned long long take_mask = __ballot(taker);
                if (!take_mask) break;
                if (pool_next >= pool_end) {
                    if (!have_ticket) {
                        unsigned t = 0;
                        if (lane == 0) t = __hip_atomic_fetch_add(&sh->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        t = uniform(t);
                        tk_seq = udiv_magic(t, P.div_bpj_m, P.div_bpj_s);
                        tk_b = t - tk_seq * P.bpj;
                        have_ticket = true;
                    }
                    const unsigned eof = uniform(__hip_atomic_load(&sh->eof, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP));     // (read BEFORE the slots)
                    unsigned rs = RTW_SLOT_OPENING;                            // (0xfffffffe: matches neither a sequence number < 2^31 nor one with the opening bit)
                    if (lane < P.n_slots) rs = __hip_atomic_load(&sh->slot(lane, P.slot_stride)->ready_seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    unsigned long long m_mine = __ballot(rs == tk_seq);
                    const unsigned long long m_opening = __ballot(rs == (tk_seq | RTW_SLOT_OPENING_BIT)), m_free = __ballot(rs == RTW_SLOT_FREE);
                    unsigned sl = 0;
                    if (!m_mine && !m_opening && !eof && tk_b == 0u && m_free) {
                        // this wave opens the job: take a free slot, claim a job from the global queues, zero the accumulators
                        sl = (unsigned)__builtin_ctzll(m_free);
                        JobSlot *S = sh->slot(sl, P.slot_stride);
                        unsigned won = 0;
                        if (lane == 0) {
                            unsigned expect = RTW_SLOT_FREE;
                            won = __hip_atomic_compare_exchange_strong(&S->ready_seq, &expect, tk_seq | RTW_SLOT_OPENING_BIT, __ATOMIC_ACQ_REL,
                                                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                        }
                        if (uniform(won)) {
                            open_job<RTW_OPEN_RETRY != 0>(P, S, lane, ctr, &sh->jobs);
                            const unsigned opened = uniform(S->job);
                            if (opened >= RTW_JOB_RETRY) {
                                if (opened == RTW_JOB_EOF && lane == 0) __hip_atomic_store(&sh->eof, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                __hip_atomic_store(&S->ready_seq, RTW_SLOT_FREE, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                                if (opened == RTW_JOB_EOF) {
                                    if (taker) alive = false;                 // the global queues are exhausted: these lanes are done
                                    have_ticket = false;
                                }
                                break;                                        // (RTW_JOB_RETRY: the ticket is kept, the slot given back: ask again in the next iteration)
                            }
                            __hip_atomic_store(&S->ready_seq, tk_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            m_mine = 1ull << sl;
                        }
                        // (lost the slot to another opener: look again in the next iteration)
                    } else if (!m_mine && !m_opening && eof) {
                        if (taker) alive = false;                             // the job of this ticket does not exist: the queues are exhausted
                        have_ticket = false;
                        break;
                    }
                    if (!m_mine) break;                                       // the job is not open yet / no slot is free: try again in the next iteration
                    sl = (unsigned)__builtin_ctzll(m_mine);
                    const JobSlot *S = sh->slot(sl, P.slot_stride);
                    clk.count(22, 1u);                                                                    // batches set up
                    pool_slot = sl; pool_b = tk_b;
                    pool_next = 0; pool_end = 64;
                    have_ticket = false;
                    if constexpr (MFMA) {
                        // Lane l prepares item l of the batch: setting up a stream is 2 splitmix64 + 1 step (~60 VALU), and a
                        // wave takes items a few lanes at a time -- almost every iteration for ~6 % of its lanes.
                        const unsigned px = lane & ((1u << P.job_shift) - 1u), chunk = tk_b * (64u >> P.job_shift) + (lane >> P.job_shift);
                        const unsigned rsh = P.rows_shift;
                        const int i0 = S->i_base + (int)(px & ((1u << rsh) - 1u)), j0 = S->j_base + (int)(px >> rsh);
                        Rng r0;
                        rng_stream(P.seed, (unsigned long long)j0 * (unsigned)P.height + (unsigned)i0, chunk, r0);
                        __builtin_amdgcn_wave_barrier();                      // (the previous batch's states have all been read)
                        pool_rng[lane] = ulonglong2{r0.x, r0.y};
                        pool_valid = __ballot((int)chunk < P.n_chunks && ((S->valid >> px) & 1u));
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                // hand out items of the wave's batch: item p = (pixel p mod job_px, chunk (64 / job_px) b + p / job_px)
                const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(take_mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)take_mask, 0u));   // takers below this lane
                const unsigned p = pool_next + rank;
                if (taker && p < pool_end) {
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
                            const unsigned rsh = P.rows_shift;
                            const int i0 = S->i_base + (int)(px & ((1u << rsh) - 1u)), j0 = S->j_base + (int)(px >> rsh);
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
                const unsigned want = (unsigned)__popcll(take_mask), have = pool_end - pool_next;
                if (PROFILE && want > have) clk.count(11, want - have);                // takers the pool could not serve in this round
                pool_next = min(pool_end, pool_next + want);
                if (want <= have) break;                                              // every taker was served (padding items: they pull again next time)
            }
            if (PROFILE) clk.count(12, (unsigned)__popcll(__ballot(need && alive && !have_item)));     // takers left without an item in this iteration
        }
        if (!__any(alive)) break;
        clk.lap(0);

        // ---- (H2) a hit starts the scatter (src/ray_color.jl:20-33) ----
        int todo = PATH_READY;    // PATH_BALL: scatter waits for a unit-ball sample; PATH_NORM: direction to normalise
        V3<T> vec = {0, 0, 0};    // PATH_BALL: n (Lambertian) / reflect(d, n) (Metal); PATH_NORM: the raw direction
        T vscale_ = 1;            // PATH_BALL: 1 (Lambertian) / fuzz (Metal)
        int kind = 0;
        if (PROFILE && __any(hit)) clk.count(23, 1u);                                                    // H2 executed
        if (hit) {
            const V4 g = CULL ? cull.exact[idx] : scene.geom[idx];    // CULL: device order
            const V4 m0 = CULL ? cull.mat0[idx] : scene.mat0[idx];
            const V4 m1 = CULL ? cull.mat1[idx] : scene.mat1[idx];
            HitRec<T> rec;
            make_hitrec<T>({g.x, g.y, g.z}, m0.x, ro, rd, t_hit, rec);
            kind = (int)m0.z;
            const DielConst<T> dc = {m0.w, m1.x, m1.y};
            todo = scatter_begin<T>(rng, kind, m0.y, rd, rec, vec, vscale_, &dc);
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
                trand(rng, r1); du = RTW_DIV(r1, w_div);
                trand(rng, r2); dv = RTW_DIV(r2, h_div);
            }
            const JobSlot *S = sh->slot((ref_depth & RTW_REF_MASK) >> 4, P.slot_stride);
            const unsigned px = ref_depth & 15u;
            const unsigned rs = P.rows_shift;
            su = (T)S->uv[px >> rs] + du;                              // T(j / W) + du,       src/render.jl:26,37
            sv = (T)S->uv[4 + (px & ((1u << rs) - 1u))] + dv;          // T((H - i) / H) + dv, src/render.jl:27,37
            new_sample = true;
            jitter = true;
            samples_left -= 1;
        }
        n_samples += (unsigned long long)__popcll(__ballot(new_sample));
        if (PROFILE && __any(new_sample)) clk.count(24, 1u);                                             // B executed

        // ---- (R) ONE rejection loop for every lane that needs a random point: unit ball for
        //      Lambertian / Metal scatter (src/rand.jl:15-22), unit disk for the lens (:31-38) ----
        const bool ball = todo == PATH_BALL;
        V3<T> rp = {0, 0, 0};
        T len2 = 0;
        {
            bool pending = ball || new_sample;
            RTW_PROBE_REJECT_TWICE();
            // (wave priority, Float32 matrix-pipe kernels -- see hit_world_mfma: this loop is pure VALU work, a filler like the block
            //  loop; 364.0 -> 363.0 ms)
            // (Rejected in round 6, in git history: ending the loop when one lane is still sampling and letting that lane continue its
            //  trials in the next iteration -- 6.35 -> 5.05 trials per iteration, but 371.2 -> 378.1 ms plain, 285.9 -> 293.4 ms group
            //  cull, 277.5 -> 282.2 ms Float64: the parked lane sits out a scan and the mask bookkeeping costs 140 scalar instructions.)
            constexpr bool use_prio = MFMA && sizeof(T) == 4 && RTW_SCAN_PRIO != 0;
            if (use_prio) __builtin_amdgcn_s_setprio(0);
#ifdef RTW_PROBE_REJ_CAP     // (rtw_probes.hpp)
            for (int rr = 0; rr < RTW_PROBE_REJ_CAP && pending; ++rr) {
                len2 = reject_trial<T>(rng, ball, rp);
                pending = !(len2 <= T(1));
            }
            if (pending) len2 = T(1);
#else
            if constexpr (PROFILE) {                     // (the same loop with a wave-uniform trip: R executed, its trials, the trials with a third draw)
                if (__any(pending)) clk.count(17, 1u);
                while (__any(pending)) {
                    clk.count(16, 1u);
                    if (__any(ball && pending)) clk.count(29, 1u);
                    if (pending) { len2 = reject_trial<T>(rng, ball, rp); pending = !(len2 <= T(1)); }
                }
            }
            while (pending) {
                len2 = reject_trial<T>(rng, ball, rp);
                pending = !(len2 <= T(1));
            }
#endif
            if (use_prio) __builtin_amdgcn_s_setprio(1);
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
        if (PROFILE && __any(todo == PATH_NORM)) clk.count(25, 1u);                                      // F's normalize executed
        if (todo == PATH_NORM) rd = normalize(vec);
        if (ball || new_sample || todo == PATH_NORM) has_ray = ref_depth > RTW_REF_MASK;   // depth <= 0: ray_color returns 0 (src/ray_color.jl:15)
        clk.lap(1);
        if (!__any(has_ray)) __builtin_amdgcn_s_sleep(2);    // every lane waits for a job slot: do not hammer LDS
    }

    if (PROFILE && lane == 0) {
        for (int k = 0; k < 32; ++k) atomicAdd(&ctr->phase[k], clk.acc[k]);
    }
    if (lane == 0) {
        // A global atomic is a 32-byte write at the memory side: the waves of a workgroup add up in LDS and the last one to
        // finish reports for all four.
        __hip_atomic_fetch_add(&sh->fin_segments, n_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __hip_atomic_fetch_add(&sh->fin_samples, n_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if ((threadIdx.x >> 6) < 4u) __hip_atomic_store(&sh->t_wave[4u + (threadIdx.x >> 6)], wall_clock64(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const unsigned nw = blockDim.x >> 6;                                     // waves of this workgroup (t_wave holds up to four)
        if (__hip_atomic_fetch_add(&sh->fin_waves, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == nw - 1u) {
            atomicAdd(&ctr->segments, __hip_atomic_load(&sh->fin_segments, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            atomicAdd(&ctr->samples, __hip_atomic_load(&sh->fin_samples, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            // the end-of-queue drain (RTW_DRAIN_PROFILE only): first wave start, last wave end, sum of the wave ends, waves by end time
            if (P.drain_profile) {
            unsigned long long t_start = ~0ull, t_end = 0ull, t_sum = 0ull;
            for (unsigned w = 0; w < nw && w < 4u; ++w) {
                const unsigned long long a = __hip_atomic_load(&sh->t_wave[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const unsigned long long e = __hip_atomic_load(&sh->t_wave[4u + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                t_start = a < t_start ? a : t_start; t_end = e > t_end ? e : t_end; t_sum += e;
            }
            const unsigned long long t0_seen = atomicMin(&ctr->t_first, t_start);
            atomicMax(&ctr->t_last, t_end);
            atomicAdd(&ctr->t_end_sum, t_sum);
            atomicAdd(&ctr->n_waves, (unsigned long long)nw);
            const unsigned long long t0 = t0_seen < t_start ? t0_seen : t_start;
            for (unsigned w = 0; w < nw && w < 4u; ++w) {
                const unsigned long long bin = (__hip_atomic_load(&sh->t_wave[4u + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - t0) / 25000ull;
                atomicAdd(&ctr->end_hist[bin < 4095ull ? (unsigned)bin : 4095u], 1u);
            }
            }
        }
    }
}

}  // namespace rtw
