// rtw_kernels.hpp -- the trace kernel (render -> ray_color -> hit/scatter fused) and the
// finalize kernel.  gfx950 only; wave = 64 lanes.
//
// Work decomposition
//   item  = (pixel, sample chunk): `chunk_spp` consecutive samples of one pixel drawn from the
//           item's own Xoroshiro128+ stream.  Items are enumerated tile-major: 64 consecutive
//           items are one 8x8 pixel tile for one chunk, so a wave starts out coherent.
//   lane  = persistent worker.  It owns one item at a time, regenerates a camera ray the moment
//           its path ends (no lock-step with the other lanes' path lengths), and pulls the next
//           item from a global queue when its chunk is finished.  The only convergent part is
//           the sphere scan, which every live lane executes every iteration.
//   out   = per-item chunk sums (3 doubles) -> finalize kernel adds them in chunk order, divides
//           by spp, applies gamma and stores RGB{T}.  Results do not depend on scheduling, grid
//           size or shard count.
#pragma once
#include "rtw_device.hpp"

namespace rtw {

struct KParams {
    int width, height, spp, max_depth;
    uint64_t seed;
    int n_chunks;      // effective (non-empty) chunks per pixel
    int chunk_spp;
    int shard_index, shard_count;
    int tiles_i, tiles_j;  // 8x8 tiles along rows (i) and columns (j)
    int n_local_tiles;
    unsigned total_items;  // n_local_tiles * n_chunks * 64
    // exact unsigned division by the two loop-invariant divisors of the item decode (host: make_udiv):
    // n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> s   for every 32-bit n
    unsigned div_chunks_m, div_chunks_s, div_tiles_m, div_tiles_s;
    int gamma;
};

struct DevCounters {
    unsigned long long next_item;
    unsigned long long segments;
    unsigned long long samples;
    unsigned long long phase[8];   // RTW_PHASE_PROFILE=1 only: wave-cycles per phase (s_memtime)
};

// Phase profiler (opt-in instantiation, never used for timed runs): s_memtime stamps around
// the phases of the lane loop, summed per wave.
template <bool ON> struct PhaseClock {
    unsigned long long t0 = 0, acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    __device__ __forceinline__ void start() { if (ON) t0 = __builtin_readcyclecounter(); }
    __device__ __forceinline__ void lap(int k) {
        if (ON) { unsigned long long t = __builtin_readcyclecounter(); acc[k] += t - t0; t0 = t; }
    }
};

__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned m, unsigned s) {
    if (s & 0x80000000u) return n;                      // divisor 1
    const unsigned t = __umulhi(n, m);
    return (t + ((n - t) >> 1)) >> s;
}

__device__ __forceinline__ unsigned lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// waves per SIMD the trace kernel is compiled for (second __launch_bounds__ argument):
// Float32 -> 7 (VGPR cap 72), Float64 -> 4 (cap 128; the double state does not fit lower caps)
#ifndef RTW_TRACE_WAVES_F32
#define RTW_TRACE_WAVES_F32 7
#endif
template <typename T> struct TraceWaves { static constexpr int value = RTW_TRACE_WAVES_F32; };
template <> struct TraceWaves<double> { static constexpr int value = 4; };
// the group-cull variant keeps the slab-test constants live across the scan: fewer waves, no spills
#ifndef RTW_TRACE_WAVES_CULL_F32
#define RTW_TRACE_WAVES_CULL_F32 5
#endif
template <typename T, bool CULL> struct TraceWavesOf { static constexpr int value = TraceWaves<T>::value; };
template <> struct TraceWavesOf<float, true> { static constexpr int value = RTW_TRACE_WAVES_CULL_F32; };
template <> struct TraceWavesOf<double, true> { static constexpr int value = 3; };
#ifndef RTW_ITEM_BATCH
#define RTW_ITEM_BATCH 64u   // work items a wave takes from the global queue per atomic
#endif

template <typename T, bool PROFILE, bool LDS_SCENE, bool CULL>
__global__ __launch_bounds__(256, (TraceWavesOf<T, CULL>::value)) void trace_kernel(KParams P, Camera<T> cam, DevScene<T> scene,
                                                   CullScene<T> cull, const T *__restrict__ puv,
                                                   double *__restrict__ partial, DevCounters *ctr) {
    using V4 = typename Vec4<T>::type;
    const unsigned lane = lane_id();
    // LDS: [per-lane candidate lists, stride 256][scene geom copy (LDS_SCENE only)]
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned short *my_list = reinterpret_cast<unsigned short *>(smem) + threadIdx.x;
    V4 *lds_geom = reinterpret_cast<V4 *>(smem + RTW_LIST_CAP * 256 * sizeof(unsigned short));
    unsigned short *lds_orig = reinterpret_cast<unsigned short *>(lds_geom + (CULL ? cull_exact_count(cull) : 0));
    if (LDS_SCENE) {
        if (CULL) stage_cull_scene<T>(cull, lds_geom, lds_orig);
        else stage_scene<T>(scene, lds_geom);
        __syncthreads();
    }
    // the wave's local pool of work items [pool_next, pool_end): one global atomic per batch
    unsigned pool_next = 0, pool_end = 0;

    // ---- per-lane state ----
    bool alive = true;        // still pulling work
    bool have_item = false;   // owns an item whose chunk sum is not yet flushed
    bool has_ray = false;     // a path is in flight
    unsigned item_slot = 0;   // where the chunk sum goes: (local pixel)*n_chunks + chunk
    int samples_left = 0;
    int s_global = 0;         // 0-based sample index within the pixel (sample 0 is un-jittered)
    T pu = 0, pv = 0;         // pixel's (u, v)  (src/render.jl:26-27)
    Rng rng = {1, 2};
    double acc_r = 0, acc_g = 0, acc_b = 0;
    V3<T> ro = {0, 0, 0}, rd = {0, 0, 1};
    double thr_r = 1, thr_g = 1, thr_b = 1;
    int depth_left = 0;
    unsigned my_segments = 0, my_samples = 0;

    const T inv_w_div = (T)(float)P.width;   // f32_image_width  (src/render.jl:16)
    const T inv_h_div = (T)(float)P.height;  // f32_image_height (src/render.jl:17)

    PhaseClock<PROFILE> clk;
    for (;;) {
        clk.start();
        // ---- (A) lanes whose chunk is done flush it and pull the next item ----
        const bool need = alive && !has_ray && samples_left == 0;
        const unsigned long long need_mask = __ballot(need);
        if (need_mask) {
            const unsigned cnt = __popcll(need_mask);
            const unsigned avail = pool_end - pool_next;
            unsigned new_base = 0;
            if (cnt > avail) {                                   // wave-uniform: refill the pool
                unsigned long long got = 0;
                if (lane == 0) got = atomicAdd(&ctr->next_item, (unsigned long long)RTW_ITEM_BATCH);
                got = __shfl(got, 0);
                new_base = got > 0xffffffffull ? 0xffffffffu : (unsigned)got;   // beyond total_items anyway
            }
            if (need) {
                if (have_item) {
                    double *dst = partial + (size_t)item_slot * 3;
                    dst[0] = acc_r; dst[1] = acc_g; dst[2] = acc_b;
                    have_item = false;
                }
                const unsigned rank = __popcll(need_mask & ((1ull << lane) - 1ull));
                const unsigned long long idx = rank < avail ? (unsigned long long)pool_next + rank
                                                            : (unsigned long long)new_base + (rank - avail);
                if (idx >= P.total_items) {
                    alive = false;
                } else {
                    // idx = (k * n_chunks + chunk) * 64 + pl
                    const unsigned pl = (unsigned)idx & 63u, q = (unsigned)idx >> 6;
                    const unsigned k = udiv_magic(q, P.div_chunks_m, P.div_chunks_s);      // local tile
                    const unsigned chunk = q - k * (unsigned)P.n_chunks;
                    const unsigned t = k * (unsigned)P.shard_count + (unsigned)P.shard_index;
                    const int tj = (int)udiv_magic(t, P.div_tiles_m, P.div_tiles_s), ti = (int)(t - (unsigned)tj * (unsigned)P.tiles_i);
                    const int i0 = ti * 8 + (int)(pl & 7u), j0 = tj * 8 + (int)(pl >> 3);  // 0-based
                    if (i0 < P.height && j0 < P.width) {
                        const int i = i0 + 1, j = j0 + 1;                // Julia's 1-based (i, j)
                        pu = puv[j0];                                    // T(j / W),       src/render.jl:26
                        pv = puv[P.width + i0];                          // T((H - i) / H), src/render.jl:27
                        (void)i; (void)j;
                        const unsigned long long pix = (unsigned long long)j0 * (unsigned)P.height + (unsigned)i0;
                        rng_stream(P.seed, pix, chunk, rng);
                        s_global = (int)chunk * P.chunk_spp;
                        const int s_end = min(P.spp, s_global + P.chunk_spp);
                        samples_left = s_end - s_global;
                        item_slot = (k * 64u + pl) * (unsigned)P.n_chunks + chunk;
                        acc_r = acc_g = acc_b = 0.0;
                        have_item = true;
                    }
                    // out-of-image pixel of an edge tile: nothing to do, pull again next round
                }
            }
            if (cnt > avail) {
                pool_next = new_base + (cnt - avail);
                pool_end = new_base > 0xffffffffu - RTW_ITEM_BATCH ? 0xffffffffu : new_base + RTW_ITEM_BATCH;
                if (pool_next > pool_end) pool_next = pool_end;
            } else {
                pool_next += cnt;
            }
        }
        if (!__any(alive)) break;
        clk.lap(0);

        // ---- (B) start the next sample (src/render.jl:29-37) ----
        if (alive && !has_ray && samples_left > 0) {
            T du = 0, dv = 0;
            if (s_global != 0) {
                T r1, r2;
                trand(rng, r1); du = r1 / inv_w_div;
                trand(rng, r2); dv = r2 / inv_h_div;
            }
            get_ray(rng, cam, pu + du, pv + dv, ro, rd);
            thr_r = thr_g = thr_b = 1.0;
            depth_left = P.max_depth;
            has_ray = depth_left > 0;    // depth <= 0: ray_color returns 0 (src/ray_color.jl:15)
            samples_left -= 1;
            s_global += 1;
            my_samples += 1;
        }

        clk.lap(1);
        // ---- (C) closest hit over the whole sphere list (src/hit.jl:38-50) ----
        T t_hit = 0;
        int idx = -1;
        if (has_ray) {
            if (CULL && LDS_SCENE)
                idx = hit_world_cull<T, 256>(cull, (const V4 *)lds_geom, (const unsigned short *)lds_orig, ro, rd, (T)1e-4,
                                             (T)__builtin_huge_val(), t_hit, my_list, clk);
            else if (CULL)
                idx = hit_world_cull<T, 256>(cull, cull.exact, cull.orig, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
            else if (LDS_SCENE)
                idx = hit_world<T, 256>(scene, (const V4 *)lds_geom, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
            else
                idx = hit_world<T, 256>(scene, scene.geom, ro, rd, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list, clk);
            my_segments += 1;
        }

        clk.lap(2);
        // ---- (D) shade (src/ray_color.jl:20-37) ----
        if (has_ray) {
            if (idx < 0) {
                const C3 sky = skycolor(rd);
                acc_r += thr_r * sky.r; acc_g += thr_g * sky.g; acc_b += thr_b * sky.b;
                has_ray = false;
            } else {
                const typename Vec4<T>::type g = CULL ? cull.exact[idx] : scene.geom[idx];    // CULL: device order
                const typename Vec4<T>::type m0 = CULL ? cull.mat0[idx] : scene.mat0[idx];
                const typename Vec4<T>::type m1 = CULL ? cull.mat1[idx] : scene.mat1[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, ro, rd, t_hit, rec);
                V3<T> nd, att;
                scatter<T>(rng, (int)m0.z, {m1.x, m1.y, m1.z}, m0.y, rd, rec, nd, att);
                thr_r = thr_r * (double)att.x; thr_g = thr_g * (double)att.y; thr_b = thr_b * (double)att.z;
                ro = rec.p; rd = nd;
                depth_left -= 1;
                if (depth_left <= 0) has_ray = false;   // recursion bottoms out with 0 radiance
            }
        }
        clk.lap(3);
    }

    if (PROFILE && lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd(&ctr->phase[k], clk.acc[k]);
    }
    // counters (one atomic per lane at the very end; the compiler reduces them per wave)
    atomicAdd(&ctr->segments, (unsigned long long)my_segments);
    atomicAdd(&ctr->samples, (unsigned long long)my_samples);
}

// One thread per local pixel: chunk sums added in chunk order, / spp, gamma, store RGB{T}
// (src/render.jl:40, src/vec.jl:22).  Column-major H x W, as Matrix{RGB{T}}.
template <typename T>
__global__ __launch_bounds__(256) void finalize_kernel(KParams P, const double *__restrict__ partial,
                                                      T *__restrict__ out) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned n_local = (unsigned)P.n_local_tiles * 64u;
    if (gid >= n_local) return;
    const unsigned k = gid >> 6, pl = gid & 63u;
    const unsigned t = k * (unsigned)P.shard_count + (unsigned)P.shard_index;
    const int ti = (int)(t % (unsigned)P.tiles_i), tj = (int)(t / (unsigned)P.tiles_i);
    const int i0 = ti * 8 + (int)(pl & 7u), j0 = tj * 8 + (int)(pl >> 3);
    if (i0 >= P.height || j0 >= P.width) return;
    const double *src = partial + (size_t)gid * (size_t)P.n_chunks * 3;
    double r = 0.0, g = 0.0, b = 0.0;
    for (int c = 0; c < P.n_chunks; ++c) { r += src[c * 3 + 0]; g += src[c * 3 + 1]; b += src[c * 3 + 2]; }
    const double n = (double)P.spp;
    r = r / n; g = g / n; b = b / n;
    if (P.gamma) { r = __builtin_sqrt(r); g = __builtin_sqrt(g); b = __builtin_sqrt(b); }
    T *dst = out + ((size_t)j0 * (size_t)P.height + (size_t)i0) * 3;
    dst[0] = (T)r; dst[1] = (T)g; dst[2] = (T)b;
}

}  // namespace rtw
