// rtw_scan_cull.hpp -- the opt-in group-cull layout (CullScene), its all-VALU scan (hit_world_cull) and the staging of a scene into LDS
// (part of the device side of the hot path, gfx950 only; see rtw_device.hpp for the numerics contract all parts share)
#pragma once
#include "rtw_scan_mfma.hpp"

namespace rtw {

// ==== opt-in accelerated scan (rtw_params.flags & RTW_FLAG_GROUP_CULL) ===========================
// a clearly separate mode.  Results are bit-identical to the plain
// scan; only spheres that provably cannot be hit are skipped.
//   * At upload the spheres are split into a BIG class (tested exactly by every lane, e.g. the
//     r = 1000 ground) and clusters of <= RTW_CULL_GS small spheres (kd median split) with an
//     axis-aligned bounding box each.  Device order: cluster-major, then the big class.
//   * level 1 (wave-uniform, SGPR data, software-pipelined like pass 1): every lane runs a slab
//     test of its RAY (t >= 0) against every cluster box, the box inflated per ray by
//         m = kappa * (|o - Cs| + Rs + 1),   kappa = 2^-8 (Float32) / 2^-22 (Float64).
//     Why this is conservative: sphere_root accepts sphere i only if its contract discriminant is
//     >= 0, and |disc_c - D| <= 20u (|o-c_i| + r_i)^2 in every numerics mode (17.2 u |o-c|^2 + 3.1 u r^2 for the reference's un-fused
//     order, see hit_world_mfma), so the line passes within
//     r_i + 4.5 sqrt(u) (|o-c_i| + r_i) <= r_i + m/2 of c_i; an accepted root is >= tmin > 0, so
//     either the closest approach is in front of the origin (that point is inside the box grown
//     by m/2) or the origin itself is within r_i + m/2 of c_i.  The slab arithmetic adds
//     relative errors of a few u to parameters of size <= |o - Cs| + Rs, far below m/2
//     (9 sqrt(u) = 2.2e-3 < kappa/2 = 1.95e-3 ... kappa covers both with the +1 term); direction
//     components smaller than 1e-9 are replaced by +-1e-9 (moves the ray by < 1e-9 t).
//     All of this assumes a unit direction; hit_world_cull widens m by 2 sqrt(|d|^2 - 1) for the rays that
//     are not (found by comparing both modes at 1080p x 1000 spp: 8 pixels differed in round 1).
//   * level 2 (per lane): the members of every touched cluster are tested with the contract
//     discriminant (sphere data gathered from LDS) and the candidates in front of the ray are
//     pushed to the lane's list.
//   * pass 2 is the same exact root selection, with the order-free acceptance rule: the
//     reference's scan returns the minimum over the spheres of their smallest root in
//     [tmin, inf) and the LAST sphere of the caller's list among exact ties, so a candidate is
//     taken if root < closest, or root == closest and it comes later in the caller's list.
#ifndef RTW_CULL_GS
#define RTW_CULL_GS 16   // spheres per cluster (multiple of 8)
#endif
#define RTW_CULL_BG 4    // cluster boxes per SGPR set: 4 x 8 floats = 2 x s_load_dwordx16
#ifndef RTW_CULL_L2
#define RTW_CULL_L2 4    // cluster members gathered from LDS per step of level 2 (8 costs 16 more VGPRs: spills)
#endif
template <typename T> struct CullScene {
    const T *box;                          // 8 T per cluster: lo.xyz, pad, hi.xyz, pad; padded + tail group
    const typename Vec4<T>::type *exact;   // (cx, cy, cz, r*r), cluster-major then big class
    const unsigned short *orig;            // index in the caller's list
    const typename Vec4<T>::type *mat0;    // device order
    const typename Vec4<T>::type *mat1;
    int n_groups_pad;                      // multiple of 2*RTW_CULL_BG
    int n_big;                             // device indices n_groups_pad*GS .. +n_big-1
    T cs[3], rs;                           // bounding sphere of the small class
    T kappa;
    const uint4 *mf_ops;                   // group cull on the matrix pipe (hit_world_mfma with MfmaCull): operands in this
    const float *mf_box;                   //   device order, one binary32 box per block of 32
    float mf_glo[3], mf_ghi[3];            //   ... and the box of the whole small class (the union of those boxes; an empty class: lo > hi)
    int mf_blocks;
    int n_huge, n_huge_exact;              // spheres of this order tested in-lane, and how many of them exactly (see MfmaCull::n_huge, n_exact)
    CullGrid grid;                         // the block vote (tables behind mf_box)
    int numerics;                          // NUM_* (see DevScene::numerics)
};
template <typename T> __host__ __device__ inline MfmaCull mfma_cull_of(const CullScene<T> &c) {
    return MfmaCull{c.mf_ops, c.mf_box, c.mf_blocks, {(float)c.cs[0], (float)c.cs[1], (float)c.cs[2]}, (float)c.rs * 1.000001f + 1e-30f, (const void *)c.mat0,
                    {c.mf_glo[0], c.mf_glo[1], c.mf_glo[2]}, {c.mf_ghi[0], c.mf_ghi[1], c.mf_ghi[2]}, c.n_huge, c.n_huge_exact, c.grid, reinterpret_cast<const unsigned *>(c.mf_box + 8 * (c.mf_blocks + 1))};
}
template <typename T> __host__ __device__ inline int cull_exact_count(const CullScene<T> &c) { return c.n_groups_pad * RTW_CULL_GS + ((c.n_big + 31) / 32) * 32; }   // (allocated in whole blocks of 32: dead slots behind the BIG class)

__device__ __forceinline__ float t_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ float t_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return __builtin_fmax(a, b); }

template <typename T, int STRIDE, typename SRC, typename ORIG>
__device__ __forceinline__ void resolve_candidates_anyorder(int num, SRC src, const typename Vec4<T>::type *rad, ORIG orig, V3<T> o, V3<T> d, T tmin, T &closest, int &idx,
                                                            const unsigned short *list, int cnt) {
    using V4 = typename Vec4<T>::type;
    int i_next = cnt > 0 ? (int)list[0] : 0;
    V4 s_next = src[i_next];
    for (int c = 0; __any(c < cnt); ++c) {
        const int i = i_next;
        const V4 s = s_next;
        i_next = (c + 1 < cnt) ? (int)list[(c + 1) * STRIDE] : 0;
        s_next = src[i_next];
        if (c < cnt) {
            T hb, disc, root;
            T rr = T(0);
            if (num == NUM_REFERENCE_FMA2) rr = rad[i].x;
            sphere_disc<T>(num, s.x, s.y, s.z, s.w, rr, o, d, hb, disc);
            if (sphere_root<T>(hb, disc, tmin, closest, root)) {       // root in [tmin, closest]
                bool take = true;
                if (root == closest && idx >= 0) take = orig[i] > orig[idx];
                if (take) { closest = root; idx = i; }
            }
        }
    }
}

template <typename T, int STRIDE, typename SRC, typename ORIG, typename CLK = NoClock>
__device__ __forceinline__ int hit_world_cull(const CullScene<T> &w, SRC src, ORIG orig, V3<T> o, V3<T> d, T tmin, T tmax,
                                              T &t_hit, unsigned short *list, CLK &&clk = NoClock()) {
    using V4 = typename Vec4<T>::type;
    constexpr int G = RTW_CULL_BG;
    constexpr int GS = RTW_CULL_GS;
    typedef const T __attribute__((address_space(4))) *cptr;
    cptr gx = (cptr)(uintptr_t)w.exact;
    T closest = tmax;
    int idx = -1, cnt = 0;
    auto push = [&](int i) {                       // lane-local: may run in divergent code
        if (cnt >= RTW_LIST_CAP) {
            resolve_candidates_anyorder<T, STRIDE>(w.numerics, src, w.mat0, orig, o, d, tmin, closest, idx, list, cnt);
            cnt = 0;
        }
        list[cnt * STRIDE] = (unsigned short)i;
        cnt += 1;
    };
    // A root >= tmin > 0 needs half_b <= 0 or disc > half_b^2 (else -half_b + sqrt(disc) <= 0):
    // spheres entirely behind the ray are not even listed.
    auto member = [&](const V4 &sp, int i) {
        T hb, disc;
        T rr = T(0);
        if (w.numerics == NUM_REFERENCE_FMA2) rr = w.mat0[i].x;
        sphere_disc<T>(w.numerics, sp.x, sp.y, sp.z, sp.w, rr, o, d, hb, disc);
        if (!(disc < T(0))) { if (!(hb > T(0)) || disc > hb * hb) push(i); }
    };

    // big class: contract discriminant for every lane (wave-uniform sphere data)
    const int big0 = w.n_groups_pad * GS;
    for (int b = 0; b < w.n_big; ++b) {
        const int i = big0 + b;
        member(V4{gx[4 * i], gx[4 * i + 1], gx[4 * i + 2], gx[4 * i + 3]}, i);
    }

    // per-ray constants of the slab test.  The margin argument above is geometric and needs |d| = 1, but the
    // reference does NOT renormalise a dielectric reflection (src/material.jl:48): along chains of internal
    // reflections s2 = |d|^2 drifts (1 + 1e-3 ... 400 occur about once per 10^9 segments of the headline scene;
    // round 1's cull lost 8 pixels of the 1080p x 1000 spp frame to it).  With eps = s2 - 1 > 0 the contract
    // discriminant accepts spheres within  r + sqrt(eps) |L| + 4.5 sqrt(u) |d| (|o - c| + r)  of the LINE
    // (L = (o - c).d / |d|, |L| <= |o - c|), so the margin grows with the ray's own eps:
    //     m = (kappa * max(1, s2) + 2 sqrt(eps+)) * (|o - Cs| + Rs + 1),   eps+ = max(s2 - 1, 0) + 4u s2
    // (kappa / 2 >= 4.5 sqrt(u) as before; the 4u s2 covers the rounding of s2 itself).  eps < 0 only shrinks
    // the accepted set.  A wildly non-unit ray simply touches every cluster.
    const V3<T> ocs = {o.x - w.cs[0], o.y - w.cs[1], o.z - w.cs[2]};
    const T s2 = dot(d, d);
    const T eps_p = (s2 > T(1) ? s2 - T(1) : T(0)) + (sizeof(T) == 4 ? T(2.4e-7) : T(4.5e-16)) * s2;
    const T margin = (w.kappa * (s2 > T(1) ? s2 : T(1)) + T(2) * t_sqrt(eps_p)) * ((t_sqrt(dot(ocs, ocs)) + w.rs) + T(1));
    auto safe_inv = [](T x) { const T e = T(1e-9); const T y = (x < e && x > -e) ? (x < T(0) ? -e : e) : x; return T(1) / y; };
    const V3<T> inv = {safe_inv(d.x), safe_inv(d.y), safe_inv(d.z)};
    const V3<T> op = {o.x + margin, o.y + margin, o.z + margin};     // lo' - o = lo - (o + m)
    const V3<T> om = {o.x - margin, o.y - margin, o.z - margin};     // hi' - o = hi - (o - m)

    struct Box { T lx, ly, lz, l_, hx, hy, hz, h_; };
    cptr gb = (cptr)(uintptr_t)w.box;
    auto ldb = [](cptr p, int k) -> Box { return Box{p[8 * k], p[8 * k + 1], p[8 * k + 2], p[8 * k + 3], p[8 * k + 4], p[8 * k + 5], p[8 * k + 6], p[8 * k + 7]}; };
    Box A[G], B[G];
#pragma unroll
    for (int k = 0; k < G; ++k) A[k] = ldb(gb, k);
    cptr pw = gb;
    auto test1 = [&](const Box &bx, uint32_t &mask) {       // one cluster box: 23 VALU ops
        const T x0 = (bx.lx - op.x) * inv.x, x1 = (bx.hx - om.x) * inv.x;
        const T y0 = (bx.ly - op.y) * inv.y, y1 = (bx.hy - om.y) * inv.y;
        const T z0 = (bx.lz - op.z) * inv.z, z1 = (bx.hz - om.z) * inv.z;
        const T tn = t_max(t_max(t_min(x0, x1), t_min(y0, y1)), t_min(z0, z1));
        const T tf = t_min(t_min(t_max(x0, x1), t_max(y0, y1)), t_max(z0, z1));
        const T sgn = tf - t_max(tn, T(0));                  // >= 0 <=> the ray (t >= 0) meets the grown box
        mask = __builtin_amdgcn_alignbit(mask, sign_word(sgn), 31);
    };
    for (int base = 0; base < w.n_groups_pad; base += RTW_SPHERE_WORD) {
        uint32_t mask = 0;
        const int left = w.n_groups_pad - base;
        const int npairs = left >= RTW_SPHERE_WORD ? RTW_SPHERE_WORD / (2 * G) : left / (2 * G);
        for (int q = 0; q < npairs; ++q, pw += 2 * G * 8) {
            test1(A[0], mask);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < G; ++k) B[k] = ldb(pw, G + k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < G; ++k) test1(A[k], mask);
            __builtin_amdgcn_sched_barrier(0);
            test1(B[0], mask);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < G; ++k) A[k] = ldb(pw, 2 * G + k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < G; ++k) test1(B[k], mask);
            __builtin_amdgcn_sched_barrier(0);
        }
        clk.lap(2);
        uint32_t m = ~mask;                       // bit 31 = cluster `base`
        if (left < RTW_SPHERE_WORD) m <<= (RTW_SPHERE_WORD - npairs * 2 * G);
        // level 2: each lane expands the clusters its ray can touch, 8 members at a time
        while (__any(m != 0u)) {
            if (m != 0u) {
                const int b = __clz((int)m);
                m &= ~(0x80000000u >> b);
                const int first = (base + b) * GS;
#pragma unroll 1
                for (int h = 0; h < GS; h += RTW_CULL_L2) {
                    V4 sp[RTW_CULL_L2];
#pragma unroll
                    for (int j = 0; j < RTW_CULL_L2; ++j) sp[j] = src[first + h + j];
#pragma unroll
                    for (int j = 0; j < RTW_CULL_L2; ++j) member(sp[j], first + h + j);
                }
            }
        }
        clk.lap(4);
    }
    resolve_candidates_anyorder<T, STRIDE>(w.numerics, src, w.mat0, orig, o, d, tmin, closest, idx, list, cnt);
    clk.lap(5);
    t_hit = closest;
    return idx;
}

template <typename T>
__device__ __forceinline__ void stage_cull_scene(const CullScene<T> &w, typename Vec4<T>::type *dst, unsigned short *dst_orig) {
    const int n = cull_exact_count(w);
    for (int i = threadIdx.x; i < n; i += blockDim.x) { dst[i] = w.exact[i]; dst_orig[i] = w.orig[i]; }
}

// Stage the scene's geom array into LDS (all threads of the block; caller synchronises).
template <typename T>
__device__ __forceinline__ void stage_scene(const DevScene<T> &w, typename Vec4<T>::type *dst) {
    const int n_alloc = scene_geom_alloc(w.n, w.n_pad);
    for (int i = threadIdx.x; i < n_alloc; i += blockDim.x) dst[i] = w.geom[i];
}

}  // namespace rtw
