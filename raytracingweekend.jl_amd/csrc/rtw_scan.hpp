// rtw_scan.hpp -- the device scene and the all-VALU closest-hit scan (hit_world)
// (part of the device side of the hot path, gfx950 only; see rtw_device.hpp for the numerics contract all parts share)
#pragma once
#include "rtw_path.hpp"

namespace rtw {

// ---- device scene ----------------------------------------------------------------------------
// geom[i] = (cx, cy, cz, r*r)   hot: 16 B (f32) / 32 B (f64) per sphere, wave-uniform reads
// mat0[i] = (r, param, kind, 1 / ir)  cold: read once per segment by the lane that hit sphere i
// mat1[i] = (ar, ag, ab, 0); for a Dielectric (its albedo is never read: attenuation is 1) (r0 front, r0 back, 0, 0)
// geom is padded to a multiple of G spheres (one scalar-load group) plus one prefetch group
// with spheres that can never be hit (r*r = -1e30 => discriminant < 0 always).
#define RTW_SPHERE_WORD 32
#define RTW_SPHERE_TAIL 8
// entries of the geom / mat arrays: the padded scan groups + one prefetch group, and at least whole blocks of 32
// (hit_world_mfma can list the padding spheres of its last block for a ray that takes every sphere)
__host__ __device__ inline int scene_geom_alloc(int n, int n_pad) {
    const int a = n_pad + RTW_SPHERE_TAIL, b = ((n + 31) / 32) * 32;
    return a > b ? a : b;
}
template <typename T> struct DevScene {
    const float *scan;   // what pass 1 streams through scalar loads (binary32 for BOTH precisions):
                         //   Float32: geom itself, 4 floats per sphere (cx, cy, cz, r^2) -- the exact contract discriminant;
                         //   Float64: 8 floats per sphere (cx, cy, cz, r^2, G, 0, 0, 0) rounded to binary32, for the
                         //   conservative binary32 filter of hit_world (G = the sphere's share of the error margin)
    const typename Vec4<T>::type *geom;
    const typename Vec4<T>::type *mat0;
    const typename Vec4<T>::type *mat1;
    int n, n_pad;   // n_pad: multiple of ScanGroup<T>::N (the tail group lies beyond n_pad)
    // pass 1 on the matrix pipe (hit_world_mfma): per block of 32 spheres two A operands of v_mfma_f32_32x32x16_f16
    // (64 lanes x 16 B each: [P1][P2]), one more block of padding for the prefetch; see the derivation there
    const uint4 *mf_ops;
    int mf_blocks;          // ceil(n / 32)
    float mf_sc;            // power of two: lengths are scaled by it before they are split into f16 pieces
    float mf_sigma2;        // mf_sc^2
    float mf_oo_keep;       // 1 - (the ray's share of the relative margin)
    float mf_o1_coef;       // absolute margin per unit of |o|_1
    float mf_o_max;         // rays with a larger |o_k| (or non-unit, non-finite ones) take every sphere as a candidate
    int n_huge, huge[2];    // spheres tested exactly by every lane instead of through the filter (a ground sphere: candidate of nearly every ray)
    int numerics;           // NUM_*: the deciding arithmetic of sphere_disc for this render (set per launch, not per upload)
};

// Candidate lists: pass 1 of the scan appends the indices of the spheres whose discriminant is
// >= 0 to a per-lane list in LDS; pass 2 resolves them in ascending sphere order.
#define RTW_LIST_CAP 16     // entries per lane (u16); a full list is resolved early (wave-wide)
// Scenes up to this many bytes of geom are also staged in LDS so that pass 2 gathers its
// candidates' spheres from LDS (latency ~100 cycles) instead of global memory (~700).
#define RTW_LDS_SCENE_MAX_BYTES (24 * 1024)

__device__ __forceinline__ uint32_t sign_word(float x) { return __float_as_uint(x); }
__device__ __forceinline__ uint32_t sign_word(double x) { return (uint32_t)((uint64_t)__double_as_longlong(x) >> 32); }

// Spheres per scalar load of the all-VALU scan.  Two groups are in SGPRs at a time (one being tested, one in flight): 2 x 16
// registers each.  Float32 used groups of 8 (2 x 32 SGPRs of the 102 a wave has) until round 3: every other long-lived scalar
// of the kernel then competes for ~30 registers, and whether the allocator spilled them around the scan loop or INSIDE it (47
// v_readlane / v_writelane per 16 spheres, +30 % kernel time) changed with unrelated edits to the kernel's epilogue.  With
// groups of 4 the loop has no spill code at all and runs 10 % faster than the best groups-of-8 build (255 vs 286 ms at 1080p x
// 300 spp); the 33 VALU instructions between a load and its use are plenty at 7 waves per SIMD.
#ifndef RTW_SCAN_PRIO
#define RTW_SCAN_PRIO 1   // wave priorities of the Float32 matrix-pipe kernels (hit_world_mfma); 0: no s_setprio at all (A/B)
#endif
template <typename T> struct ScanGroup;
template <> struct ScanGroup<float> { static constexpr int N = 4; };    // 4 x 16 B = 1 x s_load_dwordx16
template <> struct ScanGroup<double> { static constexpr int N = 4; };   // 4 x 8 floats = 2 x s_load_dwordx16

struct NoClock { static constexpr bool on() { return false; } __device__ __forceinline__ void lap(int) {} __device__ __forceinline__ void count(int, unsigned) {} };

// src/hit.jl:38-50 -- closest hit by linear scan over ALL spheres; `closest` shrinks; a later
// sphere wins an exact tie.  Same results as the plain loop, organised for the wave:
//   pass 1  (branch-free, every lane, every sphere): the discriminant of src/hit.jl:13-18 from
//           wave-uniform sphere data held in SGPRs (scalar loads, prefetched one group ahead);
//           its sign bit is shifted into a 32-sphere mask word with ONE v_alignbit per sphere.
//           disc >= 0  <=>  sign bit clear (disc is never -0: hb*hb >= +0; NaN cannot occur for
//           finite scenes).  After each word the few candidate indices go to the lane's LDS list.
//   pass 2  (every lane walks its own list, ascending sphere index): the exact root selection
//           of src/hit.jl:19-29 against the shrinking `closest`.  Sphere order is preserved, so
//           ties resolve exactly as in the reference.  `src` is the scene copy in LDS (or the
//           global array for scenes too large for LDS); the loop is software-pipelined: entry
//           c+1's index and sphere are fetched while entry c is tested.
template <typename T, int STRIDE, int NUM, typename SRC>
__device__ __forceinline__ void resolve_candidates_n(SRC src, const typename Vec4<T>::type *rad, V3<T> o, V3<T> d, T tmin, T &closest, int &idx,
                                                     const unsigned short *list, int cnt) {
    using V4 = typename Vec4<T>::type;
    auto test = [&](int c, int i, const V4 &s) {
        if (c < cnt) {
            T hb, disc, root, r = T(0);
            if constexpr (NUM == NUM_REFERENCE_FMA2) r = rad[i].x;           // (mat0[i].x: the radius itself)
            sphere_disc_n<T, NUM>(s.x, s.y, s.z, s.w, r, o, d, hb, disc);
            if (sphere_root<T>(hb, disc, tmin, closest, root)) { closest = root; idx = i; }
        }
    };
    // two entries per trip, fetched one ahead, in two fixed register sets (no copies between trips)
    int ia = cnt > 0 ? (int)list[0] : 0;
    V4 sa = src[ia];
    for (int c = 0; __any(c < cnt); c += 2) {
        const int ib = (c + 1 < cnt) ? (int)list[(c + 1) * STRIDE] : 0;
        const V4 sb = src[ib];
        test(c, ia, sa);
        ia = (c + 2 < cnt) ? (int)list[(c + 2) * STRIDE] : 0;
        sa = src[ia];
        if (__any(c + 1 < cnt)) test(c + 1, ib, sb);
    }
}
// (the numerics mode is wave-uniform: one scalar branch per call, not per candidate)
template <typename T, int STRIDE, typename SRC>
__device__ __forceinline__ void resolve_candidates(int num, SRC src, const typename Vec4<T>::type *rad, V3<T> o, V3<T> d, T tmin, T &closest, int &idx,
                                                   const unsigned short *list, int cnt) {
    if (num == NUM_REFERENCE) resolve_candidates_n<T, STRIDE, NUM_REFERENCE>(src, rad, o, d, tmin, closest, idx, list, cnt);
    else if (num == NUM_CONTRACT) resolve_candidates_n<T, STRIDE, NUM_CONTRACT>(src, rad, o, d, tmin, closest, idx, list, cnt);
    else resolve_candidates_n<T, STRIDE, NUM_REFERENCE_FMA2>(src, rad, o, d, tmin, closest, idx, list, cnt);
}

template <typename T, int STRIDE, typename SRC, typename CLK = NoClock>
__device__ __forceinline__ int hit_world(const DevScene<T> &w, SRC src, V3<T> o, V3<T> d, T tmin, T tmax, T &t_hit,
                                         unsigned short *list, CLK &&clk = NoClock()) {
    constexpr int G = ScanGroup<T>::N;
    constexpr bool F64 = sizeof(T) == 8;
    constexpr int SW = F64 ? 8 : 4;                              // floats per sphere in the scan array
    typedef const float __attribute__((address_space(4))) *cptr; // constant address space: SMEM loads
    cptr gs = (cptr)(uintptr_t)w.scan;
    struct Unit { float v[SW]; };
    auto ldg = [&](int i) -> Unit {
        Unit r;
#pragma unroll
        for (int j = 0; j < SW; ++j) r.v[j] = gs[SW * i + j];
        return r;
    };
    T closest = tmax;
    int idx = -1, cnt = 0;
    Unit A[G], B[G];
#pragma unroll
    for (int k = 0; k < G; ++k) A[k] = ldg(k);
    // Pass 1 only has to produce a SUPERSET of {spheres whose contract discriminant is >= 0}: pass 2 applies the
    // exact test to every candidate.
    //   Float32: the discriminant itself, in the render's numerics mode (contract form: 10 VALU + 1 v_alignbit per sphere; the
    //   reference's un-fused form: 16 + 1).
    //   Float64: a conservative binary32 FILTER (12 VALU + 1 v_alignbit; an FP64 instruction costs two issue slots,
    //   the exact form would be 10 x 2 + 1).  With o, c, d, r^2 rounded to binary32 (u = 2^-24) and the same
    //   operation order, the computed  W = fma(hb, hb, fma(nc, 1 - 2^-18, G))  satisfies
    //       W >= disc + 2^-18 |o - c|^2 + (G - 2^-18 r^2) - Err,
    //       Err <= u [28.5 |o - c|^2 + 12.2 |c|^2 + 6.1 r^2 + 2 G]          for |d|^2 <= 1.001
    //   (input rounding a = u (|o| + |c| + |o - c|) per component of o - c; 2 |hb| d(hb) <= u [11.3 |oc|^2 + 2.03 (|o|^2 +
    //   |c|^2)]; d(nc) <= u [4.01 r^2 + 6.02 |oc|^2 + 2.01 (|o|^2 + |c|^2)]; the two final roundings <= u [2 r^2 +
    //   3.01 |oc|^2 + 2 G]; |o|^2 <= 2 |oc|^2 + 2 |c|^2).  2^-18 = 64 u > 28.5 u, and the upload sets
    //   G = 1.01 (2^-18 r^2 + 2^-20 |c|^2 + 2^-20 r^2) + 1e-30 (rounded up), so  disc >= 0  =>  W > 0: sign bit clear.
    //   The binary64 roundings of the deciding discriminant itself (<= 20 * 2^-53 (|oc| + r)^2 in every numerics mode) vanish in the slack.  Rays that
    //   are not (nearly) unit, not finite or astronomically far take every sphere as a candidate (lane_ok).
    [[maybe_unused]] bool lane_ok = true;
    [[maybe_unused]] V3<float> of = {0, 0, 0}, df = {0, 0, 1};
    if constexpr (F64) {
        const double s2 = dot(d, d), o2 = dot(o, o);
        lane_ok = s2 <= 1.001 && o2 < 1e30;                       // (false for NaN)
        of = {(float)o.x, (float)o.y, (float)o.z};
        df = {(float)d.x, (float)d.y, (float)d.z};
    }
    auto test1 = [&](auto tag, const Unit &sp, uint32_t &mask) {
        if constexpr (F64) {
            const float ocx = of.x - sp.v[0], ocy = of.y - sp.v[1], ocz = of.z - sp.v[2];
            const float hb = __builtin_fmaf(ocz, df.z, __builtin_fmaf(ocy, df.y, ocx * df.x));
            const float nc = __builtin_fmaf(-ocz, ocz, __builtin_fmaf(-ocy, ocy, __builtin_fmaf(-ocx, ocx, sp.v[3])));
            const float m = __builtin_fmaf(nc, 0.999996185302734375f /* 1 - 2^-18 */, sp.v[4]);
            const float W = __builtin_fmaf(hb, hb, m);
            mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(W), 31);
        } else {
            T hb, disc;
            if constexpr (decltype(tag)::value == NUM_REFERENCE_FMA2) {
                // The scalar stream carries r^2, not r: pass 1 evaluates the form with only the last step contracted and adds a margin that covers what the
                // un-rounded square can change -- the two values of c differ by <= u r^2 (the rounding of r r) + 2 u (oc.oc + r^2) (their
                // own roundings), the two final fmas by <= 2 u (half_b^2 + oc.oc + r^2): < 8 u (oc.oc + r^2) = 2^-21 (oc.oc + r^2) in all.
                // A superset is all pass 1 owes; pass 2 decides with the radius itself.
                const T ocx = o.x - sp.v[0], ocy = o.y - sp.v[1], ocz = o.z - sp.v[2];
                hb = (ocx * d.x + ocy * d.y) + ocz * d.z;
                const T ococ = (ocx * ocx + ocy * ocy) + ocz * ocz;
                disc = t_fma(ococ + sp.v[3], T(4.76837158203125e-07), t_fma(hb, hb, -(ococ - sp.v[3])));
            } else {
                sphere_disc_n<T, decltype(tag)::value>(sp.v[0], sp.v[1], sp.v[2], sp.v[3], T(0), o, d, hb, disc);
            }
            mask = __builtin_amdgcn_alignbit(mask, sign_word(disc), 31);
        }
    };
    // (the whole scan loop once per numerics mode: the mode is decided outside the loop, not per sphere)
    auto scan = [&](auto tag) {
    for (int base = 0; base < w.n_pad; base += RTW_SPHERE_WORD) {
        uint32_t mask = 0;
        // the last word may be partial: n_pad is a multiple of one group (G), not of 32
        const int left = w.n_pad - base;
        const int ngroups = (left >= RTW_SPHERE_WORD ? RTW_SPHERE_WORD : left) / G;
        const int npairs = ngroups >> 1;
        for (int q = 0; q < npairs; ++q) {
            const int off = base + q * 2 * G;
            // Scalar loads return out of order, so every wait is lgkmcnt(0).  To keep a group's
            // loads in flight for a whole group of VALU work, the next group's loads are issued
            // right AFTER the wait that the current group's first use forces, never before it.
            test1(tag, A[0], mask);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < G; ++k) B[k] = ldg(off + G + k);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < G; ++k) test1(tag, A[k], mask);
            __builtin_amdgcn_sched_barrier(0);
            test1(tag, B[0], mask);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < G; ++k) A[k] = ldg(off + 2 * G + k);      // next group (tail-padded)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 1; k < G; ++k) test1(tag, B[k], mask);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ngroups & 1) {                        // odd group count: the scene's last group, already in A
#pragma unroll
            for (int k = 0; k < G; ++k) test1(tag, A[k], mask);
        }
        clk.lap(2);
        uint32_t m = ~mask;                       // bit 31 = sphere `base`, bit 0 = sphere base+31
        if constexpr (F64) { if (!lane_ok) m = 0xffffffffu; }                // no filter for this ray: every sphere
        if (left < RTW_SPHERE_WORD) m <<= (RTW_SPHERE_WORD - ngroups * G);   // partial word: align to bit 31
        auto push_first = [&]() {                 // append the lane's first remaining candidate of this word
            const int b = __clz((int)m);
            list[cnt * STRIDE] = (unsigned short)(base + b);
            cnt += 1;
            m &= ~(0x80000000u >> b);
        };
        if (!__any(m != 0u)) {
            // no lane has a candidate among these 32 spheres
        } else if (!__any(cnt + (int)__popc(m) > RTW_LIST_CAP)) {
            do { if (m != 0u) push_first(); } while (__any(m != 0u));     // the common case: a tight loop
        } else {
            while (__any(m != 0u)) {
                if (__any(cnt >= RTW_LIST_CAP)) {     // some lane's list is full: resolve all lists now
                    clk.lap(4);
                    resolve_candidates<T, STRIDE>(w.numerics, src, w.mat0, o, d, tmin, closest, idx, list, cnt);
                    cnt = 0;
                    clk.lap(5);
                }
                if (m != 0u) push_first();
            }
        }
        clk.lap(4);
    }
    };
    if constexpr (F64) scan(NumTag<NUM_REFERENCE>{});        // (the binary32 filter does not depend on the mode)
    else if (w.numerics == NUM_REFERENCE) scan(NumTag<NUM_REFERENCE>{});
    else if (w.numerics == NUM_CONTRACT) scan(NumTag<NUM_CONTRACT>{});
    else scan(NumTag<NUM_REFERENCE_FMA2>{});
    resolve_candidates<T, STRIDE>(w.numerics, src, w.mat0, o, d, tmin, closest, idx, list, cnt);
    clk.lap(5);
    t_hit = closest;
    return idx;
}

}  // namespace rtw
