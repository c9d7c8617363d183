// rtw_scan_mfma.hpp -- pass 1 of the closest-hit scan on the matrix pipe (hit_world_mfma), its exact pass 2, and the table vote of the group cull
// (part of the device side of the hot path, gfx950 only; see rtw_device.hpp for the numerics contract all parts share)
#pragma once
#include "rtw_scan.hpp"
#include "rtw_cull_tables.hpp"
#include "rtw_probes.hpp"

namespace rtw {

// ==== pass 1 on the matrix pipe ====================================================================================
// The discriminant of src/hit.jl:13-18, D = (d.(o - c))^2 - |o - c|^2 + r^2, expanded around the ray's scalar q = d.o and
// the vector p = o - q d (an algebraic identity, no unit-length assumption):
//     D = (d.c)^2 + 2 p.c + (q^2 - |o|^2) + (r^2 - |c|^2)
// Every term is BILINEAR in (ray features) x (sphere features) -- the square (d.c)^2 = sum_ij (d_i d_j)(c_i c_j) through its six
// distinct products -- so ONE K = 32 contraction gives the whole filter value
//     W = [2dx^2 2dy^2 2dz^2 4dxdy 4dxdz 4dydz | 2p | 1 | q^2 - oo'] . [cx^2 cy^2 cz^2 cxcy cxcz cycz (x 1/2) | c | k' | 1]
// (k' = r^2 - |c|^2 + Gs, oo' = |o|^2 - Gr: the sphere's and the ray's shares of the error margin), computed by two chained
// v_mfma_f32_32x32x16_f16 (the second accumulates onto the first) for 32 spheres x 32 rays.  The VALU is left with ONE
// instruction per (ray, sphere) -- the v_alignbit that collects the sign -- instead of the 11 instructions of hit_world's
// pass 1 (round 2 formed W = P1^2 + P2 from two separate products: one v_fma_f32 more per test, 22 % of the kernel's VALU
// instructions, and 16 more result registers).  An MFMA and VALU instructions do not overlap on a SIMD, whichever wave they
// come from and however they are interleaved (tools/ubench_mfma_overlap.hip, tools/ubench_mfma_pipe.hip; in the kernel itself:
// every MFMA pair executed twice / three times costs +38 % / +80 %, profiles/r05_probe_phases.txt): the scan costs
// the SUM of its MFMA and VALU issue time, so the instruction count is what there is to gain.
// The result is only a FILTER, like the binary32 filter of hit_world<double>: pass 2 applies the exact test of the render's numerics
// mode (sphere_disc) to every candidate, so pass 1 must flag a SUPERSET of {deciding discriminant >= 0} for every mode.
// Precision.  Every f32 feature x is split into two f16 pieces x = p1 + p2 + e, |e| <= eta |x| + phi (eta = 2^-22; phi = 2^-25:
// the floor once a piece is an f16 subnormal -- the instruction honours subnormal inputs, tools/ubench_mfma_f16_numerics.hip)
// and three K slots hold the cross terms a1 b1, a1 b2, a2 b1 of a feature pair (a2 b2 <= 2^-22 |a b| is dropped), so every
// product is exact in the f32 accumulator; the measured accumulation error of one MFMA is <= 2^-21.8 x max|term| (the budget
// assumes beta = 2^-20 x (sum |terms| + |C|)).  Scales: lengths by the power of two s = mf_sc (|c_k| s <= 2^8; |o_k| s <= 2^13
// for a ray that uses the filter: up to 32 x the scene's extent), so the linear features are 2 p_k s <= 2^15.5 and c_k s; the
// quadratic features are 2 m d_i d_j (<= 2.002) and c_i c_j s^2 / 2 (<= 2^15); the two LARGE constants are split over two
// scales so that no small piece lands in the f16 subnormal range:
//     k' s^2  = 2^15 k1 + 2^4 k2                     against the ray-side constants (2^15, 2^4)  (0 for a ray that is not ok)
//     (q^2 - oo') s^2 = 2^15 t1 + 2^4 (t2 + t3)       against the sphere-side constants (2^15, 2^4, 2^4)
// 18 + 9 + 2 + 3 = 32 slots:
//     slot  0-7   (MFMA 1, lanes 0-31)   xx xx xx yy yy yy zz zz        ray pieces (1 1 2 | 1 1 2 | 1 1)   sphere pieces (1 2 1 | 1 2 1 | 1 2)
//     slot  8-15  (MFMA 1, lanes 32-63)  zz xy xy xy xz xz xz yz        (2 | 1 1 2 | 1 1 2 | 1)            (1 | 1 2 1 | 1 2 1 | 1)
//     slot 16-23  (MFMA 2, lanes 0-31)   yz yz px px px py py py        (1 2 | 1 1 2 | 1 1 2)              (2 1 | 1 2 1 | 1 2 1)
//     slot 24-31  (MFMA 2, lanes 32-63)  pz pz pz k k T T T             (1 1 2 | 2^15 2^4 | t1 t2 t3)      (1 2 1 | k1 k2 | 2^15 2^4 2^4)
// Error budget in unscaled units, in multiples of 2^-22 (u = 2^-24 = 0.25, |d|^2 <= 1.001, S = (|o| + |c|)^2, |p| <= |o|):
//     splits of the quadratic features (3.01 x sum |terms| <= 1.001 |c|^2)             3.02 |c|^2
//     splits of the linear features (3.01 x 2 |p| |c|)                                 6.03 |o| |c|
//     roundings of the features themselves (d_i d_j, c_i c_j, p_k)                      0.5 |c|^2 + 0.5 |o| |c|
//     q computed in binary32 (3.01 u |o|) against 2 |hb| <= 2.001 (|o| + |c|)            3.01 (|o|^2 + |o| |c|)
//     |o|^2, q^2 and the fma that forms q^2 - oo' in binary32                          1.3 |o|^2
//     two MFMAs, beta x (sum |terms| + |C|)                     12.01 |c|^2 + 8.01 |o| |c| + 4.01 |o|^2 + 4 r^2 + 4 Gs
//     the deciding discriminant vs exact arithmetic (binary32), in EVERY numerics mode (sphere_disc_n):          4.28 S + r^2
//         contract form (three FMA chains)          15 u |o - c|^2 + 4 u r^2
//         reference order (round 5, the default)    17.2 u |o - c|^2 + 3.1 u r^2:  with e = o - c and e^ its rounded components (u |e_k| each),
//             half_b = fl(fl(fl(e1 d1) + fl(e2 d2)) + fl(e3 d3)) is within 3 u |e||d| of e^.d and e^.d within u |e||d| of e.d: 4.01 u |e|; its square
//             8.06 u |e|^2 + the rounding of the product 1.01 u |e|^2 (absent with disc = fma(half_b, half_b, -c)); oc.oc 3 u |e|^2 + 2.01 u |e|^2;
//             r r: u r^2;  c = fl(oc.oc - r^2): 1.01 u (|e|^2 + r^2);  disc = fl(half_b^2 - c): 2.03 u |e|^2 + 1.01 u r^2
//     inputs rounded from binary64 (hit_world_mfma<double>)                            1.5 S
// With |o| |c| <= (|o|^2 + |c|^2) / 2 and S <= 2 |o|^2 + 2 |c|^2:  E <= 2^-22 (35.9 |c|^2 + 28.7 |o|^2 + 5 r^2) + floors,
// floors <= phi_c (5.5 |o|_1 + |c|_1) + 1.4 phi_k, phi_c = 2^-25 / s (second pieces of the linear features; |p|_1 <= 2.74 |o|_1),
// phi_k = 2^-20 / s^2 (the 2^4-scaled pieces of k' and q^2 - oo', the quadratic features' second pieces).  The margin separates:
// the upload adds  Gs = 1.02 [(2 A_S + A_r)|c|^2 + A_r r^2 + 9 phi_c |c|_1 + 1.5 phi_k]  to k' (A_S = 32 x 2^-22 = 2^-17, A_r = 12 x
// 2^-22: the round-2 constants, kept although this formulation needs only 36 / 29 / 5 of the 76 / 64 / 12 they provide -- no
// error is amplified by a squaring any more) and the ray subtracts  oo' = |o|^2 (1 - 1.02 x 2^-16) - 9.18 phi_c |o|_1
// (mf_oo_keep, mf_o1_coef), so that  deciding discriminant >= 0 (any numerics mode)  =>  W > 0: sign bit clear.
// Rays that are not (nearly) unit (the reference does not renormalise dielectric reflections), not finite, or farther
// than 2^13 / s from the origin take EVERY sphere as a candidate (all features 0, t1 = 60000); lanes without a ray take none
// (t1 = -60000).  Padding spheres carry k' s^2 = -2^30.
// Lane layout of the instruction (A: row l & 31, k = 8 (l >> 5) + e; C/D: col l & 31, row (reg & 3) + 8 (reg >> 2) +
// 4 (l >> 5)): lanes l and l + 32 hold the SAME 32 rays of a half wave and different spheres, so the candidates go to a
// wave-shared list in LDS and pass 2 walks that list 64 candidates at a time whatever their owner (no lane waits for the
// longest per-lane list any more).  Pass 2 is order-free: the reference's scan (src/hit.jl:38-50, closest shrinking, "<="
// acceptance) returns the minimum over the spheres of their first root in [tmin, inf) -- the near root if it is >= tmin,
// else the far root if that is -- and the LAST sphere among exact ties; a sphere whose near root exceeds the running
// closest cannot win with its far root either.  That is the minimum of the 64-bit keys (root bits, ~sphere): one LDS
// atomic min per accepted candidate (Float64: min on the root bits, then max on the index among the candidates equal to it).
#define RTW_PAIR_CAP 512     // 32-bit words of a wave's candidate list area: 160 entries of (bits, lane << 16 | block << 5) + 192 single candidates; a full list is resolved early
typedef _Float16 rtw_h8 __attribute__((ext_vector_type(8)));
typedef float rtw_f16v __attribute__((ext_vector_type(16)));

// group cull on the matrix pipe: operands in the cull layout's device order, one binary32 box per block of 32 (lo.xyz, -, hi.xyz, -)
// The block vote of the group cull (hit_world_mfma<.., CULLED>), per RAY and for 32 blocks at once.  A block can be touched when its box
// overlaps the bounds [lo, hi] of the clipped ray on every axis: lo_b <= hi and hi_b >= lo.  Each axis is cut into RTW_CULL_BINS bins over
// the small class's box (the outer bins reach to infinity); two tables per axis hold, per bin, the 32-bit set of the blocks with
// lo_b <= (upper edge of the bin) and of those with hi_b >= (lower edge): six look-ups and five ANDs give the set of blocks the ray's
// bounds can overlap -- a superset of the exact box test by at most one bin width per side.  The sets of a half wave are ORed on the DPP
// network (4 steps): that is the whole vote, once per scan and group of 32 blocks, instead of 10 VALU instructions per (scan, block).
// Tables (uint32, behind the boxes: box + 8 (blocks + 1)), per group of 32 blocks RTW_CULL_TAB_WORDS words:
//     [axis][0: lo_b <= edge, indexed by the bin of hi | 1: hi_b >= edge, indexed by the bin of lo][bin], then {BIG blocks, live blocks, 0, 0},
//     then RTW_CULL_INLANE_MAX sphere indices (first group: the in-lane list, MfmaCull::n_huge of them)
// (BIG: touched by every ray; live: what a ray without the filter touches; dead blocks are in no set).
// 64 bins: one table = 64 words = one word per LDS bank, any 64 look-ups are conflict-free; 128 bins measured 8 % SLOWER (317 vs 293 ms).
// (RTW_CULL_BINS, RTW_CULL_INLANE_MAX, RTW_CULL_TAB_WORDS: rtw_cull_tables.hpp, with the host code that builds the tables)
__host__ __device__ inline int cull_tab_words(int blocks) { return ((blocks + 31) / 32 > 0 ? (blocks + 31) / 32 : 1) * RTW_CULL_TAB_WORDS; }
struct CullGrid {
    float inv[3], off[3];          // bin of a coordinate p on axis k: floor(p inv[k] + off[k]) clamped to 0 .. RTW_CULL_BINS - 1
};

struct MfmaCull {
    const uint4 *ops;
    const float *box;
    int blocks;
    float cs[3], rs;      // bounding sphere of the small class (the margin grows with the distance to it)
    const void *mat0;     // the cold rows in this (device) order: mat0[i].x = the radius (NUM_REFERENCE_FMA2)
    float glo[3], ghi[3]; // the box of the whole small class (the union of its blocks' boxes): a ray is clipped against it ONCE per scan
    int n_huge;           // spheres (device order) tested in-lane like DevScene::huge -- the huge ones and, when it fits, the whole BIG class: their indices follow the flag words of the first group's tables
    int n_exact;          // ... of which the first n_exact get the EXACT test in every lane (the huge ones: hit by most rays); the others only the
                          //     discriminant -- a lane with disc >= 0 records a list entry, pass 2 does the rest (RTW_INLANE_FILTER)
    CullGrid grid;        // the block vote: bins ...
    const unsigned *tab;  // ... and tables (global memory, or the workgroup's copy in LDS)
};

struct WaveScratch {
    unsigned *pairs;              // the wave's candidate list, 8-byte aligned: entries of two words (see resolve_pairs)
    unsigned long long *keys;     // 64 entries: Float32 (root bits << 32 | ~sphere); Float64 root bits
    unsigned *kidx;               // Float64 only: 64 entries, sphere + 1
    unsigned cap = 5 * RTW_PAIR_CAP / 16; // ENTRIES (two words each) at the start of `pairs`: 160 -- a scan of the headline scene records ~50 ...
    unsigned cap2 = 3 * RTW_PAIR_CAP / 8; // ... and single candidates (one word each) behind them: 192 -- ~70 per scan there; the exact tests run when 128 are waiting or the scan ends.  (The ray-pool kernel gives a wave half the area.)
};

// x = p1 + p2 with p1 = RN16(x), p2 = RN16(x - p1); returns p1 | p2 << 16
// Two instructions: v_cvt_f16_f32 writes p1 to the low half, v_fma_mixhi_f16 computes x * 1.0 - p1 with the f16 operand read in place
// (exact in binary32: p1 is x rounded to 11 bits) and rounds it once into the high half -- the same bits as the five instructions the
// compiler makes of the C form (convert, convert back, subtract, convert, pack): 11 splits per scan, 34 VALU instructions fewer.
__device__ __forceinline__ unsigned split_f16(float x) {
    unsigned w;
    __asm__("v_cvt_f16_f32_e32 %0, %1\n\tv_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "=&v"(w) : "v"(x));
    return w;
}
__device__ __forceinline__ float lane_get(float v, unsigned src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), __float_as_int(v)));
}
__device__ __forceinline__ double lane_get(double v, unsigned src_lane) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(unsigned)b);
    const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(unsigned)((unsigned long long)b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// (Rejected and removed in round 6, in git history: flagging GROUPS of 2 / 4 spheres per list entry -- 380.5 / 388.0 against 370.9 ms, pass 2 pays
// one exact test per member --, and collecting the signs by v_cmp into SGPR lane masks -- +4.7 %.)
#ifndef RTW_INLANE_FILTER
#define RTW_INLANE_FILTER 1     // group cull: the in-lane class beyond the huge spheres gets the discriminant only, candidates go through the list (0: the exact test in every lane, rounds 4 - 5)
#endif
#ifndef RTW_PRECHECK_GROUPS
#define RTW_PRECHECK_GROUPS 1   // groups per half block whose sign collection is skipped separately (hit_world_mfma): 1, 2 or 4
#endif
#ifndef RTW_SCAN_SKIP
#define RTW_SCAN_SKIP 1      // wave-level early-out per half block (hit_world_mfma): 372.2 vs 376.3 ms.  (Left to the compiler it is
                             // if-converted -- both sides executed -- and gains nothing: the sign collection's side is fenced by an asm.)
#endif

// Pass 2 over the wave's list (all 64 lanes; src = the scene's geom in LDS or global memory).  The block loop records ONE entry per (lane, block)
// with a candidate: (x: the lane's candidate bits of the block, bit 31 - b for b = half << 4 | result register; y: recording lane (H, j) << 16 |
// block << 5).  Here the entries are first EXPLODED into single candidates (y | b) in the second half of the list area -- the ballot / ctz /
// mbcnt / write loop that ran once per BLOCK until round 5 (25 - 35 VALU instructions per block with a candidate; now 7 there and this loop
// once per 64 entries) -- and then the exact test runs on full rounds of 64 candidates whatever their owner.  (Walking the bits of an entry in
// the lane that holds it was measured too: 3.3 rounds of exact tests at 33 % lane utilisation per scan instead of 1.5 -- candidates of one ray in
// one block come in runs, the list order of the reference's scene is a row of the lattice.)
struct NoOrig {};
template <typename T, bool WITH_R, typename SRC, typename ORIG = NoOrig>
__device__ __forceinline__ void test_singles(int num, SRC src, [[maybe_unused]] const typename Vec4<T>::type *rad, V3<T> o, V3<T> d, T tmin, const WaveScratch &ws, const unsigned *singles, unsigned n, unsigned lane, ORIG orig, unsigned *prof) {
    constexpr bool CULLED = !__is_same(ORIG, NoOrig);      // device order != the caller's order: ties go by orig[], the key carries both
    using V4 = typename Vec4<T>::type;
    for (unsigned p0 = 0; p0 < n; p0 += 64u) {
        const unsigned p = p0 + lane;
        const bool valid = p < n;
        const unsigned e = singles[valid ? p : 0u];
        if (prof) { prof[0] += 1u; prof[1] += (unsigned)__popcll(__ballot(valid)); }        // (phase-profile build: rounds of exact tests, lanes that had one)
        // candidate = recording lane (H, j) << 16 | block << 5 | b,  b = half << 4 | result register: ray j + 32 (b >> 4), sphere 32 block + 16 H + (b & 15)
        const unsigned owner = ((e >> 16) & 31u) + ((e & 16u) << 1), sph = (e & 0xffefu) + ((e >> 17) & 16u);
        const V3<T> po = {lane_get(o.x, owner), lane_get(o.y, owner), lane_get(o.z, owner)};
        const V3<T> pd = {lane_get(d.x, owner), lane_get(d.y, owner), lane_get(d.z, owner)};
        const V4 s = src[sph];
        T hb, disc, root = 0;
        if constexpr (WITH_R) sphere_disc_n<T, NUM_REFERENCE_FMA2>(s.x, s.y, s.z, s.w, rad[sph].x, po, pd, hb, disc);   // (mat0: the radius itself; the LDS copy holds r^2)
        else sphere_disc<T>(num, s.x, s.y, s.z, s.w, T(0), po, pd, hb, disc);
        const bool hit = valid && sphere_root<T>(hb, disc, tmin, (T)__builtin_huge_val(), root);
        unsigned tie = sph;                                  // larger = later in the caller's list
        if constexpr (CULLED) tie = ((unsigned)orig[sph] << 16) | sph;
        if constexpr (sizeof(T) == 4) {
            if (hit) {
                const unsigned low = CULLED ? ((0xffffu - (tie >> 16)) << 16) | (tie & 0xffffu) : 0xffffffffu - tie;
                const unsigned long long key = ((unsigned long long)__float_as_uint(root) << 32) | (unsigned long long)low;
                __hip_atomic_fetch_min(&ws.keys[owner], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else {
            const unsigned long long tb = (unsigned long long)__double_as_longlong(root);
            unsigned long long old = 0ull;
            if (hit) old = __hip_atomic_fetch_min(&ws.keys[owner], tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
            const unsigned long long cur = ws.keys[owner];
            if (hit && tb == cur && old > tb) ws.kidx[owner] = 0u;              // the candidate that lowered the minimum to its final value of this step
            __builtin_amdgcn_wave_barrier();
            if (hit && tb == cur) __hip_atomic_fetch_max(&ws.kidx[owner], tie + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_wave_barrier();
        }
    }
}
template <typename T, bool WITH_R, typename SRC, typename ORIG = NoOrig>
__device__ __forceinline__ void resolve_pairs_impl(int num, SRC src, [[maybe_unused]] const typename Vec4<T>::type *rad, V3<T> o, V3<T> d, T tmin, const WaveScratch &ws, unsigned n, unsigned lane, ORIG orig = ORIG(), unsigned *prof = nullptr) {
    __builtin_amdgcn_wave_barrier();
    const uint2 *list = reinterpret_cast<const uint2 *>(ws.pairs);
    unsigned *singles = ws.pairs + 2u * ws.cap;            // ws.cap2 words behind the ws.cap entries
    unsigned total = 0;                                    // (wave-uniform) single candidates waiting for their exact test
    for (unsigned p0 = 0; p0 < n; p0 += 64u) {
        const unsigned p = p0 + lane;
        const uint2 e = list[p < n ? p : 0u];
        unsigned m = p < n ? e.x : 0u;
        const unsigned code31 = e.y + 31u;
        for (;;) {
            const unsigned long long act = __ballot(m != 0u);
            if (prof) prof[2] += 1u;                                   // (phase-profile build: iterations of this loop)
            if (!act) break;
            if (total + 64u > ws.cap2) {
                __builtin_amdgcn_wave_barrier();
                test_singles<T, WITH_R>(num, src, rad, o, d, tmin, ws, singles, total, lane, orig, prof);
                __builtin_amdgcn_wave_barrier();
                total = 0;
            }
            if (m != 0u) {
                const unsigned z = (unsigned)__builtin_ctz(m);
                m &= m - 1u;
                singles[__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, total))] = code31 - z;     // b = 31 - z
            }
            total += (unsigned)__popcll(act);
        }
    }
    __builtin_amdgcn_wave_barrier();
    test_singles<T, WITH_R>(num, src, rad, o, d, tmin, ws, singles, total, lane, orig, prof);
    __builtin_amdgcn_wave_barrier();
}

// (NUM_REFERENCE_FMA2 reads the radius from mat0: its own copy of the loop, so that the other modes keep their registers)
template <typename T, typename SRC, typename RAD, typename ORIG = NoOrig>
__device__ __forceinline__ void resolve_pairs(int num, SRC src, RAD rad, V3<T> o, V3<T> d, T tmin, const WaveScratch &ws, unsigned n, unsigned lane, ORIG orig = ORIG(), unsigned *prof = nullptr) {
    if (num == NUM_REFERENCE_FMA2) resolve_pairs_impl<T, true>(num, src, rad(), o, d, tmin, ws, n, lane, orig, prof);
    else resolve_pairs_impl<T, false>(num, src, nullptr, o, d, tmin, ws, n, lane, orig, prof);
}

// Closest hit for the rays of a whole wave (every lane calls it, convergently; has_ray = this lane has a ray).
// With `mc` (group cull, RTW_FLAG_GROUP_CULL): the spheres come in the cull layout's device order (src, orig), and a block of
// 32 is visited only when some ray of the half wave can touch its box (the conservative margin of hit_world_cull, in binary32 with
// the Float32 kappa for both precisions): the table vote of CullGrid, once per scan.  Returns the DEVICE index.
template <typename T, typename SRC, typename ORIG = NoOrig, typename CLK = NoClock>
__device__ __forceinline__ int hit_world_mfma(const DevScene<T> &w, SRC src, V3<T> o, V3<T> d, bool has_ray, T tmin, T &t_hit,
                                              const WaveScratch &ws, unsigned lane, CLK &&clk = NoClock(),
                                              const MfmaCull *mc = nullptr, ORIG orig = ORIG()) {
    constexpr bool CULLED = !__is_same(ORIG, NoOrig);
    // radii (mat0[i].x) in the order of `src`: read by NUM_REFERENCE_FMA2 only -- fetched from the kernel arguments where that mode needs them, not held across the scan
    auto rad = [&]() -> const typename Vec4<T>::type * { if constexpr (CULLED) return (const typename Vec4<T>::type *)mc->mat0; else return w.mat0; };
    // ---- ray features (binary32) ----
    const float ox = (float)o.x, oy = (float)o.y, oz = (float)o.z, dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    const float s2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    const float oinf = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ox), __builtin_fabsf(oy)), __builtin_fabsf(oz));
    const bool ok = has_ray && s2 <= 1.0009f && oinf <= w.mf_o_max;                  // (false for NaN)
    const float q = __builtin_fmaf(oz, dz, __builtin_fmaf(oy, dy, ox * dx));          // d.o
    const float oo = __builtin_fmaf(oz, oz, __builtin_fmaf(oy, oy, ox * ox));
    const float o1 = (__builtin_fabsf(ox) + __builtin_fabsf(oy)) + __builtin_fabsf(oz);
    const float oop = __builtin_fmaf(oo, w.mf_oo_keep, -(w.mf_o1_coef * o1));         // oo' = |o|^2 - the ray's share of the margin
    const float tq = w.mf_sigma2 * __builtin_fmaf(q, q, -oop);                        // (q^2 - oo') s^2
    const float z = ok ? 1.0f : 0.0f, z2 = z + z, zs2 = z2 * w.mf_sc;
    // p = o - q d, as 2 p_k s; the quadratic features 2 m d_i d_j
    const float fp[3] = {__builtin_fmaf(-q, dx, ox) * zs2, __builtin_fmaf(-q, dy, oy) * zs2, __builtin_fmaf(-q, dz, oz) * zs2};
    const float dx2 = dx * z2, dy2 = dy * z2, dz2 = dz * z2;
    const float fq[6] = {dx2 * dx, dy2 * dy, dz2 * dz, (dx2 + dx2) * dy, (dx2 + dx2) * dz, (dy2 + dy2) * dz};
    // a lane that is not ok: all features 0 and t1 = +-60000 (exact in f16): W = +-2^15 x 60000 for EVERY sphere
    const float tx = ok ? tq : (has_ray ? 60000.0f * 32768.0f : -60000.0f * 32768.0f);
    const uint4 *ops_base = CULLED ? mc->ops : w.mf_ops;         // (wave-uniform: the block's operands are read as base[block] + lane, no 64-bit vector address arithmetic per block)
    auto pa_of = [&](int b) { return ops_base + (size_t)(unsigned)b * 128u; };
    const int n_blocks = CULLED ? mc->blocks : w.mf_blocks;
    // Group cull: the block vote.  A sphere of a block can only be hit if the RAY (t >= 0) meets the block's box grown by the margin m
    // (hit_world_cull derives m).  Round 4 ran that slab test per (lane, block): 23 VALU instructions x 17 blocks, about what the skipped
    // blocks saved.  Since round 5 the ray is clipped ONCE per scan against the box of the whole small class grown by m (the union of the
    // blocks' boxes: every grown block box lies inside it), which leaves a segment [tn, tf] of the ray; every point of the ray inside any
    // grown block box lies on that segment, hence inside the segment's axis-aligned bounds [pmin, pmax], and
    //     the ray can touch block b  =>  lo_b - m <= pmax + delta  and  hi_b + m >= pmin - delta        on every axis
    // with delta the rounding of tn, tf and the two end points (a few ulps of |o| + tf |d|: below 1e-6 of the distances m is proportional
    // to with a factor >= 2^-8), covered by using m for it: lo3 = pmin - 2m, hi3 = pmax + 2m.  That test is not run per block: the bins
    // of lo3 / hi3 index the tables of CullGrid, whose entries are the SETS of blocks passing each of the six comparisons (a superset: the
    // bin's far edge stands for the coordinate); their AND is the ray's set, the OR over a half wave is the vote -- 6 look-ups per ray and
    // scan instead of 10 instructions per (ray, block).  For the flat layer of small spheres of the reference's scenes the segment is
    // short (the ray crosses the layer), so the bounds are tight; a ray running along the layer gets loose bounds -- conservative, never
    // wrong.  A ray that misses the small class's box is in no block's set; one that does not use the filter (not ok) takes every live
    // block and every ray the BIG class, through the flag words behind the tables; lanes without a ray contribute nothing.
    // (a lambda run once per group of 32 blocks, from the ray itself: nothing of it -- bins, flags -- is held in registers across the block
    //  loop; scenes of more than 1 024 spheres repeat the clip per group, 100 instructions against 32 blocks' work)
    [[maybe_unused]] auto block_sets = [&](int base_, unsigned &v0_, unsigned &v1_) {
        float cox = ox, coy = oy, coz = oz, cdx = dx, cdy = dy, cdz = dz, cs2 = s2;
        __asm__ volatile("" : "+v"(cox), "+v"(coy), "+v"(coz), "+v"(cdx), "+v"(cdy), "+v"(cdz), "+v"(cs2));      // (not hoisted out of the group loop)
        unsigned bin_lo[3], bin_hi[3];
        const float ex = cox - mc->cs[0], ey = coy - mc->cs[1], ez = coz - mc->cs[2];
        const float eps_p = (cs2 > 1.0f ? cs2 - 1.0f : 0.0f) + 2.4e-7f * cs2;
        // (hardware approximations v_sqrt_f32 / v_rcp_f32, 1 ulp: m is inflated by 2^-10 for them, and the reciprocals only place the end
        //  points of the clip, whose rounding the 2 m of slack covers a thousandfold -- the IEEE forms cost 65 instructions per scan)
        const float m = 1.001f * (0.00390625f * (cs2 > 1.0f ? cs2 : 1.0f) + 2.0f * __builtin_amdgcn_sqrtf(eps_p)) *
                        ((__builtin_amdgcn_sqrtf(__builtin_fmaf(ez, ez, __builtin_fmaf(ey, ey, ex * ex))) + mc->rs) + 1.0f);
        // 1 / d_k with |d_k| clamped to >= 1e-9 (moves the ray by < 1e-9 t): max on the magnitude, the sign copied back (v_max_f32 |x|, v_bfi_b32, v_rcp_f32)
        auto safe_inv = [](float x) {
            const float mag = __builtin_fmaxf(__builtin_fabsf(x), 1e-9f);
            return __builtin_amdgcn_rcpf(__uint_as_float((__float_as_uint(mag) & 0x7fffffffu) | (__float_as_uint(x) & 0x80000000u)));
        };
        const float ix = safe_inv(cdx), iy = safe_inv(cdy), iz = safe_inv(cdz);
        const float x0 = ((mc->glo[0] - m) - cox) * ix, x1 = ((mc->ghi[0] + m) - cox) * ix;
        const float y0 = ((mc->glo[1] - m) - coy) * iy, y1 = ((mc->ghi[1] + m) - coy) * iy;
        const float z0 = ((mc->glo[2] - m) - coz) * iz, z1 = ((mc->ghi[2] + m) - coz) * iz;
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(x0, x1), __builtin_fminf(y0, y1)), __builtin_fminf(z0, z1)), 0.0f);
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(x0, x1), __builtin_fmaxf(y0, y1)), __builtin_fmaxf(z0, z1));
        const float m2 = m + m;
        const float ax = __builtin_fmaf(tn, cdx, cox), ay = __builtin_fmaf(tn, cdy, coy), az = __builtin_fmaf(tn, cdz, coz);
        const float bx = __builtin_fmaf(tf, cdx, cox), by = __builtin_fmaf(tf, cdy, coy), bz = __builtin_fmaf(tf, cdz, coz);
        // (for a ray that uses the filter every quantity above is finite: |o_k| <= mf_o_max, |d|^2 <= 1.0009, |1 / d_k| <= 1e9; the others are
        //  handled by the mask below, whatever their bounds came out as)
        const bool hits_class = tf >= tn;
        const float lo3[3] = {__builtin_fminf(ax, bx) - m2, __builtin_fminf(ay, by) - m2, __builtin_fminf(az, bz) - m2};
        const float hi3[3] = {__builtin_fmaxf(ax, bx) + m2, __builtin_fmaxf(ay, by) + m2, __builtin_fmaxf(az, bz) + m2};
        // the bins of the bounds on each axis (inv >= 0); a ray that misses the small class's box is in no block's set (the BIG class comes
        // in through its flag word), one that does not use the filter takes every live block
        const CullGrid &G = mc->grid;
        auto binf = [](float u) {                                                                                          // (NaN -> 0)
            const unsigned b = (unsigned)__builtin_amdgcn_fmed3f(u, 0.0f, (float)RTW_CULL_BINS - 0.5f);
            __builtin_assume(b < (unsigned)RTW_CULL_BINS);      // (a 32-bit table offset: no 64-bit index pairs held across the block loop)
            return b;
        };
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bin_lo[k] = binf(__builtin_fmaf(lo3[k], G.inv[k], G.off[k]));
            bin_hi[k] = binf(__builtin_fmaf(hi3[k], G.inv[k], G.off[k]));
        }
        const bool cells_ok = ok && hits_class;
        const unsigned flag_word = 6u * RTW_CULL_BINS + (has_ray ? (ok ? 0u : 1u) : 2u);     // the word behind the group's tables the ray ORs in: BIG / live / 0
        const unsigned *t = mc->tab + (base_ >> 5) * RTW_CULL_TAB_WORDS;
        unsigned mine = (t[0 * RTW_CULL_BINS + bin_hi[0]] & t[1 * RTW_CULL_BINS + bin_lo[0]]) & (t[2 * RTW_CULL_BINS + bin_hi[1]] & t[3 * RTW_CULL_BINS + bin_lo[1]]) &
                        (t[4 * RTW_CULL_BINS + bin_hi[2]] & t[5 * RTW_CULL_BINS + bin_lo[2]]);
        mine = (cells_ok ? mine : 0u) | t[flag_word];
        // OR over the 16 lanes of a row (xor butterfly on the DPP network), then the two rows of each half wave
        mine |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xf, 0xf, true);     // quad_perm [1,0,3,2]
        mine |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x4E, 0xf, 0xf, true);     // quad_perm [2,3,0,1]
        mine |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x141, 0xf, 0xf, true);    // row_half_mirror
        mine |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)mine, 0x140, 0xf, 0xf, true);    // row_mirror
        v0_ = (unsigned)__builtin_amdgcn_readlane((int)mine, 0) | (unsigned)__builtin_amdgcn_readlane((int)mine, 16);
        v1_ = (unsigned)__builtin_amdgcn_readlane((int)mine, 32) | (unsigned)__builtin_amdgcn_readlane((int)mine, 48);
    };
    // (measured: running the first group's vote and the first operand fetch HERE, in front of the ray operands and the huge-sphere tests,
    //  changes nothing -- 293.1 against 293.2 ms -- and costs a spilled register: the vote stays in the block loop's prologue)
    uint4 A1 = {0u, 0u, 0u, 0u}, A2 = {0u, 0u, 0u, 0u};
    if constexpr (!CULLED) { A1 = pa_of(0)[lane]; A2 = pa_of(0)[lane + 64u]; }
    // Lane (H, j) supplies slots 8H .. 8H + 7 of both MFMAs for ray j (first half wave: h = 0) / ray 32 + j (h = 1).  Every
    // lane makes, for ITS ray, the operand words of both lane groups; one v_permlane32_swap per word then hands each lane
    // group its words for both half waves:
    //     swap(X, Y):  X' = [X(0..31) | Y(0..31)],  Y' = [X(32..63) | Y(32..63)]
    // with X = the group-0 word and Y = the group-1 word of the lane's own ray, X' is the operand of the first half wave
    // (lane l < 32: its own ray's group-0 word; lane l >= 32: ray l - 32's group-1 word) and Y' that of the second.
    // With sw = (piece 1, piece 2) of a feature: (1, 1) = dup(sw), (2 of a, 1 of b) = alignbit(sw_b, sw_a, 16), (1, 2) = sw.
    unsigned sq[6], sp[3];
#pragma unroll
    for (int k = 0; k < 6; ++k) sq[k] = split_f16(fq[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) sp[k] = split_f16(fp[k]);
    const _Float16 t1 = (_Float16)(tx * (1.0f / 32768.0f));
    const float trem = tx - 32768.0f * (float)t1;                         // exact
    const unsigned x23 = split_f16(trem * (1.0f / 16.0f));               // (t2, t3)
    const unsigned sb = ok ? 0x7800u : 0u, ss = ok ? 0x4c00u : 0u;        // the ray's constants 2^15, 2^4 as f16 (0: not ok)
    auto dup = [](unsigned v) { return __builtin_amdgcn_perm(v, v, 0x01000100u); };                 // (piece 1, piece 1)
    auto cat = [](unsigned a, unsigned b) { return __builtin_amdgcn_alignbit(b, a, 16); };        // (piece 2 of a, piece 1 of b)
    // words 0-3: MFMA 1 (slots 0-7 | 8-15), words 4-7: MFMA 2 (slots 16-23 | 24-31)
    const unsigned g0[8] = {dup(sq[0]), cat(sq[0], sq[1]), sq[1], dup(sq[2]),                       // xx xx | xx yy | yy yy | zz zz
                            sq[5], dup(sp[0]), cat(sp[0], sp[1]), sp[1]};                            // yz yz | px px | px py | py py
    const unsigned g1[8] = {cat(sq[2], sq[3]), sq[3], dup(sq[4]), cat(sq[4], sq[5]),                // zz xy | xy xy | xz xz | xz yz
                            dup(sp[2]), cat(sp[2], sb), ss | ((unsigned)__builtin_bit_cast(unsigned short, t1) << 16), x23};   // pz pz | pz k | k T | T T
    unsigned h0[8], h1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const auto sw = __builtin_amdgcn_permlane32_swap(g0[k], g1[k], false, false);
        h0[k] = sw[0]; h1[k] = sw[1];
    }
    RTW_PROBE_OPERANDS_TWICE();
    rtw_h8 B1[2], B2[2];
    {
        const uint4 q10 = {h0[0], h0[1], h0[2], h0[3]}, q11 = {h1[0], h1[1], h1[2], h1[3]};
        const uint4 q20 = {h0[4], h0[5], h0[6], h0[7]}, q21 = {h1[4], h1[5], h1[6], h1[7]};
        B1[0] = __builtin_bit_cast(rtw_h8, q10); B1[1] = __builtin_bit_cast(rtw_h8, q11);
        B2[0] = __builtin_bit_cast(rtw_h8, q20); B2[1] = __builtin_bit_cast(rtw_h8, q21);
    }
    // ---- the result cells, initialised with the lane's own exact test of the scene's huge spheres (DevScene::huge / MfmaCull::huge):
    //      the same contract test and the same key / tie rule as pass 2, so the minimum over all candidates is unchanged ----
    const unsigned lane_const = lane << 16;
    unsigned total = 0;                                   // wave-uniform: entries in the wave's candidate list
    {
        unsigned long long key0 = ~0ull;
        [[maybe_unused]] unsigned kidx0 = 0u;
        using V4 = typename Vec4<T>::type;
        const int n_huge = CULLED ? (RTW_INLANE_FILTER ? mc->n_exact : mc->n_huge) : w.n_huge;
        for (int hgi = 0; hgi < n_huge; ++hgi) {
            int si;
            if constexpr (CULLED) si = __builtin_amdgcn_readfirstlane((int)mc->tab[6 * RTW_CULL_BINS + 4 + hgi]);      // (the list lies with the vote's tables)
            else si = hgi == 0 ? w.huge[0] : w.huge[1];                  // (no dynamic indexing of a by-value struct: that would live in scratch)
            const V4 sg = src[si];
            T hb_, disc_, root_ = 0;
            T rr_ = T(0);
            if (w.numerics == NUM_REFERENCE_FMA2) rr_ = rad()[si].x;
            sphere_disc<T>(w.numerics, sg.x, sg.y, sg.z, sg.w, rr_, o, d, hb_, disc_);
            if (has_ray && sphere_root<T>(hb_, disc_, tmin, (T)__builtin_huge_val(), root_)) {
                unsigned tie = (unsigned)si;                       // larger = later in the caller's list (resolve_pairs)
                if constexpr (CULLED) tie = ((unsigned)orig[si] << 16) | (unsigned)si;
                if constexpr (sizeof(T) == 4) {
                    const unsigned low = CULLED ? ((0xffffu - (tie >> 16)) << 16) | (tie & 0xffffu) : 0xffffffffu - tie;
                    const unsigned long long k = ((unsigned long long)__float_as_uint((float)root_) << 32) | (unsigned long long)low;
                    key0 = k < key0 ? k : key0;
                } else {
                    const unsigned long long tb = (unsigned long long)__double_as_longlong((double)root_);
                    if (tb < key0 || (tb == key0 && tie + 1u > kidx0)) { key0 = tb; kidx0 = tie + 1u; }
                }
            }
        }
        if constexpr (sizeof(T) == 4) ws.keys[lane] = key0;
        else { ws.keys[lane] = key0; ws.kidx[lane] = kidx0; }
    }
    // The rest of the in-lane class (group cull: the BIG class's other members -- the unit spheres of the reference's scene, each hit by a
    // tenth of the rays): every lane evaluates the DISCRIMINANT of the render's numerics for its own ray; where it is >= 0 the lane records
    // an ordinary list entry for (its ray, that sphere) and pass 2 does the root, the key and the tie rule.  (Until round 6 these spheres
    // got the whole exact test in every lane: ~48 VALU instructions each per scan against ~26 now.)
    if constexpr (CULLED && RTW_INLANE_FILTER) {
        using V4 = typename Vec4<T>::type;
        for (int hgi = mc->n_exact; hgi < mc->n_huge; ++hgi) {
            const int si = __builtin_amdgcn_readfirstlane((int)mc->tab[6 * RTW_CULL_BINS + 4 + hgi]);
            const V4 sg = src[si];
            T hb_, disc_;
            T rr_ = T(0);
            if (w.numerics == NUM_REFERENCE_FMA2) rr_ = rad()[si].x;
            sphere_disc<T>(w.numerics, sg.x, sg.y, sg.z, sg.w, rr_, o, d, hb_, disc_);
            const bool cand = has_ray && !(disc_ < T(0));
            const unsigned long long cm = __ballot(cand);
            if (cm) {
                if (total + 64u > ws.cap) { resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig); total = 0; }
                if (cand) {
                    // the entry a recording lane (H, j) would write for ray j + 32 half and sphere 32 block + 16 H + register: H, block, register from
                    // the sphere, j and the half from this lane
                    const unsigned b = ((lane >> 5) << 4) | ((unsigned)si & 15u);
                    const unsigned code = ((((unsigned)si >> 4) & 1u) << 21) | ((lane & 31u) << 16) | (((unsigned)si >> 5) << 5);
                    const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, total));
                    reinterpret_cast<uint2 *>(ws.pairs)[pos] = uint2{0x80000000u >> b, code};
                }
                total += (unsigned)__popcll(cm);
            }
        }
    }
    // Wave priority: low inside the block loop, raised for everything else (pass 2 and the divergent phases of the lane loop are
    // chains of dependent LDS / VALU instructions; a wave in the block loop issues a 32-cycle MFMA pair and waits for it
    // anyway).  Measured at Float32: 372.1 -> 368.0 ms on one box, 361.8 -> 359.9 on a faster one, group cull 345.6 -> 343.0; which of
    // the levels 1 - 3 made no difference.  Float64 (4 waves per SIMD, FP64 instructions of two issue slots in pass 2 and the
    // shading) is the other way round: 1152.7 -> 1156.8 ms with these levels, 1144.7 -> 1134.6 with the block loop HIGH and the
    // rest low -- so that is what it gets.
    constexpr bool use_prio = RTW_SCAN_PRIO != 0;
    if (use_prio) __builtin_amdgcn_s_setprio(sizeof(T) == 4 ? 0 : 1);
    // The blocks are visited in groups of 32 (CULLED: every lane looks its ray's set of blocks up, the sets of a half wave are ORed, and
    // only the blocks some ray can touch are visited; otherwise one group = every block in turn).
    for (int base = 0; base < n_blocks; base += CULLED ? 32 : n_blocks) {
    [[maybe_unused]] unsigned vote0 = 0, vote1 = 0, todo = 0;
    int blk = base;
    if constexpr (CULLED) {
        block_sets(base, vote0, vote1);
        todo = vote0 | vote1;
        clk.count(7, (unsigned)(n_blocks - base < 32 ? n_blocks - base : 32));
        clk.count(6, (unsigned)((n_blocks - base < 32 ? n_blocks - base : 32) - __popc(todo)));
        if (!todo) continue;
        blk = base + (int)__builtin_ctz(todo);
        todo &= todo - 1u;
        A1 = pa_of(blk)[lane]; A2 = pa_of(blk)[lane + 64u];
    }
    for (bool more = true; more;) {
        const int cur = blk;                                   // (this iteration's block; `blk` becomes the next one)
        [[maybe_unused]] bool do_half0 = true, do_half1 = true;          // (wave-uniform) group cull: which ray halves of the wave can touch this block
        if constexpr (CULLED) {
            // lanes l and l + 32 hold the same 32 rays of a half wave: the two MFMA pairs of a block are the two RAY halves -- each is skipped by itself
            do_half0 = ((vote0 >> (cur - base)) & 1u) != 0u;
            do_half1 = ((vote1 >> (cur - base)) & 1u) != 0u;
            more = todo != 0u;
            blk = more ? base + (int)__builtin_ctz(todo) : cur + 1;     // (the operand array has one block of padding at the end)
            todo &= todo - 1u;
        } else {
            blk = cur + 1;
            more = blk < n_blocks;
        }
        unsigned mask = 0;
        bool any_cand = false;                               // (wave-uniform)
        const rtw_f16v zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        // Sign collection of one half block's 16 filter values: one slow-class v_alignbit_b32 per value.  Most (wave, half block)
        // evaluations find no candidate in ANY lane (rays of a wave are neighbours), so the sign bits of a GROUP of values are ANDed first
        // (FMA-class v_bitop3_b32 / v_and_b32, one per two values) and the alignbits of the group run only when some lane has a
        // non-negative value in it.  RTW_PRECHECK_GROUPS groups per half block: 1 = all 16 values at once (rounds 3 - 5), 2 = two groups of 8.
        // (Left to the compiler the skip is if-converted -- both sides executed -- and gains nothing: the collecting side is fenced by an asm.)
        auto collect = [&](rtw_f16v &Wv) {
            constexpr int NG = RTW_PRECHECK_GROUPS, GS = 16 / NG;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int r0 = g * GS;
                unsigned t = __builtin_amdgcn_bitop3_b32(__float_as_uint(Wv[r0]), __float_as_uint(Wv[r0 + 1]), __float_as_uint(Wv[r0 + 2]), 0x80);
#pragma unroll
                for (int r = r0 + 3; r < r0 + GS - 1; r += 2) t = __builtin_amdgcn_bitop3_b32(t, __float_as_uint(Wv[r]), __float_as_uint(Wv[r + 1]), 0x80);
                t &= __float_as_uint(Wv[r0 + GS - 1]);
                if (RTW_SCAN_SKIP && !__any((int)t >= 0)) {
                    mask = (mask << GS) | ((1u << GS) - 1u);                       // all negative: no lane has a candidate in this group
                } else {
                    if (RTW_SCAN_SKIP) __asm__ volatile("" : "+v"(Wv));
#pragma unroll
                    for (int r = r0; r < r0 + GS; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(Wv[r]), 31);
                    any_cand = true;
                    clk.count(26, 1u);                                                 // sign collections executed (per group)
                }
            }
        };
        [[maybe_unused]] auto eval = [&](const rtw_f16v &Wv) {            // (rtw_probes.hpp)
#pragma unroll
            for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(Wv[r]), 31);
        };
        // the filter values of one half block (32 spheres x 32 rays): two chained MFMAs (rtw_probes.hpp: time probes that repeat / replace them)
        auto filter_pair = [&](const uint4 &a1, const uint4 &a2, const rtw_h8 &b1, const rtw_h8 &b2) -> rtw_f16v {
            RTW_PROBE_FILTER_PAIR_REPLACED(a1, a2, b1, b2);
            rtw_f16v Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a1), b1, zero, 0, 0, 0);
            Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a2), b2, Wp, 0, 0, 0);
            RTW_PROBE_FILTER_PAIR_AGAIN(Wp, a1, a2, b1, b2);
            return Wp;
        };
        constexpr unsigned HB = 16u;                          // mask bits per half block
        {
            if (!CULLED || do_half0) {
                rtw_f16v Wv = filter_pair(A1, A2, B1[0], B2[0]);
                RTW_PROBE_EVAL_TWICE(Wv);
                collect(Wv);
            } else {
                mask = (1u << HB) - 1u;                               // (no ray of this half can touch the block)
            }
        }
        {
            // the next block's operands are fetched as soon as this block's last use of each is issued (one block of
            // padding at the end), so only one set of A registers is live during the evaluation
            // (two copies of the prefetch: with ONE behind an `if (run) Wv = ...` the skipped side zeroes all 16 result registers -- 16 v_mov per
            //  skipped ray half in the group cull, measured in the ISA)
            if (!CULLED || do_half1) {
                rtw_f16v Wv = filter_pair(A1, A2, B1[1], B2[1]);
                __builtin_amdgcn_sched_barrier(0);
                A1 = pa_of(blk)[lane]; A2 = pa_of(blk)[lane + 64u];
                __builtin_amdgcn_sched_barrier(0);
                RTW_PROBE_EVAL_TWICE(Wv);
                collect(Wv);
            } else {
                A1 = pa_of(blk)[lane]; A2 = pa_of(blk)[lane + 64u];
                mask = (mask << HB) | ((1u << HB) - 1u);
            }
        }
        clk.lap(2);
        if (RTW_SCAN_SKIP && !any_cand) {                    // no lane has a candidate in this block: nothing to extract
            if constexpr (!CULLED) { clk.count(7, 1u); clk.count(6, 1u); }
            continue;
        }
        // (the block's mask: bit 31 - b, b = half wave << 4 | result register)
        unsigned m = ~mask;
        if constexpr (!CULLED) {                          // (phase-profile build only: blocks, and blocks without any candidate)
            clk.count(7, 1u);
            if (!__any(m != 0u)) clk.count(6, 1u);
        }
        // the lanes with a candidate in this block record their bits, ONE entry each: (bits, recording lane << 16 | block << 5) -- pass 2
        // walks the bits.  (Until round 5 every BIT became an entry here: a loop of ballot / ctz / mbcnt / write per candidate of the
        // busiest lane, ~25 - 35 VALU instructions per block with a candidate against 7 now.)
        const unsigned long long act = __ballot(m != 0u);
        clk.count(27, 1u);                                                             // blocks that record entries
        if (total + 64u > ws.cap) {
            clk.lap(4);
            { unsigned pr[3] = {0u, 0u, 0u}; resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig, clk.on() ? pr : nullptr); clk.count(13, total); clk.count(14, pr[0]); clk.count(15, pr[1]); clk.count(28, pr[2]); }
            total = 0;
            clk.lap(5);
        }
        RTW_PROBE_EXTRACT_TWICE();
        if (m != 0u) {
            const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, total));
            reinterpret_cast<uint2 *>(ws.pairs)[pos] = uint2{m, lane_const + (unsigned)cur * 32u};
        }
        total += (unsigned)__popcll(act);
        clk.lap(4);
    }
    }
    if (use_prio) __builtin_amdgcn_s_setprio(sizeof(T) == 4 ? 1 : 0);
    { unsigned pr[3] = {0u, 0u, 0u}; resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig, clk.on() ? pr : nullptr); clk.count(13, total); clk.count(14, pr[0]); clk.count(15, pr[1]); clk.count(28, pr[2]); }
    RTW_PROBE_RESOLVE_TWICE();
    clk.lap(5);
    int idx;
    if constexpr (sizeof(T) == 4) {
        const unsigned long long k = ws.keys[lane];
        idx = k == ~0ull ? -1 : (CULLED ? (int)((unsigned)k & 0xffffu) : (int)(0xffffffffu - (unsigned)k));
        t_hit = __uint_as_float((unsigned)(k >> 32));
    } else {
        const unsigned long long k = ws.keys[lane];
        const unsigned ki = ws.kidx[lane];
        idx = ki == 0u ? -1 : (CULLED ? (int)((ki - 1u) & 0xffffu) : (int)ki - 1);
        t_hit = __longlong_as_double((long long)k);
    }
    if (!has_ray) idx = -1;
    return idx;
}

}  // namespace rtw
