// rtw_path.hpp -- the path tracer's building blocks: SVector arithmetic, Xoroshiro128+ samplers, hit(::Sphere), light transport, materials, sky, camera, the exact pixel sum
// (part of the device side of the hot path, gfx950 only; see rtw_device.hpp for the numerics contract all parts share)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtw {


enum { LAMBERTIAN = 0, METAL = 1, DIELECTRIC = 2 };

template <typename T> struct V3 { T x, y, z; };
struct C3 { double r, g, b; };

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

// Correctly rounded binary32 square root and reciprocal in FEWER instructions than the compiler's IEEE expansions (14 / 10 VALU instructions:
// denormal scaling, v_sqrt / v_rcp, +-1 ulp checks with v_cndmask, fix-ups) -- 5.6 square roots and 2 reciprocals per lane-loop iteration.
// Markstein's sequences on the hardware approximations (1 ulp): every step an FMA, the last one rounds once.  Valid where nothing under- or
// overflows on the way: the arguments of a wave are range-checked, and a wave with ANY argument outside takes the compiler's sequence (a
// tangent ray's discriminant 0, denormals, inf, NaN).  Same bits as __builtin_sqrtf / 1.0f / x for EVERY binary32 argument: unit op 16 compares
// all 2^32 of them on the device (tests/test_gpu_round6.py).  RTW_EXACT_FAST_MATH=0: the compiler's sequences.
#ifndef RTW_EXACT_FAST_MATH
#define RTW_EXACT_FAST_MATH 1
#endif
__device__ __forceinline__ float t_sqrt(float x) {
#if RTW_EXACT_FAST_MATH
    if (__all(x >= 0x1p-100f && x <= 0x1p100f)) {
        const float y = __builtin_amdgcn_rsqf(x);
        float g = x * y, h = 0.5f * y;
        const float r = __builtin_fmaf(-h, g, 0.5f);
        g = __builtin_fmaf(g, r, g); h = __builtin_fmaf(h, r, h);
        const float d = __builtin_fmaf(-g, g, x);
        return __builtin_fmaf(d, h, g);
    }
#endif
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ double t_sqrt(double x) { return __builtin_sqrt(x); }
// 1 / x, one rounding (StaticArrays' inv(norm(v)))
__device__ __forceinline__ float t_rcp(float x) {
#if RTW_EXACT_FAST_MATH
    if (__all(__builtin_fabsf(x) >= 0x1p-100f && __builtin_fabsf(x) <= 0x1p100f)) {
        float y = __builtin_amdgcn_rcpf(x);
        float e = __builtin_fmaf(-x, y, 1.0f);
        y = __builtin_fmaf(e, y, y);
        e = __builtin_fmaf(-x, y, 1.0f);
        return __builtin_fmaf(e, y, y);
    }
#endif
    return 1.0f / x;
}
__device__ __forceinline__ double t_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ float t_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double t_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- SVector arithmetic (src/vec.jl:3): element-wise, one rounding each ---------------------
template <typename T> __device__ __forceinline__ V3<T> vadd(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> __device__ __forceinline__ V3<T> vsub(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> __device__ __forceinline__ V3<T> vscale(T s, V3<T> a) { return {s * a.x, s * a.y, s * a.z}; }
template <typename T> __device__ __forceinline__ V3<T> vneg(V3<T> a) { return {-a.x, -a.y, -a.z}; }
// StaticArrays dot, length 3: (a1*b1 + a2*b2) + a3*b3
template <typename T> __device__ __forceinline__ T dot(V3<T> a, V3<T> b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
// StaticArrays normalize(v) = inv(norm(v)) * v
#include "rtw_probes.hpp"
#ifdef RTW_PROBE_FASTDIV   // time probe (WRONG image): approximate reciprocal square root / reciprocal instead of the IEEE sqrt and divisions
__device__ __forceinline__ float probe_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double probe_rsq(double x) { return 1.0 / __builtin_sqrt(x); }
__device__ __forceinline__ float probe_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double probe_rcp(double x) { return 1.0 / x; }
#endif
template <typename T> __device__ __forceinline__ V3<T> normalize(V3<T> a) {
    T inv = RTW_RSQRT(dot(a, a));
    return vscale(inv, a);
}
// src/vec.jl:19-20: compared against the Float64 literal 1e-5
template <typename T> __device__ __forceinline__ bool near_zero(V3<T> a) { return (double)dot(a, a) < 1e-5; }

// ---- RNG: per-lane Xoroshiro128+ (src/init.jl:2-12, src/rand.jl:5-13; RandomNumbers.jl) -----
struct Rng { uint64_t x, y; };

// Xoroshiro128+ (55/14/36), one step:  out = x + y;  s1 = x ^ y;  x' = rotl(x, 55) ^ s1 ^ (s1 << 14);  y' = rotl(s1, 36).
// Written on 32-bit halves: the rotations and the long shift are funnel shifts (v_alignbit_b32) and each half of x' is
// one three-input xor (v_bitop3_b32) -- 10 VALU instructions + the output add instead of the 14 the compiler makes of the
// 64-bit form.  (hi:lo) >> s, low word:
__device__ __forceinline__ uint32_t funnel(uint32_t hi, uint32_t lo, uint32_t s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
__device__ __forceinline__ uint64_t rng_next(Rng &r) {
    const uint32_t xl = (uint32_t)r.x, xh = (uint32_t)(r.x >> 32), yl = (uint32_t)r.y, yh = (uint32_t)(r.y >> 32);
    const uint64_t out = r.x + r.y;                               // (callers that keep 23 bits get a 32-bit add)
    const uint32_t sl = xl ^ yl, sh = xh ^ yh;
    const uint32_t nxl = __builtin_amdgcn_bitop3_b32(funnel(xh, xl, 9), sl, sl << 14, 0x96);              // rotl 55 = rotr 9
    const uint32_t nxh = __builtin_amdgcn_bitop3_b32(funnel(xl, xh, 9), sh, funnel(sh, sl, 18), 0x96);
    r.x = ((uint64_t)nxh << 32) | nxl;
    r.y = ((uint64_t)funnel(sl, sh, 28) << 32) | funnel(sh, sl, 28);                                    // rotl 36 = swap halves, rotl 4
    return out;
}
__device__ __forceinline__ uint64_t splitmix64(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
// the independent stream of (render seed, pixel, sample chunk) -- DESIGN.md section 5
__device__ __forceinline__ void rng_stream(uint64_t seed, uint64_t pixel, uint64_t chunk, Rng &r) {
    // (the +1 is applied to registers made HERE: written as K * (x + 1), the compiler keeps the two 64-bit addends K in
    // loop-long register pairs -- spilled ones, in the trace kernel)
    uint64_t p1 = pixel + 1, c1 = chunk + 1;
    __asm__ volatile("" : "+v"(p1), "+v"(c1));
    uint64_t s = seed ^ (0xd1b54a32d192ed03ULL * p1) ^ (0x8cb92ba72f3d8dd7ULL * c1);
    r.x = splitmix64(s);
    r.y = splitmix64(s);
    (void)rng_next(r);
}
// rand(rng, Float32): low 23 bits -> [1,2) - 1;  rand(rng, Float64): low 52 bits -> [1,2) - 1
__device__ __forceinline__ void trand(Rng &r, float &out) {
    uint32_t bits = ((uint32_t)rng_next(r) & 0x007fffffu) | 0x3f800000u;
    // [1, 2) - 1, as one v_add_f32 with the inline constant: left to the compiler, this subtraction is paired with an
    // unrelated add into a v_pk_add_f32 whose constant operand (x, -1.0) it then keeps in a spilled register pair --
    // 0.8 GB of scratch writes per 1080p frame
    __asm__("v_add_f32_e32 %0, -1.0, %1" : "=v"(out) : "v"(__uint_as_float(bits)));
}
__device__ __forceinline__ void trand(Rng &r, double &out) {
    uint64_t bits = (rng_next(r) & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    out = __longlong_as_double((long long)bits) - 1.0;
}
// src/rand.jl:24  trand(T)*(max-min) + min
template <typename T> __device__ __forceinline__ T random_between(Rng &r, T mn, T mx) {
    T u; trand(r, u);
    return u * (mx - mn) + mn;
}
// One trial of the rejection samplers: src/rand.jl:15-22 (unit ball: x, y, z) and :31-38 (unit
// disk: x, y).  Returns the squared length p.p; the trial is accepted iff it is <= 1 (boundary
// inclusive).  The disk form leaves z = +0, and (x*x + y*y) + 0*0 has the bits of x*x + y*y, so
// one expression serves both; lanes of a wave that need a ball sample and lanes that need a disk
// sample run the SAME loop (rtw_kernels.hpp phase R), each consuming its own stream exactly as
// the reference's two separate loops would.
// random_between(-1, 1) = trand * (1 - (-1)) + (-1) (src/rand.jl:24) in ONE instruction: with f = the [1, 2) float made from
// the generator's bits, trand = f - 1, 2 trand and 2 trand - 1 are all exact (multiples of 2^-22 / 2^-51 below 2 in
// magnitude), so fma(f, 2, -3) -- also exact -- has the same bits as the reference's three roundings.
__device__ __forceinline__ float random_pm1(Rng &r, float) {
    const uint32_t bits = ((uint32_t)rng_next(r) & 0x007fffffu) | 0x3f800000u;
    return __builtin_fmaf(__uint_as_float(bits), 2.0f, -3.0f);
}
__device__ __forceinline__ double random_pm1(Rng &r, double) {
    const uint64_t bits = (rng_next(r) & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    return __builtin_fma(__longlong_as_double((long long)bits), 2.0, -3.0);
}
template <typename T> __device__ __forceinline__ T reject_trial(Rng &r, bool ball, V3<T> &p) {
    p.x = random_pm1(r, T(0));
    p.y = random_pm1(r, T(0));
    p.z = T(0);
    if (ball) p.z = random_pm1(r, T(0));
    return (p.x * p.x + p.y * p.y) + p.z * p.z;
}
// normalize(p) when p.p is already known: StaticArrays' inv(norm(p)) * p with norm = sqrt(p.p)
template <typename T> __device__ __forceinline__ V3<T> normalize_len2(V3<T> p, T len2) {
    return vscale(RTW_RSQRT(len2), p);
}
// src/rand.jl:15-22,29: rejection in the unit ball (x,y,z order, boundary inclusive), normalised
template <typename T> __device__ __forceinline__ V3<T> random_vec3_on_sphere(Rng &r) {
    V3<T> p; T len2;
    do { len2 = reject_trial<T>(r, true, p); } while (!(len2 <= T(1)));
    return normalize_len2(p, len2);
}
// src/rand.jl:31-38
template <typename T> __device__ __forceinline__ void random_vec2_in_disk(Rng &r, T &x, T &y) {
    V3<T> p; T len2;
    do { len2 = reject_trial<T>(r, false, p); } while (!(len2 <= T(1)));
    x = p.x; y = p.y;
}

// ---- intersection (src/hit.jl) ---------------------------------------------------------------
// The per-sphere test of src/hit.jl:13-18.  r2 = r*r is precomputed at upload (same bits as computing it here).
// The deciding arithmetic is selectable (include/rtw_hip.h, RTW_FLAG_NUMERICS_*):
//   NUM_REFERENCE (default)  as the reference evaluates it: `oc . r.dir` and `oc . oc` are StaticArrays' dot -- a callee that
//                            @fastmath does not rewrite: (x1 y1 + x2 y2) + x3 y3, no FMA --, then  c = oc.oc - r^2  and
//                            disc = half_b^2 - c  with one rounding each
//   NUM_REFERENCE_FMA2       the un-fused dots with both squares contracted, disc = fma(half_b, half_b, -c) and c = fma(-r, r, oc.oc): what LLVM makes of src/hit.jl:17-18 on an FMA target when BOTH squares carry
//                            fast-math flags (tools/llvm_fastmath_check: with the flag-less llvm.powi that Julia's pow_fast emits, neither is fused)
//   NUM_CONTRACT             rounds 1 - 4: half_b, r^2 - |oc|^2 and disc as three FMA chains
// (this file is compiled with -ffp-contract=off: the un-fused forms stay un-fused).
// (code 2 was NUM_REFERENCE_FMA -- only the last step contracted -- until ABI 3: removed, no compiler was found to emit it)
enum { NUM_REFERENCE = 0, NUM_CONTRACT = 1, NUM_REFERENCE_FMA2 = 3 };
template <int N> struct NumTag { static constexpr int value = N; };
// `r`: the sphere's radius itself -- read only by NUM_REFERENCE_FMA2 (c = fma(-r, r, oc.oc): the un-rounded square)
template <typename T, int NUM>
__device__ __forceinline__ void sphere_disc_n(T cx, T cy, T cz, T r2, [[maybe_unused]] T r, V3<T> o, V3<T> d, T &half_b, T &disc) {
    const T ocx = o.x - cx, ocy = o.y - cy, ocz = o.z - cz;
    if constexpr (NUM == NUM_CONTRACT) {
        half_b = t_fma(ocz, d.z, t_fma(ocy, d.y, ocx * d.x));
        const T nc = t_fma(-ocz, ocz, t_fma(-ocy, ocy, t_fma(-ocx, ocx, r2)));
        disc = t_fma(half_b, half_b, nc);
    } else {
        half_b = (ocx * d.x + ocy * d.y) + ocz * d.z;                       // src/hit.jl:16
        const T ococ = (ocx * ocx + ocy * ocy) + ocz * ocz;
        // (:17; a PADDING row carries r^2 = -1e30 in the hot array and an arbitrary radius in the cold one: it must stay a miss in every mode)
        const T c = (NUM == NUM_REFERENCE_FMA2 && !(r2 < T(0))) ? t_fma(-r, r, ococ) : ococ - r2;
        if constexpr (NUM == NUM_REFERENCE) disc = half_b * half_b - c;     // :18 (a == 1)
        else disc = t_fma(half_b, half_b, -c);
    }
}
// `num` is wave-uniform (a kernel argument): real scalar branches
template <typename T>
__device__ __forceinline__ void sphere_disc(int num, T cx, T cy, T cz, T r2, T r, V3<T> o, V3<T> d, T &half_b, T &disc) {
    if (num == NUM_REFERENCE) sphere_disc_n<T, NUM_REFERENCE>(cx, cy, cz, r2, r, o, d, half_b, disc);
    else if (num == NUM_CONTRACT) sphere_disc_n<T, NUM_CONTRACT>(cx, cy, cz, r2, r, o, d, half_b, disc);
    else sphere_disc_n<T, NUM_REFERENCE_FMA2>(cx, cy, cz, r2, r, o, d, half_b, disc);
}
// src/hit.jl:19-29: root selection against [tmin, closest]; returns true and the root on a hit
template <typename T>
__device__ __forceinline__ bool sphere_root(T half_b, T disc, T tmin, T closest, T &root) {
    if (disc < T(0)) return false;
    T sqrtd = t_sqrt(disc);
    root = -half_b - sqrtd;
    if (root < tmin || closest < root) {
        root = -half_b + sqrtd;
        if (root < tmin || closest < root) return false;
    }
    return true;
}

template <typename T> struct HitRec { T t; V3<T> p, n; bool front; };

// src/hit.jl:31-34 + ray_to_HitRecord :6-10 + point :3, for the sphere that won the scan
template <typename T>
__device__ __forceinline__ void make_hitrec(V3<T> c, T r, V3<T> o, V3<T> d, T t, HitRec<T> &rec) {
    rec.t = t;
    rec.p = vadd(o, vscale(t, d));
    V3<T> pc = vsub(rec.p, c);
    V3<T> n_out = {RTW_DIV(pc.x, r), RTW_DIV(pc.y, r), RTW_DIV(pc.z, r)};
    rec.front = dot(d, n_out) < T(0);
    rec.n = rec.front ? n_out : vneg(n_out);
}

// ---- light transport (src/light.jl) ----------------------------------------------------------
template <typename T> __device__ __forceinline__ V3<T> reflect(V3<T> v, V3<T> n) {   // :6
    T k = dot(vscale(T(2), v), n);
    return vsub(v, vscale(k, n));
}
// refract (:12-17) is normalize(perp + par); refract_raw is the vector before the normalize
template <typename T> __device__ __forceinline__ V3<T> refract_raw(V3<T> dir, V3<T> n, T ratio) {
    T cos_t = -dot(dir, n);
    if (!(T(1) > cos_t)) cos_t = T(1);
    V3<T> perp = vscale(ratio, vadd(dir, vscale(cos_t, n)));
    T one_m = T(1) - dot(perp, perp);
    T par_s = -t_sqrt(one_m < T(0) ? -one_m : one_m);
    return vadd(perp, vscale(par_s, n));
}
template <typename T> __device__ __forceinline__ V3<T> refract(V3<T> dir, V3<T> n, T ratio) {  // :12-17
    return normalize(refract_raw(dir, n, ratio));
}
template <typename T> __device__ __forceinline__ T reflectance(T cos_t, T ratio) {   // :19-25
    T r0 = (T(1) - ratio) / (T(1) + ratio);
    r0 = r0 * r0;
    T x = T(1) - cos_t;
    T x2 = x * x;
    T x5 = (x2 * x2) * x;
    return r0 + (T(1) - r0) * x5;
}

// ---- materials (src/material.jl:13-53) -------------------------------------------------------
// scatter is split at the random unit vector so that the trace kernel can run ONE rejection loop
// and ONE final normalize for all lanes of a wave whatever they are doing (new camera ray,
// Lambertian, Metal, refraction).  scatter() below composes the parts for one lane (T0 tests).
enum { PATH_READY = 0,   // `vec` is the final direction as the reference leaves it
       PATH_NORM = 1,    // final direction = normalize(vec)
       PATH_BALL = 2 };  // needs a unit-ball sample: scatter_finish(kind, vec, scale, p, p.p)
// What scatter(::Dielectric) derives from the sphere's `ir` alone (src/material.jl:42-43, src/light.jl:20-21): the two
// refraction ratios and Schlick's r0 for each -- two IEEE divisions per dielectric hit, in a branch that a wave takes in 97 % of
// its iterations for 5 % of its lanes.  The upload computes them once per sphere with the SAME operations in T
// (dielectric_constants: host code, -ffp-contract=off), so the bits are the ones the reference's expressions give.
template <typename T> struct DielConst { T inv_ir, r0_front, r0_back; };
template <typename T> __host__ __device__ inline DielConst<T> dielectric_constants(T ir) {
    DielConst<T> c;
    c.inv_ir = T(1) / ir;                                             // ratio for a front face: 1 / ir
    T a = (T(1) - c.inv_ir) / (T(1) + c.inv_ir); c.r0_front = a * a;  // reflectance's r0 (src/light.jl:20-21) with that ratio
    T b = (T(1) - ir) / (T(1) + ir); c.r0_back = b * b;               // ... and with ratio = ir (back face)
    return c;
}
// reflectance (src/light.jl:19-25) from a precomputed r0
template <typename T> __device__ __forceinline__ T reflectance_r0(T cos_t, T r0) {
    T x = T(1) - cos_t;
    T x2 = x * x;
    T x5 = (x2 * x2) * x;
    return r0 + (T(1) - r0) * x5;
}
// `dc`: the sphere's precomputed constants (trace kernel) or nullptr (computed here: the T0 unit ops, scatter())
template <typename T>
__device__ __forceinline__ int scatter_begin(Rng &rng, int kind, T param, V3<T> d_in, const HitRec<T> &rec,
                                             V3<T> &vec, T &scale, const DielConst<T> *dc = nullptr) {
    scale = T(1);
    if (kind == DIELECTRIC) {                                     // :41-53
        T ratio = dc ? (rec.front ? dc->inv_ir : param) : (rec.front ? (T(1) / param) : param);
        T cos_t = -dot(d_in, rec.n);
        if (!(T(1) > cos_t)) cos_t = T(1);
        T sin_t = t_sqrt(T(1) - cos_t * cos_t);
        bool refl = ratio * sin_t > T(1);
        if (!refl) {                                              // :47 short-circuit draw
            T u; trand(rng, u);
            refl = (dc ? reflectance_r0(cos_t, rec.front ? dc->r0_front : dc->r0_back) : reflectance(cos_t, ratio)) > u;
        }
        if (refl) { vec = reflect(d_in, rec.n); return PATH_READY; }   // :48, not re-normalised
        vec = refract_raw(d_in, rec.n, ratio);                          // :50
        return PATH_NORM;
    }
    // Lambertian (:13-23): n + u;  Metal (:31-34): reflect(d, n) + fuzz * u
    if (kind == LAMBERTIAN) { vec = rec.n; } else { vec = reflect(d_in, rec.n); scale = param; }
    return PATH_BALL;
}
// p: the accepted unit-ball sample, len2 = p.p.  scale == 1 for Lambertian (1 * u == u exactly).
template <typename T>
__device__ __forceinline__ int scatter_finish(int kind, V3<T> base, T scale, V3<T> p, T len2, V3<T> &vec) {
    V3<T> uvec = normalize_len2(p, len2);
    V3<T> dir = vadd(base, vscale(scale, uvec));
    if (kind == LAMBERTIAN && near_zero(dir)) { vec = base; return PATH_READY; }   // :15-16
    vec = dir;
    return PATH_NORM;
}
template <typename T> __device__ __forceinline__ V3<T> attenuation_of(int kind, V3<T> albedo) {
    return kind == DIELECTRIC ? V3<T>{T(1), T(1), T(1)} : albedo;
}
template <typename T>
__device__ __forceinline__ void scatter(Rng &rng, int kind, V3<T> albedo, T param, V3<T> d_in,
                                        const HitRec<T> &rec, V3<T> &out_d, V3<T> &att) {
    att = attenuation_of(kind, albedo);
    V3<T> vec; T scale;
    int path = scatter_begin<T>(rng, kind, param, d_in, rec, vec, scale);
    if (path == PATH_BALL) {
        V3<T> p; T len2;
        do { len2 = reject_trial<T>(rng, true, p); } while (!(len2 <= T(1)));
        path = scatter_finish<T>(kind, vec, scale, p, len2, vec);
    }
    out_d = path == PATH_NORM ? normalize(vec) : vec;
}

// ---- sky (src/ray_color.jl:1-6): Float64 constants ------------------------------------------
template <typename T> __device__ __forceinline__ C3 skycolor(V3<T> d) {
    T t = T(0.5) * (d.y + T(1));
    T omt = T(1) - t;
    return {(double)omt * 1.0 + (double)t * 0.5, (double)omt * 1.0 + (double)t * 0.7,
            (double)omt * 1.0 + (double)t * 1.0};
}

// ---- camera (src/camera.jl:43-48) ------------------------------------------------------------
template <typename T> struct Camera {
    T origin[3], llc[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    T lens_radius;
};
// get_ray after the lens sample (dx, dy): origin and the un-normalised direction (:45-47)
template <typename T>
__device__ __forceinline__ void camera_ray_raw(const Camera<T> &cam, T s, T t, T dx, T dy, V3<T> &ro, V3<T> &raw) {
    T rx = cam.lens_radius * dx, ry = cam.lens_radius * dy;
    V3<T> cu = {cam.u[0], cam.u[1], cam.u[2]}, cv = {cam.v[0], cam.v[1], cam.v[2]};
    V3<T> org = {cam.origin[0], cam.origin[1], cam.origin[2]};
    V3<T> llc = {cam.llc[0], cam.llc[1], cam.llc[2]};
    V3<T> hor = {cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]};
    V3<T> ver = {cam.vertical[0], cam.vertical[1], cam.vertical[2]};
    V3<T> offset = vadd(vscale(rx, cu), vscale(ry, cv));
    ro = vadd(org, offset);
    V3<T> dir = vadd(llc, vscale(s, hor));
    dir = vadd(dir, vscale(t, ver));
    dir = vsub(dir, org);
    raw = vsub(dir, offset);
}
template <typename T>
__device__ __forceinline__ void get_ray(Rng &rng, const Camera<T> &cam, T s, T t, V3<T> &ro, V3<T> &rd) {
    T dx, dy;
    random_vec2_in_disk(rng, dx, dy);
    V3<T> raw;
    camera_ray_raw(cam, s, t, dx, dy, ro, raw);
    rd = normalize(raw);
}

// ---- exact pixel accumulation (oracle/rtw_oracle.c fx_add) --------------
// A chunk sum (binary64) as signed 64.64 fixed point in two's complement (hi:lo).  Exact for
// magnitudes in [2^-11, 2^31); smaller ones are truncated towards zero at 2^-64.  false = the
// value is NaN, infinite or >= 2^31: it poisons the pixel.
__device__ __forceinline__ bool fx_from_double(double x, unsigned long long &lo, unsigned long long &hi) {
    const double a = __builtin_fabs(x);
    if (!(a < 2147483648.0)) return false;
    const unsigned ip = (unsigned)a;                       // trunc(|x|)
    const double fr = a - (double)ip;                      // exact, in [0, 1)
    const double y = fr * 4294967296.0;                    // exact
    const unsigned p1 = (unsigned)y;
    const double r1 = y - (double)p1;                      // exact, in [0, 1)
    const unsigned p0 = (unsigned)(r1 * 4294967296.0);     // truncated at 2^-64
    unsigned long long l = ((unsigned long long)p1 << 32) | (unsigned long long)p0, h = ip;
    if (x < 0.0) { l = 0ull - l; h = ~h + (l == 0ull ? 1ull : 0ull); }
    lo = l; hi = h;
    return true;
}
// the 128-bit sum rounded once to binary64, round to nearest, ties to even
__device__ __forceinline__ double fx_to_double(unsigned long long lo, unsigned long long hi) {
    const bool neg = (long long)hi < 0;
    if (neg) { lo = 0ull - lo; hi = ~hi + (lo == 0ull ? 1ull : 0ull); }
    if ((lo | hi) == 0ull) return 0.0;
    // normalise: shift left until bit 127 is set; the top 53 bits are the significand, the rest decides the rounding
    const int n = hi ? __clzll((long long)hi) : 64 + __clzll((long long)lo);
    if (n >= 64) { hi = lo << (n - 64); lo = 0ull; }
    else if (n > 0) { hi = (hi << n) | (lo >> (64 - n)); lo <<= n; }
    unsigned long long mant = hi >> 11;
    const unsigned rem = (unsigned)hi & 0x7ffu;
    const bool sticky = lo != 0ull;
    if (rem > 0x400u || (rem == 0x400u && (sticky || (mant & 1ull)))) mant += 1ull;
    const double v = __builtin_ldexp((double)mant, 11 - n);      // value = mant * 2^(127 - n - 52) / 2^64
    return neg ? -v : v;
}

}  // namespace rtw
