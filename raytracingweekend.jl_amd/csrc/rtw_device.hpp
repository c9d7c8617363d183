// rtw_device.hpp -- device-side building blocks of the hot path, gfx950 (MI355X) only.
//
// Each function restates one reference function (paths relative to /root/reference) under the
// numerics contract of DESIGN.md section 4: IEEE binary32/64, one rounding per written
// operation (this file is compiled with -ffp-contract=off), explicit FMA only in the
// ray-sphere discriminant.  Float32 mode is the reference's mixed precision:
// geometry / RNG / scatter in T, sky colour, throughput and accumulation in double.
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

__device__ __forceinline__ float t_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return __builtin_sqrt(x); }
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
#ifdef RTW_PROBE_FASTDIV   // time probe (WRONG image): approximate reciprocal square root / reciprocal instead of the IEEE sqrt and divisions
__device__ __forceinline__ float probe_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double probe_rsq(double x) { return 1.0 / __builtin_sqrt(x); }
__device__ __forceinline__ float probe_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double probe_rcp(double x) { return 1.0 / x; }
#endif
template <typename T> __device__ __forceinline__ V3<T> normalize(V3<T> a) {
#ifdef RTW_PROBE_FASTDIV
    T inv = probe_rsq(dot(a, a));
#else
    T inv = T(1) / t_sqrt(dot(a, a));
#endif
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
#ifdef RTW_PROBE_FASTDIV
    return vscale(probe_rsq(len2), p);
#else
    return vscale(T(1) / t_sqrt(len2), p);
#endif
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
//   NUM_REFERENCE_FMA        the same with the last step contracted: disc = fma(half_b, half_b, -c)
//   NUM_REFERENCE_FMA2       ... and c = fma(-r, r, oc.oc) as well: what LLVM makes of src/hit.jl:17-18 on an FMA target when BOTH squares carry
//                            fast-math flags (tools/llvm_fastmath_check: with the flag-less llvm.powi that Julia's pow_fast emits, neither is fused)
//   NUM_CONTRACT             rounds 1 - 4: half_b, r^2 - |oc|^2 and disc as three FMA chains
// (this file is compiled with -ffp-contract=off: the un-fused forms stay un-fused).
enum { NUM_REFERENCE = 0, NUM_CONTRACT = 1, NUM_REFERENCE_FMA = 2, NUM_REFERENCE_FMA2 = 3 };
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
    else if (num == NUM_REFERENCE_FMA) sphere_disc_n<T, NUM_REFERENCE_FMA>(cx, cy, cz, r2, r, o, d, half_b, disc);
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
#ifdef RTW_PROBE_FASTDIV
    const T ir_ = probe_rcp(r);
    V3<T> n_out = {pc.x * ir_, pc.y * ir_, pc.z * ir_};
#else
    V3<T> n_out = {pc.x / r, pc.y / r, pc.z / r};
#endif
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

struct NoClock { __device__ __forceinline__ void lap(int) {} __device__ __forceinline__ void count(int, unsigned) {} };

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
    else if (num == NUM_REFERENCE_FMA) resolve_candidates_n<T, STRIDE, NUM_REFERENCE_FMA>(src, rad, o, d, tmin, closest, idx, list, cnt);
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
                // The scalar stream carries r^2, not r: pass 1 evaluates the reference_fma form and adds a margin that covers what the
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
    else if (w.numerics == NUM_REFERENCE_FMA) scan(NumTag<NUM_REFERENCE_FMA>{});
    else scan(NumTag<NUM_REFERENCE_FMA2>{});
    resolve_candidates<T, STRIDE>(w.numerics, src, w.mat0, o, d, tmin, closest, idx, list, cnt);
#ifdef RTW_DUP_RESOLVE   // instruction-count probe: the final resolve twice (idempotent: same winner)
    { T c2 = tmax; int i2 = -1; __asm__ volatile("" : "+v"(c2), "+v"(i2));
      resolve_candidates<T, STRIDE>(w.numerics, src, w.mat0, o, d, tmin, c2, i2, list, cnt);
      __asm__ volatile("" :: "v"(c2), "v"(i2)); }
#endif
    clk.lap(5);
    t_hit = closest;
    return idx;
}

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
#define RTW_PAIR_CAP 512     // (owner, sphere) pairs per wave; a full list is resolved early
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
//     [axis][0: lo_b <= edge, indexed by the bin of hi | 1: hi_b >= edge, indexed by the bin of lo][bin], then {BIG blocks, live blocks, 0, 0}
// (BIG: touched by every ray; live: what a ray without the filter touches; dead blocks are in no set).
// 64 bins: one table = 64 words = one word per LDS bank, any 64 look-ups are conflict-free; 128 bins measured 8 % SLOWER (317 vs 293 ms).
#ifndef RTW_CULL_BINS
#define RTW_CULL_BINS 64
#endif
#define RTW_CULL_TAB_WORDS (6 * RTW_CULL_BINS + 4)
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
    int n_huge, huge[2];  // huge spheres (device order), tested in-lane like DevScene::huge
    CullGrid grid;        // the block vote: bins ...
    const unsigned *tab;  // ... and tables (global memory, or the workgroup's copy in LDS)
};

struct WaveScratch {
    unsigned *pairs;              // RTW_PAIR_CAP entries: recording lane << 16 | block << 5 | bit (see resolve_pairs)
    unsigned long long *keys;     // 64 entries: Float32 (root bits << 32 | ~sphere); Float64 root bits
    unsigned *kidx;               // Float64 only: 64 entries, sphere + 1
    unsigned cap = RTW_PAIR_CAP;  // entries in `pairs` (wave-uniform; the ray-pool kernel gives a wave 256)
};

// x = p1 + p2 with p1 = RN16(x), p2 = RN16(x - p1); returns p1 | p2 << 16
// Two instructions: v_cvt_f16_f32 writes p1 to the low half, v_fma_mixhi_f16 computes x * 1.0 - p1 with the f16 operand read in place
// (exact in binary32: p1 is x rounded to 11 bits) and rounds it once into the high half -- the same bits as the five instructions the
// compiler makes of the C form (convert, convert back, subtract, convert, pack): 11 splits per scan, 34 VALU instructions fewer.
__device__ __forceinline__ unsigned split_f16(float x) {
#ifdef RTW_SPLIT_C_FORM
    const _Float16 p1 = (_Float16)x;
    const _Float16 p2 = (_Float16)(x - (float)p1);
    return (unsigned)__builtin_bit_cast(unsigned short, p1) | ((unsigned)__builtin_bit_cast(unsigned short, p2) << 16);
#else
    unsigned w;
    __asm__("v_cvt_f16_f32_e32 %0, %1\n\tv_fma_mixhi_f16 %0, %1, 1.0, -%0 op_sel_hi:[0,0,1]" : "=&v"(w) : "v"(x));
    return w;
#endif
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

// Pass 1 flags GROUPS of RTW_SCAN_GROUP spheres (consecutive result registers of one lane = consecutive spheres): the sign
// bits of a group's filter values are ANDed with fast-class bit operations (v_bitop3_b32 / v_and_b32: 2.4 cycles) and only
// the group's bit goes through the slow-class v_alignbit_b32 (4.3 cycles) -- 16 + 8 instead of 32 instructions per block of
// 32 spheres; pass 2 applies the exact test to every member of a flagged group.  1 (default): one sphere per list entry.
// Measured (1080p x 1000 spp Float32, same box, with the wave-level early-out): single spheres 370.9 ms, groups of 2 380.5,
// of 4 388.0 -- pass 2 pays one exact test (sqrt, root selection, LDS atomic) per member of a flagged group, more than the
// alignbits saved.  Kept for A/B runs.
#ifndef RTW_SCAN_GROUP
#define RTW_SCAN_GROUP 1
#endif
static_assert(RTW_SCAN_GROUP == 1 || RTW_SCAN_GROUP == 2 || RTW_SCAN_GROUP == 4, "groups of 1, 2 or 4 result registers");
#ifndef RTW_SCAN_CMP
#define RTW_SCAN_CMP 0       // 1: sign collection by v_cmp -> SGPR lane masks instead of v_alignbit + extraction loop (experiment, rejected)
#endif
#ifndef RTW_SCAN_SKIP
#define RTW_SCAN_SKIP 1      // wave-level early-out per half block (hit_world_mfma): 372.2 vs 376.3 ms.  (Left to the compiler it is
                             // if-converted -- both sides executed -- and gains nothing: the sign collection's side is fenced by an asm.)
#endif

#ifdef RTW_CAND_HIST     // debug build: candidates of the filter per sphere, [2 i] = flagged but contract discriminant < 0, [2 i + 1] = discriminant >= 0
__device__ unsigned g_cand_hist[8192];
#endif
// Walk n entries of the wave's list (all 64 lanes; src = the scene's geom in LDS or global memory).
struct NoOrig {};
template <typename T, bool WITH_R, typename SRC, typename ORIG = NoOrig>
__device__ __forceinline__ void resolve_pairs_impl(int num, SRC src, [[maybe_unused]] const typename Vec4<T>::type *rad, V3<T> o, V3<T> d, T tmin, const WaveScratch &ws, unsigned n, unsigned lane, ORIG orig = ORIG()) {
    constexpr bool CULLED = !__is_same(ORIG, NoOrig);      // device order != the caller's order: ties go by orig[], the key carries both
    constexpr unsigned G = RTW_SCAN_GROUP;
    using V4 = typename Vec4<T>::type;
    __builtin_amdgcn_wave_barrier();
    for (unsigned p0 = 0; p0 < n; p0 += 64u) {
        const unsigned p = p0 + lane;
        const bool valid = p < n;
        const unsigned e = ws.pairs[valid ? p : 0u];
        // entry = recording lane (H, j) << 16 | block << 5 | b.   G = 1: b = half << 4 | result register: ray j + 32 (b >> 4), sphere
        // 32 block + 16 H + (b & 15).   G = 2 / 4: b = half << (3 / 2) | group: ray j + 32 half, spheres 32 block + 16 H + G group + 0..G-1
        unsigned owner, sph0;
        if constexpr (G == 1) { owner = ((e >> 16) & 31u) + ((e & 16u) << 1); sph0 = (e & 0xffefu) + ((e >> 17) & 16u); }
        else if constexpr (G == 2) { owner = ((e >> 16) & 31u) + ((e & 8u) << 2); sph0 = (e & 0xffe0u) + ((e >> 17) & 16u) + ((e & 7u) << 1); }
        else { owner = ((e >> 16) & 31u) + ((e & 4u) << 3); sph0 = (e & 0xffe0u) + ((e >> 17) & 16u) + ((e & 3u) << 2); }
        const V3<T> po = {lane_get(o.x, owner), lane_get(o.y, owner), lane_get(o.z, owner)};
        const V3<T> pd = {lane_get(d.x, owner), lane_get(d.y, owner), lane_get(d.z, owner)};
        V4 sg[G];
#pragma unroll
        for (unsigned m = 0; m < G; ++m) sg[m] = src[sph0 + m];
#pragma unroll
        for (unsigned m = 0; m < G; ++m) {
            const unsigned sph = sph0 + m;
            const V4 s = sg[m];
            T hb, disc, root = 0;
            if constexpr (WITH_R) sphere_disc_n<T, NUM_REFERENCE_FMA2>(s.x, s.y, s.z, s.w, rad[sph].x, po, pd, hb, disc);   // (mat0: the radius itself; the LDS copy holds r^2)
            else sphere_disc<T>(num, s.x, s.y, s.z, s.w, T(0), po, pd, hb, disc);
#ifdef RTW_CAND_HIST
            if (valid && sph < 4096u) atomicAdd(&g_cand_hist[2u * sph + (disc < T(0) ? 0u : 1u)], 1u);
#endif
            if (G > 1 && !__any(valid && !(disc < T(0)))) continue;           // no entry has a candidate at this position
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
    __builtin_amdgcn_wave_barrier();
}

// (NUM_REFERENCE_FMA2 reads the radius from mat0: its own copy of the loop, so that the other modes keep their registers)
template <typename T, typename SRC, typename RAD, typename ORIG = NoOrig>
__device__ __forceinline__ void resolve_pairs(int num, SRC src, RAD rad, V3<T> o, V3<T> d, T tmin, const WaveScratch &ws, unsigned n, unsigned lane, ORIG orig = ORIG()) {
    if (num == NUM_REFERENCE_FMA2) resolve_pairs_impl<T, true>(num, src, rad(), o, d, tmin, ws, n, lane, orig);
    else resolve_pairs_impl<T, false>(num, src, nullptr, o, d, tmin, ws, n, lane, orig);
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
    const uint4 *pa = (CULLED ? mc->ops : w.mf_ops) + lane;
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
    if constexpr (!CULLED) { A1 = pa[0]; A2 = pa[64]; }
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
#ifdef RTW_DUP_OPERANDS   // time probe: the ray-operand build (features, f16 splits, word assembly, lane exchange) a second time, same result
    {
        float ox2 = ox, oy2 = oy, oz2 = oz, dx2_ = dx, dy2_ = dy, dz2_ = dz;
        __asm__ volatile("" : "+v"(ox2), "+v"(oy2), "+v"(oz2), "+v"(dx2_), "+v"(dy2_), "+v"(dz2_));
        const float q_ = __builtin_fmaf(oz2, dz2_, __builtin_fmaf(oy2, dy2_, ox2 * dx2_));
        const float oo_ = __builtin_fmaf(oz2, oz2, __builtin_fmaf(oy2, oy2, ox2 * ox2));
        const float o1_ = (__builtin_fabsf(ox2) + __builtin_fabsf(oy2)) + __builtin_fabsf(oz2);
        const float oop_ = __builtin_fmaf(oo_, w.mf_oo_keep, -(w.mf_o1_coef * o1_));
        const float tq_ = w.mf_sigma2 * __builtin_fmaf(q_, q_, -oop_);
        const float fp_[3] = {__builtin_fmaf(-q_, dx2_, ox2) * zs2, __builtin_fmaf(-q_, dy2_, oy2) * zs2, __builtin_fmaf(-q_, dz2_, oz2) * zs2};
        const float ax = dx2_ * z2, ay = dy2_ * z2, az = dz2_ * z2;
        const float fq_[6] = {ax * dx2_, ay * dy2_, az * dz2_, (ax + ax) * dy2_, (ax + ax) * dz2_, (ay + ay) * dz2_};
        unsigned sq_[6], sp_[3];
#pragma unroll
        for (int k = 0; k < 6; ++k) sq_[k] = split_f16(fq_[k]);
#pragma unroll
        for (int k = 0; k < 3; ++k) sp_[k] = split_f16(fp_[k]);
        const float tx_ = ok ? tq_ : tx;
        const _Float16 t1_ = (_Float16)(tx_ * (1.0f / 32768.0f));
        const unsigned x23_ = split_f16((tx_ - 32768.0f * (float)t1_) * (1.0f / 16.0f));
        const unsigned a0[8] = {dup(sq_[0]), cat(sq_[0], sq_[1]), sq_[1], dup(sq_[2]), sq_[5], dup(sp_[0]), cat(sp_[0], sp_[1]), sp_[1]};
        const unsigned a1[8] = {cat(sq_[2], sq_[3]), sq_[3], dup(sq_[4]), cat(sq_[4], sq_[5]), dup(sp_[2]), cat(sp_[2], sb), ss | ((unsigned)__builtin_bit_cast(unsigned short, t1_) << 16), x23_};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const auto sw2 = __builtin_amdgcn_permlane32_swap(a0[k], a1[k], false, false);
            __asm__ volatile("" :: "v"(sw2[0]), "v"(sw2[1]));
        }
    }
#endif
    rtw_h8 B1[2], B2[2];
    {
        const uint4 q10 = {h0[0], h0[1], h0[2], h0[3]}, q11 = {h1[0], h1[1], h1[2], h1[3]};
        const uint4 q20 = {h0[4], h0[5], h0[6], h0[7]}, q21 = {h1[4], h1[5], h1[6], h1[7]};
        B1[0] = __builtin_bit_cast(rtw_h8, q10); B1[1] = __builtin_bit_cast(rtw_h8, q11);
        B2[0] = __builtin_bit_cast(rtw_h8, q20); B2[1] = __builtin_bit_cast(rtw_h8, q21);
    }
    // ---- the result cells, initialised with the lane's own exact test of the scene's huge spheres (DevScene::huge / MfmaCull::huge):
    //      the same contract test and the same key / tie rule as pass 2, so the minimum over all candidates is unchanged ----
    {
        unsigned long long key0 = ~0ull;
        [[maybe_unused]] unsigned kidx0 = 0u;
        using V4 = typename Vec4<T>::type;
        const int n_huge = CULLED ? mc->n_huge : w.n_huge;
        for (int hgi = 0; hgi < n_huge; ++hgi) {
            const int si = CULLED ? (hgi == 0 ? mc->huge[0] : mc->huge[1]) : (hgi == 0 ? w.huge[0] : w.huge[1]);      // (no dynamic indexing of a by-value struct: that would live in scratch)
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

    const unsigned lane_const = lane << 16;
    unsigned total = 0;                                   // wave-uniform
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
        A1 = pa[blk * 128]; A2 = pa[blk * 128 + 64];
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
        auto eval = [&](const rtw_f16v &Wv) {
            if constexpr (RTW_SCAN_GROUP == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(Wv[r]), 31);
            } else if constexpr (RTW_SCAN_GROUP == 2) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(Wv[r]) & __float_as_uint(Wv[r + 1]), 31);
            } else {
                // sign of (a & b & c & d) is set iff all four filter values are negative: no member is a candidate
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const unsigned g = __builtin_amdgcn_bitop3_b32(__float_as_uint(Wv[r]), __float_as_uint(Wv[r + 1]), __float_as_uint(Wv[r + 2]), 0x80) &
                                       __float_as_uint(Wv[r + 3]);
                    mask = __builtin_amdgcn_alignbit(mask, g, 31);
                }
            }
        };
        // Half of the (wave, block) evaluations find no candidate in ANY lane (rays of a wave are neighbours): the sign bits of
        // a half block's 16 filter values are ANDed first (8 FMA-class v_bitop3_b32 / v_and_b32) and the 16 slow-class
        // v_alignbit_b32 run only when some lane has a non-negative value.  true = no lane has a candidate in Wv.
        auto none = [&](const rtw_f16v &Wv) -> bool {
            unsigned t = __builtin_amdgcn_bitop3_b32(__float_as_uint(Wv[0]), __float_as_uint(Wv[1]), __float_as_uint(Wv[2]), 0x80);
#pragma unroll
            for (int r = 3; r < 15; r += 2) t = __builtin_amdgcn_bitop3_b32(t, __float_as_uint(Wv[r]), __float_as_uint(Wv[r + 1]), 0x80);
            t &= __float_as_uint(Wv[15]);
            return !__any((int)t >= 0);
        };
#if RTW_SCAN_CMP
        // Experiment: one v_cmp_ge_f32 per result register -> a 64-bit lane mask in SGPRs; a non-empty mask
        // records its lanes' candidates at once (entry = recording lane << 16 | block << 5 | half << 4 | register), no per-lane mask
        // word, no extraction loop.  Measured: see DESIGN.md section 6.3.
        auto record_cmp = [&](const rtw_f16v &Wv, unsigned half16, int blk_) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned long long cm = __ballot(!(Wv[r] < 0.0f));
                if (cm) {
                    if (total + 64u > ws.cap) { resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig); total = 0; }
                    if (!(Wv[r] < 0.0f))
                        ws.pairs[__builtin_amdgcn_mbcnt_hi((unsigned)(cm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)cm, total))] = lane_const + (unsigned)blk_ * 32u + half16 + (unsigned)r;
                    total += (unsigned)__popcll(cm);
                }
            }
        };
#endif
        // the filter values of one half block (32 spheres x 32 rays): two chained MFMAs.  Time probes (tools/gpu_probe_phases.sh):
        // -DRTW_DUP_MFMA=k executes the pair k more times (same result); -DRTW_PROBE_NO_MFMA replaces it by a constant "no candidate"
        // (WRONG image: only the in-lane huge spheres are ever hit -- what the kernel costs per wave-segment WITHOUT the matrix pipe).
        auto filter_pair = [&](const uint4 &a1, const uint4 &a2, const rtw_h8 &b1, const rtw_h8 &b2) -> rtw_f16v {
#ifdef RTW_PROBE_NO_MFMA
            rtw_f16v Wn = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
            const uint4 u1 = __builtin_bit_cast(uint4, b1), u2 = __builtin_bit_cast(uint4, b2);
            __asm__ volatile("" :: "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(a2.x), "v"(a2.y), "v"(a2.z), "v"(a2.w));   // (the operands stay live: the loads
            __asm__ volatile("" :: "v"(u1.x), "v"(u1.y), "v"(u1.z), "v"(u1.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w));   //  and the ray operands are still made)
            __asm__ volatile("" : "+v"(Wn));
            return Wn;
#else
            rtw_f16v Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a1), b1, zero, 0, 0, 0);
            Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a2), b2, Wp, 0, 0, 0);
#ifdef RTW_DUP_MFMA
#pragma unroll
            for (int rep = 0; rep < (RTW_DUP_MFMA + 0 > 0 ? RTW_DUP_MFMA + 0 : 1); ++rep) {
                // (the repeated pair takes its sphere operand through an opaque copy and starts from the previous result x 0: left
                //  as the same expression it is merged with the first pair -- rounds 3 and 4 measured 16 register copies, not MFMAs)
                uint4 a1c = a1;
                __asm__ volatile("" : "+v"(a1c.x), "+v"(a1c.y), "+v"(a1c.z), "+v"(a1c.w));
                __asm__ volatile("" : "+v"(Wp));
                Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a1c), b1, zero, 0, 0, 0);
                Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a2), b2, Wp, 0, 0, 0);
            }
#endif
            return Wp;
#endif
        };
        constexpr unsigned HB = 16u / RTW_SCAN_GROUP;       // mask bits per half block
        {
            rtw_f16v Wv = zero;
            if (!CULLED || do_half0) Wv = filter_pair(A1, A2, B1[0], B2[0]);
#ifdef RTW_DUP_EVAL      // time probe: the sign collection twice
            { unsigned keep = mask; eval(Wv); __asm__ volatile("" :: "v"(mask)); mask = keep; __asm__ volatile("" : "+v"(Wv)); }   // (no CSE with the real one)
#endif
#if RTW_SCAN_CMP
            if ((!CULLED || do_half0) && !(RTW_SCAN_SKIP && none(Wv))) { if (RTW_SCAN_SKIP) __asm__ volatile("" : "+v"(Wv)); record_cmp(Wv, 0u, cur); }
#else
            if ((CULLED && !do_half0) || (RTW_SCAN_SKIP && none(Wv))) mask = (1u << HB) - 1u;          // all negative
            else { if (RTW_SCAN_SKIP) __asm__ volatile("" : "+v"(Wv)); eval(Wv); any_cand = true; }      // (the asm keeps it a real branch: no if-conversion)
#endif
        }
        {
            // the next block's operands are fetched as soon as this block's last use of each is issued (one block of
            // padding at the end), so only one set of A registers is live during the evaluation
            rtw_f16v Wv = zero;
            if (!CULLED || do_half1) Wv = filter_pair(A1, A2, B1[1], B2[1]);
            __builtin_amdgcn_sched_barrier(0);
            A1 = pa[blk * 128]; A2 = pa[blk * 128 + 64];
            __builtin_amdgcn_sched_barrier(0);
#ifdef RTW_DUP_EVAL
            { unsigned keep = mask; eval(Wv); __asm__ volatile("" :: "v"(mask)); mask = keep; __asm__ volatile("" : "+v"(Wv)); }   // (no CSE with the real one)
#endif
#if RTW_SCAN_CMP
            if ((!CULLED || do_half1) && !(RTW_SCAN_SKIP && none(Wv))) { if (RTW_SCAN_SKIP) __asm__ volatile("" : "+v"(Wv)); record_cmp(Wv, 16u, cur); }
#else
            if ((CULLED && !do_half1) || (RTW_SCAN_SKIP && none(Wv))) mask = (mask << HB) | ((1u << HB) - 1u);
            else { if (RTW_SCAN_SKIP) __asm__ volatile("" : "+v"(Wv)); eval(Wv); any_cand = true; }
#endif
        }
#if RTW_SCAN_CMP
        clk.lap(2);
        continue;            // (the candidates of this block are already in the list)
#endif
        clk.lap(2);
        if (RTW_SCAN_SKIP && !any_cand) {                    // no lane has a candidate in this block: nothing to extract
            if constexpr (!CULLED) { clk.count(7, 1u); clk.count(6, 1u); }
            continue;
        }
        constexpr unsigned NB = 32u / RTW_SCAN_GROUP;     // list bits per block: bit NB - 1 - b, b = half wave << (4 | 2) | result register / group
        unsigned m = ~mask;
        if constexpr (NB < 32u) m &= (1u << NB) - 1u;
        if constexpr (!CULLED) {                          // (phase-profile build only: blocks, and blocks without any candidate)
            clk.count(7, 1u);
            if (!__any(m != 0u)) clk.count(6, 1u);
        }
        const unsigned code0 = lane_const + (unsigned)cur * 32u + (NB - 1u);      // entry = recording lane << 16 | block << 5 | b
#ifdef RTW_DUP_EXTRACT   // instruction/time probe: the extraction loop twice (the first run writes the same entries)
        { unsigned m2 = m, t2 = total;   // (probe)
          for (;;) {
              const unsigned long long act2 = __ballot(m2 != 0u);
              if (!act2 || t2 + 64u > RTW_PAIR_CAP) break;
              if (m2 != 0u) {
                  const unsigned z2 = (unsigned)__builtin_ctz(m2);
                  m2 &= m2 - 1u;
                  ws.pairs[__builtin_amdgcn_mbcnt_hi((unsigned)(act2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act2, t2))] = code0 - z2;
              }
              t2 += (unsigned)__popcll(act2);
          }
          __builtin_amdgcn_wave_barrier(); }
#endif
        for (;;) {
            const unsigned long long act = __ballot(m != 0u);
            if (!act) break;
            if (total + 64u > ws.cap) {
                clk.lap(4);
                resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig);
                total = 0;
                clk.lap(5);
            }
            if (m != 0u) {
                const unsigned z = (unsigned)__builtin_ctz(m);
                m &= m - 1u;
                const unsigned pos = __builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, total));
                ws.pairs[pos] = code0 - z;
            }
            total += (unsigned)__popcll(act);
        }
        clk.lap(4);
    }
    }
    if (use_prio) __builtin_amdgcn_s_setprio(sizeof(T) == 4 ? 1 : 0);
    resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig);
#ifdef RTW_DUP_RESOLVE_PAIRS   // instruction/time probe: the final resolve twice (idempotent: min / max of the same keys)
    resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig);
#endif
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
    int n_huge, huge[2];                   // huge spheres in this order (see DevScene::huge)
    CullGrid grid;                         // the block vote (tables behind mf_box)
    int numerics;                          // NUM_* (see DevScene::numerics)
};
template <typename T> __host__ __device__ inline MfmaCull mfma_cull_of(const CullScene<T> &c) {
    return MfmaCull{c.mf_ops, c.mf_box, c.mf_blocks, {(float)c.cs[0], (float)c.cs[1], (float)c.cs[2]}, (float)c.rs * 1.000001f + 1e-30f, (const void *)c.mat0,
                    {c.mf_glo[0], c.mf_glo[1], c.mf_glo[2]}, {c.mf_ghi[0], c.mf_ghi[1], c.mf_ghi[2]}, c.n_huge, {c.huge[0], c.huge[1]}, c.grid, reinterpret_cast<const unsigned *>(c.mf_box + 8 * (c.mf_blocks + 1))};
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
