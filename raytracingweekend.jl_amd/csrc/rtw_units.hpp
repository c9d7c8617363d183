// rtw_units.hpp -- unit-level device entry points for the T0 parity tier: one lane evaluates
// one reference function on one input, using exactly the device functions the trace kernel
// uses.  All I/O is in 8-byte slots: doubles for real values (exact for binary32 inputs),
// raw uint64 for RNG state words.  Slot layouts are documented in tests/test_gpu_units.py.
#pragma once
#include "rtw_device.hpp"

namespace rtw {

enum UnitOp {
    U_HIT_SPHERE = 0, U_REFLECT = 1, U_REFRACT = 2, U_REFLECTANCE = 3, U_SCATTER = 4,
    U_GET_RAY = 5, U_SKYCOLOR = 6, U_RNG = 7, U_HIT_WORLD = 8, U_RAY_COLOR = 9,
    U_HIT_WORLD_LDS = 10,    // hit_world with the scene staged in LDS (the trace kernel's instantiation)
    U_HIT_WORLD_CULL = 11,   // hit_world_cull (RTW_FLAG_GROUP_CULL), scene staged in LDS
    U_FX_SUM = 12,           // exact 64.64 accumulation of 8 doubles -> rounded sum, poison count
    U_HIT_WORLD_MFMA = 13,   // hit_world_mfma (pass 1 on the matrix pipe), scene staged in LDS; tmin / tmax of ray 0 serve the whole launch
    U_HIT_WORLD_MFMA_CULL = 14,   // hit_world_mfma with block culling (RTW_FLAG_GROUP_CULL on the matrix pipe), cull layout staged in LDS
    U_NEAR_ZERO = 15,        // near_zero(v) (src/vec.jl:20): squared length against the Float64 literal 1e-5
    U_EXACT_MATH = 16,       // t_sqrt / t_rcp against the compiler's IEEE sqrt / division on a RANGE of binary32 bit patterns (Float32 only)
    U_NUM_OPS = 17
};

__host__ __device__ inline int unit_in_slots(int op) {
    switch (op) {
        case U_HIT_SPHERE: return 12;   // c[3], r, o[3], d[3], tmin, tmax
        case U_REFLECT: return 6;       // v[3], n[3]
        case U_REFRACT: return 7;       // d[3], n[3], ratio
        case U_REFLECTANCE: return 2;   // cos, ratio
        case U_SCATTER: return 18;      // state[2], kind, albedo[3], param, d[3], rec{t,p[3],n[3],front}
        case U_GET_RAY: return 4;       // state[2], s, t
        case U_SKYCOLOR: return 3;      // d[3]
        case U_RNG: return 2;           // state[2]
        case U_NEAR_ZERO: return 3;     // v[3]
        case U_EXACT_MATH: return 2;    // first bit pattern, number of consecutive patterns
        case U_HIT_WORLD: case U_HIT_WORLD_LDS: case U_HIT_WORLD_CULL: case U_HIT_WORLD_MFMA: case U_HIT_WORLD_MFMA_CULL: return 8;     // o[3], d[3], tmin, tmax
        case U_RAY_COLOR: return 9;     // state[2], o[3], d[3], depth
        case U_FX_SUM: return 8;        // 8 binary64 values
    }
    return 0;
}
__host__ __device__ inline int unit_out_slots(int op) {
    switch (op) {
        case U_HIT_SPHERE: return 9;    // hit, t, p[3], n[3], front
        case U_REFLECT: return 3;
        case U_REFRACT: return 3;
        case U_REFLECTANCE: return 1;
        case U_SCATTER: return 11;      // state[2], o[3], d[3], att[3]
        case U_GET_RAY: return 8;       // state[2], o[3], d[3]
        case U_SKYCOLOR: return 3;
        case U_RNG: return 6;           // state[2], 4 uniforms
        case U_NEAR_ZERO: return 2;     // near_zero(v), squared_length(v) (src/vec.jl:19-20)
        case U_EXACT_MATH: return 4;    // mismatches of t_sqrt, of t_rcp, first bad pattern of each (or -1)
        case U_HIT_WORLD: case U_HIT_WORLD_LDS: case U_HIT_WORLD_CULL: case U_HIT_WORLD_MFMA: case U_HIT_WORLD_MFMA_CULL: return 9;     // idx, t, p[3], n[3], front
        case U_RAY_COLOR: return 6;     // state[2], colour[3], segments
        case U_FX_SUM: return 2;        // sum, poisoned
    }
    return 0;
}

template <typename T>
__device__ __forceinline__ V3<T> ld3(const double *p) { return {(T)p[0], (T)p[1], (T)p[2]}; }
template <typename T>
__device__ __forceinline__ void st3(double *p, V3<T> v) { p[0] = (double)v.x; p[1] = (double)v.y; p[2] = (double)v.z; }
__device__ __forceinline__ uint64_t as_u64(double d) { return (uint64_t)__double_as_longlong(d); }
__device__ __forceinline__ double as_f64(uint64_t u) { return __longlong_as_double((long long)u); }

template <typename T>
__global__ void unit_kernel(int op, int count, const double *__restrict__ in, double *__restrict__ out,
                            DevScene<T> scene, CullScene<T> cull, Camera<T> cam) {
    using V4 = typename Vec4<T>::type;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    __shared__ unsigned short s_list[RTW_LIST_CAP * 64];    // launched with 64-thread blocks
    extern __shared__ __attribute__((aligned(16))) unsigned char u_smem[];   // ops 10, 11: the staged scene
    unsigned short *my_list = s_list + threadIdx.x;
    const bool live = gid < count;                          // no early return: the scan is wave-cooperative
    const double *x = in + (size_t)gid * unit_in_slots(op);
    double *y = out + (size_t)gid * unit_out_slots(op);
    const bool coop = op == U_HIT_WORLD || op == U_RAY_COLOR || op == U_HIT_WORLD_LDS || op == U_HIT_WORLD_CULL || op == U_HIT_WORLD_MFMA || op == U_HIT_WORLD_MFMA_CULL;
    if (!coop && !live) return;
    switch (op) {
        case U_HIT_SPHERE: {
            V3<T> c = ld3<T>(x), o = ld3<T>(x + 4), d = ld3<T>(x + 7);
            T r = (T)x[3], tmin = (T)x[10], tmax = (T)x[11];
            T hb, disc, root;
            sphere_disc<T>(scene.numerics, c.x, c.y, c.z, r * r, r, o, d, hb, disc);
            for (int k = 0; k < 9; ++k) y[k] = 0.0;
            if (sphere_root<T>(hb, disc, tmin, tmax, root)) {
                HitRec<T> rec;
                make_hitrec<T>(c, r, o, d, root, rec);
                y[0] = 1.0; y[1] = (double)rec.t; st3(y + 2, rec.p); st3(y + 5, rec.n); y[8] = rec.front ? 1.0 : 0.0;
            }
        } break;
        case U_REFLECT: st3(y, reflect(ld3<T>(x), ld3<T>(x + 3))); break;
        case U_REFRACT: st3(y, refract(ld3<T>(x), ld3<T>(x + 3), (T)x[6])); break;
        case U_REFLECTANCE: y[0] = (double)reflectance((T)x[0], (T)x[1]); break;
        case U_SCATTER: {
            Rng rng = {as_u64(x[0]), as_u64(x[1])};
            HitRec<T> rec;
            rec.t = (T)x[10]; rec.p = ld3<T>(x + 11); rec.n = ld3<T>(x + 14); rec.front = x[17] != 0.0;
            V3<T> nd, att;
            scatter<T>(rng, (int)x[2], ld3<T>(x + 3), (T)x[6], ld3<T>(x + 7), rec, nd, att);
            y[0] = as_f64(rng.x); y[1] = as_f64(rng.y);
            st3(y + 2, rec.p); st3(y + 5, nd); st3(y + 8, att);
        } break;
        case U_GET_RAY: {
            Rng rng = {as_u64(x[0]), as_u64(x[1])};
            V3<T> o, d;
            get_ray(rng, cam, (T)x[2], (T)x[3], o, d);
            y[0] = as_f64(rng.x); y[1] = as_f64(rng.y);
            st3(y + 2, o); st3(y + 5, d);
        } break;
        case U_SKYCOLOR: {
            C3 c = skycolor(ld3<T>(x));
            y[0] = c.r; y[1] = c.g; y[2] = c.b;
        } break;
        case U_RNG: {
            Rng rng = {as_u64(x[0]), as_u64(x[1])};
            for (int k = 0; k < 4; ++k) { T u; trand(rng, u); y[2 + k] = (double)u; }
            y[0] = as_f64(rng.x); y[1] = as_f64(rng.y);
        } break;
        case U_EXACT_MATH: {
            // (the waves of this launch see consecutive patterns in their lanes: in-range and out-of-range arguments mix, as in the kernel)
            double bad_s = 0, bad_r = 0, first_s = -1, first_r = -1;
            if constexpr (sizeof(T) == 4) {
                const unsigned p0 = (unsigned)x[0], n = (unsigned)x[1];
                for (unsigned k = 0; k < n; ++k) {
                    const unsigned u = p0 + k;
                    const float v = __uint_as_float(u);
                    const unsigned a = __float_as_uint(t_sqrt(v)), b = __float_as_uint(__builtin_sqrtf(v));
                    const unsigned c = __float_as_uint(t_rcp(v)), e = __float_as_uint(1.0f / v);
                    const bool nan_ab = (a & 0x7fffffffu) > 0x7f800000u && (b & 0x7fffffffu) > 0x7f800000u;      // (any NaN equals any NaN)
                    const bool nan_ce = (c & 0x7fffffffu) > 0x7f800000u && (e & 0x7fffffffu) > 0x7f800000u;
                    if (a != b && !nan_ab) { bad_s += 1; if (first_s < 0) first_s = (double)u; }
                    if (c != e && !nan_ce) { bad_r += 1; if (first_r < 0) first_r = (double)u; }
                }
            }
            y[0] = bad_s; y[1] = bad_r; y[2] = first_s; y[3] = first_r;
        } break;
        case U_NEAR_ZERO: {
            const V3<T> v = ld3<T>(x);
            y[0] = near_zero(v) ? 1.0 : 0.0;
            y[1] = (double)dot(v, v);
        } break;
        case U_HIT_WORLD: {
            V3<T> o = {0, 0, 0}, d = {0, 0, 1};
            T tmn = 0, tmx = 0;
            if (live) { o = ld3<T>(x); d = ld3<T>(x + 3); tmn = (T)x[6]; tmx = (T)x[7]; }
            T t_hit;
            int idx = hit_world<T, 64>(scene, scene.geom, o, d, tmn, tmx, t_hit, my_list);
            if (!live) return;
            for (int k = 0; k < 9; ++k) y[k] = 0.0;
            y[0] = (double)idx;
            if (idx >= 0) {
                auto g = scene.geom[idx];
                auto m0 = scene.mat0[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, o, d, t_hit, rec);
                y[1] = (double)rec.t; st3(y + 2, rec.p); st3(y + 5, rec.n); y[8] = rec.front ? 1.0 : 0.0;
            }
        } break;
        case U_HIT_WORLD_LDS: case U_HIT_WORLD_CULL: {
            V4 *lds_geom = reinterpret_cast<V4 *>(u_smem);
            unsigned short *lds_orig = reinterpret_cast<unsigned short *>(lds_geom + (op == U_HIT_WORLD_CULL ? cull_exact_count(cull) : 0));
            if (op == U_HIT_WORLD_CULL) stage_cull_scene<T>(cull, lds_geom, lds_orig); else stage_scene<T>(scene, lds_geom);
            __syncthreads();
            V3<T> o = {0, 0, 0}, d = {0, 0, 1};
            T tmn = 0, tmx = 0;
            if (live) { o = ld3<T>(x); d = ld3<T>(x + 3); tmn = (T)x[6]; tmx = (T)x[7]; }
            T t_hit;
            int idx;
            if (op == U_HIT_WORLD_CULL)
                idx = hit_world_cull<T, 64>(cull, (const V4 *)lds_geom, (const unsigned short *)lds_orig, o, d, tmn, tmx, t_hit, my_list);
            else
                idx = hit_world<T, 64>(scene, (const V4 *)lds_geom, o, d, tmn, tmx, t_hit, my_list);
            if (!live) return;
            for (int k = 0; k < 9; ++k) y[k] = 0.0;
            y[0] = (double)(idx >= 0 && op == U_HIT_WORLD_CULL ? (int)cull.orig[idx] : idx);   // index in the caller's list
            if (idx >= 0) {
                const V4 g = op == U_HIT_WORLD_CULL ? cull.exact[idx] : scene.geom[idx];
                const V4 m0 = op == U_HIT_WORLD_CULL ? cull.mat0[idx] : scene.mat0[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, o, d, t_hit, rec);
                y[1] = (double)rec.t; st3(y + 2, rec.p); st3(y + 5, rec.n); y[8] = rec.front ? 1.0 : 0.0;
            }
        } break;
        case U_HIT_WORLD_MFMA: {
            __shared__ __attribute__((aligned(8))) unsigned s_pairs[RTW_PAIR_CAP];
            __shared__ unsigned long long s_keys[64];
            __shared__ unsigned s_kidx[64];
            V4 *lds_geom = reinterpret_cast<V4 *>(u_smem);
            stage_scene<T>(scene, lds_geom);
            __syncthreads();
            V3<T> o = {0, 0, 0}, d = {0, 0, 1};
            if (live) { o = ld3<T>(x); d = ld3<T>(x + 3); }
            const T tmn = (T)in[6];                       // wave-uniform by contract (the trace kernel passes a constant)
            T t_hit;
            const WaveScratch ws = {s_pairs, s_keys, s_kidx};
            int idx = -1;
            if (scene.mf_ops) idx = hit_world_mfma<T>(scene, (const V4 *)lds_geom, o, d, live, tmn, t_hit, ws, threadIdx.x & 63u);
            if (!live) return;
            for (int k = 0; k < 9; ++k) y[k] = 0.0;
            y[0] = (double)idx;
            if (idx >= 0) {
                const V4 g = scene.geom[idx];
                const V4 m0 = scene.mat0[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, o, d, t_hit, rec);
                y[1] = (double)rec.t; st3(y + 2, rec.p); st3(y + 5, rec.n); y[8] = rec.front ? 1.0 : 0.0;
            }
        } break;
        case U_HIT_WORLD_MFMA_CULL: {
            __shared__ __attribute__((aligned(8))) unsigned c_pairs[RTW_PAIR_CAP];
            __shared__ unsigned long long c_keys[64];
            __shared__ unsigned c_kidx[64];
            V4 *lds_geom = reinterpret_cast<V4 *>(u_smem);
            unsigned short *lds_orig = reinterpret_cast<unsigned short *>(lds_geom + cull_exact_count(cull));
            stage_cull_scene<T>(cull, lds_geom, lds_orig);
            __syncthreads();
            V3<T> o = {0, 0, 0}, d = {0, 0, 1};
            if (live) { o = ld3<T>(x); d = ld3<T>(x + 3); }
            const T tmn = (T)in[6];
            T t_hit;
            const WaveScratch ws = {c_pairs, c_keys, c_kidx};
            const MfmaCull mc = mfma_cull_of(cull);
            int idx = -1;
            if (cull.mf_ops) idx = hit_world_mfma<T>(scene, (const V4 *)lds_geom, o, d, live, tmn, t_hit, ws, threadIdx.x & 63u, NoClock(), &mc, (const unsigned short *)lds_orig);
            if (!live) return;
            for (int k = 0; k < 9; ++k) y[k] = 0.0;
            y[0] = (double)(idx >= 0 ? (int)cull.orig[idx] : idx);      // index in the caller's list
            if (idx >= 0) {
                const V4 g = cull.exact[idx];
                const V4 m0 = cull.mat0[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, o, d, t_hit, rec);
                y[1] = (double)rec.t; st3(y + 2, rec.p); st3(y + 5, rec.n); y[8] = rec.front ? 1.0 : 0.0;
            }
        } break;
        case U_FX_SUM: {
            // the trace kernel's pixel accumulation: fx_from_double -> 128-bit adds -> fx_to_double
            unsigned long long lo = 0, hi = 0, bad = 0;
            for (int k = 0; k < 8; ++k) {
                unsigned long long l, h;
                if (fx_from_double(x[k], l, h)) { const unsigned long long old = lo; lo += l; hi += h + (lo < old ? 1ull : 0ull); }
                else bad += 1;
            }
            y[0] = bad ? __builtin_nan("") : fx_to_double(lo, hi);
            y[1] = (double)bad;
        } break;
        case U_RAY_COLOR: {
            // src/ray_color.jl:14-38 as the iterative front-to-back loop of the trace kernel
            Rng rng = {1, 2};
            V3<T> o = {0, 0, 0}, d = {0, 0, 1};
            int depth = 0;
            if (live) { rng = {as_u64(x[0]), as_u64(x[1])}; o = ld3<T>(x + 2); d = ld3<T>(x + 5); depth = (int)x[8]; }
            double tr = 1, tg = 1, tb = 1, cr = 0, cg = 0, cb = 0;
            unsigned segs = 0;
            bool active = depth > 0;
            while (__any(active)) {                   // the scan is wave-cooperative: all lanes enter it
                T t_hit;
                int idx = hit_world<T, 64>(scene, scene.geom, o, d, (T)1e-4, (T)__builtin_huge_val(), t_hit, my_list);
                if (!active) continue;
                segs++;
                if (idx < 0) {
                    C3 sky = skycolor(d);
                    cr = tr * sky.r; cg = tg * sky.g; cb = tb * sky.b;
                    active = false;
                    continue;
                }
                auto g = scene.geom[idx];
                auto m0 = scene.mat0[idx];
                auto m1 = scene.mat1[idx];
                HitRec<T> rec;
                make_hitrec<T>({g.x, g.y, g.z}, m0.x, o, d, t_hit, rec);
                V3<T> nd, att;
                scatter<T>(rng, (int)m0.z, {m1.x, m1.y, m1.z}, m0.y, d, rec, nd, att);
                tr = tr * (double)att.x; tg = tg * (double)att.y; tb = tb * (double)att.z;
                o = rec.p; d = nd;
                depth -= 1;
                if (depth <= 0) active = false;
            }
            if (!live) return;
            y[0] = as_f64(rng.x); y[1] = as_f64(rng.y);
            y[2] = cr; y[3] = cg; y[4] = cb; y[5] = (double)segs;
        } break;
    }
}

}  // namespace rtw
