// rtw_probes.hpp -- time / instruction-count probes of the trace kernel (tools/gpu_probe_phases.sh builds library VARIANTS with one of
// the switches below on the GPU box; profiles/r0N_probe_phases.txt).  Every macro expands to NOTHING in the product build: none of
// this is in librtw_hip.so.  A probe's body refers to the local variables of the one site that uses it.
//   -DRTW_DUP_OPERANDS       the ray-operand build of hit_world_mfma (features, f16 splits, word assembly, lane exchange) a second time
//   -DRTW_DUP_MFMA[=k]       every MFMA pair k more times, same result      -DRTW_PROBE_NO_MFMA   a constant "no candidate" instead (WRONG image)
//   -DRTW_DUP_EVAL           the sign collection twice                      -DRTW_DUP_EXTRACT     the extraction loop twice
//   -DRTW_DUP_RESOLVE_PAIRS  the final resolve twice (idempotent)           -DRTW_DUP_REJECT      the rejection loop twice (on a copy of the generator)
//   -DRTW_PROBE_REJ_CAP=n    the rejection loop stops after n trials (WRONG image)
//   -DRTW_PROBE_NO_ACCUM     nothing is added to the pixels (WRONG image)   -DRTW_PROBE_FASTDIV   approximate 1/x, 1/sqrt(x) in the shading (WRONG image)
#pragma once

#ifdef RTW_DUP_OPERANDS
#define RTW_PROBE_OPERANDS_TWICE()                                                                                                          \
    {                                                                                                                                       \
        float ox2 = ox, oy2 = oy, oz2 = oz, dx2_ = dx, dy2_ = dy, dz2_ = dz;                                                                \
        __asm__ volatile("" : "+v"(ox2), "+v"(oy2), "+v"(oz2), "+v"(dx2_), "+v"(dy2_), "+v"(dz2_));                                         \
        const float q_ = __builtin_fmaf(oz2, dz2_, __builtin_fmaf(oy2, dy2_, ox2 * dx2_));                                                  \
        const float oo_ = __builtin_fmaf(oz2, oz2, __builtin_fmaf(oy2, oy2, ox2 * ox2));                                                    \
        const float o1_ = (__builtin_fabsf(ox2) + __builtin_fabsf(oy2)) + __builtin_fabsf(oz2);                                             \
        const float oop_ = __builtin_fmaf(oo_, w.mf_oo_keep, -(w.mf_o1_coef * o1_));                                                        \
        const float tq_ = w.mf_sigma2 * __builtin_fmaf(q_, q_, -oop_);                                                                      \
        const float fp_[3] = {__builtin_fmaf(-q_, dx2_, ox2) * zs2, __builtin_fmaf(-q_, dy2_, oy2) * zs2, __builtin_fmaf(-q_, dz2_, oz2) * zs2}; \
        const float ax_ = dx2_ * z2, ay_ = dy2_ * z2, az_ = dz2_ * z2;                                                                      \
        const float fq_[6] = {ax_ * dx2_, ay_ * dy2_, az_ * dz2_, (ax_ + ax_) * dy2_, (ax_ + ax_) * dz2_, (ay_ + ay_) * dz2_};              \
        unsigned sq_[6], sp_[3];                                                                                                            \
        for (int k = 0; k < 6; ++k) sq_[k] = split_f16(fq_[k]);                                                                             \
        for (int k = 0; k < 3; ++k) sp_[k] = split_f16(fp_[k]);                                                                             \
        const float tx_ = ok ? tq_ : tx;                                                                                                    \
        const _Float16 t1_ = (_Float16)(tx_ * (1.0f / 32768.0f));                                                                           \
        const unsigned x23_ = split_f16((tx_ - 32768.0f * (float)t1_) * (1.0f / 16.0f));                                                    \
        const unsigned a0[8] = {dup(sq_[0]), cat(sq_[0], sq_[1]), sq_[1], dup(sq_[2]), sq_[5], dup(sp_[0]), cat(sp_[0], sp_[1]), sp_[1]};   \
        const unsigned a1[8] = {cat(sq_[2], sq_[3]), sq_[3], dup(sq_[4]), cat(sq_[4], sq_[5]), dup(sp_[2]), cat(sp_[2], sb),                \
                                ss | ((unsigned)__builtin_bit_cast(unsigned short, t1_) << 16), x23_};                                      \
        for (int k = 0; k < 8; ++k) {                                                                                                       \
            const auto sw2 = __builtin_amdgcn_permlane32_swap(a0[k], a1[k], false, false);                                                  \
            __asm__ volatile("" :: "v"(sw2[0]), "v"(sw2[1]));                                                                               \
        }                                                                                                                                   \
    }
#else
#define RTW_PROBE_OPERANDS_TWICE()
#endif

// (the operands stay live: the loads and the ray operands are still made)
#ifdef RTW_PROBE_NO_MFMA
#define RTW_PROBE_FILTER_PAIR_REPLACED(a1, a2, b1, b2)                                                                                      \
    {                                                                                                                                       \
        rtw_f16v Wn = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};                                                     \
        const uint4 u1 = __builtin_bit_cast(uint4, b1), u2 = __builtin_bit_cast(uint4, b2);                                                 \
        __asm__ volatile("" :: "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w), "v"(a2.x), "v"(a2.y), "v"(a2.z), "v"(a2.w));                     \
        __asm__ volatile("" :: "v"(u1.x), "v"(u1.y), "v"(u1.z), "v"(u1.w), "v"(u2.x), "v"(u2.y), "v"(u2.z), "v"(u2.w));                     \
        __asm__ volatile("" : "+v"(Wn));                                                                                                    \
        return Wn;                                                                                                                          \
    }
#else
#define RTW_PROBE_FILTER_PAIR_REPLACED(a1, a2, b1, b2)
#endif

// (the repeated pair takes its sphere operand through an opaque copy and starts from zero again: left as the same expression it is
//  merged with the first pair -- rounds 3 and 4 measured 16 register copies, not MFMAs)
#ifdef RTW_DUP_MFMA
#define RTW_PROBE_FILTER_PAIR_AGAIN(Wp, a1, a2, b1, b2)                                                                                     \
    for (int rep = 0; rep < (RTW_DUP_MFMA + 0 > 0 ? RTW_DUP_MFMA + 0 : 1); ++rep) {                                                         \
        uint4 a1c = a1;                                                                                                                     \
        __asm__ volatile("" : "+v"(a1c.x), "+v"(a1c.y), "+v"(a1c.z), "+v"(a1c.w));                                                          \
        __asm__ volatile("" : "+v"(Wp));                                                                                                    \
        Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a1c), b1, zero, 0, 0, 0);                                    \
        Wp = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(rtw_h8, a2), b2, Wp, 0, 0, 0);                                       \
    }
#else
#define RTW_PROBE_FILTER_PAIR_AGAIN(Wp, a1, a2, b1, b2)
#endif

#ifdef RTW_DUP_EVAL      // (no CSE with the real one)
#define RTW_PROBE_EVAL_TWICE(Wv) { unsigned keep = mask; eval(Wv); __asm__ volatile("" :: "v"(mask)); mask = keep; __asm__ volatile("" : "+v"(Wv)); }
#else
#define RTW_PROBE_EVAL_TWICE(Wv)
#endif

#ifdef RTW_DUP_EXTRACT   // (the first run writes the same entry)
#define RTW_PROBE_EXTRACT_TWICE()                                                                                                           \
    {                                                                                                                                       \
        unsigned m2 = m; __asm__ volatile("" : "+v"(m2));                                                                                   \
        const unsigned long long act2 = __ballot(m2 != 0u);                                                                                 \
        if (m2 != 0u) reinterpret_cast<uint2 *>(ws.pairs)[__builtin_amdgcn_mbcnt_hi((unsigned)(act2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act2, total))] = uint2{m2, lane_const + (unsigned)cur * 32u}; \
        __builtin_amdgcn_wave_barrier();                                                                                                    \
    }
#else
#define RTW_PROBE_EXTRACT_TWICE()
#endif

#ifdef RTW_DUP_RESOLVE_PAIRS
#define RTW_PROBE_RESOLVE_TWICE() resolve_pairs<T>(w.numerics, src, rad, o, d, tmin, ws, total, lane, orig)
#else
#define RTW_PROBE_RESOLVE_TWICE()
#endif

// ---- the lane loop (rtw_kernels.hpp) ----
#ifdef RTW_DUP_REJECT
#define RTW_PROBE_REJECT_TWICE()                                                                                                            \
    {                                                                                                                                       \
        Rng r2 = rng; V3<T> q2 = {0, 0, 0}; T l2 = 0; bool p2 = pending;                                                                    \
        while (__any(p2)) { if (p2) { l2 = reject_trial<T>(r2, ball, q2); p2 = !(l2 <= T(1)); } }                                           \
        __asm__ volatile("" :: "v"(q2.x), "v"(q2.y), "v"(q2.z), "v"(l2), "v"((unsigned)r2.x), "v"((unsigned)r2.y));                         \
    }
#else
#define RTW_PROBE_REJECT_TWICE()
#endif

#ifdef RTW_PROBE_NO_ACCUM
#define RTW_PROBE_MISS(expr) false
#else
#define RTW_PROBE_MISS(expr) (expr)
#endif

// ---- the shading's divisions and square roots (rtw_path.hpp) ----
#ifdef RTW_PROBE_FASTDIV
#define RTW_DIV(a, b) ((a) * ::rtw::probe_rcp(b))
#define RTW_RSQRT(x) ::rtw::probe_rsq(x)
#else
#define RTW_DIV(a, b) ((a) / (b))                    // IEEE division
#define RTW_RSQRT(x) t_rcp(t_sqrt(x))                // StaticArrays: inv(norm(v)), two correctly rounded operations
#endif
