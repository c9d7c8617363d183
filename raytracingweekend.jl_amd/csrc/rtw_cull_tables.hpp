// rtw_cull_tables.hpp -- the tables of the group cull's block vote (rtw::CullGrid, rtw_scan_mfma.hpp), built on the host.
// Plain C++ (no HIP): rtw_scene.hip uploads what this builds, tests/cull_tables_check.cpp checks it on the CPU against the exact
// box-overlap test it stands for.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>

#ifndef RTW_CULL_BINS
#define RTW_CULL_BINS 64
#endif
#define RTW_CULL_INLANE_MAX 8
#ifndef RTW_CULL_EDGE_SLACK
#define RTW_CULL_EDGE_SLACK 1e-3      // bins by which the edges are moved outwards (the CPU check is also compiled with a NEGATIVE slack: it must then fail)
#endif
#define RTW_CULL_TAB_WORDS (6 * RTW_CULL_BINS + 4 + RTW_CULL_INLANE_MAX)

namespace rtwh {

struct CullTables {
    float inv[3], off[3];            // bin of a coordinate p on axis k: floor(p inv[k] + off[k]) clamped to 0 .. RTW_CULL_BINS - 1
    std::vector<unsigned> words;     // per group of 32 blocks RTW_CULL_TAB_WORDS words (layout: rtw_scan_mfma.hpp)
};

// bx: 8 floats per block (lo.xyz, pad, hi.xyz, pad), nb blocks -- a BIG block carries +-INFINITY, a dead one the point 1e15.
// glo / ghi: the union of the live finite boxes (none: glo > ghi).  RTW_CULL_BINS bins per axis over that box; per group of 32 blocks
// and axis the set of blocks with lo_b <= upper edge of the bin / hi_b >= lower edge of the bin.  The edges are moved outwards by 1e-3 of
// a bin: the device's binary32 evaluation of p inv + off (one fma: the product is exact, the sum is rounded once, |result| < 64) is off
// by < 1e-5 of a bin, and inv / off are used here exactly as the device gets them (rounded to binary32 FIRST); the outer bins reach to
// infinity.  Dead blocks are in no set.
inline void build_cull_tables(const float *bx, int nb, const float glo[3], const float ghi[3], int n_inlane, const int *inlane, CullTables *out) {
    const int n_grp = (nb + 31) / 32 > 0 ? (nb + 31) / 32 : 1;
    out->words.assign((size_t)n_grp * RTW_CULL_TAB_WORDS, 0u);
    std::vector<unsigned> &tab = out->words;
    for (int k = 0; k < 3; ++k) {
        const double ext = ghi[k] > glo[k] ? (double)ghi[k] - (double)glo[k] : 0.0;
        const double inv = (double)(float)(ext > 0.0 ? RTW_CULL_BINS / ext : 0.0);            // (a flat or empty class: everything in bin 0)
        const double off = (double)(float)(ext > 0.0 && std::isfinite(inv) ? -(double)glo[k] * inv : 0.0);
        out->inv[k] = std::isfinite(inv) ? (float)inv : 0.0f; out->off[k] = std::isfinite(off) ? (float)off : 0.0f;
        const double inv_u = (double)out->inv[k], off_u = (double)out->off[k];
        for (int j = 0; j < RTW_CULL_BINS; ++j) {
            const double edge_hi = (j == RTW_CULL_BINS - 1 || inv_u == 0.0) ? INFINITY : ((double)(j + 1) + RTW_CULL_EDGE_SLACK - off_u) / inv_u;
            const double edge_lo = (j == 0 || inv_u == 0.0) ? -INFINITY : ((double)j - RTW_CULL_EDGE_SLACK - off_u) / inv_u;
            for (int b = 0; b < nb; ++b) {
                const float *q = &bx[(size_t)b * 8];
                if (std::isfinite(q[0]) && q[0] >= 1e15f) continue;                                // nothing alive
                unsigned *t = &tab[(size_t)(b / 32) * RTW_CULL_TAB_WORDS];
                if ((double)q[k] <= edge_hi) t[(k * 2 + 0) * RTW_CULL_BINS + j] |= 1u << (b % 32);
                if ((double)q[4 + k] >= edge_lo) t[(k * 2 + 1) * RTW_CULL_BINS + j] |= 1u << (b % 32);
            }
        }
    }
    for (int b = 0; b < nb; ++b) {
        const float *q = &bx[(size_t)b * 8];
        unsigned *t = &tab[(size_t)(b / 32) * RTW_CULL_TAB_WORDS + 6 * RTW_CULL_BINS];
        if (!std::isfinite(q[0])) { t[0] |= 1u << (b % 32); t[1] |= 1u << (b % 32); }          // BIG: every ray; live
        else if (q[0] < 1e15f) t[1] |= 1u << (b % 32);                                          // live: what a ray without the filter takes
    }
    for (int k = 0; k < n_inlane && k < RTW_CULL_INLANE_MAX; ++k) tab[6 * RTW_CULL_BINS + 4 + k] = (unsigned)inlane[k];      // the in-lane list (first group's tables)
}

// the device's look-up (hit_world_mfma's block_sets) for one ray with bounds [lo3, hi3], restated for the host: the set of blocks of `group`
inline unsigned cull_tables_lookup(const CullTables &c, int group, const float lo3[3], const float hi3[3]) {
    auto bin = [](float u) -> unsigned {                         // v_med3_f32(u, 0, BINS - 0.5) then v_cvt_u32_f32 (NaN -> 0)
        if (!(u == u)) return 0u;
        const float m = u < 0.0f ? 0.0f : (u > (float)RTW_CULL_BINS - 0.5f ? (float)RTW_CULL_BINS - 0.5f : u);
        return (unsigned)m;
    };
    const unsigned *t = &c.words[(size_t)group * RTW_CULL_TAB_WORDS];
    unsigned set = ~0u;
    for (int k = 0; k < 3; ++k) {
        const unsigned bl = bin(std::fmaf(lo3[k], c.inv[k], c.off[k])), bh = bin(std::fmaf(hi3[k], c.inv[k], c.off[k]));
        set &= t[(k * 2 + 0) * RTW_CULL_BINS + bh] & t[(k * 2 + 1) * RTW_CULL_BINS + bl];
    }
    return set;
}

}  // namespace rtwh
