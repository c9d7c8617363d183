// cull_tables_check.cpp -- CPU check of the group cull's vote tables (raytracingweekend.jl_amd/csrc/rtw_cull_tables.hpp, plain C++):
// for random block boxes and random ray bounds the looked-up set of blocks must contain every block whose box overlaps the bounds
// (the exact test it replaces), never a dead block, and the flag words must name the BIG / live blocks.  Prints one summary line;
// exit code 1 on the first violation.  Compiled and run by tests/test_host_abi.py (no GPU, no HIP).
#include <cstdio>
#include <cstdlib>
#include <random>
#include "../raytracingweekend.jl_amd/csrc/rtw_cull_tables.hpp"

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 400;
    std::mt19937_64 gen(12345);
    auto uni = [&](double a, double b) { return a + (b - a) * (double)(gen() >> 11) * (1.0 / 9007199254740992.0); };
    long long pairs = 0, overlaps = 0, in_set = 0;
    for (int round = 0; round < rounds; ++round) {
        const int nb = 1 + (int)(gen() % 70);                                        // up to three groups of 32 blocks
        const double scale = std::pow(10.0, uni(-3, 6)), far = (round % 3 == 0) ? scale * std::pow(10.0, uni(0, 4)) : 0.0;
        const double centre[3] = {uni(-1, 1) * far, uni(-1, 1) * far, uni(-1, 1) * far};
        const bool flat = round % 7 == 0;                                            // one axis of (almost) no extent
        std::vector<float> bx((size_t)(nb + 1) * 8, 0.0f);
        float glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int b = 0; b < nb; ++b) {
            float *q = &bx[(size_t)b * 8];
            const int kind = (int)(gen() % 12);                                      // 0: BIG, 1: dead, else a box
            for (int a = 0; a < 3; ++a) {
                if (kind == 0) { q[a] = -INFINITY; q[4 + a] = INFINITY; }
                else if (kind == 1) { q[a] = 1e15f; q[4 + a] = 1e15f; }
                else {
                    const double c = centre[a] + uni(-1, 1) * scale * ((flat && a == 1) ? 0.0 : 1.0), w = uni(0, 0.3) * scale * ((flat && a == 1) ? 1e-9 : 1.0);
                    q[a] = (float)(c - w); q[4 + a] = (float)(c + w);
                    if (round % 5 == 0) { q[a] = std::round(q[a] / (float)scale * 8.0f) * (float)scale / 8.0f; q[4 + a] = std::fmax(q[a], std::round(q[4 + a] / (float)scale * 8.0f) * (float)scale / 8.0f); }   // boxes on a lattice: bin edges
                    glo[a] = std::fmin(glo[a], q[a]); ghi[a] = std::fmax(ghi[a], q[4 + a]);
                }
            }
        }
        rtwh::CullTables ct;
        const int inlane[2] = {7, 3};
        rtwh::build_cull_tables(bx.data(), nb, glo, ghi, 2, inlane, &ct);
        if (ct.words[6 * RTW_CULL_BINS + 4] != 7u || ct.words[6 * RTW_CULL_BINS + 5] != 3u) { printf("in-lane list\n"); return 1; }
        const int n_grp = (nb + 31) / 32;
        for (int g = 0; g < n_grp; ++g) {                                            // the flag words
            unsigned big = 0, live = 0;
            for (int b = 32 * g; b < nb && b < 32 * g + 32; ++b) {
                const float *q = &bx[(size_t)b * 8];
                if (!std::isfinite(q[0])) { big |= 1u << (b % 32); live |= 1u << (b % 32); } else if (q[0] < 1e15f) live |= 1u << (b % 32);
            }
            const unsigned *t = &ct.words[(size_t)g * RTW_CULL_TAB_WORDS + 6 * RTW_CULL_BINS];
            if (t[0] != big || t[1] != live || t[2] != 0u || t[3] != 0u) { printf("flag words of group %d\n", g); return 1; }
        }
        for (int ray = 0; ray < 600; ++ray) {
            float lo3[3], hi3[3];
            const int mode = (int)(gen() % 8);
            for (int a = 0; a < 3; ++a) {
                double c = centre[a] + uni(-1.3, 1.3) * scale, w = uni(0, mode == 0 ? 2.0 : 0.2) * scale;
                if (mode == 1) c = centre[a] + uni(-50, 50) * scale;                 // far outside the class's box
                lo3[a] = (float)(c - w); hi3[a] = (float)(c + w);
                if (mode == 2 && a == (int)(gen() % 3)) { const float *q = &bx[(size_t)(gen() % nb) * 8]; if (std::isfinite(q[a]) && q[a] < 1e15f) { hi3[a] = q[a]; lo3[a] = std::fmin(lo3[a], hi3[a]); } }   // touching a box's face exactly
                if (mode == 3) { lo3[a] = -INFINITY; hi3[a] = INFINITY; }
                if (mode == 4 && a == 0) { lo3[a] = 3.0e38f; hi3[a] = -3.0e38f; }      // (what a ray that misses the class used to carry)
            }
            for (int g = 0; g < n_grp; ++g) {
                const unsigned set = rtwh::cull_tables_lookup(ct, g, lo3, hi3);
                for (int b = 32 * g; b < nb && b < 32 * g + 32; ++b) {
                    const float *q = &bx[(size_t)b * 8];
                    const bool dead = std::isfinite(q[0]) && q[0] >= 1e15f;
                    bool overlap = !dead;
                    for (int a = 0; a < 3; ++a) overlap = overlap && q[a] <= hi3[a] && q[4 + a] >= lo3[a];
                    const bool have = (set >> (b % 32)) & 1u;
                    ++pairs; overlaps += overlap; in_set += have;
                    if (overlap && !have) { printf("round %d ray %d: block %d overlaps the bounds but is not in the set\n", round, ray, b); return 1; }
                    if (dead && have) { printf("round %d ray %d: dead block %d in the set\n", round, ray, b); return 1; }
                }
                if (nb % 32 && g == n_grp - 1 && (set >> (nb % 32)) != 0u) { printf("bits beyond the last block\n"); return 1; }
            }
        }
    }
    printf("cull tables: %d scenes, %lld (bounds, block) pairs, %lld overlap, %lld in the looked-up sets (%.3f x), no block lost\n", rounds, pairs, overlaps, in_set,
           overlaps ? (double)in_set / (double)overlaps : 0.0);
    return 0;
}
