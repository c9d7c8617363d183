// rtw_scene.hip -- scene upload (rtw_scene_upload_f32/_f64 and the cached uploads of the host-buffer entry points): the SoA rows the
// kernels read, the kd split and boxes of the opt-in group-cull layout, and the sphere-side operands of the matrix-pipe filter
// (hit_world_mfma, rtw_device.hpp) with their error-margin constants.  Host code only; -ffp-contract=off like the kernels.
#include "rtw_scene_view.hpp"
#include "rtw_cull_tables.hpp"

namespace rtwh {

// kd median split of the small class into clusters of <= RTW_CULL_GS spheres (ids = indices into the caller's list)
template <typename SceneT>
void kd_split(const SceneT *s, std::vector<int> &ids, int lo, int hi, std::vector<std::vector<int>> &groups) {
    const int cnt = hi - lo;
    if (cnt <= RTW_CULL_GS) {
        if (cnt > 0) groups.emplace_back(ids.begin() + lo, ids.begin() + hi);
        return;
    }
    double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
    for (int k = lo; k < hi; ++k) {
        const double c[3] = {(double)s->cx[ids[k]], (double)s->cy[ids[k]], (double)s->cz[ids[k]]};
        for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], c[a]); mx[a] = std::max(mx[a], c[a]); }
    }
    int ax = 0;
    for (int a = 1; a < 3; ++a) if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
    // left part: whole clusters -- and whole PAIRS of clusters while more than one pair is left, so that the blocks of 32 of
    // the matrix-pipe cull (two consecutive clusters) are always siblings of this tree
    const int unit = cnt > 2 * RTW_CULL_GS ? 2 * RTW_CULL_GS : RTW_CULL_GS;
    int half = ((cnt / 2 + unit - 1) / unit) * unit;
    if (half >= cnt) half = cnt - 1;
    auto key = [&](int i) { return ax == 0 ? (double)s->cx[i] : ax == 1 ? (double)s->cy[i] : (double)s->cz[i]; };
    std::nth_element(ids.begin() + lo, ids.begin() + lo + half, ids.begin() + hi, [&](int a, int b) { return key(a) < key(b); });
    kd_split(s, ids, lo, lo + half, groups);
    kd_split(s, ids, lo + half, hi, groups);
}

// the cold per-sphere rows (rtw_device.hpp "device scene"): mat0 = (r, param, kind, 1 / ir), mat1 = albedo -- or, for a Dielectric,
// Schlick's r0 for a front and a back face (dielectric_constants: the reference's own expressions evaluated once, in T)
template <typename T, typename V4, typename SceneT>
void material_rows(const SceneT *s, int i, V4 &m0, V4 &m1) {
    m0 = V4{s->r[i], s->param[i], (T)s->kind[i], (T)0};
    m1 = V4{s->ar[i], s->ag[i], s->ab[i], (T)0};
    if (s->kind[i] == rtw::DIELECTRIC) {
        const rtw::DielConst<T> c = rtw::dielectric_constants<T>(s->param[i]);
        m0.w = c.inv_ir;
        m1 = V4{c.r0_front, c.r0_back, (T)0, (T)0};
    }
}

// cluster-major arrays for the opt-in group-cull scan (rtw_device.hpp, "opt-in accelerated scan")
template <typename T, typename V4>
int build_mfma_operands(const std::vector<V4> &geom, int n, rtw_scene_dev *h, void **ops_out, int *blocks_out, int n_skip = 0, const int *skip = nullptr);

template <typename T, typename SceneT>
int build_cull(const SceneT *s, rtw_scene_dev *h) {
    using V4 = typename rtw::Vec4<T>::type;
    const int n = s->n;
    constexpr int pair = 2 * RTW_CULL_BG, GS = RTW_CULL_GS;
    // BIG class: |r| > 4 x lower-median |r| (the ground sphere, the unit spheres of scene_random_spheres)
    std::vector<int> small_ids, big_ids;
    if (n > 0) {
        std::vector<double> rr(n);
        for (int i = 0; i < n; ++i) rr[i] = std::fabs((double)s->r[i]);
        std::vector<double> tmp(rr);
        std::nth_element(tmp.begin(), tmp.begin() + (n - 1) / 2, tmp.end());
        const double thr = 4.0 * tmp[(n - 1) / 2];
        for (int i = 0; i < n; ++i) (rr[i] > thr ? big_ids : small_ids).push_back(i);
    }
    std::vector<std::vector<int>> groups;
    kd_split(s, small_ids, 0, (int)small_ids.size(), groups);
    const int ng = (int)groups.size();
    const int ng_pad = ((ng + pair - 1) / pair) * pair;
    const int n_big = (int)big_ids.size();
    const int n_exact = ng_pad * GS + ((n_big + 31) / 32) * 32;          // whole blocks of 32 (the matrix-pipe scan may list any slot of a block)
    if (n_exact >= 65536) return fail(-5, "too many spheres (%d) for the group-cull layout", n);
    std::vector<T> box((size_t)(ng_pad + RTW_CULL_BG) * 8);            // + one prefetch group
    std::vector<V4> exact(std::max(n_exact, 1)), mat0(std::max(n_exact, 1)), mat1(std::max(n_exact, 1));
    std::vector<unsigned short> orig(std::max(n_exact, 1), 0);
    // dead cluster: far away and empty; dead sphere: r^2 = -1e30 (never a candidate)
    for (auto &b : box) b = (T)1e15;                                    // dead cluster: a point far away
    for (int k = 0; k < n_exact; ++k) { exact[k] = V4{(T)0, (T)0, (T)0, (T)-1e30}; mat0[k] = V4{(T)1, (T)0, (T)0, (T)0}; mat1[k] = V4{(T)0, (T)0, (T)0, (T)0}; }
    auto put = [&](int k, int i) {
        exact[k] = V4{s->cx[i], s->cy[i], s->cz[i], s->r[i] * s->r[i]};
        material_rows<T, V4>(s, i, mat0[k], mat1[k]);
        orig[k] = (unsigned short)i;
    };
    double cs[3] = {0, 0, 0}, rs = 0;
    long nsm = 0;
    for (auto &g : groups) for (int i : g) { cs[0] += s->cx[i]; cs[1] += s->cy[i]; cs[2] += s->cz[i]; ++nsm; }
    if (nsm) { cs[0] /= nsm; cs[1] /= nsm; cs[2] /= nsm; }
    for (int gi = 0; gi < ng; ++gi) {
        const auto &g = groups[gi];
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (int j = 0; j < (int)g.size(); ++j) {
            const int i = g[j];
            const double c[3] = {(double)s->cx[i], (double)s->cy[i], (double)s->cz[i]}, ar = std::fabs((double)s->r[i]);
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], c[a] - ar); hi[a] = std::max(hi[a], c[a] + ar); }
            put(gi * GS + j, i);
            const double ex = c[0] - cs[0], ey = c[1] - cs[1], ez = c[2] - cs[2];
            rs = std::max(rs, std::sqrt(ex * ex + ey * ey + ez * ez) + ar);
        }
        for (int a = 0; a < 3; ++a) {                                    // round outwards
            T l = (T)lo[a], u2 = (T)hi[a];
            if ((double)l > lo[a]) l = std::nextafter(l, (T)-INFINITY);
            if ((double)u2 < hi[a]) u2 = std::nextafter(u2, (T)INFINITY);
            box[(size_t)gi * 8 + a] = l; box[(size_t)gi * 8 + 4 + a] = u2;
        }
        box[(size_t)gi * 8 + 3] = box[(size_t)gi * 8 + 7] = (T)0;
    }
    for (int k = 0; k < n_big; ++k) put(ng_pad * GS + k, big_ids[k]);
    h->c_groups_pad = ng_pad; h->c_big = n_big;
    h->c_cs[0] = (double)(T)cs[0]; h->c_cs[1] = (double)(T)cs[1]; h->c_cs[2] = (double)(T)cs[2];
    h->c_rs = rs * (1.0 + 1e-5) + 1e-3 * (std::fabs(cs[0]) + std::fabs(cs[1]) + std::fabs(cs[2])) * (sizeof(T) == 4 ? 1e-4 : 1e-12);
    const size_t bb = sizeof(T) * box.size(), eb = sizeof(V4) * exact.size(), ob = sizeof(unsigned short) * orig.size();
    HIP_TRY(hipMalloc(&h->c_bound, bb));
    HIP_TRY(hipMalloc(&h->c_exact, eb));
    HIP_TRY(hipMalloc(&h->c_mat0, eb));
    HIP_TRY(hipMalloc(&h->c_mat1, eb));
    HIP_TRY(hipMalloc((void **)&h->c_orig, ob));
    HIP_TRY(hipMemcpy(h->c_bound, box.data(), bb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->c_exact, exact.data(), eb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->c_mat0, mat0.data(), eb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->c_mat1, mat1.data(), eb, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->c_orig, orig.data(), ob, hipMemcpyHostToDevice));
    // Group cull on the matrix pipe (hit_world_mfma<.., CULLED>): the operands in this cluster-major order and one
    // box per block of 32 = two clusters (dead clusters left out; the BIG class: everything).  Boxes are binary32,
    // rounded outwards, for both precisions -- the slab test runs in binary32 with the Float32 margin.
    if (h->mf_ops && n_exact > 0) {
        h->c_n_inlane = 0;
        for (int k = 0; k < RTW_CULL_INLANE_MAX; ++k) h->c_inlane[k] = -1;
        for (int k = 0; k < h->n_huge; ++k) {                     // the huge spheres (tested in-lane) in this order
            int at = -1;
            for (int dI = 0; dI < n_exact; ++dI)
                if ((double)exact[dI].w > -1e29 && (int)orig[dI] == h->huge[k]) { at = dI; break; }
            if (at < 0) return fail(-9, "internal: huge sphere %d not found in the cull layout", h->huge[k]);
            h->c_inlane[h->c_n_inlane++] = at;
        }
        // The BIG class is a block of its own that every half wave visits.  When all of it fits the in-lane list (the reference's scenes: the
        // ground and three spheres of radius 1), every lane tests those spheres by itself -- the exact test, the same keys -- and the block
        // is dead: 4 MFMAs, their sign collection and a pass-2 batch per scan less.  Scheduling only.
        {
            std::vector<int> rest;
            for (int k = 0; k < n_big; ++k) {
                const int dI = ng_pad * GS + k;
                if (!((double)exact[dI].w > -1e29)) continue;
                bool have = false;
                for (int j = 0; j < h->c_n_inlane; ++j) have = have || h->c_inlane[j] == dI;
                if (!have) rest.push_back(dI);
            }
            // (Float32 only -- measured 293.1 -> 288.9 ms at configs[2]; the binary64 tests cost more than the block: 93.3 -> 93.9 ms at the published configuration)
            static const bool inlane_f64 = aid_flag("RTW_INLANE_F64");      // (A/B aid: the in-lane class for Float64 too -- with the discriminant-only form of round 6)
            if ((sizeof(T) == 4 || inlane_f64) && h->c_n_inlane + (int)rest.size() <= RTW_CULL_INLANE_MAX)
                for (int dI : rest) h->c_inlane[h->c_n_inlane++] = dI;
        }
        auto in_lane = [&](int dI) { for (int j = 0; j < h->c_n_inlane; ++j) if (h->c_inlane[j] == dI) return true; return false; };
        if (int rc = build_mfma_operands<T>(exact, n_exact, h, &h->c_mf_ops, &h->c_mf_blocks, h->c_n_inlane, h->c_inlane)) return rc;
        const int nb = h->c_mf_blocks;
        std::vector<float> bx((size_t)(nb + 1) * 8, 0.0f);
        for (int b = 0; b <= nb; ++b) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool all = false, any = false;
            for (int c = 2 * b; c < 2 * b + 2; ++c) {
                if (c >= ng_pad) {                                                            // BIG class (device indices >= ng_pad * GS): any row left for the filter?
                    if (b < nb) for (int dI = c * GS; dI < std::min((c + 1) * GS, n_exact); ++dI) if ((double)exact[dI].w > -1e29 && !in_lane(dI)) all = true;
                    continue;
                }
                if (c >= ng) continue;                                                        // dead cluster
                any = true;
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)box[(size_t)c * 8 + a]); hi[a] = std::max(hi[a], (double)box[(size_t)c * 8 + 4 + a]); }
            }
            float *q = &bx[(size_t)b * 8];
            for (int a = 0; a < 3; ++a) {
                if (all) { q[a] = -INFINITY; q[4 + a] = INFINITY; }                          // never skipped (the vote of hit_world_mfma)
                else if (!any) { q[a] = 1e15f; q[4 + a] = 1e15f; }                            // nothing alive: a point far away
                else {
                    float l = (float)lo[a], u2 = (float)hi[a];
                    if ((double)l > lo[a]) l = std::nextafter(l, -INFINITY);
                    if ((double)u2 < hi[a]) u2 = std::nextafter(u2, INFINITY);
                    q[a] = l; q[4 + a] = u2;
                }
            }
        }
        // the box of the whole small class: the union of the live, finite block boxes (none: lo > hi, every ray misses it)
        for (int a = 0; a < 3; ++a) { h->c_glo[a] = INFINITY; h->c_ghi[a] = -INFINITY; }
        for (int b = 0; b < nb; ++b) {
            const float *q = &bx[(size_t)b * 8];
            if (!std::isfinite(q[0]) || q[0] >= 1e15f) continue;                                   // BIG class / nothing alive
            for (int a = 0; a < 3; ++a) { h->c_glo[a] = std::min(h->c_glo[a], q[a]); h->c_ghi[a] = std::max(h->c_ghi[a], q[4 + a]); }
        }
        // The tables of the block vote (rtw::CullGrid): rtw_cull_tables.hpp builds them (plain C++, checked on the CPU by tests/cull_tables_check.cpp)
        CullTables ct;
        build_cull_tables(bx.data(), nb, h->c_glo, h->c_ghi, h->c_n_inlane, h->c_inlane, &ct);
        for (int k = 0; k < 3; ++k) { h->c_grid[k] = ct.inv[k]; h->c_grid[3 + k] = ct.off[k]; }
        std::vector<float> cells(ct.words.size(), 0.0f);
        memcpy(cells.data(), ct.words.data(), ct.words.size() * sizeof(unsigned));
        bx.insert(bx.end(), cells.begin(), cells.end());
        HIP_TRY(hipMalloc(&h->c_mf_box, bx.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->c_mf_box, bx.data(), bx.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

// Operands of pass 1 on the matrix pipe (rtw_device.hpp, hit_world_mfma): per block of 32 spheres the two A operands of the
// chained v_mfma_f32_32x32x16_f16 pair -- the sphere side of the K = 32 contraction
//     [cx^2 cy^2 cz^2 cxcy cxcz cycz] s^2 / 2 | [cx cy cz] s | k' s^2 (k' = r^2 - |c|^2 + the sphere's share Gs of the error margin) | 1
// every feature split into two f16 pieces, in the slot order documented there.  Row i of the instruction holds sphere
// 16 ((i >> 2) & 1) + (((i >> 3) << 2) | (i & 3)) of the block, so that result register r of lane (H, j) is sphere 16 H + r.
// `geom`: n entries; entries with r^2 < -1e29 are padding (never a candidate).  *ops_out / *blocks_out receive the device
// array; the scale constants in `h` depend on the set of spheres only, so both orders of a scene get the same ones.
template <typename T, typename V4>
int build_mfma_operands(const std::vector<V4> &geom, int n, rtw_scene_dev *h, void **ops_out, int *blocks_out, int n_skip, const int *skip) {
    *ops_out = nullptr; *blocks_out = 0;
    if (n <= 0) return 0;
    auto live = [&](int i) { return i < n && (double)geom[i].w > -1e29; };
    // `skip`: spheres whose filter ROW is disabled (written like a padding row: never flagged for a ray that uses the filter) because
    // every lane tests them exactly by itself (DevScene::huge).  They still count for the scales: both orders of a scene share those.
    auto in_filter = [&](int i) { if (!live(i)) return false; for (int k = 0; k < n_skip; ++k) if (skip[k] == i) return false; return true; };
    double emax = 0;
    for (int i = 0; i < n; ++i) {
        if (!live(i)) continue;
        const double r = std::sqrt(std::fabs((double)geom[i].w));
        emax = std::max(emax, std::max(std::max(std::fabs((double)(float)geom[i].x), std::fabs((double)(float)geom[i].y)),
                                       std::max(std::fabs((double)(float)geom[i].z), r)));
    }
    if (!(emax > 0) || !std::isfinite(emax)) return 0;
    int ex = 0;
    (void)std::frexp(emax, &ex);                         // emax <= 2^ex
    if (ex > 40 || ex < -40) return 0;                   // outside what the scaled f16 pieces cover: VALU scan
    // lengths x s: sphere coordinates and radii use 2^8 of the f16 range (their products, halved: 2^15), ray origins may use
    // 2^13 -- rays up to 32 x the scene's extent away still take the filter (2 |p_k| s <= 2 x 2.74 x 2^13 < 65504).
    // phi_c: a coordinate's second f16 piece is a subnormal below 2^-3 (absolute error 2^-25 scaled); phi_k: the same floor for
    // the 2^4-scaled pieces of k' and of the ray's constant and for the second pieces of the quadratic features.
    const double sc = std::ldexp(1.0, 8 - ex), sig2 = sc * sc;
    const double phi_c = std::ldexp(1.0, -25) / sc, phi_k = std::ldexp(1.0, -20) / sig2;
    const double A_S = std::ldexp(1.0, -17), A_r = std::ldexp(12.0, -22);
    auto split = [](double x, unsigned &p1, unsigned &p2) {            // the f16 pieces of (float)x
        const float xf = (float)x;
        const _Float16 h1 = (_Float16)xf;
        const _Float16 h2 = (_Float16)(xf - (float)h1);
        unsigned short b1, b2;
        memcpy(&b1, &h1, 2); memcpy(&b2, &h2, 2);
        p1 = b1; p2 = b2;
    };
    const int nb = (n + 31) / 32;
    std::vector<uint4> ops((size_t)(nb + 1) * 128);
    for (int blk = 0; blk <= nb; ++blk)
        for (int lane = 0; lane < 64; ++lane) {
            const int i = lane & 31, H = lane >> 5;
            const int sph = blk * 32 + 16 * ((i >> 2) & 1) + (((i >> 3) << 2) | (i & 3));
            double fq[6] = {0, 0, 0, 0, 0, 0}, fl[3] = {0, 0, 0};
            double kx = -1073741824.0;                                        // padding sphere: k' s^2 = -2^30: W = -2^30 + (q^2 - oo') s^2 < 0
            if (in_filter(sph)) {
                const double cx = (double)(float)geom[sph].x, cy = (double)(float)geom[sph].y, cz = (double)(float)geom[sph].z;
                const double r2 = (double)geom[sph].w, c2 = cx * cx + cy * cy + cz * cz;
                const double Gs = 1.02 * ((2 * A_S + A_r) * c2 + A_r * r2 + 9 * phi_c * (std::fabs(cx) + std::fabs(cy) + std::fabs(cz)) + 1.5 * phi_k);
                kx = (r2 - c2 + Gs) * sig2;
                const double hs = 0.5 * sig2;
                fq[0] = cx * cx * hs; fq[1] = cy * cy * hs; fq[2] = cz * cz * hs; fq[3] = cx * cy * hs; fq[4] = cx * cz * hs; fq[5] = cy * cz * hs;
                fl[0] = cx * sc; fl[1] = cy * sc; fl[2] = cz * sc;
            }
            unsigned q1[6], q2[6], l1[3], l2[3];
            for (int k = 0; k < 6; ++k) split(fq[k], q1[k], q2[k]);
            for (int k = 0; k < 3; ++k) split(fl[k], l1[k], l2[k]);
            // k' s^2 = 2^15 k1 + 2^4 k2, the remainder rounded UP (a larger k' only widens the filter)
            const _Float16 k1 = (_Float16)(float)(kx / 32768.0);
            const double rem = (kx - 32768.0 * (double)(float)k1) / 16.0;
            _Float16 k2 = (_Float16)(float)rem;
            if ((double)(float)k2 < rem) { unsigned short b; memcpy(&b, &k2, 2); b = (unsigned short)((float)k2 >= 0.0f ? b + 1 : b - 1); memcpy(&k2, &b, 2); }
            unsigned short kb1, kb2;
            memcpy(&kb1, &k1, 2); memcpy(&kb2, &k2, 2);
            auto pk = [](unsigned lo, unsigned hi) { return (lo & 0xffffu) | (hi << 16); };
            // sphere pieces per slot (rtw_device.hpp): a feature's three slots are (1, 2, 1) against the ray's (1, 1, 2)
            uint4 m1, m2;
            if (H == 0) {
                m1 = uint4{pk(q1[0], q2[0]), pk(q1[0], q1[1]), pk(q2[1], q1[1]), pk(q1[2], q2[2])};      // xx xx | xx yy | yy yy | zz zz
                m2 = uint4{pk(q2[5], q1[5]), pk(l1[0], l2[0]), pk(l1[0], l1[1]), pk(l2[1], l1[1])};      // yz yz | px px | px py | py py
            } else {
                m1 = uint4{pk(q1[2], q1[3]), pk(q2[3], q1[3]), pk(q1[4], q2[4]), pk(q1[4], q1[5])};      // zz xy | xy xy | xz xz | xz yz
                m2 = uint4{pk(l1[2], l2[2]), pk(l1[2], kb1), pk(kb2, 0x7800u), 0x4c004c00u};             // pz pz | pz k | k T | T T
            }
            ops[(size_t)blk * 128 + lane] = m1;
            ops[(size_t)blk * 128 + 64 + lane] = m2;
        }
    HIP_TRY(hipMalloc(ops_out, ops.size() * sizeof(uint4)));
    HIP_TRY(hipMemcpy(*ops_out, ops.data(), ops.size() * sizeof(uint4), hipMemcpyHostToDevice));
    *blocks_out = nb;
    h->mf_sc = (float)sc; h->mf_sigma2 = (float)sig2;
    float keep = (float)(1.0 - 1.02 * 2 * A_S);
    if ((double)keep > 1.0 - 1.02 * 2 * A_S) keep = std::nextafter(keep, 0.0f);
    h->mf_oo_keep = keep;
    float coef = (float)(1.02 * 9 * phi_c);
    if ((double)coef < 1.02 * 9 * phi_c) coef = std::nextafter(coef, INFINITY);
    h->mf_o1_coef = coef;
    h->mf_o_max = (float)(std::ldexp(1.0, 13) / sc);
    return 0;
}

template <typename T, typename SceneT>
int upload_scene(const SceneT *s, int device, rtw_scene_handle *out) {
    if (!s || !out) return fail(-1, "null argument");
    if (s->n < 0) return fail(-2, "scene.n < 0");
    if (s->n > 0 && (!s->cx || !s->cy || !s->cz || !s->r || !s->kind || !s->ar || !s->ag || !s->ab || !s->param))
        return fail(-1, "null scene array");
    for (int i = 0; i < s->n; ++i) {
        if (s->kind[i] < 0 || s->kind[i] > 2) return fail(-3, "sphere %d: unknown material kind %d", i, s->kind[i]);
        if (!std::isfinite((double)s->cx[i]) || !std::isfinite((double)s->cy[i]) || !std::isfinite((double)s->cz[i]) ||
            !std::isfinite((double)s->r[i]))
            return fail(-3, "sphere %d: centre / radius is not finite", i);
    }
    DeviceGuard guard;
    int dev;
    if (int rc = resolve_device(device, &dev)) return rc;
    CtxPtr ctx;
    if (int rc = get_ctx(dev, &ctx)) return rc;
    HIP_TRY(hipSetDevice(dev));
    using V4 = typename rtw::Vec4<T>::type;
    const int n = s->n;
    constexpr int grp = rtw::ScanGroup<T>::N;                                            // 8 (f32) / 4 (f64)
    const int n_pad = ((n + grp - 1) / grp) * grp;                                       // 0 spheres: no scan at all
    const int n_alloc = rtw::scene_geom_alloc(n, n_pad);                                  // prefetch tail group / whole blocks of 32
    if (n_pad >= 65536) return fail(-5, "too many spheres (%d): candidate lists hold 16-bit indices", n);
    std::vector<V4> geom(n_alloc), mat0(n_alloc), mat1(n_alloc);
    for (int i = 0; i < n_alloc; ++i) {
        if (i < n) {
            geom[i] = V4{s->cx[i], s->cy[i], s->cz[i], s->r[i] * s->r[i]};  // r^2: src/hit.jl:17
            material_rows<T, V4>(s, i, mat0[i], mat1[i]);
        } else {
            // padding sphere that can never be hit: r^2 hugely negative => disc < 0 always
            geom[i] = V4{(T)0, (T)0, (T)0, (T)-1e30};
            mat0[i] = V4{(T)1, (T)0, (T)0, (T)0};
            mat1[i] = V4{(T)0, (T)0, (T)0, (T)0};
        }
    }
    ScenePtr h(new rtw_scene_dev());                      // freed on every error path below
    memset(h.get(), 0, sizeof(rtw_scene_dev));
    h->device = dev; h->is_f64 = sizeof(T) == 8; h->n = n; h->n_pad = n_pad;
    const size_t bytes = sizeof(V4) * (size_t)n_alloc;
    HIP_TRY(hipMalloc(&h->geom, bytes));
    HIP_TRY(hipMalloc(&h->mat0, bytes));
    HIP_TRY(hipMalloc(&h->mat1, bytes));
    HIP_TRY(hipMemcpy(h->geom, geom.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->mat0, mat0.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->mat1, mat1.data(), bytes, hipMemcpyHostToDevice));
    if (sizeof(T) == 8) {
        // pass-1 filter data of hit_world<double>: centre and r^2 rounded to binary32 (to nearest) and the sphere's
        // share G of the error margin, rounded up (derivation in rtw_device.hpp)
        std::vector<float> f((size_t)n_alloc * 8, 0.0f);
        for (int i = 0; i < n_alloc; ++i) {
            float *q = &f[(size_t)i * 8];
            q[0] = (float)geom[i].x; q[1] = (float)geom[i].y; q[2] = (float)geom[i].z; q[3] = (float)geom[i].w;
            if (i < n) {
                const double r2 = (double)geom[i].w, c2 = (double)geom[i].x * geom[i].x + (double)geom[i].y * geom[i].y + (double)geom[i].z * geom[i].z;
                const double G = 1.01 * (std::ldexp(r2, -18) + std::ldexp(c2, -20) + std::ldexp(r2, -20)) + 1e-30;
                float g = (float)G;
                if ((double)g < G) g = std::nextafter(g, INFINITY);
                if (!(c2 < 1e30) || !(r2 < 1e30)) g = INFINITY;                  // astronomically large: always a candidate
                q[4] = g;
            }                                                                     // padding spheres: r^2 = -1e30, G = 0
        }
        HIP_TRY(hipMalloc(&h->scan, f.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->scan, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // "Huge" spheres: a ground sphere of radius 1000 under spheres of radius 0.2 has a non-negative discriminant for nearly every ray, so
    // it costs every scan two non-skipped half blocks, an extraction trip and a full 64-entry batch of pass 2.  At most two spheres whose
    // radius is >= 16 x the median radius are instead tested exactly by every lane for its own ray (hit_world_mfma; same contract test,
    // same tie rule), and their rows of the filter are disabled.  Scheduling only: the image cannot change.
    h->n_huge = 0;
    if (n >= 8) {
        std::vector<double> ra(n);
        for (int i = 0; i < n; ++i) ra[i] = std::fabs((double)s->r[i]);
        std::vector<double> sorted(ra);
        std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
        const double med = sorted[n / 2];
        for (int pick = 0; pick < 2; ++pick) {
            int best = -1;
            for (int i = 0; i < n; ++i) {
                if (!(ra[i] >= 16.0 * med) || !std::isfinite(ra[i])) continue;
                if (h->n_huge > 0 && h->huge[0] == i) continue;
                if (best < 0 || ra[i] > ra[best]) best = i;
            }
            if (best < 0) break;
            h->huge[h->n_huge++] = best;
        }
    }
    static const bool env_no_huge = aid_flag("RTW_NO_HUGE");      // A/B aid
    if (env_no_huge) h->n_huge = 0;
    if (int rc = build_mfma_operands<T>(geom, n, h.get(), &h->mf_ops, &h->mf_blocks, h->n_huge, h->huge)) return rc;
    if (int rc = build_cull<T>(s, h.get())) return rc;
    *out = h.release();
    return 0;
}

int upload_scene_f32(const rtw_scene_f32 *s, int device, rtw_scene_handle *out) { return upload_scene<float>(s, device, out); }
int upload_scene_f64(const rtw_scene_f64 *s, int device, rtw_scene_handle *out) { return upload_scene<double>(s, device, out); }

}  // namespace rtwh
