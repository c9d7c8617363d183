// rtw_scene_view.hpp -- the device-side views (rtw_device.hpp DevScene / CullScene) of an uploaded scene handle
#pragma once
#include "rtw_host.hpp"
#include "rtw_device.hpp"

namespace rtwh {

template <typename T>
rtw::CullScene<T> cull_scene_of(const rtw_scene_dev *h) {
    using V4 = typename rtw::Vec4<T>::type;
    rtw::CullScene<T> C;
    C.box = (const T *)h->c_bound; C.exact = (const V4 *)h->c_exact; C.orig = h->c_orig;
    C.mat0 = (const V4 *)h->c_mat0; C.mat1 = (const V4 *)h->c_mat1;
    C.n_groups_pad = h->c_groups_pad; C.n_big = h->c_big;
    C.cs[0] = (T)h->c_cs[0]; C.cs[1] = (T)h->c_cs[1]; C.cs[2] = (T)h->c_cs[2]; C.rs = (T)h->c_rs;
    C.kappa = sizeof(T) == 4 ? (T)0.00390625 : (T)2.384185791015625e-07;     // 2^-8 / 2^-22
    C.mf_ops = (const uint4 *)h->c_mf_ops; C.mf_box = (const float *)h->c_mf_box; C.mf_blocks = h->c_mf_blocks;
    for (int k = 0; k < 3; ++k) { C.mf_glo[k] = h->c_glo[k]; C.mf_ghi[k] = h->c_ghi[k]; }
    static_assert(RTW_CULL_INLANE_MAX == 8, "rtw_scene_dev::c_inlane");
    C.n_huge = h->c_mf_ops ? h->c_n_inlane : 0;
    C.n_huge_exact = h->c_mf_ops ? std::min(h->n_huge, h->c_n_inlane) : 0;      // (the huge spheres come first in c_inlane)
    for (int k = 0; k < 3; ++k) { C.grid.inv[k] = h->c_grid[k]; C.grid.off[k] = h->c_grid[3 + k]; }
    C.numerics = rtw::NUM_REFERENCE;
    return C;
}

template <typename T>
rtw::DevScene<T> dev_scene_of(const rtw_scene_dev *h) {
    using V4 = typename rtw::Vec4<T>::type;
    rtw::DevScene<T> S;
    memset(&S, 0, sizeof S);
    S.geom = (const V4 *)h->geom; S.mat0 = (const V4 *)h->mat0; S.mat1 = (const V4 *)h->mat1;
    S.scan = (const float *)(h->scan ? h->scan : h->geom);
    S.n = h->n; S.n_pad = h->n_pad;
    S.mf_ops = (const uint4 *)h->mf_ops; S.mf_blocks = h->mf_blocks;
    S.mf_sc = h->mf_sc; S.mf_sigma2 = h->mf_sigma2; S.mf_oo_keep = h->mf_oo_keep; S.mf_o1_coef = h->mf_o1_coef; S.mf_o_max = h->mf_o_max;
    S.n_huge = h->mf_ops ? h->n_huge : 0; S.huge[0] = h->huge[0]; S.huge[1] = h->huge[1];
    return S;
}

}  // namespace rtwh
