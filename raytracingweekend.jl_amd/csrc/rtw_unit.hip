// rtw_unit.hip -- the T0 unit entry points (rtw_unit_f32/_f64): host slots -> device -> unit_kernel (rtw_units.hpp) -> host slots
#include "rtw_scene_view.hpp"
#include "rtw_units.hpp"

namespace rtwh {

// T0 unit entry point: host slots -> device -> unit_kernel -> host slots
template <typename T, typename SceneT, typename CamT>
int run_unit(int op_arg, int count, const void *in, void *out, const SceneT *scene, const CamT *cam) {
    const int op = op_arg & 0xff, numerics = (op_arg >> 8) & 3;      // bits 8-9: the numerics mode of the ray-sphere test
    if (op_arg < 0 || (op_arg >> 10) != 0 || op >= rtw::U_NUM_OPS) return fail(-2, "unknown unit op %d", op_arg);
    if (numerics == 2) return fail(-2, "unknown numerics mode %d (unit op %d)", numerics, op_arg);
    if (count < 0 || (count > 0 && (!in || !out))) return fail(-1, "null argument");
    if (count == 0) return 0;
    const bool needs_scene = op == rtw::U_HIT_WORLD || op == rtw::U_RAY_COLOR || op == rtw::U_HIT_WORLD_LDS || op == rtw::U_HIT_WORLD_CULL || op == rtw::U_HIT_WORLD_MFMA || op == rtw::U_HIT_WORLD_MFMA_CULL;
    if (needs_scene && !scene) return fail(-1, "op %d needs a scene", op);
    if (op == rtw::U_GET_RAY && !cam) return fail(-1, "op %d needs a camera", op);
    DeviceGuard guard;
    int dev;
    if (int rc = resolve_device(-1, &dev)) return rc;
    CtxPtr ctx;
    if (int rc = get_ctx(dev, &ctx)) return rc;
    rtw_scene_handle h_raw = nullptr;
    rtw::DevScene<T> S;
    memset(&S, 0, sizeof S);
    rtw::CullScene<T> CS;
    memset(&CS, 0, sizeof CS);
    if (needs_scene) {
        if (int rc = upload_scene_t(scene, dev, &h_raw)) return rc;
        S = dev_scene_of<T>(h_raw);
        CS = cull_scene_of<T>(h_raw);
    }
    S.numerics = numerics; CS.numerics = numerics;
    ScenePtr h(h_raw);
    HIP_TRY(hipSetDevice(dev));
    rtw::Camera<T> C;
    memset(&C, 0, sizeof C);
    if (cam) {
        for (int k = 0; k < 3; ++k) {
            C.origin[k] = cam->origin[k]; C.llc[k] = cam->lower_left_corner[k];
            C.horizontal[k] = cam->horizontal[k]; C.vertical[k] = cam->vertical[k];
            C.u[k] = cam->u[k]; C.v[k] = cam->v[k]; C.w[k] = cam->w[k];
        }
        C.lens_radius = cam->lens_radius;
    }
    using V4 = typename rtw::Vec4<T>::type;
    size_t lds_bytes = 0;
    if (op == rtw::U_HIT_WORLD_LDS || op == rtw::U_HIT_WORLD_MFMA) lds_bytes = (size_t)rtw::scene_geom_alloc(S.n, S.n_pad) * sizeof(V4);
    if (op == rtw::U_HIT_WORLD_MFMA && !S.mf_ops) return fail(-5, "the scene has no matrix-pipe scan operands (unit op %d)", op);
    if (op == rtw::U_HIT_WORLD_MFMA_CULL && !CS.mf_ops) return fail(-5, "the scene has no matrix-pipe cull operands (unit op %d)", op);
    if (op == rtw::U_HIT_WORLD_CULL || op == rtw::U_HIT_WORLD_MFMA_CULL) {
        const size_t n_cull = (size_t)rtw::cull_exact_count(CS);
        lds_bytes = n_cull * sizeof(V4) + ((n_cull * sizeof(unsigned short) + 15) / 16) * 16;
    }
    if (lds_bytes > 60 * 1024) return fail(-5, "scene too large for the LDS-staged unit op %d (%zu bytes)", op, lds_bytes);
    const size_t in_b = (size_t)count * rtw::unit_in_slots(op) * 8, out_b = (size_t)count * rtw::unit_out_slots(op) * 8;
    double *d_in = nullptr, *d_out = nullptr;
    int rc = 0;
    hipError_t e;
    if ((e = hipMalloc(&d_in, in_b)) != hipSuccess || (e = hipMalloc(&d_out, out_b)) != hipSuccess ||
        (e = hipMemcpy(d_in, in, in_b, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail((int)e, "unit buffers: %s", hipGetErrorString(e));
    } else {
        (void)hipGetLastError();
        hipLaunchKernelGGL(rtw::unit_kernel<T>, dim3((count + 63) / 64), dim3(64), lds_bytes, 0, op, count, d_in, d_out, S, CS, C);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail((int)e, "unit kernel: %s", hipGetErrorString(e));
    }
    if (d_in) HIP_IGNORE(hipFree(d_in));
    if (d_out) HIP_IGNORE(hipFree(d_out));
    return rc;
}

int run_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene, const rtw_camera_f32 *cam) { return run_unit<float>(op, count, in, out, scene, cam); }
int run_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene, const rtw_camera_f64 *cam) { return run_unit<double>(op, count, in, out, scene, cam); }

}  // namespace rtwh
