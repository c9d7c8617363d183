// rtw_host.hpp -- host-side internals of librtw_hip.so shared by its translation units (nothing here is part of the C ABI):
//   rtw_abi.hip          the extern "C" entry points (include/rtw_hip.h), argument validation, the per-device contexts and the per-render
//                        records behind rtw_stats()
//   rtw_scene.hip        scene upload: SoA rows, kd split of the group-cull layout, the f16-split operands of the matrix-pipe filter
//   rtw_launch.hip       one render = one launch of the trace kernel (rtw_kernels.hpp / rtw_pool.hpp): geometry, job shape, counters
//   rtw_render_host.hip  the host-buffer entry points: cached per-device context (scene, stream, image), one device or a device list
//   rtw_multi.hip        what a device list needs: peer access, the on-demand RCCL binding, the un-tile kernel
//   rtw_unit.hip         the T0 unit entry points (rtw_units.hpp)
// Everything is in namespace rtwh with hidden visibility; the library exports the C ABI only.
#pragma once
#pragma GCC visibility push(default)
#include "../../include/rtw_hip.h"
#pragma GCC visibility pop

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#pragma GCC visibility push(hidden)

namespace rtw { struct DevCounters; }      // rtw_kernels.hpp (device side); the host keeps one per render record


// ---- device scene handle (the C ABI's opaque rtw_scene_handle) ------------------------------------
struct rtw_scene_dev {
    int device;
    int is_f64;
    int n, n_pad;
    void *geom, *mat0, *mat1;
    void *scan;      // Float64: the binary32 filter array of pass 1 (8 floats per sphere); Float32: null (geom itself)
    // pass 1 on the matrix pipe (hit_world_mfma): A operands per block of 32 spheres, scales and the ray's share of the margin
    void *mf_ops;    // null: the scene's extent is outside what the f16 split covers (the VALU scan is used)
    int mf_blocks;
    float mf_sc, mf_sigma2, mf_oo_keep, mf_o1_coef, mf_o_max;
    int n_huge, huge[2];   // spheres that pass the filter for nearly every ray (a ground sphere): tested exactly by every lane, their filter rows disabled
    // group-cull mode on the matrix pipe: the same operands in the cluster-major order + one box per block of 32
    void *c_mf_ops, *c_mf_box;
    float c_glo[3], c_ghi[3];   // the box of the whole small class (union of the block boxes): the ray is clipped against it once per scan
    int c_mf_blocks;
    float c_grid[6];       // the bins of the per-ray block vote (rtw::CullGrid: inv[3], off[3]); its tables lie behind the boxes in c_mf_box
    int c_n_inlane, c_inlane[8];   // spheres of the cluster-major order that every lane tests by itself (rtw::RTW_CULL_INLANE_MAX): the huge spheres and, when it fits, the whole BIG class
    // opt-in group-cull mode (RTW_FLAG_GROUP_CULL): cluster-major copies
    void *c_bound, *c_exact, *c_mat0, *c_mat1;
    unsigned short *c_orig;
    int c_groups_pad, c_big;
    double c_cs[3], c_rs;
};

namespace rtwh {

extern __thread char g_err[512];         // (__thread: no dynamic initialisation, so no init-function call through a hidden weak symbol from the other translation units)
int fail(int code, const char *fmt, ...);

// Measurement / test switches of the environment are honoured only under the master switch RTW_ENABLE_TEST_AIDS=1 (read once):
// without it a stray RTW_SCAN=valu or RTW_JOB_PIXELS=1 in a caller's environment changes nothing (include/rtw_hip.h).
bool test_aids();
const char *aid_env(const char *name);
bool aid_flag(const char *name);

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::rtwh::fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// for calls whose failure cannot be acted upon (frees on cleanup paths): RTW_DEBUG=1 reports them on stderr
#define HIP_IGNORE(expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            static const bool dbg_ = ::rtwh::aid_env("RTW_DEBUG") != nullptr;                          \
            if (dbg_) fprintf(stderr, "[rtw debug] %s -> %s (%s:%d)\n", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            (void)hipGetLastError();            /* do not leave it for a later hipGetLastError() check */ \
        }                                                                                     \
    } while (0)

// restores the caller's current device when an entry point returns
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) prev = -1; }
    ~DeviceGuard() { if (prev >= 0) HIP_IGNORE(hipSetDevice(prev)); }
};

// ---- per-render record: device counters + the events that time the trace kernel ---------------
struct RenderRec {
    int device = -1;
    rtw::DevCounters *ctr = nullptr;     // device memory
    rtw::DevCounters *h_ctr = nullptr;   // pinned host copy: filled by an asynchronous D2H behind the kernel on the render's stream (no blocking copy per render)
    size_t ctr_bytes = 0;                // how much of it that copy brought over (the head; everything with the drain profile)
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;      // around the kernel; behind the copy of the counters
    bool used = false;                   // ev1 has been recorded at least once
    bool done = true;                    // the kernel recorded by ev1 is known to have finished (no hipEventQuery needed)
    bool owned = false;                  // referenced by some thread's "last render"
    bool fresh = true;                   // the device counters have never been cleared as a whole
    int n_spheres = 0, n_chunks = 0, grid = 0, block = 256;
    ~RenderRec() {
        if (ctr) HIP_IGNORE(hipFree(ctr));
        if (h_ctr) HIP_IGNORE(hipHostFree(h_ctr));
        if (ev2) HIP_IGNORE(hipEventDestroy(ev2));
        if (ev0) HIP_IGNORE(hipEventDestroy(ev0));
        if (ev1) HIP_IGNORE(hipEventDestroy(ev1));
    }
};

// ---- persistent context of the host-buffer entry points (rtw_render_f32/_f64) -----------------------
// What a caller that renders frame after frame through the Julia `render()` shim pays per call, besides the kernel, is the
// image D2H: the uploaded scene (kd split + matrix-pipe operands: a dozen hipMalloc + synchronous copies), the stream and the
// device image are kept per device and reused while the scene's bytes are the same.
struct HostCtx {
    int device = -1;
    bool busy = false;                       // in use by a render call (guarded by DeviceCtx::mu)
    hipStream_t stream = nullptr;
    rtw_scene_handle scene = nullptr;        // the cached upload ...
    std::vector<unsigned char> scene_key;    // ... and the exact bytes it was made from (precision tag, n, the nine arrays)
    void *d_img = nullptr;  size_t d_cap = 0;      // device image / compact shard
    void *d_aux = nullptr;  size_t aux_cap = 0;    // multi-device root: the gathered compact shards
    hipEvent_t done_ev = nullptr;                  // multi-device: this shard has arrived in the root's gather buffer
    void *h_stage = nullptr; size_t stage_cap = 0; // multi-device without peer access: pinned staging of this shard
    unsigned long long last_use = 0;               // pool eviction: least recently used idle entry
    ~HostCtx();
};

struct DeviceCtx {
    int device = -1;
    int num_cus = 0;
    size_t lds_per_cu = 0;                         // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor (160 KB on MI355X)
    std::vector<std::pair<std::pair<const void *, size_t>, int>> occupancy;   // (kernel, dynamic LDS) -> workgroups per CU (asked of the runtime once; guarded by mu)
    std::mutex mu;
    std::vector<std::unique_ptr<RenderRec>> recs;
    std::vector<std::unique_ptr<HostCtx>> host;    // at most RTW_HOST_CTX_POOL cached entries
    std::map<int, bool> peer;                      // peer device -> access enabled in both directions (ensure_peer)
    unsigned long long use_clock = 0;
};
#define RTW_HOST_CTX_POOL 8          // cached host contexts per device at most (what concurrent callers can hold)
#define RTW_HOST_CTX_IDLE_KEEP 3     // ... of which idle ones kept for other scenes (acquire_host)
using CtxPtr = std::shared_ptr<DeviceCtx>;       // holders keep a context alive across a concurrent rtw_shutdown()


extern std::atomic<unsigned> g_generation;      // bumped by rtw_shutdown: invalidates every thread's "last render"

void release_last();
// what rtw_stats() reports: the records of the last render issued from this thread
struct LastRender {
    unsigned generation = 0;
    bool resolved = false;
    std::vector<RenderRec *> recs;          // pending (device-resident call) or already summed into `agg`
    std::vector<CtxPtr> ctxs;               // the contexts that own `recs` (kept alive; parallel to recs)
    rtw_stats_t agg;
    std::vector<std::pair<int, double>> per_device;   // (device ordinal, kernel ms) of every shard of the last render, in shard order (rtw_stats_devices)
    ~LastRender();                          // a thread that exits hands its records back
};
extern thread_local LastRender g_last;

int get_ctx(int device, CtxPtr *out);
int acquire_rec(DeviceCtx *ctx, RenderRec **out);           // a record nobody references whose previous kernel (if any) has finished; the device must be current
void release_rec(const CtxPtr &ctx, RenderRec *r, bool finished);
int resolve_device(int device, int *out);
int validate_params(const rtw_params *p, int *n_chunks, int *chunk_spp);
long long local_tiles(const rtw_params *p);

template <typename SceneT> bool s_has_bad_scene(const SceneT *s) {
    return s->n > 0 && (!s->cx || !s->cy || !s->cz || !s->r || !s->kind || !s->ar || !s->ag || !s->ab || !s->param);
}

struct SceneDeleter { void operator()(rtw_scene_dev *h) const { rtw_scene_free(h); } };
using ScenePtr = std::unique_ptr<rtw_scene_dev, SceneDeleter>;

// rtw_scene.hip
int upload_scene_f32(const rtw_scene_f32 *s, int device, rtw_scene_handle *out);
int upload_scene_f64(const rtw_scene_f64 *s, int device, rtw_scene_handle *out);
inline int upload_scene_t(const rtw_scene_f32 *s, int device, rtw_scene_handle *out) { return upload_scene_f32(s, device, out); }
inline int upload_scene_t(const rtw_scene_f64 *s, int device, rtw_scene_handle *out) { return upload_scene_f64(s, device, out); }

// rtw_launch.hip -- enqueue one render (this shard's tiles) on `stream`; `rec` receives the counters and the kernel's events
int launch_render_f32(rtw_scene_handle scene, const rtw_camera_f32 *cam, const rtw_params *p, void *d_out, hipStream_t stream, RenderRec **rec_out, CtxPtr *ctx_out);
int launch_render_f64(rtw_scene_handle scene, const rtw_camera_f64 *cam, const rtw_params *p, void *d_out, hipStream_t stream, RenderRec **rec_out, CtxPtr *ctx_out);
inline int launch_render_t(rtw_scene_handle s, const rtw_camera_f32 *c, const rtw_params *p, void *d, hipStream_t st, RenderRec **r, CtxPtr *x) { return launch_render_f32(s, c, p, d, st, r, x); }
inline int launch_render_t(rtw_scene_handle s, const rtw_camera_f64 *c, const rtw_params *p, void *d, hipStream_t st, RenderRec **r, CtxPtr *x) { return launch_render_f64(s, c, p, d, st, r, x); }
int resolve_rec(RenderRec *r, rtw_stats_t *agg);            // wait for a record's kernel and add its counters to `agg`

// rtw_render_host.hip
int render_host_f32(const rtw_scene_f32 *scene, const rtw_camera_f32 *cam, const rtw_params *p, float *out);
int render_host_f64(const rtw_scene_f64 *scene, const rtw_camera_f64 *cam, const rtw_params *p, double *out);

// rtw_multi.hip
int ensure_peer(const CtxPtr &ctx, int dev, int root, bool *direct);
struct RcclSet;                                               // the communicators of one device list (one rank per device)
int rccl_acquire(const std::vector<int> &devs, std::shared_ptr<RcclSet> *out, std::unique_lock<std::mutex> *use);
int rccl_reduce_frames(RcclSet &set, const std::vector<const void *> &send, void *recv_root, size_t count, bool f64, const std::vector<hipStream_t> &streams);
void rccl_shutdown();
int launch_untile(bool f64, const void *gather, void *frame, int W, int H, long n_tiles, int n_shards, long pad_tiles, hipStream_t stream);

// rtw_unit.hip
int run_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene, const rtw_camera_f32 *cam);
int run_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene, const rtw_camera_f64 *cam);

}  // namespace rtwh

#pragma GCC visibility pop
