// rtw_hip.hip -- C ABI of librtw_hip.so (include/rtw_hip.h) over the gfx950 kernels.
// Replaces /root/reference/src/render.jl:8-44 behind a ccall-able boundary.  No CPU fallback:
// every compute entry point needs a HIP device and reports an error otherwise.
//
// Concurrency: every render call owns its own record (device counters + events) taken from a
// mutex-guarded per-device pool, the trace kernel keeps all per-render state in LDS / registers
// and there is no shared device workspace, so renders may be in flight concurrently on any mix
// of streams, host threads and devices.  The caller's current HIP device is restored on return.
#include "../../include/rtw_hip.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types and prototypes only: librccl is loaded on demand (dlopen), never linked
#include <dlfcn.h>

#include <algorithm>
#include <map>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <type_traits>
#include <mutex>
#include <thread>
#include <vector>

#include "rtw_kernels.hpp"
#include "rtw_pool.hpp"
#include "rtw_units.hpp"

namespace {

thread_local char g_err[512] = "";

// Measurement / test switches of the environment are honoured only under the master switch RTW_ENABLE_TEST_AIDS=1 (read once):
// without it a stray RTW_SCAN=valu or RTW_JOB_PIXELS=1 in a caller's environment changes nothing (include/rtw_hip.h).
bool test_aids() {
    static const bool on = [] { const char *e = getenv("RTW_ENABLE_TEST_AIDS"); return e != nullptr && atoi(e) != 0; }();
    return on;
}
const char *aid_env(const char *name) { return test_aids() ? getenv(name) : nullptr; }
bool aid_flag(const char *name) { const char *e = aid_env(name); return e != nullptr && atoi(e) != 0; }

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// for calls whose failure cannot be acted upon (frees on cleanup paths): RTW_DEBUG=1 reports them on stderr
#define HIP_IGNORE(expr)                                                                      \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) {                                                               \
            static const bool dbg_ = aid_env("RTW_DEBUG") != nullptr;                          \
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
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool used = false;                   // ev1 has been recorded at least once
    bool done = true;                    // the kernel recorded by ev1 is known to have finished (no hipEventQuery needed)
    bool owned = false;                  // referenced by some thread's "last render"
    int n_spheres = 0, n_chunks = 0, grid = 0, block = 256;
    ~RenderRec() {
        if (ctr) HIP_IGNORE(hipFree(ctr));
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
    std::mutex mu;
    std::vector<std::unique_ptr<RenderRec>> recs;
    std::vector<std::unique_ptr<HostCtx>> host;    // at most RTW_HOST_CTX_POOL cached entries
    std::map<int, bool> peer;                      // peer device -> access enabled in both directions (ensure_peer)
    unsigned long long use_clock = 0;
};
#define RTW_HOST_CTX_POOL 8
using CtxPtr = std::shared_ptr<DeviceCtx>;       // holders keep a context alive across a concurrent rtw_shutdown()

std::mutex g_mu;
std::vector<CtxPtr> g_ctx;
std::atomic<unsigned> g_generation{1};      // bumped by rtw_shutdown: invalidates every thread's "last render"

void release_last();
// what rtw_stats() reports: the records of the last render issued from this thread
struct LastRender {
    unsigned generation = 0;
    bool resolved = false;
    std::vector<RenderRec *> recs;          // pending (device-resident call) or already summed into `agg`
    std::vector<CtxPtr> ctxs;               // the contexts that own `recs` (kept alive; parallel to recs)
    rtw_stats_t agg;
    ~LastRender();                          // a thread that exits hands its records back
};
thread_local LastRender g_last;

int get_ctx(int device, CtxPtr *out) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &c : g_ctx)
        if (c->device == device) { *out = c; return 0; }
    CtxPtr c(new DeviceCtx());
    c->device = device;
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(-20, "device %d is %s; librtw_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    c->num_cus = prop.multiProcessorCount;
    *out = c;
    g_ctx.push_back(std::move(c));
    return 0;
}

// a record nobody references whose previous kernel (if any) has finished; the device must be current
int acquire_rec(DeviceCtx *ctx, RenderRec **out) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (auto &r : ctx->recs) {
        if (r->owned) continue;
        if (r->used && !r->done) {
            if (hipEventQuery(r->ev1) != hipSuccess) { (void)hipGetLastError(); continue; }   // still in flight
            r->done = true;
        }
        r->owned = true;
        *out = r.get();
        return 0;
    }
    std::unique_ptr<RenderRec> r(new RenderRec());
    r->device = ctx->device;
    HIP_TRY(hipMalloc(&r->ctr, sizeof(rtw::DevCounters)));
    HIP_TRY(hipEventCreate(&r->ev0));
    HIP_TRY(hipEventCreate(&r->ev1));
    r->owned = true;
    *out = r.get();
    ctx->recs.push_back(std::move(r));
    return 0;
}

void release_rec(const CtxPtr &ctx, RenderRec *r, bool finished) {
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (finished) r->done = true;
    r->owned = false;
}

void release_last() {
    if (g_last.generation == g_generation.load())
        for (size_t k = 0; k < g_last.recs.size(); ++k) release_rec(g_last.ctxs[k], g_last.recs[k], false);
    g_last.recs.clear();
    g_last.ctxs.clear();                   // (a stale generation: the records died with their contexts' pools; only the shared_ptrs are dropped)
    g_last.resolved = false;
    g_last.generation = g_generation.load();
    memset(&g_last.agg, 0, sizeof g_last.agg);
}
LastRender::~LastRender() {
    if (generation == g_generation.load())
        for (size_t k = 0; k < recs.size(); ++k) release_rec(ctxs[k], recs[k], false);
}

int resolve_device(int device, int *out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(e != hipSuccess ? (int)e : -21, "no HIP device available (%s); librtw_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0) { HIP_TRY(hipGetDevice(&device)); }
    if (device >= n) return fail(-22, "device %d out of range (%d devices)", device, n);
    *out = device;
    return 0;
}

}  // namespace

// ---- device scene handle ---------------------------------------------------------------------
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
    int c_mf_blocks;
    int c_huge[2];         // the huge spheres' indices in the cluster-major order (n_huge of them)
    // opt-in group-cull mode (RTW_FLAG_GROUP_CULL): cluster-major copies
    void *c_bound, *c_exact, *c_mat0, *c_mat1;
    unsigned short *c_orig;
    int c_groups_pad, c_big;
    double c_cs[3], c_rs;
};

namespace {

template <typename SceneT> bool s_has_bad_scene(const SceneT *s) {
    return s->n > 0 && (!s->cx || !s->cy || !s->cz || !s->r || !s->kind || !s->ar || !s->ag || !s->ab || !s->param);
}

struct SceneDeleter { void operator()(rtw_scene_dev *h) const { rtw_scene_free(h); } };
using ScenePtr = std::unique_ptr<rtw_scene_dev, SceneDeleter>;

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
        for (int k = 0; k < h->n_huge; ++k) {                     // the huge spheres (tested in-lane) in this order
            h->c_huge[k] = -1;
            for (int dI = 0; dI < n_exact; ++dI)
                if ((double)exact[dI].w > -1e29 && (int)orig[dI] == h->huge[k]) { h->c_huge[k] = dI; break; }
            if (h->c_huge[k] < 0) return fail(-9, "internal: huge sphere %d not found in the cull layout", h->huge[k]);
        }
        if (int rc = build_mfma_operands<T>(exact, n_exact, h, &h->c_mf_ops, &h->c_mf_blocks, h->n_huge, h->c_huge)) return rc;
        const int nb = h->c_mf_blocks;
        std::vector<float> bx((size_t)(nb + 1) * 8, 0.0f);
        for (int b = 0; b <= nb; ++b) {
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            bool all = false, any = false;
            for (int c = 2 * b; c < 2 * b + 2; ++c) {
                if (c >= ng_pad) { if (b < nb && c * GS < n_exact) all = true; continue; }   // BIG class (device indices >= ng_pad * GS)
                if (c >= ng) continue;                                                        // dead cluster
                any = true;
                for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], (double)box[(size_t)c * 8 + a]); hi[a] = std::max(hi[a], (double)box[(size_t)c * 8 + 4 + a]); }
            }
            float *q = &bx[(size_t)b * 8];
            for (int a = 0; a < 3; ++a) {
                if (all) { q[a] = -3.0e38f; q[4 + a] = 3.0e38f; }
                else if (!any) { q[a] = 1e15f; q[4 + a] = 1e15f; }                            // nothing alive: a point far away
                else {
                    float l = (float)lo[a], u2 = (float)hi[a];
                    if ((double)l > lo[a]) l = std::nextafter(l, -INFINITY);
                    if ((double)u2 < hi[a]) u2 = std::nextafter(u2, INFINITY);
                    q[a] = l; q[4 + a] = u2;
                }
            }
        }
        HIP_TRY(hipMalloc(&h->c_mf_box, bx.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(h->c_mf_box, bx.data(), bx.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    return 0;
}

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
    C.n_huge = h->c_mf_ops ? h->n_huge : 0; C.huge[0] = h->c_huge[0]; C.huge[1] = h->c_huge[1];
    C.numerics = rtw::NUM_REFERENCE;
    return C;
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

// magic number for exact unsigned 32-bit division by an invariant d >= 1 (Granlund-Montgomery / Hacker's
// Delight "add" form): n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> s for all 32-bit n
void make_udiv(unsigned d, unsigned *m, unsigned *s) {
    if (d <= 1) { *m = 0; *s = 0x80000000u; return; }   // flag: identity
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                        // l = ceil(log2 d) >= 1
    *m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    *s = l - 1;
}

int validate_params(const rtw_params *p, int *n_chunks, int *chunk_spp) {
    if (!p) return fail(-1, "null params");
    if (p->width <= 0 || p->height <= 0) return fail(-2, "width/height must be positive (got %d x %d)", p->width, p->height);
    if (p->spp <= 0) return fail(-2, "spp must be positive (got %d)", p->spp);
    if (p->max_depth < 0 || p->max_depth >= (1 << 22)) return fail(-2, "max_depth must be in [0, 2^22)");
    if (p->shard_count <= 0 || p->shard_index < 0 || p->shard_index >= p->shard_count)
        return fail(-2, "bad shard %d of %d", p->shard_index, p->shard_count);
    if (p->n_chunks < 0) return fail(-2, "n_chunks must be >= 0");
    if (p->flags & ~(RTW_FLAG_GROUP_CULL | RTW_FLAG_COMPACT_TILES | RTW_FLAG_SCAN_VALU | RTW_FLAG_RAY_POOL | RTW_FLAG_RCCL_REDUCE | RTW_FLAG_NUMERICS_CONTRACT | RTW_FLAG_NUMERICS_REFERENCE_FMA)) return fail(-2, "unknown flags 0x%x", p->flags);
    if ((p->flags & RTW_FLAG_NUMERICS_CONTRACT) && (p->flags & RTW_FLAG_NUMERICS_REFERENCE_FMA)) return fail(-2, "RTW_FLAG_NUMERICS_CONTRACT and RTW_FLAG_NUMERICS_REFERENCE_FMA exclude each other");
    // default rule: about 4 samples per chunk, between 16 and 256 chunks (never more than spp):
    // enough items for load balance, few enough stream set-ups (1 sample per chunk costs 7 % at Float64)
    int nch = p->n_chunks > 0 ? p->n_chunks : std::min(p->spp, std::max(16, std::min(256, p->spp / 4)));
    if (nch > p->spp) nch = p->spp;
    int cs = (p->spp + nch - 1) / nch;
    *chunk_spp = cs;
    *n_chunks = (p->spp + cs - 1) / cs;
    return 0;
}

long long local_tiles(const rtw_params *p) {
    const long long n_tiles = (long long)((p->height + 7) / 8) * ((p->width + 7) / 8);
    return n_tiles > p->shard_index ? (n_tiles - p->shard_index + p->shard_count - 1) / p->shard_count : 0;
}

// Enqueue one render (this shard's tiles) on `stream`; `rec` receives the counters and the kernel's events.
template <typename T, typename CamT>
int launch_render(rtw_scene_handle scene, const CamT *cam, const rtw_params *p, void *d_out, hipStream_t stream, RenderRec **rec_out, CtxPtr *ctx_out) {
    if (!scene || !cam || !d_out) return fail(-1, "null argument");
    if (scene->is_f64 != (sizeof(T) == 8)) return fail(-4, "scene handle precision does not match the call");
    int nch, cs;
    if (int rc = validate_params(p, &nch, &cs)) return rc;
    if (p->device >= 0 && p->device != scene->device)
        return fail(-4, "params.device %d != scene device %d", p->device, scene->device);
    CtxPtr ctx;
    if (int rc = get_ctx(scene->device, &ctx)) return rc;
    *ctx_out = ctx;
    HIP_TRY(hipSetDevice(scene->device));

    rtw::KParams K;
    memset(&K, 0, sizeof K);
    K.width = p->width; K.height = p->height; K.spp = p->spp; K.max_depth = p->max_depth;
    K.seed = p->seed; K.n_chunks = nch; K.chunk_spp = cs;
    K.shard_index = p->shard_index; K.shard_count = p->shard_count;
    K.tiles_i = (p->height + 7) / 8; K.tiles_j = (p->width + 7) / 8;
    const long long n_local = local_tiles(p);
    K.gamma = p->gamma;
    K.out_layout = (p->flags & RTW_FLAG_COMPACT_TILES) ? 1 : 0;
    make_udiv((unsigned)K.tiles_i, &K.div_tiles_m, &K.div_tiles_s);

    rtw::Camera<T> C;
    for (int k = 0; k < 3; ++k) {
        C.origin[k] = cam->origin[k]; C.llc[k] = cam->lower_left_corner[k];
        C.horizontal[k] = cam->horizontal[k]; C.vertical[k] = cam->vertical[k];
        C.u[k] = cam->u[k]; C.v[k] = cam->v[k]; C.w[k] = cam->w[k];
    }
    C.lens_radius = cam->lens_radius;
    using V4 = typename rtw::Vec4<T>::type;
    rtw::DevScene<T> S = dev_scene_of<T>(scene);
    // the deciding arithmetic of the ray-sphere test (include/rtw_hip.h RTW_FLAG_NUMERICS_*): a property of the render, not of the upload
    S.numerics = (p->flags & RTW_FLAG_NUMERICS_CONTRACT) ? rtw::NUM_CONTRACT : (p->flags & RTW_FLAG_NUMERICS_REFERENCE_FMA) ? rtw::NUM_REFERENCE_FMA : rtw::NUM_REFERENCE;

    // persistent grid: enough 256-thread blocks to fill every CU at the kernel's occupancy
    static const bool phase_profile = aid_env("RTW_PHASE_PROFILE") != nullptr;   // debugging aid, not for timed runs
    const size_t list_bytes = (size_t)RTW_LIST_CAP * 256 * sizeof(unsigned short);
    const size_t shared_bytes = (sizeof(rtw::WgShared<T>) + 15) / 16 * 16;
    const bool cull = (p->flags & RTW_FLAG_GROUP_CULL) != 0;
    rtw::CullScene<T> CS = cull_scene_of<T>(scene);
    CS.numerics = S.numerics;
    const size_t n_cull = (size_t)rtw::cull_exact_count(CS);
    const size_t geom_bytes = cull ? n_cull * sizeof(V4) + ((n_cull * sizeof(unsigned short) + 15) / 16) * 16
                                   : (size_t)rtw::scene_geom_alloc(scene->n, scene->n_pad) * sizeof(V4);
    const bool lds_scene = geom_bytes <= RTW_LDS_SCENE_MAX_BYTES;
    // the plain scan runs pass 1 on the matrix pipe (RTW_SCAN=valu: the all-VALU scan, for A/B measurements)
    static const bool force_valu = aid_env("RTW_SCAN") != nullptr && strcmp(aid_env("RTW_SCAN"), "valu") == 0;
    // (group cull: on the matrix pipe too when the scene has the operands; RTW_FLAG_SCAN_VALU selects the all-VALU cull scan)
    const bool mfma = (cull ? scene->c_mf_ops != nullptr : scene->mf_ops != nullptr) && !force_valu && !(p->flags & RTW_FLAG_SCAN_VALU);
    const size_t lds_bytes = list_bytes + shared_bytes + (mfma ? rtw::mfma_cell_bytes<T>() : 0) + (lds_scene ? geom_bytes : 0);
    typedef void (*kern_t)(rtw::KParams, rtw::Camera<T>, rtw::DevScene<T>, rtw::CullScene<T>, T *, rtw::DevCounters *);
    kern_t kern;
    if (cull && mfma && phase_profile) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, true, true, true, true> : (kern_t)rtw::trace_kernel<T, false, false, true, true>;
    else if (cull && mfma) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, true, true> : (kern_t)rtw::trace_kernel<T, false, false, true, true>;
    else if (cull && phase_profile) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, true, true, true> : (kern_t)rtw::trace_kernel<T, false, false, true>;
    else if (cull) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, true> : (kern_t)rtw::trace_kernel<T, false, false, true>;
    else if (mfma && phase_profile) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, true, true, false, true> : (kern_t)rtw::trace_kernel<T, true, false, false, true>;
    else if (mfma) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, false, true> : (kern_t)rtw::trace_kernel<T, false, false, false, true>;
    else if (phase_profile) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, true, true, false> : (kern_t)rtw::trace_kernel<T, true, false, false>;
    else kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, false> : (kern_t)rtw::trace_kernel<T, false, false, false>;
    // The ray-pool kernel (rtw_pool.hpp; opt-in: RTW_FLAG_RAY_POOL, or RTW_POOL=1 in the environment for A/B runs): Float32 plain
    // scans on the matrix pipe, when the pool, the rings and the scene copy fit the 160 KB of LDS of a CU (one workgroup of
    // RTW_POOL_W waves per CU); everything else runs the lane-loop kernel above.
    static const bool env_pool = aid_flag("RTW_POOL");
    size_t pool_lds = 0;
    bool pool = false;
    typedef void (*pool_kern_t)(rtw::KParams, rtw::Camera<T>, rtw::DevScene<T>, T *, rtw::DevCounters *);
    pool_kern_t pool_kern = nullptr;
    if constexpr (sizeof(T) == 4) {
        pool_lds = rtw::pool_fixed_lds_bytes<T, RTW_POOL_W, RTW_POOL_R>() + rtw::pool_scene_lds_bytes<T>(scene->n, scene->n_pad);
        pool = mfma && !cull && (env_pool || (p->flags & RTW_FLAG_RAY_POOL)) && pool_lds <= 160u * 1024u &&
               cs <= RTW_POOL_MAX_CHUNK_SPP;
        pool_kern = phase_profile ? (pool_kern_t)rtw::trace_pool_kernel<T, RTW_POOL_W, RTW_POOL_R, true> : (pool_kern_t)rtw::trace_pool_kernel<T, RTW_POOL_W, RTW_POOL_R, false>;
    }
    const int block_threads = pool ? RTW_POOL_W * 64 : 256;
    int blocks_per_cu = 0;
    if (pool) {
        HIP_TRY(hipFuncSetAttribute((const void *)pool_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool_lds));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, pool_kern, block_threads, pool_lds));
    } else {
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, 256, lds_bytes));
    }
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    long long grid = (long long)ctx->num_cus * blocks_per_cu;
    // Job size.  A job is owned by one workgroup, so its size sets the end-of-queue drain; smaller jobs also store the
    // image in smaller pieces (more partial-line writes).  2x2 pixels (a batch = 4 pixels x 16 chunks) when the chunks
    // fill such batches, else 4x4 (x 4 chunks); ONE pixel (x 64 chunks) when a workgroup would otherwise see fewer than
    // 150 jobs (small frames, shards of a multi-GPU render).  Measured at 1080p x 1000 spp / 250 chunks
    // (tools/gpu_drain.py): drain 4.4 / 9.7 / 30 ms of idle wave slots for 1 / 4 / 16-pixel jobs; full frame 859 / 859 /
    // 871 ms; a 1/8 shard 115.6 / 119.8 / 137.7 ms; HBM writes 148 / 72 / 45 MB per frame.
    // (Slots per workgroup: 24 / 12 / 4 -- one-pixel jobs need many slots in flight; with 6 they ran 33 % slower.)
    int job_shift = nch >= 16 ? 2 : 4;
    if (nch >= 64 && n_local * 16 < 150 * grid) job_shift = 0;
    // (measurement aid for A/B runs, tools/gpu_ab.sh: RTW_JOB_PIXELS = 1, 4, 8 or 16; any other value is ignored)
    static const int env_job_pixels = [] { const char *e = aid_env("RTW_JOB_PIXELS"); const int v = e ? atoi(e) : 0; return (v == 1 || v == 4 || v == 8 || v == 16) ? v : 0; }();
    const int job_pixels = p->job_pixels ? p->job_pixels : env_job_pixels;
    if (job_pixels == 16 || job_pixels == 8 || job_pixels == 4 || job_pixels == 1) {
        job_shift = job_pixels == 16 ? 4 : job_pixels == 8 ? 3 : job_pixels == 4 ? 2 : 0;
    } else if (job_pixels != 0) {
        return fail(-2, "job_pixels must be 0 (automatic), 1, 4, 8 or 16");
    }
    const long long total_jobs = n_local * (64 >> job_shift);
    const long long cpb = 64 >> job_shift;
    const long long bpj = (nch + cpb - 1) / cpb;
    // (claim_job packs a queue position into 28 bits; queue 0 is the longest: every 8th tile column, or every 8th tile of a shard)
    const long long queue0_jobs = (p->shard_count == 1 ? (long long)((K.tiles_j + 7) / 8) * K.tiles_i : (n_local + 7) / 8) * (64 >> job_shift);
    if (total_jobs >= (1ll << 31) || queue0_jobs >= (1ll << 28) || total_jobs * bpj >= (1ll << 40))
        return fail(-5, "render too large for one call: %lld pixel-block jobs", total_jobs);
    K.total_jobs = (unsigned)total_jobs; K.local_tiles = (unsigned)n_local; K.bpj = (unsigned)bpj; K.job_shift = (unsigned)job_shift;
    K.rows_shift = (unsigned)std::min(job_shift, 3);       // 4 x 1, 8 x 1, 8 x 2 pixels: whole column strips
    static const int env_rows_shift = aid_env("RTW_ROWS_SHIFT") ? atoi(aid_env("RTW_ROWS_SHIFT")) : -1;             // measurement aid: job shape
    if (env_rows_shift >= 0 && env_rows_shift <= job_shift && env_rows_shift <= 3 && job_shift - env_rows_shift <= 2) K.rows_shift = (unsigned)env_rows_shift;
    K.slot_stride = (unsigned)(sizeof(rtw::JobSlot) + 64u * (1u << job_shift));
    K.n_slots = std::min(24u, (unsigned)RTW_SLOT_BYTES / K.slot_stride);             // 24 / 12 / 7 / 4 slots of 1 / 4 / 8 / 16 pixels
    make_udiv(K.n_slots, &K.div_slots_m, &K.div_slots_s);
    make_udiv((unsigned)bpj, &K.div_bpj_m, &K.div_bpj_s);
    const long long max_useful = pool ? (total_jobs * bpj * 64 + RTW_POOL_R - 1) / RTW_POOL_R       // one item per slot of the pool
                                      : (total_jobs * bpj + 3) / 4;                                // one batch per wave, 4 waves per block
    if (grid > max_useful) grid = max_useful;
    if (grid < 1) grid = 1;

    RenderRec *rec;
    if (int rc = acquire_rec(ctx.get(), &rec)) return rc;
    *rec_out = rec;
    rec->n_spheres = scene->n; rec->n_chunks = nch; rec->grid = (int)grid; rec->block = block_threads;
    HIP_TRY(hipMemsetAsync(rec->ctr, 0, sizeof(rtw::DevCounters), stream));
    HIP_TRY(hipMemsetAsync(&rec->ctr->t_first, 0xff, sizeof(unsigned long long), stream));
    // pixels of other shards read 0 in the full-frame layout (the sum over the shards is the image)
    if (K.out_layout == 0 && p->shard_count > 1)
        HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)p->width * p->height * 3 * sizeof(T), stream));
    HIP_TRY(hipEventRecord(rec->ev0, stream));
    if (total_jobs > 0) {
        (void)hipGetLastError();           // (hipEventQuery's hipErrorNotReady in acquire_rec must not be mistaken for a launch failure)
        if (pool) hipLaunchKernelGGL(pool_kern, dim3((unsigned)grid), dim3((unsigned)block_threads), pool_lds, stream, K, C, S, (T *)d_out, rec->ctr);
        else hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, K, C, S, CS, (T *)d_out, rec->ctr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(rec->ev1, stream));
    rec->used = true; rec->done = false;
    return 0;
}

// wait for a record's kernel and add its counters to `agg`
int resolve_rec(RenderRec *r, rtw_stats_t *agg) {
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev1));
    r->done = true;
    float k_ms = 0;
    HIP_TRY(hipEventElapsedTime(&k_ms, r->ev0, r->ev1));
    rtw::DevCounters c;                    // (16 KB incl. the drain histogram)
    HIP_TRY(hipMemcpy(&c, r->ctr, sizeof c, hipMemcpyDeviceToHost));
    if (aid_env("RTW_PHASE_PROFILE") && r->block != 256) {
        const unsigned long long *pp = reinterpret_cast<const unsigned long long *>(c.end_hist + 256);
        static const char *names[6] = {"SCAN", "LM", "END", "DIEL", "REJ", "WAIT"};
        const double tot = (double)pp[26];
        fprintf(stderr, "[rtw pool profile] wave-cycles %.4g; idle %.1f%%; lost pops %llu; blocks without a candidate %.1f%% of %llu\n", tot, 100.0 * (double)pp[24] / tot, (unsigned long long)pp[25],
                pp[28] ? 100.0 * (double)pp[27] / (double)pp[28] : 0.0, (unsigned long long)pp[28]);
        for (int k = 0; k < 6; ++k)
            fprintf(stderr, "[rtw pool profile]   %-5s batches %10llu  mean fill %5.1f  %5.1f%% of wave-cycles  %7.0f cycles/batch\n", names[k], (unsigned long long)pp[4 * k],
                    pp[4 * k] ? (double)pp[4 * k + 1] / (double)pp[4 * k] : 0.0, 100.0 * (double)pp[4 * k + 2] / tot, pp[4 * k] ? (double)pp[4 * k + 2] / (double)pp[4 * k] : 0.0);
    } else if (aid_env("RTW_PHASE_PROFILE")) {
        double tot = 0;
        for (int k = 0; k < 6; ++k) tot += (double)c.phase[k];
        fprintf(stderr, "[rtw phase profile] wave-cycles: pull %.1f%%  sample+scatter finish %.1f%%  scan-pass1/level1 %.1f%%  extract/level2 %.1f%%  resolve %.1f%%  shade %.1f%%  (total %.3g)\n",
                100 * c.phase[0] / tot, 100 * c.phase[1] / tot, 100 * c.phase[2] / tot, 100 * c.phase[4] / tot,
                100 * c.phase[5] / tot, 100 * c.phase[3] / tot, tot);
        if (c.phase[7])
            fprintf(stderr, "[rtw phase profile] matrix-pipe scan: %.1f%% of the (wave, block of 32 spheres) evaluations found no candidate in any lane (%llu of %llu)\n",
                    100.0 * (double)c.phase[6] / (double)c.phase[7], (unsigned long long)c.phase[6], (unsigned long long)c.phase[7]);
    }
#ifdef RTW_CAND_HIST
    {
        static std::vector<unsigned> hh(8192);
        HIP_TRY(hipMemcpyFromSymbol(hh.data(), HIP_SYMBOL(rtw::g_cand_hist), 8192 * sizeof(unsigned)));
        unsigned long long fl = 0, tr = 0;
        for (int i = 0; i < 4096; ++i) { fl += hh[2 * i]; tr += hh[2 * i + 1]; }
        fprintf(stderr, "[rtw cand hist] cumulative: %llu candidates with discriminant < 0 (filter margin), %llu with discriminant >= 0; per segment %.4f / %.4f\n", fl, tr,
                (double)fl / (double)std::max<unsigned long long>(1, c.segments), (double)tr / (double)std::max<unsigned long long>(1, c.segments));
        fprintf(stderr, "[rtw cand hist] first spheres (false, true):");
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %d:(%u,%u)", i, hh[2 * i], hh[2 * i + 1]);
        fprintf(stderr, " ... last:");
        for (int i = std::max(0, r->n_spheres - 4); i < r->n_spheres; ++i) fprintf(stderr, " %d:(%u,%u)", i, hh[2 * i], hh[2 * i + 1]);
        fprintf(stderr, "\n");
    }
#endif
    if (c.end_hist[0] == 0xdeadbeefu) {        // (RTW_POOL_WATCHDOG builds: the pool kernel gave up; its state)
        fprintf(stderr, "[rtw pool watchdog]");
        for (int k = 1; k <= 113; ++k) fprintf(stderr, " %u", c.end_hist[k]);
        fprintf(stderr, "\n");
    }
    if (aid_env("RTW_DRAIN_PROFILE") && c.n_waves) {
        const double span = (double)(c.t_last - c.t_first) * 1e-5, mean_end = ((double)c.t_end_sum / (double)c.n_waves - (double)c.t_first) * 1e-5;
        fprintf(stderr, "[rtw drain profile] %llu waves: kernel span %.2f ms, mean wave end at %.2f ms -> %.2f ms (%.1f %%) of idle wave slots at the end of the queue\n",
                (unsigned long long)c.n_waves, span, mean_end, span - mean_end, 100.0 * (span - mean_end) / span);
        int last = 4095;
        while (last > 0 && !c.end_hist[last]) --last;
        fprintf(stderr, "[rtw drain profile] waves ending per 0.25 ms bin, last 64 bins (ending at %.2f ms):", (last + 1) * 0.25);
        for (int b = std::max(0, last - 63); b <= last; ++b) fprintf(stderr, " %u", c.end_hist[b]);
        fprintf(stderr, "\n");
    }
    agg->samples += c.samples;
    agg->segments += c.segments;
    agg->sphere_tests += c.segments * (uint64_t)r->n_spheres;
    agg->kernel_ms = std::max(agg->kernel_ms, (double)k_ms);
    agg->total_ms = std::max(agg->total_ms, (double)k_ms);
    agg->n_chunks = r->n_chunks;
    agg->grid_blocks = std::max(agg->grid_blocks, r->grid);
    agg->block_threads = std::max(agg->block_threads, r->block);
    return 0;
}

template <typename T, typename CamT>
int render_device(rtw_scene_handle scene, const CamT *cam, const rtw_params *p, void *d_out, void *stream_v) {
    DeviceGuard guard;
    if (p && p->n_devices > 1) return fail(-2, "the device-resident entry point renders on the scene's device only (n_devices = %d)", p->n_devices);
    RenderRec *rec = nullptr;
    CtxPtr ctx;
    release_last();
    int rc = launch_render<T>(scene, cam, p, d_out, (hipStream_t)stream_v, &rec, &ctx);
    if (rec) { g_last.recs.push_back(rec); g_last.ctxs.push_back(ctx); }       // (also on a late error: released by the next call)
    return rc;
}


// ---- multi-device plumbing ---------------------------------------------------------------------------------------
// Peer access between a shard's device and the gather root, enabled once per pair in both directions.  hipMemcpyPeerAsync works
// without it too (the runtime then stages through host memory by itself), but "peer copies over xGMI" is only true when the access is
// enabled -- so it is asked for explicitly, and when the platform says no the shard takes the DOCUMENTED fallback: D2H into its own
// pinned staging buffer, H2D on the root's stream (gather_path bit RTW_GATHER_HOST_STAGED in rtw_stats_t).
// Test aids (one-GPU boxes): RTW_DEBUG_REMOTE_SHARDS=1 treats every shard but the first as if it were on another device (own image
// buffer + copy into the gather buffer), RTW_DEBUG_NO_PEER=1 forces the host-staged fallback.
int ensure_peer(const CtxPtr &ctx, int dev, int root, bool *direct) {
    *direct = false;
    static const bool no_peer = aid_flag("RTW_DEBUG_NO_PEER");
    if (no_peer) return 0;
    if (dev == root) { *direct = true; return 0; }
    std::lock_guard<std::mutex> lk(ctx->mu);
    auto it = ctx->peer.find(root);
    if (it != ctx->peer.end()) { *direct = it->second; return 0; }
    int can_fwd = 0, can_back = 0;
    bool ok = hipDeviceCanAccessPeer(&can_fwd, dev, root) == hipSuccess && hipDeviceCanAccessPeer(&can_back, root, dev) == hipSuccess && can_fwd && can_back;
    if (ok) {
        const int pair[2][2] = {{dev, root}, {root, dev}};
        for (auto &pr : pair) {
            hipError_t e = hipSetDevice(pr[0]);
            if (e == hipSuccess) e = hipDeviceEnablePeerAccess(pr[1], 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            if (e != hipSuccess) { (void)hipGetLastError(); ok = false; }
        }
    } else {
        (void)hipGetLastError();
    }
    ctx->peer[root] = ok;
    *direct = ok;
    return 0;
}

// RCCL, loaded on demand (librccl.so is 570 MB: a caller that renders on one device never pays for it).  RTW_RCCL_LIB overrides the path.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
std::map<std::vector<int>, std::vector<ncclComm_t>> g_rccl_comms;      // device list -> one communicator per device (ncclCommInitAll)

int rccl_load() {       // g_rccl_mu held
    if (g_rccl.handle) return 0;
    const char *env = getenv("RTW_RCCL_LIB");
    const char *names[] = {env, "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
    void *h = nullptr;
    for (const char *nm : names) { if (nm && *nm && (h = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break; }
    if (!h) return fail(-30, "RTW_FLAG_RCCL_REDUCE: cannot load librccl (%s)", dlerror());
    RcclApi a;
    a.handle = h;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.Reduce = (decltype(a.Reduce))dlsym(h, "ncclReduce");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.Reduce || !a.GroupStart || !a.GroupEnd || !a.GetErrorString) { dlclose(h); return fail(-30, "librccl lacks a symbol this library needs"); }
    g_rccl = a;
    return 0;
}
#define NCCL_TRY(expr)                                                                        \
    do {                                                                                      \
        ncclResult_t _r = (expr);                                                             \
        if (_r != ncclSuccess) return fail(1000 + (int)_r, "%s: %s", #expr, g_rccl.GetErrorString(_r)); \
    } while (0)

// the communicators of a device list (created once, kept until rtw_shutdown)
int rccl_comms(const std::vector<int> &devs, std::vector<ncclComm_t> *out) {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (int rc = rccl_load()) return rc;
    auto it = g_rccl_comms.find(devs);
    if (it == g_rccl_comms.end()) {
        std::vector<int> sorted(devs);
        std::sort(sorted.begin(), sorted.end());
        if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
            return fail(-2, "RTW_FLAG_RCCL_REDUCE needs distinct devices (a communicator has one rank per GPU)");
        std::vector<ncclComm_t> comms(devs.size());
        NCCL_TRY(g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        it = g_rccl_comms.emplace(devs, std::move(comms)).first;
    }
    *out = it->second;
    return 0;
}

// ---- host-buffer path --------------------------------------------------------------------------------------------
HostCtx::~HostCtx() {
    if (device >= 0) HIP_IGNORE(hipSetDevice(device));
    if (scene) rtw_scene_free(scene);
    if (d_img) HIP_IGNORE(hipFree(d_img));
    if (d_aux) HIP_IGNORE(hipFree(d_aux));
    if (h_stage) HIP_IGNORE(hipHostFree(h_stage));
    if (done_ev) HIP_IGNORE(hipEventDestroy(done_ev));
    if (stream) HIP_IGNORE(hipStreamDestroy(stream));
}

template <typename SceneT>
void scene_key_of(const SceneT *s, bool f64, std::vector<unsigned char> &key) {
    using T = typename std::remove_cv<typename std::remove_pointer<decltype(s->cx)>::type>::type;
    const size_t n = (size_t)(s->n > 0 ? s->n : 0);
    key.clear();
    key.reserve(16 + n * (8 * sizeof(T) + sizeof(int32_t)));
    auto put = [&](const void *p, size_t b) { const unsigned char *q = (const unsigned char *)p; key.insert(key.end(), q, q + b); };
    const int32_t head[2] = {f64 ? 1 : 0, s->n};
    put(head, sizeof head);
    if (n == 0) return;
    const T *arrs[8] = {s->cx, s->cy, s->cz, s->r, s->ar, s->ag, s->ab, s->param};
    for (const T *a : arrs) put(a, n * sizeof(T));
    put(s->kind, n * sizeof(int32_t));
}

// the caller holds a HostCtx exclusively between acquire and release
struct HostLease {
    CtxPtr ctx;
    HostCtx *hc = nullptr;
    bool pooled = false;
    ~HostLease() {
        if (!hc) return;
        if (pooled) { std::lock_guard<std::mutex> lk(ctx->mu); hc->busy = false; }
        else delete hc;                               // more concurrent host renders than pool entries: a temporary
    }
};

int acquire_host(int device, const std::vector<unsigned char> &key, HostLease *out) {
    int dev;
    if (int rc = resolve_device(device, &dev)) return rc;
    CtxPtr ctx;
    if (int rc = get_ctx(dev, &ctx)) return rc;
    out->ctx = ctx;
    {
        std::lock_guard<std::mutex> lk(ctx->mu);
        HostCtx *pick = nullptr;
        for (auto &h : ctx->host) if (!h->busy && h->scene_key == key) { pick = h.get(); break; }     // same scene: nothing to upload
        if (!pick && ctx->host.size() < RTW_HOST_CTX_POOL) {
            // a scene this device has not seen (or whose entries are all busy): a NEW entry while the pool has room -- a caller that
            // alternates between two scenes keeps both uploaded
            ctx->host.emplace_back(new HostCtx());
            pick = ctx->host.back().get();
            pick->device = dev;
        }
        if (!pick)                                                                                  // pool full: the least recently used idle entry
            for (auto &h : ctx->host) if (!h->busy && (!pick || h->last_use < pick->last_use)) pick = h.get();
        if (pick) { pick->busy = true; pick->last_use = ++ctx->use_clock; out->hc = pick; out->pooled = true; }
    }
    if (!out->hc) { out->hc = new HostCtx(); out->hc->device = dev; out->pooled = false; }
    HostCtx *hc = out->hc;
    HIP_TRY(hipSetDevice(dev));
    if (!hc->stream) HIP_TRY(hipStreamCreateWithFlags(&hc->stream, hipStreamNonBlocking));
    if (!hc->done_ev) HIP_TRY(hipEventCreateWithFlags(&hc->done_ev, hipEventDisableTiming));
    return 0;
}

template <typename T, typename SceneT>
int ensure_scene(HostCtx *hc, const SceneT *scene, const std::vector<unsigned char> &key) {
    if (hc->scene && hc->scene_key == key) return 0;
    if (hc->scene) { rtw_scene_free(hc->scene); hc->scene = nullptr; hc->scene_key.clear(); }
    if (int rc = upload_scene<T>(scene, hc->device, &hc->scene)) return rc;
    hc->scene_key = key;
    return 0;
}

int ensure_dev(void **p, size_t *cap, size_t bytes) {
    if (*cap >= bytes && *p) return 0;
    if (*p) { HIP_IGNORE(hipFree(*p)); *p = nullptr; *cap = 0; }
    HIP_TRY(hipMalloc(p, bytes));
    *cap = bytes;
    return 0;
}

// Device image -> the caller's (pageable) buffer: ONE hipMemcpyAsync on the context's stream.  Measured on the MI355X box
// (tools/ubench_d2h.hip, 24.9 MB): straight into pageable memory 0.45 ms -- as fast as into pinned memory -- against 1.28 ms
// through a pinned staging buffer + memcpy and 0.6 - 1.0 ms for chunked staging overlapped with 1 - 4 memcpy threads.
int copy_out(HostCtx *hc, const void *d_src, void *out, size_t bytes) {
    if (bytes == 0) return 0;
    HIP_TRY(hipMemcpyAsync(out, d_src, bytes, hipMemcpyDeviceToHost, hc->stream));
    HIP_TRY(hipStreamSynchronize(hc->stream));
    return 0;
}

// compact tile-major shards (shard r at r * pad_tiles tiles) -> the column-major frame (multi-device root)
template <typename T>
__global__ void untile_kernel(const T *__restrict__ gather, T *__restrict__ frame, int W, int H, int tiles_i, long n_tiles, int n_shards, long pad_tiles) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;         // (tile, slot)
    const long t = g >> 6;
    if (t >= n_tiles) return;
    const int slot = (int)(g & 63);
    const long r = t % n_shards, k = t / n_shards;
    const int tj = (int)(t / tiles_i), ti = (int)(t % tiles_i);
    const int i0 = ti * 8 + (slot & 7), j0 = tj * 8 + (slot >> 3);
    if (i0 >= H || j0 >= W) return;
    const T *src = gather + ((r * pad_tiles + k) * 64 + slot) * 3;
    T *dst = frame + ((size_t)j0 * H + i0) * 3;
    dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2];
}

template <typename T, typename SceneT, typename CamT>
int render_host(const SceneT *scene, const CamT *cam, const rtw_params *p, T *out) {
    if (!scene || !cam || !p || !out) return fail(-1, "null argument");
    int nch, cs;
    if (int rc = validate_params(p, &nch, &cs)) return rc;
    DeviceGuard guard;
    release_last();
    // the device list (n_devices / device_ids; Julia keyword devices=:all)
    std::vector<int> devs;
    if (p->n_devices == -1) {
        int n = 0;
        const hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0)
            return fail(e != hipSuccess ? (int)e : -21, "no HIP device available (%s); librtw_hip has no CPU fallback",
                        e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
        for (int d = 0; d < n; ++d) devs.push_back(d);
    } else if (p->n_devices > 1 || (p->n_devices == 1 && p->device_ids)) {
        if (!p->device_ids) return fail(-1, "n_devices = %d but device_ids is null", p->n_devices);
        devs.assign(p->device_ids, p->device_ids + p->n_devices);
    } else if (p->n_devices < -1) {
        return fail(-2, "bad n_devices %d", p->n_devices);
    }
    if (s_has_bad_scene(scene)) return fail(-1, "null scene array");
    std::vector<unsigned char> key;
    scene_key_of(scene, sizeof(T) == 8, key);

    if (devs.size() <= 1 && !((p->flags & RTW_FLAG_RCCL_REDUCE) && devs.size() == 1)) {
        // ---- one device: render into the cached device image, one D2H ----
        HostLease L;
        if (int rc = acquire_host(devs.size() == 1 ? devs[0] : p->device, key, &L)) return rc;
        HostCtx *hc = L.hc;
        if (int rc = ensure_scene<T>(hc, scene, key)) return rc;
        rtw_params q = *p;
        q.device = hc->device; q.n_devices = 0; q.device_ids = nullptr;
        q.flags &= ~RTW_FLAG_RCCL_REDUCE;              // (one device: nothing to reduce)
        const bool compact = (q.flags & RTW_FLAG_COMPACT_TILES) != 0;
        const size_t elems = compact ? (size_t)local_tiles(&q) * 64 * 3 : (size_t)q.width * (size_t)q.height * 3;
        if (elems == 0) { g_last.resolved = true; return 0; }
        if (int rc = ensure_dev(&hc->d_img, &hc->d_cap, elems * sizeof(T))) return rc;
        RenderRec *rec = nullptr;
        CtxPtr rctx;
        int rc = launch_render<T>(hc->scene, cam, &q, hc->d_img, hc->stream, &rec, &rctx);
        if (!rc) rc = copy_out(hc, hc->d_img, out, elems * sizeof(T));
        if (rc) (void)hipStreamSynchronize(hc->stream);           // nothing of this call may still be in flight when the lease ends
        if (!rc) rc = resolve_rec(rec, &g_last.agg);
        if (rec) release_rec(rctx, rec, rc == 0);
        g_last.resolved = rc == 0;
        return rc;
    }

    // ---- several devices: shard r renders tiles t = r (mod N) on its own device and stream, then ONE of
    //   (default)             compact tile-major shards gathered in HBM of the first device -- peer copies over xGMI with peer access
    //                         enabled per pair (ensure_peer; host-staged fallback when the platform refuses it); a shard on the root
    //                         device renders straight into the gather buffer --, un-tiled there by one small kernel;
    //   RTW_FLAG_RCCL_REDUCE  zero-padded full frames summed onto the first device by ONE ncclReduce over xGMI (BASELINE configs[3]'s
    //                         "RCCL reduce of per-tile framebuffers": x + 0 == x, so the sum is the image bit for bit);
    //      and the frame crosses PCIe once ----
    if (p->shard_count != 1) return fail(-2, "n_devices > 1 cannot be combined with shard_index/shard_count");
    if (p->flags & RTW_FLAG_COMPACT_TILES) return fail(-2, "n_devices > 1 writes the full frame (RTW_FLAG_COMPACT_TILES is a per-shard layout)");
    const int N = (int)devs.size();
    const bool use_rccl = (p->flags & RTW_FLAG_RCCL_REDUCE) != 0;
    static const bool dbg_remote = aid_flag("RTW_DEBUG_REMOTE_SHARDS");
    std::vector<ncclComm_t> comms;
    if (use_rccl) { if (int rc = rccl_comms(devs, &comms)) return rc; }
    std::vector<HostLease> L(N);
    for (int r = 0; r < N; ++r)
        if (int rc = acquire_host(devs[r], key, &L[r])) return fail(rc, "device %d (shard %d of %d): %s", devs[r], r, N, std::string(g_err).c_str());
    {   // scene uploads of the devices that do not have it yet, in parallel (a cache miss on 8 devices is 8 x (kd split + a dozen copies))
        std::vector<int> up_rc(N, 0);
        std::vector<std::string> up_err(N);
        std::vector<std::thread> th;
        for (int r = 0; r < N; ++r) {
            if (L[r].hc->scene && L[r].hc->scene_key == key) continue;
            th.emplace_back([&, r] { up_rc[r] = ensure_scene<T>(L[r].hc, scene, key); if (up_rc[r]) up_err[r] = g_err; });
        }
        for (auto &t : th) t.join();
        for (int r = 0; r < N; ++r)
            if (up_rc[r]) return fail(up_rc[r], "device %d (shard %d of %d): %s", devs[r], r, N, up_err[r].c_str());
    }
    HostCtx *root = L[0].hc;
    rtw_params q0 = *p;
    q0.shard_index = 0; q0.shard_count = N;
    const long pad_tiles = local_tiles(&q0);                               // shard 0 owns the most tiles
    const size_t shard_bytes = (size_t)pad_tiles * 192 * sizeof(T), frame_bytes = (size_t)p->width * p->height * 3 * sizeof(T);
    HIP_TRY(hipSetDevice(root->device));
    if (!use_rccl) { if (int rc = ensure_dev(&root->d_aux, &root->aux_cap, shard_bytes * N)) return rc; }
    if (int rc = ensure_dev(&root->d_img, &root->d_cap, frame_bytes)) return rc;
    std::vector<RenderRec *> recs(N, nullptr);
    std::vector<CtxPtr> rctx(N);
    int rc = 0, gather_path = 0;
    for (int r = 0; r < N && !rc; ++r) {
        HostCtx *hc = L[r].hc;
        rtw_params q = *p;
        q.device = hc->device; q.shard_index = r; q.shard_count = N; q.n_devices = 0; q.device_ids = nullptr;
        q.flags &= ~RTW_FLAG_RCCL_REDUCE;
        const hipError_t e0 = hipSetDevice(hc->device);
        if (e0 != hipSuccess) { rc = fail((int)e0, "hipSetDevice(%d): %s", hc->device, hipGetErrorString(e0)); break; }
        if (use_rccl) {
            // this shard's tiles in the full-frame layout, zero elsewhere (launch_render clears the frame of a sharded render first)
            if ((rc = ensure_dev(&hc->d_img, &hc->d_cap, frame_bytes))) break;
            rc = launch_render<T>(hc->scene, cam, &q, hc->d_img, hc->stream, &recs[r], &rctx[r]);
            continue;
        }
        q.flags |= RTW_FLAG_COMPACT_TILES;
        const size_t my_bytes = (size_t)local_tiles(&q) * 192 * sizeof(T);
        char *slot = (char *)root->d_aux + (size_t)r * shard_bytes;
        if (my_bytes == 0) continue;
        const bool remote = hc->device != root->device || (dbg_remote && hc != root);
        void *d_out = slot;
        if (remote) {
            if ((rc = ensure_dev(&hc->d_img, &hc->d_cap, my_bytes))) break;        // (the shard buffer belongs to ITS device)
            d_out = hc->d_img;
        }
        if ((rc = launch_render<T>(hc->scene, cam, &q, d_out, hc->stream, &recs[r], &rctx[r]))) break;
        hipError_t e = hipSuccess;
        bool staged = false;
        if (remote) {
            bool direct = false;
            if ((rc = ensure_peer(L[r].ctx, hc->device, root->device, &direct))) break;
            e = hipSetDevice(hc->device);                 // (ensure_peer switches devices while it enables the access)
            if (e != hipSuccess) { rc = fail((int)e, "hipSetDevice(%d): %s", hc->device, hipGetErrorString(e)); break; }
            if (direct) {
                gather_path |= RTW_GATHER_PEER;
                e = hipSetDevice(hc->device);
                if (e == hipSuccess) e = hipMemcpyPeerAsync(slot, root->device, d_out, hc->device, my_bytes, hc->stream);
            } else {
                // the documented fallback: through this shard's pinned staging buffer
                gather_path |= RTW_GATHER_HOST_STAGED;
                staged = true;
                if (hc->stage_cap < my_bytes) {
                    if (hc->h_stage) { HIP_IGNORE(hipHostFree(hc->h_stage)); hc->h_stage = nullptr; hc->stage_cap = 0; }
                    e = hipHostMalloc(&hc->h_stage, my_bytes, hipHostMallocDefault);
                    if (e == hipSuccess) hc->stage_cap = my_bytes;
                }
                if (e == hipSuccess) e = hipMemcpyAsync(hc->h_stage, d_out, my_bytes, hipMemcpyDeviceToHost, hc->stream);
            }
        } else if (hc != root) {
            gather_path |= RTW_GATHER_SAME_DEVICE;
        }
        if (e == hipSuccess) e = hipEventRecord(hc->done_ev, hc->stream);
        if (e == hipSuccess && hc != root) {
            e = hipSetDevice(root->device);
            if (e == hipSuccess) e = hipStreamWaitEvent(root->stream, hc->done_ev, 0);
            if (e == hipSuccess && staged) e = hipMemcpyAsync(slot, hc->h_stage, my_bytes, hipMemcpyHostToDevice, root->stream);
        }
        if (e != hipSuccess) rc = fail((int)e, "device %d (shard %d of %d): gather failed: %s", hc->device, r, N, hipGetErrorString(e));
    }
    if (!rc && use_rccl) {
        gather_path |= RTW_GATHER_RCCL;
        const ncclDataType_t dt = sizeof(T) == 8 ? ncclFloat64 : ncclFloat32;
        const size_t count = (size_t)p->width * p->height * 3;
        ncclResult_t nr = g_rccl.GroupStart();
        for (int r = 0; r < N && nr == ncclSuccess; ++r)
            nr = g_rccl.Reduce(L[r].hc->d_img, root->d_img, count, dt, ncclSum, 0, comms[r], L[r].hc->stream);
        const ncclResult_t ne = g_rccl.GroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) rc = fail(1000 + (int)nr, "ncclReduce over %d devices: %s", N, g_rccl.GetErrorString(nr));
        if (!rc) { const hipError_t e = hipSetDevice(root->device); if (e != hipSuccess) rc = fail((int)e, "hipSetDevice: %s", hipGetErrorString(e)); }
        if (!rc) rc = copy_out(root, root->d_img, out, frame_bytes);
    } else if (!rc) {
        hipError_t e = hipSetDevice(root->device);
        const long n_tiles = (long)((p->height + 7) / 8) * ((p->width + 7) / 8);
        if (e == hipSuccess && n_tiles > 0) {
            (void)hipGetLastError();
            hipLaunchKernelGGL(untile_kernel<T>, dim3((unsigned)((n_tiles * 64 + 255) / 256)), dim3(256), 0, root->stream,
                               (const T *)root->d_aux, (T *)root->d_img, p->width, p->height, (p->height + 7) / 8, n_tiles, N, pad_tiles);
            e = hipGetLastError();
        }
        if (e != hipSuccess) rc = fail((int)e, "un-tile kernel: %s", hipGetErrorString(e));
        if (!rc) rc = copy_out(root, root->d_img, out, frame_bytes);
    }
    // every stream of this call drains before the leases end, also on an error
    for (int r = 0; r < N; ++r) { HIP_IGNORE(hipSetDevice(L[r].hc->device)); HIP_IGNORE(hipStreamSynchronize(L[r].hc->stream)); }
    rtw_stats_t &a = g_last.agg;                       // sums over the devices; times: the maximum
    for (int r = 0; r < N; ++r) {
        if (!recs[r]) continue;
        rtw_stats_t st;
        memset(&st, 0, sizeof st);
        if (!rc) rc = resolve_rec(recs[r], &st);
        release_rec(rctx[r], recs[r], rc == 0);
        a.samples += st.samples; a.segments += st.segments; a.sphere_tests += st.sphere_tests;
        a.kernel_ms = std::max(a.kernel_ms, st.kernel_ms); a.total_ms = std::max(a.total_ms, st.total_ms);
        a.n_chunks = st.n_chunks; a.grid_blocks = std::max(a.grid_blocks, st.grid_blocks); a.block_threads = std::max(a.block_threads, st.block_threads);
    }
    a.gather_path = gather_path;
    g_last.resolved = rc == 0;
    return rc;
}

// T0 unit entry point: host slots -> device -> unit_kernel -> host slots
template <typename T, typename SceneT, typename CamT>
int run_unit(int op_arg, int count, const void *in, void *out, const SceneT *scene, const CamT *cam) {
    const int op = op_arg & 0xff, numerics = (op_arg >> 8) & 3;      // bits 8-9: the numerics mode of the ray-sphere test
    if (op_arg < 0 || (op_arg >> 10) != 0 || op >= rtw::U_NUM_OPS) return fail(-2, "unknown unit op %d", op_arg);
    if (numerics > rtw::NUM_REFERENCE_FMA) return fail(-2, "unknown numerics mode %d (unit op %d)", numerics, op_arg);
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
        if (int rc = upload_scene<T>(scene, dev, &h_raw)) return rc;
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

}  // namespace

extern "C" {

int rtw_abi_version(void) { return RTW_ABI_VERSION; }

int rtw_device_count(int *count) {
    if (!count) return fail(-1, "null argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail((int)e, "hipGetDeviceCount failed: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

const char *rtw_last_error(void) { return g_err; }

int rtw_scene_upload_f32(const rtw_scene_f32 *s, int device, rtw_scene_handle *out) { return upload_scene<float>(s, device, out); }
int rtw_scene_upload_f64(const rtw_scene_f64 *s, int device, rtw_scene_handle *out) { return upload_scene<double>(s, device, out); }

int rtw_scene_free(rtw_scene_handle h) {
    if (!h) return 0;
    DeviceGuard guard;
    HIP_IGNORE(hipSetDevice(h->device));
    void *ptrs[] = {h->geom, h->mat0, h->mat1, h->scan, h->mf_ops, h->c_mf_ops, h->c_mf_box, h->c_bound, h->c_exact, h->c_mat0, h->c_mat1, h->c_orig};
    for (void *q : ptrs) if (q) HIP_IGNORE(hipFree(q));
    delete h;
    return 0;
}

int rtw_render_device_f32(rtw_scene_handle s, const rtw_camera_f32 *c, const rtw_params *p, void *d_out, void *stream) {
    return render_device<float>(s, c, p, d_out, stream);
}
int rtw_render_device_f64(rtw_scene_handle s, const rtw_camera_f64 *c, const rtw_params *p, void *d_out, void *stream) {
    return render_device<double>(s, c, p, d_out, stream);
}
int rtw_render_f32(const rtw_scene_f32 *s, const rtw_camera_f32 *c, const rtw_params *p, float *out) {
    return render_host<float>(s, c, p, out);
}
int rtw_render_f64(const rtw_scene_f64 *s, const rtw_camera_f64 *c, const rtw_params *p, double *out) {
    return render_host<double>(s, c, p, out);
}

int rtw_stats(rtw_stats_t *out) {
    if (!out) return fail(-1, "null argument");
    if (g_last.generation != g_generation.load() || (g_last.recs.empty() && !g_last.resolved))
        return fail(-6, "no render has been issued from this thread");
    DeviceGuard guard;
    if (!g_last.resolved) {
        memset(&g_last.agg, 0, sizeof g_last.agg);
        for (RenderRec *r : g_last.recs)
            if (int rc = resolve_rec(r, &g_last.agg)) return rc;
        g_last.resolved = true;
    }
    *out = g_last.agg;
    return 0;
}

int rtw_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene, const rtw_camera_f32 *cam) {
    return run_unit<float>(op, count, in, out, scene, cam);
}
int rtw_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene, const rtw_camera_f64 *cam) {
    return run_unit<double>(op, count, in, out, scene, cam);
}

int rtw_shutdown(void) {
    DeviceGuard guard;
    std::lock_guard<std::mutex> lk(g_mu);
    g_generation.fetch_add(1);                 // every thread's "last render" is now stale (rtw_stats reports -6)
    for (auto &c : g_ctx) {
        HIP_IGNORE(hipSetDevice(c->device));
        std::lock_guard<std::mutex> lk2(c->mu);
        // Records that some thread still references (`owned`: a host render in flight on another thread, or a thread's last render)
        // are NOT destroyed here: that thread holds a CtxPtr, the DeviceCtx -- and with it these records -- lives until it lets go.
        c->recs.erase(std::remove_if(c->recs.begin(), c->recs.end(), [](const std::unique_ptr<RenderRec> &r) { return !r->owned; }), c->recs.end());
        // (host contexts in use by a render in flight on another thread stay alive with their DeviceCtx in the same way)
        c->host.erase(std::remove_if(c->host.begin(), c->host.end(), [](const std::unique_ptr<HostCtx> &h) { return !h->busy; }), c->host.end());
    }
    g_ctx.clear();
    {
        std::lock_guard<std::mutex> lr(g_rccl_mu);
        for (auto &kv : g_rccl_comms)
            for (ncclComm_t cm : kv.second) (void)g_rccl.CommDestroy(cm);
        g_rccl_comms.clear();
    }
    return 0;
}

}  // extern "C"
