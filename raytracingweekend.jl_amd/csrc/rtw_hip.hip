// rtw_hip.hip -- C ABI of librtw_hip.so (include/rtw_hip.h) over the gfx950 kernels.
// Replaces /root/reference/src/render.jl:8-44 behind a ccall-able boundary.  No CPU fallback:
// every compute entry point needs a HIP device and reports an error otherwise.
#include "../../include/rtw_hip.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "rtw_kernels.hpp"
#include "rtw_units.hpp"

namespace {

thread_local char g_err[512] = "";

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

// ---- per-device context: cached workspace, counters, events ---------------------------------
struct DeviceCtx {
    int device = -1;
    double *partial = nullptr;
    size_t partial_bytes = 0;
    void *puv = nullptr;
    size_t puv_bytes = 0;
    rtw::DevCounters *ctr = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    int num_cus = 0;
    // last render
    bool pending = false;
    uint64_t last_samples_expected = 0;
    int last_n = 0, last_chunks = 0, last_grid = 0, last_block = 0;
};

std::mutex g_mu;
std::vector<DeviceCtx *> g_ctx;
thread_local DeviceCtx *g_last = nullptr;

int get_ctx(int device, DeviceCtx **out) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (DeviceCtx *c : g_ctx)
        if (c->device == device) { *out = c; return 0; }
    DeviceCtx *c = new DeviceCtx();
    c->device = device;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(-20, "device %d is %s; librtw_hip is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    c->num_cus = prop.multiProcessorCount;
    HIP_TRY(hipMalloc(&c->ctr, sizeof(rtw::DevCounters)));
    HIP_TRY(hipEventCreate(&c->ev0));
    HIP_TRY(hipEventCreate(&c->ev1));
    HIP_TRY(hipEventCreate(&c->ev2));
    g_ctx.push_back(c);
    *out = c;
    return 0;
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
    // opt-in group-cull mode (RTW_FLAG_GROUP_CULL): cluster-major copies
    void *c_bound, *c_exact, *c_mat0, *c_mat1;
    unsigned short *c_orig;
    int c_groups_pad, c_big;
    double c_cs[3], c_rs;
};

namespace {

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
    int half = ((cnt / 2 + RTW_CULL_GS - 1) / RTW_CULL_GS) * RTW_CULL_GS;      // left part: whole clusters
    if (half >= cnt) half = cnt - 1;
    auto key = [&](int i) { return ax == 0 ? (double)s->cx[i] : ax == 1 ? (double)s->cy[i] : (double)s->cz[i]; };
    std::nth_element(ids.begin() + lo, ids.begin() + lo + half, ids.begin() + hi, [&](int a, int b) { return key(a) < key(b); });
    kd_split(s, ids, lo, lo + half, groups);
    kd_split(s, ids, lo + half, hi, groups);
}

// cluster-major arrays for the opt-in group-cull scan (rtw_device.hpp, "opt-in accelerated scan")
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
    const int n_exact = ng_pad * GS + n_big;
    if (n_exact >= 65536) return fail(-5, "too many spheres (%d) for the group-cull layout", n);
    std::vector<T> box((size_t)(ng_pad + RTW_CULL_BG) * 8);            // + one prefetch group
    std::vector<V4> exact(std::max(n_exact, 1)), mat0(std::max(n_exact, 1)), mat1(std::max(n_exact, 1));
    std::vector<unsigned short> orig(std::max(n_exact, 1), 0);
    // dead cluster: far away and empty; dead sphere: r^2 = -1e30 (never a candidate)
    for (auto &b : box) b = (T)1e15;                                    // dead cluster: a point far away
    for (int k = 0; k < n_exact; ++k) { exact[k] = V4{(T)0, (T)0, (T)0, (T)-1e30}; mat0[k] = V4{(T)1, (T)0, (T)0, (T)0}; mat1[k] = V4{(T)0, (T)0, (T)0, (T)0}; }
    auto put = [&](int k, int i) {
        exact[k] = V4{s->cx[i], s->cy[i], s->cz[i], s->r[i] * s->r[i]};
        mat0[k] = V4{s->r[i], s->param[i], (T)s->kind[i], (T)0};
        mat1[k] = V4{s->ar[i], s->ag[i], s->ab[i], (T)0};
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
    return C;
}

template <typename T, typename SceneT>
int upload_scene(const SceneT *s, int device, rtw_scene_handle *out) {
    if (!s || !out) return fail(-1, "null argument");
    if (s->n < 0) return fail(-2, "scene.n < 0");
    if (s->n > 0 && (!s->cx || !s->cy || !s->cz || !s->r || !s->kind || !s->ar || !s->ag || !s->ab || !s->param))
        return fail(-1, "null scene array");
    for (int i = 0; i < s->n; ++i) {
        if (s->kind[i] < 0 || s->kind[i] > 2) return fail(-3, "sphere %d: unknown material kind %d", i, s->kind[i]);
    }
    int dev;
    if (int rc = resolve_device(device, &dev)) return rc;
    DeviceCtx *ctx;
    if (int rc = get_ctx(dev, &ctx)) return rc;
    HIP_TRY(hipSetDevice(dev));
    using V4 = typename rtw::Vec4<T>::type;
    const int n = s->n;
    constexpr int pair = 2 * rtw::ScanGroup<T>::N;                                       // 16 (f32) / 8 (f64)
    const int n_pad = ((n + pair - 1) / pair) * pair;                                    // 0 spheres: no scan at all
    const int n_alloc = n_pad + RTW_SPHERE_TAIL;                                          // prefetch tail group
    if (n_pad >= 65536) return fail(-5, "too many spheres (%d): candidate lists hold 16-bit indices", n);
    std::vector<V4> geom(n_alloc), mat0(n_alloc), mat1(n_alloc);
    for (int i = 0; i < n_alloc; ++i) {
        if (i < n) {
            geom[i] = V4{s->cx[i], s->cy[i], s->cz[i], s->r[i] * s->r[i]};  // r^2: src/hit.jl:17
            mat0[i] = V4{s->r[i], s->param[i], (T)s->kind[i], (T)0};
            mat1[i] = V4{s->ar[i], s->ag[i], s->ab[i], (T)0};
        } else {
            // padding sphere that can never be hit: r^2 hugely negative => disc < 0 always
            geom[i] = V4{(T)0, (T)0, (T)0, (T)-1e30};
            mat0[i] = V4{(T)1, (T)0, (T)0, (T)0};
            mat1[i] = V4{(T)0, (T)0, (T)0, (T)0};
        }
    }
    rtw_scene_dev *h = new rtw_scene_dev();
    h->device = dev; h->is_f64 = sizeof(T) == 8; h->n = n; h->n_pad = n_pad;
    h->geom = h->mat0 = h->mat1 = nullptr;
    h->c_bound = h->c_exact = h->c_mat0 = h->c_mat1 = nullptr; h->c_orig = nullptr;
    const size_t bytes = sizeof(V4) * (size_t)n_alloc;
    HIP_TRY(hipMalloc(&h->geom, bytes));
    HIP_TRY(hipMalloc(&h->mat0, bytes));
    HIP_TRY(hipMalloc(&h->mat1, bytes));
    HIP_TRY(hipMemcpy(h->geom, geom.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->mat0, mat0.data(), bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->mat1, mat1.data(), bytes, hipMemcpyHostToDevice));
    if (int rc = build_cull<T>(s, h)) { rtw_scene_free(h); return rc; }
    *out = h;
    return 0;
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
    if (p->max_depth < 0) return fail(-2, "max_depth must be >= 0");
    if (p->shard_count <= 0 || p->shard_index < 0 || p->shard_index >= p->shard_count)
        return fail(-2, "bad shard %d of %d", p->shard_index, p->shard_count);
    if (p->n_chunks < 0) return fail(-2, "n_chunks must be >= 0");
    if (p->flags & ~RTW_FLAG_GROUP_CULL) return fail(-2, "unknown flags 0x%x", p->flags);
    int nch = p->n_chunks > 0 ? p->n_chunks : (p->spp < 128 ? p->spp : 128);
    if (nch > p->spp) nch = p->spp;
    int cs = (p->spp + nch - 1) / nch;
    *chunk_spp = cs;
    *n_chunks = (p->spp + cs - 1) / cs;
    return 0;
}

template <typename T, typename CamT>
int render_device(rtw_scene_handle scene, const CamT *cam, const rtw_params *p, void *d_out, void *stream_v) {
    if (!scene || !cam || !d_out) return fail(-1, "null argument");
    if (scene->is_f64 != (sizeof(T) == 8)) return fail(-4, "scene handle precision does not match the call");
    int nch, cs;
    if (int rc = validate_params(p, &nch, &cs)) return rc;
    if (p->device >= 0 && p->device != scene->device)
        return fail(-4, "params.device %d != scene device %d", p->device, scene->device);
    DeviceCtx *ctx;
    if (int rc = get_ctx(scene->device, &ctx)) return rc;
    HIP_TRY(hipSetDevice(scene->device));
    hipStream_t stream = (hipStream_t)stream_v;

    rtw::KParams K;
    K.width = p->width; K.height = p->height; K.spp = p->spp; K.max_depth = p->max_depth;
    K.seed = p->seed; K.n_chunks = nch; K.chunk_spp = cs;
    K.shard_index = p->shard_index; K.shard_count = p->shard_count;
    K.tiles_i = (p->height + 7) / 8; K.tiles_j = (p->width + 7) / 8;
    const long long n_tiles = (long long)K.tiles_i * K.tiles_j;
    const long long n_local = n_tiles > p->shard_index ? (n_tiles - p->shard_index + p->shard_count - 1) / p->shard_count : 0;
    const long long total_items = n_local * nch * 64;
    if (total_items >= (1ll << 31)) return fail(-5, "render too large for one call: %lld work items", total_items);
    K.n_local_tiles = (int)n_local; K.total_items = (unsigned)total_items; K.gamma = p->gamma;
    make_udiv((unsigned)nch, &K.div_chunks_m, &K.div_chunks_s);
    make_udiv((unsigned)K.tiles_i, &K.div_tiles_m, &K.div_tiles_s);

    // per-column / per-row (u, v) of src/render.jl:26-27: T(j / W) and T((H - i) / H), Float64 division
    // then conversion, evaluated here on the host (exactly the arithmetic the reference performs)
    {
        std::vector<T> puv((size_t)p->width + p->height);
        for (int j0 = 0; j0 < p->width; ++j0) puv[j0] = (T)((double)(j0 + 1) / (double)p->width);
        for (int i0 = 0; i0 < p->height; ++i0) puv[(size_t)p->width + i0] = (T)((double)(p->height - (i0 + 1)) / (double)p->height);
        const size_t pb = puv.size() * sizeof(T);
        if (ctx->puv_bytes < pb) {
            if (ctx->puv) { HIP_TRY(hipFree(ctx->puv)); ctx->puv = nullptr; ctx->puv_bytes = 0; }
            HIP_TRY(hipMalloc(&ctx->puv, pb));
            ctx->puv_bytes = pb;
        }
        HIP_TRY(hipMemcpyAsync(ctx->puv, puv.data(), pb, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));          // puv is a stack-local host buffer
    }

    const size_t need = (size_t)(total_items > 0 ? total_items : 1) * 3 * sizeof(double);
    if (ctx->partial_bytes < need) {
        if (ctx->partial) { HIP_TRY(hipFree(ctx->partial)); ctx->partial = nullptr; ctx->partial_bytes = 0; }
        HIP_TRY(hipMalloc(&ctx->partial, need));
        ctx->partial_bytes = need;
    }

    rtw::Camera<T> C;
    for (int k = 0; k < 3; ++k) {
        C.origin[k] = cam->origin[k]; C.llc[k] = cam->lower_left_corner[k];
        C.horizontal[k] = cam->horizontal[k]; C.vertical[k] = cam->vertical[k];
        C.u[k] = cam->u[k]; C.v[k] = cam->v[k]; C.w[k] = cam->w[k];
    }
    C.lens_radius = cam->lens_radius;
    rtw::DevScene<T> S;
    using V4 = typename rtw::Vec4<T>::type;
    S.geom = (const V4 *)scene->geom; S.mat0 = (const V4 *)scene->mat0; S.mat1 = (const V4 *)scene->mat1;
    S.n = scene->n; S.n_pad = scene->n_pad;

    // persistent grid: enough 256-thread blocks to fill every CU at the kernel's occupancy
    static const bool phase_profile = getenv("RTW_PHASE_PROFILE") != nullptr;   // debugging aid, not for timed runs
    const size_t list_bytes = (size_t)RTW_LIST_CAP * 256 * sizeof(unsigned short);
    const bool cull = (p->flags & RTW_FLAG_GROUP_CULL) != 0;
    const rtw::CullScene<T> CS = cull_scene_of<T>(scene);
    const size_t n_cull = (size_t)rtw::cull_exact_count(CS);
    const size_t geom_bytes = cull ? n_cull * sizeof(V4) + ((n_cull * sizeof(unsigned short) + 15) / 16) * 16
                                   : (size_t)(scene->n_pad + RTW_SPHERE_TAIL) * sizeof(V4);
    const bool lds_scene = geom_bytes <= RTW_LDS_SCENE_MAX_BYTES;
    const size_t lds_bytes = list_bytes + (lds_scene ? geom_bytes : 0);
    typedef void (*kern_t)(rtw::KParams, rtw::Camera<T>, rtw::DevScene<T>, rtw::CullScene<T>, const T *, double *, rtw::DevCounters *);
    kern_t kern;
    if (cull && phase_profile) kern = (kern_t)rtw::trace_kernel<T, true, true, true>;      // profiling aid: LDS path only
    else if (cull) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, true> : (kern_t)rtw::trace_kernel<T, false, false, true>;
    else if (phase_profile) kern = lds_scene ? (kern_t)rtw::trace_kernel<T, true, true, false> : (kern_t)rtw::trace_kernel<T, true, false, false>;
    else kern = lds_scene ? (kern_t)rtw::trace_kernel<T, false, true, false> : (kern_t)rtw::trace_kernel<T, false, false, false>;
    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, 256, lds_bytes));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    long long grid = (long long)ctx->num_cus * blocks_per_cu;
    const long long max_useful = (total_items + 255) / 256;
    if (grid > max_useful) grid = max_useful;
    if (grid < 1) grid = 1;

    HIP_TRY(hipMemsetAsync(ctx->ctr, 0, sizeof(rtw::DevCounters), stream));
    HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)p->width * p->height * 3 * sizeof(T), stream));
    HIP_TRY(hipEventRecord(ctx->ev0, stream));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, K, C, S, CS, (const T *)ctx->puv, ctx->partial, ctx->ctr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(ctx->ev1, stream));
    const unsigned n_pix_local = (unsigned)n_local * 64u;
    if (n_pix_local > 0) {
        hipLaunchKernelGGL(rtw::finalize_kernel<T>, dim3((n_pix_local + 255) / 256), dim3(256), 0, stream, K,
                           (const double *)ctx->partial, (T *)d_out);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(ctx->ev2, stream));
    ctx->pending = true;
    ctx->last_n = scene->n; ctx->last_chunks = nch; ctx->last_grid = (int)grid; ctx->last_block = 256;
    g_last = ctx;
    return 0;
}

template <typename T, typename SceneT, typename CamT>
int render_host(const SceneT *scene, const CamT *cam, const rtw_params *p, T *out) {
    if (!scene || !cam || !p || !out) return fail(-1, "null argument");
    rtw_scene_handle h = nullptr;
    int rc = upload_scene<T>(scene, p->device, &h);
    if (rc) return rc;
    void *d_out = nullptr;
    const size_t bytes = (size_t)p->width * (size_t)p->height * 3 * sizeof(T);
    hipError_t e = (p->width > 0 && p->height > 0) ? hipMalloc(&d_out, bytes) : hipSuccess;
    if (e != hipSuccess) { rtw_scene_free(h); return fail((int)e, "hipMalloc(image) failed: %s", hipGetErrorString(e)); }
    rc = d_out ? render_device<T>(h, cam, p, d_out, nullptr) : fail(-2, "width/height must be positive");
    if (!rc) {
        e = hipMemcpy(out, d_out, bytes, hipMemcpyDeviceToHost);   // blocking: waits for the render
        if (e != hipSuccess) rc = fail((int)e, "hipMemcpy(image D2H) failed: %s", hipGetErrorString(e));
    }
    if (d_out) hipFree(d_out);
    rtw_scene_free(h);
    return rc;
}

// T0 unit entry point: host slots -> device -> unit_kernel -> host slots
template <typename T, typename SceneT, typename CamT>
int run_unit(int op, int count, const void *in, void *out, const SceneT *scene, const CamT *cam) {
    if (op < 0 || op >= rtw::U_NUM_OPS) return fail(-2, "unknown unit op %d", op);
    if (count < 0 || (count > 0 && (!in || !out))) return fail(-1, "null argument");
    if (count == 0) return 0;
    const bool needs_scene = op == rtw::U_HIT_WORLD || op == rtw::U_RAY_COLOR;
    if (needs_scene && !scene) return fail(-1, "op %d needs a scene", op);
    if (op == rtw::U_GET_RAY && !cam) return fail(-1, "op %d needs a camera", op);
    int dev;
    if (int rc = resolve_device(-1, &dev)) return rc;
    DeviceCtx *ctx;
    if (int rc = get_ctx(dev, &ctx)) return rc;
    rtw_scene_handle h = nullptr;
    rtw::DevScene<T> S{nullptr, nullptr, nullptr, 0, 0};
    if (needs_scene) {
        if (int rc = upload_scene<T>(scene, dev, &h)) return rc;
        using V4 = typename rtw::Vec4<T>::type;
        S.geom = (const V4 *)h->geom; S.mat0 = (const V4 *)h->mat0; S.mat1 = (const V4 *)h->mat1;
        S.n = h->n; S.n_pad = h->n_pad;
    }
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
    const size_t in_b = (size_t)count * rtw::unit_in_slots(op) * 8, out_b = (size_t)count * rtw::unit_out_slots(op) * 8;
    double *d_in = nullptr, *d_out = nullptr;
    int rc = 0;
    hipError_t e;
    if ((e = hipMalloc(&d_in, in_b)) != hipSuccess || (e = hipMalloc(&d_out, out_b)) != hipSuccess ||
        (e = hipMemcpy(d_in, in, in_b, hipMemcpyHostToDevice)) != hipSuccess) {
        rc = fail((int)e, "unit buffers: %s", hipGetErrorString(e));
    } else {
        hipLaunchKernelGGL(rtw::unit_kernel<T>, dim3((count + 63) / 64), dim3(64), 0, 0, op, count, d_in, d_out, S, C);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipMemcpy(out, d_out, out_b, hipMemcpyDeviceToHost)) != hipSuccess)
            rc = fail((int)e, "unit kernel: %s", hipGetErrorString(e));
    }
    if (d_in) hipFree(d_in);
    if (d_out) hipFree(d_out);
    if (h) rtw_scene_free(h);
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
    hipSetDevice(h->device);
    if (h->geom) hipFree(h->geom);
    if (h->c_bound) hipFree(h->c_bound);
    if (h->c_exact) hipFree(h->c_exact);
    if (h->c_mat0) hipFree(h->c_mat0);
    if (h->c_mat1) hipFree(h->c_mat1);
    if (h->c_orig) hipFree(h->c_orig);
    if (h->mat0) hipFree(h->mat0);
    if (h->mat1) hipFree(h->mat1);
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
    DeviceCtx *ctx = g_last;
    if (!ctx) return fail(-6, "no render has been issued from this thread");
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(ctx->ev2));
    float k_ms = 0, t_ms = 0;
    HIP_TRY(hipEventElapsedTime(&k_ms, ctx->ev0, ctx->ev1));
    HIP_TRY(hipEventElapsedTime(&t_ms, ctx->ev0, ctx->ev2));
    rtw::DevCounters c;
    HIP_TRY(hipMemcpy(&c, ctx->ctr, sizeof c, hipMemcpyDeviceToHost));
    if (getenv("RTW_PHASE_PROFILE")) {
        double tot = 0;
        for (int k = 0; k < 6; ++k) tot += (double)c.phase[k];
        fprintf(stderr, "[rtw phase profile] wave-cycles: pull %.1f%%  raygen %.1f%%  scan-pass1/level1 %.1f%%  extract/level2 %.1f%%  resolve %.1f%%  shade %.1f%%  (total %.3g)\n",
                100 * c.phase[0] / tot, 100 * c.phase[1] / tot, 100 * c.phase[2] / tot, 100 * c.phase[4] / tot,
                100 * c.phase[5] / tot, 100 * c.phase[3] / tot, tot);
    }
    memset(out, 0, sizeof *out);
    out->samples = c.samples;
    out->segments = c.segments;
    out->sphere_tests = c.segments * (uint64_t)ctx->last_n;
    out->kernel_ms = k_ms;
    out->total_ms = t_ms;
    out->n_chunks = ctx->last_chunks;
    out->grid_blocks = ctx->last_grid;
    out->block_threads = ctx->last_block;
    return 0;
}

int rtw_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene, const rtw_camera_f32 *cam) {
    return run_unit<float>(op, count, in, out, scene, cam);
}
int rtw_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene, const rtw_camera_f64 *cam) {
    return run_unit<double>(op, count, in, out, scene, cam);
}

int rtw_shutdown(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (DeviceCtx *c : g_ctx) {
        hipSetDevice(c->device);
        if (c->partial) hipFree(c->partial);
        if (c->puv) hipFree(c->puv);
        if (c->ctr) hipFree(c->ctr);
        if (c->ev0) hipEventDestroy(c->ev0);
        if (c->ev1) hipEventDestroy(c->ev1);
        if (c->ev2) hipEventDestroy(c->ev2);
        delete c;
    }
    g_ctx.clear();
    g_last = nullptr;
    return 0;
}

}  // extern "C"
