// rtw_abi.hip -- C ABI of librtw_hip.so (include/rtw_hip.h) over the gfx950 kernels.
// Replaces /root/reference/src/render.jl:8-44 behind a ccall-able boundary.  No CPU fallback:
// every compute entry point needs a HIP device and reports an error otherwise.
//
// Concurrency: every render call owns its own record (device counters + events) taken from a
// mutex-guarded per-device pool, the trace kernel keeps all per-render state in LDS / registers
// and there is no shared device workspace, so renders may be in flight concurrently on any mix
// of streams, host threads and devices.  The caller's current HIP device is restored on return.
#include "rtw_host.hpp"
#include "rtw_kernels.hpp"      // (rtw::DevCounters: the size of a render record's device counters)

namespace rtwh {

__thread char g_err[512] = "";

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

std::mutex g_mu;
std::vector<CtxPtr> g_ctx;
std::atomic<unsigned> g_generation{1};
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
    c->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
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
            if (hipEventQuery(r->ev2) != hipSuccess) { (void)hipGetLastError(); continue; }   // still in flight
            r->done = true;
        }
        r->owned = true;
        *out = r.get();
        return 0;
    }
    std::unique_ptr<RenderRec> r(new RenderRec());
    r->device = ctx->device;
    HIP_TRY(hipMalloc(&r->ctr, sizeof(rtw::DevCounters)));
    HIP_TRY(hipHostMalloc((void **)&r->h_ctr, sizeof(rtw::DevCounters), hipHostMallocDefault));
    r->fresh = true;        // (its first render clears ALL of the counters, on its own stream: a hipMemset here would run on the null stream, which
                            //  the non-blocking streams of the renders do not wait for -- it could land in the middle of the first kernel)
    HIP_TRY(hipEventCreate(&r->ev0));
    HIP_TRY(hipEventCreate(&r->ev1));
    HIP_TRY(hipEventCreateWithFlags(&r->ev2, hipEventDisableTiming));
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
    g_last.per_device.clear();
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

int validate_params(const rtw_params *p, int *n_chunks, int *chunk_spp) {
    if (!p) return fail(-1, "null params");
    if (p->width <= 0 || p->height <= 0) return fail(-2, "width/height must be positive (got %d x %d)", p->width, p->height);
    if (p->spp <= 0) return fail(-2, "spp must be positive (got %d)", p->spp);
    if (p->max_depth < 0 || p->max_depth >= (1 << 22)) return fail(-2, "max_depth must be in [0, 2^22)");
    if (p->shard_count <= 0 || p->shard_index < 0 || p->shard_index >= p->shard_count)
        return fail(-2, "bad shard %d of %d", p->shard_index, p->shard_count);
    if (p->n_chunks < 0) return fail(-2, "n_chunks must be >= 0");
    if (p->flags & ~(RTW_FLAG_GROUP_CULL | RTW_FLAG_COMPACT_TILES | RTW_FLAG_SCAN_VALU | RTW_FLAG_RAY_POOL | RTW_FLAG_RCCL_REDUCE | RTW_FLAG_NUMERICS_CONTRACT | RTW_FLAG_NUMERICS_REFERENCE_FMA2)) return fail(-2, "unknown flags 0x%x", p->flags);
    { const int nm = p->flags & (RTW_FLAG_NUMERICS_CONTRACT | RTW_FLAG_NUMERICS_REFERENCE_FMA2);
      if (nm & (nm - 1)) return fail(-2, "the RTW_FLAG_NUMERICS_* bits exclude each other (flags 0x%x)", p->flags); }
    // default rule (ABI 4): one sample per chunk up to 256 chunks -- min(spp, 256).  The finer the items, the shorter the tail of a frame
    // and the less a job slot waits for a straggler; a chunk's stream set-up is made by the whole wave for a batch at a time.  Measured
    // (round 6, new scheduler; 4-sample chunks -> 1-sample chunks): 1080p x 64 spp 28.6 -> 24.9 ms, 1080p x 200 spp 76.0 -> 74.6 ms,
    // 320 x 180 x 64 spp 1.26 -> 0.84 ms, Float64 1080p x 200 spp 56.8 -> 57.0 ms.  1000 spp: 250 chunks of 4 under both rules.
    // (ABI <= 3: min(spp, clamp(spp / 4, 16, 256)).)
    int nch = p->n_chunks > 0 ? p->n_chunks : std::min(p->spp, 256);
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

template <typename CamT>
int render_device(rtw_scene_handle scene, const CamT *cam, const rtw_params *p, void *d_out, void *stream_v) {
    DeviceGuard guard;
    if (p && p->n_devices > 1) return fail(-2, "the device-resident entry point renders on the scene's device only (n_devices = %d)", p->n_devices);
    RenderRec *rec = nullptr;
    CtxPtr ctx;
    release_last();
    int rc = launch_render_t(scene, cam, p, d_out, (hipStream_t)stream_v, &rec, &ctx);
    if (rec) { g_last.recs.push_back(rec); g_last.ctxs.push_back(ctx); }       // (also on a late error: released by the next call)
    return rc;
}

}  // namespace rtwh

using namespace rtwh;

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

int rtw_scene_upload_f32(const rtw_scene_f32 *s, int device, rtw_scene_handle *out) { return upload_scene_f32(s, device, out); }
int rtw_scene_upload_f64(const rtw_scene_f64 *s, int device, rtw_scene_handle *out) { return upload_scene_f64(s, device, out); }

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
    return render_device(s, c, p, d_out, stream);
}
int rtw_render_device_f64(rtw_scene_handle s, const rtw_camera_f64 *c, const rtw_params *p, void *d_out, void *stream) {
    return render_device(s, c, p, d_out, stream);
}
int rtw_render_f32(const rtw_scene_f32 *s, const rtw_camera_f32 *c, const rtw_params *p, float *out) {
    return render_host_f32(s, c, p, out);
}
int rtw_render_f64(const rtw_scene_f64 *s, const rtw_camera_f64 *c, const rtw_params *p, double *out) {
    return render_host_f64(s, c, p, out);
}

int rtw_stats(rtw_stats_t *out) {
    if (!out) return fail(-1, "null argument");
    if (g_last.generation != g_generation.load() || (g_last.recs.empty() && !g_last.resolved))
        return fail(-6, "no render has been issued from this thread");
    DeviceGuard guard;
    if (!g_last.resolved) {
        memset(&g_last.agg, 0, sizeof g_last.agg);
        g_last.per_device.clear();
        for (RenderRec *r : g_last.recs) {
            rtw_stats_t one;
            memset(&one, 0, sizeof one);
            if (int rc = resolve_rec(r, &one)) return rc;
            g_last.per_device.emplace_back(r->device, one.kernel_ms);
            rtw_stats_t &a = g_last.agg;
            a.samples += one.samples; a.segments += one.segments; a.sphere_tests += one.sphere_tests;
            a.kernel_ms = std::max(a.kernel_ms, one.kernel_ms); a.total_ms = std::max(a.total_ms, one.total_ms);
            a.n_chunks = one.n_chunks; a.grid_blocks = std::max(a.grid_blocks, one.grid_blocks); a.block_threads = std::max(a.block_threads, one.block_threads);
        }
        g_last.resolved = true;
    }
    *out = g_last.agg;
    return 0;
}

int rtw_stats_devices(int32_t capacity, int32_t *count, int32_t *devices, double *kernel_ms) {
    if (!count || capacity < 0 || (capacity > 0 && (!devices || !kernel_ms))) return fail(-1, "null argument");
    rtw_stats_t st;
    if (int rc = rtw_stats(&st)) return rc;                    // (resolves a pending device-resident render)
    *count = (int32_t)g_last.per_device.size();
    for (int32_t k = 0; k < capacity && k < *count; ++k) { devices[k] = g_last.per_device[(size_t)k].first; kernel_ms[k] = g_last.per_device[(size_t)k].second; }
    return 0;
}

int rtw_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene, const rtw_camera_f32 *cam) {
    return run_unit_f32(op, count, in, out, scene, cam);
}
int rtw_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene, const rtw_camera_f64 *cam) {
    return run_unit_f64(op, count, in, out, scene, cam);
}

int rtw_shutdown(void) {
    DeviceGuard guard;
    {
        // g_mu is held for the context table only: a render that owns an RCCL communicator set takes g_mu (get_ctx) while it holds the
        // set, so waiting for that set below with g_mu held would be a lock-order inversion (render: set -> g_mu; shutdown: g_mu -> set)
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
    }
    rccl_shutdown();           // (waits for a render that is using a communicator set; g_mu is NOT held here)
    return 0;
}

}  // extern "C"
