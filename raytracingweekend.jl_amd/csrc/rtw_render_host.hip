// rtw_render_host.hip -- the host-buffer entry points rtw_render_f32/_f64 (what the Julia ccall binds; replaces src/render.jl:8-44):
// a cached per-device context (uploaded scene, stream, device image), one device or a device list (tiles dealt round-robin, shards
// gathered on the first device by peer copies or ONE RCCL reduce: rtw_multi.hip), one D2H of the frame.
#include "rtw_host.hpp"

namespace rtwh {

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
        size_t idle = 0;
        for (auto &h : ctx->host) if (!h->busy) ++idle;
        if (!pick && ctx->host.size() < RTW_HOST_CTX_POOL && idle < RTW_HOST_CTX_IDLE_KEEP) {
            // a scene this device has not seen (or whose entries are all busy): a NEW entry while the pool has room and fewer than
            // RTW_HOST_CTX_IDLE_KEEP idle entries exist -- a caller that alternates between two or three scenes keeps them all uploaded,
            // concurrent callers get an entry each, but an animation (a new scene every frame from one thread) recycles the least
            // recently used idle entry instead of building eight contexts of ~200 MB each
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
    if (int rc = upload_scene_t(scene, hc->device, &hc->scene)) return rc;
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
        int rc = launch_render_t(hc->scene, cam, &q, hc->d_img, hc->stream, &rec, &rctx);
        if (!rc) rc = copy_out(hc, hc->d_img, out, elems * sizeof(T));
        if (rc) (void)hipStreamSynchronize(hc->stream);           // nothing of this call may still be in flight when the lease ends
        if (!rc) rc = resolve_rec(rec, &g_last.agg);
        if (!rc) g_last.per_device.emplace_back(hc->device, g_last.agg.kernel_ms);
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
    // (RCCL: the communicator set of this device list, locked for this render until its streams have drained -- RcclSet, rtw_multi.hip)
    std::shared_ptr<RcclSet> rccl_set;
    std::unique_lock<std::mutex> rccl_use;
    std::vector<HostLease> L(N);
    for (int r = 0; r < N; ++r)
        if (int rc = acquire_host(devs[r], key, &L[r])) return fail(rc, "device %d (shard %d of %d): %s", devs[r], r, N, std::string(g_err).c_str());
    if (use_rccl) { if (int rc = rccl_acquire(devs, &rccl_set, &rccl_use)) return rc; }       // (after the devices are known to exist; a host context is never waited for, so the order cannot deadlock)
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
            rc = launch_render_t(hc->scene, cam, &q, hc->d_img, hc->stream, &recs[r], &rctx[r]);
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
        if ((rc = launch_render_t(hc->scene, cam, &q, d_out, hc->stream, &recs[r], &rctx[r]))) break;
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
        std::vector<const void *> send(N);
        std::vector<hipStream_t> streams(N);
        for (int r = 0; r < N; ++r) { send[r] = L[r].hc->d_img; streams[r] = L[r].hc->stream; }
        rc = rccl_reduce_frames(*rccl_set, send, root->d_img, (size_t)p->width * p->height * 3, sizeof(T) == 8, streams);
        if (!rc) { const hipError_t e = hipSetDevice(root->device); if (e != hipSuccess) rc = fail((int)e, "hipSetDevice: %s", hipGetErrorString(e)); }
        if (!rc) rc = copy_out(root, root->d_img, out, frame_bytes);
    } else if (!rc) {
        const hipError_t e = hipSetDevice(root->device);
        const long n_tiles = (long)((p->height + 7) / 8) * ((p->width + 7) / 8);
        if (e != hipSuccess) rc = fail((int)e, "hipSetDevice(%d): %s", root->device, hipGetErrorString(e));
        if (!rc) rc = launch_untile(sizeof(T) == 8, root->d_aux, root->d_img, p->width, p->height, n_tiles, N, pad_tiles, root->stream);
        if (!rc) rc = copy_out(root, root->d_img, out, frame_bytes);
    }
    // every stream of this call drains before the leases end, also on an error
    for (int r = 0; r < N; ++r) { HIP_IGNORE(hipSetDevice(L[r].hc->device)); HIP_IGNORE(hipStreamSynchronize(L[r].hc->stream)); }
    if (rccl_use.owns_lock()) rccl_use.unlock();           // the communicators are free for the next render of this device list
    rtw_stats_t &a = g_last.agg;                       // sums over the devices; times: the maximum
    for (int r = 0; r < N; ++r) {
        if (!recs[r]) {                                   // a shard that owns no tile (a tiny frame on many devices): launched nothing --
            if (!rc) g_last.per_device.emplace_back(L[r].hc->device, 0.0);     // still one entry per shard, in shard order (include/rtw_hip.h)
            continue;
        }
        rtw_stats_t st;
        memset(&st, 0, sizeof st);
        if (!rc) rc = resolve_rec(recs[r], &st);
        release_rec(rctx[r], recs[r], rc == 0);
        if (!rc) g_last.per_device.emplace_back(recs[r]->device, st.kernel_ms);
        a.samples += st.samples; a.segments += st.segments; a.sphere_tests += st.sphere_tests;
        a.kernel_ms = std::max(a.kernel_ms, st.kernel_ms); a.total_ms = std::max(a.total_ms, st.total_ms);
        a.n_chunks = st.n_chunks; a.grid_blocks = std::max(a.grid_blocks, st.grid_blocks); a.block_threads = std::max(a.block_threads, st.block_threads);
    }
    a.gather_path = gather_path;
    g_last.resolved = rc == 0;
    return rc;
}

int render_host_f32(const rtw_scene_f32 *scene, const rtw_camera_f32 *cam, const rtw_params *p, float *out) { return render_host<float>(scene, cam, p, out); }
int render_host_f64(const rtw_scene_f64 *scene, const rtw_camera_f64 *cam, const rtw_params *p, double *out) { return render_host<double>(scene, cam, p, out); }

}  // namespace rtwh
