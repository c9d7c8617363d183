// rtw_multi.hip -- what a device list needs beyond one device: peer access per device pair, the on-demand binding of RCCL (dlopen:
// never linked) with one communicator set per device list, and the kernel that un-tiles gathered compact shards into the frame.
#include "rtw_host.hpp"
#include <dlfcn.h>

namespace rtwh {

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

// ---- RCCL, loaded on demand (librccl.so is 570 MB: a caller that renders on one device never pays for it; RTW_RCCL_LIB overrides
//      the path).  The library is neither linked nor are its headers needed to BUILD this file: the handful of types and constants this
//      file uses are declared here, with the values of rccl.h (NCCL's public ABI: stable since NCCL 2.0). -----------------------------
typedef struct ncclComm *ncclComm_t;
typedef int ncclResult_t;                                       // enum in rccl.h; ncclSuccess == 0
enum { RTW_NCCL_SUCCESS = 0, RTW_NCCL_SUM = 0, RTW_NCCL_FLOAT32 = 7, RTW_NCCL_FLOAT64 = 8 };
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, int /*ncclDataType_t*/, int /*ncclRedOp_t*/, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
// The communicators of one device list.  A communicator takes ONE collective at a time: `use` serialises the renders that share a
// set -- held from ncclGroupStart until every stream of the render has drained (two host threads rendering with the same device list
// would otherwise submit collectives on the same ncclComm_t at once) -- and rtw_shutdown() takes it before ncclCommDestroy, i.e.
// waits for a render in flight; `dead` tells a render that found the set before the shutdown to make a new one.
struct RcclSet {
    std::vector<ncclComm_t> comms;
    std::mutex use;
    bool dead = false;
};
std::mutex g_rccl_mu;
RcclApi g_rccl;
std::map<std::vector<int>, std::shared_ptr<RcclSet>> g_rccl_sets;      // device list -> its communicators (ncclCommInitAll), kept until rtw_shutdown()

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

// The communicator set of a device list, created on first use; returns with `use` locked on the set (see RcclSet).
int rccl_acquire(const std::vector<int> &devs, std::shared_ptr<RcclSet> *out, std::unique_lock<std::mutex> *use) {
    for (;;) {
        std::shared_ptr<RcclSet> set;
        {
            std::lock_guard<std::mutex> lk(g_rccl_mu);
            if (int rc = rccl_load()) return rc;
            auto it = g_rccl_sets.find(devs);
            if (it == g_rccl_sets.end()) {
                std::vector<int> sorted(devs);
                std::sort(sorted.begin(), sorted.end());
                if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end())
                    return fail(-2, "RTW_FLAG_RCCL_REDUCE needs distinct devices (a communicator has one rank per GPU)");
                std::shared_ptr<RcclSet> fresh(new RcclSet());
                fresh->comms.resize(devs.size());
                const ncclResult_t r = g_rccl.CommInitAll(fresh->comms.data(), (int)devs.size(), devs.data());
                if (r != RTW_NCCL_SUCCESS) return fail(1000 + (int)r, "ncclCommInitAll over %d devices: %s", (int)devs.size(), g_rccl.GetErrorString(r));
                it = g_rccl_sets.emplace(devs, std::move(fresh)).first;
            }
            set = it->second;
        }
        std::unique_lock<std::mutex> lk(set->use);             // (not under g_rccl_mu: another render may hold it for a whole frame)
        if (set->dead) continue;                                // destroyed by a shutdown in between: make a new one
        *out = std::move(set);
        *use = std::move(lk);
        return 0;
    }
}

// ONE ncclReduce(sum, root = rank 0) of `count` elements per rank, rank r on streams[r]; the caller holds set.use
int rccl_reduce_frames(RcclSet &set, const std::vector<const void *> &send, void *recv_root, size_t count, bool f64, const std::vector<hipStream_t> &streams) {
    const int N = (int)set.comms.size();
    ncclResult_t nr = g_rccl.GroupStart();
    for (int r = 0; r < N && nr == RTW_NCCL_SUCCESS; ++r)
        nr = g_rccl.Reduce(send[r], recv_root, count, f64 ? RTW_NCCL_FLOAT64 : RTW_NCCL_FLOAT32, RTW_NCCL_SUM, 0, set.comms[r], streams[r]);
    const ncclResult_t ne = g_rccl.GroupEnd();
    if (nr == RTW_NCCL_SUCCESS) nr = ne;
    if (nr != RTW_NCCL_SUCCESS) return fail(1000 + (int)nr, "ncclReduce over %d devices: %s", N, g_rccl.GetErrorString(nr));
    return 0;
}

// rtw_shutdown(): every set leaves the map; each is destroyed once no render uses it any more
void rccl_shutdown() {
    std::vector<std::shared_ptr<RcclSet>> sets;
    {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        for (auto &kv : g_rccl_sets) sets.push_back(kv.second);
        g_rccl_sets.clear();
    }
    for (auto &s : sets) {
        std::lock_guard<std::mutex> use(s->use);                // waits for a render in flight on this set
        for (ncclComm_t cm : s->comms) (void)g_rccl.CommDestroy(cm);
        s->comms.clear();
        s->dead = true;
    }
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

int launch_untile(bool f64, const void *gather, void *frame, int W, int H, long n_tiles, int n_shards, long pad_tiles, hipStream_t stream) {
    if (n_tiles <= 0) return 0;
    (void)hipGetLastError();
    const dim3 grid((unsigned)((n_tiles * 64 + 255) / 256)), block(256);
    if (f64) hipLaunchKernelGGL(untile_kernel<double>, grid, block, 0, stream, (const double *)gather, (double *)frame, W, H, (H + 7) / 8, n_tiles, n_shards, pad_tiles);
    else hipLaunchKernelGGL(untile_kernel<float>, grid, block, 0, stream, (const float *)gather, (float *)frame, W, H, (H + 7) / 8, n_tiles, n_shards, pad_tiles);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "un-tile kernel: %s", hipGetErrorString(e));
    return 0;
}

}  // namespace rtwh
