// rtw_launch.hip -- one render = ONE launch of the trace kernel (rtw_kernels.hpp; opt-in: the ray-pool kernel of rtw_pool.hpp): which
// instantiation, the persistent grid, the job shape, the per-render counters and events, and what rtw_stats() reads back.
#include "rtw_scene_view.hpp"
#include "rtw_kernels.hpp"
#ifdef RTW_WITH_POOL          // `make POOL=1`: the ray-pool kernel (a measured 16 - 19 % LOSS on this chip, DESIGN_LOG R4) is not in the default library
#include "rtw_pool.hpp"
#endif

namespace rtwh {

// magic number for exact unsigned 32-bit division by an invariant d >= 1 (Granlund-Montgomery / Hacker's
// Delight "add" form): n / d == (umulhi(n, m) + ((n - umulhi(n, m)) >> 1)) >> s for all 32-bit n
void make_udiv(unsigned d, unsigned *m, unsigned *s) {
    if (d <= 1) { *m = 0; *s = 0x80000000u; return; }   // flag: identity
    unsigned l = 0;
    while ((1ull << l) < d) ++l;                        // l = ceil(log2 d) >= 1
    *m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
    *s = l - 1;
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
    S.numerics = (p->flags & RTW_FLAG_NUMERICS_CONTRACT) ? rtw::NUM_CONTRACT : (p->flags & RTW_FLAG_NUMERICS_REFERENCE_FMA2) ? rtw::NUM_REFERENCE_FMA2 : rtw::NUM_REFERENCE;

    // persistent grid: enough 256-thread blocks to fill every CU at the kernel's occupancy
    static const bool phase_profile = aid_env("RTW_PHASE_PROFILE") != nullptr;   // debugging aid, not for timed runs
    const size_t list_bytes = (size_t)RTW_LIST_CAP * 256 * sizeof(unsigned short);
    const size_t shared_bytes = (sizeof(rtw::WgShared<T>) + 15) / 16 * 16;
    const bool cull = (p->flags & RTW_FLAG_GROUP_CULL) != 0;
    rtw::CullScene<T> CS = cull_scene_of<T>(scene);
    CS.numerics = S.numerics;
    const size_t n_cull = (size_t)rtw::cull_exact_count(CS);
    // the plain scan runs pass 1 on the matrix pipe (RTW_SCAN=valu: the all-VALU scan, for A/B measurements)
    static const bool force_valu = aid_env("RTW_SCAN") != nullptr && strcmp(aid_env("RTW_SCAN"), "valu") == 0;
    // (group cull: on the matrix pipe too when the scene has the operands; RTW_FLAG_SCAN_VALU selects the all-VALU cull scan)
    const bool mfma = (cull ? scene->c_mf_ops != nullptr : scene->mf_ops != nullptr) && !force_valu && !(p->flags & RTW_FLAG_SCAN_VALU);
    // (group cull on the matrix pipe: the tables of the block vote travel with the scene copy)
    const size_t geom_bytes = cull ? n_cull * sizeof(V4) + ((n_cull * sizeof(unsigned short) + 15) / 16) * 16 +
                                         (mfma ? (size_t)rtw::cull_tab_words(scene->c_mf_blocks) * sizeof(unsigned) : 0)
                                   : (size_t)rtw::scene_geom_alloc(scene->n, scene->n_pad) * sizeof(V4);
    const bool lds_scene = geom_bytes <= RTW_LDS_SCENE_MAX_BYTES;
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
    // the default numerics mode of the headline variants (scene in LDS, matrix pipe): an instance with the mode fixed at compile time
    if (S.numerics == rtw::NUM_REFERENCE && lds_scene && mfma && !phase_profile)
        kern = cull ? (kern_t)rtw::trace_kernel<T, false, true, true, true, rtw::NUM_REFERENCE> : (kern_t)rtw::trace_kernel<T, false, true, false, true, rtw::NUM_REFERENCE>;
    // The ray-pool kernel (rtw_pool.hpp; opt-in: RTW_FLAG_RAY_POOL, or RTW_POOL=1 in the environment for A/B runs) exists in `make POOL=1`
    // builds only: Float32 plain scans on the matrix pipe, when the pool, the rings and the scene copy fit the 160 KB of LDS of a CU (one
    // workgroup of RTW_POOL_W waves per CU); everything else runs the lane-loop kernel above.  The default library refuses the flag.
    [[maybe_unused]] size_t pool_lds = 0;
    bool pool = false;
    int block_threads = 256;
    int blocks_per_cu = 0;
#ifdef RTW_WITH_POOL
    static const bool env_pool = aid_flag("RTW_POOL");
    typedef void (*pool_kern_t)(rtw::KParams, rtw::Camera<T>, rtw::DevScene<T>, T *, rtw::DevCounters *);
    pool_kern_t pool_kern = nullptr;
    if constexpr (sizeof(T) == 4) {
        pool_lds = rtw::pool_fixed_lds_bytes<T, RTW_POOL_W, RTW_POOL_R>() + rtw::pool_scene_lds_bytes<T>(scene->n, scene->n_pad);
        pool = mfma && !cull && (env_pool || (p->flags & RTW_FLAG_RAY_POOL)) && pool_lds <= ctx->lds_per_cu &&
               cs <= RTW_POOL_MAX_CHUNK_SPP;
        pool_kern = phase_profile ? (pool_kern_t)rtw::trace_pool_kernel<T, RTW_POOL_W, RTW_POOL_R, true> : (pool_kern_t)rtw::trace_pool_kernel<T, RTW_POOL_W, RTW_POOL_R, false>;
    }
    if (pool) {
        block_threads = RTW_POOL_W * 64;
        // "everything else runs the lane-loop kernel": also a device (or a runtime) that refuses this much dynamic LDS
        hipError_t e = hipFuncSetAttribute((const void *)pool_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pool_lds);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, pool_kern, block_threads, pool_lds);
        if (e != hipSuccess || blocks_per_cu < 1) { (void)hipGetLastError(); pool = false; block_threads = 256; blocks_per_cu = 0; }
    }
#else
    if (p->flags & RTW_FLAG_RAY_POOL)
        return fail(-7, "RTW_FLAG_RAY_POOL: this build of librtw_hip has no ray-pool kernel (a measured loss on MI355X; `make POOL=1` builds it in)");
#endif
    if (!pool) {
        // (the runtime's answer for a (kernel, LDS size) pair does not change: asked once per device context -- 4 us per render otherwise)
        std::lock_guard<std::mutex> lk(ctx->mu);
        for (auto &o : ctx->occupancy) if (o.first.first == (const void *)kern && o.first.second == lds_bytes) blocks_per_cu = o.second;
        if (blocks_per_cu < 1) {
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, 256, lds_bytes));
            ctx->occupancy.push_back({{(const void *)kern, lds_bytes}, blocks_per_cu});
        }
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
    // Round 6, with out-of-order job slots and static first claims (tools/gpu_small_sweep.sh, kernel us for 1 / 4 / 8 / 16 pixels):
    //   1/8 shard of 1080p x 1000 spp (250 chunks)   51.1 / 53.0 / -- / 73.8 ms      one-pixel jobs: the shortest drain
    //   1/8 shard of 1080p x 64 spp (64 chunks)      4.28 / 4.08 / -- / 5.22 ms      (1 pixel x 64 chunks = ONE batch per job: every batch opens a job)
    //   320 x 180 x 64 spp (64 chunks; 11 jobs of 4 pixels per workgroup)            1008 / 838 / 1002 / 997 us
    //   200 x 112 x 32 spp Float64 (32 chunks; 5 per workgroup)                      572 / 377 / 419 / 334 us
    //   96 x 54 x 16 spp (16 chunks; 1 per workgroup)                                527 / 155 / 91 / 58 us
    // -> one pixel only when its batches are many (>= 2 per job); the LARGEST jobs when a workgroup sees only a handful (a frame of
    //    a few hundred microseconds is a latency chain per wave: the fewest job openings win).
    int job_shift = nch >= 16 ? 2 : 4;
    if (nch >= 128 && n_local * 16 < 150 * grid) job_shift = 0;
    else if (n_local * 16 < 8 * grid) job_shift = 4;
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
    if (total_jobs >= (1ll << 30) || queue0_jobs >= (1ll << 28) || total_jobs * bpj >= (1ll << 40))
        return fail(-5, "render too large for one call: %lld pixel-block jobs", total_jobs);
    K.total_jobs = (unsigned)total_jobs; K.local_tiles = (unsigned)n_local; K.bpj = (unsigned)bpj; K.job_shift = (unsigned)job_shift;
    K.rows_shift = (unsigned)std::min(job_shift, 3);       // 4 x 1, 8 x 1, 8 x 2 pixels: whole column strips
    static const int env_rows_shift = aid_env("RTW_ROWS_SHIFT") ? atoi(aid_env("RTW_ROWS_SHIFT")) : -1;             // measurement aid: job shape
    if (env_rows_shift >= 0 && env_rows_shift <= job_shift && env_rows_shift <= 3 && job_shift - env_rows_shift <= 2) K.rows_shift = (unsigned)env_rows_shift;
    K.slot_stride = (unsigned)(sizeof(rtw::JobSlot) + 64u * (1u << job_shift));
    K.n_slots = std::min(24u, (unsigned)RTW_SLOT_BYTES / K.slot_stride);             // 24 / 12 / 7 / 4 slots of 1 / 4 / 8 / 16 pixels
    make_udiv((unsigned)bpj, &K.div_bpj_m, &K.div_bpj_s);
    long long max_useful = (total_jobs * bpj + 3) / 4;                                             // one batch per wave, 4 waves per block
#ifdef RTW_WITH_POOL
    if (pool) max_useful = (total_jobs * bpj * 64 + RTW_POOL_R - 1) / RTW_POOL_R;                  // one item per slot of the pool
#endif
    if (grid > max_useful) grid = max_useful;
    // (measurement aid: RTW_GRID_BLOCKS caps the persistent grid -- fewer waves per SIMD, same image)
    static const long long env_grid = aid_env("RTW_GRID_BLOCKS") ? atoll(aid_env("RTW_GRID_BLOCKS")) : 0;
    if (env_grid > 0 && grid > env_grid) grid = env_grid;
    if (grid < 1) grid = 1;

    RenderRec *rec;
    if (int rc = acquire_rec(ctx.get(), &rec)) return rc;
    *rec_out = rec;
    rec->n_spheres = scene->n; rec->n_chunks = nch; rec->grid = (int)grid; rec->block = block_threads;
    // the counters: queue heads + segment / sample counts (+ the phase cells) are cleared per render; the drain clocks and their 16 KB
    // histogram only when the drain is profiled (the kernel touches them only then)
    static const bool drain_profile = aid_env("RTW_DRAIN_PROFILE") != nullptr;
    K.drain_profile = drain_profile ? 1 : 0;
    rec->ctr_bytes = (drain_profile || phase_profile || pool) ? sizeof(rtw::DevCounters) : offsetof(rtw::DevCounters, t_first);    // (the ray-pool kernel keeps its stage profile / watchdog state in the histogram cells)
    HIP_TRY(hipMemsetAsync(rec->ctr, 0, rec->fresh ? sizeof(rtw::DevCounters) : rec->ctr_bytes, stream));
    rec->fresh = false;
    if (drain_profile) HIP_TRY(hipMemsetAsync(&rec->ctr->t_first, 0xff, sizeof(unsigned long long), stream));
    // pixels of other shards read 0 in the full-frame layout (the sum over the shards is the image)
    if (K.out_layout == 0 && p->shard_count > 1)
        HIP_TRY(hipMemsetAsync(d_out, 0, (size_t)p->width * p->height * 3 * sizeof(T), stream));
    HIP_TRY(hipEventRecord(rec->ev0, stream));
    if (total_jobs > 0) {
        (void)hipGetLastError();           // (hipEventQuery's hipErrorNotReady in acquire_rec must not be mistaken for a launch failure)
#ifdef RTW_WITH_POOL
        if (pool) hipLaunchKernelGGL(pool_kern, dim3((unsigned)grid), dim3((unsigned)block_threads), pool_lds, stream, K, C, S, (T *)d_out, rec->ctr);
        else
#endif
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds_bytes, stream, K, C, S, CS, (T *)d_out, rec->ctr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipEventRecord(rec->ev1, stream));
    HIP_TRY(hipMemcpyAsync(rec->h_ctr, rec->ctr, rec->ctr_bytes, hipMemcpyDeviceToHost, stream));     // (into pinned memory: truly asynchronous)
    HIP_TRY(hipEventRecord(rec->ev2, stream));
    rec->used = true; rec->done = false;
    return 0;
}

// wait for a record's kernel and add its counters to `agg`
int resolve_rec(RenderRec *r, rtw_stats_t *agg) {
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipEventSynchronize(r->ev2));
    r->done = true;
    float k_ms = 0;
    HIP_TRY(hipEventElapsedTime(&k_ms, r->ev0, r->ev1));
    const rtw::DevCounters &c = *r->h_ctr;         // (the copy that followed the kernel on its stream; the drain part only with RTW_DRAIN_PROFILE)
    if (aid_env("RTW_DEBUG")) {
        rtw::DevCounters chk;
        HIP_TRY(hipMemcpy(&chk, r->ctr, offsetof(rtw::DevCounters, t_first), hipMemcpyDeviceToHost));
        fprintf(stderr, "[rtw debug] resolve rec %p dev %d grid %d: pinned copy samples %llu segments %llu | device now samples %llu segments %llu%s\n", (void *)r, r->device, r->grid,
                (unsigned long long)c.samples, (unsigned long long)c.segments, (unsigned long long)chk.samples, (unsigned long long)chk.segments, c.samples != chk.samples ? "  <-- STALE" : "");
    }
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
        if (c.phase[8])
            fprintf(stderr, "[rtw phase profile] lane loop: %llu wave-iterations, lane utilisation at the scan %.2f %% (%llu lanes with a ray), %.2f %% of the iterations without any ray; "
                            "lane-iterations lost at batch boundaries %.2f %% (pool short) + %.2f %% (no usable batch)\n", (unsigned long long)c.phase[8],
                    100.0 * (double)c.phase[9] / (64.0 * (double)c.phase[8]), (unsigned long long)c.phase[9], 100.0 * (double)c.phase[10] / (double)c.phase[8],
                    100.0 * (double)c.phase[11] / (64.0 * (double)c.phase[8]), 100.0 * (double)c.phase[12] / (64.0 * (double)c.phase[8]));
        if (c.phase[8]) {
            fprintf(stderr, "[rtw phase counts] per wave-iteration:");
            static const char *nm[32] = {0, 0, 0, 0, 0, 0, "blocks_without_candidate", "blocks", "iterations", "lanes_with_ray", "iterations_without_ray", "takers_pool_short", "takers_unserved",
                                         "list_entries", "exact_test_rounds", "exact_tests", "reject_trials", "R_executed", "H1_executed", "H1_rounds", "A_executed", "jobs_stored", "batches_set_up",
                                         "H2_executed", "B_executed", "F_normalize_executed", "sign_collections", "blocks_recording", "explode_iterations", "reject_trials_3_draws", 0, 0};
            for (int k = 6; k < 32; ++k) if (nm[k]) fprintf(stderr, " %s=%.4f", nm[k], (double)c.phase[k] / (double)c.phase[8]);
            fprintf(stderr, "\n");
        }
        if (c.phase[8] && c.phase[13])
            fprintf(stderr, "[rtw phase profile] pass 2: %.1f list entries, %.2f exact-test rounds and %.1f exact tests per wave-iteration (lanes busy in a round: %.1f %%)\n",
                    (double)c.phase[13] / (double)c.phase[8], (double)c.phase[14] / (double)c.phase[8], (double)c.phase[15] / (double)c.phase[8], 100.0 * (double)c.phase[15] / (64.0 * (double)c.phase[14]));
        if (c.phase[7])
            fprintf(stderr, "[rtw phase profile] matrix-pipe scan: %.1f%% of the (wave, block of 32 spheres) evaluations found no candidate in any lane (%llu of %llu)\n",
                    100.0 * (double)c.phase[6] / (double)c.phase[7], (unsigned long long)c.phase[6], (unsigned long long)c.phase[7]);
    }
    if (r->ctr_bytes == sizeof(rtw::DevCounters) && c.end_hist[0] == 0xdeadbeefu) {        // (RTW_POOL_WATCHDOG builds: the pool kernel gave up; its state)
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

int launch_render_f32(rtw_scene_handle scene, const rtw_camera_f32 *cam, const rtw_params *p, void *d_out, hipStream_t stream, RenderRec **rec_out, CtxPtr *ctx_out) {
    return launch_render<float>(scene, cam, p, d_out, stream, rec_out, ctx_out);
}
int launch_render_f64(rtw_scene_handle scene, const rtw_camera_f64 *cam, const rtw_params *p, void *d_out, hipStream_t stream, RenderRec **rec_out, CtxPtr *ctx_out) {
    return launch_render<double>(scene, cam, p, d_out, stream, rec_out, ctx_out);
}

}  // namespace rtwh
