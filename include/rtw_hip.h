/*
 * rtw_hip.h -- C ABI of librtw_hip.so, the MI355X (gfx950) implementation of the hot path
 *              render -> ray_color -> hit/scatter  of claforte/RayTracingWeekend.jl.
 *
 * The reference has NO FFI: its boundary for this path is the exported Julia function
 *     render(scene::HittableList, cam::Camera{T}, image_width=400, n_samples=1) -> Matrix{RGB{T}}
 * (/root/reference/src/render.jl:8-44, export at src/RayTracingWeekend.jl:25).  This header is
 * what a Julia `ccall` shim for that function binds (julia/RTWeekendHIP.jl):
 * plain pointers and sizes only, no torch / HIP types in any signature.
 *
 * Conventions
 *   - every pointer is owned by the caller; the library reads inputs and writes outputs only
 *     during the call and retains nothing (device-side handles excepted, see below);
 *   - return 0 = OK, negative = argument/validation error, positive = hipError_t;
 *     rtw_last_error() returns a thread-local message valid until the next call on that thread;
 *   - never throws, never calls exit/abort;
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails.
 */
#ifndef RTW_HIP_H
#define RTW_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTW_ABI_VERSION 4   /* round 6: no field moved; two flags of version 3 are gone from the default library -- RTW_FLAG_NUMERICS_REFERENCE_FMA
                               (64: a numerics mode no compiler was found to emit; the bit is now an unknown flag, -2) and RTW_FLAG_RAY_POOL (8:
                               still defined, but the kernel is a `make POOL=1` build option; the default library answers -7) -- and the default
                               chunk rule of rtw_params.n_chunks changed (below).  Version 3 (round 5): the DEFAULT image became the reference's
                               own un-fused order of hit(::Sphere) (RTW_FLAG_NUMERICS_*; version 2's image = RTW_FLAG_NUMERICS_CONTRACT) and the
                               measurement switches of the environment are honoured only under RTW_ENABLE_TEST_AIDS=1 */

/* Material kinds: Lambertian / Metal / Dielectric (src/material.jl:3-5, 25-29, 37-39). */
enum { RTW_LAMBERTIAN = 0, RTW_METAL = 1, RTW_DIELECTRIC = 2 };

/* A HittableList of Sphere{T} (src/structs.jl:10,31-35) flattened to SoA.  Host pointers. */
typedef struct {
    int32_t n;
    const float *cx, *cy, *cz, *r;   /* Sphere.center, Sphere.radius (may be negative)       */
    const int32_t *kind;             /* RTW_* of Sphere.mat                                  */
    const float *ar, *ag, *ab;       /* albedo (Lambertian, Metal)                           */
    const float *param;              /* Metal.fuzz / Dielectric.ir / 0                       */
} rtw_scene_f32;

typedef struct {
    int32_t n;
    const double *cx, *cy, *cz, *r;
    const int32_t *kind;
    const double *ar, *ag, *ab;
    const double *param;
} rtw_scene_f64;

/* Camera{T}: the 22 scalars in the field order of src/camera.jl:2-9. */
typedef struct {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
} rtw_camera_f32;

typedef struct {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius;
} rtw_camera_f64;

/* rtw_params.flags.  RTW_FLAG_GROUP_CULL: opt-in accelerated closest-hit scan:
 * spheres are clustered at upload (kd clusters with boxes) and spheres that a ray provably cannot hit are never
 * tested: a block of 32 spatially sorted spheres is visited only when some ray of the half wave can touch its box
 * (default: per-ray block sets looked up from tables, ORed over the half wave),
 * or, with RTW_FLAG_SCAN_VALU, per ray and cluster of 16 on the vector ALUs (the round-1/2 form).
 * Bit-identical images; default (0) is the reference's plain linear scan. */
#define RTW_FLAG_GROUP_CULL 1
/* RTW_FLAG_COMPACT_TILES (device-resident entry points): write only this shard's 8x8 tiles, tile-major
 * and compact -- local tile k (global tile k*shard_count + shard_index, tiles numbered column-major like
 * the image) at [k*64 + (i mod 8) + 8*(j mod 8)]*3 -- instead of the zero-padded full frame: the buffer is
 * ceil((n_tiles - shard_index) / shard_count) * 192 elements and a gather moves 1/shard_count of the frame. */
#define RTW_FLAG_COMPACT_TILES 2
/* RTW_FLAG_SCAN_VALU: run the plain scan entirely on the vector ALUs (the contract discriminant for every
 * sphere and every ray, 11 instructions each).  Default (0): pass 1 of the plain scan is a conservative filter
 * on the matrix pipe (v_mfma_f32_32x32x16_f16 over f16-split features), ~2.3x faster; the
 * exact contract test still decides every hit, so the image is the same bit for bit.  For A/B measurements
 * and as the reference the filter is tested against; also what the library uses by itself for scenes whose
 * extent the f16 split cannot cover (|coordinates| or radii beyond 2^40 or all below 2^-40). */
#define RTW_FLAG_SCAN_VALU 4
/* RTW_FLAG_RAY_POOL (Float32 plain scans on the matrix pipe; ignored otherwise): trace with the ray-pool kernel (rtw_pool.hpp) -- the
 * rays of a workgroup are parked in LDS between the stages scan / shade / path end and every stage runs on full waves of one
 * kind, instead of every lane keeping its own ray from the camera to the sky (rtw_kernels.hpp, the default).  Scheduling only:
 * the image is identical bit for bit.  Measured 14 % SLOWER than the default on MI355X (DESIGN.md section 6.4 says why); kept as
 * an independent cross-check of the default kernel and as the starting point for hardware with more LDS per CU. */
#define RTW_FLAG_RAY_POOL 8
/* RTW_FLAG_RCCL_REDUCE (rtw_render_f32/_f64 with a device list): every device renders its tiles into a zero-padded full frame and
 * ONE ncclReduce(sum, root = the first device of the list) over xGMI assembles the image (BASELINE configs[3]: "tile-sharded + RCCL
 * framebuffer reduce"; x + 0 == x, so the sum is the image bit for bit).  librccl is loaded on demand (dlopen; RTW_RCCL_LIB overrides
 * the path), one communicator per device list is kept until rtw_shutdown().  The devices of the list must be distinct.  Default (0):
 * compact shards gathered with peer copies (1/N of a frame per device instead of a whole one). */
#define RTW_FLAG_RCCL_REDUCE 16
/* The deciding arithmetic of the ray-sphere test, src/hit.jl:16-18.  In Float32 the choice is visible: on
 * scene_random_spheres the contract form traces 4 % fewer ray segments per sample than the reference's own order and its image is
 * brighter by 0.003 in the mean (fewer tmin re-hits of the r = 1000 ground sphere); Float64 images agree to the last few ulps.
 *   default (neither bit)            `oc . r.dir` and `oc . oc` as StaticArrays' dot evaluates them -- (x1 y1 + x2 y2) + x3 y3, no FMA:
 *                                    a callee, which @fastmath does not rewrite --, c = oc.oc - r^2, disc = half_b^2 - c, one rounding each
 *   RTW_FLAG_NUMERICS_REFERENCE_FMA2 the un-fused dots with BOTH squares contracted, disc = fma(half_b, half_b, -c) and c = fma(-r, r, oc.oc): what LLVM emits for an FMA target when BOTH squares of lines 17-18 carry
 *                                    fast-math flags (tools/llvm_fastmath_check/: with the flag-less llvm.powi of Julia's pow_fast neither site is fused)
 *   RTW_FLAG_NUMERICS_CONTRACT       ABI 2's arithmetic: half_b, r^2 - |oc|^2 and disc as three FMA chains
 * The bits exclude each other.  tools/julia_kat.jl + tools/check_julia_kat.py decide between them on a Julia box. */
#define RTW_FLAG_NUMERICS_CONTRACT 32
/* (64 was RTW_FLAG_NUMERICS_REFERENCE_FMA in ABI 3 -- only the last step contracted: removed in ABI 4, neither LLVM experiment of
 *  tools/llvm_fastmath_check/ emits it; the bit is rejected as unknown) */
#define RTW_FLAG_NUMERICS_REFERENCE_FMA2 128
/* Measurement / test switches of the ENVIRONMENT (RTW_SCAN, RTW_POOL, RTW_JOB_PIXELS, RTW_ROWS_SHIFT, RTW_NO_HUGE,
 * RTW_DEBUG_REMOTE_SHARDS, RTW_DEBUG_NO_PEER, RTW_PHASE_PROFILE, RTW_DRAIN_PROFILE, RTW_DEBUG) are honoured only when the master switch
 * RTW_ENABLE_TEST_AIDS=1 is set too (read once per process).  Without it a stray variable changes nothing: a render's kernel choice, launch
 * geometry and gather path depend on rtw_params alone. */
/* rtw_stats_t.gather_path (bits): how the shards of the last multi-device render reached the first device */
#define RTW_GATHER_PEER 1         /* hipMemcpyPeerAsync with peer access enabled in both directions (xGMI)      */
#define RTW_GATHER_HOST_STAGED 2  /* no peer access on this platform: D2H into pinned memory, H2D on the root   */
#define RTW_GATHER_RCCL 4         /* ncclReduce (RTW_FLAG_RCCL_REDUCE)                                           */
#define RTW_GATHER_SAME_DEVICE 8  /* a shard on the root's own device rendered straight into the gather buffer   */

/* Positional arguments of render() plus the keyword extras of the shim. */
typedef struct {
    int32_t width;        /* image_width  (src/render.jl:8)                                   */
    int32_t height;       /* image_width div 16//9 (src/render.jl:11-12); computed by caller  */
    int32_t spp;          /* n_samples    (src/render.jl:9)                                   */
    int32_t max_depth;    /* ray_color depth; reference default 16 (src/ray_color.jl:14)      */
    uint64_t seed;        /* render seed; the stream of (pixel, chunk) derives from it        */
    int32_t n_chunks;     /* sample chunks per pixel, each with its own RNG stream;
                             0 = default rule min(spp, 256) (ABI <= 3: min(spp, clamp(spp / 4, 16, 256)): the default images of
                             renders with 17 .. 1023 spp changed with ABI 4; 1000 spp: 250 chunks either way).  Part of the image definition
                             (the sample radiances themselves are added exactly, in any order). */
    int32_t shard_index;  /* this call renders the 8x8 pixel tiles t with                      */
    int32_t shard_count;  /*   t mod shard_count == shard_index; other pixels are written 0    */
    int32_t device;       /* HIP device ordinal; -1 = current device                          */
    int32_t gamma;        /* 1 = sqrt per channel (rgb_gamma2, src/vec.jl:22); 0 = linear mean */
    int32_t flags;        /* 0, or RTW_FLAG_* (opt-in modes; the image is identical in every mode)    */
    int32_t n_devices;    /* rtw_render_f32/_f64 only (Julia keyword `devices`): 0 (or 1 with device_ids null) = the one
                             device named by `device`; N >= 1 = the N ordinals in device_ids; -1 = every visible
                             device.  The 8x8 tiles are dealt round-robin to the devices (a stream each); the
                             shards are gathered in HBM of the first device of the list -- peer copies with peer
                             access enabled per device pair (xGMI), a host-staged copy where the platform refuses
                             it; RTW_FLAG_RCCL_REDUCE: one ncclReduce instead -- and the frame is copied to `out`
                             once.  The image is identical for every device list; rtw_stats_t.gather_path says
                             which path ran. */
    int32_t job_pixels;   /* 0 = automatic.  1, 4, 8 or 16: pixels per work-queue job (1x1, 4x1, 8x1, 8x2: rows x columns).
                             Scheduling granularity only -- the image is identical for every value.      */
    const int32_t *device_ids; /* n_devices > 1: HIP ordinals; an ordinal may repeat (its shards then run
                             concurrently on that device)                                               */
} rtw_params;

/* Counters of the most recent render issued from the calling thread (summed over its devices;
 * times are the maximum over the devices). */
typedef struct {
    uint64_t samples;       /* pixel samples taken by this shard                              */
    uint64_t segments;      /* ray segments == closest-hit scans (src/hit.jl:38-50)           */
    uint64_t sphere_tests;  /* segments * n spheres (src/hit.jl:12-35 evaluations)            */
    double kernel_ms;       /* HIP-event time of the trace kernel (the only kernel of a render) */
    double total_ms;        /* same (kept from ABI 1, where a second kernel stored the image)  */
    int32_t n_chunks;       /* chunks per pixel actually used                                 */
    int32_t grid_blocks;    /* trace kernel launch geometry                                   */
    int32_t block_threads;
    int32_t gather_path;    /* multi-device renders: RTW_GATHER_* bits; 0 otherwise                  */
} rtw_stats_t;

int rtw_abi_version(void);
int rtw_device_count(int *count);
const char *rtw_last_error(void);

/* render(scene, cam, width, spp) for elem_type Float32 / Float64 -- replaces
 * /root/reference/src/render.jl:8-44.  `out` is a HOST buffer of height*width*3 elements in
 * the memory layout of the returned Matrix{RGB{T}}: pixel (i,j) (1-based row, column) at
 * ((j-1)*height + (i-1))*3.  Blocking.  Uploads the scene, renders, copies the image back.
 * The library keeps the uploaded scene (recognised by its bytes), a stream and the device image per
 * device between calls: from the second call of a scene on, a call costs its kernel + one D2H of the
 * image (0.5 ms at 1920x1080 Float32).  rtw_shutdown() releases them. */
int rtw_render_f32(const rtw_scene_f32 *scene, const rtw_camera_f32 *cam, const rtw_params *p,
                   float *out);
int rtw_render_f64(const rtw_scene_f64 *scene, const rtw_camera_f64 *cam, const rtw_params *p,
                   double *out);

/* Device-resident variant of the same call for callers that keep the image in HBM (bench.py,
 * multi-process sharding over RCCL).  `scene` is a handle from rtw_scene_upload_*; `d_out` is a
 * DEVICE pointer to height*width*3 elements; `hip_stream` is a hipStream_t passed as void*
 * (NULL = the null stream).  Asynchronous with respect to the host: work is enqueued on the
 * stream and nothing is waited for; rtw_stats() synchronises with it.  Re-entrant: any number of
 * renders may be in flight on any mix of streams, host threads and devices (each call owns its
 * counters; there is no shared device workspace). */
typedef struct rtw_scene_dev *rtw_scene_handle;
int rtw_scene_upload_f32(const rtw_scene_f32 *scene, int device, rtw_scene_handle *out);
int rtw_scene_upload_f64(const rtw_scene_f64 *scene, int device, rtw_scene_handle *out);
int rtw_scene_free(rtw_scene_handle scene);
int rtw_render_device_f32(rtw_scene_handle scene, const rtw_camera_f32 *cam, const rtw_params *p,
                          void *d_out, void *hip_stream);
int rtw_render_device_f64(rtw_scene_handle scene, const rtw_camera_f64 *cam, const rtw_params *p,
                          void *d_out, void *hip_stream);

/* Counters/timings of the last render issued from this thread (waits for it to finish). */
int rtw_stats(rtw_stats_t *out);

/* The shards of the last render issued from this thread, one entry per shard in shard order (a device list of N: N entries; one
 * device: one): the HIP ordinal it ran on and the HIP-event time of its trace kernel.  *count receives the number of shards; at most
 * `capacity` entries are written.  (rtw_stats_t.kernel_ms is the maximum of these.) */
int rtw_stats_devices(int32_t capacity, int32_t *count, int32_t *devices, double *kernel_ms);

/* Unit-level device entry points used by the parity tests (tier T0): each evaluates the
 * device implementation of one reference function on `count` inputs, one lane per input.
 * All pointers are HOST pointers; layouts are documented in tests/test_gpu_units.py.
 *   op: 0 hit_sphere  1 reflect  2 refract  3 reflectance  4 scatter  5 get_ray  6 skycolor
 *       7 rng_f (uniforms from a stream state)  8 hit_world  9 ray_color
 *       10 hit_world, scene staged in LDS  11 hit_world_cull (RTW_FLAG_GROUP_CULL)
 *       12 exact 64.64 fixed-point accumulation of 8 doubles
 *       13 hit_world_mfma (pass 1 on the matrix pipe: the trace kernel's plain scan), scene staged in LDS;
 *          tmin of ray 0 serves the whole launch, tmax is +inf
 *       14 the same with block culling (RTW_FLAG_GROUP_CULL on the matrix pipe), cull layout staged in LDS
 *       15 near_zero(v) (src/vec.jl:19-20)
 *       16 (Float32) the kernels' short correctly-rounded sqrt / reciprocal against the compiler's IEEE sequences on a range of binary32 bit patterns:
 *          in = (first pattern, count) per item, out = (mismatches sqrt, mismatches 1/x, first bad pattern of each or -1)
 *   bits 8-9 of `op`: the numerics mode of the ray-sphere test for ops 0, 8 - 11, 13, 14 (0 reference, 1 contract, 3 reference_fma2; 2 is rejected)  */
int rtw_unit_f32(int op, int count, const void *in, void *out, const rtw_scene_f32 *scene,
                 const rtw_camera_f32 *cam);
int rtw_unit_f64(int op, int count, const void *in, void *out, const rtw_scene_f64 *scene,
                 const rtw_camera_f64 *cam);

/* Frees the cached per-device render records (counters, events).  Optional.  Afterwards
 * rtw_stats() reports "no render" on every thread until that thread renders again. */
int rtw_shutdown(void);

#ifdef __cplusplus
}
#endif
#endif
