/*
 * rtw_oracle.h -- CPU ORACLE for the hot path  render -> ray_color -> hit/scatter
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and only as the
 * checker / CPU baseline.  The product (librtw_hip.so) never links or calls it.
 *
 * It is a plain-C restatement of the reference's algorithm (reference = /root/reference,
 * pure Julia; `julia` is not installed in this image, so the reference itself cannot be run
 * or built here -- there is no oracle/_ref).  Every function cites the reference file:line
 * it follows.
 *
 * PARITY PINNING STATUS
 *   pinned   : reflect KAT (test/runtests.jl:180), near_zero / rgb KATs (:131,:135),
 *              refract KATs (src/pluto_RayTracingWeekend.jl:603-615), hand-derivable KATs
 *              (SURVEY.md A.4).
 *   UNPINNED : "parity unpinned" for everything that depends on third-party arithmetic that
 *              is absent from /root/reference: RandomNumbers.jl 1.5.3 Xoroshiro128Plus
 *              (seeding + Float32/Float64 extraction), StaticArrays 1.2.13 normalize/dot,
 *              Julia Base tand / @fastmath contraction.  Their published algorithms are
 *              restated from memory; the reference holds no test or golden vector for them.
 *              tools/julia_kat.jl dumps the vectors that would pin them on a Julia box.
 *
 * NUMERICS (shared with the HIP kernels)
 *   - IEEE-754 binary32/binary64, round-to-nearest-even, correctly rounded + - * / sqrt,
 *     subnormals kept.  Compiled with -ffp-contract=off: NO implicit FMA anywhere.
 *   - One rounding per written operation, left to right as the reference writes it -- INCLUDING, since round 5, the
 *     ray-sphere discriminant of src/hit.jl:16-18 (RTW_NUMERICS_REFERENCE, the default):
 *        half_b = (oc.x*d.x + oc.y*d.y) + oc.z*d.z          StaticArrays' dot: a callee, @fastmath does not reach it
 *        c      = ((oc.x^2 + oc.y^2) + oc.z^2) - r*r
 *        disc   = half_b*half_b - c
 *   - NUMERICS MODES (rtwo_params.numerics; rtwo_set_numerics for the unit-level exports): the other evaluations an
 *     LLVM build of that function could produce, kept selectable so that one Julia run decides (tools/julia_kat.jl):
 *        RTW_NUMERICS_CONTRACT        rounds 1-4: half_b, r^2 - |oc|^2 and disc as three FMA chains
 *        RTW_NUMERICS_REFERENCE_FMA   un-fused dots, disc = fma(half_b, half_b, -c)   (oracle only since round 6: a mode of the library until ABI 3;
 *                                     kept here so that tools/check_julia_kat.py can still name it if a Julia build turns out to emit it)
 *        RTW_NUMERICS_REFERENCE_FMA2  ... and c = fma(-r, r, oc.oc) too
 *     In Float32 the choice is NOT noise: the contract form traces 4 % fewer segments per sample on
 *     scene_random_spheres and shifts the image mean by +0.003 (fewer tmin re-hits of the r = 1000 ground sphere).
 *   - Float32 mode is the reference's *mixed* precision: geometry, RNG floats and
 *     scatter in binary32; sky colour, attenuation product and pixel accumulation in binary64.
 */
#ifndef RTW_ORACLE_H
#define RTW_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material kinds (src/material.jl:3,25,37) */
enum { RTW_LAMBERTIAN = 0, RTW_METAL = 1, RTW_DIELECTRIC = 2 };

/* SoA scene: what a flattened HittableList of Sphere{T} looks like (src/structs.jl:31-35). */
typedef struct {
    int32_t n;
    const float *cx, *cy, *cz, *r;  /* centre, radius (radius may be negative)            */
    const int32_t *kind;            /* RTW_LAMBERTIAN / RTW_METAL / RTW_DIELECTRIC        */
    const float *ar, *ag, *ab;      /* albedo (unused for dielectric)                     */
    const float *param;             /* Metal: fuzz; Dielectric: ir; Lambertian: unused    */
} rtwo_scene_f32;

typedef struct {
    int32_t n;
    const double *cx, *cy, *cz, *r;
    const int32_t *kind;
    const double *ar, *ag, *ab;
    const double *param;
} rtwo_scene_f64;

/* Camera{T}: 22 scalars in the field order of src/camera.jl:2-9 */
typedef struct {
    float origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    float lens_radius;
} rtwo_camera_f32;

typedef struct {
    double origin[3], lower_left_corner[3], horizontal[3], vertical[3], u[3], v[3], w[3];
    double lens_radius;
} rtwo_camera_f64;

enum { RTW_RNG_PIXEL_STREAM = 0, RTW_RNG_REF_SERIAL = 1 };
enum { RTW_PRODUCT_REFERENCE = 0, RTW_PRODUCT_FORWARD = 1 };
/* the deciding arithmetic of hit(::Sphere) (src/hit.jl:16-18): see "NUMERICS MODES" above and sphere_test_ */
enum { RTW_NUMERICS_REFERENCE = 0, RTW_NUMERICS_CONTRACT = 1, RTW_NUMERICS_REFERENCE_FMA = 2, RTW_NUMERICS_REFERENCE_FMA2 = 3 };

typedef struct {
    int32_t width, height;   /* height = width div 16//9 (src/render.jl:11-12); both passed    */
    int32_t spp;             /* n_samples (src/render.jl:9)                                    */
    int32_t max_depth;       /* ray_color's depth (default 16, src/ray_color.jl:14)            */
    uint64_t seed;           /* PIXEL_STREAM: stream seed.  REF_SERIAL: ignored (thread index) */
    int32_t rng_mode;        /* RTW_RNG_*                                                      */
    int32_t ref_threads;     /* REF_SERIAL: Threads.nthreads() being mirrored                  */
    int32_t n_chunks;        /* PIXEL_STREAM: sample chunks per pixel (>=1), one RNG stream each; the
                                sample radiances are added exactly in 64.64 fixed point
                                (rtw_oracle.c fx_add), so the chunking only selects the streams */
    int32_t product_order;   /* RTW_PRODUCT_REFERENCE: att1*(att2*(...*sky)) as the recursion
                                unwinds (src/ray_color.jl:31); RTW_PRODUCT_FORWARD:
                                ((att1*att2)*...)*sky, what the iterative GPU loop computes   */
    int32_t omp_threads;     /* worker threads for the timing leg (<=0: all)                   */
    int32_t gamma;           /* 1: sqrt per channel (rgb_gamma2, src/vec.jl:22); 0: linear mean */
    int32_t numerics;        /* RTW_NUMERICS_*: the deciding arithmetic of hit(::Sphere)               */
} rtwo_params;

typedef struct {
    uint64_t samples;        /* pixel samples taken                                           */
    uint64_t segments;       /* calls of hit(world, ...) == ray segments traced               */
    uint64_t sphere_tests;   /* segments * n                                                  */
    uint64_t rng_draws;      /* u64 outputs consumed                                          */
    uint64_t cand_disc;      /* sphere tests with disc >= 0 (line meets sphere)               */
    uint64_t cand_forward;   /* ... of which half_b < 0 or origin inside (can yield a root)   */
} rtwo_stats;

/* render(): out is H x W x 3 of T, Julia column-major Matrix{RGB{T}}:
 * pixel (i,j), 1-based row i / column j, lives at ((j-1)*H + (i-1))*3 (src/render.jl:15,40). */
int rtwo_render_f32(const rtwo_scene_f32 *, const rtwo_camera_f32 *, const rtwo_params *,
                    float *out, rtwo_stats *stats);
int rtwo_render_f64(const rtwo_scene_f64 *, const rtwo_camera_f64 *, const rtwo_params *,
                    double *out, rtwo_stats *stats);

/* radiances of all spp samples of pixel (i, j) (1-based) in PIXEL_STREAM mode, sample order: out[3 s + channel] */
int rtwo_pixel_samples_f32(const rtwo_scene_f32 *, const rtwo_camera_f32 *, const rtwo_params *, int i, int j, double *out);
int rtwo_pixel_samples_f64(const rtwo_scene_f64 *, const rtwo_camera_f64 *, const rtwo_params *, int i, int j, double *out);

/* ---- RNG (src/init.jl:2-12, src/rand.jl:2-13; RandomNumbers.jl 1.5.3 restated) ---------- */
uint64_t rtwo_splitmix64(uint64_t *state);                            /* one SplitMix64 step: state += golden gamma, returns the mixed output */
void rtwo_rng_seed(uint64_t seed, uint64_t state[2]);                 /* Xoroshiro128Plus(seed) */
void rtwo_rng_stream(uint64_t seed, uint64_t pixel, uint64_t chunk, uint64_t state[2]);
uint64_t rtwo_rng_next(uint64_t state[2]);
float rtwo_rng_f32(uint64_t state[2]);
double rtwo_rng_f64(uint64_t state[2]);

/* ---- unit-level entry points (T0 parity tier) ------------------------------------------- */
void rtwo_reflect_f32(const float v[3], const float n[3], float out[3]);
void rtwo_reflect_f64(const double v[3], const double n[3], double out[3]);
void rtwo_refract_f32(const float d[3], const float n[3], float ratio, float out[3]);
void rtwo_refract_f64(const double d[3], const double n[3], double ratio, double out[3]);
float rtwo_reflectance_f32(float cos_theta, float ratio);
double rtwo_reflectance_f64(double cos_theta, double ratio);
int rtwo_near_zero_f32(const float v[3]);
int rtwo_near_zero_f64(const double v[3]);
void rtwo_skycolor_f32(const float dir[3], double out[3]);
void rtwo_skycolor_f64(const double dir[3], double out[3]);
/* hit(::Sphere): returns 1 on hit; rec = {t, p[3], n[3], front_face} (8 values) */
int rtwo_hit_sphere_f32(const float c[3], float r, const float o[3], const float d[3],
                        float tmin, float tmax, float rec[8]);
int rtwo_hit_sphere_f64(const double c[3], double r, const double o[3], const double d[3],
                        double tmin, double tmax, double rec[8]);
/* hit(::HittableList): returns index of the closest sphere or -1 */
/* n rays at once (rays = n x {o[3], d[3]}): idx[i] = -1 on a miss, t[i] = the hit distance */
void rtwo_hit_world_batch_f32(const rtwo_scene_f32 *, const float *rays, long n, float tmin, float tmax,
                              int32_t *idx, float *t);
void rtwo_hit_world_batch_f64(const rtwo_scene_f64 *, const double *rays, long n, double tmin, double tmax,
                              int32_t *idx, double *t);
int rtwo_hit_world_f32(const rtwo_scene_f32 *, const float o[3], const float d[3],
                       float tmin, float tmax, float rec[8]);
int rtwo_hit_world_f64(const rtwo_scene_f64 *, const double o[3], const double d[3],
                       double tmin, double tmax, double rec[8]);
/* scatter(): consumes randoms from state; returns 1; out = {origin[3], dir[3], att[3]} */
int rtwo_scatter_f32(int kind, const float albedo[3], float param, const float d[3],
                     const float rec[8], uint64_t state[2], float out[9]);
int rtwo_scatter_f64(int kind, const double albedo[3], double param, const double d[3],
                     const double rec[8], uint64_t state[2], double out[9]);
/* get_ray (src/camera.jl:43-48); out = {origin[3], dir[3]} */
void rtwo_get_ray_f32(const rtwo_camera_f32 *, float s, float t, uint64_t state[2], float out[6]);
void rtwo_get_ray_f64(const rtwo_camera_f64 *, double s, double t, uint64_t state[2], double out[6]);
/* ray_color for one ray (src/ray_color.jl:14-38); colour is binary64 in both modes (F5) */
void rtwo_ray_color_f32(const rtwo_scene_f32 *, const float o[3], const float d[3], int depth,
                        int product_order, uint64_t state[2], double out[3]);
void rtwo_ray_color_f64(const rtwo_scene_f64 *, const double o[3], const double d[3], int depth,
                        int product_order, uint64_t state[2], double out[3]);

/* ---- host-side producers mirrored for the Julia-less harness ----------------------------- */
/* default_camera (src/camera.jl:18-36) */
void rtwo_default_camera_f32(const float lookfrom[3], const float lookat[3], const float vup[3],
                             float vfov, float aspect, float aperture, float focus_dist,
                             rtwo_camera_f32 *out);
void rtwo_default_camera_f64(const double lookfrom[3], const double lookat[3], const double vup[3],
                             double vfov, double aspect, double aperture, double focus_dist,
                             rtwo_camera_f64 *out);
/* scene_random_spheres (src/scenes.jl:49-84) drawn from a fresh Xoroshiro128Plus(seed).
 * Arrays must hold >= 488 entries; returns the sphere count. */
int rtwo_scene_random_spheres_f32(uint64_t seed, float *cx, float *cy, float *cz, float *r,
                                  int32_t *kind, float *ar, float *ag, float *ab, float *param);
int rtwo_scene_random_spheres_f64(uint64_t seed, double *cx, double *cy, double *cz, double *r,
                                  int32_t *kind, double *ar, double *ag, double *ab, double *param);

int rtwo_max_threads(void);
/* the numerics mode of the unit-level exports above (hit_sphere, hit_world, hit_world_batch, ray_color); process-wide */
int rtwo_set_numerics(int mode);
int rtwo_get_numerics(void);
/* exact fixed-point sum of binary64 values, rounded once (the PIXEL_STREAM pixel accumulation) */
double rtwo_fx_sum(const double *x, int n, int *poisoned);

#ifdef __cplusplus
}
#endif
#endif
