/*
 * rtw_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY; see rtw_oracle.h for the rules,
 * the pinning status ("parity unpinned" at the RandomNumbers.jl / StaticArrays / Base boundary)
 * and the numerics contract).
 *
 * Build: make -C oracle            (gcc -O2 -ffp-contract=off -fopenmp -mfma)
 */
#include "rtw_oracle.h"

#include <math.h>
#include <omp.h>
#include <stddef.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * Xoroshiro128Plus as shipped by RandomNumbers.jl 1.5.3 (src/proto/Manifest.toml:737-741 pins
 * the version; the package source is NOT under /root/reference).  Restated from its published
 * algorithm [UNVERIFIED here -- no Julia]: 2016 constants (55, 14, 36); integer seeds are
 * expanded with SplitMix64 into the two state words followed by one discarded output.
 * Reference call sites: src/init.jl:9 (Xoroshiro128Plus(i)), src/rand.jl:2 (seed!(rng, i)),
 * src/rand.jl:7,12 (rand(rng), rand(rng, T)).
 * ------------------------------------------------------------------------------------------ */
typedef struct { uint64_t x, y; } orng;
typedef struct { orng rng; uint64_t draws, segments; int numerics; } octx;
typedef struct { double r, g, b; } c3;
static _Thread_local uint64_t g_cand_disc = 0, g_cand_fwd = 0;   /* workload statistics */
static int g_unit_numerics = RTW_NUMERICS_REFERENCE;             /* what the unit-level exports use (rtwo_set_numerics) */
int rtwo_set_numerics(int mode) {
    if (mode < 0 || mode > RTW_NUMERICS_REFERENCE_FMA2) return -2;
    g_unit_numerics = mode;
    return 0;
}
int rtwo_get_numerics(void) { return g_unit_numerics; }

static inline uint64_t rotl64(uint64_t v, int k) { return (v << k) | (v >> (64 - k)); }

static inline uint64_t rng_next(orng *r) {
    uint64_t x = r->x, y = r->y;
    uint64_t out = x + y;
    uint64_t s1 = x ^ y;
    r->x = rotl64(x, 55) ^ s1 ^ (s1 << 14);
    r->y = rotl64(s1, 36);
    return out;
}

static inline uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

static inline void rng_seed_int(uint64_t seed, orng *r) {
    uint64_t s = seed;
    r->x = splitmix64(&s);
    r->y = splitmix64(&s);
    (void)rng_next(r);
}

/* PIXEL_STREAM: the independent stream of (render seed, pixel, sample chunk).  Not a reference
 * construct (the reference has one serial stream per Julia thread); it is the
 * partition-invariant keying the device uses, defined here and in DESIGN.md section 5. */
static inline void rng_stream(uint64_t seed, uint64_t pixel, uint64_t chunk, orng *r) {
    uint64_t s = seed ^ (0xd1b54a32d192ed03ULL * (pixel + 1)) ^ (0x8cb92ba72f3d8dd7ULL * (chunk + 1));
    r->x = splitmix64(&s);
    r->y = splitmix64(&s);
    (void)rng_next(r);
}

/* rand(rng, Float32): low 23 bits of the UInt64 output's low word into [1,2), minus 1
 * rand(rng, Float64): low 52 bits into [1,2), minus 1          (Julia 1.6/1.7 Random) */
static inline float rng_f32(orng *r) {
    uint32_t bits = ((uint32_t)rng_next(r) & 0x007fffffu) | 0x3f800000u;
    float f;
    memcpy(&f, &bits, 4);
    return f - 1.0f;
}
static inline double rng_f64(orng *r) {
    uint64_t bits = (rng_next(r) & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;
    double d;
    memcpy(&d, &bits, 8);
    return d - 1.0;
}

uint64_t rtwo_splitmix64(uint64_t *state) { return splitmix64(state); }
void rtwo_rng_seed(uint64_t seed, uint64_t state[2]) {
    orng r; rng_seed_int(seed, &r); state[0] = r.x; state[1] = r.y;
}
void rtwo_rng_stream(uint64_t seed, uint64_t pixel, uint64_t chunk, uint64_t state[2]) {
    orng r; rng_stream(seed, pixel, chunk, &r); state[0] = r.x; state[1] = r.y;
}
uint64_t rtwo_rng_next(uint64_t state[2]) {
    orng r = {state[0], state[1]}; uint64_t o = rng_next(&r); state[0] = r.x; state[1] = r.y; return o;
}
float rtwo_rng_f32(uint64_t state[2]) {
    orng r = {state[0], state[1]}; float o = rng_f32(&r); state[0] = r.x; state[1] = r.y; return o;
}
double rtwo_rng_f64(uint64_t state[2]) {
    orng r = {state[0], state[1]}; double o = rng_f64(&r); state[0] = r.x; state[1] = r.y; return o;
}

/* tand (src/camera.jl:23): Julia's tand is exact at multiples of 45 degrees and < 1 ulp
 * elsewhere; evaluated in extended precision and rounded once to T.  [UNVERIFIED vs Julia] */
static long double tand_ld(long double x) {
    long double m = fmodl(x, 180.0L);
    if (m == 0.0L) return 0.0L;
    if (m == 45.0L || m == -135.0L) return 1.0L;
    if (m == -45.0L || m == 135.0L) return -1.0L;
    return tanl(m * (3.14159265358979323846264338327950288L / 180.0L));
}
static inline float tand_f32(float x) { return (float)tand_ld((long double)x); }
static inline double tand_f64(double x) { return (double)tand_ld((long double)x); }

int rtwo_max_threads(void) { return omp_get_max_threads(); }

/* ------------------------------------------------------------------------------------------
 * PIXEL_STREAM pixel accumulation: the sample radiances of one pixel
 * are added EXACTLY -- each binary64 radiance is converted to a signed 64.64 fixed-point number
 * (every double of magnitude in [2^-11, 2^31) is represented exactly; smaller magnitudes are
 * truncated towards zero at 2^-64) and the fixed-point numbers are added as 128-bit integers.
 * The pixel sum is that integer rounded ONCE to binary64 (round to nearest, ties to even).
 * Integer addition is associative, so the result does not depend on the order in which the
 * samples finish -- which is what lets the device add them with LDS atomics in any order.
 * The reference itself adds the samples sequentially in Float64 (src/render.jl:29-39); this
 * is at least as accurate (one rounding per pixel instead of one per sample).
 * A radiance that is NaN, infinite or >= 2^31 in magnitude poisons the pixel (NaN output).
 * ------------------------------------------------------------------------------------------ */
typedef unsigned __int128 u128;
typedef struct { u128 v[3]; uint32_t poison; } fxacc;

static inline int fx_from_double(double x, u128 *out) {
    double a = fabs(x);
    if (!(a < 2147483648.0)) return 0;                       /* NaN, Inf, >= 2^31 */
    uint32_t ip = (uint32_t)a;                               /* trunc(|x|) < 2^31 */
    double fr = a - (double)ip;                              /* exact, in [0, 1) */
    double y = fr * 4294967296.0;                            /* exact scaling by 2^32 */
    uint32_t p1 = (uint32_t)y;                               /* bits 2^-1 .. 2^-32 */
    double r1 = y - (double)p1;                              /* exact, in [0, 1) */
    uint32_t p0 = (uint32_t)(r1 * 4294967296.0);             /* bits 2^-33 .. 2^-64, truncated */
    u128 m = ((u128)ip << 64) | ((u128)p1 << 32) | (u128)p0;
    *out = x < 0.0 ? (u128)0 - m : m;
    return 1;
}
static inline void fx_add(fxacc *acc, int ch, double x) {
    u128 q;
    if (fx_from_double(x, &q)) acc->v[ch] += q; else acc->poison++;
}
static inline double fx_to_double(u128 a) {
    int neg = (int)(a >> 127);
    if (neg) a = (u128)0 - a;
    if (a == 0) return 0.0;
    uint64_t hi = (uint64_t)(a >> 64), lo = (uint64_t)a;
    int p = hi ? 127 - __builtin_clzll(hi) : 63 - __builtin_clzll(lo);   /* index of the top set bit */
    double m;
    int sh = 0;
    if (p <= 52) {
        m = (double)lo;                                                  /* < 2^53: exact */
    } else {
        sh = p - 52;
        uint64_t mant = (uint64_t)(a >> sh);                             /* top 53 bits */
        u128 rem = a & ((((u128)1) << sh) - 1), half = ((u128)1) << (sh - 1);
        if (rem > half || (rem == half && (mant & 1))) mant++;           /* ties to even */
        m = (double)mant;                                                /* <= 2^53: exact */
    }
    double v = ldexp(m, sh - 64);
    return neg ? -v : v;
}

/* unit-level exports of the fixed-point accumulation (tests/test_oracle_kats.py) */
double rtwo_fx_sum(const double *x, int n, int *poisoned) {
    fxacc a; memset(&a, 0, sizeof a);
    for (int i = 0; i < n; ++i) fx_add(&a, 0, x[i]);
    if (poisoned) *poisoned = (int)a.poison;
    return a.poison ? NAN : fx_to_double(a.v[0]);
}


/* ---- Float32 (mixed precision) ------------------------------------------------ */
#define T float
#define SUF f32
#define SQRT_T sqrtf
#define FMA_T fmaf
#define RAND_T rng_f32
#define TAND_T tand_f32
#define T_INF ((float)INFINITY)
#define SCENE_T rtwo_scene_f32
#define CAMERA_T rtwo_camera_f32
#include "rtw_oracle_impl.h"
#undef T
#undef SUF
#undef SQRT_T
#undef FMA_T
#undef RAND_T
#undef TAND_T
#undef T_INF
#undef SCENE_T
#undef CAMERA_T

/* ---- Float64 ----------------------------------------------------------------------------- */
#define T double
#define SUF f64
#define SQRT_T sqrt
#define FMA_T fma
#define RAND_T rng_f64
#define TAND_T tand_f64
#define T_INF ((double)INFINITY)
#define SCENE_T rtwo_scene_f64
#define CAMERA_T rtwo_camera_f64
#include "rtw_oracle_impl.h"
