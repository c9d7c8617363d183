/*
 * rtw_oracle_impl.h -- type-generic body of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 * Included twice by rtw_oracle.c with
 *     T    = float  / double          the reference's elem_type
 *     SUF  = f32    / f64
 *     SQRT_T, FMA_T                   sqrtf/sqrt, fmaf/fma
 *     RAND_T                          rng_f32 / rng_f64
 *     SCENE_T, CAMERA_T               rtwo_scene_f32 ...
 * Every function cites the reference file:line (paths relative to /root/reference) it restates.
 * Compiled with -ffp-contract=off: one rounding per written operation.
 */

#define CAT2_(a, b) a##b
#define CAT2(a, b) CAT2_(a, b)
#define FN(name) CAT2(name, SUF)

typedef struct { T x, y, z; } FN(v3_);
#define V3 FN(v3_)

/* src/vec.jl:3, StaticArrays SVector arithmetic: element-wise, one rounding each */
static inline V3 FN(vadd_)(V3 a, V3 b) { V3 r = {a.x + b.x, a.y + b.y, a.z + b.z}; return r; }
static inline V3 FN(vsub_)(V3 a, V3 b) { V3 r = {a.x - b.x, a.y - b.y, a.z - b.z}; return r; }
static inline V3 FN(vscale_)(T s, V3 a) { V3 r = {s * a.x, s * a.y, s * a.z}; return r; }
static inline V3 FN(vneg_)(V3 a) { V3 r = {-a.x, -a.y, -a.z}; return r; }
/* StaticArrays dot for length 3: (a1*b1 + a2*b2) + a3*b3, no FMA */
static inline T FN(dot_)(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline V3 FN(cross_)(V3 a, V3 b) {
    V3 r = {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
    return r;
}
/* StaticArrays 1.2.13 normalize(v) = inv(norm(v)) * v, norm = sqrt(sum abs2)  [UNVERIFIED] */
static inline V3 FN(normalize_)(V3 a) {
    T inv = (T)1 / SQRT_T(FN(dot_)(a, a));
    return FN(vscale_)(inv, a);
}
/* src/vec.jl:19-20: squared_length, near_zero compares against the Float64 literal 1e-5 */
static inline int FN(near_zero_)(V3 a) { return (double)FN(dot_)(a, a) < 1e-5; }

/* ---- RNG samplers (src/rand.jl:5-38) ----------------------------------------------------- */
static inline T FN(trand_)(octx *c) { c->draws++; return RAND_T(&c->rng); }
/* src/rand.jl:24  trand(T)*(max-min) + min */
static inline T FN(random_between_)(octx *c, T mn, T mx) { return FN(trand_)(c) * (mx - mn) + mn; }
/* src/rand.jl:15-22 with :25 (draw order x,y,z; boundary inclusive) */
static inline V3 FN(random_vec3_in_sphere_)(octx *c) {
    for (;;) {
        V3 p;
        p.x = FN(random_between_)(c, (T)-1, (T)1);
        p.y = FN(random_between_)(c, (T)-1, (T)1);
        p.z = FN(random_between_)(c, (T)-1, (T)1);
        if (FN(dot_)(p, p) <= (T)1) return p;
    }
}
/* src/rand.jl:29 */
static inline V3 FN(random_vec3_on_sphere_)(octx *c) {
    return FN(normalize_)(FN(random_vec3_in_sphere_)(c));
}
/* src/rand.jl:31-38 with :26 (draw order x,y) */
static inline void FN(random_vec2_in_disk_)(octx *c, T *px, T *py) {
    for (;;) {
        T x = FN(random_between_)(c, (T)-1, (T)1);
        T y = FN(random_between_)(c, (T)-1, (T)1);
        if (x * x + y * y <= (T)1) { *px = x; *py = y; return; }
    }
}

/* ---- intersection (src/hit.jl) ----------------------------------------------------------- */
typedef struct { T t; V3 p, n; int front; } FN(hitrec_);
#define HREC FN(hitrec_)

/* src/hit.jl:12-29: the quadratic and the root selection.  The deciding arithmetic is selectable (rtw_oracle.h,
 * "NUMERICS MODES"):
 *   RTW_NUMERICS_REFERENCE      what src/hit.jl:16-18 evaluates as written: `oc . r.dir` and `oc . oc` are calls of
 *                               StaticArrays' dot -- a callee, which @fastmath does not rewrite -- i.e. the un-fused
 *                               (x1 y1 + x2 y2) + x3 y3; `- s.radius^2` and `half_b^2 - a*c` are one rounding each
 *                               (x^2 = llvm.powi(x, 2) = x * x WITHOUT fast-math flags: no contraction with the fsub)
 *   RTW_NUMERICS_REFERENCE_FMA  the same, but the last step contracted: disc = fma(half_b, half_b, -c) -- what LLVM emits
 *                               on an FMA target if the square does carry the `contract` flag
 *   RTW_NUMERICS_REFERENCE_FMA2 ... and c = fma(-r, r, oc.oc) as well (both fsub-of-a-square sites contracted)
 *   RTW_NUMERICS_CONTRACT       rounds 1-4: three FMA chains (a legal @fastmath contraction only if @fastmath reached into dot) */
static inline __attribute__((always_inline)) int FN(sphere_test_)(int numerics, V3 c, T r, V3 o, V3 d, T tmin, T tmax, T *t_out) {
    V3 oc = FN(vsub_)(o, c);                                            /* :13 */
    T half_b, disc, nc;
    if (numerics == RTW_NUMERICS_CONTRACT) {
        half_b = FMA_T(oc.z, d.z, FMA_T(oc.y, d.y, oc.x * d.x));        /* :16 */
        nc = FMA_T(-oc.z, oc.z, FMA_T(-oc.y, oc.y, FMA_T(-oc.x, oc.x, r * r))); /* -(:17) */
        disc = FMA_T(half_b, half_b, nc);                               /* :18 (a == 1) */
    } else {
        half_b = FN(dot_)(oc, d);                                       /* :16  StaticArrays dot, no FMA */
        T ococ = FN(dot_)(oc, oc);
        T cc = numerics == RTW_NUMERICS_REFERENCE_FMA2 ? FMA_T(-r, r, ococ) : ococ - r * r;   /* :17 */
        disc = numerics == RTW_NUMERICS_REFERENCE ? half_b * half_b - cc : FMA_T(half_b, half_b, -cc);   /* :18 (a == 1: a*c == c) */
        nc = -cc;
    }
    if (disc < (T)0) return 0;                                          /* :19 */
    g_cand_disc++;                       /* workload statistics only */
    if (half_b < (T)0 || nc > (T)0) g_cand_fwd++;
    T sqrtd = SQRT_T(disc);                                             /* :20 */
    T root = -half_b - sqrtd;                                           /* :23 */
    if (root < tmin || tmax < root) {                                   /* :24 */
        root = -half_b + sqrtd;                                         /* :25 */
        if (root < tmin || tmax < root) return 0;                       /* :26-27 */
    }
    *t_out = root;                                                      /* :31 */
    return 1;
}
/* src/hit.jl:31-34 + ray_to_HitRecord :6-10 + point :3 */
static inline void FN(make_rec_)(V3 c, T r, V3 o, V3 d, T t, HREC *rec) {
    rec->t = t;
    rec->p = FN(vadd_)(o, FN(vscale_)(t, d));                           /* :32, :3 */
    V3 pc = FN(vsub_)(rec->p, c);
    V3 n_out = {pc.x / r, pc.y / r, pc.z / r};                          /* :33 (negative r flips) */
    rec->front = FN(dot_)(d, n_out) < (T)0;                             /* :7 */
    rec->n = rec->front ? n_out : FN(vneg_)(n_out);                     /* :8 */
}
static inline int FN(hit_sphere_)(int numerics, V3 c, T r, V3 o, V3 d, T tmin, T tmax, HREC *rec) {
    T t;
    if (!FN(sphere_test_)(numerics, c, r, o, d, tmin, tmax, &t)) return 0;
    FN(make_rec_)(c, r, o, d, t, rec);
    return 1;
}

/* src/hit.jl:38-50: closest hit, linear scan, `closest` shrinks, later sphere wins exact ties.
 * The reference materialises a HitRecord per accepted candidate; only the last one survives,
 * so it is built once for the winner (same values). */
static inline __attribute__((always_inline)) int FN(hit_world_n_)(const int numerics, const SCENE_T *w, V3 o, V3 d, T tmin, T tmax, HREC *best) {
    T closest = tmax;
    int idx = -1;
    for (int i = 0; i < w->n; ++i) {
        V3 c = {w->cx[i], w->cy[i], w->cz[i]};
        T t;
        if (FN(sphere_test_)(numerics, c, w->r[i], o, d, tmin, closest, &t)) { closest = t; idx = i; }
    }
    if (idx >= 0) {
        V3 c = {w->cx[idx], w->cy[idx], w->cz[idx]};
        FN(make_rec_)(c, w->r[idx], o, d, closest, best);
    }
    return idx;
}
/* (the mode is decided once per scan, outside the loop over the spheres: each call below inlines the loop with a CONSTANT mode) */
static inline int FN(hit_world_)(int numerics, const SCENE_T *w, V3 o, V3 d, T tmin, T tmax, HREC *best) {
    switch (numerics) {
        case RTW_NUMERICS_CONTRACT: return FN(hit_world_n_)(RTW_NUMERICS_CONTRACT, w, o, d, tmin, tmax, best);
        case RTW_NUMERICS_REFERENCE_FMA: return FN(hit_world_n_)(RTW_NUMERICS_REFERENCE_FMA, w, o, d, tmin, tmax, best);
        case RTW_NUMERICS_REFERENCE_FMA2: return FN(hit_world_n_)(RTW_NUMERICS_REFERENCE_FMA2, w, o, d, tmin, tmax, best);
        default: return FN(hit_world_n_)(RTW_NUMERICS_REFERENCE, w, o, d, tmin, tmax, best);
    }
}

/* ---- light transport (src/light.jl) ------------------------------------------------------ */
/* src/light.jl:6   v - (2v.n)*n */
static inline V3 FN(reflect_)(V3 v, V3 n) {
    V3 v2 = FN(vscale_)((T)2, v);
    T k = FN(dot_)(v2, n);
    return FN(vsub_)(v, FN(vscale_)(k, n));
}
/* src/light.jl:12-17 */
static inline V3 FN(refract_)(V3 dir, V3 n, T ratio) {
    T cos_t = -FN(dot_)(dir, n);
    if ((T)1 > cos_t) { /* min_fast(x,y) = ifelse(y > x, x, y) */ } else { cos_t = (T)1; }
    V3 perp = FN(vscale_)(ratio, FN(vadd_)(dir, FN(vscale_)(cos_t, n)));            /* :14 */
    T one_m = (T)1 - FN(dot_)(perp, perp);
    T par_s = -SQRT_T(one_m < (T)0 ? -one_m : one_m);                               /* :15 */
    V3 par = FN(vscale_)(par_s, n);
    return FN(normalize_)(FN(vadd_)(perp, par));                                    /* :16 */
}
/* src/light.jl:19-25 Schlick; ^2 and ^5 as powi: x*x, ((x*x)*(x*x))*x */
static inline T FN(reflectance_)(T cos_t, T ratio) {
    T r0 = ((T)1 - ratio) / ((T)1 + ratio);
    r0 = r0 * r0;
    T x = (T)1 - cos_t;
    T x2 = x * x;
    T x5 = (x2 * x2) * x;
    return r0 + ((T)1 - r0) * x5;
}

/* ---- materials (src/material.jl) --------------------------------------------------------- */
typedef struct { V3 o, d, att; } FN(scat_);
#define SCAT FN(scat_)

static inline SCAT FN(scatter_)(octx *c, int kind, V3 albedo, T param, V3 d_in, const HREC *rec) {
    SCAT s;
    s.o = rec->p;
    if (kind == RTW_LAMBERTIAN) {                                       /* src/material.jl:13-23 */
        V3 dir = FN(vadd_)(rec->n, FN(random_vec3_on_sphere_)(c));
        if (FN(near_zero_)(dir)) dir = rec->n; else dir = FN(normalize_)(dir);
        s.d = dir;
        s.att = albedo;
    } else if (kind == RTW_METAL) {                                     /* :31-34, never absorbs */
        V3 refl = FN(reflect_)(d_in, rec->n);
        V3 fz = FN(vscale_)(param, FN(random_vec3_on_sphere_)(c));
        s.d = FN(normalize_)(FN(vadd_)(refl, fz));
        s.att = albedo;
    } else {                                                            /* Dielectric :41-53 */
        V3 one = {(T)1, (T)1, (T)1};
        s.att = one;
        T ratio = rec->front ? ((T)1 / param) : param;                  /* :43 */
        T cos_t = -FN(dot_)(d_in, rec->n);                              /* :44 */
        if (!((T)1 > cos_t)) cos_t = (T)1;
        T sin_t = SQRT_T((T)1 - cos_t * cos_t);                         /* :45 */
        int cannot = ratio * sin_t > (T)1;                              /* :46 */
        /* :47 short-circuit: the random number is drawn only when refraction is possible */
        if (cannot || FN(reflectance_)(cos_t, ratio) > FN(trand_)(c))
            s.d = FN(reflect_)(d_in, rec->n);                           /* :48 (not re-normalised) */
        else
            s.d = FN(refract_)(d_in, rec->n, ratio);                    /* :50 */
    }
    return s;
}

/* ---- integrator (src/ray_color.jl) ------------------------------------------------------- */
/* src/ray_color.jl:1-6: Float64 constants; t and (1-t) in T, then promoted */
static inline c3 FN(skycolor_)(V3 d) {
    T t = (T)0.5 * (d.y + (T)1);
    T omt = (T)1 - t;
    c3 r = {(double)omt * 1.0 + (double)t * 0.5, (double)omt * 1.0 + (double)t * 0.7,
            (double)omt * 1.0 + (double)t * 1.0};
    return r;
}

static inline void FN(mat_of_)(const SCENE_T *w, int i, V3 *albedo, T *param) {
    albedo->x = w->ar[i]; albedo->y = w->ag[i]; albedo->z = w->ab[i];
    *param = w->param[i];
}

/* src/ray_color.jl:14-38, reference order: attenuation applied as the recursion unwinds */
static c3 FN(ray_color_rec_)(octx *c, const SCENE_T *w, V3 o, V3 d, int depth) {
    c3 zero = {0.0, 0.0, 0.0};
    if (depth <= 0) return zero;                                        /* :15-17 */
    HREC rec;
    c->segments++;
    int idx = FN(hit_world_)(c->numerics, w, o, d, (T)1e-4, T_INF, &rec); /* :19 */
    if (idx < 0) return FN(skycolor_)(d);                               /* :36 */
    V3 albedo; T param;
    FN(mat_of_)(w, idx, &albedo, &param);
    SCAT s = FN(scatter_)(c, w->kind[idx], albedo, param, d, &rec);     /* :29 */
    c3 in = FN(ray_color_rec_)(c, w, s.o, s.d, depth - 1);              /* :31 */
    c3 out = {(double)s.att.x * in.r, (double)s.att.y * in.g, (double)s.att.z * in.b};
    return out;
}

/* same light transport, product formed front to back: ((1*att1)*att2...)*sky */
static c3 FN(ray_color_fwd_)(octx *c, const SCENE_T *w, V3 o, V3 d, int depth) {
    c3 thr = {1.0, 1.0, 1.0};
    c3 zero = {0.0, 0.0, 0.0};
    for (;;) {
        if (depth <= 0) return zero;
        HREC rec;
        c->segments++;
        int idx = FN(hit_world_)(c->numerics, w, o, d, (T)1e-4, T_INF, &rec);
        if (idx < 0) {
            c3 sky = FN(skycolor_)(d);
            c3 out = {thr.r * sky.r, thr.g * sky.g, thr.b * sky.b};
            return out;
        }
        V3 albedo; T param;
        FN(mat_of_)(w, idx, &albedo, &param);
        SCAT s = FN(scatter_)(c, w->kind[idx], albedo, param, d, &rec);
        thr.r = thr.r * (double)s.att.x;
        thr.g = thr.g * (double)s.att.y;
        thr.b = thr.b * (double)s.att.z;
        o = s.o; d = s.d;
        depth -= 1;
    }
}

/* ---- camera (src/camera.jl:43-48) -------------------------------------------------------- */
static inline void FN(get_ray_)(octx *c, const CAMERA_T *cam, T s, T t, V3 *ro, V3 *rd_out) {
    T dx, dy;
    FN(random_vec2_in_disk_)(c, &dx, &dy);                              /* always drawn, :44 */
    T rx = cam->lens_radius * dx, ry = cam->lens_radius * dy;
    V3 cu = {cam->u[0], cam->u[1], cam->u[2]}, cv = {cam->v[0], cam->v[1], cam->v[2]};
    V3 org = {cam->origin[0], cam->origin[1], cam->origin[2]};
    V3 llc = {cam->lower_left_corner[0], cam->lower_left_corner[1], cam->lower_left_corner[2]};
    V3 hor = {cam->horizontal[0], cam->horizontal[1], cam->horizontal[2]};
    V3 ver = {cam->vertical[0], cam->vertical[1], cam->vertical[2]};
    V3 offset = FN(vadd_)(FN(vscale_)(rx, cu), FN(vscale_)(ry, cv));   /* :45 */
    *ro = FN(vadd_)(org, offset);
    V3 dir = FN(vadd_)(llc, FN(vscale_)(s, hor));                       /* :46-47 left to right */
    dir = FN(vadd_)(dir, FN(vscale_)(t, ver));
    dir = FN(vsub_)(dir, org);
    dir = FN(vsub_)(dir, offset);
    *rd_out = FN(normalize_)(dir);
}

/* one pixel sample (src/render.jl:30-38); s is 0-based, sample 0 is un-jittered */
static inline c3 FN(sample_)(octx *c, const SCENE_T *w, const CAMERA_T *cam, const rtwo_params *P,
                             T u, T v, int s) {
    T du = (T)0, dv = (T)0;
    if (s != 0) {
        du = FN(trand_)(c) / (T)(float)P->width;                        /* :34 (Float32 divisor) */
        dv = FN(trand_)(c) / (T)(float)P->height;                       /* :35 */
    }
    V3 o, d;
    FN(get_ray_)(c, cam, u + du, v + dv, &o, &d);                       /* :37 */
    return P->product_order == RTW_PRODUCT_FORWARD
               ? FN(ray_color_fwd_)(c, w, o, d, P->max_depth)
               : FN(ray_color_rec_)(c, w, o, d, P->max_depth);          /* :38 */
}

static inline void FN(store_)(const rtwo_params *P, T *out, int i, int j, c3 acc) {
    /* :40 rgb_gamma2(accum / n_samples), stored as RGB{T}; column-major H x W */
    double n = (double)P->spp;
    double r = acc.r / n, g = acc.g / n, b = acc.b / n;
    if (P->gamma) { r = sqrt(r); g = sqrt(g); b = sqrt(b); }
    size_t p = ((size_t)(j - 1) * (size_t)P->height + (size_t)(i - 1)) * 3;
    out[p] = (T)r; out[p + 1] = (T)g; out[p + 2] = (T)b;
}

/* src/render.jl:8-44 */
int FN(rtwo_render_)(const SCENE_T *w, const CAMERA_T *cam, const rtwo_params *P, T *out,
                     rtwo_stats *stats) {
    if (!w || !cam || !P || !out) return -1;
    if (P->width <= 0 || P->height <= 0 || P->spp <= 0 || w->n < 0) return -2;
    const int W = P->width, H = P->height;
    uint64_t tot_draws = 0, tot_segments = 0, tot_cd = 0, tot_cf = 0;
    int nthr = P->omp_threads > 0 ? P->omp_threads : omp_get_max_threads();

    if (P->rng_mode == RTW_RNG_REF_SERIAL) {
        /* Threads.@threads :static over rows (src/render.jl:23): thread k of n owns a
         * contiguous block, the first H mod n threads one extra row; its RNG is
         * Xoroshiro128Plus(k) freshly seeded by reseed!() (src/rand.jl:2, src/render.jl:21). */
        int n = P->ref_threads > 0 ? P->ref_threads : 1;
        int len = H / n, rem = H % n;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthr) reduction(+ : tot_draws, tot_segments, tot_cd, tot_cf)
        for (int k = 1; k <= n; ++k) {
            int f = 1 + (k - 1) * len, l = f + len - 1;
            if (rem > 0) {
                if (k <= rem) { f += k - 1; l += k; } else { f += rem; l += rem; }
            }
            octx c; c.draws = 0; c.segments = 0; c.numerics = P->numerics;
            g_cand_disc = 0; g_cand_fwd = 0;
            rng_seed_int((uint64_t)k, &c.rng);
            for (int i = f; i <= l; ++i)
                for (int j = 1; j <= W; ++j) {
                    c3 acc = {0.0, 0.0, 0.0};
                    T u = (T)((double)j / (double)W);                   /* :26 */
                    T v = (T)((double)(H - i) / (double)H);             /* :27 */
                    for (int s = 0; s < P->spp; ++s) {
                        c3 col = FN(sample_)(&c, w, cam, P, u, v, s);
                        acc.r += col.r; acc.g += col.g; acc.b += col.b;
                    }
                    FN(store_)(P, out, i, j, acc);
                }
            tot_draws += c.draws; tot_segments += c.segments; tot_cd += g_cand_disc; tot_cf += g_cand_fwd;
        }
    } else {
        /* PIXEL_STREAM: one independent Xoroshiro128+ stream per (pixel, sample chunk); the
         * radiances of ALL the pixel's samples are added exactly and the sum is rounded once.
         * This is what a parallel device can reproduce in any completion order. */
        int nch = P->n_chunks > 0 ? P->n_chunks : 1;
        int cs = (P->spp + nch - 1) / nch;
        int nch_eff = (P->spp + cs - 1) / cs;
        long npix = (long)W * H;
#pragma omp parallel for schedule(dynamic, 64) num_threads(nthr) reduction(+ : tot_draws, tot_segments, tot_cd, tot_cf)
        for (long pix = 0; pix < npix; ++pix) {
            int j = (int)(pix / H) + 1, i = (int)(pix % H) + 1;
            T u = (T)((double)j / (double)W);
            T v = (T)((double)(H - i) / (double)H);
            fxacc fx; memset(&fx, 0, sizeof fx);
            octx c; c.draws = 0; c.segments = 0; c.numerics = P->numerics;
            g_cand_disc = 0; g_cand_fwd = 0;
            for (int ch = 0; ch < nch_eff; ++ch) {
                rng_stream(P->seed, (uint64_t)pix, (uint64_t)ch, &c.rng);
                int s1 = (ch + 1) * cs < P->spp ? (ch + 1) * cs : P->spp;
                for (int s = ch * cs; s < s1; ++s) {
                    c3 col = FN(sample_)(&c, w, cam, P, u, v, s);
                    /* exact (order-independent) addition of the sample radiances: rtw_oracle.c fx_add */
                    fx_add(&fx, 0, col.r); fx_add(&fx, 1, col.g); fx_add(&fx, 2, col.b);
                }
            }
            c3 acc = {fx_to_double(fx.v[0]), fx_to_double(fx.v[1]), fx_to_double(fx.v[2])};
            if (fx.poison) acc.r = acc.g = acc.b = NAN;
            FN(store_)(P, out, i, j, acc);
            tot_draws += c.draws; tot_segments += c.segments; tot_cd += g_cand_disc; tot_cf += g_cand_fwd;
        }
    }
    if (stats) {
        stats->cand_disc = tot_cd; stats->cand_forward = tot_cf;
        stats->samples = (uint64_t)W * H * (uint64_t)P->spp;
        stats->segments = tot_segments;
        stats->sphere_tests = tot_segments * (uint64_t)w->n;
        stats->rng_draws = tot_draws;
    }
    return 0;
}

/* The radiance of every sample of pixel (i, j) (1-based, src/render.jl:23-24) in PIXEL_STREAM mode, in sample order:
 * out[3 s .. 3 s + 2].  For the tests that bound the exact pixel sum against the reference's running
 * `accum_color += ray_color(...)` (src/render.jl:38) on the SAME samples. */
int FN(rtwo_pixel_samples_)(const SCENE_T *w, const CAMERA_T *cam, const rtwo_params *P, int i, int j, double *out) {
    if (!w || !cam || !P || !out) return -1;
    const int W = P->width, H = P->height;
    if (i < 1 || i > H || j < 1 || j > W || P->spp <= 0) return -2;
    int nch = P->n_chunks > 0 ? P->n_chunks : 1;
    int cs = (P->spp + nch - 1) / nch;
    int nch_eff = (P->spp + cs - 1) / cs;
    long pix = (long)(j - 1) * H + (i - 1);
    T u = (T)((double)j / (double)W);
    T v = (T)((double)(H - i) / (double)H);
    octx c; c.draws = 0; c.segments = 0; c.numerics = P->numerics;
    for (int ch = 0; ch < nch_eff; ++ch) {
        rng_stream(P->seed, (uint64_t)pix, (uint64_t)ch, &c.rng);
        int s1 = (ch + 1) * cs < P->spp ? (ch + 1) * cs : P->spp;
        for (int s = ch * cs; s < s1; ++s) {
            c3 col = FN(sample_)(&c, w, cam, P, u, v, s);
            out[3 * s] = col.r; out[3 * s + 1] = col.g; out[3 * s + 2] = col.b;
        }
    }
    return 0;
}

/* ---- host-side producers ----------------------------------------------------------------- */
/* src/camera.jl:18-36 */
void FN(rtwo_default_camera_)(const T lookfrom[3], const T lookat[3], const T vup[3], T vfov,
                              T aspect, T aperture, T focus_dist, CAMERA_T *out) {
    T viewport_height = (T)2 * TAND_T(vfov / (T)2);                     /* :23 */
    T viewport_width = aspect * viewport_height;                        /* :24 */
    V3 lf = {lookfrom[0], lookfrom[1], lookfrom[2]}, la = {lookat[0], lookat[1], lookat[2]};
    V3 up = {vup[0], vup[1], vup[2]};
    V3 w = FN(normalize_)(FN(vsub_)(lf, la));                           /* :26 */
    V3 u = FN(normalize_)(FN(cross_)(up, w));                           /* :27 */
    V3 v = FN(cross_)(w, u);                                            /* :28 */
    V3 hor = FN(vscale_)(focus_dist * viewport_width, u);               /* :31 */
    V3 ver = FN(vscale_)(focus_dist * viewport_height, v);              /* :32 */
    V3 h2 = {hor.x / (T)2, hor.y / (T)2, hor.z / (T)2}, v2 = {ver.x / (T)2, ver.y / (T)2, ver.z / (T)2};
    V3 llc = FN(vsub_)(FN(vsub_)(FN(vsub_)(lf, h2), v2), FN(vscale_)(focus_dist, w)); /* :33 */
    out->origin[0] = lf.x; out->origin[1] = lf.y; out->origin[2] = lf.z;
    out->lower_left_corner[0] = llc.x; out->lower_left_corner[1] = llc.y; out->lower_left_corner[2] = llc.z;
    out->horizontal[0] = hor.x; out->horizontal[1] = hor.y; out->horizontal[2] = hor.z;
    out->vertical[0] = ver.x; out->vertical[1] = ver.y; out->vertical[2] = ver.z;
    out->u[0] = u.x; out->u[1] = u.y; out->u[2] = u.z;
    out->v[0] = v.x; out->v[1] = v.y; out->v[2] = v.z;
    out->w[0] = w.x; out->w[1] = w.y; out->w[2] = w.z;
    out->lens_radius = aperture / (T)2;                                 /* :34 */
}

/* src/scenes.jl:49-84, drawn from Xoroshiro128Plus(seed) (== reseed!() then build on thread 1) */
int FN(rtwo_scene_random_spheres_)(uint64_t seed, T *cx, T *cy, T *cz, T *r, int32_t *kind,
                                   T *ar, T *ag, T *ab, T *param) {
    octx c; c.draws = 0; c.segments = 0; c.numerics = 0;
    rng_seed_int(seed, &c.rng);
    int n = 0;
#define PUSH(X, Y, Z, R, K, A0, A1, A2, P)                                                   \
    do { cx[n] = (X); cy[n] = (Y); cz[n] = (Z); r[n] = (R); kind[n] = (K); ar[n] = (A0);      \
         ag[n] = (A1); ab[n] = (A2); param[n] = (P); ++n; } while (0)
    PUSH((T)0, (T)-1000, (T)-1, (T)1000, RTW_LAMBERTIAN, (T)0.5, (T)0.5, (T)0.5, (T)0); /* :53 */
    for (int a = -11; a <= 10; ++a)
        for (int b = -11; b <= 10; ++b) {                               /* :56 a outer, b inner */
            T choose = FN(trand_)(&c);                                  /* :57 */
            T x = (T)a + (T)0.9 * FN(trand_)(&c);                       /* :58 */
            T y = (T)0.2;
            T z = (T)b + (T)0.9 * FN(trand_)(&c);
            T dx = x - (T)4, dy = y - (T)0.2, dz = z - (T)0;
            if (SQRT_T((dx * dx + dy * dy) + dz * dz) < (T)0.9) continue; /* :61 */
            if (choose < (T)0.8) {                                      /* :63-66 */
                T a0 = FN(trand_)(&c), a1 = FN(trand_)(&c), a2 = FN(trand_)(&c);
                T b0 = FN(trand_)(&c), b1 = FN(trand_)(&c), b2 = FN(trand_)(&c);
                PUSH(x, y, z, (T)0.2, RTW_LAMBERTIAN, a0 * b0, a1 * b1, a2 * b2, (T)0);
            } else if (choose < (T)0.95) {                              /* :67-71 */
                T a0 = FN(random_between_)(&c, (T)0.5, (T)1.0);
                T a1 = FN(random_between_)(&c, (T)0.5, (T)1.0);
                T a2 = FN(random_between_)(&c, (T)0.5, (T)1.0);
                T fuzz = FN(random_between_)(&c, (T)0.0, (T)5.0);
                PUSH(x, y, z, (T)0.2, RTW_METAL, a0, a1, a2, fuzz);
            } else {                                                    /* :72-75 */
                PUSH(x, y, z, (T)0.2, RTW_DIELECTRIC, (T)1, (T)1, (T)1, (T)1.5);
            }
        }
    PUSH((T)0, (T)1, (T)0, (T)1, RTW_DIELECTRIC, (T)1, (T)1, (T)1, (T)1.5);          /* :78 */
    PUSH((T)-4, (T)1, (T)0, (T)1, RTW_LAMBERTIAN, (T)0.4, (T)0.2, (T)0.1, (T)0);     /* :79 */
    PUSH((T)4, (T)1, (T)0, (T)1, RTW_METAL, (T)0.7, (T)0.6, (T)0.5, (T)0);           /* :81 */
#undef PUSH
    return n;
}

/* ---- unit-level exports ------------------------------------------------------------------ */
static inline V3 FN(ld3_)(const T *p) { V3 r = {p[0], p[1], p[2]}; return r; }
static inline void FN(st3_)(T *p, V3 v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

void FN(rtwo_reflect_)(const T v[3], const T n[3], T out[3]) {
    FN(st3_)(out, FN(reflect_)(FN(ld3_)(v), FN(ld3_)(n)));
}
void FN(rtwo_refract_)(const T d[3], const T n[3], T ratio, T out[3]) {
    FN(st3_)(out, FN(refract_)(FN(ld3_)(d), FN(ld3_)(n), ratio));
}
T FN(rtwo_reflectance_)(T cos_theta, T ratio) { return FN(reflectance_)(cos_theta, ratio); }
int FN(rtwo_near_zero_)(const T v[3]) { return FN(near_zero_)(FN(ld3_)(v)); }
void FN(rtwo_skycolor_)(const T dir[3], double out[3]) {
    c3 c = FN(skycolor_)(FN(ld3_)(dir));
    out[0] = c.r; out[1] = c.g; out[2] = c.b;
}
static inline void FN(rec_out_)(const HREC *h, T rec[8]) {
    rec[0] = h->t; FN(st3_)(rec + 1, h->p); FN(st3_)(rec + 4, h->n); rec[7] = (T)h->front;
}
int FN(rtwo_hit_sphere_)(const T c[3], T r, const T o[3], const T d[3], T tmin, T tmax, T rec[8]) {
    HREC h;
    if (!FN(hit_sphere_)(g_unit_numerics, FN(ld3_)(c), r, FN(ld3_)(o), FN(ld3_)(d), tmin, tmax, &h)) return 0;
    FN(rec_out_)(&h, rec);
    return 1;
}
int FN(rtwo_hit_world_)(const SCENE_T *w, const T o[3], const T d[3], T tmin, T tmax, T rec[8]) {
    HREC h;
    int idx = FN(hit_world_)(g_unit_numerics, w, FN(ld3_)(o), FN(ld3_)(d), tmin, tmax, &h);
    if (idx >= 0) FN(rec_out_)(&h, rec);
    return idx;
}
/* batched closest hit for the randomized scan stress tests: rays = n x (o[3], d[3]); idx[i] = -1 on a miss */
void FN(rtwo_hit_world_batch_)(const SCENE_T *w, const T *rays, long n, T tmin, T tmax, int32_t *idx, T *t) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < n; ++i) {
        HREC h;
        h.t = (T)0;
        idx[i] = FN(hit_world_)(g_unit_numerics, w, FN(ld3_)(rays + 6 * i), FN(ld3_)(rays + 6 * i + 3), tmin, tmax, &h);
        t[i] = idx[i] >= 0 ? h.t : (T)0;
    }
}
int FN(rtwo_scatter_)(int kind, const T albedo[3], T param, const T d[3], const T rec[8],
                      uint64_t state[2], T out[9]) {
    octx c; c.draws = 0; c.segments = 0; c.numerics = g_unit_numerics; c.rng.x = state[0]; c.rng.y = state[1];
    HREC h; h.t = rec[0]; h.p = FN(ld3_)(rec + 1); h.n = FN(ld3_)(rec + 4); h.front = rec[7] != (T)0;
    SCAT s = FN(scatter_)(&c, kind, FN(ld3_)(albedo), param, FN(ld3_)(d), &h);
    FN(st3_)(out, s.o); FN(st3_)(out + 3, s.d); FN(st3_)(out + 6, s.att);
    state[0] = c.rng.x; state[1] = c.rng.y;
    return 1;
}
void FN(rtwo_get_ray_)(const CAMERA_T *cam, T s, T t, uint64_t state[2], T out[6]) {
    octx c; c.draws = 0; c.segments = 0; c.numerics = g_unit_numerics; c.rng.x = state[0]; c.rng.y = state[1];
    V3 o, d;
    FN(get_ray_)(&c, cam, s, t, &o, &d);
    FN(st3_)(out, o); FN(st3_)(out + 3, d);
    state[0] = c.rng.x; state[1] = c.rng.y;
}
void FN(rtwo_ray_color_)(const SCENE_T *w, const T o[3], const T d[3], int depth, int product_order,
                         uint64_t state[2], double out[3]) {
    octx c; c.draws = 0; c.segments = 0; c.numerics = g_unit_numerics; c.rng.x = state[0]; c.rng.y = state[1];
    c3 col = product_order == RTW_PRODUCT_FORWARD
                 ? FN(ray_color_fwd_)(&c, w, FN(ld3_)(o), FN(ld3_)(d), depth)
                 : FN(ray_color_rec_)(&c, w, FN(ld3_)(o), FN(ld3_)(d), depth);
    out[0] = col.r; out[1] = col.g; out[2] = col.b;
    state[0] = c.rng.x; state[1] = c.rng.y;
}

#undef V3
#undef HREC
#undef SCAT
#undef FN
#undef CAT2
#undef CAT2_
