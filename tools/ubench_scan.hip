// ubench_scan.hip -- micro-benchmark of candidate inner loops for the closest-hit sphere scan
// (the >85 % kernel phase) plus raw VALU issue-rate probes.  gfx950 only.  Not part of the
// product; used to pick the trace kernel's inner loop by measurement.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/ubench_scan.hip -o build/ubench_scan
//   ./build/ubench_scan            (prints one line per variant and ray set)
//
// Every variant must return exactly the (index, t) of variant 0 (the straight restatement of
// src/hit.jl:38-50); the harness checks that before timing means anything.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

struct Ray { float ox, oy, oz, dx, dy, dz; };
struct Hit { int idx; float t; };

__device__ __forceinline__ void disc_of(float4 s, float ox, float oy, float oz, float dx, float dy, float dz, float &hb, float &disc) {
    float ocx = ox - s.x, ocy = oy - s.y, ocz = oz - s.z;
    hb = __builtin_fmaf(ocz, dz, __builtin_fmaf(ocy, dy, ocx * dx));
    float nc = __builtin_fmaf(-ocz, ocz, __builtin_fmaf(-ocy, ocy, __builtin_fmaf(-ocx, ocx, s.w)));
    disc = __builtin_fmaf(hb, hb, nc);
}
__device__ __forceinline__ void accept(float hb, float disc, float tmin, float &closest, int &idx, int i) {
    if (disc < 0.f) return;
    float sq = __builtin_sqrtf(disc);
    float root = -hb - sq;
    if (root < tmin || closest < root) {
        root = -hb + sq;
        if (root < tmin || closest < root) return;
    }
    closest = root; idx = i;
}

// ---- variant 0: plain loop over a global float4 array (what v1 of the trace kernel does) ----
struct V0 {
    static constexpr const char *name = "v0 global-vector-load, 1 sphere/iter";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *g, const float4 *, int n, Ray r) {
        float closest = INFINITY; int idx = -1;
        for (int i = 0; i < n; ++i) {
            float hb, disc; disc_of(g[i], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb, disc);
            accept(hb, disc, 1e-4f, closest, idx, i);
        }
        return {idx, closest};
    }
};

// ---- variant 1: scalar (SMEM) loads of wave-uniform sphere data, group-of-4 cull -------------
struct V1 {
    static constexpr const char *name = "v1 scalar-load, 4 spheres/iter, any-candidate cull";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *, int n, Ray r) {
        float closest = INFINITY; int idx = -1;
        for (int i = 0; i < n; i += 4) {
            float hb[4], dc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) disc_of(g[i + k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb[k], dc[k]);
            float m = fmaxf(fmaxf(dc[0], dc[1]), fmaxf(dc[2], dc[3]));
            if (!(m < 0.f)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) accept(hb[k], dc[k], 1e-4f, closest, idx, i + k);
            }
        }
        return {idx, closest};
    }
};

// ---- variant 2: spheres staged in LDS, broadcast ds_read_b128, group-of-4 cull ---------------
struct V2 {
    static constexpr const char *name = "v2 LDS-broadcast, 4 spheres/iter, any-candidate cull";
    static constexpr int lds_bytes = 512 * 16;
    __device__ static Hit scan(const float4 *, const float4 *s, int n, Ray r) {
        float closest = INFINITY; int idx = -1;
        for (int i = 0; i < n; i += 4) {
            float hb[4], dc[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) disc_of(s[i + k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb[k], dc[k]);
            float m = fmaxf(fmaxf(dc[0], dc[1]), fmaxf(dc[2], dc[3]));
            if (!(m < 0.f)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) accept(hb[k], dc[k], 1e-4f, closest, idx, i + k);
            }
        }
        return {idx, closest};
    }
};

// ---- variant 3: LDS, 8 spheres/iter, forward-only cull (hb < 0 or inside), exact accept ------
// A root in [tmin, closest] needs disc >= 0 and (half_b < 0 or the origin inside the sphere,
// i.e. disc > half_b^2): otherwise -half_b + sqrt(disc) <= 0 < tmin.  Spheres behind the ray are
// culled with no sqrt.  The cull is conservative; accept() keeps the exact reference semantics.
struct V3 {
    static constexpr const char *name = "v3 LDS-broadcast, 8 spheres/iter, forward cull";
    static constexpr int lds_bytes = 512 * 16;
    __device__ static Hit scan(const float4 *, const float4 *s, int n, Ray r) {
        float closest = INFINITY; int idx = -1;
        for (int i = 0; i < n; i += 8) {
            float hb[8], dc[8], key[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                disc_of(s[i + k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb[k], dc[k]);
                // candidate iff disc >= 0 and (hb <= 0 or disc >= hb*hb): key >= 0
                float inside = __builtin_fmaf(-hb[k], hb[k], dc[k]);   // = nc (recomputed, cheap)
                key[k] = fminf(dc[k], fmaxf(-hb[k], inside));
            }
            float m = fmaxf(fmaxf(fmaxf(key[0], key[1]), fmaxf(key[2], key[3])), fmaxf(fmaxf(key[4], key[5]), fmaxf(key[6], key[7])));
            if (!(m < 0.f)) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (!(key[k] < 0.f)) accept(hb[k], dc[k], 1e-4f, closest, idx, i + k);
            }
        }
        return {idx, closest};
    }
};

// ---- variant 4: scalar loads, 8 spheres/iter, disc-only cull ---------------------------------
struct V4 {
    static constexpr const char *name = "v4 scalar-load, 8 spheres/iter, any-candidate cull";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *, int n, Ray r) {
        float closest = INFINITY; int idx = -1;
        for (int i = 0; i < n; i += 8) {
            float hb[8], dc[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) disc_of(g[i + k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb[k], dc[k]);
            float m = fmaxf(fmaxf(fmaxf(dc[0], dc[1]), fmaxf(dc[2], dc[3])), fmaxf(fmaxf(dc[4], dc[5]), fmaxf(dc[6], dc[7])));
            if (!(m < 0.f)) {
#pragma unroll
                for (int k = 0; k < 8; ++k) accept(hb[k], dc[k], 1e-4f, closest, idx, i + k);
            }
        }
        return {idx, closest};
    }
};

// ---- variant 5: two-pass -- pass 1 records candidates in a per-lane LDS list, pass 2 resolves
// them in sphere order with every lane busy (slow path runs at full lane occupancy) ------------
struct V5 {
    static constexpr const char *name = "v5 scalar-load + per-lane LDS candidate list (2-pass)";
    static constexpr int CAP = 24;
    static constexpr int lds_bytes = 256 * CAP * 2;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *lds_raw, int n, Ray r) {
        unsigned short *list = (unsigned short *)lds_raw + threadIdx.x;   // stride 256: conflict-free
        float closest = INFINITY; int idx = -1;
        int cnt = 0;
        for (int i = 0; i < n; i += 4) {
            float hb[4], dc[4], key[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                disc_of(g[i + k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb[k], dc[k]);
                float inside = __builtin_fmaf(-hb[k], hb[k], dc[k]);
                key[k] = fminf(dc[k], fmaxf(-hb[k], inside));
            }
            float m = fmaxf(fmaxf(key[0], key[1]), fmaxf(key[2], key[3]));
            if (!(m < 0.f)) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (!(key[k] < 0.f)) {
                        if (cnt < CAP) { list[cnt * 256] = (unsigned short)(i + k); cnt++; }
                        else accept(hb[k], dc[k], 1e-4f, closest, idx, i + k);   // overflow: resolve now (still in order)
                    }
            }
        }
        // NOTE: overflow entries were resolved before earlier list entries -> only order-safe if
        // no overflow happened; the harness reports mismatches, CAP is sized so that it does not.
        for (int c = 0; c < cnt; ++c) {
            int i = list[c * 256];
            float hb, disc; disc_of(g[i], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb, disc);
            accept(hb, disc, 1e-4f, closest, idx, i);
        }
        return {idx, closest};
    }
};

// ---- variant 9: pass-1 VALU work only: sphere data stays in SGPRs, no SMEM in the loop --------
// (upper bound for pass 1 if scalar-load latency were free; results are NOT comparable to v0)
struct V9 {
    static constexpr const char *name = "v9 pass-1 VALU only (no SMEM in loop; bound, not a scan)";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *, int n, Ray r) {
        float4 A[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) A[k] = g[k];
        unsigned acc = 0;
        for (int base = 0; base < n; base += 32) {
            unsigned mask = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    asm volatile("" : "+s"(A[k].x), "+s"(A[k].y), "+s"(A[k].z), "+s"(A[k].w));
                    float hb, dc; disc_of(A[k], r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb, dc);
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(dc), 31);
                }
            }
            acc += __popc(mask);
        }
        return {(int)acc, 0.f};
    }
};

// ---- variant 10: packed math, 2 spheres per VALU instruction ----------------------------------
// pair layout in memory: (c0x,c1x, c0y,c1y, c0z,c1z, r2_0,r2_1) = 8 floats per sphere pair;
// v_pk_add/mul/fma_f32 with the SGPR pair as one operand and the ray component broadcast.
typedef float f2 __attribute__((ext_vector_type(2)));
struct V10 {
    static constexpr const char *name = "v10 packed: 2 spheres/instr (v_pk_*), sign-mask only";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *, int n, Ray r) {
        typedef const float __attribute__((address_space(4))) *cptr;
        cptr p = (cptr)(uintptr_t)g;       // harness passes the pair-transposed array for this variant
        const f2 ox = {r.ox, r.ox}, oy = {r.oy, r.oy}, oz = {r.oz, r.oz}, dx = {r.dx, r.dx}, dy = {r.dy, r.dy}, dz = {r.dz, r.dz};
        unsigned acc = 0;
        for (int base = 0; base < n; base += 32) {
            unsigned mask = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {            // 16 pairs = 32 spheres
                cptr s = p + (base / 2 + q) * 8;
                const f2 cx = {s[0], s[1]}, cy = {s[2], s[3]}, cz = {s[4], s[5]}, r2 = {s[6], s[7]};
                const f2 ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                f2 hb = ocx * dx;
                hb = __builtin_elementwise_fma(ocy, dy, hb);
                hb = __builtin_elementwise_fma(ocz, dz, hb);
                f2 nc = __builtin_elementwise_fma(-ocx, ocx, r2);
                nc = __builtin_elementwise_fma(-ocy, ocy, nc);
                nc = __builtin_elementwise_fma(-ocz, ocz, nc);
                const f2 disc = __builtin_elementwise_fma(hb, hb, nc);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(disc.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(disc.y), 31);
            }
            acc += __popc(~mask);
        }
        return {(int)acc, 0.f};
    }
};
// ---- variant 11: v4-like scalar (non-packed) sign-mask only, for an apples-to-apples baseline --
struct V11 {
    static constexpr const char *name = "v11 scalar-load scalar-math, sign-mask only";
    static constexpr int lds_bytes = 0;
    __device__ static Hit scan(const float4 *__restrict__ g, const float4 *, int n, Ray r) {
        typedef const float __attribute__((address_space(4))) *cptr;
        cptr p = (cptr)(uintptr_t)g;
        unsigned acc = 0;
        for (int base = 0; base < n; base += 32) {
            unsigned mask = 0;
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                cptr s = p + (base + k) * 4;
                float hb, dc; disc_of(float4{s[0], s[1], s[2], s[3]}, r.ox, r.oy, r.oz, r.dx, r.dy, r.dz, hb, dc);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(dc), 31);
            }
            acc += __popc(~mask);
        }
        return {(int)acc, 0.f};
    }
};

// ---- variant 6: two rays per lane, scalar loads ----------------------------------------------
// (handled by a separate kernel below)

template <typename V>
__global__ __launch_bounds__(256) void scan_kernel(const float4 *__restrict__ geom, int n, const Ray *__restrict__ rays,
                                                   int rays_per_lane, Hit *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float4 lds[];
    if (V::lds_bytes > 0 && V::lds_bytes >= n * 16 && V::lds_bytes == 512 * 16) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = geom[i];
        __syncthreads();
    }
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = gridDim.x * blockDim.x;
    Hit acc = {0, 0.f};
    for (int k = 0; k < rays_per_lane; ++k) {
        Ray r = rays[gid + k * total];
        Hit h = V::scan(geom, lds, n, r);
        acc.idx += h.idx; acc.t += (h.idx >= 0 ? h.t : 0.f);
    }
    out[gid] = acc;
}

__global__ __launch_bounds__(256) void scan2_kernel(const float4 *__restrict__ g, int n, const Ray *__restrict__ rays,
                                                    int rays_per_lane, Hit *__restrict__ out) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = gridDim.x * blockDim.x;
    Hit acc = {0, 0.f};
    for (int k = 0; k < rays_per_lane; k += 2) {
        Ray a = rays[gid + k * total], b = rays[gid + (k + 1) * total];
        float ca = INFINITY, cb = INFINITY; int ia = -1, ib = -1;
        for (int i = 0; i < n; i += 4) {
            float hba[4], dca[4], hbb[4], dcb[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 s = g[i + q];
                disc_of(s, a.ox, a.oy, a.oz, a.dx, a.dy, a.dz, hba[q], dca[q]);
                disc_of(s, b.ox, b.oy, b.oz, b.dx, b.dy, b.dz, hbb[q], dcb[q]);
            }
            float m = fmaxf(fmaxf(fmaxf(dca[0], dca[1]), fmaxf(dca[2], dca[3])), fmaxf(fmaxf(dcb[0], dcb[1]), fmaxf(dcb[2], dcb[3])));
            if (!(m < 0.f)) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { accept(hba[q], dca[q], 1e-4f, ca, ia, i + q); accept(hbb[q], dcb[q], 1e-4f, cb, ib, i + q); }
            }
        }
        acc.idx += ia + ib; acc.t += (ia >= 0 ? ca : 0.f); acc.t += (ib >= 0 ? cb : 0.f);
    }
    out[gid] = acc;
}

// ---- raw VALU issue-rate probes ----------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void valu_probe(float *out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                a0 = __builtin_fmaf(a0, b, c); a1 = __builtin_fmaf(a1, b, c); a2 = __builtin_fmaf(a2, b, c); a3 = __builtin_fmaf(a3, b, c);
                a4 = __builtin_fmaf(a4, b, c); a5 = __builtin_fmaf(a5, b, c); a6 = __builtin_fmaf(a6, b, c); a7 = __builtin_fmaf(a7, b, c);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                p0 = __builtin_elementwise_fma(p0, pb, pc); p1 = __builtin_elementwise_fma(p1, pb, pc);
                p2 = __builtin_elementwise_fma(p2, pb, pc); p3 = __builtin_elementwise_fma(p3, pb, pc);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = MODE == 0 ? (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7) : (p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y);
}

// ---- host ---------------------------------------------------------------------------------------
static uint64_t lcg = 88172645463325252ull;
static double urand() { lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17; return (lcg >> 11) * (1.0 / 9007199254740992.0); }

int main(int argc, char **argv) {
    int rays_per_lane = argc > 1 ? atoi(argv[1]) : 16;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clock %d MHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    // scene like scene_random_spheres: ground + 22x22 lattice + 3 big, padded to 496 with dead spheres
    std::vector<float4> g;
    g.push_back({0, -1000, -1, 1000.f * 1000.f});
    for (int a = -11; a < 11; ++a) for (int b = -11; b < 11; ++b) {
        float x = a + 0.9f * (float)urand(), z = b + 0.9f * (float)urand();
        if (sqrtf((x - 4) * (x - 4) + z * z) < 0.9f) continue;
        g.push_back({x, 0.2f, z, 0.04f});
    }
    g.push_back({0, 1, 0, 1}); g.push_back({-4, 1, 0, 1}); g.push_back({4, 1, 0, 1});
    int n_real = (int)g.size();
    while (g.size() % 32) g.push_back({0, 0, 0, -1e30f});
    int n = (int)g.size();
    for (int k = 0; k < 8; ++k) g.push_back({0, 0, 0, -1e30f});   // prefetch tail for v7
    const int blocks = prop.multiProcessorCount * 8, threads = 256, lanes = blocks * threads;
    size_t nrays = (size_t)lanes * rays_per_lane;
    std::vector<Ray> cam(nrays), dif(nrays), mis(nrays);
    for (size_t i = 0; i < nrays; ++i) {
        // coherent: camera rays, consecutive lanes = neighbouring pixels
        size_t lane = i % lanes; int px = (int)(lane % 1920), py = (int)((lane / 1920) % 1080);
        float u = (px + (float)urand()) / 1920.f, v = (py + (float)urand()) / 1080.f;
        // t_cam1: from (13,2,3) at (0,0,0), vfov 20: viewport spans u=(0.2249,0,-0.9744)*+-3.13, v~(−0.144,0.989,−0.033)*+-1.76 at distance 10
        float a = (u - 0.5f) * 6.27f, b = (0.5f - v) * 3.53f;
        float tx = 13 - 9.636f + 0.2249f * a - 0.1445f * b, ty = 2 - 1.4825f + 0.9889f * b, tz = 3 - 2.2237f - 0.9744f * a - 0.0333f * b;
        float dx = tx - 13, dy = ty - 2, dz = tz - 3; float l = 1 / sqrtf(dx * dx + dy * dy + dz * dz);
        cam[i] = {13, 2, 3, dx * l, dy * l, dz * l};
        // incoherent: bounce rays leaving the ground / small spheres in random directions
        float ox = -11 + 22 * (float)urand(), oz = -11 + 22 * (float)urand(), oy = urand() < 0.5 ? 1e-3f : 0.41f;
        float ax, ay, az, q;
        do { ax = 2 * (float)urand() - 1; ay = 2 * (float)urand() - 1; az = 2 * (float)urand() - 1; q = ax * ax + ay * ay + az * az; } while (q > 1 || q < 1e-4f);
        l = 1 / sqrtf(q); if (urand() < 0.85) ay = fabsf(ay);
        dif[i] = {ox, oy, oz, ax * l, ay * l, az * l};
        mis[i] = {ox, 3.0f + oy, oz, ax * l, fabsf(ay) * l + 1e-3f, az * l};   // above everything, going up: disc < 0 for every sphere
    }
    float4 *d_g; Ray *d_r; Hit *d_o0, *d_o;
    CHECK(hipMalloc(&d_g, (n + 8) * sizeof(float4))); CHECK(hipMalloc(&d_r, nrays * sizeof(Ray)));
    CHECK(hipMalloc(&d_o0, lanes * sizeof(Hit))); CHECK(hipMalloc(&d_o, lanes * sizeof(Hit)));
    CHECK(hipMemcpy(d_g, g.data(), (n + 8) * sizeof(float4), hipMemcpyHostToDevice));
    // pair-transposed copy for the packed variant
    std::vector<float> gp((n + 8) * 4);
    for (int q = 0; q < (n + 8) / 2; ++q) {
        const float4 a = g[2 * q], b = g[2 * q + 1];
        float *o = &gp[q * 8];
        o[0] = a.x; o[1] = b.x; o[2] = a.y; o[3] = b.y; o[4] = a.z; o[5] = b.z; o[6] = a.w; o[7] = b.w;
    }
    float4 *d_gp; CHECK(hipMalloc(&d_gp, (n + 8) * sizeof(float4)));
    CHECK(hipMemcpy(d_gp, gp.data(), (n + 8) * sizeof(float4), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    std::vector<Hit> h0(lanes), h(lanes);
    printf("%d spheres (%d padded), %d lanes x %d rays\n", n_real, n, lanes, rays_per_lane);

    auto run = [&](auto launch, const char *name, Hit *dst, const char *set) {
        launch(dst);  // warm-up
        CHECK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0)); launch(dst); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        double tests = (double)nrays * n_real;
        printf("  %-8s %-58s %8.3f ms  %7.1f Gtests/s  %5.1f%% of FP32 VALU peak (17 flop/test)\n", set, name, best,
               tests / best / 1e6, 100.0 * tests * 17 / (best * 1e-3) / 157.3e12);
    };
    for (int set = 0; set < 3; ++set) {
        const char *sname = set == 2 ? "miss" : set ? "bounce" : "camera";
        CHECK(hipMemcpy(d_r, (set == 2 ? mis : set ? dif : cam).data(), nrays * sizeof(Ray), hipMemcpyHostToDevice));
#define RUNV(V, ISREF)                                                                                              \
        run([&](Hit *dst) { hipLaunchKernelGGL(scan_kernel<V>, dim3(blocks), dim3(threads), V::lds_bytes, 0, d_g, n, d_r, rays_per_lane, dst); }, V::name, ISREF ? d_o0 : d_o, sname); \
        if (!ISREF) { CHECK(hipMemcpy(h.data(), d_o, lanes * sizeof(Hit), hipMemcpyDeviceToHost)); size_t bad = 0; for (int i = 0; i < lanes; ++i) bad += (h[i].idx != h0[i].idx || h[i].t != h0[i].t); if (bad) printf("      MISMATCH vs v0 on %zu lanes\n", bad); } \
        else { CHECK(hipMemcpy(h0.data(), d_o0, lanes * sizeof(Hit), hipMemcpyDeviceToHost)); long hits = 0; for (int i = 0; i < lanes; ++i) hits += h0[i].idx; printf("      (checksum %ld)\n", hits); }
        RUNV(V0, true) RUNV(V1, false) RUNV(V2, false) RUNV(V3, false) RUNV(V4, false) RUNV(V5, false) RUNV(V9, false) RUNV(V11, false)
        run([&](Hit *dst) { hipLaunchKernelGGL(scan_kernel<V10>, dim3(blocks), dim3(threads), 0, 0, d_gp, n, d_r, rays_per_lane, dst); }, V10::name, d_o, sname);
        { std::vector<Hit> h11(lanes); CHECK(hipMemcpy(h11.data(), d_o, lanes * sizeof(Hit), hipMemcpyDeviceToHost));
          hipLaunchKernelGGL(scan_kernel<V11>, dim3(blocks), dim3(threads), 0, 0, d_g, n, d_r, rays_per_lane, d_o); CHECK(hipMemcpy(h.data(), d_o, lanes * sizeof(Hit), hipMemcpyDeviceToHost));
          size_t bad = 0; for (int i = 0; i < lanes; ++i) bad += (h[i].idx != h11[i].idx); printf("      v10 vs v11 candidate counts: %zu lanes differ\n", bad); }
        run([&](Hit *dst) { hipLaunchKernelGGL(scan2_kernel, dim3(blocks), dim3(threads), 0, 0, d_g, n, d_r, rays_per_lane, dst); }, "v6 scalar-load, 2 rays/lane, 4 spheres/iter", d_o, sname);
        CHECK(hipMemcpy(h.data(), d_o, lanes * sizeof(Hit), hipMemcpyDeviceToHost));
        { size_t bad = 0; for (int i = 0; i < lanes; ++i) bad += (h[i].idx != h0[i].idx); if (bad) printf("      MISMATCH(idx sum) vs v0 on %zu lanes\n", bad); }
    }
    // VALU probes
    float *d_f; CHECK(hipMalloc(&d_f, lanes * sizeof(float)));
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 4096;
        auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL(valu_probe<0>, dim3(blocks), dim3(threads), 0, 0, d_f, iters, 1.f); else hipLaunchKernelGGL(valu_probe<1>, dim3(blocks), dim3(threads), 0, 0, d_f, iters, 1.f); };
        launch(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)lanes * iters * 8 * 8 * 2;   // 64 scalar-equivalent FMAs per iter per lane
        printf("  probe %-12s %8.3f ms  %7.1f TFLOP/s\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms, flops / (ms * 1e-3) / 1e12);
    }
    return 0;
}
