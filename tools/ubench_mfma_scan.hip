// ubench_mfma_scan.hip -- can pass 1 of the closest-hit scan run on the matrix pipe?
// The discriminant hb^2 - |o-c|^2 + r^2 is bilinear in (ray features) x (sphere features):
//     -hb = [dx dy dz -o.d] . [cx cy cz 1],     m = [2ox 2oy 2oz 1] . [cx cy cz r^2-|c|^2]  - |o|^2,     W = hb^2 + m
// Two v_mfma_f32_16x16x4_f32 give 16 spheres x 16 rays; the VALU is left with fma + sub + alignbit per value
// (3 instead of 11).  This measures cycles per sphere per wave for both forms, alone and with a VALU filler that
// stands for the rest of the kernel (1 623 non-scan instructions per 496 spheres = 52 per 16 spheres).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int FILL>
__device__ __forceinline__ void filler(float (&f)[4], float x) {
#pragma unroll
    for (int k = 0; k < FILL; ++k) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(f[k & 3]) : "v"(x));
}

template <int FILL>
__global__ __launch_bounds__(256) void mfma_scan(unsigned *out, int iters, const float *__restrict__ feat, int ngroups) {
    const int lane = threadIdx.x & 63, R = lane >> 4, c = lane & 15;
    float T1[4], T2[4], Too[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { T1[t] = 0.1f * t + 1e-3f * lane; T2[t] = 0.2f * t - 1e-3f * lane; Too[t] = 0.01f * t + blockIdx.x * 1e-6f; }
    float f[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    unsigned mask = 0, acc = 0;
    const float *p1 = feat + R * 16 + c, *p2 = feat + (R == 3 ? 4 : R) * 16 + c;
    for (int it = 0; it < iters; ++it) {
        float a1 = p1[0], a2 = p2[0];
        for (int g = 0; g < ngroups; ++g) {
            const float n1 = p1[(g + 1) * 80], n2 = p2[(g + 1) * 80];      // next group (array is padded by one group)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 z = {0, 0, 0, 0};
                const f4 h = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, T1[t], z, 0, 0, 0);
                const f4 m = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, T2[t], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float W = __builtin_fmaf(h[r], h[r], m[r]) - Too[t];
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(W), 31);
                }
            }
            filler<FILL>(f, a1);
            if (g & 1) { acc ^= mask; mask = 0; }
            a1 = n1; a2 = n2;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ __float_as_uint(f[0] + f[1] + f[2] + f[3]);
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// 32 spheres x 32 rays per v_mfma_f32_32x32x16_f16 (K = 4 features x 4 split terms: every f32 feature as two f16 pieces)
template <int FILL>
__global__ __launch_bounds__(256) void mfma16_scan(unsigned *out, int iters, const float *__restrict__ feat, int nblocks) {
    const int lane = threadIdx.x & 63;
    h8 B1[2], B2[2];
    float Too[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { B1[h][e] = (_Float16)(0.1f * e + 1e-3f * lane + h); B2[h][e] = (_Float16)(0.2f * e - 1e-3f * lane - h); }
        Too[h] = 0.01f * h + blockIdx.x * 1e-6f;
    }
    float f[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    unsigned acc = 0;
    const h8 *pa = (const h8 *)feat + lane;
    for (int it = 0; it < iters; ++it) {
        h8 a1 = pa[0], a2 = pa[64];
        for (int g = 0; g < nblocks; ++g) {
            const h8 n1 = pa[(g + 1) * 128], n2 = pa[(g + 1) * 128 + 64];
            unsigned mask = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const f16v hh = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, B1[h], z, 0, 0, 0);
                const f16v m = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, B2[h], z, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float W = __builtin_fmaf(hh[r], hh[r], m[r]) - Too[h];
                    mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(W), 31);
                }
            }
            filler<2 * FILL>(f, Too[0]);
            acc ^= mask;
            a1 = n1; a2 = n2;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ __float_as_uint(f[0] + f[1] + f[2] + f[3]);
}

template <int FILL>
__global__ __launch_bounds__(256) void valu_scan(unsigned *out, int iters, const float *__restrict__ g4, int ngroups) {
    typedef const float __attribute__((address_space(4))) *cptr;
    cptr gs = (cptr)(uintptr_t)g4;
    float ox = threadIdx.x * 1e-3f, oy = 0.5f + blockIdx.x * 1e-6f, oz = 0.25f, dx = 0.6f, dy = 0.0f, dz = 0.8f;
    float f[4] = {0.1f, 0.2f, 0.3f, 0.4f};
    unsigned mask = 0, acc = 0;
    for (int it = 0; it < iters; ++it)
        for (int g = 0; g < ngroups; ++g) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = g * 16 + k;
                const float cx = gs[4 * i], cy = gs[4 * i + 1], cz = gs[4 * i + 2], r2 = gs[4 * i + 3];
                const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                const float hb = __builtin_fmaf(ocz, dz, __builtin_fmaf(ocy, dy, ocx * dx));
                const float nc = __builtin_fmaf(-ocz, ocz, __builtin_fmaf(-ocy, ocy, __builtin_fmaf(-ocx, ocx, r2)));
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(__builtin_fmaf(hb, hb, nc)), 31);
            }
            filler<FILL>(f, ox);
            if (g & 1) { acc ^= mask; mask = 0; }
        }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ __float_as_uint(f[0] + f[1] + f[2] + f[3]);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int ngroups = 31, iters = 64;
    static float h[64 * 8 * 20];
    for (int i = 0; i < 64 * 8 * 20; ++i) h[i] = 0.01f * (i % 97) - 0.3f;
    float *g; unsigned *d; (void)hipMalloc(&g, sizeof h); (void)hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {4, 5, 6, 7}) {
        const int blocks = p.multiProcessorCount * wps;
        for (int m = 0; m < 8; ++m) {
            auto launch = [&]() {
                switch (m) {
                case 0: hipLaunchKernelGGL(valu_scan<0>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 1: hipLaunchKernelGGL(valu_scan<52>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 2: hipLaunchKernelGGL(mfma_scan<0>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 3: hipLaunchKernelGGL(mfma_scan<52>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 4: hipLaunchKernelGGL(mfma_scan<80>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 5: hipLaunchKernelGGL(mfma_scan<120>, dim3(blocks), dim3(256), 0, 0, d, iters, g, ngroups); break;
                case 6: hipLaunchKernelGGL(mfma16_scan<0>, dim3(blocks), dim3(256), 0, 0, d, iters, g, (ngroups + 1) / 2); break;
                default: hipLaunchKernelGGL(mfma16_scan<52>, dim3(blocks), dim3(256), 0, 0, d, iters, g, (ngroups + 1) / 2); break;
                }
            };
            static const char *names[] = {"valu", "valu+52", "mfma", "mfma+52", "mfma+80", "mfma+120", "f16x2-32x32", "f16x2-32x32+52"};
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double sph = (double)blocks * 4 * iters * (m >= 6 ? (ngroups + 1) / 2 * 32 : ngroups * 16);     // wave-spheres
            printf("waves/SIMD %d %-9s %8.3f ms  %6.2f cycles per sphere per wave-slot @2.4GHz  (%.1f %% of FP32 peak at 17 flop)\n", wps, names[m], ms,
                   p.multiProcessorCount * 4 * 2.4e9 / (sph / (ms * 1e-3)), 100.0 * sph * 64 * 17 / (ms * 1e-3) / 157.3e12);
        }
    }
    return 0;
}
