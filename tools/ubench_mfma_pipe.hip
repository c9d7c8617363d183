// ubench_mfma_pipe.hip -- how should the block loop of hit_world_mfma (rtw_device.hpp) be scheduled?
// One block = 32 spheres x 64 rays = 4 x v_mfma_f32_32x32x16_f16 (two halves x P1, P2) + 64 VALU (one fma + one alignbit per
// result register).  Variants (ORDER):
//   0  the round-2 loop: M M | eval(32 VALU) | M M | eval(32)           one result set, every eval waits for its MFMAs
//   1  two result sets, coarse: [Mb Mb] evalA [Ma' Ma'] evalB           the MFMAs of the next half are issued before the eval
//   2  two result sets, fine:   Mb 16V Mb 16V  Ma' 16V Ma' 16V          one MFMA per 16 VALU
//   3  two result sets, 8-wise: M 8V ... (the other two quarters of each eval after its second MFMA)
// FILL = independent v_fma_f32 per block standing for the rest of the kernel (round 2: ~1 300 per 16 blocks = 81 per block).
// Output: SIMD cycles per block per wave at 2.4 GHz for W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

template <int FILL>
__device__ __forceinline__ void filler(float (&f)[8], float x) {
#pragma unroll
    for (int k = 0; k < FILL; ++k) asm volatile("v_fma_f32 %0, %0, %1, 0.5" : "+v"(f[k & 7]) : "v"(x));
}

__device__ __forceinline__ void eval16(const f16v &P1, const f16v &P2, unsigned &mask) {
#pragma unroll
    for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(__builtin_fmaf(P1[r], P1[r], P2[r])), 31);
}
// two independent alignbit chains (even / odd result registers), merged by the caller
__device__ __forceinline__ void eval16x2(const f16v &P1, const f16v &P2, unsigned &m0, unsigned &m1) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        m0 = __builtin_amdgcn_alignbit(m0, __float_as_uint(__builtin_fmaf(P1[r], P1[r], P2[r])), 31);
        m1 = __builtin_amdgcn_alignbit(m1, __float_as_uint(__builtin_fmaf(P1[r + 1], P1[r + 1], P2[r + 1])), 31);
    }
}

template <int ORDER, int FILL, int W>
__global__ __launch_bounds__(256, W) void scan(unsigned *out, int iters, const uint4 *__restrict__ feat, int nblocks, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63;
    h8 B1[2], B2[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) { B1[h][e] = (_Float16)(0.1f * e + 1e-3f * lane + h); B2[h][e] = (_Float16)(0.2f * e - 1e-3f * lane - h); }
    float f[8] = {0.1f, 0.2f, 0.3f, 0.4f, 0.5f, 0.6f, 0.7f, 0.8f};
    unsigned acc = 0;
    const unsigned long long t_start = __builtin_readcyclecounter();
    const uint4 *pa = feat + lane;
    const f16v z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4 A1 = pa[0], A2 = pa[64];
        if constexpr (ORDER >= 4) {
            // components: 4 = the four MFMAs only; 5 = the 64 eval VALU only (static inputs); 6 = 32 fma only; 7 = 32 alignbit only
            f16v P1 = z, P2 = z;
#pragma unroll
            for (int r = 0; r < 16; ++r) { P1[r] = 0.25f * r + lane; P2[r] = -1.5f * r; }
            for (int g = 0; g < nblocks; ++g) {
                unsigned mask = 0;
                if constexpr (ORDER == 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const f16v Q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[h], z, 0, 0, 0);
                        const f16v Q2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[h], z, 0, 0, 0);
                        asm volatile("" :: "v"(Q1), "v"(Q2));
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        asm volatile("" : "+v"(P1), "+v"(P2));
                        if constexpr (ORDER == 5) eval16(P1, P2, mask);
                        if constexpr (ORDER == 6) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) { float w = __builtin_fmaf(P1[r], P1[r], P2[r]); asm volatile("" :: "v"(w)); }
                        }
                        if constexpr (ORDER == 7) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(P1[r]), 31);
                        }
                    }
                }
                filler<FILL>(f, B1[0][0]);
                acc ^= mask;
            }
        } else if constexpr (ORDER == 0) {
            for (int g = 0; g < nblocks; ++g) {
                unsigned mask = 0;
                {
                    const f16v P1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[0], z, 0, 0, 0);
                    const f16v P2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[0], z, 0, 0, 0);
                    eval16(P1, P2, mask);
                }
                {
                    const f16v P1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[1], z, 0, 0, 0);
                    const f16v P2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[1], z, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    A1 = pa[(g + 1) * 128]; A2 = pa[(g + 1) * 128 + 64];
                    __builtin_amdgcn_sched_barrier(0);
                    eval16(P1, P2, mask);
                }
                filler<FILL>(f, B1[0][0]);
                acc ^= mask;
            }
        } else {
            // software pipeline over half blocks: set A = half 0, set B = half 1
            f16v PA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[0], z, 0, 0, 0);
            f16v PA2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[0], z, 0, 0, 0);
            for (int g = 0; g < nblocks; ++g) {
                unsigned m0 = 0, m1 = 0;
                const uint4 N1 = pa[(g + 1) * 128], N2 = pa[(g + 1) * 128 + 64];
                const f16v PB1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[1], z, 0, 0, 0);
                const f16v PB2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[1], z, 0, 0, 0);
                eval16x2(PA1, PA2, m0, m1);
                A1 = N1; A2 = N2;
                PA1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A1), B1[0], z, 0, 0, 0);
                PA2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, A2), B2[0], z, 0, 0, 0);
                eval16x2(PB1, PB2, m0, m1);
                if constexpr (ORDER == 1) {
                    SGB(0x008, 2); SGB(0x002, 32); SGB(0x008, 2); SGB(0x002, 32);
                } else if constexpr (ORDER == 2) {
                    SGB(0x008, 1); SGB(0x002, 16); SGB(0x008, 1); SGB(0x002, 16); SGB(0x008, 1); SGB(0x002, 16); SGB(0x008, 1); SGB(0x002, 16);
                } else {
                    SGB(0x008, 1); SGB(0x002, 8); SGB(0x008, 1); SGB(0x002, 24); SGB(0x008, 1); SGB(0x002, 8); SGB(0x008, 1); SGB(0x002, 24);
                }
                __builtin_amdgcn_sched_barrier(0);
                filler<FILL>(f, B1[0][0]);
                acc ^= (m0 << 16) ^ m1;
            }
            acc ^= __float_as_uint(PA1[0] + PA2[3]);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc ^ __float_as_uint(f[0] + f[1] + f[2] + f[3] + f[4] + f[5] + f[6] + f[7]);
    if (lane == 0) atomicAdd(cyc, __builtin_readcyclecounter() - t_start);
}

template <int ORDER, int FILL, int W>
float run(unsigned *d, const uint4 *g, int cus, int iters, int nblocks) {
    static unsigned long long *cyc = nullptr;
    if (!cyc) (void)hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = cus * W;
    hipLaunchKernelGGL((scan<ORDER, FILL, W>), dim3(blocks), dim3(256), 0, 0, d, iters, g, nblocks, cyc);
    (void)hipDeviceSynchronize();
    (void)hipMemset(cyc, 0, 8);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((scan<ORDER, FILL, W>), dim3(blocks), dim3(256), 0, 0, d, iters, g, nblocks, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc = 0; (void)hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    const double c24 = ms * 1e-3 * 2.4e9 / ((double)iters * nblocks * W);     // SIMD cycles per block per wave if the clock were 2.4 GHz
    const double ctrue = (double)hc / ((double)blocks * 4) / ((double)iters * nblocks * W);   // from s_memtime: wave residency / waves per SIMD
    printf("order %d fill %3d waves/SIMD %d : %8.3f ms  %7.1f cycles per (block, wave) @2.4GHz   %7.1f by s_memtime  (clock %.2f GHz)\n", ORDER, FILL, W, ms, c24, ctrue, 2.4 * ctrue / c24);
    return ms;
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int nblocks = 16, iters = 2000;
    static uint4 h[(16 + 2) * 128];
    for (int i = 0; i < (16 + 2) * 128; ++i) { h[i].x = 0x2e662e66u + i; h[i].y = 0x30003000u + 3 * i; h[i].z = 0x2c002c00u + 7 * i; h[i].w = 0x34003400u; }
    uint4 *g; unsigned *d; (void)hipMalloc(&g, sizeof h); (void)hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    const int cus = p.multiProcessorCount;
#define ROW(FILL, W) run<0, FILL, W>(d, g, cus, iters, nblocks); run<1, FILL, W>(d, g, cus, iters, nblocks); run<2, FILL, W>(d, g, cus, iters, nblocks); run<3, FILL, W>(d, g, cus, iters, nblocks);
#define COMP(FILL, W) run<4, FILL, W>(d, g, cus, iters, nblocks); run<5, FILL, W>(d, g, cus, iters, nblocks); run<6, FILL, W>(d, g, cus, iters, nblocks); run<7, FILL, W>(d, g, cus, iters, nblocks);
    ROW(0, 4) COMP(0, 4) COMP(0, 1) COMP(0, 2) ROW(0, 1) ROW(0, 2) ROW(80, 4) COMP(80, 4)
    return 0;
}
