// ubench_mfma_overlap.hip -- do f32-input MFMAs (v_mfma_f32_16x16x4_f32) overlap with f32 VALU work of OTHER waves on the
// same SIMD?  8 waves/SIMD; `nm` of them run an MFMA stream, the rest an independent v_fma_f32 stream; equal instruction
// counts chosen so that each stream alone takes about the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
template <int KIND> __device__ __forceinline__ f4 mm(float a, float b, f4 c) {
    if constexpr (KIND == 0) return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    else if constexpr (KIND == 1) { h4 x = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b}; return __builtin_amdgcn_mfma_f32_16x16x16f16(x, x, c, 0, 0, 0); }
    else { h8 x = {(_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b, (_Float16)a, (_Float16)b}; return __builtin_amdgcn_mfma_f32_16x16x32_f16(x, x, c, 0, 0, 0); }
}
template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int iters, int mode) {
    // mode 0: every wave MFMA; 1: every wave VALU; 2: even waves MFMA, odd waves VALU (both full length);
    const int wave = (threadIdx.x >> 6) + 4 * blockIdx.x;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 1) == 0);
    float a = threadIdx.x * 1e-3f, b = 0.5f;
    float r = 0;
    if (mode == 3) {            // every wave: 4 MFMAs and 48 independent fma per iteration, interleaved
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4_ = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
#define FMA12 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
                             "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4_), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a))
        for (int i = 0; i < iters; ++i) {
            c0 = mm<KIND>(a, b, c0); FMA12;
            c1 = mm<KIND>(a, b, c1); FMA12;
            c2 = mm<KIND>(a, b, c2); FMA12;
            c3 = mm<KIND>(a, b, c3); FMA12;
        }
        r = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3 + f4_ + f5 + f6 + f7;
    } else if (do_mfma) {
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
        for (int i = 0; i < iters; ++i) {
            c0 = mm<KIND>(a, b, c0); c1 = mm<KIND>(a, b, c1); c2 = mm<KIND>(a, b, c2); c3 = mm<KIND>(a, b, c3);
        }
        r = c0[0] + c1[1] + c2[2] + c3[3];
    } else {
        float f0 = a, f1 = a + 1, f2 = a + 2, f3 = a + 3, f4_ = a + 4, f5 = a + 5, f6 = a + 6, f7 = a + 7;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 6; ++u) {     // 48 fma per iteration ~ 4 MFMA x 32 cycles / 2.67
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4_), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(b), "v"(a));
            }
        }
        r = f0 + f1 + f2 + f3 + f4_ + f5 + f6 + f7;
    }
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float *d; (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 20000;
    for (int kind = 0; kind < 3; ++kind)
    for (int wps : {4, 8})
        for (int mode = 0; mode < 4; ++mode) {
            const int blocks = p.multiProcessorCount * wps;
            auto launch = [&]() {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
                else if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
                else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, d, iters, mode);
            };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("%s  ", kind == 0 ? "f32 16x16x4 " : kind == 1 ? "f16 16x16x16" : "f16 16x16x32");
            static const char *names[] = {"all waves MFMA (4 per iteration)", "all waves VALU (48 fma per iteration)", "half MFMA, half VALU", "every wave both, interleaved"};
            printf("waves/SIMD %d  %-38s %8.3f ms   %7.1f SIMD cycles per wave-iteration\n", wps, names[mode], ms, ms * 1e-3 * 2.4e9 / iters / wps);
        }
    return 0;
}
