// ubench_sgpr.hip -- does an SGPR source operand limit the VALU issue rate on gfx950?
// 8 independent accumulators per lane, v_fma_f32 with (VGPR,VGPR) or (SGPR,VGPR) sources, and a
// mix with fraction k/8 of SGPR-sourced instructions.  hipcc --offload-arch=gfx950 -O3 ...
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NS>   // NS of every 8 FMAs read an SGPR
__global__ __launch_bounds__(256) void probe(float *out, int iters, float sb, float sc) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
    float vb = sb + 1e-9f * threadIdx.x, vc = sc;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k < NS) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(sb), "v"(vc));
                else        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(vb), "v"(vc));
            }
        }
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// VOP2 forms: v_fmac_f32 (dst += s*v) and v_subrev with SGPR
template <int MODE>
__global__ __launch_bounds__(256) void probe2(float *out, int iters, float sb, float sc) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = threadIdx.x + k;
    float vb = sb + 1e-9f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (MODE == 0) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "s"(sb), "v"(vb));
                if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(vb), "v"(vb));
                if (MODE == 2) asm volatile("v_subrev_f32 %0, %1, %0" : "+v"(a[k]) : "s"(sc));
                if (MODE == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(vb));
                if (MODE == 4) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a[k]) : "v"(vb));
            }
        }
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, threads = 256, iters = 2048;
    float *d; CHECK(hipMalloc(&d, blocks * threads * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char *name) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double insts = (double)blocks * threads / 64 * iters * 64;   // wave-instructions
        printf("%-44s %7.3f ms  %6.2f G wave-inst/s  = %5.2f cycles/inst/SIMD @2.4GHz\n", name, ms, insts / ms / 1e6,
               1024.0 * 2.4e9 / (insts / (ms * 1e-3)));
    };
#define P(NS) time([&] { hipLaunchKernelGGL(probe<NS>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001f, 1e-9f); }, "v_fma_f32 VOP3, " #NS "/8 with SGPR src")
    P(0); P(2); P(4); P(6); P(8);
#define Q(M, name) time([&] { hipLaunchKernelGGL(probe2<M>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001f, 1e-9f); }, name)
    Q(0, "v_fmac_f32 VOP2 (SGPR, VGPR)"); Q(1, "v_fmac_f32 VOP2 (VGPR, VGPR)"); Q(2, "v_subrev_f32 VOP2 (SGPR)"); Q(3, "v_sub_f32 VOP2 (VGPR)"); Q(4, "v_alignbit_b32 VOP3");
    return 0;
}
