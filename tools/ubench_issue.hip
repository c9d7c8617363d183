// ubench_issue.hip -- VALU issue rate on gfx950 as a function of (a) the fraction and placement of
// SGPR-sourced instructions in a wave's stream and (b) waves per SIMD.  Feeds DESIGN.md section 4.1.
// build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_issue tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// PAT: bit k set => instruction k of each group of 11 reads an SGPR operand
template <unsigned PAT>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float s0, float s1, float s2, float s3) {
    float a[11];
    for (int k = 0; k < 11; ++k) a[k] = threadIdx.x + k;
    float vb = s0 + 1e-9f * threadIdx.x, vc = s1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                if ((PAT >> k) & 1u) {
                    if (k % 4 == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(s0), "v"(vc));
                    else if (k % 4 == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(s1), "v"(vc));
                    else if (k % 4 == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(s2), "v"(vc));
                    else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "s"(s3), "v"(vc));
                } else {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(vb), "v"(vc));
                }
            }
        }
    }
    float s = 0; for (int k = 0; k < 11; ++k) s += a[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// dependent chain like the real discriminant: 3 sub(S) -> mul, fma, fma -> fma(-),fma(-),fma(S) -> fma -> alignbit
__global__ __launch_bounds__(256) void probe_disc(float *out, int iters, float cx, float cy, float cz, float r2) {
    float ox = threadIdx.x * 1e-3f, oy = 0.5f, oz = 0.25f, dx = 0.6f, dy = 0.0f, dz = 0.8f;
    unsigned mask = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float ocx, ocy, ocz, hb, nc, disc;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocx) : "v"(ox), "s"(cx));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocy) : "v"(oy), "s"(cy));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocz) : "v"(oz), "s"(cz));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hb) : "v"(ocx), "v"(dx));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hb) : "v"(ocy), "v"(dy));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hb) : "v"(ocz), "v"(dz));
            asm volatile("v_fma_f32 %0, -%1, %1, %2" : "=v"(nc) : "v"(ocx), "s"(r2));
            asm volatile("v_fma_f32 %0, -%1, %1, %0" : "+v"(nc) : "v"(ocy));
            asm volatile("v_fma_f32 %0, -%1, %1, %0" : "+v"(nc) : "v"(ocz));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(disc) : "v"(hb), "v"(nc));
            asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mask) : "v"(disc));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)mask;
}
// same work, sphere constants first moved to VGPRs once (upper bound of an all-VGPR stream)
__global__ __launch_bounds__(256) void probe_disc_v(float *out, int iters, float cx_, float cy_, float cz_, float r2_) {
    float ox = threadIdx.x * 1e-3f, oy = 0.5f, oz = 0.25f, dx = 0.6f, dy = 0.0f, dz = 0.8f;
    float cx = cx_ + 1e-9f * threadIdx.x, cy = cy_ + 1e-9f * threadIdx.x, cz = cz_ + 1e-9f * threadIdx.x, r2 = r2_ + 1e-9f * threadIdx.x;
    unsigned mask = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            float ocx, ocy, ocz, hb, nc, disc;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocx) : "v"(ox), "v"(cx));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocy) : "v"(oy), "v"(cy));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(ocz) : "v"(oz), "v"(cz));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(hb) : "v"(ocx), "v"(dx));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hb) : "v"(ocy), "v"(dy));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(hb) : "v"(ocz), "v"(dz));
            asm volatile("v_fma_f32 %0, -%1, %1, %2" : "=v"(nc) : "v"(ocx), "v"(r2));
            asm volatile("v_fma_f32 %0, -%1, %1, %0" : "+v"(nc) : "v"(ocy));
            asm volatile("v_fma_f32 %0, -%1, %1, %0" : "+v"(nc) : "v"(ocz));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(disc) : "v"(hb), "v"(nc));
            asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(mask) : "v"(disc));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)mask;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int threads = 256;
    float *d; CHECK(hipMalloc(&d, (size_t)prop.multiProcessorCount * 8 * threads * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int wps : {1, 2, 4, 7, 8}) {
        const int blocks = prop.multiProcessorCount * wps;
        const int iters = 4096 / wps;
        auto time = [&](auto launch, const char *name, double inst_per_iter) {
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)blocks * threads / 64 * iters * inst_per_iter;   // wave-instructions
            printf("wps %d  %-34s %8.3f ms  %5.2f cycles/inst/SIMD @2.4GHz\n", wps, name, ms, 1024.0 * 2.4e9 / (insts / (ms * 1e-3)));
        };
#define P(PAT, name) time([&] { hipLaunchKernelGGL(probe<PAT>, dim3(blocks), dim3(threads), 0, 0, d, iters, 1.0000001f, 1e-9f, 1.0000002f, 1.0000003f); }, name, 44.0)
        P(0x000u, "11 fma: VVVVVVVVVVV");
        P(0x00fu, "11 fma: SSSSVVVVVVV");
        P(0x249u, "11 fma: SVVSVVSVVSV");
        P(0x555u, "11 fma: SVSVSVSVSVS (6S)");
        P(0x7ffu, "11 fma: SSSSSSSSSSS");
        time([&] { hipLaunchKernelGGL(probe_disc, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.1f, 0.2f, 0.3f, 0.04f); }, "disc chain (4 SGPR ops of 11)", 88.0);
        time([&] { hipLaunchKernelGGL(probe_disc_v, dim3(blocks), dim3(threads), 0, 0, d, iters, 0.1f, 0.2f, 0.3f, 0.04f); }, "disc chain, all-VGPR operands", 88.0);
    }
    return 0;
}
