// ubench_pk.hip -- the scan's discriminant chain for sphere PAIRS with packed FP32 (v_pk_add/mul/fma_f32), sphere
// data as SGPR pairs, against the scalar chain.  cycles per SPHERE per SIMD at nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
struct Sph { float cx, cy, cz, r2; };

__global__ __launch_bounds__(256) void scalar_chain(unsigned *out, int iters, const float *__restrict__ g, int n) {
    typedef const float __attribute__((address_space(4))) *cptr;
    cptr gs = (cptr)(uintptr_t)g;
    float ox = threadIdx.x * 1e-3f, oy = 0.5f + blockIdx.x * 1e-6f, oz = 0.25f, dx = 0.6f, dy = 0.0f, dz = 0.8f;
    unsigned mask = 0;
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float cx = gs[4 * (i + k)], cy = gs[4 * (i + k) + 1], cz = gs[4 * (i + k) + 2], r2 = gs[4 * (i + k) + 3];
                const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
                const float hb = __builtin_fmaf(ocz, dz, __builtin_fmaf(ocy, dy, ocx * dx));
                const float nc = __builtin_fmaf(-ocz, ocz, __builtin_fmaf(-ocy, ocy, __builtin_fmaf(-ocx, ocx, r2)));
                const float disc = __builtin_fmaf(hb, hb, nc);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(disc), 31);
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = mask;
}
// pair layout: [cx0 cx1 cy0 cy1 cz0 cz1 r20 r21] per 2 spheres
__global__ __launch_bounds__(256) void packed_chain(unsigned *out, int iters, const float *__restrict__ g, int n) {
    typedef const float __attribute__((address_space(4))) *cptr;
    cptr gs = (cptr)(uintptr_t)g;
    const float ox = threadIdx.x * 1e-3f, oy = 0.5f + blockIdx.x * 1e-6f, oz = 0.25f, dx = 0.6f, dy = 0.0f, dz = 0.8f;
    const f2 OX = {ox, ox}, OY = {oy, oy}, OZ = {oz, oz}, DX = {dx, dx}, DY = {dy, dy}, DZ = {dz, dz};
    unsigned mask = 0;
    for (int it = 0; it < iters; ++it)
        for (int i = 0; i < n; i += 8) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int b = 4 * (i + 2 * k);
                const f2 cx = {gs[b], gs[b + 1]}, cy = {gs[b + 2], gs[b + 3]}, cz = {gs[b + 4], gs[b + 5]}, r2 = {gs[b + 6], gs[b + 7]};
                const f2 ocx = OX - cx, ocy = OY - cy, ocz = OZ - cz;
                const f2 hb = __builtin_elementwise_fma(ocz, DZ, __builtin_elementwise_fma(ocy, DY, ocx * DX));
                const f2 nc = __builtin_elementwise_fma(-ocz, ocz, __builtin_elementwise_fma(-ocy, ocy, __builtin_elementwise_fma(-ocx, ocx, r2)));
                const f2 disc = __builtin_elementwise_fma(hb, hb, nc);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(disc.x), 31);
                mask = __builtin_amdgcn_alignbit(mask, __float_as_uint(disc.y), 31);
            }
        }
    out[blockIdx.x * 256 + threadIdx.x] = mask;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int n = 488, iters = 64;
    float h[4 * 496];
    for (int i = 0; i < 4 * 496; ++i) h[i] = 0.01f * (i % 97) - 0.3f;
    float *g; unsigned *d; (void)hipMalloc(&g, sizeof h); (void)hipMemcpy(g, h, sizeof h, hipMemcpyHostToDevice);
    (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int wps : {5, 6, 7}) {
        const int blocks = p.multiProcessorCount * wps;
        for (int m = 0; m < 2; ++m) {
            auto launch = [&]() { if (m == 0) hipLaunchKernelGGL(scalar_chain, dim3(blocks), dim3(256), 0, 0, d, iters, g, n); else hipLaunchKernelGGL(packed_chain, dim3(blocks), dim3(256), 0, 0, d, iters, g, n); };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double sph = (double)blocks * 4 * iters * n;     // wave-spheres
            printf("wps %d %-8s %8.3f ms  %6.2f cycles per sphere per SIMD @2.4GHz  (%.1f %% of FP32 peak at 17 flop)\n", wps, m ? "packed" : "scalar", ms,
                   1024.0 * 2.4e9 / (sph / (ms * 1e-3)), 100.0 * sph * 64 * 17 / (ms * 1e-3) / 157.3e12);
        }
    }
    return 0;
}
