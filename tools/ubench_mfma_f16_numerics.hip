// ubench_mfma_f16_numerics.hip -- what v_mfma_f32_32x32x16_f16 does to its inputs and its sum:
//  (1) operand / result layout check, (2) are f16 subnormal inputs honoured, (3) accumulation error of the 16-term sum.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
// A[32][16], B[16][32] row-major f16 (as float on the host side), D[32][32]
__global__ void mm(const _Float16 *A, const _Float16 *B, const float *C, float *D) {
    const int l = threadIdx.x, i = l & 31, H = l >> 5;
    h8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[i * 16 + 8 * H + e]; b[e] = B[(8 * H + e) * 32 + i]; }
    f16v c;
    for (int r = 0; r < 16; ++r) c[r] = C[((r & 3) + 8 * (r >> 2) + 4 * H) * 32 + i];
    const f16v d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * H) * 32 + i] = d[r];
}
int main() {
    _Float16 *dA, *dB; float *dC, *dD;
    (void)hipMalloc(&dA, 512 * 2); (void)hipMalloc(&dB, 512 * 2); (void)hipMalloc(&dC, 1024 * 4); (void)hipMalloc(&dD, 1024 * 4);
    std::vector<_Float16> A(512), B(512); std::vector<float> C(1024), D(1024);
    std::mt19937_64 rng(1);
    auto run = [&]() {
        (void)hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        (void)hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mm, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
        (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    };
    // (1) layout: small integers, asymmetric
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (_Float16)((i * 3 + k) % 7 - 3);
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)((k * 5 + j * 2) % 9 - 4);
    for (int i = 0; i < 1024; ++i) C[i] = (float)(i % 11);
    run();
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = C[i * 32 + j];
        for (int k = 0; k < 16; ++k) s += (double)(float)A[i * 16 + k] * (double)(float)B[k * 32 + j];
        if (s != D[i * 32 + j]) ++bad;
    }
    printf("layout check: %d mismatches of 1024\n", bad);
    // (2) subnormal inputs: a = 2^-20 (f16 subnormal), b = 2^10 -> product 2^-10 if honoured, 0 if flushed
    for (auto &x : A) x = (_Float16)0; for (auto &x : B) x = (_Float16)0; for (auto &x : C) x = 0;
    A[0] = (_Float16)9.5367431640625e-07f; B[0] = (_Float16)1024.0f;           // row 0 x col 0, k = 0
    A[16 + 1] = (_Float16)0.5f; B[32 + 1] = (_Float16)5.9604644775390625e-08f; // row 1 x col 1, k = 1: 0.5 * 2^-24 = 2^-25
    run();
    printf("subnormal A input: D[0][0] = %g (expected %g)   subnormal B input: D[1][1] = %g (expected %g)\n", D[0], 9.5367431640625e-07 * 1024, D[33], 0.5 * 5.9604644775390625e-08);
    // (3) accumulation error: random magnitudes over a wide range with cancellation; exact sum in long double
    double worst_rel_sum = 0, worst_rel_max = 0; long n_inexact = 0, n_tot = 0;
    for (int rep = 0; rep < 400; ++rep) {
        std::uniform_real_distribution<double> U(-1, 1); std::uniform_int_distribution<int> E(-6, 8);
        for (auto &x : A) x = (_Float16)(float)std::ldexp(U(rng), E(rng));
        for (auto &x : B) x = (_Float16)(float)std::ldexp(U(rng), E(rng));
        for (auto &x : C) x = (rep & 1) ? (float)std::ldexp(U(rng), E(rng) + 6) : 0.0f;
        if (rep % 4 == 2)   // strong cancellation: second half of K = minus the first half, plus a small tail
            for (int i = 0; i < 32; ++i) for (int k = 8; k < 15; ++k) { A[i * 16 + k] = A[i * 16 + k - 8]; }
        if (rep % 4 == 2) for (int k = 8; k < 15; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (_Float16)(-(float)B[(k - 8) * 32 + j]);
        run();
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            long double s = C[i * 32 + j], sa = std::fabs(C[i * 32 + j]), mx = sa;
            for (int k = 0; k < 16; ++k) { long double p = (long double)(float)A[i * 16 + k] * (long double)(float)B[k * 32 + j]; s += p; sa += fabsl(p); if (fabsl(p) > mx) mx = fabsl(p); }
            const long double err = fabsl((long double)D[i * 32 + j] - s);
            ++n_tot; if (err != 0) ++n_inexact;
            // subtract the unavoidable final rounding (half an ulp of the result)
            const long double ulp_half = std::ldexp(1.0L, (s == 0 ? -150 : (int)std::floor(std::log2((double)fabsl(s)))) - 24);
            const long double extra = err > ulp_half ? err - ulp_half : 0;
            if (sa > 0 && (double)(extra / sa) > worst_rel_sum) worst_rel_sum = (double)(extra / sa);
            if (mx > 0 && (double)(extra / mx) > worst_rel_max) worst_rel_max = (double)(extra / mx);
        }
    }
    printf("accumulation: %ld of %ld results inexact; worst error beyond the final half-ulp: %.3g x sum|terms| (2^%.1f), %.3g x max|term| (2^%.1f)\n",
           n_inexact, n_tot, worst_rel_sum, std::log2(worst_rel_sum), worst_rel_max, std::log2(worst_rel_max));
    return 0;
}
