// ubench_bank.hip -- do VGPR bank conflicts (register number mod 4) limit the VALU issue rate on gfx950?
// 8 independent v_fma_f32 chains; the three sources of every instruction are either in three different
// banks or all in the same bank.  Also VOP2 forms (v_fmac, v_sub with SGPR) as in the scan's inner loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256) void probe(float *out, int iters, float s0) {
    float r = 0;
    // v8..v15 accumulators; sources v16..v27
    asm volatile(
        "v_mov_b32 v8, %1\n v_mov_b32 v9, %1\n v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n"
        "v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0\n"
        "s_mov_b32 s40, %2\n"
        "1:\n"
        : "=v"(r) : "v"((float)threadIdx.x), "s"(iters)
        : "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","s40","scc","memory");
    if (MODE == 0) {   // dst = dst * vB + vC, three sources in three different banks (dst bank b, src b+1, b+2)
        asm volatile(REP8(
            "v_fma_f32 v8, v8, v17, v22\n v_fma_f32 v9, v9, v18, v23\n v_fma_f32 v10, v10, v19, v20\n v_fma_f32 v11, v11, v16, v21\n"
            "v_fma_f32 v12, v12, v17, v22\n v_fma_f32 v13, v13, v18, v23\n v_fma_f32 v14, v14, v19, v20\n v_fma_f32 v15, v15, v16, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 1) {   // all three sources in the SAME bank
        asm volatile(REP8(
            "v_fma_f32 v8, v8, v16, v20\n v_fma_f32 v9, v9, v17, v21\n v_fma_f32 v10, v10, v18, v22\n v_fma_f32 v11, v11, v19, v23\n"
            "v_fma_f32 v12, v12, v16, v24\n v_fma_f32 v13, v13, v17, v25\n v_fma_f32 v14, v14, v18, v26\n v_fma_f32 v15, v15, v19, v27\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 2) {   // two sources same bank, third different
        asm volatile(REP8(
            "v_fma_f32 v8, v8, v16, v21\n v_fma_f32 v9, v9, v17, v22\n v_fma_f32 v10, v10, v18, v23\n v_fma_f32 v11, v11, v19, v20\n"
            "v_fma_f32 v12, v12, v16, v21\n v_fma_f32 v13, v13, v17, v22\n v_fma_f32 v14, v14, v18, v23\n v_fma_f32 v15, v15, v19, v20\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 3) {   // VOP2 v_fmac (dst += a*b), sources in different banks from dst
        asm volatile(REP8(
            "v_fmac_f32 v8, v17, v22\n v_fmac_f32 v9, v18, v23\n v_fmac_f32 v10, v19, v20\n v_fmac_f32 v11, v16, v21\n"
            "v_fmac_f32 v12, v17, v22\n v_fmac_f32 v13, v18, v23\n v_fmac_f32 v14, v19, v20\n v_fmac_f32 v15, v16, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 4) {   // VOP2 v_fmac, all same bank
        asm volatile(REP8(
            "v_fmac_f32 v8, v16, v20\n v_fmac_f32 v9, v17, v21\n v_fmac_f32 v10, v18, v22\n v_fmac_f32 v11, v19, v23\n"
            "v_fmac_f32 v12, v16, v24\n v_fmac_f32 v13, v17, v25\n v_fmac_f32 v14, v18, v26\n v_fmac_f32 v15, v19, v27\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 5) {   // v_fma with the same register twice (the scan's fma(-oc, oc, acc)): v_fma dst, -vA, vA, dst
        asm volatile(REP8(
            "v_fma_f32 v8, -v17, v17, v8\n v_fma_f32 v9, -v18, v18, v9\n v_fma_f32 v10, -v19, v19, v10\n v_fma_f32 v11, -v16, v16, v11\n"
            "v_fma_f32 v12, -v17, v17, v12\n v_fma_f32 v13, -v18, v18, v13\n v_fma_f32 v14, -v19, v19, v14\n v_fma_f32 v15, -v16, v16, v15\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 6) {   // VOP2 v_sub with SGPR
        asm volatile(REP8(
            "v_subrev_f32 v8, s41, v8\n v_subrev_f32 v9, s41, v9\n v_subrev_f32 v10, s41, v10\n v_subrev_f32 v11, s41, v11\n"
            "v_subrev_f32 v12, s41, v12\n v_subrev_f32 v13, s41, v13\n v_subrev_f32 v14, s41, v14\n v_subrev_f32 v15, s41, v15\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15","s41");
    } else if (MODE == 7) {   // v_alignbit chain as in the scan (dst, dst, src, 31)
        asm volatile(REP8(
            "v_alignbit_b32 v8, v8, v17, 31\n v_alignbit_b32 v9, v9, v18, 31\n v_alignbit_b32 v10, v10, v19, 31\n v_alignbit_b32 v11, v11, v16, 31\n"
            "v_alignbit_b32 v12, v12, v17, 31\n v_alignbit_b32 v13, v13, v18, 31\n v_alignbit_b32 v14, v14, v19, 31\n v_alignbit_b32 v15, v15, v16, 31\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    }
    else if (MODE == 8) {
        asm volatile(REP8(
            "v_cvt_pkrtz_f16_f32 v8, v8, v17\n v_cvt_pkrtz_f16_f32 v9, v9, v18\n v_cvt_pkrtz_f16_f32 v10, v10, v19\n v_cvt_pkrtz_f16_f32 v11, v11, v16\n"
            "v_cvt_pkrtz_f16_f32 v12, v12, v17\n v_cvt_pkrtz_f16_f32 v13, v13, v18\n v_cvt_pkrtz_f16_f32 v14, v14, v19\n v_cvt_pkrtz_f16_f32 v15, v15, v16\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 9) {
        asm volatile(REP8(
            "v_perm_b32 v8, v8, v17, v22\n v_perm_b32 v9, v9, v18, v23\n v_perm_b32 v10, v10, v19, v20\n v_perm_b32 v11, v11, v16, v21\n"
            "v_perm_b32 v12, v12, v17, v22\n v_perm_b32 v13, v13, v18, v23\n v_perm_b32 v14, v14, v19, v20\n v_perm_b32 v15, v15, v16, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 10) {
        asm volatile(REP8(
            "v_and_or_b32 v8, v8, v17, v22\n v_and_or_b32 v9, v9, v18, v23\n v_and_or_b32 v10, v10, v19, v20\n v_and_or_b32 v11, v11, v16, v21\n"
            "v_and_or_b32 v12, v12, v17, v22\n v_and_or_b32 v13, v13, v18, v23\n v_and_or_b32 v14, v14, v19, v20\n v_and_or_b32 v15, v15, v16, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 11) {
        asm volatile(REP8(
            "v_lshl_or_b32 v8, v8, 1, v22\n v_lshl_or_b32 v9, v9, 1, v23\n v_lshl_or_b32 v10, v10, 1, v20\n v_lshl_or_b32 v11, v11, 1, v21\n"
            "v_lshl_or_b32 v12, v12, 1, v22\n v_lshl_or_b32 v13, v13, 1, v23\n v_lshl_or_b32 v14, v14, 1, v20\n v_lshl_or_b32 v15, v15, 1, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 12) {
        asm volatile(REP8(
            "v_max3_f32 v8, v8, v17, v22\n v_max3_f32 v9, v9, v18, v23\n v_max3_f32 v10, v10, v19, v20\n v_max3_f32 v11, v11, v16, v21\n"
            "v_max3_f32 v12, v12, v17, v22\n v_max3_f32 v13, v13, v18, v23\n v_max3_f32 v14, v14, v19, v20\n v_max3_f32 v15, v15, v16, v21\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 13) {   // v_cmp to VCC + v_addc (shift the compare result in as carry): 2 instructions per bit
        asm volatile(REP8(
            "v_cmp_gt_f32 vcc, 0, v17\n v_addc_co_u32 v8, vcc, v8, v8, vcc\n v_cmp_gt_f32 vcc, 0, v18\n v_addc_co_u32 v9, vcc, v9, v9, vcc\n"
            "v_cmp_gt_f32 vcc, 0, v19\n v_addc_co_u32 v10, vcc, v10, v10, vcc\n v_cmp_gt_f32 vcc, 0, v16\n v_addc_co_u32 v11, vcc, v11, v11, vcc\n")
            ::: "v8","v9","v10","v11","vcc");
    } else if (MODE == 14) {   // v_lshrrev_b32 (extract sign) alone
        asm volatile(REP8(
            "v_lshrrev_b32 v8, 31, v17\n v_lshrrev_b32 v9, 31, v18\n v_lshrrev_b32 v10, 31, v19\n v_lshrrev_b32 v11, 31, v16\n"
            "v_lshrrev_b32 v12, 31, v17\n v_lshrrev_b32 v13, 31, v18\n v_lshrrev_b32 v14, 31, v19\n v_lshrrev_b32 v15, 31, v16\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    } else if (MODE == 15) {   // v_mov_b32_sdwa: top byte of the source into byte 0 of dst, rest preserved
        asm volatile(REP8(
            "v_mov_b32_sdwa v8, v17 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v9, v18 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
            "v_mov_b32_sdwa v10, v19 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v11, v16 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
            "v_mov_b32_sdwa v12, v17 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v13, v18 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n"
            "v_mov_b32_sdwa v14, v19 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n v_mov_b32_sdwa v15, v16 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_3\n")
            ::: "v8","v9","v10","v11","v12","v13","v14","v15");
    }
    asm volatile(
        "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n"
        "v_add_f32 %0, v8, v9\n v_add_f32 %0, %0, v10\n v_add_f32 %0, %0, v11\n v_add_f32 %0, %0, v12\n v_add_f32 %0, %0, v13\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v15\n"
        : "=v"(r) :: "s40", "scc", "v8","v9","v10","v11","v12","v13","v14","v15");
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + s0;
}

int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    float *d; CHECK(hipMalloc(&d, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char *names[] = {"v_fma_f32 VOP3, 3 sources in 3 banks", "v_fma_f32 VOP3, 3 sources in 1 bank", "v_fma_f32 VOP3, 2 of 3 in one bank",
                           "v_fmac_f32 VOP2, different banks", "v_fmac_f32 VOP2, same bank", "v_fma_f32 dst, -a, a, dst", "v_subrev_f32 VOP2 with SGPR", "v_alignbit_b32 dst,dst,src,31",
                           "v_cvt_pkrtz_f16_f32", "v_perm_b32", "v_and_or_b32", "v_lshl_or_b32", "v_max3_f32", "v_cmp_gt_f32 + v_addc_co_u32 (per pair)", "v_lshrrev_b32", "v_mov_b32_sdwa byte insert"};
    for (int wps : {7}) {
        const int blocks = prop.multiProcessorCount * wps, iters = 2048;
        for (int m = 0; m < 16; ++m) {
            auto launch = [&]() {
                switch (m) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 4: hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 5: hipLaunchKernelGGL(probe<5>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 6: hipLaunchKernelGGL(probe<6>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 7: hipLaunchKernelGGL(probe<7>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 8: hipLaunchKernelGGL(probe<8>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 9: hipLaunchKernelGGL(probe<9>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 10: hipLaunchKernelGGL(probe<10>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 11: hipLaunchKernelGGL(probe<11>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 12: hipLaunchKernelGGL(probe<12>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 13: hipLaunchKernelGGL(probe<13>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    case 14: hipLaunchKernelGGL(probe<14>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                    default: hipLaunchKernelGGL(probe<15>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.f); break;
                }
            };
            launch(); (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double insts = (double)blocks * 4 * iters * 64.0;
            printf("wps %d  %-40s %8.3f ms  %5.2f cycles/inst/SIMD @2.4GHz\n", wps, names[m], ms, 1024.0 * 2.4e9 / (insts / (ms * 1e-3)));
        }
    }
    return 0;
}
