// ubench_pipes.hip -- do different instruction classes issue concurrently from different waves of one SIMD?
// Half of the waves run stream A, the other half stream B (wave-uniform branch); compare with all-A and all-B.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
#define FMA "v_fma_f32 v8, v8, v17, v22\n v_fma_f32 v9, v9, v18, v23\n v_fma_f32 v10, v10, v19, v20\n v_fma_f32 v11, v11, v16, v21\n v_fma_f32 v12, v12, v17, v22\n v_fma_f32 v13, v13, v18, v23\n v_fma_f32 v14, v14, v19, v20\n v_fma_f32 v15, v15, v16, v21\n"
#define ALN "v_alignbit_b32 v8, v8, v17, 31\n v_alignbit_b32 v9, v9, v18, 31\n v_alignbit_b32 v10, v10, v19, 31\n v_alignbit_b32 v11, v11, v16, 31\n v_alignbit_b32 v12, v12, v17, 31\n v_alignbit_b32 v13, v13, v18, 31\n v_alignbit_b32 v14, v14, v19, 31\n v_alignbit_b32 v15, v15, v16, 31\n"
#define PKF "v_pk_fma_f32 v[32:33], v[32:33], v[48:49], v[50:51]\n v_pk_fma_f32 v[34:35], v[34:35], v[48:49], v[50:51]\n v_pk_fma_f32 v[36:37], v[36:37], v[48:49], v[50:51]\n v_pk_fma_f32 v[38:39], v[38:39], v[48:49], v[50:51]\n v_pk_fma_f32 v[40:41], v[40:41], v[48:49], v[50:51]\n v_pk_fma_f32 v[42:43], v[42:43], v[48:49], v[50:51]\n v_pk_fma_f32 v[44:45], v[44:45], v[48:49], v[50:51]\n v_pk_fma_f32 v[46:47], v[46:47], v[48:49], v[50:51]\n"
#define F64 "v_fma_f64 v[32:33], v[32:33], v[48:49], v[50:51]\n v_fma_f64 v[34:35], v[34:35], v[48:49], v[50:51]\n v_fma_f64 v[36:37], v[36:37], v[48:49], v[50:51]\n v_fma_f64 v[38:39], v[38:39], v[48:49], v[50:51]\n v_fma_f64 v[40:41], v[40:41], v[48:49], v[50:51]\n v_fma_f64 v[42:43], v[42:43], v[48:49], v[50:51]\n v_fma_f64 v[44:45], v[44:45], v[48:49], v[50:51]\n v_fma_f64 v[46:47], v[46:47], v[48:49], v[50:51]\n"
#define XOR "v_xor_b32 v8, v8, v17\n v_xor_b32 v9, v9, v18\n v_xor_b32 v10, v10, v19\n v_xor_b32 v11, v11, v16\n v_xor_b32 v12, v12, v17\n v_xor_b32 v13, v13, v18\n v_xor_b32 v14, v14, v19\n v_xor_b32 v15, v15, v16\n"
#define CLOB "v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","s40","scc","memory"
#define INIT "s_mov_b32 s40, %1\n v_mov_b32 v8, 1.0\n v_mov_b32 v9, 1.0\n v_mov_b32 v10, 1.0\n v_mov_b32 v11, 1.0\n v_mov_b32 v12, 1.0\n v_mov_b32 v13, 1.0\n v_mov_b32 v14, 1.0\n v_mov_b32 v15, 1.0\n v_mov_b32 v16, 1.0\n v_mov_b32 v17, 1.0\n v_mov_b32 v18, 1.0\n v_mov_b32 v19, 1.0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n" \
             "v_mov_b32 v32, 0\n v_mov_b32 v33, 0\n v_mov_b32 v34, 0\n v_mov_b32 v35, 0\n v_mov_b32 v36, 0\n v_mov_b32 v37, 0\n v_mov_b32 v38, 0\n v_mov_b32 v39, 0\n v_mov_b32 v40, 0\n v_mov_b32 v41, 0\n v_mov_b32 v42, 0\n v_mov_b32 v43, 0\n v_mov_b32 v44, 0\n v_mov_b32 v45, 0\n v_mov_b32 v46, 0\n v_mov_b32 v47, 0\n v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"
#define LOOP(BODY) asm volatile(INIT "1:\n" REP8(BODY) "s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n v_mov_b32 %0, v8\n" : "=v"(r) : "s"(iters) : CLOB)
// KIND: which stream each wave runs.  split: waves with (wave index & 1) run B, the others A.
template <int A, int B>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float r = 0;
    const int wave = (blockIdx.x * 4 + (threadIdx.x >> 6));
    const int which = (wave & 1) ? B : A;
    if (__builtin_amdgcn_readfirstlane(which) == 0) LOOP(FMA);
    else if (__builtin_amdgcn_readfirstlane(which) == 1) LOOP(ALN);
    else if (__builtin_amdgcn_readfirstlane(which) == 2) LOOP(PKF);
    else if (__builtin_amdgcn_readfirstlane(which) == 3) LOOP(F64);
    else LOOP(XOR);
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int A, int B> float run(float *d, int blocks, int iters) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL((k<A, B>), dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    float *d; (void)hipMalloc(&d, (size_t)p.multiProcessorCount * 8 * 256 * 4);
    const int blocks = p.multiProcessorCount * 8, iters = 2048;   // 8 waves per SIMD: 4 run A, 4 run B
    const char *nm[] = {"fma_f32", "alignbit", "pk_fma_f32", "fma_f64", "xor"};
    printf("8 waves/SIMD, 64 instructions per loop iteration per wave; time for all waves to finish\n");
#define R(A, B) printf("  %-10s | %-10s : %7.3f ms\n", nm[A], nm[B], run<A, B>(d, blocks, iters))
    R(0, 0); R(1, 1); R(2, 2); R(3, 3); R(4, 4);
    R(0, 1); R(0, 2); R(0, 3); R(0, 4); R(1, 2); R(1, 3); R(1, 4);
    return 0;
}
