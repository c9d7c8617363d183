// ubench_write_size.hip -- calibrate rocprofv3 WRITE_SIZE on gfx950 and check HW_REG_XCC_ID.
//   k_full   : every lane stores 4 consecutive bytes: 24.9 MB in whole 128-byte lines                      (expect 1.0 x)
//   k_pieces : the same bytes, but each 128-byte line is written in 24-byte pieces by DIFFERENT workgroups at different times
//              (piece p of every line by pass p of the kernel's grid-stride loop): what the trace kernel's 2x2-pixel jobs do
//   k_pieces_xcd : like k_pieces, but the pieces of a line are written by workgroups of ONE XCD (line -> xcd = line % 8)
//   k_xcc    : prints (blockIdx.x, HW_REG_XCC_ID) for the first blocks
// run under: rocprofv3 --pmc WRITE_SIZE --kernel-trace ...
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned xcc() { return (unsigned)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u; }
__global__ void k_full(float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)i;
}
// n_lines lines of 32 floats; piece p = floats [6p, 6p + 6) (p = 0..4; piece 5 = floats 30, 31).  `gap` = s_sleep(127) repetitions
// between the passes (127 x 64 cycles each): 1 -> microseconds, 400 -> ~1.4 ms per gap like the jobs of the trace kernel.
__global__ void k_pieces(float *out, size_t n_lines, int by_xcd, int gap) {
    const unsigned me = xcc() & 7u;
    const size_t per_block = blockDim.x / 8;
    for (int p = 0; p < 6; ++p) {
        if (!by_xcd) {
            for (size_t line = (size_t)blockIdx.x * per_block + threadIdx.x / 8; line < n_lines; line += (size_t)gridDim.x * per_block) {
                const int f = 6 * p + (threadIdx.x & 7);
                if ((threadIdx.x & 7) < 6 && f < 32) out[line * 32 + f] = (float)(line + p);
            }
        } else {
            // line 8 m + x is written only by blocks running on XCD x (blocks are dealt round robin: gridDim.x / 8 blocks per XCD)
            for (size_t m = (size_t)(blockIdx.x / 8) * per_block + threadIdx.x / 8; 8 * m + me < n_lines; m += (size_t)(gridDim.x / 8) * per_block) {
                const size_t line = 8 * m + me;
                const int f = 6 * p + (threadIdx.x & 7);
                if ((threadIdx.x & 7) < 6 && f < 32) out[line * 32 + f] = (float)(line + p);
            }
        }
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);
    }
}
// Strips: the buffer is cut into runs of `len` floats; each run is written by ONE store instruction of a few lanes, runs in a
// scattered order (run r by (block, lane group) hash) so that neighbouring runs are written at unrelated times by unrelated
// workgroups.  `shift` floats of misalignment (8 floats = 32 B).
__global__ void k_strips(float *out, size_t n, int len, int shift) {
    const size_t n_runs = (n - shift) / len;
    const int per_wave = 64 / len > 0 ? 64 / len : 1;          // runs per store instruction
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 64, n_waves = (size_t)gridDim.x * blockDim.x / 64;
    const int lane = threadIdx.x & 63;
    for (size_t it = 0; ; ++it) {
        // a pseudo-random permutation of the runs: r = (a * idx) mod n_runs with a odd multiplier coprime to n_runs (n_runs prime-ish not needed: collisions only re-write)
        const size_t idx = (it * n_waves + wave) * per_wave + lane / len;
        if ((it * n_waves) * per_wave >= n_runs) break;
        if (len <= 64) {
            if (idx < n_runs && lane < per_wave * len) { const size_t r = (idx * 2654435761ull) % n_runs; out[shift + r * len + lane % len] = (float)r; }
        } else {
            if (idx < n_runs) { const size_t r = (idx * 2654435761ull) % n_runs; for (int q = lane; q < len; q += 64) out[shift + r * len + q] = (float)r; }
        }
        __builtin_amdgcn_s_sleep(20);
    }
}
__global__ void k_xcc(unsigned *out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc(); }
int main() {
    const size_t n = (size_t)1920 * 1080 * 3, n_lines = n / 32;
    float *d; (void)hipMalloc(&d, n * 4);
    unsigned *x; (void)hipMalloc(&x, 64 * 4);
    hipLaunchKernelGGL(k_full, dim3(2048), dim3(256), 0, 0, d, n);
    (void)hipDeviceSynchronize();
    for (int gap : {1, 400, 4000})
        for (int by_xcd : {0, 1}) {
            hipLaunchKernelGGL(k_pieces, dim3(1280), dim3(256), 0, 0, d, n_lines, by_xcd, gap);
            (void)hipDeviceSynchronize();
        }
    printf("k_pieces launches in order: (gap 1, any XCD) (gap 1, one XCD per line) (gap 400, any) (gap 400, one) (gap 4000, any) (gap 4000, one)\n");
    for (int len : {6, 12, 24, 48, 96})
        for (int shift : {0, 8}) {
            hipLaunchKernelGGL(k_strips, dim3(1280), dim3(256), 0, 0, d, n, len, shift);
            (void)hipDeviceSynchronize();
        }
    printf("k_strips launches in order: len (floats) 6, 12, 24, 48, 96 x shift (floats) 0, 8\n");
    hipLaunchKernelGGL(k_xcc, dim3(64), dim3(64), 0, 0, x);
    unsigned h[64]; (void)hipMemcpy(h, x, sizeof h, hipMemcpyDeviceToHost);
    printf("XCC_ID of blocks 0..31:"); for (int i = 0; i < 32; ++i) printf(" %u", h[i]); printf("\n");
    printf("bytes written by each kernel: %zu\n", n * 4);
    return 0;
}
