// Per-CU global store throughput by access shape (MI355X): one 512-thread workgroup per CU, every wave streams its own region.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/store_rate.hip -o tools/probes/store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
// SHAPE: bytes contiguous per row segment (64, 128, 256, 512, 1024); a wave-instruction covers 1024 / SEG rows of a matrix whose
// row pitch is PITCH bytes.  Each wave writes ITERS instructions walking down its own rows.
template <int SEG, bool NT>
__global__ __launch_bounds__(512) void store_kernel(char* base, long pitch, int iters, long wave_stride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = SEG / 16;      // lanes per row
    constexpr int ROWS = 64 / LPR;     // rows per instruction
    char* p = base + ((long)blockIdx.x * 8 + wave) * wave_stride + (long)(lane / LPR) * pitch + (lane % LPR) * 16;
    u32x4_t v = {(unsigned)lane, 1u, 2u, 3u};
    for (int i = 0; i < iters; ++i) {
        if (NT) __builtin_nontemporal_store(v, (u32x4_t*)p);
        else *(u32x4_t*)p = v;
        p += (long)ROWS * pitch;
    }
}
template <int SEG, bool NT>
float run(char* buf, long pitch, int iters, int grid) {
    constexpr int ROWS = 64 / (SEG / 16);
    const long wave_stride = (long)ROWS * pitch * iters;
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(grid), dim3(512), 0, 0, buf, pitch, iters, wave_stride);
    hipEventRecord(a);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((store_kernel<SEG, NT>), dim3(grid), dim3(512), 0, 0, buf, pitch, iters, wave_stride);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms / reps;
}
int main(int argc, char** argv) {
    const int grid = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;  // 64 KiB per wave, 512 KiB per CU
    const long bytes = (long)grid * 8 * iters * 1024;
    char* buf;
    const long cap = 8L << 30;
    hipMalloc(&buf, cap);
    printf("grid %d, %d stores of 1 KiB per wave, %.1f MB per launch\n", grid, iters, bytes / 1e6);
#define RUN(SEG, NT, PITCH)                                                                                      \
    {                                                                                                            \
        const long pitch = (PITCH);                                                                              \
        if ((long)grid * 8 * (64 / (SEG / 16)) * pitch * iters <= cap) {                                         \
            const float ms = run<SEG, NT>(buf, pitch, iters, grid);                                              \
            printf("seg %4d B pitch %5ld nt %d: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz  (%.0f ns per wave-instruction per CU)\n", SEG, pitch, NT, \
                   ms * 1e3, bytes / ms / 1e9, bytes / (ms * 1e-3) / grid / 2.4e9, ms * 1e6 / (8.0 * iters));  \
        }                                                                                                        \
    }
    RUN(1024, false, 1024)
    RUN(1024, true, 1024)
    RUN(512, false, 4608)
    RUN(256, false, 4608)
    RUN(128, false, 4608)
    RUN(128, true, 4608)
    RUN(64, false, 4608)
    RUN(128, false, 1536)
    RUN(256, false, 3072)
    return 0;
}
