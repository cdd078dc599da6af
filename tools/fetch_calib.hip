// FETCH_SIZE / WRITE_SIZE calibration on a KNOWN byte count in the access patterns the GEMM engines use (MI355X_MICROARCH.md:
// "Other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel reads (or
// writes) exactly BYTES bytes of a buffer much larger than L2 + Infinity Cache, once:
//   k_ld16     global_load_dwordx4, 16 B per lane, fully coalesced                      (the documented x2 case)
//   k_dma128   global_load_lds_dwordx4, one instruction = 8 rows x 128 B  (gemm8: BK = 64 K-contiguous pieces, full lines)
//   k_dma64    global_load_lds_dwordx4, one instruction = 16 rows x 64 B  (gemm2 / gemm5 at BK = 32: half lines)
//   k_dma256   global_load_lds_dwordx4, one instruction = 4 rows x 256 B  (TR operands: dgrad / wgrad pieces)
//   k_st16     global_store_dwordx4, 16 B per lane
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib ; rocprofv3 --pmc FETCH_SIZE ... -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) char* lds_ptr;
__device__ __forceinline__ void glds16(const void* g, unsigned lds) {
    unsigned keep;
    lds = __builtin_amdgcn_readfirstlane(lds);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
constexpr long ROW = 6144;  // bytes per matrix row (K = 3072 bf16): rows of one piece are ROW apart, like a GEMM operand
__global__ void k_ld16(const uint4* __restrict__ p, uint4* out, long n16) {
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc.x ^= v.x, acc.y ^= v.y, acc.z ^= v.z, acc.w ^= v.w;
    }
    if (acc.x == 0x12345) out[0] = acc;
}
// RB = bytes per row piece (64 / 128 / 256): one wave-instruction covers 1024 / RB rows; the wave walks the row's K extent
template <int RB>
__global__ void k_dma(const char* __restrict__ p, float* out, long nrows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int RPI = 1024 / RB, LPR = RB / 16;  // rows per instruction, lanes per row
    const unsigned lds = (unsigned)(uintptr_t)(lds_ptr)smem + wave * 1024;
    for (long r0 = ((long)blockIdx.x * nw + wave) * RPI; r0 < nrows; r0 += (long)gridDim.x * nw * RPI) {
        const char* src = p + (r0 + lane / LPR) * ROW + (lane % LPR) * 16;
        for (int k = 0; k < ROW; k += RB) glds16(src + k, lds);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (smem[threadIdx.x] == 0x7f && out) out[0] = 1.f;
}
__global__ void k_st16(uint4* __restrict__ p, long n16) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, (unsigned)i);
}
int main() {
    const long nrows = 160 * 1024, bytes = nrows * ROW;  // 1.0 GB >> 32 MiB L2 + 256 MiB Infinity Cache
    char* buf;
    float* out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_ld16, dim3(2048), dim3(256), 0, 0, (const uint4*)buf, (uint4*)out, bytes / 16);
        hipLaunchKernelGGL(k_dma<128>, dim3(1024), dim3(256), 4096, 0, buf, out, nrows);
        hipLaunchKernelGGL(k_dma<64>, dim3(1024), dim3(256), 4096, 0, buf, out, nrows);
        hipLaunchKernelGGL(k_dma<256>, dim3(1024), dim3(256), 4096, 0, buf, out, nrows);
        hipLaunchKernelGGL(k_st16, dim3(2048), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("bytes per kernel: %ld\n", bytes);
    return 0;
}
