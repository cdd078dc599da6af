#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_char_ptr;
__device__ __forceinline__ void blds16(unsigned voff, i32x4 rsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}
__global__ void k(const uint32_t* src, uint32_t nbytes, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char smem[2048];
    uint32_t* s32 = (uint32_t*)smem;
    for (int i = threadIdx.x; i < 512; i += 64) s32[i] = 0xABABABABu;
    __syncthreads();
    const uint64_t a = (uint64_t)src;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)nbytes);
    r.w = 0x00020000;
    const int lane = threadIdx.x;
    unsigned voff = (lane & 1) ? 0x80000000u : lane * 16;
    if (lane == 62) voff = nbytes - 16;  // last valid
    if (lane == 60) voff = nbytes - 8;   // straddles the end
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    blds16(voff, r, lds_base);
    blds16(voff, r, lds_base + 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = s32[i];
}
int main() {
    const int n = 4096;
    uint32_t *d, *o, h[n], ho[512];
    for (int i = 0; i < n; ++i) h[i] = 0x1000000u + i;
    hipMalloc(&d, n * 4); hipMalloc(&o, 2048);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, (uint32_t)(2048), o);
    hipMemcpy(ho, o, 2048, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int half = 0; half < 2; ++half)
    for (int l = 0; l < 64; ++l) {
        uint32_t* p = ho + half * 256 + l * 4;
        if (l < 4 || l >= 58) printf("half %d lane %d: %08x %08x %08x %08x\n", half, l, p[0], p[1], p[2], p[3]);
        uint32_t exp0 = (l & 1) ? 0 : 0x1000000u + l * 4;
        if (l == 62) exp0 = 0x1000000u + (2048 - 16) / 4;
        if (l == 60) continue;
        if (p[0] != exp0) ++bad;
    }
    printf("%s: %d mismatches\n", bad ? "FAIL" : "PASS", bad);
    return bad != 0;
}
