// Does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?  (gfx950)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_offset.hip -o tools/probes/lds_dma_offset
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) char* lds_ptr;
__global__ void k(const unsigned* g, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* s = (unsigned*)smem;
    for (int i = threadIdx.x; i < 2048; i += 64) s[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds = (unsigned)(uintptr_t)(lds_ptr)smem;
    const unsigned voff = threadIdx.x * 16;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(voff), "s"(g), "s"(lds) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 2048; i += 64) out[i] = s[i];
}
int main() {
    std::vector<unsigned> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned *g, *o;
    hipMalloc(&g, 4096 * 4);
    hipMalloc(&o, 2048 * 4);
    hipMemcpy(g, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, g, o);
    std::vector<unsigned> r(2048);
    hipMemcpy(r.data(), o, 2048 * 4, hipMemcpyDeviceToHost);
    int first = -1;
    for (int i = 0; i < 2048; ++i)
        if (r[i] != 0xdeadbeefu) { first = i; break; }
    printf("first written LDS dword: %d (byte %d), value there: %u (= global dword index; 256 means the source moved by 1024 bytes too)\n", first, first * 4, first >= 0 ? r[first] : 0);
    printf("LDS dword 0: %08x, dword 256: %08x, dword 511: %08x\n", r[0], r[256], r[511]);
    return 0;
}
