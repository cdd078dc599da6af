// Diagnostic: which XCD (HW_REG_XCC_ID) does each workgroup of a 1-D grid land on?  Build: hipcc --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(x & 0xf);
    }
}
int main() {
    for (int cfg = 0; cfg < 3; ++cfg) {
        dim3 grid = cfg == 0 ? dim3(256, 1, 1) : cfg == 1 ? dim3(756, 1, 1) : dim3(54, 4, 1);
        int n = grid.x * grid.y * grid.z;
        int *d, *h = (int*)malloc(n * sizeof(int));
        hipMalloc(&d, n * sizeof(int));
        hipLaunchKernelGGL(probe, grid, dim3(512), 147456, 0, d);
        hipMemcpy(h, d, n * sizeof(int), hipMemcpyDeviceToHost);
        int match = 0;
        for (int i = 0; i < n; ++i) match += (h[i] == (i % 8));
        printf("grid (%d,%d,%d): xcc of first 24 linear blocks:", grid.x, grid.y, grid.z);
        for (int i = 0; i < 24 && i < n; ++i) printf(" %d", h[i]);
        printf("  | blocks with xcc == linear_id %% 8: %d / %d\n", match, n);
        hipFree(d);
        free(h);
    }
    return 0;
}
