// Ordered fold of per-workgroup partial sums (deterministic mode of the BatchNorm reductions); shared by elementwise.hip (BatchNorm
// kernels) and head.hip (BatchNorm + classifier tail).  Each translation unit gets its own copy (anonymous namespace).
#pragma once
#include "common.h"

namespace {
// deterministic mode: sums[i] = sum over the workgroups of the reduce pass of part[wg][i], in a fixed order (strided subsets
// per column in wg order, folded in subset order) -- the float atomics of the default mode arrive in any order, and a fixed-point
// integer sum has no range for both the forward moments (up to 1e10) and the backward ones (down to 1e-9)
__global__ __launch_bounds__(1024) void bn_part_fold_kernel(const float* __restrict__ part, double* __restrict__ sums, int nwg, int n) {
    // 64 columns x 16 row subsets per workgroup; every thread keeps four independent partial sums so that its loads overlap
    __shared__ double sub[16][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    if (c < n) {
        int w = g;
        for (; w + 48 < nwg; w += 64) {
            t0 += (double)part[(size_t)w * n + c], t1 += (double)part[(size_t)(w + 16) * n + c];
            t2 += (double)part[(size_t)(w + 32) * n + c], t3 += (double)part[(size_t)(w + 48) * n + c];
        }
        for (; w < nwg; w += 16) t0 += (double)part[(size_t)w * n + c];
    }
    sub[g][threadIdx.x & 63] = (t0 + t1) + (t2 + t3);
    __syncthreads();
    if (g == 0 && c < n) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += sub[q][threadIdx.x];
        sums[c] = t;
    }
}
}  // namespace
