// Stand-alone probe of the 4-wave / one-wave-per-SIMD GEMM K-loop (csrc/gen_gemm4.py): C = A B^T + bias, bf16 in, fp32 accumulate, bf16 out.
//   python3 ../../instageo-e2e-geospatial-ml_amd/csrc/gen_gemm4.py gemm4_gen.inc [key=value ...] && hipcc -O3 --offload-arch=gfx950 gemm4_probe.hip -o gemm4_probe
//   ./gemm4_probe [M N K]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <vector>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;
#include "gemm4_gen.inc"

struct G4P {
    const bf16_t* A;
    const bf16_t* B;
    bf16_t* C;
    const float* bias;
    int M, N, K;
    long lda, ldb, ldo;
    int alias;  // timing experiments (garbage results): bit 0 = every tile reads the A rows of row block 0, bit 1 = the B rows of column block 0
};

__device__ __forceinline__ unsigned pack_bf2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    bf2 r = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, r);
}

typedef __attribute__((address_space(3))) char* lds_char_ptr;
constexpr int G4_STAGES = 2 * 65536, G4_SMEM = G4_STAGES + 4 * 4096;

__global__ __launch_bounds__(256) void gemm4_kernel(G4P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N >> 8, tiles_m = (p.M + 255) >> 8, ntiles = tiles_m * tiles_n;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    if (my_tiles <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    const int lda2 = (int)(p.lda * 2), ldb2 = (int)(p.ldb * 2);
    const int nk = p.K >> 6, npair = (nk >> 1) - 2;

    const unsigned rowv = wave * 64 + (lane >> 3);
    const unsigned c16 = ((lane & 7) ^ ((lane >> 3) & 7)) << 4;
    const unsigned swz = ((lane >> 4) ^ (lane & 7)) << 4;
    const unsigned fa = lds_base + (wr * 128 + (lane & 15)) * 128 + swz;
    const unsigned fb = lds_base + (wc * 128 + (lane & 15)) * 128 + swz;
    const unsigned ldsw = lds_base + wave * 8192;

    int tile = tlo + jx;
    {
        const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
        const char* aptr = (const char*)p.A + (long)((p.alias & 1) ? 0 : bm) * 256 * lda2;
        const char* bptr = (const char*)p.B + (long)((p.alias & 2) ? 0 : bn) * 256 * ldb2;
        const int vrc = min(256, p.M - bm * 256) - 1;
        asm volatile(G4_ASM_PROLOGUE ::[aptr] "s"(aptr), [bptr] "s"(bptr), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc), [vrn] "s"(vrc), [ldsw] "s"(ldsw),
                     [rowv] "v"(rowv), [c16] "v"(c16), [fa] "v"(fa), [fb] "v"(fb)
                     : G4_CLOBBERS);
    }
    for (int t = 0; t < my_tiles; ++t, tile += nbx) {
        const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
        const int tn = (t + 1 < my_tiles) ? tile + nbx : tile;
        const int bm2 = tn / tiles_n, bn2 = tn - bm2 * tiles_n;
        const char* aptr = (const char*)p.A + (long)((p.alias & 1) ? 0 : bm) * 256 * lda2 + 256;
        const char* bptr = (const char*)p.B + (long)((p.alias & 2) ? 0 : bn) * 256 * ldb2 + 256;
        const char* anext = (const char*)p.A + (long)((p.alias & 1) ? 0 : bm2) * 256 * lda2;
        const char* bnext = (const char*)p.B + (long)((p.alias & 2) ? 0 : bn2) * 256 * ldb2;
        const int vrc = min(256, p.M - bm * 256) - 1, vrn = min(256, p.M - bm2 * 256) - 1;
        asm volatile(G4_ASM_TILE ::[aptr] "s"(aptr), [bptr] "s"(bptr), [anext] "s"(anext), [bnext] "s"(bnext), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc),
                     [vrn] "s"(vrn), [npair] "s"(npair), [ldsw] "s"(ldsw), [wave] "s"(wave), [rowv] "v"(rowv), [c16] "v"(c16), [fa] "v"(fa), [fb] "v"(fb)
                     : G4_CLOBBERS);
        // ---- epilogue: + bias, bf16, staged through a wave-private 4 KiB slab (16 rows x 128 columns), 16-byte row-contiguous stores
        char* st = smem + G4_STAGES + wave * 4096;
        const int n0 = bn * 256 + wc * 128;
        f32x4 bv[8];
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + ni * 16 + 4 * (lane >> 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
            bv[ni] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
        const int erow = lane & 15;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            f32x4 tt[8];
            g4_acc_row(mi, tt);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const f32x4 v = tt[ni] + bv[ni];
                uint2 u;
                u.x = pack_bf2(v[0], v[1]), u.y = pack_bf2(v[2], v[3]);
                const int chunk = ni * 2 + (lane >> 5);
                *reinterpret_cast<uint2*>(st + erow * 256 + (((chunk ^ erow) & 15) << 4) + ((lane >> 4) & 1) * 8) = u;
            }
            const int m0 = bm * 256 + wr * 128 + mi * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = j * 4 + (lane >> 4), ch = lane & 15;
                const uint4 u = *reinterpret_cast<const uint4*>(st + r * 256 + (((ch ^ r) & 15) << 4));
                if (m0 + r < p.M) *reinterpret_cast<uint4*>(p.C + (size_t)(m0 + r) * p.ldo + n0 + ch * 8) = u;
            }
        }
    }
}

// naive reference for sampled rows
__global__ void ref_kernel(G4P p, const int* rows, int nrows, float* out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, ri = blockIdx.y;
    if (n >= p.N) return;
    const int m = rows[ri];
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) {
        const float a = __uint_as_float((unsigned)p.A[(size_t)m * p.lda + k] << 16), b = __uint_as_float((unsigned)p.B[(size_t)n * p.ldb + k] << 16);
        s += a * b;
    }
    out[(size_t)ri * p.N + n] = s + (p.bias ? p.bias[n] : 0.f);
}

static bf16_t f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (bf16_t)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}
static float bf2f(bf16_t h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

int main(int argc, char** argv) {
    int M = argc > 1 ? atoi(argv[1]) : 42552, N = argc > 2 ? atoi(argv[2]) : 2304, K = argc > 3 ? atoi(argv[3]) : 768;
    const int reps = argc > 4 ? atoi(argv[4]) : 60;
    const int alias = argc > 5 ? atoi(argv[5]) : 0;
    printf("gemm4 probe: M=%d N=%d K=%d alias=%d\n", M, N, K, alias);
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<float> hbias(N);
    uint32_t s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f; };
    for (auto& v : hA) v = f2bf(rnd());
    for (auto& v : hB) v = f2bf(rnd() * 0.05f);
    for (auto& v : hbias) v = rnd();
    bf16_t *dA, *dB, *dC;
    float* dbias;
    CK(hipMalloc(&dA, hA.size() * 2 + (1 << 20)));  // slack: the L2 prefetch of the last rows runs past K
    CK(hipMalloc(&dB, hB.size() * 2 + (1 << 20)));
    CK(hipMalloc(&dC, (size_t)M * N * 2));
    CK(hipMalloc(&dbias, N * 4));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dbias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0xFF, (size_t)M * N * 2));
    G4P p{dA, dB, dC, dbias, M, N, K, K, K, N, alias};
    CK(hipFuncSetAttribute((const void*)gemm4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G4_SMEM));
    const int ntiles = ((M + 255) / 256) * (N / 256);
    const int grid = ntiles < 256 ? ntiles : 256;
    hipLaunchKernelGGL(gemm4_kernel, dim3(grid), dim3(256), G4_SMEM, 0, p);
    CK(hipDeviceSynchronize());
    // check sampled rows
    std::vector<int> rows;
    for (int i = 0; i < 61; ++i) rows.push_back((int)(((long)i * 7919 + 13) % M));
    rows.push_back(0), rows.push_back(M - 1), rows.push_back(255), rows.push_back(256 < M ? 256 : 0), rows.push_back(M > 300 ? M - 257 : 0);
    int* drows;
    float* dref;
    CK(hipMalloc(&drows, rows.size() * 4));
    CK(hipMalloc(&dref, rows.size() * (size_t)N * 4));
    CK(hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ref_kernel, dim3((N + 255) / 256, (unsigned)rows.size()), dim3(256), 0, 0, p, drows, (int)rows.size(), dref);
    CK(hipDeviceSynchronize());
    std::vector<float> href(rows.size() * (size_t)N);
    std::vector<bf16_t> hC((size_t)M * N);
    CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hC.data(), dC, hC.size() * 2, hipMemcpyDeviceToHost));
    double maxerr = 0;
    long bad = 0;
    for (size_t ri = 0; ri < rows.size(); ++ri)
        for (int n = 0; n < N; ++n) {
            const float r = href[ri * N + n], c = bf2f(hC[(size_t)rows[ri] * N + n]);
            const double e = fabs(r - c), tol = 0.02 + 0.01 * fabs(r);
            if (!(e <= tol)) {
                if (bad < 8) printf("  mismatch row %d col %d: ref %f got %f\n", rows[ri], n, r, c);
                ++bad;
            }
            if (e > maxerr) maxerr = e;
        }
    printf("check: %zu rows x %d cols, max err %.4g, mismatches %ld -> %s\n", rows.size(), N, maxerr, bad, bad ? "FAIL" : "ok");
    // timing
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm4_kernel, dim3(grid), dim3(256), G4_SMEM, 0, p);
    double best = 1e30, sum = 0;
    const int rounds = 8;
    std::vector<double> all;
    for (int r = 0; r < rounds; ++r) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm4_kernel, dim3(grid), dim3(256), G4_SMEM, 0, p);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        sum += us;
        all.push_back(us);
        if (us < best) best = us;
    }
    std::sort(all.begin(), all.end());
    const double med = all[all.size() / 2];
    const double fl = 2.0 * M * N * K;
    printf("time: median %.1f us (%.0f TF/s), best %.1f us (%.0f TF/s)\n", med, fl / med / 1e6, best, fl / best / 1e6);
    return bad ? 2 : 0;
}
