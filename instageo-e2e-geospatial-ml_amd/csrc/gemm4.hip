// v4: the forward ("NT", both operands K-contiguous) linear GEMMs of the timm Block (pritvhi.py:446-456: qkv / proj / fc1 / fc2 and their data
// gradients through the transposed weight copy) as a 256 x 256 x 64 kernel with ONE wave per SIMD -- 4 waves, 128 x 128 of the tile per wave.
//
//   C[m][n] = sum_k A[m][k] * B[n][k]      A = activations [M][K] (bf16), B = nn.Linear weight [N][K] (bf16), fp32 accumulate
//
// * Why a second engine next to gemm8.hip (2 waves per SIMD, 128 x 64 per wave): per K-tile a CU reads 128 KiB of fragments from LDS instead of
//   192, has one barrier instead of four, and no wave ever waits for its SIMD partner's read phase.  hipcc cannot hold 256 accumulators + 128
//   fragment registers without spilling (rounds 2 and 5 tried), so the K-loop of a tile is ONE inline-asm block with hand-assigned registers,
//   written by gen_gemm4.py (the Makefile runs it): accumulators in a0..a255, two fragment sets in v128..v255, the LDS-DMA issues woven in
//   behind every 4th MFMA of the loop's second half, fragment reads behind every 3rd.  This file holds the C++ around it: the persistent tile
//   loop, the operand addressing and the epilogues, which read the accumulators back with v_accvgpr_read_b32 (g4_acc_row).
// * LDS (all 160 KiB): a ring of three A slots + two B stages of 256 rows x 128 B, 16-byte chunks swizzled by row & 7 on the DMA source address
//   and on the fragment reads (as gemm8.hip).  The A slot whose K-tile was consumed last is free until the next tile's first iteration: it is
//   the epilogue's staging buffer (4 x 8 KiB, wave-private).  The ring state (three slot offsets) lives in registers across tiles.
// * Epilogues (same contracts as gemm8.hip's kinds; the bias is added here, not in the accumulator init):
//     kind 0: bf16 store of act(acc + bias) [+ the gelu' copy for the backward pass]
//     kind 1: fp32 out = resid + acc + bias
//     kind 2: bf16 store of acc * dact (dact = the gelu' the forward saved) with optional fused column sums (the bias gradient of fc1)
//   The split precision mode (bf16x3) stays on gemm8.hip's paired K-tiles.
#include <stdlib.h>

#include "common.h"
#include "gemm4_gen.inc"

namespace {

typedef __attribute__((address_space(3))) char* lds_char_ptr;
constexpr int G4_SMEM = 5 * 32768;  // A slots at 0 / 32 / 64 KiB, B stages at 96 / 128 KiB (gen_gemm4.py)

template <bool NT = false, typename T>
__device__ __forceinline__ void g4_store(T* ptr, const T& v) {
    if constexpr (NT) {
        static_assert(sizeof(T) == 16, "16-byte stores");
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), reinterpret_cast<u32x4_t*>(ptr));
    } else {
        *ptr = v;
    }
}

// NSEG = 1: plain bf16 operands.  NSEG = 2: the split precision mode (bf16x3) on PAIRED K-tiles (gen_gemm4.py, second half): operands are
// (hi, lo) pairs with lo a 32-bit distance above hi, a K-tile covers 32 reduction elements with hi and lo side by side in the LDS row, three
// products of 64 MFMAs per K-tile (hi hi, hi(B) lo(A), lo(B) hi(A)); outputs are split again (kinds 0, 2) or fp32 (kind 1).
template <int KIND, int ACT, bool DACT, int NSEG = 1>
__global__ __launch_bounds__(256) void gemm4_kernel(G8Params p) {
    constexpr bool PAIR = NSEG == 2;
    static_assert(!(PAIR && DACT), "gemm4: the paired form does not save gelu' (four staging slabs): that kind stays on gemm8.hip");
    constexpr int KB = PAIR ? 64 : 128;  // bytes of one operand row per K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = p.N >> 8, tiles_m = (p.M + 255) >> 8, ntiles = tiles_m * tiles_n;
    // persistent, XCD-aware (as gemm8.hip): workgroups are dealt round-robin over the 8 XCDs; XCD x owns a contiguous tile range
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    if (my_tiles <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    const int lda2 = (int)(p.lda * 2), ldb2 = (int)(p.ldb * 2);
    const int nk = PAIR ? p.K >> 5 : p.K >> 6, npair = (nk >> 1) - 2;

    // lane constants of the asm blocks (gen_gemm4.py: setup): DMA row and source chunk, fragment read bases, DMA destination of this wave
    const unsigned rowv = wave * 64 + (lane >> 3);
    const unsigned sc = (lane & 7) ^ ((lane >> 3) & 7);  // source chunk of this lane's LDS chunk; paired: chunks 0-3 = hi, 4-7 = lo of the same 32 elements
    const unsigned c16 = sc << 4;
    const unsigned c16a = PAIR ? ((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.a[2] - (const char*)p.a[0]) : 0u) : c16;
    const unsigned c16b = PAIR ? ((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.b[1] - (const char*)p.b[0]) : 0u) : c16;
    const unsigned swz = ((lane >> 4) ^ (lane & 7)) << 4;
    const unsigned fa = lds_base + (wr * 128 + (lane & 15)) * 128 + swz;
    const unsigned fb = lds_base + (wc * 128 + (lane & 15)) * 128 + swz;
    const unsigned ldsw = lds_base + wave * 8192;

    unsigned a0 = 0, a1 = 32768, a2 = 65536;  // A ring: slot offsets of K-tiles kt, kt + 1, kt + 2 (rotated by the asm blocks)
    int tile = tlo + jx;
    {
        const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
        const char* aptr = (const char*)p.a[0] + (long)bm * 256 * lda2;
        const char* bptr = (const char*)p.b[0] + (long)bn * 256 * ldb2;
        const int vrc = min(256, p.M - bm * 256) - 1;
        if constexpr (PAIR)
            asm volatile(G4P_ASM_PROLOGUE ::[aptr] "s"(aptr), [bptr] "s"(bptr), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc), [vrn] "s"(vrc),
                         [ldsw] "s"(ldsw), [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [rowv] "v"(rowv), [c16a] "v"(c16a), [c16b] "v"(c16b), [fa] "v"(fa),
                         [fb] "v"(fb)
                         : G4P_CLOBBERS);
        else
            asm volatile(G4_ASM_PROLOGUE ::[aptr] "s"(aptr), [bptr] "s"(bptr), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc), [vrn] "s"(vrc), [ldsw] "s"(ldsw),
                         [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [rowv] "v"(rowv), [c16] "v"(c16), [fa] "v"(fa), [fb] "v"(fb)
                         : G4_CLOBBERS);
    }
    for (int t = 0; t < my_tiles; ++t, tile += nbx) {
        const int bm = tile / tiles_n, bn = tile - bm * tiles_n;
        const int tn = (t + 1 < my_tiles) ? tile + nbx : tile;  // no next tile: the loop's last two DMA rounds re-fetch this tile's first K-tiles
        const int bm2 = tn / tiles_n, bn2 = tn - bm2 * tiles_n;
        const char* aptr = (const char*)p.a[0] + (long)bm * 256 * lda2 + 2 * KB;  // K-tile 2 (0 and 1 are in flight)
        const char* bptr = (const char*)p.b[0] + (long)bn * 256 * ldb2 + 2 * KB;
        const char* anext = (const char*)p.a[0] + (long)bm2 * 256 * lda2;
        const char* bnext = (const char*)p.b[0] + (long)bn2 * 256 * ldb2;
        const int vrc = min(256, p.M - bm * 256) - 1, vrn = min(256, p.M - bm2 * 256) - 1;
        if constexpr (PAIR)
            asm volatile(G4P_ASM_TILE : [a0] "+s"(a0), [a1] "+s"(a1), [a2] "+s"(a2) : [aptr] "s"(aptr), [bptr] "s"(bptr), [anext] "s"(anext), [bnext] "s"(bnext),
                         [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc), [vrn] "s"(vrn), [npair] "s"(npair), [ldsw] "s"(ldsw), [rowv] "v"(rowv),
                         [c16a] "v"(c16a), [c16b] "v"(c16b), [fa] "v"(fa), [fb] "v"(fb)
                         : G4P_CLOBBERS);
        else
            asm volatile(G4_ASM_TILE : [a0] "+s"(a0), [a1] "+s"(a1), [a2] "+s"(a2) : [aptr] "s"(aptr), [bptr] "s"(bptr), [anext] "s"(anext), [bnext] "s"(bnext), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [vrc] "s"(vrc),
                         [vrn] "s"(vrn), [npair] "s"(npair), [ldsw] "s"(ldsw), [wave] "s"(wave), [rowv] "v"(rowv), [c16] "v"(c16), [fa] "v"(fa), [fb] "v"(fb)
                         : G4_CLOBBERS);
        // ================= epilogue (the next tile's K-tiles 0 and 1 are in flight) =================
        // accumulator block (mi, ni) of this lane: C[m = mi 16 + (lane & 15)][n = ni 16 + 4 (lane >> 4) .. + 3] of the wave's 128 x 128
        // (the lane index is re-materialised behind an empty asm: otherwise hipcc hoists the epilogue's per-lane address arithmetic out of the
        // tile loop, keeps it live across the K-loop block -- which clobbers v90..v255 -- and spills it)
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
#define lane lane_e
        char* st = smem + a2 + wave * 8192;  // the A slot this tile's last K-tile has left
        const int n0 = bn * 256 + wc * 128, mw = bm * 256 + wr * 128;
        const int erow = lane & 15, eq = lane >> 4;
        if constexpr (KIND == 0) {
            // bf16 store of act(acc + bias): the 16 x 128 row block is staged as bf16 (4 KiB; + 4 KiB for the gelu' copy), 16-byte chunk ^= row,
            // and leaves as row-contiguous 16-byte stores (a wave-instruction = 4 rows x 256 B)
            f32x4 bv[8];
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + ni * 16 + 4 * eq) : make_float4(0.f, 0.f, 0.f, 0.f);
                bv[ni] = f32x4{b4.x, b4.y, b4.z, b4.w};
            }
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                f32x4 tt[8];
                g4_acc_row(mi, tt);
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) {
                    const f32x4 a = tt[ni] + bv[ni];
                    float v[4] = {a[0], a[1], a[2], a[3]}, d[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (ACT == 1) {
                        f32x2 g0, g1, d0, d1;
                        gelu_erf_pair<DACT>(f32x2{v[0], v[1]}, g0, d0);
                        gelu_erf_pair<DACT>(f32x2{v[2], v[3]}, g1, d1);
                        v[0] = g0.x, v[1] = g0.y, v[2] = g1.x, v[3] = g1.y;
                        if constexpr (DACT) d[0] = d0.x, d[1] = d0.y, d[2] = d1.x, d[3] = d1.y;
                    }
                    const int chunk = ni * 2 + (eq >> 1);
                    const int off = erow * 256 + (((chunk ^ erow) & 15) << 4) + (eq & 1) * 8;
                    uint2 u;
                    u.x = pack_bf2(v[0], v[1]), u.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(st + off) = u;
                    if constexpr (PAIR) {  // split output: lo = bf16(v - hi) in the second slab
                        uint2 ul;
                        ul.x = pack_bf2(v[0] - __uint_as_float(u.x << 16), v[1] - __uint_as_float(u.x & 0xffff0000u));
                        ul.y = pack_bf2(v[2] - __uint_as_float(u.y << 16), v[3] - __uint_as_float(u.y & 0xffff0000u));
                        *reinterpret_cast<uint2*>(st + 4096 + off) = ul;
                    }
                    if constexpr (DACT) {
                        uint2 ud;
                        ud.x = pack_bf2(d[0], d[1]), ud.y = pack_bf2(d[2], d[3]);
                        *reinterpret_cast<uint2*>(st + 4096 + off) = ud;
                    }
                }
                const int m0 = mw + mi * 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = j * 4 + eq, ch = lane & 15;
                    const int off = r * 256 + (((ch ^ r) & 15) << 4);
                    const uint4 ux = *reinterpret_cast<const uint4*>(st + off);
                    uint4 ud = ux;
                    if constexpr (DACT || PAIR) ud = *reinterpret_cast<const uint4*>(st + 4096 + off);
                    if (m0 + r < p.M) {
                        const size_t o = (size_t)(m0 + r) * p.ldo + n0 + ch * 8;
                        g4_store(reinterpret_cast<uint4*>(p.out_hi + o), ux);
                        if constexpr (PAIR) g4_store(reinterpret_cast<uint4*>(p.out_lo + o), ud);
                        // gelu' is read exactly once, by the backward pass: streaming (non-temporal) store
                        if constexpr (DACT) g4_store<true>(reinterpret_cast<uint4*>(p.dact_hi + o), ud);
                    }
                }
            }
        } else if constexpr (KIND == 1) {
            // fp32 out = resid + acc + bias: the 16 x 128 fp32 row block is staged (8 KiB, 16-byte chunk ^= row & 7) and read back row-contiguous
            // (32 lanes = one 512-byte row); the residual rows of row block mi + 1 are loaded before row block mi is stored
            const int rc = lane & 31, rr = lane >> 5;
            const float4 b4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n0 + rc * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            // wave-uniform bases + 32-bit lane offsets (scalar-base addressing: half the address registers of per-lane 64-bit pointers)
            const float* rbase = p.resid + (size_t)mw * p.ldo + n0;
            float* obase = p.outf + (size_t)mw * p.ldo + n0;
            const int mlast = p.M - 1 - mw;  // last valid row of the wave's 128 (>= 0: a tile has at least one row; may be < 0 for wr = 1 -> clamp)
            const unsigned ldo4 = (unsigned)p.ldo;
            float4 rs[2][8];
#define G4_RESID_LOAD(MI, DST)                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) {                                             \
        const int r_ = max(min((MI)*16 + i_ * 2 + rr, mlast), -mw);                                \
        DST[i_] = *reinterpret_cast<const float4*>(rbase + (long)r_ * ldo4 + rc * 4);              \
    }
            G4_RESID_LOAD(0, rs[0])
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                f32x4 tt[8];
                g4_acc_row(mi, tt);
                if (mi + 1 < 8) G4_RESID_LOAD(mi + 1, rs[(mi + 1) & 1])
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) {
                    const int c4 = ni * 4 + eq;
                    *reinterpret_cast<f32x4*>(st + erow * 512 + ((c4 ^ (erow & 7)) << 4)) = tt[ni];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int r = i * 2 + rr;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(st + r * 512 + ((rc ^ (r & 7)) << 4));
                    const int rl = mi * 16 + r;
                    if (rl <= mlast) {
                        const float4 rv = rs[mi & 1][i];
                        g4_store(reinterpret_cast<float4*>(obase + (unsigned)rl * ldo4 + rc * 4),
                                 make_float4(rv.x + a[0] + b4.x, rv.y + a[1] + b4.y, rv.z + a[2] + b4.z, rv.w + a[3] + b4.w));
                    }
                }
            }
#undef G4_RESID_LOAD
        } else {
            // data gradient with an elementwise factor: dx = acc * dact (the gelu' the forward saved) [+ fused column sums of dx].  The fp32 row
            // block is staged and read back 8 columns per lane (16 lanes = one row), so the factor is one coalesced 16-byte load per 8 values;
            // the four factor loads of a row block go out before its staging round trip
            const int rcol = (lane & 15) * 8;
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
            uint4 fh[2][4], fl[2][4];
            // gelu' was saved by the forward pass and is read exactly once: streaming (non-temporal) loads; the four loads of row block mi + 1 go
            // out before row block mi is staged and stored (one wave per SIMD: nothing else hides their latency)
#define G4_FACTOR_LOAD(MI, DST, DSTL)                                                                                                    \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                                                                             \
        const int m_ = min(mw + (MI)*16 + j_ * 4 + eq, p.M - 1);                                                                   \
        DST[j_] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p.dact_hi + (size_t)m_ * p.ldo + n0 + rcol))); \
        if constexpr (PAIR)                                                                                                        \
            DSTL[j_] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p.dact_lo + (size_t)m_ * p.ldo + n0 + rcol))); \
    }
            G4_FACTOR_LOAD(0, fh[0], fl[0])
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const int m0 = mw + mi * 16;
                if (mi + 1 < 8) G4_FACTOR_LOAD(mi + 1, fh[(mi + 1) & 1], fl[(mi + 1) & 1])
                f32x4 tt[8];
                g4_acc_row(mi, tt);
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) {
                    const int c4 = ni * 4 + eq;
                    *reinterpret_cast<f32x4*>(st + erow * 512 + ((c4 ^ (erow & 7)) << 4)) = tt[ni];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = j * 4 + eq, cc = 2 * (lane & 15);
                    const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + r * 512 + ((cc ^ (r & 7)) << 4));
                    const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + r * 512 + (((cc + 1) ^ (r & 7)) << 4));
                    float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    float f[8];
                    unpack8(fh[mi & 1][j], f);
                    if constexpr (PAIR) {
                        float f2[8];
                        unpack8(fl[mi & 1][j], f2);
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] += f2[e];
                    }
                    if (m0 + r < p.M) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] *= f[e], cs[e] += v[e];
                        const uint4 u = pack8(v);
                        g4_store(reinterpret_cast<uint4*>(p.out_hi + (size_t)(m0 + r) * p.ldo + n0 + rcol), u);
                        if constexpr (PAIR) {
                            float hv[8], rv[8];
                            unpack8(u, hv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) rv[e] = v[e] - hv[e];
                            g4_store(reinterpret_cast<uint4*>(p.out_lo + (size_t)(m0 + r) * p.ldo + n0 + rcol), pack8(rv));
                        }
                    }
                }
            }
#undef G4_FACTOR_LOAD
            if (p.colsum) {  // lanes with equal (lane & 15) own the same 8 columns: fold over lane >> 4, one atomic per column
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = cs[j];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane < 16) ig_red_add(p.colsum + n0 + rcol + j, v);
                }
            }
        }
#undef lane
    }
}

// IG_GEMM4: 0 = off (gemm8.hip serves every shape), 1 = default
inline int g4_env() {
    const char* e = getenv("IG_GEMM4");
    return e ? atoi(e) : 1;
}

template <int KIND, int ACT, bool DACT, int NSEG = 1>
int g4_launch(const G8Params& p, int grid, hipStream_t st) {
    auto kern = gemm4_kernel<KIND, ACT, DACT, NSEG>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G4_SMEM) != hipSuccess) {
            ig_set_error("gemm4: could not reserve %d bytes of LDS", G4_SMEM);
            return IG_ERR_HIP;
        }
        attr_done = true;
    }
    if (NSEG == 1) ig_note_kernel("gemm4_kernel<%d,%d,%s>", KIND, ACT, DACT ? "true" : "false");
    else ig_note_kernel("gemm4_kernel<%d,%d,%s,%d>", KIND, ACT, DACT ? "true" : "false", NSEG);
    ig_note_grid(grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), G4_SMEM, st, p);
    return ig_check_launch("gemm4");
}

}  // namespace
IG_DET_TU(gemm4)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

// IG_ERR_UNSUPPORTED (no error string) when the shape / mode is not covered: ig_gemm8_nt goes on to its own instances.
int ig_gemm4_nt(const G8Params& p, void* stream) {
    if (!g4_env()) return IG_ERR_UNSUPPORTED;
    if (p.M <= 0 || (p.nseg != 1 && p.nseg != 3)) return IG_ERR_UNSUPPORTED;
    const bool split = p.nseg == 3;  // bf16x3: (a[0], b[0]) = hi hi, (a[1], b[1]) = hi lo, (a[2], b[2]) = lo hi  (gemm8.hip)
    if (split) {
        // the paired form needs both lo tensors above their hi tensors, 16-byte aligned, near enough for the 32-bit lane offsets (ops.BT allocates
        // hi and lo as one block); IG_G8_PAIR=0 (the three-pass A/B arm of gemm8.hip) also keeps the split mode off this engine
        const char* e = getenv("IG_G8_PAIR");
        const long dA = (const char*)p.a[2] - (const char*)p.a[0], dB = (const char*)p.b[1] - (const char*)p.b[0];
        const bool pair = (!e || atoi(e) != 0) && dA > 0 && dB > 0 && !(dA & 15) && !(dB & 15) && dA + 257L * p.lda * 2 < (1L << 32) &&
                          dB + 257L * p.ldb * 2 < (1L << 32);
        if (!pair) return IG_ERR_UNSUPPORTED;
    }
    if ((p.N & 255) || (p.K & 127) || p.K < 256) return IG_ERR_UNSUPPORTED;  // 256-wide tiles; an even number (>= 4) of K-tiles
    if (p.lda * 2 >= (1L << 24) || p.ldb * 2 >= (1L << 24)) return IG_ERR_UNSUPPORTED;  // 24-bit offset multiply
    if ((p.lda & 7) || (p.ldb & 7) || (p.ldo & 7) || ((uintptr_t)p.a[0] & 15) || ((uintptr_t)p.b[0] & 15)) return IG_ERR_UNSUPPORTED;
    const int ntiles = ((p.M + 255) >> 8) * (p.N >> 8);
    if (ntiles < 128 && g4_env() != 2) return IG_ERR_UNSUPPORTED;  // below one tile per CU the 128 x 128 instance of gemm8.hip takes over (as there)
    // Routing by measurement (same-process A/B at M = 42552, profiles/r06_gemm4_ring_vs_stages.txt): the plain bf16 store (qkv -4 %, the N = 768 /
    // K = 3072 data gradient -9 %), the fp32 residual kind (fc2 -8 %, proj -4 %) and dx * gelu' (-6 %) win; the GELU epilogues are bound by VALU
    // issue -- one wave per SIMD issues a vector instruction every 4 cycles where the two co-resident waves of gemm8.hip issue one every 2 --
    // and come out equal (+0.5 %): they stay on the 8-phase engine.  IG_GEMM4=2 forces every kind this engine has (tests).
    if (g4_env() != 2 && p.act != 0) return IG_ERR_UNSUPPORTED;
    const int grid = ig_tile_grid(ntiles, 1);
    hipStream_t st = (hipStream_t)stream;
    if (p.kind == 0) {
        if (!p.out_hi || split != (p.out_lo != nullptr)) return IG_ERR_UNSUPPORTED;
        const bool dact = p.dact_hi != nullptr;
        if (split) {
            if (dact) return IG_ERR_UNSUPPORTED;  // gelu' saved in two halves needs four staging slabs: gemm8.hip
            return p.act == 0 ? g4_launch<0, 0, false, 2>(p, grid, st) : g4_launch<0, 1, false, 2>(p, grid, st);
        }
        if (p.dact_lo) return IG_ERR_UNSUPPORTED;
        if (p.act == 0) return dact ? IG_ERR_UNSUPPORTED : g4_launch<0, 0, false>(p, grid, st);
        return dact ? g4_launch<0, 1, true>(p, grid, st) : g4_launch<0, 1, false>(p, grid, st);
    }
    if (p.kind == 1) {
        if (!p.outf || !p.resid) return IG_ERR_UNSUPPORTED;
        return split ? g4_launch<1, 0, false, 2>(p, grid, st) : g4_launch<1, 0, false>(p, grid, st);
    }
    if (p.kind == 2) {
        if (!p.dact_hi || !p.out_hi || p.bias || split != (p.out_lo != nullptr) || split != (p.dact_lo != nullptr)) return IG_ERR_UNSUPPORTED;
        return split ? g4_launch<2, 0, false, 2>(p, grid, st) : g4_launch<2, 0, false>(p, grid, st);
    }
    return IG_ERR_UNSUPPORTED;
}
