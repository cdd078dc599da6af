// Generic bf16 MFMA "gather-GEMM" for gfx950 and every matmul-shaped op of the Prithvi path.
//
//   C[m][n] = sum_seg sum_k A_seg(m,k) * B_seg(n,k)           (fp32 accumulate)
//
// * 128x128 output tile, BK=64, 256 threads = 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 bf16.
// * Operands are fetched in 16-byte units (8 bf16) through *loader functors* that return a global
//   pointer or NULL (zero fill): plain row-major matrices, NHWC convolution gathers (3x3 conv, stride-2
//   transposed conv and their gradients) and weight views all go through the same kernel.
// * Each operand is either K-contiguous (LDS image [row][64 k], ds_read_b128 fragments) or
//   K-strided / "TR" (LDS image [64 k][128 rows], ds_read_b64_tr_b16 hardware-transposed fragments).
//   NT = linear/conv forward, (A plain, B TR) = dgrad, (A TR, B TR) = wgrad.
// * NSEG=3 runs the split-bf16 (hi*hi + hi*lo + lo*hi) precision mode through the same loop.
// * The MFMA is issued as D = Bfrag x Afrag so that a lane owns 4 consecutive n of one row m:
//   epilogues store 8-byte (bf16) / 16-byte (fp32) row-contiguous pieces.
// * register-staged global->LDS double buffering: loads of tile t+1 are in flight during the MFMAs of
//   tile t; one barrier per K-step.  XOR swizzles keep both fragment read kinds bank-conflict free.
//
// Reference ops replaced (file:line in /root/reference): nn.Conv3d patch embed pritvhi.py:243-268;
// timm Block linears (qkv/proj/fc1/fc2) pritvhi.py:446-456; nn.ConvTranspose2d / nn.Conv2d of the
// decode head model.py:361-375.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTHR = 256;
constexpr int TILE_BYTES = 16384;           // one operand tile
constexpr int SMEM_BYTES = 4 * TILE_BYTES;  // 2 buffers x (A + B)

// LDS images ------------------------------------------------------------------------------------
// K-contiguous tile: up to 128 rows x 8 chunks(16 B); chunk ^= row&7  (ds_read_b128 conflict-free)
__device__ __forceinline__ int lds_kc(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }
// TR tile: 64 k-rows x UPR chunks (UPR = 16 / 12 / 8 for a 128 / 96 / 64-column tile).  A ds_read_b64_tr_b16 pass covers
// k-rows {k..k+3, k+8..k+11} x 32 B; the chunk-pair XOR key separates the rows that alias in the 256-byte bank span:
//   256 B rows: all 8 alias        -> 3-bit key (k&3, k>>3)
//   192 B rows: k and k+8 alias    -> 1-bit key (k>>3)      (rows k..k+3 sit 64 B apart)
//   128 B rows: k, k+2, k+8, k+10  -> 2-bit key (k>>1, k>>3)
template <int UPR>
__device__ __forceinline__ int lds_trw(int k, int c) {
    static_assert(UPR == 16 || UPR == 12 || UPR == 8, "unsupported TR tile width");
    int key;
    if constexpr (UPR == 16) key = (k & 3) | (((k >> 3) & 1) << 2);
    else if constexpr (UPR == 12) key = (k >> 3) & 1;
    else key = ((k >> 1) & 1) | (((k >> 3) & 1) << 1);
    return k * (UPR * 16) + ((c ^ (key << 1)) << 4);
}
__device__ __forceinline__ int lds_tr(int k, int c) { return lds_trw<16>(k, c); }

template <bool TR, int UPR = 16>
__device__ __forceinline__ bf16x8_t read_frag(const char* tile, int row0, int s, int lane) {
    if constexpr (!TR) {
        int r = row0 + (lane & 15);
        int c = s * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8_t*>(tile + lds_kc(r, c));
    } else {
        int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
        int k0 = s * 32 + 8 * g + q;
        int col = row0 + 4 * p;
        int chunk = col >> 3, sub = (col & 7) * 2;
        typedef __attribute__((address_space(3))) s16x4* lds_ptr;
        s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + lds_trw<UPR>(k0, chunk) + sub));
        s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + lds_trw<UPR>(k0 + 4, chunk) + sub));
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        s16x8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8_t, r);
    }
}

// 64-byte-row K-contiguous image (BK = 32): physical chunk of logical chunk c in row r (conflict-free for ds_read_b128)
__device__ __forceinline__ int kc32_off(int r, int c) { return r * 64 + ((c ^ ((0x1230 >> (4 * ((r >> 2) & 3))) & 3)) << 4); }

// v1: 4 waves laid out WM x WN (2 x 2, 4 x 1 or 1 x 4), each owning MT x NT MFMA tiles of 16 x 16 => block tile
// (16 WM MT) x (16 WN NT).  The decode head's channel counts are multiples of 48 (48, 96, 192, 384): 48-, 96- and
// 64-wide tiles avoid the 25-60 % of wasted MFMA columns (or rows, for the weight gradients) a fixed 128 x 128 tile
// would spend on them, and the smaller LDS footprint lets a third workgroup share the CU.
template <class AL, class BL, class EP, bool A_TR, bool B_TR, int NSEG, int MT, int NT, int WM, int BKT = 64>
__global__ __launch_bounds__(NTHR) void gemm_kernel(AL al, BL bl, EP ep, int M, int N, int K, int tiles_n, int kchunk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WN = 4 / WM;
    const int wm = wave / WN, wn = wave % WN;
    constexpr int BME = WM * MT * 16, BNE = WN * NT * 16;  // block tile
    constexpr int UPA = BME / 8, UPB = BNE / 8;            // 16-byte units per k-row of a TR tile
    constexpr int UPA_L = UPA <= 8 ? 8 : UPA <= 12 ? 12 : 16, UPB_L = UPB <= 8 ? 8 : UPB <= 12 ? 12 : 16;  // LDS row pitch
    // BKT = K-step: 64, or 32 for the small-channel convolutions (half the LDS => twice the resident workgroups; those
    // tiles have only a few K-steps and every step exposes a global-load round trip, so concurrency is what hides it)
    constexpr int UPK = BKT / 8;                           // 16-byte units per K-contiguous row
    constexpr int UA = BME * UPK, UB = BNE * UPK;          // units per operand tile (same count for TR images)
    constexpr int NA = (UA + NTHR - 1) / NTHR, NB = (UB + NTHR - 1) / NTHR;  // staged units per thread
    constexpr int TA_BYTES = A_TR ? BKT * 16 * UPA_L : BKT * 2 * BME, TB_BYTES = B_TR ? BKT * 16 * UPB_L : BKT * 2 * BNE;
    constexpr int BUF_BYTES = TA_BYTES + TB_BYTES;
    // XCD-aware tile order: hardware deals consecutive workgroups round-robin over the 8 XCDs (private L2s);
    // remap so that each XCD owns a CONTIGUOUS run of tiles (neighbours share the A row panel / B panels).
    int bid = blockIdx.x;
    int ysplit = blockIdx.y;
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    if constexpr (EP::kStagedAtomic) {
        // split-K weight gradients on a 1-D grid: `bid` now enumerates the (split, tile) pairs XCD-contiguously in split-major
        // order, so one XCD works inside one (at most two) K ranges -- its resident workgroups stream the same operand rows --
        // instead of every XCD touching every range (see gemm2_kernel)
        if (ep.pairs > 0) {
            ysplit = bid / ep.ntiles;
            bid -= ysplit * ep.ntiles;
            ep.split = ysplit;
        }
    }
    const int bm = bid / tiles_n, bn = bid - bm * tiles_n;
    al.init(blockIdx.z);
    bl.init(blockIdx.z);
    ep.init(blockIdx.z);
    if (al.kdim() >= 0) K = al.kdim();
    // split-K (wgrad): split ysplit owns k-tiles [kt0, kt0+nk) of every segment
    const int nk_all = (K + BKT - 1) / BKT;
    const int kt0 = ysplit * kchunk;
    const int nk = min(kchunk, nk_all - kt0);
    if (nk <= 0) return;
    const int total = nk * NSEG;

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[NA], rb[NB];
    // row decode hoisted out of the K loop (K-contiguous operands: a thread keeps the same rows for the whole tile);
    // the convolution gathers were issue-bound on integer divisions recomputed per 16-byte unit per K-step
    typename AL::Row arow[NA];
    typename BL::Row brow[NB];
    typename AL::Col acolT[NA];  // TR operands: a thread keeps the same column units for the whole tile
    typename BL::Col bcolT[NB];
    if constexpr (!A_TR) {
#pragma unroll
        for (int i = 0; i < NA; ++i) arow[i] = al.row(bm * BME + (tid + i * NTHR) / UPK);
    } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) acolT[i] = al.col(bm * UPA + (tid + i * NTHR) % UPA);
    }
    if constexpr (!B_TR) {
#pragma unroll
        for (int i = 0; i < NB; ++i) brow[i] = bl.row(bn * BNE + (tid + i * NTHR) / UPK);
    } else {
#pragma unroll
        for (int i = 0; i < NB; ++i) bcolT[i] = bl.col(bn * UPB + (tid + i * NTHR) % UPB);
    }

    // Loads are UNCONDITIONAL (invalid units read a dummy valid address and are zeroed by a select): a
    // branch around each load makes hipcc wait vmcnt(0) per load and serialises the whole tile fetch.
#define GEMM_GLOAD(IT)                                                                                   \
    {                                                                                                     \
        const int seg_ = (NSEG == 1) ? 0 : (IT) / nk;                                                     \
        const int kt_ = kt0 + (IT)-seg_ * nk;                                                             \
        const typename AL::Col acol_ = al.col(kt_ * UPK + (tid % UPK));                                   \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                  \
            const int u = tid + i * NTHR;                                                                 \
            bool ok;                                                                                      \
            const bf16_t* p;                                                                              \
            if constexpr (A_TR) p = al.ptr_tr(seg_, kt_ * BKT + u / UPA, acolT[i], ok);                   \
            else p = al.at(seg_, arow[i], acol_, ok);                                                     \
            if (UA % NTHR) ok = ok && u < UA; /* unit beyond the tile */                                  \
            uint4 v = *reinterpret_cast<const uint4*>(p);                                                 \
            ra[i] = ok ? v : make_uint4(0, 0, 0, 0);                                                      \
        }                                                                                                 \
        const typename BL::Col bcol_ = bl.col(kt_ * UPK + (tid % UPK));                                   \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                  \
            const int u = tid + i * NTHR;                                                                 \
            bool ok;                                                                                      \
            const bf16_t* p;                                                                              \
            if constexpr (B_TR) p = bl.ptr_tr(seg_, kt_ * BKT + u / UPB, bcolT[i], ok);                   \
            else p = bl.at(seg_, brow[i], bcol_, ok);                                                     \
            if (UB % NTHR) ok = ok && u < UB;                                                             \
            uint4 v = *reinterpret_cast<const uint4*>(p);                                                 \
            rb[i] = ok ? v : make_uint4(0, 0, 0, 0);                                                      \
        }                                                                                                 \
    }
#define GEMM_LSTORE(BUF)                                                                                 \
    {                                                                                                     \
        char* ta_ = smem + (BUF)*BUF_BYTES;                                                               \
        char* tb_ = ta_ + TA_BYTES;                                                                       \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                  \
            const int u = tid + i * NTHR;                                                                 \
            int oa;                                                                                       \
            if constexpr (A_TR) oa = lds_trw<UPA_L>(u / UPA, u % UPA);                                    \
            else oa = BKT == 64 ? lds_kc(u >> 3, u & 7) : kc32_off(u >> 2, u & 3);                        \
            if ((UA % NTHR) == 0 || u < UA) *reinterpret_cast<uint4*>(ta_ + oa) = ra[i];                  \
        }                                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                  \
            const int u = tid + i * NTHR;                                                                 \
            int ob;                                                                                       \
            if constexpr (B_TR) ob = lds_trw<UPB_L>(u / UPB, u % UPB);                                    \
            else ob = BKT == 64 ? lds_kc(u >> 3, u & 7) : kc32_off(u >> 2, u & 3);                        \
            if ((UB % NTHR) == 0 || u < UB) *reinterpret_cast<uint4*>(tb_ + ob) = rb[i];                  \
        }                                                                                                 \
    }

    GEMM_GLOAD(0);
    GEMM_LSTORE(0);
    __syncthreads();
    for (int it = 0; it < total; ++it) {
        const int cur = it & 1;
        if (it + 1 < total) GEMM_GLOAD(it + 1);
        const char* ta = smem + cur * BUF_BYTES;
        const char* tb = ta + TA_BYTES;
#pragma unroll
        for (int s = 0; s < BKT / 32; ++s) {
            bf16x8_t af[MT], bf[NT];
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if constexpr (!A_TR && BKT == 32)
                    af[t] = *reinterpret_cast<const bf16x8_t*>(ta + kc32_off(wm * (MT * 16) + t * 16 + (lane & 15), lane >> 4));
                else af[t] = read_frag<A_TR, UPA_L>(ta, wm * (MT * 16) + t * 16, s, lane);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if constexpr (!B_TR && BKT == 32)
                    bf[t] = *reinterpret_cast<const bf16x8_t*>(tb + kc32_off(wn * (NT * 16) + t * 16 + (lane & 15), lane >> 4));
                else bf[t] = read_frag<B_TR, UPB_L>(tb, wn * (NT * 16) + t * 16, s, lane);
            }
#pragma unroll
            for (int tn = 0; tn < NT; ++tn)
#pragma unroll
                for (int tm = 0; tm < MT; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
        }
        if (it + 1 < total) GEMM_LSTORE(cur ^ 1);
        __syncthreads();
    }
#undef GEMM_GLOAD
#undef GEMM_LSTORE

    if constexpr (EP::kStagedAtomic) {
        // wgrad: stage the fp32 tile in LDS (16-byte slots XOR-swizzled by row) and issue the global atomics
        // row-contiguously: every wave-instruction adds 256 contiguous bytes (the full-rate atomic shape).
        float* st = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int tm = 0; tm < MT; ++tm) {
            const int ml = wm * (MT * 16) + tm * 16 + (lane & 15);
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) {
                const int c4 = (wn * (NT * 16) + tn * 16 + 4 * (lane >> 4)) >> 2;
                *reinterpret_cast<f32x4*>(st + ml * 128 + ((c4 ^ (ml & 7)) << 2)) = acc[tn][tm];
            }
        }
        __syncthreads();
        for (int rr = wave; rr < BME; rr += NTHR / 64) {
            const int m = bm * BME + rr;
            if (m >= M) break;
            for (int nl = lane; nl < BNE; nl += 64) {
                const int n = bn * BNE + nl;
                if (n < N) ep.add(m, n, st[rr * 128 + ((((nl >> 2) ^ (rr & 7)) << 2) | (nl & 3))]);
            }
        }
        return;
    }
    // epilogue: lane owns C[m][n..n+3]
#pragma unroll
    for (int tm = 0; tm < MT; ++tm) {
        int m = bm * BME + wm * (MT * 16) + tm * 16 + (lane & 15);
        if (m >= M) continue;
#pragma unroll
        for (int tn = 0; tn < NT; ++tn) {
            int n = bn * BNE + wn * (NT * 16) + tn * 16 + 4 * (lane >> 4);
            if (n < N) ep.store(m, n, acc[tn][tm]);
        }
    }
}

// ==============================================================================================
// v2: 256x128 tile, 8 waves (4x2, 64x64 each), LDS-DMA (global_load_lds_dwordx4) into a 3-stage LDS ring with
// counted vmcnt + raw s_barrier: two K-steps (96 KiB) stay in flight per CU across the barrier, no VGPR staging and
// no ds_write pass.  The LDS image is lane-linear per wave-instruction (1 KiB = 8 rows x 128 B, or 4 k-rows x
// 256 B for TR operands); the XOR swizzle is applied on the SOURCE chunk, the fragment reads use the same
// involution.  Invalid units read a 16-byte zero page.  The LDS-DMA is issued from inline asm so that hipcc does
// not drain it with vmcnt(0) at the next ds_read / barrier (cdna_hip_programming.md 5, "Pipelining across barriers").
// ==============================================================================================
constexpr int BM2 = 256, NTHR2 = 512;
typedef __attribute__((address_space(3))) char* lds_char_ptr;

// Geometry of one ring stage for K-step BKT (64: 48 KiB/stage, 1 workgroup/CU -- used by the LDS-staged wgrad epilogue;
// 32: 24 KiB/stage, 72 KiB ring => 2 workgroups (16 waves) per CU so one workgroup's barrier / DMA waits are covered by
// the other's MFMAs)
template <int BKT>
struct G2 {
    static constexpr int UPR = BKT / 8;                // 16-byte units per K-contiguous row
    static constexpr int RPI = 64 / UPR;               // rows per 1-KiB wave-instruction (K-contiguous image)
    static constexpr int A_BYTES = BM2 * BKT * 2;
    static constexpr int B_BYTES = BN * BKT * 2;
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int NA = A_BYTES / 1024, NB = B_BYTES / 1024;  // wave-instructions per stage
    static constexpr int PER_WAVE = (NA + NB) / 8;
    static constexpr int SMEM = 3 * STAGE;
    static constexpr int TRH = BKT * 256;              // bytes of one 128-column half of a TR image
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);  // wave-uniform by construction; keeps the "s" operand in an SGPR
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ int tr_key(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }
// K-contiguous image swizzle: physical 16-byte chunk of logical chunk c in row r (an involution in c)
template <int BKT>
__device__ __forceinline__ int kc_swz(int r, int c) {
    if constexpr (BKT == 64) return c ^ (r & 7);
    else return c ^ ((0x1230 >> (4 * ((r >> 2) & 3))) & 3);  // f = {0,3,2,1}: conflict-free for ds_read_b128 on 64-byte rows
}
template <int BKT, bool TR>
__device__ __forceinline__ bf16x8_t read_frag2(const char* tile, int row0, int s, int lane) {
    if constexpr (!TR) {
        const int r = row0 + (lane & 15);
        const int c = s * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8_t*>(tile + r * (BKT * 2) + (kc_swz<BKT>(r, c) << 4));
    } else {
        return read_frag<true>(tile, row0, s, lane);
    }
}

// DUALK = 2 (staged-atomic weight gradients only): ONE 16-wave workgroup per CU instead of two 8-wave ones.  Its two wave
// groups run this kernel's body on their own rings over the two halves of the workgroup's K slice (they meet at the same
// barriers, so both execute the same number of K-steps; the shorter half idles through its last one) and fold their two fp32
// tiles through LDS before the atomic pass: the split-K atomic volume (#workgroups x tile) is halved -- with the adds
// switched off the weight gradient ran at 890-990 TFLOP/s against 675-785.
template <class AL, class BL, class EP, bool A_TR, bool B_TR, int NSEG, int BKT, int DUALK = 1>
__global__ __launch_bounds__(NTHR2 * DUALK, (BKT == 32 ? 4 : 2)) void gemm2_kernel(AL al, BL bl, EP ep, int M, int N, int K, int tiles_n,
                                                                               int ntiles, int kchunk, const bf16_t* zero_page) {
    using G = G2<BKT>;
    static_assert(DUALK == 1 || (DUALK == 2 && EP::kStagedAtomic && BKT == 32), "the dual-group form exists for the staged-atomic epilogue");
    extern __shared__ __attribute__((aligned(16))) char smem_all[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave16 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = DUALK == 2 ? wave16 >> 3 : 0;
    const int wave = wave16 & 7;
    char* smem = smem_all + grp * G::SMEM;  // this wave group's ring
    const int wm = wave >> 1, wn = wave & 1;
    // PERSISTENT tile schedule: workgroups are dealt round-robin over the 8 XCDs (private L2s; XCD = linear workgroup
    // id % 8, tools/xcc_probe.hip); XCD x owns the contiguous tile range [tlo, tlo+tcnt) and its workgroups walk it with
    // stride nbx, so the CUs of one XCD always work on neighbouring tiles (shared A row panel / B panels).  With
    // gridDim.x == ntiles this degenerates to one tile per workgroup.
    const int nb = gridDim.x, xcd = blockIdx.x & 7;
    int jx = blockIdx.x >> 3;
    int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    int ysplit = blockIdx.y;
    if constexpr (EP::kStagedAtomic) {
        // Split-K weight gradients, one (split, tile) pair per workgroup on a 1-D grid: the pairs are ordered split-major and every
        // XCD takes a contiguous run of them, so an XCD works on ONE (at most two) token ranges and a contiguous tile range
        // inside it.  Its ~27 concurrent workgroups then stream the SAME rows of dy and x (L2 hits), and each operand slab of a
        // token range is fetched by the one or two XCDs that own it instead of by all eight (dealing the tiles of every split over
        // all XCDs made every XCD read the whole of x: 412 MB per launch against 160 MB of algorithmic bytes).
        if (ep.pairs > 0) {
            const int P = ntiles * ep.pairs, qP = P >> 3, rP = P & 7;
            const int plo = xcd * qP + min(xcd, rP), pcnt = qP + (xcd < rP ? 1 : 0);
            if (jx >= pcnt) return;
            const int pr = plo + jx;
            ysplit = pr / ntiles;
            tlo = pr - ysplit * ntiles, tcnt = 1, jx = 0, nbx = 1;
            ep.split = ysplit;
        }
    }
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    al.init(blockIdx.z);
    bl.init(blockIdx.z);
    ep.init(blockIdx.z);
    if (al.kdim() >= 0) K = al.kdim();
    const int nk_all = (K + BKT - 1) / BKT;
    int kt0 = ysplit * kchunk;
    int nk = min(kchunk, nk_all - kt0);
    if (nk <= 0 || my_tiles <= 0) return;
    int G_loop = 0;  // DUALK == 2: iterations (barriers) both wave groups execute
    if constexpr (DUALK == 2) {
        const int nkh = (nk + 1) >> 1;  // group 0: [kt0, kt0 + nkh), group 1: the rest (possibly one step shorter, or empty)
        G_loop = nkh * NSEG;
        kt0 += grp * nkh;
        nk = grp == 0 ? nkh : nk - nkh;
    }
    const int total = nk * NSEG;      // K-steps per tile
    const int G_ = my_tiles * total;  // flattened (tile, K-step) sequence of this workgroup (wave group)
    if constexpr (DUALK == 1) G_loop = G_;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // one stage = NA + NB wave-instructions ("pieces") of 1 KiB (ids 0..NA-1 -> A, then B); wave w issues ids i*8 + w.
    // The source pointer of a piece advances by a constant per K-step, so it is decoded once per (tile, segment) and
    // then stepped with one 64-bit add: recomputing row*ld + bounds for every piece and K-step cost more issue cycles
    // than the 16 MFMAs of the step (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost').  Needs K % BKT == 0 and a
    // loader whose address is linear in k (kLinearK); otherwise the generic per-step decode below is used.
#define GEMM2_PIECE_PTR(SEG, KT, BM_, BN_, ID, P, OK)                                                       \
    {                                                                                                       \
        if ((ID) < G::NA) {                                                                                 \
            if constexpr (!A_TR) {                                                                          \
                const int row = (ID)*G::RPI + lane / G::UPR;                                                \
                P = al.ptr(SEG, (BM_)*BM2 + row, (KT)*G::UPR + kc_swz<BKT>(row, lane % G::UPR), OK);        \
            } else {                                                                                        \
                const int k = ((ID) % (BKT / 4)) * 4 + (lane >> 4);                                         \
                P = al.ptr(SEG, (KT)*BKT + k, (BM_)*32 + ((ID) / (BKT / 4)) * 16 + ((lane & 15) ^ (tr_key(k) << 1)), OK); \
            }                                                                                               \
        } else {                                                                                            \
            const int id2 = (ID)-G::NA;                                                                     \
            if constexpr (!B_TR) {                                                                          \
                const int row = id2 * G::RPI + lane / G::UPR;                                               \
                P = bl.ptr(SEG, (BN_)*BN + row, (KT)*G::UPR + kc_swz<BKT>(row, lane % G::UPR), OK);         \
            } else {                                                                                        \
                const int k = id2 * 4 + (lane >> 4);                                                        \
                P = bl.ptr(SEG, (KT)*BKT + k, (BN_)*16 + ((lane & 15) ^ (tr_key(k) << 1)), OK);             \
            }                                                                                               \
        }                                                                                                   \
    }
    constexpr bool kFast = AL::kLinearK && BL::kLinearK;  // a K tail (K % BKT != 0) takes the generic decode for that step only
    const char* pp[G::PER_WAVE];   // fast path: current source of this wave's pieces
    long pstep[G::PER_WAVE];       // bytes per K-step (0 for out-of-range rows, which read the zero page)
#define GEMM2_SETUP(IT, BM_, BN_)                                                                           \
    {                                                                                                       \
        const int seg_ = (NSEG == 1) ? 0 : (IT) / nk;                                                       \
        const int kt_ = kt0 + (IT)-seg_ * nk;                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PER_WAVE; ++i) {                                           \
            const int id = i * 8 + wave;                                                                    \
            bool ok;                                                                                        \
            const bf16_t* p;                                                                                \
            GEMM2_PIECE_PTR(seg_, kt_, BM_, BN_, id, p, ok);                                                \
            pp[i] = ok ? (const char*)p : (const char*)zero_page;                                           \
            const bool tr_ = id < G::NA ? A_TR : B_TR;                                                      \
            const long ld_ = id < G::NA ? al.kstride() : bl.kstride();                                      \
            pstep[i] = !ok ? 0L : tr_ ? (long)BKT * ld_ * 2 : (long)BKT * 2;                                \
        }                                                                                                   \
    }
    // Issue of one K-step's pieces: BEGIN decodes the step (and re-seeds the piece pointers at a tile/segment start),
    // ONE(i) issues piece i.  In the main loop the pieces are spread BETWEEN the MFMA groups of the step: issued in one
    // burst right after the barrier, every wave of the SIMD sits in its ~100-cycle DMA issues at the same time and the
    // matrix pipe idles; spread out, one wave's DMA issue overlaps the other waves' MFMAs.
#define GEMM2_ISSUE_BEGIN(IT, SLOT, BM_, BN_)                                                              \
    const unsigned sbase_ = lds_base + (SLOT)*G::STAGE;                                                     \
    const int seg_ = (NSEG == 1) ? 0 : (IT) / nk;                                                           \
    const int kt_ = kt0 + (IT)-seg_ * nk;                                                                   \
    const int ibm_ = (BM_), ibn_ = (BN_);                                                                   \
    const bool fast_ = kFast && (kt_ + 1) * BKT <= K;                                                       \
    if (fast_ && (IT) == seg_ * nk) GEMM2_SETUP(IT, BM_, BN_);
#define GEMM2_ISSUE_ONE(I)                                                                                  \
    {                                                                                                       \
        if (fast_) {                                                                                        \
            glds16(pp[I], sbase_ + ((I)*8 + wave) * 1024);                                                  \
            pp[I] += pstep[I];                                                                              \
        } else {                                                                                            \
            const int id = (I)*8 + wave;                                                                    \
            bool ok;                                                                                        \
            const bf16_t* p;                                                                                \
            GEMM2_PIECE_PTR(seg_, kt_, ibm_, ibn_, id, p, ok);                                              \
            glds16(ok ? p : zero_page, sbase_ + id * 1024);                                                 \
        }                                                                                                   \
    }
#define GEMM2_ISSUE(IT, SLOT, BM_, BN_)                                                                    \
    {                                                                                                       \
        GEMM2_ISSUE_BEGIN(IT, SLOT, BM_, BN_)                                                               \
        _Pragma("unroll") for (int i = 0; i < G::PER_WAVE; ++i) GEMM2_ISSUE_ONE(i)                          \
    }

    // issue-side cursor (runs two K-steps ahead of the compute-side cursor, across tile boundaries)
    int it_i = 0;
    int tile_i = tlo + jx;
    int bm_i = tile_i / tiles_n, bn_i = tile_i - bm_i * tiles_n;
#define GEMM2_ADVANCE_ISSUE()                                  \
    {                                                           \
        if (++it_i == total) {                                  \
            it_i = 0;                                           \
            tile_i += nbx;                                      \
            bm_i = tile_i / tiles_n, bn_i = tile_i - bm_i * tiles_n; \
        }                                                       \
    }
    if (G_ > 0) {
        GEMM2_ISSUE(it_i, 0, bm_i, bn_i);
        GEMM2_ADVANCE_ISSUE();
    }
    if (G_ > 1) {
        GEMM2_ISSUE(it_i, 1, bm_i, bn_i);
        GEMM2_ADVANCE_ISSUE();
    }
    int slot = 0, it_c = 0;
    int tile_c = tlo + jx;
    for (int g = 0; g < G_loop; ++g) {
        // K-step g landed for THIS wave once at most the DMAs of step g+1 are outstanding (vmcnt also counts the
        // epilogue's stores, which only makes the wait at a tile boundary conservative); the barrier then covers
        // the other waves' pieces (RAW) and everybody's reads of the slot that is refilled next (WAR)
        if (g + 1 < G_) {
            if constexpr (G::PER_WAVE == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if constexpr (DUALK == 2) {
            if (g >= G_) continue;  // the shorter wave group only keeps the barrier count (its last iteration, or all of them)
        }
        const bool do_issue = g + 2 < G_;
        const int s2 = slot >= 1 ? slot - 1 : slot + 2;  // (slot + 2) % 3
        GEMM2_ISSUE_BEGIN(do_issue ? it_i : 0, s2, bm_i, bn_i)
        // spreading pays for the K-contiguous operands (forward GEMMs); with transposed operands the scheduling fences
        // cost more than the overlap gains (measured), so those issue the whole step right after the barrier
        constexpr bool kSpread = !A_TR && !B_TR;
        if constexpr (!kSpread) {
            if (do_issue) {
#pragma unroll
                for (int i = 0; i < G::PER_WAVE; ++i) GEMM2_ISSUE_ONE(i)
            }
        }
        const char* ta = smem + slot * G::STAGE;
        const char* tb = ta + G::A_BYTES;
#pragma unroll
        for (int s = 0; s < BKT / 32; ++s) {
            bf16x8_t af[4], bf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (!A_TR) af[t] = read_frag2<BKT, false>(ta, wm * 64 + t * 16, s, lane);
                else af[t] = read_frag2<BKT, true>(ta + (wm >> 1) * G::TRH, (wm & 1) * 64 + t * 16, s, lane);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) bf[t] = read_frag2<BKT, B_TR>(tb, wn * 64 + t * 16, s, lane);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tn][tm], 0, 0, 0);
                if (kSpread && s * 4 + tn < G::PER_WAVE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (do_issue) GEMM2_ISSUE_ONE(s * 4 + tn)
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (do_issue) GEMM2_ADVANCE_ISSUE();
        slot = slot == 2 ? 0 : slot + 1;
        if constexpr (DUALK == 2) continue;  // one tile per workgroup: the epilogue follows the loop, entered by both groups together
        if (++it_c < total) continue;
        // ---- tile finished: epilogue (next tile's first K-steps are already in flight) ----
        it_c = 0;
        const int bm = tile_c / tiles_n, bn = tile_c - bm * tiles_n;
        tile_c += nbx;
        if constexpr (EP::kStagedAtomic) {
            // launched with one tile per workgroup: the LDS ring is idle here.  The 256 x 128 fp32 tile is staged in two
            // 128-row halves (64 KiB each) so that the 72 KiB BK=32 ring (2 workgroups/CU) is enough.
#pragma unroll
            for (int hpass = 0; hpass < 2; ++hpass) {
                __syncthreads();
                float* st = reinterpret_cast<float*>(smem);
                if ((wm >> 1) == hpass) {
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) {
                        const int ml = (wm & 1) * 64 + tm * 16 + (lane & 15);
#pragma unroll
                        for (int tn = 0; tn < 4; ++tn) {
                            const int c4 = (wn * 64 + tn * 16 + 4 * (lane >> 4)) >> 2;
                            *reinterpret_cast<f32x4*>(st + ml * 128 + ((c4 ^ (ml & 7)) << 2)) = acc[tn][tm];
                        }
                    }
                }
                __syncthreads();
                for (int rr = wave; rr < 128; rr += NTHR2 / 64) {
                    const int m = bm * BM2 + hpass * 128 + rr;
                    if (m >= M) break;
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        const int nl = half * 64 + lane;
                        const int n = bn * BN + nl;
                        if (n < N) ep.add(m, n, st[rr * 128 + ((((nl >> 2) ^ (rr & 7)) << 2) | (nl & 3))]);
                    }
                }
            }
            return;
        } else {
            f32x4 cs[4];  // per-lane partial column sums (4 consecutive n per n-tile) over this wave's 64 rows
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) cs[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) {
                int m = bm * BM2 + wm * 64 + tm * 16 + (lane & 15);
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    int n = bn * BN + wn * 64 + tn * 16 + 4 * (lane >> 4);
                    if (m < M && n < N) {
                        if constexpr (EP::kColSum) cs[tn] += ep.store_ret(m, n, acc[tn][tm]);
                        else ep.store(m, n, acc[tn][tm]);
                    }
                    acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            if constexpr (EP::kColSum) {
                if (ep.colsum) {  // wave-uniform: reduce over the 16 lanes that share (lane >> 4), one atomic per column
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        const int n = bn * BN + wn * 64 + tn * 16 + 4 * (lane >> 4);
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            float v = cs[tn][c];
                            v += __shfl_xor(v, 1, 64);
                            v += __shfl_xor(v, 2, 64);
                            v += __shfl_xor(v, 4, 64);
                            v += __shfl_xor(v, 8, 64);
                            if ((lane & 15) == 0 && n + c < N) ig_red_add(ep.colsum + n + c, v);
                        }
                    }
                }
            }
            asm volatile("" ::: "memory");
        }
    }
    if constexpr (DUALK == 2) {
        // both wave groups hold a partial 256 x 128 fp32 tile: stage 128 rows of each (2 x 64 KiB of the 144 KiB of rings), add
        // the pair and issue ONE row-contiguous atomic pass
        const int bm = tile_c / tiles_n, bn = tile_c - bm * tiles_n;
#pragma unroll
        for (int hpass = 0; hpass < 2; ++hpass) {
            __syncthreads();
            float* st = reinterpret_cast<float*>(smem_all) + grp * (128 * 128);
            if ((wm >> 1) == hpass) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) {
                    const int ml = (wm & 1) * 64 + tm * 16 + (lane & 15);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        const int c4 = (wn * 64 + tn * 16 + 4 * (lane >> 4)) >> 2;
                        *reinterpret_cast<f32x4*>(st + ml * 128 + ((c4 ^ (ml & 7)) << 2)) = acc[tn][tm];
                    }
                }
            }
            __syncthreads();
            const float* s0 = reinterpret_cast<const float*>(smem_all);
            const float* s1 = s0 + 128 * 128;
            for (int rr = wave16; rr < 128; rr += 16) {
                const int m = bm * BM2 + hpass * 128 + rr;
                if (m >= M) break;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int nl = half * 64 + lane;
                    const int n = bn * BN + nl;
                    const int o = rr * 128 + ((((nl >> 2) ^ (rr & 7)) << 2) | (nl & 3));
                    if (n < N) ep.add(m, n, s0[o] + s1[o]);
                }
            }
        }
    }
#undef GEMM2_ISSUE
#undef GEMM2_ISSUE_ONE
#undef GEMM2_ISSUE_BEGIN
#undef GEMM2_SETUP
#undef GEMM2_PIECE_PTR
#undef GEMM2_ADVANCE_ISSUE
}

// ==============================================================================================
// v5: "ping-pong" 256x256 tile.  8 waves = 2 row groups (wr) x 4 column waves (wc); a wave owns 128 x 64
// (acc[4][8], 128 accumulator registers, one workgroup per CU => 256 registers per wave).  BK = 32, LDS-DMA into a
// 4-stage ring (4 x 32 KiB), two K-steps in flight.  Every K-step is two PHASES (the two 64-row halves of the wave's
// rows); a phase = [fragment ds_reads + 2 DMA pieces] s_barrier [16 MFMAs under s_setprio 1] s_barrier.  The two row
// groups run ONE barrier apart (group 1 executes an extra barrier first): between two barriers one group issues
// memory instructions while the other group's waves -- one per SIMD -- own the matrix pipe, so the ~100-cycle DMA
// issues and the fragment-read latency of one group are hidden under the MFMAs of the other (the structure of the
// CDNA guide's 8-phase template; here with BK = 32 phases and the loaders / epilogues of v2).
//   RAW: a wave retires its own pieces of step g+1 with vmcnt(4) in phase (g,1) before that phase's first barrier;
//        step g+1 is first read two barriers later.      WAR: slot (g+2)%4 was last read during step g-2.
// ==============================================================================================
constexpr int NTHR5 = 512;
constexpr int G5_OP = 256 * 32 * 2;     // one operand image: 16 KiB
constexpr int G5_STAGE = 2 * G5_OP;     // A + B
constexpr int G5_SMEM = 4 * G5_STAGE;   // 128 KiB

template <class AL, class BL, class EP, bool A_TR, bool B_TR, int NSEG>
__global__ __launch_bounds__(NTHR5, 2) void gemm5_kernel(AL al, BL bl, EP ep, int M, int N, int K, int tiles_n, int ntiles,
                                                         int kchunk, const bf16_t* zero_page) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    al.init(blockIdx.z);
    bl.init(blockIdx.z);
    ep.init(blockIdx.z);
    const int nk_all = (K + 31) / 32;  // a K tail takes the generic per-piece decode for that step
    const int kt0 = blockIdx.y * kchunk;
    const int nk = min(kchunk, nk_all - kt0);
    if (nk <= 0 || my_tiles <= 0) return;
    const int total = nk * NSEG;
    const int G_ = my_tiles * total;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // 32 pieces of 1 KiB per stage: ids 0..15 -> A, 16..31 -> B; wave w owns ids i*8 + w (i = 0, 1: A; 2, 3: B)
    const char* pp[4];
    long pstep[4];
    int kpos_i = 0, seg_i = 0, tile_i = tlo + jx;  // issue cursor: K-step inside the segment, segment, tile
    int bm_i = tile_i / tiles_n, bn_i = tile_i - bm_i * tiles_n;
    const bool has_tail = (K & 31) != 0 && kt0 + nk == nk_all;  // this workgroup's last K-step of a segment is partial
    int kb[4];  // this lane's k offset inside a K-step for each piece (K-tail predicate)
#define GEMM5_SETUP(SEG, KT)                                                                                \
    {                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
            const int id = (i & 1) * 8 + wave;                                                              \
            bool ok;                                                                                        \
            const bf16_t* p;                                                                                \
            if (i < 2) {                                                                                    \
                if constexpr (!A_TR) {                                                                      \
                    const int row = id * 16 + (lane >> 2), ch = kc_swz<32>(row, lane & 3);                  \
                    p = al.ptr(SEG, bm_i * 256 + row, (KT)*4 + ch, ok);                                     \
                    ok = bm_i * 256 + row < M;                                                              \
                    kb[i] = ch * 8;                                                                         \
                } else {                                                                                    \
                    const int k = (id & 7) * 4 + (lane >> 4), cu = bm_i * 32 + (id >> 3) * 16 + ((lane & 15) ^ (tr_key(k) << 1)); \
                    p = al.ptr(SEG, (KT)*32 + k, cu, ok);                                                   \
                    ok = cu * 8 < M;                                                                        \
                    kb[i] = k;                                                                              \
                }                                                                                           \
                pstep[i] = !ok ? 0L : A_TR ? 64L * al.kstride() : 64L;                                      \
            } else {                                                                                        \
                if constexpr (!B_TR) {                                                                      \
                    const int row = id * 16 + (lane >> 2), ch = kc_swz<32>(row, lane & 3);                  \
                    p = bl.ptr(SEG, bn_i * 256 + row, (KT)*4 + ch, ok);                                     \
                    ok = bn_i * 256 + row < N;                                                              \
                    kb[i] = ch * 8;                                                                         \
                } else {                                                                                    \
                    const int k = (id & 7) * 4 + (lane >> 4), cu = bn_i * 32 + (id >> 3) * 16 + ((lane & 15) ^ (tr_key(k) << 1)); \
                    p = bl.ptr(SEG, (KT)*32 + k, cu, ok);                                                   \
                    ok = cu * 8 < N;                                                                        \
                    kb[i] = k;                                                                              \
                }                                                                                           \
                pstep[i] = !ok ? 0L : B_TR ? 64L * bl.kstride() : 64L;                                      \
            }                                                                                               \
            pp[i] = ok ? (const char*)p : (const char*)zero_page;                                           \
        }                                                                                                   \
    }
    // pieces 2h, 2h+1 of the stage for the issue cursor's K-step go out in phase h.  The piece pointers are stepped for
    // every K-step; only the (wave-uniform, rare) last partial step masks the lanes whose k index is beyond K.
#define GEMM5_ISSUE_HALF(SLOT, H)                                                                           \
    {                                                                                                       \
        const unsigned sb_ = lds_base + (SLOT)*G5_STAGE + ((H) ? G5_OP : 0);                                \
        if ((H) == 0 && kpos_i == 0) GEMM5_SETUP(seg_i, kt0);                                               \
        const char *q0 = pp[2 * (H)], *q1 = pp[2 * (H) + 1];                                                \
        if (has_tail && kpos_i == nk - 1) {                                                                 \
            const int k0_ = (kt0 + kpos_i) * 32;                                                            \
            q0 = (k0_ + kb[2 * (H)] < K) ? q0 : (const char*)zero_page;                                     \
            q1 = (k0_ + kb[2 * (H) + 1] < K) ? q1 : (const char*)zero_page;                                 \
        }                                                                                                   \
        glds16(q0, sb_ + (0 * 8 + wave) * 1024);                                                            \
        glds16(q1, sb_ + (1 * 8 + wave) * 1024);                                                            \
        pp[2 * (H)] += pstep[2 * (H)], pp[2 * (H) + 1] += pstep[2 * (H) + 1];                               \
        if ((H) == 1 && ++kpos_i == nk) {                                                                   \
            kpos_i = 0;                                                                                     \
            if (NSEG == 1 || ++seg_i == NSEG) {                                                             \
                seg_i = 0;                                                                                  \
                tile_i += nbx;                                                                              \
                bm_i = tile_i / tiles_n, bn_i = tile_i - bm_i * tiles_n;                                    \
            }                                                                                               \
        }                                                                                                   \
    }
    // prologue: two stages in flight, the first one landed for everybody
    GEMM5_ISSUE_HALF(0, 0);
    GEMM5_ISSUE_HALF(0, 1);
    if (G_ > 1) {
        GEMM5_ISSUE_HALF(1, 0);
        GEMM5_ISSUE_HALF(1, 1);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
    if (wr == 1) asm volatile("s_barrier" ::: "memory");  // stagger: group 1 runs one barrier behind group 0

    int it_c = 0, tile_c = tlo + jx;
    bf16x8_t bf[4];
    for (int g = 0; g < G_; ++g) {
        const int slot = g & 3;
        const char* ta = smem + slot * G5_STAGE;
        const char* tb = ta + G5_OP;
        const bool do_issue = g + 2 < G_;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8_t af[4];
            if (h == 0) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    if constexpr (!B_TR) bf[t] = read_frag2<32, false>(tb, wc * 64 + t * 16, 0, lane);
                    else bf[t] = read_frag2<32, true>(tb + (wc >> 1) * 8192, (wc & 1) * 64 + t * 16, 0, lane);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (!A_TR) af[t] = read_frag2<32, false>(ta, wr * 128 + h * 64 + t * 16, 0, lane);
                else af[t] = read_frag2<32, true>(ta + wr * 8192, h * 64 + t * 16, 0, lane);
            }
            if (do_issue) GEMM5_ISSUE_HALF((g + 2) & 3, h);
            if (h == 1 && g + 1 < G_) {
                if (do_issue) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
                    acc[tn][h * 4 + tm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[tn], af[tm], acc[tn][h * 4 + tm], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            asm volatile("s_barrier" ::: "memory");
        }
        if (++it_c < total) continue;
        it_c = 0;
        if constexpr (EP::kStagedAtomic) break;  // one tile per workgroup: staged epilogue after the loop
        const int bm = tile_c / tiles_n, bn = tile_c - bm * tiles_n;
        tile_c += nbx;
        f32x4 cs[4];
#pragma unroll
        for (int tn = 0; tn < 4; ++tn) cs[tn] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tm = 0; tm < 8; ++tm) {
            const int m = bm * 256 + wr * 128 + tm * 16 + (lane & 15);
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
                const int n = bn * 256 + wc * 64 + tn * 16 + 4 * (lane >> 4);
                if (m < M && n < N) {
                    if constexpr (EP::kColSum) cs[tn] += ep.store_ret(m, n, acc[tn][tm]);
                    else ep.store(m, n, acc[tn][tm]);
                }
                acc[tn][tm] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            asm volatile("" ::: "memory");
        }
        if constexpr (EP::kColSum) {
            if (ep.colsum) {
#pragma unroll
                for (int tn = 0; tn < 4; ++tn) {
                    const int n = bn * 256 + wc * 64 + tn * 16 + 4 * (lane >> 4);
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float v = cs[tn][c];
                        v += __shfl_xor(v, 1, 64);
                        v += __shfl_xor(v, 2, 64);
                        v += __shfl_xor(v, 4, 64);
                        v += __shfl_xor(v, 8, 64);
                        if ((lane & 15) == 0 && n + c < N) ig_red_add(ep.colsum + n + c, v);
                    }
                }
            }
        }
    }
    if (wr == 0) asm volatile("s_barrier" ::: "memory");  // balance group 1's extra barrier: the groups are aligned again
    if constexpr (EP::kStagedAtomic) {
        // weight gradient (one tile per workgroup, split-K over blockIdx.y): the fp32 tile goes through the idle ring in two
        // 128-row passes (128 KiB each) and is added with row-contiguous 256-byte atomic wave-instructions
        const int bm = tile_c / tiles_n, bn = tile_c - bm * tiles_n;
        float* st = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            __syncthreads();
            if (wr == pass) {
#pragma unroll
                for (int tm = 0; tm < 8; ++tm) {
                    const int ml = tm * 16 + (lane & 15);
#pragma unroll
                    for (int tn = 0; tn < 4; ++tn) {
                        const int c4 = (wc * 64 + tn * 16 + 4 * (lane >> 4)) >> 2;
                        *reinterpret_cast<f32x4*>(st + ml * 256 + ((c4 ^ (ml & 7)) << 2)) = acc[tn][tm];
                    }
                }
            }
            __syncthreads();
            for (int rr = wave; rr < 128; rr += NTHR5 / 64) {
                const int m = bm * 256 + pass * 128 + rr;
                if (m >= M) break;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int nl = q * 64 + lane;
                    const int n = bn * 256 + nl;
                    if (n < N) ep.add(m, n, st[rr * 256 + ((((nl >> 2) ^ (rr & 7)) << 2) | (nl & 3))]);
                }
            }
        }
    }
#undef GEMM5_ISSUE_HALF
#undef GEMM5_SETUP
}

// FDiv / make_fdiv (division by a launch-time constant as multiply-high + shift): common.h

// ------------------------------------------------------------------------------------ loaders
// segment base pointer by compare/select: a dynamically indexed base[seg] of a by-value loader that init() has modified
// lives in scratch memory (the split-precision ConvTranspose kernels carried 96-240 bytes of scratch per lane for it)
__device__ __forceinline__ const bf16_t* bsel(const bf16_t* const (&b)[3], int seg) { return seg == 0 ? b[0] : seg == 1 ? b[1] : b[2]; }
struct PlainLoader {
    static constexpr const char* kName = "PlainLoader";  // row-major [R][ld], logical width C (multiple of 8)
    static constexpr bool kLinearK = true;  // ptr() is affine in the K index (v2 steps piece pointers instead of re-decoding)
    const bf16_t* base[3];
    int R, C;
    long ld;
    __device__ long kstride() const { return ld; }  // elements between consecutive k-rows of a TR (k-major) operand
    __device__ void init(int) {}
    __device__ int kdim() const { return -1; }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const {
        int c = c8 * 8;
        ok = (r < R) & (c < C);
        return bsel(base, seg) + (ok ? (long)r * ld + c : 0L);
    }
    struct Row { long off; bool ok; };
    struct Col { int c; bool ok; };
    __device__ Row row(int r) const { return Row{(long)r * ld, r < R}; }
    __device__ Col col(int c8) const { return Col{c8 * 8, c8 * 8 < C}; }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const {
        ok = rw.ok & cl.ok;
        return bsel(base, seg) + (ok ? rw.off + cl.c : 0L);
    }
    // TR operand: k-row r, hoisted column decode
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const {
        ok = (r < R) & cl.ok;
        return bsel(base, seg) + (ok ? (long)r * ld + cl.c : 0L);
    }
};

// NHWC 3x3 pad-1 gather: row r = pixel (b,y,x), column unit -> (tap, channel).  sign=+1 reads
// (y+ky-1, x+kx-1) (conv forward / wgrad input side), sign=-1 reads (y+1-ky, x+1-kx) (dgrad).
struct Conv3Loader {
    static constexpr const char* kName = "Conv3Loader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Mtot, H, W, C, sign;
    FDiv f_hw, f_w, f_c;  // set by finish()
    void finish() { f_hw = make_fdiv(H * W), f_w = make_fdiv(W), f_c = make_fdiv(C); }
    __device__ void init(int) {}
    __device__ int kdim() const { return -1; }
    // split decode: the ROW (pixel) part is done once per tile -- element offset of the pixel and a 9-bit mask of the
    // taps that stay inside the image -- the COLUMN (tap, channel) part once per K-step; at() is then a shift, an and,
    // a 64-bit add and a select per 16-byte unit (it used to redo the bounds tests and a 64-bit multiply).
    struct Row { long off; unsigned mask; };
    struct Col { int tap, delta, dy, dx; bool ok; };
    __device__ void pixel(int r, int& y, int& x) const {
        const int b = f_hw.div(r), rem = r - b * (H * W);
        y = f_w.div(rem), x = rem - y * W;
    }
    __device__ Row row(int r) const {
        int y, x;
        pixel(r, y, x);
        unsigned mask = 0;
        if (r < Mtot) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bool v = ((unsigned)(y + sign * (ky - 1)) < (unsigned)H) & ((unsigned)(x + sign * (kx - 1)) < (unsigned)W);
                    mask |= (v ? 1u : 0u) << (ky * 3 + kx);
                }
        }
        return Row{(long)r * C, mask};
    }
    __device__ Col col(int c8) const {
        const int k = c8 * 8;
        const int tap = f_c.div(k), c = k - tap * C;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;  // tap / 3 for tap < 32
        const int dy = sign * (ky - 1), dx = sign * (kx - 1);
        return Col{tap, (dy * W + dx) * C + c, dy, dx, k < 9 * C};
    }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const {
        ok = cl.ok & ((rw.mask >> cl.tap) & 1u);
        return bsel(base, seg) + (ok ? rw.off + cl.delta : 0L);
    }
    // TR operand (weight gradient): k-row = pixel r (changes every K-step), column decode hoisted
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const {
        int y, x;
        pixel(r, y, x);
        ok = cl.ok & (r < Mtot) & ((unsigned)(y + cl.dy) < (unsigned)H) & ((unsigned)(x + cl.dx) < (unsigned)W);
        return bsel(base, seg) + (ok ? (long)r * C + cl.delta : 0L);
    }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const { return ptr_tr(seg, r, col(c8), ok); }
};

// NHWC KS x KS convolution with padding 1 (nn.Conv2d(kernel_size=KS, padding=1), model.py:370-375 with the 600M variants'
// kernel sizes 5 / 7, model.py:169-177): the output grid is (H + 3 - KS)^2, so rows and source live on DIFFERENT grids.
//   sign = +1: row = output pixel (b, yo, xo) on (Hr, Wr), reads x(yo + ky - 1, xo + kx - 1) on (Hs, Ws)   (forward, wgrad input)
//   sign = -1: row = input pixel (b, yi, xi) on (Hr, Wr), reads dy(yi + 1 - ky, xi + 1 - kx) on (Hs, Ws)   (dgrad)
// Column unit -> (tap = ky KS + kx, channel).  Validity is tested per unit from the row's (y, x) (no tap mask: 49 taps).
struct ConvKLoader {
    static constexpr const char* kName = "ConvKLoader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Mtot, Hr, Wr, Hs, Ws, C, KS, sign;
    FDiv f_hw, f_w, f_c, f_ks;
    void finish() { f_hw = make_fdiv(Hr * Wr), f_w = make_fdiv(Wr), f_c = make_fdiv(C), f_ks = make_fdiv(KS); }
    __device__ void init(int) {}
    __device__ int kdim() const { return -1; }
    struct Row { long off; int y, x; bool ok; };
    struct Col { int dy, dx, c; bool ok; };
    __device__ Row row(int r) const {
        const int b = f_hw.div(r), rem = r - b * (Hr * Wr);
        const int y = f_w.div(rem), x = rem - y * Wr;
        return Row{(long)b * Hs * Ws * C, y, x, r < Mtot};
    }
    __device__ Col col(int c8) const {
        const int k = c8 * 8;
        const int tap = f_c.div(k), c = k - tap * C;
        const int ky = f_ks.div(tap), kx = tap - ky * KS;
        return Col{sign * (ky - 1), sign * (kx - 1), c, k < KS * KS * C};
    }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const {
        const int ys = rw.y + cl.dy, xs = rw.x + cl.dx;
        ok = rw.ok & cl.ok & ((unsigned)ys < (unsigned)Hs) & ((unsigned)xs < (unsigned)Ws);
        return bsel(base, seg) + (ok ? rw.off + ((long)ys * Ws + xs) * C + cl.c : 0L);
    }
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const { return at(seg, row(r), cl, ok); }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const { return at(seg, row(r), col(c8), ok); }
};

// ConvTranspose2d(k3,s2,p1,op1) forward, sub-pixel phase z=(py,px): output (2iy+py, 2ix+px) reads taps
// ky in {1} (py=0) or {0,2} (py=1); tap ky==0 reads input row iy+1, otherwise iy (same for x).
struct ConvTFwdALoader {
    static constexpr const char* kName = "ConvTFwdALoader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Mtot, H, W, C;
    FDiv f_hw, f_w, f_c;
    void finish() { f_hw = make_fdiv(H * W), f_w = make_fdiv(W), f_c = make_fdiv(C); }
    int nky, nkx, ky0, kx0, K;
    __device__ void init(int z) {
        z = 3 - z;  // heaviest phase (4 taps) first: blockIdx.z is dispatched slowest, so the launch ends on the 1-tap phase
        int py = z >> 1, px = z & 1;
        nky = py ? 2 : 1, nkx = px ? 2 : 1;
        ky0 = py ? 0 : 1, kx0 = px ? 0 : 1;  // tap lists: {1} or {0,2}
        K = nky * nkx * C;
    }
    __device__ int kdim() const { return K; }
    struct Row { long off; unsigned mask; };  // mask bit (dy*2 + dx): input pixel (y+dy, x+dx) exists
    struct Col { int idx, delta; bool ok; };
    __device__ Row row(int r) const {
        const int b = f_hw.div(r), rem = r - b * (H * W);
        const int y = f_w.div(rem), x = rem - y * W;
        unsigned mask = 0;
        if (r < Mtot) mask = 1u | ((x + 1 < W) ? 2u : 0u) | ((y + 1 < H) ? 4u : 0u) | ((x + 1 < W && y + 1 < H) ? 8u : 0u);
        return Row{(long)r * C, mask};
    }
    __device__ Col col(int c8) const {
        const int k = c8 * 8;
        const int tl = f_c.div(k), c = k - tl * C;
        const int tyi = nkx == 2 ? (tl >> 1) : tl, txi = tl - tyi * nkx;
        const int dy = (ky0 + 2 * tyi) == 0, dx = (kx0 + 2 * txi) == 0;  // tap 0 reads the next input row / column
        return Col{dy * 2 + dx, (dy * W + dx) * C + c, k < K};
    }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const {
        ok = cl.ok & ((rw.mask >> cl.idx) & 1u);
        return bsel(base, seg) + (ok ? rw.off + cl.delta : 0L);
    }
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const { return at(seg, row(r), cl, ok); }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const { return at(seg, row(r), col(c8), ok); }
};
// matching weight view: n = co, k = (local tap, ci) of storage Wc[co][tap][ci]
struct ConvTFwdBLoader {
    static constexpr const char* kName = "ConvTFwdBLoader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Cout, C;
    FDiv f_c;  // set by finish(): multiply-high division by C (a runtime division per 16-byte unit sat in every K-step)
    void finish() { f_c = make_fdiv(C); }
    int nky, nkx, ky0, kx0, K;
    __device__ void init(int z) {
        z = 3 - z;  // heaviest phase (4 taps) first: blockIdx.z is dispatched slowest, so the launch ends on the 1-tap phase
        int py = z >> 1, px = z & 1;
        nky = py ? 2 : 1, nkx = px ? 2 : 1;
        ky0 = py ? 0 : 1, kx0 = px ? 0 : 1;
        K = nky * nkx * C;
    }
    __device__ int kdim() const { return K; }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const {
        int k = c8 * 8;
        int tl = f_c.div(k), c = k - tl * C;
        int tyi = tl >> (nkx - 1), txi = tl - tyi * nkx;  // nkx is 1 or 2
        int tap = (ky0 + 2 * tyi) * 3 + (kx0 + 2 * txi);
        ok = (r < Cout) & (k < K);
        return bsel(base, seg) + (ok ? ((long)r * 9 + tap) * C + c : 0L);
    }
    // (row, column) decode split: the K loop re-uses a row decode across K-steps and a column decode across rows
    struct Row { int r; };
    struct Col { int c8; };
    __device__ Row row(int r) const { return Row{r}; }
    __device__ Col col(int c8) const { return Col{c8}; }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const { return ptr(seg, rw.r, cl.c8, ok); }
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const { return ptr(seg, r, cl.c8, ok); }
};

// dgrad weight view (TR operand): reduce row = (tap, co), contiguous columns = ci of Wc[co][tap][ci]
struct ConvWgtTRLoader {
    static constexpr const char* kName = "ConvWgtTRLoader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Cout, Cin;
    int ntaps = 9;  // 3 x 3; 25 / 49 for the 5 x 5 / 7 x 7 convolutions of the 600M head (ig_convk_*)
    FDiv f_co;  // set by finish()
    void finish() { f_co = make_fdiv(Cout); }
    __device__ void init(int) {}
    __device__ int kdim() const { return -1; }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const {
        int c = c8 * 8;
        int tap = f_co.div(r), co = r - tap * Cout;
        ok = (tap < ntaps) & (c < Cin);
        return bsel(base, seg) + (ok ? ((long)co * ntaps + tap) * Cin + c : 0L);
    }
    // (row, column) decode split: the K loop re-uses a row decode across K-steps and a column decode across rows
    struct Row { int r; };
    struct Col { int c8; };
    __device__ Row row(int r) const { return Row{r}; }
    __device__ Col col(int c8) const { return Col{c8}; }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const { return ptr(seg, rw.r, cl.c8, ok); }
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const { return ptr(seg, r, cl.c8, ok); }
};

// ConvTranspose dgrad A: row = input pixel (b,iy,ix), k = (tap, co): reads dOut(2iy-1+ky, 2ix-1+kx).
// With fixed_tap >= 0 (wgrad, TR operand) the column unit is co only and the tap comes from init(z).
struct ConvTGradLoader {
    static constexpr const char* kName = "ConvTGradLoader";
    static constexpr bool kLinearK = false;
    __device__ long kstride() const { return 0; }
    const bf16_t* base[3];
    int Mtot, H, W, Cout;  // H,W = input resolution; dOut is (2H,2W)
    int fixed_tap;         // -1: k=(tap,co); -2: take tap from blockIdx.z
    FDiv f_hw, f_w, f_c;
    void finish() { f_hw = make_fdiv(H * W), f_w = make_fdiv(W), f_c = make_fdiv(Cout); }
    int tapz;
    __device__ void init(int z) { tapz = z; }
    __device__ int kdim() const { return -1; }
    struct Row { long off; unsigned mask; };  // off = element offset of output pixel (2y, 2x); mask bit tap: (2y-1+ky, 2x-1+kx) exists
    struct Col { int tap, delta, ky, kx; bool ok; };
    __device__ void pixel(int r, int& b, int& y, int& x) const {
        b = f_hw.div(r);
        const int rem = r - b * (H * W);
        y = f_w.div(rem), x = rem - y * W;
    }
    __device__ Row row(int r) const {
        int b, y, x;
        pixel(r, b, y, x);
        unsigned mask = 0;
        if (r < Mtot) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const bool v = ((unsigned)(2 * y - 1 + ky) < (unsigned)(2 * H)) & ((unsigned)(2 * x - 1 + kx) < (unsigned)(2 * W));
                    mask |= (v ? 1u : 0u) << (ky * 3 + kx);
                }
        }
        return Row{(((long)(b * 2 * H + 2 * y)) * (2 * W) + 2 * x) * Cout, mask};
    }
    __device__ Col col(int c8) const {
        const int k = c8 * 8;
        const int tap = (fixed_tap == -1) ? f_c.div(k) : tapz;
        const int c = (fixed_tap == -1) ? k - tap * Cout : k;
        const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
        return Col{tap, ((ky - 1) * (2 * W) + (kx - 1)) * Cout + c, ky, kx, (tap < 9) && (c < Cout)};
    }
    __device__ const bf16_t* at(int seg, const Row& rw, const Col& cl, bool& ok) const {
        ok = cl.ok & ((rw.mask >> cl.tap) & 1u);
        return bsel(base, seg) + (ok ? rw.off + cl.delta : 0L);
    }
    __device__ const bf16_t* ptr_tr(int seg, int r, const Col& cl, bool& ok) const {
        int b, y, x;
        pixel(r, b, y, x);
        ok = cl.ok & (r < Mtot) & ((unsigned)(2 * y - 1 + cl.ky) < (unsigned)(2 * H)) & ((unsigned)(2 * x - 1 + cl.kx) < (unsigned)(2 * W));
        return bsel(base, seg) + (ok ? (((long)(b * 2 * H + 2 * y)) * (2 * W) + 2 * x) * Cout + cl.delta : 0L);
    }
    __device__ const bf16_t* ptr(int seg, int r, int c8, bool& ok) const { return ptr_tr(seg, r, col(c8), ok); }
};

// ---------------------------------------------------------------------------------- epilogues
// bf16/split store: out = drop(act(acc + bias)); optional pre-activation copy; optional ConvT phase row map
struct EpStore {
    static constexpr const char* kName = "EpStore";
    static constexpr bool kColSum = false;
    static constexpr bool kStagedAtomic = false;
    bf16_t *out_hi, *out_lo;
    bf16_t *pre_hi, *pre_lo;
    const float* bias;
    const float *col_scale, *col_shift;  // optional: v = relu((acc+bias)*scale[n] + shift[n]) (eval-mode BatchNorm+ReLU)
    long ldo;
    int act;  // 0 none, 1 exact GELU
    uint32_t drop_seed, drop_thresh;
    const uint32_t* drop_seed_dev;  // optional per-step offset read on the device (graph-replay safe)
    float drop_inv;
    int phase_map, H, W;  // phase_map=1: row (b,iy,ix) -> (b, 2iy+py, 2ix+px) of a (2H,2W) image
    FDiv f_hw, f_w;       // multiply-high division by H*W and W (set with phase_map; a runtime integer division per store was ~80 instructions)
    int py, px;
    __device__ void init(int z) {
        if (phase_map) z = 3 - z;  // same phase order as the ConvT loaders (heaviest first)
        py = z >> 1, px = z & 1;
        if (drop_seed_dev) drop_seed += *drop_seed_dev;
    }
    __device__ void store(int m, int n, f32x4 a) const {
        long row = m;
        if (phase_map) {
            int hw = H * W;
            int b = f_hw.div(m), rem = m - b * hw;
            int y = f_w.div(rem), x = rem - y * W;
            row = ((long)(b * 2 * H + 2 * y + py)) * (2 * W) + 2 * x + px;
        }
        float v[4] = {a[0], a[1], a[2], a[3]};
        if (bias) {
            float4 bb = *reinterpret_cast<const float4*>(bias + n);
            v[0] += bb.x, v[1] += bb.y, v[2] += bb.z, v[3] += bb.w;
        }
        if (col_scale) {
            float4 sc = *reinterpret_cast<const float4*>(col_scale + n);
            float4 sh = *reinterpret_cast<const float4*>(col_shift + n);
            v[0] = fmaxf(v[0] * sc.x + sh.x, 0.f), v[1] = fmaxf(v[1] * sc.y + sh.y, 0.f);
            v[2] = fmaxf(v[2] * sc.z + sh.z, 0.f), v[3] = fmaxf(v[3] * sc.w + sh.w, 0.f);
        }
        size_t idx = (size_t)row * ldo + n;
        if (act == 1) {
            // packed fp32 pairs (v_pk_fma_f32): the scalar form cost as many issue cycles as the K loop of a K = 768 tile
            f32x2 g0, g1, d0, d1;
            if (pre_hi) {  // training: also save gelu'(pre-activation), the only thing backward needs of it
                gelu_erf_pair<true>(f32x2{v[0], v[1]}, g0, d0);
                gelu_erf_pair<true>(f32x2{v[2], v[3]}, g1, d1);
                const float dg[4] = {d0.x, d0.y, d1.x, d1.y};
                store4_split(pre_hi, pre_lo, idx, dg);
            } else {
                gelu_erf_pair<false>(f32x2{v[0], v[1]}, g0, d0);
                gelu_erf_pair<false>(f32x2{v[2], v[3]}, g1, d1);
            }
            v[0] = g0.x, v[1] = g0.y, v[2] = g1.x, v[3] = g1.y;
        }
        if (drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, drop_thresh, drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
        store4_split(out_hi, out_lo, idx, v);
    }
};

// dgrad store with an elementwise factor: mode 1: * dact[m][n] (the gelu' saved by the forward); mode 2: * dropout mask(idx)
struct EpGradStore {
    static constexpr const char* kName = "EpGradStore";
    static constexpr bool kStagedAtomic = false;
    static constexpr bool kColSum = true;  // optional fused column sums of the stored values (bias gradient)
    float* colsum;
    bf16_t *out_hi, *out_lo;
    const bf16_t *pre_hi, *pre_lo;
    long ldo;
    int mode;
    uint32_t drop_seed, drop_thresh;
    const uint32_t* drop_seed_dev;
    float drop_inv;
    __device__ void init(int) {
        if (drop_seed_dev) drop_seed += *drop_seed_dev;
    }
    __device__ f32x4 store_ret(int m, int n, f32x4 a) const {
        size_t idx = (size_t)m * ldo + n;
        float v[4] = {a[0], a[1], a[2], a[3]};
        if (mode == 1) {  // elementwise factor saved by the forward epilogue (gelu'), 4 values = one 8-byte load
            const uint2 u = *reinterpret_cast<const uint2*>(pre_hi + idx);
            float f[4] = {__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                          __uint_as_float(u.y & 0xffff0000u)};
            if (pre_lo) {
                const uint2 l = *reinterpret_cast<const uint2*>(pre_lo + idx);
                f[0] += __uint_as_float(l.x << 16), f[1] += __uint_as_float(l.x & 0xffff0000u);
                f[2] += __uint_as_float(l.y << 16), f[3] += __uint_as_float(l.y & 0xffff0000u);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= f[i];
        } else if (mode == 2 && drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, drop_thresh, drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
        store4_split(out_hi, out_lo, idx, v);
        return f32x4{v[0], v[1], v[2], v[3]};
    }
    __device__ void store(int m, int n, f32x4 a) const { (void)store_ret(m, n, a); }
};

// fp32 residual: out[m][n] = resid[m][n] + acc + bias[n]
struct EpResidual {
    static constexpr const char* kName = "EpResidual";
    static constexpr bool kColSum = false;
    static constexpr bool kStagedAtomic = false;
    float* out;
    const float* resid;
    const float* bias;
    long ldo;
    __device__ void init(int) {}
    __device__ void store(int m, int n, f32x4 a) const {
        size_t idx = (size_t)m * ldo + n;
        float4 r = *reinterpret_cast<const float4*>(resid + idx);
        float4 bb = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0, 0, 0, 0);
        float4 o = make_float4(r.x + a[0] + bb.x, r.y + a[1] + bb.y, r.z + a[2] + bb.z, r.w + a[3] + bb.w);
        *reinterpret_cast<float4*>(out + idx) = o;
    }
};

// patch embed: token row m=(b, tp) -> x[b*Ntok + 1 + tp][n] = acc + bias[n] + pos[1+tp][n]   (pritvhi.py:513-517)
struct EpPatchEmbed {
    static constexpr const char* kName = "EpPatchEmbed";
    static constexpr bool kColSum = false;
    static constexpr bool kStagedAtomic = false;
    float* x;
    const float* bias;
    const float* pos;  // [Ntok][D]
    int tokens_per_chip;  // T*g*g
    long D;
    __device__ void init(int) {}
    __device__ void store(int m, int n, f32x4 a) const {
        int b = m / tokens_per_chip, tp = m - b * tokens_per_chip;
        size_t row = (size_t)b * (tokens_per_chip + 1) + 1 + tp;
        float4 bb = *reinterpret_cast<const float4*>(bias + n);
        float4 pp = *reinterpret_cast<const float4*>(pos + (size_t)(1 + tp) * D + n);
        float4 o = make_float4(a[0] + bb.x + pp.x, a[1] + bb.y + pp.y, a[2] + bb.z + pp.z, a[3] + bb.w + pp.w);
        *reinterpret_cast<float4*>(x + row * D + n) = o;
    }
};

// wgrad: fp32 atomic accumulate into the gradient buffer; column offset z*zstride (ConvT taps)
struct EpAtomic {
    static constexpr const char* kName = "EpAtomic";
    static constexpr bool kColSum = false;
    static constexpr bool kStagedAtomic = true;
    float* out;
    long ldo;
    long zstride;
    long zoff;
    // deterministic split-K: with `partial` set, split blockIdx.y STORES its tile into slab blockIdx.y of a workspace (plain,
    // row-contiguous stores) and splitk_reduce_kernel adds the slabs to `out` in split order afterwards -- no float atomics
    float* partial;
    long slab;
    // gemm2, one tile per workgroup: `pairs` = number of K splits when the launch is ONE-dimensional and the (split, tile) pairs
    // are dealt to the XCDs in split-major order (see gemm2_kernel); `split` = this workgroup's split (set by the kernel)
    int pairs = 0;
    int split = -1;
    int ntiles = 0;  // v1 engine: output tiles of the launch (the kernel is not told otherwise)
    __device__ void init(int z) { zoff = (long)z * zstride; }
    __device__ void add(int m, int n, float v) const {
        if (partial) partial[(size_t)(split >= 0 ? split : (int)blockIdx.y) * slab + (size_t)m * ldo + zoff + n] = v;
        else ig_red_add(out + (size_t)m * ldo + zoff + n, v);
    }
    __device__ void store(int m, int n, f32x4 a) const {
        float* p = out + (size_t)m * ldo + zoff + n;
        ig_red_add(p + 0, a[0]);
        ig_red_add(p + 1, a[1]);
        ig_red_add(p + 2, a[2]);
        ig_red_add(p + 3, a[3]);
    }
};

// out[i] += sum_s partial[s][i] in split order (fixed summation order: bit-reproducible weight gradients)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float4* __restrict__ part, float4* __restrict__ out, long n4, int ks,
                                                            long slab4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 s = part[i];
        for (int k = 1; k < ks; ++k) {
            const float4 t = part[k * slab4 + i];
            s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
        }
        float4 o = out[i];
        o.x += s.x, o.y += s.y, o.z += s.z, o.w += s.w;
        out[i] = o;
    }
}
// workspace of the split-K partial tiles: per (device, stream), grown on demand (runtime.hip: ig_scratch slot 5; a buffer a captured
// hipGraph may hold is never freed there).  While a stream capture is active nothing may be allocated: the caller then falls back to
// atomics unless the workspace already fits.
inline float* splitk_workspace(size_t bytes, hipStream_t st) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    return (float*)ig_scratch2(5, bytes, !capturing, st);
}
constexpr bool splitk_partial_enabled() { return true; }  // split-K partials + ordered fold (the float-atomic form was an A/B arm)

// ------------------------------------------------------------------------------------ launch
// Engine choice for the plain-matrix GEMMs that the 8-phase / 4-wave engines do not cover: v2 everywhere except the shapes where the
// ping-pong v5 measured faster on the same box (tools/gemm_bench.py): dgrad without an elementwise factor (+8-9 %) and the residual
// GEMM with a long reduction (fc2, K = 4D: +9 %).  (The IG_GEMM switch that forced one engine was an A/B arm and is gone.)
inline int gemm_version() { return 2; }
inline int gemm_version_prefer5(bool prefer) { return prefer ? 5 : 2; }
inline int conv_version(int n_out) {
    (void)n_out;
    return 1;  // measured: v1 is faster for every conv stage (gather address math per LDS-DMA issue)
}
inline const bf16_t* zero_page() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return (const bf16_t*)z;
}

constexpr int tr_pitch(int upr) { return upr <= 8 ? 8 : upr <= 12 ? 12 : 16; }

template <class AL, class BL, class EP, bool A_TR, bool B_TR>
int launch_gemm(const AL& al, const BL& bl, const EP& ep_in, int M, int N, int K, int Z, bool split, hipStream_t st,
                const char* what, bool allow_ksplit = false, int force_ver = 0) {
    if (M <= 0 || N <= 0 || K <= 0) return IG_OK;
    EP ep = ep_in;
    // Split-K weight gradients (atomic epilogue): `ks` splits store their tiles into workspace slabs and one reduce launch adds them
    // to the gradient in a fixed order (deterministic; the split-K float atomics were ~25 us of each ~100 us launch).
    int partial_ks = 0;
    auto prep_partial = [&](int ks) -> int {
        if constexpr (EP::kStagedAtomic) {
            partial_ks = 0;
            // Z > 1 (the taps of the ConvTranspose weight gradient in blockIdx.z): the z slices must tile the rows of `out` exactly,
            // because the reduce adds WHOLE slabs -- an element no workgroup stored would be garbage
            const bool z_ok = Z == 1 || ((long)Z * ep.zstride == ep.ldo && (long)N == ep.zstride);
            if (ks > 1 && z_ok && splitk_partial_enabled() && ((long)M * ep.ldo) % 4 == 0 && (((uintptr_t)ep.out) & 15) == 0) {
                float* ws = splitk_workspace((size_t)ks * M * ep.ldo * sizeof(float), st);
                if (ws) {  // (no workspace -- allocation failed or a capture is active: the atomic form, still correct)
                    ep.partial = ws, ep.slab = (long)M * ep.ldo;
                    partial_ks = ks;
                }
            }
        }
        return IG_OK;
    };
    auto finish_partial = [&]() {
        if constexpr (EP::kStagedAtomic) {
            if (partial_ks > 1) {
                const long n4 = (long)M * ep.ldo / 4;
                const int blocks = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float4*)ep.partial, (float4*)ep.out, n4,
                                   partial_ks, n4);
            }
        }
    };
    // v2 (256x128, LDS-DMA ring) wins on the encoder linears; the head convolutions (Cout 48..384, huge M) are
    // better served by the 128x128 register-staged tile at 2 workgroups/CU until a narrow-N tile exists
    int ver = force_ver ? force_ver : gemm_version();
    if (!EP::kStagedAtomic && AL::kLinearK && BL::kLinearK) {
        // small problems (e.g. the YAML's batch 16: M = 3152 rows): the big tiles leave most CUs idle -- a 256x256 tile
        // needs >= 192 tiles to be worth one workgroup per CU, a 256x128 tile >= 256; below that the 128x128 engine's
        // 2-4x larger tile count wins (measured at B = 16: fc2 94.9 us on v5)
        if (ver == 5 && (long)ig_cdiv(M, 256) * ig_cdiv(N, 256) < 192) ver = 2;
        if (ver == 2 && (long)ig_cdiv(M, 256) * ig_cdiv(N, 128) < 256) ver = 1;
    }
    if constexpr (EP::kColSum) {
        if (ver == 1 && ep.colsum) ver = 2;  // the fused column sums live in the v2 / v5 epilogues only
    }
    if constexpr (AL::kLinearK && BL::kLinearK) {
        if (ver == 5) {
            const int tm5 = ig_cdiv(M, 256), tn5 = ig_cdiv(N, 256), ntiles = tm5 * tn5;
            dim3 grid5(ig_tile_grid(ntiles, 1), 1, Z);  // persistent: one workgroup per CU
            int kchunk5 = ig_cdiv(K, 32);
            if constexpr (EP::kStagedAtomic) {  // one tile per workgroup, split-K over blockIdx.y, one workgroup per CU
                grid5.x = ntiles;
                const int nk32 = kchunk5;
                int ks = 256 / (ntiles * Z);
                if (ks > nk32 / 16) ks = nk32 / 16;
                if (ks < 1) ks = 1;
                kchunk5 = ig_cdiv(nk32, ks);
                grid5.y = ig_cdiv(nk32, kchunk5);
            }
            const bf16_t* zp5 = zero_page();
            if (!zp5) {
                ig_set_error("%s: could not allocate the zero page", what);
                return IG_ERR_HIP;
            }
#define IG_LAUNCH_V5(NSEG_)                                                                                           \
    {                                                                                                                  \
        auto kern = gemm5_kernel<AL, BL, EP, A_TR, B_TR, NSEG_>;                                                       \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G5_SMEM);         \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        ig_note_kernel("gemm5_kernel<%s,%s,%s,%s,%s,%d>", AL::kName, BL::kName, EP::kName, A_TR ? "true" : "false", B_TR ? "true" : "false", NSEG_); \
        ig_note_grid((int)grid5.x);                                                                                    \
        hipLaunchKernelGGL(kern, grid5, dim3(NTHR5), G5_SMEM, st, al, bl, ep, M, N, K, tn5, ntiles, kchunk5, zp5);     \
    }
            if (prep_partial((int)grid5.y) != IG_OK) return IG_ERR_HIP;
            if (split) IG_LAUNCH_V5(3) else IG_LAUNCH_V5(1)
#undef IG_LAUNCH_V5
            finish_partial();
            return ig_check_launch(what);
        }
    }
    if (ver == 5) ver = 2;  // shapes / epilogues v5 does not cover
    // v1 tile shape: the candidate with the fewest padded rows/columns, ties to the larger tile.  Weight gradients
    // (atomic epilogue) vary the tile height (M = Cout), everything else the width (N = Cout).  Code 1 = 48.
    int mt = 4, nt = 4;
    int bm_rows = BM2, bn_cols = BN;
    if (ver == 1) {
        static const int cand[4] = {128, 96, 64, 48};
        static const int code[4] = {4, 3, 2, 1};
        const int dim = EP::kStagedAtomic ? M : N;
        long best = -1;
        int pick = 0;
        // cost = padded extent / relative tile efficiency, from the rates the head stages reach (configs[2] bench): forward / data
        // gradient 128-wide ~800, 96-wide (K-step 32) ~620, 48-wide ~450 TFLOP/s; weight gradient 128-row ~580, 96-row ~450,
        // 48-row ~280.  So 576 channels take 5 x 128 (11 % padding) rather than 6 x 96, and 144 take 2 x 96 rather than 3 x 48.
        static const int eff_n[4] = {100, 76, 60, 55}, eff_m[4] = {100, 80, 60, 50};
        const int* eff = EP::kStagedAtomic ? eff_m : eff_n;
        for (int c = 0; c < 4; ++c) {
            if (EP::kStagedAtomic && cand[c] == 64) continue;
            long padded = (long)ig_cdiv(dim, cand[c]) * cand[c];
            padded = padded * 100 / eff[c];
            if (best < 0 || padded < best) best = padded, pick = c;
        }
        if (EP::kStagedAtomic) mt = code[pick], bm_rows = cand[pick], bn_cols = 128;
        else nt = code[pick], bn_cols = cand[pick], bm_rows = nt == 1 ? 256 : 128;
    }
    int tm = ig_cdiv(M, bm_rows), tn = ig_cdiv(N, bn_cols);
    // K-step 32 for the narrow (<= 96 columns) non-atomic v1 tiles
    const bool bk32 = ver == 1 && !EP::kStagedAtomic && nt <= 3;
    int nk_all = ig_cdiv(K, bk32 ? 32 : BK);
    int ksplit = 1;
    if (allow_ksplit) {  // atomic epilogues only: fill the chip once; every extra split is one more atomic pass
        int tiles = tm * tn * Z;
        int slots = ver == 2 ? 256 : 512;
        ksplit = slots / tiles;  // floor: tiles*ksplit must not spill into a second, mostly empty round
        if (ksplit > nk_all / 8) ksplit = nk_all / 8;
        if (ksplit < 1) ksplit = 1;
    }
    int kchunk = ig_cdiv(nk_all, ksplit);
    ksplit = ig_cdiv(nk_all, kchunk);
    dim3 grid(tm * tn, ksplit, Z);
    if (ver == 2) {
        const int ntiles = tm * tn;
        const bf16_t* zp = zero_page();
        if (!zp) {
            ig_set_error("%s: could not allocate the zero page", what);
            return IG_ERR_HIP;
        }
#define IG_LAUNCH_V2(NSEG_, BKT_)                                                                                        \
    {                                                                                                                     \
        auto kern = gemm2_kernel<AL, BL, EP, A_TR, B_TR, NSEG_, BKT_>;                                                    \
        static bool attr_done = false;                                                                                    \
        if (!attr_done) {                                                                                                 \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, G2<BKT_>::SMEM);     \
            attr_done = true;                                                                                             \
        }                                                                                                                 \
        ig_note_kernel("gemm2_kernel<%s,%s,%s,%s,%s,%d,%d,1>", AL::kName, BL::kName, EP::kName, A_TR ? "true" : "false", B_TR ? "true" : "false", NSEG_, BKT_); \
        ig_note_grid((int)grid.x);                                                                                        \
        hipLaunchKernelGGL(kern, grid, dim3(NTHR2), G2<BKT_>::SMEM, st, al, bl, ep, M, N, K, tn, ntiles, kchunk, zp);     \
    }
        if constexpr (EP::kStagedAtomic) {
            // one tile per workgroup (split-K over blockIdx.y); the fp32 tile is staged through the ring in two halves
            const int nk32 = ig_cdiv(K, 32);
            const bool dual = !split && AL::kLinearK && BL::kLinearK && nk32 >= 64;
            int ks = (dual ? 256 : 512) / (tm * tn * Z);  // two 8-wave workgroups per CU, or one 16-wave workgroup (two K halves)
            if (ks > nk32 / (dual ? 32 : 16)) ks = nk32 / (dual ? 32 : 16);
            if (ks < 1) ks = 1;
            kchunk = ig_cdiv(nk32, ks);
            grid.y = ig_cdiv(nk32, kchunk);
            if constexpr (AL::kLinearK && BL::kLinearK) {
                if (dual) {
                    auto kern = gemm2_kernel<AL, BL, EP, A_TR, B_TR, 1, 32, 2>;
                    static bool attr_dual = false;
                    if (!attr_dual) {
                        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G2<32>::SMEM);
                        attr_dual = true;
                    }
                    ig_note_kernel("gemm2_kernel<%s,%s,%s,%s,%s,1,32,2>", AL::kName, BL::kName, EP::kName, A_TR ? "true" : "false", B_TR ? "true" : "false");
                    if (prep_partial((int)grid.y) != IG_OK) return IG_ERR_HIP;
                    {  // 1-D grid, (split, tile) pairs dealt split-major to the XCDs (see gemm2_kernel)
                        ep.pairs = (int)grid.y;
                        grid.x = grid.x * grid.y, grid.y = 1;
                    }
                    hipLaunchKernelGGL(kern, grid, dim3(2 * NTHR2), 2 * G2<32>::SMEM, st, al, bl, ep, M, N, K, tn, ntiles, kchunk, zp);
                    finish_partial();
                    return ig_check_launch(what);
                }
            }
            if (prep_partial((int)grid.y) != IG_OK) return IG_ERR_HIP;
            {
                if (grid.y > 1) {
                    ep.pairs = (int)grid.y;
                    grid.x = grid.x * grid.y, grid.y = 1;
                }
            }
            if (split) IG_LAUNCH_V2(3, 32) else IG_LAUNCH_V2(1, 32)
            finish_partial();
        } else {
            // persistent: two workgroups per CU walk the tile list; BK = 32 keeps the ring at 72 KiB
            grid.x = ig_tile_grid((int)grid.x, 2);  // persistent: two workgroups per CU
            kchunk = ig_cdiv(K, 32);
            if (split) IG_LAUNCH_V2(3, 32) else IG_LAUNCH_V2(1, 32)
        }
#undef IG_LAUNCH_V2
        return ig_check_launch(what);
    }
    dim3 block(NTHR);
#define IG_LAUNCH_V1K(NSEG_, MT_, NT_, WM_, BKT_)                                                                      \
    {                                                                                                                  \
        auto kern = gemm_kernel<AL, BL, EP, A_TR, B_TR, NSEG_, MT_, NT_, WM_, BKT_>;                                   \
        constexpr int bme_ = WM_ * MT_ * 16, bne_ = (4 / WM_) * NT_ * 16;                                              \
        constexpr int ta_ = A_TR ? BKT_ * 16 * tr_pitch(bme_ / 8) : BKT_ * 2 * bme_;                                   \
        constexpr int tb_ = B_TR ? BKT_ * 16 * tr_pitch(bne_ / 8) : BKT_ * 2 * bne_;                                   \
        constexpr int ring_ = 2 * (ta_ + tb_), stage_ = EP::kStagedAtomic ? bme_ * 512 : 0;                            \
        constexpr int lds_ = ring_ > stage_ ? ring_ : stage_;                                                          \
        static bool attr_done = false;                                                                                 \
        if (!attr_done) {                                                                                              \
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_);            \
            attr_done = true;                                                                                          \
        }                                                                                                              \
        ig_note_kernel("gemm_kernel<%s,%s,%s,%s,%s,%d,%d,%d,%d,%d>", AL::kName, BL::kName, EP::kName, A_TR ? "true" : "false", B_TR ? "true" : "false", NSEG_, MT_, NT_, WM_, BKT_); \
        hipLaunchKernelGGL(kern, grid, block, lds_, st, al, bl, ep, M, N, K, tn, kchunk);                              \
    }
#define IG_LAUNCH_V1(NSEG_, MT_, NT_, WM_) IG_LAUNCH_V1K(NSEG_, MT_, NT_, WM_, 64)
    if constexpr (EP::kStagedAtomic) {
        if (prep_partial((int)grid.y) != IG_OK) return IG_ERR_HIP;
        if (grid.y > 1) {  // 1-D grid of (split, tile) pairs, split-major per XCD (see gemm_kernel)
            ep.pairs = (int)grid.y, ep.ntiles = (int)grid.x;
            grid.x = grid.x * grid.y, grid.y = 1;
        }
        if (mt == 1) {  // 48 x 128
            if (split) IG_LAUNCH_V1(3, 3, 2, 1) else IG_LAUNCH_V1(1, 3, 2, 1)
        } else if (mt == 3) {
            if (split) IG_LAUNCH_V1(3, 3, 4, 2) else IG_LAUNCH_V1(1, 3, 4, 2)
        } else {
            if (split) IG_LAUNCH_V1(3, 4, 4, 2) else IG_LAUNCH_V1(1, 4, 4, 2)
        }
    } else if (bk32) {
        // small-channel convolution: K-step 32 (grid / kchunk were computed for it above)
        if (nt == 1) {
            if (split) IG_LAUNCH_V1K(3, 4, 3, 4, 32) else IG_LAUNCH_V1K(1, 4, 3, 4, 32)
        } else if (nt == 2) {
            if (split) IG_LAUNCH_V1K(3, 4, 2, 2, 32) else IG_LAUNCH_V1K(1, 4, 2, 2, 32)
        } else {
            if (split) IG_LAUNCH_V1K(3, 4, 3, 2, 32) else IG_LAUNCH_V1K(1, 4, 3, 2, 32)
        }
    } else {
        if (nt == 1) {  // 256 x 48
            if (split) IG_LAUNCH_V1(3, 4, 3, 4) else IG_LAUNCH_V1(1, 4, 3, 4)
        } else if (nt == 2) {
            if (split) IG_LAUNCH_V1(3, 4, 2, 2) else IG_LAUNCH_V1(1, 4, 2, 2)
        } else if (nt == 3) {
            if (split) IG_LAUNCH_V1(3, 4, 3, 2) else IG_LAUNCH_V1(1, 4, 3, 2)
        } else {
            if (split) IG_LAUNCH_V1(3, 4, 4, 2) else IG_LAUNCH_V1(1, 4, 4, 2)
        }
    }
#undef IG_LAUNCH_V1K
#undef IG_LAUNCH_V1
    finish_partial();
    return ig_check_launch(what);
}

// segment pointer sets: A uses (hi,hi,lo), B uses (hi,lo,hi)
inline void seg_a(const bf16_t* (&b)[3], const void* hi, const void* lo) {
    b[0] = (const bf16_t*)hi, b[1] = (const bf16_t*)hi, b[2] = (const bf16_t*)(lo ? lo : hi);
}
inline void seg_b(const bf16_t* (&b)[3], const void* hi, const void* lo) {
    b[0] = (const bf16_t*)hi, b[1] = (const bf16_t*)(lo ? lo : hi), b[2] = (const bf16_t*)hi;
}
inline PlainLoader plain_a(const void* hi, const void* lo, int R, int C, long ld) {
    PlainLoader l;
    seg_a(l.base, hi, lo);
    l.R = R, l.C = C, l.ld = ld;
    return l;
}
inline PlainLoader plain_b(const void* hi, const void* lo, int R, int C, long ld) {
    PlainLoader l;
    seg_b(l.base, hi, lo);
    l.R = R, l.C = C, l.ld = ld;
    return l;
}
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace
IG_DET_TU(gemm)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

#define IG_SPLIT_CONSISTENT(a_lo, b_lo) \
    IG_REQUIRE(((a_lo) == nullptr) == ((b_lo) == nullptr), "split (lo) pointers must be given for all bf16 operands or none")

extern "C" {

// y[M][N] = act(x[M][K] @ w[N][K]^T + bias)            act: 0 none, 1 GELU (pre-activation copy optional)
int ig_linear_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                  void* y_hi, void* y_lo, void* pre_hi, void* pre_lo, int M, int N, int K, int act, void* stream) {
    IG_REQUIRE(x_hi && w_hi && y_hi, "ig_linear_fwd: null pointer");
    IG_REQUIRE(N % 8 == 0 && K % 8 == 0, "ig_linear_fwd: N and K must be multiples of 8 (got %d, %d)", N, K);
    IG_REQUIRE(aligned16(x_hi) && aligned16(w_hi) && aligned16(y_hi), "ig_linear_fwd: pointers must be 16-byte aligned");
    IG_SPLIT_CONSISTENT(x_lo, w_lo);
    IG_REQUIRE((x_lo == nullptr) == (y_lo == nullptr), "ig_linear_fwd: input and output must both be split or both plain");
    {  // 256 x 256 x 64 8-phase engine (gemm8.hip) for the shapes it covers
        G8Params g{};
        seg_a(g.a, x_hi, x_lo), seg_b(g.b, w_hi, w_lo);
        g.nseg = x_lo ? 3 : 1, g.M = M, g.N = N, g.K = K, g.lda = K, g.ldb = K, g.ldo = N, g.kind = 0, g.act = act, g.bias = bias;
        g.out_hi = (bf16_t*)y_hi, g.out_lo = (bf16_t*)y_lo, g.dact_hi = (bf16_t*)pre_hi, g.dact_lo = (bf16_t*)pre_lo;
        const int rc = ig_gemm8_nt(g, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    EpStore ep{};
    ep.out_hi = (bf16_t*)y_hi, ep.out_lo = (bf16_t*)y_lo, ep.pre_hi = (bf16_t*)pre_hi, ep.pre_lo = (bf16_t*)pre_lo;
    ep.bias = bias, ep.ldo = N, ep.act = act;
    return launch_gemm<PlainLoader, PlainLoader, EpStore, false, false>(
        plain_a(x_hi, x_lo, M, K, K), plain_b(w_hi, w_lo, N, K, K), ep, M, N, K, 1, x_lo != nullptr, (hipStream_t)stream,
        "ig_linear_fwd");
}

// out_f32[M][N] = resid[M][N] + x @ w^T + bias      (residual-stream update; out may alias resid)
int ig_linear_residual_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                           const float* resid, float* out, int M, int N, int K, void* stream) {
    IG_REQUIRE(x_hi && w_hi && resid && out, "ig_linear_residual_fwd: null pointer");
    IG_REQUIRE(N % 8 == 0 && K % 8 == 0, "ig_linear_residual_fwd: N and K must be multiples of 8");
    IG_SPLIT_CONSISTENT(x_lo, w_lo);
    if (aligned16(x_hi) && aligned16(w_hi) && aligned16(resid) && aligned16(out)) {
        G8Params g{};
        seg_a(g.a, x_hi, x_lo), seg_b(g.b, w_hi, w_lo);
        g.nseg = x_lo ? 3 : 1, g.M = M, g.N = N, g.K = K, g.lda = K, g.ldb = K, g.ldo = N, g.kind = 1, g.bias = bias;
        g.outf = out, g.resid = resid;
        const int rc = ig_gemm8_nt(g, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    EpResidual ep{out, resid, bias, (long)N};
    return launch_gemm<PlainLoader, PlainLoader, EpResidual, false, false>(
        plain_a(x_hi, x_lo, M, K, K), plain_b(w_hi, w_lo, N, K, K), ep, M, N, K, 1, x_lo != nullptr, (hipStream_t)stream,
        "ig_linear_residual_fwd", false, gemm_version_prefer5(K >= 2048 && x_lo == nullptr));
}

// dx[M][K] = dy[M][N] @ w[N][K]       mode 0: plain store, 1: * gelu'(pre[M][K])
int ig_linear_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo,
                    const void* pre_hi, const void* pre_lo, float* dx_colsum, int M, int N, int K, int mode, void* stream) {
    IG_REQUIRE(dy_hi && w_hi && dx_hi, "ig_linear_dgrad: null pointer");
    IG_REQUIRE(N % 8 == 0 && K % 8 == 0, "ig_linear_dgrad: N and K must be multiples of 8");
    IG_REQUIRE(mode == 0 || (mode == 1 && pre_hi), "ig_linear_dgrad: mode 1 needs the saved activation-derivative tensor (dact)");
    IG_SPLIT_CONSISTENT(dy_lo, w_lo);
    EpGradStore ep{};
    ep.out_hi = (bf16_t*)dx_hi, ep.out_lo = (bf16_t*)dx_lo, ep.pre_hi = (const bf16_t*)pre_hi, ep.pre_lo = (const bf16_t*)pre_lo;
    ep.ldo = K, ep.mode = mode;
    ep.colsum = dx_colsum;  // optional: dx_colsum[k] += sum_m dx[m][k] (bias gradient of the layer that produced x)
    // C[m][k] = sum_n dy[m][n] * w[n][k]: reduce dim = N; B operand is TR (rows n, contiguous k)
    return launch_gemm<PlainLoader, PlainLoader, EpGradStore, false, true>(
        plain_a(dy_hi, dy_lo, M, N, N), plain_b(w_hi, w_lo, N, K, K), ep, M, K, N, 1, dy_lo != nullptr, (hipStream_t)stream,
        "ig_linear_dgrad", false, gemm_version_prefer5(mode == 0 && dx_colsum == nullptr && dy_lo == nullptr));
}

// ig_linear_dgrad with the weight given TRANSPOSED: wt[K][N] = w^T.  Both operands are then contiguous in the reduce dimension N
// (the forward form), which the 256 x 256 x 64 engine covers; other shapes run the generic engines in the same form.
int ig_linear_dgrad_wt(const void* dy_hi, const void* dy_lo, const void* wt_hi, const void* wt_lo, void* dx_hi, void* dx_lo,
                       const void* pre_hi, const void* pre_lo, float* dx_colsum, int M, int N, int K, int mode, void* stream) {
    IG_REQUIRE(dy_hi && wt_hi && dx_hi, "ig_linear_dgrad_wt: null pointer");
    IG_REQUIRE(N % 8 == 0 && K % 8 == 0, "ig_linear_dgrad_wt: N and K must be multiples of 8");
    IG_REQUIRE(mode == 0 || (mode == 1 && pre_hi), "ig_linear_dgrad_wt: mode 1 needs the saved activation-derivative tensor (dact)");
    IG_REQUIRE(aligned16(dy_hi) && aligned16(wt_hi) && aligned16(dx_hi), "ig_linear_dgrad_wt: pointers must be 16-byte aligned");
    IG_SPLIT_CONSISTENT(dy_lo, wt_lo);
    IG_REQUIRE((dy_lo == nullptr) == (dx_lo == nullptr), "ig_linear_dgrad_wt: input and output must both be split or both plain");
    if (mode == 1 || dx_colsum == nullptr) {
        G8Params g{};
        seg_a(g.a, dy_hi, dy_lo), seg_b(g.b, wt_hi, wt_lo);
        g.nseg = dy_lo ? 3 : 1, g.M = M, g.N = K, g.K = N, g.lda = N, g.ldb = N, g.ldo = K;
        g.kind = mode == 1 ? 2 : 0, g.act = 0, g.bias = nullptr;
        g.out_hi = (bf16_t*)dx_hi, g.out_lo = (bf16_t*)dx_lo;
        if (mode == 1) g.dact_hi = (bf16_t*)pre_hi, g.dact_lo = (bf16_t*)pre_lo, g.colsum = dx_colsum;
        const int rc = ig_gemm8_nt(g, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    EpGradStore ep{};
    ep.out_hi = (bf16_t*)dx_hi, ep.out_lo = (bf16_t*)dx_lo, ep.pre_hi = (const bf16_t*)pre_hi, ep.pre_lo = (const bf16_t*)pre_lo;
    ep.ldo = K, ep.mode = mode;
    ep.colsum = dx_colsum;
    return launch_gemm<PlainLoader, PlainLoader, EpGradStore, false, false>(
        plain_a(dy_hi, dy_lo, M, N, N), plain_b(wt_hi, wt_lo, K, N, N), ep, M, K, N, 1, dy_lo != nullptr, (hipStream_t)stream,
        "ig_linear_dgrad_wt");
}

// dw[N][K] += dy[M][N]^T @ x[M][K]   (fp32 atomic accumulate)
int ig_linear_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, int M, int N,
                    int K, void* stream) {
    IG_REQUIRE(dy_hi && x_hi && dw, "ig_linear_wgrad: null pointer");
    IG_REQUIRE(N % 8 == 0 && K % 8 == 0, "ig_linear_wgrad: N and K must be multiples of 8");
    IG_SPLIT_CONSISTENT(dy_lo, x_lo);
    {  // 8-phase engine with transposed fragment reads (gemm8w.hip) for the shapes it covers
        const int rc = ig_wgrad8_group(1, &dy_hi, &dy_lo, &x_hi, &x_lo, &dw, &N, &K, M, 0, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    EpAtomic ep{dw, (long)K, 0, 0, nullptr, 0};
    // K-steps of 32 per workgroup if the 256 x 128 engine ran this problem (its split-K rule, see launch_gemm)
    const int nk32 = ig_cdiv(M, 32), tiles2 = ig_cdiv(N, 256) * ig_cdiv(K, 128);
    int ks2 = 512 / tiles2;
    if (ks2 > nk32 / 16) ks2 = nk32 / 16;
    if (ks2 < 1) ks2 = 1;
    const int v2_steps = nk32 / ks2;
    return launch_gemm<PlainLoader, PlainLoader, EpAtomic, true, true>(
        plain_a(dy_hi, dy_lo, M, N, N), plain_b(x_hi, x_lo, M, K, K), ep, N, K, M, 1, dy_lo != nullptr, (hipStream_t)stream,
        "ig_linear_wgrad", true,
        // engine: the 256 x 128 engine in its dual-group form (half the split-K atomic volume, launch_gemm) wins from M = 2048 up
        // on every shape measured (M = 3152 .. 21168: +6 .. +20 % over the 128 x 128 engine, proj 768 x 768 included from
        // M = 6304) except the smallest output at the YAML's batch (M = 3152, 768 x 768: 151 vs 175 TFLOP/s); below M = 2048
        // the dual form does not apply and the old rule stands (128 x 128 unless a 256 x 128 workgroup gets >= 50 K-steps)
        nk32 >= 64 ? (((long)N * K <= (1L << 20) && M < 6000) ? 1 : 2) : (((long)N * K <= (1L << 20) || v2_steps < 50) ? 1 : 2));
}

// n weight gradients that share the token count M in ONE launch: dw[g][N[g]][K[g]] += dy[g][M][N[g]]^T @ x[g][M][K[g]].  The pointer
// and size arrays are HOST arrays of n entries (dy_lo / x_lo: NULL or arrays whose entries are all NULL or all set).
int ig_linear_wgrad_group(int n, const void* const* dy_hi, const void* const* dy_lo, const void* const* x_hi, const void* const* x_lo,
                          float* const* dw, const int* N, const int* K, int M, int overwrite, void* stream) {
    IG_REQUIRE(n > 0 && n <= 16 && dy_hi && x_hi && dw && N && K, "ig_linear_wgrad_group: 1..16 GEMMs and non-null arrays");
    for (int g = 0; g < n; ++g) IG_REQUIRE(dy_hi[g] && x_hi[g] && dw[g], "ig_linear_wgrad_group: null pointer in GEMM %d", g);
    if (M <= 0 && !overwrite) return IG_OK;
    if (M > 0) {
        const int rc = ig_wgrad8_group(n, dy_hi, dy_lo, x_hi, x_lo, dw, N, K, M, overwrite, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    for (int g = 0; g < n; ++g) {
        // the per-GEMM engines accumulate: "overwrite" = clear first
        if (overwrite && hipMemsetAsync(dw[g], 0, (size_t)N[g] * K[g] * sizeof(float), (hipStream_t)stream) != hipSuccess) {
            ig_set_error("ig_linear_wgrad_group: hipMemsetAsync failed");
            return IG_ERR_HIP;
        }
        if (M <= 0) continue;
        const int rc = ig_linear_wgrad(dy_hi[g], dy_lo ? dy_lo[g] : nullptr, x_hi[g], x_lo ? x_lo[g] : nullptr, dw[g], M, N[g], K[g], stream);
        if (rc != IG_OK) return rc;
    }
    return IG_OK;
}

// Patch embedding (pritvhi.py:243-268,513-517): x[b][1+tp][:] = patches[b*TP+tp] @ w^T + bias + pos[1+tp]
int ig_patch_embed_fwd(const void* p_hi, const void* p_lo, const void* w_hi, const void* w_lo, const float* bias,
                       const float* pos, float* x, int batch, int tokens_per_chip, int D, int K, void* stream) {
    IG_REQUIRE(p_hi && w_hi && bias && pos && x, "ig_patch_embed_fwd: null pointer");
    IG_REQUIRE(D % 8 == 0 && K % 8 == 0, "ig_patch_embed_fwd: D and K must be multiples of 8");
    IG_SPLIT_CONSISTENT(p_lo, w_lo);
    int M = batch * tokens_per_chip;
    EpPatchEmbed ep{x, bias, pos, tokens_per_chip, (long)D};
    return launch_gemm<PlainLoader, PlainLoader, EpPatchEmbed, false, false>(
        plain_a(p_hi, p_lo, M, K, K), plain_b(w_hi, w_lo, D, K, K), ep, M, D, K, 1, p_lo != nullptr, (hipStream_t)stream,
        "ig_patch_embed_fwd");
}

// ---- 3x3 pad-1 convolution, NHWC activations, weight storage Wc[Cout][9][Cin] (tap = ky*3+kx) ----
// model.py:370-375 (nn.Conv2d(k=3,padding=1))
int ig_conv3x3_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
                   const float* bn_scale, const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout,
                   void* stream) {
    IG_REQUIRE(x_hi && w_hi && y_hi, "ig_conv3x3_fwd: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_conv3x3_fwd: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(x_lo, w_lo);
    IG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ig_conv3x3_fwd: bn_scale and bn_shift go together");
    if (!x_lo && !y_lo) {  // narrow last stage: halo-tile direct convolution (conv_direct.hip)
        const int rc = ig_conv3x3_direct(x_hi, w_hi, bias, bn_scale, bn_shift, y_hi, B, H, W, Cin, Cout, 0, 0, nullptr, 0.f, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    } else if (x_lo && y_lo) {  // the same stage with split operands
        const int rc = ig_conv3x3_direct_split(x_hi, x_lo, w_hi, w_lo, bias, bn_scale, bn_shift, y_hi, y_lo, B, H, W, Cin, Cout, 0, 0, nullptr, 0.f, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    {  // wide stages: implicit GEMM on the 8-phase schedule with gathering LDS-DMA (conv8.hip)
        const int rc = ig_conv8(0, 1, x_hi, x_lo, w_hi, w_lo, bias, bn_scale, bn_shift, y_hi, y_lo, B, H, W, Cin, Cout, 0, nullptr, 0.f, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    Conv3Loader al{};
    seg_a(al.base, x_hi, x_lo);
    al.Mtot = B * H * W, al.H = H, al.W = W, al.C = Cin, al.sign = 1;
    al.finish();
    EpStore ep{};
    ep.out_hi = (bf16_t*)y_hi, ep.out_lo = (bf16_t*)y_lo, ep.bias = bias, ep.ldo = Cout;
    IG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ig_conv3x3_fwd: bn_scale and bn_shift go together");
    ep.col_scale = bn_scale, ep.col_shift = bn_shift;  // eval-mode BatchNorm + ReLU folded into the epilogue
    return launch_gemm<Conv3Loader, PlainLoader, EpStore, false, false>(
        al, plain_b(w_hi, w_lo, Cout, 9 * Cin, 9L * Cin), ep, al.Mtot, Cout, 9 * Cin, 1, x_lo != nullptr,
        (hipStream_t)stream, "ig_conv3x3_fwd", false, conv_version(Cout));
}

// The same convolution in front of a training-mode BatchNorm: where the direct kernel runs (the 48-channel last stage) it also
// leaves sums[2 Cout] = per-channel sum and sum of squares of the stored outputs and sets *fused = 1 (host int); otherwise
// *fused = 0 and the caller runs the statistics pass (ig_bn_relu_fwd with y == NULL).
int ig_conv3x3_fwd_stats(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, void* y_hi,
                         void* y_lo, double* sums, int* fused, int B, int H, int W, int Cin, int Cout, void* stream) {
    IG_REQUIRE(x_hi && w_hi && y_hi && sums && fused, "ig_conv3x3_fwd_stats: null pointer");
    *fused = 0;
    if (!x_lo && !y_lo && Cin % 8 == 0 && Cout % 8 == 0 && (x_lo == nullptr) == (w_lo == nullptr)) {
        const int rc = ig_conv3x3_direct(x_hi, w_hi, bias, nullptr, nullptr, y_hi, B, H, W, Cin, Cout, 0, 0, nullptr, 0.f, stream, sums, fused);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    } else if (x_lo && w_lo && y_lo && Cin % 8 == 0 && Cout % 8 == 0) {
        const int rc = ig_conv3x3_direct_split(x_hi, x_lo, w_hi, w_lo, bias, nullptr, nullptr, y_hi, y_lo, B, H, W, Cin, Cout, 0, 0, nullptr, 0.f, stream, sums, fused);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    *fused = 0;
    return ig_conv3x3_fwd(x_hi, x_lo, w_hi, w_lo, bias, nullptr, nullptr, y_hi, y_lo, B, H, W, Cin, Cout, stream);
}

// Inference: the last Conv2d(k=3, padding=1) (+ eval-mode BatchNorm + ReLU) and the Conv2d(k=1) classifier on top of it.  Where the direct
// 48-channel kernel runs and ncls <= 2 the classifier is applied in the convolution's epilogue (*fused = 1, HOST int; y may be NULL and
// is then not written); otherwise *fused = 0 and NOTHING has been computed: the caller runs ig_conv3x3_fwd and ig_classifier_fwd.
int ig_conv3x3_cls_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                       const float* bn_shift, void* y_hi, const float* cls_w, const float* cls_b, float* logits, int* fused, int B, int H,
                       int W, int Cin, int Cout, int ncls, void* stream) {
    IG_REQUIRE(x_hi && w_hi && cls_w && cls_b && logits && fused, "ig_conv3x3_cls_fwd: null pointer");
    IG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ig_conv3x3_cls_fwd: bn_scale and bn_shift go together");
    *fused = 0;
    if (x_lo || w_lo || Cin != Cout) return IG_OK;
    const int rc = ig_conv3x3_cls_direct(x_hi, w_hi, bias, bn_scale, bn_shift, y_hi, cls_w, cls_b, logits, B, H, W, Cin, ncls, stream);
    if (rc == IG_ERR_UNSUPPORTED) return IG_OK;
    if (rc == IG_OK) *fused = 1;
    return rc;
}

// dx = conv_dgrad(dy, w) [* dropout mask of the conv input when drop_p > 0]
int ig_conv3x3_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo,
                     int B, int H, int W, int Cin, int Cout, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                     void* stream) {
    IG_REQUIRE(dy_hi && w_hi && dx_hi, "ig_conv3x3_dgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_conv3x3_dgrad: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(dy_lo, w_lo);
    IG_REQUIRE(drop_p <= 0.f || (double)B * H * W * Cin < 4294967296.0, "ig_conv3x3_dgrad: dropout needs < 2^32 elements");
    if (!dy_lo && !dx_lo) {
        const int rc = ig_conv3x3_direct(dy_hi, w_hi, nullptr, nullptr, nullptr, dx_hi, B, H, W, Cin, Cout, 1, drop_seed, drop_seed_dev,
                                         drop_p, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    } else if (dy_lo && dx_lo) {
        const int rc = ig_conv3x3_direct_split(dy_hi, dy_lo, w_hi, w_lo, nullptr, nullptr, nullptr, dx_hi, dx_lo, B, H, W, Cin, Cout, 1, drop_seed,
                                               drop_seed_dev, drop_p, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    {
        const int rc = ig_conv8(0, -1, dy_hi, dy_lo, w_hi, w_lo, nullptr, nullptr, nullptr, dx_hi, dx_lo, B, H, W, Cout, Cin, drop_seed,
                                drop_seed_dev, drop_p, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    Conv3Loader al{};
    seg_a(al.base, dy_hi, dy_lo);
    al.Mtot = B * H * W, al.H = H, al.W = W, al.C = Cout, al.sign = -1;
    al.finish();
    ConvWgtTRLoader bl{};
    seg_b(bl.base, w_hi, w_lo);
    bl.Cout = Cout, bl.Cin = Cin;
    bl.finish();
    EpGradStore ep{};
    IG_REQUIRE(drop_p <= 0.f || (double)B * H * W * Cin < 4294967296.0, "ig_conv3x3_dgrad: dropout needs < 2^32 elements");
    ep.out_hi = (bf16_t*)dx_hi, ep.out_lo = (bf16_t*)dx_lo, ep.ldo = Cin, ep.mode = 2;
    ep.drop_seed = drop_seed, ep.drop_seed_dev = drop_seed_dev;
    ep.drop_thresh = ig_drop_thresh16(drop_p);
    ep.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    return launch_gemm<Conv3Loader, ConvWgtTRLoader, EpGradStore, false, true>(
        al, bl, ep, al.Mtot, Cin, 9 * Cout, 1, dy_lo != nullptr, (hipStream_t)stream, "ig_conv3x3_dgrad", false, conv_version(Cin));
}

// split-operand weight gradients of the narrow stages: three launches of the bf16 direct kernels (round 4: +6.6 % on the bf16x3 step)
static constexpr bool direct_x3_env() { return true; }

// dWc[Cout][9][Cin] += sum_pixels dy[p][co] * x[shift_tap(p)][ci]
int ig_conv3x3_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B,
                     int H, int W, int Cin, int Cout, void* stream) {
    IG_REQUIRE(dy_hi && x_hi && dw, "ig_conv3x3_wgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_conv3x3_wgrad: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(dy_lo, x_lo);
    // dbias (optional): the convolution's bias gradient, dbias[co] += sum_pixels dy[p][co] -- fused into the direct kernels,
    // otherwise one column-sum pass over dy
    {  // wide stages: the 8-phase weight-gradient engine with a gathering B operand (gemm8w.hip)
        const int rc = ig_wgrad8_conv(0, dy_hi, dy_lo, x_hi, x_lo, dw, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK && dbias) return ig_colsum(dy_hi, dy_lo, dbias, (long)B * H * W, Cout, stream);
            return rc;
        }
    }
    if (!dy_lo) {  // narrow stages: register-resident partial sums over halo tiles (conv_direct.hip)
        int fused = 0;
        const int rc = ig_conv3x3_wgrad_direct(dy_hi, x_hi, dw, dbias, &fused, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK && dbias && !fused) return ig_colsum(dy_hi, dy_lo, dbias, (long)B * H * W, Cout, stream);
            return rc;
        }
    } else if (direct_x3_env()) {
        // split operands on the narrow stages: the bf16x3 product dy^T x = dy_hi^T x_hi + dy_hi^T x_lo + dy_lo^T x_hi (lo x lo dropped, as in
        // every split GEMM here) is a SUM of three bf16 products accumulated in fp32 -- and the direct kernel accumulates into dw: three
        // launches on the operand pairs (1.7 ms at 48 channels, B = 216, against 4.4 ms on the gather GEMM, whose 128 x 128 tiles fit a 48-row
        // output badly).  The bias gradient is linear in dy: it rides on the launches that carry dy_hi and dy_lo against x_hi.
        int fused = 0, f2 = 0;
        int rc = ig_conv3x3_wgrad_direct(dy_hi, x_hi, dw, dbias, &fused, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK) rc = ig_conv3x3_wgrad_direct(dy_hi, x_lo, dw, nullptr, &f2, B, H, W, Cin, Cout, stream);
            if (rc == IG_OK) rc = ig_conv3x3_wgrad_direct(dy_lo, x_hi, dw, fused ? dbias : nullptr, &f2, B, H, W, Cin, Cout, stream);
            if (rc == IG_OK && dbias && !fused) return ig_colsum(dy_hi, dy_lo, dbias, (long)B * H * W, Cout, stream);
            return rc;
        }
    }
    if (dbias) {
        const int rc = ig_colsum(dy_hi, dy_lo, dbias, (long)B * H * W, Cout, stream);
        if (rc != IG_OK) return rc;
    }
    int Mtot = B * H * W;
    Conv3Loader bl{};
    seg_b(bl.base, x_hi, x_lo);
    bl.Mtot = Mtot, bl.H = H, bl.W = W, bl.C = Cin, bl.sign = 1;
    bl.finish();
    EpAtomic ep{dw, 9L * Cin, 0, 0, nullptr, 0};
    return launch_gemm<PlainLoader, Conv3Loader, EpAtomic, true, true>(
        plain_a(dy_hi, dy_lo, Mtot, Cout, Cout), bl, ep, Cout, 9 * Cin, Mtot, 1, dy_lo != nullptr, (hipStream_t)stream,
        "ig_conv3x3_wgrad", true, 1);
}

// ---- nn.Conv2d(kernel_size=KS, padding=1), KS odd >= 3, NHWC, weight storage Wc[Cout][KS*KS][Cin]: the 5 x 5 / 7 x 7 convolutions of
// the 600M variants' decode head (model.py:169-177, 370-375).  x (B,H,W,Cin) -> y (B,Ho,Wo,Cout), Ho = H + 3 - KS. ----
int ig_convk_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                 const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout, int KS, void* stream) {
    IG_REQUIRE(x_hi && w_hi && y_hi, "ig_convk_fwd: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convk_fwd: channels must be multiples of 8");
    IG_REQUIRE(KS >= 3 && (KS & 1) && KS <= 9 && H + 3 - KS > 0 && W + 3 - KS > 0, "ig_convk_fwd: kernel size %d does not fit a %d x %d input", KS, H, W);
    IG_SPLIT_CONSISTENT(x_lo, w_lo);
    IG_REQUIRE((x_lo == nullptr) == (y_lo == nullptr), "ig_convk_fwd: input and output must both be split or both plain");
    IG_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "ig_convk_fwd: bn_scale and bn_shift go together");
    const int Ho = H + 3 - KS, Wo = W + 3 - KS;
    ConvKLoader al{};
    seg_a(al.base, x_hi, x_lo);
    al.Mtot = B * Ho * Wo, al.Hr = Ho, al.Wr = Wo, al.Hs = H, al.Ws = W, al.C = Cin, al.KS = KS, al.sign = 1;
    al.finish();
    EpStore ep{};
    ep.out_hi = (bf16_t*)y_hi, ep.out_lo = (bf16_t*)y_lo, ep.bias = bias, ep.ldo = Cout;
    ep.col_scale = bn_scale, ep.col_shift = bn_shift;
    return launch_gemm<ConvKLoader, PlainLoader, EpStore, false, false>(
        al, plain_b(w_hi, w_lo, Cout, KS * KS * Cin, (long)KS * KS * Cin), ep, al.Mtot, Cout, KS * KS * Cin, 1, x_lo != nullptr,
        (hipStream_t)stream, "ig_convk_fwd", false, 1);
}

// dx (B,H,W,Cin) = conv_dgrad(dy (B,Ho,Wo,Cout), w) [* dropout mask of the conv input when drop_p > 0]
int ig_convk_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo, int B, int H,
                   int W, int Cin, int Cout, int KS, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream) {
    IG_REQUIRE(dy_hi && w_hi && dx_hi, "ig_convk_dgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convk_dgrad: channels must be multiples of 8");
    IG_REQUIRE(KS >= 3 && (KS & 1) && KS <= 9 && H + 3 - KS > 0 && W + 3 - KS > 0, "ig_convk_dgrad: kernel size %d does not fit a %d x %d input", KS, H, W);
    IG_SPLIT_CONSISTENT(dy_lo, w_lo);
    IG_REQUIRE(drop_p <= 0.f || (double)B * H * W * Cin < 4294967296.0, "ig_convk_dgrad: dropout needs < 2^32 elements");
    const int Ho = H + 3 - KS, Wo = W + 3 - KS;
    ConvKLoader al{};
    seg_a(al.base, dy_hi, dy_lo);
    al.Mtot = B * H * W, al.Hr = H, al.Wr = W, al.Hs = Ho, al.Ws = Wo, al.C = Cout, al.KS = KS, al.sign = -1;
    al.finish();
    ConvWgtTRLoader bl{};
    seg_b(bl.base, w_hi, w_lo);
    bl.Cout = Cout, bl.Cin = Cin, bl.ntaps = KS * KS;
    bl.finish();
    EpGradStore ep{};
    ep.out_hi = (bf16_t*)dx_hi, ep.out_lo = (bf16_t*)dx_lo, ep.ldo = Cin, ep.mode = 2;
    ep.drop_seed = drop_seed, ep.drop_seed_dev = drop_seed_dev;
    ep.drop_thresh = ig_drop_thresh16(drop_p);
    ep.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    return launch_gemm<ConvKLoader, ConvWgtTRLoader, EpGradStore, false, true>(
        al, bl, ep, al.Mtot, Cin, KS * KS * Cout, 1, dy_lo != nullptr, (hipStream_t)stream, "ig_convk_dgrad", false, 1);
}

// dWc[Cout][KS*KS][Cin] += sum over output pixels of dy[p][co] * x[shift_tap(p)][ci];  dbias[co] += sum_p dy[p][co] (optional)
int ig_convk_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B, int H,
                   int W, int Cin, int Cout, int KS, void* stream) {
    IG_REQUIRE(dy_hi && x_hi && dw, "ig_convk_wgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convk_wgrad: channels must be multiples of 8");
    IG_REQUIRE(KS >= 3 && (KS & 1) && KS <= 9 && H + 3 - KS > 0 && W + 3 - KS > 0, "ig_convk_wgrad: kernel size %d does not fit a %d x %d input", KS, H, W);
    IG_SPLIT_CONSISTENT(dy_lo, x_lo);
    const int Ho = H + 3 - KS, Wo = W + 3 - KS, Mo = B * Ho * Wo;
    if (dbias) {
        const int rc = ig_colsum(dy_hi, dy_lo, dbias, (long)Mo, Cout, stream);
        if (rc != IG_OK) return rc;
    }
    ConvKLoader bl{};
    seg_b(bl.base, x_hi, x_lo);
    bl.Mtot = Mo, bl.Hr = Ho, bl.Wr = Wo, bl.Hs = H, bl.Ws = W, bl.C = Cin, bl.KS = KS, bl.sign = 1;
    bl.finish();
    EpAtomic ep{dw, (long)KS * KS * Cin, 0, 0, nullptr, 0};
    return launch_gemm<PlainLoader, ConvKLoader, EpAtomic, true, true>(
        plain_a(dy_hi, dy_lo, Mo, Cout, Cout), bl, ep, Cout, KS * KS * Cin, Mo, 1, dy_lo != nullptr, (hipStream_t)stream,
        "ig_convk_wgrad", true, 1);
}

// ---- ConvTranspose2d(k=3,s=2,p=1,op=1), NHWC, weight storage Wc[Cout][9][Cin]  (model.py:361-368) ----
// y (2H,2W) = drop(convT(x) + bias): four sub-pixel phase GEMMs in one launch (blockIdx.z = phase)
int ig_convT_fwd(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, void* y_hi,
                 void* y_lo, int B, int H, int W, int Cin, int Cout, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                 void* stream) {
    IG_REQUIRE(x_hi && w_hi && y_hi, "ig_convT_fwd: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convT_fwd: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(x_lo, w_lo);
    IG_REQUIRE(drop_p <= 0.f || (double)B * 4 * H * W * Cout < 4294967296.0, "ig_convT_fwd: dropout needs < 2^32 elements");
    if (!x_lo && !y_lo) {  // last stage (96 -> 48): direct sub-pixel kernel (conv_direct.hip)
        const int rc = ig_convT_fwd_direct(x_hi, w_hi, bias, y_hi, B, H, W, Cin, Cout, drop_seed, drop_seed_dev, drop_p, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    {
        const int rc = ig_conv8(1, 1, x_hi, x_lo, w_hi, w_lo, bias, nullptr, nullptr, y_hi, y_lo, B, H, W, Cin, Cout, drop_seed, drop_seed_dev,
                                drop_p, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    ConvTFwdALoader al{};
    seg_a(al.base, x_hi, x_lo);
    al.Mtot = B * H * W, al.H = H, al.W = W, al.C = Cin;
    al.finish();
    ConvTFwdBLoader bl{};
    seg_b(bl.base, w_hi, w_lo);
    bl.Cout = Cout, bl.C = Cin;
    bl.finish();
    EpStore ep{};
    ep.out_hi = (bf16_t*)y_hi, ep.out_lo = (bf16_t*)y_lo, ep.bias = bias, ep.ldo = Cout;
    IG_REQUIRE(drop_p <= 0.f || (double)B * 4 * H * W * Cout < 4294967296.0, "ig_convT_fwd: dropout needs < 2^32 elements");
    ep.phase_map = 1, ep.H = H, ep.W = W;
    ep.f_hw = make_fdiv(H * W), ep.f_w = make_fdiv(W);
    ep.drop_seed = drop_seed, ep.drop_seed_dev = drop_seed_dev;
    ep.drop_thresh = ig_drop_thresh16(drop_p);
    ep.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    return launch_gemm<ConvTFwdALoader, ConvTFwdBLoader, EpStore, false, false>(
        al, bl, ep, al.Mtot, Cout, 4 * Cin, 4, x_lo != nullptr, (hipStream_t)stream, "ig_convT_fwd", false, conv_version(Cout));
}

// dx (H,W,Cin) = stride-2 gather of dy (2H,2W,Cout) against Wc
int ig_convT_dgrad(const void* dy_hi, const void* dy_lo, const void* w_hi, const void* w_lo, void* dx_hi, void* dx_lo,
                   int B, int H, int W, int Cin, int Cout, void* stream) {
    IG_REQUIRE(dy_hi && w_hi && dx_hi, "ig_convT_dgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convT_dgrad: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(dy_lo, w_lo);
    if (!dy_lo && !dx_lo) {  // last stage (96 -> 48): direct stride-2 gather over phase planes (conv_direct.hip)
        const int rc = ig_convT_dgrad_direct(dy_hi, w_hi, dx_hi, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    {
        const int rc = ig_conv8(2, 1, dy_hi, dy_lo, w_hi, w_lo, nullptr, nullptr, nullptr, dx_hi, dx_lo, B, H, W, Cout, Cin, 0, nullptr, 0.f, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    ConvTGradLoader al{};
    seg_a(al.base, dy_hi, dy_lo);
    al.Mtot = B * H * W, al.H = H, al.W = W, al.Cout = Cout, al.fixed_tap = -1;
    al.finish();
    ConvWgtTRLoader bl{};
    seg_b(bl.base, w_hi, w_lo);
    bl.Cout = Cout, bl.Cin = Cin;
    bl.finish();
    EpGradStore ep{};
    ep.out_hi = (bf16_t*)dx_hi, ep.out_lo = (bf16_t*)dx_lo, ep.ldo = Cin, ep.mode = 0;
    return launch_gemm<ConvTGradLoader, ConvWgtTRLoader, EpGradStore, false, true>(
        al, bl, ep, al.Mtot, Cin, 9 * Cout, 1, dy_lo != nullptr, (hipStream_t)stream, "ig_convT_dgrad", false, conv_version(Cin));
}

// dWc[Cout][tap][Cin] += sum_{input pixels} dy[shift_tap(p)][co] * x[p][ci]     (blockIdx.z = tap)
int ig_convT_wgrad(const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, float* dbias, int B,
                   int H, int W, int Cin, int Cout, void* stream) {
    IG_REQUIRE(dy_hi && x_hi && dw, "ig_convT_wgrad: null pointer");
    IG_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0, "ig_convT_wgrad: channels must be multiples of 8");
    IG_SPLIT_CONSISTENT(dy_lo, x_lo);
    // dbias (optional): dbias[co] += sum over the (2H, 2W) output pixels of dy -- fused into the direct kernel, otherwise one
    // column-sum pass over dy
    {  // wide stages: the 8-phase weight-gradient engine with a gathering A operand (gemm8w.hip)
        const int rc = ig_wgrad8_conv(1, dy_hi, dy_lo, x_hi, x_lo, dw, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK && dbias) return ig_colsum(dy_hi, dy_lo, dbias, 4L * B * H * W, Cout, stream);
            return rc;
        }
    }
    if (!dy_lo) {  // last stage (96 -> 48): register-resident partial sums (conv_direct.hip)
        int fused = 0;
        const int rc = ig_convT_wgrad_direct(dy_hi, x_hi, dw, dbias, &fused, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK && dbias && !fused) return ig_colsum(dy_hi, dy_lo, dbias, 4L * B * H * W, Cout, stream);
            return rc;
        }
    } else if (direct_x3_env()) {  // split operands: three launches of the bf16 kernel on (hi, hi), (hi, lo), (lo, hi) -- see ig_conv3x3_wgrad
        int fused = 0, f2 = 0;
        int rc = ig_convT_wgrad_direct(dy_hi, x_hi, dw, dbias, &fused, B, H, W, Cin, Cout, stream);
        if (rc != IG_ERR_UNSUPPORTED) {
            if (rc == IG_OK) rc = ig_convT_wgrad_direct(dy_hi, x_lo, dw, nullptr, &f2, B, H, W, Cin, Cout, stream);
            if (rc == IG_OK) rc = ig_convT_wgrad_direct(dy_lo, x_hi, dw, fused ? dbias : nullptr, &f2, B, H, W, Cin, Cout, stream);
            if (rc == IG_OK && dbias && !fused) return ig_colsum(dy_hi, dy_lo, dbias, 4L * B * H * W, Cout, stream);
            return rc;
        }
    }
    if (dbias) {
        const int rc = ig_colsum(dy_hi, dy_lo, dbias, 4L * B * H * W, Cout, stream);
        if (rc != IG_OK) return rc;
    }
    int Mtot = B * H * W;
    ConvTGradLoader al{};
    seg_a(al.base, dy_hi, dy_lo);
    al.Mtot = Mtot, al.H = H, al.W = W, al.Cout = Cout, al.fixed_tap = -2;
    al.finish();
    EpAtomic ep{dw, 9L * Cin, (long)Cin, 0, nullptr, 0};
    return launch_gemm<ConvTGradLoader, PlainLoader, EpAtomic, true, true>(
        al, plain_b(x_hi, x_lo, Mtot, Cin, Cin), ep, Cout, Cin, Mtot, 9, dy_lo != nullptr, (hipStream_t)stream,
        "ig_convT_wgrad", true, 1);
}

}  // extern "C"
