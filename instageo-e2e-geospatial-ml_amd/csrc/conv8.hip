// conv8: the decode head's wide convolutions (model.py:349-390: nn.ConvTranspose2d(k3,s2,p1,op1) :361-368 and
// nn.Conv2d(k3,padding=1) :370-375, forward and data gradient) as implicit GEMMs on the 8-phase LDS-DMA schedule of gemm8.hip.
//
//   out[m][n] = sum_k A(m, k) * Wp[n][k]          m = pixel of the row grid, k = (tap, channel), n = output channel
//
// * A is GATHERED by the LDS-DMA itself: `buffer_load_dwordx4 ... offen lds` with a 32-bit per-lane offset into the NHWC
//   activation tensor = (row's pixel offset, decoded once per tile and half) + (tap displacement + channel of the lane's
//   16-byte K chunk, one 4-byte entry of a table in LDS).  Taps that fall off the image get an offset beyond the buffer
//   descriptor's bound, which the hardware answers with ZEROS written to LDS (tools/probes/buffer_lds_oob.hip) -- no zero page,
//   no 64-bit address arithmetic, no branch.
// * Wp is the weight tensor Wc[Cout][9][Cin] re-packed per call (conv8_pack_*_kernel, a few MB) into plain K-contiguous
//   [N][Kpad] matrices: per sub-pixel phase for the ConvTranspose forward (1 / 2 / 2 / 4 taps), transposed (n = ci,
//   k = (tap, co)) for the data gradients; K is padded to a multiple of 128 with zeros (the table marks the pad chunks invalid).
// * Tiles: 2 row groups x WC column waves; a wave owns (2 MT 16) rows x (NT0 + NT1) 16 columns.  The two B half-tiles may be
//   UNEQUAL (NT0 != NT1): the schedule's phases are split by the A halves, both B halves are read in the same phase, so the
//   counted waits only need the per-wave totals (2 AI + NT0 + NT1 LDS-DMA instructions in flight).  Instances: 256 x 256,
//   256 x 192, 256 x 128 (8 waves) -- the head's widths 384, 192, 768 (T = 1) and 2304, 1152, 576 (T = 3) tile without padding;
//   a ragged last column tile (N = 144 on 192) clamps its weight rows and skips the dead columns in the epilogue.
// * The four sub-pixel phases of a ConvTranspose forward are tiles of ONE persistent launch with their own K length; a row
//   tile's phases and column tiles are neighbours in the tile list, so the XCD that owns them reads x once; the phase a
//   workgroup gets rotates with the row tile, so every workgroup sees the 1-, 2- and 4-tap tiles equally often.
// * conv4_kernel<NI, PAIR> (round 6, further down): the same implicit GEMM on FOUR waves (256 x 192 / 256 x 96 tiles) with a generated
//   assembly K-loop; it takes the launches whose width tiles by 192 or 96, conv8_kernel the rest (144 / 128 / 256-wide, small launches).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct C8Params {
    const bf16_t* a[2];  // gathered activation tensor (NHWC): hi, lo (NULL: plain bf16)
    unsigned a_bytes;    // bytes of one activation tensor (the buffer descriptor's bound; < 2^31)
    const bf16_t* b[2];  // packed weights: hi, lo
    const int* ktab;     // per 16-byte K chunk: (displacement in 16-byte units << 8) | mask bit (31: padding, always invalid)
    int ktab_n;
    int M, N, C;         // rows, output channels, channels of A
    int H, W, sm;        // row grid per image; source grid = (sm H, sm W), row (y, x) sits at source (sm y, sm x)
    int nphase, nbits;
    unsigned ddy_code, ddx_code;  // 2 bits per mask bit: displacement + 1 (source-grid pixels)
    // ktab starts with a 32-word header: per phase p (8 words each) {K-tiles of 64 (even), first table entry, packed row length in
    // elements, element offset of the packed block inside b[] (lo, hi), 0, 0, 0}; ktab_n counts the entries after the header
    FDiv f_hw, f_w;               // division by H W and W
    const float *bias, *scale, *shift;  // bias (accumulator init); optional y = relu(v scale + shift) (eval-mode BatchNorm + ReLU)
    bf16_t *out_hi, *out_lo;
    int ldo;
    int phase_map;  // 1: row (b, y, x) of phase (py, px) -> output pixel (b, 2y + py, 2x + px) of the (2H, 2W) image
    uint32_t drop_seed, drop_thresh;
    const uint32_t* drop_seed_dev;
    float drop_inv;
};

__device__ __forceinline__ void c8_glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    // M0 = LDS destination of the DMA.  It is neither saved nor restored (2 SALU fewer per issue; round 5: the s_nop 4 -> 0 and this together
    // are worth 3 % on the grouped weight gradients): hipcc keeps nothing in M0 in these kernels -- gfx9 LDS instructions do not read it -- and
    // tests/test_cpu_host.py::test_m0_is_only_written_by_the_lds_dma_helpers checks the ISA for any other M0 reference
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
// LDS-DMA through a buffer descriptor: lanes whose offset lies beyond num_records write zeros
__device__ __forceinline__ void c8_blds16(unsigned voff, i32x4 rsrc, unsigned lds_dst) {
    // M0 = LDS destination of the DMA.  It is neither saved nor restored (2 SALU fewer per issue; round 5: the s_nop 4 -> 0 and this together
    // are worth 3 % on the grouped weight gradients): hipcc keeps nothing in M0 in these kernels -- gfx9 LDS instructions do not read it -- and
    // tests/test_cpu_host.py::test_m0_is_only_written_by_the_lds_dma_helpers checks the ISA for any other M0 reference
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ i32x4 c8_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}
template <class T>
__device__ __forceinline__ T c8_sel4(const T (&v)[4], int i) {
    return i == 0 ? v[0] : i == 1 ? v[1] : i == 2 ? v[2] : v[3];
}

template <int AI>
struct CurA {  // issue cursor of an A half-tile (kt, seg, i, left, ph, nk, toff wave-uniform; rowoff, mask, e per lane)
    int kt, seg, i, nk, toff;
    unsigned rowoff[AI], mask[AI];
    int e;
};
struct CurB {
    const char* base;
    int kt, seg, i, nk, ldb2, vr;
};

template <int WC, int MT, int NT0, int NT1, bool SPLIT_OUT>
struct C8Geo {
    static_assert((2 * MT) % WC == 0, "conv8: every wave issues the same number of A pieces");
    static constexpr int NW = 2 * WC, NTHR = 64 * NW, MH = MT * 16, BM = 4 * MH, NTW = NT0 + NT1, WN = 16 * NTW, BN = WC * WN;
    static constexpr int AI = 2 * MT / WC;
    static constexpr int HALF_A = 2 * MH * 128, HALF_B0 = WC * NT0 * 16 * 128, HALF_B1 = WC * NT1 * 16 * 128;
    static constexpr int OFF_B0 = 2 * HALF_A, OFF_B1 = OFF_B0 + HALF_B0, BUF = OFF_B1 + HALF_B1;
    static constexpr int PITCH = WN * 2 + 16, STG = 16 * PITCH * (SPLIT_OUT ? 2 : 1);
    static constexpr int OFF_STAGE = 2 * BUF, OFF_TAB = OFF_STAGE + NW * STG;
    static constexpr int VMW = 2 * AI + NT0 + NT1;  // LDS-DMA instructions of four half-tiles: what the steady-state waits leave in flight
};

template <int WC, int MT, int NT0, int NT1, int NSEG, bool SPLIT_OUT>
__global__ __launch_bounds__(128 * WC, 2) void conv8_kernel(C8Params p) {
    using Geo = C8Geo<WC, MT, NT0, NT1, SPLIT_OUT>;
    constexpr int NW = Geo::NW, MH = Geo::MH, BM = Geo::BM, NTW = Geo::NTW, WN = Geo::WN, BN = Geo::BN, AI = Geo::AI;
    constexpr int HALF_A = Geo::HALF_A, OFF_B0 = Geo::OFF_B0, OFF_B1 = Geo::OFF_B1, BUF = Geo::BUF, VMW = Geo::VMW;
    constexpr int PITCH = Geo::PITCH, STG = Geo::STG, OFF_STAGE = Geo::OFF_STAGE, OFF_TAB = Geo::OFF_TAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, tpb = tiles_n * p.nphase, ntiles = tiles_m * tpb;
    // persistent, XCD-aware: workgroups are dealt round-robin over the 8 XCDs; XCD x owns a contiguous tile range
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0), nbx_nom = max(nb >> 3, 1);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    if (my_tiles <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    // per-phase constants {K-tiles, first table entry, packed row length, packed block offset (2 words)} in LDS in front of the
    // chunk table: they are read at tile changes only and would otherwise sit in 20 SGPRs for the whole kernel
    int* phc = reinterpret_cast<int*>(smem + OFF_TAB);
    int* tab = phc + 32;
    for (int i = tid; i < p.ktab_n + 32; i += Geo::NTHR) phc[i] = p.ktab[i];
    auto phase_nk = [&](int ph) { return __builtin_amdgcn_readfirstlane(phc[ph * 8]); };

    // tile t of the list -> (row tile, phase, column tile); the phase of slot s rotates with the row tile (see the header)
#define C8_DECODE(T, BM_, PH_, BN_)                                            \
    {                                                                          \
        BM_ = (T) / tpb;                                                       \
        const int r_ = (T)-BM_ * tpb, s_ = r_ / tiles_n;                       \
        BN_ = r_ - s_ * tiles_n;                                               \
        PH_ = p.nphase == 1 ? 0 : ((s_ + (BM_ * tpb) / nbx_nom) & 3);          \
    }
    int Gtot = 0;
    for (int i = 0; i < my_tiles; ++i) {
        int bm_, ph_, bn_;
        C8_DECODE(tlo + jx + i * nbx, bm_, ph_, bn_)
        Gtot += p.ktab[ph_ * 8] * NSEG;  // (paired: two K-tiles of 32 per table K-tile of 64)
    }
    // NSEG = 2: the split precision mode with PAIRED K-tiles (gemm8.hip): a K-tile covers 32 reduction elements, its 128-byte LDS row is
    // [hi | lo]; the lo lanes of a DMA instruction add the distance of the lo tensor to their offset.  For the gathered operand that needs
    // ONE buffer descriptor over both tensors (lo above hi, the host checks the distance), and the "off the image" offset moves above it.
    constexpr bool PAIR = NSEG == 2;
    const unsigned dA = PAIR ? (unsigned)((const char*)p.a[1] - (const char*)p.a[0]) : 0u;
    const i32x4 rs_hi = c8_rsrc(p.a[0], PAIR ? dA + p.a_bytes : p.a_bytes);
    const i32x4 rs_lo = c8_rsrc(NSEG == 3 ? p.a[1] : p.a[0], p.a_bytes);
    constexpr unsigned OOB = PAIR ? 0xfffffff0u : 0x80000000u;

    // fragment read offsets inside a half-tile (k-substep 1: ^ 64)
    const int sw = ((lane >> 4) ^ (lane & 7)) << 4;
    const int aoff = (wr * MH + (lane & 15)) * 128 + sw;
    const int boff0 = OFF_B0 + (wc * NT0 * 16 + (lane & 15)) * 128 + sw;
    const int boff1 = OFF_B1 + (wc * NT1 * 16 + (lane & 15)) * 128 + sw;
    // LDS-DMA lane constants
    const int sc = (lane & 7) ^ ((lane >> 3) & 7);  // source chunk of this lane's LDS slot (bank swizzle on the source side)
    const unsigned c16 = PAIR ? (unsigned)((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.b[1] - (const char*)p.b[0]) : 0u) : (unsigned)(sc << 4);
    const unsigned a_lo_add = PAIR && sc >= 4 ? dA : 0u;
    const int tsc = PAIR ? (sc & 3) : sc;  // table entry of this lane inside a K-tile's group of TPK entries
    constexpr int TPK = PAIR ? 4 : 8;
    const unsigned ldsw_a = lds_base + wave * (AI * 1024);
    const unsigned ldsw_b0 = lds_base + OFF_B0 + wave * (NT0 * 1024);
    const unsigned ldsw_b1 = lds_base + OFF_B1 + wave * (NT1 * 1024);
    int arow[AI];  // tile row of A piece i of half 0 (half 1: + MH)
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int hl = wave * (AI * 8) + i * 8 + (lane >> 3);
        arow[i] = (hl / MH) * (2 * MH) + hl % MH;
    }
    int bcol0[NT0], bcol1[NT1 > 0 ? NT1 : 1];  // tile column of B piece i of half 0 / 1
#pragma unroll
    for (int i = 0; i < NT0; ++i) {
        const int hl = wave * (NT0 * 8) + i * 8 + (lane >> 3);
        bcol0[i] = (hl / (NT0 * 16)) * WN + hl % (NT0 * 16);
    }
#pragma unroll
    for (int i = 0; i < NT1; ++i) {
        const int hl = wave * (NT1 * 8) + i * 8 + (lane >> 3);
        bcol1[i] = (hl / (NT1 * 16)) * WN + NT0 * 16 + hl % (NT1 * 16);
    }
    const int Hs = p.H * p.sm, Ws = p.W * p.sm;

    CurA<AI> cA0, cA1;
    CurB cB;  // both B half-tiles of a K-tile are always issued together: one cursor
#define C8_REBASE_A(CUR, HF)                                                                                      \
    {                                                                                                          \
        int bm_, ph_, bn_;                                                                                     \
        C8_DECODE(tlo + jx + (CUR).i * nbx, bm_, ph_, bn_)                                                       \
        (CUR).nk = phase_nk(ph_) * (PAIR ? 2 : 1), (CUR).toff = __builtin_amdgcn_readfirstlane(phc[ph_ * 8 + 1]);                     \
        _Pragma("unroll") for (int i_ = 0; i_ < AI; ++i_) {                                                    \
            const int m_ = bm_ * BM + arow[i_] + (HF)*MH;                                                       \
            const int b_ = p.f_hw.div(m_), rem_ = m_ - b_ * (p.H * p.W);                                       \
            const int y_ = p.f_w.div(rem_), x_ = rem_ - y_ * p.W;                                              \
            const int sy_ = y_ * p.sm, sx_ = x_ * p.sm;                                                        \
            unsigned mk_ = 0;                                                                                  \
            if (m_ < p.M) {                                                                                    \
                for (int t_ = 0; t_ < p.nbits; ++t_) {                                                         \
                    const int dy_ = (int)((p.ddy_code >> (2 * t_)) & 3u) - 1, dx_ = (int)((p.ddx_code >> (2 * t_)) & 3u) - 1; \
                    const bool v_ = ((unsigned)(sy_ + dy_) < (unsigned)Hs) & ((unsigned)(sx_ + dx_) < (unsigned)Ws); \
                    mk_ |= (v_ ? 1u : 0u) << t_;                                                               \
                }                                                                                              \
            }                                                                                                  \
            (CUR).mask[i_] = mk_;                                                                                \
            (CUR).rowoff[i_] = (unsigned)((b_ * Hs + sy_) * Ws + sx_) * (unsigned)(p.C * 2);                     \
        }                                                                                                      \
    }
#define C8_REBASE_B(CUR)                                                                                         \
    {                                                                                                          \
        int bm_, ph_, bn_;                                                                                     \
        C8_DECODE(tlo + jx + (CUR).i * nbx, bm_, ph_, bn_)                                                       \
        (CUR).nk = phase_nk(ph_) * (PAIR ? 2 : 1);                                                               \
        (CUR).ldb2 = __builtin_amdgcn_readfirstlane(phc[ph_ * 8 + 2]) * 2;                                       \
        const long bo_ = (long)(unsigned)__builtin_amdgcn_readfirstlane(phc[ph_ * 8 + 3]) |                      \
                         ((long)__builtin_amdgcn_readfirstlane(phc[ph_ * 8 + 4]) << 32);                         \
        const bf16_t* b_ = (NSEG == 3 && (CUR).seg == 1) ? p.b[1] : p.b[0];                                      \
        (CUR).base = (const char*)(b_ + bo_) + (long)bn_ * BN * (CUR).ldb2;                                      \
        (CUR).vr = min(BN, p.N - bn_ * BN); /* ragged last column tile: rows past N re-read the last valid one */ \
    }
#define C8_ADVANCE(CUR, REBASE)                                       \
    if (++(CUR).kt == (CUR).nk) {                                       \
        (CUR).kt = 0;                                                 \
        if (NSEG != 3 || ++(CUR).seg == NSEG) (CUR).seg = 0, (CUR).i++;   \
        REBASE                                                      \
    }
    // the two LDS-DMA instructions of this wave for A half H of the cursor's K-tile into buffer BUFI
#define C8_ISSUE_A(CUR, HF, BUFI)                                                                                 \
    if ((CUR).i < my_tiles) {                                                                                  \
        const int e_ = (CUR).e;                                                                                  \
        const int delta_ = (e_ >> 8) * 16;                                                                      \
        const unsigned bit_ = (unsigned)e_ & 31u;                                                              \
        const i32x4 rs_ = (NSEG == 3 && (CUR).seg == 2) ? rs_lo : rs_hi;                                         \
        _Pragma("unroll") for (int i_ = 0; i_ < AI; ++i_) {                                                    \
            const bool ok_ = ((CUR).mask[i_] >> bit_) & 1u;                                                      \
            const unsigned voff_ = ok_ ? (CUR).rowoff[i_] + (unsigned)delta_ + a_lo_add : OOB;                   \
            c8_blds16(voff_, rs_, ldsw_a + (BUFI)*BUF + (HF)*HALF_A + i_ * 1024);                               \
        }                                                                                                      \
        C8_ADVANCE(CUR, C8_REBASE_A(CUR, HF))                                                                       \
        (CUR).e = tab[(CUR).toff + (CUR).kt * TPK + tsc];                                                            \
    }
    // B pieces of the cursor's K-tile into buffer BUFI in two parts (as gemm8.hip, schedule 4): part 0 in the read phase of big phase 2,
    // part 1 behind that phase's MFMAs.  Only the 256-wide instances (two column halves of two pieces each) move their second half
    // there: convT dgrad 768 -> 384 244 -> 217 us; with one piece to move (256 x 192: 24 MFMAs per phase) the tail made the instances
    // 1-3 % slower, so they keep all of B in the read phase (same-box A/B of two builds, tools/head_bench.py).
    constexpr bool B_TAIL = NT1 >= 2;
#define C8_ISSUE_B(CUR, BUFI, PART)                                                                            \
    if ((CUR).i < my_tiles) {                                                                                  \
        if ((PART) == 0) {                                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < NT0; ++i_)                                                 \
                c8_glds16_s((unsigned)__mul24(min(bcol0[i_], (CUR).vr - 1), (CUR).ldb2) + c16, (CUR).base, ldsw_b0 + (BUFI)*BUF + i_ * 1024); \
        }                                                                                                      \
        if ((PART) == (B_TAIL ? 1 : 0)) {                                                                      \
            _Pragma("unroll") for (int i_ = 0; i_ < NT1; ++i_)                                                 \
                c8_glds16_s((unsigned)__mul24(min(bcol1[i_], (CUR).vr - 1), (CUR).ldb2) + c16, (CUR).base, ldsw_b1 + (BUFI)*BUF + i_ * 1024); \
            (CUR).base += PAIR ? 64 : 128;                                                                     \
            C8_ADVANCE(CUR, C8_REBASE_B(CUR))                                                                  \
        }                                                                                                      \
    }

    f32x4 acc[2][NTW][MT];  // [h][nt][mt]
    int tile_i = 0;         // index of the tile being accumulated in this workgroup's list
    int c_bm, c_ph, c_bn;
    C8_DECODE(tlo + jx, c_bm, c_ph, c_bn)
    int per_tile2 = phase_nk(c_ph) * NSEG / 2;
#define C8_INIT_ACC()                                                                                            \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_)       \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) acc[h_][nt_][mt_] = f32x4{0.f, 0.f, 0.f, 0.f};
    C8_INIT_ACC()

    __syncthreads();  // the chunk table is in LDS
    // prologue: K-tile 0 complete + A0, B0, B1 of K-tile 1 (the state the steady-state schedule leaves behind)
    cA0.kt = cA1.kt = cB.kt = 0;
    cA0.seg = cA1.seg = cB.seg = 0;
    cA0.i = cA1.i = cB.i = 0;
    C8_REBASE_A(cA0, 0)
    C8_REBASE_A(cA1, 1)
    C8_REBASE_B(cB)
    cA0.e = tab[cA0.toff + tsc];
    cA1.e = cA0.e;
    C8_ISSUE_A(cA0, 0, 0)
    C8_ISSUE_B(cB, 0, 0)
    C8_ISSUE_B(cB, 0, 1)
    C8_ISSUE_A(cA1, 1, 0)
    C8_ISSUE_A(cA0, 0, 1)
    C8_ISSUE_B(cB, 1, 0)
    C8_ISSUE_B(cB, 1, 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW - AI) : "memory");
    asm volatile("s_barrier" ::: "memory");

    bf16x8_t af[MT][2], bfr[NTW][2];
#define C8_READ_A(BUFI, HF)                                                                                     \
    _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)       \
        af[mt_][s_] = *reinterpret_cast<const bf16x8_t*>(smem + (BUFI)*BUF + (HF)*HALF_A + mt_ * 2048 + (aoff ^ (s_ * 64)));
#define C8_READ_B(BUFI)                                                                                         \
    _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)      \
        bfr[nt_][s_] = *reinterpret_cast<const bf16x8_t*>(                                                      \
            smem + (BUFI)*BUF + (nt_ < NT0 ? nt_ * 2048 + (boff0 ^ (s_ * 64)) : (nt_ - NT0) * 2048 + (boff1 ^ (s_ * 64))));
    // one "big phase": all MFMAs of A half H between one barrier pair; the reads were retired BEFORE the first barrier
    // (lgkmcnt(0)), so a half-tile may be refilled in the very next phase
#define C8_MFMA(HF, TAIL)                                                                                       \
    {                                                                                                           \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                      \
        asm volatile("s_barrier" ::: "memory");                                                                 \
        __builtin_amdgcn_s_setprio(1);                                                                          \
        if constexpr (PAIR) { /* fragments [0] = hi, [1] = lo: hi hi, hi(B) lo(A), lo(B) hi(A) -- conv4's order */                 \
            _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_) \
                _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_)                                            \
                    acc[HF][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt_][t_ == 2], af[mt_][t_ == 1], acc[HF][nt_][mt_], 0, 0, 0); \
        } else {                                                                                                \
        _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_) _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                    \
                acc[HF][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt_][s_], af[mt_][s_], acc[HF][nt_][mt_], 0, 0, 0); \
        }                                                                                                       \
        __builtin_amdgcn_s_setprio(0);                                                                          \
        TAIL                                                                                                    \
        asm volatile("s_barrier" ::: "memory");                                                                 \
    }
    // B_TAIL instances: the wait stands BEFORE its phase's issues: R1 retires A1(t) with the AI + NT0 + NT1 younger operations of
    // A0 B (t + 1) in flight, R2 retires A0 B (t + 1) with the AI of A1(t + 1); last iteration: VMW - AI, AI, 0, 0.  The others keep the
    // round-4 order (issue, then wait with VMW operations in flight; last iteration: VMW, AI, 0, 0).
#define C8_WAIT(CNT, LASTCNT)                                                                \
    {                                                                                        \
        if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LASTCNT) : "memory");             \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CNT) : "memory");                      \
    }

    int it_c = 0;  // iterations (K-tile pairs) done of the current tile
    const int iters = Gtot >> 1;
    bool staggered = false;
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
        if (!staggered) {  // (re-)establish the stagger: group 1 runs one barrier behind group 0
            if (wr == 1) asm volatile("s_barrier" ::: "memory");
            staggered = true;
        }
        // two big phases per K-tile: BP1 reads A0 B0 B1 + issues A1 of the next K-tile; BP2 reads A1 + issues A0 and the first part of B of
        // K-tile + 2 into the slots BP1 has just retired, the second part of B behind its MFMAs (the read phases bound the intervals)
        C8_READ_B(0)
        C8_READ_A(0, 0)
        if constexpr (B_TAIL) C8_WAIT(VMW - AI, VMW - AI)
        C8_ISSUE_A(cA1, 1, 1)
        if constexpr (!B_TAIL) C8_WAIT(VMW, VMW)
        C8_MFMA(0, )
        C8_READ_A(0, 1)
        if constexpr (B_TAIL) C8_WAIT(AI, AI)
        C8_ISSUE_A(cA0, 0, 0)
        C8_ISSUE_B(cB, 0, 0)
        if constexpr (!B_TAIL) C8_WAIT(VMW, AI)
        C8_MFMA(1, C8_ISSUE_B(cB, 0, 1))
        C8_READ_B(1)
        C8_READ_A(1, 0)
        if constexpr (B_TAIL) C8_WAIT(VMW - AI, 0)
        C8_ISSUE_A(cA1, 1, 0)
        if constexpr (!B_TAIL) C8_WAIT(VMW, 0)
        C8_MFMA(0, )
        C8_READ_A(1, 1)
        if constexpr (B_TAIL) C8_WAIT(AI, 0)
        C8_ISSUE_A(cA0, 0, 1)
        C8_ISSUE_B(cB, 1, 0)
        if constexpr (!B_TAIL) C8_WAIT(VMW, 0)
        C8_MFMA(1, C8_ISSUE_B(cB, 1, 1))
        if (++it_c < per_tile2) continue;
        // ================= tile finished: epilogue (the next tile's first K-tiles are in flight) =================
        it_c = 0;
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // re-align the groups: both epilogues run concurrently
        staggered = false;
        const int bm = c_bm, ph = c_ph, bn = c_bn;
        ++tile_i;
        if (!last) {
            C8_DECODE(tlo + jx + tile_i * nbx, c_bm, c_ph, c_bn)
            per_tile2 = phase_nk(c_ph) * NSEG / 2;
        }
        char* st = smem + OFF_STAGE + wave * STG;
        const int erow = lane & 15, eq = lane >> 4;
        const int n0 = bn * BN + wc * WN;
        const int py = ph >> 1, px = ph & 1;
        f32x4 bias4[NTW];  // bias in MFMA layout (4 consecutive n per lane): live during the epilogue only
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            bias4[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n0 + nt * 16 + 4 * eq < p.N) {
                const float4 b_ = *reinterpret_cast<const float4*>(p.bias + n0 + nt * 16 + 4 * eq);
                bias4[nt] = f32x4{b_.x, b_.y, b_.z, b_.w};
            }
        }
        // output row of tile row m: identity, or the sub-pixel phase map of the ConvTranspose forward
        auto out_row = [&](int m) -> long {
            if (!p.phase_map) return (long)m;
            const int b_ = p.f_hw.div(m), rem_ = m - b_ * (p.H * p.W);
            const int y_ = p.f_w.div(rem_), x_ = rem_ - y_ * p.W;
            return ((long)(b_ * 2 * p.H + 2 * y_ + py)) * (2 * p.W) + 2 * x_ + px;
        };
        constexpr int UPR = WN / 8, NU = 16 * UPR, NIT = (NU + 63) / 64;  // 16-byte units per staged row / per pass
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m0 = bm * BM + wr * (2 * MH) + h * MH + mt * 16;
                const long orow_e = p.drop_thresh ? out_row(m0 + erow) : 0L;
#pragma unroll
                for (int nt = 0; nt < NTW; ++nt) {
                    const f32x4 a = acc[h][nt][mt] + bias4[nt];
                    float v[4] = {a[0], a[1], a[2], a[3]};
                    if (p.scale && n0 + nt * 16 + 4 * eq < p.N) {  // eval-mode BatchNorm + ReLU (inference only: the constants come from L1 per use, no registers held)
                        const float4 s4 = *reinterpret_cast<const float4*>(p.scale + n0 + nt * 16 + 4 * eq);
                        const float4 t4 = *reinterpret_cast<const float4*>(p.shift + n0 + nt * 16 + 4 * eq);
                        v[0] = fmaxf(v[0] * s4.x + t4.x, 0.f), v[1] = fmaxf(v[1] * s4.y + t4.y, 0.f);
                        v[2] = fmaxf(v[2] * s4.z + t4.z, 0.f), v[3] = fmaxf(v[3] * s4.w + t4.w, 0.f);
                    }
                    if (p.drop_thresh) {
                        float mk[4];
                        dropout_scale4(drop_seed, (uint32_t)(orow_e * p.ldo + n0 + nt * 16 + 4 * eq), p.drop_thresh, p.drop_inv, mk);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] *= mk[j];
                    }
                    uint2 po;
                    po.x = pack_bf2(v[0], v[1]), po.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(st + erow * PITCH + nt * 32 + eq * 8) = po;
                    if constexpr (SPLIT_OUT) {
                        uint2 pl;
                        pl.x = pack_bf2(v[0] - __uint_as_float(po.x << 16), v[1] - __uint_as_float(po.x & 0xffff0000u));
                        pl.y = pack_bf2(v[2] - __uint_as_float(po.y << 16), v[3] - __uint_as_float(po.y & 0xffff0000u));
                        *reinterpret_cast<uint2*>(st + 16 * PITCH + erow * PITCH + nt * 32 + eq * 8) = pl;
                    }
                }
#pragma unroll
                for (int itu = 0; itu < NIT; ++itu) {
                    const int u = itu * 64 + lane;
                    if (NU % 64 == 0 || u < NU) {
                        const int r = u / UPR, ch = u - r * UPR;
                        const uint4 ux = *reinterpret_cast<const uint4*>(st + r * PITCH + ch * 16);
                        uint4 ul = ux;
                        if constexpr (SPLIT_OUT) ul = *reinterpret_cast<const uint4*>(st + 16 * PITCH + r * PITCH + ch * 16);
                        const int m = m0 + r;
                        if (m < p.M && n0 + ch * 8 < p.N) {
                            const size_t o = (size_t)out_row(m) * p.ldo + n0 + ch * 8;
                            *reinterpret_cast<uint4*>(p.out_hi + o) = ux;
                            if constexpr (SPLIT_OUT) *reinterpret_cast<uint4*>(p.out_lo + o) = ul;
                        }
                    }
                }
            }
        C8_INIT_ACC()
    }
    if (staggered && wr == 0) asm volatile("s_barrier" ::: "memory");  // (unreachable in practice: every tile ends re-aligned)
#undef C8_WAIT
#undef C8_MFMA
#undef C8_READ_A
#undef C8_READ_B
#undef C8_ISSUE_A
#undef C8_ISSUE_B
#undef C8_ADVANCE
#undef C8_REBASE_A
#undef C8_REBASE_B
#undef C8_INIT_ACC
#undef C8_DECODE
}

// =====================================================================================================================================
// conv4: the same implicit GEMM on FOUR waves, one per SIMD, 256 x 192 tile, wave (wr, wc) owns 128 rows x 96 columns in 192 accumulator
// registers; the K-loop of a tile is one generated assembly block (gen_gemm4.py, the "c" form: gemm4.hip's ring of three A slots + two
// B stages, the A pieces gathered through the buffer descriptor with three vector instructions of address arithmetic each).  Conv2d forward /
// data gradient, ConvTranspose forward (four phases) / data gradient, plain bf16 operands, widths that tile by 192.  Same packed
// weights, chunk table and epilogue features as conv8_kernel, and the same results bit for bit (same MFMA, same K order).
#include "gemm4_gen.inc"
constexpr int C4_OFF_TAB = 147456;  // 144 KiB: 3 x 32 KiB A slots, 2 x 24 KiB B stages, then the chunk table (32-word header + entries)
constexpr int C4_TAB_MAX = 4096 - 32;

// NI: 16-column blocks per wave: 6 = the 256 x 192 tile, 3 = 256 x 96 (N = 96, 288).  PAIR: the split precision mode (bf16x3) on paired K-tiles
// (32 reduction elements, LDS row = [hi | lo], three products per K-tile: gen_gemm4.py "cp" form; operands and outputs are (hi, lo) pairs with
// lo a 32-bit distance above hi, as conv8_kernel's NSEG = 2).
template <int NI, bool PAIR = false>
__global__ __launch_bounds__(256) void conv4_kernel(C8Params p) {
    static_assert(NI == 6 || NI == 3, "conv4: 256 x 192 or 256 x 96 tiles");
    constexpr int BM = 256, BN = 32 * NI, WNC = 16 * NI, PITCH = WNC * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM, tpb = tiles_n * p.nphase, ntiles = tiles_m * tpb;  // (a ragged last column tile: N = 144)
    // persistent, XCD-aware (as conv8_kernel): XCD x owns a contiguous tile range; the column tiles and phases of a row tile are neighbours
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0), nbx_nom = max(nb >> 3, 1);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    if (my_tiles <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    int* phc = reinterpret_cast<int*>(smem + C4_OFF_TAB);  // 32-word header of per-phase constants, then the chunk table
    for (int i = tid; i < p.ktab_n + 32; i += 256) phc[i] = p.ktab[i];
    const unsigned dA = PAIR ? (unsigned)((const char*)p.a[1] - (const char*)p.a[0]) : 0u;  // one descriptor spans hi and lo (the host checks)
    const unsigned abytes = p.a_bytes + dA;
    const char* abase = (const char*)p.a[0];
    // lane constants of the asm blocks (gen_gemm4.py: c_setup / cp_setup)
    const unsigned sc = (lane & 7) ^ ((lane >> 3) & 7);  // source chunk of this lane's LDS chunk; paired: chunks 0-3 = hi, 4-7 = lo of the same 32 elements
    const unsigned c16 = PAIR ? ((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.b[1] - (const char*)p.b[0]) : 0u) : sc << 4;
    const unsigned aloadd = PAIR && sc >= 4 ? dA : 0u;
    const unsigned swz = ((lane >> 4) ^ (lane & 7)) << 4;
    const unsigned fa = lds_base + (wr * 128 + (lane & 15)) * 128 + swz;
    const unsigned fb = lds_base + (wc * WNC + (lane & 15)) * 128 + swz;
    const unsigned ldsw = lds_base + wave * 8192, ldswb = lds_base + wave * (NI * 1024);
    const unsigned browv = wave * (NI * 8) + (lane >> 3);
    const unsigned vtl = lds_base + C4_OFF_TAB + 128 + (PAIR ? sc & 3 : sc) * 4;  // this lane's entry of table position 0
    const int Hs = p.H * p.sm, Ws = p.W * p.sm;
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;

    // tile t of the list -> (row tile, phase, column tile); the phase of slot s rotates with the row tile (conv8_kernel's order)
    auto decode_tile = [&](int t, int& bm, int& ph, int& bn) {
        bm = t / tpb;
        const int r = t - bm * tpb, s_ = r / tiles_n;
        bn = r - s_ * tiles_n;
        ph = p.nphase == 1 ? 0 : ((s_ + (bm * tpb) / nbx_nom) & 3);
    };
    // per-phase constants: K-tiles, byte offset of the phase's table, packed row pitch in bytes, the phase's packed block
    struct PhaseK { int nk, toff4, ldb2; const char* b; };
    auto phase_k = [&](int ph) {
        PhaseK k;
        k.nk = __builtin_amdgcn_readfirstlane(phc[ph * 8]) * (PAIR ? 2 : 1);  // (paired: two 32-element K-tiles per table K-tile)
        k.toff4 = __builtin_amdgcn_readfirstlane(phc[ph * 8 + 1]) * 4;
        k.ldb2 = __builtin_amdgcn_readfirstlane(phc[ph * 8 + 2]) * 2;
        const long bo = (long)(unsigned)__builtin_amdgcn_readfirstlane(phc[ph * 8 + 3]) | ((long)__builtin_amdgcn_readfirstlane(phc[ph * 8 + 4]) << 32);
        k.b = (const char*)(p.b[0] + bo);
        return k;
    };
    auto uni = [](const char* q) {  // a wave-uniform pointer the compiler can keep in scalar registers (the asm blocks take it as "s")
        const uint64_t v = (uint64_t)q;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (const char*)(((uint64_t)hi << 32) | lo);
    };
    // rows of A piece i of this wave: tile row wave*64 + i*8 + (lane >> 3): byte offset of its source pixel, INVERTED tap mask (bit t set:
    // tap t falls off the image; bit 31 always set: padding chunks; all ones for rows past M)
    auto decode_rows = [&](int bm, unsigned (&ro)[8], unsigned (&im)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = bm * BM + wave * 64 + i * 8 + (lane >> 3);
            const int b_ = p.f_hw.div(m), rem = m - b_ * (p.H * p.W);
            const int y = p.f_w.div(rem), x = rem - y * p.W;
            const int sy = y * p.sm, sx = x * p.sm;
            unsigned mk = 0;
            if (m < p.M) {
                for (int t = 0; t < p.nbits; ++t) {
                    const int dy = (int)((p.ddy_code >> (2 * t)) & 3u) - 1, dx = (int)((p.ddx_code >> (2 * t)) & 3u) - 1;
                    const bool v = ((unsigned)(sy + dy) < (unsigned)Hs) & ((unsigned)(sx + dx) < (unsigned)Ws);
                    mk |= (v ? 1u : 0u) << t;
                }
            }
            im[i] = ~mk;
            ro[i] = m < p.M ? (unsigned)((b_ * Hs + sy) * Ws + sx) * (unsigned)(p.C * 2) : 0u;
        }
    };
    __syncthreads();  // the chunk table is in LDS

    unsigned a0 = 0, a1 = 32768, a2 = 65536;  // A ring (rotated by the asm blocks)
    unsigned ro[8], im[8], ron[8], imn[8];
    int tile = tlo + jx;
    int c_bm, c_ph, c_bn;
    decode_tile(tile, c_bm, c_ph, c_bn);
    decode_rows(c_bm, ro, im);
    {
        const PhaseK k = phase_k(c_ph);
        const int vrb = min(BN, p.N - c_bn * BN) - 1;
        const char* bptr = uni(k.b + (long)c_bn * BN * k.ldb2);
#define C4_PRO_OPERANDS ::[abase] "s"(abase), [abytes] "s"(abytes), [bptr] "s"(bptr), [ldb2] "s"(k.ldb2), [vrb] "s"(vrb), [ldsw] "s"(ldsw), \
                     [ldswb] "s"(ldswb), [nk] "s"(k.nk), [toff4] "s"(k.toff4), [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [browv] "v"(browv), \
                     [c16] "v"(c16), [c16b] "v"(c16), [aloadd] "v"(aloadd), [fa] "v"(fa), [fb] "v"(fb), [vtl] "v"(vtl), [ro0] "v"(ro[0]), [ro1] "v"(ro[1]), [ro2] "v"(ro[2]), [ro3] "v"(ro[3]), \
                     [ro4] "v"(ro[4]), [ro5] "v"(ro[5]), [ro6] "v"(ro[6]), [ro7] "v"(ro[7]), [im0] "v"(im[0]), [im1] "v"(im[1]), [im2] "v"(im[2]), \
                     [im3] "v"(im[3]), [im4] "v"(im[4]), [im5] "v"(im[5]), [im6] "v"(im[6]), [im7] "v"(im[7])
        if constexpr (PAIR && NI == 6) asm volatile(G4CP6_ASM_PROLOGUE C4_PRO_OPERANDS : G4CP6_CLOBBERS);
        else if constexpr (PAIR) asm volatile(G4CP3_ASM_PROLOGUE C4_PRO_OPERANDS : G4CP3_CLOBBERS);
        else if constexpr (NI == 6) asm volatile(G4C6_ASM_PROLOGUE C4_PRO_OPERANDS : G4C6_CLOBBERS);
        else asm volatile(G4C3_ASM_PROLOGUE C4_PRO_OPERANDS : G4C3_CLOBBERS);
#undef C4_PRO_OPERANDS
    }
    for (int t = 0; t < my_tiles; ++t) {
        const int bm = c_bm, ph = c_ph, bn = c_bn;
        const bool more = t + 1 < my_tiles;
        const int tn = more ? tile + nbx : tile;
        decode_tile(tn, c_bm, c_ph, c_bn);
        decode_rows(c_bm, ron, imn);
        if (!more) {  // no next tile: the last two DMA rounds fill zeros
#pragma unroll
            for (int i = 0; i < 8; ++i) imn[i] = 0xffffffffu;
        }
        const PhaseK k = phase_k(ph), kn = phase_k(c_ph);
        const char* bptr = uni(k.b + (long)bn * BN * k.ldb2 + (PAIR ? 128 : 256));  // K-tile 2 (0 and 1 are in flight)
        const char* bnext = uni(kn.b + (long)c_bn * BN * kn.ldb2);
        const int npair = (k.nk >> 1) - 2;
        const int vrb = min(BN, p.N - bn * BN) - 1, vrbn = min(BN, p.N - c_bn * BN) - 1;
#define C4_TILE_OPERANDS \
                     : [a0] "+s"(a0), [a1] "+s"(a1), [a2] "+s"(a2) \
                     : [abase] "s"(abase), [abytes] "s"(abytes), [bptr] "s"(bptr), [bnext] "s"(bnext), [ldb2] "s"(k.ldb2), [ldb2n] "s"(kn.ldb2), [vrb] "s"(vrb), [vrbn] "s"(vrbn), \
                       [ldsw] "s"(ldsw), [ldswb] "s"(ldswb), [nk] "s"(k.nk), [toff4] "s"(k.toff4), [toffn4] "s"(kn.toff4), [npair] "s"(npair), \
                       [browv] "v"(browv), [c16] "v"(c16), [c16b] "v"(c16), [aloadd] "v"(aloadd), [fa] "v"(fa), [fb] "v"(fb), [vtl] "v"(vtl), [ro0] "v"(ro[0]), [ro1] "v"(ro[1]), \
                       [ro2] "v"(ro[2]), [ro3] "v"(ro[3]), [ro4] "v"(ro[4]), [ro5] "v"(ro[5]), [ro6] "v"(ro[6]), [ro7] "v"(ro[7]), [im0] "v"(im[0]), \
                       [im1] "v"(im[1]), [im2] "v"(im[2]), [im3] "v"(im[3]), [im4] "v"(im[4]), [im5] "v"(im[5]), [im6] "v"(im[6]), [im7] "v"(im[7]), \
                       [ron0] "v"(ron[0]), [ron1] "v"(ron[1]), [ron2] "v"(ron[2]), [ron3] "v"(ron[3]), [ron4] "v"(ron[4]), [ron5] "v"(ron[5]), \
                       [ron6] "v"(ron[6]), [ron7] "v"(ron[7]), [imn0] "v"(imn[0]), [imn1] "v"(imn[1]), [imn2] "v"(imn[2]), [imn3] "v"(imn[3]), \
                       [imn4] "v"(imn[4]), [imn5] "v"(imn[5]), [imn6] "v"(imn[6]), [imn7] "v"(imn[7])
        if constexpr (PAIR && NI == 6) asm volatile(G4CP6_ASM_TILE C4_TILE_OPERANDS : G4CP6_CLOBBERS);
        else if constexpr (PAIR) asm volatile(G4CP3_ASM_TILE C4_TILE_OPERANDS : G4CP3_CLOBBERS);
        else if constexpr (NI == 6) asm volatile(G4C6_ASM_TILE C4_TILE_OPERANDS : G4C6_CLOBBERS);
        else asm volatile(G4C3_ASM_TILE C4_TILE_OPERANDS : G4C3_CLOBBERS);
#undef C4_TILE_OPERANDS
        // ================= epilogue (the next tile's K-tiles 0 and 1 are in flight) =================
        // accumulator block (mi, ni) of this lane: out[m = mi 16 + (lane & 15)][n = ni 16 + 4 (lane >> 4) .. + 3] of the wave's 128 x 16 NI
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        char* st = smem + a2 + wave * 8192;  // the A slot this tile's last K-tile has left
        const int erow = lane_e & 15, eq = lane_e >> 4;
        const int n0 = bn * BN + wc * WNC, mw = bm * BM + wr * 128;
        const int py = ph >> 1, px = ph & 1;
        // output row of tile row m: identity, or the sub-pixel phase map of the ConvTranspose forward
        auto out_row = [&](int m) -> long {
            if (!p.phase_map) return (long)m;
            const int b_ = p.f_hw.div(m), rem_ = m - b_ * (p.H * p.W);
            const int y_ = p.f_w.div(rem_), x_ = rem_ - y_ * p.W;
            return ((long)(b_ * 2 * p.H + 2 * y_ + py)) * (2 * p.W) + 2 * x_ + px;
        };
        f32x4 bias4[NI];
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            bias4[ni] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (p.bias && n0 + ni * 16 + 4 * eq < p.N) {
                const float4 b_ = *reinterpret_cast<const float4*>(p.bias + n0 + ni * 16 + 4 * eq);
                bias4[ni] = f32x4{b_.x, b_.y, b_.z, b_.w};
            }
        }
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            f32x4 tt[8];
            g4_acc_row(mi, tt);
            const int m0 = mw + mi * 16;
            const long orow_e = p.drop_thresh ? out_row(m0 + erow) : 0L;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                const f32x4 a = tt[ni] + bias4[ni];
                float v[4] = {a[0], a[1], a[2], a[3]};
                if (p.scale && n0 + ni * 16 + 4 * eq < p.N) {  // eval-mode BatchNorm + ReLU (inference only)
                    const float4 s4 = *reinterpret_cast<const float4*>(p.scale + n0 + ni * 16 + 4 * eq);
                    const float4 t4 = *reinterpret_cast<const float4*>(p.shift + n0 + ni * 16 + 4 * eq);
                    v[0] = fmaxf(v[0] * s4.x + t4.x, 0.f), v[1] = fmaxf(v[1] * s4.y + t4.y, 0.f);
                    v[2] = fmaxf(v[2] * s4.z + t4.z, 0.f), v[3] = fmaxf(v[3] * s4.w + t4.w, 0.f);
                }
                if (p.drop_thresh) {
                    float mk[4];
                    dropout_scale4(drop_seed, (uint32_t)(orow_e * p.ldo + n0 + ni * 16 + 4 * eq), p.drop_thresh, p.drop_inv, mk);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] *= mk[j];
                }
                uint2 po;
                po.x = pack_bf2(v[0], v[1]), po.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(st + erow * PITCH + ni * 32 + eq * 8) = po;
                if constexpr (PAIR) {  // split output: lo = bf16(v - hi) in the second slab
                    uint2 pl;
                    pl.x = pack_bf2(v[0] - __uint_as_float(po.x << 16), v[1] - __uint_as_float(po.x & 0xffff0000u));
                    pl.y = pack_bf2(v[2] - __uint_as_float(po.y << 16), v[3] - __uint_as_float(po.y & 0xffff0000u));
                    *reinterpret_cast<uint2*>(st + 16 * PITCH + erow * PITCH + ni * 32 + eq * 8) = pl;
                }
            }
            constexpr int UPR = NI * 2, NU = 16 * UPR;  // sixteen-byte units per staged row, per row block
#pragma unroll
            for (int itu = 0; itu < (NU + 63) / 64; ++itu) {
                const int u = itu * 64 + lane_e;
                if (NU % 64 == 0 || u < NU) {
                    const int r = u / UPR, ch = u - r * UPR;
                    const uint4 ux = *reinterpret_cast<const uint4*>(st + r * PITCH + ch * 16);
                    uint4 ul = ux;
                    if constexpr (PAIR) ul = *reinterpret_cast<const uint4*>(st + 16 * PITCH + r * PITCH + ch * 16);
                    const int m = m0 + r;
                    if (m < p.M && n0 + ch * 8 < p.N) {
                        const size_t o = (size_t)out_row(m) * p.ldo + n0 + ch * 8;
                        *reinterpret_cast<uint4*>(p.out_hi + o) = ux;
                        if constexpr (PAIR) *reinterpret_cast<uint4*>(p.out_lo + o) = ul;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) ro[i] = ron[i], im[i] = imn[i];
        tile = tn;
    }
}

// ---- weight packing + chunk table (one launch per convolution call) -------------------------------------------------------------
struct C8Plan {
    int nphase;
    int ntaps[4];
    int tap[4][9];   // weight tap (ky 3 + kx) of K position tl
    int bit[4][9];   // row-mask bit of K position tl
    int kpad[4];     // packed row length (multiple of 128)
    long boff[4];    // element offset of the phase block
    int toff[4];     // first table entry
    int ddy[9], ddx[9];  // displacement of mask bit b in source-grid pixels
    int N, C;            // packed rows, channels per tap
    int Ws;              // source grid width
    int transposed;      // 0: Wp[n = co][tl C + ci] = Wc[co][tap][ci];  1: Wp[n = ci][tl C + co] = Wc[co][tap][ci]
    int rows_total;      // nphase * N
    int ktab_n;
};

// chunk table behind its 32-word header of per-phase constants (one workgroup)
__device__ void conv8_build_table(int* __restrict__ ktab, const C8Plan& pl) {
    if (threadIdx.x < 32) {
        const int ph = threadIdx.x >> 3, f = threadIdx.x & 7, q = ph < pl.nphase ? ph : 0;
        const int v = f == 0 ? pl.kpad[q] / 64 : f == 1 ? pl.toff[q] : f == 2 ? pl.kpad[q] : f == 3 ? (int)(pl.boff[q] & 0xffffffffL)
                      : f == 4 ? (int)(pl.boff[q] >> 32) : 0;
        ktab[threadIdx.x] = v;
    }
    for (int ph = 0; ph < pl.nphase; ++ph)
        for (int j = threadIdx.x; j < pl.kpad[ph] / 8; j += blockDim.x) {
            const int k = j * 8, tl = k / pl.C;
            int e = 31;
            if (tl < pl.ntaps[ph]) {
                const int b = pl.bit[ph][tl], c = k - tl * pl.C;
                const int d16 = ((pl.ddy[b] * pl.Ws + pl.ddx[b]) * pl.C + c) / 8;  // 16-byte units (C % 8 == 0)
                e = (int)((unsigned)d16 << 8) | b;
            }
            ktab[32 + pl.toff[ph] + j] = e;
        }
}
// straight pack: one workgroup per (phase, n) row; 16-byte units.  Also builds the table (blockIdx.x == rows_total).
__global__ __launch_bounds__(256) void conv8_pack_rows_kernel(const bf16_t* __restrict__ w_hi, const bf16_t* __restrict__ w_lo,
                                                              bf16_t* __restrict__ d_hi, bf16_t* __restrict__ d_lo, int* __restrict__ ktab, C8Plan pl) {
    const int row = blockIdx.x;
    if (row == pl.rows_total) {  // chunk table behind its 32-word header of per-phase constants
        if (blockIdx.y == 0) conv8_build_table(ktab, pl);
        return;
    }
    const int ph = row / pl.N, n = row - ph * pl.N;
    const bf16_t* src = blockIdx.y ? w_lo : w_hi;
    bf16_t* dst = (blockIdx.y ? d_lo : d_hi) + pl.boff[ph] + (long)n * pl.kpad[ph];
    for (int j = threadIdx.x; j < pl.kpad[ph] / 8; j += blockDim.x) {
        const int k = j * 8, tl = k / pl.C;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (tl < pl.ntaps[ph]) u = *reinterpret_cast<const uint4*>(src + ((long)n * 9 + pl.tap[ph][tl]) * pl.C + (k - tl * pl.C));
        *reinterpret_cast<uint4*>(dst + k) = u;
    }
}
// transposed pack (data gradients): per tap a [C = Cout][N = Cin] -> [N][C] tile transpose through LDS; blockIdx.z = tap (9: the
// zero padding of the rows' tails)
__global__ __launch_bounds__(256) void conv8_pack_tr_kernel(const bf16_t* __restrict__ w_hi, const bf16_t* __restrict__ w_lo,
                                                            bf16_t* __restrict__ d_hi, bf16_t* __restrict__ d_lo, int* __restrict__ ktab, C8Plan pl) {
    __shared__ bf16_t tile[64][66];
    const int t = threadIdx.x;
    if (blockIdx.z == gridDim.z - 1) {  // last z slice: the chunk table behind its header (one launch instead of two per data gradient)
        if (blockIdx.x == 0 && blockIdx.y == 0) conv8_build_table(ktab, pl);
        return;
    }
    const int split = blockIdx.z / 10, tap = blockIdx.z - split * 10;
    const bf16_t* src = split ? w_lo : w_hi;
    bf16_t* dst = split ? d_lo : d_hi;
    const int N = pl.N, C = pl.C, kpad = pl.kpad[0];
    if (tap == 9) {
        const int padn = kpad - 9 * C;  // < 128
        if (padn <= 0 || blockIdx.y != 0) return;
        const int n0 = blockIdx.x * 64;
        for (int i = t; i < 64 * (padn / 8); i += 256) {
            const int r = i / (padn / 8), j = i - r * (padn / 8);
            if (n0 + r < N) *reinterpret_cast<uint4*>(dst + (long)(n0 + r) * kpad + 9 * C + j * 8) = make_uint4(0, 0, 0, 0);
        }
        return;
    }
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64;  // n = ci (columns of the source), c = co (rows of the source)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (t >> 3) + i * 32, ch = t & 7;
        uint4 u = make_uint4(0, 0, 0, 0);
        if (c0 + r < C && n0 + ch * 8 < N) u = *reinterpret_cast<const uint4*>(src + ((long)(c0 + r) * 9 + tap) * N + n0 + ch * 8);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            tile[r][ch * 8 + 2 * j] = (bf16_t)(w[j] & 0xffffu);
            tile[r][ch * 8 + 2 * j + 1] = (bf16_t)(w[j] >> 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = (t >> 3) + i * 32, ch = t & 7;  // output row = source column c (ci), output columns = source rows (co)
        if (n0 + c < N && c0 + ch * 8 < C) {
            uint32_t w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[ch * 8 + 2 * j][c] | ((uint32_t)tile[ch * 8 + 2 * j + 1][c] << 16);
            *reinterpret_cast<uint4*>(dst + (long)(n0 + c) * kpad + (long)tap * C + c0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
}

// IG_CONV8: 0 = off, 1 = default (launches with enough tiles), 2 = every covered shape (tests).  Read per call.
inline int c8_env() {
    const char* e = getenv("IG_CONV8");
    return e ? atoi(e) : 1;
}

template <int WC, int MT, int NT0, int NT1, int NSEG, bool SPLIT_OUT>
int c8_launch(const C8Params& p, int grid, hipStream_t st) {
    using Geo = C8Geo<WC, MT, NT0, NT1, SPLIT_OUT>;
    auto kern = conv8_kernel<WC, MT, NT0, NT1, NSEG, SPLIT_OUT>;
    const int smem = Geo::OFF_TAB + (p.ktab_n + 32) * 4;
    if (smem > 160 * 1024) return IG_ERR_UNSUPPORTED;
    static int attr_done = 0;
    if (attr_done < smem) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
            ig_set_error("conv8: could not reserve 160 KiB of LDS");
            return IG_ERR_HIP;
        }
        attr_done = 160 * 1024;
    }
    ig_note_kernel("conv8_kernel<%d,%d,%d,%d,%d,%s>", WC, MT, NT0, NT1, NSEG, SPLIT_OUT ? "true" : "false");
    ig_note_grid(grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo::NTHR), smem, st, p);
    return ig_check_launch("conv8");
}

template <int WC, int MT, int NT0, int NT1>
int c8_launch_seg(const C8Params& p, int grid, hipStream_t st) {
    if (p.a[1]) {
        // paired K-tiles when one buffer descriptor can span hi and lo of the gathered tensor (ops.BT allocates them as one block; the packed
        // weights always are); IG_G8_PAIR=0: the three-pass form (A/B runs)
        const char* e = getenv("IG_G8_PAIR");
        const long dA = (const char*)p.a[1] - (const char*)p.a[0], dB = (const char*)p.b[1] - (const char*)p.b[0];
        const bool pair = (!e || atoi(e) != 0) && dA > 0 && dB > 0 && !(dA & 15) && !(dB & 15) && dA + (long)p.a_bytes < 0xfffffff0L && dB + (1L << 24) < (1L << 32);
        if (pair) return c8_launch<WC, MT, NT0, NT1, 2, true>(p, grid, st);
        return c8_launch<WC, MT, NT0, NT1, 3, true>(p, grid, st);
    }
    return c8_launch<WC, MT, NT0, NT1, 1, false>(p, grid, st);
}

struct C8Shape {
    int bm, bn;
    float eff;      // relative main-loop efficiency of the instance (narrower tiles re-read A fragments more often)
    int lds, lds3;  // LDS in front of the chunk table: plain / split output
};
#define C8_SHAPE(WC, MT, NT0, NT1, EFF) \
    {C8Geo<WC, MT, NT0, NT1, false>::BM, C8Geo<WC, MT, NT0, NT1, false>::BN, EFF, C8Geo<WC, MT, NT0, NT1, false>::OFF_TAB, C8Geo<WC, MT, NT0, NT1, true>::OFF_TAB}
// 8-wave instances only: a 6-wave workgroup (192 x 144 / 192 x 96 were built and measured: 400-650 TFLOP/s) leaves two of the four
// SIMDs with one wave, i.e. without the partner whose MFMAs cover its LDS reads and DMA issues.
const C8Shape kShapes[] = {C8_SHAPE(4, 4, 2, 2, 1.00f), C8_SHAPE(4, 4, 2, 1, 0.92f), C8_SHAPE(4, 4, 1, 1, 0.72f)};

}  // namespace
IG_DET_TU(conv8)

// kind 0: Conv2d 3x3 pad 1 forward (sign +1) / data gradient (sign -1);  1: ConvTranspose forward (4 phases);  2: its data gradient.
// x: (B, Hx, Wx, C) activations to gather from; rows = B H W pixels of the ROW grid (H, W); N output channels.
// w: Wc[Cout_w][9][Cin_w].  IG_ERR_UNSUPPORTED (no error string) when the shape is not covered: the caller falls back.
int ig_conv8(int kind, int sign, const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
             const float* scale, const float* shift, void* y_hi, void* y_lo, int B, int H, int W, int C, int N, unsigned drop_seed,
             const unsigned* drop_seed_dev, float drop_p, void* stream) {
    const int env = c8_env();
    if (!env) return IG_ERR_UNSUPPORTED;
    if ((x_lo == nullptr) != (w_lo == nullptr) || (x_lo == nullptr) != (y_lo == nullptr)) return IG_ERR_UNSUPPORTED;
    if (C % 8 || N % 8 || B <= 0) return IG_ERR_UNSUPPORTED;
    const long M = (long)B * H * W;
    const int sm = kind == 2 ? 2 : 1;
    const double a_bytes = (double)M * sm * sm * C * 2.0;
    if (M >= (1L << 30) || a_bytes >= 2147483648.0) return IG_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;

    C8Plan pl{};
    pl.N = N, pl.C = C, pl.Ws = W * sm;
    pl.nphase = kind == 1 ? 4 : 1;
    pl.transposed = (kind == 0 && sign < 0) || kind == 2;
    int nbits = 9;
    if (kind == 1) {
        // phase (py, px): output (2y + py, 2x + px) takes taps ky in {1} (py = 0) or {0, 2} (py = 1); tap ky == 0 reads input row
        // y + 1, every other tap row y (same for x).  Mask bit = dy 2 + dx of the 2 x 2 input neighbourhood.
        nbits = 4;
        for (int b = 0; b < 4; ++b) pl.ddy[b] = b >> 1, pl.ddx[b] = b & 1;
        for (int ph = 0; ph < 4; ++ph) {
            const int py = ph >> 1, px = ph & 1;
            const int nky = py ? 2 : 1, nkx = px ? 2 : 1, ky0 = py ? 0 : 1, kx0 = px ? 0 : 1;
            pl.ntaps[ph] = nky * nkx;
            for (int ty = 0; ty < nky; ++ty)
                for (int tx = 0; tx < nkx; ++tx) {
                    const int ky = ky0 + 2 * ty, kx = kx0 + 2 * tx, tl = ty * nkx + tx;
                    pl.tap[ph][tl] = ky * 3 + kx;
                    pl.bit[ph][tl] = (ky == 0 ? 2 : 0) + (kx == 0 ? 1 : 0);
                }
        }
    } else {
        const int s = kind == 0 ? sign : 1;
        pl.ntaps[0] = 9;
        for (int t = 0; t < 9; ++t) {
            pl.tap[0][t] = t, pl.bit[0][t] = t;
            pl.ddy[t] = s * (t / 3 - 1), pl.ddx[t] = s * (t % 3 - 1);
        }
    }
    long elems = 0;
    int tent = 0;
    for (int ph = 0; ph < pl.nphase; ++ph) {
        pl.kpad[ph] = (pl.ntaps[ph] * C + 127) / 128 * 128;
        pl.boff[ph] = elems, pl.toff[ph] = tent;
        elems += (long)N * pl.kpad[ph];
        tent += pl.kpad[ph] / 8;
    }
    pl.rows_total = pl.nphase * N, pl.ktab_n = tent;

    const int slots = ig_cu_count();
    // conv4_kernel<NI> (4 waves, generated K-loop): plain-bf16 launches whose width tiles by 192 (NI = 6) or by 96 (NI = 3).  IG_GEMM4 = 0: off,
    // 2: every covered shape (tests); default: (nearly) one tile per CU or more (tools/head_bench.py, B = 216: Conv2d 384 forward
    // 481 -> 375 us, data gradient 473 -> 394; Conv2d 192: 530 -> 415, 567 -> 451; ConvTranspose data gradients 288 -> 235, 347 -> 309,
    // 768 -> 384: 206 -> 194; ConvTranspose forward 768 -> 384: 367 -> 280 (conv8), 384 -> 192: 417 -> 376, 192 -> 96: 642 -> 578 (the round-1
    // engine); T = 3, B = 72, 256 x 96 tiles: Conv2d 288 forward 1887 -> 1345, data gradient 1920 -> 1517, ConvTranspose forward 576 -> 288: 1203 -> 854)
    bool use4 = false;
    // 256 x 192 tiles, 256 x 96 (N = 96: the 192 -> 96 ConvTranspose forward; 288 = 3 x 96), or ONE ragged 192-wide tile (N = 144: T = 3's last stage)
    const int ni4 = N % 192 == 0 ? 6 : N % 96 == 0 ? 3 : 6;
    const long nt4 = (M + 255) / 256 * ((N + 32 * ni4 - 1) / (32 * ni4)) * pl.nphase;
    {
        const char* e4 = getenv("IG_GEMM4");
        const int g4 = e4 ? atoi(e4) : 1;
        bool ok4 = g4 && (N % 96 == 0 || N == 144) && tent <= C4_TAB_MAX && a_bytes < 2147483648.0 - 16777216.0 && nt4 < (1L << 30);
        if (w_lo) {  // the split mode: paired K-tiles, hi and lo of the gathered tensor under ONE descriptor below 2 GiB
            const char* ep = getenv("IG_G8_PAIR");
            const long dA = (const char*)x_lo - (const char*)x_hi, dB = (long)(((size_t)elems * 2 + 255) / 256 * 256);
            ok4 = ok4 && (!ep || atoi(ep) != 0) && dA > 0 && !(dA & 15) && (double)dA + a_bytes < 2147483648.0 - 16777216.0 &&
                  dB + (1L << 24) < (1L << 32);
        }
        for (int ph = 0; ph < pl.nphase; ++ph) ok4 = ok4 && pl.kpad[ph] / 64 >= 4;
        use4 = ok4 && (g4 == 2 || nt4 >= slots - slots / 8);
    }
    // instance: the one with the least (rounds x tile area / efficiency) among those whose width divides N
    int best = -1;
    double best_cost = 0;
    for (int i = 0; i < (int)(sizeof(kShapes) / sizeof(kShapes[0])); ++i) {
        const C8Shape& s = kShapes[i];
        const int tn = (N + s.bn - 1) / s.bn;
        const double util = (double)N / ((double)tn * s.bn);  // ragged last column tile: dead MFMA columns
        // ragged last column tile: N = 144 on one 192-wide tile pays (measured); 288 on two does not; the 128-wide instance never does
        // (N = 96 on it: 3360 vs 2192 us in the split mode, 797 vs 610 us plain)
        // (split mode, paired K-tiles: N = 96 on the 128-wide instance now beats the round-1 engine -- bf16x3 step + 0.7 % together with the
        // ConvTranspose rule below)
        if (util < (env == 2 ? 0.5 : s.bn == 128 ? (w_lo ? 0.74 : 1.0) : tn == 1 ? 0.74 : 0.80)) continue;
        if ((w_lo ? s.lds3 : s.lds) + (tent + 32) * 4 > 160 * 1024) continue;
        const long tiles = (M + s.bm - 1) / s.bm * tn * pl.nphase;
        const long rounds = (tiles + slots - 1) / slots;
        const double cost = (double)rounds * s.bm * s.bn / s.eff;
        if (best < 0 || cost < best_cost) best = i, best_cost = cost;
    }
    if (best < 0 && !use4) return IG_ERR_UNSUPPORTED;
    const C8Shape& sh = kShapes[best < 0 ? 0 : best];
    const long ntiles = (M + sh.bm - 1) / sh.bm * ((N + sh.bn - 1) / sh.bn) * pl.nphase;
    // (nearly) one tile per CU or more: 252 tiles pay (ConvTranspose dgrad 2304 -> 1152 at B = 36: 477 -> 319 us), 196 do not (B = 16: -0.9 %
    // of the step): below that the round-1 engine's finer tiles
    if (env != 2 && ntiles < slots - slots / 8 && !use4) return IG_ERR_UNSUPPORTED;
    // ConvTranspose forward: the statically dealt 1- / 2- / 4-tap phase tiles leave the workgroups 10-25 % out of balance (the
    // round-1 engine's phases are dispatched heaviest-first by the hardware).  Measured per stage (tools/head_bench.py): it pays with
    // long reductions and many tiles (768 -> 384 at B = 216, 1152 -> 576, 576 -> 288) and at the widths the round-1 tiles fit badly
    // (N = 144).
    if (kind == 1 && env != 2 && !use4) {
        const bool pays = ntiles >= 4L * slots && (sh.bn >= 192 || w_lo) && (C >= 512 || (N % 64) != 0);
        if (!pays) return IG_ERR_UNSUPPORTED;
    }
    if (ntiles >= (1L << 30)) return IG_ERR_UNSUPPORTED;
    for (int ph = 0; ph < pl.nphase; ++ph)
        if ((long)pl.kpad[ph] * 2 * 256 >= (1L << 24)) return IG_ERR_UNSUPPORTED;  // 24-bit offset multiply of the B pieces

    // scratch: packed hi [, lo], table
    const size_t wbytes = ((size_t)elems * 2 + 255) / 256 * 256;
    const size_t need = wbytes * (w_lo ? 2 : 1) + (size_t)(tent + 32) * 4;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    char* scratch = (char*)ig_scratch2(1, need, !capturing, st);
    if (!scratch) return IG_ERR_UNSUPPORTED;
    bf16_t* p_hi = (bf16_t*)scratch;
    bf16_t* p_lo = w_lo ? (bf16_t*)(scratch + wbytes) : nullptr;
    int* ktab = (int*)(scratch + wbytes * (w_lo ? 2 : 1));

    if (pl.transposed)  // (z: 10 slices per precision half -- 9 taps + the zero padding -- and one more for the chunk table)
        hipLaunchKernelGGL(conv8_pack_tr_kernel, dim3((N + 63) / 64, (C + 63) / 64, 10 * (w_lo ? 2 : 1) + 1), dim3(256), 0, st, (const bf16_t*)w_hi,
                           (const bf16_t*)w_lo, p_hi, p_lo, ktab, pl);
    else
        hipLaunchKernelGGL(conv8_pack_rows_kernel, dim3(pl.rows_total + 1, w_lo ? 2 : 1), dim3(256), 0, st, (const bf16_t*)w_hi, (const bf16_t*)w_lo,
                           p_hi, p_lo, ktab, pl);

    C8Params p{};
    p.a[0] = (const bf16_t*)x_hi, p.a[1] = (const bf16_t*)x_lo;
    p.a_bytes = (unsigned)a_bytes;
    p.b[0] = p_hi, p.b[1] = p_lo;
    p.ktab = ktab, p.ktab_n = tent;
    p.M = (int)M, p.N = N, p.C = C, p.H = H, p.W = W, p.sm = sm;
    p.nphase = pl.nphase, p.nbits = nbits;
    for (int b = 0; b < nbits; ++b) p.ddy_code |= (unsigned)(pl.ddy[b] + 1) << (2 * b), p.ddx_code |= (unsigned)(pl.ddx[b] + 1) << (2 * b);
    p.f_hw = make_fdiv(H * W), p.f_w = make_fdiv(W);
    p.bias = bias, p.scale = scale, p.shift = shift;
    p.out_hi = (bf16_t*)y_hi, p.out_lo = (bf16_t*)y_lo, p.ldo = N;
    p.phase_map = kind == 1;
    p.drop_seed = drop_seed, p.drop_seed_dev = drop_seed_dev;
    p.drop_thresh = ig_drop_thresh16(drop_p);
    p.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    {
        if (use4) {  // (also where conv8 would take 256 x 256: 768 -> 384 data gradient 206 -> 194 us)
            static bool attr4_done = false;
            if (!attr4_done) {
                if (hipFuncSetAttribute((const void*)conv4_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)conv4_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)conv4_kernel<6, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                    hipFuncSetAttribute((const void*)conv4_kernel<3, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    ig_set_error("conv4: could not reserve 160 KiB of LDS");
                    return IG_ERR_HIP;
                }
                attr4_done = true;
            }
            const int grid4 = ig_tile_grid((int)nt4, 1);
            ig_note_kernel(w_lo ? "conv4_kernel<%d,true>" : "conv4_kernel<%d>", ni4);
            ig_note_grid(grid4);
            if (w_lo && ni4 == 6) hipLaunchKernelGGL((conv4_kernel<6, true>), dim3(grid4), dim3(256), C4_OFF_TAB + (tent + 32) * 4, st, p);
            else if (w_lo) hipLaunchKernelGGL((conv4_kernel<3, true>), dim3(grid4), dim3(256), C4_OFF_TAB + (tent + 32) * 4, st, p);
            else if (ni4 == 6) hipLaunchKernelGGL(conv4_kernel<6>, dim3(grid4), dim3(256), C4_OFF_TAB + (tent + 32) * 4, st, p);
            else hipLaunchKernelGGL(conv4_kernel<3>, dim3(grid4), dim3(256), C4_OFF_TAB + (tent + 32) * 4, st, p);
            return ig_check_launch("conv4");
        }
    }
    const int grid = ig_tile_grid((int)ntiles, 1);
    switch (best) {
        case 0: return c8_launch_seg<4, 4, 2, 2>(p, grid, st);
        case 1: return c8_launch_seg<4, 4, 2, 1>(p, grid, st);
        default: return c8_launch_seg<4, 4, 1, 1>(p, grid, st);
    }
}
