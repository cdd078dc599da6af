// v8w: weight gradients on the 256 x 256 x 64 "8-phase" schedule of gemm8.hip, as grouped launches with ordered split-K folds.
//
//   mode 0  dW_g[n][k] += sum_m dy_g[m][n] * x_g[m][k]       the linears of a timm Block (pritvhi.py:446-456: qkv / proj / fc1 / fc2;
//                                                            autograd's grad_weight of F.linear), g = 0 .. ng-1 share the token count
//   mode 1  dWc[co][tap][ci] += sum_p dy[p][co] * x[p + shift(tap)][ci]        nn.Conv2d(k=3, padding=1)      (model.py:370-375)
//           dWc[co][tap][ci] += sum_p dy[up(p) + shift(tap)][co] * x[p][ci]    nn.ConvTranspose2d(k3,s2,p1,op1) (model.py:361-368)
//           runs as mode 1 with the ROLES SWAPPED -- rows = ci (x, plain: 256-row tiles), columns = (tap, co) (dy, gathered at the
//           stride-2 positions, the lane's tap from its column), the fold stores the tile transposed into dWc[co][tap][ci]
//
// * The reduction runs over tokens / pixels, so BOTH operands are reduce-strided ("TR"): an operand half-tile in LDS is
//   [64 tokens][128 columns] (256-byte rows, 16 KiB), filled by LDS-DMA in full 256-byte source rows (two cache lines per
//   token and half-tile), and the MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose).  The bank swizzle is the
//   chunk-pair key of gemm.hip's TR image (lds_trw<16>), applied on the DMA source chunk and on the fragment reads.
// * Tile shapes: the kernel body is a template on MT (16-row blocks of dW per wave and half: 4 -> 256 rows, 3 -> 192) and NT1 (16-column
//   blocks per wave in the second B half: 2 -> 256 columns, 1 -> 192).  The LDS image and the number of LDS-DMA instructions per wave and
//   half-tile are the same for all of them (the lanes of the unused columns are masked), so the barrier / vmcnt protocol is shared.
//   The linears use 256 x 256; the head's channel counts (multiples of 48 / 144) use 192-row tiles.
// * Convolutions: the shifted operand is GATHERED by the LDS-DMA through a buffer descriptor (`buffer_load_dwordx4 ... offen lds`:
//   border taps get an out-of-range offset and the hardware writes zeros, see conv8.hip): per lane a 32-bit byte offset = pixel row
//   (kept incrementally per K-tile: p, y, x advance by 64 tokens) + the tap's displacement and the lane's channel chunk (per segment).
//   mode 1 gathers x (the lane's tap follows from its output column); the ConvTranspose runs in mode 1 with the roles swapped.
// * Work = (output tile, token range) SEGMENTS from a host-built table: the tiles of all GEMMs of the group form one list, so
//   a group of four GEMMs (108 tiles at D = 768) runs with 2 token splits on 216 CUs where four separate launches needed
//   7-28 splits each -- the split-K fold (slab stores + ordered reduce) shrinks from 4 x 32 MB to 54 MB per block and every
//   workgroup runs a 160-K-tile main loop instead of 12-48.  The table holds OFFSETS; the operand pointers travel as kernel
//   arguments, so a plan depends on shapes only (bounded cache, no device allocation per pointer set).
// * Deterministic: every segment stores its fp32 partial tile to its own slab in the accumulator's own (lane-linear) layout
//   -- 16-byte coalesced stores, no LDS staging -- and wgrad8_reduce_kernel adds a tile's slabs in slab order.
// * NSEG = 2 is the split precision mode (bf16x3: hi hi + hi lo + lo hi) in its "paired" form (round 5; the three-pass form it replaces
//   ran the loop once per operand pair): a K-tile covers 32 tokens and its 64 LDS rows are [hi tokens 0-31 | lo tokens 0-31] -- waves 0-3
//   fetch from the hi tensors, waves 4-7 the same token rows from the lo tensors (wave-uniform: no address arithmetic per lane), k-substep
//   0 of the fragment reads is hi, 1 is lo, and a big phase issues its MFMAs three times (hi hi, hi(B) lo(A), lo(B) hi(A) -- gemm4w's order): twice the
//   LDS-DMA traffic and fragment reads of the plain kernel for three times its MFMAs.
// * Same barrier / vmcnt protocol as gemm8_kernel (SCHED 4): two big phases per K-tile, the two row groups one
//   barrier apart, counted vmcnt(6) / vmcnt(2) in front of the issues, the second B half issued behind the MFMAs of big phase 2.
#include <stdlib.h>
#include <string.h>

#include <list>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

namespace {

constexpr int W_HALF = 16384;  // one half-tile: 64 tokens x 128 columns
constexpr int W_BUF = 65536;   // A0 A1 B0 B1
constexpr int W_SMEM = 131072;
constexpr int W_MAXSEG = 6;    // segments per workgroup
constexpr int W_MAXG = 4;      // GEMMs per group
constexpr long W_SLAB = 65536;  // floats per slab slot (the 256 x 256 tile; smaller tiles use a prefix)
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;
typedef int i32x4 __attribute__((ext_vector_type(4)));

struct WSeg {  // one (workgroup, output tile, token range); 64 bytes
    long aoff, boff;  // byte offset of the tile's column block in a row of dy / x; for a plain operand plus the first token's row
    long slab;        // index of the segment's partial tile in the slab workspace
    int lda2, ldb2;   // row pitch of dy / x in bytes
    int nkt;          // K-tiles (64 tokens) of this segment, even; 0 = end of this workgroup's list
    int rows;         // valid tokens from the first one (< 64 nkt on the ragged tail)
    int tok0;         // first token (gathered operands decode their pixel from it)
    short g, tap;     // GEMM of the group (pointer set); mode 2: tap of the tile
    int jcol0;        // mode 1: first column of the tile in (tap, ci) space
    int acols, bcols;  // valid columns of the tile's A / B block (multiples of 8): lanes beyond them are masked
    int pad;
};
static_assert(sizeof(WSeg) == 64, "WSeg layout");

struct WTile {  // reduce table; 40 bytes
    long out_off;  // element offset of the tile origin inside dW of GEMM g (transposed fold: first row = ci of the tile)
    long first;    // first slab of this tile (its slabs are consecutive, in token order)
    int ldo, nslab;
    short g, rows, cols, pad;  // valid rows / columns of the tile
    int jcol0, pad2;           // transposed fold: first column of the tile in (tap, co) space
};
static_assert(sizeof(WTile) == 40, "WTile layout");

struct WArgs {  // operand pointers of the launch (the tables hold offsets)
    const char* a[W_MAXG][2];  // dy hi / lo
    const char* b[W_MAXG][2];  // x hi / lo
};
struct WConv {  // modes 1 / 2: geometry of the pixel grid the tokens run over
    int M, H, W;            // tokens, grid of one image
    int Cg;                 // channels of the gathered operand (mode 1: Cin of x, or Cout of dy with sm = 2; mode 2: Cout of dy)
    int sm;                 // mode 1: the gathered tensor lives on the (sm H, sm W) grid, token (y, x) sits at (sm y, sm x) (2: ConvTranspose)
    int Cout_t, Cin_t;      // transposed fold (ConvTranspose with swapped roles): dWc[co][tap][ci], row pitch 9 Cin_t
    unsigned g_bytes;       // bytes of the gathered tensor (buffer descriptor bound, < 2^31)
    FDiv f_hw, f_w, f_c;    // division by H W, W, Cg
};
struct WDw {
    float* dw[W_MAXG];
};

__device__ __forceinline__ void w_glds_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    // M0 = LDS destination of the DMA.  It is neither saved nor restored (2 SALU fewer per issue; round 5: the s_nop 4 -> 0 and this together
    // are worth 3 % on the grouped weight gradients): hipcc keeps nothing in M0 in these kernels -- gfx9 LDS instructions do not read it -- and
    // tests/test_cpu_host.py::test_m0_is_only_written_by_the_lds_dma_helpers checks the ISA for any other M0 reference
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void w_glds_v(const void* gsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ void w_blds(unsigned voff, i32x4 rsrc, unsigned lds_dst) {
    // M0 = LDS destination of the DMA.  It is neither saved nor restored (2 SALU fewer per issue; round 5: the s_nop 4 -> 0 and this together
    // are worth 3 % on the grouped weight gradients): hipcc keeps nothing in M0 in these kernels -- gfx9 LDS instructions do not read it -- and
    // tests/test_cpu_host.py::test_m0_is_only_written_by_the_lds_dma_helpers checks the ISA for any other M0 reference
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds_dst) : "memory", "m0");
}
__device__ __forceinline__ i32x4 w_rsrc(const void* base, unsigned bytes) {
    const uint64_t a = (uint64_t)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffu));
    r.y = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu));
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}

struct WCur {  // issue cursor of one operand (wave-uniform part)
    const char* base;
    int kt, rows, ld2, seg, left;
};
struct WRows {  // gathered operand: the two token rows this lane fetches per K-tile (rows krow, krow + 4 of the K-tile)
    int p[2], y[2], x[2], b[2];
};

// MODE 0: plain x plain (linears), 1: B gathered (Conv2d 3x3), 2: A gathered (ConvTranspose).  MT: 16-row blocks per wave and A half
// (4: 256-row tile, 3: 192); NT1: 16-column blocks per wave in the second B half (2: 256-column tile, 1: 192).
template <int NSEG, int MODE, int MT, int NT1>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const WSeg* __restrict__ segs, float* __restrict__ slabs,
                                                        const bf16_t* __restrict__ zero_page, WArgs args, WConv cv) {
    constexpr int NTW = 2 + NT1;        // 16-column blocks per wave
    constexpr int ACOLS = 2 * MT * 16;  // columns of an A half actually used (of the 128 in the image)
    constexpr int B1COLS = NT1 * 64;    // columns of the second B half actually used
    constexpr bool MASKED = MODE != 0 || MT != 4 || NT1 != 2;
    static_assert(NSEG == 1 || NSEG == 2, "gemm8w: plain (1) or paired split (2)");
    constexpr bool PAIR = NSEG == 2;
    constexpr int KT = PAIR ? 32 : 64;  // tokens per K-tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const WSeg* __restrict__ my = segs + (size_t)blockIdx.x * W_MAXSEG;
    int Gtot = 0;
#pragma unroll
    for (int s = 0; s < W_MAXSEG; ++s) Gtot += my[s].nkt * NSEG;
    Gtot = __builtin_amdgcn_readfirstlane(Gtot);
    if (Gtot <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    // ---- fragment read offsets inside a half-tile (ds_read_b64_tr_b16: 16 lanes read a 4-token x 16-column block) ----
    const int g4 = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const int rkey = qq | ((g4 & 1) << 2);  // key of token rows 8 g4 + qq (+4, +32)
    const int rlow = ((pp >> 1) << 4) | ((pp & 1) << 3);
    const int k0off = (8 * g4 + qq) * 256;
    // A: column wr*MT*16 + mt*16 + 4pp of the half -> 16-byte chunk pair index (wr MT + mt) ^ key
    // B: half 0: column wc*32 + nt*16 + 4pp -> pair (wc 2 + nt);  half 1: column wc*NT1*16 + nt*16 + 4pp -> pair (wc NT1 + nt)
#define W_AOFF(mt_) (k0off + (((((wr * MT) + (mt_)) ^ rkey) << 5) | rlow))
#define W_BOFF(nt_) (2 * W_HALF + ((nt_) < 2 ? 0 : W_HALF) + k0off + (((((nt_) < 2 ? wc * 2 + (nt_) : wc * NT1 + (nt_)-2) ^ rkey) << 5) | rlow))
    // ---- LDS-DMA lane constants: instruction i of this wave fills token rows wave*8 + i*4 + (lane >> 4) ----
    const int lo_wave = PAIR && wave >= 4 ? 1 : 0;                  // paired: this wave fills the lo rows (32-63) of every half-tile
    const int krow = (PAIR ? (wave & 3) : wave) * 8 + (lane >> 4);  // token of the K-tile behind LDS row wave*8 + i*4 + (lane >> 4)
    const int dkey = (lane >> 4) | ((wave & 1) << 2);
    const int lch = (lane & 15) ^ (dkey << 1);  // logical source chunk (8 columns) of this lane's physical chunk
    const int lch16 = lch << 4;
    const unsigned ldsw = lds_base + wave * 2048;

    WCur cA0, cA1, cB;  // the two B halves of a K-tile are always issued together: one cursor
    WRows rA0, rA1, rB;  // row state of the gathered operand (MODE 1: rB)
    // per-segment lane constants of the gathered operand
    bool okA[2] = {true, true}, okB[2] = {true, true};  // this lane's column exists in half 0 / 1 of the tile
    int shB[2] = {0, 0}, dyB[2] = {0, 0}, dxB[2] = {0, 0};  // MODE 1: byte displacement (tap shift + channel), tap displacement per B half
    i32x4 rs_hi = {0, 0, 0, 0}, rs_lo = {0, 0, 0, 0};
    if constexpr (MODE != 0) {  // one gathered tensor per launch (the group holds ONE convolution)
        rs_hi = w_rsrc(MODE == 1 ? args.b[0][0] : args.a[0][0], cv.g_bytes);
        rs_lo = w_rsrc(MODE == 1 ? args.b[0][1] : args.a[0][1], cv.g_bytes);
    }
    const int C2g = cv.Cg * 2;

#define W_DECODE_ROWS(R, TOK0)                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
        const int p_ = (TOK0) + krow + i_ * 4;                                               \
        const int b_ = cv.f_hw.div(p_), rem_ = p_ - b_ * (cv.H * cv.W);                      \
        (R).p[i_] = p_, (R).b[i_] = b_;                                                      \
        (R).y[i_] = cv.f_w.div(rem_), (R).x[i_] = rem_ - (R).y[i_] * cv.W;                   \
    }
#define W_ADVANCE_ROWS(R)                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                       \
        (R).p[i_] += KT;                                                                     \
        const int xs_ = (R).x[i_] + KT, q_ = cv.f_w.div(xs_);                                \
        (R).x[i_] = xs_ - __mul24(q_, cv.W);                                                 \
        int y_ = (R).y[i_] + q_;                                                             \
        while (y_ >= cv.H) y_ -= cv.H, (R).b[i_]++;                                          \
        (R).y[i_] = y_;                                                                      \
    }
    auto rebaseA = [&](WCur& C, WRows& R, int hg) {
        const WSeg* s_ = my + C.seg;
        const int g_ = s_->g;
        C.ld2 = s_->lda2, C.kt = s_->nkt * NSEG, C.rows = s_->rows;
        C.base = args.a[g_][lo_wave] + s_->aoff;
        if constexpr (MASKED) okA[hg] = lch * 8 < ACOLS && hg * ACOLS + lch * 8 < s_->acols;
    };
    auto rebaseB = [&](WCur& C, WRows& R) {
        const WSeg* s_ = my + C.seg;
        const int g_ = s_->g;
        C.ld2 = s_->ldb2, C.kt = s_->nkt * NSEG, C.rows = s_->rows;
        if constexpr (MODE == 1) {
            C.base = nullptr;
            W_DECODE_ROWS(R, s_->tok0)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int j = s_->jcol0 + g * 128 + lch * 8;  // column in (tap, ci) space
                const int tap = cv.f_c.div(j), ci = j - tap * cv.Cg;
                const int ky = (tap * 11) >> 5, kx = tap - ky * 3;
                dyB[g] = ky - 1, dxB[g] = kx - 1;
                shB[g] = ((dyB[g] * cv.sm * cv.W + dxB[g]) * cv.Cg + ci) * 2;
                okB[g] = lch * 8 < (g ? B1COLS : 128) && g * 128 + lch * 8 < s_->bcols && tap < 9;
            }
        } else {
            C.base = args.b[g_][lo_wave] + s_->boff;
            if constexpr (MASKED) {
#pragma unroll
                for (int g = 0; g < 2; ++g) okB[g] = lch * 8 < (g ? B1COLS : 128) && g * 128 + lch * 8 < s_->bcols;
            }
        }
    };
    auto advance = [&](WCur& C) -> bool {  // one K-tile issued; true: the cursor moved to its next segment (rebase)
        C.left--;
        if (C.base) C.base += (long)C.ld2 * KT;
        C.rows -= KT;
        if (--C.kt == 0) {
            C.seg++;
            return C.left > 0;
        }
        return false;
    };
    // plain operand: the two wave-instructions of this wave for one half-tile
    auto issue_plain = [&](const WCur& C, unsigned dst, int colbyte, bool ok) {
        // a masked lane (column outside the tile / the half) re-reads the first chunk of the tile's block: every wave must issue both
        // instructions (the counted vmcnt waits rely on it), the address stays inside the tensor, the LDS columns it fills are never used
        const int cb = (!MASKED || ok) ? colbyte + lch16 : 0;
        if (C.rows >= KT) {
#pragma unroll
            for (int i = 0; i < 2; ++i) w_glds_s((unsigned)(__mul24(krow + i * 4, C.ld2) + cb), C.base, dst + i * 1024);
        } else {  // ragged tail of the token range: rows past the end come from the zero page
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = krow + i * 4;
                const char* p_ = r < C.rows ? C.base + (unsigned)(__mul24(r, C.ld2) + cb) : (const char*)zero_page + ((lane & 15) << 4);
                w_glds_v(p_, dst + i * 1024);
            }
        }
    };
    // A half hg of the cursor's K-tile into buffer buf
    auto issueA = [&](WCur& C, WRows& R, int hg, int buf) {
        if (C.left <= 0) return;
        const unsigned dst = ldsw + buf * W_BUF + hg * W_HALF;
        issue_plain(C, dst, hg * (ACOLS * 2), okA[hg]);
        if (advance(C)) rebaseA(C, R, hg);
    };
    // B half g of the cursor's K-tile into buffer buf; the cursor advances behind half 1 (the halves of a K-tile are issued in two
    // different phases: half 0 in the read phase of big phase 2, half 1 behind that phase's MFMAs -- see the main loop)
    auto issueB = [&](WCur& C, WRows& R, int buf, int g) {
        if (C.left <= 0) return;
        const unsigned dst = ldsw + buf * W_BUF + 2 * W_HALF + g * W_HALF;
        if constexpr (MODE == 1) {
            const i32x4 rs = lo_wave ? rs_lo : rs_hi;
            // source pixel of token p = (b, y, x): p itself (sm = 1), or (b, 2y, 2x) of the (2H, 2W) image = 4 p - 2 x (ConvTranspose, sm = 2):
            // shifts and one 32-bit multiply per row (the per-half work is two adds, two compares, a select)
            const int sh = cv.sm - 1, Hs = cv.H << sh, Ws = cv.W << sh;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ysh = R.y[i] << sh, xsh = R.x[i] << sh;
                const unsigned rowb = (unsigned)(((R.p[i] << (2 * sh)) - (xsh & -sh)) * C2g);
                const bool v = okB[g] & (R.p[i] < cv.M) & ((unsigned)(ysh + dyB[g]) < (unsigned)Hs) & ((unsigned)(xsh + dxB[g]) < (unsigned)Ws);
                const unsigned voff = v ? rowb + (unsigned)shB[g] : 0x80000000u;
                w_blds(voff, rs, dst + i * 1024);
            }
            if (g == 1) W_ADVANCE_ROWS(R)
        } else {
            issue_plain(C, dst, g * 256, okB[g]);
        }
        if (g == 1 && advance(C)) rebaseB(C, R);
    };

    f32x4 acc[2][NTW][MT];  // [h][nt][mt]
#define W_ZERO_ACC()                                                                                             \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_)       \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) acc[h_][nt_][mt_] = f32x4{0.f, 0.f, 0.f, 0.f};
    W_ZERO_ACC()

    // prologue: K-tile 0 complete + A0, B0, B1 of K-tile 1 (the state the steady-state schedule leaves behind)
    cA0.seg = cA1.seg = cB.seg = 0;
    cA0.left = cA1.left = cB.left = Gtot;
    rebaseA(cA0, rA0, 0);
    rebaseA(cA1, rA1, 1);
    rebaseB(cB, rB);
    issueA(cA0, rA0, 0, 0);
    issueB(cB, rB, 0, 0);
    issueB(cB, rB, 0, 1);
    issueA(cA1, rA1, 1, 0);
    issueA(cA0, rA0, 0, 1);
    issueB(cB, rB, 1, 0);
    issueB(cB, rB, 1, 1);
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    bf16x8_t af[MT][2], bfr[NTW][2];
#define W_TR(DST, ADDR)                                                                                           \
    {                                                                                                             \
        const s16x4 v0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (ADDR)));                \
        const s16x4 v1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (ADDR) + 1024));         \
        typedef __attribute__((ext_vector_type(8))) short s16x8_;                                                 \
        const s16x8_ r_ = {v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};                       \
        DST = __builtin_bit_cast(bf16x8_t, r_);                                                                   \
    }
#define W_READ_A(BUF, H)                                                                                          \
    _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)          \
        W_TR(af[mt_][s_], (BUF)*W_BUF + (H)*W_HALF + s_ * 8192 + W_AOFF(mt_))
#define W_READ_B(BUF)                                                                                             \
    _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)        \
        W_TR(bfr[nt_][s_], (BUF)*W_BUF + s_ * 8192 + W_BOFF(nt_))
    // "big phase": all MFMAs of one A half between one barrier pair; reads are retired BEFORE the first barrier
    // TAIL: LDS-DMA issues of this wave placed behind its MFMAs (as gemm8.hip, schedule 4: the read phases bound the intervals)
#define W_MFMA2(H, TAIL)                                                                                          \
    {                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        asm volatile("s_barrier" ::: "memory");                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        if constexpr (PAIR) { /* fragments [0] = hi, [1] = lo */                                                   \
            _Pragma("unroll") for (int t_ = 0; t_ < 3; ++t_) _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_)   \
                _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_)                                              \
                    acc[H][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt_][t_ == 2], af[mt_][t_ == 1], acc[H][nt_][mt_], 0, 0, 0); \
        } else {                                                                                                  \
        _Pragma("unroll") for (int nt_ = 0; nt_ < NTW; ++nt_) _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) \
            _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)                                                      \
                acc[H][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[nt_][s_], af[mt_][s_], acc[H][nt_][mt_], 0, 0, 0); \
        }                                                                                                         \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        TAIL                                                                                                      \
        asm volatile("s_barrier" ::: "memory");                                                                   \
    }
    // the wait stands BEFORE its phase's issues: R1 retires A1(t) with the 6 younger operations A0 B.0 (R2) B.1 (M2 tail) of K-tile t + 1
    // in flight, R2 retires A0 B (t + 1) with the 2 of A1(t + 1); last iteration: 6 2 0 0
#define W_WAIT(CNT, LASTCNT)                                                     \
    {                                                                            \
        if (last) asm volatile("s_waitcnt vmcnt(" #LASTCNT ")" ::: "memory");    \
        else asm volatile("s_waitcnt vmcnt(" #CNT ")" ::: "memory");             \
    }

    int cseg = 0, it_c = 0;
    int seg_iters = (my[0].nkt * NSEG) >> 1;
    const int iters = Gtot >> 1;
    bool staggered = false;
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
        if (!staggered) {  // (re-)establish the stagger: group 1 runs one barrier behind group 0
            if (wr == 1) asm volatile("s_barrier" ::: "memory");
            staggered = true;
        }
        W_READ_B(0)
        W_READ_A(0, 0)
        W_WAIT(6, 6)
        issueA(cA1, rA1, 1, 1);
        W_MFMA2(0, )
        W_READ_A(0, 1)
        W_WAIT(2, 2)
        issueA(cA0, rA0, 0, 0);
        issueB(cB, rB, 0, 0);
        W_MFMA2(1, issueB(cB, rB, 0, 1);)
        W_READ_B(1)
        W_READ_A(1, 0)
        W_WAIT(6, 0)
        issueA(cA1, rA1, 1, 0);
        W_MFMA2(0, )
        W_READ_A(1, 1)
        W_WAIT(2, 0)
        issueA(cA0, rA0, 0, 1);
        issueB(cB, rB, 1, 0);
        W_MFMA2(1, issueB(cB, rB, 1, 1);)
        if (++it_c < seg_iters) continue;
        // ===== segment finished: store the partial tile (the next segment's first K-tiles are in flight) =====
        it_c = 0;
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // re-align the groups: both store bursts run concurrently
        staggered = false;
        f32x4* sl = reinterpret_cast<f32x4*>(slabs + my[cseg].slab * W_SLAB) + tid;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) sl[((h * NTW + nt) * MT + mt) * 512] = acc[h][nt][mt];
        W_ZERO_ACC()
        ++cseg;
        if (!last) seg_iters = (my[cseg].nkt * NSEG) >> 1;
    }
    if (staggered && wr == 0) asm volatile("s_barrier" ::: "memory");  // (every segment ends re-aligned: not reached)
#undef W_WAIT
#undef W_MFMA2
#undef W_READ_A
#undef W_READ_B
#undef W_TR
#undef W_ZERO_ACC
#undef W_AOFF
#undef W_BOFF
#undef W_DECODE_ROWS
#undef W_ADVANCE_ROWS
}

// ===================================================================================================================================
// gemm4w: the plain 256 x 256 linears on FOUR waves, one per SIMD, each owning a 128 x 128 quadrant of the tile in 256 accumulator
// registers (round 6; the forward-side twin is gemm4.hip).  The K-loop of a segment is one generated assembly block (gen_gemm4.py, the
// "w" form: three A slots + two B stages of the same [64 tokens][128 columns] TR image, 32 fragment reads and 8 LDS-DMA wave-instructions
// per half-iteration between 64 MFMAs); ragged token tails are zero-filled by the buffer descriptor's bound instead of a zero page.
// Same segment tables, same slab workspace, same ordered fold (the slab holds the accumulators in this kernel's lane-linear order:
// wgrad8_reduce_kernel<.., W4 = true>).
#include "gemm4_gen.inc"
constexpr int W4_SMEM = 5 * 32768;

// PAIR: the split precision mode in its paired form (K-tiles of 32 tokens, LDS rows [hi | lo]: waves 0, 1 fetch hi, waves 2, 3 lo; three
// products per K-tile -- gen_gemm4.py, "wp" form).
template <bool PAIR>
__global__ __launch_bounds__(256) void gemm4w_kernel(const WSeg* __restrict__ segs, float* __restrict__ slabs, WArgs args) {
    constexpr int KT = PAIR ? 32 : 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const WSeg* __restrict__ my = segs + (size_t)blockIdx.x * W_MAXSEG;
    if (my[0].nkt <= 0) return;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    // lane constants of the asm blocks (gen_gemm4.py: w_setup): fragment reads as in gemm8w_kernel (16 lanes read 4 tokens x 16 columns)
    const unsigned g4 = lane >> 4, li = lane & 15, qq = li >> 2, pp = li & 3;
    const unsigned rkey5 = (qq | ((g4 & 1) << 2)) << 5;
    const unsigned tbase = lds_base + (8 * g4 + qq) * 256 + pp * 8;
    const unsigned fah = wr * W_HALF, fbh = wc * W_HALF;
    // LDS-DMA: instruction i of this wave fills LDS rows wave*16 + i*4 + (lane >> 4) of a half-tile (paired: token (wave & 1)*16 + ..., hi / lo)
    const int lo_wave = PAIR ? wave >> 1 : 0;
    const unsigned rowv = (PAIR ? (wave & 1) : wave) * 16 + (lane >> 4);
    const unsigned lch0 = ((lane & 15) ^ ((lane >> 4) << 1)) << 4;
    const unsigned ldsw = lds_base + wave * 4096;
    unsigned a0 = 0, a1 = 32768, a2 = 65536;  // A ring (rotated by the asm blocks)

    // descriptor of a segment's operand stream from K-tile `skip` on: base, valid bytes (rows x pitch, clamped)
#define W4_DESC(S, OPER, OFF, LD, SKIP, PTR, NB)                                                  \
    const char* PTR = args.OPER[(S)->g][lo_wave] + (S)->OFF + (long)(SKIP) * KT * (S)->LD;       \
    const unsigned NB = (unsigned)max((S)->rows - (SKIP) * KT, 0) * (unsigned)(S)->LD;
#define W4_LANE_OPERANDS [rowv] "v"(rowv), [lch0] "v"(lch0), [rkey5] "v"(rkey5), [tbase] "v"(tbase), [fah] "v"(fah), [fbh] "v"(fbh)
    {
        const WSeg* s0 = my;
        W4_DESC(s0, a, aoff, lda2, 0, ra, na)
        W4_DESC(s0, b, boff, ldb2, 0, rb, nb)
        const int lda2 = s0->lda2, ldb2 = s0->ldb2;
#define W4_PRO_OPERANDS                                                                                                                 \
    ::[ra] "s"(ra), [na] "s"(na), [rb] "s"(rb), [nb] "s"(nb), [lda2] "s"(lda2), [ldb2] "s"(ldb2), [ldsw] "s"(ldsw), [a0] "s"(a0), [a1] "s"(a1), \
        [a2] "s"(a2), W4_LANE_OPERANDS
        if constexpr (PAIR) asm volatile(G4WP_ASM_PROLOGUE W4_PRO_OPERANDS : G4WP_CLOBBERS);
        else asm volatile(G4W_ASM_PROLOGUE W4_PRO_OPERANDS : G4W_CLOBBERS);
#undef W4_PRO_OPERANDS
    }
    for (int sg = 0; sg < W_MAXSEG; ++sg) {
        const WSeg* s = my + sg;
        const int nkt = s->nkt;
        if (nkt <= 0) break;
        const bool more = sg + 1 < W_MAXSEG && my[sg + 1].nkt > 0;
        const WSeg* sn = more ? s + 1 : s;
        W4_DESC(s, a, aoff, lda2, 2, ra, na)
        W4_DESC(s, b, boff, ldb2, 2, rb, nb)
        W4_DESC(sn, a, aoff, lda2, 0, ran, nan0)
        W4_DESC(sn, b, boff, ldb2, 0, rbn, nbn0)
        const unsigned nan = more ? nan0 : 0u, nbn = more ? nbn0 : 0u;  // no next segment: the last two DMA rounds fill zeros (bound 0)
        const int lda2 = s->lda2, ldb2 = s->ldb2, lda2n = sn->lda2, ldb2n = sn->ldb2;
        const int npair = PAIR ? nkt - 2 : (nkt >> 1) - 2;  // K-tile pairs between the peeled first and last pair
#define W4_SEG_OPERANDS                                                                                                                   \
    : [a0] "+s"(a0), [a1] "+s"(a1), [a2] "+s"(a2)                                                                                         \
    : [ra] "s"(ra), [na] "s"(na), [rb] "s"(rb), [nb] "s"(nb), [ran] "s"(ran), [nan] "s"(nan), [rbn] "s"(rbn), [nbn] "s"(nbn),             \
      [lda2] "s"(lda2), [ldb2] "s"(ldb2), [lda2n] "s"(lda2n), [ldb2n] "s"(ldb2n), [npair] "s"(npair), [ldsw] "s"(ldsw), W4_LANE_OPERANDS
        if constexpr (PAIR) asm volatile(G4WP_ASM_SEG W4_SEG_OPERANDS : G4WP_CLOBBERS);
        else if (nkt == 2) asm volatile(G4W_ASM_SEG_SHORT W4_SEG_OPERANDS : G4W_CLOBBERS);
        else asm volatile(G4W_ASM_SEG W4_SEG_OPERANDS : G4W_CLOBBERS);
#undef W4_SEG_OPERANDS
        // the partial tile leaves in the accumulators' own order: slab element (mi 8 + ni) 256 + tid (16-byte coalesced stores); the next
        // segment's K-tiles 0 and 1 are in flight.  (The thread index is re-materialised behind an empty asm: see gemm4.hip.)
        int tid_e = tid;
        asm volatile("" : "+v"(tid_e));
        f32x4* sl = reinterpret_cast<f32x4*>(slabs + s->slab * W_SLAB) + tid_e;
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) {
            f32x4 tt[8];
            g4_acc_row(mi, tt);
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) sl[(mi * 8 + ni) * 256] = tt[ni];
        }
    }
#undef W4_LANE_OPERANDS
#undef W4_DESC
}

// dW tile (+)= sum of its slabs, in slab (= token) order.  Slab element e = a * 512 + tid holds accumulator a = (h NTW + nt) MT + mt
// of thread tid: dW rows h*2MT16 + wr*MT16 + mt*16 + (lane & 15), columns (nt < 2: wc*32 + nt*16 | 128 + wc*NT1*16 + (nt-2)*16) + 4 (lane >> 4) .. +3.
// W4: the slab was written by gemm4w_kernel -- element e = a * 256 + tid, a = mi 8 + ni: rows (wave >> 1) 128 + mi 16 + (lane & 15), columns
// (wave & 1) 128 + ni 16 + 4 (lane >> 4) .. +3.
template <int MT, int NT1, bool TRANS, bool W4 = false>
__global__ __launch_bounds__(256) void wgrad8_reduce_kernel(const WTile* __restrict__ tiles, const float4* __restrict__ slabs, WDw dws,
                                                            int overwrite, int cout_t, int cin_t) {
    constexpr int NTW = 2 + NT1, NA = 2 * NTW * MT;
    const WTile T = tiles[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;  // 0 .. NA * 512 - 1
    if (e >= NA * 512) return;
    int row, col;
    if constexpr (W4) {
        const int a = e >> 8, tid = e & 255, wave = tid >> 6, lane = tid & 63;
        row = (wave >> 1) * 128 + (a >> 3) * 16 + (lane & 15);
        col = (wave & 1) * 128 + (a & 7) * 16 + 4 * (lane >> 4);
    } else {
        const int a = e >> 9, tid = e & 511, wave = tid >> 6, lane = tid & 63;
        const int h = a / (NTW * MT), nt = (a / MT) % NTW, mt = a % MT;
        row = h * (2 * MT * 16) + (wave >> 2) * (MT * 16) + mt * 16 + (lane & 15);
        col = (nt < 2 ? (wave & 3) * 32 + nt * 16 : 128 + (wave & 3) * (NT1 * 16) + (nt - 2) * 16) + 4 * (lane >> 4);
    }
    if (row >= T.rows || col >= T.cols) return;
    const float4* p = slabs + T.first * (W_SLAB / 4) + e;
    float4 s = p[0];
    for (int k = 1; k < T.nslab; ++k) {
        const float4 t = p[(long)k * (W_SLAB / 4)];
        s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
    }
    float* dw = T.g == 0 ? dws.dw[0] : T.g == 1 ? dws.dw[1] : T.g == 2 ? dws.dw[2] : dws.dw[3];
    if constexpr (TRANS) {  // tile rows = ci, columns = (tap, co): dWc[co][tap][ci] (4 consecutive co per thread: scalar stores)
        const int j = T.jcol0 + col, tap = j / cout_t, co = j - tap * cout_t;
        float* o = dw + ((long)co * 9 + tap) * cin_t + T.out_off + row;
        const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float* oe = o + (long)e * 9 * cin_t;
            *oe = overwrite ? v[e] : *oe + v[e];
        }
        return;
    }
    float4* o = reinterpret_cast<float4*>(dw + T.out_off + (long)row * T.ldo + col);
    if (!overwrite) {  // overwrite: dW = sum (a fresh step: the caller neither zeroed dW nor wants its old contents read)
        const float4 v = *o;
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    *o = s;
}

// ------------------------------------------------------------------------------------------------ host side
struct WPlan {
    WSeg* segs = nullptr;    // device, nwg * W_MAXSEG
    WTile* tiles = nullptr;  // device
    int nwg = 0, ntiles = 0;
    long nslab = 0;
    int min_nkt = 0;  // shortest segment in K-tiles; 0: gemm4w_kernel cannot take the plan (32-bit descriptor bounds)
};
typedef std::vector<long> WKey;
// Plans depend on shapes only (the tables hold offsets), so the cache is small: bounded, least-recently-used eviction, one lock.
struct PlanCache {
    std::mutex mu;
    std::map<WKey, std::pair<WPlan, std::list<WKey>::iterator>> map;
    std::list<WKey> lru;  // front = most recent
    static constexpr size_t kMax = 96;
};
PlanCache& plan_cache() {
    static PlanCache c;
    return c;
}
// slab workspace: per (device, stream), grown on demand (runtime.hip: ig_scratch slot 4); nothing is allocated while a stream capture is
// active (the first eager step on that stream has sized it by then)
float* slab_workspace(long nslab, bool capturing, hipStream_t st) {
    return (float*)ig_scratch2(4, (size_t)nslab * W_SLAB * sizeof(float), !capturing, st);
}
const bf16_t* w_zero_page() {
    constexpr int kMaxDev = 16;
    static void* z[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    if (!z[dev]) {
        if (hipMalloc(&z[dev], 256) != hipSuccess) return nullptr;
        (void)hipMemset(z[dev], 0, 256);
    }
    return (const bf16_t*)z[dev];
}
inline int wg8_env() {  // IG_WGRAD8: 0 = off (the BK = 32 ring engine of gemm.hip), 1 = default.  Read per call (A/B runs, tests).
    const char* e = getenv("IG_WGRAD8");
    return e ? atoi(e) : 1;
}
constexpr int wg8_rem_env() { return 1; }  // ragged token splits (uniform-only splits were an A/B arm)
constexpr int w_rem_aligned_env() { return 1; }  // remainder workgroups in step (the walking remainder was the A/B arm: profiles/r06_gemm8w_xcd_rectangles.txt)
constexpr int IG_W_DEAL_ENV() { return 1; }  // per-XCD rectangles (the round-3 dealing was the A/B arm: profiles/r06_gemm8w_xcd_rectangles.txt)

struct TileRef {  // one output tile of the launch
    int g;         // GEMM (pointer set)
    long aoff, boff;  // byte offset of its column block in a row of dy / x
    int acols, bcols;
    long out_off;
    int ldo;
    int tap, jcol0;
};

inline int wg4_env() {  // IG_GEMM4 (shared with gemm4.hip): 0 = the 8-wave kernel for the linears too (tests, A/B runs), otherwise gemm4w_kernel
    const char* e = getenv("IG_GEMM4");
    return e ? atoi(e) : 1;
}

template <int NSEG, int MODE, int MT, int NT1, bool TRANS>
int w_launch(const WPlan& pl, float* ws, const bf16_t* zp, const WArgs& args, const WConv& cv, const WDw& dws, int overwrite, hipStream_t st) {
    if constexpr (MODE == 0 && MT == 4 && NT1 == 2 && !TRANS) {
        if (pl.min_nkt >= 2 && wg4_env()) {
            static bool attr4_done = false;
            if (!attr4_done) {
                if (hipFuncSetAttribute((const void*)gemm4w_kernel<NSEG == 2>, hipFuncAttributeMaxDynamicSharedMemorySize, W4_SMEM) != hipSuccess) {
                    ig_set_error("gemm4w: could not reserve %d bytes of LDS", W4_SMEM);
                    return IG_ERR_HIP;
                }
                attr4_done = true;
            }
            ig_note_kernel("gemm4w_kernel<%s>", NSEG == 2 ? "true" : "false");
            ig_note_grid(pl.nwg);
            hipLaunchKernelGGL(gemm4w_kernel<NSEG == 2>, dim3(pl.nwg), dim3(256), W4_SMEM, st, (const WSeg*)pl.segs, ws, args);
            hipLaunchKernelGGL((wgrad8_reduce_kernel<4, 2, false, true>), dim3(64, pl.ntiles), dim3(256), 0, st, (const WTile*)pl.tiles,
                               (const float4*)ws, dws, overwrite, 0, 0);
            return ig_check_launch("gemm4w");
        }
    }
    auto kern = gemm8w_kernel<NSEG, MODE, MT, NT1>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM) != hipSuccess) {
            ig_set_error("gemm8w: could not reserve %d bytes of LDS", W_SMEM);
            return IG_ERR_HIP;
        }
        attr_done = true;
    }
    ig_note_kernel("gemm8w_kernel<%d,%d,%d,%d>", NSEG, MODE, MT, NT1);
    ig_note_grid(pl.nwg);
    hipLaunchKernelGGL(kern, dim3(pl.nwg), dim3(512), W_SMEM, st, (const WSeg*)pl.segs, ws, zp, args, cv);
    constexpr int NA = 2 * (2 + NT1) * MT;
    hipLaunchKernelGGL((wgrad8_reduce_kernel<MT, NT1, TRANS>), dim3(NA * 2, pl.ntiles), dim3(256), 0, st, (const WTile*)pl.tiles, (const float4*)ws,
                       dws, overwrite, cv.Cout_t, cv.Cin_t);
    return ig_check_launch("gemm8w");
}

// Plan (cached by `key`) + launch.  tiles: the output tiles of the launch; every tile reduces over the same M tokens.
template <int MODE, int MT, int NT1, bool TRANS = false>
int w_run(const WKey& key, const std::vector<TileRef>& tl, int M, int lda2_of_g[W_MAXG], int ldb2_of_g[W_MAXG], bool split, const WArgs& args,
          const WConv& cv, const WDw& dws, int overwrite, hipStream_t st, const char* what) {
    const long ntiles = (long)tl.size();
    int ncu = ig_cu_count() - ig_reserved_cus();
    if (ncu < 8) ncu = 8;
    if (ntiles > (long)ncu * W_MAXSEG || ntiles < 1) return IG_ERR_UNSUPPORTED;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    PlanCache& cache = plan_cache();
    WPlan pl;
    {
        std::lock_guard<std::mutex> lock(cache.mu);
        auto it = cache.map.find(key);
        if (it != cache.map.end()) {
            cache.lru.splice(cache.lru.begin(), cache.lru, it->second.second);
            pl = it->second.first;
        } else {
            if (capturing) return IG_ERR_UNSUPPORTED;  // a miss allocates and copies synchronously: not inside a stream capture
            // ---- plan.  P = K-tile pairs (128 tokens) per tile, T tiles, C workgroups (one per CU).
            // * uniform: every tile is cut into ks = C / T token ranges; the (split, tile) pairs are listed split-major and each XCD
            //   (workgroup id % 8) takes a contiguous run, so the workgroups that share an L2 stream the SAME token range of
            //   neighbouring tiles (tiles of one GEMM row share the dy column block).  108 tiles (one Block at D = 768): 216 of 256 CUs.
            // * with remainder (IG_WGRAD8_REM, default on): the ks main ranges are shortened to c = ceil(T P / C) pairs -- the length
            //   that balances all C workgroups -- and the last r = P - ks c pairs of every tile go to the R = C - T ks workgroups
            //   left over, which walk the tiles' remainders in tile order (3-4 short segments each, again in step with one another).
            //   Makespan 84 -> 71 pairs at T = 108; the price is one more slab per tile in the fold.
            // * more tiles than CUs: ks = 1 and a workgroup walks several tiles.
            const int P = (M + 127) >> 7;
            const int T = (int)ntiles;
            struct SegRef { int t, p0, np; };
            std::vector<std::vector<SegRef>> wl;  // logical workgroups, in the order they are dealt to the XCDs
            int ks = ncu / T;
            if (ks < 1) ks = 1;
            if (ks > P) ks = P;
            bool rem = false;
            int c = 0, r = 0, R = 0, q = 0;
            if (T <= ncu && wg8_rem_env()) {
                c = (int)(((long)T * P + ncu - 1) / ncu);
                r = P - ks * c, R = ncu - T * ks;
                if (r > 0 && R > 0) {
                    q = (int)(((long)T * r + R - 1) / R);
                    rem = (q + r - 1) / r + 1 <= W_MAXSEG;
                }
            }
            // Remainder workgroups that run IN STEP (round 6).  The walk below gives every remainder workgroup q token pairs of the
            // concatenated tile remainders: it enters its first tile at an offset of its own, so no two of them are ever at the same token of
            // a shared operand slab at the same time -- their fetches (T r 2 slab pairs against the 48 r a Block needs) were most of the
            // kernel's excess traffic.  When the counts divide over the 8 XCDs the remainder is cut differently: every XCD's nr remainder
            // workgroups take WHOLE tile remainders of that XCD's share of the tiles, d = ceil(share / nr) tiles each, one after the other and
            // all starting at the same token -- tiles that are neighbours in the (rectangle-ordered) list, so each time step's nr tiles share
            // their operand slabs in step.  r is chosen so that d remainders weigh about one main segment: P = ks c + r with d r ~ c.
            bool aligned = false;
            if (rem && ((long)T * ks) % 8 == 0 && R % 8 == 0 && w_rem_aligned_env()) {
                const int nr = R / 8, share = (T + 7) / 8, d = (share + nr - 1) / nr;
                if (d >= 1 && d <= W_MAXSEG && (long)nr * d * 8 >= T) {
                    int r2 = (int)((P + (ks * d + 1) / 2) / (ks * d + 1));
                    while (r2 > 1 && (P - r2) % ks) --r2;
                    const int c2 = (P - r2) / ks;
                    if (r2 >= 1 && (P - r2) % ks == 0 && c2 >= 1 && c2 <= c + c / 16) {  // at most ~6 % more than the balanced main segment
                        c = c2, r = r2, aligned = true;
                        for (int sp = 0; sp < ks; ++sp)
                            for (int t = 0; t < T; ++t) wl.push_back({SegRef{t, sp * c, c}});
                        for (int x = 0; x < 8; ++x) {
                            const int t0 = (int)((long)T * x / 8), t1 = (int)((long)T * (x + 1) / 8);  // this XCD's share of the tile remainders
                            for (int j = 0; j < nr; ++j) {
                                std::vector<SegRef> v;
                                for (int t = t0 + j; t < t1; t += nr) v.push_back(SegRef{t, ks * c, r});
                                wl.push_back(v);  // (an empty list is a workgroup without work: it returns at once)
                            }
                        }
                    }
                }
            }
            if (aligned) {
            } else if (rem) {
                for (int sp = 0; sp < ks; ++sp)
                    for (int t = 0; t < T; ++t) wl.push_back({SegRef{t, sp * c, c}});
                for (int j = 0; j < R; ++j) {
                    std::vector<SegRef> v;
                    long u = (long)j * q, ue = u + q < (long)T * r ? u + q : (long)T * r;
                    while (u < ue) {
                        const int t = (int)(u / r), o = (int)(u - (long)t * r);
                        const int np = (int)((r - o) < (ue - u) ? (r - o) : (ue - u));
                        v.push_back(SegRef{t, ks * c + o, np});
                        u += np;
                    }
                    if (!v.empty()) wl.push_back(v);
                }
            } else {
                const int chunk = (P + ks - 1) / ks;
                ks = (P + chunk - 1) / chunk;
                const long Q = (long)T * ks;                    // (split, tile) pairs
                const int per_wg = (int)((Q + ncu - 1) / ncu);  // 1 unless there are more tiles than CUs
                if (per_wg > W_MAXSEG) return IG_ERR_UNSUPPORTED;
                for (long q0 = 0; q0 < Q; q0 += per_wg) {
                    std::vector<SegRef> v;
                    for (long qi = q0; qi < q0 + per_wg && qi < Q; ++qi) {
                        const int sp = (int)(qi / T), t = (int)(qi - (long)sp * T);
                        v.push_back(SegRef{t, sp * chunk, (P - sp * chunk) < chunk ? (P - sp * chunk) : chunk});
                    }
                    wl.push_back(v);
                }
            }
            // XCD x owns a contiguous eighth of the logical workgroup list.  With the remainder workgroups at the END of the list the eighths
            // straddle the token splits (XCD 3: the last 12 tiles of split 0 + the first 20 of split 1 -- nothing shared between them); when the
            // counts divide, every XCD gets an equal run of main segments of ONE split and an equal share of the remainder workgroups instead.
            if (rem && ((long)T * ks) % 8 == 0 && (wl.size() - (size_t)T * ks) % 8 == 0 && IG_W_DEAL_ENV()) {
                const size_t nm = (size_t)T * ks / 8, nr = (wl.size() - (size_t)T * ks) / 8;
                std::vector<std::vector<SegRef>> w2;
                for (int x = 0; x < 8; ++x) {
                    for (size_t i = 0; i < nm; ++i) w2.push_back(wl[x * nm + i]);
                    for (size_t i = 0; i < nr; ++i) w2.push_back(wl[(size_t)T * ks + x * nr + i]);
                }
                wl.swap(w2);
            }
            const long nq = (long)wl.size();
            const int nwg = 8 * (int)((nq + 7) / 8);
            std::vector<WSeg> hs((size_t)nwg * W_MAXSEG);
            memset(hs.data(), 0, hs.size() * sizeof(WSeg));
            const int spt = ks + 2;  // slab slots per tile (a tile's remainder may be cut once by a workgroup boundary)
            std::vector<int> tcount(T, 0);
            int min_nkt = 1 << 30;
            const long qx = nq >> 3, rx = nq & 7;
            for (int b = 0; b < nwg; ++b) {
                const int xcd = b & 7, j = b >> 3;
                const long lo = xcd * qx + (xcd < rx ? xcd : rx), cnt = qx + (xcd < rx ? 1 : 0);
                if (j >= cnt) continue;
                const std::vector<SegRef>& v = wl[lo + j];
                for (size_t si = 0; si < v.size() && si < (size_t)W_MAXSEG; ++si) {
                    const TileRef& tr = tl[v[si].t];
                    const long tok0 = (long)v[si].p0 * 128;
                    WSeg& d = hs[(size_t)b * W_MAXSEG + si];
                    d.lda2 = lda2_of_g[tr.g], d.ldb2 = ldb2_of_g[tr.g];
                    d.aoff = tr.aoff + tok0 * d.lda2;
                    d.boff = tr.boff + (MODE == 1 ? 0 : tok0 * d.ldb2);
                    d.slab = (long)v[si].t * spt + tcount[v[si].t]++;
                    d.nkt = v[si].np * 2;
                    if (d.nkt < min_nkt) min_nkt = d.nkt;
                    const long left = (long)M - tok0;
                    d.rows = (int)(left < (long)d.nkt * 64 ? left : (long)d.nkt * 64);
                    d.tok0 = (int)tok0;
                    d.g = (short)tr.g, d.tap = (short)tr.tap, d.jcol0 = tr.jcol0;
                    d.acols = tr.acols, d.bcols = tr.bcols;
                }
            }
            std::vector<WTile> ht(ntiles);
            for (long t = 0; t < ntiles; ++t) {
                const TileRef& tr = tl[t];
                ht[t].out_off = tr.out_off;
                ht[t].first = t * spt, ht[t].ldo = tr.ldo, ht[t].nslab = tcount[t];
                ht[t].g = (short)tr.g, ht[t].rows = (short)tr.acols, ht[t].cols = (short)tr.bcols, ht[t].pad = 0;
                ht[t].jcol0 = tr.jcol0, ht[t].pad2 = 0;
                if (tcount[t] > spt || tcount[t] < 1) {
                    ig_set_error("%s: internal plan error (tile %ld has %d segments)", what, t, tcount[t]);
                    return IG_ERR_ARG;
                }
            }
            for (int g = 0; g < W_MAXG; ++g)  // gemm4w_kernel's descriptor bound (valid rows x pitch) is 32 bits
                if ((long)M * lda2_of_g[g] >= (1L << 32) || (long)M * ldb2_of_g[g] >= (1L << 32)) min_nkt = 0;
            pl.nwg = nwg, pl.ntiles = (int)ntiles, pl.nslab = ntiles * spt, pl.min_nkt = min_nkt;
            if (hipMalloc((void**)&pl.segs, hs.size() * sizeof(WSeg)) != hipSuccess ||
                hipMalloc((void**)&pl.tiles, ht.size() * sizeof(WTile)) != hipSuccess) {
                ig_set_error("%s: could not allocate the segment tables", what);
                return IG_ERR_HIP;
            }
            // synchronous copies from host vectors (first call of a shape only)
            if (hipMemcpy(pl.segs, hs.data(), hs.size() * sizeof(WSeg), hipMemcpyHostToDevice) != hipSuccess ||
                hipMemcpy(pl.tiles, ht.data(), ht.size() * sizeof(WTile), hipMemcpyHostToDevice) != hipSuccess) {
                ig_set_error("%s: could not upload the segment tables", what);
                return IG_ERR_HIP;
            }
            if (cache.map.size() >= PlanCache::kMax) {  // evict the least recently used plan (its launches may still be queued: drain first)
                (void)hipDeviceSynchronize();
                const WKey old = cache.lru.back();
                auto ot = cache.map.find(old);
                (void)hipFree(ot->second.first.segs);
                (void)hipFree(ot->second.first.tiles);
                cache.map.erase(ot);
                cache.lru.pop_back();
            }
            cache.lru.push_front(key);
            cache.map.emplace(key, std::make_pair(pl, cache.lru.begin()));
        }
    }
    float* ws = slab_workspace(pl.nslab, capturing, st);
    const bf16_t* zp = w_zero_page();
    if (!ws || !zp) {
        if (capturing) return IG_ERR_UNSUPPORTED;
        ig_set_error("%s: could not allocate the slab workspace (%ld slabs)", what, pl.nslab);
        return IG_ERR_HIP;
    }
    if (split) return w_launch<2, MODE, MT, NT1, TRANS>(pl, ws, zp, args, cv, dws, overwrite, st);
    return w_launch<1, MODE, MT, NT1, TRANS>(pl, ws, zp, args, cv, dws, overwrite, st);
}

}  // namespace

// Grouped linear weight gradients.  IG_ERR_UNSUPPORTED (no error string) when a shape is not covered: the caller falls back to one
// ig_linear_wgrad per GEMM.
int ig_wgrad8_group(int n, const void* const* dy_hi, const void* const* dy_lo, const void* const* x_hi, const void* const* x_lo,
                    float* const* dw, const int* N, const int* K, int M, int overwrite, void* stream) {
    if (!wg8_env() || n <= 0 || n > W_MAXG || M <= 0) return IG_ERR_UNSUPPORTED;
    const bool split = dy_lo && dy_lo[0];
    WArgs args{};
    WDw dws{};
    int lda2[W_MAXG] = {}, ldb2[W_MAXG] = {};
    std::vector<TileRef> tl;
    for (int g = 0; g < n; ++g) {
        if (N[g] <= 0 || K[g] <= 0 || (N[g] & 255) || (K[g] & 255)) return IG_ERR_UNSUPPORTED;
        if ((long)N[g] * 2 * 64 >= (1L << 24) || (long)K[g] * 2 * 64 >= (1L << 24)) return IG_ERR_UNSUPPORTED;  // 24-bit offset multiply
        if ((((uintptr_t)dy_hi[g]) | ((uintptr_t)x_hi[g]) | ((uintptr_t)dw[g])) & 15) return IG_ERR_UNSUPPORTED;
        if (split != ((dy_lo && dy_lo[g]) != 0) || split != ((x_lo && x_lo[g]) != 0)) return IG_ERR_UNSUPPORTED;
        if (split && ((((uintptr_t)dy_lo[g]) | ((uintptr_t)x_lo[g])) & 15)) return IG_ERR_UNSUPPORTED;
        args.a[g][0] = (const char*)dy_hi[g], args.a[g][1] = split ? (const char*)dy_lo[g] : (const char*)dy_hi[g];
        args.b[g][0] = (const char*)x_hi[g], args.b[g][1] = split ? (const char*)x_lo[g] : (const char*)x_hi[g];
        dws.dw[g] = dw[g];
        lda2[g] = N[g] * 2, ldb2[g] = K[g] * 2;
        // Tile order: the LONGER dimension of the tile grid is the outer loop, so that a run of consecutive tiles (what one XCD gets) is a
        // compact rectangle: 27 tiles of fc1's 12 x 3 grid = 9 rows x 3 columns share 9 + 3 operand slabs, 27 of fc2's 3 x 12 grid in
        // row-major order 3 + 12 (round 6; a model of the plan, DESIGN 9: 1.32 -> 1.20 x the algorithmic bytes under ideal L2 sharing)
        const int tR = N[g] >> 8, tC = K[g] >> 8;
        const bool rows_outer = tR >= tC || !IG_W_DEAL_ENV();
        for (int o = 0; o < (rows_outer ? tR : tC); ++o)
            for (int i = 0; i < (rows_outer ? tC : tR); ++i) {
                const int tm = rows_outer ? o : i, tn = rows_outer ? i : o;
                tl.push_back(TileRef{g, (long)tm * 512, (long)tn * 512, 256, 256, (long)tm * 256 * K[g] + (long)tn * 256, K[g], 0, 0});
            }
    }
    int ncu = ig_cu_count() - ig_reserved_cus();
    if (ncu < 8) ncu = 8;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    WKey key = {0, n, M, ncu, split, wg8_rem_env() + 2 * IG_W_DEAL_ENV() + 4 * w_rem_aligned_env(), dev_};
    for (int g = 0; g < n; ++g) key.push_back(N[g]), key.push_back(K[g]);
    WConv cv{};
    return w_run<0, 4, 2>(key, tl, M, lda2, ldb2, split, args, cv, dws, overwrite, (hipStream_t)stream, "ig_linear_wgrad_group");
}

// Default routing of the convolution weight gradients (IG_WGRAD8_CONV=1), by measurement against the round-1 engines / the direct
// kernels (tools/head_bench.py, microseconds old -> new; IG_WGRAD8_CONV=2 forces every covered shape):
//   B = 216, T = 1   Conv2d 384: 673 -> 575 (192 x 256 tiles); 192: the direct kernel stays (573 vs 569); 96 / 48: direct kernels (1306 / 2559 forced)
//                    ConvTranspose (roles swapped) 768 -> 384: 455 -> 325 (256 x 256 tiles, 692 TFLOP/s), 384 -> 192: 490 -> 381, 192 -> 96: 646 -> 462;
//                    96 -> 48: the direct kernel stays (373 vs 960)
//   B = 36, T = 3    Conv2d 1152: 1152 -> 940, 576: 1193 -> 839, 144: 1551 -> 1378; 288 (1.5 row tiles): 1088 vs 1165 forced -> stays
//                    ConvTranspose 1152 -> 576: 698 -> 482, 576 -> 288: 841 -> 515, 288 -> 144: 1163 -> 817; 2304 -> 1152 (M = 7056): 652 vs 717 forced -> stays
//   bf16x3           Conv2d 192: 2820 -> 1900 (the direct kernels do not take split operands)
static bool c8w_pays(int kind, bool swap, bool split, int Cin, int Cout, double util, long ntl, int ncu) {
    if (kind == 0) return (util >= 0.90 && Cin >= (split ? 192 : 384)) || Cout == 144;
    if (swap) return util >= 0.70 && Cin >= 192 && Cout >= 96;
    return Cin >= (split ? 384 : 1024) && util >= 0.95 && ntl <= ncu;
}

// Weight gradient of the decode head's convolutions on the same engine.  kind 0: nn.Conv2d(k=3, padding=1): dy (B,H,W,Cout), x (B,H,W,Cin);
// kind 1: nn.ConvTranspose2d(k3,s2,p1,op1): dy (B,2H,2W,Cout), x (B,H,W,Cin).  dWc[Cout][9][Cin] += ...  IG_ERR_UNSUPPORTED (no error
// string) when the shape is not covered.
int ig_wgrad8_conv(int kind, const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, int B, int H, int W,
                   int Cin, int Cout, void* stream) {
    const int env = wg8_env();
    if (!env || B <= 0) return IG_ERR_UNSUPPORTED;
    const char* ce = getenv("IG_WGRAD8_CONV");  // 0: off, 1: default, 2: every covered shape
    const int cenv = ce ? atoi(ce) : 1;
    if (!cenv) return IG_ERR_UNSUPPORTED;
    if ((Cin % 8) || (Cout % 8)) return IG_ERR_UNSUPPORTED;
    const bool split = dy_lo != nullptr;
    if (split != (x_lo != nullptr)) return IG_ERR_UNSUPPORTED;
    if ((((uintptr_t)dy_hi) | ((uintptr_t)x_hi) | ((uintptr_t)dw) | ((uintptr_t)dy_lo) | ((uintptr_t)x_lo)) & 15) return IG_ERR_UNSUPPORTED;
    const long M = (long)B * H * W;
    const double g_bytes = kind == 0 ? (double)M * Cin * 2.0 : (double)M * 4.0 * Cout * 2.0;
    if (M >= (1L << 30) || g_bytes >= 2147483648.0) return IG_ERR_UNSUPPORTED;
    if ((long)Cout * 2 * 64 >= (1L << 24) || (long)Cin * 2 * 64 >= (1L << 24)) return IG_ERR_UNSUPPORTED;
    // ConvTranspose: roles swapped (rows = ci from x, columns = (tap, co) from dy gathered at the stride-2 positions, transposed fold):
    // one GEMM with 256- or 192-row tiles.  (The per-tap form of round 4 -- nine GEMMs with 192-row tiles of a gathered A, 455 against
    // 325 us at 768 -> 384 -- was an A/B arm and is gone.)
    const bool swap = kind == 1;
    const int nrows = swap ? Cin : Cout;                                  // rows of the GEMM's output
    const int ncols = kind == 0 ? 9 * Cin : (swap ? 9 * Cout : Cin);      // its columns
    auto waste = [](int n, int b) { return (double)((n + b - 1) / b * b) / n; };
    // tile shape: 256 x 256 carries 64 MFMAs per wave and K-tile, 192 x 256 48, 192 x 192 36 at the same LDS-DMA / barrier cost: the larger
    // tile wins unless it leaves more than ~12 % of itself empty
    const int BMt = (swap || kind == 0) && waste(nrows, 256) <= waste(nrows, 192) + 1e-9 ? 256 : (waste(nrows, 256) < waste(nrows, 192) - 0.12 ? 256 : 192);
    const int BNt = waste(ncols, 256) <= waste(ncols, 192) + 0.12 ? 256 : 192;
    const int tiles_m = (nrows + BMt - 1) / BMt, tiles_n = (ncols + BNt - 1) / BNt;
    const double util = (double)nrows * ncols / ((double)tiles_m * BMt * tiles_n * BNt);
    int ncu = ig_cu_count() - ig_reserved_cus();
    if (ncu < 8) ncu = 8;
    const long ntl = (long)tiles_m * tiles_n * ((kind == 0 || swap) ? 1 : 9);
    if (cenv != 2) {
        if (M < 8192 || ntl > (long)ncu * W_MAXSEG) return IG_ERR_UNSUPPORTED;
        if (!c8w_pays(kind, swap, split, Cin, Cout, util, ntl, ncu)) return IG_ERR_UNSUPPORTED;
    }
    WArgs args{};
    WDw dws{};
    // operand A = the plain matrix whose columns are the ROWS of the output, operand B = the other one
    const void *ah = swap ? x_hi : dy_hi, *al = swap ? x_lo : dy_lo, *bh = swap ? dy_hi : x_hi, *bl = swap ? dy_lo : x_lo;
    args.a[0][0] = (const char*)ah, args.a[0][1] = split ? (const char*)al : (const char*)ah;
    args.b[0][0] = (const char*)bh, args.b[0][1] = split ? (const char*)bl : (const char*)bh;
    dws.dw[0] = dw;
    int lda2[W_MAXG] = {(swap ? Cin : Cout) * 2, 0, 0, 0}, ldb2[W_MAXG] = {(swap ? Cout : Cin) * 2, 0, 0, 0};
    std::vector<TileRef> tl;
    const int ntap = (kind == 0 || swap) ? 1 : 9;
    for (int tap = 0; tap < ntap; ++tap)
        for (int tm = 0; tm < tiles_m; ++tm)
            for (int tn = 0; tn < tiles_n; ++tn) {
                TileRef t{};
                t.g = 0;
                t.aoff = (long)tm * BMt * 2, t.boff = (kind == 0 || swap) ? 0 : (long)tn * BNt * 2;
                t.acols = nrows - tm * BMt < BMt ? nrows - tm * BMt : BMt;
                t.bcols = ncols - tn * BNt < BNt ? ncols - tn * BNt : BNt;
                t.ldo = 9 * Cin;
                t.out_off = swap ? (long)tm * BMt : (long)tm * BMt * t.ldo + (kind == 0 ? 0 : (long)tap * Cin) + (long)tn * BNt;
                t.tap = tap, t.jcol0 = tn * BNt;
                tl.push_back(t);
            }
    WConv cv{};
    cv.M = (int)M, cv.H = H, cv.W = W, cv.Cg = kind == 0 ? Cin : Cout;
    cv.sm = swap ? 2 : 1, cv.Cout_t = Cout, cv.Cin_t = Cin;
    cv.g_bytes = (unsigned)g_bytes;
    cv.f_hw = make_fdiv(H * W), cv.f_w = make_fdiv(W), cv.f_c = make_fdiv(cv.Cg);
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    const WKey key = {1 + kind + (swap ? 2 : 0), (long)M, ncu, split, wg8_rem_env(), dev_, H, W, Cin, Cout, BMt, BNt};
    hipStream_t st = (hipStream_t)stream;
    const int M_ = (int)M;
#define IG_W8C(MODE, TRANS, WHAT)                                                                                                  \
    {                                                                                                                              \
        if (BMt == 256 && BNt == 256) return w_run<MODE, 4, 2, TRANS>(key, tl, M_, lda2, ldb2, split, args, cv, dws, 0, st, WHAT);  \
        if (BMt == 256) return w_run<MODE, 4, 1, TRANS>(key, tl, M_, lda2, ldb2, split, args, cv, dws, 0, st, WHAT);                \
        if (BNt == 256) return w_run<MODE, 3, 2, TRANS>(key, tl, M_, lda2, ldb2, split, args, cv, dws, 0, st, WHAT);                \
        return w_run<MODE, 3, 1, TRANS>(key, tl, M_, lda2, ldb2, split, args, cv, dws, 0, st, WHAT);                                \
    }
    if (kind == 0) IG_W8C(1, false, "ig_conv3x3_wgrad")
    IG_W8C(1, true, "ig_convT_wgrad")
#undef IG_W8C
}
