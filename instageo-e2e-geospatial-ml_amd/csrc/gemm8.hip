// v8: the forward ("NT", both operands K-contiguous) linear GEMM of the timm Block (pritvhi.py:446-456: qkv / proj /
// fc1 / fc2) and of the patch embedding as a 256 x 256 x 64 "8-phase" ping-pong kernel for gfx950.
//
//   C[m][n] = sum_seg sum_k A_seg[m][k] * B_seg[n][k]      A = activations [M][K], B = nn.Linear weight [N][K]
//
// * One 512-thread workgroup per CU (8 waves = 2 row groups x 4 column waves), persistent over an XCD-contiguous tile
//   list.  A wave owns 128 x 64 of the tile: 4 quadrants (h, g) of 64 x 32 = 32 MFMA 16x16x32 accumulators (128 VGPRs).
// * K-step 64: an operand half-tile is 128 rows x 128 B = 16 KiB and is moved global -> LDS by LDS-DMA in FULL cache
//   lines (one global_load_lds_dwordx4 = 8 rows x 128 B; BK = 32 moved half lines).  LDS = 2 buffers x {A0 A1 B0 B1}
//   = 128 KiB + 8 x 4 KiB epilogue staging = 160 KiB.  The image is lane-linear; the bank swizzle (16-byte chunk ^=
//   row & 7) is applied on the SOURCE address and by the same involution on the fragment reads (conflict-free
//   ds_read_b128).
// * The halves interleave over the waves: A half h = tile rows with ((row >> 6) & 1) == h, B half g = tile columns with
//   ((col >> 5) & 1) == g, so quadrant (h, g) of EVERY wave reads half-tiles (A_h, B_g) and a half-tile is dead for the
//   whole workgroup as soon as its quadrant phase is over -- it is refilled two phases later (K-tile t+2), four
//   half-tiles (64 KiB) in flight.  Per K-tile 4 phases:
//       P1: read B0 A0, MFMA A0xB0 | P2: read B1, MFMA A0xB1 | P3: read A1, MFMA A1xB1 | P4: MFMA A1xB0
//   a phase = [ds_reads + 2 DMA issues + counted vmcnt] s_barrier [16 MFMAs, s_setprio 1] s_barrier.  The two row
//   groups run ONE barrier apart, so one group's MFMAs cover the other group's LDS reads and DMA issues.
//       RAW: a half-tile is read one phase after the counted vmcnt(8) (+ barrier) that retires it;
//       WAR: a half-tile is re-issued >= 2 phases after its last ds_read.
// * Epilogue: the groups are re-aligned first (both epilogues run concurrently on the SIMD's two waves), the bias is the
//   accumulator INIT (no epilogue loads), results go through a wave-private 4 KiB LDS staging slab and leave as
//   row-contiguous 16-byte stores (8 full 128-byte lines per wave-instruction instead of 16 x 32-byte pieces).
// * NSEG = 3 runs the split-bf16 (hi*hi + hi*lo + lo*hi) precision mode through the same loop, one operand pair after the other: three
//   times the LDS-DMA traffic and fragment reads for three times the MFMAs.
// * NSEG = 2 ("paired", round 5) is the same precision mode for operands whose lo tensor lies within 4 GiB ABOVE its hi tensor (ops.BT
//   allocates them as one block): a K-tile covers 32 reduction elements and its 128-byte LDS row is [hi k..k+31 | lo k..k+31] -- the
//   lanes of one LDS-DMA instruction fetch their 16-byte chunk from hi or lo (the lo lanes carry the tensors' distance in their 32-bit
//   offset).  The fragment reads are unchanged (k-substep 0 = hi, 1 = lo) and a big phase issues 48 MFMAs (hi hi, hi lo, lo hi) on the
//   fragments that the plain kernel feeds to 32: twice the traffic and reads of the plain kernel for three times its MFMAs, and an MFMA
//   phase (768 cycles) that is as long as the other group's read phase (760-970, see SCHED 4) instead of shorter.
#include <stdlib.h>

#include "common.h"

#ifdef IG_G8_PROF
// Diagnostic builds only (csrc: `make prof` -> libinstageo_hip_g8prof{1,2}.so, tools/gemm8_phase_prof.py): s_memtime stamps at the phase
// boundaries around one tile transition of workgroup 0; they go to a buffer nothing else reads.  IG_G8_PROF=1 also drains the stores
// behind the epilogue (how long until they are acknowledged), 2 leaves the kernel's own waits alone.
__device__ unsigned long long g_g8prof[8 * 64];
#define G8P_MARK(I) { if (g8p_on) g_g8prof[wave * 64 + (I)] = __builtin_readcyclecounter(); }
#define G8P_REAL(I) { if (g8p_on) g_g8prof[wave * 64 + (I)] = __builtin_amdgcn_s_memrealtime(); }  // 100 MHz: the clock = d(memtime) / d(memrealtime) x 100 MHz
extern "C" int ig_debug_g8prof(unsigned long long* dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_g8prof), sizeof(g_g8prof)) == hipSuccess ? 0 : -1; }
#else
#define G8P_MARK(I)
#define G8P_REAL(I)
#endif
namespace {

// Epilogue stores.  Non-temporal stores drain the write burst of 256 CUs faster in isolation (round 4, same-box A/B of two builds at
// M = 42552: proj 87 -> 72 us, fc1 + gelu' 256 -> 239, d_fc1 164 -> 157, qkv / fc2 unchanged) but over the whole step the gain is gone
// (44.36 -> 44.28 ms): the consumer of the tile then misses in L2 / MALL.  -DIG_G8_NT builds them all non-temporal (A/B builds); by
// default only the tensor nobody reads before the backward pass -- gelu' saved by fc1 -- takes the non-temporal path.
template <bool NT = false, typename T>
__device__ __forceinline__ void g8_store(T* ptr, const T& v) {
#ifdef IG_G8_NT
    constexpr bool nt = true;
#else
    constexpr bool nt = NT;
#endif
    if constexpr (nt) {
        static_assert(sizeof(T) == 16, "16-byte stores");
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, v), reinterpret_cast<u32x4_t*>(ptr));
    } else {
        *ptr = v;
    }
}

// Geometry: WC column waves x 2 row groups; a wave owns (2 MT 16) x 64 of the (MT 64) x (WC 64) tile.  Two instances:
//   MT = 4, WC = 4: 256 x 256, 8 waves, 160 KiB of LDS, one workgroup per CU            (the shapes with >= 128 such tiles)
//   MT = 2, WC = 2: 128 x 128, 4 waves,  80 KiB of LDS, TWO workgroups per CU           (small batches: M = 16 x 197 rows leave
//                   39-156 tiles of 256 x 256 for 256 CUs; the 128 x 128 tiles are 4 x as many and two workgroups share a CU)
// In both, a half-tile is (MT WR 16 = WC 32) rows x 128 B and every wave issues TWO LDS-DMA instructions per half-tile, so the
// barrier / vmcnt protocol is identical.
template <int MT, int WC>
struct G8Geo {
    static_assert((MT == 4 && WC == 4) || (MT == 2 && WC == 2), "gemm8: 256 x 256 (MT 4, WC 4) or 128 x 128 (MT 2, WC 2)");
    static constexpr int NW = 2 * WC, NTHR = 64 * NW;
    static constexpr int BM = 2 * 2 * MT * 16, BN = WC * 64;
    static constexpr int HALF = MT * 2 * 16 * 128;  // one half-tile: (BM / 2) rows x 64 k  ( == (BN / 2) x 64 k )
    static constexpr int BUF = 4 * HALF;            // A0 A1 B0 B1
    static constexpr int STAGE = 2 * BUF;           // epilogue staging: NW waves x 4 KiB
    static constexpr int SMEM = STAGE + NW * 4096;
};
typedef __attribute__((address_space(3))) char* lds_char_ptr;

// LDS-DMA with a scalar base + 32-bit per-lane offset (the K advance is one scalar add per issue)
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    // M0 = LDS destination of the DMA.  It is neither saved nor restored (2 SALU fewer per issue; round 5: the s_nop 4 -> 0 and this together
    // are worth 3 % on the grouped weight gradients): hipcc keeps nothing in M0 in these kernels -- gfx9 LDS instructions do not read it -- and
    // tests/test_cpu_host.py::test_m0_is_only_written_by_the_lds_dma_helpers checks the ISA for any other M0 reference
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_dst) : "memory", "m0");
}

struct Cur {  // issue cursor of one half-tile type (all wave-uniform)
    const char* base;
    int kt, seg, tile, vr, left;
};

// SCHED 2: two "big phases" of 32 MFMAs per K-tile (4 barriers per K-tile instead of the 8 of the header comment's 16-MFMA phases:
// +2-4 %, round 2; that four-phase schedule and one that balanced its LDS reads were removed).  SCHED 4 (default, round 5): the same
// big phases with (i) every counted wait BEFORE its phase's issues and (ii) the B1 half-tile of K-tile t + 2 issued BEHIND the MFMAs of
// big phase 2 instead of in its read phase.  In-kernel stamps (profiles/r05_gemm8_phase_*.txt): a read phase -- 8-16 fragment reads,
// 2-6 LDS-DMA issues at ~95 cycles each, the wait -- takes 760-970 cycles, the other group's 32 MFMAs 640, so every interval is as long
// as its READ side and a wave waits ~500 cycles at the closing barrier behind its MFMAs: that is where two of the six issues of the
// long read phase go.  WAR: the slot was last read two intervals earlier, by both groups.  The DMA stream of a wave in program order:
// [R1: A1(t+1)] [R2: A0 B0 (t+2)] [M2: B1(t+2)]; R1's wait retires A1(t) with 6 younger operations in flight, R2's retires
// A0 B0 B1 (t+1) with 2 (last iteration: 6 2 0 0).  qkv 142.0 -> 135.4 us, fc2 193.8 -> 190.4 at M = 42552 (same box, tools/gemm8_bench.py
// --sched); also measured, not kept: A1 behind the MFMAs of big phase 1 as well (140.1), the waits moved alone (141.9), the issues in front
// of their phase's fragment reads (134.0 against 135.1: noise).  In-kernel clock under the loop: 2.12-2.15 GHz (s_memtime / s_memrealtime);
// a 2-K-tile iteration takes ~7950 cycles against the 4096 of its MFMAs: the matrix pipe is ~50 % busy; inside an MFMA phase the 32 MFMAs
// issue in 620-660 cycles (512 back to back).
// (Timing ablations of round 2 -- no LDS-DMA / no fragment reads / no barriers / no epilogue stores: of 71 us (qkv, B = 108) the epilogue was
// 13.7, LDS-DMA 10.2, fragment reads 2.5, barriers 1.7 and the MFMAs themselves 43 us -- are in profiles/r02_v8_ablation_qkv.log; the build
// switch that produced them is gone.)
template <int KIND, int NSEG, int ACT, bool DACT, bool SPLIT_OUT, int SCHED = 4, int MT = 4, int WC = 4>
__global__ __launch_bounds__(128 * WC, 2) void gemm8_kernel(G8Params p) {
    using Geo = G8Geo<MT, WC>;
    constexpr int G8_HALF = Geo::HALF, G8_BUF = Geo::BUF, G8_STAGE = Geo::STAGE, BM = Geo::BM, BN = Geo::BN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave / WC, wc = wave % WC;
    const int tiles_n = p.N / BN, tiles_m = (p.M + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    // persistent, XCD-aware: workgroups are dealt round-robin over the 8 XCDs; XCD x owns a contiguous tile range
    const int nb = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
    const int nbx = (nb >> 3) + (xcd < (nb & 7) ? 1 : 0);
    const int qT = ntiles >> 3, rT = ntiles & 7;
    const int tlo = xcd * qT + min(xcd, rT), tcnt = qT + (xcd < rT ? 1 : 0);
    const int my_tiles = tcnt > jx ? (tcnt - jx + nbx - 1) / nbx : 0;
    if (my_tiles <= 0) return;
    constexpr bool PAIR = NSEG == 2;
    const int nk = PAIR ? p.K >> 5 : p.K >> 6;     // K-tiles per segment (even: K % 128 == 0)
    const int per_tile = nk * (NSEG == 3 ? 3 : 1);  // K-tiles per output tile
    const int Gtot = my_tiles * per_tile;
    const int lda2 = (int)(p.lda * 2), ldb2 = (int)(p.ldb * 2);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    // fragment read offsets inside a half-tile (k-substep 1: ^ 64)
    const int sw = ((lane >> 4) ^ (lane & 7)) << 4;
    const int aoff = (wr * (MT * 16) + (lane & 15)) * 128 + sw;
    const int boff = 2 * G8_HALF + (wc * 32 + (lane & 15)) * 128 + sw;
    // LDS-DMA lane constants: LDS row of instruction i of this wave = wave*16 + i*8 + (lane >> 3)
    const int rbaseA = ((16 * wave) / (MT * 16)) * (2 * MT * 16) + (16 * wave) % (MT * 16) + (lane >> 3);  // + h*MT*16 + i*8  -> tile row
    const int rbaseB = (wave >> 1) * 64 + (wave & 1) * 16 + (lane >> 3);   // + g*32 + i*8  -> tile column
    // source chunk of this lane's LDS chunk; paired: chunks 0-3 = hi, 4-7 = lo of the same 32 reduction elements
    const int sc = (lane & 7) ^ (lane >> 3);
    const unsigned c16a = PAIR ? (unsigned)((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.a[2] - (const char*)p.a[0]) : 0u) : (unsigned)(sc << 4);
    const unsigned c16b = PAIR ? (unsigned)((sc & 3) << 4) + (sc >= 4 ? (unsigned)((const char*)p.b[1] - (const char*)p.b[0]) : 0u) : (unsigned)(sc << 4);
    const unsigned ldsw = lds_base + wave * 2048;

    Cur cA0, cA1, cB0, cB1;
#define G8_REBASE(C, ISA)                                                                   \
    {                                                                                       \
        const int bm_ = (C).tile / tiles_n, bn_ = (C).tile - bm_ * tiles_n;                 \
        const int s_ = NSEG != 3 ? 0 : (C).seg;                                             \
        if (ISA) {                                                                          \
            const bf16_t* b_ = s_ == 0 ? p.a[0] : s_ == 1 ? p.a[1] : p.a[2];                \
            (C).base = (const char*)b_ + (long)bm_ * BM * lda2;                             \
            (C).vr = min(BM, p.M - bm_ * BM);                                               \
        } else {                                                                            \
            const bf16_t* b_ = s_ == 0 ? p.b[0] : s_ == 1 ? p.b[1] : p.b[2];                \
            (C).base = (const char*)b_ + (long)bn_ * BN * ldb2;                             \
            (C).vr = BN;                                                                    \
        }                                                                                   \
    }
#define G8_INIT(C, ISA)                                        \
    {                                                          \
        (C).kt = 0, (C).seg = 0, (C).tile = tlo + jx, (C).left = Gtot; \
        G8_REBASE(C, ISA)                                      \
    }
    // issue the two wave-instructions of this wave for half-tile (ISA ? A : B, HG) of the cursor's K-tile into buffer BUF
#define G8_ISSUE(C, ISA, HG, BUF)                                                                              \
    if ((C).left > 0) {                                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                     \
            int row_ = ((ISA) ? rbaseA + (HG)*(MT * 16) : rbaseB + (HG)*32) + i_ * 8;                          \
            row_ = min(row_, (C).vr - 1);                                                                      \
            const unsigned voff_ = (unsigned)__mul24(row_, (ISA) ? lda2 : ldb2) + ((ISA) ? c16a : c16b);       \
            glds16_s(voff_, (C).base, ldsw + (BUF)*G8_BUF + ((ISA) ? 0 : 2 * G8_HALF) + (HG)*G8_HALF + i_ * 1024); \
        }                                                                                                      \
        (C).left--;                                                                                            \
        (C).base += PAIR ? 64 : 128;                                                                           \
        if (++(C).kt == nk) {                                                                                  \
            (C).kt = 0;                                                                                        \
            if (NSEG != 3 || ++(C).seg == NSEG) {                                                              \
                (C).seg = 0;                                                                                   \
                (C).tile += nbx;                                                                               \
            }                                                                                                  \
            G8_REBASE(C, ISA)                                                                                  \
        }                                                                                                      \
    }

    f32x4 acc[2][2][2][MT];  // [h][g][nt][mt]
    f32x4 binit[2][2];      // accumulator init = bias in MFMA layout (4 consecutive n per lane)
    int tile_c = tlo + jx;
#define G8_LOAD_BIAS(TILE)                                                                                       \
    {                                                                                                            \
        const int bn_ = (TILE) % tiles_n;                                                                        \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) {   \
            if (p.bias) {                                                                                        \
                const float4 b4_ = *reinterpret_cast<const float4*>(p.bias + bn_ * BN + wc * 64 + g_ * 32 + nt_ * 16 + 4 * (lane >> 4)); \
                binit[g_][nt_] = f32x4{b4_.x, b4_.y, b4_.z, b4_.w};                                              \
            } else {                                                                                             \
                binit[g_][nt_] = f32x4{0.f, 0.f, 0.f, 0.f};                                                      \
            }                                                                                                    \
        }                                                                                                        \
    }
#define G8_INIT_ACC()                                                                                            \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)            \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) \
            acc[h_][g_][nt_][mt_] = binit[g_][nt_];
    G8_LOAD_BIAS(tile_c)
    G8_INIT_ACC()

    // prologue: K-tile 0 complete + A0, B0 of K-tile 1 (the state the steady-state schedule leaves behind)
    G8_INIT(cA0, true)
    G8_INIT(cA1, true)
    G8_INIT(cB0, false)
    G8_INIT(cB1, false)
    G8_ISSUE(cA0, true, 0, 0)
    G8_ISSUE(cB0, false, 0, 0)
    G8_ISSUE(cB1, false, 1, 0)
    G8_ISSUE(cA1, true, 1, 0)
    G8_ISSUE(cA0, true, 0, 1)
    G8_ISSUE(cB0, false, 0, 1)
    if constexpr (SCHED >= 2) G8_ISSUE(cB1, false, 1, 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    bf16x8_t af[MT][2], bf0[2][2], bf1[2][2];
#define G8_READ_A(BUF, H)                                                                                          \
    _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) \
        af[mt_][s_] = *reinterpret_cast<const bf16x8_t*>(smem + (BUF)*G8_BUF + (H)*G8_HALF + mt_ * 2048 + (aoff ^ (s_ * 64)));
#define G8_READ_B(BUF, G, DST)                                                                                     \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_) \
        DST[nt_][s_] = *reinterpret_cast<const bf16x8_t*>(smem + (BUF)*G8_BUF + (G)*G8_HALF + nt_ * 2048 + (boff ^ (s_ * 64)));
// SCHED 2 "big phase": 32 MFMAs (two quadrants) between one barrier pair; reads are retired BEFORE the first barrier
    // (lgkmcnt(0)), so a half-tile may be refilled in the very next phase
#define G8_MFMA2(H) G8_MFMA2T(H, 60, )
    // PM: first of three stamp slots of the diagnostic build; TAIL: LDS-DMA issues of this wave placed behind its MFMAs (SCHED 4)
#define G8_MFMA_TERM(H, T)                                                                                         \
    _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)           \
        _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_)                                                       \
            acc[H][g_][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_ ? bf1[nt_][(T) == 1] : bf0[nt_][(T) == 1], af[mt_][(T) == 2], acc[H][g_][nt_][mt_], 0, 0, 0);
#define G8_MFMA2T(H, PM, TAIL)                                                                                     \
    {                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                         \
        G8P_MARK(PM)                                                                                               \
        asm volatile("s_barrier" ::: "memory");                                                      \
        G8P_MARK(PM + 1)                                                                                           \
        __builtin_amdgcn_s_setprio(1);                                                                             \
        if constexpr (PAIR) { /* fragments [0] = hi, [1] = lo: hi hi, lo(B) hi(A), hi(B) lo(A), each term over all 16 accumulators.  (Issuing  \
            the first term in FRONT of the opening barrier -- behind a sched_barrier, hipcc moves MFMAs across an asm barrier -- to fill   \
            the pipe behind the other group's last MFMAs: qkv 362 -> 366 us, fc2 443 -> 442: nothing.) */                              \
            G8_MFMA_TERM(H, 0)                                                                                     \
            G8_MFMA_TERM(H, 1)                                                                                     \
            G8_MFMA_TERM(H, 2)                                                                                     \
        } else {                                                                                                   \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)       \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)  \
                acc[H][g_][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_ ? bf1[nt_][s_] : bf0[nt_][s_], af[mt_][s_], acc[H][g_][nt_][mt_], 0, 0, 0); \
        }                                                                                                          \
        __builtin_amdgcn_s_setprio(0);                                                                             \
        G8P_MARK(PM + 2)                                                                                           \
        TAIL                                                                                                       \
        asm volatile("s_barrier" ::: "memory");                                                      \
    }
#define G8_WAITN(CNT, LASTCNT)                                                               \
    {                                                                                        \
        if (last) asm volatile("s_waitcnt vmcnt(" #LASTCNT ")" ::: "memory");                \
        else asm volatile("s_waitcnt vmcnt(" #CNT ")" ::: "memory");                         \
    }
#define G8_WAIT(LASTCNT)                                                                     \
    {                                                                                        \
        if (last) asm volatile("s_waitcnt vmcnt(" #LASTCNT ")" ::: "memory");                \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                                \
    }

#ifdef IG_G8_PROF
    int g8p_tiles = 0;    // tiles finished by this workgroup
    bool g8p_on = false;  // the stamps of this iteration are recorded
    int g8p_base = 0;
#endif
    int it_c = 0;  // iterations (K-tile pairs) done of the current tile
    const int iters = Gtot >> 1, per_tile2 = per_tile >> 1;
    bool staggered = false;
    for (int it = 0; it < iters; ++it) {
        const bool last = it == iters - 1;
#ifdef IG_G8_PROF
        // recorded: the first iteration behind the 2nd tile's epilogue (slots 2-14) and the 2nd iteration of the 3rd tile (18-30, a plain one)
        g8p_on = blockIdx.x == 0 && g8p_tiles == 2 && it_c <= 1;
        g8p_base = it_c == 0 ? 0 : 16;
        G8P_MARK(g8p_base + 2)
        if (g8p_base) G8P_REAL(56)
#endif
        if (!staggered) {  // (re-)establish the stagger: group 1 runs one barrier behind group 0
            if (wr == 1) asm volatile("s_barrier" ::: "memory");
            staggered = true;
        }
        if constexpr (SCHED == 2) {
            // two big phases per K-tile: BP1 reads A0 B0 B1 + issues A1 of the next K-tile; BP2 reads A1 + issues A0 B0 B1 of
            // K-tile + 2 into the slots BP1 has just retired.  Every wait is vmcnt(8) (see the header comment).
            G8_READ_B(0, 0, bf0)
            G8_READ_B(0, 1, bf1)
            G8_READ_A(0, 0)
            G8_ISSUE(cA1, true, 1, 1)
            G8_WAIT(8)
            G8_MFMA2(0)
            G8_READ_A(0, 1)
            G8_ISSUE(cA0, true, 0, 0)
            G8_ISSUE(cB0, false, 0, 0)
            G8_ISSUE(cB1, false, 1, 0)
            G8_WAIT(2)
            G8_MFMA2(1)
            G8_READ_B(1, 0, bf0)
            G8_READ_B(1, 1, bf1)
            G8_READ_A(1, 0)
            G8_ISSUE(cA1, true, 1, 0)
            G8_WAIT(0)
            G8_MFMA2(0)
            G8_READ_A(1, 1)
            G8_ISSUE(cA0, true, 0, 1)
            G8_ISSUE(cB0, false, 0, 1)
            G8_ISSUE(cB1, false, 1, 1)
            G8_WAIT(0)
            G8_MFMA2(1)
        }
        else if constexpr (SCHED == 4) {
            G8_READ_B(0, 0, bf0)
            G8_READ_B(0, 1, bf1)
            G8_READ_A(0, 0)
            G8P_MARK(g8p_base + 3)
            G8_WAITN(6, 6)
            G8_ISSUE(cA1, true, 1, 1)
            G8P_MARK(g8p_base + 4)
            G8_MFMA2T(0, (g8p_base ? 40 : 60), )
            G8P_MARK(g8p_base + 5)
            G8_READ_A(0, 1)
            G8P_MARK(g8p_base + 6)
            G8_WAITN(2, 2)
            G8_ISSUE(cA0, true, 0, 0)
            G8_ISSUE(cB0, false, 0, 0)
            G8P_MARK(g8p_base + 7)
            G8_MFMA2T(1, (g8p_base ? 44 : 60), G8_ISSUE(cB1, false, 1, 0))
            G8P_MARK(g8p_base + 8)
            G8_READ_B(1, 0, bf0)
            G8_READ_B(1, 1, bf1)
            G8_READ_A(1, 0)
            G8P_MARK(g8p_base + 9)
            G8_WAITN(6, 0)
            G8_ISSUE(cA1, true, 1, 0)
            G8P_MARK(g8p_base + 10)
            G8_MFMA2T(0, (g8p_base ? 48 : 60), )
            G8P_MARK(g8p_base + 11)
            G8_READ_A(1, 1)
            G8P_MARK(g8p_base + 12)
            G8_WAITN(2, 0)
            G8_ISSUE(cA0, true, 0, 1)
            G8_ISSUE(cB0, false, 0, 1)
            G8P_MARK(g8p_base + 13)
            G8_MFMA2T(1, (g8p_base ? 52 : 60), G8_ISSUE(cB1, false, 1, 1))
            G8P_MARK(g8p_base + 14)
#ifdef IG_G8_PROF
            if (g8p_base) G8P_REAL(57)
#endif
        }
        if (++it_c < per_tile2) continue;
        // ================= tile finished: epilogue (the next tile's first K-tiles are in flight) =================
        it_c = 0;
#ifdef IG_G8_PROF
        g8p_on = blockIdx.x == 0 && g8p_tiles == 1;
        G8P_MARK(32)
#endif
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // re-align the groups: both epilogues run concurrently
        staggered = false;
        G8P_MARK(33)
        const int bm = tile_c / tiles_n, bn = tile_c - bm * tiles_n;
        tile_c += nbx;
        if (!last) G8_LOAD_BIAS(tile_c)  // next tile's accumulator init: in flight during the epilogue
        char* st = smem + G8_STAGE + wave * 4096;
        const int erow = lane & 15, eq = lane >> 4;
        const int n0 = bn * BN + wc * 64;
        if constexpr (KIND == 0) {
            // bf16 (split) store of act(acc): two 2 KiB staging slots (16 rows x 64 bf16, chunk ^= row & 7)
            const int rrow = lane >> 3, rch = ((lane & 7) ^ (lane >> 3)) << 4, rcol = (lane & 7) * 8;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int m0 = bm * BM + wr * (2 * MT * 16) + h * (MT * 16) + mt * 16;
                    uint2 po[2][2], pd[2][2], pol[2][2], pdl[2][2];
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const f32x4 a = acc[h][g][nt][mt];
                            float v[4] = {a[0], a[1], a[2], a[3]}, d[4] = {0.f, 0.f, 0.f, 0.f};
                            if constexpr (ACT == 1) {
                                f32x2 g0, g1, d0, d1;
                                gelu_erf_pair<DACT>(f32x2{v[0], v[1]}, g0, d0);
                                gelu_erf_pair<DACT>(f32x2{v[2], v[3]}, g1, d1);
                                v[0] = g0.x, v[1] = g0.y, v[2] = g1.x, v[3] = g1.y;
                                if constexpr (DACT) d[0] = d0.x, d[1] = d0.y, d[2] = d1.x, d[3] = d1.y;
                            }
                            po[g][nt].x = pack_bf2(v[0], v[1]), po[g][nt].y = pack_bf2(v[2], v[3]);
                            if constexpr (SPLIT_OUT) {
                                pol[g][nt].x = pack_bf2(v[0] - __uint_as_float(po[g][nt].x << 16), v[1] - __uint_as_float(po[g][nt].x & 0xffff0000u));
                                pol[g][nt].y = pack_bf2(v[2] - __uint_as_float(po[g][nt].y << 16), v[3] - __uint_as_float(po[g][nt].y & 0xffff0000u));
                            }
                            if constexpr (DACT) {
                                pd[g][nt].x = pack_bf2(d[0], d[1]), pd[g][nt].y = pack_bf2(d[2], d[3]);
                                if constexpr (SPLIT_OUT) {
                                    pdl[g][nt].x = pack_bf2(d[0] - __uint_as_float(pd[g][nt].x << 16), d[1] - __uint_as_float(pd[g][nt].x & 0xffff0000u));
                                    pdl[g][nt].y = pack_bf2(d[2] - __uint_as_float(pd[g][nt].y << 16), d[3] - __uint_as_float(pd[g][nt].y & 0xffff0000u));
                                }
                            }
                        }
                        // one pass = two staged arrays (X -> slot 0, Y -> slot 1) written in MFMA layout, read back row-contiguous
#define G8_STAGE_PASS(PX, PY, HAVE_Y, DSTX, DSTY, NTY)                                                                    \
    {                                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 2; ++g) _Pragma("unroll") for (int nt = 0; nt < 2; ++nt) {               \
            const int chunk_ = g * 4 + nt * 2 + (eq >> 1);                                                             \
            const int off_ = erow * 128 + ((chunk_ ^ (erow & 7)) << 4) + (eq & 1) * 8;                                 \
            *reinterpret_cast<uint2*>(st + off_) = PX[g][nt];                                                          \
            if (HAVE_Y) *reinterpret_cast<uint2*>(st + 2048 + off_) = PY[g][nt];                                       \
        }                                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                             \
            const int r_ = i_ * 8 + rrow;                                                                              \
            const uint4 ux_ = *reinterpret_cast<const uint4*>(st + r_ * 128 + rch);                                    \
            uint4 uy_ = ux_;                                                                                           \
            if (HAVE_Y) uy_ = *reinterpret_cast<const uint4*>(st + 2048 + r_ * 128 + rch);                             \
            const int m_ = m0 + r_;                                                                                    \
            if (m_ < p.M) {                                                                                            \
                const size_t o_ = (size_t)m_ * p.ldo + n0 + rcol;                                                      \
                g8_store(reinterpret_cast<uint4*>((DSTX) + o_), ux_);                                                  \
                if (HAVE_Y) g8_store<NTY>(reinterpret_cast<uint4*>((DSTY) + o_), uy_);                                 \
            }                                                                                                          \
        }                                                                                                              \
    }
                    if constexpr (SPLIT_OUT) {
                        G8_STAGE_PASS(po, pol, true, p.out_hi, p.out_lo, false)
                        if constexpr (DACT) G8_STAGE_PASS(pd, pdl, true, p.dact_hi, p.dact_lo, false)
                    } else if constexpr (DACT) {
                        G8_STAGE_PASS(po, pd, true, p.out_hi, p.dact_hi, true)
                    } else {
                        G8_STAGE_PASS(po, po, false, p.out_hi, p.out_hi, false)
                    }
#undef G8_STAGE_PASS
                }
        } else if constexpr (KIND == 2) {
            // data gradient with an elementwise factor: dx = acc * dact (the gelu' the forward saved), optional fused column
            // sums of dx (the bias gradient of the producing layer).  The fp32 tile rows are staged (16 rows x 64 fp32) and read
            // back row-contiguous, 8 columns per lane, so the factor is ONE coalesced 16-byte load per 8 values (in the MFMA
            // layout it was an 8-byte load per 4 values behind the stores).
            const int rrow = lane >> 3, rcol = (lane & 7) * 8;
            float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            constexpr int PF = SPLIT_OUT ? 2 : 4;  // row tiles per batch of factor loads (16 / 32 registers of hi [+ lo] words)
            constexpr int NRT = 2 * MT;            // row tiles of a wave: t -> (h, mt) = (t / MT, t % MT)
#pragma unroll
            for (int bt = 0; bt < NRT / PF; ++bt) {
                // the factor loads of a batch go out FIRST: interleaved with the stores below, every load sat behind the previous
                // row tile's stores in the in-order vmcnt and its latency was paid 16 times per tile
                uint4 fh[PF][2], fl[PF][2];
#pragma unroll
                for (int j = 0; j < PF; ++j)
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int t = bt * PF + j;
                        const int m = min(bm * BM + wr * (2 * MT * 16) + (t / MT) * (MT * 16) + (t % MT) * 16 + i * 8 + rrow, p.M - 1);
                        const size_t o = (size_t)m * p.ldo + n0 + rcol;
                        // gelu' was saved by the forward pass and is read exactly once: streaming (non-temporal) loads
                        typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
                        fh[j][i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p.dact_hi + o)));
                        if constexpr (SPLIT_OUT) fl[j][i] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p.dact_lo + o)));
                    }
#pragma unroll
                for (int j = 0; j < PF; ++j) {
                    const int t = bt * PF + j, h = t / MT, mt = t % MT;
                    const int m0 = bm * BM + wr * (2 * MT * 16) + h * (MT * 16) + mt * 16;
#pragma unroll
                    for (int g = 0; g < 2; ++g)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const int c4 = g * 8 + nt * 4 + eq;
                            *reinterpret_cast<f32x4*>(st + erow * 256 + ((c4 ^ (erow & 7)) << 4)) = acc[h][g][nt][mt];
                        }
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const int r = i * 8 + rrow;
                        const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + r * 256 + (((2 * (lane & 7)) ^ (r & 7)) << 4));
                        const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + r * 256 + (((2 * (lane & 7) + 1) ^ (r & 7)) << 4));
                        const int m = m0 + r;
                        float v[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        float f[8];
                        unpack8(fh[j][i], f);
                        if constexpr (SPLIT_OUT) {
                            float f2[8];
                            unpack8(fl[j][i], f2);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += f2[e];
                        }
                        if (m < p.M) {
                            const size_t o = (size_t)m * p.ldo + n0 + rcol;
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] *= f[e], cs[e] += v[e];
                            const uint4 u = pack8(v);
                            g8_store(reinterpret_cast<uint4*>(p.out_hi + o), u);
                            if constexpr (SPLIT_OUT) {
                                float hv[8], rv[8];
                                unpack8(u, hv);
#pragma unroll
                                for (int e = 0; e < 8; ++e) rv[e] = v[e] - hv[e];
                                g8_store(reinterpret_cast<uint4*>(p.out_lo + o), pack8(rv));
                            }
                        }
                    }
                }
            }
            if (p.colsum) {  // lanes with equal (lane & 7) own the same 8 columns: fold over lane >> 3, one atomic per column
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float v = cs[j];
                    v += __shfl_xor(v, 8, 64);
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    if (lane < 8) ig_red_add(p.colsum + n0 + rcol + j, v);
                }
            }
        } else {
            // fp32 residual: out = resid + acc (bias is in the accumulator init); 16 rows x 64 fp32 = 4 KiB staged.  The residual
            // rows of row tile t + 1 are loaded before row tile t is stored (their latency hides behind the staging round trip;
            // loaded after the stores they would wait for them in the in-order vmcnt).
            const int rrow = lane >> 4, rc = lane & 15;
            float4 rs[2][4];
#define G8_RESID_LOAD(T, DST)                                                                                              \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                                     \
        const int m_ = min(bm * BM + wr * (2 * MT * 16) + ((T) / MT) * (MT * 16) + ((T) % MT) * 16 + i_ * 4 + rrow, p.M - 1); \
        DST[i_] = *reinterpret_cast<const float4*>(p.resid + (size_t)m_ * p.ldo + n0 + rc * 4);                            \
    }
            G8_RESID_LOAD(0, rs[0])
#pragma unroll
            for (int t = 0; t < 2 * MT; ++t) {
                const int h = t / MT, mt = t % MT;
                const int m0 = bm * BM + wr * (2 * MT * 16) + h * (MT * 16) + mt * 16;
                if (t + 1 < 2 * MT) G8_RESID_LOAD(t + 1, rs[(t + 1) & 1])
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int c4 = g * 8 + nt * 4 + eq;
                        *reinterpret_cast<f32x4*>(st + erow * 256 + ((c4 ^ (erow & 7)) << 4)) = acc[h][g][nt][mt];
                    }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 4 + rrow;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(st + r * 256 + ((rc ^ (r & 7)) << 4));
                    const int m = m0 + r;
                    if (m < p.M) {
                        const float4 rv = rs[t & 1][i];
                        g8_store(reinterpret_cast<float4*>(p.outf + (size_t)m * p.ldo + n0 + rc * 4), make_float4(rv.x + a[0], rv.y + a[1], rv.z + a[2], rv.w + a[3]));
                    }
                }
            }
#undef G8_RESID_LOAD
        }
#ifdef IG_G8_PROF
        G8P_MARK(34)
#if IG_G8_PROF == 1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        G8P_MARK(35)
#endif
        ++g8p_tiles;
#endif
        G8_INIT_ACC()
    }
    if (staggered && wr == 0) asm volatile("s_barrier" ::: "memory");  // (unreachable in practice: every tile ends re-aligned)
#undef G8_WAIT
#undef G8_MFMA2
#undef G8_MFMA2T
#undef G8_MFMA_TERM
#undef G8_WAITN
#undef G8_READ_A
#undef G8_READ_B
#undef G8_ISSUE
#undef G8_INIT
#undef G8_REBASE
#undef G8_LOAD_BIAS
#undef G8_INIT_ACC
}

// IG_GEMM8: 0 = off, 1 = default (shapes with enough tiles), 2 = every covered shape.  Read per call (tests and A/B benches flip it).
inline int g8_env() {
    const char* e = getenv("IG_GEMM8");
    return e ? atoi(e) : 1;
}

template <int KIND, int NSEG, int ACT, bool DACT, bool SPLIT_OUT, int SCHED = 1, int MT = 4, int WC = 4>
int g8_launch_v(const G8Params& p, int grid, hipStream_t st) {
    using Geo = G8Geo<MT, WC>;
    auto kern = gemm8_kernel<KIND, NSEG, ACT, DACT, SPLIT_OUT, SCHED, MT, WC>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Geo::SMEM) != hipSuccess) {
            ig_set_error("gemm8: could not reserve %d bytes of LDS", Geo::SMEM);
            return IG_ERR_HIP;
        }
        attr_done = true;
    }
    if (MT == 4) ig_note_kernel("gemm8_kernel<%d,%d,%d,%s,%s,%d>", KIND, NSEG, ACT, DACT ? "true" : "false", SPLIT_OUT ? "true" : "false", SCHED);
    else ig_note_kernel("gemm8_kernel<%d,%d,%d,%s,%s,%d,%d,%d>", KIND, NSEG, ACT, DACT ? "true" : "false", SPLIT_OUT ? "true" : "false", SCHED, MT, WC);
    ig_note_grid(grid);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(Geo::NTHR), Geo::SMEM, st, p);
    return ig_check_launch("gemm8");
}

// small = the 128 x 128 instance (two workgroups per CU): it keeps the round-2 placement of the LDS-DMA issues (SCHED 2: with schedule 4 its
// launches were 9-10 % slower in the B = 16 step, profiles/r05_configs/step_b16_sched4_small_instance.txt: 24.8 -> 27.2 us); the 256 x 256
// instance runs schedule 4 (same-box A/B of the two: profiles/r05_gemm8_sched_ab_*.log).
template <int KIND, int NSEG, int ACT, bool DACT, bool SPLIT_OUT>
int g8_launch(const G8Params& p, int grid, hipStream_t st, bool small) {
    if (small) return g8_launch_v<KIND, NSEG, ACT, DACT, SPLIT_OUT, 2, 2, 2>(p, grid, st);
    return g8_launch_v<KIND, NSEG, ACT, DACT, SPLIT_OUT, 4>(p, grid, st);
}

}  // namespace
IG_DET_TU(gemm8)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

// IG_ERR_UNSUPPORTED (no error string) when the shape is not covered: the caller falls back to the generic engines.
int ig_gemm8_nt(const G8Params& p, void* stream) {
    if (!g8_env()) return IG_ERR_UNSUPPORTED;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0) return IG_ERR_UNSUPPORTED;
    // (Round 6, measured and removed -- profiles/r06_tail_round_split.txt: sending the row tiles of a nearly empty last round to a second launch
    // on the 128 x 128 instance.  At M = 22064 (B = 112: 783 tiles = 3.06 rounds) qkv 86.4 -> 80.6 us, fc2 127.8 -> 123.1, but proj 47.1 -> 50.1 and
    // the whole B = 112 step 4287 -> 4267 chips/s: a lone 128 x 128 tile still walks its 12 serialized K-tiles in ~25 us, as long as a round
    // of big tiles.  The batch-size cliff needs shorter tails, not smaller tiles.)
    {  // the 4-wave kernel (gemm4.hip) takes the plain bf16 shapes it is routed for
        const int rc4 = ig_gemm4_nt(p, stream);
        if (rc4 != IG_ERR_UNSUPPORTED) return rc4;
    }
    if ((p.K & 127) || (p.N & 127)) return IG_ERR_UNSUPPORTED;
    if (p.lda * 2 * 256 >= (1L << 24) || p.ldb * 2 * 256 >= (1L << 24)) return IG_ERR_UNSUPPORTED;  // 24-bit offset multiply
    const int ntiles256 = (p.N & 255) ? 0 : ((p.M + 255) >> 8) * (p.N >> 8);
    constexpr int min_tiles = 128;
    // The 256 x 256 instance needs enough tiles for one workgroup per CU.  Threshold swept on the whole step: 128 beats 192 at the
    // YAML's batch 16 (117-156 tiles: 1707 -> 1770 chips/s) and for the 300M model at B = 54 (168 tiles: 1200 -> 1279), neutral at
    // B = 32 / 48 / 108; 96 starts to lose at B = 48.  Below it (and for N = 128 mod 256) the 128 x 128 instance takes over: four
    // times the tiles, two workgroups per CU.
    const bool small = ntiles256 < min_tiles && g8_env() != 2;
    // (Round 4, measured and not kept: taking the 128 x 128 instance whenever its round count beats the big one's by a cost model --
    // 261 tiles of 256 x 256 on 256 CUs are two rounds for 1.02 rounds of work at B = 112, 1044 quarter-tiles are three rounds of 512 --
    // made the cliff WORSE: 4142 -> 4005 chips/s at B = 112, 4599 -> 4343 at B = 221; a small-instance round costs more than half a big one.)
    const int ntiles = small ? ((p.M + 127) >> 7) * (p.N >> 7) : ntiles256;
    const int grid = ig_tile_grid(ntiles, small ? 2 : 1);
    hipStream_t st = (hipStream_t)stream;
    const bool split_in = p.nseg == 3;
    // paired form of the split mode (NSEG = 2 in the kernel): both lo tensors must lie above their hi tensors, 16-byte aligned, near enough
    // for the 32-bit lane offsets (largest: 255 rows + one row + the distance).  IG_G8_PAIR=0: the three-pass form (A/B runs)
    bool pair = false;
    if (split_in) {
        const char* e = getenv("IG_G8_PAIR");
        const long dA = (const char*)p.a[2] - (const char*)p.a[0], dB = (const char*)p.b[1] - (const char*)p.b[0];
        pair = (!e || atoi(e) != 0) && dA > 0 && dB > 0 && !(dA & 15) && !(dB & 15) && dA + 257L * p.lda * 2 < (1L << 32) && dB + 257L * p.ldb * 2 < (1L << 32);
    }
    if (p.kind == 0) {
        const bool split_out = p.out_lo != nullptr;
        const bool dact = p.dact_hi != nullptr;
        if (split_in != split_out) return IG_ERR_UNSUPPORTED;
        if (p.act == 0 && dact) return IG_ERR_UNSUPPORTED;
        if (!split_in) {
            if (p.act == 0) return g8_launch<0, 1, 0, false, false>(p, grid, st, small);
            if (!dact) return g8_launch<0, 1, 1, false, false>(p, grid, st, small);
            return g8_launch<0, 1, 1, true, false>(p, grid, st, small);
        }
        if (pair) {
            if (p.act == 0) return g8_launch<0, 2, 0, false, true>(p, grid, st, small);
            if (!dact) return g8_launch<0, 2, 1, false, true>(p, grid, st, small);
            return g8_launch<0, 2, 1, true, true>(p, grid, st, small);
        }
        if (p.act == 0) return g8_launch<0, 3, 0, false, true>(p, grid, st, small);
        if (!dact) return g8_launch<0, 3, 1, false, true>(p, grid, st, small);
        return g8_launch<0, 3, 1, true, true>(p, grid, st, small);
    }
    if (p.kind == 1) {
        if (!split_in) return g8_launch<1, 1, 0, false, false>(p, grid, st, small);
        if (pair) return g8_launch<1, 2, 0, false, false>(p, grid, st, small);
        return g8_launch<1, 3, 0, false, false>(p, grid, st, small);
    }
    if (p.kind == 2) {
        if (!p.dact_hi || (split_in != (p.out_lo != nullptr)) || (split_in != (p.dact_lo != nullptr))) return IG_ERR_UNSUPPORTED;
        if (!split_in) return g8_launch<2, 1, 0, false, false>(p, grid, st, small);
        if (pair) return g8_launch<2, 2, 0, false, true>(p, grid, st, small);
        return g8_launch<2, 3, 0, false, true>(p, grid, st, small);
    }
    return IG_ERR_UNSUPPORTED;
}
