// v8w: the weight gradients of the timm Block linears (pritvhi.py:446-456: qkv / proj / fc1 / fc2; autograd's grad_weight of
// F.linear) as ONE grouped launch on the 256 x 256 x 64 "8-phase" schedule of gemm8.hip.
//
//   dW_g[n][k] += sum_m dy_g[m][n] * x_g[m][k]          g = 0 .. ng-1, all GEMMs of a group share the token count M
//
// * The reduction runs over tokens, so BOTH operands are reduce-strided ("TR"): an operand half-tile in LDS is
//   [64 tokens][128 columns] (256-byte rows, 16 KiB), filled by LDS-DMA in full 256-byte source rows (two cache lines per
//   token and half-tile), and the MFMA fragments come from ds_read_b64_tr_b16 (hardware transpose).  The bank swizzle is the
//   chunk-pair key of gemm.hip's TR image (lds_trw<16>), applied on the DMA source chunk and on the fragment reads.
// * Work = (output tile, token range) SEGMENTS from a host-built table: the tiles of all GEMMs of the group form one list, so
//   a group of four GEMMs (108 tiles at D = 768) runs with 2 token splits on 216 CUs where four separate launches needed
//   7-28 splits each -- the split-K fold (slab stores + ordered reduce) shrinks from 4 x 32 MB to 54 MB per block and every
//   workgroup runs a 160-K-tile main loop instead of 12-48.
// * Deterministic: every segment stores its fp32 partial tile to its own slab in the accumulator's own (lane-linear) layout
//   -- 16-byte coalesced stores, no LDS staging -- and wgrad8_reduce_kernel adds a tile's slabs in slab order.
// * Same barrier / vmcnt protocol as gemm8_kernel (SCHED 2): two big phases of 32 MFMAs per K-tile, the two row groups one
//   barrier apart, counted vmcnt(8), a half-tile refilled the phase after its last read was retired.
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#include "common.h"

namespace {

constexpr int W_HALF = 16384;  // one half-tile: 64 tokens x 128 columns
constexpr int W_BUF = 65536;   // A0 A1 B0 B1
constexpr int W_SMEM = 131072;
constexpr int W_MAXSEG = 6;    // segments per workgroup
typedef __attribute__((address_space(3))) char* lds_char_ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

struct WSeg {  // one (workgroup, output tile, token range); 64 bytes
    const char* a[2];  // dy hi / lo at (first token, tile row block): the columns of dy are the ROWS of dW
    const char* b[2];  // x hi / lo at (first token, tile column block)
    long slab;         // index of the segment's 256 x 256 fp32 partial tile in the slab workspace
    int lda2, ldb2;    // row pitch of dy / x in bytes
    int nkt;           // K-tiles (64 tokens) of this segment, even; 0 = end of this workgroup's list
    int rows;          // valid tokens from the first one (< 64 nkt on the ragged tail: the rest reads the zero page)
    long pad;
};
static_assert(sizeof(WSeg) == 64, "WSeg layout");

struct WTile {  // reduce table
    float* out;  // dW at the tile origin
    long first;  // first slab of this tile (its slabs are consecutive, in token order)
    int ldo, nslab;
    long pad;
};
static_assert(sizeof(WTile) == 32, "WTile layout");

__device__ __forceinline__ void w_glds_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void w_glds_v(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

struct WCur {  // issue cursor of one half-tile type (all wave-uniform)
    const char* base;
    int kt, rows, ld2, seg, pseg, left;
};

template <int NSEG>
__global__ __launch_bounds__(512, 2) void gemm8w_kernel(const WSeg* __restrict__ segs, float* __restrict__ slabs,
                                                        const bf16_t* __restrict__ zero_page) {
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
    // A: column wr*64 + mt*16 + 4pp of the half -> 16-byte chunk wr*8 + mt*2 + (pp>>1); chunk pair index (wr<<2 | mt) ^ key
    const int aoff = k0off + ((((wr << 2) ^ rkey) << 5) | rlow);              // ^ (mt << 5)
    // B: column wc*32 + nt*16 + 4pp -> chunk wc*4 + nt*2 + (pp>>1); chunk pair index (wc<<1 | nt) ^ key
    const int boff = 2 * W_HALF + k0off + ((((wc << 1) ^ rkey) << 5) | rlow);  // ^ (nt << 5)
    // ---- LDS-DMA lane constants: instruction i of this wave fills token rows wave*8 + i*4 + (lane >> 4) ----
    const int krow = wave * 8 + (lane >> 4);
    const int dkey = (lane >> 4) | ((wave & 1) << 2);
    const int lch16 = ((lane & 15) ^ (dkey << 1)) << 4;  // logical source chunk of this lane's physical chunk
    const unsigned ldsw = lds_base + wave * 2048;

    WCur cA0, cA1, cB0, cB1;
#define W_REBASE(C, ISA)                                                                      \
    {                                                                                         \
        const WSeg* s_ = my + (C).seg;                                                        \
        const int pa_ = NSEG == 1 ? 0 : ((C).pseg == 2 ? 1 : 0);                              \
        const int pb_ = NSEG == 1 ? 0 : ((C).pseg == 1 ? 1 : 0);                              \
        (C).base = (ISA) ? (pa_ ? s_->a[1] : s_->a[0]) : (pb_ ? s_->b[1] : s_->b[0]);         \
        (C).ld2 = (ISA) ? s_->lda2 : s_->ldb2;                                                \
        (C).kt = s_->nkt;                                                                     \
        (C).rows = s_->rows;                                                                  \
    }
#define W_INIT(C, ISA)                                    \
    {                                                     \
        (C).seg = 0, (C).pseg = 0, (C).left = Gtot;       \
        W_REBASE(C, ISA)                                  \
    }
    // the two wave-instructions of this wave for half-tile (ISA ? A : B, HG) of the cursor's K-tile into buffer BUF
#define W_ISSUE(C, ISA, HG, BUF)                                                                                  \
    if ((C).left > 0) {                                                                                           \
        const unsigned dst_ = ldsw + (BUF)*W_BUF + ((ISA) ? 0 : 2 * W_HALF) + (HG)*W_HALF;                        \
        if ((C).rows >= 64) {                                                                                     \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
                const unsigned voff_ = (unsigned)(__mul24(krow + i_ * 4, (C).ld2) + (HG)*256 + lch16);            \
                w_glds_s(voff_, (C).base, dst_ + i_ * 1024);                                                      \
            }                                                                                                     \
        } else { /* ragged tail of the token range: rows past the end come from the zero page */                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                    \
                const int r_ = krow + i_ * 4;                                                                     \
                const char* p_ = r_ < (C).rows ? (C).base + (unsigned)(__mul24(r_, (C).ld2) + (HG)*256 + lch16)   \
                                               : (const char*)zero_page + ((lane & 15) << 4);                     \
                w_glds_v(p_, dst_ + i_ * 1024);                                                                   \
            }                                                                                                     \
        }                                                                                                         \
        (C).left--;                                                                                               \
        (C).base += (long)(C).ld2 << 6;                                                                           \
        (C).rows -= 64;                                                                                           \
        if (--(C).kt == 0) {                                                                                      \
            if (NSEG == 1 || ++(C).pseg == NSEG) {                                                                \
                (C).pseg = 0;                                                                                     \
                (C).seg++;                                                                                        \
            }                                                                                                     \
            if ((C).left > 0) W_REBASE(C, ISA)                                                                    \
        }                                                                                                         \
    }

    f32x4 acc[2][2][2][4];  // [h][g][nt][mt]
#define W_ZERO_ACC()                                                                                             \
    _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)            \
        _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_)  \
            acc[h_][g_][nt_][mt_] = f32x4{0.f, 0.f, 0.f, 0.f};
    W_ZERO_ACC()

    // prologue: K-tile 0 complete + A0, B0, B1 of K-tile 1 (the state the steady-state schedule leaves behind)
    W_INIT(cA0, true)
    W_INIT(cA1, true)
    W_INIT(cB0, false)
    W_INIT(cB1, false)
    W_ISSUE(cA0, true, 0, 0)
    W_ISSUE(cB0, false, 0, 0)
    W_ISSUE(cB1, false, 1, 0)
    W_ISSUE(cA1, true, 1, 0)
    W_ISSUE(cA0, true, 0, 1)
    W_ISSUE(cB0, false, 0, 1)
    W_ISSUE(cB1, false, 1, 1)
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    bf16x8_t af[4][2], bf0[2][2], bf1[2][2];
#define W_TR(DST, ADDR)                                                                                           \
    {                                                                                                             \
        const s16x4 v0_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (ADDR)));                \
        const s16x4 v1_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(smem + (ADDR) + 1024));         \
        typedef __attribute__((ext_vector_type(8))) short s16x8_;                                                 \
        const s16x8_ r_ = {v0_[0], v0_[1], v0_[2], v0_[3], v1_[0], v1_[1], v1_[2], v1_[3]};                       \
        DST = __builtin_bit_cast(bf16x8_t, r_);                                                                   \
    }
#define W_READ_A(BUF, H)                                                                                          \
    _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)          \
        W_TR(af[mt_][s_], (BUF)*W_BUF + (H)*W_HALF + s_ * 8192 + (aoff ^ (mt_ << 5)))
#define W_READ_B(BUF, G, DST)                                                                                     \
    _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)          \
        W_TR(DST[nt_][s_], (BUF)*W_BUF + (G)*W_HALF + s_ * 8192 + (boff ^ (nt_ << 5)))
    // "big phase": 32 MFMAs (two quadrants) between one barrier pair; reads are retired BEFORE the first barrier
#define W_MFMA2(H)                                                                                                \
    {                                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
        asm volatile("s_barrier" ::: "memory");                                                                   \
        __builtin_amdgcn_s_setprio(1);                                                                            \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_) _Pragma("unroll") for (int nt_ = 0; nt_ < 2; ++nt_)      \
            _Pragma("unroll") for (int mt_ = 0; mt_ < 4; ++mt_) _Pragma("unroll") for (int s_ = 0; s_ < 2; ++s_)  \
                acc[H][g_][nt_][mt_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(g_ ? bf1[nt_][s_] : bf0[nt_][s_], af[mt_][s_], acc[H][g_][nt_][mt_], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                            \
        asm volatile("s_barrier" ::: "memory");                                                                   \
    }
#define W_WAIT(LASTCNT)                                                          \
    {                                                                            \
        if (last) asm volatile("s_waitcnt vmcnt(" #LASTCNT ")" ::: "memory");    \
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                    \
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
        W_READ_B(0, 0, bf0)
        W_READ_B(0, 1, bf1)
        W_READ_A(0, 0)
        W_ISSUE(cA1, true, 1, 1)
        W_WAIT(8)
        W_MFMA2(0)
        W_READ_A(0, 1)
        W_ISSUE(cA0, true, 0, 0)
        W_ISSUE(cB0, false, 0, 0)
        W_ISSUE(cB1, false, 1, 0)
        W_WAIT(2)
        W_MFMA2(1)
        W_READ_B(1, 0, bf0)
        W_READ_B(1, 1, bf1)
        W_READ_A(1, 0)
        W_ISSUE(cA1, true, 1, 0)
        W_WAIT(0)
        W_MFMA2(0)
        W_READ_A(1, 1)
        W_ISSUE(cA0, true, 0, 1)
        W_ISSUE(cB0, false, 0, 1)
        W_ISSUE(cB1, false, 1, 1)
        W_WAIT(0)
        W_MFMA2(1)
        if (++it_c < seg_iters) continue;
        // ===== segment finished: store the partial tile (the next segment's first K-tiles are in flight) =====
        it_c = 0;
        if (wr == 0) asm volatile("s_barrier" ::: "memory");  // re-align the groups: both store bursts run concurrently
        staggered = false;
        f32x4* sl = reinterpret_cast<f32x4*>(slabs + my[cseg].slab * 65536L) + tid;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) sl[(((h * 2 + g) * 2 + nt) * 4 + mt) * 512] = acc[h][g][nt][mt];
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
#undef W_ISSUE
#undef W_INIT
#undef W_REBASE
#undef W_ZERO_ACC
}

// dW tile (+)= sum of its slabs, in slab (= token) order.  Slab element e = a * 512 + tid holds accumulator a = ((h*2+g)*2+nt)*4+mt
// of thread tid: dW rows h*128 + wr*64 + mt*16 + (lane & 15), columns g*128 + wc*32 + nt*16 + 4 (lane >> 4) .. +3.
__global__ __launch_bounds__(256) void wgrad8_reduce_kernel(const WTile* __restrict__ tiles, const float4* __restrict__ slabs, int overwrite) {
    const WTile T = tiles[blockIdx.y];
    const int e = blockIdx.x * 256 + threadIdx.x;  // 0 .. 16383
    const float4* p = slabs + T.first * 16384L + e;
    float4 s = p[0];
    for (int k = 1; k < T.nslab; ++k) {
        const float4 t = p[(long)k * 16384L];
        s.x += t.x, s.y += t.y, s.z += t.z, s.w += t.w;
    }
    const int a = e >> 9, tid = e & 511, wave = tid >> 6, lane = tid & 63;
    const int row = (a >> 4) * 128 + (wave >> 2) * 64 + (a & 3) * 16 + (lane & 15);
    const int col = ((a >> 3) & 1) * 128 + (wave & 3) * 32 + ((a >> 2) & 1) * 16 + 4 * (lane >> 4);
    float4* o = reinterpret_cast<float4*>(T.out + (long)row * T.ldo + col);
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
};
typedef std::vector<long> WKey;
std::map<WKey, WPlan>& plan_cache() {
    static std::map<WKey, WPlan> c;
    return c;
}
// slab workspace, per device: grown on demand; superseded buffers are kept (a captured graph may still hold their address) and
// nothing is allocated while a stream capture is active (the first eager step has sized it by then)
float* slab_workspace(long nslab, hipStream_t st) {
    constexpr int kMaxDev = 16;
    static float* ws[kMaxDev] = {};
    static long cap[kMaxDev] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
    if (nslab > cap[dev]) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return nullptr;
        void* p = nullptr;
        if (hipMalloc(&p, (size_t)nslab * 65536 * sizeof(float)) != hipSuccess) return nullptr;
        ws[dev] = (float*)p, cap[dev] = nslab;
    }
    return ws[dev];
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

inline int wg8_rem_env() {  // IG_WGRAD8_REM=0: uniform token splits only (A/B runs)
    const char* e = getenv("IG_WGRAD8_REM");
    return e ? atoi(e) : 1;
}

}  // namespace

// Grouped weight gradients.  IG_ERR_UNSUPPORTED (no error string) when a shape is not covered: the caller falls back to one
// ig_linear_wgrad per GEMM.
int ig_wgrad8_group(int n, const void* const* dy_hi, const void* const* dy_lo, const void* const* x_hi, const void* const* x_lo,
                    float* const* dw, const int* N, const int* K, int M, int overwrite, void* stream) {
    if (!wg8_env() || n <= 0 || M <= 0) return IG_ERR_UNSUPPORTED;
    const bool split = dy_lo && dy_lo[0];
    long ntiles = 0;
    for (int g = 0; g < n; ++g) {
        if (N[g] <= 0 || K[g] <= 0 || (N[g] & 255) || (K[g] & 255)) return IG_ERR_UNSUPPORTED;
        if ((long)N[g] * 2 * 64 >= (1L << 24) || (long)K[g] * 2 * 64 >= (1L << 24)) return IG_ERR_UNSUPPORTED;  // 24-bit offset multiply
        if ((((uintptr_t)dy_hi[g]) | ((uintptr_t)x_hi[g]) | ((uintptr_t)dw[g])) & 15) return IG_ERR_UNSUPPORTED;
        if (split != ((dy_lo && dy_lo[g]) != 0) || split != ((x_lo && x_lo[g]) != 0)) return IG_ERR_UNSUPPORTED;
        if (split && ((((uintptr_t)dy_lo[g]) | ((uintptr_t)x_lo[g])) & 15)) return IG_ERR_UNSUPPORTED;
        ntiles += (long)(N[g] >> 8) * (K[g] >> 8);
    }
    int ncu = ig_cu_count() - ig_reserved_cus();
    if (ncu < 8) ncu = 8;
    if (ntiles > (long)ncu * W_MAXSEG) return IG_ERR_UNSUPPORTED;
    WKey key;
    int dev_ = 0;
    (void)hipGetDevice(&dev_);
    key.push_back(n), key.push_back(M), key.push_back(ncu), key.push_back(split), key.push_back(wg8_rem_env()), key.push_back(dev_);
    for (int g = 0; g < n; ++g) {
        key.push_back((long)(uintptr_t)dy_hi[g]), key.push_back((long)(uintptr_t)(split ? dy_lo[g] : nullptr));
        key.push_back((long)(uintptr_t)x_hi[g]), key.push_back((long)(uintptr_t)(split ? x_lo[g] : nullptr));
        key.push_back((long)(uintptr_t)dw[g]), key.push_back(N[g]), key.push_back(K[g]);
    }
    hipStream_t st = (hipStream_t)stream;
    auto& cache = plan_cache();
    auto it = cache.find(key);
    if (it == cache.end()) {
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
        struct TileRef { int g, tm, tn; };
        std::vector<TileRef> tl;
        for (int g = 0; g < n; ++g)
            for (int tm = 0; tm < (N[g] >> 8); ++tm)
                for (int tn = 0; tn < (K[g] >> 8); ++tn) tl.push_back({g, tm, tn});
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
        if (rem) {
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
            for (long q0 = 0; q0 < Q; q0 += per_wg) {
                std::vector<SegRef> v;
                for (long qi = q0; qi < q0 + per_wg && qi < Q; ++qi) {
                    const int sp = (int)(qi / T), t = (int)(qi - (long)sp * T);
                    v.push_back(SegRef{t, sp * chunk, (P - sp * chunk) < chunk ? (P - sp * chunk) : chunk});
                }
                wl.push_back(v);
            }
        }
        const long nq = (long)wl.size();
        const int nwg = 8 * (int)((nq + 7) / 8);
        std::vector<WSeg> hs((size_t)nwg * W_MAXSEG);
        memset(hs.data(), 0, hs.size() * sizeof(WSeg));
        const int spt = ks + 2;  // slab slots per tile (a tile's remainder may be cut once by a workgroup boundary)
        std::vector<std::vector<std::pair<int, long>>> tslabs(T);  // per tile: (first pair, slab) of its segments
        std::vector<int> tcount(T, 0);
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
                const long ao = (tok0 * N[tr.g] + (long)tr.tm * 256) * 2, bo = (tok0 * K[tr.g] + (long)tr.tn * 256) * 2;
                d.a[0] = (const char*)dy_hi[tr.g] + ao, d.a[1] = split ? (const char*)dy_lo[tr.g] + ao : d.a[0];
                d.b[0] = (const char*)x_hi[tr.g] + bo, d.b[1] = split ? (const char*)x_lo[tr.g] + bo : d.b[0];
                d.slab = (long)v[si].t * spt + tcount[v[si].t]++;
                d.lda2 = N[tr.g] * 2, d.ldb2 = K[tr.g] * 2;
                d.nkt = v[si].np * 2;
                const long left = (long)M - tok0;
                d.rows = (int)(left < (long)d.nkt * 64 ? left : (long)d.nkt * 64);
            }
        }
        std::vector<WTile> ht(ntiles);
        for (long t = 0; t < ntiles; ++t) {
            const TileRef& tr = tl[t];
            ht[t].out = dw[tr.g] + (long)tr.tm * 256 * K[tr.g] + (long)tr.tn * 256;
            ht[t].first = t * spt, ht[t].ldo = K[tr.g], ht[t].nslab = tcount[t], ht[t].pad = 0;
            if (tcount[t] > spt || tcount[t] < 1) {
                ig_set_error("ig_linear_wgrad_group: internal plan error (tile %ld has %d segments)", t, tcount[t]);
                return IG_ERR_ARG;
            }
        }
        WPlan pl;
        pl.nwg = nwg, pl.ntiles = (int)ntiles, pl.nslab = ntiles * spt;
        if (hipMalloc((void**)&pl.segs, hs.size() * sizeof(WSeg)) != hipSuccess ||
            hipMalloc((void**)&pl.tiles, ht.size() * sizeof(WTile)) != hipSuccess) {
            ig_set_error("ig_linear_wgrad_group: could not allocate the segment tables");
            return IG_ERR_HIP;
        }
        // synchronous copies from host vectors (first call of a configuration only)
        if (hipMemcpy(pl.segs, hs.data(), hs.size() * sizeof(WSeg), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(pl.tiles, ht.data(), ht.size() * sizeof(WTile), hipMemcpyHostToDevice) != hipSuccess) {
            ig_set_error("ig_linear_wgrad_group: could not upload the segment tables");
            return IG_ERR_HIP;
        }
        it = cache.emplace(key, pl).first;
    }
    const WPlan& pl = it->second;
    float* ws = slab_workspace(pl.nslab, st);
    const bf16_t* zp = w_zero_page();
    if (!ws || !zp) {
        ig_set_error("ig_linear_wgrad_group: could not allocate the slab workspace (%ld slabs)", pl.nslab);
        return IG_ERR_HIP;
    }
    static bool attr_done[2] = {false, false};
    auto k1 = gemm8w_kernel<1>;
    auto k3 = gemm8w_kernel<3>;
    if (!attr_done[split]) {
        if (hipFuncSetAttribute(split ? (const void*)k3 : (const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM) != hipSuccess) {
            ig_set_error("gemm8w: could not reserve %d bytes of LDS", W_SMEM);
            return IG_ERR_HIP;
        }
        attr_done[split] = true;
    }
    ig_note_kernel("gemm8w_kernel<%d>", split ? 3 : 1);
    ig_note_grid(pl.nwg);
    if (split) hipLaunchKernelGGL(k3, dim3(pl.nwg), dim3(512), W_SMEM, st, (const WSeg*)pl.segs, ws, zp);
    else hipLaunchKernelGGL(k1, dim3(pl.nwg), dim3(512), W_SMEM, st, (const WSeg*)pl.segs, ws, zp);
    hipLaunchKernelGGL(wgrad8_reduce_kernel, dim3(64, pl.ntiles), dim3(256), 0, st, (const WTile*)pl.tiles, (const float4*)ws, overwrite);
    return ig_check_launch("ig_linear_wgrad_group");
}
