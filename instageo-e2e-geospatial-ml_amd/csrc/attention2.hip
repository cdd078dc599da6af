// Attention forward, second generation (head_dim 64): F.scaled_dot_product_attention of timm's Attention (pritvhi.py:446-456).
//
// The first kernel (attention.hip) gives every wave a 16-query tile and spends ~96 VALU instructions per 8 MFMAs: it is bound
// by instruction issue (9 % of the MFMA peak).  This one is built around v_mfma_f32_32x32x16_bf16:
// * a workgroup = 7 waves = 224 queries of one (batch, head); a wave owns 32 queries (Q fragments live in registers);
// * K and V of the head are brought into LDS by LDS-DMA in full 128-byte rows (one global_load_lds_dwordx4 = 8 keys x 128 B),
//   224 keys per chunk -- the whole head for T = 1 (197 tokens), three chunks for T = 3 (589 tokens).  The bank swizzles are
//   applied on the SOURCE chunk: K rows are read with ds_read_b128 by 32 lanes = 32 different keys (chunk ^= (key >> 1) & 7 is
//   conflict-free for that lane grouping), V with ds_read_b64_tr_b16 (64-byte halves flipped by key bit 1);
// * S^T[key][query] = K Q^T puts the query on the lane and the 32 keys of a tile in the 16 accumulator registers of the two
//   lane halves: row max / row sum are 15 in-lane operations + ONE v_permlane32_swap, and the probability tile is directly the
//   B operand of the next product O^T[d][query] += V^T[d][key] P^T[key][query] (k order permuted consistently on the V^T
//   fragments: element j of lane half h = key 16 s + 8 (j >> 2) + 4 h + (j & 3));
// * online softmax in the base-2 domain with a LAZY rescale: the running maximum only moves (and O, l are only rescaled) when
//   a tile's maximum exceeds it by more than 2^8 -- a wave-uniform branch that is taken on the first tile and then almost never;
// * SPLIT = the bf16x3 precision mode (hi*hi + hi*lo + lo*hi for both products; K/V hi and lo images in LDS).
#include <stdlib.h>

#include "common.h"

#ifdef IG_A2_PROF
// timing instrumentation of attn2_bwd_fused_kernel (debug builds only: tools/attn_phase_prof.py): s_memtime at phase boundaries of the
// waves of workgroup (0, 0, 0); MARKV makes the clock read wait for a value (an MFMA result), so the phase includes the pipeline drain
__device__ unsigned long long g_a2prof[8 * 48];
#define A2P_INIT_X const bool a2p_on = blockIdx.x == 0 && blockIdx.y == 1 && blockIdx.z == 1 && (threadIdx.x & 63) == 0;
#define A2P_MARK(I) { if (a2p_on) g_a2prof[(threadIdx.x >> 6) * 48 + (I)] = __builtin_readcyclecounter(); }
#define A2P_MARKV(I, V) { float a2p_v = (V); asm volatile("" ::"v"(a2p_v)); if (a2p_on) g_a2prof[(threadIdx.x >> 6) * 48 + (I)] = __builtin_readcyclecounter(); }
extern "C" int ig_debug_a2prof(unsigned long long* dst) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_a2prof), sizeof(g_a2prof)) == hipSuccess ? 0 : -1; }
#else
#define A2P_INIT_X
#define A2P_MARK(I)
#define A2P_MARKV(I, V)
#endif

namespace {

constexpr int A2_WAVES = 7, A2_THREADS = 448, A2_QB = 224, A2_CH = 224;
constexpr int A2_TILE = A2_CH * 128;  // one K or V image: 224 keys x 128 B
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
typedef __attribute__((address_space(3))) char* lds_char_ptr;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ int k_swz(int row) { return (row >> 1) & 7; }           // K image: 16-byte chunk ^= this
__device__ __forceinline__ int v_swz(int row) { return ((row >> 1) & 1) << 2; }    // V image: flips the 64-byte half

__device__ __forceinline__ f32x16 mfma32(bf16x8_t a, bf16x8_t b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
template <bool SPLIT>
__device__ __forceinline__ f32x16 mma3(bf16x8_t ah, bf16x8_t al, bf16x8_t bh, bf16x8_t bl, f32x16 c) {
    c = mfma32(ah, bh, c);
    if constexpr (SPLIT) {
        c = mfma32(ah, bl, c);
        c = mfma32(al, bh, c);
    }
    return c;
}

// One token row of 64 head-dim columns from two 32x32 accumulator tiles (lane half lh holds columns 32 dh + 8 g + 4 lh .. + 3 in
// registers 4 g .. 4 g + 3) as 16-byte stores: per pair of groups (g = 2 gp, 2 gp + 1) the lane halves exchange one 4-column piece --
// v_permlane32_swap(x, y): the lower lanes receive x of lane + 32 in result[1], the upper lanes y of lane - 32 in result[0] -- then
// half 0 owns the 8 columns of the even group and half 1 those of the odd one: 4 store instructions of 32 rows x 32 contiguous bytes
// instead of 8 of 32 rows x 16 (forward 70.5 -> 65.7 us at B = 216).  Both lanes of a pair must be active (same token row).
template <bool SPLIT>
__device__ __forceinline__ void a2_store_row64(bf16_t* hi, bf16_t* lo, size_t orow, const f32x16 (&acc)[2], float scale, int lh) {
#pragma unroll
    for (int dh = 0; dh < 2; ++dh)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            float va[4], vc[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) va[e] = acc[dh][8 * gp + e] * scale, vc[e] = acc[dh][8 * gp + 4 + e] * scale;
            const uint32_t a0 = pack_bf2(va[0], va[1]), a1 = pack_bf2(va[2], va[3]), c0 = pack_bf2(vc[0], vc[1]), c1 = pack_bf2(vc[2], vc[3]);
            {
                const auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1, false, false);
                *reinterpret_cast<uint4*>(hi + orow + 32 * dh + 16 * gp + 8 * lh) = lh == 0 ? make_uint4(a0, a1, s0[1], s1[1]) : make_uint4(s0[0], s1[0], c0, c1);
            }
            if constexpr (SPLIT) {
                const uint32_t la0 = pack_bf2(va[0] - __uint_as_float(a0 << 16), va[1] - __uint_as_float(a0 & 0xffff0000u));
                const uint32_t la1 = pack_bf2(va[2] - __uint_as_float(a1 << 16), va[3] - __uint_as_float(a1 & 0xffff0000u));
                const uint32_t lc0 = pack_bf2(vc[0] - __uint_as_float(c0 << 16), vc[1] - __uint_as_float(c0 & 0xffff0000u));
                const uint32_t lc1 = pack_bf2(vc[2] - __uint_as_float(c1 << 16), vc[3] - __uint_as_float(c1 & 0xffff0000u));
                const auto s0 = __builtin_amdgcn_permlane32_swap(la0, lc0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(la1, lc1, false, false);
                *reinterpret_cast<uint4*>(lo + orow + 32 * dh + 16 * gp + 8 * lh) = lh == 0 ? make_uint4(la0, la1, s0[1], s1[1]) : make_uint4(s0[0], s1[0], lc0, lc1);
            }
        }
}

template <bool SPLIT>
__global__ __launch_bounds__(A2_THREADS) void attn2_fwd_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                               bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                               float* __restrict__ lse, int N, int H, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_hi = smem;
    char* v_hi = smem + A2_TILE;
    char* k_lo = smem + 2 * A2_TILE;  // SPLIT only
    char* v_lo = smem + 3 * A2_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * 64;  // elements between consecutive tokens of qkv [B][N][3][H][64]
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * 64;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * 64 : nullptr;
    const int q0 = blockIdx.x * A2_QB + wave * 32;
    const bool active = q0 < N;  // wave-uniform: a wave without queries still helps to load and keeps the barriers

    // Q^T fragments (B operand): lane holds Q[q0 + lr][16 s + 8 lh + j]
    bf16x8_t qh[4], ql[4];
    {
        const int qr = min(q0 + lr, N - 1);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(base_hi + (long)qr * RS + 16 * s + 8 * lh));
            if constexpr (SPLIT) ql[s] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(base_lo + (long)qr * RS + 16 * s + 8 * lh));
            else ql[s] = qh[s];
        }
    }
    f32x16 o[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = scale * 1.44269504088896340736f;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    for (int c0 = 0; c0 < N; c0 += A2_CH) {
        if (c0 > 0) __syncthreads();  // the previous chunk is fully consumed
        // ---- LDS-DMA: 28 wave-instructions per image (8 keys x 128 B each), 4 per wave ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (wave * 4 + i) * 8 + (lane >> 3);  // key inside the chunk
            const int key = min(c0 + row, N - 1);
            const long goff = (long)key * RS;
            const int ck = ((lane & 7) ^ k_swz(row)) * 8, cv = ((lane & 7) ^ v_swz(row)) * 8;  // logical element offsets
            const unsigned dst = lds_base + (wave * 4 + i) * 1024;
            glds16(base_hi + goff + H * 64 + ck, dst);
            glds16(base_hi + goff + 2 * H * 64 + cv, dst + A2_TILE);
            if constexpr (SPLIT) {
                glds16(base_lo + goff + H * 64 + ck, dst + 2 * A2_TILE);
                glds16(base_lo + goff + 2 * H * 64 + cv, dst + 3 * A2_TILE);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!active) continue;
        const int ntile = min(A2_CH / 32, (N - c0 + 31) >> 5);
        // software pipeline: the S^T MFMAs of tile kt + 1 are issued BEFORE the softmax of tile kt -- the matrix pipe works on
        // them while the wave issues the ~90 VALU instructions of the softmax (one wave's MFMA and VALU streams overlap only if
        // independent instructions are adjacent in program order)
#define A2_QK(KT, DST)                                                                                       \
    {                                                                                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) DST[r_] = 0.f;                                     \
        const int krow_ = (KT)*32 + lr;                                                                      \
        _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) {                                                   \
            const int off_ = krow_ * 128 + (((2 * s_ + lh) ^ k_swz(krow_)) << 4);                            \
            const bf16x8_t kh_ = *reinterpret_cast<const bf16x8_t*>(k_hi + off_);                            \
            const bf16x8_t kl_ = SPLIT ? *reinterpret_cast<const bf16x8_t*>(k_lo + off_) : kh_;              \
            DST = mma3<SPLIT>(kh_, kl_, qh[s_], ql[s_], DST);                                                \
        }                                                                                                    \
    }
        f32x16 st_next;
        A2_QK(0, st_next)
        for (int kt = 0; kt < ntile; ++kt) {
            f32x16 st = st_next;
            if (kt + 1 < ntile) A2_QK(kt + 1, st_next)
            const int key0 = c0 + kt * 32;
            if (key0 + 32 > N) {  // tail tile: keys >= N carry no weight
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + 8 * (r >> 2) + 4 * lh + (r & 3) >= N) st[r] = -INFINITY;
            }
            // ---- online softmax, base 2, lazy rescale ----
            float mx = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, st[r]);
            {
                const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
                mx = fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
            }
            const float m_tile = mx * c2;
            if (__builtin_amdgcn_ballot_w64(m_tile > m_run + 8.0f) != 0) {  // wave-uniform; always on the first tile (m_run = -inf)
                const float m_new = fmaxf(m_run, m_tile);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                l_run *= alpha;
                m_run = m_new;
            }
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -m_run));
                st[r] = pv;
                psum += pv;
            }
            l_run += psum;
            // ---- P^T as the B operand of the next product: registers 8 s .. 8 s + 7 = k-step s ----
            bf16x8_t ph[2], pl[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float f[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = st[8 * s + j];
                const uint4 u = pack8(f);
                ph[s] = __builtin_bit_cast(bf16x8_t, u);
                if constexpr (SPLIT) {
                    float hv[8], rv[8];
                    unpack8(u, hv);
#pragma unroll
                    for (int j = 0; j < 8; ++j) rv[j] = f[j] - hv[j];
                    pl[s] = __builtin_bit_cast(bf16x8_t, pack8(rv));
                } else {
                    pl[s] = ph[s];
                }
            }
            // ---- O^T[d][query] += V^T[d][key] P^T[key][query]; V^T fragments by hardware-transposed reads ----
            const int g16 = lane >> 4, i16 = lane & 15;
#pragma unroll
            for (int dh = 0; dh < 2; ++dh)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    // lane (group g16, i16): 4 keys x 16 d block; d = 32 dh + 16 (g16 & 1) + i16, lane half = g16 >> 1
                    const int r0 = kt * 32 + 16 * s + 4 * (g16 >> 1) + (i16 >> 2);
                    const int r1 = r0 + 8;
                    const int cb = (32 * dh + 16 * (g16 & 1) + 4 * (i16 & 3)) * 2;  // byte offset of the 4-column piece in the row
                    const int o0 = r0 * 128 + ((((cb >> 4) ^ v_swz(r0)) << 4) | (cb & 15));
                    const int o1 = r1 * 128 + ((((cb >> 4) ^ v_swz(r1)) << 4) | (cb & 15));
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(v_hi + o0));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(v_hi + o1));
                    const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    bf16x8_t vh = __builtin_bit_cast(bf16x8_t, av), vl = vh;
                    if constexpr (SPLIT) {
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(v_lo + o0));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(v_lo + o1));
                        const s16x8 bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        vl = __builtin_bit_cast(bf16x8_t, bv);
                    }
                    o[dh] = mma3<SPLIT>(vh, vl, ph[s], pl[s], o[dh]);
                }
        }
    }
#undef A2_QK
    if (!active) return;
    {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(l_run), __float_as_uint(l_run), false, false);
        l_run = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    const int q = q0 + lr;
    if (q < N) {
        const float inv = 1.f / l_run;
        const size_t orow = ((size_t)b * N + q) * ((size_t)H * 64) + h * 64;
        a2_store_row64<SPLIT>(out_hi, out_lo, orow, o, inv, lh);
        if (lse && lh == 0) lse[((long)b * H + h) * N + q] = (m_run + __log2f(l_run)) * 0.69314718055994530942f;
    }
}


// ======================================================================================================================
// backward.  Same workgroup shape (7 waves x 32 rows of one (batch, head)), same LDS images; ONE swizzle serves both access kinds:
// chunk ^= brev3((row >> 1) & 7) is conflict-free for the 32-rows-per-instruction ds_read_b128 AND puts rows r, r + 2 of a
// ds_read_b64_tr_b16 block into different 64-byte halves.
//   dQ pass (a wave owns 32 queries, K / V of the head stream through LDS):
//       S^T = K Q^T, P^T = exp2(S^T c2 - lse), dP^T = V dO^T, dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T      (+ delta = rowsum(dO o O))
//   dK/dV pass (a wave owns 32 keys, Q / dO stream through LDS):
//       S = Q K^T, P = exp2(S c2 - lse_q), dP = dO V^T, dS = P o (dP - delta_q), dV^T += dO^T P, dK^T += Q^T dS
// Every second product takes the first one's accumulator tile as its B operand (contraction over the register index).
// ======================================================================================================================
__device__ __forceinline__ int sw2(int row) { return (0x73516240 >> (4 * ((row >> 1) & 7))) & 7; }

// LDS-DMA of one 224-row image [row][64] from a token-major tensor slice (row stride rs elements), rows clamped to N - 1
__device__ __forceinline__ void dma_image(unsigned lds_img, const bf16_t* gbase, long rs, int c0, int N, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int tok = min(c0 + row, N - 1);
        glds16(gbase + (long)tok * rs + (((lane & 7) ^ sw2(row)) * 8), lds_img + (wave * 4 + i) * 1024);
    }
}
// A operand from image rows: lane holds T[row0 + lr][16 s + 8 lh + j]
__device__ __forceinline__ bf16x8_t rows_frag(const char* img, int row0, int s, int lr, int lh) {
    const int row = row0 + lr;
    return *reinterpret_cast<const bf16x8_t*>(img + row * 128 + (((2 * s + lh) ^ sw2(row)) << 4));
}
// A operand T^T[d][row] for the accumulator-as-operand k order: d = dbase + (lane & 31), element j = row row0 + 16 s + 8 (j >> 2) + 4 lh + (j & 3)
__device__ __forceinline__ bf16x8_t tr_frag(const char* img, int row0, int s, int dbase, int lane) {
    const int g16 = lane >> 4, i16 = lane & 15;
    const int r0 = row0 + 16 * s + 4 * (g16 >> 1) + (i16 >> 2), r1 = r0 + 8;
    const int cb = (dbase + 16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r0 * 128 + ((((cb >> 4) ^ sw2(r0)) << 4) | (cb & 15))));
    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(img + r1 * 128 + ((((cb >> 4) ^ sw2(r1)) << 4) | (cb & 15))));
    const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    return __builtin_bit_cast(bf16x8_t, av);
}
// B operand of this lane's own token row: T[tok][16 s + 8 lh + j]
__device__ __forceinline__ bf16x8_t own_frag(const bf16_t* gbase, long rs, int tok, int s, int lh) {
    return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(gbase + (long)tok * rs + 16 * s + 8 * lh));
}
// accumulator registers 8 s .. 8 s + 7 -> bf16 (hi [, lo]) B operand of k-step s
template <bool SPLIT>
__device__ __forceinline__ void pack_step(const f32x16& x, int s, bf16x8_t& hi, bf16x8_t& lo) {
    float f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = x[8 * s + j];
    const uint4 u = pack8(f);
    hi = __builtin_bit_cast(bf16x8_t, u);
    if constexpr (SPLIT) {
        float hv[8], rv[8];
        unpack8(u, hv);
#pragma unroll
        for (int j = 0; j < 8; ++j) rv[j] = f[j] - hv[j];
        lo = __builtin_bit_cast(bf16x8_t, pack8(rv));
    } else {
        lo = hi;
    }
}

template <bool SPLIT, int LB>
__global__ __launch_bounds__(A2_THREADS, LB) void attn2_bwd_dq_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                  const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                                  const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo,
                                                                  const float* __restrict__ lse, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                                  float scale, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_hi = smem;
    char* v_hi = smem + A2_TILE;
    char* k_lo = smem + 2 * A2_TILE;
    char* v_lo = smem + 3 * A2_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * 64, OS = (long)H * 64;
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * 64;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * 64 : nullptr;
    const bf16_t* dob_hi = do_hi + (long)b * N * OS + h * 64;
    const bf16_t* dob_lo = SPLIT ? do_lo + (long)b * N * OS + h * 64 : nullptr;
    const int q0 = blockIdx.x * A2_QB + wave * 32;
    const bool active = q0 < N;
    const int q = q0 + lr, qr = min(q, N - 1);

    bf16x8_t qh[4], ql[4], dh[4], dl[4];
    float my_delta = 0.f;
    {
        const bf16_t* ob_hi = o_hi + (long)b * N * OS + h * 64;
        const bf16_t* ob_lo = SPLIT ? o_lo + (long)b * N * OS + h * 64 : nullptr;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qh[s] = own_frag(base_hi, RS, qr, s, lh);
            dh[s] = own_frag(dob_hi, OS, qr, s, lh);
            ql[s] = SPLIT ? own_frag(base_lo, RS, qr, s, lh) : qh[s];
            dl[s] = SPLIT ? own_frag(dob_lo, OS, qr, s, lh) : dh[s];
            const bf16x8_t oh = own_frag(ob_hi, OS, qr, s, lh);
            const bf16x8_t ol = SPLIT ? own_frag(ob_lo, OS, qr, s, lh) : oh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ov = SPLIT ? (float)oh[e] + (float)ol[e] : (float)oh[e];
                const float dv = SPLIT ? (float)dh[s][e] + (float)dl[s][e] : (float)dh[s][e];
                my_delta = fmaf(ov, dv, my_delta);
            }
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(my_delta), __float_as_uint(my_delta), false, false);
        my_delta = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0 && q < N) delta[((long)b * H + h) * N + q] = my_delta;
    }
    const float c2 = scale * 1.44269504088896340736f;
    const float my_lse = q < N ? lse[((long)b * H + h) * N + q] * 1.44269504088896340736f : INFINITY;  // +inf -> P = 0 on padded queries
    f32x16 dq[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    for (int c0 = 0; c0 < N; c0 += A2_CH) {
        if (c0 > 0) __syncthreads();
        dma_image(lds_base, base_hi + H * 64, RS, c0, N, wave, lane);
        dma_image(lds_base + A2_TILE, base_hi + 2 * H * 64, RS, c0, N, wave, lane);
        if constexpr (SPLIT) {
            dma_image(lds_base + 2 * A2_TILE, base_lo + H * 64, RS, c0, N, wave, lane);
            dma_image(lds_base + 3 * A2_TILE, base_lo + 2 * H * 64, RS, c0, N, wave, lane);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!active) continue;
        const int ntile = min(A2_CH / 32, (N - c0 + 31) >> 5);
        for (int kt = 0; kt < ntile; ++kt) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t kh = rows_frag(k_hi, kt * 32, s, lr, lh);
                const bf16x8_t kl = SPLIT ? rows_frag(k_lo, kt * 32, s, lr, lh) : kh;
                st = mma3<SPLIT>(kh, kl, qh[s], ql[s], st);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t vh = rows_frag(v_hi, kt * 32, s, lr, lh);
                const bf16x8_t vl = SPLIT ? rows_frag(v_lo, kt * 32, s, lr, lh) : vh;
                dp = mma3<SPLIT>(vh, vl, dh[s], dl[s], dp);
            }
            const int key0 = c0 + kt * 32;
            const bool tail = key0 + 32 > N;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pv = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -my_lse));
                if (tail && key0 + 8 * (r >> 2) + 4 * lh + (r & 3) >= N) pv = 0.f;
                st[r] = pv * (dp[r] - my_delta);  // dS^T (the softmax scale is applied once to the finished dQ tile)
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t sh, sl;
                pack_step<SPLIT>(st, s, sh, sl);
#pragma unroll
                for (int dhf = 0; dhf < 2; ++dhf) {
                    const bf16x8_t th = tr_frag(k_hi, kt * 32, s, 32 * dhf, lane);
                    const bf16x8_t tl = SPLIT ? tr_frag(k_lo, kt * 32, s, 32 * dhf, lane) : th;
                    dq[dhf] = mma3<SPLIT>(th, tl, sh, sl, dq[dhf]);
                }
            }
        }
    }
    const bool valid = active && q < N;
    if (valid) {
        const size_t orow = ((size_t)b * N + q) * RS + h * 64;
        a2_store_row64<SPLIT>(dqkv_hi, dqkv_lo, orow, dq, scale, lh);
    }
    if (!dbias) return;  // wave-uniform
    // Bias gradient of the fused qkv Linear (column sums of dqkv over the tokens) from what this workgroup already holds, as in the
    // single-pass kernel: Q part = sum over its queries of dQ; V part = sum_key dV = sum_q dO (the rows of P sum to one) -> the dO
    // rows of its queries; K part = 0 exactly (the rows of dS sum to zero).  This replaces an ig_colsum pass over all of dqkv per Block
    // (T = 3, B = 36: 12 x 43 us; bf16x3 at B = 216: 12 x 145 us).  Lanes = queries: a transposing butterfly over the 32 lanes of a
    // half-wave (31 shuffles per 32 values) leaves value index lr in lane lr; waves are folded through the (dead) K image.
    __syncthreads();  // every wave has left the key loop: the K image is dead
    float* scr = reinterpret_cast<float*>(smem);  // [wave][part][64]
    // value index lr -> column: dQ register (dhf, 4 g + e) = index 16 dhf + 4 g + e holds column 32 dhf + 8 g + 4 lh + e;
    // dO fragment (s, e) = index 8 s + e holds column 16 s + 8 lh + e.  (One set of 32 values at a time: registers.)
    {
        float v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = valid ? dq[i >> 4][i & 15] * scale : 0.f;
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            const bool up = (lr & o) != 0;
#pragma unroll
            for (int i = 0; i < o; ++i) {
                const float snd = up ? v[i] : v[i + o], kp = up ? v[i + o] : v[i];
                v[i] = kp + __shfl_xor(snd, o, 64);
            }
        }
        scr[(wave * 2 + 0) * 64 + 32 * (lr >> 4) + 8 * ((lr >> 2) & 3) + 4 * lh + (lr & 3)] = v[0];
    }
    {
        float v[32];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float dv = SPLIT ? (float)dh[s][e] + (float)dl[s][e] : (float)dh[s][e];
                v[s * 8 + e] = valid ? dv : 0.f;
            }
#pragma unroll
        for (int o = 16; o >= 1; o >>= 1) {
            const bool up = (lr & o) != 0;
#pragma unroll
            for (int i = 0; i < o; ++i) {
                const float snd = up ? v[i] : v[i + o], kp = up ? v[i + o] : v[i];
                v[i] = kp + __shfl_xor(snd, o, 64);
            }
        }
        scr[(wave * 2 + 1) * 64 + 16 * (lr >> 3) + 8 * lh + (lr & 7)] = v[0];
    }
    __syncthreads();
    if (tid < 128) {
        const int part = tid >> 6, col = tid & 63;  // part 0 = Q, 1 = V
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < A2_WAVES; ++w) t += scr[(w * 2 + part) * 64 + col];
        ig_red_add(dbias + (size_t)(part * 2) * H * 64 + h * 64 + col, t);
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(A2_THREADS) void attn2_bwd_dkv_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                   const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                                   float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_hi = smem;
    char* d_hi = smem + A2_TILE;
    char* q_lo = smem + 2 * A2_TILE;
    char* d_lo = smem + 3 * A2_TILE;
    float* s_lse = reinterpret_cast<float*>(smem + (SPLIT ? 4 : 2) * A2_TILE);  // [224] lse * log2(e)  (+inf on padded queries)
    float* s_del = s_lse + A2_CH;                                                  // [224] delta
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * 64, OS = (long)H * 64;
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * 64;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * 64 : nullptr;
    const bf16_t* dob_hi = do_hi + (long)b * N * OS + h * 64;
    const bf16_t* dob_lo = SPLIT ? do_lo + (long)b * N * OS + h * 64 : nullptr;
    const int k0 = blockIdx.x * A2_QB + wave * 32;
    const bool active = k0 < N;
    const int key = k0 + lr, kr = min(key, N - 1);

    bf16x8_t kh[4], kl[4], vh[4], vl[4];  // K^T / V^T B operands of this lane's key
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kh[s] = own_frag(base_hi + H * 64, RS, kr, s, lh);
        vh[s] = own_frag(base_hi + 2 * H * 64, RS, kr, s, lh);
        kl[s] = SPLIT ? own_frag(base_lo + H * 64, RS, kr, s, lh) : kh[s];
        vl[s] = SPLIT ? own_frag(base_lo + 2 * H * 64, RS, kr, s, lh) : vh[s];
    }
    const float c2 = scale * 1.44269504088896340736f;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[i][r] = 0.f, dv[i][r] = 0.f;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    for (int c0 = 0; c0 < N; c0 += A2_CH) {
        if (c0 > 0) __syncthreads();
        dma_image(lds_base, base_hi, RS, c0, N, wave, lane);
        dma_image(lds_base + A2_TILE, dob_hi, OS, c0, N, wave, lane);
        if constexpr (SPLIT) {
            dma_image(lds_base + 2 * A2_TILE, base_lo, RS, c0, N, wave, lane);
            dma_image(lds_base + 3 * A2_TILE, dob_lo, OS, c0, N, wave, lane);
        }
        if (tid < A2_CH) {
            const int qq = c0 + tid;
            s_lse[tid] = qq < N ? lse[((long)b * H + h) * N + qq] * 1.44269504088896340736f : INFINITY;
            s_del[tid] = qq < N ? delta[((long)b * H + h) * N + qq] : 0.f;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!active) continue;
        const int ntile = min(A2_CH / 32, (N - c0 + 31) >> 5);
        for (int qt = 0; qt < ntile; ++qt) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t ah = rows_frag(q_hi, qt * 32, s, lr, lh);
                const bf16x8_t al = SPLIT ? rows_frag(q_lo, qt * 32, s, lr, lh) : ah;
                st = mma3<SPLIT>(ah, al, kh[s], kl[s], st);  // S[q][key]
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const bf16x8_t ah = rows_frag(d_hi, qt * 32, s, lr, lh);
                const bf16x8_t al = SPLIT ? rows_frag(d_lo, qt * 32, s, lr, lh) : ah;
                dp = mma3<SPLIT>(ah, al, vh[s], vl[s], dp);  // dP[q][key]
            }
            f32x16 ds;
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // accumulator rows 8 g + 4 lh + (0..3) = four consecutive queries
                const float4 l4 = *reinterpret_cast<const float4*>(s_lse + qt * 32 + 8 * g + 4 * lh);
                const float4 d4 = *reinterpret_cast<const float4*>(s_del + qt * 32 + 8 * g + 4 * lh);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[4 * g + e], c2, -ll[e]));  // lse = +inf on padded queries -> 0
                    st[4 * g + e] = pv;
                    ds[4 * g + e] = pv * (dp[4 * g + e] - dd[e]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t ph, pl, sh, sl;
                pack_step<SPLIT>(st, s, ph, pl);
                pack_step<SPLIT>(ds, s, sh, sl);
#pragma unroll
                for (int dhf = 0; dhf < 2; ++dhf) {
                    const bf16x8_t th = tr_frag(d_hi, qt * 32, s, 32 * dhf, lane);  // dO^T
                    const bf16x8_t tl = SPLIT ? tr_frag(d_lo, qt * 32, s, 32 * dhf, lane) : th;
                    dv[dhf] = mma3<SPLIT>(th, tl, ph, pl, dv[dhf]);
                    const bf16x8_t uh = tr_frag(q_hi, qt * 32, s, 32 * dhf, lane);  // Q^T
                    const bf16x8_t ul = SPLIT ? tr_frag(q_lo, qt * 32, s, 32 * dhf, lane) : uh;
                    dk[dhf] = mma3<SPLIT>(uh, ul, sh, sl, dk[dhf]);
                }
            }
        }
    }
    if (!active || key >= N) return;
    const size_t orow = ((size_t)b * N + key) * RS + h * 64;
    a2_store_row64<SPLIT>(dqkv_hi, dqkv_lo, orow + (size_t)H * 64, dk, scale, lh);
    a2_store_row64<SPLIT>(dqkv_hi, dqkv_lo, orow + 2 * (size_t)H * 64, dv, 1.f, lh);
}


// ----------------------------------------------------------------------------------------------------------------------
// Fused single-pass backward for one-chunk heads (N <= 224, plain bf16): the key-owner pass above + dQ in the same sweep.
//   dS[q][key] has the keys on the LANES; dQ^T[d][q] += K^T[d][key] dS^T[key][q] contracts over keys, so the tile is transposed
//   once through a wave-private LDS slab ([key][q] bf16, 72-byte pitch: 8-byte row writes and hardware-transposed reads are both
//   conflict-free) and the 7 waves' partial dQ tiles are folded into ONE fp32 image dQ[224][64] in LDS.  LDS float atomics are far
//   too slow for that (ds_add_f32: measured ~240 cycles per wave-instruction -- 858 us per launch); instead the waves walk the
//   query tiles in ROTATED order (wave w takes tile (w + i) mod 7 in step i, a barrier per step), so that no two waves touch
//   the same tile in a step and the fold is a plain 16-byte read-add-write (pitch 68 words: conflict-free).
//   5 MFMA products per (key block, query tile) instead of 7 and Q, K, V, dO read once: 262 MB instead of 426 MB per launch at
//   B = 108.  delta = rowsum(dO o O) is computed by the workgroup itself.
constexpr int A2F_DQP = 68;                          // dQ image pitch in floats (row = query)
constexpr int A2F_SLAB = 32 * 72;                    // per-wave dS slab
constexpr int A2F_OFF_K = 2 * A2_TILE;               // K image (transposed fragments of the wave's own keys)
constexpr int A2F_OFF_DQ = 2 * A2_TILE;              // fp32 dQ image: ALIASES the K image (dead once the K^T fragments are in registers)
constexpr int A2F_OFF_SLAB = A2F_OFF_DQ + A2_CH * A2F_DQP * 4;
constexpr int A2F_OFF_LSE = A2F_OFF_SLAB + A2_WAVES * A2F_SLAB;
constexpr int A2F_SMEM = A2F_OFF_LSE + 2 * A2_CH * 4;

// B operand X[n = lane & 31][k] from a [k rows][32 n] bf16 slab of pitch 72 B, accumulator k order (see tr_frag)
__device__ __forceinline__ bf16x8_t tr_slab(const char* slab, int s, int lane) {
    const int g16 = lane >> 4, i16 = lane & 15;
    const int r0 = 16 * s + 4 * (g16 >> 1) + (i16 >> 2), r1 = r0 + 8;
    const int cb = (16 * (g16 & 1) + 4 * (i16 & 3)) * 2;
    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(slab + r0 * 72 + cb));
    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(slab + r1 * 72 + cb));
    const s16x8 av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    return __builtin_bit_cast(bf16x8_t, av);
}

__global__ __launch_bounds__(A2_THREADS) void attn2_bwd_fused_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ do_hi,
                                                                     const bf16_t* __restrict__ o_hi, const float* __restrict__ lse,
                                                                     float* __restrict__ delta, bf16_t* __restrict__ dqkv_hi, int N, int H,
                                                                     float scale, float* __restrict__ dbias) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    A2P_INIT_X
    char* q_img = smem;
    char* d_img = smem + A2_TILE;
    char* k_img = smem + A2F_OFF_K;
    float* s_dq = reinterpret_cast<float*>(smem + A2F_OFF_DQ);
    float* s_lse = reinterpret_cast<float*>(smem + A2F_OFF_LSE);
    float* s_del = s_lse + A2_CH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    char* slab = smem + A2F_OFF_SLAB + wave * A2F_SLAB;
    const int lr = lane & 31, lh = lane >> 5;
    const long RS = 3L * H * 64, OS = (long)H * 64;
    // (A persistent loop over the (batch, head) items saves the dispatch of a workgroup and the launch skew of its seven waves per head --
    // 138 -> 135.5 us at B = 108 -- but costs ~40 registers: together with the prefetched fragments below the body spills (289 us at
    // B = 216 against 276 us either way alone); as a non-inlined item function 352 us.  One workgroup per item it stays.)
    const int h = blockIdx.y, b = blockIdx.z;
    const bf16_t* base = qkv_hi + (long)b * N * RS + h * 64;
    const bf16_t* dob = do_hi + (long)b * N * OS + h * 64;
    const bf16_t* ob = o_hi + (long)b * N * OS + h * 64;
    const int k0 = wave * 32;
    const bool active = k0 < N;
    const int key = k0 + lr, kr = min(key, N - 1);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;

    dma_image(lds_base, base, RS, 0, N, wave, lane);
    dma_image(lds_base + A2_TILE, dob, OS, 0, N, wave, lane);
    dma_image(lds_base + A2F_OFF_K, base + H * 64, RS, 0, N, wave, lane);
    bf16x8_t kh[4], vh[4];  // K^T / V^T B operands of this lane's key
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        kh[s] = own_frag(base + H * 64, RS, kr, s, lh);
        vh[s] = own_frag(base + 2 * H * 64, RS, kr, s, lh);
    }
    {   // delta = rowsum(dO o O) and lse of the wave's own 32 QUERY rows (same row numbers as its keys): 8 + 8 coalesced 16-byte
        // loads per lane issued behind the DMAs, two half-row partial sums folded with one permlane swap
        float dl = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8_t oh = own_frag(ob, OS, kr, s, lh), dh = own_frag(dob, OS, kr, s, lh);
#pragma unroll
            for (int e = 0; e < 8; ++e) dl = fmaf((float)oh[e], (float)dh[e], dl);
        }
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(dl), __float_as_uint(dl), false, false);
        dl = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        if (lh == 0) {
            if (key < N) delta[((long)b * H + h) * N + key] = dl;
            s_lse[key] = key < N ? lse[((long)b * H + h) * N + key] * 1.44269504088896340736f : INFINITY;  // +inf -> P = 0 on padded queries
            s_del[key] = key < N ? dl : 0.f;
        }
    }
    const float c2 = scale * 1.44269504088896340736f;
    const float kmask = key < N ? 1.f : 0.f;  // padded keys must not reach dQ
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[i][r] = 0.f, dv[i][r] = 0.f;
    A2P_MARK(0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    A2P_MARK(1)
    __syncthreads();
    A2P_MARK(2)
    const int ntile = (N + 31) >> 5;  // = number of active waves: tile qt is first touched by wave qt in step 0
    bf16x8_t kt[2][2];                // K^T A operands of the wave's 32 keys: [k-step][d half]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dhf = 0; dhf < 2; ++dhf) kt[s][dhf] = tr_frag(k_img, k0, s, 32 * dhf, lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();  // the dQ image takes the K image's place
    A2P_MARK(3)
    for (int step = 0; step < A2_WAVES; ++step) {
        const int qt = (wave + step) % A2_WAVES;
        A2P_MARK(4 + step * 6)
        if (active && qt < ntile) {
            f32x16 st, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.f, dp[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) st = mfma32(rows_frag(q_img, qt * 32, s, lr, lh), kh[s], st);  // S[q][key]
#pragma unroll
            for (int s = 0; s < 4; ++s) dp = mfma32(rows_frag(d_img, qt * 32, s, lr, lh), vh[s], dp);  // dP[q][key]
            // the transposed Q / dO fragments of the dV / dK products do not depend on the softmax: their LDS reads are issued HERE, so
            // that their latency runs under the exp / dS arithmetic instead of in front of every MFMA pair (phase timing, tools/
            // attn_phase_prof.py: "pack + dV, dK" was 1325 cycles per step and wave for 256 cycles of MFMAs)
            bf16x8_t tdo[2][2], tq[2][2];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int dhf = 0; dhf < 2; ++dhf) {
                    tdo[s][dhf] = tr_frag(d_img, qt * 32, s, 32 * dhf, lane);
                    tq[s][dhf] = tr_frag(q_img, qt * 32, s, 32 * dhf, lane);
                }
            A2P_MARKV(5 + step * 6, st[0] + dp[0])
            f32x16 ds;
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // accumulator rows 8 g + 4 lh + (0..3) = four consecutive queries
                const float4 l4 = *reinterpret_cast<const float4*>(s_lse + qt * 32 + 8 * g + 4 * lh);
                const float4 d4 = *reinterpret_cast<const float4*>(s_del + qt * 32 + 8 * g + 4 * lh);
                const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float pv = __builtin_amdgcn_exp2f(fmaf(st[4 * g + e], c2, -ll[e])) * kmask;
                    st[4 * g + e] = pv;
                    ds[4 * g + e] = pv * (dp[4 * g + e] - dd[e]);
                }
            }
            A2P_MARKV(6 + step * 6, ds[0] + ds[15])
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t ph, pl, sh, sl;
                pack_step<false>(st, s, ph, pl);
                pack_step<false>(ds, s, sh, sl);
                // dS rows of this lane's key: queries 16 s + 4 lh + (0..3) and 16 s + 8 + 4 lh + (0..3)
                const uint4 su = __builtin_bit_cast(uint4, sh);
                *reinterpret_cast<uint2*>(slab + lr * 72 + (16 * s + 4 * lh) * 2) = make_uint2(su.x, su.y);
                *reinterpret_cast<uint2*>(slab + lr * 72 + (16 * s + 8 + 4 * lh) * 2) = make_uint2(su.z, su.w);
#pragma unroll
                for (int dhf = 0; dhf < 2; ++dhf) {
                    dv[dhf] = mfma32(tdo[s][dhf], ph, dv[dhf]);  // dV^T += dO^T P
                    dk[dhf] = mfma32(tq[s][dhf], sh, dk[dhf]);   // dK^T += Q^T dS
                }
            }
            A2P_MARKV(7 + step * 6, dv[0][0] + dk[0][0] + dv[1][15] + dk[1][15])
            // dQ^T[d][q tile] partial of this key block, folded into the workgroup's fp32 image (this wave owns tile qt in this step)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // the dQ partials of the earlier steps for this query tile (final since the last barrier) become the accumulator INIT of this
            // step's dQ product: their LDS reads run under the dV / dK MFMAs, and the read-add-write after the product is a plain write
            f32x16 dq0, dq1;  // (loaded behind the dV / dK MFMAs: their issue covers the LDS latency; earlier, the 32 registers spill)
            {
                const float* row = s_dq + (qt * 32 + lr) * A2F_DQP + 4 * lh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 o0 = make_float4(0.f, 0.f, 0.f, 0.f), o1 = o0;
                    if (step > 0) o0 = *reinterpret_cast<const float4*>(row + 8 * g), o1 = *reinterpret_cast<const float4*>(row + 32 + 8 * g);
                    dq0[4 * g] = o0.x, dq0[4 * g + 1] = o0.y, dq0[4 * g + 2] = o0.z, dq0[4 * g + 3] = o0.w;
                    dq1[4 * g] = o1.x, dq1[4 * g + 1] = o1.y, dq1[4 * g + 2] = o1.z, dq1[4 * g + 3] = o1.w;
                }
            }
            const bf16x8_t b0 = tr_slab(slab, 0, lane), b1 = tr_slab(slab, 1, lane);
            dq0 = mfma32(kt[0][0], b0, dq0);
            dq1 = mfma32(kt[0][1], b0, dq1);
            dq0 = mfma32(kt[1][0], b1, dq0);
            dq1 = mfma32(kt[1][1], b1, dq1);
            {
                float* row = s_dq + (qt * 32 + lr) * A2F_DQP + 4 * lh;  // lane = query; registers 4 g .. 4 g + 3 = columns 8 g + 4 lh + (0..3)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    *reinterpret_cast<float4*>(row + 8 * g) = make_float4(dq0[4 * g], dq0[4 * g + 1], dq0[4 * g + 2], dq0[4 * g + 3]);
                    *reinterpret_cast<float4*>(row + 32 + 8 * g) = make_float4(dq1[4 * g], dq1[4 * g + 1], dq1[4 * g + 2], dq1[4 * g + 3]);
                }
            }
            A2P_MARK(8 + step * 6)
        }
        __syncthreads();
        A2P_MARK(9 + step * 6)
    }
    // dQ write-out: item = (query, 8-column chunk); 8 lanes cover one 128-byte row segment of dqkv's Q slot.  A thread's chunk is
    // always c = tid & 7 (the stride is a multiple of 8), so its partial column sums stay in eight registers: the qkv BIAS gradient.
    float cbq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, cbv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int it = tid; it < N * 8; it += A2_THREADS) {
        const int q = it >> 3, c = it & 7;
        const float4 v0 = *reinterpret_cast<const float4*>(s_dq + q * A2F_DQP + 8 * c), v1 = *reinterpret_cast<const float4*>(s_dq + q * A2F_DQP + 8 * c + 4);
        const float f[8] = {v0.x * scale, v0.y * scale, v0.z * scale, v0.w * scale, v1.x * scale, v1.y * scale, v1.z * scale, v1.w * scale};
        *reinterpret_cast<uint4*>(dqkv_hi + ((size_t)b * N + q) * RS + h * 64 + 8 * c) = pack8(f);
#pragma unroll
        for (int e = 0; e < 8; ++e) cbq[e] += f[e];
    }
    A2P_MARK(46)
    if (dbias) {  // wave-uniform
        // Bias gradient of the fused qkv Linear = column sums of dqkv over the tokens.  Its three parts for this (batch, head):
        //   Q: sum_q dQ[q][d]                      -- the partial sums above;
        //   K: sum_key dK[key][d] = sum_q Q[q][d] * (sum_key dS[q][key]) = 0 EXACTLY: the rows of dS sum to zero (softmax Jacobian;
        //      a key bias shifts every logit of a row by the same amount) -- nothing to add;
        //   V: sum_key dV[key][d] = sum_q dO[q][d] * (sum_key P[q][key]) = sum_q dO[q][d]: column sums of the dO image in LDS.
        for (int it = tid; it < N * 8; it += A2_THREADS) {
            const int q = it >> 3, c = it & 7;
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(d_img + q * 128 + ((c ^ sw2(q)) << 4)), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) cbv[e] += f[e];
        }
        A2P_MARKV(40, cbv[0] + cbv[7])
        // Every thread's 8 + 8 partial sums go to LDS as they are (the Q image is dead: [thread][16] floats = exactly its 28 KB) and 128
        // threads add the 56 partials of their column (7 waves x 8 lane groups own the same 8-column chunk) in a fixed order.  (Folding over
        // the lanes with 48 __shfl_xor first cost 2.7 k of the workgroup's 41 k cycles: each is a ds_bpermute round trip.)
        float* scr = reinterpret_cast<float*>(q_img);  // [tid][16]: 0..7 = dQ sums, 8..15 = dO sums of chunk tid & 7
        *reinterpret_cast<float4*>(scr + tid * 16) = make_float4(cbq[0], cbq[1], cbq[2], cbq[3]);
        *reinterpret_cast<float4*>(scr + tid * 16 + 4) = make_float4(cbq[4], cbq[5], cbq[6], cbq[7]);
        *reinterpret_cast<float4*>(scr + tid * 16 + 8) = make_float4(cbv[0], cbv[1], cbv[2], cbv[3]);
        *reinterpret_cast<float4*>(scr + tid * 16 + 12) = make_float4(cbv[4], cbv[5], cbv[6], cbv[7]);
        A2P_MARK(41)
        __syncthreads();
        A2P_MARK(42)
        if (tid < 128) {
            const int part = tid >> 6, col = tid & 63;  // part 0 = Q, 1 = V
            float t = 0.f;
            for (int g = 0; g < A2_THREADS / 8; ++g) t += scr[(g * 8 + (col >> 3)) * 16 + part * 8 + (col & 7)];
            ig_red_add(dbias + (size_t)(part * 2) * H * 64 + h * 64 + col, t);
        }
    }
    A2P_MARK(47)
    if (active && key < N) {
        const size_t orow = ((size_t)b * N + key) * RS + h * 64;
        a2_store_row64<false>(dqkv_hi, nullptr, orow + (size_t)H * 64, dk, scale, lh);
        a2_store_row64<false>(dqkv_hi, nullptr, orow + 2 * (size_t)H * 64, dv, 1.f, lh);
    }
}

}  // namespace
IG_DET_TU(attention2)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

int ig_attention2_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H, void* stream) {
    IG_REQUIRE(N >= 1 && H <= 65535 && B <= 65535, "ig_attention_fwd: B and H must be <= 65535 (grid dimensions), got B = %d, H = %d", B, H);
    IG_REQUIRE((((uintptr_t)qkv_hi | (uintptr_t)qkv_lo) & 15) == 0, "ig_attention_fwd: qkv must be 16-byte aligned%s", "");
    const bool split = qkv_lo != nullptr;
    const dim3 grid((N + A2_QB - 1) / A2_QB, H, B);
    const int lds = (split ? 4 : 2) * A2_TILE;
    const float scale = 0.125f;  // 64^-0.5
    hipStream_t st = (hipStream_t)stream;
    if (split) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn2_fwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr = true;
        }
        ig_note_kernel("attn2_fwd_kernel<true>");
        hipLaunchKernelGGL(attn2_fwd_kernel<true>, grid, dim3(A2_THREADS), lds, st, (const bf16_t*)qkv_hi, (const bf16_t*)qkv_lo,
                           (bf16_t*)out_hi, (bf16_t*)out_lo, lse, N, H, scale);
    } else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn2_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr = true;
        }
        ig_note_kernel("attn2_fwd_kernel<false>");
        hipLaunchKernelGGL(attn2_fwd_kernel<false>, grid, dim3(A2_THREADS), lds, st, (const bf16_t*)qkv_hi, (const bf16_t*)qkv_lo,
                           (bf16_t*)out_hi, (bf16_t*)out_lo, lse, N, H, scale);
    }
    return ig_check_launch("ig_attention_fwd(attn2)");
}

// dbias (optional): the qkv bias gradient, dbias[3][H][64] += column sums of dqkv over the B * N tokens -- fused into the single-pass
// kernel and into the dQ kernel of the two-pass form (the K third is identically zero; the K third is identically zero)
int ig_attention2_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                      const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, float* dbias, int B, int N, int H,
                      void* stream) {
    IG_REQUIRE(N >= 1 && H <= 65535 && B <= 65535, "ig_attention_bwd: B and H must be <= 65535 (grid dimensions), got B = %d, H = %d", B, H);
    IG_REQUIRE((((uintptr_t)qkv_hi | (uintptr_t)qkv_lo | (uintptr_t)dout_hi | (uintptr_t)dout_lo | (uintptr_t)out_hi | (uintptr_t)out_lo) & 15) == 0,
               "ig_attention_bwd: the tensors must be 16-byte aligned%s", "");
    const bool split = qkv_lo != nullptr;
    const dim3 grid((N + A2_QB - 1) / A2_QB, H, B);
    const int lds_q = (split ? 4 : 2) * A2_TILE, lds_kv = lds_q + 2 * A2_CH * (int)sizeof(float);
    const float scale = 0.125f;
    hipStream_t st = (hipStream_t)stream;
    if (!split && N <= A2_CH) {  // one chunk of keys, plain bf16: the single-pass kernel
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)attn2_bwd_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A2F_SMEM);
            attr = true;
        }
        ig_note_kernel("attn2_bwd_fused_kernel");
        hipLaunchKernelGGL(attn2_bwd_fused_kernel, dim3(1, H, B), dim3(A2_THREADS), A2F_SMEM, st, (const bf16_t*)qkv_hi, (const bf16_t*)dout_hi,
                           (const bf16_t*)out_hi, lse, delta, (bf16_t*)dqkv_hi, N, H, scale, dbias);
        return ig_check_launch("ig_attention_bwd(attn2 fused)");
    }
    // plain bf16: two dQ workgroups per CU at 128 registers (56 B of scratch; measured 75 against 85 us for one at 143); the qkv bias
    // gradient comes out of the dQ kernel (a column-sum pass over dqkv was 41.9 us at T = 3 / B = 36)
    constexpr int dqlb = 4;
    float* dbias_k = dbias;
#define IG_A2_BWD(SPLIT_)                                                                                                          \
    {                                                                                                                              \
        static bool attr = false;                                                                                                  \
        if (!attr) {                                                                                                               \
            (void)hipFuncSetAttribute((const void*)attn2_bwd_dq_kernel<SPLIT_, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q); \
            (void)hipFuncSetAttribute((const void*)attn2_bwd_dq_kernel<SPLIT_, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q); \
            (void)hipFuncSetAttribute((const void*)attn2_bwd_dkv_kernel<SPLIT_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv); \
            attr = true;                                                                                                           \
        }                                                                                                                          \
        if (dqlb == 4 && !SPLIT_)                                                                                                  \
            hipLaunchKernelGGL((attn2_bwd_dq_kernel<SPLIT_, 4>), grid, dim3(A2_THREADS), lds_q, st, (const bf16_t*)qkv_hi,          \
                               (const bf16_t*)qkv_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, (const bf16_t*)out_hi,       \
                               (const bf16_t*)out_lo, lse, delta, (bf16_t*)dqkv_hi, (bf16_t*)dqkv_lo, N, H, scale, dbias_k);       \
        else                                                                                                                       \
            hipLaunchKernelGGL((attn2_bwd_dq_kernel<SPLIT_, 2>), grid, dim3(A2_THREADS), lds_q, st, (const bf16_t*)qkv_hi,          \
                               (const bf16_t*)qkv_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, (const bf16_t*)out_hi,       \
                               (const bf16_t*)out_lo, lse, delta, (bf16_t*)dqkv_hi, (bf16_t*)dqkv_lo, N, H, scale, dbias_k);       \
        ig_note_kernel("attn2_bwd_dq_kernel<%s>+attn2_bwd_dkv_kernel<%s>", SPLIT_ ? "true" : "false", SPLIT_ ? "true" : "false");  \
        hipLaunchKernelGGL(attn2_bwd_dkv_kernel<SPLIT_>, grid, dim3(A2_THREADS), lds_kv, st, (const bf16_t*)qkv_hi,                   \
                           (const bf16_t*)qkv_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, lse, delta, (bf16_t*)dqkv_hi,     \
                           (bf16_t*)dqkv_lo, N, H, scale);                                                                         \
    }
    if (split) IG_A2_BWD(true) else IG_A2_BWD(false)
#undef IG_A2_BWD
    return ig_check_launch("ig_attention_bwd(attn2)");
}
