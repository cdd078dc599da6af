// Fused multi-head self-attention for the Prithvi ViT blocks on gfx950 (head_dim = 64, no mask, no dropout).
// Replaces timm Attention's F.scaled_dot_product_attention called from pritvhi.py:446-456.
//
// Layout: qkv [B][N][3][H][64] bf16 (the timm reshape(B,N,3,H,hd) of the fused qkv Linear), out [B][N][H*64].
// Structure (all three kernels): one workgroup = NW waves (<=16) of one (batch, head); every wave OWNS a 16-row
// tile (queries in fwd / dQ, keys in dK/dV) whose operands live in registers, while the other side is STREAMED
// through LDS in 32-row tiles shared by all waves.  All products are MFMA 16x16x32 bf16 with the streamed index
// on the accumulator ROW, so the probability tile is directly the B operand of the next product (no LDS round
// trip); the transposed operands (V^T, K^T, dO^T, Q^T) come from ds_read_b64_tr_b16 on the row-major LDS tile.
// N = 197 / 589 tokens are handled by zero-filled tails and -inf / +inf masks.
// SPLIT=true is the bf16x3 precision mode (hi*hi + hi*lo + lo*hi).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HD = 64;
constexpr int KT = 32;                 // streamed rows per tile
constexpr int TILE = KT * HD * 2;      // 4 KiB per bf16 tile
// tiles streamed per barrier pair (NTL): one cooperative load + 2 barriers cover 32 * NTL rows

__device__ __forceinline__ int lds_kc(int r, int c) { return r * 128 + ((c ^ (r & 7)) << 4); }

typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

// row fragment: lane holds T[row0 + (l&15)][32*s + 8*(l>>4) + j]
__device__ __forceinline__ bf16x8_t frag_rows(const char* tile, int row0, int s, int lane) {
    return *reinterpret_cast<const bf16x8_t*>(tile + lds_kc(row0 + (lane & 15), s * 4 + (lane >> 4)));
}
// transposed fragment for the "accumulator as operand" k order: lane (g=l>>4,i=l&15) holds
// T[4g + j][col0 + i] (j<4) and T[16 + 4g + (j-4)][col0 + i] (j>=4)
__device__ __forceinline__ bf16x8_t frag_tr(const char* tile, int col0, int lane) {
    int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    int col = col0 + 4 * p;
    int chunk = col >> 3, sub = (col & 7) * 2;
    s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + lds_kc(4 * g + q, chunk) + sub));
    s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(tile + lds_kc(16 + 4 * g + q, chunk) + sub));
    s16x8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    return __builtin_bit_cast(bf16x8_t, r);
}

template <bool SPLIT>
__device__ __forceinline__ f32x4 mma(bf16x8_t ah, bf16x8_t al, bf16x8_t bh, bf16x8_t bl, f32x4 acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    if constexpr (SPLIT) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
    }
    return acc;
}

// pack two accumulator tiles (rows 4g+reg of sub-tile 0 and 1) into the next product's B operand (hi [+ lo])
template <bool SPLIT>
__device__ __forceinline__ void pack_acc(const f32x4& a0, const f32x4& a1, bf16x8_t& hi, bf16x8_t& lo) {
    float f[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    uint4 u = pack8(f);
    hi = __builtin_bit_cast(bf16x8_t, u);
    if constexpr (SPLIT) {
        float h[8], r[8];
        unpack8(u, h);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = f[i] - h[i];
        uint4 v = pack8(r);
        lo = __builtin_bit_cast(bf16x8_t, v);
    }
}

// Cooperative, register-staged load of one GROUP of streamed tiles: NT tensors x NTL tiles of 32 x 64 bf16, laid out in
// LDS as tile index (tensor * NTL + j).  fetch() issues every global load back to back (unconditional loads from a
// clamped address + select: a branch around a load would serialise them on vmcnt(0)); commit() writes the staged
// registers to LDS.  With nthr * NR >= UPG a whole group is one pass, so the NEXT group's fetch can stay in flight
// while the current one is consumed; smaller workgroups fall back to synchronous passes.
template <int NT, int NTL, int NR_>
struct TileGroup {
    static constexpr int UPG = NT * NTL * KT * 8;  // 16-byte units per group
    static constexpr int NR = NR_;                 // staged units per thread and pass (one pass needs nthr * NR >= UPG)
    uint4 v[NR];

    __device__ __forceinline__ void fetch(int u0, const bf16_t* b0, const bf16_t* b1, const bf16_t* b2, const bf16_t* b3, long rs0,
                                          long rs1, int tile0, int nrows, int tid, int nthr) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (u0 + i * nthr >= UPG) break;  // uniform
            const int u = u0 + i * nthr + tid;
            const int t = u >> 8, tensor = t / NTL, j = t - tensor * NTL, r = (u >> 3) & 31, c = u & 7;
            const int row = (tile0 + j) * KT + r;
            const bool ok = u < UPG && row < nrows;
            const bf16_t* base = tensor == 0 ? b0 : tensor == 1 ? b1 : tensor == 2 ? b2 : b3;
            const long rs = (tensor & 1) ? rs1 : rs0;
            const uint4 x = *reinterpret_cast<const uint4*>(base + (ok ? (long)row * rs + c * 8 : 0L));
            v[i] = ok ? x : make_uint4(0, 0, 0, 0);
        }
    }
    __device__ __forceinline__ void commit(int u0, char* smem, int tid, int nthr) const {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (u0 + i * nthr >= UPG) break;
            const int u = u0 + i * nthr + tid;
            if (u < UPG) *reinterpret_cast<uint4*>(smem + (u >> 8) * TILE + lds_kc((u >> 3) & 31, u & 7)) = v[i];
        }
    }
    // synchronous variant for small workgroups
    __device__ __forceinline__ void load_sync(char* smem, const bf16_t* b0, const bf16_t* b1, const bf16_t* b2, const bf16_t* b3,
                                              long rs0, long rs1, int tile0, int nrows, int tid, int nthr) {
        for (int u0 = 0; u0 < UPG; u0 += NR * nthr) {
            fetch(u0, b0, b1, b2, b3, rs0, rs1, tile0, nrows, tid, nthr);
            commit(u0, smem, tid, nthr);
        }
    }
};

// register fragment of the wave-owned 16-row tile straight from global memory (zero beyond nrows)
__device__ __forceinline__ bf16x8_t load_own(const bf16_t* base, long row_stride, int row0, int nrows, int s, int lane) {
    int r = row0 + (lane & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r < nrows) v = *reinterpret_cast<const uint4*>(base + (long)r * row_stride + s * 32 + 8 * (lane >> 4));
    return __builtin_bit_cast(bf16x8_t, v);
}

// ------------------------------------------------------------------------------------------------------
// forward: O = softmax(scale * Q K^T) V ; LSE saved for backward
// ------------------------------------------------------------------------------------------------------
template <bool SPLIT, int NTLP, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_fwd_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                        bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                        float* __restrict__ lse, int N, int H, float scale) {
    constexpr int NTL = SPLIT ? 2 : NTLP;
    // staging registers: one pass must cover the group with >= 13 waves (MAXT = 1024) or >= 7 waves (MAXT = 512)
    constexpr int NRS = ((SPLIT ? 4 : 2) * NTL * KT * 8 + (MAXT == 1024 ? 832 : 448) - 1) / (MAXT == 1024 ? 832 : 448);
    __shared__ __attribute__((aligned(16))) char smem[(SPLIT ? 4 : 2) * NTL * TILE];
    char* k_hi0 = smem;
    char* v_hi0 = smem + NTL * TILE;
    char* k_lo0 = smem + 2 * NTL * TILE;  // only SPLIT
    char* v_lo0 = smem + 3 * NTL * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * HD;
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : nullptr;
    const int q0 = (blockIdx.x * nw + wave) * 16;
    const int g = lane >> 4;

    bf16x8_t qh[2], ql[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qh[s] = load_own(base_hi, RS, q0, N, s, lane);
        ql[s] = SPLIT ? load_own(base_lo, RS, q0, N, s, lane) : qh[s];
    }
    f32x4 o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;  // running max in the base-2 domain (m_run = max(s) * c2)
    const float c2 = scale * 1.44269504088896340736f;

    const int ntiles = (N + KT - 1) / KT;
    TileGroup<SPLIT ? 4 : 2, NTL, NRS> tg;
    const int nthr = blockDim.x;
    // whole group in one pass -> prefetch the next group during compute (not in split mode: no registers to spare)
    const bool pre = !SPLIT && nthr * tg.NR >= tg.UPG;
    const bf16_t *gk_hi = base_hi + H * HD, *gv_hi = base_hi + 2 * H * HD;
    const bf16_t *gk_lo = SPLIT ? base_lo + H * HD : nullptr, *gv_lo = SPLIT ? base_lo + 2 * H * HD : nullptr;
    if (pre) tg.fetch(0, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, 0, N, tid, nthr);
    for (int kt0 = 0; kt0 < ntiles; kt0 += NTL) {
        __syncthreads();  // previous tiles fully consumed
        if (pre) tg.commit(0, smem, tid, nthr);
        else tg.load_sync(smem, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, kt0, N, tid, nthr);
        __syncthreads();
        if (pre && kt0 + NTL < ntiles) tg.fetch(0, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, kt0 + NTL, N, tid, nthr);
        for (int j = 0; j < NTL && kt0 + j < ntiles; ++j) {
        const int kt = kt0 + j;
        const char *k_hi = k_hi0 + j * TILE, *v_hi = v_hi0 + j * TILE, *k_lo = k_lo0 + j * TILE, *v_lo = v_lo0 + j * TILE;
        // S^T[key][q] for two 16-key sub-tiles
        f32x4 st[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            st[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t kh = frag_rows(k_hi, sub * 16, s, lane);
                bf16x8_t kl = SPLIT ? frag_rows(k_lo, sub * 16, s, lane) : kh;
                st[sub] = mma<SPLIT>(kh, kl, qh[s], ql[s], st[sub]);
            }
        }
        // online softmax over this lane's query column (keys live in regs and in lanes l^16, l^32, l^48), in the
        // base-2 domain: p = exp2(s*c - m) with c = scale*log2(e) folded into one fma per element (v_exp_f32 IS exp2);
        // the key < N mask is only evaluated on the tail tile
        if (kt * KT + KT > N) {
            asm volatile("" ::: "memory");  // keep this a (uniform) branch: if-converted, the 7 compares + selects ran on every tile
#pragma unroll
            for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * KT + sub * 16 + 4 * g + r >= N) st[sub][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3])),
                         fmaxf(fmaxf(st[1][0], st[1][1]), fmaxf(st[1][2], st[1][3])));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx * c2);  // c2 > 0: max commutes with the scaling
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // m_run=-inf on the first tile -> 0
        float psum = 0.f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[sub][r], c2, -m_new));
                st[sub][r] = p;
                psum += p;
            }
        l_run = l_run * alpha + psum;  // per-lane partial; alpha is uniform over the 4 lanes of a query
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] *= alpha;
        bf16x8_t ph, pl;
        pack_acc<SPLIT>(st[0], st[1], ph, pl);
        if constexpr (!SPLIT) pl = ph;
        // O^T[d][q] += V^T[d][key] P^T[key][q]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x8_t vh = frag_tr(v_hi, dt * 16, lane);
            bf16x8_t vl = SPLIT ? frag_tr(v_lo, dt * 16, lane) : vh;
            o[dt] = mma<SPLIT>(vh, vl, ph, pl, o[dt]);
        }
        }  // j
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const int q = q0 + (lane & 15);
    if (q < N) {
        float inv = 1.f / l_run;
        long orow = ((long)b * N + q) * ((long)H * HD) + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float f[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
            store4_split(out_hi, out_lo, (size_t)orow + dt * 16 + 4 * g, f);
        }
        if (lse && g == 0) lse[((long)b * H + h) * N + q] = (m_run + __log2f(l_run)) * 0.69314718055994530942f;
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, query-owner pass: dQ = scale * dS K   (streams K,V tiles; recomputes P^T from LSE)
// ------------------------------------------------------------------------------------------------------
template <bool SPLIT, int NTLP, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_bwd_dq_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                           const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                           const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                           float scale) {
    constexpr int NTL = SPLIT ? 2 : NTLP;
    // staging registers: one pass must cover the group with >= 13 waves (MAXT = 1024) or >= 7 waves (MAXT = 512)
    constexpr int NRS = ((SPLIT ? 4 : 2) * NTL * KT * 8 + (MAXT == 1024 ? 832 : 448) - 1) / (MAXT == 1024 ? 832 : 448);
    __shared__ __attribute__((aligned(16))) char smem[(SPLIT ? 4 : 2) * NTL * TILE];
    char* k_hi0 = smem;
    char* v_hi0 = smem + NTL * TILE;
    char* k_lo0 = smem + 2 * NTL * TILE;
    char* v_lo0 = smem + 3 * NTL * TILE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : nullptr;
    const bf16_t* dob_hi = do_hi + (long)b * N * OS + h * HD;
    const bf16_t* dob_lo = SPLIT ? do_lo + (long)b * N * OS + h * HD : nullptr;
    const int q0 = (blockIdx.x * nw + wave) * 16;
    const int g = lane >> 4;
    const int q = q0 + (lane & 15);

    bf16x8_t qh[2], ql[2], dh[2], dl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        qh[s] = load_own(base_hi, RS, q0, N, s, lane);
        dh[s] = load_own(dob_hi, OS, q0, N, s, lane);
        ql[s] = SPLIT ? load_own(base_lo, RS, q0, N, s, lane) : qh[s];
        dl[s] = SPLIT ? load_own(dob_lo, OS, q0, N, s, lane) : dh[s];
    }
    const float c2 = scale * 1.44269504088896340736f;  // P = exp2(s*c2 - lse*log2e): one fma + v_exp_f32 per element
    const float my_lse = q < N ? lse[((long)b * H + h) * N + q] * 1.44269504088896340736f : INFINITY;  // +inf -> P = 0 on padded queries
    // delta[q] = sum_d dO[q][d] * O[q][d], computed here from the dO fragments this wave owns anyway (its four k-group lanes
    // hold the 64 values of a row) and stored for the key-owner pass that runs next -- a separate delta kernel was 17 us per
    // block of launch + one more pass over O and dO
    float my_delta = 0.f;
    {
        const bf16_t* ob_hi = o_hi + (long)b * N * OS + h * HD;
        const bf16_t* ob_lo = SPLIT ? o_lo + (long)b * N * OS + h * HD : nullptr;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8_t oh = load_own(ob_hi, OS, q0, N, s, lane);
            const bf16x8_t ol = SPLIT ? load_own(ob_lo, OS, q0, N, s, lane) : oh;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ov = SPLIT ? (float)oh[e] + (float)ol[e] : (float)oh[e];
                const float dv = SPLIT ? (float)dh[s][e] + (float)dl[s][e] : (float)dh[s][e];
                my_delta = fmaf(ov, dv, my_delta);
            }
        }
        my_delta += __shfl_xor(my_delta, 16, 64);
        my_delta += __shfl_xor(my_delta, 32, 64);
        if (g == 0 && q < N) delta[((long)b * H + h) * N + q] = my_delta;
    }
    f32x4 dq[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dq[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (N + KT - 1) / KT;
    TileGroup<SPLIT ? 4 : 2, NTL, NRS> tg;
    const int nthr = blockDim.x;
    // whole group in one pass -> prefetch the next group during compute (not in split mode: no registers to spare)
    const bool pre = !SPLIT && nthr * tg.NR >= tg.UPG;
    const bf16_t *gk_hi = base_hi + H * HD, *gv_hi = base_hi + 2 * H * HD;
    const bf16_t *gk_lo = SPLIT ? base_lo + H * HD : nullptr, *gv_lo = SPLIT ? base_lo + 2 * H * HD : nullptr;
    if (pre) tg.fetch(0, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, 0, N, tid, nthr);
    for (int kt0 = 0; kt0 < ntiles; kt0 += NTL) {
        __syncthreads();  // previous tiles fully consumed
        if (pre) tg.commit(0, smem, tid, nthr);
        else tg.load_sync(smem, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, kt0, N, tid, nthr);
        __syncthreads();
        if (pre && kt0 + NTL < ntiles) tg.fetch(0, gk_hi, gv_hi, gk_lo, gv_lo, RS, RS, kt0 + NTL, N, tid, nthr);
        for (int j = 0; j < NTL && kt0 + j < ntiles; ++j) {
        const int kt = kt0 + j;
        const char *k_hi = k_hi0 + j * TILE, *v_hi = v_hi0 + j * TILE, *k_lo = k_lo0 + j * TILE, *v_lo = v_lo0 + j * TILE;
        const bool tail = kt * KT + KT > N;  // only the last tile has keys beyond N
        f32x4 ds[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t kh = frag_rows(k_hi, sub * 16, s, lane);
                bf16x8_t kl = SPLIT ? frag_rows(k_lo, sub * 16, s, lane) : kh;
                st = mma<SPLIT>(kh, kl, qh[s], ql[s], st);
                bf16x8_t vh = frag_rows(v_hi, sub * 16, s, lane);
                bf16x8_t vl = SPLIT ? frag_rows(v_lo, sub * 16, s, lane) : vh;
                dp = mma<SPLIT>(vh, vl, dh[s], dl[s], dp);
            }
            if (tail) {  // keys beyond N: S = -inf -> P = 0 (a real branch: the selects would otherwise run on every tile)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * KT + sub * 16 + 4 * g + r >= N) st[r] = -INFINITY;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -my_lse));
                ds[sub][r] = p * (dp[r] - my_delta);  // the softmax scale is applied once to the finished dQ tile
            }
        }
        bf16x8_t sh, sl;
        pack_acc<SPLIT>(ds[0], ds[1], sh, sl);
        if constexpr (!SPLIT) sl = sh;
        // dQ^T[d][q] += K^T[d][key] dS^T[key][q]
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x8_t kh = frag_tr(k_hi, dt * 16, lane);
            bf16x8_t kl = SPLIT ? frag_tr(k_lo, dt * 16, lane) : kh;
            dq[dt] = mma<SPLIT>(kh, kl, sh, sl, dq[dt]);
        }
        }  // j
    }
    if (q < N) {
        long orow = ((long)b * N + q) * RS + h * HD;  // q slot of dqkv
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float f[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
            store4_split(dqkv_hi, dqkv_lo, (size_t)orow + dt * 16 + 4 * g, f);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// backward, key-owner pass: dV = P^T dO ; dK = scale * dS^T Q   (streams Q,dO tiles)
// ------------------------------------------------------------------------------------------------------
template <bool SPLIT, int NTLP, int MAXT>
__global__ __launch_bounds__(MAXT) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                            const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                            const float* __restrict__ lse, const float* __restrict__ delta,
                                                            bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                            float scale) {
    constexpr int NTL = SPLIT ? 2 : NTLP;
    // staging registers: one pass must cover the group with >= 13 waves (MAXT = 1024) or >= 7 waves (MAXT = 512)
    constexpr int NRS = ((SPLIT ? 4 : 2) * NTL * KT * 8 + (MAXT == 1024 ? 832 : 448) - 1) / (MAXT == 1024 ? 832 : 448);
    __shared__ __attribute__((aligned(16))) char smem[(SPLIT ? 4 : 2) * NTL * TILE + 2 * NTL * KT * 4];
    char* q_hi0 = smem;
    char* d_hi0 = smem + NTL * TILE;
    char* q_lo0 = smem + 2 * NTL * TILE;
    char* d_lo0 = smem + 3 * NTL * TILE;
    float* s_lse0 = reinterpret_cast<float*>(smem + (SPLIT ? 4 : 2) * NTL * TILE);
    float* s_del0 = s_lse0 + NTL * KT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int h = blockIdx.y, b = blockIdx.z;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* base_hi = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* base_lo = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : nullptr;
    const bf16_t* dob_hi = do_hi + (long)b * N * OS + h * HD;
    const bf16_t* dob_lo = SPLIT ? do_lo + (long)b * N * OS + h * HD : nullptr;
    const int k0 = (blockIdx.x * nw + wave) * 16;
    const int g = lane >> 4;
    const float c2 = scale * 1.44269504088896340736f;

    bf16x8_t kh[2], kl[2], vh[2], vl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        kh[s] = load_own(base_hi + H * HD, RS, k0, N, s, lane);
        vh[s] = load_own(base_hi + 2 * H * HD, RS, k0, N, s, lane);
        kl[s] = SPLIT ? load_own(base_lo + H * HD, RS, k0, N, s, lane) : kh[s];
        vl[s] = SPLIT ? load_own(base_lo + 2 * H * HD, RS, k0, N, s, lane) : vh[s];
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dk[i] = dv[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ntiles = (N + KT - 1) / KT;
    TileGroup<SPLIT ? 4 : 2, NTL, NRS> tg;
    const int nthr = blockDim.x;
    const bool pre = !SPLIT && nthr * tg.NR >= tg.UPG;
    if (pre) tg.fetch(0, base_hi, dob_hi, base_lo, dob_lo, RS, OS, 0, N, tid, nthr);
    for (int qt0 = 0; qt0 < ntiles; qt0 += NTL) {
        __syncthreads();
        if (pre) tg.commit(0, smem, tid, nthr);
        else tg.load_sync(smem, base_hi, dob_hi, base_lo, dob_lo, RS, OS, qt0, N, tid, nthr);
        if (tid < NTL * KT) {
            int qq = qt0 * KT + tid;
            s_lse0[tid] = qq < N ? lse[((long)b * H + h) * N + qq] * 1.44269504088896340736f : INFINITY;  // base-2 domain
            s_del0[tid] = qq < N ? delta[((long)b * H + h) * N + qq] : 0.f;
        }
        __syncthreads();
        if (pre && qt0 + NTL < ntiles) tg.fetch(0, base_hi, dob_hi, base_lo, dob_lo, RS, OS, qt0 + NTL, N, tid, nthr);
        for (int j = 0; j < NTL && qt0 + j < ntiles; ++j) {
        const char *q_hi = q_hi0 + j * TILE, *d_hi = d_hi0 + j * TILE, *q_lo = q_lo0 + j * TILE, *d_lo = d_lo0 + j * TILE;
        const float *s_lse = s_lse0 + j * KT, *s_del = s_del0 + j * KT;
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bf16x8_t ah = frag_rows(q_hi, sub * 16, s, lane);
                bf16x8_t al = SPLIT ? frag_rows(q_lo, sub * 16, s, lane) : ah;
                st = mma<SPLIT>(ah, al, kh[s], kl[s], st);  // S[q][key]
                bf16x8_t bh = frag_rows(d_hi, sub * 16, s, lane);
                bf16x8_t bl = SPLIT ? frag_rows(d_lo, sub * 16, s, lane) : bh;
                dp = mma<SPLIT>(bh, bl, vh[s], vl[s], dp);  // dP[q][key]
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int ql_ = sub * 16 + 4 * g + r;
                float p = __builtin_amdgcn_exp2f(fmaf(st[r], c2, -s_lse[ql_]));  // padded queries: lse=+inf -> 0
                pp[sub][r] = p;
                ds[sub][r] = p * (dp[r] - s_del[ql_]);  // scale applied once to the finished dK tile
            }
        }
        bf16x8_t ph, pl, sh, sl;
        pack_acc<SPLIT>(pp[0], pp[1], ph, pl);
        pack_acc<SPLIT>(ds[0], ds[1], sh, sl);
        if constexpr (!SPLIT) pl = ph, sl = sh;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            bf16x8_t th = frag_tr(d_hi, dt * 16, lane);  // dO^T
            bf16x8_t tl = SPLIT ? frag_tr(d_lo, dt * 16, lane) : th;
            dv[dt] = mma<SPLIT>(th, tl, ph, pl, dv[dt]);
            bf16x8_t uh = frag_tr(q_hi, dt * 16, lane);  // Q^T
            bf16x8_t ul = SPLIT ? frag_tr(q_lo, dt * 16, lane) : uh;
            dk[dt] = mma<SPLIT>(uh, ul, sh, sl, dk[dt]);
        }
        }  // j
    }
    const int key = k0 + (lane & 15);
    if (key < N) {
        long orow = ((long)b * N + key) * RS + h * HD;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            float fk[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
            float fv[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
            store4_split(dqkv_hi, dqkv_lo, (size_t)orow + (long)H * HD + dt * 16 + 4 * g, fk);
            store4_split(dqkv_hi, dqkv_lo, (size_t)orow + 2L * H * HD + dt * 16 + 4 * g, fv);
        }
    }
}

// Workgroup geometry: up to 16 waves (one 16-row tile each) per workgroup, i.e. one workgroup per (batch, head) for N = 197
// (13 waves) and three for N = 589.  Smaller workgroups (2-3 per CU) measured slower: the kernels are issue-bound per tile.
inline void wave_geometry(int N, int& nblk, int& nw) {
    const int tiles = (N + 15) / 16;
    nblk = (tiles + 15) / 16;
    nw = (tiles + nblk - 1) / nblk;
}

inline bool attn_generic_env() {  // read per call (tests flip it)
    const char* e = getenv("IG_ATTN_GENERIC");
    return e && atoi(e) != 0;
}

}  // namespace

extern "C" {

// out[B][N][H*64] = softmax(q k^T / sqrt(64)) v ; lse[B][H][N] (may be NULL for inference)
int ig_attention_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H,
                     int head_dim, void* stream) {
    IG_REQUIRE(qkv_hi && out_hi, "ig_attention_fwd: null pointer");
    IG_REQUIRE(head_dim == HD || head_dim == 80, "ig_attention_fwd: head_dim must be 64 or 80 (got %d)", head_dim);
    IG_REQUIRE((qkv_lo == nullptr) == (out_lo == nullptr), "ig_attention_fwd: split pointers must be given for all tensors or none");
    if (B == 0 || N == 0) return IG_OK;
    if (head_dim != HD || attn_generic_env())  // 600M variants (16 heads of 80), or IG_ATTN_GENERIC=1 (tests, A/B)
        return ig_attention_generic_fwd(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, head_dim, stream);
    {
        const int rc = ig_attention2_fwd(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    int nblk, nw;
    wave_geometry(N, nblk, nw);
    dim3 grid(nblk, H, B), block(nw * 64);
    float scale = 1.0f / sqrtf((float)head_dim);
#define IG_ATTN_FWD(SPLIT_, NTL_, MAXT_)                                                                                     \
    {                                                                                                                         \
        ig_note_kernel("attn_fwd_kernel<%s,%d,%d>", SPLIT_ ? "true" : "false", NTL_, MAXT_);                                      \
        hipLaunchKernelGGL((attn_fwd_kernel<SPLIT_, NTL_, MAXT_>), grid, block, 0, (hipStream_t)stream, (const bf16_t*)qkv_hi,   \
                           (const bf16_t*)qkv_lo, (bf16_t*)out_hi, (bf16_t*)out_lo, lse, N, H, scale);                        \
    }
    if (qkv_lo) IG_ATTN_FWD(true, 2, 1024)
    else IG_ATTN_FWD(false, 4, 1024)
#undef IG_ATTN_FWD
    return ig_check_launch("ig_attention_fwd");
}

// dqkv[B][N][3][H][64] from dout, qkv, out, lse ; delta: device scratch float[B*H*N]
int ig_attention_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                     const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, float* dqkv_colsum, int B, int N,
                     int H, int head_dim, void* stream) {
    IG_REQUIRE(qkv_hi && out_hi && dout_hi && lse && delta && dqkv_hi, "ig_attention_bwd: null pointer");
    IG_REQUIRE(head_dim == HD || head_dim == 80, "ig_attention_bwd: head_dim must be 64 or 80 (got %d)", head_dim);
    bool split = qkv_lo != nullptr;
    IG_REQUIRE(split == (out_lo != nullptr) && split == (dout_lo != nullptr) && split == (dqkv_lo != nullptr),
               "ig_attention_bwd: split pointers must be given for all tensors or none");
    if (B == 0 || N == 0) return IG_OK;
    if (head_dim != HD || attn_generic_env()) {
        const int rc = ig_attention_generic_bwd(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, B, N, H, head_dim, stream);
        if (rc != IG_OK || !dqkv_colsum) return rc;
        return ig_colsum(dqkv_hi, dqkv_lo, dqkv_colsum, (long)B * N, 3 * H * head_dim, stream);
    }
    {
        const int rc = ig_attention2_bwd(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, dqkv_colsum, B, N, H, stream);
        if (rc != IG_ERR_UNSUPPORTED) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    // delta is produced by the query-owner pass (first launch) and consumed by the key-owner pass (second)
    int nblk, nw;
    wave_geometry(N, nblk, nw);
    dim3 grid(nblk, H, B), block(nw * 64);
    float scale = 1.0f / sqrtf((float)head_dim);
#define IG_ATTN_BWD(SPLIT_, NTL_, MAXT_)                                                                                     \
    {                                                                                                                         \
        hipLaunchKernelGGL((attn_bwd_dq_kernel<SPLIT_, NTL_, MAXT_>), grid, block, 0, st, (const bf16_t*)qkv_hi,                  \
                           (const bf16_t*)qkv_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, (const bf16_t*)out_hi,          \
                           (const bf16_t*)out_lo, lse, delta, (bf16_t*)dqkv_hi,                                                   \
                           (bf16_t*)dqkv_lo, N, H, scale);                                                                    \
        ig_note_kernel("attn_bwd_dq_kernel<%s,%d,%d>+attn_bwd_dkv_kernel<%s,%d,%d>", SPLIT_ ? "true" : "false", NTL_, MAXT_, SPLIT_ ? "true" : "false", NTL_, MAXT_); \
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<SPLIT_, NTL_, MAXT_>), grid, block, 0, st, (const bf16_t*)qkv_hi,                 \
                           (const bf16_t*)qkv_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, lse, delta, (bf16_t*)dqkv_hi,  \
                           (bf16_t*)dqkv_lo, N, H, scale);                                                                    \
    }
    if (split) IG_ATTN_BWD(true, 2, 1024)
    else IG_ATTN_BWD(false, 4, 1024)
#undef IG_ATTN_BWD
    if (dqkv_colsum) {
        const int rc = ig_colsum(dqkv_hi, dqkv_lo, dqkv_colsum, (long)B * N, 3 * H * head_dim, stream);
        if (rc != IG_OK) return rc;
    }
    return ig_check_launch("ig_attention_bwd");
}

}  // extern "C"
