// Attention for head dimensions other than 64 (any multiple of 16 up to 128): the Prithvi-EO-2.0 600M variants run 16 heads of
// 80 (model.py:154-167; timm Attention called from pritvhi.py:446-456 -> F.scaled_dot_product_attention).
//
// Two sets of kernels.  Round 4 (plain bf16): the streamed side of the head staged in LDS, transposed fragments by
// ds_read_b64_tr_b16 (attng_*_lds_kernel, further down).  Round 3 (kept for the split-precision mode, whose hi + lo images of a
// 257-token head do not fit the LDS, and for A/B runs): built on v_mfma_f32_16x16x16_bf16 (K = 16: 80 = 5 steps, no padding), without LDS: a wave owns a 16-row tile of one
// (batch, head) -- queries in the forward and the dQ pass, keys in the dK / dV pass -- and streams the other side in 16-row
// tiles.  Products are oriented so that the accumulator tile of the first product is directly the B operand of the second
// (rows 4 (lane >> 4) + j, column lane & 15 = its k / column layout):
//   forward        S^T[key][q] = K Q^T        P^T -> O^T[d][q]  += V^T[d][key] P^T[key][q]
//   dK / dV pass   S[q][key]   = Q K^T        P, dS -> dV^T[d][key] += dO^T[d][q] P[q][key],  dK^T[d][key] += Q^T[d][q] dS[q][key]
//   dQ pass        S^T, dP^T                  dS^T -> dQ^T[d][q] += K^T[d][key] dS^T[key][q]
// Row-major operands (Q, K, V, dO rows) are 8-byte fragment loads; the transposed A operands (V^T, dO^T, Q^T, K^T) are four
// 2-byte loads per fragment straight from global memory (L2 hits: the head's tensors are a few hundred KiB).  This is the
// coverage path for the 600M shapes, not a tuned kernel: the 64-wide heads of every benchmarked config run attention2.hip.
// SPLIT = the bf16x3 precision mode (hi*hi + hi*lo + lo*hi for every product).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4v;

__device__ __forceinline__ f32x4 mfma16(s16x4v a, s16x4v b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }
template <bool SPLIT>
__device__ __forceinline__ f32x4 mma16(s16x4v ah, s16x4v al, s16x4v bh, s16x4v bl, f32x4 c) {
    c = mfma16(ah, bh, c);
    if constexpr (SPLIT) {
        c = mfma16(ah, bl, c);
        c = mfma16(al, bh, c);
    }
    return c;
}
// 4 consecutive elements of one row (8-byte load)
__device__ __forceinline__ s16x4v row4(const bf16_t* base, long row_stride, int row, int col) {
    return *reinterpret_cast<const s16x4v*>(base + (long)row * row_stride + col);
}
// 4 consecutive ROWS of one column (transposed fragment): rows r0 .. r0 + 3 clamped to nrows - 1
__device__ __forceinline__ s16x4v col4(const bf16_t* base, long row_stride, int r0, int nrows, int col) {
    s16x4v v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (short)base[(long)min(r0 + j, nrows - 1) * row_stride + col];
    return v;
}
// pack an accumulator tile into the next product's B operand (hi [+ lo])
template <bool SPLIT>
__device__ __forceinline__ void pack4(const f32x4& a, s16x4v& hi, s16x4v& lo) {
    const uint32_t u0 = pack_bf2(a[0], a[1]), u1 = pack_bf2(a[2], a[3]);
    hi = __builtin_bit_cast(s16x4v, make_uint2(u0, u1));
    if constexpr (SPLIT) {
        const uint32_t v0 = pack_bf2(a[0] - __uint_as_float(u0 << 16), a[1] - __uint_as_float(u0 & 0xffff0000u));
        const uint32_t v1 = pack_bf2(a[2] - __uint_as_float(u1 << 16), a[3] - __uint_as_float(u1 & 0xffff0000u));
        lo = __builtin_bit_cast(s16x4v, make_uint2(v0, v1));
    } else {
        lo = hi;
    }
}
// reduce over the four lane groups that share lane & 15
__device__ __forceinline__ float grp_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float grp_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

constexpr int AG_WPB = 4;  // waves (16-row tiles) per workgroup

template <int HD, bool SPLIT>
__global__ __launch_bounds__(AG_WPB * 64) void attng_fwd_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                               bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                               float* __restrict__ lse, int N, int H, float scale) {
    constexpr int NS = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * AG_WPB + wave) * 16;
    if (q0 >= N) return;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* ql = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : qh;
    const bf16_t *kh = qh + H * HD, *kl = ql + H * HD, *vh = qh + 2 * H * HD, *vl = ql + 2 * H * HD;
    const int qr = min(q0 + li, N - 1);
    s16x4v bqh[NS], bql[NS];  // Q^T as B operand: column q = li, k = d = 16 s + 4 g + j
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        bqh[s] = row4(qh, RS, qr, 16 * s + 4 * g);
        bql[s] = SPLIT ? row4(ql, RS, qr, 16 * s + 4 * g) : bqh[s];
    }
    f32x4 acc[NS];  // O^T: rows d = 16 dt + 4 g + j, column q = li
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = scale * 1.44269504088896340736f;
    for (int k0 = 0; k0 < N; k0 += 16) {
        const int kr = min(k0 + li, N - 1);
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};  // S^T: rows key = k0 + 4 g + j, column q
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const s16x4v ah = row4(kh, RS, kr, 16 * s + 4 * g);
            const s16x4v al = SPLIT ? row4(kl, RS, kr, 16 * s + 4 * g) : ah;
            st = mma16<SPLIT>(ah, al, bqh[s], bql[s], st);
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = (k0 + 4 * g + j < N) ? st[j] * c2 : -INFINITY;
            mx = fmaxf(mx, st[j]);
        }
        mx = grp_max(mx);
        const float m_new = fmaxf(m_run, mx);  // finite: every tile holds at least one valid key
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        float ps = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            st[j] = __builtin_amdgcn_exp2f(st[j] - m_new);
            ps += st[j];
        }
        l_run = l_run * alpha + grp_sum(ps);
        m_run = m_new;
        s16x4v bph, bpl;
        pack4<SPLIT>(st, bph, bpl);
#pragma unroll
        for (int dt = 0; dt < NS; ++dt) {
            acc[dt] *= alpha;
            const s16x4v ah = col4(vh, RS, k0 + 4 * g, N, 16 * dt + li);
            const s16x4v al = SPLIT ? col4(vl, RS, k0 + 4 * g, N, 16 * dt + li) : ah;
            acc[dt] = mma16<SPLIT>(ah, al, bph, bpl, acc[dt]);
        }
    }
    if (q0 + li >= N) return;
    const float inv = 1.0f / l_run;
    const size_t orow = ((size_t)b * N + q0 + li) * OS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float f[4] = {acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv};
        store4_split(out_hi, SPLIT ? out_lo : nullptr, orow + 16 * dt + 4 * g, f);
    }
    if (lse && g == 0) lse[((long)b * H + h) * N + q0 + li] = (m_run + __builtin_amdgcn_logf(l_run)) * 0.69314718055994530942f;
}

// delta[b][h][q] = sum_d dO[q][d] * O[q][d]
template <int HD, bool SPLIT>
__global__ __launch_bounds__(256) void attng_delta_kernel(const bf16_t* __restrict__ o_hi, const bf16_t* __restrict__ o_lo,
                                                          const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                          float* __restrict__ delta, int N, int H, long total) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;  // (b, q, h)
    if (i >= total) return;
    const int h = (int)(i % H);
    const long bq = i / H;
    const int q = (int)(bq % N);
    const long b = bq / N;
    const size_t row = (size_t)bq * H * HD + (size_t)h * HD;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < HD; c += 8) {
        float a[8], d[8];
        load8_split(o_hi, SPLIT ? o_lo : nullptr, row + c, a);
        load8_split(do_hi, SPLIT ? do_lo : nullptr, row + c, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(a[e], d[e], s);
    }
    delta[(b * H + h) * N + q] = s;
}

// key-owner pass: dK, dV of 16 keys
template <int HD, bool SPLIT>
__global__ __launch_bounds__(AG_WPB * 64) void attng_bwd_dkv_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                   const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                                   float scale) {
    constexpr int NS = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = (blockIdx.x * AG_WPB + wave) * 16;
    if (k0 >= N) return;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* ql = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : qh;
    const bf16_t *kh = qh + H * HD, *kl = ql + H * HD, *vh = qh + 2 * H * HD, *vl = ql + 2 * H * HD;
    const bf16_t* dh = do_hi + (long)b * N * OS + h * HD;
    const bf16_t* dl = SPLIT ? do_lo + (long)b * N * OS + h * HD : dh;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    const int kr = min(k0 + li, N - 1);
    const float kvalid = (k0 + li < N) ? 1.f : 0.f;
    s16x4v bkh[NS], bkl[NS], bvh[NS], bvl[NS];  // K^T / V^T as B operands: column key = li, k = d
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        bkh[s] = row4(kh, RS, kr, 16 * s + 4 * g), bvh[s] = row4(vh, RS, kr, 16 * s + 4 * g);
        bkl[s] = SPLIT ? row4(kl, RS, kr, 16 * s + 4 * g) : bkh[s];
        bvl[s] = SPLIT ? row4(vl, RS, kr, 16 * s + 4 * g) : bvh[s];
    }
    f32x4 dk[NS], dv[NS];  // dK^T / dV^T: rows d = 16 dt + 4 g + j, column key
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}, dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = scale * 1.44269504088896340736f;
    for (int q0 = 0; q0 < N; q0 += 16) {
        const int qr = min(q0 + li, N - 1);
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};  // rows q = q0 + 4 g + j, column key
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const s16x4v aqh = row4(qh, RS, qr, 16 * s + 4 * g), adh = row4(dh, OS, qr, 16 * s + 4 * g);
            const s16x4v aql = SPLIT ? row4(ql, RS, qr, 16 * s + 4 * g) : aqh;
            const s16x4v adl = SPLIT ? row4(dl, OS, qr, 16 * s + 4 * g) : adh;
            st = mma16<SPLIT>(aqh, aql, bkh[s], bkl[s], st);
            dp = mma16<SPLIT>(adh, adl, bvh[s], bvl[s], dp);
        }
        f32x4 ds;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = q0 + 4 * g + j;
            const bool ok = q < N;
            const float l2 = ok ? lrow[q] * 1.44269504088896340736f : INFINITY;  // +inf -> P = 0 on padded queries
            const float de = ok ? drow[q] : 0.f;
            const float pv = __builtin_amdgcn_exp2f(fmaf(st[j], c2, -l2)) * kvalid;
            st[j] = pv;
            ds[j] = pv * (dp[j] - de);
        }
        s16x4v bph, bpl, bsh, bsl;
        pack4<SPLIT>(st, bph, bpl);
        pack4<SPLIT>(ds, bsh, bsl);
#pragma unroll
        for (int dt = 0; dt < NS; ++dt) {
            const s16x4v adh = col4(dh, OS, q0 + 4 * g, N, 16 * dt + li), aqh = col4(qh, RS, q0 + 4 * g, N, 16 * dt + li);
            const s16x4v adl = SPLIT ? col4(dl, OS, q0 + 4 * g, N, 16 * dt + li) : adh;
            const s16x4v aql = SPLIT ? col4(ql, RS, q0 + 4 * g, N, 16 * dt + li) : aqh;
            dv[dt] = mma16<SPLIT>(adh, adl, bph, bpl, dv[dt]);
            dk[dt] = mma16<SPLIT>(aqh, aql, bsh, bsl, dk[dt]);
        }
    }
    if (k0 + li >= N) return;
    const size_t orow = ((size_t)b * N + k0 + li) * RS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float fk[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
        const float fv[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
        store4_split(dqkv_hi, SPLIT ? dqkv_lo : nullptr, orow + (size_t)H * HD + 16 * dt + 4 * g, fk);
        store4_split(dqkv_hi, SPLIT ? dqkv_lo : nullptr, orow + 2 * (size_t)H * HD + 16 * dt + 4 * g, fv);
    }
}

// query-owner pass: dQ of 16 queries
template <int HD, bool SPLIT>
__global__ __launch_bounds__(AG_WPB * 64) void attng_bwd_dq_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ qkv_lo,
                                                                  const bf16_t* __restrict__ do_hi, const bf16_t* __restrict__ do_lo,
                                                                  const float* __restrict__ lse, const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dqkv_hi, bf16_t* __restrict__ dqkv_lo, int N, int H,
                                                                  float scale) {
    constexpr int NS = HD / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * AG_WPB + wave) * 16;
    if (q0 >= N) return;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t* ql = SPLIT ? qkv_lo + (long)b * N * RS + h * HD : qh;
    const bf16_t *kh = qh + H * HD, *kl = ql + H * HD, *vh = qh + 2 * H * HD, *vl = ql + 2 * H * HD;
    const bf16_t* dh = do_hi + (long)b * N * OS + h * HD;
    const bf16_t* dl = SPLIT ? do_lo + (long)b * N * OS + h * HD : dh;
    const int qr = min(q0 + li, N - 1);
    const bool qok = q0 + li < N;
    const float l2 = qok ? lse[((long)b * H + h) * N + q0 + li] * 1.44269504088896340736f : INFINITY;
    const float de = qok ? delta[((long)b * H + h) * N + q0 + li] : 0.f;
    s16x4v bqh[NS], bql[NS], bdh[NS], bdl[NS];  // Q^T / dO^T as B operands: column q = li, k = d
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        bqh[s] = row4(qh, RS, qr, 16 * s + 4 * g), bdh[s] = row4(dh, OS, qr, 16 * s + 4 * g);
        bql[s] = SPLIT ? row4(ql, RS, qr, 16 * s + 4 * g) : bqh[s];
        bdl[s] = SPLIT ? row4(dl, OS, qr, 16 * s + 4 * g) : bdh[s];
    }
    f32x4 dq[NS];  // dQ^T: rows d = 16 dt + 4 g + j, column q
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = scale * 1.44269504088896340736f;
    for (int k0 = 0; k0 < N; k0 += 16) {
        const int kr = min(k0 + li, N - 1);
        f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};  // rows key = k0 + 4 g + j, column q
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const s16x4v akh = row4(kh, RS, kr, 16 * s + 4 * g), avh = row4(vh, RS, kr, 16 * s + 4 * g);
            const s16x4v akl = SPLIT ? row4(kl, RS, kr, 16 * s + 4 * g) : akh;
            const s16x4v avl = SPLIT ? row4(vl, RS, kr, 16 * s + 4 * g) : avh;
            st = mma16<SPLIT>(akh, akl, bqh[s], bql[s], st);
            dp = mma16<SPLIT>(avh, avl, bdh[s], bdl[s], dp);
        }
        f32x4 ds;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float pv = (k0 + 4 * g + j < N) ? __builtin_amdgcn_exp2f(fmaf(st[j], c2, -l2)) : 0.f;
            ds[j] = pv * (dp[j] - de);
        }
        s16x4v bsh, bsl;
        pack4<SPLIT>(ds, bsh, bsl);
#pragma unroll
        for (int dt = 0; dt < NS; ++dt) {
            const s16x4v akh = col4(kh, RS, k0 + 4 * g, N, 16 * dt + li);
            const s16x4v akl = SPLIT ? col4(kl, RS, k0 + 4 * g, N, 16 * dt + li) : akh;
            dq[dt] = mma16<SPLIT>(akh, akl, bsh, bsl, dq[dt]);
        }
    }
    if (!qok) return;
    const size_t orow = ((size_t)b * N + q0 + li) * RS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float f[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
        store4_split(dqkv_hi, SPLIT ? dqkv_lo : nullptr, orow + 16 * dt + 4 * g, f);
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// LDS-staged variants (plain bf16): a workgroup = 8 waves = 128 owner rows of one (batch, head); the streamed side of the head is
// staged ONCE per workgroup and chunk of AGL_CH rows into two row-major LDS images, and every fragment of the inner loop comes from
// LDS: row operands as 8-byte reads, the transposed A operands (V^T, dO^T, Q^T, K^T) as ds_read_b64_tr_b16 on the same row-major
// image (16 lanes read a 4-row x 16-column block, lane i receives column i) -- the kernels above fetch them as four 2-byte loads from
// L2 per fragment.  Image pitches: 2 HD + 16 bytes = 176 for HD = 80 (44 dwords: the sixteen rows of an 8-byte row read fall on
// disjoint banks) for images read by rows, 2 HD = 160 for an image only read transposed (rows q, q + 4 of a tr block: 40-dword steps).
constexpr int AGL_WPB = 8;
constexpr int AGL_CH = 272;  // rows of the streamed side per chunk (17 tiles: the 257 tokens of a 600M chip in one chunk)
typedef __attribute__((address_space(3))) s16x4v* agl_lds_ptr;

template <int HD>
__device__ __forceinline__ void agl_fill(char* img, int pitch, const bf16_t* src, long row_stride, int r0, int nrows, int N, int tid, int nthr) {
    constexpr int UPR = HD / 8;  // 16-byte units per row
    for (int u = tid; u < nrows * UPR; u += nthr) {
        const int r = u / UPR, c = u - r * UPR;
        const uint4 v = *reinterpret_cast<const uint4*>(src + (long)min(r0 + r, N - 1) * row_stride + c * 8);
        *reinterpret_cast<uint4*>(img + r * pitch + c * 16) = v;
    }
}
__device__ __forceinline__ s16x4v agl_row4(const char* img, int pitch, int row, int col) {
    return *reinterpret_cast<const s16x4v*>(img + row * pitch + col * 2);
}
// transposed fragment: 4 consecutive ROWS r0 + 4 g .. + 3 of column c0 + li  (A operand X^T[m = column][k = row])
__device__ __forceinline__ s16x4v agl_col4(const char* img, int pitch, int r0, int c0, int li, int g) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((agl_lds_ptr)(img + (r0 + 4 * g + (li >> 2)) * pitch + (c0 + 4 * (li & 3)) * 2));
}

template <int HD>
__global__ __launch_bounds__(AGL_WPB * 64) void attng_fwd_lds_kernel(const bf16_t* __restrict__ qkv_hi, bf16_t* __restrict__ out_hi,
                                                                    float* __restrict__ lse, int N, int H, float scale) {
    constexpr int NS = HD / 16, PK = 2 * HD + 16, PV = 2 * HD;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_img = smem;
    char* v_img = smem + AGL_CH * PK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * AGL_WPB + wave) * 16;
    const bool active = q0 < N;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t *kh = qh + H * HD, *vh = qh + 2 * H * HD;
    const int qr = min(q0 + li, N - 1);
    s16x4v bq[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[s] = row4(qh, RS, qr, 16 * s + 4 * g);
    f32x4 acc[NS];
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float c2 = scale * 1.44269504088896340736f;
    for (int c0 = 0; c0 < N; c0 += AGL_CH) {
        const int nr = min(AGL_CH, (N - c0 + 15) / 16 * 16);
        __syncthreads();
        agl_fill<HD>(k_img, PK, kh, RS, c0, nr, N, tid, AGL_WPB * 64);
        agl_fill<HD>(v_img, PV, vh, RS, c0, nr, N, tid, AGL_WPB * 64);
        __syncthreads();
        if (!active) continue;
        for (int kk = 0; kk < nr; kk += 16) {
            const int k0 = c0 + kk;
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) st = mfma16(agl_row4(k_img, PK, kk + li, 16 * s + 4 * g), bq[s], st);
            float mx = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st[j] = (k0 + 4 * g + j < N) ? st[j] * c2 : -INFINITY;
                mx = fmaxf(mx, st[j]);
            }
            mx = grp_max(mx);
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                st[j] = __builtin_amdgcn_exp2f(st[j] - m_new);
                ps += st[j];
            }
            l_run = l_run * alpha + grp_sum(ps);
            m_run = m_new;
            s16x4v bp, bpl;
            pack4<false>(st, bp, bpl);
#pragma unroll
            for (int dt = 0; dt < NS; ++dt) {
                acc[dt] *= alpha;
                acc[dt] = mfma16(agl_col4(v_img, PV, kk, 16 * dt, li, g), bp, acc[dt]);
            }
        }
    }
    if (!active || q0 + li >= N) return;
    const float inv = 1.0f / l_run;
    const size_t orow = ((size_t)b * N + q0 + li) * OS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float f[4] = {acc[dt][0] * inv, acc[dt][1] * inv, acc[dt][2] * inv, acc[dt][3] * inv};
        store4_split(out_hi, nullptr, orow + 16 * dt + 4 * g, f);
    }
    if (lse && g == 0) lse[((long)b * H + h) * N + q0 + li] = (m_run + __builtin_amdgcn_logf(l_run)) * 0.69314718055994530942f;
}

// key-owner pass (dK, dV of 16 keys per wave): Q and dO of the head staged in LDS (both read by rows AND transposed: pitch 2 HD + 16)
template <int HD>
__global__ __launch_bounds__(AGL_WPB * 64) void attng_bwd_dkv_lds_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ do_hi,
                                                                        const float* __restrict__ lse, const float* __restrict__ delta,
                                                                        bf16_t* __restrict__ dqkv_hi, int N, int H, float scale) {
    constexpr int NS = HD / 16, PQ = 2 * HD + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* q_img = smem;
    char* d_img = smem + AGL_CH * PQ;
    float* s_l = reinterpret_cast<float*>(smem + 2 * AGL_CH * PQ);
    float* s_d = s_l + AGL_CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int k0 = (blockIdx.x * AGL_WPB + wave) * 16;
    const bool active = k0 < N;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t *kh = qh + H * HD, *vh = qh + 2 * H * HD;
    const bf16_t* dh = do_hi + (long)b * N * OS + h * HD;
    const float* lrow = lse + ((long)b * H + h) * N;
    const float* drow = delta + ((long)b * H + h) * N;
    const int kr = min(k0 + li, N - 1);
    const float kvalid = (k0 + li < N) ? 1.f : 0.f;
    s16x4v bk[NS], bv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bk[s] = row4(kh, RS, kr, 16 * s + 4 * g), bv[s] = row4(vh, RS, kr, 16 * s + 4 * g);
    f32x4 dk[NS], dv[NS];
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}, dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = scale * 1.44269504088896340736f;
    for (int c0 = 0; c0 < N; c0 += AGL_CH) {
        const int nr = min(AGL_CH, (N - c0 + 15) / 16 * 16);
        __syncthreads();
        agl_fill<HD>(q_img, PQ, qh, RS, c0, nr, N, tid, AGL_WPB * 64);
        agl_fill<HD>(d_img, PQ, dh, OS, c0, nr, N, tid, AGL_WPB * 64);
        for (int r = tid; r < nr; r += AGL_WPB * 64) {
            const bool ok = c0 + r < N;
            s_l[r] = ok ? lrow[c0 + r] * 1.44269504088896340736f : INFINITY;  // +inf -> P = 0 on padded queries
            s_d[r] = ok ? drow[c0 + r] : 0.f;
        }
        __syncthreads();
        if (!active) continue;
        for (int qq = 0; qq < nr; qq += 16) {
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};  // rows q = qq + 4 g + j, column key
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                st = mfma16(agl_row4(q_img, PQ, qq + li, 16 * s + 4 * g), bk[s], st);
                dp = mfma16(agl_row4(d_img, PQ, qq + li, 16 * s + 4 * g), bv[s], dp);
            }
            const float4 l4 = *reinterpret_cast<const float4*>(s_l + qq + 4 * g), d4 = *reinterpret_cast<const float4*>(s_d + qq + 4 * g);
            const float ll[4] = {l4.x, l4.y, l4.z, l4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w};
            f32x4 ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = __builtin_amdgcn_exp2f(fmaf(st[j], c2, -ll[j])) * kvalid;
                st[j] = pv;
                ds[j] = pv * (dp[j] - dd[j]);
            }
            s16x4v bp, bpl, bs, bsl;
            pack4<false>(st, bp, bpl);
            pack4<false>(ds, bs, bsl);
#pragma unroll
            for (int dt = 0; dt < NS; ++dt) {
                dv[dt] = mfma16(agl_col4(d_img, PQ, qq, 16 * dt, li, g), bp, dv[dt]);
                dk[dt] = mfma16(agl_col4(q_img, PQ, qq, 16 * dt, li, g), bs, dk[dt]);
            }
        }
    }
    if (!active || k0 + li >= N) return;
    const size_t orow = ((size_t)b * N + k0 + li) * RS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float fk[4] = {dk[dt][0] * scale, dk[dt][1] * scale, dk[dt][2] * scale, dk[dt][3] * scale};
        const float fv[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
        store4_split(dqkv_hi, nullptr, orow + (size_t)H * HD + 16 * dt + 4 * g, fk);
        store4_split(dqkv_hi, nullptr, orow + 2 * (size_t)H * HD + 16 * dt + 4 * g, fv);
    }
}

// query-owner pass (dQ of 16 queries per wave): K (rows and transposed) and V (rows) of the head staged in LDS
template <int HD>
__global__ __launch_bounds__(AGL_WPB * 64) void attng_bwd_dq_lds_kernel(const bf16_t* __restrict__ qkv_hi, const bf16_t* __restrict__ do_hi,
                                                                       const float* __restrict__ lse, const float* __restrict__ delta,
                                                                       bf16_t* __restrict__ dqkv_hi, int N, int H, float scale) {
    constexpr int NS = HD / 16, PK = 2 * HD + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* k_img = smem;
    char* v_img = smem + AGL_CH * PK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * AGL_WPB + wave) * 16;
    const bool active = q0 < N;
    const long RS = 3L * H * HD, OS = (long)H * HD;
    const bf16_t* qh = qkv_hi + (long)b * N * RS + h * HD;
    const bf16_t *kh = qh + H * HD, *vh = qh + 2 * H * HD;
    const bf16_t* dh = do_hi + (long)b * N * OS + h * HD;
    const int qr = min(q0 + li, N - 1);
    const bool qok = q0 + li < N;
    const float l2 = qok ? lse[((long)b * H + h) * N + q0 + li] * 1.44269504088896340736f : INFINITY;
    const float de = qok ? delta[((long)b * H + h) * N + q0 + li] : 0.f;
    s16x4v bq[NS], bd[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) bq[s] = row4(qh, RS, qr, 16 * s + 4 * g), bd[s] = row4(dh, OS, qr, 16 * s + 4 * g);
    f32x4 dq[NS];
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float c2 = scale * 1.44269504088896340736f;
    for (int c0 = 0; c0 < N; c0 += AGL_CH) {
        const int nr = min(AGL_CH, (N - c0 + 15) / 16 * 16);
        __syncthreads();
        agl_fill<HD>(k_img, PK, kh, RS, c0, nr, N, tid, AGL_WPB * 64);
        agl_fill<HD>(v_img, PK, vh, RS, c0, nr, N, tid, AGL_WPB * 64);
        __syncthreads();
        if (!active) continue;
        for (int kk = 0; kk < nr; kk += 16) {
            const int k0 = c0 + kk;
            f32x4 st = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};  // rows key = k0 + 4 g + j, column q
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                st = mfma16(agl_row4(k_img, PK, kk + li, 16 * s + 4 * g), bq[s], st);
                dp = mfma16(agl_row4(v_img, PK, kk + li, 16 * s + 4 * g), bd[s], dp);
            }
            f32x4 ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float pv = (k0 + 4 * g + j < N) ? __builtin_amdgcn_exp2f(fmaf(st[j], c2, -l2)) : 0.f;
                ds[j] = pv * (dp[j] - de);
            }
            s16x4v bs, bsl;
            pack4<false>(ds, bs, bsl);
#pragma unroll
            for (int dt = 0; dt < NS; ++dt) dq[dt] = mfma16(agl_col4(k_img, PK, kk, 16 * dt, li, g), bs, dq[dt]);
        }
    }
    if (!active || !qok) return;
    const size_t orow = ((size_t)b * N + q0 + li) * RS + h * HD;
#pragma unroll
    for (int dt = 0; dt < NS; ++dt) {
        const float f[4] = {dq[dt][0] * scale, dq[dt][1] * scale, dq[dt][2] * scale, dq[dt][3] * scale};
        store4_split(dqkv_hi, nullptr, orow + 16 * dt + 4 * g, f);
    }
}

constexpr bool agl_enabled() { return true; }  // plain bf16: the LDS-staged kernels (round 4: backward 587 -> 252 us); split mode: the register / L2 kernels above
template <class K>
inline bool agl_attr(K kern, int bytes) {
    return hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
}

template <int HD>
int attng_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H, hipStream_t st) {
    const dim3 grid(ig_cdiv(ig_cdiv(N, 16), AG_WPB), H, B), block(AG_WPB * 64);
    const float scale = 1.0f / sqrtf((float)HD);
    if (!qkv_lo && agl_enabled()) {
        constexpr int smem = AGL_CH * (2 * HD + 16) + AGL_CH * 2 * HD;
        static bool attr = false;
        if (!attr) attr = agl_attr(attng_fwd_lds_kernel<HD>, smem);
        if (attr) {
            ig_note_kernel("attng_fwd_lds_kernel<%d>", HD);
            hipLaunchKernelGGL((attng_fwd_lds_kernel<HD>), dim3(ig_cdiv(ig_cdiv(N, 16), AGL_WPB), H, B), dim3(AGL_WPB * 64), smem, st,
                               (const bf16_t*)qkv_hi, (bf16_t*)out_hi, lse, N, H, scale);
            return ig_check_launch("ig_attention_fwd");
        }
    }
    ig_note_kernel("attng_fwd_kernel<%d,%s>", HD, qkv_lo ? "true" : "false");
    if (qkv_lo)
        hipLaunchKernelGGL((attng_fwd_kernel<HD, true>), grid, block, 0, st, (const bf16_t*)qkv_hi, (const bf16_t*)qkv_lo, (bf16_t*)out_hi,
                           (bf16_t*)out_lo, lse, N, H, scale);
    else
        hipLaunchKernelGGL((attng_fwd_kernel<HD, false>), grid, block, 0, st, (const bf16_t*)qkv_hi, (const bf16_t*)nullptr, (bf16_t*)out_hi,
                           (bf16_t*)nullptr, lse, N, H, scale);
    return ig_check_launch("ig_attention_fwd");
}

template <int HD>
int attng_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi, const void* dout_lo,
              const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, int B, int N, int H, hipStream_t st) {
    const dim3 grid(ig_cdiv(ig_cdiv(N, 16), AG_WPB), H, B), block(AG_WPB * 64);
    const float scale = 1.0f / sqrtf((float)HD);
    const long total = (long)B * N * H;
    if (!qkv_lo && agl_enabled()) {
        constexpr int smem_kv = 2 * AGL_CH * (2 * HD + 16) + 2 * AGL_CH * 4, smem_q = 2 * AGL_CH * (2 * HD + 16);
        static bool attr = false;
        if (!attr) attr = agl_attr(attng_bwd_dkv_lds_kernel<HD>, smem_kv) && agl_attr(attng_bwd_dq_lds_kernel<HD>, smem_q);
        if (attr) {
            const dim3 gl(ig_cdiv(ig_cdiv(N, 16), AGL_WPB), H, B), bl(AGL_WPB * 64);
            ig_note_kernel("attng_bwd_dkv_lds_kernel<%d>+attng_bwd_dq_lds_kernel", HD);
            hipLaunchKernelGGL((attng_delta_kernel<HD, false>), dim3(ig_cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)out_hi,
                               (const bf16_t*)out_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, delta, N, H, total);
            hipLaunchKernelGGL((attng_bwd_dkv_lds_kernel<HD>), gl, bl, smem_kv, st, (const bf16_t*)qkv_hi, (const bf16_t*)dout_hi, lse,
                               (const float*)delta, (bf16_t*)dqkv_hi, N, H, scale);
            hipLaunchKernelGGL((attng_bwd_dq_lds_kernel<HD>), gl, bl, smem_q, st, (const bf16_t*)qkv_hi, (const bf16_t*)dout_hi, lse,
                               (const float*)delta, (bf16_t*)dqkv_hi, N, H, scale);
            return ig_check_launch("ig_attention_bwd");
        }
    }
    ig_note_kernel("attng_bwd_dkv_kernel<%d,%s>+attng_bwd_dq_kernel", HD, qkv_lo ? "true" : "false");
#define AG_BWD(SPLIT_)                                                                                                             \
    {                                                                                                                              \
        hipLaunchKernelGGL((attng_delta_kernel<HD, SPLIT_>), dim3(ig_cdiv(total, 256)), dim3(256), 0, st, (const bf16_t*)out_hi,   \
                           (const bf16_t*)out_lo, (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, delta, N, H, total);             \
        hipLaunchKernelGGL((attng_bwd_dkv_kernel<HD, SPLIT_>), grid, block, 0, st, (const bf16_t*)qkv_hi, (const bf16_t*)qkv_lo,   \
                           (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, lse, (const float*)delta, (bf16_t*)dqkv_hi,             \
                           (bf16_t*)dqkv_lo, N, H, scale);                                                                         \
        hipLaunchKernelGGL((attng_bwd_dq_kernel<HD, SPLIT_>), grid, block, 0, st, (const bf16_t*)qkv_hi, (const bf16_t*)qkv_lo,    \
                           (const bf16_t*)dout_hi, (const bf16_t*)dout_lo, lse, (const float*)delta, (bf16_t*)dqkv_hi,             \
                           (bf16_t*)dqkv_lo, N, H, scale);                                                                         \
    }
    if (qkv_lo) AG_BWD(true)
    else AG_BWD(false)
#undef AG_BWD
    return ig_check_launch("ig_attention_bwd");
}

}  // namespace

// head dimensions served here: 80 (Prithvi-EO-2.0 600M) and, for A/B and tests against the tuned kernels, 64
int ig_attention_generic_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H,
                             int head_dim, void* stream) {
    if (head_dim == 80) return attng_fwd<80>(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, (hipStream_t)stream);
    if (head_dim == 64) return attng_fwd<64>(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, (hipStream_t)stream);
    return IG_ERR_UNSUPPORTED;
}
int ig_attention_generic_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                             const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, int B, int N, int H,
                             int head_dim, void* stream) {
    if (head_dim == 80)
        return attng_bwd<80>(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, B, N, H, (hipStream_t)stream);
    if (head_dim == 64)
        return attng_bwd<64>(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, B, N, H, (hipStream_t)stream);
    return IG_ERR_UNSUPPORTED;
}
