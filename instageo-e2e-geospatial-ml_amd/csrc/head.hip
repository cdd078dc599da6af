// Classifier (1x1 conv), segmentation loss, argmax and streaming confusion matrix for gfx950.
// HBM-bound per-pixel kernels: one thread per pixel, 16-byte NHWC reads, plane-coalesced NCHW logits.
//   K14 nn.Conv2d(48T -> ncls, k=1) preceded by Dropout(0.1)      model.py:388-389
//   K15 CrossEntropyLoss(weight, ignore_index, 'none') + loss[mask].mean()   segmentation.py:85-87,117-122
//   K16 argmax / softmax                                              segmentation.py:125-126, infer_utils.py:99-101
//   K17 np.bincount(y_true*k + y_pred) confusion matrix (int64)       metrics.py:86-108
#include "common.h"
#include "bn_fold.h"

namespace {

// m = b * HW + pix.  A 64-bit integer division is ~150 instructions on this hardware and sat in per-pixel loops; the 32-bit form
// (pixel counts below 2^31: every real batch) is ~30.
__device__ __forceinline__ void split_pixel(long m, long HW, long& b, long& pix) {
    if ((unsigned long)(m | HW) < (1ul << 31)) {
        const unsigned bu = (unsigned)m / (unsigned)HW;
        b = bu, pix = (long)((unsigned)m - bu * (unsigned)HW);
    } else {
        b = m / HW, pix = m - b * HW;
    }
}

// Grid-wide sums in a FIXED order (the reported loss statistics are bit-identical from run to run): thread i < NV of every workgroup
// hands in its workgroup's partial t; partials go to scratch[workgroup][i]; the workgroup that takes the last ticket adds them up in
// workgroup order (16 strided subsets per value, folded in subset order) and gets the totals back in t of its threads i < NV.
// Returns true in that workgroup only.  NV <= 16, blockDim.x == 256.  Launches that share the scratch are ordered on one stream.
template <int NV>
__device__ __forceinline__ bool ordered_grid_totals(double& t, double* __restrict__ scratch, unsigned* __restrict__ ticket) {
    static_assert(NV <= 16, "at most 16 values");
    __shared__ int s_last;
    __shared__ double s_sub[16][16];
    if (threadIdx.x < NV) scratch[(size_t)blockIdx.x * NV + threadIdx.x] = t;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return false;
    __threadfence();
    const int i = threadIdx.x & 15, g = threadIdx.x >> 4;
    double a = 0.0;
    if (i < NV)
        for (unsigned b = g; b < gridDim.x; b += 16) a += *reinterpret_cast<volatile double*>(scratch + (size_t)b * NV + i);
    s_sub[g][i] = a;
    __syncthreads();
    if (threadIdx.x < NV) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += s_sub[q][threadIdx.x];
        t = s;
    }
    if (threadIdx.x == 0) atomicExch(ticket, 0u);
    return true;
}

constexpr int TPB = 256;
constexpr int MAXC = 16;  // max classes held in registers

// Copy `total` 16-byte units (units per pixel row) from global to LDS rows of `pitch` bytes: consecutive threads take
// consecutive units; four loads are in flight per thread before the LDS stores.
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ src, char* dst, int total, int units, int pitch) {
    for (int base = threadIdx.x; base < total; base += TPB * 4) {
        uint4 b0 = make_uint4(0, 0, 0, 0), b1 = b0, b2 = b0, b3 = b0;
        const int u0 = base, u1 = base + TPB, u2 = base + 2 * TPB, u3 = base + 3 * TPB;
        if (u0 < total) b0 = *reinterpret_cast<const uint4*>(src + (size_t)u0 * 8);
        if (u1 < total) b1 = *reinterpret_cast<const uint4*>(src + (size_t)u1 * 8);
        if (u2 < total) b2 = *reinterpret_cast<const uint4*>(src + (size_t)u2 * 8);
        if (u3 < total) b3 = *reinterpret_cast<const uint4*>(src + (size_t)u3 * 8);
        if (u0 < total) *reinterpret_cast<uint4*>(dst + (u0 / units) * pitch + (u0 % units) * 16) = b0;
        if (u1 < total) *reinterpret_cast<uint4*>(dst + (u1 / units) * pitch + (u1 % units) * 16) = b1;
        if (u2 < total) *reinterpret_cast<uint4*>(dst + (u2 / units) * pitch + (u2 % units) * 16) = b2;
        if (u3 < total) *reinterpret_cast<uint4*>(dst + (u3 / units) * pitch + (u3 % units) * 16) = b3;
    }
}

// The same for a window of `cu` units starting at unit `c0` of rows that are `units` units long (channel-chunked staging).
__device__ __forceinline__ void stage_cols(const bf16_t* __restrict__ src, char* dst, int nrows, int units, int c0, int cu, int pitch) {
    const int total = nrows * cu;
    for (int base = threadIdx.x; base < total; base += TPB * 4) {
        uint4 b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = base + k * TPB;
            b[k] = make_uint4(0, 0, 0, 0);
            if (u < total) b[k] = *reinterpret_cast<const uint4*>(src + ((size_t)(u / cu) * units + c0 + (u % cu)) * 8);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int u = base + k * TPB;
            if (u < total) *reinterpret_cast<uint4*>(dst + (u / cu) * pitch + (u % cu) * 16) = b[k];
        }
    }
}

// logits[b][n][pix] = bias[n] + sum_c drop(f[b][pix][c]) * w[n][c]
// One thread per pixel, but the workgroup's 256 pixels x C channels are first copied to LDS with fully coalesced 16-byte
// loads (a thread reading its own 96-byte pixel row straight from global touched 48 cache lines per wave-instruction);
// LDS rows are padded by 16 bytes to spread the per-pixel ds_read_b128 over the banks.  The pixels are staged in CHUNKS of
// CLS_CHUNK channels (accumulators stay in registers across chunks): staging all 144 channels of the multi-temporal head took
// 78 KiB -- one workgroup per CU -- and the kernel ran at 0.8 TB/s (780 us) where the 48-channel case streams at 3.4.
constexpr int CLS_CHUNK = 48;
constexpr int CLS_IG = 8;  // wide backward: iterations per staged dlogits group
__global__ __launch_bounds__(TPB) void classifier_fwd_kernel(const bf16_t* __restrict__ f_hi, const bf16_t* __restrict__ f_lo,
                                                             const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ logits, long M, long HW, int C, int ncls,
                                                             uint32_t drop_seed, const uint32_t* drop_seed_dev, uint32_t drop_thresh,
                                                             float drop_inv, const float* __restrict__ bn_scale,
                                                             const float* __restrict__ bn_shift) {
    // [ncls][C] + [ncls] (padded to 4) [+ BatchNorm scale [C], shift [C]] | staged pixel chunk (hi [, lo])
    extern __shared__ __attribute__((aligned(16))) float sw[];
    if (drop_seed_dev) drop_seed += *drop_seed_dev;
    const int units = C / 8;
    const int cmax = min(units, CLS_CHUNK / 8), pitch = cmax * 16 + 16;  // bytes per staged pixel row
    const float* sbn = sw + ncls * C + ((ncls + 3) & ~3);
    char* stage = reinterpret_cast<char*>(sw + ncls * C + ((ncls + 3) & ~3) + (bn_scale ? 2 * C : 0));
    for (int i = threadIdx.x; i < ncls * C; i += TPB) sw[i] = w[i];
    for (int i = threadIdx.x; i < ncls; i += TPB) sw[ncls * C + i] = bias[i];
    if (bn_scale)
        for (int i = threadIdx.x; i < C; i += TPB) {
            sw[ncls * C + ((ncls + 3) & ~3) + i] = bn_scale[i];
            sw[ncls * C + ((ncls + 3) & ~3) + C + i] = bn_shift[i];
        }
    const long m0 = blockIdx.x * (long)TPB;
    const int npix = (int)min((long)TPB, M - m0);
    const long m = m0 + threadIdx.x;
    float acc[MAXC];
#pragma unroll
    for (int n = 0; n < MAXC; ++n) acc[n] = 0.f;
    const char* row = stage + threadIdx.x * pitch;
    for (int c0 = 0; c0 < units; c0 += cmax) {
        const int cu = min(cmax, units - c0);
        if (c0) __syncthreads();  // the previous chunk has been consumed
        if (cu == units) {  // one chunk = whole rows: contiguous copy
            stage_rows(f_hi + (size_t)m0 * C, stage, npix * units, units, pitch);
            if (f_lo) stage_rows(f_lo + (size_t)m0 * C, stage + (size_t)TPB * pitch, npix * units, units, pitch);
        } else {
            stage_cols(f_hi + (size_t)m0 * C, stage, npix, units, c0, cu, pitch);
            if (f_lo) stage_cols(f_lo + (size_t)m0 * C, stage + (size_t)TPB * pitch, npix, units, c0, cu, pitch);
        }
        __syncthreads();
        if (m >= M) continue;
        for (int c8 = 0; c8 < cu; ++c8) {
            float f[8];
            unpack8(*reinterpret_cast<const uint4*>(row + c8 * 16), f);
            if (f_lo) {
                float g[8];
                unpack8(*reinterpret_cast<const uint4*>(row + (size_t)TPB * pitch + c8 * 16), g);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] += g[j];
            }
            if (bn_scale) {  // the operand is the last Conv2d's output: training-mode BatchNorm + ReLU applied here, not in a pass of its own
                const float* sc = sbn + (c0 + c8) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sc[C + j]), 0.f);
            }
            if (drop_thresh) {
                const size_t idx = (size_t)m * C + (c0 + c8) * 8;
                float mk[8];
                dropout_scale4(drop_seed, (uint32_t)idx, drop_thresh, drop_inv, mk);
                dropout_scale4(drop_seed, (uint32_t)idx + 4u, drop_thresh, drop_inv, mk + 4);
#pragma unroll
                for (int j = 0; j < 8; ++j) f[j] *= mk[j];
            }
#pragma unroll
            for (int n = 0; n < MAXC; ++n) {
                if (n < ncls) {
                    const float* wr = sw + n * C + (c0 + c8) * 8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[n] += f[j] * wr[j];
                }
            }
        }
    }
    if (m >= M) return;
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
        if (n < ncls) acc[n] += sw[ncls * C + n];
    long b, pix;
    split_pixel(m, HW, b, pix);
#pragma unroll
    for (int n = 0; n < MAXC; ++n)
        if (n < ncls) logits[(b * ncls + n) * HW + pix] = acc[n];
}

// Training-mode BatchNorm2d + ReLU in front of the classifier, fused into its kernels (MODE template argument of the backward kernels):
//   MODE 0  plain classifier backward (f = the classifier's input, df = its gradient)
//   MODE 1  f = x, the last Conv2d's output: a = relu(x*scale + shift) is recomputed, dW / db are accumulated from it, and the
//           BatchNorm backward's two per-channel sums (dyr, dyr * xhat; dyr = d a * [a > 0]) leave as per-workgroup partials (or double
//           atomics) -- nothing of pixel size is written
//   MODE 2  the same recomputation, writes dx = scale*(dyr - k1 - xhat*k2) (k = sums / n) to df; no dW / db
// so the activation between BatchNorm and the classifier and its gradient never exist in HBM (7 passes over the largest tensor of the
// head instead of 11 with the separate BatchNorm kernels).
struct ClsBn {
    const float *scale, *shift, *mean, *rstd;
    double* sums;   // [2C]: MODE 1 adds to it when part == NULL, MODE 2 reads it
    float* part;    // MODE 1, deterministic mode: [workgroups][2C] partials (bn_part_fold_kernel sums them in index order)
    float *dgamma, *dbeta;
    double inv_n;
};

// df[m][c] = drop_mask * sum_n dl[b][n][pix] * w[n][c] * gscale ; dW[n][c] += sum_m dl*f_dropped ; db[n] += sum_m dl
// gscale = 1/(*count) when count != NULL (fused trainer: dlogits left un-normalised by the loss kernel).
//
// These kernels look HBM-bound (one pass over the head's largest tensor) but are VALU-issue-bound unless the per-pixel overhead is
// kept out of the loop (round 4: the previous pair of kernels spent ~40 instructions per element on 64-bit index arithmetic, pixel
// -> (image, offset) divisions and per-class branches, and ran at 1.1-3.3 TB/s).  Structure:
//   * a thread owns ONE unit of VEC channels (8, or 4 when 16 classes would not leave room for the NC x VEC weights AND the NC x VEC
//     dW accumulators in registers) and walks pixels with stride nsl = 256 / (C / VEC): loads stay fully coalesced, the class weights,
//     the BatchNorm constants and the dW partials never leave registers, the element index advances by one add per pixel;
//   * the dlogits of IG iterations (IG * nsl consecutive pixels) are staged through LDS with coalesced loads, pre-scaled and padded to
//     NC classes (the C / VEC threads of a pixel would each have loaded all of them, from NCHW planes); the (image, offset) split is one
//     scalar division per group;
//   * NC is a compile-time class bucket: the class loops are straight-line packed FMAs (padded classes multiply zeros);
//   * at the end the dW partials of the pixel slices are folded through a 32 KiB slab, G classes per round, then leave as one
//     order-independent add per element and workgroup (ig_red_add); db through 2^44 fixed point (NaN / Inf / huge bypass it).
constexpr int CLS_SLAB = 8192;  // floats
template <int VEC>
__device__ __forceinline__ void cls_load(const bf16_t* hi, const bf16_t* lo, size_t idx, float* f) {
    if constexpr (VEC == 8) {
        load8_split(hi, lo, idx, f);
    } else {
        const uint2 u = *reinterpret_cast<const uint2*>(hi + idx);
        f[0] = __uint_as_float(u.x << 16), f[1] = __uint_as_float(u.x & 0xffff0000u);
        f[2] = __uint_as_float(u.y << 16), f[3] = __uint_as_float(u.y & 0xffff0000u);
        if (lo) {
            const uint2 v = *reinterpret_cast<const uint2*>(lo + idx);
            f[0] += __uint_as_float(v.x << 16), f[1] += __uint_as_float(v.x & 0xffff0000u);
            f[2] += __uint_as_float(v.y << 16), f[3] += __uint_as_float(v.y & 0xffff0000u);
        }
    }
}
template <int VEC>
__device__ __forceinline__ void cls_store(bf16_t* hi, bf16_t* lo, size_t idx, const float* f) {
    if constexpr (VEC == 8) store8_split(hi, lo, idx, f);
    else store4_split(hi, lo, idx, f);
}

// MODE: 0 plain (df + dW + db)                 1 BatchNorm reduce pass (dW + db + BatchNorm sums; ClsBn above)
//       2 BatchNorm apply pass (dx only)         3 plain, df only            4 plain, dW + db only
//       5 BatchNorm reduce pass in "S form" (below)
// Modes 4 / 5 take NC classes per workgroup, blockIdx.y selects the class group (more than 8 classes: NC x VEC weights AND NC x VEC
// accumulators do not fit the registers at a useful occupancy; a 16-class instance ran at one wave per SIMD, 0.6 TB/s).
// S form: with pm = dropout mask * [a > 0] the per-class sums S1[n][c] = sum_m g[m][n] pm[m][c] and S2[n][c] = sum_m g[m][n] pm[m][c]
// (x[m][c] - mu[c]) give everything the reduce pass owes WITHOUT forming d a = sum_n g w (needs all classes in one thread):
//   dW[n][c] = sum_m g a msk = scale[c] S2[n][c] + beta[c] S1[n][c]          (a = scale (x - mu) + beta where a > 0)
//   sum_m dyr[c] = sum_n w[n][c] S1[n][c],   sum_m dyr x_hat[c] = rstd[c] sum_n w[n][c] S2[n][c]
// so each class group adds its part to dW, db and the BatchNorm partials independently.
template <int NC, int VEC, int MODE>
__global__ __launch_bounds__(TPB, VEC == 8 ? 1 : 2) void cls_bwd_kernel(const float* __restrict__ dl, const bf16_t* __restrict__ f_hi,
                                                         const bf16_t* __restrict__ f_lo, const float* __restrict__ w,
                                                         bf16_t* __restrict__ df_hi, bf16_t* __restrict__ df_lo, float* __restrict__ dw,
                                                         float* __restrict__ db, const double* count, long M, long HW, int C, int ncls,
                                                         uint32_t drop_seed, const uint32_t* drop_seed_dev, uint32_t drop_thresh,
                                                         float drop_inv, int iters, int ig, ClsBn bn) {
    constexpr bool HAS_BN = MODE == 1 || MODE == 2 || MODE == 5;
    constexpr bool WANT_O = MODE <= 3;                            // d(classifier input) is formed
    constexpr bool WANT_DW = MODE == 0 || MODE == 1 || MODE >= 4; // dW / db are accumulated
    constexpr bool SFORM = MODE == 5;
    // sdbq[16] (64-bit fixed point) | slab[CLS_SLAB] | sdl[NC][ig * nsl]
    extern __shared__ __attribute__((aligned(16))) float sm[];
    unsigned long long* sdbq = reinterpret_cast<unsigned long long*>(sm);
    float* slab = sm + 32;
    float* sdl = slab + CLS_SLAB;
    if (drop_seed_dev) drop_seed += *drop_seed_dev;
    const int nu = C / VEC, nsl = TPB / nu;
    const int u = threadIdx.x % nu, sl = threadIdx.x / nu;
    const bool live = sl < nsl;
    const int c0 = u * VEC;
    const int n0 = blockIdx.y * NC;          // first class of this workgroup's group (modes 4 / 5; otherwise 0)
    const int ncl = min(NC, ncls - n0);      // classes of the group
    if (threadIdx.x < 16) sdbq[threadIdx.x] = 0ull;
    const float gscale = count ? (float)(1.0 / fmax(count[1], 1.0)) : 1.f;
    float wr[WANT_O ? NC : 1][VEC], dwa[WANT_DW ? NC : 1][VEC], s1[SFORM ? NC : 1][VEC];
#pragma unroll
    for (int n = 0; n < NC; ++n) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if constexpr (WANT_O) wr[n][j] = n < ncl ? w[(size_t)(n0 + n) * C + c0 + j] : 0.f;
            if constexpr (WANT_DW) dwa[n][j] = 0.f;
            if constexpr (SFORM) s1[n][j] = 0.f;
        }
    }
    float bsc[VEC], bsh[VEC], bmu[VEC], brs[VEC], bs[VEC], bq[VEC];  // BatchNorm constants of this unit; MODE 2: bmu = ca, brs = cb
    if constexpr (HAS_BN) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = c0 + j;
            bsc[j] = bn.scale[c], bsh[j] = bn.shift[c], bmu[j] = bn.mean[c], brs[j] = bn.rstd[c], bs[j] = 0.f, bq[j] = 0.f;
            if constexpr (MODE == 2) {  // dx = sc*dyr - ca - cb*x  with  ca = sc*k1 - mu*cb,  cb = sc*rs*k2
                const float k1 = (float)(bn.sums[c] * bn.inv_n), k2 = (float)(bn.sums[C + c] * bn.inv_n);
                brs[j] = bsc[j] * brs[j] * k2;
                bmu[j] = bsc[j] * k1 - bmu[j] * brs[j];
            }
        }
        if (MODE == 2 && blockIdx.x == 0)
            for (int c = threadIdx.x; c < C; c += TPB) {
                if (bn.dbeta) atomicAdd(bn.dbeta + c, (float)bn.sums[c]);  // one contributor per element: order-independent
                if (bn.dgamma) atomicAdd(bn.dgamma + c, (float)bn.sums[C + c]);
            }
    }
    const long mb = (long)blockIdx.x * nsl * iters;
    const int gpx = ig * nsl;
    const size_t estep = (size_t)nsl * C;
    constexpr bool PIPE = VEC == 4 && !(MODE == 5 && NC == 16);
    constexpr int PF = PIPE ? 4 : 2;  // pixels per trip
    auto process = [&](float* f, const size_t idx, const int px) __attribute__((always_inline)) {
            float o[VEC], msk[VEC], x[VEC];
            if (drop_thresh) {
                dropout_scale4(drop_seed, (uint32_t)idx, drop_thresh, drop_inv, msk);
                if constexpr (VEC == 8) dropout_scale4(drop_seed, (uint32_t)idx + 4u, drop_thresh, drop_inv, msk + 4);
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) msk[j] = 1.f;
            }
            if constexpr (SFORM) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    x[j] = f[j];
                    msk[j] = fmaf(x[j], bsc[j], bsh[j]) > 0.f ? msk[j] : 0.f;  // pm
                    f[j] = msk[j] * (x[j] - bmu[j]);                         // pm (x - mu)
                }
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const float g = sdl[n * gpx + px];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) s1[n][j] = fmaf(g, msk[j], s1[n][j]), dwa[n][j] = fmaf(g, f[j], dwa[n][j]);
                }
            } else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    if constexpr (HAS_BN) x[j] = f[j], f[j] = fmaxf(fmaf(x[j], bsc[j], bsh[j]), 0.f);
                    f[j] *= msk[j];
                    o[j] = 0.f;
                }
#pragma unroll
                for (int n = 0; n < NC; ++n) {
                    const float g = sdl[n * gpx + px];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        if constexpr (WANT_O) o[j] = fmaf(g, wr[n][j], o[j]);
                        if constexpr (WANT_DW) dwa[n][j] = fmaf(g, f[j], dwa[n][j]);
                    }
                }
                if constexpr (WANT_O) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) o[j] *= msk[j];
                }
                if constexpr (MODE == 0 || MODE == 3) cls_store<VEC>(df_hi, df_lo, idx, o);
                if constexpr (MODE == 1) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float dyr = fmaf(x[j], bsc[j], bsh[j]) > 0.f ? o[j] : 0.f;
                        bs[j] += dyr;
                        bq[j] = fmaf(dyr, x[j] - bmu[j], bq[j]);  // x_hat = (x - mu) * rstd: the factor rstd is applied once, below
                    }
                }
                if constexpr (MODE == 2) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        const float dyr = fmaf(x[j], bsc[j], bsh[j]) > 0.f ? o[j] : 0.f;
                        o[j] = fmaf(bsc[j], dyr, -fmaf(brs[j], x[j], bmu[j]));
                    }
                    cls_store<VEC>(df_hi, df_lo, idx, o);
                }
            }

    };
    for (int it0 = 0; it0 < iters; it0 += ig) {
        const long mg = mb + (long)it0 * nsl;
        if (mg >= M) break;  // uniform
        const int itn = min(ig, iters - it0);
        const int npx = (int)min((long)itn * nsl, M - mg);
        size_t e = (size_t)(mg + sl) * C + c0;
        float fa[PF][VEC];
        if (PIPE && live && sl < npx) {  // first trip of the group: in flight across the dlogits staging below
            const int nv0 = min(itn, (npx - sl + nsl - 1) / nsl);
#pragma unroll
            for (int k = 0; k < PF; ++k) cls_load<VEC>(f_hi, f_lo, k < nv0 ? e + (size_t)k * estep : e, fa[k]);
        }
        __syncthreads();  // the previous group's dlogits have been consumed (first group: sdbq is zero)
        {
            const long b0 = mg / HW, pix0 = mg - b0 * HW;
#pragma unroll
            for (int n = 0; n < NC; ++n) {
                float part = 0.f;
                for (int px = threadIdx.x; px < gpx; px += TPB) {
                    float g = 0.f;
                    if (n < ncl && px < npx) {
                        long b = b0, pix = pix0 + px;
                        while (pix >= HW) pix -= HW, ++b;
                        g = dl[(b * ncls + n0 + n) * HW + pix] * gscale;
                    }
                    sdl[n * gpx + px] = g;
                    part += g;
                }
                if constexpr (WANT_DW) {  // db: the staged values are summed here (per wave, then 2^44 fixed point: any order, same bits)
                    if (n < ncl) {        // -- not per pixel and thread in the loop below (NC adds and NC registers less)
                        part = wave_sum(part);
                        if ((threadIdx.x & 63) == 0) {  // non-finite / out-of-range partials bypass the fixed-point slot (ig_red_add keeps them visible)
                            if (fabsf(part) < 5.0e5f) atomicAdd(sdbq + n, (unsigned long long)__float2ll_rn(part * 17592186044416.f));
                            else ig_red_add(db + n0 + n, part);
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (live) {
            const int nv = min(itn, (npx - sl + nsl - 1) / nsl);  // this thread's pixels in the group
            if constexpr (PIPE) {
                // trips of PF pixels; the loads of trip t + 1 are issued before trip t is computed: the 8-byte loads of the 4-channel form
                // left too little in flight (16-class df pass 709 -> 381 us); the 8-channel form at <= 4 classes is at the HBM rate with
                // two pixels per step and lost occupancy to the second buffer (462 -> 708 us)
                for (int t = 0; t < nv; t += PF) {
                    float fb[PF][VEC];
                    if (t + PF < nv) {
#pragma unroll
                        for (int k = 0; k < PF; ++k) cls_load<VEC>(f_hi, f_lo, t + PF + k < nv ? e + (size_t)(PF + k) * estep : e, fb[k]);
                    }
#pragma unroll
                    for (int k = 0; k < PF; ++k) {
                        if (t + k < nv) process(fa[k], e + (size_t)k * estep, (t + k) * nsl + sl);
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int k = 0; k < PF; ++k)
#pragma unroll
                        for (int j = 0; j < VEC; ++j) fa[k][j] = fb[k][j];
                    e += (size_t)PF * estep;
                }
            } else {
                for (int t = 0; t < nv; t += 2, e += 2 * estep) {  // two pixels in flight per thread
                    const bool two = t + 1 < nv;
                    float f[2][VEC];
                    cls_load<VEC>(f_hi, f_lo, e, f[0]);
                    cls_load<VEC>(f_hi, f_lo, two ? e + estep : e, f[1]);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        if (q == 1 && !two) break;
                        process(f[q], e + q * estep, (t + q) * nsl + sl);
                    }
                }
            }
        }
    }
    if constexpr (WANT_DW) {  // (nothing is reduced in the passes that only write df / dx)
        if constexpr (SFORM) {  // S1, S2 -> dW and this class group's part of the BatchNorm sums
#pragma unroll
            for (int n = 0; n < NC; ++n)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float wn = n < ncl ? w[(size_t)(n0 + n) * C + c0 + j] : 0.f;
                    bs[j] = fmaf(wn, s1[n][j], bs[j]);
                    bq[j] = fmaf(wn, dwa[n][j], bq[j]);
                    dwa[n][j] = fmaf(bsc[j], dwa[n][j], fmaf(bmu[j], bsc[j], bsh[j]) * s1[n][j]);  // beta = shift + mu * scale
                }
        }
        // ---- fold the dW partials of the pixel slices, G classes per round through the slab [nsl][G][C]
        const int G = max(1, min(NC, CLS_SLAB / (nsl * C)));
        for (int g0 = 0; g0 < ncl; g0 += G) {
            __syncthreads();
            if (live) {
#pragma unroll
                for (int n = 0; n < NC; ++n)
                    if (n >= g0 && n < g0 + G) {
#pragma unroll
                        for (int j = 0; j < VEC; ++j) slab[((size_t)sl * G + (n - g0)) * C + c0 + j] = dwa[n][j];
                    }
            }
            __syncthreads();
            const int nn = min(G, ncl - g0);
            for (int i = threadIdx.x; i < nn * C; i += TPB) {
                const int k = i / C, c = i - k * C;
                float t = 0.f;
                for (int s2 = 0; s2 < nsl; ++s2) t += slab[((size_t)s2 * G + k) * C + c];
                ig_red_add(dw + (size_t)(n0 + g0 + k) * C + c, t);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < ncl; i += TPB) ig_red_add(db + n0 + i, (float)((double)(long long)sdbq[i] * 5.684341886080802e-14));
        if constexpr (MODE == 1 || SFORM) {  // BatchNorm sums of this workgroup: slab per pixel slice [nsl][2][C], folded in slice order
            if (live) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) slab[(size_t)sl * 2 * C + c0 + j] = bs[j], slab[(size_t)sl * 2 * C + C + c0 + j] = bq[j] * brs[j];
            }
            __syncthreads();
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            for (int i = threadIdx.x; i < 2 * C; i += TPB) {
                float t = 0.f;
                for (int q = 0; q < nsl; ++q) t += slab[(size_t)q * 2 * C + i];
                if (bn.part) bn.part[wg * 2 * C + i] = t;
                else atomicAdd(bn.sums + i, (double)t);
            }
        }
    }
}

// stats[0] += sum w_y*nll over valid pixels ; stats[1] += #valid.  dlogits (optional) is left UN-normalised:
// dl[n] = w_y*(softmax_n - [n==y]) (0 on ignored pixels); divide by stats[1] downstream.
// preds (optional): int64 argmax (first maximal index, as torch.argmax); confusion (optional): int64 [k][k].
// NCB: class bucket (2: the two-class flood task -- straight-line class loops and the confusion counts in four registers per thread:
// the LDS histogram has only four counters there, and 256 threads x 4 pixels of same-address LDS atomics per iteration were the kernel;
// 16: everything else)
template <typename LABEL, int VEC, int NCB>
__global__ __launch_bounds__(TPB) void ce_loss_kernel(const float* __restrict__ logits, const LABEL* __restrict__ labels,
                                                      const float* __restrict__ cw, long ignore_index, double* __restrict__ stats,
                                                      float* __restrict__ dlogits, long long* __restrict__ preds,
                                                      signed char* __restrict__ preds_i8, unsigned long long* __restrict__ confusion,
                                                      long M, long HW, int ncls, unsigned long long* __restrict__ acc, unsigned* __restrict__ arrived) {
    extern __shared__ unsigned int hist[];  // [ncls*ncls]
    for (int i = threadIdx.x; i < ncls * ncls; i += TPB) hist[i] = 0u;
    __syncthreads();
    // grid-stride over pixels: ~1k workgroups end in one round of global atomics each (a workgroup per 256 pixels made
    // the 21k same-address double atomics the whole cost of the kernel).  VEC = 4 (HW % 4 == 0): a thread takes four
    // consecutive pixels of one image per iteration with 16-byte loads/stores -- the scalar loop had too few bytes in
    // flight per CU to cover the HBM latency.
    float my_loss = 0.f, my_cnt = 0.f;
    unsigned cnt4[4] = {0u, 0u, 0u, 0u};  // NCB == 2: confusion counts [y][argmax] of this thread
    for (long m0 = (blockIdx.x * (long)TPB + threadIdx.x) * VEC; m0 < M; m0 += (long)gridDim.x * TPB * VEC) {
        long b, pix;
        split_pixel(m0, HW, b, pix);
        float z[NCB][VEC];
        LABEL lab[VEC];
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
            if (n < ncls) {
                const float* src = logits + (b * ncls + n) * HW + pix;
                if constexpr (VEC == 4) {
                    const float4 v = *reinterpret_cast<const float4*>(src);
                    z[n][0] = v.x, z[n][1] = v.y, z[n][2] = v.z, z[n][3] = v.w;
                } else {
                    z[n][0] = *src;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) lab[e] = labels[m0 + e];
        int amv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float mx = -INFINITY;
            int am = 0;
#pragma unroll
            for (int n = 0; n < NCB; ++n)
                if (n < ncls && z[n][e] > mx) mx = z[n][e], am = n;
            float se = 0.f;
#pragma unroll
            for (int n = 0; n < NCB; ++n)
                if (n < ncls) se += __expf(z[n][e] - mx);
            const float lse = mx + __logf(se);
            const long y = (long)lab[e];
            const bool valid = (y != ignore_index) && y >= 0 && y < ncls;
            const float wy = valid ? (cw ? cw[y] : 1.f) : 0.f;
            if (valid) {
                float zy = 0.f;
#pragma unroll
                for (int n = 0; n < NCB; ++n)
                    if (n == (int)y) zy = z[n][e];
                my_loss += wy * (lse - zy);
                my_cnt += 1.f;
                if constexpr (NCB == 2) {
                    const int ci = (int)y * ncls + am;
#pragma unroll
                    for (int q = 0; q < 4; ++q) cnt4[q] += ci == q ? 1u : 0u;
                } else {
                    if (confusion) atomicAdd(hist + (int)y * ncls + am, 1u);
                }
            }
#pragma unroll
            for (int n = 0; n < NCB; ++n)
                if (n < ncls) z[n][e] = wy * (__expf(z[n][e] - lse) - (n == (int)y ? 1.f : 0.f));  // z now holds dlogits
            amv[e] = am;
        }
        if (dlogits) {
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                if (n < ncls) {
                    float* dst = dlogits + (b * ncls + n) * HW + pix;
                    if constexpr (VEC == 4)
                        *reinterpret_cast<float4*>(dst) = make_float4(z[n][0], z[n][1], z[n][2], z[n][3]);
                    else
                        *dst = z[n][0];
                }
            }
        }
        if (preds) {
            if constexpr (VEC == 4) {
                *reinterpret_cast<longlong2*>(preds + m0) = make_longlong2(amv[0], amv[1]);
                *reinterpret_cast<longlong2*>(preds + m0 + 2) = make_longlong2(amv[2], amv[3]);
            } else {
                preds[m0] = amv[0];
            }
        }
        if (preds_i8) {
            if constexpr (VEC == 4)
                *reinterpret_cast<char4*>(preds_i8 + m0) = make_char4((signed char)amv[0], (signed char)amv[1], (signed char)amv[2], (signed char)amv[3]);
            else
                preds_i8[m0] = (signed char)amv[0];
        }
    }
    // (loss, count) of the launch as INTEGER atomics (the loss partial of a workgroup in 2^28 fixed point, the count exactly), so
    // the reported loss is bit-identical from run to run whatever order the workgroups finish in; the workgroup that arrives last
    // converts the totals, adds them to stats and re-arms the scratch.  Waves are folded in index order.
    if constexpr (NCB == 2) {
        if (confusion) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                unsigned c = cnt4[q];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
                if ((threadIdx.x & 63) == 0 && q < ncls * ncls && c) atomicAdd(hist + q, c);
            }
        }
    }
    __shared__ float wred[TPB / 64][2];
    my_loss = wave_sum(my_loss);
    my_cnt = wave_sum(my_cnt);
    if ((threadIdx.x & 63) == 0) wred[threadIdx.x >> 6][0] = my_loss, wred[threadIdx.x >> 6][1] = my_cnt;
    __syncthreads();
    if (stats && threadIdx.x == 0) {
        float l = 0.f, c = 0.f;
        for (int w = 0; w < TPB / 64; ++w) l += wred[w][0], c += wred[w][1];
        // a NaN / Inf / absurdly large partial (diverged step) cannot go through the integer sum (NaN would convert to 0): it raises a
        // poison word instead and the reported loss becomes NaN / +-Inf, as the reference's (Lightning logs the NaN loss)
        unsigned* poison = reinterpret_cast<unsigned*>(acc + 3);
        unsigned long long r0 = 0ull;
        unsigned pz0 = 0u;
        if (fabsf(l) < 3.0e7f) r0 = atomicAdd(acc, (unsigned long long)__double2ll_rn((double)l * 268435456.0));
        else pz0 = atomicOr(poison, l != l ? 1u : (l > 0.f ? 2u : 4u));
        const unsigned long long r1 = atomicAdd(acc + 1, (unsigned long long)c);
        asm volatile("" ::"v"(r0), "v"(r1), "v"(pz0));  // the adds have been performed before this workgroup counts itself in
        if (atomicAdd(arrived, 1u) == gridDim.x - 1) {
            const long long tl = (long long)atomicExch(acc, 0ull);
            const unsigned long long tc = atomicExch(acc + 1, 0ull);
            const unsigned pz = atomicExch(poison, 0u);
            atomicExch(arrived, 0u);  // launches that share the scratch are ordered on one stream
            double tot = (double)tl * (1.0 / 268435456.0);
            if (pz) tot = ((pz & 1u) || (pz & 6u) == 6u) ? __longlong_as_double(0x7ff8000000000000LL) : ((pz & 2u) ? 1.0 : -1.0) * __longlong_as_double(0x7ff0000000000000LL);
            stats[0] += tot, stats[1] += (double)tc;
        }
    }
    if (confusion)
        for (int i = threadIdx.x; i < ncls * ncls; i += TPB)
            if (hist[i]) atomicAdd(confusion + i, (unsigned long long)hist[i]);
}

// Knowledge-distillation term (segmentation.py:352-378): KLDivLoss(batchmean)(log_softmax(student), softmax(teacher)) over
// the valid pixels: kl_sum += sum_valid sum_c t_c (log t_c - log s_c); its gradient w.r.t. the student logits,
// (s_c - t_c) per valid pixel (un-normalised: the shared 1 / #valid is applied downstream like the CE term), is ADDED to
// dlogits, which ig_ce_loss has filled with the cross-entropy part.
template <typename LABEL>
__global__ __launch_bounds__(TPB) void kd_loss_kernel(const float* __restrict__ student, const float* __restrict__ teacher,
                                                      const LABEL* __restrict__ labels, long ignore_index, double* __restrict__ kl_sum,
                                                      float* __restrict__ dlogits, long M, long HW, int ncls, double* __restrict__ scratch, unsigned* __restrict__ ticket) {
    __shared__ double red[TPB / 64];
    double my = 0.0;
    for (long m = blockIdx.x * (long)TPB + threadIdx.x; m < M; m += (long)gridDim.x * TPB) {
        {  // the validity predicate of ce_loss_kernel: ignored AND out-of-range labels carry neither loss nor gradient
            const long y = (long)labels[m];
            if (y == ignore_index || y < 0 || y >= ncls) continue;
        }
        long b, pix;
        split_pixel(m, HW, b, pix);
        float zs[MAXC], zt[MAXC];
        float ms = -INFINITY, mt = -INFINITY;
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) {
                zs[n] = student[(b * ncls + n) * HW + pix];
                zt[n] = teacher[(b * ncls + n) * HW + pix];
                ms = fmaxf(ms, zs[n]), mt = fmaxf(mt, zt[n]);
            }
        float ses = 0.f, set = 0.f;
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) ses += __expf(zs[n] - ms), set += __expf(zt[n] - mt);
        const float lses = ms + __logf(ses), lset = mt + __logf(set);
        float kl = 0.f;
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) {
                const float lt = zt[n] - lset, ls = zs[n] - lses;
                const float t = __expf(lt);
                kl += t * (lt - ls);
                if (dlogits) dlogits[(b * ncls + n) * HW + pix] += __expf(ls) - t;
            }
        my += (double)kl;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) my += __shfl_xor(my, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = my;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
    if (ordered_grid_totals<1>(t, scratch, ticket) && threadIdx.x == 0) *kl_sum += t;
}

// K16b streaming ROC-AUC histograms (metrics.py:214-256, called with softmax probabilities at segmentation.py:153-156):
// for every valid pixel and class c: bin = int((clamp(p_c, lo, hi) - lo) / (hi - lo) * (nbins - 1)) in float32 (numpy 2
// scalar arithmetic on float32 probabilities), hist[y == c ? 0 : 1][c][bin] += 1.  The reference does this in a Python
// loop per pixel; here one thread per pixel, histograms in LDS when they fit (2 * ncls * nbins * 4 B <= 64 KiB).
template <typename LABEL, bool USE_LDS>
__global__ __launch_bounds__(TPB) void auc_update_kernel(const float* __restrict__ logits, const LABEL* __restrict__ labels,
                                                         long ignore_index, unsigned long long* __restrict__ hist, long M, long HW,
                                                         int ncls, int nbins, float lo, float hi) {
    extern __shared__ unsigned int sh[];  // [2][ncls][nbins] when USE_LDS
    const int nh = 2 * ncls * nbins;
    if (USE_LDS) {
        for (int i = threadIdx.x; i < nh; i += TPB) sh[i] = 0u;
        __syncthreads();
    }
    const float inv = (float)(nbins - 1);
    for (long m = blockIdx.x * (long)TPB + threadIdx.x; m < M; m += (long)gridDim.x * TPB) {
        const long y = (long)labels[m];
        if (y == ignore_index) continue;
        long b, pix;
        split_pixel(m, HW, b, pix);
        float z[MAXC];
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) {
                z[n] = logits[(b * ncls + n) * HW + pix];
                mx = fmaxf(mx, z[n]);
            }
        float se = 0.f;
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) {
                z[n] = __expf(z[n] - mx);
                se += z[n];
            }
#pragma unroll
        for (int n = 0; n < MAXC; ++n)
            if (n < ncls) {
                float p = z[n] / se;
                p = fminf(hi, fmaxf(lo, p));
                const int bin = (int)((p - lo) / (hi - lo) * inv);
                const int idx = ((y == n ? 0 : 1) * ncls + n) * nbins + bin;
                if (USE_LDS) atomicAdd(sh + idx, 1u);
                else atomicAdd(hist + idx, 1ull);
            }
    }
    if (USE_LDS) {
        __syncthreads();
        for (int i = threadIdx.x; i < nh; i += TPB)
            if (sh[i]) atomicAdd(hist + i, (unsigned long long)sh[i]);
    }
}

// predict_step (segmentation.py:202-213): softmax(logits, dim=1)[:, cls]
__global__ __launch_bounds__(TPB) void softmax_prob_kernel(const float* __restrict__ logits, float* __restrict__ out, long M, long HW,
                                                           int ncls, int cls) {
    const long m = blockIdx.x * (long)TPB + threadIdx.x;
    if (m >= M) return;
    long b, pix;
    split_pixel(m, HW, b, pix);
    float mx = -INFINITY, zc = 0.f;
    for (int n = 0; n < ncls; ++n) mx = fmaxf(mx, logits[(b * ncls + n) * HW + pix]);
    float se = 0.f;
    for (int n = 0; n < ncls; ++n) {
        const float e = __expf(logits[(b * ncls + n) * HW + pix] - mx);
        se += e;
        if (n == cls) zc = e;
    }
    out[m] = zc / se;
}

// Regression head (regression.py:141-191): masked MSE on the single-channel output, optional log1p label scale,
// gradient of the mean squared error (left UN-normalised like the CE kernel: dpred = 2 (pred - label'), divide by stats[1]
// downstream) and the streaming sums of RunningRegressionMetrics (metrics.py:330-352) on the de-scaled values:
//   msums = { n, sum x, sum y, sum xy, sum x^2, sum y^2, sum |e|, sum e^2, #(|e| <= ee_bias + ee_coef x) }, x = label, y = prediction
__global__ __launch_bounds__(TPB) void mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ labels, float ignore_value,
                                                       int use_log, double* __restrict__ stats, float* __restrict__ dpred,
                                                       double* __restrict__ msums, float ee_bias, float ee_coef, int include_ee,
                                                       long M, double* __restrict__ scratch, unsigned* __restrict__ ticket) {
    __shared__ double red[TPB / 64][10];
    double a[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // [0] = sum of squared error in the (possibly log) training domain
    for (long m = blockIdx.x * (long)TPB + threadIdx.x; m < M; m += (long)gridDim.x * TPB) {
        const float lab = labels[m];
        const bool valid = lab != ignore_value;
        const float p = pred[m];
        const float t = use_log ? log1pf(lab) : lab;
        const float d = p - t;
        if (dpred) dpred[m] = valid ? 2.f * d : 0.f;
        if (valid) {
            a[0] += (double)(d * d);
            const float y = use_log ? expm1f(p) : p;   // de-scaled prediction
            const float x = use_log ? expm1f(t) : lab;  // the reference round-trips the label through the scaler too
            const float e = fabsf(y - x);
            a[1] += 1.0, a[2] += x, a[3] += y, a[4] += (double)(x * y), a[5] += (double)(x * x), a[6] += (double)(y * y);
            a[7] += e, a[8] += (double)(e * e);
            if (include_ee && e <= ee_bias + ee_coef * x) a[9] += 1.0;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        double v = a[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x < 10)
        for (int w = 0; w < TPB / 64; ++w) t += red[w][threadIdx.x];
    if (ordered_grid_totals<10>(t, scratch, ticket) && threadIdx.x < 10) {
        const int i = threadIdx.x;
        if (i == 0) stats[0] += t;
        else if (i == 1) {
            stats[1] += t;
            if (msums) msums[0] += t;
        } else if (msums) msums[i - 1] += t;
    }
}

// Distillation term of the regression task (regression.py:477-509, 522-534): mean over the VALID pixels (labels != ignore) of
// (student - teacher')^2 with teacher' = log1p(teacher) under use_log_scale; sum[0] += the numerator, dpred += 2 (s - t').
__global__ __launch_bounds__(TPB) void kd_mse_loss_kernel(const float* __restrict__ pred, const float* __restrict__ teacher,
                                                          const float* __restrict__ labels, float ignore_value, int use_log,
                                                          double* __restrict__ sum, float* __restrict__ dpred, long M, double* __restrict__ scratch, unsigned* __restrict__ ticket) {
    __shared__ double red[TPB / 64];
    double a = 0.0;
    for (long m = blockIdx.x * (long)TPB + threadIdx.x; m < M; m += (long)gridDim.x * TPB) {
        if (labels[m] == ignore_value) continue;
        const float t = use_log ? log1pf(teacher[m]) : teacher[m];
        const float d = pred[m] - t;
        a += (double)(d * d);
        if (dpred) dpred[m] += 2.f * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0)
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
    if (ordered_grid_totals<1>(t, scratch, ticket) && threadIdx.x == 0) *sum += t;
}

// argmax over classes -> int8 class map (infer_utils.py:99-101)
__global__ void argmax_kernel(const float* __restrict__ logits, signed char* __restrict__ out, long M, long HW, int ncls) {
    long m = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (m >= M) return;
    long b, pix;
    split_pixel(m, HW, b, pix);
    float mx = -INFINITY;
    int am = 0;
    for (int n = 0; n < ncls; ++n) {
        float z = logits[(b * ncls + n) * HW + pix];
        if (z > mx) mx = z, am = n;
    }
    out[m] = (signed char)am;
}

// confusion[y*k + p] += 1 over pixels with y != ignore  (metrics.py:86-108) for externally produced predictions
__global__ __launch_bounds__(TPB) void confusion_kernel(const long long* __restrict__ y_true, const long long* __restrict__ y_pred,
                                                        unsigned long long* __restrict__ confusion, long n, int k, long ignore_index,
                                                        int has_ignore) {
    extern __shared__ unsigned int hist[];
    for (int i = threadIdx.x; i < k * k; i += TPB) hist[i] = 0u;
    __syncthreads();
    for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        long long t = y_true[i], p = y_pred[i];
        if (has_ignore && t == ignore_index) continue;
        if (t < 0 || t >= k || p < 0 || p >= k) continue;
        atomicAdd(hist + (int)t * k + (int)p, 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < k * k; i += TPB)
        if (hist[i]) atomicAdd(confusion + i, (unsigned long long)hist[i]);
}

inline uint32_t thresh_of(float p) { return ig_drop_thresh16(p); }

}  // namespace
IG_DET_TU(head)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

extern "C" {

static int classifier_fwd_impl(const void* f_hi, const void* f_lo, const float* bn_scale, const float* bn_shift, const float* w,
                               const float* bias, float* logits, int B, long HW, int C, int ncls, unsigned drop_seed,
                               const unsigned* drop_seed_dev, float drop_p, void* stream) {
    IG_REQUIRE(f_hi && w && bias && logits, "ig_classifier_fwd: null pointer");
    IG_REQUIRE(C % 8 == 0 && ncls >= 1 && ncls <= MAXC, "ig_classifier_fwd: need C %% 8 == 0 and 1 <= ncls <= %d (C=%d ncls=%d)", MAXC, C, ncls);
    long M = (long)B * HW;
    if (M == 0) return IG_OK;
    const int cmax = C / 8 < CLS_CHUNK / 8 ? C / 8 : CLS_CHUNK / 8;
    size_t sm = ((size_t)ncls * C + ((ncls + 3) & ~3) + (bn_scale ? 2 * (size_t)C : 0)) * sizeof(float) + (size_t)TPB * (cmax * 16 + 16) * (f_lo ? 2 : 1);
    IG_REQUIRE(sm <= 160 * 1024, "ig_classifier_fwd: ncls x C = %d x %d weights do not fit the LDS", ncls, C);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)classifier_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    hipLaunchKernelGGL(classifier_fwd_kernel, dim3((unsigned)((M + TPB - 1) / TPB)), dim3(TPB), sm, (hipStream_t)stream,
                       (const bf16_t*)f_hi, (const bf16_t*)f_lo, w, bias, logits, M, HW, C, ncls, drop_seed, drop_seed_dev, thresh_of(drop_p),
                       drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f, bn_scale, bn_shift);
    return ig_check_launch("ig_classifier_fwd");
}

int ig_classifier_fwd(const void* f_hi, const void* f_lo, const float* w, const float* bias, float* logits, int B, long HW, int C,
                      int ncls, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream) {
    return classifier_fwd_impl(f_hi, f_lo, nullptr, nullptr, w, bias, logits, B, HW, C, ncls, drop_seed, drop_seed_dev, drop_p, stream);
}

// The training-mode tail of the head in one pass: logits = bias + w . drop(relu(x*scale + shift)), x = the last Conv2d's output,
// scale / shift = the batch statistics' affine (ig_bn_relu_fwd with y == NULL computes them without an apply pass).
int ig_classifier_bn_fwd(const void* x_hi, const void* x_lo, const float* scale, const float* shift, const float* w, const float* bias,
                         float* logits, int B, long HW, int C, int ncls, unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p,
                         void* stream) {
    IG_REQUIRE(scale && shift, "ig_classifier_bn_fwd: null pointer");
    return classifier_fwd_impl(x_hi, x_lo, scale, shift, w, bias, logits, B, HW, C, ncls, drop_seed, drop_seed_dev, drop_p, stream);
}

// count: NULL, or the loss kernel's stats buffer (uses stats[1] = #valid pixels to normalise dlogits)
// mode 0: plain; 1 / 2: the reduce / apply passes of the fused BatchNorm + classifier backward (ClsBn)
static int classifier_bwd_impl(int mode, ClsBn bn, int* nwg_out, const float* dlogits, const void* f_hi, const void* f_lo, const float* w,
                               void* df_hi, void* df_lo, float* dw, float* db, const double* count, int B, long HW, int C, int ncls,
                               unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream) {
    IG_REQUIRE(dlogits && f_hi && w && (df_hi || mode == 1) && ((dw && db) || mode == 2), "ig_classifier_bwd: null pointer");
    IG_REQUIRE(C % 8 == 0 && ncls >= 1 && ncls <= MAXC, "ig_classifier_bwd: need C %% 8 == 0 and 1 <= ncls <= %d", MAXC);
    long M = (long)B * HW;
    if (M == 0) return IG_OK;
    const int nc = ncls <= 2 ? 2 : ncls <= 4 ? 4 : ncls <= 8 ? 8 : 16;  // class bucket (padded classes multiply zeros)
    const bool wide = ncls > 4;          // dW passes split the classes into groups of 8 (blockIdx.y), 4 channels per thread
    const int vec = wide ? 4 : 8;
    IG_REQUIRE(C / vec <= TPB, "ig_classifier_bwd: C must be <= %d", TPB * vec);
    const long nsl = TPB / (C / vec);
    long iters = (M + nsl * 1024 - 1) / (nsl * 1024);  // ~1k workgroups: one round of dW / db adds each
    if (iters < 8) iters = 8;
    const long ppb = nsl * iters;  // pixels per workgroup
    const unsigned gx = (unsigned)((M + ppb - 1) / ppb);
    const unsigned gy_split = wide ? (unsigned)((ncls + 7) / 8) : 1u;  // class groups of the dW passes
    // > 8 classes: the dW passes take all 16 class slots in one group (default: 13 classes, S form: 722 against 1040 us for two groups of 8, each of
    // which repeats the loads, the dropout hash and the mask)
    constexpr bool one_group = true;
    if (nwg_out) {  // geometry query of the reduce pass (the caller sizes the partial-sum scratch before the launch)
        *nwg_out = (int)(gx * (nc == 16 && one_group ? 1u : gy_split));
        return IG_OK;
    }
    long ig = 4096 / (nc * nsl);  // iterations per staged dlogits group: ~16 KiB of LDS (at most 8 x 16 x 128 floats = 64 KiB)
    const long ig_max = wide ? 32 : 8;
    ig = ig < 8 ? 8 : ig > ig_max ? ig_max : ig & ~7L;  // whole trips of the prefetching loop (4 pixels)
    const size_t sm = (32 + (size_t)CLS_SLAB + (size_t)nc * ig * nsl) * sizeof(float);
    const float inv = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    constexpr bool sform = true;  // the reduce pass runs in S form at every class count (round 4: 111 -> 83 us at 2 classes)
    // narrow heads (C = 16 with > 8 classes, C = 8 with > 4) stage more than the default 64 KiB limit of dynamic LDS: 32 KiB slab + ig >= 8
    // groups of nc x nsl floats; every instantiation raises its limit once (as classifier_fwd_kernel does)
#define IG_CLS_BWD(NC, VEC, MD, GY)                                                                                                      \
    do {                                                                                                                                 \
        auto kern_ = cls_bwd_kernel<NC, VEC, MD>;                                                                                        \
        static bool attr_done_ = false;                                                                                                  \
        if (!attr_done_) {                                                                                                               \
            IG_REQUIRE(hipFuncSetAttribute((const void*)kern_, hipFuncAttributeMaxDynamicSharedMemorySize, 128 << 10) == hipSuccess,     \
                       "ig_classifier_bwd: could not raise the dynamic LDS limit%s", "");                                               \
            attr_done_ = true;                                                                                                           \
        }                                                                                                                                \
        hipLaunchKernelGGL(kern_, dim3(gx, GY), dim3(TPB), sm, (hipStream_t)stream, dlogits, (const bf16_t*)f_hi,                        \
                           (const bf16_t*)f_lo, w, (bf16_t*)df_hi, (bf16_t*)df_lo, dw, db, count, M, HW, C, ncls, drop_seed,             \
                           drop_seed_dev, thresh_of(drop_p), inv, (int)iters, (int)ig, bn);                                              \
    } while (0)
#define IG_CLS_BWD_M(NC)                             \
    do {                                             \
        if (mode == 0) IG_CLS_BWD(NC, 8, 0, 1);      \
        else if (mode == 1) {                        \
            if (sform) IG_CLS_BWD(NC, 8, 5, 1);      \
            else IG_CLS_BWD(NC, 8, 1, 1);            \
        }                                            \
        else IG_CLS_BWD(NC, 8, 2, 1);                \
    } while (0)
#define IG_CLS_BWD_W(NC)                                      \
    do {                                                      \
        if (mode == 0) {                                      \
            IG_CLS_BWD(NC, 4, 3, 1);                          \
            if (NC == 16 && one_group) IG_CLS_BWD(NC, 4, 4, 1); \
            else IG_CLS_BWD(8, 4, 4, gy_split);               \
        } else if (mode == 1) {                               \
            if (NC == 16 && one_group) IG_CLS_BWD(NC, 4, 5, 1); \
            else IG_CLS_BWD(8, 4, 5, gy_split);               \
        }                                                     \
        else IG_CLS_BWD(NC, 4, 2, 1);                         \
    } while (0)
    if (nc == 2) IG_CLS_BWD_M(2);
    else if (nc == 4) IG_CLS_BWD_M(4);
    else if (nc == 8) IG_CLS_BWD_W(8);
    else IG_CLS_BWD_W(16);
#undef IG_CLS_BWD_W
#undef IG_CLS_BWD_M
#undef IG_CLS_BWD
    return ig_check_launch("ig_classifier_bwd");
}

int ig_classifier_bwd(const float* dlogits, const void* f_hi, const void* f_lo, const float* w, void* df_hi, void* df_lo, float* dw,
                      float* db, const double* count, int B, long HW, int C, int ncls, unsigned drop_seed,
                      const unsigned* drop_seed_dev, float drop_p, void* stream) {
    return classifier_bwd_impl(0, ClsBn{}, nullptr, dlogits, f_hi, f_lo, w, df_hi, df_lo, dw, db, count, B, HW, C, ncls, drop_seed,
                               drop_seed_dev, drop_p, stream);
}

// Backward of the fused tail (ig_classifier_bn_fwd): x = the last Conv2d's output (saved), scale / shift / mean / rstd from the forward's
// statistics.  Pass 1 recomputes the activation, accumulates dW / db of the classifier and the BatchNorm backward's per-channel sums;
// pass 2 recomputes it again and writes dx; dgamma / dbeta are added once.  sums: device scratch double[2C].
int ig_classifier_bn_bwd(const float* dlogits, const void* x_hi, const void* x_lo, const float* scale, const float* shift,
                         const float* mean, const float* rstd, const float* w, void* dx_hi, void* dx_lo, float* dw, float* db,
                         float* dgamma, float* dbeta, double* sums, const double* count, int B, long HW, int C, int ncls,
                         unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream) {
    IG_REQUIRE(scale && shift && mean && rstd && sums && dx_hi, "ig_classifier_bn_bwd: null pointer");
    const long M = (long)B * HW;
    if (M == 0) return IG_OK;
    int nwg = 0;
    int rc = classifier_bwd_impl(1, ClsBn{}, &nwg, dlogits, x_hi, x_lo, w, nullptr, nullptr, dw, db, count, B, HW, C, ncls, drop_seed,
                                 drop_seed_dev, drop_p, stream);
    if (rc != IG_OK) return rc;
    ClsBn bn{scale, shift, mean, rstd, sums, nullptr, dgamma, dbeta, 1.0 / (double)M};
    if (ig_deterministic()) {
        bn.part = (float*)ig_scratch(0, (size_t)nwg * 2 * C * sizeof(float), (hipStream_t)stream);
        IG_REQUIRE(bn.part, "ig_classifier_bn_bwd: scratch allocation failed");
    } else {
        (void)hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(double), (hipStream_t)stream);
    }
    rc = classifier_bwd_impl(1, bn, nullptr, dlogits, x_hi, x_lo, w, nullptr, nullptr, dw, db, count, B, HW, C, ncls, drop_seed,
                             drop_seed_dev, drop_p, stream);
    if (rc != IG_OK) return rc;
    if (bn.part)
        hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * C, 64)), dim3(1024), 0, (hipStream_t)stream, bn.part, sums, nwg, 2 * C);
    return classifier_bwd_impl(2, bn, nullptr, dlogits, x_hi, x_lo, w, dx_hi, dx_lo, nullptr, nullptr, count, B, HW, C, ncls, drop_seed,
                               drop_seed_dev, drop_p, stream);
}

// scratch of ordered_grid_totals: 1024 workgroups x 16 partials + the ticket, per (device, stream) (runtime.hip: ig_scratch slot 2, zero-filled
// when it is made; its kernels re-arm the ticket).  Launches that share it are ordered on their stream; two streams never share it.  Not to be
// made during a graph capture (the first call on a stream is a warm-up call).
static double* loss_scratch(unsigned** ticket, hipStream_t st) {
    double* buf = (double*)ig_scratch(2, (1024 * 16 + 2) * sizeof(double), st);
    if (!buf) return nullptr;
    *ticket = reinterpret_cast<unsigned*>(buf + 1024 * 16);
    return buf;
}

// label_dtype: 0 = int64, 1 = int32, 2 = float32 (reference labels are float tensors cast with .long())
int ig_ce_loss(const float* logits, const void* labels, int label_dtype, const float* class_weights, long ignore_index,
               double* stats, float* dlogits, long long* preds, signed char* preds_i8, unsigned long long* confusion, int B,
               long HW, int ncls, void* stream) {
    IG_REQUIRE(logits && labels, "ig_ce_loss: null pointer");
    IG_REQUIRE(ncls >= 1 && ncls <= MAXC, "ig_ce_loss: 1 <= ncls <= %d (got %d)", MAXC, ncls);
    long M = (long)B * HW;
    if (M == 0) return IG_OK;
    size_t sm = (size_t)ncls * ncls * sizeof(unsigned);
    // scratch of the order-independent (loss, count) sums: two 64-bit accumulators + the arrival counter, per (device, stream) (ig_scratch
    // slot 3, zero-filled when it is made; the kernel re-arms it); never made during a graph capture (the first call on a stream is a warm-up)
    unsigned long long* acc_stream = nullptr;
    if (stats) {
        acc_stream = (unsigned long long*)ig_scratch(3, 4 * sizeof(unsigned long long), (hipStream_t)stream);
        if (!acc_stream) {
            ig_set_error("ig_ce_loss: scratch allocation failed");
            return IG_ERR_HIP;
        }
    }
    unsigned long long* acc = acc_stream;
    unsigned* arrived = acc ? reinterpret_cast<unsigned*>(acc + 2) : nullptr;
    const bool vec4 = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits) |
                                      reinterpret_cast<uintptr_t>(preds) | reinterpret_cast<uintptr_t>(preds_i8)) % 16 == 0;
    const int vec = vec4 ? 4 : 1;
    long nblk = (M + (long)TPB * vec - 1) / ((long)TPB * vec);
    if (nblk > 1024) nblk = 1024;  // grid-stride: one round of global atomics per workgroup
    dim3 grid((unsigned)nblk), block(TPB);
    hipStream_t st = (hipStream_t)stream;
#define IG_CE(LT)                                                                                                              \
    {                                                                                                                          \
        if (vec4 && ncls <= 2)                                                                                                 \
            hipLaunchKernelGGL((ce_loss_kernel<LT, 4, 2>), grid, block, sm, st, logits, (const LT*)labels, class_weights,      \
                               ignore_index, stats, dlogits, preds, preds_i8, confusion, M, HW, ncls, acc, arrived); \
        else if (vec4)                                                                                                         \
            hipLaunchKernelGGL((ce_loss_kernel<LT, 4, 16>), grid, block, sm, st, logits, (const LT*)labels, class_weights,     \
                               ignore_index, stats, dlogits, preds, preds_i8, confusion, M, HW, ncls, acc, arrived); \
        else                                                                                                                   \
            hipLaunchKernelGGL((ce_loss_kernel<LT, 1, 16>), grid, block, sm, st, logits, (const LT*)labels, class_weights,     \
                               ignore_index, stats, dlogits, preds, preds_i8, confusion, M, HW, ncls, acc, arrived); \
    }
    if (label_dtype == 0)
        IG_CE(long long)
    else if (label_dtype == 1)
        IG_CE(int)
    else if (label_dtype == 2)
        IG_CE(float)
    else {
        ig_set_error("ig_ce_loss: unsupported label dtype %d", label_dtype);
        return IG_ERR_UNSUPPORTED;
    }
#undef IG_CE
    return ig_check_launch("ig_ce_loss");
}

// kl_sum: device double += KL sum over valid pixels; dlogits (optional) += softmax(student) - softmax(teacher) on valid pixels
int ig_kd_loss(const float* student_logits, const float* teacher_logits, const void* labels, int label_dtype, long ignore_index,
               double* kl_sum, float* dlogits, int B, long HW, int ncls, void* stream) {
    IG_REQUIRE(student_logits && teacher_logits && labels && kl_sum, "ig_kd_loss: null pointer");
    IG_REQUIRE(ncls >= 1 && ncls <= MAXC, "ig_kd_loss: 1 <= ncls <= %d (got %d)", MAXC, ncls);
    const long M = (long)B * HW;
    if (M == 0) return IG_OK;
    long nblk = (M + TPB - 1) / TPB;
    if (nblk > 1024) nblk = 1024;
    hipStream_t st = (hipStream_t)stream;
    unsigned* ticket = nullptr;
    double* scratch = loss_scratch(&ticket, (hipStream_t)stream);
    IG_REQUIRE(scratch, "ig_kd_loss: scratch allocation failed");
    if (label_dtype == 0)
        hipLaunchKernelGGL(kd_loss_kernel<long long>, dim3((unsigned)nblk), dim3(TPB), 0, st, student_logits, teacher_logits,
                           (const long long*)labels, ignore_index, kl_sum, dlogits, M, HW, ncls, scratch, ticket);
    else if (label_dtype == 1)
        hipLaunchKernelGGL(kd_loss_kernel<int>, dim3((unsigned)nblk), dim3(TPB), 0, st, student_logits, teacher_logits, (const int*)labels,
                           ignore_index, kl_sum, dlogits, M, HW, ncls, scratch, ticket);
    else if (label_dtype == 2)
        hipLaunchKernelGGL(kd_loss_kernel<float>, dim3((unsigned)nblk), dim3(TPB), 0, st, student_logits, teacher_logits,
                           (const float*)labels, ignore_index, kl_sum, dlogits, M, HW, ncls, scratch, ticket);
    else {
        ig_set_error("ig_kd_loss: unsupported label dtype %d", label_dtype);
        return IG_ERR_UNSUPPORTED;
    }
    return ig_check_launch("ig_kd_loss");
}

// hist: device uint64 [2][ncls][nbins] (0 = positives of class c, 1 = negatives); label_dtype as ig_ce_loss
int ig_auc_update(const float* logits, const void* labels, int label_dtype, long ignore_index, unsigned long long* hist, int B,
                  long HW, int ncls, int nbins, float min_score, float max_score, void* stream) {
    IG_REQUIRE(logits && labels && hist, "ig_auc_update: null pointer");
    IG_REQUIRE(ncls >= 1 && ncls <= MAXC, "ig_auc_update: 1 <= ncls <= %d (got %d)", MAXC, ncls);
    IG_REQUIRE(nbins >= 2 && max_score > min_score, "ig_auc_update: need nbins >= 2 and max_score > min_score");
    const long M = (long)B * HW;
    if (M == 0) return IG_OK;
    long nblk = (M + TPB - 1) / TPB;
    if (nblk > 1024) nblk = 1024;
    const size_t lds = 2 * (size_t)ncls * nbins * sizeof(unsigned int);
    const bool use_lds = lds <= 65536;
    hipStream_t st = (hipStream_t)stream;
#define IG_AUC(LT)                                                                                                              \
    {                                                                                                                            \
        if (use_lds)                                                                                                             \
            hipLaunchKernelGGL((auc_update_kernel<LT, true>), dim3((unsigned)nblk), dim3(TPB), lds, st, logits, (const LT*)labels, \
                               ignore_index, hist, M, HW, ncls, nbins, min_score, max_score);                                     \
        else                                                                                                                     \
            hipLaunchKernelGGL((auc_update_kernel<LT, false>), dim3((unsigned)nblk), dim3(TPB), 0, st, logits, (const LT*)labels, \
                               ignore_index, hist, M, HW, ncls, nbins, min_score, max_score);                                     \
    }
    if (label_dtype == 0) IG_AUC(long long)
    else if (label_dtype == 1) IG_AUC(int)
    else if (label_dtype == 2) IG_AUC(float)
    else {
        ig_set_error("ig_auc_update: unsupported label dtype %d", label_dtype);
        return IG_ERR_UNSUPPORTED;
    }
#undef IG_AUC
    return ig_check_launch("ig_auc_update");
}

int ig_softmax_prob(const float* logits, float* out, int B, long HW, int ncls, int cls, void* stream) {
    IG_REQUIRE(logits && out, "ig_softmax_prob: null pointer");
    IG_REQUIRE(ncls >= 1 && cls >= 0 && cls < ncls, "ig_softmax_prob: need 0 <= cls < ncls");
    const long M = (long)B * HW;
    if (M == 0) return IG_OK;
    hipLaunchKernelGGL(softmax_prob_kernel, dim3((unsigned)((M + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, logits, out, M, HW,
                       ncls, cls);
    return ig_check_launch("ig_softmax_prob");
}

// stats: double[2] += (sum of squared error, #valid); msums: optional double[9] (see the kernel)
int ig_mse_loss(const float* pred, const float* labels, float ignore_value, int use_log_scale, double* stats, float* dpred,
                double* msums, float ee_bias, float ee_coef, int include_ee, long n, void* stream) {
    IG_REQUIRE(pred && labels && stats, "ig_mse_loss: null pointer");
    if (n == 0) return IG_OK;
    long nblk = (n + TPB - 1) / TPB;
    if (nblk > 1024) nblk = 1024;
    unsigned* ticket = nullptr;
    double* scratch = loss_scratch(&ticket, (hipStream_t)stream);
    IG_REQUIRE(scratch, "ig_mse_loss: scratch allocation failed");
    hipLaunchKernelGGL(mse_loss_kernel, dim3((unsigned)nblk), dim3(TPB), 0, (hipStream_t)stream, pred, labels, ignore_value, use_log_scale,
                       stats, dpred, msums, ee_bias, ee_coef, include_ee, n, scratch, ticket);
    return ig_check_launch("ig_mse_loss");
}

int ig_kd_mse_loss(const float* pred, const float* teacher, const float* labels, float ignore_value, int use_log_scale, double* sum,
                   float* dpred, long n, void* stream) {
    IG_REQUIRE(pred && teacher && labels && sum, "ig_kd_mse_loss: null pointer");
    if (n == 0) return IG_OK;
    long nblk = (n + TPB - 1) / TPB;
    if (nblk > 1024) nblk = 1024;
    unsigned* ticket = nullptr;
    double* scratch = loss_scratch(&ticket, (hipStream_t)stream);
    IG_REQUIRE(scratch, "ig_kd_mse_loss: scratch allocation failed");
    hipLaunchKernelGGL(kd_mse_loss_kernel, dim3((unsigned)nblk), dim3(TPB), 0, (hipStream_t)stream, pred, teacher, labels, ignore_value,
                       use_log_scale, sum, dpred, n, scratch, ticket);
    return ig_check_launch("ig_kd_mse_loss");
}

int ig_argmax_i8(const float* logits, signed char* out, int B, long HW, int ncls, void* stream) {
    IG_REQUIRE(logits && out, "ig_argmax_i8: null pointer");
    IG_REQUIRE(ncls >= 1 && ncls <= 127, "ig_argmax_i8: 1 <= ncls <= 127");
    long M = (long)B * HW;
    if (M == 0) return IG_OK;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((M + TPB - 1) / TPB)), dim3(TPB), 0, (hipStream_t)stream, logits, out, M, HW,
                       ncls);
    return ig_check_launch("ig_argmax_i8");
}

int ig_confusion_update(const long long* y_true, const long long* y_pred, unsigned long long* confusion, long n, int k,
                        long ignore_index, int has_ignore, void* stream) {
    IG_REQUIRE(y_true && y_pred && confusion, "ig_confusion_update: null pointer");
    IG_REQUIRE(k >= 1 && k <= 64, "ig_confusion_update: 1 <= k <= 64");
    if (n == 0) return IG_OK;
    long g = (n + TPB - 1) / TPB;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(confusion_kernel, dim3((unsigned)g), dim3(TPB), (size_t)k * k * sizeof(unsigned), (hipStream_t)stream, y_true,
                       y_pred, confusion, n, k, ignore_index, has_ignore);
    return ig_check_launch("ig_confusion_update");
}

}  // extern "C"
