// HBM-bound kernels of the Prithvi segmentation path (gfx950): chip normalisation, patch gather,
// LayerNorm fwd/bwd, BatchNorm(+ReLU) fwd/bwd, column sums, bf16 splitting, AdamW.
// All are streaming kernels: 16-byte accesses per lane, wave64 shuffles for row reductions, one
// atomic per (block, column) for column reductions.  Reference call sites are cited per kernel.
#include <algorithm>

#include "common.h"
#include "bn_fold.h"

namespace {

constexpr int TPB = 256;

inline int grid_for(long n, int per_block, int cap = 1 << 20) {
    long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// ----------------------------------------------------------------------------------------------
// K0: chip normalisation + layout  (instageo/model/dataloader.py:495-524, process_data :707-750)
//   src (B, T*C, H, W) int16|f32, band index t*C+c  ->  dst (B, C, T, H, W) f32 = (src*mult - mean_c)/std_c
// ----------------------------------------------------------------------------------------------
template <typename SRC>
__global__ void normalize_kernel(const SRC* __restrict__ src, float* __restrict__ dst, const float* __restrict__ mean,
                                 const float* __restrict__ stdv, double mult, int use_mult, int T, int C, long HW, long total4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        long e = i * 4;  // linear index in dst (B,C,T,HW)
        long pix = e % HW;
        long r = e / HW;
        int t = (int)(r % T);
        r /= T;
        int c = (int)(r % C);
        long b = r / C;
        const SRC* s = src + ((b * T + t) * C + c) * HW + pix;
        float m = mean[c], sd = stdv[c];
        float4 o;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // reference: data * constant_multiplier in float64 (dataloader.py:737), then float32
            float x = use_mult ? (float)((double)s[j] * mult) : (float)s[j];
            v[j] = (x - m) / sd;
        }
        o = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(dst + e) = o;
    }
}

// ----------------------------------------------------------------------------------------------
// K0 with the training-time data movement fused in: RandomCrop(im) + hflip/vflip + Normalize in one pass
// (dataloader.py:58-77 crop_image_and_label, :80-141 RandomHorizontal/VerticalFlip, :495-524 normalise).
//   src (B, T*C, Hs, Ws) int16|f32 ; params[b] = {top, left, hflip, vflip}
//   dst[b][c][t][y][x] = (src[b][t*C+c][top + y'][left + x'] * mult - mean_c) / std_c,
//   y' = vflip ? im-1-y : y, x' = hflip ? im-1-x : x   (crop first, then the flips, as the reference order)
//   labels (optional, f32 (B, Hs, Ws)) get the same crop + flips -> (B, im, im)
// One thread = 4 consecutive OUTPUT pixels of a row (16-byte store; reads are contiguous, reversed when hflip).
// ----------------------------------------------------------------------------------------------
template <typename SRC>
__global__ void crop_flip_normalize_kernel(const SRC* __restrict__ src, float* __restrict__ dst, const float* __restrict__ mean,
                                           const float* __restrict__ stdv, double mult, int use_mult, const int* __restrict__ params,
                                           const float* __restrict__ lab_in, float* __restrict__ lab_out, int T, int C, int Hs,
                                           int Ws, int im, long total4, long img4, long chip_stride, long lab_stride, int pw) {
    const int q = im / 4;  // quads per output row
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const bool is_lab = i >= img4;
        long r = is_lab ? i - img4 : i;
        const int xq = (int)(r % q);
        r /= q;
        const int y = (int)(r % im);
        r /= im;
        int t = 0, c = 0;
        long b;
        if (!is_lab) {
            t = (int)(r % T);
            r /= T;
            c = (int)(r % C);
            b = r / C;
        } else {
            b = r;
        }
        // pw = 4: {top, left, hflip, vflip} per chip (training crops); pw = 2: {top, left} window origins into ONE shared tile
        // (chip_stride = lab_stride = 0: sliding-window inference, dataloader.py:655-664)
        const int top = params[b * pw + 0], left = params[b * pw + 1], hf = pw == 4 ? params[b * 4 + 2] : 0, vf = pw == 4 ? params[b * 4 + 3] : 0;
        const int sy = top + (vf ? im - 1 - y : y);
        float v[4];
        if (!is_lab) {
            const SRC* row = src + b * chip_stride + ((long)t * C + c) * ((long)Hs * Ws) + (long)sy * Ws + left;
            const float m = mean[c], sd = stdv[c];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xq * 4 + j;
                const SRC sv = row[hf ? im - 1 - x : x];
                const float xv = use_mult ? (float)((double)sv * mult) : (float)sv;
                v[j] = (xv - m) / sd;
            }
            *reinterpret_cast<float4*>(dst + ((((long)b * C + c) * T + t) * im + y) * im + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            const float* row = lab_in + b * lab_stride + (long)sy * Ws + left;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xq * 4 + j;
                v[j] = row[hf ? im - 1 - x : x];
            }
            *reinterpret_cast<float4*>(lab_out + ((long)b * im + y) * im + xq * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// mode=stats (pipeline_utils.py:207-254): for every chip b and channel c the mean and the BIASED variance over
// (T, H, W); sums[c] += mean_bc, sums[C+c] += var_bc (the reference averages per-chip statistics over the chips:
// std = sqrt(mean_b var_bc), not the pooled standard deviation).  One workgroup per (b, c), two passes over the
// chip's T*H*W values (the second one hits L2), fp64 accumulation.
// ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void chip_stats_kernel(const float* __restrict__ x, double* __restrict__ sums, int C, long n) {
    __shared__ double red[TPB / 64];
    __shared__ double s_mean;
    const long bc = blockIdx.x;
    const int c = (int)(bc % C);
    const float* p = x + bc * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double acc = 0.0;
    for (long i = threadIdx.x * 4L; i < n; i += TPB * 4L) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        acc += (double)v.x + (double)v.y + (double)v.z + (double)v.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        s_mean = t / (double)n;
    }
    __syncthreads();
    const double mu = s_mean;
    acc = 0.0;
    for (long i = threadIdx.x * 4L; i < n; i += TPB * 4L) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        const double d0 = v.x - mu, d1 = v.y - mu, d2 = v.z - mu, d3 = v.w - mu;
        acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __syncthreads();
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < TPB / 64; ++w) t += red[w];
        atomicAdd(sums + c, mu);
        atomicAdd(sums + C + c, t / (double)n);
    }
}

// class counts of a float label map (np.unique(label, return_counts=True) of pipeline_utils.py:240-243):
// counts[v - lo] += 1 for integer-valued labels lo <= v < lo + nbins; anything else goes to counts[nbins]
__global__ __launch_bounds__(TPB) void label_hist_kernel(const float* __restrict__ lab, unsigned long long* __restrict__ counts,
                                                         long n, int lo, int nbins) {
    extern __shared__ unsigned int h[];  // nbins + 1
    for (int i = threadIdx.x; i <= nbins; i += TPB) h[i] = 0u;
    __syncthreads();
    for (long i = blockIdx.x * (long)TPB + threadIdx.x; i < n; i += (long)gridDim.x * TPB) {
        const float v = lab[i];
        const int k = (int)v - lo;
        const bool ok = (float)(int)v == v && k >= 0 && k < nbins;
        atomicAdd(h + (ok ? k : nbins), 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i <= nbins; i += TPB)
        if (h[i]) atomicAdd(counts + i, (unsigned long long)h[i]);
}

// ----------------------------------------------------------------------------------------------
// K1 gather: (B,C,T,H,W) f32 -> patches [B*T*gh*gw][C*p*p] bf16/split, k = c*p*p + iy*p + ix,
// token order (t, row, col)  (pritvhi.py:266-268 flatten(2).transpose(1,2) of Conv3d k=s=(1,p,p))
// ----------------------------------------------------------------------------------------------
__global__ void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                int C, int T, int H, int W, int p, int gh, int gw, long units) {
    const int Kp = C * p * p;
    const int upr = p / 8;  // 16-byte units per patch row
    for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        long e = u * 8;
        long row = e / Kp;
        int k = (int)(e - row * Kp);
        int c = k / (p * p);
        int rem = k - c * p * p;
        int iy = rem / p, ix = rem - iy * p;
        long tok = row;
        int px = (int)(tok % gw);
        tok /= gw;
        int py = (int)(tok % gh);
        tok /= gh;
        int t = (int)(tok % T);
        long b = tok / T;
        const float* s = img + (((b * C + c) * T + t) * H + (py * p + iy)) * (long)W + px * p + ix;
        float4 a = *reinterpret_cast<const float4*>(s);
        float4 bq = *reinterpret_cast<const float4*>(s + 4);
        float f[8] = {a.x, a.y, a.z, a.w, bq.x, bq.y, bq.z, bq.w};
        store8_split(out_hi, out_lo, (size_t)e, f);
        (void)upr;
    }
}

// any patch size with C p p a multiple of 8 (the 600M variants' patch 14: 8-element units straddle image rows): per-element decode
__global__ void patchify_generic_kernel(const float* __restrict__ img, bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                        int C, int T, int H, int W, int p, int gh, int gw, long units) {
    const int Kp = C * p * p;
    for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < units; u += (long)gridDim.x * blockDim.x) {
        const long e = u * 8;
        const long row = e / Kp;
        const int k0 = (int)(e - row * Kp);
        long tok = row;
        const int px = (int)(tok % gw);
        tok /= gw;
        const int py = (int)(tok % gh);
        tok /= gh;
        const int t = (int)(tok % T);
        const long b = tok / T;
        float f[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + j;
            const int c = k / (p * p), rem = k - c * p * p;
            const int iy = rem / p, ix = rem - iy * p;
            f[j] = img[(((b * C + c) * T + t) * H + (py * p + iy)) * (long)W + px * p + ix];
        }
        store8_split(out_hi, out_lo, (size_t)e, f);
    }
}

// cls rows: x[b][0][:] = cls + pos[0]   (pritvhi.py:520-522)
__global__ void cls_rows_kernel(float* __restrict__ x, const float* __restrict__ cls, const float* __restrict__ pos, int B,
                                long row_stride, int D) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * D) return;
    int d = (int)(i % D);
    long b = i / D;
    x[b * row_stride + d] = cls[d] + pos[d];
}

// ----------------------------------------------------------------------------------------------
// K3: LayerNorm forward (nn.LayerNorm eps=1e-5 inside timm Block and PrithviViT.norm, pritvhi.py:448-459,529)
// one wave per row; fp32 in; bf16/split out; saves mean and rstd.  feat_T>0 selects the K9 layout
// (model.py:406-413): token (b, 1+t*G+p) channel d -> out[(b*G+p)*(D*T) + d*T + t], cls row dropped.
// ----------------------------------------------------------------------------------------------
template <int MAXV>
__global__ __launch_bounds__(TPB) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, bf16_t* __restrict__ out_hi,
                                                            bf16_t* __restrict__ out_lo, float* __restrict__ mean_o,
                                                            float* __restrict__ rstd_o, int M, int D, float eps, int feat_T,
                                                            int feat_G, int ntok) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nv = D / 4;  // float4 chunks per row
    const float* xr = x + (size_t)row * D;
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + i * 64;
        if (c < nv) {
            v[i] = *reinterpret_cast<const float4*>(xr + c * 4);
            s += v[i].x + v[i].y + v[i].z + v[i].w;
        }
    }
    float mu = wave_sum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + i * 64;
        if (c < nv) {
            float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
            q += a * a + b * b + cc * cc + d * d;
        }
    }
    float rstd = rsqrtf(wave_sum(q) / D + eps);
    if (lane == 0) {
        if (mean_o) mean_o[row] = mu;
        if (rstd_o) rstd_o[row] = rstd;
    }
    long obase;
    int tfr = 0;
    if (feat_T > 0) {
        int b = row / ntok, tok = row - b * ntok;
        if (tok == 0) return;  // cls token dropped
        int tp = tok - 1;
        tfr = tp / feat_G;
        int p = tp - tfr * feat_G;
        obase = ((long)b * feat_G + p) * ((long)D * feat_T);
    } else {
        obase = (long)row * D;
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int c = lane + i * 64;
        if (c < nv) {
            float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
            float4 bb = *reinterpret_cast<const float4*>(beta + c * 4);
            float o[4] = {(v[i].x - mu) * rstd * g.x + bb.x, (v[i].y - mu) * rstd * g.y + bb.y,
                          (v[i].z - mu) * rstd * g.z + bb.z, (v[i].w - mu) * rstd * g.w + bb.w};
            if (feat_T <= 1) {
                store4_split(out_hi, out_lo, (size_t)obase + c * 4, o);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) store1_split(out_hi, out_lo, (size_t)obase + (long)(c * 4 + j) * feat_T + tfr, o[j]);
            }
        }
    }
}

// LayerNorm backward.  dy: bf16/split [M][D] (or the K9 feature layout when feat_T>0, cls rows = 0).
//   dx[row] = (accumulate ? dx[row] : 0) + rstd*(dy*g - mean(dy*g) - xhat*mean(dy*g*xhat))
//   dgamma += sum_rows dy*xhat ; dbeta += sum_rows dy ; optional: dxb = bf16(dx), dcol += sum_rows dx
// Each block owns ROWS_PER_BLOCK rows; each wave keeps its column partials in registers, then one
// LDS reduction + one fp32 atomic per (block, column).
constexpr int LNB_TPB = 256;  // measured: 4 waves x 32 rows per workgroup beats 16-row, 512- and 1024-thread variants
template <int MAXV>
__global__ __launch_bounds__(LNB_TPB) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy_hi, const bf16_t* __restrict__ dy_lo,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, int accumulate, bf16_t* __restrict__ dxb_hi,
                                                            bf16_t* __restrict__ dxb_lo, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dcol, int M, int D,
                                                            int rows_per_block, int feat_T, int feat_G, int ntok) {
    extern __shared__ float red[];  // [3][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nv = D / 4;
    float4 ag[MAXV], ab[MAXV], ac[MAXV];
#pragma unroll
    for (int i = 0; i < MAXV; ++i) ag[i] = ab[i] = ac[i] = make_float4(0, 0, 0, 0);
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(M, row0 + rows_per_block);
    for (int row = row0 + wave; row < row1; row += LNB_TPB / 64) {
        const float mu = mean[row], rs = rstd[row];
        bool zero_dy = false;
        long dbase = (long)row * D;
        int tfr = 0;
        if (feat_T > 0) {
            int b = row / ntok, tok = row - b * ntok;
            if (tok == 0) zero_dy = true;
            else {
                int tp = tok - 1;
                tfr = tp / feat_G;
                int p = tp - tfr * feat_G;
                dbase = ((long)b * feat_G + p) * ((long)D * feat_T);
            }
        }
        float4 xh[MAXV], dyv[MAXV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + i * 64;
            if (c < nv) {
                float4 xv = *reinterpret_cast<const float4*>(x + (size_t)row * D + c * 4);
                xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                if (!zero_dy) {
                    if (feat_T <= 1) {
                        const bf16_t* ph = dy_hi + dbase + c * 4;
                        uint2 u = *reinterpret_cast<const uint2*>(ph);
                        d[0] = __uint_as_float(u.x << 16), d[1] = __uint_as_float(u.x & 0xffff0000u);
                        d[2] = __uint_as_float(u.y << 16), d[3] = __uint_as_float(u.y & 0xffff0000u);
                        if (dy_lo) {
                            uint2 w = *reinterpret_cast<const uint2*>(dy_lo + dbase + c * 4);
                            d[0] += __uint_as_float(w.x << 16), d[1] += __uint_as_float(w.x & 0xffff0000u);
                            d[2] += __uint_as_float(w.y << 16), d[3] += __uint_as_float(w.y & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[j] = load1_split(dy_hi, dy_lo, (size_t)dbase + (long)(c * 4 + j) * feat_T + tfr);
                    }
                }
                dyv[i] = make_float4(d[0], d[1], d[2], d[3]);
                float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
                float a0 = d[0] * g.x, a1 = d[1] * g.y, a2 = d[2] * g.z, a3 = d[3] * g.w;
                s1 += a0 + a1 + a2 + a3;
                s2 += a0 * xh[i].x + a1 * xh[i].y + a2 * xh[i].z + a3 * xh[i].w;
            }
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int c = lane + i * 64;
            if (c < nv) {
                float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
                float4 r;
                r.x = rs * (dyv[i].x * g.x - s1 - xh[i].x * s2);
                r.y = rs * (dyv[i].y * g.y - s1 - xh[i].y * s2);
                r.z = rs * (dyv[i].z * g.z - s1 - xh[i].z * s2);
                r.w = rs * (dyv[i].w * g.w - s1 - xh[i].w * s2);
                float* dp = dx + (size_t)row * D + c * 4;
                if (accumulate) {
                    float4 o = *reinterpret_cast<const float4*>(dp);
                    r.x += o.x, r.y += o.y, r.z += o.z, r.w += o.w;
                }
                *reinterpret_cast<float4*>(dp) = r;
                if (dxb_hi) {
                    float f[4] = {r.x, r.y, r.z, r.w};
                    store4_split(dxb_hi, dxb_lo, (size_t)row * D + c * 4, f);
                }
                ag[i].x += dyv[i].x * xh[i].x, ag[i].y += dyv[i].y * xh[i].y, ag[i].z += dyv[i].z * xh[i].z, ag[i].w += dyv[i].w * xh[i].w;
                ab[i].x += dyv[i].x, ab[i].y += dyv[i].y, ab[i].z += dyv[i].z, ab[i].w += dyv[i].w;
                ac[i].x += r.x, ac[i].y += r.y, ac[i].z += r.z, ac[i].w += r.w;
            }
        }
    }
    // cross-wave column reduction: the waves add their partials into the LDS image one after the other (plain read-modify-write in
    // wave order, so the sum does not depend on which wave gets there first; this variant only serves the small / odd widths)
    for (int i = threadIdx.x; i < 3 * D; i += LNB_TPB) red[i] = 0.f;
    __syncthreads();
    for (int w = 0; w < LNB_TPB / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                int c = lane + i * 64;
                if (c < nv) {
                    float4* r0 = reinterpret_cast<float4*>(red + c * 4);
                    float4* r1 = reinterpret_cast<float4*>(red + D + c * 4);
                    float4* r2 = reinterpret_cast<float4*>(red + 2 * D + c * 4);
                    float4 t0 = *r0, t1 = *r1, t2 = *r2;
                    t0.x += ag[i].x, t0.y += ag[i].y, t0.z += ag[i].z, t0.w += ag[i].w;
                    t1.x += ab[i].x, t1.y += ab[i].y, t1.z += ab[i].z, t1.w += ab[i].w;
                    t2.x += ac[i].x, t2.y += ac[i].y, t2.z += ac[i].z, t2.w += ac[i].w;
                    *r0 = t0, *r1 = t1, *r2 = t2;
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < D; i += LNB_TPB) {
        if (dgamma) ig_red_add(dgamma + i, red[i]);
        if (dbeta) ig_red_add(dbeta + i, red[D + i]);
        if (dcol) ig_red_add(dcol + i, red[2 * D + i]);
    }
}

// exact variant: D == NCH*256, so every lane owns NCH full float4 chunks -- no per-chunk guards (a guard per load made
// hipcc branch around each load and wait vmcnt(0) 52 times per row; 180 VGPRs) and the feature layout is a template flag
template <int NCH, bool FEAT>
__global__ __launch_bounds__(LNB_TPB) void layernorm_bwd_exact_kernel(const bf16_t* __restrict__ dy_hi, const bf16_t* __restrict__ dy_lo,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            float* __restrict__ dx, int accumulate, bf16_t* __restrict__ dxb_hi,
                                                            bf16_t* __restrict__ dxb_lo, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ dcol, int M, int D,
                                                            int rows_per_block, int feat_T, int feat_G, int ntok) {
    extern __shared__ float red[];  // [3][D]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        float4 ag[NCH], ab[NCH], ac[NCH];
#pragma unroll
    for (int i = 0; i < NCH; ++i) ag[i] = ab[i] = ac[i] = make_float4(0, 0, 0, 0);
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(M, row0 + rows_per_block);
    // TWO rows per wave and iteration, every load of both rows (x, dy and -- when accumulating -- the old dx) issued before
    // the first reduction: one exposed memory latency per pair of rows instead of two per row
    constexpr int NW = LNB_TPB / 64;
    for (int rbase = row0 + wave; rbase < row1; rbase += 2 * NW) {
        float4 xh[2][NCH], dyv[2][NCH], old[2][NCH];
        float mu[2], rs[2];
        bool live[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int rowq = rbase + q * NW;
            live[q] = rowq < row1;
            const int row = live[q] ? rowq : rbase;  // dead second row: harmless duplicate loads, no stores
            mu[q] = mean[row], rs[q] = rstd[row];
            bool zero_dy = false;
            long dbase = (long)row * D;
            int tfr = 0;
            if (FEAT) {
                int b = row / ntok, tok = row - b * ntok;
                if (tok == 0) zero_dy = true;
                else {
                    int tp = tok - 1;
                    tfr = tp / feat_G;
                    int p = tp - tfr * feat_G;
                    dbase = ((long)b * feat_G + p) * ((long)D * feat_T);
                }
            }
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                {   // the saved forward input is read exactly once: streaming (non-temporal) load, it should not displace the gradients that
                    // the neighbouring kernels hand to each other through L2 / MALL
                    const f32x4 xv4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x + (size_t)row * D + c * 4));
                    xh[q][i] = make_float4(xv4[0], xv4[1], xv4[2], xv4[3]);
                }
                old[q][i] = accumulate ? *reinterpret_cast<const float4*>(dx + (size_t)row * D + c * 4) : make_float4(0, 0, 0, 0);
                float d[4] = {0.f, 0.f, 0.f, 0.f};
                if (!zero_dy) {
                    if (!FEAT || feat_T <= 1) {
                        uint2 u = *reinterpret_cast<const uint2*>(dy_hi + dbase + c * 4);
                        d[0] = __uint_as_float(u.x << 16), d[1] = __uint_as_float(u.x & 0xffff0000u);
                        d[2] = __uint_as_float(u.y << 16), d[3] = __uint_as_float(u.y & 0xffff0000u);
                        if (dy_lo) {
                            uint2 w = *reinterpret_cast<const uint2*>(dy_lo + dbase + c * 4);
                            d[0] += __uint_as_float(w.x << 16), d[1] += __uint_as_float(w.x & 0xffff0000u);
                            d[2] += __uint_as_float(w.y << 16), d[3] += __uint_as_float(w.y & 0xffff0000u);
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) d[j] = load1_split(dy_hi, dy_lo, (size_t)dbase + (long)(c * 4 + j) * feat_T + tfr);
                    }
                }
                dyv[q][i] = make_float4(d[0], d[1], d[2], d[3]);
            }
        }
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                const float4 xv = xh[q][i];
                xh[q][i] = make_float4((xv.x - mu[q]) * rs[q], (xv.y - mu[q]) * rs[q], (xv.z - mu[q]) * rs[q], (xv.w - mu[q]) * rs[q]);
                const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
                const float a0 = dyv[q][i].x * g.x, a1 = dyv[q][i].y * g.y, a2 = dyv[q][i].z * g.z, a3 = dyv[q][i].w * g.w;
                s1[q] += a0 + a1 + a2 + a3;
                s2[q] += a0 * xh[q][i].x + a1 * xh[q][i].y + a2 * xh[q][i].z + a3 * xh[q][i].w;
            }
#pragma unroll
        for (int q = 0; q < 2; ++q) s1[q] = wave_sum(s1[q]) / D, s2[q] = wave_sum(s2[q]) / D;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!live[q]) continue;  // wave-uniform
            const int row = rbase + q * NW;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const int c = lane + i * 64;
                const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
                float4 r;
                r.x = rs[q] * (dyv[q][i].x * g.x - s1[q] - xh[q][i].x * s2[q]) + old[q][i].x;
                r.y = rs[q] * (dyv[q][i].y * g.y - s1[q] - xh[q][i].y * s2[q]) + old[q][i].y;
                r.z = rs[q] * (dyv[q][i].z * g.z - s1[q] - xh[q][i].z * s2[q]) + old[q][i].z;
                r.w = rs[q] * (dyv[q][i].w * g.w - s1[q] - xh[q][i].w * s2[q]) + old[q][i].w;
                *reinterpret_cast<float4*>(dx + (size_t)row * D + c * 4) = r;
                if (dxb_hi) {
                    float f[4] = {r.x, r.y, r.z, r.w};
                    store4_split(dxb_hi, dxb_lo, (size_t)row * D + c * 4, f);
                }
                ag[i].x += dyv[q][i].x * xh[q][i].x, ag[i].y += dyv[q][i].y * xh[q][i].y, ag[i].z += dyv[q][i].z * xh[q][i].z,
                    ag[i].w += dyv[q][i].w * xh[q][i].w;
                ab[i].x += dyv[q][i].x, ab[i].y += dyv[q][i].y, ab[i].z += dyv[q][i].z, ab[i].w += dyv[q][i].w;
                ac[i].x += r.x, ac[i].y += r.y, ac[i].z += r.z, ac[i].w += r.w;
            }
        }
    }
    // cross-wave column reduction through LDS with PLAIN stores and loads: every wave writes its 3 x D partials to its own
    // slab, then each thread sums one column over the waves.  (36 ds_add_f32 per lane cost 25 us of a 68 us launch: LDS
    // float atomics retire a few lanes per cycle.)
    constexpr int NWV = LNB_TPB / 64;
    float* mine = red + (size_t)wave * 3 * D;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
        const int c = lane + i * 64;
        *reinterpret_cast<float4*>(mine + c * 4) = ag[i];
        *reinterpret_cast<float4*>(mine + D + c * 4) = ab[i];
        *reinterpret_cast<float4*>(mine + 2 * D + c * 4) = ac[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += LNB_TPB) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) t += red[(size_t)w * 3 * D + i];
        float* dst = i < D ? dgamma : i < 2 * D ? dbeta : dcol;
        if (dst) ig_red_add(dst + (i < D ? i : i < 2 * D ? i - D : i - 2 * D), t);
    }
}

// column sums of a bf16/split matrix: out[c] += sum_m x[m][c]   (bias gradients)
// grid (row chunk, 1024-column chunk); threads tile (row slice, 8-column unit) with 4 independent 16-byte loads in
// flight; per-block LDS reduction over the row slices (plain stores + loads), then one atomic per (block, column)
__global__ __launch_bounds__(TPB) void colsum_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo,
                                                     float* __restrict__ out, long M, int C, int rows_per_block, long ld) {
    __shared__ float red[TPB * 8];  // [row slice][columns of this block]: plain stores, no LDS float atomics (slow)
    const int c0 = blockIdx.y * 1024;
    const int cw = min(1024, C - c0);  // columns of this block (128 units: two row slices per 256 threads)
    const int units = cw / 8;
    const int tu = min(units, TPB), nslice = TPB / tu;
    const int u = threadIdx.x % tu, sl = threadIdx.x / tu;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    if (sl < nslice) {
        // units <= tu here (cw <= 1024 and tu = min(units, 256) with units <= 128), so one pass covers the block's columns
        const size_t col = (size_t)c0 + u * 8;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long r = r0 + sl;
        for (; r + 3L * nslice < r1; r += 4L * nslice) {
            float f0[8], f1[8], f2[8], f3[8];
            load8_split(hi, lo, (size_t)r * ld + col, f0);
            load8_split(hi, lo, (size_t)(r + nslice) * ld + col, f1);
            load8_split(hi, lo, (size_t)(r + 2L * nslice) * ld + col, f2);
            load8_split(hi, lo, (size_t)(r + 3L * nslice) * ld + col, f3);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (f0[j] + f1[j]) + (f2[j] + f3[j]);
        }
        for (; r < r1; r += nslice) {
            float f[8];
            load8_split(hi, lo, (size_t)r * ld + col, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += f[j];
        }
        float* dst = red + (size_t)sl * cw + u * 8;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cw; i += TPB) {
        float t = 0.f;
        for (int q = 0; q < nslice; ++q) t += red[(size_t)q * cw + i];
        ig_red_add(out + c0 + i, t);
    }
}

// f32 -> bf16 hi (+ lo)
__global__ void split_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        store1_split(hi, lo, (size_t)i, src[i]);
    }
}
// bf16 hi (+ lo) -> f32
__global__ void merge_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo, float* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        dst[i] = load1_split(hi, lo, (size_t)i);
    }
}

// patch-embed gradient prep: dx f32 [B][ntok][D] -> dpe bf16/split [B*(ntok-1)][D] (cls row removed),
// dcls[d] += sum_b dx[b][0][d], dbias[d] += sum over non-cls rows (cls_token / conv bias gradients)
__global__ __launch_bounds__(TPB) void patch_grad_prep_kernel(const float* __restrict__ dx, bf16_t* __restrict__ hi,
                                                              bf16_t* __restrict__ lo, float* __restrict__ dcls,
                                                              float* __restrict__ dbias, int B, int ntok, int D) {
    // one block per (chunk of 16 tokens, batch group: b = blockIdx.y, + gridDim.y, ...); threads over D in float4; four token rows in flight
    // per thread and ONE round of bias-gradient adds per workgroup (one block per (b, 32 tokens) with one load at a time was all latency
    // and 1.2 M same-address adds: 190 us for 130 MB)
    const int t0 = blockIdx.x * 16;
    const int t1 = min(ntok, t0 + 16);
    for (int c = threadIdx.x; c < D / 4; c += TPB) {
        float4 acc = make_float4(0, 0, 0, 0);
      for (int b = blockIdx.y; b < B; b += gridDim.y) {
        int t = t0;
        if (t == 0) {  // the cls token's gradient
            const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)b * ntok) * D + c * 4);
            ig_red_add(dcls + c * 4 + 0, v.x), ig_red_add(dcls + c * 4 + 1, v.y);
            ig_red_add(dcls + c * 4 + 2, v.z), ig_red_add(dcls + c * 4 + 3, v.w);
            t = 1;
        }
        for (; t + 3 < t1; t += 4) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const float4*>(dx + ((size_t)b * ntok + t + k) * D + c * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
                store4_split(hi, lo, ((size_t)b * (ntok - 1) + (t + k - 1)) * D + c * 4, f);
                acc.x += v[k].x, acc.y += v[k].y, acc.z += v[k].z, acc.w += v[k].w;
            }
        }
        for (; t < t1; ++t) {
            const float4 v = *reinterpret_cast<const float4*>(dx + ((size_t)b * ntok + t) * D + c * 4);
            const float f[4] = {v.x, v.y, v.z, v.w};
            store4_split(hi, lo, ((size_t)b * (ntok - 1) + (t - 1)) * D + c * 4, f);
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
      }
        ig_red_add(dbias + c * 4 + 0, acc.x), ig_red_add(dbias + c * 4 + 1, acc.y);
        ig_red_add(dbias + c * 4 + 2, acc.z), ig_red_add(dbias + c * 4 + 3, acc.w);
    }
}

// ----------------------------------------------------------------------------------------------
// K13: BatchNorm2d (+ReLU) on NHWC bf16/split tensors [M][C]   (model.py:376-377; eps 1e-5, momentum 0.1)
// ----------------------------------------------------------------------------------------------
// pass 1: per-channel sum / sum of squares: fp32 partials per thread, LDS reduction per block, one fp64
// atomic per (block, channel, statistic)
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const bf16_t* __restrict__ hi, const bf16_t* __restrict__ lo,
                                                       double* __restrict__ sums, long M, int C, int rows_per_block, float* __restrict__ part) {
    extern __shared__ float red[];  // [2C]
    const int units = C / 8;
    const int tu = min(units, TPB), nslice = TPB / tu;
    const int u = threadIdx.x % tu, sl = threadIdx.x / tu;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    const bool direct = units <= TPB;  // one column pass: every (slice, column) cell is written exactly once
    if (!direct) {
        for (int i = threadIdx.x; i < 2 * C; i += TPB) red[i] = 0.f;
        __syncthreads();
    }
    if (sl < nslice) {
        for (int ub = u; ub < units; ub += tu) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            long r = r0 + sl;
            for (; r + 3L * nslice < r1; r += 4L * nslice) {  // four rows in flight per thread
                float f[4][8];
#pragma unroll
                for (int k = 0; k < 4; ++k) load8_split(hi, lo, (size_t)(r + (long)k * nslice) * C + ub * 8, f[k]);
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) s[j] += f[k][j], q[j] = fmaf(f[k][j], f[k][j], q[j]);
            }
            for (; r < r1; r += nslice) {
                float f[8];
                load8_split(hi, lo, (size_t)r * C + ub * 8, f);
#pragma unroll
                for (int j = 0; j < 8; ++j) s[j] += f[j], q[j] = fmaf(f[j], f[j], q[j]);
            }
            if (direct) {  // slab per row slice: plain stores (LDS float atomics retire a few lanes per cycle)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(size_t)sl * 2 * C + ub * 8 + j] = s[j], red[(size_t)sl * 2 * C + C + ub * 8 + j] = q[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    atomicAdd(red + ub * 8 + j, s[j]);
                    atomicAdd(red + C + ub * 8 + j, q[j]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += TPB) {
        float t = red[i];
        if (direct)
            for (int q2 = 1; q2 < nslice; ++q2) t += red[(size_t)q2 * 2 * C + i];
        if (part) part[(size_t)blockIdx.x * 2 * C + i] = t;  // deterministic mode: bn_part_fold_kernel sums the workgroups in index order
        else atomicAdd(sums + i, (double)t);
    }
}
// finalize: train -> batch statistics (+ running update), eval -> running statistics
//   scale = gamma*rstd ; shift = beta - mean*scale ; saves mean/rstd for backward
__global__ void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean_o, float* __restrict__ rstd_o, double n, int C, float eps,
                                   float momentum, int training, int update_running) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mu, var;
    if (training) {
        double m = sums[c] / n;
        double v = sums[C + c] / n - m * m;
        if (v < 0) v = 0;
        mu = (float)m, var = (float)v;
        if (update_running) {
            double unbiased = n > 1 ? v * n / (n - 1) : v;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mu;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    } else {
        mu = running_mean[c], var = running_var[c];
    }
    float rs = rsqrtf(var + eps);
    float sc = gamma[c] * rs;
    scale[c] = sc;
    shift[c] = beta[c] - mu * sc;
    if (mean_o) mean_o[c] = mu;
    if (rstd_o) rstd_o[c] = rs;
}
// y = relu(x*scale + shift).  Thread = one fixed 8-channel unit walking rows (stride = row slices per block): the
// per-channel constants are loaded once, not per element, and four rows are in flight per thread.
__global__ __launch_bounds__(TPB) void bn_relu_apply_kernel(const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            bf16_t* __restrict__ yh, bf16_t* __restrict__ yl, long M, int C,
                                                            int rows_per_block) {
    const int units = C / 8;
    const int tu = min(units, TPB), nslice = TPB / tu;
    const int u = threadIdx.x % tu, sl = threadIdx.x / tu;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    if (sl >= nslice) return;
    for (int ub = u; ub < units; ub += tu) {
        float sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sc[j] = scale[ub * 8 + j], sh[j] = shift[ub * 8 + j];
        long r = r0 + sl;
        for (; r + 3L * nslice < r1; r += 4L * nslice) {
            float f[4][8];
#pragma unroll
            for (int k = 0; k < 4; ++k) load8_split(xh, xl, (size_t)(r + (long)k * nslice) * C + ub * 8, f[k]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int j = 0; j < 8; ++j) f[k][j] = fmaxf(fmaf(f[k][j], sc[j], sh[j]), 0.f);
                store8_split(yh, yl, (size_t)(r + (long)k * nslice) * C + ub * 8, f[k]);
            }
        }
        for (; r < r1; r += nslice) {
            float f[8];
            load8_split(xh, xl, (size_t)r * C + ub * 8, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) f[j] = fmaxf(fmaf(f[j], sc[j], sh[j]), 0.f);
            store8_split(yh, yl, (size_t)r * C + ub * 8, f);
        }
    }
}
// backward pass 1: sums[c] += dyr ; sums[C+c] += dyr*xhat   (dyr = dy * [bn(x) > 0])
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl,
                                                            const bf16_t* __restrict__ dyh, const bf16_t* __restrict__ dyl,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            double* __restrict__ sums, long M, int C, int rows_per_block, float* __restrict__ part) {
    extern __shared__ float red[];  // [2C]
    const int units = C / 8;
    const int tu = min(units, TPB), nslice = TPB / tu;
    const int u = threadIdx.x % tu, sl = threadIdx.x / tu;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    const bool direct = units <= TPB;  // one column pass: every (slice, column) cell is written exactly once
    if (!direct) {
        for (int i = threadIdx.x; i < 2 * C; i += TPB) red[i] = 0.f;
        __syncthreads();
    }
    if (sl < nslice) {
        for (int ub = u; ub < units; ub += tu) {
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0}, q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sc[j] = scale[ub * 8 + j], sh[j] = shift[ub * 8 + j], mu[j] = mean[ub * 8 + j], rs[j] = rstd[ub * 8 + j];
            long r = r0 + sl;
            for (; r + 1L * nslice < r1; r += 2L * nslice) {  // two rows (four loads) in flight per thread
                float x[2][8], d[2][8];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    load8_split(xh, xl, (size_t)(r + (long)k * nslice) * C + ub * 8, x[k]);
                    load8_split(dyh, dyl, (size_t)(r + (long)k * nslice) * C + ub * 8, d[k]);
                }
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float dyr = fmaf(x[k][j], sc[j], sh[j]) > 0.f ? d[k][j] : 0.f;
                        s[j] += dyr;
                        q[j] += dyr * (x[k][j] - mu[j]) * rs[j];
                    }
            }
            for (; r < r1; r += nslice) {
                float x[8], d[8];
                load8_split(xh, xl, (size_t)r * C + ub * 8, x);
                load8_split(dyh, dyl, (size_t)r * C + ub * 8, d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dyr = fmaf(x[j], sc[j], sh[j]) > 0.f ? d[j] : 0.f;
                    s[j] += dyr;
                    q[j] += dyr * (x[j] - mu[j]) * rs[j];
                }
            }
            if (direct) {  // slab per row slice: plain stores (LDS float atomics retire a few lanes per cycle)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[(size_t)sl * 2 * C + ub * 8 + j] = s[j], red[(size_t)sl * 2 * C + C + ub * 8 + j] = q[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    atomicAdd(red + ub * 8 + j, s[j]);
                    atomicAdd(red + C + ub * 8 + j, q[j]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += TPB) {
        float t = red[i];
        if (direct)
            for (int q2 = 1; q2 < nslice; ++q2) t += red[(size_t)q2 * 2 * C + i];
        if (part) part[(size_t)blockIdx.x * 2 * C + i] = t;  // deterministic mode: bn_part_fold_kernel sums the workgroups in index order
        else atomicAdd(sums + i, (double)t);
    }
}
// backward pass 2: dx = scale*(dyr - sum_dy/n - xhat*sum_dyxhat/n); also emits dgamma/dbeta once (block 0).
// Same thread -> fixed-channel-unit walk as the forward apply: six per-channel constants live in registers.
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const bf16_t* __restrict__ xh, const bf16_t* __restrict__ xl,
                                                           const bf16_t* __restrict__ dyh, const bf16_t* __restrict__ dyl,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const double* __restrict__ sums, bf16_t* __restrict__ dxh,
                                                           bf16_t* __restrict__ dxl, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, long M, int C, double n, int rows_per_block) {
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) atomicAdd(dbeta + c, (float)sums[c]);  // one contributor per element: order-independent
            if (dgamma) atomicAdd(dgamma + c, (float)sums[C + c]);
        }
    }
    const int units = C / 8;
    const int tu = min(units, TPB), nslice = TPB / tu;
    const int u = threadIdx.x % tu, sl = threadIdx.x / tu;
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    if (sl >= nslice) return;
    const double inv_n = 1.0 / n;
    for (int ub = u; ub < units; ub += tu) {
        // dx = sc*dyr - a - b*x with  xhat = (x - mu)*rs :  a = sc*(k1 - mu*rs*k2),  b = sc*rs*k2
        float sc[8], sh[8], ca[8], cb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = ub * 8 + j;
            const float k1 = (float)(sums[c] * inv_n), k2 = (float)(sums[C + c] * inv_n);
            sc[j] = scale[c], sh[j] = shift[c];
            cb[j] = sc[j] * rstd[c] * k2;
            ca[j] = sc[j] * k1 - mean[c] * cb[j];
        }
        long r = r0 + sl;
        for (; r + 1L * nslice < r1; r += 2L * nslice) {
            float x[2][8], d[2][8];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                load8_split(xh, xl, (size_t)(r + (long)k * nslice) * C + ub * 8, x[k]);
                load8_split(dyh, dyl, (size_t)(r + (long)k * nslice) * C + ub * 8, d[k]);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                float o[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float dyr = fmaf(x[k][j], sc[j], sh[j]) > 0.f ? d[k][j] : 0.f;
                    o[j] = fmaf(sc[j], dyr, -fmaf(cb[j], x[k][j], ca[j]));
                }
                store8_split(dxh, dxl, (size_t)(r + (long)k * nslice) * C + ub * 8, o);
            }
        }
        for (; r < r1; r += nslice) {
            float x[8], d[8], o[8];
            load8_split(xh, xl, (size_t)r * C + ub * 8, x);
            load8_split(dyh, dyl, (size_t)r * C + ub * 8, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dyr = fmaf(x[j], sc[j], sh[j]) > 0.f ? d[j] : 0.f;
                o[j] = fmaf(sc[j], dyr, -fmaf(cb[j], x[j], ca[j]));
            }
            store8_split(dxh, dxl, (size_t)r * C + ub * 8, o);
        }
    }
}

// ----------------------------------------------------------------------------------------------
// K18: AdamW (torch.optim.AdamW defaults, base.py:124-126) on the flat parameter buffer, fused with the
// optional weight clamp (base.py:103-113) and the bf16 (hi/lo) shadow refresh used by the MFMA kernels.
// hyper (device): [0]=lr [1]=beta1 [2]=beta2 [3]=eps [4]=weight_decay [5]=bias_corr1 [6]=sqrt(bias_corr2)
//                 [7]=clip_lo [8]=clip_hi [9]=clip_enabled  [10]=step (as float)
//                 [11]=1-beta1 [12]=1-beta2 (host computes them in double like torch does)
//                 [13]=gradient scale (1/world_size after a SUM all-reduce; 0 is treated as 1)
// ----------------------------------------------------------------------------------------------
__global__ void adamw_advance_kernel(float* hyper) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float step = hyper[10] + 1.f;
        hyper[10] = step;
        hyper[5] = (float)(1.0 - pow((double)hyper[1], (double)step));
        hyper[6] = (float)sqrt(1.0 - pow((double)hyper[2], (double)step));
    }
}
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             bf16_t* __restrict__ sh, bf16_t* __restrict__ sl, const float* __restrict__ hyper, long n4) {
    const float lr = hyper[0], b2 = hyper[2], eps = hyper[3], wd = hyper[4], bc1 = hyper[5], bc2s = hyper[6];
    const float clo = hyper[7], chi = hyper[8];
    const float omb1 = hyper[11], omb2 = hyper[12];
    const float gscale = hyper[13] == 0.f ? 1.f : hyper[13];
    const bool clip = hyper[9] != 0.f;
    const float step_size = lr / bc1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        // streaming accesses (non-temporal): 30 bytes per parameter that nobody reads again before the next optimizer step -- kept out of
        // the L2 / MALL that the backward kernels running beside the overlapped launches live in
        const f32x4 P4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + i);
        const f32x4 G4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(g) + i);
        const f32x4 M4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(m) + i);
        const f32x4 V4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(v) + i);
        const float4 P = make_float4(P4[0], P4[1], P4[2], P4[3]), G = make_float4(G4[0], G4[1], G4[2], G4[3]);
        const float4 Mv = make_float4(M4[0], M4[1], M4[2], M4[3]), V = make_float4(V4[0], V4[1], V4[2], V4[3]);
        float pp[4] = {P.x, P.y, P.z, P.w}, gg[4] = {G.x * gscale, G.y * gscale, G.z * gscale, G.w * gscale}, mm[4] = {Mv.x, Mv.y, Mv.z, Mv.w},
              vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pp[j] = pp[j] * (1.f - lr * wd);
            mm[j] = mm[j] + (gg[j] - mm[j]) * omb1;  // lerp_(grad, 1-beta1)
            vv[j] = vv[j] * b2 + omb2 * gg[j] * gg[j];
            float denom = sqrtf(vv[j]) / bc2s + eps;
            pp[j] = pp[j] - step_size * (mm[j] / denom);
            if (clip) pp[j] = fminf(fmaxf(pp[j], clo), chi);
        }
        __builtin_nontemporal_store(f32x4{pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<f32x4*>(p) + i);
        __builtin_nontemporal_store(f32x4{mm[0], mm[1], mm[2], mm[3]}, reinterpret_cast<f32x4*>(m) + i);
        __builtin_nontemporal_store(f32x4{vv[0], vv[1], vv[2], vv[3]}, reinterpret_cast<f32x4*>(v) + i);
        if (sh) store4_split(sh, sl, (size_t)i * 4, pp);
    }
}

// dst[b][c][r] = src[b][r][c], bf16, 64 x 64 tiles through LDS (16-byte global loads and stores on both sides).  Used once per
// optimizer step on the Block linears' weights: the transposed copy makes their data gradients K-contiguous GEMMs (gemm8.hip).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int R, int C,
                                                             long src_stride, long dst_stride) {
    __shared__ bf16_t tile[64][66];
    const int t = threadIdx.x;
    const long b = blockIdx.z;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const bf16_t* s = src + b * src_stride;
    bf16_t* d = dst + b * dst_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (t >> 3) + i * 32, ch = t & 7;
        const uint4 u = *reinterpret_cast<const uint4*>(s + (long)(r0 + r) * C + c0 + ch * 8);
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
        const int c = (t >> 3) + i * 32, ch = t & 7;  // output row = source column c, output columns = source rows ch*8 ..
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = (uint32_t)tile[ch * 8 + 2 * j][c] | ((uint32_t)tile[ch * 8 + 2 * j + 1][c] << 16);
        *reinterpret_cast<uint4*>(d + (long)(c0 + c) * R + r0 + ch * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}


}  // namespace
IG_DET_TU(elementwise)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

#define ST(s) ((hipStream_t)(s))

extern "C" {

// src_dtype: 0 = int16, 1 = float32.  mult_enabled=0 skips the constant multiplier.
int ig_normalize_chips(const void* src, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                       float* dst, int B, int T, int C, int H, int W, void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(src && mean && stdv && dst, "ig_normalize_chips: null pointer");
    IG_REQUIRE(B >= 0 && T > 0 && C > 0 && H > 0 && W > 0, "ig_normalize_chips: bad dims");
    long HW = (long)H * W;
    IG_REQUIRE(HW % 4 == 0, "ig_normalize_chips: H*W must be a multiple of 4 (got %ld)", HW);
    long total4 = (long)B * T * C * HW / 4;
    if (total4 == 0) return IG_OK;
    int grid = grid_for(total4, TPB, 8192);
    if (src_dtype == 0)
        hipLaunchKernelGGL(normalize_kernel<int16_t>, dim3(grid), dim3(TPB), 0, ST(stream), (const int16_t*)src, dst, mean, stdv,
                           mult, mult_enabled, T, C, HW, total4);
    else if (src_dtype == 1)
        hipLaunchKernelGGL(normalize_kernel<float>, dim3(grid), dim3(TPB), 0, ST(stream), (const float*)src, dst, mean, stdv, mult,
                           mult_enabled, T, C, HW, total4);
    else {
        ig_set_error("ig_normalize_chips: unsupported src_dtype %d", src_dtype);
        return IG_ERR_UNSUPPORTED;
    }
    return ig_check_launch("ig_normalize_chips");
}

int ig_crop_flip_normalize(const void* src, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                           const int* params, float* dst, const float* labels_in, float* labels_out, int B, int T, int C, int Hs,
                           int Ws, int im, void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(src && mean && stdv && params && dst, "ig_crop_flip_normalize: null pointer");
    IG_REQUIRE((labels_in == nullptr) == (labels_out == nullptr), "ig_crop_flip_normalize: labels_in and labels_out go together");
    IG_REQUIRE(B >= 0 && T > 0 && C > 0 && im > 0 && im <= Hs && im <= Ws, "ig_crop_flip_normalize: need 0 < im <= Hs, Ws");
    IG_REQUIRE(im % 4 == 0, "ig_crop_flip_normalize: im must be a multiple of 4 (got %d)", im);
    const long img4 = (long)B * T * C * im * (im / 4);
    const long total4 = img4 + (labels_in ? (long)B * im * (im / 4) : 0L);
    if (total4 == 0) return IG_OK;
    const int grid = grid_for(total4, TPB, 8192);
    if (src_dtype == 0)
        hipLaunchKernelGGL(crop_flip_normalize_kernel<int16_t>, dim3(grid), dim3(TPB), 0, ST(stream), (const int16_t*)src, dst, mean,
                           stdv, mult, mult_enabled, params, labels_in, labels_out, T, C, Hs, Ws, im, total4, img4,
                           (long)T * C * Hs * Ws, (long)Hs * Ws, 4);
    else if (src_dtype == 1)
        hipLaunchKernelGGL(crop_flip_normalize_kernel<float>, dim3(grid), dim3(TPB), 0, ST(stream), (const float*)src, dst, mean, stdv,
                           mult, mult_enabled, params, labels_in, labels_out, T, C, Hs, Ws, im, total4, img4, (long)T * C * Hs * Ws,
                           (long)Hs * Ws, 4);
    else {
        ig_set_error("ig_crop_flip_normalize: unsupported src_dtype %d", src_dtype);
        return IG_ERR_UNSUPPORTED;
    }
    return ig_check_launch("ig_crop_flip_normalize");
}

// Sliding-window gather + normalise: n windows of ONE tile in one launch (the reference's process_test loops over the origins
// in Python and crops / normalises window by window, dataloader.py:618-669)
int ig_normalize_windows(const void* tile, int src_dtype, const float* mean, const float* stdv, double mult, int mult_enabled,
                         const int* origins, float* dst, const float* labels_tile, float* labels_out, int n, int T, int C, int Hs,
                         int Ws, int crop, void* stream) {
    if (n == 0) return IG_OK;  // no windows: origins / dst are null
    IG_REQUIRE(tile && mean && stdv && origins && dst, "ig_normalize_windows: null pointer");
    IG_REQUIRE((labels_tile == nullptr) == (labels_out == nullptr), "ig_normalize_windows: labels_tile and labels_out go together");
    IG_REQUIRE(n >= 0 && T > 0 && C > 0 && crop > 0 && crop <= Hs && crop <= Ws, "ig_normalize_windows: need 0 < crop <= Hs, Ws");
    IG_REQUIRE(crop % 4 == 0, "ig_normalize_windows: crop must be a multiple of 4 (got %d)", crop);
    const long img4 = (long)n * T * C * crop * (crop / 4);
    const long total4 = img4 + (labels_tile ? (long)n * crop * (crop / 4) : 0L);
    if (total4 == 0) return IG_OK;
    const int grid = grid_for(total4, TPB, 8192);
    if (src_dtype == 0)
        hipLaunchKernelGGL(crop_flip_normalize_kernel<int16_t>, dim3(grid), dim3(TPB), 0, ST(stream), (const int16_t*)tile, dst, mean,
                           stdv, mult, mult_enabled, origins, labels_tile, labels_out, T, C, Hs, Ws, crop, total4, img4, 0L, 0L, 2);
    else if (src_dtype == 1)
        hipLaunchKernelGGL(crop_flip_normalize_kernel<float>, dim3(grid), dim3(TPB), 0, ST(stream), (const float*)tile, dst, mean, stdv,
                           mult, mult_enabled, origins, labels_tile, labels_out, T, C, Hs, Ws, crop, total4, img4, 0L, 0L, 2);
    else {
        ig_set_error("ig_normalize_windows: unsupported src_dtype %d", src_dtype);
        return IG_ERR_UNSUPPORTED;
    }
    return ig_check_launch("ig_normalize_windows");
}

int ig_chip_stats(const float* x, double* sums, int B, int C, long n_per_channel, void* stream) {
    IG_REQUIRE(x && sums, "ig_chip_stats: null pointer");
    IG_REQUIRE(C > 0 && n_per_channel > 0 && n_per_channel % 4 == 0, "ig_chip_stats: T*H*W must be a positive multiple of 4");
    if (B == 0) return IG_OK;
    hipLaunchKernelGGL(chip_stats_kernel, dim3((unsigned)((long)B * C)), dim3(TPB), 0, ST(stream), x, sums, C, n_per_channel);
    return ig_check_launch("ig_chip_stats");
}

int ig_label_hist(const float* labels, unsigned long long* counts, long n, int lo, int nbins, void* stream) {
    IG_REQUIRE(labels && counts, "ig_label_hist: null pointer");
    IG_REQUIRE(nbins > 0 && nbins <= 4096, "ig_label_hist: 1 <= nbins <= 4096");
    if (n == 0) return IG_OK;
    long nb = (n + TPB - 1) / TPB;
    if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(label_hist_kernel, dim3((unsigned)nb), dim3(TPB), (nbins + 1) * sizeof(unsigned int), ST(stream), labels, counts,
                       n, lo, nbins);
    return ig_check_launch("ig_label_hist");
}

int ig_patchify(const float* img, void* out_hi, void* out_lo, int B, int C, int T, int H, int W, int p, void* stream) {
    IG_REQUIRE(img && out_hi, "ig_patchify: null pointer");
    IG_REQUIRE(p > 0 && (C * p * p) % 8 == 0, "ig_patchify: C * p * p must be a multiple of 8 (C=%d p=%d)", C, p);
    int gh = H / p, gw = W / p;
    long units = (long)B * T * gh * gw * C * p * p / 8;
    if (units == 0) return IG_OK;
    if (p % 8 == 0 && W % 4 == 0)
        hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(units, TPB, 16384)), dim3(TPB), 0, ST(stream), img, (bf16_t*)out_hi,
                           (bf16_t*)out_lo, C, T, H, W, p, gh, gw, units);
    else  // patch 14 (600M variants): 8-element units straddle image rows
        hipLaunchKernelGGL(patchify_generic_kernel, dim3(grid_for(units, TPB, 16384)), dim3(TPB), 0, ST(stream), img, (bf16_t*)out_hi,
                           (bf16_t*)out_lo, C, T, H, W, p, gh, gw, units);
    return ig_check_launch("ig_patchify");
}

int ig_cls_rows(float* x, const float* cls, const float* pos, int B, int ntok, int D, void* stream) {
    IG_REQUIRE(x && cls && pos, "ig_cls_rows: null pointer");
    long n = (long)B * D;
    if (n == 0) return IG_OK;
    hipLaunchKernelGGL(cls_rows_kernel, dim3(grid_for(n, TPB)), dim3(TPB), 0, ST(stream), x, cls, pos, B, (long)ntok * D, D);
    return ig_check_launch("ig_cls_rows");
}

// feat_T = 0: out[M][D]; feat_T >= 1: K9 feature-image layout [B][G][D*T] with the cls row dropped
int ig_layernorm_fwd(const float* x, const float* gamma, const float* beta, void* out_hi, void* out_lo, float* mean,
                     float* rstd, int M, int D, float eps, int feat_T, int feat_G, int ntok, void* stream) {
    IG_REQUIRE(x && gamma && beta && out_hi, "ig_layernorm_fwd: null pointer");
    IG_REQUIRE(D % 4 == 0 && D <= 2048, "ig_layernorm_fwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
    if (M == 0) return IG_OK;
    dim3 grid(ig_cdiv(M, TPB / 64));
    if (D <= 1024)
        hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, dim3(TPB), 0, ST(stream), x, gamma, beta, (bf16_t*)out_hi, (bf16_t*)out_lo,
                           mean, rstd, M, D, eps, feat_T, feat_G, ntok);
    else
        hipLaunchKernelGGL(layernorm_fwd_kernel<8>, grid, dim3(TPB), 0, ST(stream), x, gamma, beta, (bf16_t*)out_hi, (bf16_t*)out_lo,
                           mean, rstd, M, D, eps, feat_T, feat_G, ntok);
    return ig_check_launch("ig_layernorm_fwd");
}

int ig_layernorm_bwd(const void* dy_hi, const void* dy_lo, const float* x, const float* mean, const float* rstd,
                     const float* gamma, float* dx, int accumulate, void* dxb_hi, void* dxb_lo, float* dgamma, float* dbeta,
                     float* dcol, int M, int D, int feat_T, int feat_G, int ntok, void* stream) {
    IG_REQUIRE(dy_hi && x && mean && rstd && gamma && dx, "ig_layernorm_bwd: null pointer");
    IG_REQUIRE(D % 4 == 0 && D <= 2048, "ig_layernorm_bwd: D must be a multiple of 4 and <= 2048 (got %d)", D);
    if (M == 0) return IG_OK;
    // 48 rows per workgroup (measured best of 16..128 at M = 21168, tools/ln_bench.py).  For small M keep ~200 workgroups
    // (every workgroup ends in 3 D atomics, so more is not better): M = 3152 (the YAML's batch 16) 16 rows -> 21 us against
    // 38 us at 48 and 24 us at 8; M = 10638 stays at 48 (60 us; 32 rows: 75 us)
    // From M = 21168 up the best grid stays at ~440 workgroups (1.7 per CU): M = 42552 (batch 216) 96 rows 113 us against 119 us at 48
    // and 147-210 us at 144-240 rows (r03: IG_LNB_RPB sweep)
    int rpb = (int)((((long)M + 209) / 210 + 7) / 8 * 8);
    if (rpb < 8) rpb = 8;
    if (rpb > 48) rpb = std::max(48, (int)(((long)M / 444 + 4) / 8 * 8));  // nearest multiple of 8 to M / 444
    dim3 grid(ig_cdiv(M, rpb));
    size_t sm = 3 * (size_t)D * sizeof(float);
    const size_t sm_exact = (LNB_TPB / 64) * sm;  // one slab per wave (plain-store reduction)
#define IG_LNB_EXACT(NCH_)                                                                                              \
    {                                                                                                                 \
        if (feat_T > 0)                                                                                               \
            hipLaunchKernelGGL((layernorm_bwd_exact_kernel<NCH_, true>), grid, dim3(LNB_TPB), sm_exact, ST(stream),  \
                               (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, x, mean, rstd, gamma, dx, accumulate,      \
                               (bf16_t*)dxb_hi, (bf16_t*)dxb_lo, dgamma, dbeta, dcol, M, D, rpb, feat_T, feat_G, ntok); \
        else                                                                                                          \
            hipLaunchKernelGGL((layernorm_bwd_exact_kernel<NCH_, false>), grid, dim3(LNB_TPB), sm_exact, ST(stream), \
                               (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, x, mean, rstd, gamma, dx, accumulate,      \
                               (bf16_t*)dxb_hi, (bf16_t*)dxb_lo, dgamma, dbeta, dcol, M, D, rpb, feat_T, feat_G, ntok); \
        return ig_check_launch("ig_layernorm_bwd");                                                                   \
    }
    if (D == 256) IG_LNB_EXACT(1)
    if (D == 768) IG_LNB_EXACT(3)
    if (D == 1024) IG_LNB_EXACT(4)
    if (D == 1280) IG_LNB_EXACT(5)
#undef IG_LNB_EXACT
    if (D <= 1024)
        hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, dim3(LNB_TPB), sm, ST(stream), (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, x,
                           mean, rstd, gamma, dx, accumulate, (bf16_t*)dxb_hi, (bf16_t*)dxb_lo, dgamma, dbeta, dcol, M, D, rpb,
                           feat_T, feat_G, ntok);
    else
        hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(LNB_TPB), sm, ST(stream), (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, x,
                           mean, rstd, gamma, dx, accumulate, (bf16_t*)dxb_hi, (bf16_t*)dxb_lo, dgamma, dbeta, dcol, M, D, rpb,
                           feat_T, feat_G, ntok);
    return ig_check_launch("ig_layernorm_bwd");
}

int ig_colsum(const void* hi, const void* lo, float* out, long M, int C, void* stream) { return ig_colsum_ld(hi, lo, out, M, C, C, stream); }
}  // extern "C"
// internal: column sums of the first C columns of a row-major matrix with row pitch ld (elements)
int ig_colsum_ld(const void* hi, const void* lo, float* out, long M, int C, long ld, void* stream) {
    IG_REQUIRE(hi && out, "ig_colsum: null pointer");
    IG_REQUIRE(C % 8 == 0 && ld % 8 == 0 && ld >= C, "ig_colsum: C and the row pitch must be multiples of 8 (got %d, %ld)", C, ld);
    if (M == 0) return IG_OK;
    // ~768 workgroups in total: fewer leave HBM idle, more are bound by the final global atomics (measured, tools/colsum_bench.py)
    const long cchunks = ig_cdiv(C, 1024);
    int rpb = (int)(((M * cchunks + 767) / 768 + 31) / 32 * 32);
    hipLaunchKernelGGL(colsum_kernel, dim3(ig_cdiv(M, rpb), ig_cdiv(C, 1024)), dim3(TPB), 0, ST(stream), (const bf16_t*)hi, (const bf16_t*)lo, out, M,
                       C, rpb, ld);
    return ig_check_launch("ig_colsum");
}
extern "C" {

int ig_split_bf16(const float* src, void* hi, void* lo, long n, void* stream) {
    IG_REQUIRE(src && hi, "ig_split_bf16: null pointer");
    if (n == 0) return IG_OK;
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(n, TPB, 16384)), dim3(TPB), 0, ST(stream), src, (bf16_t*)hi, (bf16_t*)lo, n);
    return ig_check_launch("ig_split_bf16");
}

int ig_transpose_bf16(const void* src_hi, const void* src_lo, void* dst_hi, void* dst_lo, int R, int C, int batch, long src_stride,
                      long dst_stride, void* stream) {
    IG_REQUIRE(src_hi && dst_hi, "ig_transpose_bf16: null pointer");
    IG_REQUIRE((src_lo == nullptr) == (dst_lo == nullptr), "ig_transpose_bf16: source and destination must both be split or both plain");
    IG_REQUIRE(R > 0 && C > 0 && R % 64 == 0 && C % 64 == 0, "ig_transpose_bf16: R and C must be multiples of 64 (got %d, %d)", R, C);
    IG_REQUIRE(batch >= 1 && batch <= 65535, "ig_transpose_bf16: 1 <= batch <= 65535");
    IG_REQUIRE((((uintptr_t)src_hi | (uintptr_t)dst_hi | (uintptr_t)src_lo | (uintptr_t)dst_lo) & 15) == 0 && src_stride % 8 == 0 &&
                   dst_stride % 8 == 0, "ig_transpose_bf16: pointers and strides must be 16-byte aligned");
    const dim3 grid(C / 64, R / 64, batch);
    hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_hi, (bf16_t*)dst_hi, R, C, src_stride,
                       dst_stride);
    if (src_lo)
        hipLaunchKernelGGL(transpose_bf16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src_lo, (bf16_t*)dst_lo, R, C,
                           src_stride, dst_stride);
    return ig_check_launch("ig_transpose_bf16");
}

int ig_merge_bf16(const void* hi, const void* lo, float* dst, long n, void* stream) {
    IG_REQUIRE(hi && dst, "ig_merge_bf16: null pointer");
    if (n == 0) return IG_OK;
    hipLaunchKernelGGL(merge_kernel, dim3(grid_for(n, TPB, 16384)), dim3(TPB), 0, ST(stream), (const bf16_t*)hi, (const bf16_t*)lo,
                       dst, n);
    return ig_check_launch("ig_merge_bf16");
}

int ig_patch_grad_prep(const float* dx, void* hi, void* lo, float* dcls, float* dbias, int B, int ntok, int D, void* stream) {
    IG_REQUIRE(dx && hi && dcls && dbias, "ig_patch_grad_prep: null pointer");
    IG_REQUIRE(D % 4 == 0, "ig_patch_grad_prep: D must be a multiple of 4");
    if (B == 0) return IG_OK;
    const int gy = B <= 8 ? B : B <= 64 ? (B + 1) / 2 : (B + 7) / 8;  // batch groups: enough workgroups at small B, few adds at large B
    hipLaunchKernelGGL(patch_grad_prep_kernel, dim3(ig_cdiv(ntok, 16), gy), dim3(TPB), 0, ST(stream), dx, (bf16_t*)hi, (bf16_t*)lo,
                       dcls, dbias, B, ntok, D);
    return ig_check_launch("ig_patch_grad_prep");
}

// rows per workgroup: reductions end in global atomics -> ~768 workgroups (see ig_colsum); streaming apply passes give
// every thread 8 rows of its channel unit
// LDS of the BN reductions: one [2C] slab per row slice when a workgroup covers all channels in one pass
static inline size_t bn_red_bytes(int C) {
    const int units = C / 8;
    return (units <= TPB ? (size_t)(TPB / units) : (size_t)1) * 2 * C * sizeof(float);
}
static inline int bn_reduce_rows(long M) { return (int)(((M + 767) / 768 + 31) / 32 * 32); }
static inline int bn_apply_rows(int C) {
    const int units = C / 8, tu = units < TPB ? units : TPB;
    return (TPB / tu) * 8;
}

// BatchNorm(+ReLU) forward.  sums: device scratch double[2*C] (zeroed here).  training=1: batch statistics
// and (update_running=1) running-stat update; training=0: running statistics.
int ig_bn_relu_fwd(const void* x_hi, const void* x_lo, const float* gamma, const float* beta, float* running_mean,
                   float* running_var, void* y_hi, void* y_lo, float* scale, float* shift, float* mean, float* rstd,
                   double* sums, long M, int C, float eps, float momentum, int training, int update_running, void* stream) {
    IG_REQUIRE(x_hi && gamma && beta && running_mean && running_var && scale && shift && sums, "ig_bn_relu_fwd: null pointer");
    IG_REQUIRE(C % 8 == 0 && C <= 4096, "ig_bn_relu_fwd: C must be a multiple of 8 and <= 4096 (got %d)", C);
    if (M == 0) return IG_OK;
    if (training == 1) {  // (training == 2: a producer has already left the statistics in sums -- ig_conv3x3_fwd_stats)
        const int rpb = bn_reduce_rows(M), nwg = ig_cdiv(M, rpb);
        float* part = nullptr;
        if (ig_deterministic()) {
            part = (float*)ig_scratch(0, (size_t)nwg * 2 * C * sizeof(float), ST(stream));
            IG_REQUIRE(part, "ig_bn_relu_fwd: scratch allocation failed");
        } else {
            (void)hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(double), ST(stream));
        }
        hipLaunchKernelGGL(bn_stats_kernel, dim3(nwg), dim3(TPB), bn_red_bytes(C), ST(stream), (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                           sums, M, C, rpb, part);
        if (part) hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * C, 64)), dim3(1024), 0, ST(stream), part, sums, nwg, 2 * C);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ig_cdiv(C, TPB)), dim3(TPB), 0, ST(stream), sums, gamma, beta, running_mean,
                       running_var, scale, shift, mean, rstd, (double)M, C, eps, momentum, training, update_running);
    if (!y_hi) return ig_check_launch("ig_bn_relu_fwd");  // statistics only: the consumer applies scale / shift itself (ig_classifier_bn_fwd)
    const int arpb = bn_apply_rows(C);
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(ig_cdiv(M, arpb)), dim3(TPB), 0, ST(stream), (const bf16_t*)x_hi,
                       (const bf16_t*)x_lo, scale, shift, (bf16_t*)y_hi, (bf16_t*)y_lo, M, C, arpb);
    return ig_check_launch("ig_bn_relu_fwd");
}

// Training-mode statistics -> scale / shift / mean / rstd (+ running update) from sums[2C] that a producer has already filled
// (ig_conv3x3_fwd_stats): the finalize step of ig_bn_relu_fwd alone.
int ig_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                   float* shift, float* mean, float* rstd, long M, int C, float eps, float momentum, int update_running, void* stream) {
    IG_REQUIRE(sums && gamma && beta && running_mean && running_var && scale && shift, "ig_bn_finalize: null pointer");
    if (M == 0 || C == 0) return IG_OK;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ig_cdiv(C, TPB)), dim3(TPB), 0, ST(stream), sums, gamma, beta, running_mean, running_var,
                       scale, shift, mean, rstd, (double)M, C, eps, momentum, 1, update_running);
    return ig_check_launch("ig_bn_finalize");
}

// BatchNorm(+ReLU) backward (training statistics): x = saved conv output, dy = grad of the ReLU output
int ig_bn_relu_bwd(const void* x_hi, const void* x_lo, const void* dy_hi, const void* dy_lo, const float* scale,
                   const float* shift, const float* mean, const float* rstd, void* dx_hi, void* dx_lo, float* dgamma,
                   float* dbeta, double* sums, long M, int C, void* stream) {
    IG_REQUIRE(x_hi && dy_hi && scale && shift && mean && rstd && dx_hi && sums, "ig_bn_relu_bwd: null pointer");
    IG_REQUIRE(C % 8 == 0 && C <= 4096, "ig_bn_relu_bwd: C must be a multiple of 8 and <= 4096 (got %d)", C);
    if (M == 0) return IG_OK;
    const int rpb = bn_reduce_rows(M), nwg = ig_cdiv(M, rpb);
    float* part = nullptr;
    if (ig_deterministic()) {
        part = (float*)ig_scratch(0, (size_t)nwg * 2 * C * sizeof(float), ST(stream));
        IG_REQUIRE(part, "ig_bn_relu_bwd: scratch allocation failed");
    } else {
        (void)hipMemsetAsync(sums, 0, 2 * (size_t)C * sizeof(double), ST(stream));
    }
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nwg), dim3(TPB), bn_red_bytes(C), ST(stream), (const bf16_t*)x_hi, (const bf16_t*)x_lo,
                       (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, scale, shift, mean, rstd, sums, M, C, rpb, part);
    if (part) hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * C, 64)), dim3(1024), 0, ST(stream), part, sums, nwg, 2 * C);
    const int arpb = bn_apply_rows(C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ig_cdiv(M, arpb)), dim3(TPB), 0, ST(stream), (const bf16_t*)x_hi,
                       (const bf16_t*)x_lo, (const bf16_t*)dy_hi, (const bf16_t*)dy_lo, scale, shift, mean, rstd, sums, (bf16_t*)dx_hi,
                       (bf16_t*)dx_lo, dgamma, dbeta, M, C, (double)M, arpb);
    return ig_check_launch("ig_bn_relu_bwd");
}

// eval-mode BatchNorm as a per-channel affine: scale = gamma*rsqrt(running_var+eps), shift = beta - running_mean*scale
// (used by ig_conv3x3_fwd's fused BN+ReLU epilogue at inference)
int ig_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float* scale,
                      float* shift, int C, float eps, void* stream) {
    IG_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "ig_bn_eval_affine: null pointer");
    if (C == 0) return IG_OK;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(ig_cdiv(C, TPB)), dim3(TPB), 0, ST(stream), (const double*)nullptr, gamma, beta,
                       (float*)running_mean, (float*)running_var, scale, shift, (float*)nullptr, (float*)nullptr, 1.0, C, eps, 0.f, 0, 0);
    return ig_check_launch("ig_bn_eval_affine");
}

// hyper: device float[16] (layout above).  ig_adamw_advance increments the step and the bias corrections
// on the device so that a captured graph can be replayed without host-side scalars.
int ig_adamw_advance(float* hyper, void* stream) {
    IG_REQUIRE(hyper, "ig_adamw_advance: null pointer");
    hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(64), 0, ST(stream), hyper);
    return ig_check_launch("ig_adamw_advance");
}
int ig_adamw_step(float* p, const float* g, float* m, float* v, void* shadow_hi, void* shadow_lo, const float* hyper, long n,
                  void* stream) {
    IG_REQUIRE(p && g && m && v && hyper, "ig_adamw_step: null pointer");
    IG_REQUIRE(n % 4 == 0, "ig_adamw_step: n must be a multiple of 4 (pad the flat buffer)");
    if (n == 0) return IG_OK;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n / 4, TPB, 8192)), dim3(TPB), 0, ST(stream), p, g, m, v, (bf16_t*)shadow_hi,
                       (bf16_t*)shadow_lo, hyper, n / 4);
    return ig_check_launch("ig_adamw_step");
}

}  // extern "C"
