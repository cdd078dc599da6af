// Photometric / resampling augmentations of the training input pipeline on the device (second part).
// Reference: instageo/model/dataloader.py:144-386 -- RandomRotation, RandomBrightnessContrast, RandomGaussianBlur,
// RandomGaussianNoise work chip by chip and band by band on PIL images in DataLoader workers; here one launch handles a whole
// batch that is already resident as raw-domain float32 (B, T*C, S, S) (the output of ig_crop_flip_normalize with identity
// statistics).  Every random decision (apply?, angle, factors, seed) is drawn on the HOST and handed over per chip, like the
// crop origins and flips.  All four are HBM-bound streaming passes (8-12 bytes per pixel and band).
#include "common.h"

namespace {

constexpr int ATPB = 256;

static inline int aug_grid(long n) {
    long g = (n + ATPB - 1) / ATPB;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

// Nearest-neighbour rotation about the chip centre with constant fill: Pillow's Image.rotate as torchvision's
// transforms.functional.rotate calls it (dataloader.py:183-186).  prm[b] = {apply, a0, a1, a2, a3, a4, a5, 0}: the six 16.16
// fixed-point coefficients of Pillow's affine walk (Geometry.c affine_fixed; pixel centres folded into a2 / a5), computed on
// the host from the drawn angle.  Band CT of a chip is its label (own fill value).
__global__ __launch_bounds__(ATPB) void aug_rotate_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          const float* __restrict__ lab_src, float* __restrict__ lab_dst,
                                                          const int* __restrict__ prm, float fill, float lab_fill, int CT, int S,
                                                          int planes, long total) {
    const long ss = (long)S * S;
    for (long i = blockIdx.x * (long)ATPB + threadIdx.x; i < total; i += (long)gridDim.x * ATPB) {
        const long plane = i / ss;
        const int pix = (int)(i - plane * ss);
        const int b = (int)(plane / planes), band = (int)(plane - (long)b * planes);
        const bool is_lab = band == CT;
        const float* s = is_lab ? lab_src + (long)b * ss : src + ((long)b * CT + band) * ss;
        float* d = is_lab ? lab_dst + (long)b * ss : dst + ((long)b * CT + band) * ss;
        const int* q = prm + b * 8;
        float v;
        if (!q[0]) {
            v = s[pix];
        } else {
            const int y = pix / S, x = pix - y * S;
            const int xin = (q[3] + q[2] * y + q[1] * x) >> 16;
            const int yin = (q[6] + q[5] * y + q[4] * x) >> 16;
            v = (xin >= 0 && xin < S && yin >= 0 && yin < S) ? s[yin * S + xin] : (is_lab ? lab_fill : fill);
        }
        d[pix] = v;
    }
}

// arr*bright, then contrast about the band mean of THAT product, clamp to [0, max] (dataloader.py:230-238).  One workgroup per
// (chip, band): pass 1 = mean (fp64 accumulation of the fp32 products), pass 2 = apply; the band (<= 200 KiB) stays in L2.
__global__ __launch_bounds__(ATPB) void aug_brightness_contrast_kernel(float* __restrict__ buf, const float* __restrict__ prm, float max_pixel,
                                                                       int CT, int n) {
    const int b = blockIdx.x / CT;
    const float* q = prm + b * 4;
    if (q[0] == 0.f) return;
    const float bright = q[1], contrast = q[2];
    float* p = buf + (long)blockIdx.x * n;
    double acc = 0.0;
    for (int i = threadIdx.x * 4; i < n; i += ATPB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(p + i);
        acc += (double)(v.x * bright) + (double)(v.y * bright) + (double)(v.z * bright) + (double)(v.w * bright);
    }
    __shared__ double red[ATPB / 64];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < ATPB / 64; ++w) tot += red[w];
    const float mean = (float)(tot / (double)n);
    for (int i = threadIdx.x * 4; i < n; i += ATPB * 4) {
        float4 v = *reinterpret_cast<const float4*>(p + i);
        float* e = reinterpret_cast<float*>(&v);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float t = e[k] * bright;
            t = (t - mean) * contrast + mean;
            e[k] = fminf(fmaxf(t, 0.f), max_pixel);
        }
        *reinterpret_cast<float4*>(p + i) = v;
    }
}

// clip to [0, max] / max, k x k Gaussian with reflect padding, clamp to [0, 1], * max, truncate to uint16 (dataloader.py:296-314;
// torchvision's gaussian_blur = reflect pad + depth-wise conv2d with the outer product of the two 1-D kernels, given in k2).
__global__ __launch_bounds__(ATPB) void aug_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ apply,
                                                        const float* __restrict__ k2, int ksize, float max_pixel, int CT, int S, long total) {
    const long ss = (long)S * S;
    const int half = ksize >> 1;
    for (long i = blockIdx.x * (long)ATPB + threadIdx.x; i < total; i += (long)gridDim.x * ATPB) {
        const long plane = i / ss;
        const int pix = (int)(i - plane * ss);
        const int b = (int)(plane / CT);
        const float* s = src + plane * ss;
        if (!apply[b]) {
            dst[i] = s[pix];
            continue;
        }
        const int y = pix / S, x = pix - y * S;
        float acc = 0.f;
        for (int dy = 0; dy < ksize; ++dy) {
            int yy = y + dy - half;
            yy = yy < 0 ? -yy : (yy >= S ? 2 * S - 2 - yy : yy);
            for (int dx = 0; dx < ksize; ++dx) {
                int xx = x + dx - half;
                xx = xx < 0 ? -xx : (xx >= S ? 2 * S - 2 - xx : xx);
                const float a = fminf(fmaxf(s[yy * S + xx], 0.f), max_pixel) / max_pixel;
                acc += k2[dy * ksize + dx] * a;
            }
        }
        acc = fminf(fmaxf(acc, 0.f), 1.f) * max_pixel;
        dst[i] = (float)(unsigned short)acc;
    }
}

__device__ inline unsigned aug_hash(unsigned x) {
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

// clip / max, + N(0, std), clamp to [0, 1], * max, truncate to uint16 (dataloader.py:356-368).  The standard-normal field is either
// given (noise, same shape as buf: tests, or a torch.randn tensor drawn by the caller) or generated from a per-chip seed by a
// counter hash + Box-Muller (torch's CPU randn stream cannot be replayed on the device).
__global__ __launch_bounds__(ATPB) void aug_noise_kernel(float* __restrict__ buf, const int* __restrict__ prm, const float* __restrict__ noise,
                                                         float noise_std, float max_pixel, long per_chip, long total) {
    for (long i = blockIdx.x * (long)ATPB + threadIdx.x; i < total; i += (long)gridDim.x * ATPB) {
        const int b = (int)(i / per_chip);
        if (!prm[2 * b]) continue;
        float z;
        if (noise) {
            z = noise[i];
        } else {
            const unsigned idx = (unsigned)(i - (long)b * per_chip);
            const unsigned h1 = aug_hash(idx * 2u + 0x9e3779b9u * (unsigned)prm[2 * b + 1]);
            const unsigned h2 = aug_hash(idx * 2u + 1u + 0x85ebca6bu * (unsigned)prm[2 * b + 1]);
            const float u1 = ((float)(h1 >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0, 1]
            const float u2 = (float)(h2 >> 8) * (1.0f / 16777216.0f);
            z = sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
        }
        float a = fminf(fmaxf(buf[i], 0.f), max_pixel) / max_pixel;
        a = fminf(fmaxf(a + z * noise_std, 0.f), 1.f) * max_pixel;
        buf[i] = (float)(unsigned short)a;
    }
}

}  // namespace

extern "C" {

int ig_aug_rotate(const float* src, float* dst, const float* labels_in, float* labels_out, const int* params, float fill, float label_fill,
                  int B, int CT, int S, void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(src && dst && params && src != dst, "ig_aug_rotate: null pointer or in-place call (src and dst must differ)");
    IG_REQUIRE((labels_in == nullptr) == (labels_out == nullptr), "ig_aug_rotate: labels_in and labels_out go together");
    IG_REQUIRE(B >= 0 && CT > 0 && S > 0 && S <= 4096, "ig_aug_rotate: need B >= 0, CT > 0, 0 < S <= 4096");
    const int planes = CT + (labels_in ? 1 : 0);
    const long total = (long)B * planes * S * S;
    if (total == 0) return IG_OK;
    ig_note_kernel("aug_rotate_kernel");
    hipLaunchKernelGGL(aug_rotate_kernel, dim3(aug_grid(total)), dim3(ATPB), 0, (hipStream_t)stream, src, dst, labels_in, labels_out, params,
                       fill, label_fill, CT, S, planes, total);
    return ig_check_launch("ig_aug_rotate");
}

int ig_aug_brightness_contrast(float* buf, const float* params, float max_pixel, int B, int CT, int S, void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(buf && params, "ig_aug_brightness_contrast: null pointer");
    IG_REQUIRE(B >= 0 && CT > 0 && S > 0 && (S * S) % 4 == 0, "ig_aug_brightness_contrast: S*S must be a positive multiple of 4");
    if (B == 0) return IG_OK;
    ig_note_kernel("aug_brightness_contrast_kernel");
    hipLaunchKernelGGL(aug_brightness_contrast_kernel, dim3((unsigned)(B * CT)), dim3(ATPB), 0, (hipStream_t)stream, buf, params, max_pixel, CT,
                       S * S);
    return ig_check_launch("ig_aug_brightness_contrast");
}

int ig_aug_blur(const float* src, float* dst, const int* apply, const float* kernel2d, int ksize, float max_pixel, int B, int CT, int S,
                void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(src && dst && apply && kernel2d && src != dst, "ig_aug_blur: null pointer or in-place call (src and dst must differ)");
    IG_REQUIRE(ksize > 0 && (ksize & 1) && ksize / 2 < S, "ig_aug_blur: kernel size must be odd and smaller than 2*S (got %d)", ksize);
    IG_REQUIRE(B >= 0 && CT > 0 && S > 0 && max_pixel > 0.f, "ig_aug_blur: bad sizes");
    const long total = (long)B * CT * S * S;
    if (total == 0) return IG_OK;
    ig_note_kernel("aug_blur_kernel");
    hipLaunchKernelGGL(aug_blur_kernel, dim3(aug_grid(total)), dim3(ATPB), 0, (hipStream_t)stream, src, dst, apply, kernel2d, ksize, max_pixel, CT,
                       S, total);
    return ig_check_launch("ig_aug_blur");
}

int ig_aug_noise(float* buf, const int* params, const float* noise, float noise_std, float max_pixel, int B, int CT, int S, void* stream) {
    if (B == 0) return IG_OK;  // an empty batch carries null data pointers
    IG_REQUIRE(buf && params, "ig_aug_noise: null pointer");
    IG_REQUIRE(B >= 0 && CT > 0 && S > 0 && max_pixel > 0.f, "ig_aug_noise: bad sizes");
    const long per_chip = (long)CT * S * S, total = (long)B * per_chip;
    if (total == 0) return IG_OK;
    ig_note_kernel("aug_noise_kernel");
    hipLaunchKernelGGL(aug_noise_kernel, dim3(aug_grid(total)), dim3(ATPB), 0, (hipStream_t)stream, buf, params, noise, noise_std, max_pixel,
                       per_chip, total);
    return ig_check_launch("ig_aug_noise");
}

}  // extern "C"
