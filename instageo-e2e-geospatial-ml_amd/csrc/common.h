// Shared device/host helpers for the instageo MI355X (gfx950) hot-path library.
// Wave = 64 lanes everywhere.  bf16 tensors are raw uint16_t in memory.
//
// "Split" tensors: in precision mode bf16x3 every bf16 activation/weight tensor is a pair
// (hi, lo) of identically shaped bf16 arrays with value = hi + lo (~16 mantissa bits).  A NULL lo
// pointer means plain bf16.  MFMA products are then formed as hi*hi + hi*lo + lo*hi (3 segments).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define IG_OK 0
#define IG_ERR_ARG -1
#define IG_ERR_HIP -2
#define IG_ERR_UNSUPPORTED -3

// host side ------------------------------------------------------------------------------------
void ig_set_error(const char* fmt, ...);
int ig_check_launch(const char* what);
// conv_direct.hip: IG_ERR_UNSUPPORTED (no error string) when the shape is not covered
int ig_conv3x3_direct(const void* x, const void* w, const float* bias, const float* bn_scale, const float* bn_shift, void* y,
                      int B, int H, int W, int Cin, int Cout, int dgrad, unsigned drop_seed, const unsigned* drop_seed_dev,
                      float drop_p, void* stream, double* stat_sums = nullptr, int* stats_fused = nullptr);
int ig_conv3x3_direct_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                            const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout, int dgrad, unsigned drop_seed,
                            const unsigned* drop_seed_dev, float drop_p, void* stream, double* stat_sums = nullptr, int* stats_fused = nullptr);
int ig_conv3x3_cls_direct(const void* x, const void* w, const float* bias, const float* bn_scale, const float* bn_shift, void* y,
                          const float* cls_w, const float* cls_b, float* logits, int B, int H, int W, int C, int ncls, void* stream);
int ig_convT_fwd_direct(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                        unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream);
// elementwise.hip (public C ABI): out[c] += sum_m x[m][c]
extern "C" int ig_colsum(const void* hi, const void* lo, float* out, long M, int C, void* stream);
int ig_colsum_ld(const void* hi, const void* lo, float* out, long M, int C, long ld, void* stream);  // first C columns, row pitch ld
int ig_convT_dgrad_direct(const void* dy, const void* w, void* dx, int B, int H, int W, int Cin, int Cout, void* stream);
int ig_convT_wgrad_direct(const void* dy, const void* x, float* dw, float* dbias, int* bias_fused, int B, int H, int W, int Cin,
                          int Cout, void* stream);
int ig_conv3x3_wgrad_direct(const void* dy, const void* x, float* dw, float* dbias, int* bias_fused, int B, int H, int W, int Cin,
                            int Cout, void* stream);

// conv8.hip: the decode head's wide ConvTranspose / Conv2d 3x3 forward and data-gradient passes as implicit GEMMs on the 8-phase
// schedule with gathering LDS-DMA.  kind 0: Conv2d 3x3 pad 1 (sign +1 forward, -1 data gradient), 1: ConvTranspose forward,
// 2: ConvTranspose data gradient.  (H, W) = row grid, C = channels of the gathered tensor, N = output channels.
// IG_ERR_UNSUPPORTED (no error string) when the shape is not covered.
int ig_conv8(int kind, int sign, const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias,
             const float* scale, const float* shift, void* y_hi, void* y_lo, int B, int H, int W, int C, int N, unsigned drop_seed,
             const unsigned* drop_seed_dev, float drop_p, void* stream);
// gemm8.hip: the 256 x 256 x 64 8-phase engine for the K-contiguous ("NT") linear GEMMs.  kind 0: bf16 (split) store of
// act(acc + bias) [+ gelu' copy]; kind 1: fp32 out = resid + acc + bias; kind 2: bf16 (split) store of acc * dact (dact_hi /
// dact_lo are INPUTS here) with optional fused column sums.  a / b: segment pointer sets (nseg 1 or 3).
struct G8Params {
    const bf16_t* a[3];
    const bf16_t* b[3];
    int nseg, M, N, K;
    long lda, ldb, ldo;
    int kind, act;
    const float* bias;
    bf16_t *out_hi, *out_lo, *dact_hi, *dact_lo;
    float* outf;
    const float* resid;
    float* colsum;
};
int ig_gemm8_nt(const G8Params& p, void* stream);  // IG_ERR_UNSUPPORTED (no error string) when the shape is not covered
// gemm4.hip: the same contract on the 4-wave (one wave per SIMD, 128 x 128 per wave) kernel with a generated K-loop: plain bf16 operands,
// N % 256 == 0, K % 128 == 0, >= 128 tiles; ig_gemm8_nt tries it first
int ig_gemm4_nt(const G8Params& p, void* stream);
// gemm8w.hip: grouped linear weight gradients dW_g += dy_g^T x_g (shared token count M) on the 8-phase schedule with transposed
// fragment reads; IG_ERR_UNSUPPORTED (no error string) when a shape is not covered (N, K multiples of 256)
int ig_wgrad8_group(int n, const void* const* dy_hi, const void* const* dy_lo, const void* const* x_hi, const void* const* x_lo,
                    float* const* dw, const int* N, const int* K, int M, int overwrite, void* stream);
// gemm8w.hip: weight gradient of nn.Conv2d(k=3, padding=1) (kind 0) / nn.ConvTranspose2d(k3,s2,p1,op1) (kind 1) on the same engine
// (gathering LDS-DMA); dWc[Cout][9][Cin] += ...; IG_ERR_UNSUPPORTED (no error string) when the shape is not covered
int ig_wgrad8_conv(int kind, const void* dy_hi, const void* dy_lo, const void* x_hi, const void* x_lo, float* dw, int B, int H, int W,
                   int Cin, int Cout, void* stream);
// runtime.hip: compute units the persistent kernels leave free (for RCCL's kernels when world > 1); ig_set_reserved_cus()
// attention2.hip: second-generation attention forward (32x32x16 MFMA, whole-head K/V in LDS); IG_ERR_UNSUPPORTED -> first generation
int ig_attention2_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H, void* stream);
int ig_attention2_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                      const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, float* dbias, int B, int N, int H,
                      void* stream);
// attention_g.hip: any head dimension that is a multiple of 16 (instantiated: 80 for the 600M variants, 64 for A/B runs)
int ig_attention_generic_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H,
                             int head_dim, void* stream);
int ig_attention_generic_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                             const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, int B, int N, int H,
                             int head_dim, void* stream);
// runtime.hip: name of the kernel an entry point launched last on this thread (rocprofv3's demangled name without namespaces and
// spaces), so bench.py can key its per-kernel roofline table by the names the rocprof summaries under profiles/ use
void ig_note_kernel(const char* fmt, ...);
void ig_note_grid(int workgroups);  // workgroups of the last persistent GEMM launch (ig_last_grid: the reserved-CU rule is testable)
int ig_reserved_cus();
int ig_cu_count();
int ig_tile_grid(int ntiles, int per_cu);

#define IG_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            ig_set_error(__VA_ARGS__); \
            return IG_ERR_ARG;         \
        }                              \
    } while (0)

static inline int ig_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Division by a launch-time constant as multiply-high + shift (dividends < 2^31): the pixel -> (b, y, x) and
// k -> (tap, channel) decodes of the convolution gathers were integer divisions per 16-byte unit.
struct FDiv {
    unsigned mul, shr, d;
    __device__ __forceinline__ int div(int n) const { return d == 1 ? n : (int)(__umulhi((unsigned)n, mul) >> shr); }
};
static inline FDiv make_fdiv(int d) {
    FDiv f{0u, 0u, (unsigned)d};
    if (d > 1) {
        unsigned lg = 0;
        while ((1u << lg) < (unsigned)d) ++lg;  // ceil(log2 d)
        const unsigned p = 31 + lg;
        f.mul = (unsigned)((((unsigned long long)1 << p) + (unsigned)d - 1) / (unsigned)d);
        f.shr = p - 32;
    }
    return f;
}

// ---- run-to-run deterministic reductions (opt-in: ig_set_deterministic) -------------------------------------------------------
// Floating-point atomics make a sum depend on the order its contributors arrive in.  In deterministic mode every multi-contributor
// reduction into the flat gradient buffer goes to a parallel int64 "shadow" of that buffer as a FIXED-POINT integer add (integer
// addition is associative, so any arrival order gives the same bits); ig_det_fold() then adds shadow * 2^-44 into the gradients and
// clears the shadow.  2^-44 = 5.7e-14 is the rounding step of one contribution, +-5.2e5 the range of one gradient element.
// Grid-wide statistics (BatchNorm sums) go through per-workgroup partials and an ordered fold instead (their magnitudes span too
// many decades for one fixed-point scale).  Targets outside the registered buffer keep the float atomic.  Each translation unit holds its own __constant__ copy of the descriptor (no relocatable device code);
// IG_DET_TU(name) defines the function runtime.hip calls to update that copy.
struct IgDet {
    long long* shadow;  // NULL: mode off
    const float* base;  // flat gradient buffer the shadow parallels
    long n;
};
static __constant__ IgDet g_igdet;
#define IG_DET_TU(name)                                                                                      \
    int ig_det_sync_##name(const IgDet* d, hipStream_t st) {                                                 \
        return hipMemcpyToSymbolAsync(HIP_SYMBOL(g_igdet), d, sizeof(IgDet), 0, hipMemcpyHostToDevice, st) == hipSuccess ? IG_OK : IG_ERR_HIP; \
    }
bool ig_deterministic();  // host-side view of the mode (runtime.hip)
// per-device grow-only scratch buffers (slot 0: BatchNorm partial sums); never freed, a superseded buffer stays allocated because
// launches in flight may still use it; NULL on allocation failure.  Not to be grown during a graph capture (first calls are warm-ups).
void* ig_scratch(int slot, size_t bytes, hipStream_t st);                   // per (device, stream, slot): see runtime.hip
void* ig_scratch2(int slot, size_t bytes, bool may_grow, hipStream_t st);  // may_grow = false: NULL instead of an allocation (stream captures)

// device side ----------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// out-of-order-safe reduction adds (see IgDet above).  The branch is wave-uniform (a constant-memory flag).
__device__ __forceinline__ void ig_red_add(float* p, float v) {
    if (g_igdet.shadow) {
        const long i = p - g_igdet.base;
        // NaN, Inf and |v| >= 5e5 do not fit the fixed-point shadow (the conversion would turn NaN into 0 and saturate the rest: a
        // diverged step would report finite gradients): they take the float atomic -- NaN / Inf absorb whatever order they arrive in
        if ((unsigned long)i < (unsigned long)g_igdet.n && fabsf(v) < 5.0e5f) {
            atomicAdd(reinterpret_cast<unsigned long long*>(g_igdet.shadow + i), (unsigned long long)__float2ll_rn(v * 17592186044416.f));
            return;
        }
    }
    atomicAdd(p, v);
}

__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;  // round-to-nearest-even, NaN preserving (v_cvt_pk_bf16_f32)
    return __builtin_bit_cast(bf16_t, b);
}

// pack two floats into one dword of 2 x bf16 (lo = a, hi = b): ONE v_cvt_pk_bf16_f32 (converting the halves separately and
// or-ing them costs two conversions plus a shift/or -- and every epilogue packs 64 values per lane)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32pair_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
    const f32pair_t f = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
}

__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
    f[0] = __uint_as_float(u.x << 16);
    f[1] = __uint_as_float(u.x & 0xffff0000u);
    f[2] = __uint_as_float(u.y << 16);
    f[3] = __uint_as_float(u.y & 0xffff0000u);
    f[4] = __uint_as_float(u.z << 16);
    f[5] = __uint_as_float(u.z & 0xffff0000u);
    f[6] = __uint_as_float(u.w << 16);
    f[7] = __uint_as_float(u.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
    uint4 u;
    u.x = pack_bf2(f[0], f[1]);
    u.y = pack_bf2(f[2], f[3]);
    u.z = pack_bf2(f[4], f[5]);
    u.w = pack_bf2(f[6], f[7]);
    return u;
}

// Load 8 consecutive logical elements (16-byte aligned) of a possibly split tensor as floats.
__device__ __forceinline__ void load8_split(const bf16_t* hi, const bf16_t* lo, size_t idx, float* f) {
    uint4 u = *reinterpret_cast<const uint4*>(hi + idx);
    unpack8(u, f);
    if (lo) {
        float g[8];
        uint4 v = *reinterpret_cast<const uint4*>(lo + idx);
        unpack8(v, g);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] += g[i];
    }
}

// Store 8 consecutive logical elements to a possibly split tensor.
__device__ __forceinline__ void store8_split(bf16_t* hi, bf16_t* lo, size_t idx, const float* f) {
    uint4 u = pack8(f);
    *reinterpret_cast<uint4*>(hi + idx) = u;
    if (lo) {
        float h[8], r[8];
        unpack8(u, h);
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = f[i] - h[i];
        *reinterpret_cast<uint4*>(lo + idx) = pack8(r);
    }
}

__device__ __forceinline__ void store4_split(bf16_t* hi, bf16_t* lo, size_t idx, const float* f) {
    uint2 u;
    u.x = pack_bf2(f[0], f[1]);
    u.y = pack_bf2(f[2], f[3]);
    *reinterpret_cast<uint2*>(hi + idx) = u;
    if (lo) {
        float r0 = f[0] - __uint_as_float(u.x << 16), r1 = f[1] - __uint_as_float(u.x & 0xffff0000u);
        float r2 = f[2] - __uint_as_float(u.y << 16), r3 = f[3] - __uint_as_float(u.y & 0xffff0000u);
        uint2 v;
        v.x = pack_bf2(r0, r1);
        v.y = pack_bf2(r2, r3);
        *reinterpret_cast<uint2*>(lo + idx) = v;
    }
}

__device__ __forceinline__ float load1_split(const bf16_t* hi, const bf16_t* lo, size_t idx) {
    float v = bf2f(hi[idx]);
    if (lo) v += bf2f(lo[idx]);
    return v;
}

__device__ __forceinline__ void store1_split(bf16_t* hi, bf16_t* lo, size_t idx, float v) {
    bf16_t h = f2bf(v);
    hi[idx] = h;
    if (lo) lo[idx] = f2bf(v - bf2f(h));
}

// wave-wide reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Counter-based dropout mask (stateless: the backward pass regenerates it from the element index, no mask tensor in
// HBM).  Elements are handled four at a time: two murmur3-finaliser hashes of (seed, idx/2) give four 16-bit uniforms;
// keep iff u16 >= p * 65536.  (A 64-bit splitmix hash per element made the ConvTranspose epilogue issue-bound.)
// idx4 must be a multiple of 4 and the tensor must have < 2^32 elements.
__device__ __forceinline__ uint32_t ig_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
__device__ __forceinline__ void dropout_scale4(uint32_t seed, uint32_t idx4, uint32_t thresh16, float inv_keep, float* m) {
    const uint32_t base = seed * 0x9E3779B1u + (idx4 >> 1);
    const uint32_t h0 = ig_fmix32(base), h1 = ig_fmix32(base + 1u);
    m[0] = (h0 & 0xffffu) >= thresh16 ? inv_keep : 0.0f;
    m[1] = (h0 >> 16) >= thresh16 ? inv_keep : 0.0f;
    m[2] = (h1 & 0xffffu) >= thresh16 ? inv_keep : 0.0f;
    m[3] = (h1 >> 16) >= thresh16 ? inv_keep : 0.0f;
}
static inline uint32_t ig_drop_thresh16(float p) { return p > 0.f ? (uint32_t)((double)p * 65536.0 + 0.5) : 0u; }

// erf by Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 1 rcp + 1 exp + 5 fma instead of the ~40-instruction
// branchy libm erff -- the GELU epilogue of the fc1 GEMM (64 values per lane) was costing as much as its K loop.
// v_rcp_f32 / v_exp_f32 are used raw (1 ulp): __frcp_rn expands to the ~10-instruction IEEE division sequence.
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float r = 1.0f - p * t * __builtin_amdgcn_exp2f(ax * ax * -1.44269504088896340736f);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }
// Two elements at a time in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: one issue slot per PAIR), GELU and optionally its
// derivative from ONE exponential: erf(x/sqrt2) = 1 - poly(t) e,  pdf(x) = e / sqrt(2 pi),  e = exp(-x^2/2).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <bool WITH_GRAD>
__device__ __forceinline__ void gelu_erf_pair(f32x2 x, f32x2& g, f32x2& dg) {
    const f32x2 a = __builtin_elementwise_abs(x) * 0.70710678118654752440f;
    const f32x2 den = a * 0.3275911f + 1.0f;
    const f32x2 t = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
    f32x2 p = t * 1.061405429f + (-1.453152027f);
    p = p * t + 1.421413741f;
    p = p * t + (-0.284496736f);
    p = p * t + 0.254829592f;
    const f32x2 na = a * a * (-1.44269504088896340736f);
    const f32x2 e = {__builtin_amdgcn_exp2f(na.x), __builtin_amdgcn_exp2f(na.y)};
    const f32x2 r = 1.0f - p * t * e;
    const f32x2 erfv = {copysignf(r.x, x.x), copysignf(r.y, x.y)};
    const f32x2 cdf = erfv * 0.5f + 0.5f;
    g = x * cdf;
    if constexpr (WITH_GRAD) dg = x * (e * 0.39894228040143267794f) + cdf;
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f));
    float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);
    return cdf + x * pdf;
}
