// Multi-head self-attention entry points of the Prithvi ViT blocks on gfx950 (no mask, no dropout): timm Attention's
// F.scaled_dot_product_attention called from pritvhi.py:446-456.
//
// Layout: qkv [B][N][3][H][hd] bf16 (the timm reshape(B,N,3,H,hd) of the fused qkv Linear), out [B][N][H*hd]; the split (hi, lo)
// tensor pairs select the bf16x3 precision mode.  Kernels: attention2.hip (head_dim 64: 32x32x16 MFMA, whole-head K / V images in
// LDS), attention_g.hip (any head_dim that is a multiple of 16; instantiated for 80 -- the 600M variants -- and 64).  The
// first-generation kernels of round 1 (16-query tiles per wave, 9 % of the MFMA peak) were the fallback behind attention2.hip until
// round 4 and are gone: attention2.hip covers every head_dim 64 case.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HD = 64;

inline bool attn_generic_env() {  // IG_ATTN_GENERIC=1: head_dim 64 on the generic kernels (cross-check in the tests); read per call
    const char* e = getenv("IG_ATTN_GENERIC");
    return e && atoi(e) != 0;
}

}  // namespace

extern "C" {

// out[B][N][H*hd] = softmax(q k^T / sqrt(hd)) v ; lse[B][H][N] (may be NULL for inference)
int ig_attention_fwd(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, float* lse, int B, int N, int H,
                     int head_dim, void* stream) {
    IG_REQUIRE(qkv_hi && out_hi, "ig_attention_fwd: null pointer");
    IG_REQUIRE(head_dim == HD || head_dim == 80, "ig_attention_fwd: head_dim must be 64 or 80 (got %d)", head_dim);
    IG_REQUIRE((qkv_lo == nullptr) == (out_lo == nullptr), "ig_attention_fwd: split pointers must be given for all tensors or none");
    if (B == 0 || N == 0) return IG_OK;
    if (head_dim != HD || attn_generic_env())  // 600M variants (16 heads of 80), or IG_ATTN_GENERIC=1
        return ig_attention_generic_fwd(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, head_dim, stream);
    return ig_attention2_fwd(qkv_hi, qkv_lo, out_hi, out_lo, lse, B, N, H, stream);
}

// dqkv[B][N][3][H][hd] from dout, qkv, out, lse ; delta: device scratch float[B*H*N]; dqkv_colsum (optional): qkv bias gradient
int ig_attention_bwd(const void* qkv_hi, const void* qkv_lo, const void* out_hi, const void* out_lo, const void* dout_hi,
                     const void* dout_lo, const float* lse, float* delta, void* dqkv_hi, void* dqkv_lo, float* dqkv_colsum, int B, int N,
                     int H, int head_dim, void* stream) {
    IG_REQUIRE(qkv_hi && out_hi && dout_hi && lse && delta && dqkv_hi, "ig_attention_bwd: null pointer");
    IG_REQUIRE(head_dim == HD || head_dim == 80, "ig_attention_bwd: head_dim must be 64 or 80 (got %d)", head_dim);
    const bool split = qkv_lo != nullptr;
    IG_REQUIRE(split == (out_lo != nullptr) && split == (dout_lo != nullptr) && split == (dqkv_lo != nullptr),
               "ig_attention_bwd: split pointers must be given for all tensors or none");
    if (B == 0 || N == 0) return IG_OK;
    if (head_dim != HD || attn_generic_env()) {
        const int rc = ig_attention_generic_bwd(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, B, N, H, head_dim, stream);
        if (rc != IG_OK || !dqkv_colsum) return rc;
        return ig_colsum(dqkv_hi, dqkv_lo, dqkv_colsum, (long)B * N, 3 * H * head_dim, stream);
    }
    return ig_attention2_bwd(qkv_hi, qkv_lo, out_hi, out_lo, dout_hi, dout_lo, lse, delta, dqkv_hi, dqkv_lo, dqkv_colsum, B, N, H, stream);
}

}  // extern "C"
