// Direct (halo-tile) 3x3 pad-1 convolution for the narrow last stage of the decode head (C = Cin = Cout = 48), gfx950.
//   model.py:370-375 nn.Conv2d(C, C, kernel_size=3, padding=1) of the 224 x 224 x 48 stage, forward and data gradient.
//
// Why not the implicit GEMM of gemm.hip: at 48 output channels a 256 x 48 GEMM tile is 10.6 MFLOP, and gathering its
// A operand (every input pixel is fetched for 9 taps), re-reading the 41 KiB of weights per tile, the per-tile address
// decode and the 21 k tile launches cost more than its MFMAs.  Here
//   * the whole weight tensor Wc[C][9*C] sits in LDS for the lifetime of a persistent workgroup,
//   * a workgroup walks 16 x 16-pixel output tiles; the 18 x 18 x C input halo of a tile is staged ONCE in LDS (zero-filled
//     outside the image) and all 9 taps read it from there,
//   * the reduction index is k = tap * C + c, consumed 32 at a time by v_mfma_f32_16x16x32_bf16: a lane's 8-element
//     k-group never straddles a tap (C % 8 == 0), so its operand is ONE ds_read_b128 at  pixel(row + dy, col + dx) * pitch
//     + c * 2; the (dy, dx, c) part of that address is tile-independent and precomputed per lane and K-step,
//   * pixel pitch 2C bytes and weight pitch 64 * KSTEPS + 32 bytes make every ds_read_b128 bank-conflict free,
//   * the halo of the tile after next is prefetched into registers while the current tile's MFMAs run, and those loads
//     are issued ahead of the epilogue's stores; two workgroups per CU (2 x 76 KiB of LDS),
//   * output channels are interleaved over the MFMA row blocks so that a lane stores 16 contiguous bytes; bias / folded
//     BatchNorm constants are read from LDS (a global read in the epilogue waits on vmcnt, i.e. on the draining stores).
// Measured at 108 x 224 x 224 x 48: 345-365 us against 614 us for the implicit GEMM; with the MFMAs switched off 210-260 us
// and with the stores off 215 us remain -- at 1.04 GB of compulsory traffic (216 FLOP/B, below the chip's 312) this stage is
// bound by HBM streaming, not by the matrix cores (SQ counters: MFMA busy 93 us, waves parked on waitcnt/barrier 39 %).
// Tried without gain: staggered start of the two co-resident workgroups; L2 prefetch touches 1-3 tiles ahead; a loader wave
// feeding a 3-stage LDS-DMA halo ring to 4 or 8 MFMA waves (one workgroup per CU: 364 / 330 us against 333 us) -- the
// load-only weight-gradient kernels below gain 10-50 % from such a ring, this kernel is bound by its 520 MB of stores.
// The data gradient is the same kernel over dy with the weights gathered as W'[ci][8 - tap][co] at LDS-fill time.
// Results are those of the implicit-GEMM path up to fp32 summation order (same bf16 operands, fp32 accumulation).
#include "common.h"
#include "bn_fold.h"

namespace {

// First tile of a persistent workgroup.  Workgroups are dealt round-robin over the 8 XCDs (private L2s): with t = blockIdx.x
// the eight neighbours of a tile run on eight different XCDs and every halo is fetched from HBM/MALL again (FETCH_SIZE showed
// 655 MB against 520 MB of input).  Give XCD k the k-th contiguous eighth of each round of gridDim.x tiles instead.
__device__ __forceinline__ int xcd_first_tile() {
    const int nb = gridDim.x;
    return (nb & 7) ? (int)blockIdx.x : (int)(blockIdx.x & 7) * (nb >> 3) + (int)(blockIdx.x >> 3);
}

constexpr int CD_TPB = 256;
constexpr int TH = 16, TW = 16;    // output tile
constexpr int HH = TH + 2, HW_ = TW + 2;  // halo

template <int C>
struct CDCfg {
    static constexpr int KG = 9 * C / 8;            // 8-element k-groups
    static constexpr int KSTEPS = (KG + 3) / 4;     // K-steps of 32
    static constexpr int NB = C / 16;               // 16-wide output-channel blocks
    // pitches chosen against the ds_read_b128 lane groups of gfx950 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32): with lane =
    // (k-group g, row j) a 24-dword pixel pitch and a 232-dword weight pitch are conflict-free, the 'obvious' +16-byte pads
    // (28 / 228 dwords) are 2-way (tools/lds_bank_model.py)
    static constexpr int WP = KSTEPS * 64 + 32;     // weight row pitch (bytes)
    static constexpr int PP = 2 * C;                // halo pixel pitch (bytes)
    static constexpr int W_BYTES = C * WP;
    static constexpr int H_BYTES = HH * HW_ * PP;
    static constexpr int UNITS = C / 8;              // 16-byte units per pixel
    static constexpr int HUNITS = HH * HW_ * UNITS;  // 16-byte units per halo
    static constexpr int ROUNDS = (HUNITS + CD_TPB - 1) / CD_TPB;
    static constexpr int PAR_OFF = W_BYTES + H_BYTES + 16;  // after the zero slot: bias | scale | shift, fp32 [C] each
    static constexpr int SMEM = PAR_OFF + 3 * C * 4;
    static constexpr int NPAIR = NB / 2;  // channel blocks (2p, 2p+1) are interleaved so that a lane owns 8 consecutive channels
    // MFMA row position of output channel c.  A lane (g = lane / 16) holds rows 4g..4g+3 of every 16-row block; placing
    // channels 32p + 8g + {0..3} in block 2p and 32p + 8g + {4..7} in block 2p+1 turns two 8-byte stores into one 16-byte
    // store.  An odd last block keeps the identity order.
    __host__ __device__ static constexpr int pos_of(int c) {
        return c < NPAIR * 32 ? (c / 32) * 32 + ((c % 8) / 4) * 16 + ((c % 32) / 8) * 4 + c % 4 : c;
    }
};

struct CDParams {
    const bf16_t* x;   // [B][H][W][C] input (forward) or dy (data gradient)
    const bf16_t* w;   // Wc[C][9][C]
    bf16_t* y;         // [B][H][W][C]
    const float* bias;                    // optional, forward
    const float *col_scale, *col_shift;   // optional eval-mode BatchNorm + ReLU, forward
    int B, H, W;
    int tiles_x, tiles_y;
    long ntiles;
    int dgrad;  // 1: weights gathered transposed + flipped
    uint32_t drop_seed, drop_thresh;
    const uint32_t* drop_seed_dev;
    float drop_inv;
    // optional (conv3x3_direct_kernel): per-workgroup partial sums [gridDim.x][2C] of the STORED (bf16-rounded) outputs and their
    // squares -- the statistics pass of the training-mode BatchNorm that follows, without its read of the tensor
    float* stats_part;
    // optional (conv3x3_direct_kernel<C, NCLS > 0>, inference): the 1 x 1 classifier on top of the finished pixels (eval-mode BatchNorm +
    // ReLU already folded in): logits[b][n][oy][ox] = cls_b[n] + sum_c cls_w[n][c] * v[c]; y may then be NULL (the activation is not stored)
    const float *cls_w, *cls_b;
    float* logits;
};

template <int C, int NCLS = 0>
__global__ __launch_bounds__(CD_TPB, 2) void conv3x3_direct_kernel(CDParams p) {
    using G = CDCfg<C>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    char* hal = smem + G::W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;

    // ---- weights -> LDS (once per workgroup); pad k-groups and the zero slot are cleared first
    for (int i = tid; i < G::PAR_OFF / 16; i += CD_TPB) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (!p.dgrad) {
        for (int u = tid; u < C * G::KG; u += CD_TPB) {
            const int co = u / G::KG, g = u - co * G::KG;
            *reinterpret_cast<uint4*>(wl + G::pos_of(co) * G::WP + g * 16) = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * G::KG + g) * 8);
        }
    } else {
        // W'[ci][tap'][co] = Wc[co][8 - tap'][ci]: coalesced 16-byte reads of Wc, eight 2-byte LDS scatters each
        for (int u = tid; u < C * G::KG; u += CD_TPB) {
            const int co = u / G::KG, gk = u - co * G::KG;
            const int tap = gk / G::UNITS, ci0 = (gk - tap * G::UNITS) * 8;
            const uint4 v = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * G::KG + gk) * 8);
            const uint32_t q[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                reinterpret_cast<bf16_t*>(wl + G::pos_of(ci0 + i) * G::WP)[(8 - tap) * C + co] = (bf16_t)(q[i >> 1] >> ((i & 1) * 16));
        }
    }

    // per-channel epilogue constants live in LDS: reading them from global inside the tile loop made every epilogue wait on
    // vmcnt, i.e. on the previous tile's still-draining stores
    float* par = reinterpret_cast<float*>(smem + G::PAR_OFF);
    for (int c = tid; c < C; c += CD_TPB) {
        par[c] = p.bias ? p.bias[c] : 0.f;
        par[C + c] = p.col_scale ? p.col_scale[c] : 1.f;
        par[2 * C + c] = p.col_scale ? p.col_shift[c] : 0.f;
    }
    const bool has_bn = p.col_scale != nullptr;

    // ---- tile-independent per-thread tables
    // halo fill: unit u = round * 256 + tid -> (halo pixel, 16-byte unit); LDS offset = u * 16 (pitch = UNITS * 16).  Global
    // offsets are relative to the halo's top-left pixel (ty0 - 1, tx0 - 1), i.e. non-negative: uniform base + 32-bit offset.
    static_assert(G::PP == G::UNITS * 16, "halo units must be contiguous in LDS");
    int h_goff[G::ROUNDS], h_yx[G::ROUNDS];
#pragma unroll
    for (int r = 0; r < G::ROUNDS; ++r) {
        const int u = r * CD_TPB + tid;
        const int hp = u / G::UNITS, c8 = u - hp * G::UNITS;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        h_goff[r] = (hy * p.W + hx) * C + c8 * 8;
        h_yx[r] = u < G::HUNITS ? ((hy - 1) << 16) | ((hx - 1) & 0xffff) : 0x7fff0000;  // sentinel row: never inside the image
    }
    const bool last_round_on = (G::ROUNDS - 1) * CD_TPB + tid < G::HUNITS;
    // operand reads: pixel-side address of this lane's k-group per K-step (relative to the px-block's first pixel)
    const int g = lane >> 4, j = lane & 15;
    int offk[G::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
        const int kg = ks * 4 + g;
        const int tap = kg / G::UNITS, cg = kg - tap * G::UNITS;
        const int dy = tap / 3, dx = tap - dy * 3;
        offk[ks] = (dy * HW_ + dx) * G::PP + cg * 16;
    }
    const int pix_lane = ((wave * 4) * HW_ + j) * G::PP;              // px-block 0 of this wave, pixel j
    const int zero_slot = G::H_BYTES;                                  // relative to hal
    const char* wl_lane = wl + j * G::WP + g * 16;

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        ty0 = ty * TH, tx0 = (r - ty * p.tiles_x) * TW;
    };
    uint4 pre[G::ROUNDS];
    auto fetch = [&](int t) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const bf16_t* base = p.x + (((long)b * p.H + ty0 - 1) * p.W + tx0 - 1) * C;  // halo origin (may lie outside: masked)
        if (ty0 >= 1 && tx0 >= 1 && ty0 + TH < p.H && tx0 + TW < p.W) {
            // interior tile (3 of 4 at 224 x 224): the whole halo is inside the image, no per-unit bounds tests
#pragma unroll
            for (int r = 0; r < G::ROUNDS - 1; ++r) pre[r] = *reinterpret_cast<const uint4*>(base + (unsigned)h_goff[r]);
            pre[G::ROUNDS - 1] = make_uint4(0, 0, 0, 0);
            if (last_round_on) pre[G::ROUNDS - 1] = *reinterpret_cast<const uint4*>(base + (unsigned)h_goff[G::ROUNDS - 1]);
        } else {
#pragma unroll
            for (int r = 0; r < G::ROUNDS; ++r) {
                const int gy = ty0 + (h_yx[r] >> 16), gx = tx0 + (int)(short)(h_yx[r] & 0xffff);
                const bool ok = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
                pre[r] = make_uint4(0, 0, 0, 0);
                if (ok) pre[r] = *reinterpret_cast<const uint4*>(base + h_goff[r]);
            }
        }
    };

    auto halo_to_lds = [&]() {
#pragma unroll
        for (int r = 0; r < G::ROUNDS; ++r)
            if (r * CD_TPB + tid < G::HUNITS) *reinterpret_cast<uint4*>(hal + (r * CD_TPB + tid) * 16) = pre[r];
    };
    // one output group = 4 consecutive channels starting at `n` of pixel `pix`, finished and stored (8 bytes) -- or, for an
    // interleaved block pair, 8 consecutive channels (16 bytes)
    auto finish4 = [&](f32x4 a, int n, size_t idx, float* v) {
        const float4 bb = *reinterpret_cast<const float4*>(par + n);
        v[0] = a[0] + bb.x, v[1] = a[1] + bb.y, v[2] = a[2] + bb.z, v[3] = a[3] + bb.w;
        if (has_bn) {
            const float4 sc = *reinterpret_cast<const float4*>(par + C + n);
            const float4 sh = *reinterpret_cast<const float4*>(par + 2 * C + n);
            v[0] = fmaxf(v[0] * sc.x + sh.x, 0.f), v[1] = fmaxf(v[1] * sc.y + sh.y, 0.f);
            v[2] = fmaxf(v[2] * sc.z + sh.z, 0.f), v[3] = fmaxf(v[3] * sc.w + sh.w, 0.f);
        }
        if (p.drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, p.drop_thresh, p.drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
    };

    // BatchNorm statistics of this lane's channels (fixed over the tiles: 8 per block pair + 4 of an unpaired last block)
    constexpr int NS = G::NPAIR * 8 + (G::NB & 1) * 4;
    float st_s[NS], st_q[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st_s[k] = 0.f, st_q[k] = 0.f;
    const bool want_stats = p.stats_part != nullptr;
    // classifier weights of this lane's channels (registers: NCLS <= 2)
    float cw[NCLS > 0 ? NCLS : 1][NS];
    if constexpr (NCLS > 0) {
#pragma unroll
        for (int n = 0; n < NCLS; ++n)
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int c = k < G::NPAIR * 8 ? (k / 8) * 32 + 8 * g + k % 8 : (G::NB - 1) * 16 + 4 * g + (k - G::NPAIR * 8);
                cw[n][k] = p.cls_w[n * C + c];
            }
    }
    // output element offset of this lane's pixel (row 4*wave of the tile, column j) relative to the tile origin, + channel 8g
    const int e_lane = ((wave * 4) * p.W + j) * C;
    const int nt = (int)p.ntiles, gstep = (int)gridDim.x;
    int t = xcd_first_tile();
    if (t < nt) {
        fetch(t);
        halo_to_lds();
        if (t + gstep < nt) fetch(t + gstep);
    }
    __syncthreads();
    for (; t < nt; t += gstep) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const int tn = t + gstep, tnn = tn + gstep;

        // ---- 4 px-blocks (rows 4*wave .. +3 of the tile) x NB channel blocks per wave
        f32x4 acc[G::NB][4];
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            bf16x8_t wf[G::NB], pf[4];
            const bool padded = G::KG % 4 != 0 && ks == G::KSTEPS - 1 && g >= G::KG % 4;  // these k-groups read zeros
            const int po = padded ? zero_slot : pix_lane + offk[ks];
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb) wf[nb] = *reinterpret_cast<const bf16x8_t*>(wl_lane + nb * 16 * G::WP + ks * 64);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) pf[mb] = *reinterpret_cast<const bf16x8_t*>(hal + (padded ? po : po + mb * HW_ * G::PP));
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], pf[mb], acc[nb][mb], 0, 0, 0);
        }

        // ---- the halo is free once every wave has finished its reads: stage the next one, then put the loads of the
        // tile after it in flight BEFORE this tile's stores -- the memory pipeline serves a wave's requests in order, and
        // loads queued behind 12 stores per lane did not land within one tile's MFMAs
        __syncthreads();
        if (tn < nt) halo_to_lds();
        __syncthreads();
        if (tnn < nt) fetch(tnn);

        // ---- epilogue (its stores drain under the next tile's MFMAs): lane holds, of pixel (row 4*wave + mb, column j),
        // channels 32p + 8g .. +7 from block pair p and channels 16*nb + 4g .. +3 from an unpaired last block
        const int ox = tx0 + j;
        const size_t origin = (((size_t)b * p.H + ty0) * p.W + tx0) * C;  // uniform: first element of the output tile
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int oy = ty0 + wave * 4 + mb;
            const bool inside = oy < p.H && ox < p.W;
            float lg[NCLS > 0 ? NCLS : 1];
#pragma unroll
            for (int n = 0; n < (NCLS > 0 ? NCLS : 1); ++n) lg[n] = 0.f;
            if (inside) {
                const size_t pixc = origin + (unsigned)(e_lane + mb * p.W * C);  // element index of channel 0 of the pixel
#pragma unroll
                for (int pr = 0; pr < G::NPAIR; ++pr) {
                    const int n = pr * 32 + 8 * g;
                    const size_t idx = pixc + n;
                    float v[8];
                    finish4(acc[2 * pr][mb], n, idx, v);
                    finish4(acc[2 * pr + 1][mb], n + 4, idx + 4, v + 4);
                    const uint4 pk = pack8(v);
                    if (NCLS == 0 || p.y) *reinterpret_cast<uint4*>(p.y + idx) = pk;
                    if (want_stats) {
                        float r[8];
                        unpack8(pk, r);
#pragma unroll
                        for (int i = 0; i < 8; ++i) st_s[pr * 8 + i] += r[i], st_q[pr * 8 + i] = fmaf(r[i], r[i], st_q[pr * 8 + i]);
                    }
                    if constexpr (NCLS > 0) {
#pragma unroll
                        for (int n2 = 0; n2 < NCLS; ++n2)
#pragma unroll
                            for (int i = 0; i < 8; ++i) lg[n2] = fmaf(v[i], cw[n2][pr * 8 + i], lg[n2]);
                    }
                }
                if (G::NB & 1) {
                    const int n = (G::NB - 1) * 16 + 4 * g;
                    const size_t idx = pixc + n;
                    float v[4];
                    finish4(acc[G::NB - 1][mb], n, idx, v);
                    if (NCLS == 0 || p.y) store4_split(p.y, nullptr, idx, v);
                    if (want_stats) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float r = bf2f(f2bf(v[i]));
                            st_s[G::NPAIR * 8 + i] += r, st_q[G::NPAIR * 8 + i] = fmaf(r, r, st_q[G::NPAIR * 8 + i]);
                        }
                    }
                    if constexpr (NCLS > 0) {
#pragma unroll
                        for (int n2 = 0; n2 < NCLS; ++n2)
#pragma unroll
                            for (int i = 0; i < 4; ++i) lg[n2] = fmaf(v[i], cw[n2][G::NPAIR * 8 + i], lg[n2]);
                    }
                }
            }
            if constexpr (NCLS > 0) {  // the four k-group lanes of a pixel (lane = 16 g + j) hold its 48 channels: fold, g == 0 stores
#pragma unroll
                for (int n2 = 0; n2 < NCLS; ++n2) {
                    float t = lg[n2];
                    t += __shfl_xor(t, 16, 64);
                    t += __shfl_xor(t, 32, 64);
                    if (g == 0 && inside) p.logits[((size_t)b * NCLS + n2) * p.H * p.W + (size_t)oy * p.W + ox] = t + p.cls_b[n2];
                }
            }
        }
    }
    if (want_stats) {  // lanes j of a k-group -> wave -> workgroup, all in a fixed order (the tile schedule is static too)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) st_s[k] += __shfl_xor(st_s[k], o, 64), st_q[k] += __shfl_xor(st_q[k], o, 64);
        }
        __syncthreads();  // the halo buffer is dead
        float* red = reinterpret_cast<float*>(hal);  // [wave][g][NS][2]
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) red[((wave * 4 + g) * NS + k) * 2] = st_s[k], red[((wave * 4 + g) * NS + k) * 2 + 1] = st_q[k];
        }
        __syncthreads();
        for (int c = tid; c < C; c += CD_TPB) {
            int gg, k;
            if (c < G::NPAIR * 32) gg = (c % 32) / 8, k = (c / 32) * 8 + c % 8;
            else gg = (c - (G::NB - 1) * 16) / 4, k = G::NPAIR * 8 + (c - (G::NB - 1) * 16) % 4;
            float ss = 0.f, qq = 0.f;
            for (int w = 0; w < CD_TPB / 64; ++w) ss += red[((w * 4 + gg) * NS + k) * 2], qq += red[((w * 4 + gg) * NS + k) * 2 + 1];
            p.stats_part[(size_t)blockIdx.x * 2 * C + c] = ss;
            p.stats_part[(size_t)blockIdx.x * 2 * C + C + c] = qq;
        }
    }
}

template <int C, int NCLS = 0>
int launch_direct(CDParams p, hipStream_t st, const char* what, double* stat_sums = nullptr) {
    using G = CDCfg<C>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_direct_kernel<C, NCLS>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
        attr_done = true;
    }
    long nwg = 512;  // two persistent workgroups per CU
    if (nwg > p.ntiles) nwg = p.ntiles;
    if (stat_sums) {
        p.stats_part = (float*)ig_scratch(0, (size_t)nwg * 2 * C * sizeof(float), st);
        if (!p.stats_part) {
            ig_set_error("%s: scratch allocation failed", what);
            return IG_ERR_HIP;
        }
    }
    ig_note_kernel("conv3x3_direct_kernel<%d,%d>", C, NCLS);
    hipLaunchKernelGGL((conv3x3_direct_kernel<C, NCLS>), dim3((unsigned)nwg), dim3(CD_TPB), G::SMEM, st, p);
    if (stat_sums) hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * C, 64)), dim3(1024), 0, st, p.stats_part, stat_sums, (int)nwg, 2 * C);
    return ig_check_launch(what);
}

// ---- split precision (bf16x3) form of conv3x3_direct_kernel, C = 48 -----------------------------------------------------------------
// The split mode ran this stage on the implicit GEMM (2.4-2.7 ms per launch at 216 x 224 x 224 x 48: a 48-wide output fills 37 % of its
// 128-wide tiles, and the three operand pairs were three passes).  Here the hi AND lo images of the weights (2 x 43.5 KiB) and of the
// halo (2 x 30.4 KiB) sit in LDS -- 148 KiB, one 8-wave workgroup per CU, a wave owns 2 rows of the 16 x 16 tile -- and every fragment
// pair feeds three MFMAs (hi hi, lo(w) hi(x), hi(w) lo(x); lo lo is dropped as in every split GEMM here); the epilogue stores hi and lo.
// Everything else (halo prefetch through registers two tiles ahead, channel interleave for 16-byte stores, fused BatchNorm statistics of
// the STORED value hi + lo, the data-gradient form with the weights gathered transposed and flipped) follows the plain kernel.
struct CDSplit {
    const bf16_t *x_lo, *w_lo;
    bf16_t* y_lo;
};

template <int C>
__global__ __launch_bounds__(512, 1) void conv3x3_direct_split_kernel(CDParams p, CDSplit q) {
    using G = CDCfg<C>;
    constexpr int TPB = 512, NWV = 8, RPW = TH / NWV;
    constexpr int ROUNDS = (G::HUNITS + TPB - 1) / TPB;
    constexpr int HSTR = G::H_BYTES + 16;  // halo image + its zero slot
    constexpr int PAR_OFF = 2 * G::W_BYTES + 2 * HSTR;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl0 = smem;
    char* wl1 = smem + G::W_BYTES;
    char* hal0 = smem + 2 * G::W_BYTES;
    char* hal1 = hal0 + HSTR;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;

    for (int i = tid; i < PAR_OFF / 16; i += TPB) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const bf16_t* wsrc = half ? q.w_lo : p.w;
        char* wl = half ? wl1 : wl0;
        if (!p.dgrad) {
            for (int u = tid; u < C * G::KG; u += TPB) {
                const int co = u / G::KG, g = u - co * G::KG;
                *reinterpret_cast<uint4*>(wl + G::pos_of(co) * G::WP + g * 16) = *reinterpret_cast<const uint4*>(wsrc + ((size_t)co * G::KG + g) * 8);
            }
        } else {  // W'[ci][tap'][co] = Wc[co][8 - tap'][ci]
            for (int u = tid; u < C * G::KG; u += TPB) {
                const int co = u / G::KG, gk = u - co * G::KG;
                const int tap = gk / G::UNITS, ci0 = (gk - tap * G::UNITS) * 8;
                const uint4 v = *reinterpret_cast<const uint4*>(wsrc + ((size_t)co * G::KG + gk) * 8);
                const uint32_t qv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    reinterpret_cast<bf16_t*>(wl + G::pos_of(ci0 + i) * G::WP)[(8 - tap) * C + co] = (bf16_t)(qv[i >> 1] >> ((i & 1) * 16));
            }
        }
    }
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    for (int c = tid; c < C; c += TPB) {
        par[c] = p.bias ? p.bias[c] : 0.f;
        par[C + c] = p.col_scale ? p.col_scale[c] : 1.f;
        par[2 * C + c] = p.col_scale ? p.col_shift[c] : 0.f;
    }
    const bool has_bn = p.col_scale != nullptr;

    int h_goff[ROUNDS], h_yx[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int u = r * TPB + tid;
        const int hp = u / G::UNITS, c8 = u - hp * G::UNITS;
        const int hy = hp / HW_, hx = hp - hy * HW_;
        h_goff[r] = (hy * p.W + hx) * C + c8 * 8;
        h_yx[r] = u < G::HUNITS ? ((hy - 1) << 16) | ((hx - 1) & 0xffff) : 0x7fff0000;
    }
    const int g = lane >> 4, j = lane & 15;
    int offk[G::KSTEPS];
#pragma unroll
    for (int ks = 0; ks < G::KSTEPS; ++ks) {
        const int kg = ks * 4 + g;
        const int tap = kg / G::UNITS, cg = kg - tap * G::UNITS;
        const int dy = tap / 3, dx = tap - dy * 3;
        offk[ks] = (dy * HW_ + dx) * G::PP + cg * 16;
    }
    const int pix_lane = ((wave * RPW) * HW_ + j) * G::PP;
    const int zero_slot = G::H_BYTES;
    const int w_lane = j * G::WP + g * 16;

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        ty0 = ty * TH, tx0 = (r - ty * p.tiles_x) * TW;
    };
    uint4 pre0[ROUNDS], pre1[ROUNDS];
    auto fetch = [&](int t) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const long org = (((long)b * p.H + ty0 - 1) * p.W + tx0 - 1) * C;  // halo origin (may lie outside: masked)
        const bf16_t *b0 = p.x + org, *b1 = q.x_lo + org;
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const int gy = ty0 + (h_yx[r] >> 16), gx = tx0 + (int)(short)(h_yx[r] & 0xffff);
            const bool ok = ((unsigned)gy < (unsigned)p.H) & ((unsigned)gx < (unsigned)p.W);
            pre0[r] = pre1[r] = make_uint4(0, 0, 0, 0);
            if (ok) pre0[r] = *reinterpret_cast<const uint4*>(b0 + h_goff[r]), pre1[r] = *reinterpret_cast<const uint4*>(b1 + h_goff[r]);
        }
    };
    auto halo_to_lds = [&]() {
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r)
            if (r * TPB + tid < G::HUNITS) {
                *reinterpret_cast<uint4*>(hal0 + (r * TPB + tid) * 16) = pre0[r];
                *reinterpret_cast<uint4*>(hal1 + (r * TPB + tid) * 16) = pre1[r];
            }
    };
    auto finish4 = [&](f32x4 a, int n, size_t idx, float* v) {
        const float4 bb = *reinterpret_cast<const float4*>(par + n);
        v[0] = a[0] + bb.x, v[1] = a[1] + bb.y, v[2] = a[2] + bb.z, v[3] = a[3] + bb.w;
        if (has_bn) {
            const float4 sc = *reinterpret_cast<const float4*>(par + C + n);
            const float4 sh = *reinterpret_cast<const float4*>(par + 2 * C + n);
            v[0] = fmaxf(v[0] * sc.x + sh.x, 0.f), v[1] = fmaxf(v[1] * sc.y + sh.y, 0.f);
            v[2] = fmaxf(v[2] * sc.z + sh.z, 0.f), v[3] = fmaxf(v[3] * sc.w + sh.w, 0.f);
        }
        if (p.drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, p.drop_thresh, p.drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
    };
    constexpr int NS = G::NPAIR * 8 + (G::NB & 1) * 4;
    float st_s[NS], st_q[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st_s[k] = 0.f, st_q[k] = 0.f;
    const bool want_stats = p.stats_part != nullptr;
    const int e_lane = ((wave * RPW) * p.W + j) * C;
    const int nt = (int)p.ntiles, gstep = (int)gridDim.x;
    int t = xcd_first_tile();
    if (t < nt) {
        fetch(t);
        halo_to_lds();
        if (t + gstep < nt) fetch(t + gstep);
    }
    __syncthreads();
    for (; t < nt; t += gstep) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const int tn = t + gstep, tnn = tn + gstep;
        f32x4 acc[G::NB][RPW];
#pragma unroll
        for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < RPW; ++mb) acc[nb][mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KSTEPS; ++ks) {
            bf16x8_t wf0[G::NB], wf1[G::NB], pf0[RPW], pf1[RPW];
            const bool padded = G::KG % 4 != 0 && ks == G::KSTEPS - 1 && g >= G::KG % 4;  // these k-groups read zeros
            const int po = padded ? zero_slot : pix_lane + offk[ks];
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb) {
                wf0[nb] = *reinterpret_cast<const bf16x8_t*>(wl0 + w_lane + nb * 16 * G::WP + ks * 64);
                wf1[nb] = *reinterpret_cast<const bf16x8_t*>(wl1 + w_lane + nb * 16 * G::WP + ks * 64);
            }
#pragma unroll
            for (int mb = 0; mb < RPW; ++mb) {
                pf0[mb] = *reinterpret_cast<const bf16x8_t*>(hal0 + (padded ? po : po + mb * HW_ * G::PP));
                pf1[mb] = *reinterpret_cast<const bf16x8_t*>(hal1 + (padded ? po : po + mb * HW_ * G::PP));
            }
#pragma unroll
            for (int nb = 0; nb < G::NB; ++nb)
#pragma unroll
                for (int mb = 0; mb < RPW; ++mb) {
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nb], pf0[mb], acc[nb][mb], 0, 0, 0);
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf1[nb], pf0[mb], acc[nb][mb], 0, 0, 0);
                    acc[nb][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[nb], pf1[mb], acc[nb][mb], 0, 0, 0);
                }
        }
        __syncthreads();
        if (tn < nt) halo_to_lds();
        __syncthreads();
        if (tnn < nt) fetch(tnn);

        const int ox = tx0 + j;
        const size_t origin = (((size_t)b * p.H + ty0) * p.W + tx0) * C;
#pragma unroll
        for (int mb = 0; mb < RPW; ++mb) {
            const int oy = ty0 + wave * RPW + mb;
            if (oy < p.H && ox < p.W) {
                const size_t pixc = origin + (unsigned)(e_lane + mb * p.W * C);
#pragma unroll
                for (int pr = 0; pr < G::NPAIR; ++pr) {
                    const int n = pr * 32 + 8 * g;
                    const size_t idx = pixc + n;
                    float v[8], r[8], d[8];
                    finish4(acc[2 * pr][mb], n, idx, v);
                    finish4(acc[2 * pr + 1][mb], n + 4, idx + 4, v + 4);
                    const uint4 pk = pack8(v);
                    unpack8(pk, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) d[i] = v[i] - r[i];
                    const uint4 pl = pack8(d);
                    *reinterpret_cast<uint4*>(p.y + idx) = pk;
                    *reinterpret_cast<uint4*>(q.y_lo + idx) = pl;
                    if (want_stats) {
                        unpack8(pl, d);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float sv = r[i] + d[i];
                            st_s[pr * 8 + i] += sv, st_q[pr * 8 + i] = fmaf(sv, sv, st_q[pr * 8 + i]);
                        }
                    }
                }
                if (G::NB & 1) {
                    const int n = (G::NB - 1) * 16 + 4 * g;
                    const size_t idx = pixc + n;
                    float v[4];
                    finish4(acc[G::NB - 1][mb], n, idx, v);
                    store4_split(p.y, q.y_lo, idx, v);
                    if (want_stats) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float hi = bf2f(f2bf(v[i])), sv = hi + bf2f(f2bf(v[i] - hi));
                            st_s[G::NPAIR * 8 + i] += sv, st_q[G::NPAIR * 8 + i] = fmaf(sv, sv, st_q[G::NPAIR * 8 + i]);
                        }
                    }
                }
            }
        }
    }
    if (want_stats) {  // lanes j of a k-group -> wave -> workgroup, all in a fixed order (the tile schedule is static too)
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) st_s[k] += __shfl_xor(st_s[k], o, 64), st_q[k] += __shfl_xor(st_q[k], o, 64);
        }
        __syncthreads();  // the halo buffers are dead
        float* red = reinterpret_cast<float*>(hal0);  // [wave][g][NS][2]
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) red[((wave * 4 + g) * NS + k) * 2] = st_s[k], red[((wave * 4 + g) * NS + k) * 2 + 1] = st_q[k];
        }
        __syncthreads();
        for (int c = tid; c < C; c += TPB) {
            int gg, k;
            if (c < G::NPAIR * 32) gg = (c % 32) / 8, k = (c / 32) * 8 + c % 8;
            else gg = (c - (G::NB - 1) * 16) / 4, k = G::NPAIR * 8 + (c - (G::NB - 1) * 16) % 4;
            float ss = 0.f, qq = 0.f;
            for (int w = 0; w < NWV; ++w) ss += red[((w * 4 + gg) * NS + k) * 2], qq += red[((w * 4 + gg) * NS + k) * 2 + 1];
            p.stats_part[(size_t)blockIdx.x * 2 * C + c] = ss;
            p.stats_part[(size_t)blockIdx.x * 2 * C + C + c] = qq;
        }
    }
}

template <int C>
int launch_direct_split(CDParams p, CDSplit q, hipStream_t st, const char* what, double* stat_sums) {
    using G = CDCfg<C>;
    constexpr int smem = 2 * G::W_BYTES + 2 * (G::H_BYTES + 16) + 3 * C * 4;
    static_assert(smem <= 160 * 1024, "hi + lo images must fit the LDS");
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_direct_split_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    long nwg = ig_cu_count();  // one persistent workgroup per CU
    if (nwg > p.ntiles) nwg = p.ntiles;
    if (stat_sums) {
        p.stats_part = (float*)ig_scratch(0, (size_t)nwg * 2 * C * sizeof(float), st);
        if (!p.stats_part) {
            ig_set_error("%s: scratch allocation failed", what);
            return IG_ERR_HIP;
        }
    }
    ig_note_kernel("conv3x3_direct_split_kernel<%d>", C);
    hipLaunchKernelGGL((conv3x3_direct_split_kernel<C>), dim3((unsigned)nwg), dim3(512), smem, st, p, q);
    if (stat_sums) hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * C, 64)), dim3(1024), 0, st, p.stats_part, stat_sums, (int)nwg, 2 * C);
    return ig_check_launch(what);
}

// ---------------------------------------------------------------------------------------------- weight gradient

// dWc[co][tap][ci] += sum_pixels dy[p][co] * x[p + off(tap)][ci] for the 48- and 96-channel stages.  The implicit GEMM needs
// a split-K grid with an atomic pass per split and gathers x nine times (843 us at 108 x 224 x 224 x 48); here a persistent
// workgroup keeps a 48 x 9*CIN partial sum in registers (3 x 7 MFMA accumulators per wave; 4 waves for CIN = 48, 8 for 96)
// over ALL its tiles and adds it to dWc once at the end.  A workgroup covers a 48-wide slice of the output channels
// (blockIdx.y); with Cout = 96 the x halo is therefore staged by two workgroups.  Per 16 x 16 tile the x halo and the dy
// tile are staged in LDS ([pixel][channel]); the reduction runs over the tile's 256 pixels, so both operands are k-strided
// and read with ds_read_b64_tr_b16.  The k order inside a K-step (two tile rows) is chosen so that the 8 k-rows a 32-lane
// half reads together are 8 CONSECUTIVE pixels: with a pixel pitch = 8 (mod 16) dwords they fall on 8 distinct 32-byte bank
// windows (k-rows 8 apart would collide whatever the pitch).
struct CWParams {
    const bf16_t* x;   // [B][H][W][CIN] conv input
    const bf16_t* dy;  // [B][H][W][Cout] gradient of the conv output
    float* dw;         // [Cout][9][CIN] fp32, accumulated
    float* dbias;      // optional [Cout]: += column sums of dy (DMA-ring kernels only)
    int B, H, W, Cout;
    int tiles_x, tiles_y;
    long ntiles;
};

// ---------------------------------------------------------------------------------------------- ConvTranspose2d forward
// nn.ConvTranspose2d(k=3, s=2, p=1, op=1) of the last stage (112 x 112 x 96 -> 224 x 224 x 48; model.py:361-368), weights
// Wc[Cout][9][Cin].  Output (2iy+py, 2ix+px) reads tap rows ky = 1 (py = 0, input row iy) or ky = 0 / 2 (py = 1, input rows
// iy+1 / iy), the same along x: every tap belongs to exactly one of the four sub-pixel phases, so one sweep over the 9 taps
// x Cin of the LDS-resident weights produces all four phases of a block of 16 input pixels.
// One 8-wave workgroup per CU: a 16 x 16 input tile (+1 row/column of halo, zero-filled outside) in LDS, each wave owns two
// rows of 16 input pixels = 2 x 4 phases x 3 channel blocks of accumulators; next-but-one tile prefetched in registers.
// The stage moves 780 MB for 112 GFLOP (144 FLOP/B): it is HBM-bound, the implicit GEMM ran it at 607 us.
constexpr int CT_TPB = 512;
constexpr int CT_T = 16;          // input tile edge
constexpr int CT_P = CT_T + 1;    // patch edge

struct CTParams {
    const bf16_t* x;   // [B][H][W][CIN]
    const bf16_t* w;   // Wc[COUT][9][CIN]
    bf16_t* y;         // [B][2H][2W][COUT]
    const float* bias;
    int B, H, W;
    int tiles_x, tiles_y;
    long ntiles;
    uint32_t drop_seed, drop_thresh;
    const uint32_t* drop_seed_dev;
    float drop_inv;
};

// ---------------------------------------------------------------------------------------------- ConvTranspose2d weight gradient
// dWc[co][tap][ci] += sum over base pixels (iy, ix) of dy[2iy + py(tap)][2ix + px(tap)][co] * x[iy + (ky==0)][ix + (kx==0)][ci]
// for the 96 -> 48 stage (the implicit GEMM ran one split-K GEMM per tap: 440 us).  Same scheme as the 3x3 weight gradient:
// register-resident 48 x 864 partial sum per persistent workgroup, one atomic pass at the end.  Six waves: wave w owns the
// input-channel block ci = 16w .. 16w+15 of ALL nine taps (9 x 3 accumulators), so the dy fragments of the four sub-pixel
// phases (4 x 3 per K-step) are shared by its nine taps and the code is identical for every wave.  Tile = 8 x 16 base pixels:
// x patch 9 x 17 (zero-filled outside), dy tile 16 x 32 output pixels stored as four phase planes of 8 x 16 pixels so that
// the 8 pixels a 32-lane half reads together are consecutive (see the 3x3 weight gradient).
constexpr int TW_TPB = 384;
constexpr int TWH = 8, TWW = 16;  // base-pixel tile

struct CTWParams {
    const bf16_t* x;   // [B][H][W][CIN]
    const bf16_t* dy;  // [B][2H][2W][COUT]
    float* dw;         // [COUT][9][CIN]
    float* dbias;      // optional [COUT]: += column sums of dy (DMA-ring kernel only)
    int B, H, W;
    int tiles_x, tiles_y;
    long ntiles;
};

// ---- LDS-DMA variant of the ConvTranspose weight gradient ------------------------------------------------------
// The register-prefetch kernel of round 2 (removed in round 5) kept one tile (83 KB) of loads in flight per CU and nothing while it computed: one
// workgroup per CU reached 2 TB/s (the two-workgroup 3x3 kernels reach 4).  Here the tile is 4 x 16 base pixels (48 KiB
// stage, both images lane-linear so that global_load_lds_dwordx4 can fill them: x patch 5 x 17 pixels x 224 B, dy as four
// phase planes of 4 x 16 pixels x 96 B) and a 3-stage ring keeps two tiles in flight while a third is consumed -- the
// protocol of gemm2_kernel: every wave issues exactly 8 DMAs per tile, waits for its own with a counted vmcnt, one barrier
// per tile covers the other waves' pieces (RAW) and the slot that is refilled next (WAR).  No global stores in the loop, so
// the vmcnt count is exact.
__device__ __forceinline__ void cd_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    lds_dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}

template <int CIN, int COUT>
__global__ __launch_bounds__(TW_TPB, 1) void convT_wgrad_dma_kernel(CTWParams p, const bf16_t* zero_page) {
    static_assert(CIN == 96 && COUT == 48, "unit tables below are written for 96 -> 48");
    constexpr int T4 = 4;                       // base-pixel rows per tile
    constexpr int PPX = 2 * CIN + 32, PPD = 2 * COUT;
    constexpr int PH = T4 + 1, PW = TWW + 1;
    constexpr int XUP = PPX / 16;               // 14 units per patch pixel (12 data + 2 pad)
    constexpr int XUNITS = PH * PW * XUP;       // 1190
    constexpr int XSLOTS = (XUNITS + 63) / 64;  // 19 wave-DMAs
    constexpr int PLANE = T4 * TWW * PPD;       // 6144 B = 384 units
    constexpr int DUNITS = 4 * PLANE / 16;      // 1536
    constexpr int DSLOTS = DUNITS / 64;         // 24
    constexpr int NW = TW_TPB / 64;             // 6 waves
    constexpr int PER_WAVE = (XSLOTS + DSLOTS + NW - 1) / NW;  // 8
    constexpr int STAGE = PER_WAVE * NW * 1024; // 48 KiB (slots beyond 43 are scratch)
    constexpr int D_OFF = XSLOTS * 1024;
    constexpr int CB = COUT / 16;
    constexpr int KS = T4 * TWW / 32;           // 2 K-steps per tile
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;

    // per-lane unit tables: element offset from the tile origin of its image, image kind, pixel offset for the bounds test
    int u_off[PER_WAVE], u_meta[PER_WAVE];  // meta: 0 = always zero page; else (kind << 16) | (dy << 8) | dx, kind 1 = x, 2 = dy
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int slot = i * NW + wave, u = slot * 64 + lane;
        u_off[i] = 0, u_meta[i] = 0;
        if (slot < XSLOTS) {
            const int hp = u / XUP, c = u - hp * XUP;
            const int hy = hp / PW, hx = hp - hy * PW;
            if (u < XUNITS && c < CIN / 8) u_off[i] = (hy * p.W + hx) * CIN + c * 8, u_meta[i] = (1 << 16) | (hy << 8) | hx;
        } else if (slot < XSLOTS + DSLOTS) {
            const int ud = u - XSLOTS * 64;
            const int plane = ud / (PLANE / 16), rem = ud - plane * (PLANE / 16);
            const int pix = rem / (COUT / 8), c8 = rem - pix * (COUT / 8);
            const int iy = pix / TWW, ix = pix - iy * TWW;
            u_off[i] = ((2 * iy + (plane >> 1)) * 2 * p.W + 2 * ix + (plane & 1)) * COUT + c8 * 8;
            u_meta[i] = (2 << 16) | (iy << 8) | ix;
        }
    }
    auto tile_coords = [&](long t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = (int)(t / per_img);
        const int r = (int)(t - (long)b * per_img);
        const int ty = r / p.tiles_x;
        ty0 = ty * T4, tx0 = (r - ty * p.tiles_x) * TWW;
    };
    auto issue = [&](long t, int slot3) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const bf16_t* xb = p.x + (((size_t)b * p.H + ty0) * p.W + tx0) * CIN;
        const bf16_t* db = p.dy + (((size_t)b * 2 * p.H + 2 * ty0) * (2 * p.W) + 2 * tx0) * COUT;
        const unsigned sbase = lds_base + slot3 * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            const int m = u_meta[i];
            const bool ok = (m != 0) & (ty0 + ((m >> 8) & 0xff) < p.H) & (tx0 + (m & 0xff) < p.W);
            const bf16_t* src = ((m >> 16) == 1 ? xb : db) + u_off[i];
            cd_glds16(ok ? src : zero_page, sbase + (i * NW + wave) * 1024);
        }
    };
    // operand addresses (see convT_wgrad_direct_kernel): k-row of lane (g, q) in read h of K-step s = base pixel
    // (row 2s + (g>>1), column 8h + 4(g&1) + q)
    const int a_base = D_OFF + (((g >> 1) * TWW + 4 * (g & 1) + q) * PPD) + pq * 8;
    const int b_base = (((g >> 1) * PW + 4 * (g & 1) + q) * PPX) + pq * 8 + wave * 32;
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto tr_frag = [&](const char* base, int off, int pitch) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off + 8 * pitch));
        const s16x8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8_t, r);
    };
    f32x4 acc[9][CB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) acc[tap][cb] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ConvT bias gradient = column sums of dy over all four phases: wave ph (< 4) adds the row sums of phase ph's dy fragments
    // through an all-ones MFMA operand (3 extra MFMAs per K-step) -- no separate column-sum pass over the 520 MB of dy
    f32x4 rs[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) rs[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rs = p.dbias != nullptr && wave < 4;
    typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

    const long t0 = xcd_first_tile(), gstep = gridDim.x;
    const long mine = t0 < p.ntiles ? (p.ntiles - t0 + gstep - 1) / gstep : 0;
    if (mine > 0) issue(t0, 0);
    if (mine > 1) issue(t0 + gstep, 1);
    int slot = 0;
    for (long n = 0; n < mine; ++n) {
        if (n + 1 < mine) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (n + 2 < mine) issue(t0 + (n + 2) * gstep, slot >= 1 ? slot - 1 : slot + 2);
        const char* st = smem + slot * STAGE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                bf16x8_t af[CB];
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) af[cb] = tr_frag(st, ph * PLANE + a_base + s * 2 * TWW * PPD + cb * 32, PPD);
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int ky = tap / 3, kx = tap % 3;
                    if ((ky != 1) * 2 + (kx != 1) != ph) continue;
                    const bf16x8_t bf = tr_frag(st, b_base + (s * 2 * PW + (ky == 0) * PW + (kx == 0)) * PPX, PPX);
#pragma unroll
                    for (int cb = 0; cb < CB; ++cb) acc[tap][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cb], bf, acc[tap][cb], 0, 0, 0);
                }
            }
            if (do_rs) {  // one wave-uniform block per K-step, outside the phase loop: re-read phase `wave`'s dy fragments
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const bf16x8_t a2 = tr_frag(st, wave * PLANE + a_base + s * 2 * TWW * PPD + cb * 32, PPD);
                    rs[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, ones, rs[cb], 0, 0, 0);
                }
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                ig_red_add(p.dw + (size_t)(cb * 16 + 4 * g + r) * (9 * CIN) + tap * CIN + wave * 16 + i16, acc[tap][cb][r]);
    if (p.dbias != nullptr) {
        // the four phase waves' partial sums are folded through LDS first: 4 x 48 same-address atomics per workgroup, issued
        // by all 256 workgroups as they finish together, were 60 us of serialised atomics at the tail of a 200 us kernel
        __syncthreads();  // every wave is done with the ring
        float* red = reinterpret_cast<float*>(smem);
        if (do_rs && i16 == 0) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[wave * COUT + cb * 16 + 4 * g + r] = rs[cb][r];
        }
        __syncthreads();
        if (tid < COUT) ig_red_add(p.dbias + tid, red[tid] + red[COUT + tid] + red[2 * COUT + tid] + red[3 * COUT + tid]);
    }
}

// ---- LDS-DMA variant of the 3x3 weight gradient for 96 input channels (one 8-wave workgroup per CU) ------------------
// Same ring protocol as convT_wgrad_dma_kernel.  Tile = 8 x 16 pixels: x halo 10 x 18 pixels x 224 B (40 wave-DMAs), dy tile
// 8 x 16 pixels x 96 B of this workgroup's 48-channel slice (12 wave-DMAs); 3 stages x 52 KiB.  Waves 0-3 issue 7 DMAs per
// tile, waves 4-7 issue 6 (52 = 4 x 7 + 4 x 6), so the counted vmcnt differs by wave.
template <int CIN, int R8>  // R8 = tile rows: 8 for 48 / 96 input channels, 4 for 192 (three stages must fit 160 KiB)
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad_dma_kernel(CWParams p, const bf16_t* zero_page) {
    constexpr int COB = 48;
    constexpr int PPX = 2 * CIN + ((CIN / 2) % 16 == 8 ? 0 : 32), PPD = 2 * COB;
    static_assert((PPX / 4) % 16 == 8, "x pixel pitch must be = 8 (mod 16) dwords");
    constexpr int PHh = R8 + 2;                 // halo rows
    constexpr int XUP = PPX / 16;               // units per halo pixel (CIN / 8 data + 2 pad)
    constexpr int XUNITS = PHh * HW_ * XUP;
    constexpr int XSLOTS = (XUNITS + 63) / 64;  // 40 (96 ch, 8 rows) / 44 (192 ch, 4 rows)
    constexpr int DUNITS = R8 * TW * (COB / 8);
    constexpr int DSLOTS = DUNITS / 64;         // 12 / 6
    constexpr int SLOTS = XSLOTS + DSLOTS;      // 52 / 50
    constexpr int NW = 8, PER_WAVE = (SLOTS + NW - 1) / NW;  // 7
    static_assert(3 * SLOTS * 1024 <= 160 * 1024, "three stages must fit the LDS");
    constexpr int FULL_WAVES = SLOTS - (PER_WAVE - 1) * NW;  // waves 0 .. FULL_WAVES-1 issue PER_WAVE DMAs per tile, the rest one less
    constexpr int STAGE = SLOTS * 1024, D_OFF = XSLOTS * 1024;
    constexpr int CB = COB / 16, CIB = CIN / 16, NBLK = 9 * CIB, NBW = (NBLK + NW - 1) / NW;
    constexpr int KS = R8 * TW / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i16 = lane & 15, q = i16 >> 2, pq = i16 & 3;
    const int co0 = blockIdx.y * COB;

    int u_off[PER_WAVE], u_meta[PER_WAVE];  // meta: 0 = zero page; else (kind << 16) | (row << 8) | col, kind 1 = x halo, 2 = dy
#pragma unroll
    for (int i = 0; i < PER_WAVE; ++i) {
        const int slot = i * NW + wave, u = slot * 64 + lane;
        u_off[i] = 0, u_meta[i] = 0;
        if (slot < XSLOTS) {
            const int hp = u / XUP, c = u - hp * XUP;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            if (u < XUNITS && c < CIN / 8) u_off[i] = (hy * p.W + hx) * CIN + c * 8, u_meta[i] = (1 << 16) | (hy << 8) | hx;
        } else if (slot < SLOTS) {
            const int ud = u - XSLOTS * 64;
            const int pix = ud / (COB / 8), c8 = ud - pix * (COB / 8);
            const int ty = pix / TW, tx = pix - ty * TW;
            u_off[i] = (ty * p.W + tx) * p.Cout + co0 + c8 * 8;
            u_meta[i] = (2 << 16) | (ty << 8) | tx;
        }
    }
    auto tile_coords = [&](long t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = (int)(t / per_img);
        const int r = (int)(t - (long)b * per_img);
        const int ty = r / p.tiles_x;
        ty0 = ty * R8, tx0 = (r - ty * p.tiles_x) * TW;
    };
    auto issue = [&](long t, int slot3) {
        int b, ty0, tx0;
        tile_coords(t, b, ty0, tx0);
        const bf16_t* xb = p.x + (((long)b * p.H + ty0 - 1) * p.W + tx0 - 1) * CIN;  // halo origin (outside units are masked)
        const bf16_t* db = p.dy + (((size_t)b * p.H + ty0) * p.W + tx0) * p.Cout;
        const unsigned sbase = lds_base + slot3 * STAGE;
#pragma unroll
        for (int i = 0; i < PER_WAVE; ++i) {
            if (i * NW + wave >= SLOTS) break;  // wave-uniform: waves 4-7 have one DMA less
            const int m = u_meta[i];
            const int r = (m >> 8) & 0xff, c = m & 0xff;
            const bool isx = (m >> 16) == 1;
            const bool ok = (m != 0) & (isx ? ((unsigned)(ty0 + r - 1) < (unsigned)p.H) & ((unsigned)(tx0 + c - 1) < (unsigned)p.W)
                                            : (ty0 + r < p.H) & (tx0 + c < p.W));
            const bf16_t* src = (isx ? xb : db) + u_off[i];
            cd_glds16(ok ? src : zero_page, sbase + (i * NW + wave) * 1024);
        }
    };
    const int a_base = D_OFF + (((g >> 1) * TW + 4 * (g & 1) + q) * PPD) + pq * 8;
    const int b_lane = (((g >> 1) * HW_ + 4 * (g & 1) + q) * PPX) + pq * 8;
    int b_base[NBW];
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int nb = min(wave * NBW + b, NBLK - 1);
        const int tap = nb / CIB, cib = nb - tap * CIB;
        const int dy = tap / 3, dx = tap - dy * 3;
        b_base[b] = b_lane + (dy * HW_ + dx) * PPX + cib * 32;
    }
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto tr_frag = [&](const char* base, int off, int pitch) {
        const s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off));
        const s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(base + off + 8 * pitch));
        const s16x8 r = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        return __builtin_bit_cast(bf16x8_t, r);
    };
    f32x4 acc[CB][NBW];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int b = 0; b < NBW; ++b) acc[cb][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    // bias gradient of this convolution = column sums of dy = row sums of the A operand: wave cb (< CB) multiplies the dy
    // fragment of channel block cb, which it holds anyway, by an all-ones operand (one extra MFMA per K-step) instead of a
    // separate column-sum pass over dy
    f32x4 rs = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_rs = p.dbias != nullptr && wave < CB;
    typedef __attribute__((ext_vector_type(8))) short ones_s16x8;
    const bf16x8_t ones = __builtin_bit_cast(bf16x8_t, ones_s16x8{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80});

    const long t0 = xcd_first_tile(), gstep = gridDim.x;
    const long mine = t0 < p.ntiles ? (p.ntiles - t0 + gstep - 1) / gstep : 0;
    if (mine > 0) issue(t0, 0);
    if (mine > 1) issue(t0 + gstep, 1);
    int slot = 0;
    for (long n = 0; n < mine; ++n) {
        if (n + 1 < mine) {
            if (wave < FULL_WAVES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_WAVE - 1) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        if (n + 2 < mine) issue(t0 + (n + 2) * gstep, slot >= 1 ? slot - 1 : slot + 2);
        const char* st = smem + slot * STAGE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8_t af[CB];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) af[cb] = tr_frag(st, a_base + s * 2 * TW * PPD + cb * 32, PPD);
            if (do_rs) {
#pragma unroll
                for (int cb = 0; cb < CB; ++cb)
                    if (cb == wave) rs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cb], ones, rs, 0, 0, 0);
            }
#pragma unroll
            for (int b = 0; b < NBW; ++b) {
                const bf16x8_t bf = tr_frag(st, b_base[b] + s * 2 * HW_ * PPX, PPX);
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[cb][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[cb], bf, acc[cb][b], 0, 0, 0);
            }
        }
        slot = slot == 2 ? 0 : slot + 1;
    }
#pragma unroll
    for (int b = 0; b < NBW; ++b) {
        const int nb = wave * NBW + b;
        if (nb < NBLK) {
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int r = 0; r < 4; ++r) ig_red_add(p.dw + (size_t)(co0 + cb * 16 + 4 * g + r) * (9 * CIN) + nb * 16 + i16, acc[cb][b][r]);
        }
    }
    if (do_rs && i16 == 0) {  // every column of the ones-product holds the row sum: lanes with column 0 own rows 4g .. 4g+3
#pragma unroll
        for (int r = 0; r < 4; ++r) ig_red_add(p.dbias + co0 + wave * 16 + 4 * g + r, rs[r]);
    }
}

// ---- LDS-DMA variant of the ConvTranspose forward: 8 MFMA waves + 1 loader wave per CU --------------------------------
// convT_direct_kernel (one workgroup per CU, register prefetch, nothing in flight while it waits) moves its 830 MB at
// 2.3 TB/s.  Here the tile is 8 x 16 input pixels (patch 9 x 17 pixels x 224 B = 34 wave-DMAs), each MFMA wave owns one input
// row (4 phases x 3 channel blocks), and a ninth wave only issues the patch DMAs into a 2-stage ring and waits for them
// (exact vmcnt: it never stores), so the MFMA waves never wait on memory and their stores drain under the next tile.
template <int CIN, int COUT>
__global__ __launch_bounds__(576, 1) void convT_direct_dma_kernel(CTParams p, const bf16_t* zero_page) {
    constexpr int NCW = 8, R8 = 8;
    constexpr int KSUB = CIN / 32;
    constexpr int WP = 9 * KSUB * 64 + 32, PP = 2 * CIN + 32;
    static_assert((WP / 4) % 16 == 8 && (PP / 4) % 16 == 8, "pitches must be = 8 (mod 16) dwords for ds_read_b128");
    constexpr int NB = COUT / 16, NPAIR = NB / 2;
    constexpr int W_BYTES = COUT * WP;
    constexpr int PH = R8 + 1, PW = CT_T + 1;
    constexpr int XUP = PP / 16;                     // 14 units per patch pixel
    constexpr int XUNITS = PH * PW * XUP;            // 2142
    constexpr int XSLOTS = (XUNITS + 63) / 64;       // 34
    constexpr int STAGE = XSLOTS * 1024;
    constexpr int PAR_OFF = W_BYTES + 2 * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    char* wl = smem;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;

    auto pos_of = [](int c) { return c < NPAIR * 32 ? (c / 32) * 32 + ((c % 8) / 4) * 16 + ((c % 32) / 8) * 4 + c % 4 : c; };
    constexpr int WUNITS = 9 * CIN / 8;
    for (int u = tid; u < COUT * WUNITS; u += 576) {
        const int co = u / WUNITS, k8 = u - co * WUNITS;
        *reinterpret_cast<uint4*>(wl + pos_of(co) * WP + k8 * 16) = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * WUNITS + k8) * 8);
    }
    for (int c = tid; c < COUT; c += 576) par[c] = p.bias ? p.bias[c] : 0.f;
    __syncthreads();

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        ty0 = ty * R8, tx0 = (r - ty * p.tiles_x) * CT_T;
    };
    const int nt = (int)p.ntiles, gstep = (int)gridDim.x;
    const int t0 = xcd_first_tile();
    const int mine = t0 < nt ? (nt - t0 + gstep - 1) / gstep : 0;

    if (wave == NCW) {
        // ================================= loader wave =================================
        int u_off[XSLOTS], u_yx[XSLOTS];
#pragma unroll
        for (int i = 0; i < XSLOTS; ++i) {
            const int u = i * 64 + lane;
            const int hp = u / XUP, c = u - hp * XUP;
            const int hy = hp / PW, hx = hp - hy * PW;
            u_off[i] = (hy * p.W + hx) * CIN + c * 8;
            u_yx[i] = (u < XUNITS && c < CIN / 8) ? (hy << 8) | hx : 0xffff;  // pad units and the tail: always the zero page
        }
        auto issue = [&](int t, int st) {
            int b, ty0, tx0;
            tile_coords(t, b, ty0, tx0);
            const bf16_t* base = p.x + (((size_t)b * p.H + ty0) * p.W + tx0) * CIN;
            const unsigned sbase = lds_base + W_BYTES + st * STAGE;
#pragma unroll
            for (int i = 0; i < XSLOTS; ++i) {
                const bool ok = u_yx[i] != 0xffff && (ty0 + (u_yx[i] >> 8) < p.H) & (tx0 + (u_yx[i] & 0xff) < p.W);
                cd_glds16(ok ? base + u_off[i] : zero_page, sbase + i * 1024);
            }
        };
        if (mine > 0) issue(t0, 0);
        for (int n = 0; n < mine; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile n has landed (nothing else is outstanding with two stages)
            asm volatile("s_barrier" ::: "memory");
            if (n + 1 < mine) issue(t0 + (n + 1) * gstep, (n + 1) & 1);  // the consumers are past tile n-1: its stage is free
        }
        return;
    }

    // ================================= MFMA waves: input row `wave` of the tile =================================
    const int g = lane >> 4, j = lane & 15;
    const int x_lane = (wave * PW + j) * PP + g * 16;
    const char* w_lane = wl + j * WP + g * 16;
    auto finish4 = [&](f32x4 a, int n, size_t idx, float* v) {
        const float4 bb = *reinterpret_cast<const float4*>(par + n);
        v[0] = a[0] + bb.x, v[1] = a[1] + bb.y, v[2] = a[2] + bb.z, v[3] = a[3] + bb.w;
        if (p.drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, p.drop_thresh, p.drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
    };
    for (int n = 0; n < mine; ++n) {
        asm volatile("s_barrier" ::: "memory");
        int b, ty0, tx0;
        tile_coords(t0 + n * gstep, b, ty0, tx0);
        const char* xs = smem + W_BYTES + (n & 1) * STAGE + x_lane;
        f32x4 acc[4][NB];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[ph][nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            const int ph = (ky != 1) * 2 + (kx != 1);
            const int xoff = ((ky == 0) * PW + (kx == 0)) * PP;
#pragma unroll
            for (int ks = 0; ks < KSUB; ++ks) {
                bf16x8_t wf[NB];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) wf[nb] = *reinterpret_cast<const bf16x8_t*>(w_lane + nb * 16 * WP + (tap * KSUB + ks) * 64);
                const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(xs + xoff + ks * 64);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[ph][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], pf, acc[ph][nb], 0, 0, 0);
            }
        }
        const int iy = ty0 + wave, ix = tx0 + j;
        if (iy < p.H && ix < p.W) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const size_t pix = ((size_t)b * 2 * p.H + 2 * iy + (ph >> 1)) * (2 * p.W) + 2 * ix + (ph & 1);
#pragma unroll
                for (int pr = 0; pr < NPAIR; ++pr) {
                    const int nn = pr * 32 + 8 * g;
                    const size_t idx = pix * COUT + nn;
                    float v[8];
                    finish4(acc[ph][2 * pr], nn, idx, v);
                    finish4(acc[ph][2 * pr + 1], nn + 4, idx + 4, v + 4);
                    *reinterpret_cast<uint4*>(p.y + idx) = pack8(v);
                }
                if (NB & 1) {
                    const int nn = (NB - 1) * 16 + 4 * g;
                    const size_t idx = pix * COUT + nn;
                    float v[4];
                    finish4(acc[ph][NB - 1], nn, idx, v);
                    store4_split(p.y, nullptr, idx, v);
                }
            }
        }
    }
}

// ---- 3x3 forward / data gradient at 96 channels: 48-channel output slices, 8 MFMA waves + 1 loader wave per CU ------------
// All of Wc[96][864] (166 KiB) does not fit the LDS next to a halo, so a workgroup owns a 48-wide slice of the output channels
// (blockIdx.y; 84.5 KiB of weights) and the input halo is staged by both slices' workgroups (the second read is an L2 / MALL
// hit).  Tile = 8 x 16 pixels, one row per MFMA wave; the halo (10 x 18 pixels x 192 B, lane-linear, 34 wave-DMAs) goes
// through a 2-stage ring filled by the loader wave (see convT_direct_dma_kernel).  The 224-byte pixel pitch that is conflict-free
// by itself would not leave room for the second stage; at the unpadded 192 bytes the pixel-side ds_read_b128 is 2-way conflicted
// unless the 16-byte chunks of a pixel are swizzled: chunk ^= 2 for halo columns with (hx >> 2) & 1 -- applied on the SOURCE chunk by
// the loader wave and by the same involution on the fragment reads -- is conflict-free for all three tap columns and K-substeps
// (tools/lds_bank_model.py): forward 524 -> 496 us, data gradient 541 -> 510 at B = 216.  Two rows per MFMA wave (4 MFMA waves + the
// loader: every weight fragment feeds two pixel rows, 5 fragment reads per 6 MFMAs instead of 4 per 3) was measured SLOWER (530-540 us):
// it is the eight waves' latency hiding that the kernel needs, not LDS bandwidth.
template <int C>
__global__ __launch_bounds__(576, 1) void conv3x3_direct_slice_kernel(CDParams p, const bf16_t* zero_page) {
    static_assert(C == 96, "written for 96 channels (48-wide output slices)");
    constexpr int NCW = 8, R8 = 8, COB = 48;
    constexpr int UNITS = C / 8;                 // 12 units per pixel
    constexpr int KSTEPS = 9 * C / 32;           // 27, three per tap
    constexpr int WP = KSTEPS * 64 + 32;         // 440 dwords: conflict-free weight reads
    constexpr int PP = 2 * C;                    // 192 B (see above)
    constexpr int NB = COB / 16, NPAIR = NB / 2;
    constexpr int W_BYTES = COB * WP;
    constexpr int PHh = R8 + 2;
    constexpr int HUNITS = PHh * HW_ * UNITS;    // 2160
    constexpr int HSLOTS = (HUNITS + 63) / 64;   // 34
    constexpr int STAGE = HSLOTS * 1024;
    constexpr int PAR_OFF = W_BYTES + 2 * STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    char* wl = smem;
    float* par = reinterpret_cast<float*>(smem + PAR_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int co0 = blockIdx.y * COB;
    uint32_t drop_seed = p.drop_seed;
    if (p.drop_seed_dev) drop_seed += *p.drop_seed_dev;

    auto pos_of = [](int c) { return c < NPAIR * 32 ? (c / 32) * 32 + ((c % 8) / 4) * 16 + ((c % 32) / 8) * 4 + c % 4 : c; };
    constexpr int KG = 9 * UNITS;  // 108 k-groups per weight row
    for (int i = tid; i < W_BYTES / 16; i += 576) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if (!p.dgrad) {
        for (int u = tid; u < COB * KG; u += 576) {
            const int co = u / KG, gk = u - co * KG;
            *reinterpret_cast<uint4*>(wl + pos_of(co) * WP + gk * 16) = *reinterpret_cast<const uint4*>(p.w + ((size_t)(co0 + co) * KG + gk) * 8);
        }
    } else {
        // rows = this slice's input channels ci0 .. ci0+47: W'[ci][tap'][co] = Wc[co][8 - tap'][ci], co over all C
        constexpr int SU = COB / 8;  // 16-byte units of the slice per (co, tap)
        for (int u = tid; u < C * 9 * SU; u += 576) {
            const int co = u / (9 * SU), r = u - co * (9 * SU), tap = r / SU, c8 = r - tap * SU;
            const uint4 v = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * 9 + tap) * C + co0 + c8 * 8);
            const uint32_t qv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 8; ++i)
                reinterpret_cast<bf16_t*>(wl + pos_of(c8 * 8 + i) * WP)[(8 - tap) * C + co] = (bf16_t)(qv[i >> 1] >> ((i & 1) * 16));
        }
    }
    for (int c = tid; c < COB; c += 576) {
        par[c] = p.bias ? p.bias[co0 + c] : 0.f;
        par[COB + c] = p.col_scale ? p.col_scale[co0 + c] : 1.f;
        par[2 * COB + c] = p.col_scale ? p.col_shift[co0 + c] : 0.f;
    }
    const bool has_bn = p.col_scale != nullptr;
    __syncthreads();

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        ty0 = ty * R8, tx0 = (r - ty * p.tiles_x) * TW;
    };
    const int nt = (int)p.ntiles, gstep = (int)gridDim.x;
    const int t0 = xcd_first_tile();
    const int mine = t0 < nt ? (nt - t0 + gstep - 1) / gstep : 0;

    if (wave == NCW) {
        // ================================= loader wave =================================
        int u_off[HSLOTS], u_yx[HSLOTS];
#pragma unroll
        for (int i = 0; i < HSLOTS; ++i) {
            const int u = i * 64 + lane;
            const int hp = u / UNITS, c8 = u - hp * UNITS;
            const int hy = hp / HW_, hx = hp - hy * HW_;
            u_off[i] = (hy * p.W + hx) * C + (c8 ^ (((hx >> 2) & 1) << 1)) * 8;  // bank swizzle on the source chunk
            u_yx[i] = u < HUNITS ? (hy << 8) | hx : 0xffff;
        }
        auto issue = [&](int t, int st) {
            int b, ty0, tx0;
            tile_coords(t, b, ty0, tx0);
            const bf16_t* base = p.x + (((long)b * p.H + ty0 - 1) * p.W + tx0 - 1) * C;  // halo origin (outside units masked)
            const unsigned sbase = lds_base + W_BYTES + st * STAGE;
#pragma unroll
            for (int i = 0; i < HSLOTS; ++i) {
                const int hy = u_yx[i] >> 8, hx = u_yx[i] & 0xff;
                const bool ok = u_yx[i] != 0xffff && ((unsigned)(ty0 + hy - 1) < (unsigned)p.H) & ((unsigned)(tx0 + hx - 1) < (unsigned)p.W);
                cd_glds16(ok ? base + u_off[i] : zero_page, sbase + i * 1024);
            }
        };
        if (mine > 0) issue(t0, 0);
        for (int n = 0; n < mine; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (n + 1 < mine) issue(t0 + (n + 1) * gstep, (n + 1) & 1);
        }
        return;
    }

    // ================================= MFMA waves: row `wave` of the tile =================================
    const int g = lane >> 4, j = lane & 15;
    int x_lane[3];  // per tap column dx: this lane's pixel + its swizzled chunk (halo column j + dx)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) x_lane[dx] = (wave * HW_ + j + dx) * PP + ((g ^ ((((j + dx) >> 2) & 1) << 1)) << 4);
    const char* w_lane = wl + j * WP + g * 16;
    auto finish4 = [&](f32x4 a, int n, size_t idx, float* v) {
        const float4 bb = *reinterpret_cast<const float4*>(par + n);
        v[0] = a[0] + bb.x, v[1] = a[1] + bb.y, v[2] = a[2] + bb.z, v[3] = a[3] + bb.w;
        if (has_bn) {
            const float4 sc = *reinterpret_cast<const float4*>(par + COB + n);
            const float4 sh = *reinterpret_cast<const float4*>(par + 2 * COB + n);
            v[0] = fmaxf(v[0] * sc.x + sh.x, 0.f), v[1] = fmaxf(v[1] * sc.y + sh.y, 0.f);
            v[2] = fmaxf(v[2] * sc.z + sh.z, 0.f), v[3] = fmaxf(v[3] * sc.w + sh.w, 0.f);
        }
        if (p.drop_thresh) {
            float mk[4];
            dropout_scale4(drop_seed, (uint32_t)idx, p.drop_thresh, p.drop_inv, mk);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] *= mk[i];
        }
    };
    // BatchNorm statistics of this lane's 12 channels of the slice (see conv3x3_direct_kernel)
    constexpr int NS = NPAIR * 8 + (NB & 1) * 4;
    float st_s[NS], st_q[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) st_s[k] = 0.f, st_q[k] = 0.f;
    const bool want_stats = p.stats_part != nullptr;
    for (int n = 0; n < mine; ++n) {
        asm volatile("s_barrier" ::: "memory");
        int b, ty0, tx0;
        tile_coords(t0 + n * gstep, b, ty0, tx0);
        const char* xs = smem + W_BYTES + (n & 1) * STAGE;
        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int tap = ks / 3, dy = tap / 3, dx = tap % 3;
            bf16x8_t wf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[nb] = *reinterpret_cast<const bf16x8_t*>(w_lane + nb * 16 * WP + ks * 64);
            const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(xs + x_lane[dx] + dy * HW_ * PP + (ks % 3) * 64);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], pf, acc[nb], 0, 0, 0);
        }
        const int oy = ty0 + wave, ox = tx0 + j;
        if (oy < p.H && ox < p.W) {
            const size_t pixc = (((size_t)b * p.H + oy) * p.W + ox) * C + co0;
#pragma unroll
            for (int pr = 0; pr < NPAIR; ++pr) {
                const int nn = pr * 32 + 8 * g;
                const size_t idx = pixc + nn;
                float v[8];
                finish4(acc[2 * pr], nn, idx, v);
                finish4(acc[2 * pr + 1], nn + 4, idx + 4, v + 4);
                const uint4 pk = pack8(v);
                *reinterpret_cast<uint4*>(p.y + idx) = pk;
                if (want_stats) {
                    float r[8];
                    unpack8(pk, r);
#pragma unroll
                    for (int i = 0; i < 8; ++i) st_s[pr * 8 + i] += r[i], st_q[pr * 8 + i] = fmaf(r[i], r[i], st_q[pr * 8 + i]);
                }
            }
            if (NB & 1) {
                const int nn = (NB - 1) * 16 + 4 * g;
                const size_t idx = pixc + nn;
                float v[4];
                finish4(acc[NB - 1], nn, idx, v);
                store4_split(p.y, nullptr, idx, v);
                if (want_stats) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float r = bf2f(f2bf(v[i]));
                        st_s[NPAIR * 8 + i] += r, st_q[NPAIR * 8 + i] = fmaf(r, r, st_q[NPAIR * 8 + i]);
                    }
                }
            }
        }
    }
    if (want_stats) {  // the loader wave has left: the barriers below count the eight MFMA waves
#pragma unroll
        for (int k = 0; k < NS; ++k) {
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) st_s[k] += __shfl_xor(st_s[k], o, 64), st_q[k] += __shfl_xor(st_q[k], o, 64);
        }
        asm volatile("s_barrier" ::: "memory");  // every wave has finished reading the halo stages
        float* red = reinterpret_cast<float*>(smem + W_BYTES);  // [wave][g][NS][2]
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < NS; ++k) red[((wave * 4 + g) * NS + k) * 2] = st_s[k], red[((wave * 4 + g) * NS + k) * 2 + 1] = st_q[k];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        for (int c = tid; c < COB; c += NCW * 64) {
            int gg, k;
            if (c < NPAIR * 32) gg = (c % 32) / 8, k = (c / 32) * 8 + c % 8;
            else gg = (c - (NB - 1) * 16) / 4, k = NPAIR * 8 + (c - (NB - 1) * 16) % 4;
            float ss = 0.f, qq = 0.f;
            for (int w = 0; w < NCW; ++w) ss += red[((w * 4 + gg) * NS + k) * 2], qq += red[((w * 4 + gg) * NS + k) * 2 + 1];
            p.stats_part[(size_t)blockIdx.x * 2 * C + co0 + c] = ss;
            p.stats_part[(size_t)blockIdx.x * 2 * C + C + co0 + c] = qq;
        }
    }
}

// ---- ConvTranspose data gradient, 96 -> 48 stage: dx[a][b][ci] = sum_{tap,co} dy[2a-1+ky][2b-1+kx][co] * Wc[co][tap][ci] --------
// A stride-2 3x3 gather over dy (K = 9 * 48 = 432).  The loader wave de-interleaves dy into four sub-pixel phase planes
// (plane (py,px)[r][c] = dy[2(a0+r)-py][2(b0+c)-px], 5 x 17 pixels x 96 B each) so that the 16 pixels of an MFMA operand are
// consecutive in LDS; tap (ky,kx) of base pixel (a0+i, b0+j) reads plane ((ky!=1),(kx!=1)) at (i + (ky==2), j + (kx==2)).
// All of W'[ci][tap*48+co] (96 rows x 928 B) is LDS-resident; tile = 4 x 16 base pixels, 8 MFMA waves = 4 rows x 2 halves of the
// 96 output channels, 2-stage ring (see convT_direct_dma_kernel).
struct CTDParams {
    const bf16_t* dy;  // [B][2H][2W][COUT]
    const bf16_t* w;   // Wc[COUT][9][CIN]
    bf16_t* dx;        // [B][H][W][CIN]
    int B, H, W;
    int tiles_x, tiles_y;
    long ntiles;
};

template <int CIN, int COUT>
__global__ __launch_bounds__(576, 1) void convT_dgrad_direct_kernel(CTDParams p, const bf16_t* zero_page) {
    static_assert(CIN == 96 && COUT == 48, "written for the 96 -> 48 stage");
    constexpr int NCW = 8, R4 = 4;
    constexpr int UN = COUT / 8;                   // 6 units per dy pixel = k-groups per tap
    constexpr int KG = 9 * UN, KSTEPS = (KG + 3) / 4;  // 54 k-groups, 14 K-steps (the last half padded)
    constexpr int WP = KSTEPS * 64 + 32;           // 232 dwords
    constexpr int PPD = 2 * COUT;                  // 24 dwords: conflict-free
    constexpr int PR = R4 + 1, PC = TW + 1;        // plane rows / columns
    constexpr int PLANE = PR * PC * PPD;           // 8160 B
    constexpr int DUNITS = 4 * PLANE / 16;         // 2040
    constexpr int DSLOTS = (DUNITS + 63) / 64;     // 32
    constexpr int STAGE = DSLOTS * 1024;
    constexpr int W_BYTES = CIN * WP;
    constexpr int ZERO_OFF = W_BYTES + 2 * STAGE;
    constexpr int NB = 3;                          // 16-channel blocks per wave (half of the 96 output channels)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef __attribute__((address_space(3))) char* lds_char_ptr;
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_char_ptr)smem;
    char* wl = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- W'[ci][tap*COUT + co] -> LDS (rows interleaved per 48-channel half: a lane owns 8 consecutive output channels)
    auto pos48 = [](int c) { return c < 32 ? ((c % 8) / 4) * 16 + (c / 8) * 4 + c % 4 : c; };
    for (int i = tid; i < (W_BYTES) / 16; i += 576) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
    if (tid == 0) *reinterpret_cast<uint4*>(smem + ZERO_OFF) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    constexpr int CU = CIN / 8;
    for (int u = tid; u < COUT * 9 * CU; u += 576) {
        const int co = u / (9 * CU), r = u - co * (9 * CU), tap = r / CU, c8 = r - tap * CU;
        const uint4 v = *reinterpret_cast<const uint4*>(p.w + ((size_t)co * 9 + tap) * CIN + c8 * 8);
        const uint32_t qv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ci = c8 * 8 + i;
            reinterpret_cast<bf16_t*>(wl + ((ci / 48) * 48 + pos48(ci % 48)) * WP)[tap * COUT + co] = (bf16_t)(qv[i >> 1] >> ((i & 1) * 16));
        }
    }
    __syncthreads();

    auto tile_coords = [&](int t, int& b, int& ty0, int& tx0) {
        const int per_img = p.tiles_x * p.tiles_y;
        b = t / per_img;
        const int r = t - b * per_img;
        const int ty = r / p.tiles_x;
        ty0 = ty * R4, tx0 = (r - ty * p.tiles_x) * TW;
    };
    const int nt = (int)p.ntiles, gstep = (int)gridDim.x;
    const int t0 = xcd_first_tile();
    const int mine = t0 < nt ? (nt - t0 + gstep - 1) / gstep : 0;

    if (wave == NCW) {
        // ================================= loader wave =================================
        int u_off[DSLOTS], u_yx[DSLOTS];  // yx: (row offset << 8) | column offset of the dy pixel from (2a0 - 1, 2b0 - 1)
#pragma unroll
        for (int i = 0; i < DSLOTS; ++i) {
            const int u = i * 64 + lane;
            const int plane = u / (PLANE / 16), rem = u - plane * (PLANE / 16);
            const int pix = rem / UN, c8 = rem - pix * UN;
            const int r = pix / PC, c = pix - r * PC;
            const int py = plane >> 1, px = plane & 1;
            const int dyo = 2 * r - py + 1, dxo = 2 * c - px + 1;
            const bool used = u < DUNITS && r < R4 + py && c < TW + px;
            u_off[i] = (dyo * 2 * p.W + dxo) * COUT + c8 * 8;
            u_yx[i] = used ? (dyo << 8) | dxo : 0xffff;
        }
        auto issue = [&](int t, int st) {
            int b, ty0, tx0;
            tile_coords(t, b, ty0, tx0);
            const bf16_t* base = p.dy + (((long)b * 2 * p.H + 2 * ty0 - 1) * (2 * p.W) + 2 * tx0 - 1) * COUT;  // (2a0-1, 2b0-1)
            const unsigned sbase = lds_base + W_BYTES + st * STAGE;
#pragma unroll
            for (int i = 0; i < DSLOTS; ++i) {
                const int oy = 2 * ty0 - 1 + (u_yx[i] >> 8), ox = 2 * tx0 - 1 + (u_yx[i] & 0xff);
                const bool ok = u_yx[i] != 0xffff && ((unsigned)oy < (unsigned)(2 * p.H)) & ((unsigned)ox < (unsigned)(2 * p.W));
                cd_glds16(ok ? base + u_off[i] : zero_page, sbase + i * 1024);
            }
        };
        if (mine > 0) issue(t0, 0);
        for (int n = 0; n < mine; ++n) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");
            if (n + 1 < mine) issue(t0 + (n + 1) * gstep, (n + 1) & 1);
        }
        return;
    }

    // ================================= MFMA waves: row (wave >> 1), channel half (wave & 1) =================================
    const int g = lane >> 4, j = lane & 15;
    const int row = wave >> 1, half = wave & 1;
    int offk[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
        const int kg = ks * 4 + g;
        const int tap = kg / UN, cg = kg - tap * UN;
        const int ky = tap / 3, kx = tap - ky * 3;
        offk[ks] = ((ky != 1) * 2 + (kx != 1)) * PLANE + ((ky == 2) * PC + (kx == 2)) * PPD + cg * 16;
    }
    const int x_lane = (row * PC + j) * PPD;
    const char* w_lane = wl + (half * 48 + j) * WP + g * 16;
    const char* zero_ptr = smem + ZERO_OFF;
    for (int n = 0; n < mine; ++n) {
        asm volatile("s_barrier" ::: "memory");
        int b, ty0, tx0;
        tile_coords(t0 + n * gstep, b, ty0, tx0);
        const char* st = smem + W_BYTES + (n & 1) * STAGE + x_lane;
        f32x4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const bool padded = KG % 4 != 0 && ks == KSTEPS - 1 && g >= KG % 4;
            bf16x8_t wf[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) wf[nb] = *reinterpret_cast<const bf16x8_t*>(w_lane + nb * 16 * WP + ks * 64);
            const bf16x8_t pf = *reinterpret_cast<const bf16x8_t*>(padded ? zero_ptr : st + offk[ks]);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nb], pf, acc[nb], 0, 0, 0);
        }
        const int a = ty0 + row, bx = tx0 + j;
        if (a < p.H && bx < p.W) {
            const size_t pixc = (((size_t)b * p.H + a) * p.W + bx) * CIN + half * 48;
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = acc[0][i], v[4 + i] = acc[1][i];
            *reinterpret_cast<uint4*>(p.dx + pixc + 8 * g) = pack8(v);
            const float v2[4] = {acc[2][0], acc[2][1], acc[2][2], acc[2][3]};
            store4_split(p.dx, nullptr, pixc + 32 + 4 * g, v2);
        }
    }
}

}  // namespace
IG_DET_TU(conv_direct)  // constant-memory descriptor of the deterministic-reduction mode (common.h)

static const bf16_t* cd_zero_page() {
    static void* z = nullptr;
    if (!z) {
        if (hipMalloc(&z, 256) != hipSuccess) return nullptr;
        (void)hipMemset(z, 0, 256);
    }
    return (const bf16_t*)z;
}

// Called by ig_conv3x3_fwd / ig_conv3x3_dgrad (gemm.hip) for the shapes this kernel covers; returns IG_ERR_UNSUPPORTED
// (without setting the error string) when it does not, and the caller falls through to the implicit GEMM.
// Inference tail: the 48-channel last Conv2d (+ bias, + eval-mode BatchNorm + ReLU) with the 1 x 1 classifier applied to the finished
// pixels in the same epilogue: the head's largest activation is neither written nor read again.  y may be NULL.  IG_ERR_UNSUPPORTED
// (no error string) when the shape is not covered (C != 48, more than 2 classes): the caller runs the two kernels.
int ig_conv3x3_cls_direct(const void* x, const void* w, const float* bias, const float* bn_scale, const float* bn_shift, void* y,
                          const float* cls_w, const float* cls_b, float* logits, int B, int H, int W, int C, int ncls, void* stream) {
    if (C != 48 || ncls < 1 || ncls > 2 || (long)B * H * W * C >= (1L << 31)) return IG_ERR_UNSUPPORTED;
    CDParams p{};
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.y = (bf16_t*)y;
    p.bias = bias, p.col_scale = bn_scale, p.col_shift = bn_shift;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    p.drop_inv = 1.f;
    p.cls_w = cls_w, p.cls_b = cls_b, p.logits = logits;
    if (p.ntiles == 0) return IG_OK;
    if (ncls == 1) return launch_direct<48, 1>(p, (hipStream_t)stream, "ig_conv3x3_cls_fwd(direct)");
    return launch_direct<48, 2>(p, (hipStream_t)stream, "ig_conv3x3_cls_fwd(direct)");
}

int ig_conv3x3_direct(const void* x, const void* w, const float* bias, const float* bn_scale, const float* bn_shift, void* y,
                      int B, int H, int W, int Cin, int Cout, int dgrad, unsigned drop_seed, const unsigned* drop_seed_dev,
                      float drop_p, void* stream, double* stat_sums, int* stats_fused) {
    if (stats_fused) *stats_fused = 0;
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    if (!enabled || Cin != Cout || (Cin != 48 && Cin != 96)) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * Cin >= (1L << 31)) return IG_ERR_UNSUPPORTED;  // 32-bit halo offsets
    CDParams p{};
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.y = (bf16_t*)y;
    p.bias = bias, p.col_scale = bn_scale, p.col_shift = bn_shift;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    p.dgrad = dgrad;
    p.drop_seed = drop_seed, p.drop_seed_dev = drop_seed_dev;
    p.drop_thresh = ig_drop_thresh16(drop_p);
    p.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if (p.ntiles == 0) return IG_OK;
    if (Cin == 96) {
        p.tiles_y = (H + 7) / 8;  // 8 x 16 tiles
        p.ntiles = (long)B * p.tiles_x * p.tiles_y;
        constexpr int smem96 = 48 * (27 * 64 + 32) + 2 * 34 * 1024 + 3 * 48 * 4;
        const bf16_t* zp = cd_zero_page();
        if (!zp) {
            ig_set_error("ig_conv3x3: could not allocate the zero page");
            return IG_ERR_HIP;
        }
        static bool attr96 = false;
        if (!attr96) {
            (void)hipFuncSetAttribute((const void*)conv3x3_direct_slice_kernel<96>, hipFuncAttributeMaxDynamicSharedMemorySize, smem96);
            attr96 = true;
        }
        const long nwg = p.ntiles < 128 ? p.ntiles : 128;
        const bool st96 = stat_sums && stats_fused && !dgrad;
        if (st96) {
            p.stats_part = (float*)ig_scratch(0, (size_t)nwg * 2 * 96 * sizeof(float), (hipStream_t)stream);
            if (!p.stats_part) {
                ig_set_error("ig_conv3x3_fwd: scratch allocation failed");
                return IG_ERR_HIP;
            }
            *stats_fused = 1;
        }
        ig_note_kernel("conv3x3_direct_slice_kernel<96>");
        hipLaunchKernelGGL(conv3x3_direct_slice_kernel<96>, dim3((unsigned)nwg, 2), dim3(576), smem96, (hipStream_t)stream, p, zp);
        if (st96)
            hipLaunchKernelGGL(bn_part_fold_kernel, dim3(ig_cdiv(2 * 96, 64)), dim3(1024), 0, (hipStream_t)stream, p.stats_part, stat_sums, (int)nwg, 2 * 96);
        return ig_check_launch(dgrad ? "ig_conv3x3_dgrad(direct, slices)" : "ig_conv3x3_fwd(direct, slices)");
    }
    const bool st = stat_sums && stats_fused && !dgrad;
    if (st) *stats_fused = 1;
    return launch_direct<48>(p, (hipStream_t)stream, dgrad ? "ig_conv3x3_dgrad(direct)" : "ig_conv3x3_fwd(direct)", st ? stat_sums : nullptr);
}

// Split precision form (all of x, w, y as hi + lo pairs) of ig_conv3x3_direct: the 48-channel stage only (the hi + lo images of the
// 96-channel stage do not fit the LDS); IG_ERR_UNSUPPORTED (no error string) otherwise.
int ig_conv3x3_direct_split(const void* x_hi, const void* x_lo, const void* w_hi, const void* w_lo, const float* bias, const float* bn_scale,
                            const float* bn_shift, void* y_hi, void* y_lo, int B, int H, int W, int Cin, int Cout, int dgrad, unsigned drop_seed,
                            const unsigned* drop_seed_dev, float drop_p, void* stream, double* stat_sums, int* stats_fused) {
    if (stats_fused) *stats_fused = 0;
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    if (!enabled || Cin != Cout || Cin != 48 || !x_lo || !w_lo || !y_lo) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * Cin >= (1L << 31)) return IG_ERR_UNSUPPORTED;  // 32-bit halo offsets
    CDParams p{};
    p.x = (const bf16_t*)x_hi, p.w = (const bf16_t*)w_hi, p.y = (bf16_t*)y_hi;
    p.bias = bias, p.col_scale = bn_scale, p.col_shift = bn_shift;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    p.dgrad = dgrad;
    p.drop_seed = drop_seed, p.drop_seed_dev = drop_seed_dev;
    p.drop_thresh = ig_drop_thresh16(drop_p);
    p.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if (p.ntiles == 0) return IG_OK;
    const CDSplit q{(const bf16_t*)x_lo, (const bf16_t*)w_lo, (bf16_t*)y_lo};
    const bool st = stat_sums && stats_fused && !dgrad;
    if (st) *stats_fused = 1;
    return launch_direct_split<48>(p, q, (hipStream_t)stream, dgrad ? "ig_conv3x3_dgrad(direct, split)" : "ig_conv3x3_fwd(direct, split)", st ? stat_sums : nullptr);
}

// Called by ig_conv3x3_wgrad (gemm.hip); IG_ERR_UNSUPPORTED when the shape is not covered.
int ig_conv3x3_wgrad_direct(const void* dy, const void* x, float* dw, float* dbias, int* bias_fused, int B, int H, int W, int Cin,
                            int Cout, void* stream) {
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    *bias_fused = 0;
    if (!enabled || Cout % 48 != 0 || (Cin != 48 && Cin != 96 && Cin != 192)) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * (Cin > Cout ? Cin : Cout) >= (1L << 31)) return IG_ERR_UNSUPPORTED;
    CWParams p{};
    p.x = (const bf16_t*)x, p.dy = (const bf16_t*)dy, p.dw = dw, p.dbias = nullptr;
    p.B = B, p.H = H, p.W = W, p.Cout = Cout;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + TH - 1) / TH;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    if (p.ntiles == 0) return IG_OK;
    const int rows = Cin == 192 ? 4 : Cin == 48 ? 12 : 8;  // tile rows of the DMA kernels (x 16 columns)
    p.tiles_y = (H + rows - 1) / rows;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    p.dbias = dbias;
    *bias_fused = dbias != nullptr;
    const bf16_t* zp = cd_zero_page();
    if (!zp) {
        ig_set_error("ig_conv3x3_wgrad: could not allocate the zero page");
        return IG_ERR_HIP;
    }
    constexpr int smem48b = 3 * 42 * 1024, smem96 = 3 * 52 * 1024, smem192 = 3 * 50 * 1024;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_dma_kernel<48, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, smem48b);
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_dma_kernel<96, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, smem96);
        (void)hipFuncSetAttribute((const void*)conv3x3_wgrad_dma_kernel<192, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, smem192);
        attr_done = true;
    }
    const int nslices = Cout / 48;
    long nwg = 256 / nslices;
    if (nwg > p.ntiles) nwg = p.ntiles;
    const dim3 grid((unsigned)nwg, nslices);
    ig_note_kernel("conv3x3_wgrad_dma_kernel<%d,%d>", Cin, Cin == 48 ? rows : Cin == 96 ? 8 : 4);
    if (Cin == 48) hipLaunchKernelGGL((conv3x3_wgrad_dma_kernel<48, 12>), grid, dim3(512), smem48b, (hipStream_t)stream, p, zp);
    else if (Cin == 96) hipLaunchKernelGGL((conv3x3_wgrad_dma_kernel<96, 8>), grid, dim3(512), smem96, (hipStream_t)stream, p, zp);
    else hipLaunchKernelGGL((conv3x3_wgrad_dma_kernel<192, 4>), grid, dim3(512), smem192, (hipStream_t)stream, p, zp);
    return ig_check_launch("ig_conv3x3_wgrad(direct, dma)");
}

// Called by ig_convT_fwd (gemm.hip); IG_ERR_UNSUPPORTED when the shape is not covered.
int ig_convT_fwd_direct(const void* x, const void* w, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                        unsigned drop_seed, const unsigned* drop_seed_dev, float drop_p, void* stream) {
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    if (!enabled || Cin != 96 || Cout != 48) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * Cin >= (1L << 31)) return IG_ERR_UNSUPPORTED;
    CTParams p{};
    p.x = (const bf16_t*)x, p.w = (const bf16_t*)w, p.y = (bf16_t*)y, p.bias = bias;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + CT_T - 1) / CT_T, p.tiles_y = (H + CT_T - 1) / CT_T;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    p.drop_seed = drop_seed, p.drop_seed_dev = drop_seed_dev;
    p.drop_thresh = ig_drop_thresh16(drop_p);
    p.drop_inv = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if (p.ntiles == 0) return IG_OK;
    p.tiles_y = (H + 7) / 8;  // 8 x 16 input tiles
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    constexpr int smem_d = 48 * (27 * 64 + 32) + 2 * 34 * 1024 + 48 * 4;
    const bf16_t* zp = cd_zero_page();
    if (!zp) {
        ig_set_error("ig_convT_fwd: could not allocate the zero page");
        return IG_ERR_HIP;
    }
    static bool attr_d = false;
    if (!attr_d) {
        (void)hipFuncSetAttribute((const void*)convT_direct_dma_kernel<96, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, smem_d);
        attr_d = true;
    }
    const long nwg_d = p.ntiles < 256 ? p.ntiles : 256;
    ig_note_kernel("convT_direct_dma_kernel<96,48>");
    hipLaunchKernelGGL((convT_direct_dma_kernel<96, 48>), dim3((unsigned)nwg_d), dim3(576), smem_d, (hipStream_t)stream, p, zp);
    return ig_check_launch("ig_convT_fwd(direct, dma)");
}

// Called by ig_convT_wgrad (gemm.hip); IG_ERR_UNSUPPORTED when the shape is not covered.
int ig_convT_wgrad_direct(const void* dy, const void* x, float* dw, float* dbias, int* bias_fused, int B, int H, int W, int Cin,
                          int Cout, void* stream) {
    *bias_fused = 0;
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    if (!enabled || Cin != 96 || Cout != 48) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * 4 * Cout >= (1L << 31)) return IG_ERR_UNSUPPORTED;
    CTWParams p{};
    p.x = (const bf16_t*)x, p.dy = (const bf16_t*)dy, p.dw = dw;
    p.dbias = dbias;
    *bias_fused = p.dbias != nullptr;
    p.B = B, p.H = H, p.W = W;
    const int th = 4;
    p.tiles_x = (W + TWW - 1) / TWW, p.tiles_y = (H + th - 1) / th;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    if (p.ntiles == 0) return IG_OK;
    long nwg = p.ntiles < 256 ? p.ntiles : 256;
    constexpr int smem = 3 * 48 * 1024;
    const bf16_t* zp = cd_zero_page();
    if (!zp) {
        ig_set_error("ig_convT_wgrad: could not allocate the zero page");
        return IG_ERR_HIP;
    }
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convT_wgrad_dma_kernel<96, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    ig_note_kernel("convT_wgrad_dma_kernel<96,48>");
    hipLaunchKernelGGL((convT_wgrad_dma_kernel<96, 48>), dim3((unsigned)nwg), dim3(TW_TPB), smem, (hipStream_t)stream, p, zp);
    return ig_check_launch("ig_convT_wgrad(direct, dma)");
}

// Called by ig_convT_dgrad (gemm.hip); IG_ERR_UNSUPPORTED when the shape is not covered.
int ig_convT_dgrad_direct(const void* dy, const void* w, void* dx, int B, int H, int W, int Cin, int Cout, void* stream) {
    const char* e_dir = getenv("IG_CONV_DIRECT");  // read per call, like every engine switch (tests set it for single calls)
    const int enabled = e_dir ? atoi(e_dir) : 1;
    if (!enabled || Cin != 96 || Cout != 48) return IG_ERR_UNSUPPORTED;
    if ((long)B * H * W * 4 * Cout >= (1L << 31)) return IG_ERR_UNSUPPORTED;
    CTDParams p{};
    p.dy = (const bf16_t*)dy, p.w = (const bf16_t*)w, p.dx = (bf16_t*)dx;
    p.B = B, p.H = H, p.W = W;
    p.tiles_x = (W + TW - 1) / TW, p.tiles_y = (H + 3) / 4;
    p.ntiles = (long)B * p.tiles_x * p.tiles_y;
    if (p.ntiles == 0) return IG_OK;
    const bf16_t* zp = cd_zero_page();
    if (!zp) {
        ig_set_error("ig_convT_dgrad: could not allocate the zero page");
        return IG_ERR_HIP;
    }
    constexpr int smem = 96 * (14 * 64 + 32) + 2 * 32 * 1024 + 16;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)convT_dgrad_direct_kernel<96, 48>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    const long nwg = p.ntiles < 256 ? p.ntiles : 256;
    ig_note_kernel("convT_dgrad_direct_kernel<96,48>");
    hipLaunchKernelGGL((convT_dgrad_direct_kernel<96, 48>), dim3((unsigned)nwg), dim3(576), smem, (hipStream_t)stream, p, zp);
    return ig_check_launch("ig_convT_dgrad(direct)");
}
