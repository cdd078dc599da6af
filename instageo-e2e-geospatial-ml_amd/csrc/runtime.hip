// Error reporting, launch checks and device queries for the C-ABI library.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void ig_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

static thread_local char g_kernel[256] = "";
static thread_local int g_grid = 0;
void ig_note_grid(int workgroups) { g_grid = workgroups; }
void ig_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}

int ig_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ig_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return IG_ERR_HIP;
    }
    return IG_OK;
}

// Compute units left free by the persistent (one workgroup per CU) kernels.  With data parallelism RCCL's all-reduce kernels
// are launched on a side stream while the backward GEMMs run; a grid that pins every CU makes them wait for a whole kernel.
static int g_reserved_cus = -1;
int ig_reserved_cus() {
    if (g_reserved_cus < 0) {
        const char* e = getenv("IG_RESERVED_CUS");
        g_reserved_cus = e ? atoi(e) : 0;
        if (g_reserved_cus < 0) g_reserved_cus = 0;
    }
    return g_reserved_cus;
}
// Grid of a persistent tile-walking GEMM (per_cu workgroups per CU; XCD x owns a contiguous eighth of the tile list and its
// workgroups deal it round-robin): the FEWEST workgroups per XCD that keep the number of rounds, so whatever the tile count
// leaves over is free for RCCL.  The reservation is soft here -- an extra round costs 25-100 % of the launch at the benchmark's
// tile counts (84 x 3 / 84 x 9 / 84 x 12 tiles on 256 CUs), so it is honoured only when it is free; IG_RESERVED_STRICT=1 makes it
// strict.  Kernels whose work divides evenly (split-K weight gradients) always honour it.
int ig_tile_grid(int ntiles, int per_cu) {
    const char* e = getenv("IG_RESERVED_STRICT");  // read per call: a test switch
    const int strict = e ? atoi(e) : 0;
    const int slots_x = ig_cu_count() / 8 * per_cu;
    int avail_x = slots_x;
    if (strict) avail_x -= (ig_reserved_cus() * per_cu + 7) / 8;
    if (avail_x < 1) avail_x = 1;
    const int tmax = (ntiles + 7) / 8;  // tiles of the fullest XCD
    const int rounds = (tmax + avail_x - 1) / avail_x;
    const int nbx = (tmax + rounds - 1) / rounds;
    const int grid = 8 * nbx;
    return grid < ntiles ? grid : ntiles;
}
int ig_cu_count() {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        else cus = 256;
    }
    return cus;
}

// ---- deterministic reduction mode (common.h: IgDet) -----------------------------------------------------------------------------
int ig_det_sync_elementwise(const IgDet*, hipStream_t);
int ig_det_sync_head(const IgDet*, hipStream_t);
int ig_det_sync_conv_direct(const IgDet*, hipStream_t);
int ig_det_sync_gemm(const IgDet*, hipStream_t);
int ig_det_sync_gemm8(const IgDet*, hipStream_t);
int ig_det_sync_gemm4(const IgDet*, hipStream_t);
int ig_det_sync_attention2(const IgDet*, hipStream_t);
IG_DET_TU(runtime)
static IgDet g_det_host = {nullptr, nullptr, 0};
bool ig_deterministic() { return g_det_host.shadow != nullptr; }
// Scratch buffers are keyed by (device, STREAM, slot): two streams of one device -- a distillation teacher's forward beside the student's
// step (segmentation.py:216-451), a graph replay beside eager inference -- never share a buffer, so a kernel on one stream cannot read
// partial sums, packed weights, tickets or split-K slabs that a launch on another stream is overwriting.  Slots: 0 BatchNorm
// partial sums, 1 packed conv8 weights, 2 / 3 loss tickets and accumulators, 4 weight-gradient slabs (gemm8w.hip), 5 split-K partials (gemm.hip).
// A stream gets its entry on first use (up to IG_SCRATCH_STREAMS = 96 per device, then NULL: the callers report it).  A fresh buffer is zero-filled.
// Nothing is allocated while the stream is capturing (NULL + an error text: warm up on the capture stream first).  Growing frees the old buffer
// (hipFree synchronises the device, so no kernel in flight still reads it) -- unless a capture has ever been served from this entry: a
// captured graph may have baked the address into its kernel nodes, so that buffer is kept for the life of the process.
void* ig_scratch(int slot, size_t bytes, hipStream_t st) { return ig_scratch2(slot, bytes, true, st); }
void* ig_scratch2(int slot, size_t bytes, bool may_grow, hipStream_t st) {
    constexpr int IG_SCRATCH_STREAMS = 96, IG_SCRATCH_SLOTS = 6;  // (PyTorch's stream pools hold 2 x 32 streams + the default one)
    struct Entry {
        hipStream_t stream;
        bool used, captured;
        void* buf[IG_SCRATCH_SLOTS];
        size_t cap[IG_SCRATCH_SLOTS];
    };
    static Entry tab[16][IG_SCRATCH_STREAMS] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || slot < 0 || slot >= IG_SCRATCH_SLOTS) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    Entry* e = nullptr;
    for (int i = 0; i < IG_SCRATCH_STREAMS && !e; ++i) {
        if (tab[dev][i].used && tab[dev][i].stream == st) e = &tab[dev][i];
        else if (!tab[dev][i].used) {
            e = &tab[dev][i];
            e->used = true, e->stream = st;
        }
    }
    if (!e) {
        ig_set_error("scratch: more than %d streams use the library on device %d", IG_SCRATCH_STREAMS, dev);
        return nullptr;
    }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    if (capturing) e->captured = true;
    if (e->cap[slot] < bytes) {
        if (!may_grow) return nullptr;
        if (capturing) {
            ig_set_error("scratch: stream %p is capturing and has no %zu-byte buffer in slot %d yet: scratch is per (device, stream) -- run one warm-up "
                         "call on the capture stream first (torch.cuda.graph(g, stream=warmup_stream))", (void*)st, bytes, slot);
            return nullptr;
        }
        void* p = nullptr;
        const size_t want = bytes + bytes / 2;  // geometric growth: the kept buffers of a capturing entry stay below three times the largest request
        if (e->buf[slot] && !e->captured) (void)hipFree(e->buf[slot]);
        e->buf[slot] = nullptr, e->cap[slot] = 0;
        if (hipMalloc(&p, want) != hipSuccess) return nullptr;
        if (hipMemset(p, 0, want) != hipSuccess) {
            (void)hipFree(p);
            return nullptr;
        }
        e->buf[slot] = p, e->cap[slot] = want;
    }
    return e->buf[slot];
}

namespace {
// grad[i] += shadow[i] * 2^-44 ; shadow[i] = 0 over a table of flat ranges (blockIdx.y = range; two elements per thread where the
// range start is even: 16-byte shadow loads)
__global__ __launch_bounds__(256) void det_fold_kernel(float* __restrict__ grad, long long* __restrict__ shadow, const long* __restrict__ ranges,
                                                       long lo1, long hi1) {
    const long lo = ranges ? ranges[2 * blockIdx.y] : lo1, hi = ranges ? ranges[2 * blockIdx.y + 1] : hi1;
    for (long i = lo + (blockIdx.x * 256L + threadIdx.x) * 2; i < hi; i += (long)gridDim.x * 512) {
        if (i + 1 < hi && ((i & 1) == 0)) {
            longlong2 s = *reinterpret_cast<const longlong2*>(shadow + i);
            if (s.x | s.y) {
                if (s.x) grad[i] += (float)((double)s.x * 5.684341886080802e-14);
                if (s.y) grad[i + 1] += (float)((double)s.y * 5.684341886080802e-14);
                *reinterpret_cast<longlong2*>(shadow + i) = make_longlong2(0, 0);
            }
        } else {
            for (long j = i; j < min(i + 2, hi); ++j) {
                const long long s = shadow[j];
                if (s) grad[j] += (float)((double)s * 5.684341886080802e-14), shadow[j] = 0;
            }
        }
    }
}
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long* __restrict__ ranges) {
    const long lo = ranges[2 * blockIdx.y], hi = ranges[2 * blockIdx.y + 1];
    for (long i = lo + (blockIdx.x * 256L + threadIdx.x) * 4; i < hi; i += (long)gridDim.x * 1024) {
        if (i + 3 < hi && ((i & 3) == 0)) *reinterpret_cast<float4*>(base + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        else
            for (long j = i; j < min(i + 4, hi); ++j) base[j] = 0.f;
    }
}
}  // namespace

extern "C" {

// shadow: zero-initialised int64[n] on the device that parallels the flat fp32 gradient buffer grad_base[n]; NULL turns the mode
// off.  The descriptor lives in constant memory of every kernel module: call it when no kernel of the library is in flight
// (it is stream-ordered on `stream`), before any graph capture.
int ig_set_deterministic(void* shadow, const void* grad_base, long n, void* stream) {
    IG_REQUIRE(shadow == nullptr || (grad_base != nullptr && n > 0), "ig_set_deterministic: shadow without a gradient buffer");
    IG_REQUIRE(((uintptr_t)shadow & 15) == 0, "ig_set_deterministic: shadow must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    g_det_host.shadow = (long long*)shadow;
    g_det_host.base = shadow ? (const float*)grad_base : nullptr;
    g_det_host.n = shadow ? n : 0;
    int rc = ig_det_sync_runtime(&g_det_host, st);
    rc |= ig_det_sync_elementwise(&g_det_host, st);
    rc |= ig_det_sync_head(&g_det_host, st);
    rc |= ig_det_sync_conv_direct(&g_det_host, st);
    rc |= ig_det_sync_gemm(&g_det_host, st);
    rc |= ig_det_sync_gemm8(&g_det_host, st);
    rc |= ig_det_sync_gemm4(&g_det_host, st);
    rc |= ig_det_sync_attention2(&g_det_host, st);
    if (rc != IG_OK) {
        ig_set_error("ig_set_deterministic: hipMemcpyToSymbol failed: %s", hipGetErrorString(hipGetLastError()));
        return IG_ERR_HIP;
    }
    if (hipStreamSynchronize(st) != hipSuccess) return IG_ERR_HIP;
    return IG_OK;
}
int ig_get_deterministic(void) { return ig_deterministic() ? 1 : 0; }

// Adds the fixed-point shadow sums of gradient elements [lo, hi) into the gradient buffer and clears them (no-op when the mode is off).
int ig_det_fold(long lo, long hi, void* stream) {
    if (!ig_deterministic() || hi <= lo) return IG_OK;
    IG_REQUIRE(lo >= 0 && hi <= g_det_host.n, "ig_det_fold: range [%ld, %ld) outside the registered buffer of %ld", lo, hi, g_det_host.n);
    const int gx = (int)std::min<long>(((hi - lo + 1) / 2 + 255) / 256, 2048);
    ig_note_kernel("det_fold_kernel");
    det_fold_kernel<<<dim3(gx, 1), 256, 0, (hipStream_t)stream>>>(const_cast<float*>(g_det_host.base), g_det_host.shadow, nullptr, lo, hi);
    return ig_check_launch("ig_det_fold");
}
// The same over a table of n ranges in DEVICE memory (int64 [n][2] = lo, hi; inside the registered buffer -- not checked) in ONE
// launch; `longest` = the longest range (sizes the grid).  A caller that knows which elements only ever receive ordered writes
// (the weights of the linears on the grouped 8-phase weight-gradient path) leaves them out of the table.
int ig_det_fold_ranges(int n, const long* ranges_dev, long longest, void* stream) {
    if (!ig_deterministic() || n <= 0 || longest <= 0) return IG_OK;
    IG_REQUIRE(ranges_dev && n <= 65535, "ig_det_fold_ranges: null table or more than 65535 ranges");
    const int gx = (int)std::min<long>(((longest + 1) / 2 + 255) / 256, 2048);
    ig_note_kernel("det_fold_kernel");
    det_fold_kernel<<<dim3(gx, n), 256, 0, (hipStream_t)stream>>>(const_cast<float*>(g_det_host.base), g_det_host.shadow, ranges_dev, 0, 0);
    return ig_check_launch("ig_det_fold_ranges");
}

int ig_zero_ranges(float* base, int n, const long* ranges_dev, long longest, void* stream) {
    if (n <= 0 || longest <= 0) return IG_OK;
    IG_REQUIRE(base && ranges_dev && n <= 65535, "ig_zero_ranges: null pointer or more than 65535 ranges");
    const int gx = (int)std::min<long>((longest + 1023) / 1024, 2048);
    zero_ranges_kernel<<<dim3(gx, n), 256, 0, (hipStream_t)stream>>>(base, ranges_dev);
    return ig_check_launch("ig_zero_ranges");
}

int ig_set_reserved_cus(int n) {
    IG_REQUIRE(n >= 0 && n < 128, "ig_set_reserved_cus: n must be in [0, 128) (got %d)", n);
    g_reserved_cus = n;
    return IG_OK;
}
int ig_get_reserved_cus(void) { return ig_reserved_cus(); }

const char* ig_last_error(void) { return g_err; }
const char* ig_last_kernel(void) { return g_kernel; }
int ig_note_reset(void) {
    g_kernel[0] = 0;
    return IG_OK;
}
int ig_last_grid(void) { return g_grid; }

int ig_version(void) { return 100; }  // 0.1.0

#ifndef IG_HEADER_STAMP
#error "build through the Makefile: IG_HEADER_STAMP (MD5 prefix of include/instageo_hip.h) is not defined"
#endif
int ig_header_stamp(void) { return (int)(IG_HEADER_STAMP); }

// name: buffer >= 64 bytes; returns 0 or IG_ERR_HIP when no usable device is present
int ig_device_info(int device, char* name, int name_len, int* cu_count, int* lds_per_block, long* hbm_bytes) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        ig_set_error("ig_device_info: %s", hipGetErrorString(e));
        return IG_ERR_HIP;
    }
    if (name && name_len > 0) {
        strncpy(name, p.gcnArchName, name_len - 1);
        name[name_len - 1] = 0;
    }
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_per_block) *lds_per_block = (int)p.sharedMemPerBlock;
    if (hbm_bytes) *hbm_bytes = (long)p.totalGlobalMem;
    return IG_OK;
}

}  // extern "C"
