// Error reporting, launch checks and device queries for the C-ABI library.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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

extern "C" {

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
