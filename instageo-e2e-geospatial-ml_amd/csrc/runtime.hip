// Error reporting, launch checks and device queries for the C-ABI library.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void ig_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
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

extern "C" {

const char* ig_last_error(void) { return g_err; }

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
