// erf_bn_stats_finalize_static: erf_bn_stats_finalize for callers that cannot allocate (code under HIP-graph
// capture, tight serving loops, harnesses that forbid hipMalloc).  Same kernels, same results; the per-workgroup
// partial sums go to a fixed array in the code object's own device memory instead of a caller workspace.
//
// The array exists once per GPU the library runs on (each device loads its own copy of the code object) and keeps
// its size until the process ends.  In host code its NAME is only a handle: the device address is looked up with
// hipGetSymbolAddress, once per device, with that device current, and only that address is ever handed to a kernel.
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU (two streams, two threads) race.
// Nothing has to start at zero -- the reduction kernel writes every partial row the finalize kernel reads.
#include "../../include/erfnet_mi355.h"
#include "common.cuh"
#include "train.cuh"
#include <mutex>

#define ERF_STATIC_MAX_C 128
#define ERF_STATIC_GROUPS RED_MAX_GROUPS
#define ERF_STATIC_BYTES (ERF_STATIC_GROUPS * 2 * ERF_STATIC_MAX_C * 8 + 4 * ERF_STATIC_MAX_C * 4)

static __device__ __align__(256) unsigned char scratch[ERF_STATIC_BYTES];

#define ERF_MAX_DEVICES 64
static void* g_scratch_ptr[ERF_MAX_DEVICES];
static std::mutex g_scratch_lock;

// Device address of this GPU's copy of `scratch` (nullptr: the lookup failed).
static void* scratch_of_current_device() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= ERF_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> hold(g_scratch_lock);
    if (!g_scratch_ptr[dev]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        g_scratch_ptr[dev] = ptr;
    }
    return g_scratch_ptr[dev];
}

extern "C" {

int erf_static_scratch_max_channels(void) { return ERF_STATIC_MAX_C; }

int erf_bn_stats_finalize_static(const void* z, int dtype, long long pixels, int C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                                 float* scale, float* shift, void* stream) {
    if (!z || !gamma || !beta || !mean || !invstd || !scale || !shift || (dtype != ERF_F32 && dtype != ERF_BF16) || pixels <= 0)
        return ERF_EINVAL;
    if (C <= 0 || C > ERF_STATIC_MAX_C) return ERF_ETOOLARGE;         // the fixed array holds 512 partial rows of 2 * 128 doubles
    if (bn_workspace_bytes(C) > (size_t)ERF_STATIC_BYTES) return ERF_ETOOLARGE;
    void* ws = scratch_of_current_device();
    if (!ws) return ERF_ELAUNCH;
    int groups = 0;
    int rc = bn_stats_launch(z, dtype, pixels, C, (float*)ws, &groups, (hipStream_t)stream);
    if (rc) return rc;
    return bn_finalize_launch((const float*)ws, groups, C, pixels, gamma, beta, running_mean, running_var, momentum, eps, mean,
                              invstd, scale, shift, (hipStream_t)stream);
}

}  // extern "C"
