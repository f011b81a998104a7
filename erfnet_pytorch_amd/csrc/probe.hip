// Peak probes: what this part sustains on the two resources every roofline fraction in this repository is priced against.
//
//   erf_probe_mfma  a matrix-instruction-only loop on every SIMD of the chip: eight independent accumulator chains per wave,
//                   two waves per SIMD, operands in registers, nothing else in the loop.  bf16: v_mfma_f32_16x16x32_bf16
//                   (16384 flop per instruction), fp32: v_mfma_f32_16x16x4_f32 (2048 flop).  The caller times the launch and
//                   divides erf_probe_mfma_flop() by it.
//   erf_probe_copy  a streaming copy, 16 bytes per lane per access, four accesses in flight per lane, grid-stride: HBM read +
//                   write bandwidth.
//
// No counterpart in the reference (it has no kernels of its own); these exist for the measurement section of bench.py.
#include "common.cuh"

namespace {

constexpr int PROBE_THREADS = 512;          // 8 waves per workgroup = 2 per SIMD
constexpr int PROBE_CHAINS = 8;

template <bool BF16>
__global__ __launch_bounds__(PROBE_THREADS) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
    f32x4 acc[PROBE_CHAINS];
#pragma unroll
    for (int i = 0; i < PROBE_CHAINS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x;
    // small finite operands: the accumulators stay finite over any iteration count (products are 0 or +-2^-20)
    const float va = (lane & 1) ? 0.0009765625f : 0.f, vb = (lane & 2) ? -0.0009765625f : 0.0009765625f;
    i32x4 a, b;
    if (BF16) {
        const int pa = pack_bf16x2(va, va), pb = pack_bf16x2(vb, vb);
        a = i32x4{pa, pa, pa, pa};
        b = i32x4{pb, pb, pb, pb};
    } else {
        a = __builtin_bit_cast(i32x4, f32x4{va, va, va, va});
        b = __builtin_bit_cast(i32x4, f32x4{vb, vb, vb, vb});
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < PROBE_CHAINS; ++i) {
            if (BF16) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PROBE_CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) sink[0] = s;                       // keeps the chains alive; never true
}

__global__ __launch_bounds__(256) void probe_copy_kernel(const i32x4* __restrict__ src, i32x4* __restrict__ dst, long long chunks) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < chunks; i += 4 * stride) {
        const i32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const i32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        __builtin_nontemporal_store(a, dst + i);
        __builtin_nontemporal_store(b, dst + i + stride);
        __builtin_nontemporal_store(c, dst + i + 2 * stride);
        __builtin_nontemporal_store(d, dst + i + 3 * stride);
    }
    for (; i < chunks; i += stride) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

extern "C" {

long long erf_probe_mfma_flop(int dtype, int workgroups, int iters) {
    if ((dtype != 0 && dtype != 1) || workgroups <= 0 || iters <= 0) return 0;
    const long long per_instr = dtype == 1 ? 2LL * 16 * 16 * 32 : 2LL * 16 * 16 * 4;
    return per_instr * PROBE_CHAINS * (PROBE_THREADS / 64) * (long long)workgroups * iters;
}

int erf_probe_mfma(int dtype, int workgroups, int iters, float* sink, void* stream) {
    if ((dtype != 0 && dtype != 1) || workgroups <= 0 || workgroups > 65536 || iters <= 0 || !sink) return ERF_EINVAL;
    if (dtype == 1)
        hipLaunchKernelGGL(probe_mfma_kernel<true>, dim3(workgroups), dim3(PROBE_THREADS), 0, (hipStream_t)stream, iters, sink);
    else
        hipLaunchKernelGGL(probe_mfma_kernel<false>, dim3(workgroups), dim3(PROBE_THREADS), 0, (hipStream_t)stream, iters, sink);
    return erf_check_launch();
}

int erf_probe_copy(const void* src, void* dst, long long bytes, int workgroups, void* stream) {
    if (!src || !dst || bytes <= 0 || (bytes & 15) || workgroups <= 0 || workgroups > 1 << 20) return ERF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return ERF_EINVAL;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src, (i32x4*)dst, bytes / 16);
    return erf_check_launch();
}

}  // extern "C"
