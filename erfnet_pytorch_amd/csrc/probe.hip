// Peak probes: what this part sustains on the two resources every roofline fraction in this repository is priced against.
//
//   erf_probe_mfma  a matrix-instruction-only loop on every SIMD of the chip: eight independent accumulator chains per wave,
//                   two waves per SIMD, operands in registers, nothing else in the loop.  bf16: v_mfma_f32_16x16x32_bf16
//                   (16384 flop per instruction), fp32: v_mfma_f32_16x16x4_f32 (2048 flop).  The caller times the launch and
//                   divides erf_probe_mfma_flop() by it.  dtype 2: the bf16 loop with operands that differ from lane to lane,
//                   register to register and chain to chain (mantissas from a hash), i.e. with the switching activity of real
//                   data -- the constant, mostly-zero operands of dtype 1 are the cheapest case for the part's power management.
//                   Workgroup 0 also leaves, in sink[1] and sink[2], how far s_memtime and s_memrealtime (100 MHz) moved over
//                   its loop: the tick rate of the counter the kernels' cycle stamps use.
//   erf_probe_copy  a streaming copy, 16 bytes per lane per access, four accesses in flight per lane, grid-stride: HBM read +
//                   write bandwidth.
//
// No counterpart in the reference (it has no kernels of its own); these exist for the measurement section of bench.py.
#include "common.cuh"

namespace {

constexpr int PROBE_THREADS = 512;          // 8 waves per workgroup = 2 per SIMD
constexpr int PROBE_CHAINS = 8;

__device__ __forceinline__ float probe_value(unsigned key) {      // +-2^-6 * [1, 2): a different mantissa for every key
    key = key * 2654435761u + 0x9E3779B9u;
    key ^= key >> 15;
    key *= 2246822519u;
    key ^= key >> 13;
    return __uint_as_float((key & 0x807F0000u) | 0x3C800000u);
}

template <bool BF16, bool VARIED>
__global__ __launch_bounds__(PROBE_THREADS) void probe_mfma_kernel(int iters, float* __restrict__ sink) {
    f32x4 acc[PROBE_CHAINS];
#pragma unroll
    for (int i = 0; i < PROBE_CHAINS; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x;
    // small finite operands: the accumulators stay finite over any iteration count (products are 0 or +-2^-20; varied: sums
    // of +-2^-12 products with random signs)
    const float va = (lane & 1) ? 0.0009765625f : 0.f, vb = (lane & 2) ? -0.0009765625f : 0.0009765625f;
    i32x4 a[4], b[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (VARIED) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a[r][e] = pack_bf16x2(probe_value(lane * 64 + r * 8 + e * 2), probe_value(lane * 64 + r * 8 + e * 2 + 1));
                b[r][e] = pack_bf16x2(probe_value(lane * 64 + 32 + r * 8 + e * 2), probe_value(lane * 64 + 32 + r * 8 + e * 2 + 1));
            }
        } else if (BF16) {
            const int pa = pack_bf16x2(va, va), pb = pack_bf16x2(vb, vb);
            a[r] = i32x4{pa, pa, pa, pa};
            b[r] = i32x4{pb, pb, pb, pb};
        } else {
            a[r] = __builtin_bit_cast(i32x4, f32x4{va, va, va, va});
            b[r] = __builtin_bit_cast(i32x4, f32x4{vb, vb, vb, vb});
        }
        asm volatile("" : "+v"(a[r]), "+v"(b[r]));
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < PROBE_CHAINS; ++i) {
            if (BF16) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i & 3]), __builtin_bit_cast(bf16x8, b[(i >> 1) & 3]),
                                                                 acc[i], 0, 0, 0);
            } else {
                const f32x4 af = __builtin_bit_cast(f32x4, a[i & 3]), bf = __builtin_bit_cast(f32x4, b[(i >> 1) & 3]);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc[i], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PROBE_CHAINS; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s == 12345.678f) sink[0] = s;                       // keeps the chains alive; never true
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sink[1] = (float)(t1 - t0);
        sink[2] = (float)(r1 - r0);
    }
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
    if (dtype < 0 || dtype > 2 || workgroups <= 0 || iters <= 0) return 0;
    const long long per_instr = dtype != 0 ? 2LL * 16 * 16 * 32 : 2LL * 16 * 16 * 4;
    return per_instr * PROBE_CHAINS * (PROBE_THREADS / 64) * (long long)workgroups * iters;
}

int erf_probe_mfma(int dtype, int workgroups, int iters, float* sink, void* stream) {
    if (dtype < 0 || dtype > 2 || workgroups <= 0 || workgroups > 65536 || iters <= 0 || !sink) return ERF_EINVAL;
    if (dtype == 2)
        hipLaunchKernelGGL((probe_mfma_kernel<true, true>), dim3(workgroups), dim3(PROBE_THREADS), 0, (hipStream_t)stream, iters, sink);
    else if (dtype == 1)
        hipLaunchKernelGGL((probe_mfma_kernel<true, false>), dim3(workgroups), dim3(PROBE_THREADS), 0, (hipStream_t)stream, iters, sink);
    else
        hipLaunchKernelGGL((probe_mfma_kernel<false, false>), dim3(workgroups), dim3(PROBE_THREADS), 0, (hipStream_t)stream, iters, sink);
    return erf_check_launch();
}

int erf_probe_copy(const void* src, void* dst, long long bytes, int workgroups, void* stream) {
    if (!src || !dst || bytes <= 0 || (bytes & 15) || workgroups <= 0 || workgroups > 1 << 20) return ERF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return ERF_EINVAL;
    hipLaunchKernelGGL(probe_copy_kernel, dim3(workgroups), dim3(256), 0, (hipStream_t)stream, (const i32x4*)src, (i32x4*)dst, bytes / 16);
    return erf_check_launch();
}

}  // extern "C"
