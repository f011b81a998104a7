// Building blocks shared by the LDS-tiled factorized-convolution kernels.
#pragma once
#include "common.cuh"
#include <type_traits>

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

static __device__ __attribute__((aligned(1024))) char g_zero_page[1024];     // never written

template <int PIXB>
__device__ __forceinline__ int swz(int pix, int chunk) {       // an involution: swz(pix, swz(pix, c)) == c
    constexpr int CH = PIXB / 16;
    if constexpr (CH >= 16) {
        return ((chunk & 15) ^ (pix & 15)) | (chunk & ~15);
    } else {
        constexpr int PPR = CH >= 8 ? 1 : 8 / CH;      // LDS serves 128 bytes a cycle: 8 consecutive pixels must hit 8 distinct 16-byte slots
        return chunk ^ ((pix / PPR) & (CH - 1));
    }
}

template <int IMM>
__device__ __forceinline__ i32x4 lds_read_b128(int addr) {
    i32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void tie_reg(i32x4& v) { asm volatile("" : "+v"(v)); }

#define ERF_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <typename T> struct ChunkOps;
template <> struct ChunkOps<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void unpack(const i32x4& r, float* f) {
        const f32x4 v = __builtin_bit_cast(f32x4, r);
#pragma unroll
        for (int e = 0; e < 4; ++e) f[e] = v[e];
    }
    __device__ static __forceinline__ i32x4 pack(const float* f) {
        f32x4 v{f[0], f[1], f[2], f[3]};
        return __builtin_bit_cast(i32x4, v);
    }
};
template <> struct ChunkOps<bf16raw> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const i32x4& r, float* f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = __uint_as_float(((uint32_t)r[e]) << 16);
            f[2 * e + 1] = __uint_as_float(((uint32_t)r[e]) & 0xFFFF0000u);
        }
    }
    __device__ static __forceinline__ i32x4 pack(const float* f) {
        i32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = pack_bf16x2(f[2 * e], f[2 * e + 1]);
        return r;
    }
};


// ---- division of tile indices by launch constants without the ~40-instruction emulated divide:
// q = (mulhi(n, mul) + n) >> shr  (Granlund-Montgomery round-up form; exact for n < 2^31).
struct FastDiv {
    uint32_t mul, shr, div;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    f.shr = l;
    f.mul = (uint32_t)((((uint64_t)1 << 32) * (((uint64_t)1 << l) - d)) / d + 1);
    f.div = d;
    return f;
}
__device__ __forceinline__ uint32_t fast_div(uint32_t n, const FastDiv& f) { return (__umulhi(n, f.mul) + n) >> f.shr; }

// ---- raw buffer addressing: a 32-bit per-lane byte offset plus a scalar byte offset; a lane whose
// offset is >= the buffer size reads zeros / stores nothing, which is how out-of-image pixels are masked.
#define ERF_OOB 0x80000000u
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}
