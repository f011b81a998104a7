// Shared device helpers for the gfx950 ERFNet kernels.
//
// Activations are NHWC ("pixel-major": all channels of one pixel are contiguous) in
// either fp32 or bf16.  Every contraction runs on 16x16 MFMA tiles with OUTPUT
// CHANNELS on the MFMA row axis and PIXELS on the column axis:
//
//     D[co][pixel] += sum_k  W[co][k] * X[k][pixel],   k = (tap, ci)
//
// so that after the MFMA each lane owns 4 consecutive output channels of one pixel
// (C/D map: col = lane & 15, row = 4 * (lane >> 4) + reg) and can store them as one
// 8-byte (bf16) or 16-byte (fp32) piece of that pixel's channel vector.
//
// One K-step consumes, per lane, one 16-byte "chunk" of consecutive input channels
// (8 bf16 or 4 fp32) of the lane's pixel; the four 16-lane groups q = lane >> 4 hold
// four consecutive chunks.  For bf16 that is exactly one v_mfma_f32_16x16x32_bf16
// (k = 8q + e).  For fp32 it is four v_mfma_f32_16x16x4_f32, instruction e taking
// element e of every lane's chunk (k = 4q + e within the 16-wide step) -- the k order
// inside a step is a fixed permutation shared by both operands, so the sum is exact fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

typedef uint16_t bf16raw;

#define ERF_OK 0
#define ERF_EINVAL (-1)
#define ERF_EUNSUPPORTED (-2)
#define ERF_ELAUNCH (-3)
#define ERF_ETOOLARGE (-4)

__device__ __forceinline__ float bf16_to_f32(bf16raw v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16raw f32_to_bf16(float f) {
    __bf16 h = (__bf16)f;                       // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
    return __builtin_bit_cast(bf16raw, h);
}

// Two / four fp32 -> packed bf16 (RNE), one v_cvt_pk_bf16_f32 per pair.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ int pack_bf16x2(float lo, float hi) {
    const f32x2 v{lo, hi};
    return __builtin_bit_cast(int, __builtin_convertvector(v, bf16x2_t));
}
// ReLU of two packed bf16 values: bf16 is sign-magnitude, so max with 0 as 16-bit integers clears exactly the negative ones.
typedef __attribute__((ext_vector_type(2))) short i16x2;
__device__ __forceinline__ int relu_bf16x2(int packed) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2, packed), i16x2{0, 0}));
}
// max(packed pair, floor pair) as 16-bit integers: floor 0 is ReLU, floor 0x80008000 (-32768 twice) the identity, so a launch
// argument can switch the activation without a branch.  Pass SCALARS: a bit cast of a vector element misreads with this compiler.
__device__ __forceinline__ int max_bf16x2(int packed, int floor_pk) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(i16x2, packed), __builtin_bit_cast(i16x2, floor_pk)));
}
__device__ __forceinline__ i32x2 pack_bf16x4(const f32x4& v) {
    i32x2 r;
    r[0] = pack_bf16x2(v[0], v[1]);
    r[1] = pack_bf16x2(v[2], v[3]);
    return r;
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int EPC = 4;               // elements per 16-byte chunk
    static constexpr int KSTEP = 16;            // input channels one K-step consumes
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
    __device__ static __forceinline__ float to_f32(float v) { return v; }
    __device__ static __forceinline__ float from_f32(float v) { return v; }
};
template <> struct Elem<bf16raw> {
    static constexpr int EPC = 8;
    static constexpr int KSTEP = 32;
    __device__ static __forceinline__ float ld(const bf16raw* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16raw* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ float to_f32(bf16raw v) { return bf16_to_f32(v); }
    __device__ static __forceinline__ bf16raw from_f32(float v) { return f32_to_bf16(v); }
};

// acc += A(16 x Kstep) * B(Kstep x 16), fragments as 16 raw bytes per lane.
template <typename T>
__device__ __forceinline__ f32x4 mma_step(const i32x4& a, const i32x4& b, f32x4 acc);
template <>
__device__ __forceinline__ f32x4 mma_step<bf16raw>(const i32x4& a, const i32x4& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma_step<float>(const i32x4& a, const i32x4& b, f32x4 acc) {
    f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], acc, 0, 0, 0);
    return acc;
}

// 4 consecutive channels of one pixel: load / store as one vector access.
template <typename T> __device__ __forceinline__ f32x4 load4(const T* p);
template <> __device__ __forceinline__ f32x4 load4<float>(const float* p) { return *(const f32x4*)p; }
template <> __device__ __forceinline__ f32x4 load4<bf16raw>(const bf16raw* p) {
    i32x2 r = *(const i32x2*)p;
    f32x4 o;
    o[0] = __uint_as_float(((uint32_t)r[0]) << 16);
    o[1] = __uint_as_float(((uint32_t)r[0]) & 0xFFFF0000u);
    o[2] = __uint_as_float(((uint32_t)r[1]) << 16);
    o[3] = __uint_as_float(((uint32_t)r[1]) & 0xFFFF0000u);
    return o;
}
template <typename T> __device__ __forceinline__ void store4(T* p, f32x4 v);
template <> __device__ __forceinline__ void store4<float>(float* p, f32x4 v) { *(f32x4*)p = v; }
template <> __device__ __forceinline__ void store4<bf16raw>(bf16raw* p, f32x4 v) {
    *(i32x2*)p = pack_bf16x4(v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Developer knobs (tile-count sweeps, kernel bypasses, phase ablation) are read from the environment only in builds
// made with -DERF_DEBUG.  The product library never looks at the environment: every call computes what its arguments say.
#ifdef ERF_DEBUG
#include <stdlib.h>
static inline int erf_debug_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int erf_debug_env(const char*, int dflt) { return dflt; }
#endif

static inline int erf_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ERF_OK : ERF_ELAUNCH;
}
