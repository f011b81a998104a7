// Bandwidth-bound helper kernels: the 3->16 stem downsampler, the max-pool half of the
// 16->64 / 64->128 downsamplers, layout conversion, arg-max and the IoU histogram.
#include "common.cuh"
#include "misc.cuh"

// ------------------------------------------------------------------------------------------
// Stem: NCHW [N,3,Hin,Win] -> NHWC [N,Hin/2,Win/2,16].  One thread per output pixel:
// 13 conv3x3-s2 channels (27 inputs each) and 3 max-pool channels, affine + ReLU, one
// 32-byte (bf16) / 64-byte (fp32) store.  The 351 weights sit in LDS and are read as
// wave-wide broadcasts.
// uint8 images (TIN = uint8_t; LAY 0: planar [N,3,H,W], 1: interleaved [N,H,W,3], the layout image decoders deliver) are
// converted on load the way the reference's input transform does it (ToTensor(), train/main.py:47,66 / eval/eval_iou.py:31:
// value / 255 as a correctly rounded fp32 division): the 256 possible quotients are formed once per workgroup in LDS.
template <typename TIN, typename T, int LAY = 0>
__global__ __launch_bounds__(256) void stem_kernel(const TIN* __restrict__ x, T* __restrict__ y,
                                                   const float* __restrict__ w, const float* __restrict__ scale,
                                                   const float* __restrict__ shift, int N, int Hin, int Win, int relu,
                                                   float* __restrict__ raw_out) {
    __shared__ float lut[sizeof(TIN) == 1 ? 256 : 1];
    if constexpr (sizeof(TIN) == 1) {
        lut[threadIdx.x] = (float)threadIdx.x / 255.0f;
        __syncthreads();
    }
    // The 351 weights and 32 affine coefficients are read at wave-uniform, compile-time offsets, i.e. through the
    // scalar cache into SGPRs: every FMA takes its weight as a scalar operand (no LDS traffic, no VGPRs).
    const int Ho = Hin >> 1, Wo = Win >> 1;
    const long long P = (long long)N * Ho * Wo;
    const long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= P) return;
    const int wo = (int)(pix % Wo);
    const long long rest = pix / Wo;
    const int ho = (int)(rest % Ho);
    const int n = (int)(rest / Ho);
    float in[27];
    const long long plane = (long long)Hin * Win;
    const TIN* xb = x + (long long)n * 3 * plane;
    if constexpr (sizeof(TIN) == 4) {
        // fp32 image: the three taps of a row come from two aligned 8-byte loads, columns (2wo-2, 2wo-1) and (2wo, 2wo+1):
        // a wave's lanes then cover one contiguous 512-byte run per instruction, 18 instructions instead of 27 four-byte
        // loads two elements apart -- the kernel was bound by the address path of those, not by memory or arithmetic.
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int hi = 2 * ho - 1 + th;
                const bool rowok = hi >= 0 && hi < Hin;
                const float* row = (const float*)xb + c * plane + (long long)(rowok ? hi : 0) * Win + 2 * wo;
                const f32x2 lft = *(const f32x2*)(wo > 0 ? row - 2 : row);
                const f32x2 rgt = *(const f32x2*)row;
                in[c * 9 + th * 3 + 0] = (rowok && wo > 0) ? lft[1] : 0.f;
                in[c * 9 + th * 3 + 1] = rowok ? rgt[0] : 0.f;
                in[c * 9 + th * 3 + 2] = rowok ? rgt[1] : 0.f;
            }
    } else if constexpr (sizeof(TIN) == 1 && LAY == 0) {
        // planar bytes: the three taps of a row are bytes 1..3 of ONE (unaligned) 4-byte load at column 2wo-2 (bytes 0..1 of the
        // load at column 0 for the first output column, whose left tap is the zero padding): 9 loads instead of 27
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int th = 0; th < 3; ++th) {
                const int hi = 2 * ho - 1 + th;
                const bool rowok = hi >= 0 && hi < Hin;
                const uint8_t* row = (const uint8_t*)xb + c * plane + (long long)(rowok ? hi : 0) * Win + 2 * wo;
                unsigned word;
                __builtin_memcpy(&word, wo > 0 ? row - 2 : row, 4);
                if (wo == 0) word <<= 16;
                in[c * 9 + th * 3 + 0] = (rowok && wo > 0) ? lut[(word >> 8) & 255u] : 0.f;
                in[c * 9 + th * 3 + 1] = rowok ? lut[(word >> 16) & 255u] : 0.f;
                in[c * 9 + th * 3 + 2] = rowok ? lut[word >> 24] : 0.f;
            }
    } else if constexpr (sizeof(TIN) == 1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int th = 0; th < 3; ++th)
#pragma unroll
                for (int tw = 0; tw < 3; ++tw) {
                    const int hi = 2 * ho - 1 + th, wi = 2 * wo - 1 + tw;
                    const bool ok = hi >= 0 && hi < Hin && wi >= 0 && wi < Win;
                    const long long at = LAY ? ((long long)(ok ? hi : 0) * Win + (ok ? wi : 0)) * 3 + c
                                             : c * plane + (long long)(ok ? hi : 0) * Win + (ok ? wi : 0);
                    const float v = lut[xb[at]];
                    in[c * 9 + th * 3 + tw] = ok ? v : 0.f;
                }
    } else {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
            for (int tw = 0; tw < 3; ++tw) {
                const int hi = 2 * ho - 1 + th, wi = 2 * wo - 1 + tw;
                float v = 0.f;
                if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win) v = Elem<TIN>::ld(xb + c * plane + (long long)hi * Win + wi);
                in[c * 9 + th * 3 + tw] = v;
            }
    }
    float out[16];
#pragma unroll
    for (int co = 0; co < 13; ++co) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 27; ++k) a = fmaf(w[co * 27 + k], in[k], a);
        out[co] = a;
    }
    // max-pool window (2ho..2ho+1, 2wo..2wo+1) = taps (th,tw) in {1,2}^2 of the 3x3 patch
#pragma unroll
    for (int c = 0; c < 3; ++c)
        out[13 + c] = fmaxf(fmaxf(in[c * 9 + 4], in[c * 9 + 5]), fmaxf(in[c * 9 + 7], in[c * 9 + 8]));
    if (raw_out) {
#pragma unroll
        for (int c = 0; c < 16; ++c) raw_out[pix * 16 + c] = out[c];
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = out[c] * (scale ? scale[c] : 1.f) + (shift ? shift[c] : 0.f);
        out[c] = relu ? fmaxf(v, 0.f) : v;
    }
    T* yp = y + pix * 16;
    if constexpr (sizeof(T) == 2) {                    // the pixel's 32 bytes as two 16-byte stores
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            i32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(out[8 * h + 2 * e], out[8 * h + 2 * e + 1]);
            *(i32x4*)(yp + 8 * h) = v;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; c += 4) store4<T>(yp + c, f32x4{out[c], out[c + 1], out[c + 2], out[c + 3]});
    }
}

int stem_launch(const void* x, int x_dtype, void* y, const float* w, const float* scale, const float* shift, int dtype,
                int N, int Hin, int Win, int relu, hipStream_t stream) {
    if ((Hin | Win) & 1) return ERF_EINVAL;
    const long long P = (long long)N * (Hin / 2) * (Win / 2);
    const unsigned grid = (unsigned)((P + 255) / 256);
#define STEM(TIN, T) hipLaunchKernelGGL((stem_kernel<TIN, T>), dim3(grid), dim3(256), 0, stream, (const TIN*)x, (T*)y, w, scale, shift, N, Hin, Win, relu, (float*)nullptr)
    if (x_dtype == 0 && dtype == 0) STEM(float, float);
    else if (x_dtype == 0 && dtype == 1) STEM(float, bf16raw);
    else if (x_dtype == 1 && dtype == 1) STEM(bf16raw, bf16raw);
    else if (x_dtype == 1 && dtype == 0) STEM(bf16raw, float);
#define STEM8(T, LAY) hipLaunchKernelGGL((stem_kernel<uint8_t, T, LAY>), dim3(grid), dim3(256), 0, stream, (const uint8_t*)x, (T*)y, w, scale, shift, N, Hin, Win, relu, (float*)nullptr)
    else if (x_dtype == 2 && dtype == 0) STEM8(float, 0);
    else if (x_dtype == 2 && dtype == 1) STEM8(bf16raw, 0);
    else if (x_dtype == 3 && dtype == 0) STEM8(float, 1);
    else if (x_dtype == 3 && dtype == 1) STEM8(bf16raw, 1);
#undef STEM8
    else return ERF_EUNSUPPORTED;
#undef STEM
    return erf_check_launch();
}

// ------------------------------------------------------------------------------------------
// Max-pool 2x2 half of a downsampler: x NHWC [N,Hin,Win,Cin] -> channels [yoff, yoff+Cin) of
// y NHWC [N,Hin/2,Win/2,ldy], with the block's affine + ReLU applied.  4 channels per thread.
template <typename T>
__global__ __launch_bounds__(256) void pool_affine_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          int N, int Hin, int Win, int Cin, int ldy, int yoff, int relu) {
    const int Ho = Hin >> 1, Wo = Win >> 1, C4 = Cin >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C4) * 4;
    const long long pix = idx / C4;
    const int wo = (int)(pix % Wo);
    const long long rest = pix / Wo;
    const int ho = (int)(rest % Ho);
    const int n = (int)(rest / Ho);
    const T* p00 = x + (((long long)n * Hin + 2 * ho) * Win + 2 * wo) * Cin + c;
    f32x4 a = load4<T>(p00), b = load4<T>(p00 + Cin), d = load4<T>(p00 + (long long)Win * Cin),
          e = load4<T>(p00 + (long long)Win * Cin + Cin);
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float m = fmaxf(fmaxf(a[k], b[k]), fmaxf(d[k], e[k]));
        const float sc = scale ? scale[yoff + c + k] : 1.f, sh = shift ? shift[yoff + c + k] : 0.f;
        m = m * sc + sh;
        v[k] = relu ? fmaxf(m, 0.f) : m;
    }
    store4<T>(y + pix * ldy + yoff + c, v);
}

int pool_affine_launch(const void* x, void* y, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
                       int Cin, int ldy, int yoff, int relu, hipStream_t stream) {
    if (((Hin | Win) & 1) || (Cin & 3) || (yoff & 3) || (ldy & 3)) return ERF_EINVAL;
    const long long total = (long long)N * (Hin / 2) * (Win / 2) * (Cin / 4);
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == 1)
        hipLaunchKernelGGL(pool_affine_kernel<bf16raw>, dim3(grid), dim3(256), 0, stream, (const bf16raw*)x, (bf16raw*)y,
                           scale, shift, N, Hin, Win, Cin, ldy, yoff, relu);
    else
        hipLaunchKernelGGL(pool_affine_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, (float*)y, scale,
                           shift, N, Hin, Win, Cin, ldy, yoff, relu);
    return erf_check_launch();
}

// ------------------------------------------------------------------------------------------
// Layout conversion through a 32x(32+1) LDS tile: [n][c][hw] <-> [n][hw][c].
template <typename TS, typename TD>
__global__ void transpose_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long long R, long long Cc,
                                 unsigned tiles_c, unsigned tiles_r) {
    // src: [n][R][Cc] -> dst: [n][Cc][R]; blockIdx.x enumerates (n, tile_r, tile_c)
    __shared__ float tile[32][33];
    const unsigned tc = blockIdx.x % tiles_c;
    const unsigned rest = blockIdx.x / tiles_c;
    const unsigned tr = rest % tiles_r;
    const unsigned n = rest / tiles_r;
    const long long c0 = (long long)tc * 32;
    const long long r0 = (long long)tr * 32;
    const TS* s = src + (long long)n * R * Cc;
    TD* d = dst + (long long)n * R * Cc;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const long long r = r0 + i;
        const long long c = c0 + threadIdx.x;
        if (r < R && c < Cc) tile[i][threadIdx.x] = Elem<TS>::ld(s + r * Cc + c);
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const long long c = c0 + i;
        const long long r = r0 + threadIdx.x;
        if (r < R && c < Cc) Elem<TD>::st(d + c * R + r, tile[threadIdx.x][i]);
    }
}

int transpose_launch(const void* src, int sdt, void* dst, int ddt, int N, long long R, long long Cc, hipStream_t stream) {
    const long long tiles_c = (Cc + 31) / 32, tiles_r = (R + 31) / 32;
    const long long blocks = tiles_c * tiles_r * N;
    if (blocks <= 0 || blocks > 0x7FFFFFFFLL) return ERF_EINVAL;
    dim3 grid((unsigned)blocks), block(32, 8);
#define TR(TS, TD) hipLaunchKernelGGL((transpose_kernel<TS, TD>), grid, block, 0, stream, (const TS*)src, (TD*)dst, R, Cc, (unsigned)tiles_c, (unsigned)tiles_r)
    if (sdt == 0 && ddt == 0) TR(float, float);
    else if (sdt == 0 && ddt == 1) TR(float, bf16raw);
    else if (sdt == 1 && ddt == 0) TR(bf16raw, float);
    else if (sdt == 1 && ddt == 1) TR(bf16raw, bf16raw);
    else return ERF_EUNSUPPORTED;
#undef TR
    return erf_check_launch();
}

// ------------------------------------------------------------------------------------------
// arg-max over the class planes of NCHW logits; first maximum wins (strict > while scanning up).
template <typename T, typename TL>
__global__ __launch_bounds__(256) void argmax_kernel(const T* __restrict__ z, TL* __restrict__ lab, int C, long long HW,
                                                     long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long n = idx / HW, hw = idx % HW;
    const T* p = z + n * C * HW + hw;
    float best = Elem<T>::ld(p);
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = Elem<T>::ld(p + c * HW);
        if (v > best) { best = v; bi = c; }
    }
    lab[idx] = (TL)bi;
}

int argmax_launch(const void* z, int dtype, void* lab, int label_bytes, int N, int C, long long HW, hipStream_t stream) {
    const long long total = (long long)N * HW;
    const unsigned grid = (unsigned)((total + 255) / 256);
#define AM(T, TL) hipLaunchKernelGGL((argmax_kernel<T, TL>), dim3(grid), dim3(256), 0, stream, (const T*)z, (TL*)lab, C, HW, total)
    if (dtype == 0 && label_bytes == 8) AM(float, long long);
    else if (dtype == 0 && label_bytes == 1) AM(float, unsigned char);
    else if (dtype == 1 && label_bytes == 8) AM(bf16raw, long long);
    else if (dtype == 1 && label_bytes == 1) AM(bf16raw, unsigned char);
    else return ERF_EUNSUPPORTED;
#undef AM
    return erf_check_launch();
}

// ------------------------------------------------------------------------------------------
// ToLabel() + Relabel(olabel, nlabel) of the loaders (train/transform.py:53-68): 8-bit label maps as decoded from the
// label images -> int64 class indices, with one value mapped to another.  16 labels per thread and trip.
__global__ __launch_bounds__(256) void labels_from_u8_kernel(const unsigned char* __restrict__ src, long long* __restrict__ dst,
                                                             long long count, int olabel, int nlabel) {
    const long long groups = count >> 4;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += stride) {
        const i32x4 v = *(const i32x4*)(src + g * 16);
#pragma unroll
        for (int k = 0; k < 16; k += 2) {
            int a = (v[k >> 2] >> (8 * (k & 3))) & 255, b = (v[k >> 2] >> (8 * (k & 3) + 8)) & 255;
            a = a == olabel ? nlabel : a;
            b = b == olabel ? nlabel : b;
            *(i32x4*)(dst + g * 16 + k) = i32x4{a, 0, b, 0};
        }
    }
    for (long long i = groups * 16 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int a = src[i];
        dst[i] = a == olabel ? nlabel : a;
    }
}

int labels_from_u8_launch(const void* src, long long* dst, long long count, int olabel, int nlabel, hipStream_t stream) {
    if (count == 0) return ERF_OK;
    long long blocks = (count / 16 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(labels_from_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned char*)src, dst, count, olabel, nlabel);
    return erf_check_launch();
}

// ------------------------------------------------------------------------------------------
// Confusion-matrix histogram: per-workgroup nc x nc counters in LDS (atomics), flushed with one
// global 64-bit atomic per non-zero cell.  Grid-stride, 8 pixels per thread per trip.
template <typename TP, typename TLB>
__global__ __launch_bounds__(1024) void iou_hist_kernel(const TP* __restrict__ pred, const TLB* __restrict__ label,
                                                       long long count, int nc, int olabel, unsigned nlabel,
                                                       unsigned long long* __restrict__ hist) {
    // one private copy of the counters per wave (less contention on the LDS atomics); a thread takes 8 consecutive
    // pixels per trip (vector loads) and merges runs of equal (label, prediction) pairs before touching LDS --
    // label maps are piecewise constant, so most trips end in one or two atomics
    __shared__ unsigned int sh[4][32 * 32 + 1];
    const int cells = nc * nc + 1;
    for (int i = threadIdx.x; i < 4 * (32 * 32 + 1); i += blockDim.x) (&sh[0][0])[i] = 0;
    __syncthreads();
    unsigned int* mine = sh[(threadIdx.x >> 6) & 3];
    const long long groups = count >> 3;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long gidx = (long long)blockIdx.x * blockDim.x + threadIdx.x; gidx < groups; gidx += stride) {
        TP pv[8];
        TLB lv[8];
        if constexpr (sizeof(TP) == 1) *(unsigned long long*)pv = *(const unsigned long long*)(pred + gidx * 8);
        else {
#pragma unroll
            for (int k = 0; k < 8; k += 2) *(i32x4*)(pv + k) = *(const i32x4*)(pred + gidx * 8 + k);
        }
        if constexpr (sizeof(TLB) == 1) *(unsigned long long*)lv = *(const unsigned long long*)(label + gidx * 8);
        else {
#pragma unroll
            for (int k = 0; k < 8; k += 2) *(i32x4*)(lv + k) = *(const i32x4*)(label + gidx * 8 + k);
        }
        // keys in 32-bit arithmetic (a 64-bit value is in range iff its high word is zero and its low word is below nc);
        // equal neighbours are merged without branches: a pixel adds its run length when the next key differs
        int key[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long g = (unsigned long long)(long long)lv[k], q = (unsigned long long)(long long)pv[k];
            const bool re = olabel >= 0 && (long long)g == (long long)olabel;               // Relabel(olabel, nlabel) on the target
            const unsigned glo = re ? nlabel : (unsigned)g, qlo = (unsigned)q;
            const bool ok = (unsigned)(g >> 32) == 0u && (unsigned)(q >> 32) == 0u && glo < (unsigned)nc && qlo < (unsigned)nc;
            key[k] = ok ? (int)(glo * (unsigned)nc + qlo) : nc * nc;
        }
        unsigned int run = 1;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            if (key[k] != key[k + 1]) { atomicAdd(&mine[key[k]], run); run = 1; }
            else ++run;
        }
        atomicAdd(&mine[key[7]], run);
    }
    // tail (count not a multiple of 8)
    for (long long i = groups * 8 + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        long long g = (long long)label[i];
        const long long q = (long long)pred[i];
        if (olabel >= 0 && g == olabel) g = nlabel;
        atomicAdd(&mine[(g >= 0 && g < nc && q >= 0 && q < nc) ? (int)(g * nc + q) : nc * nc], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
        const unsigned int v = sh[0][i] + sh[1][i] + sh[2][i] + sh[3][i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
}

int iou_hist_launch(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count, int nc,
                    int olabel, int nlabel, long long* hist, hipStream_t stream) {
    if (nc < 1 || nc > 32 || count < 0) return ERF_EINVAL;
    if (olabel >= 0 && (nlabel < 0 || nlabel >= nc)) return ERF_EINVAL;          // olabel < 0: no relabelling
    if (count == 0) return ERF_OK;
    // One workgroup of 16 waves per CU at most: every workgroup ends with one 64-bit global atomic per non-zero cell, and
    // atomics on one address are served one after the other -- 1024 small workgroups spent most of the launch there.
    long long blocks = (count + 1024 * 8 * 4 - 1) / (1024 * 8 * 4);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
#define IH(TP, TLB) hipLaunchKernelGGL((iou_hist_kernel<TP, TLB>), dim3((unsigned)blocks), dim3(1024), 0, stream, (const TP*)pred, (const TLB*)label, count, nc, olabel, (unsigned)nlabel, (unsigned long long*)hist)
    if (pred_bytes == 1 && label_bytes == 8) IH(unsigned char, long long);
    else if (pred_bytes == 8 && label_bytes == 8) IH(long long, long long);
    else if (pred_bytes == 1 && label_bytes == 1) IH(unsigned char, unsigned char);
    else if (pred_bytes == 8 && label_bytes == 1) IH(long long, unsigned char);
    else return ERF_EUNSUPPORTED;
#undef IH
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// iouEval.addBatch for inputs given as N x nClasses x H x W planes (one-hot or soft scores) instead of label maps:
//     tp_c += sum x_c * y_c,  fp_c += sum x_c * (1 - y_c - y_ignore),  fn_c += sum (1 - x_c) * y_c      (c < classes)
// Either side may still be a label map (kind 1: int64, kind 2: uint8), which is its one-hot expansion on the fly.
// blockIdx.y = class, blockIdx.x = pixel slice; fp64 sums per thread, one partial row per block, summed in block order.
#define IOU_PLANE_ROWS 256
template <int XK, int YK>
__global__ __launch_bounds__(256) void iou_planes_kernel(const void* __restrict__ xv, const void* __restrict__ yv, int N, int nc,
                                                         long long HW, int ignore, double* __restrict__ partials) {
    const int c = blockIdx.y, classes = gridDim.y;
    const long long total = (long long)N * HW;
    double tp = 0, fp = 0, fn = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / HW, hw = i - n * HW;
        float x, y, ig = 0.f;
        if (XK == 0) x = ((const float*)xv)[(n * nc + c) * HW + hw];
        else if (XK == 1) x = ((const long long*)xv)[i] == c ? 1.f : 0.f;
        else x = ((const unsigned char*)xv)[i] == c ? 1.f : 0.f;
        if (YK == 0) {
            y = ((const float*)yv)[(n * nc + c) * HW + hw];
            if (ignore >= 0) ig = ((const float*)yv)[(n * nc + ignore) * HW + hw];
        } else {
            const long long t = YK == 1 ? ((const long long*)yv)[i] : (long long)((const unsigned char*)yv)[i];
            y = t == c ? 1.f : 0.f;
            ig = (ignore >= 0 && t == ignore) ? 1.f : 0.f;
        }
        tp += (double)(x * y);
        fp += (double)(x * (1.f - y - ig));
        fn += (double)((1.f - x) * y);
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = tp; red[1][threadIdx.x] = fp; red[2][threadIdx.x] = fn;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x < 3) partials[((long long)blockIdx.x * 3 + threadIdx.x) * classes + c] = red[threadIdx.x][0];
}

__global__ void iou_planes_finalize_kernel(const double* __restrict__ partials, int rows, int classes, double* __restrict__ sums) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;             // (kind, class)
    if (i >= 3 * classes) return;
    double a = 0;
    for (int r = 0; r < rows; ++r) a += partials[(long long)r * 3 * classes + i];
    sums[i] += a;
}

size_t iou_planes_workspace_bytes(int nc) { return (size_t)IOU_PLANE_ROWS * 3 * (nc > 0 ? nc : 1) * sizeof(double); }

int iou_planes_launch(const void* x, int x_kind, const void* y, int y_kind, int N, int nc, long long HW, int ignore, double* sums,
                      double* workspace, hipStream_t stream) {
    if (nc <= 0 || nc > 64 || ignore >= nc || x_kind < 0 || x_kind > 2 || y_kind < 0 || y_kind > 2) return ERF_EINVAL;
    const int classes = ignore >= 0 ? ignore : nc;                    // the reference keeps channels [0, ignoreIndex)
    if (classes <= 0) return ERF_EINVAL;
    const long long total = (long long)N * HW;
    long long rows = (total + 256 * 16 - 1) / (256 * 16);
    if (rows > IOU_PLANE_ROWS) rows = IOU_PLANE_ROWS;
    if (rows < 1) rows = 1;
    const dim3 grid((unsigned)rows, (unsigned)classes);
#define IP(XK, YK) hipLaunchKernelGGL((iou_planes_kernel<XK, YK>), grid, dim3(256), 0, stream, x, y, N, nc, HW, ignore, workspace)
    switch (x_kind * 3 + y_kind) {
        case 0: IP(0, 0); break; case 1: IP(0, 1); break; case 2: IP(0, 2); break;
        case 3: IP(1, 0); break; case 4: IP(1, 1); break; case 5: IP(1, 2); break;
        case 6: IP(2, 0); break; case 7: IP(2, 1); break; default: IP(2, 2); break;
    }
#undef IP
    hipLaunchKernelGGL(iou_planes_finalize_kernel, dim3((3 * classes + 63) / 64), dim3(64), 0, stream, workspace, (int)rows, classes, sums);
    return erf_check_launch();
}
