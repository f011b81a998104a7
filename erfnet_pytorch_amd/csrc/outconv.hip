// Decoder.output_conv: ConvTranspose2d(16, nc, 2, stride 2) with the per-pixel arg-max fused.
//
// Every input pixel owns its 2x2 output patch, so the layer is one small GEMM per pixel:
// 16 input channels -> 4 * nc outputs (phase = 2a + b, the position inside the patch, x class).
// The weight rows are packed so that MFMA tile c, row 4*phase + e is class 4c + e: after the MFMAs
// lane (q, r) holds ALL classes of phase q of input pixel r, i.e. of output pixel (2i + (q>>1), 2j + (q&1)).
// The arg-max over the classes therefore needs no cross-lane traffic at all, and the labels come from
// the fp32 accumulators -- the logits need not be materialised.  Byte labels of four horizontally
// adjacent output pixels are gathered with two lane exchanges and leave as one 4-byte store.
#include "common.cuh"
#include "misc.cuh"
#include "conv1d_tile.cuh"

// STAGE (fp32 logits, Win a multiple of 16): a tile's 16 input pixels are 32 consecutive output columns of two rows; the
// wave parks its logits in LDS as [class][row][32 columns] and writes them back as 16-byte pieces, 128 contiguous bytes
// per (class, row), instead of one scattered 4-byte store per lane and class.
// LOGITS = false: labels only (the inference path) -- no per-class branches at all: classes beyond nc (last channel tile
// only) are masked by a select, so a tile is ~150 straight-line instructions instead of ~90 wave-uniform branches.
template <typename T, int CT, int PT, bool STAGE, bool LOGITS = true>
__global__ __launch_bounds__(256) void outconv_kernel(const T* __restrict__ x, const i32x4* __restrict__ w,
                                                      const float* __restrict__ bias, void* __restrict__ logits,
                                                      int logits_mode, void* __restrict__ labels, int label_bytes, int N,
                                                      int Hin, int Win, int nc, FastDiv fW, FastDiv fH) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = 16 / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const unsigned P = (unsigned)N * Hin * Win;
    const unsigned tile0 = (blockIdx.x * 4 + wave) * (16 * PT);
    const int Hout = 2 * Hin, Wout = 2 * Win;

    __shared__ float stage[STAGE ? 4 * CT * 4 * 64 : 1];
    float* const st = stage + (STAGE ? wave * CT * 4 * 64 : 0);

    i32x4 a[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a[c] = w[c * 64 + lane];
    float bs[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[c][e] = 4 * c + e < nc ? bias[4 * c + e] : 0.f;

    // all PT pixel fragments are requested before the first is used (unconditional loads from clamped addresses, masked
    // afterwards: a load under a branch would be waited for tile by tile)
    i32x4 bq[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const unsigned pix = tile0 + t * 16 + r16;
        bq[t] = *(const i32x4*)(x + (long long)(pix < P ? pix : 0) * 16 + (q < CH ? q : 0) * EPC);
    }
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const unsigned pix = tile0 + t * 16 + r16;
        const bool pv = pix < P;
        const i32x4 b = (pv && q < CH) ? bq[t] : i32x4{0, 0, 0, 0};
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mma_step<T>(a[c], b, f32x4{0.f, 0.f, 0.f, 0.f});

        const unsigned pp = pv ? pix : 0;
        const unsigned rest = fast_div(pp, fW);
        const int j = (int)(pp - rest * fW.div);
        const unsigned n = fast_div(rest, fH);
        const int i = (int)(rest - n * fH.div);
        const int oh = 2 * i + (q >> 1), ow = 2 * j + (q & 1);

        float bv = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cls = 4 * c + e;
                if constexpr (!LOGITS) {
                    float val = acc[c][e] + bs[c][e];
                    if (c == CT - 1 && e > 0) val = cls < nc ? val : -INFINITY;      // ct = ceil(nc / 4): only these can lie beyond nc
                    const bool better = val > bv;                // first maximum wins, as torch.max(1)
                    bv = better ? val : bv;
                    bi = better ? cls : bi;
                } else if (cls <= 4 * (CT - 1) || cls < nc) {    // wave-uniform; folds at compile time except for the last tile's classes 1..3
                    const float val = acc[c][e] + bs[c][e];
                    if constexpr (STAGE) {
                        st[cls * 64 + (q >> 1) * 32 + 2 * r16 + (q & 1)] = val;
                    } else if (logits_mode && pv) {
                        const long long o = (((long long)n * nc + cls) * Hout + oh) * Wout + ow;
                        if (logits_mode == 2) ((float*)logits)[o] = val;
                        else Elem<T>::st((T*)logits + o, val);
                    }
                    if (val > bv) { bv = val; bi = cls; }        // first maximum wins, as torch.max(1)
                }
            }
        if constexpr (STAGE) {
            // the tile is whole (P is a multiple of 16) and lies in one input row: lane 0's pixel gives its origin
            const unsigned p0 = tile0 + t * 16;
            __builtin_amdgcn_wave_barrier();               // the wave's own LDS traffic is in order: its writes are visible to its reads
            if (p0 < P) {
                const unsigned rest0 = fast_div(p0, fW);
                const int j0 = (int)(p0 - rest0 * fW.div);
                const unsigned n0 = fast_div(rest0, fH);
                const int i0 = (int)(rest0 - n0 * fH.div);
                const int sub = lane >> 4, row = (lane >> 3) & 1, col4 = (lane & 7) * 4;
                for (int c0 = 0; c0 < nc; c0 += 4) {
                    const int cls = c0 + sub;
                    if (cls < nc) {
                        const f32x4 v = *(const f32x4*)(st + cls * 64 + row * 32 + col4);
                        *(f32x4*)((float*)logits + (((long long)n0 * nc + cls) * Hout + 2 * i0 + row) * Wout + 2 * j0 + col4) = v;
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();               // ... and the next tile's writes stay behind these reads
        }
        if (labels) {
            const long long o = ((long long)n * Hout + oh) * Wout + ow;
            if (label_bytes == 1) {
                // bytes of (b = 0, b = 1) x (j, j + 1): j even and Win even keep the four in one row and one tile
                const int right = __shfl_xor(bi, 16, 64);
                const int pair = bi | (right << 8);
                const int next = __shfl_down(pair, 1, 64);
                if (pv && !(q & 1) && !(r16 & 1)) *(unsigned int*)((unsigned char*)labels + o) = (unsigned)(pair | (next << 16));
            } else if (pv) {
                ((long long*)labels)[o] = (long long)bi;
            }
        }
    }
}

template <typename T>
static int outconv_dispatch(int ct, dim3 grid, hipStream_t stream, const T* x, const i32x4* w, const float* bias,
                            void* logits, int logits_mode, void* labels, int label_bytes, int N, int Hin, int Win, int nc) {
    const FastDiv fW = make_fastdiv(Win), fH = make_fastdiv(Hin);
#define OC(CTV)                                                                                                     \
    case CTV:                                                                                                       \
        if (logits_mode == 2 && (Win & 15) == 0)                                                                    \
            hipLaunchKernelGGL((outconv_kernel<T, CTV, 4, true>), grid, dim3(256), 0, stream, x, w, bias, logits,       \
                               logits_mode, labels, label_bytes, N, Hin, Win, nc, fW, fH);                              \
        else if (logits_mode == 0)                                                                                  \
            hipLaunchKernelGGL((outconv_kernel<T, CTV, 4, false, false>), grid, dim3(256), 0, stream, x, w, bias,       \
                               logits, logits_mode, labels, label_bytes, N, Hin, Win, nc, fW, fH);                      \
        else                                                                                                        \
            hipLaunchKernelGGL((outconv_kernel<T, CTV, 4, false>), grid, dim3(256), 0, stream, x, w, bias, logits,      \
                               logits_mode, labels, label_bytes, N, Hin, Win, nc, fW, fH);                              \
        break;
    switch (ct) {
        OC(1) OC(2) OC(3) OC(4) OC(5) OC(6) OC(7) OC(8)
        default: return ERF_EUNSUPPORTED;
    }
#undef OC
    return erf_check_launch();
}

int outconv_launch(const void* x, const void* packed, const float* bias, void* logits, int logits_f32, void* labels,
                   int label_bytes, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream) {
    if (nc < 1 || nc > 32) return ERF_EUNSUPPORTED;
    if (labels && label_bytes != 1 && label_bytes != 8) return ERF_EINVAL;
    if (Win & 1) return ERF_EUNSUPPORTED;                       // the 4-byte label stores pair columns j, j + 1
    const int ct = (nc + 3) / 4;
    const long long P = (long long)N * Hin * Win;
    if (P >= (1LL << 31)) return ERF_EUNSUPPORTED;
    const long long gx = (P + 4 * 16 * 4 - 1) / (4 * 16 * 4);
    if (gx <= 0) return ERF_EINVAL;
    const int mode = logits ? (logits_f32 ? 2 : 1) : 0;
    if (dtype == 1)
        return outconv_dispatch<bf16raw>(ct, dim3((unsigned)gx), stream, (const bf16raw*)x, (const i32x4*)packed, bias,
                                         logits, mode, labels, label_bytes, N, Hin, Win, nc);
    return outconv_dispatch<float>(ct, dim3((unsigned)gx), stream, (const float*)x, (const i32x4*)packed, bias, logits, mode,
                                   labels, label_bytes, N, Hin, Win, nc);
}
