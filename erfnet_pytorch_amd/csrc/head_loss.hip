// Decoder.output_conv + weighted cross entropy as one pass each way (training):
//
//   forward : logits of a 2x2 output patch are formed in registers exactly as in outconv.hip (lane (q, r) holds all
//             classes of phase q of input pixel r) and go straight into  num += w[t] * (logsumexp(z) - z[t]),
//             den += w[t]  -- the [N, nc, 2H, 2W] fp32 logits tensor is never written or read back;
//   backward: the logits are formed again, dlogit = coef * w[t] * (softmax(z) - onehot(t)) leaves as the bf16 / fp32
//             "patch" tensor [N, H, W, ld] (channel 4 * class + phase) that the output layer's weight- and data-gradient
//             kernels take, staged per wave in LDS so that a tile's 16 pixels leave as one contiguous run.
//
// Follows train/erfnet.py:171,179 (ConvTranspose2d(16, nc, 2, stride 2)) and train/main.py:121,202 (CrossEntropyLoss(weight)
// on targets[:, 0]).  Per-block partial (num, den) pairs are summed by ce_finalize (fixed order).
#include "common.cuh"
#include "misc.cuh"
#include "train.cuh"
#include "conv1d_tile.cuh"

#define HL_PT 4
// Class loops: with ct = ceil(nc / 4) channel tiles only classes 4 (ct - 1) + 1 .. + 3 can lie beyond nc, so `k <= 4 (CT - 1)`
// folds at compile time and a tile carries 3 wave-uniform branches per loop instead of 4 ct.

template <typename T, int CT>
__device__ __forceinline__ void head_logits(const T* __restrict__ x, const i32x4 (&a)[CT], const float (&bs)[CT][4], unsigned pix,
                                            bool pv, int q, float (&val)[CT * 4]) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = 16 / EPC;
    i32x4 b = {0, 0, 0, 0};
    if (pv && q < CH) b = *(const i32x4*)(x + (long long)pix * 16 + q * EPC);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
        const f32x4 acc = mma_step<T>(a[c], b, f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int e = 0; e < 4; ++e) val[4 * c + e] = acc[e] + bs[c][e];
    }
}

template <typename T, int CT, bool BWD>
__global__ __launch_bounds__(256) void head_ce_kernel(const T* __restrict__ x, const i32x4* __restrict__ w,
                                                      const float* __restrict__ bias, const long long* __restrict__ target,
                                                      const float* __restrict__ weight, float* __restrict__ partials,
                                                      const float* __restrict__ coef_ptr, T* __restrict__ patches, int ld, int N,
                                                      int Hin, int Win, int nc, FastDiv fW, FastDiv fH) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const unsigned P = (unsigned)N * Hin * Win;
    const unsigned tile0 = (blockIdx.x * 4 + wave) * (16 * HL_PT);
    const int Hout = 2 * Hin, Wout = 2 * Win;
    extern __shared__ __attribute__((aligned(16))) char hl_smem[];
    T* const st = (T*)hl_smem + (BWD ? wave * 16 * ld : 0);          // backward: [16 pixels][ld] of this wave

    i32x4 a[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a[c] = w[c * 64 + lane];
    float bs[CT][4];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) bs[c][e] = 4 * c + e < nc ? bias[4 * c + e] : 0.f;
    const float coef = BWD ? coef_ptr[0] : 0.f;
    if (BWD) {                                                       // channels >= 4 * nc of a patch stay zero
        for (int i = lane; i < 16 * ld; i += 64) st[i] = Elem<T>::from_f32(0.f);
        __builtin_amdgcn_wave_barrier();
    }
    float num = 0.f, den = 0.f;

#pragma unroll
    for (int t = 0; t < HL_PT; ++t) {
        const unsigned pix = tile0 + t * 16 + r16;
        const bool pv = pix < P;
        float val[CT * 4];
        head_logits<T, CT>(x, a, bs, pix, pv, q, val);
        const unsigned pp = pv ? pix : 0;
        const unsigned rest = fast_div(pp, fW);
        const int j = (int)(pp - rest * fW.div);
        const unsigned n = fast_div(rest, fH);
        const int i = (int)(rest - n * fH.div);
        const long long o = ((long long)n * Hout + 2 * i + (q >> 1)) * Wout + 2 * j + (q & 1);
        const long long tg64 = pv ? target[o] : 0;
        const int tg = (tg64 >= -100 && tg64 < 0x7fffffff) ? (int)tg64 : -1;
        float m = -INFINITY;
#pragma unroll
        for (int k = 0; k < CT * 4; ++k)
            if (k <= 4 * (CT - 1) || k < nc) m = fmaxf(m, val[k]);
        float s = 0.f, zt = 0.f;
#pragma unroll
        for (int k = 0; k < CT * 4; ++k)
            if (k <= 4 * (CT - 1) || k < nc) {
                s += __expf(val[k] - m);
                zt = k == tg ? val[k] : zt;
            }
        // a label outside [0, nc) other than torch's ignore_index (-100) makes torch raise; here it poisons the sums with NaN
        const bool tv = pv && tg >= 0 && tg < nc;
        const float wt = tv ? (weight ? weight[tg] : 1.f) : ((!pv || tg == -100) ? 0.f : __builtin_nanf(""));
        if (!BWD) {
            num += wt * (m + __logf(s) - zt);
            den += wt;
        } else {
            const float inv = 1.f / s, cw = coef * wt;
#pragma unroll
            for (int k = 0; k < CT * 4; ++k)
                if (k <= 4 * (CT - 1) || k < nc) st[r16 * ld + 4 * k + q] = Elem<T>::from_f32(cw * (__expf(val[k] - m) * inv - (k == tg ? 1.f : 0.f)));
            __builtin_amdgcn_wave_barrier();
            // the tile's 16 pixels x ld channels are one contiguous run of the patch tensor
            const unsigned p0 = tile0 + t * 16;
            const int chunks = 16 * ld * (int)sizeof(T) / 16;
            for (int c = lane; c < chunks; c += 64) {
                const int px = c * 16 / (ld * (int)sizeof(T));
                if (p0 + px < P) *((i32x4*)(patches + (long long)p0 * ld) + c) = *((const i32x4*)st + c);
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (!BWD) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { num += __shfl_xor(num, off, 64); den += __shfl_xor(den, off, 64); }
        float* red = (float*)hl_smem;
        if (lane == 0) { red[2 * wave] = num; red[2 * wave + 1] = den; }
        __syncthreads();
        if (threadIdx.x == 0) {
            partials[2 * blockIdx.x] = (red[0] + red[2]) + (red[4] + red[6]);
            partials[2 * blockIdx.x + 1] = (red[1] + red[3]) + (red[5] + red[7]);
        }
    }
}

static long long head_blocks(long long P) { return (P + 4 * 16 * HL_PT - 1) / (4 * 16 * HL_PT); }

size_t head_ce_workspace_bytes(int N, int Hin, int Win) { return (size_t)head_blocks((long long)N * Hin * Win) * 2 * sizeof(float); }

template <typename T, bool BWD>
static int head_ce_dispatch(int ct, unsigned grid, size_t lds, hipStream_t stream, const T* x, const i32x4* w, const float* bias,
                            const long long* target, const float* weight, float* partials, const float* coef, T* patches, int ld,
                            int N, int Hin, int Win, int nc) {
    const FastDiv fW = make_fastdiv(Win), fH = make_fastdiv(Hin);
#define HC(CTV)                                                                                                            \
    case CTV:                                                                                                              \
        hipLaunchKernelGGL((head_ce_kernel<T, CTV, BWD>), dim3(grid), dim3(256), lds, stream, x, w, bias, target, weight,   \
                           partials, coef, patches, ld, N, Hin, Win, nc, fW, fH);                                          \
        break;
    switch (ct) {
        HC(1) HC(2) HC(3) HC(4) HC(5) HC(6) HC(7) HC(8)
        default: return ERF_EUNSUPPORTED;
    }
#undef HC
    return erf_check_launch();
}

// out3 = (sum w*nll, sum w, their ratio); workspace: head_ce_workspace_bytes
int head_ce_fwd_launch(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                       float* out3, float* workspace, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream) {
    if (nc < 1 || nc > 32) return ERF_EUNSUPPORTED;
    const long long P = (long long)N * Hin * Win;
    if (P >= (1LL << 31) || head_blocks(P) > 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    const unsigned grid = (unsigned)head_blocks(P);
    const int ct = (nc + 3) / 4;
    int rc;
    if (dtype == 1)
        rc = head_ce_dispatch<bf16raw, false>(ct, grid, 64, stream, (const bf16raw*)x, (const i32x4*)packed, bias, target, weight,
                                              workspace, nullptr, nullptr, 0, N, Hin, Win, nc);
    else
        rc = head_ce_dispatch<float, false>(ct, grid, 64, stream, (const float*)x, (const i32x4*)packed, bias, target, weight,
                                            workspace, nullptr, nullptr, 0, N, Hin, Win, nc);
    if (rc) return rc;
    return ce_partials_finalize_launch(workspace, (long long)grid, out3, stream);
}

int head_ce_bwd_launch(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                       const float* coef, void* patches, int ld, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream) {
    if (nc < 1 || nc > 32 || ld < 4 * nc || (ld & 15)) return ERF_EUNSUPPORTED;
    const long long P = (long long)N * Hin * Win;
    if (P >= (1LL << 31)) return ERF_EUNSUPPORTED;
    const unsigned grid = (unsigned)head_blocks(P);
    const int ct = (nc + 3) / 4;
    const size_t lds = (size_t)4 * 16 * ld * (dtype == 1 ? 2 : 4);
    if (dtype == 1)
        return head_ce_dispatch<bf16raw, true>(ct, grid, lds, stream, (const bf16raw*)x, (const i32x4*)packed, bias, target, weight,
                                               nullptr, coef, (bf16raw*)patches, ld, N, Hin, Win, nc);
    return head_ce_dispatch<float, true>(ct, grid, lds, stream, (const float*)x, (const i32x4*)packed, bias, target, weight, nullptr,
                                         coef, (float*)patches, ld, N, Hin, Win, nc);
}
