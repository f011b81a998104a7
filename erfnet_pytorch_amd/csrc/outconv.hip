// Decoder.output_conv: ConvTranspose2d(16, nc, 2, stride 2) with the per-pixel arg-max fused.
//
// Every input pixel owns its 2x2 output patch, so the layer is one small GEMM per pixel:
// 16 input channels -> 4 * NCP "virtual channels" v = phase * NCP + class, phase = 2a + b the
// position inside the patch, NCP = nc rounded up to a multiple of 4 (so the 4 accumulator
// registers of a lane always belong to one phase).  With output channels on the MFMA row axis
// a lane ends up with 4 consecutive classes of one phase of one input pixel; the arg-max over
// the classes of a phase is finished with two cross-lane exchanges (lane ^ 16, lane ^ 32).
// Labels come from the fp32 accumulators, so the logits need not be materialised at all.
#include "common.cuh"
#include "misc.cuh"

template <typename T, int CT, int PT>
__global__ __launch_bounds__(256) void outconv_kernel(const T* __restrict__ x, const i32x4* __restrict__ w,
                                                      const float* __restrict__ bias, void* __restrict__ logits,
                                                      int logits_mode, void* __restrict__ labels, int label_bytes, int N,
                                                      int Hin, int Win, int nc, int ncp) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = 16 / EPC;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const long long P = (long long)N * Hin * Win;
    const long long tile0 = ((long long)blockIdx.x * 4 + wave) * (16 * PT);
    const int Hout = 2 * Hin, Wout = 2 * Win;

    i32x4 a[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) a[c] = w[c * 64 + lane];

#pragma unroll
    for (int t = 0; t < PT; ++t) {
        const long long pix = tile0 + t * 16 + r16;
        const bool pv = pix < P;
        i32x4 b = {0, 0, 0, 0};
        if (pv && q < CH) b = *(const i32x4*)(x + pix * 16 + q * EPC);
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mma_step<T>(a[c], b, f32x4{0.f, 0.f, 0.f, 0.f});

        const long long pp = pv ? pix : 0;
        const int j = (int)(pp % Win);
        const long long rest = pp / Win;
        const int i = (int)(rest % Hin);
        const int n = (int)(rest / Hin);

        float bv[4];
        int bi[4];
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) { bv[ph] = -INFINITY; bi[ph] = 0x7FFFFFFF; }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int v0 = c * 16 + q * 4;
            const int ph = v0 / ncp, cls0 = v0 % ncp;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cls = cls0 + e;
                const bool ok = ph < 4 && cls < nc;
                const float val = acc[c][e] + (ok ? bias[cls] : 0.f);
                if (ok && pv && logits_mode) {
                    const long long o = (((long long)n * nc + cls) * Hout + 2 * i + (ph >> 1)) * Wout + 2 * j + (ph & 1);
                    if (logits_mode == 2) ((float*)logits)[o] = val;
                    else Elem<T>::st((T*)logits + o, val);
                }
#pragma unroll
                for (int pq = 0; pq < 4; ++pq)
                    if (ok && ph == pq && val > bv[pq]) { bv[pq] = val; bi[pq] = cls; }
            }
        }
        if (labels) {
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    const float ov = __shfl_xor(bv[pq], o, 64);
                    const int oi = __shfl_xor(bi[pq], o, 64);
                    if (ov > bv[pq] || (ov == bv[pq] && oi < bi[pq])) { bv[pq] = ov; bi[pq] = oi; }
                }
            }
            int mine = bi[0];
#pragma unroll
            for (int pq = 1; pq < 4; ++pq) mine = (q == pq) ? bi[pq] : mine;
            if (pv) {
                const long long o = ((long long)n * Hout + 2 * i + (q >> 1)) * Wout + 2 * j + (q & 1);
                if (label_bytes == 1) ((unsigned char*)labels)[o] = (unsigned char)mine;
                else ((long long*)labels)[o] = (long long)mine;
            }
        }
    }
}

template <typename T>
static int outconv_dispatch(int ct, dim3 grid, hipStream_t stream, const T* x, const i32x4* w, const float* bias,
                            void* logits, int logits_mode, void* labels, int label_bytes, int N, int Hin, int Win, int nc,
                            int ncp) {
#define OC(CTV)                                                                                                     \
    case CTV:                                                                                                       \
        hipLaunchKernelGGL((outconv_kernel<T, CTV, 2>), grid, dim3(256), 0, stream, x, w, bias, logits, logits_mode, \
                           labels, label_bytes, N, Hin, Win, nc, ncp);                                              \
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
    const int ncp = (nc + 3) & ~3;
    const int ct = (4 * ncp + 15) / 16;
    const long long P = (long long)N * Hin * Win;
    const long long gx = (P + 4 * 16 * 2 - 1) / (4 * 16 * 2);
    if (gx <= 0 || gx > 0x7FFFFFFFLL) return ERF_EINVAL;
    const int mode = logits ? (logits_f32 ? 2 : 1) : 0;
    if (dtype == 1)
        return outconv_dispatch<bf16raw>(ct, dim3((unsigned)gx), stream, (const bf16raw*)x, (const i32x4*)packed, bias,
                                         logits, mode, labels, label_bytes, N, Hin, Win, nc, ncp);
    return outconv_dispatch<float>(ct, dim3((unsigned)gx), stream, (const float*)x, (const i32x4*)packed, bias, logits, mode,
                                   labels, label_bytes, N, Hin, Win, nc, ncp);
}
