// Gather-form implicit-GEMM convolution on 16x16 MFMA tiles ("gconv").
//
// One description covers every convolution of the model that is not the hot
// factorized 3x1/1x3 pair (those have their own LDS-tiled kernel in conv1d.hip, and
// fall back to this one for shapes that kernel does not take):
//
//   for a base grid (n, i, j), i < Hb, j < Wb, and each phase ph:
//     out[n][i*So + oh_ph][j*So + ow_ph][co] =
//        act( scale[co] * sum_{t < ntaps_ph} sum_ci  Wpk_ph[co][t][ci] * x[n][i*Si + dh_t][j*Si + dw_t][ci]
//             + shift[co] (+ residual) )
//   with x read as zero outside [0,Hin) x [0,Win).
//
//   strided conv 3x3 s2 p1     : Si=2, So=1, one phase, 9 taps (dh,dw) in {-1,0,1}^2
//   transposed conv 3x3 s2 p1+1: Si=1, So=2, four phases (output parity), 1/2/2/4 taps
//   transposed conv 2x2 s2     : Si=1, So=2, four phases, one tap each
//   1x1 conv                   : one phase, one tap
//   3x1 / 1x3 dilated          : one phase, taps (-d,0,d) on one axis
//
// Weights arrive pre-packed as MFMA A-fragments: [phase][ct][kstep][lane][16 B]
// (see pack.hip).  The B fragments (pixels) are gathered straight from global memory,
// 16 bytes per lane; there is no LDS staging here.
#pragma once
#include "common.cuh"

#define GC_MAX_TAPS 9
#define GC_MAX_PHASES 4

struct GConvPhase {
    int oh, ow, ntaps;
    int dh[GC_MAX_TAPS];
    int dw[GC_MAX_TAPS];
    int w_chunk_off;        // offset of this phase's packed weights, in 16-byte units
    int ksteps;             // ceil(ntaps * CH / 4)
};

struct GConvParams {
    const void* x;
    void* y;
    const void* w;
    const float* scale;     // may be null (= 1)
    const float* shift;     // may be null (= 0)
    const void* res;        // residual, same geometry as y (NHWC modes only); may be null
    const void* gate;       // NHWC mode: output zeroed where gate <= 0, before the residual add; may be null
    int N, Hin, Win, ldx, xoff;     // input: channel stride and first channel
    int Hb, Wb;                     // base grid
    int Hout, Wout, ldy, yoff;      // output: channel stride and first channel (NHWC) / total channels (NCHW)
    int Cout;                       // valid output channels
    int Si, So;
    int relu;
    int out_mode;                   // 0: NHWC as T, 1: NCHW as T, 2: NCHW as fp32
    int nphase;
    GConvPhase ph[GC_MAX_PHASES];
};

template <typename T, int CIN, int CT, int PT>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvParams p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = CIN / EPC;               // 16-byte chunks per pixel
    static_assert(CIN % EPC == 0, "Cin must be a whole number of 16-byte chunks");
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = lane >> 4, r16 = lane & 15;
    const GConvPhase& ph = p.ph[blockIdx.y];

    __shared__ int s_dh[GC_MAX_TAPS + 1], s_dw[GC_MAX_TAPS + 1];
    if (threadIdx.x < GC_MAX_TAPS) {
        s_dh[threadIdx.x] = ph.dh[threadIdx.x];
        s_dw[threadIdx.x] = ph.dw[threadIdx.x];
    }
    __syncthreads();

    const long long P = (long long)p.N * p.Hb * p.Wb;
    const long long tile0 = ((long long)blockIdx.x * 4 + wave) * (16 * PT);

    // this lane's pixels (one per pixel tile)
    int pn[PT], pi[PT], pj[PT];
    bool pv[PT];
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        long long pix = tile0 + t * 16 + r16;
        pv[t] = pix < P;
        long long pp = pv[t] ? pix : 0;
        pj[t] = (int)(pp % p.Wb);
        long long rest = pp / p.Wb;
        pi[t] = (int)(rest % p.Hb);
        pn[t] = (int)(rest / p.Hb);
    }

    f32x4 acc[CT][PT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};

    const i32x4* wbase = (const i32x4*)p.w + ph.w_chunk_off;
    const T* xb = (const T*)p.x + p.xoff;
    const int nchunks = ph.ntaps * CH;
    const int KS = ph.ksteps;

    // The gathers are unconditional (masked lanes read pixel 0 and are zeroed afterwards) and two K-steps are taken per
    // trip, so that their requests are in flight together instead of one memory round trip per step.
    constexpr int U = 2;
    for (int ks0 = 0; ks0 < KS; ks0 += U) {
        i32x4 b[U][PT], a[U][CT];
        bool ok[U][PT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int ks = ks0 + u;
            const int g = ks * 4 + q;
            const bool gv = ks < KS && g < nchunks;
            const int tap = gv ? g / CH : 0;
            const int chunk = g % CH;
            const int dh = s_dh[tap], dw = s_dw[tap];
#pragma unroll
            for (int t = 0; t < PT; ++t) {
                const int hi = pi[t] * p.Si + dh, wi = pj[t] * p.Si + dw;
                ok[u][t] = gv && pv[t] && hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win;
                const long long off = ok[u][t] ? (((long long)pn[t] * p.Hin + hi) * p.Win + wi) * p.ldx + chunk * EPC : 0;
                b[u][t] = *(const i32x4*)(xb + off);
            }
            const int ksc = ks < KS ? ks : KS - 1;               // the odd tail re-reads the last step's weights (its pixels are zero)
#pragma unroll
            for (int c = 0; c < CT; ++c) a[u][c] = wbase[((long long)c * KS + ksc) * 64 + lane];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int t = 0; t < PT; ++t) b[u][t] = ok[u][t] ? b[u][t] : i32x4{0, 0, 0, 0};
#pragma unroll
            for (int c = 0; c < CT; ++c)
#pragma unroll
                for (int t = 0; t < PT; ++t) acc[c][t] = mma_step<T>(a[u][c], b[u][t], acc[c][t]);
        }
    }

    // epilogue: lane holds channels c*16 + 4q + {0..3} of pixel tile t, pixel r16
#pragma unroll
    for (int t = 0; t < PT; ++t) {
        if (!pv[t]) continue;
        const int ho = pi[t] * p.So + ph.oh, wo = pj[t] * p.So + ph.ow;
        const long long opix = ((long long)pn[t] * p.Hout + ho) * p.Wout + wo;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int co = c * 16 + q * 4;
            if (co >= p.Cout) continue;
            f32x4 v = acc[c][t];
            f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            const bool full = co + 4 <= p.Cout;
            if (full) {
                if (p.scale) sc = *(const f32x4*)(p.scale + co);
                if (p.shift) sh = *(const f32x4*)(p.shift + co);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < p.Cout) {
                        if (p.scale) sc[e] = p.scale[co + e];
                        if (p.shift) sh[e] = p.shift[co + e];
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
            if (p.out_mode == 0) {
                T* yp = (T*)p.y + opix * p.ldy + p.yoff + co;
                if (full) {
                    if (p.gate) {
                        f32x4 gv = load4<T>((const T*)p.gate + opix * p.ldy + p.yoff + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] : 0.f;
                    }
                    if (p.res) {
                        f32x4 rv = load4<T>((const T*)p.res + opix * p.ldy + p.yoff + co);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] += rv[e];
                    }
                    if (p.relu)
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    store4<T>(yp, v);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (co + e < p.Cout) {
                            float o = v[e];
                            if (p.gate && !(Elem<T>::ld((const T*)p.gate + opix * p.ldy + p.yoff + co + e) > 0.f)) o = 0.f;
                            if (p.res) o += Elem<T>::ld((const T*)p.res + opix * p.ldy + p.yoff + co + e);
                            if (p.relu) o = fmaxf(o, 0.f);
                            Elem<T>::st(yp + e, o);
                        }
                }
            } else {
                // NCHW planes: [n][co][ho][wo]
                const long long plane = (long long)p.Hout * p.Wout;
                const long long base = ((long long)pn[t] * p.ldy + p.yoff + co) * plane + (long long)ho * p.Wout + wo;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (co + e < p.Cout) {
                        float o = p.relu ? fmaxf(v[e], 0.f) : v[e];
                        if (p.out_mode == 2) ((float*)p.y)[base + e * plane] = o;
                        else Elem<T>::st((T*)p.y + base + e * plane, o);
                    }
            }
        }
    }
}

// host-side launcher: picks the <CIN, CT, PT> instance
int gconv_launch(const GConvParams& p, int dtype, int cin, hipStream_t stream);
// number of 16-byte chunks the packed weights of one phase occupy
static inline long long gconv_phase_chunks(int ct, int ksteps) { return (long long)ct * ksteps * 64; }
static inline int gconv_ksteps(int dtype, int cin, int ntaps) {
    const int epc = dtype == 1 ? 8 : 4;
    return (ntaps * (cin / epc) + 3) / 4;
}
