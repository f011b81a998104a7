// Weight gradients of every convolution as a gather-form MFMA GEMM whose contraction index is
// the PIXEL:   dW[row][tap*CIN + ci] = sum_p dz[outpix(p)][row] * x[inpix(p, tap)][ci].
//
// Per workgroup: a contiguous range of base pixels, walked in steps of 32.  Each step stages
//   A = dz rows      [32 pixels][COT*16 channels]
//   B = gathered x   [32 pixels][ntaps*CIN (padded to 16)]
// in LDS, pixel-major, exactly as they lie in the NHWC tensors.  The MFMA wants its 8 (bf16) /
// 4 (fp32) K values per lane to be consecutive PIXELS of one channel, i.e. the transpose of that
// image: bf16 fragments come from ds_read_b64_tr_b16 (hardware 4x16 transpose), fp32 fragments
// from four strided ds_read_b32.  Accumulators stay in registers over the whole pixel range; each
// workgroup then writes one fp32 partial tile and a second kernel sums the partials in a fixed
// order and scatters them into the torch weight layout (no atomics: bit-reproducible).
#include "common.cuh"
#include "train.cuh"
#include "conv1d_tile.cuh"
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct WgradParams {
    const void* x;
    const void* dz;
    float* partial;
    int N, Hin, Win, ldx, xoff, Hb, Wb, Hout, Wout, lddz, dzoff, Si, So;
    int oh, ow, ntaps;
    int dh[GC_MAX_TAPS], dw[GC_MAX_TAPS];
    long long P;
    int steps_per_wg;
    int ncols_pad;
};

template <typename T>
__device__ __forceinline__ i32x4 load_frag(const char* lds, int row_bytes, int col0, int kbase, int lane);

// bf16: pixels kbase + 8q .. +7 of column col0 + (lane & 15)
template <>
__device__ __forceinline__ i32x4 load_frag<bf16raw>(const char* lds, int row_bytes, int col0, int kbase, int lane) {
    const int j = lane & 15, q = lane >> 4;
    const char* p = lds + (kbase + 8 * q + (j >> 2)) * row_bytes + (col0 + 4 * (j & 3)) * 2;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * row_bytes));
    i32x2 l2 = __builtin_bit_cast(i32x2, lo), h2 = __builtin_bit_cast(i32x2, hi);
    return i32x4{l2[0], l2[1], h2[0], h2[1]};
}

// fp32: pixels kbase + 4q .. +3 of column col0 + (lane & 15)
template <>
__device__ __forceinline__ i32x4 load_frag<float>(const char* lds, int row_bytes, int col0, int kbase, int lane) {
    const int j = lane & 15, q = lane >> 4;
    const char* p = lds + (kbase + 4 * q) * row_bytes + (col0 + j) * 4;
    i32x4 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = *(const int*)(p + e * row_bytes);
    return r;
}

template <typename T, int CIN, int COT, int CPW>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = CIN / EPC;                    // chunks per gathered pixel
    constexpr int ACH = COT * 16 / EPC;              // chunks per dz pixel
    constexpr int KPIX = 32;
    constexpr int SUB = Elem<T>::KSTEP;              // pixels one mma_step consumes (32 bf16, 16 fp32)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_row = COT * 16 * (int)sizeof(T);
    const int b_row = p.ncols_pad * (int)sizeof(T);
    char* a_lds = smem;
    char* b_lds = smem + KPIX * a_row;
    __shared__ int s_n[KPIX], s_i[KPIX], s_j[KPIX];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntc = p.ncols_pad >> 4;

    // zero the padding columns of B once
    const int ncols = p.ntaps * CIN;
    for (int id = tid; id < KPIX * (p.ncols_pad - ncols); id += 256) {
        const int px = id / (p.ncols_pad - ncols), c = ncols + id % (p.ncols_pad - ncols);
        *(T*)(b_lds + px * b_row + c * (int)sizeof(T)) = Elem<T>::from_f32(0.f);
    }

    f32x4 acc[COT][CPW];
#pragma unroll
    for (int r = 0; r < COT; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* xg = (const T*)p.x + p.xoff;
    const T* dzg = (const T*)p.dz + p.dzoff;
    const long long step0 = (long long)blockIdx.x * p.steps_per_wg;

    for (int s = 0; s < p.steps_per_wg; ++s) {
        const long long pix0 = (step0 + s) * KPIX;
        if (pix0 >= p.P) break;
        __syncthreads();                              // previous step's fragments are consumed
        if (tid < KPIX) {
            const long long pix = pix0 + tid;
            if (pix < p.P) {
                s_j[tid] = (int)(pix % p.Wb);
                const long long rest = pix / p.Wb;
                s_i[tid] = (int)(rest % p.Hb);
                s_n[tid] = (int)(rest / p.Hb);
            } else {
                s_n[tid] = -1; s_i[tid] = 0; s_j[tid] = 0;
            }
        }
        __syncthreads();
        // Both images are gathered in batches of U chunks per thread: all loads of a batch are issued before the first
        // LDS store waits for one, so a step pays a few memory round trips instead of one per chunk (the trip counts are
        // run-time values, which keeps the compiler from doing this itself).
        constexpr int U = 4;
        // A: dz rows of the 32 output pixels
        for (int base = 0; base < KPIX * ACH; base += 256 * U) {
            i32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = base + u * 256 + tid;
                v[u] = i32x4{0, 0, 0, 0};
                if (id < KPIX * ACH) {
                    const int px = id / ACH, ch = id % ACH;
                    const int n = s_n[px];
                    if (n >= 0) {
                        const long long op = ((long long)n * p.Hout + s_i[px] * p.So + p.oh) * p.Wout + s_j[px] * p.So + p.ow;
                        v[u] = *(const i32x4*)(dzg + op * p.lddz + ch * EPC);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = base + u * 256 + tid;
                if (id < KPIX * ACH) *(i32x4*)(a_lds + (id / ACH) * a_row + (id % ACH) * 16) = v[u];
            }
        }
        // B: gathered input pixels, one CIN-wide column block per tap
        const int btotal = p.ntaps * KPIX * CH;
        for (int base = 0; base < btotal; base += 256 * U) {
            i32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = base + u * 256 + tid;
                v[u] = i32x4{0, 0, 0, 0};
                if (id < btotal) {
                    const int tap = id / (KPIX * CH), rem = id % (KPIX * CH);
                    const int px = rem / CH, ch = rem % CH;
                    const int n = s_n[px];
                    if (n >= 0) {
                        const int hi = s_i[px] * p.Si + p.dh[tap], wi = s_j[px] * p.Si + p.dw[tap];
                        if (hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win)
                            v[u] = *(const i32x4*)(xg + (((long long)n * p.Hin + hi) * p.Win + wi) * p.ldx + ch * EPC);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int id = base + u * 256 + tid;
                if (id < btotal) {
                    const int tap = id / (KPIX * CH), rem = id % (KPIX * CH);
                    *(i32x4*)(b_lds + (rem / CH) * b_row + (tap * CIN) * (int)sizeof(T) + (rem % CH) * 16) = v[u];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < KPIX; kb += SUB) {
            i32x4 af[COT];
#pragma unroll
            for (int r = 0; r < COT; ++r) af[r] = load_frag<T>(a_lds, a_row, r * 16, kb, lane);
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int ct = wave + c * 4;
                if (ct < ntc) {
                    const i32x4 bf = load_frag<T>(b_lds, b_row, ct * 16, kb, lane);
#pragma unroll
                    for (int r = 0; r < COT; ++r) acc[r][c] = mma_step<T>(af[r], bf, acc[r][c]);
                }
            }
        }
    }

    // partial tile [COT*16][ncols_pad] of this workgroup
    float* out = p.partial + (long long)blockIdx.x * (COT * 16) * p.ncols_pad;
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ct = wave + c * 4;
        if (ct >= ntc) continue;
#pragma unroll
        for (int r = 0; r < COT; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(long long)(r * 16 + q * 4 + e) * p.ncols_pad + ct * 16 + j] = acc[r][c][e];
    }
}

// ---- One tap per workgroup ("tap-split") ----------------------------------------------------------------------------
// The kernel above is bound by the latency of a step (index arithmetic, gather, LDS, fragments: ~4 us for ~0.2 us of
// matrix work) and cannot hide it with more workgroups, because every workgroup owns a whole [rows][taps * CIN] tile of
// partial sums: twice the workgroups is twice the partial-sum traffic.  Here a workgroup owns ONE tap's [rows][CIN]
// column block of that tile: grid = (pixel slices, taps of all phases).  The partial sums of a slice are the same
// bytes as before, but 3..9 times as many workgroups are in flight behind them (and the four phases of a transposed
// convolution are one launch).  A step takes 128 (bf16) / 64 (fp32) base pixels, every thread decodes its own pixels
// with multiply-high divisions and has all its loads in flight before the first LDS store.  CIN = 16 has a single column
// tile: the four waves then split the pixel sub-steps and meet in LDS (fixed order) before the partial tile is written.
struct WTapEntry { float* partial; int oh, ow, dh, dw, col0, ncols_pad; };
struct WTapParams {
    const void* x;
    const void* dz;
    int N, Hin, Win, ldx, xoff, Hb, Wb, Hout, Wout, lddz, dzoff, Si, So;
    unsigned P;                     // base pixels
    int ppw;                        // base pixels per workgroup (a multiple of 128)
    int rows_pad;
    FastDiv fW, fH;
    WTapEntry e[GC_MAX_TAPS];
};

template <typename T, int CIN, int COT>
struct WTapCfg {
    static constexpr int EPC = Elem<T>::EPC;
    static constexpr int KPIX = 16 * EPC;                         // 128 pixels (bf16) / 64 (fp32) per step
    static constexpr int A_ROW = COT * 16 * (int)sizeof(T) + 16;  // LDS row pitch: + 16 B spreads the transposed reads over the banks
    static constexpr int B_ROW = CIN * (int)sizeof(T) + 16;
    static constexpr int LDS = KPIX * (A_ROW + B_ROW);
};

template <typename T, int CIN, int COT>
__global__ __launch_bounds__(256) void wgrad_tap_kernel(const WTapParams p) {
    using Cfg = WTapCfg<T, CIN, COT>;
    constexpr int EPC = Cfg::EPC, KPIX = Cfg::KPIX, A_ROW = Cfg::A_ROW, B_ROW = Cfg::B_ROW;
    constexpr int SUB = Elem<T>::KSTEP;                 // pixels one mma_step consumes
    constexpr int CH = CIN / EPC, ACH = COT * 16 / EPC;
    constexpr int NCT = CIN / 16;                       // column tiles of this tap
    constexpr int WK = NCT >= 4 ? 1 : 4 / NCT;          // wave groups that split the pixel sub-steps
    constexpr int CPW = NCT >= 4 ? NCT / 4 : 1;         // column tiles per wave
    constexpr int NA = KPIX * ACH / 256, NB = KPIX * CH / 256;
    static_assert(CIN % 16 == 0 && (NCT >= 4 ? NCT % 4 == 0 : 4 % NCT == 0), "column tiles must divide over four waves");
    static_assert(NA * 256 == KPIX * ACH && NB * 256 == KPIX * CH, "whole chunks per thread");
    static_assert(WK == 1 || 4 * COT * 1024 <= Cfg::LDS, "cross-wave sum must fit the staging buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const a_lds = smem;
    char* const b_lds = smem + KPIX * A_ROW;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const WTapEntry& e = p.e[blockIdx.y];
    const T* const xg = (const T*)p.x + p.xoff;
    const T* const dzg = (const T*)p.dz + p.dzoff;
    const unsigned pixbeg = blockIdx.x * (unsigned)p.ppw;
    const unsigned pixend = pixbeg + (unsigned)p.ppw < p.P ? pixbeg + (unsigned)p.ppw : p.P;

    f32x4 acc[COT][CPW];
#pragma unroll
    for (int r = 0; r < COT; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (unsigned pix0 = pixbeg; pix0 < pixend; pix0 += KPIX) {
        i32x4 va[NA], vb[NB];
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int id = u * 256 + tid;
            const unsigned pix = pix0 + id / ACH;
            va[u] = i32x4{0, 0, 0, 0};
            if (pix < pixend) {
                const unsigned rest = fast_div(pix, p.fW);
                const int j = (int)(pix - rest * p.fW.div);
                const unsigned n = fast_div(rest, p.fH);
                const int i = (int)(rest - n * p.fH.div);
                const long long op = ((long long)n * p.Hout + i * p.So + e.oh) * p.Wout + j * p.So + e.ow;
                va[u] = *(const i32x4*)(dzg + op * p.lddz + (id % ACH) * EPC);
            }
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int id = u * 256 + tid;
            const unsigned pix = pix0 + id / CH;
            vb[u] = i32x4{0, 0, 0, 0};
            if (pix < pixend) {
                const unsigned rest = fast_div(pix, p.fW);
                const int j = (int)(pix - rest * p.fW.div);
                const unsigned n = fast_div(rest, p.fH);
                const int i = (int)(rest - n * p.fH.div);
                const int hi = i * p.Si + e.dh, wi = j * p.Si + e.dw;
                if (hi >= 0 && hi < p.Hin && wi >= 0 && wi < p.Win)
                    vb[u] = *(const i32x4*)(xg + (((long long)n * p.Hin + hi) * p.Win + wi) * p.ldx + (id % CH) * EPC);
            }
        }
        __syncthreads();                              // the previous step's fragments are consumed
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int id = u * 256 + tid;
            *(i32x4*)(a_lds + (id / ACH) * A_ROW + (id % ACH) * 16) = va[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int id = u * 256 + tid;
            *(i32x4*)(b_lds + (id / CH) * B_ROW + (id % CH) * 16) = vb[u];
        }
        __syncthreads();
        const int wk = WK == 1 ? 0 : wave % WK;
#pragma unroll
        for (int kk = 0; kk < KPIX / (SUB * WK); ++kk) {
            const int kb = (kk * WK + wk) * SUB;
            i32x4 af[COT];
#pragma unroll
            for (int r = 0; r < COT; ++r) af[r] = load_frag<T>(a_lds, A_ROW, r * 16, kb, lane);
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const int ct = WK == 1 ? wave + 4 * c : wave / WK;
                const i32x4 bf = load_frag<T>(b_lds, B_ROW, ct * 16, kb, lane);
#pragma unroll
                for (int r = 0; r < COT; ++r) acc[r][c] = mma_step<T>(af[r], bf, acc[r][c]);
            }
        }
    }

    if (WK > 1) {                                     // the wave groups of a column tile meet in LDS, summed in group order
        __syncthreads();
        f32x4* const red = (f32x4*)smem;
#pragma unroll
        for (int r = 0; r < COT; ++r) red[(wave * COT + r) * 64 + lane] = acc[r][0];
        __syncthreads();
        if (wave % WK != 0) return;
#pragma unroll
        for (int r = 0; r < COT; ++r) {
            f32x4 s = red[(wave * COT + r) * 64 + lane];
#pragma unroll
            for (int g = 1; g < WK; ++g) {
                const f32x4 o = red[((wave + g) * COT + r) * 64 + lane];
#pragma unroll
                for (int k = 0; k < 4; ++k) s[k] += o[k];
            }
            acc[r][0] = s;
        }
    }
    // this tap's column block of the slice's partial tile [rows_pad][ncols_pad]
    float* const out = e.partial + (long long)blockIdx.x * p.rows_pad * e.ncols_pad + e.col0;
    const int q = lane >> 4, j = lane & 15;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ct = WK == 1 ? wave + 4 * c : wave / WK;
#pragma unroll
        for (int r = 0; r < COT; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) out[(long long)(r * 16 + q * 4 + k) * e.ncols_pad + ct * 16 + j] = acc[r][c][k];
    }
}

// dW[map(row, tap, ci)] = sum_wg partial[wg][row][tap*CIN + ci]
// Block = E consecutive elements x 256 / E slices of the workgroup range; every slice sums its partials in a
// fixed order (four chains, so the loads overlap), the slices meet in LDS and are added in slice order.
// E = 32 for the usual few hundred partials, E = 8 (32 slices) when small tiles were spread over many workgroups.
template <int E>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nwg, int rows_pad,
                                                           int ncols_pad, int cin, const PackParams m, float* __restrict__ dw) {
    constexpr int S = 256 / E;
    __shared__ double sh[S][E];
    const int total = m.cout_virtual * m.ntaps * cin;
    const int e = threadIdx.x % E, sl = threadIdx.x / E;
    const int idx = blockIdx.x * E + e;
    bool valid = idx < total;
    int col = 0, row = 0, tap = 0, ci = 0, phase = 0, cls = 0;
    if (valid) {
        col = idx % (m.ntaps * cin); row = idx / (m.ntaps * cin);
        tap = col / cin; ci = col % cin;
        phase = row / m.ncp; cls = row % m.ncp;
        valid = cls < m.nc_valid && !(m.cin_valid && ci >= m.cin_valid);
    }
    double acc = 0.0;
    if (valid) {
        const long long stride = (long long)rows_pad * ncols_pad;
        const float* src = partial + (long long)row * ncols_pad + col;
        const int per = (nwg + S - 1) / S;
        int g = sl * per;
        const int g1 = g + per < nwg ? g + per : nwg;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (; g + 4 <= g1; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += src[(g + u) * stride];
        }
        for (; g < g1; ++g) s4[0] += src[g * stride];
        acc = ((double)s4[0] + (double)s4[1]) + ((double)s4[2] + (double)s4[3]);
    }
    sh[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && valid) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) s += sh[k][e];
        dw[(long long)cls * m.s_co + (long long)phase * m.s_phase + (long long)ci * m.s_ci + m.toff[tap]] = (float)s;
    }
}

// The same reduction for a table of descriptors that travels as the kernel's argument (no upload, nothing for a captured
// graph to read from host memory later): blockIdx.y = entry, blockIdx.x = block of 32 elements.
#define ERF_WGRAD_REDUCE_CHUNK 26                                    /* 26 * 152 B stays below the 4 KB of kernel arguments */
struct WgradReduceTable { WgradReduceDesc d[ERF_WGRAD_REDUCE_CHUNK]; };
static_assert(sizeof(WgradReduceTable) <= 4096, "descriptor table must fit the kernel-argument segment");

__global__ __launch_bounds__(256) void wgrad_reduce_many_kernel(const WgradReduceTable table) {
    constexpr int E = 32, S = 8;
    __shared__ double sh[S][E];
    const WgradReduceDesc& d = table.d[blockIdx.y];
    const PackParams& m = d.m;
    const int total = m.cout_virtual * m.ntaps * d.cin;
    if ((int)blockIdx.x * E >= total) return;
    const int e = threadIdx.x % E, sl = threadIdx.x / E;
    const int idx = blockIdx.x * E + e;
    bool valid = idx < total;
    int col = 0, row = 0, tap = 0, ci = 0, phase = 0, cls = 0;
    if (valid) {
        col = idx % (m.ntaps * d.cin); row = idx / (m.ntaps * d.cin);
        tap = col / d.cin; ci = col % d.cin;
        phase = row / m.ncp; cls = row % m.ncp;
        valid = cls < m.nc_valid && !(m.cin_valid && ci >= m.cin_valid);
    }
    double acc = 0.0;
    if (valid) {
        const long long stride = (long long)d.rows_pad * d.ncols_pad;
        const float* src = d.partial + (long long)row * d.ncols_pad + col;
        const int per = (d.nwg + S - 1) / S;
        int g = sl * per;
        const int g1 = g + per < d.nwg ? g + per : d.nwg;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        for (; g + 4 <= g1; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s4[u] += src[(g + u) * stride];
        }
        for (; g < g1; ++g) s4[0] += src[g * stride];
        acc = ((double)s4[0] + (double)s4[1]) + ((double)s4[2] + (double)s4[3]);
    }
    sh[sl][e] = acc;
    __syncthreads();
    if (sl == 0 && valid) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < S; ++k) s += sh[k][e];
        d.dw[(long long)cls * m.s_co + (long long)phase * m.s_phase + (long long)ci * m.s_ci + m.toff[tap]] = (float)s;
    }
}

int wgrad_reduce_many_launch(const void* table_host, int count, hipStream_t stream) {
    if (!table_host || count <= 0) return ERF_EINVAL;
    const WgradReduceDesc* src = (const WgradReduceDesc*)table_host;
    for (int base = 0; base < count; base += ERF_WGRAD_REDUCE_CHUNK) {
        const int n = count - base < ERF_WGRAD_REDUCE_CHUNK ? count - base : ERF_WGRAD_REDUCE_CHUNK;
        WgradReduceTable t;
        int max_total = 0;
        for (int i = 0; i < n; ++i) {
            t.d[i] = src[base + i];
            if (!t.d[i].partial || !t.d[i].dw) return ERF_EINVAL;
            const int total = t.d[i].m.cout_virtual * t.d[i].m.ntaps * t.d[i].cin;
            max_total = total > max_total ? total : max_total;
        }
        for (int i = n; i < ERF_WGRAD_REDUCE_CHUNK; ++i) t.d[i] = t.d[0];
        if (max_total <= 0) return ERF_EINVAL;
        hipLaunchKernelGGL(wgrad_reduce_many_kernel, dim3((max_total + 31) / 32, n), dim3(256), 0, stream, t);
        const int rc = erf_check_launch();
        if (rc) return rc;
    }
    return ERF_OK;
}

// While a descriptor slot is set for the calling thread, the next reduction is not launched: its arguments are written
// to the slot (host memory) for a later wgrad_reduce_many_launch.
static thread_local WgradReduceDesc* t_reduce_slot = nullptr;
void wgrad_reduce_defer(WgradReduceDesc* slot) { t_reduce_slot = slot; }

// A recorder for entry points that issue several reductions (one per weight phase): while it is open on the calling
// thread, reductions are appended to its table instead of being launched (the table full: launched as usual).
static thread_local WgradReduceDesc* t_rec = nullptr;
static thread_local int t_rec_cap = 0, t_rec_n = 0;
void wgrad_reduce_record_begin(WgradReduceDesc* table, int capacity) { t_rec = table; t_rec_cap = capacity; t_rec_n = 0; }
int wgrad_reduce_record_end() { const int n = t_rec_n; t_rec = nullptr; t_rec_cap = t_rec_n = 0; return n; }
bool wgrad_reduce_recording() { return t_rec != nullptr; }

int wgrad_reduce_launch(const float* partial, int nwg, int rows_pad, int ncols_pad, int cin, const PackParams& m, float* dw,
                        hipStream_t stream) {
    if (t_reduce_slot) {
        WgradReduceDesc& d = *t_reduce_slot;
        d.partial = partial; d.dw = dw; d.nwg = nwg; d.rows_pad = rows_pad; d.ncols_pad = ncols_pad; d.cin = cin; d.m = m;
        t_reduce_slot = nullptr;
        return ERF_OK;
    }
    if (t_rec && t_rec_n < t_rec_cap) {
        WgradReduceDesc& d = t_rec[t_rec_n++];
        d.partial = partial; d.dw = dw; d.nwg = nwg; d.rows_pad = rows_pad; d.ncols_pad = ncols_pad; d.cin = cin; d.m = m;
        return ERF_OK;
    }
    const int total = m.cout_virtual * m.ntaps * cin;
    if (nwg >= 1024)
        hipLaunchKernelGGL(wgrad_reduce_kernel<8>, dim3((total + 7) / 8), dim3(256), 0, stream, partial, nwg, rows_pad, ncols_pad, cin, m, dw);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel<32>, dim3((total + 31) / 32), dim3(256), 0, stream, partial, nwg, rows_pad, ncols_pad, cin, m, dw);
    return erf_check_launch();
}

// Workgroups over the pixel range.  Each step is a short, latency-bound load -> LDS -> MFMA round, so small output tiles
// (few KB of partial sums per workgroup) take up to 2048 workgroups for occupancy; large tiles stay at 512 so that the
// partial sums the reduction reads back stay within ~24 MB.
static void plan(long long P, size_t tile_bytes, int* nwg, int* spw) {
    static const int forced = erf_debug_env("ERF_WGRAD_WGS", 0);
    static const int smin = erf_debug_env("ERF_WGRAD_SMIN", 8);
    long long target = forced;
    if (target <= 0) {
        target = (long long)((24u << 20) / (tile_bytes ? tile_bytes : 1));
        target = target < 512 ? 512 : (target > 2048 ? 2048 : target);
    }
    const long long steps = (P + 31) / 32;
    long long s = (steps + target - 1) / target;
    if (s < smin) s = smin;
    *spw = (int)s;
    *nwg = (int)((steps + s - 1) / s);
}

// ---- tap-split path: which shapes it takes and how it cuts the pixel range
static bool tap_instance(int cin, int cot) {
    switch (cin) {
        case 16: return cot == 1 || cot == 2 || cot == 3 || cot == 5 || cot == 8;
        case 64: return cot == 1 || cot == 4;
        case 128: return cot == 1 || cot == 2 || cot == 4 || cot == 8;
    }
    return false;
}
static bool tap_path(int rows, int cin, long long P) {
    static const int off = erf_debug_env("ERF_WGRAD_TAP_OFF", 0);
    return !off && P > 0 && P < (1LL << 31) - 65536 && tap_instance(cin, (rows + 15) / 16);
}
// About eight resident workgroups per CU over the taps of the largest phase, at least two steps each, and at most
// 12 MB of partial sums per phase.
static void tap_plan(long long P, size_t tile_bytes, int max_taps, int* nwg, int* ppw) {
    long long nk = 2048 / (max_taps > 0 ? max_taps : 1);
    const long long cap = (long long)((12u << 20) / (tile_bytes ? tile_bytes : 1));
    if (nk > cap) nk = cap;
    if (nk < 1) nk = 1;
    long long per = (P + nk - 1) / nk;
    per = (per + 127) / 128 * 128;
    if (per < 256) per = 256;
    *ppw = (int)per;
    *nwg = (int)((P + per - 1) / per);
}

size_t wgrad_workspace_bytes(int rows, int cin, int max_taps_per_phase, long long base_pixels) {
    int nwg, spw;
    const int rows_pad = (rows + 15) / 16 * 16;
    const int ncols_pad = (max_taps_per_phase * cin + 15) / 16 * 16;
    const size_t tile = (size_t)rows_pad * ncols_pad * sizeof(float);
    plan(base_pixels, tile, &nwg, &spw);
    size_t need = (size_t)nwg * tile;
    if (tap_path(rows, cin, base_pixels)) {          // a caller's workspace serves either path
        int nk, ppw;
        tap_plan(base_pixels, tile, max_taps_per_phase, &nk, &ppw);
        if ((size_t)nk * tile > need) need = (size_t)nk * tile;
    }
    return need;
}

template <typename T, int CIN, int COT, int CPW>
static int launch_phase(const WgradParams& wp, int nwg, hipStream_t stream) {
    if constexpr (CIN % Elem<T>::EPC != 0) {
        return ERF_EUNSUPPORTED;
    } else {
    const size_t lds = (size_t)32 * (COT * 16 + wp.ncols_pad) * sizeof(T);
    auto kern = wgrad_kernel<T, CIN, COT, CPW>;
    if (lds > 64 * 1024) {
        if (lds > 150 * 1024) return ERF_EUNSUPPORTED;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return ERF_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(256), lds, stream, wp);
    return erf_check_launch();
    }
}

#define WG_CASE(CINV, COTV, CPWV)                                                                         \
    if (cin == CINV && cot == COTV && cpw <= CPWV)                                                        \
        return dtype == 1 ? launch_phase<bf16raw, CINV, COTV, CPWV>(wp, nwg, stream)                      \
                          : launch_phase<float, CINV, COTV, CPWV>(wp, nwg, stream);

static int dispatch(const WgradParams& wp, int dtype, int cin, int cot, int cpw, int nwg, hipStream_t stream) {
    WG_CASE(16, 1, 1)       // conv1d C=16 (3 col tiles)
    WG_CASE(64, 4, 3)       // conv1d C=64 (12 col tiles)
    WG_CASE(128, 8, 6)      // conv1d C=128 (24 col tiles)
    WG_CASE(16, 3, 3)       // downsampler 16->64 (9 col tiles)
    WG_CASE(64, 4, 9)       // downsampler 64->128 (36 col tiles)
    WG_CASE(128, 4, 8)      // upsampler 128->64 (<= 32 col tiles per phase)
    WG_CASE(64, 1, 4)       // upsampler 64->16 (<= 16 col tiles per phase)
    WG_CASE(16, 2, 1)       // output_conv, nc <= 8
    WG_CASE(16, 5, 1)       // output_conv, nc <= 20
    WG_CASE(16, 8, 1)       // output_conv, nc <= 32
    WG_CASE(8, 1, 2)        // stem, bf16 image padded to 8 channels (5 col tiles)
    WG_CASE(4, 1, 1)        // stem, fp32 image padded to 4 channels (3 col tiles)
    WG_CASE(128, 1, 2)      // encoder output_conv 1x1, nc <= 16
    WG_CASE(128, 2, 2)      // encoder output_conv 1x1, nc <= 32
    return ERF_EUNSUPPORTED;
}

template <typename T, int CIN, int COT>
static int launch_tap(const WTapParams& tp, int nwg, int ntaps, hipStream_t stream) {
    using Cfg = WTapCfg<T, CIN, COT>;
    auto kern = wgrad_tap_kernel<T, CIN, COT>;
    if (Cfg::LDS > 64 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(nwg, ntaps), dim3(256), Cfg::LDS, stream, tp);
    return erf_check_launch();
}

#define WT_CASE(CINV, COTV)                                                                               \
    if (cin == CINV && cot == COTV)                                                                       \
        return dtype == 1 ? launch_tap<bf16raw, CINV, COTV>(tp, nwg, ntaps, stream)                       \
                          : launch_tap<float, CINV, COTV>(tp, nwg, ntaps, stream);

static int dispatch_tap(const WTapParams& tp, int dtype, int cin, int cot, int nwg, int ntaps, hipStream_t stream) {
    WT_CASE(16, 1) WT_CASE(16, 2) WT_CASE(16, 3) WT_CASE(16, 5) WT_CASE(16, 8)
    WT_CASE(64, 1) WT_CASE(64, 4)
    WT_CASE(128, 1) WT_CASE(128, 2) WT_CASE(128, 4) WT_CASE(128, 8)
    return ERF_EUNSUPPORTED;
}

// All phases and taps in one launch, one workspace slice per phase; the per-phase reductions follow (or are recorded).
static int wgrad_tap_launch(const GConvParams& geo, const void* dz, int lddz, int dzoff, int cot, int cin, int max_taps,
                            const PackParams* maps, float* dw, float* workspace, size_t workspace_bytes, int dtype,
                            hipStream_t stream) {
    const long long P = (long long)geo.N * geo.Hb * geo.Wb;
    const int rows_pad = cot * 16;
    int nwg, ppw;
    tap_plan(P, (size_t)rows_pad * max_taps * cin * sizeof(float), max_taps, &nwg, &ppw);
    const size_t per_phase = (size_t)nwg * rows_pad * max_taps * cin * sizeof(float);
    if (per_phase * geo.nphase > workspace_bytes) return ERF_EUNSUPPORTED;
    if (dzoff + rows_pad > lddz) return ERF_EUNSUPPORTED;      // a pixel's row chunks are read whole
    WTapParams tp{};
    tp.x = geo.x; tp.dz = dz;
    tp.N = geo.N; tp.Hin = geo.Hin; tp.Win = geo.Win; tp.ldx = geo.ldx; tp.xoff = geo.xoff; tp.Hb = geo.Hb; tp.Wb = geo.Wb;
    tp.Hout = geo.Hout; tp.Wout = geo.Wout; tp.lddz = lddz; tp.dzoff = dzoff; tp.Si = geo.Si; tp.So = geo.So;
    tp.P = (unsigned)P; tp.ppw = ppw; tp.rows_pad = rows_pad;
    tp.fW = make_fastdiv((uint32_t)geo.Wb); tp.fH = make_fastdiv((uint32_t)geo.Hb);
    int ntaps = 0;
    for (int ph = 0; ph < geo.nphase; ++ph) {
        const GConvPhase& g = geo.ph[ph];
        for (int t = 0; t < g.ntaps; ++t) {
            if (ntaps >= GC_MAX_TAPS) return ERF_EUNSUPPORTED;
            WTapEntry& e = tp.e[ntaps++];
            e.partial = workspace + (per_phase / sizeof(float)) * ph;
            e.oh = g.oh; e.ow = g.ow; e.dh = g.dh[t]; e.dw = g.dw[t]; e.col0 = t * cin; e.ncols_pad = g.ntaps * cin;
        }
    }
    int rc = dispatch_tap(tp, dtype, cin, cot, nwg, ntaps, stream);
    if (rc) return rc;
    for (int ph = 0; ph < geo.nphase; ++ph) {
        rc = wgrad_reduce_launch(workspace + (per_phase / sizeof(float)) * ph, nwg, rows_pad, geo.ph[ph].ntaps * cin, cin, maps[ph],
                                 dw, stream);
        if (rc) return rc;
    }
    return ERF_OK;
}

int wgrad_launch(const GConvParams& geo, const void* dz, int lddz, int dzoff, int rows, int cin, const PackParams* maps,
                 float* dw, float* workspace, size_t workspace_bytes, int dtype, hipStream_t stream) {
    const long long P = (long long)geo.N * geo.Hb * geo.Wb;
    const int cot = (rows + 15) / 16;
    int max_taps = 1;
    for (int ph = 0; ph < geo.nphase; ++ph) max_taps = geo.ph[ph].ntaps > max_taps ? geo.ph[ph].ntaps : max_taps;
    const int epc = dtype == 1 ? 8 : 4;
    if (cin % epc) return ERF_EUNSUPPORTED;
    if (tap_path(rows, cin, P)) {
        const int rc = wgrad_tap_launch(geo, dz, lddz, dzoff, cot, cin, max_taps, maps, dw, workspace, workspace_bytes, dtype, stream);
        if (rc != ERF_EUNSUPPORTED) return rc;
    }
    int nwg, spw;                       // same plan as wgrad_workspace_bytes (the caller sized the workspace with it)
    plan(P, (size_t)cot * 16 * ((max_taps * cin + 15) / 16 * 16) * sizeof(float), &nwg, &spw);
    // phases get a workspace slice each when the caller left room for that (their reductions can then be put off);
    // with one shared workspace a phase's partial tiles must be reduced before the next phase overwrites them
    const size_t per_phase = (size_t)nwg * cot * 16 * ((max_taps * cin + 15) / 16 * 16) * sizeof(float);
    const bool sliced = geo.nphase > 1 && per_phase * geo.nphase <= workspace_bytes;
    WgradReduceDesc* const rec = t_rec;
    if (geo.nphase > 1 && !sliced) t_rec = nullptr;
    struct Restore { WgradReduceDesc* r; ~Restore() { t_rec = r; } } restore{rec};
    for (int ph = 0; ph < geo.nphase; ++ph) {
        const GConvPhase& g = geo.ph[ph];
        float* const ws_ph = sliced ? workspace + (per_phase / sizeof(float)) * ph : workspace;
        WgradParams wp{};
        wp.x = geo.x; wp.dz = dz; wp.partial = ws_ph;
        wp.N = geo.N; wp.Hin = geo.Hin; wp.Win = geo.Win; wp.ldx = geo.ldx; wp.xoff = geo.xoff; wp.Hb = geo.Hb; wp.Wb = geo.Wb;
        wp.Hout = geo.Hout; wp.Wout = geo.Wout; wp.lddz = lddz; wp.dzoff = dzoff; wp.Si = geo.Si; wp.So = geo.So;
        wp.oh = g.oh; wp.ow = g.ow; wp.ntaps = g.ntaps;
        for (int t = 0; t < g.ntaps; ++t) { wp.dh[t] = g.dh[t]; wp.dw[t] = g.dw[t]; }
        wp.P = P; wp.steps_per_wg = spw;
        wp.ncols_pad = (g.ntaps * cin + 15) / 16 * 16;
        const size_t need = (size_t)nwg * cot * 16 * wp.ncols_pad * sizeof(float);
        if (need > workspace_bytes) return ERF_EINVAL;
        const int cpw = (wp.ncols_pad / 16 + 3) / 4;
        int rc = dispatch(wp, dtype, cin, cot, cpw, nwg, stream);
        if (rc) return rc;
        const PackParams& m = maps[ph];
        rc = wgrad_reduce_launch(ws_ph, nwg, cot * 16, wp.ncols_pad, cin, m, dw, stream);
        if (rc) return rc;
    }
    return ERF_OK;
}
