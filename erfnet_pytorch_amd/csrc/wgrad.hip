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

size_t wgrad_workspace_bytes(int rows, int cin, int max_taps_per_phase, long long base_pixels) {
    int nwg, spw;
    const int rows_pad = (rows + 15) / 16 * 16;
    const int ncols_pad = (max_taps_per_phase * cin + 15) / 16 * 16;
    plan(base_pixels, (size_t)rows_pad * ncols_pad * sizeof(float), &nwg, &spw);
    return (size_t)nwg * rows_pad * ncols_pad * sizeof(float);
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

int wgrad_launch(const GConvParams& geo, const void* dz, int lddz, int dzoff, int rows, int cin, const PackParams* maps,
                 float* dw, float* workspace, size_t workspace_bytes, int dtype, hipStream_t stream) {
    const long long P = (long long)geo.N * geo.Hb * geo.Wb;
    const int cot = (rows + 15) / 16;
    int max_taps = 1;
    for (int ph = 0; ph < geo.nphase; ++ph) max_taps = geo.ph[ph].ntaps > max_taps ? geo.ph[ph].ntaps : max_taps;
    int nwg, spw;                       // same plan as wgrad_workspace_bytes (the caller sized the workspace with it)
    plan(P, (size_t)cot * 16 * ((max_taps * cin + 15) / 16 * 16) * sizeof(float), &nwg, &spw);
    const int epc = dtype == 1 ? 8 : 4;
    if (cin % epc) return ERF_EUNSUPPORTED;
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
