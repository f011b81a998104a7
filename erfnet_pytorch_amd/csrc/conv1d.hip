// Factorized 3x1 / 1x3 (dilated) convolution C -> C: the hot kernel of ERFNet
// (non_bottleneck_1d, 93.8 % of the network's multiply-accumulates).
//
// Shape of the problem: out[pixel][co] = sum_{tap<3} sum_ci W[co][tap][ci] * in[pixel + (tap-1)*d*e_axis][ci]
// -- a GEMM with M = pixels (10^5..10^6), N = C (16/64/128) and K = 3C, where the three taps
// are three shifted views of the same NHWC input.
//
// Design for gfx950
//  * persistent workgroups; each wave keeps its slice of the packed weights (MFMA A fragments)
//    in REGISTERS for the whole launch -- weights come from L2 once per workgroup, never from LDS;
//  * "lattice" tiles: along the convolution axis a tile takes Tm positions spaced d apart (one
//    residue class mod d is an ordinary undilated 1-D problem), along the other axis Tn
//    consecutive positions.  The halo is therefore 2 lattice positions for EVERY dilation, and
//    the input tile ((Tm+2) x Tn pixels) staged once in LDS serves all three taps: output pixel
//    m = i*Tn + j reads LDS pixel m + tap*Tn;
//  * B fragments are 16-byte ds_read_b128 reads; the chunk index inside a pixel is XOR-ed with
//    (pixel & 15) (or the equivalent for pixels narrower than a 256-byte bank row) so the 16
//    lanes of a read group hit 16 different 16-byte slots.  All tile offsets are multiples of 16
//    pixels, so the swizzled lane offset is tile- and tap-invariant;
//  * fused epilogue: per-channel scale/shift (conv bias, folded eval BatchNorm), residual add,
//    ReLU, bf16/fp32 store of 4 consecutive channels per lane;
//  * 2 workgroups per CU (single LDS buffer each): one stages while the other computes.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"
#include <stdlib.h>
struct Conv1dParams {
    const void* x;
    void* y;
    const void* w;
    const float* scale;
    const float* shift;
    const void* res;
    const void* gate;       // optional: output is zeroed where gate <= 0 (ReLU backward), before the residual add
    int d, relu;
    int A;                  // length of the convolution axis (H or W)
    int Bn;                 // length of the other axis (W, or N*H rows for axis W)
    int nimg;               // independent images along neither axis (N for axis H, 1 for axis W)
    long long img_stride;   // pixels per image (H*W) for axis H
    int stride_u;           // global pixels between lattice positions (d*W or d)
    int stride_v;           // global pixels between positions on the other axis (1 or W)
    int Tm, Tn, log2Tn;     // tile: Tm lattice positions x Tn other-axis positions (Tm*Tn = M)
    int total_chunks;       // (Tm+2) * Tn * CH
    int CA, CB;             // tiles along the lattice (per residue) / along the other axis
    int ntiles;
    int dbg;                // diagnostic bit mask (ERF_CONV1D_DBG): 1 skip loads, 2 skip MFMA, 4 skip stores, 8 skip weights
};

// Byte offset of 16-byte chunk `chunk` of LDS pixel `pix` (chunk index swizzled by the pixel).
template <int PIXB>
__device__ __forceinline__ int lds_off(int pix, int chunk) {
    constexpr int CH = PIXB / 16;
    if constexpr (CH >= 16) {
        return pix * PIXB + (((chunk & 15) ^ (pix & 15)) | (chunk & ~15)) * 16;
    } else {
        constexpr int PPR = 16 / CH;                       // pixels per 256-byte bank row
        return pix * PIXB + (chunk ^ ((pix / PPR) & (CH - 1))) * 16;
    }
}

template <typename T, int C, int WM, int WN, int M, int MAXK>
__global__ __launch_bounds__(WM * WN * 64, 2)
void conv1d_kernel(const Conv1dParams p) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = C / EPC;
    constexpr int PIXB = C * (int)sizeof(T);
    constexpr int NT = WM * WN * 64;
    constexpr int CT = C / 16;
    constexpr int CTW = CT / WN;
    constexpr int PTW = (M / 16) / WM;
    constexpr int KS = (3 * CH + 3) / 4;
    static_assert(CT % WN == 0 && (M / 16) % WM == 0, "tile split");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;

    // ---- weights of this wave's channel slice -> registers, once per launch
    i32x4 wreg[CTW][KS];
    {
        const i32x4* wp = (const i32x4*)p.w;
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[c][ks] = wp[((wn * CTW + c) * KS + ks) * 64 + lane];
    }

    const T* xg = (const T*)p.x;
    T* yg = (T*)p.y;
    const T* rg = (const T*)p.res;
    const T* gg = (const T*)p.gate;
    const int Tn = p.Tn, log2Tn = p.log2Tn;
    const int tn16 = Tn >> 4;

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        // ---- tile geometry (wave-uniform)
        int t = tile;
        const int cb = t % p.CB; t /= p.CB;
        const int ca = t % p.CA; t /= p.CA;
        const int rho = t % p.d;
        const int img = t / p.d;
        const int a_first = rho + (ca * p.Tm - 1) * p.d;          // lattice coordinate of LDS row u = 0
        const int b0 = cb * Tn;
        const long long gp0 = (long long)img * p.img_stride + (long long)a_first * (p.stride_u / p.d) + (long long)b0 * p.stride_v;
        const int ulo = a_first < 0 ? 1 : 0;
        int uhi = (p.A - a_first + p.d - 1) / p.d;                // LDS rows u with a_first + u*d < A
        if (uhi > p.Tm + 2) uhi = p.Tm + 2;
        const int vhi = p.Bn - b0 < Tn ? p.Bn - b0 : Tn;

        // ---- stage the input tile: global -> registers -> LDS (zeros outside the image)
        i32x4 stage[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int id = tid + k * NT;
            stage[k] = i32x4{0, 0, 0, 0};
            if (id < p.total_chunks) {
                const int chunk = id % CH, pix = id / CH;
                const int v = pix & (Tn - 1), u = pix >> log2Tn;
                if (u >= ulo && u < uhi && v < vhi && !(p.dbg & 1)) {
                    const long long gp = gp0 + (long long)u * p.stride_u + (long long)v * p.stride_v;
                    stage[k] = *(const i32x4*)(xg + gp * C + chunk * EPC);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            const int id = tid + k * NT;
            if (id < p.total_chunks) *(i32x4*)(smem + lds_off<PIXB>(id / CH, id % CH)) = stage[k];
        }
        __syncthreads();

        // ---- MFMA: all three taps out of the staged tile.  LDS pixel of (pixel tile t, tap) is
        // 16 * (t + tap * Tn/16) + r16; `lane_pix` is made opaque per tile so the 8 x 12 read
        // addresses are recomputed (one add each) instead of being hoisted into registers.
        int lane_pix = r16;
        asm volatile("" : "+v"(lane_pix));
        f32x4 acc[PTW][CTW];
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[pt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(p.dbg & 2))
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt) {
            const int t16 = wm * PTW + pt;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int g = ks * 4 + q;
                const int tap = (CH >= 4) ? (ks * 4) / CH : g / CH;       // wave-uniform when CH >= 4
                const int chunk = g % CH;
                const int pix = (t16 + tap * tn16) * 16 + lane_pix;
                i32x4 b = *(const i32x4*)(smem + lds_off<PIXB>(pix, chunk));
                if constexpr ((3 * CH) % 4 != 0) {
                    if (g >= 3 * CH) b = i32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[pt][c] = mma_step<T>(wreg[c][ks], b, acc[pt][c]);
            }
        }

        // ---- epilogue: lane owns channels (wn*CTW + c)*16 + 4q .. +3 of pixel m = 16*t16 + r16
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt) {
            const int m = (wm * PTW + pt) * 16 + r16;
            const int i = m >> log2Tn, j = m & (Tn - 1);
            const bool ok = (a_first + (i + 1) * p.d < p.A) && (j < vhi);
            if (!ok || (p.dbg & 4)) continue;
            const long long opix = gp0 + (long long)(i + 1) * p.stride_u + (long long)j * p.stride_v;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int co = (wn * CTW + c) * 16 + q * 4;
                f32x4 v = acc[pt][c];
                if (p.scale) {
                    const f32x4 sc = *(const f32x4*)(p.scale + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= sc[e];
                }
                if (p.shift) {
                    const f32x4 sh = *(const f32x4*)(p.shift + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += sh[e];
                }
                if (gg) {
                    const f32x4 gv = load4<T>(gg + opix * C + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = gv[e] > 0.f ? v[e] : 0.f;
                }
                if (rg) {
                    const f32x4 rv = load4<T>(rg + opix * C + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += rv[e];
                }
                if (p.relu)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                store4<T>(yg + opix * C + co, v);
            }
        }
        __syncthreads();        // every wave is done reading the tile before it is overwritten
    }
}

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <typename T, int C, int WM, int WN, int M, int MAXK>
static int launch_cfg(Conv1dParams p, int N, int H, int W, int axis, hipStream_t stream) {
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = C / EPC;
    constexpr int PIXB = C * (int)sizeof(T);
    constexpr int NT = WM * WN * 64;
    const int d = p.d;
    if (axis == 0) {
        p.A = H; p.Bn = W; p.nimg = N; p.img_stride = (long long)H * W; p.stride_u = d * W; p.stride_v = 1;
    } else {
        p.A = W; p.Bn = N * H; p.nimg = 1; p.img_stride = 0; p.stride_u = d; p.stride_v = W;
    }
    // Tm: power of two, at most the lattice points per residue class (min 2), at most M/16; the
    // row-major axis (H) keeps Tm <= 8 so the contiguous runs along W stay long.
    const int npts = (p.A + d - 1) / d;
    const int tm_max = axis == 0 ? 8 : 32;
    int Tm = 2;
    while (Tm * 2 <= npts && Tm * 2 <= tm_max && Tm * 2 <= M / 16) Tm *= 2;
    const int Tn = M / Tm;
    p.Tm = Tm; p.Tn = Tn; p.log2Tn = ilog2(Tn);
    p.total_chunks = (Tm + 2) * Tn * CH;
    if (p.total_chunks > MAXK * NT) return ERF_EUNSUPPORTED;
    p.CA = (npts + Tm - 1) / Tm;
    p.CB = (p.Bn + Tn - 1) / Tn;
    const long long nt = (long long)p.nimg * d * p.CA * p.CB;
    if (nt > 0x7FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    const size_t lds = (size_t)(Tm + 2) * Tn * PIXB;
    if (lds > 160 * 1024) return ERF_EUNSUPPORTED;
    int wg_per_cu = (int)(160 * 1024 / lds);
    const int wave_limit = 8 / (NT / 64) > 0 ? 8 / (NT / 64) : 1;       // 2 waves per SIMD
    if (wg_per_cu > wave_limit) wg_per_cu = wave_limit;
    if (wg_per_cu < 1) wg_per_cu = 1;
    p.dbg = erf_debug_env("ERF_CONV1D_DBG", 0);
    wg_per_cu = erf_debug_env("ERF_CONV1D_WGPCU", wg_per_cu);
    int grid = 256 * wg_per_cu;
    if (grid > p.ntiles) grid = p.ntiles;
    auto kern = conv1d_kernel<T, C, WM, WN, M, MAXK>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return ERF_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, stream, p);
    return erf_check_launch();
}

int conv1d_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                  const void* gate, int dtype, int N, int H, int W, int C, int axis, int dilation, int relu, hipStream_t stream) {
    Conv1dParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift; p.res = residual; p.gate = gate;
    p.d = dilation; p.relu = relu;
    if ((long long)N * H * W >= (1LL << 31)) return ERF_EUNSUPPORTED;
    // buffer-addressed DMA kernel first; the register-staged kernel below takes what it declines
    const int rc = conv1d_buf_launch(x, y, packed, scale, shift, residual, gate, nullptr, nullptr, dtype, N, H, W, C, axis, dilation,
                                     relu, stream);
    if (rc != ERF_EUNSUPPORTED) return rc;
    if (dtype == 1) {
        if (C == 128) return launch_cfg<bf16raw, 128, 1, 4, 128, 12>(p, N, H, W, axis, stream);
        if (C == 64) return launch_cfg<bf16raw, 64, 4, 1, 128, 8>(p, N, H, W, axis, stream);
        if (C == 16) return launch_cfg<bf16raw, 16, 4, 1, 512, 8>(p, N, H, W, axis, stream);
    } else {
        if (C == 128) return launch_cfg<float, 128, 1, 8, 128, 12>(p, N, H, W, axis, stream);
        if (C == 64) return launch_cfg<float, 64, 2, 2, 128, 12>(p, N, H, W, axis, stream);
        if (C == 16) return launch_cfg<float, 16, 4, 1, 512, 12>(p, N, H, W, axis, stream);
    }
    return ERF_EUNSUPPORTED;
}
