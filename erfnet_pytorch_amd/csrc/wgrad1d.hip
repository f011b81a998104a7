// Weight gradient of the factorized 3x1 / 1x3 (dilated) convolution, bf16 activations:
//
//     dW[co][tap][ci] = sum_p dz[p][co] * x[p + (tap-1)*d*e_axis][ci]
//
// a GEMM whose contraction index is the PIXEL.  Same tiling as the forward kernel: a tile takes Tm
// lattice positions (spaced d apart along the convolution axis) x Tn positions of the other axis, so
// the x tile with its 2-position halo, staged once, serves all three taps (tap t = tile pixel + t*Tn).
//  * x and dz tiles go global -> LDS by DMA (buffer_load ... lds), double buffered, pixel-major with
//    the 16-byte chunk index XOR-ed by the pixel index exactly as in the forward kernel; out-of-image
//    pixels are lanes with an offset beyond the buffer: they read zeros and so add nothing;
//  * the MFMA wants 8 consecutive PIXELS of one channel per lane, the transpose of that layout:
//    both operands come from ds_read_b64_tr_b16 (hardware 4x16 transpose), conflict-free thanks to
//    the XOR (the 8 pixel rows a 32-lane group touches land in 8 different bank groups);
//  * accumulators stay in registers over ALL tiles of the (persistent) workgroup; there is no
//    per-tile epilogue.  Each workgroup writes one fp32 partial tile at the end and
//    wgrad_reduce_kernel sums the partials in a fixed order (bit-reproducible, no atomics).
//  * one launch takes up to ERF_W1D_MAX layers of the same shape (blockIdx.y = layer): the four convolutions of a
//    non_bottleneck_1d block share the chip, each with a quarter of the workgroups and four times the pixels per
//    workgroup -- a quarter of the partial tiles to write and to sum afterwards.
// Wave layout: column groups (3*C/16 column tiles split over waves) x KW waves along the pixel axis.
#include "common.cuh"
#include "train.cuh"
#include "conv1d_tile.cuh"
#include <stdlib.h>

struct W1dParams {
    const void* x;
    const void* dz;
    float* partial;
    int d;
    int A, Bn, nimg;
    long long img_stride;
    int stride_u, stride_v;
    int Tm, Tn, log2Tn;
    int CA, CB, ntiles;
    FastDiv fCB, fCA, fD;
    uint32_t bytes, pad;
};

#define ERF_W1D_MAX 4
struct W1dTable { W1dParams p[ERF_W1D_MAX]; };

struct W1dGeo {
    uint32_t soff;
    int ulo, uhi, vhi;
};

template <int PIXB>
__device__ __forceinline__ W1dGeo w1d_geo(const W1dParams& p, int tile) {
    W1dGeo g;
    uint32_t t = (uint32_t)tile;
    uint32_t nq = fast_div(t, p.fCB);
    const int cb = (int)(t - nq * p.fCB.div); t = nq;
    nq = fast_div(t, p.fCA);
    const int ca = (int)(t - nq * p.fCA.div); t = nq;
    nq = fast_div(t, p.fD);
    const int rho = (int)(t - nq * p.fD.div);
    const int img = (int)nq;
    const int a_first = rho + (ca * p.Tm - 1) * p.d;              // lattice coordinate of tile row u = 0
    const int b0 = cb * p.Tn;
    // offset of tile pixel (u = 1, v = 0); the x descriptor starts one lattice step (pad) earlier, so row u = 0 is offset 0
    g.soff = (uint32_t)(img * (int)p.img_stride + (a_first + p.d) * (p.stride_u / p.d) + b0 * p.stride_v) * PIXB;
    g.ulo = a_first < 0 ? 1 : 0;
    int uhi = (int)fast_div((uint32_t)(p.A - a_first + p.d - 1), p.fD);
    if (uhi > p.Tm + 2) uhi = p.Tm + 2;
    if (tile >= p.ntiles) uhi = 0;
    g.uhi = uhi > g.ulo ? uhi : g.ulo;
    const int vhi = p.Bn - b0;
    g.vhi = vhi < p.Tn ? vhi : p.Tn;
    return g;
}

template <int IMM>
__device__ __forceinline__ i32x2 lds_read_tr16(int addr) {
    i32x2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(IMM) : "memory");
    return r;
}
__device__ __forceinline__ void tie_reg2(i32x2& v) { asm volatile("" : "+v"(v)); }

template <int C, int CG, int KW, int CPW, int M, int KIX, int KIZ, int OCC>
__global__ __launch_bounds__(CG * KW * 64, OCC)
void wgrad1d_kernel(const W1dTable tab) {
#if defined(__HIP_DEVICE_COMPILE__)
    const W1dParams& p = tab.p[blockIdx.y];
    constexpr int CH = C / 8;
    constexpr int PIXB = C * 2;
    constexpr int NW = CG * KW;
    constexpr int CT = C / 16;                       // row tiles (co) = ci tiles per tap
    constexpr int NCOL = 3 * CT;
    constexpr int XB = KIX * NW * 1024, ZB = KIZ * NW * 1024;
    constexpr int KI = KIX + KIZ;
    constexpr int SPW = (M / 32) / KW;               // 32-pixel K-steps per wave per tile
    constexpr int NF = CT + CPW;                     // fragments per K-step
    static_assert(CG * CPW == NCOL && (M / 32) % KW == 0 && M * CH <= KIZ * NW * 64, "layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave % CG, kw = wave / CG;
    const int q = lane >> 4, j = lane & 15;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, p.bytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(p.dz, p.bytes);

    // ---- launch-invariant per-lane addressing
    uint32_t x_rel[KIX], z_rel[KIZ];
    int x_uv[KIX], z_uv[KIZ];
#pragma unroll
    for (int k = 0; k < KIX; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int u = pix >> p.log2Tn, v = pix & (p.Tn - 1);
        x_rel[k] = (uint32_t)(u * p.stride_u + v * p.stride_v) * PIXB + swz<PIXB>(pix, slot) * 16;
        x_uv[k] = ((u < p.Tm + 2 ? u : 255) << 16) | v;
    }
#pragma unroll
    for (int k = 0; k < KIZ; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int i = pix >> p.log2Tn, v = pix & (p.Tn - 1);
        z_rel[k] = (uint32_t)(i * p.stride_u + v * p.stride_v) * PIXB + swz<PIXB>(pix, slot) * 16;
        z_uv[k] = ((i < p.Tm ? i : 255) << 16) | v;
    }
    // transposed-read offsets: lane (q, j) reads 8 bytes of pixel row 8q + (j >> 2) (+4 for the second read),
    // channels 4*(j & 3) .. +3 of a 16-channel tile
    const int row1 = kw * 32 + 8 * q + (j >> 2);
    int a_off[CT][2], b_off[CPW][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = row1 + 4 * h;
#pragma unroll
        for (int r = 0; r < CT; ++r) {
            const int chunk = 2 * r + ((j & 3) >> 1);
            a_off[r][h] = 2 * XB + row * PIXB + swz<PIXB>(row, chunk) * 16 + (j & 1) * 8;
        }
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int ct = cg * CPW + c, tap = ct / CT, cit = ct % CT;
            const int chunk = 2 * cit + ((j & 3) >> 1);
            b_off[c][h] = (row + tap * p.Tn) * PIXB + swz<PIXB>(row, chunk) * 16 + (j & 1) * 8;     // Tn is a multiple of 16: same swizzle key
        }
    }
#pragma unroll
    for (int k = 0; k < KIX; ++k) asm volatile("" : "+v"(x_rel[k]), "+v"(x_uv[k]));
#pragma unroll
    for (int k = 0; k < KIZ; ++k) asm volatile("" : "+v"(z_rel[k]), "+v"(z_uv[k]));
#pragma unroll
    for (int r = 0; r < CT; ++r) asm volatile("" : "+v"(a_off[r][0]), "+v"(a_off[r][1]));
#pragma unroll
    for (int c = 0; c < CPW; ++c) asm volatile("" : "+v"(b_off[c][0]), "+v"(b_off[c][1]));

    auto dma = [&](const W1dGeo& g, int cur) {
#pragma unroll
        for (int k = 0; k < KIX; ++k) {
            const int u = x_uv[k] >> 16, v = x_uv[k] & 0xFFFF;
            const bool ok = (unsigned)(u - g.ulo) < (unsigned)(g.uhi - g.ulo) && v < g.vhi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + cur * XB + (k * NW + wave) * 1024),
                                                     16, ok ? x_rel[k] : ERF_OOB, g.soff, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < KIZ; ++k) {
            const int i = z_uv[k] >> 16, v = z_uv[k] & 0xFFFF;
            const bool ok = i < g.uhi - 1 && v < g.vhi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_z, (__attribute__((address_space(3))) void*)(smem + 2 * XB + cur * ZB + (k * NW + wave) * 1024),
                                                     16, ok ? z_rel[k] : ERF_OOB, g.soff, 0, 0);
        }
    };

    f32x4 acc[CT][CPW];
#pragma unroll
    for (int r = 0; r < CT; ++r)
#pragma unroll
        for (int c = 0; c < CPW; ++c) acc[r][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int G = gridDim.x;
    int tile = blockIdx.x;
    dma(w1d_geo<PIXB>(p, tile), 0);
    dma(w1d_geo<PIXB>(p, tile + G), 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");
    asm volatile("s_barrier" ::: "memory");

    int cur = 0;
    for (; tile < p.ntiles; tile += G, cur ^= 1) {
        int ao[CT][2], bo[CPW][2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r = 0; r < CT; ++r) ao[r][h] = a_off[r][h] + cur * ZB;
#pragma unroll
            for (int c = 0; c < CPW; ++c) bo[c][h] = b_off[c][h] + cur * XB;
        }
        // fragments of K-step s+1 are requested before the MFMAs of K-step s when the outstanding-read
        // counter (4 bits) can express "step s has arrived"; otherwise one step at a time, and the
        // second wave of the SIMD covers the wait
        constexpr bool PIPE = 2 * NF <= 15;
        i32x2 fr[PIPE ? 2 : 1][NF][2];
        auto issue = [&](auto si, i32x2 (*dst)[2]) {
            constexpr int imm = decltype(si)::value * KW * 32 * PIXB;
            static_assert(imm < 65536, "ds offset field");
            static_for<0, CT>([&](auto r) {
                dst[decltype(r)::value][0] = lds_read_tr16<imm>(ao[decltype(r)::value][0]);
                dst[decltype(r)::value][1] = lds_read_tr16<imm>(ao[decltype(r)::value][1]);
            });
            static_for<0, CPW>([&](auto c) {
                dst[CT + decltype(c)::value][0] = lds_read_tr16<imm>(bo[decltype(c)::value][0]);
                dst[CT + decltype(c)::value][1] = lds_read_tr16<imm>(bo[decltype(c)::value][1]);
            });
        };
        if constexpr (PIPE) issue(std::integral_constant<int, 0>{}, fr[0]);
        static_for<0, SPW>([&](auto si) {
            constexpr int s = decltype(si)::value;
            i32x2 (*cf)[2] = fr[PIPE ? (s & 1) : 0];
            if constexpr (!PIPE) {
                issue(si, fr[0]);
                wait_lgkm<0>();
            } else if constexpr (s + 1 < SPW) {
                issue(std::integral_constant<int, s + 1>{}, fr[(s + 1) & 1]);
                wait_lgkm<(PIPE ? 2 * NF : 0)>();
            } else {
                wait_lgkm<0>();
            }
            static_for<0, NF>([&](auto f) {
                tie_reg2(cf[decltype(f)::value][0]);
                tie_reg2(cf[decltype(f)::value][1]);
            });
#pragma unroll
            for (int c = 0; c < CPW; ++c) {
                const i32x4 bf{cf[CT + c][0][0], cf[CT + c][0][1], cf[CT + c][1][0], cf[CT + c][1][1]};
#pragma unroll
                for (int r = 0; r < CT; ++r) {
                    const i32x4 af{cf[r][0][0], cf[r][0][1], cf[r][1][0], cf[r][1][1]};
                    acc[r][c] = mma_step<bf16raw>(af, bf, acc[r][c]);
                }
            }
        });
        ERF_LDS_BARRIER();                 // every wave is done reading both tiles
        dma(w1d_geo<PIXB>(p, tile + 2 * G), cur);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");      // this wave's share of tile k+1 has landed
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // partial tile [C][3C] of this (workgroup, pixel-split wave)
    float* out = p.partial + ((long long)blockIdx.x * KW + kw) * C * (3 * C);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int ct = cg * CPW + c;
#pragma unroll
        for (int r = 0; r < CT; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) out[(r * 16 + q * 4 + e) * (3 * C) + ct * 16 + j] = acc[r][c][e];
    }
#endif
}

static inline int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// Geometry of one layer (everything but the output pointer and the grid).
template <int C, int CG, int KW, int CPW, int M, int KIX, int KIZ>
static int prep_w1d(W1dParams& p, int N, int H, int W, int axis) {
    constexpr int CH = C / 8, PIXB = C * 2, NW = CG * KW;
    const int d = p.d;
    if (axis == 0) {
        p.A = H; p.Bn = W; p.nimg = N; p.img_stride = (long long)H * W; p.stride_u = d * W; p.stride_v = 1;
    } else {
        p.A = W; p.Bn = N * H; p.nimg = 1; p.img_stride = 0; p.stride_u = d; p.stride_v = W;
    }
    const int npts = (p.A + d - 1) / d;
    const int tm_max = axis == 0 ? 8 : 32;
    int Tm = 2;
    while (Tm * 2 <= npts && Tm * 2 <= tm_max && Tm * 2 <= M / 16) Tm *= 2;
    const int Tn = M / Tm;
    p.Tm = Tm; p.Tn = Tn; p.log2Tn = ilog2i(Tn);
    if ((Tm + 2) * Tn * CH > KIX * NW * 64) return ERF_EUNSUPPORTED;
    p.CA = (npts + Tm - 1) / Tm;
    p.CB = (p.Bn + Tn - 1) / Tn;
    const long long nt = (long long)p.nimg * d * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA); p.fD = make_fastdiv(d);
    const long long bytes = (long long)N * H * W * PIXB, pad = (long long)p.stride_u * PIXB;
    if (bytes + (long long)(Tm + 3) * pad >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.bytes = (uint32_t)bytes; p.pad = (uint32_t)pad;
    return ERF_OK;
}

// `count` layers of one shape in one launch; layer i reduces into dw[i] (through wgrad_reduce_launch, which a caller may
// have put off with wgrad_reduce_defer -- then only for count == 1).  descs != nullptr: the reductions' arguments go there.
template <int C, int CG, int KW, int CPW, int M, int KIX, int KIZ>
static int launch_w1d(const W1dJob* jobs, int count, int N, int H, int W, const PackParams* maps, size_t workspace_bytes,
                      WgradReduceDesc* descs, hipStream_t stream) {
    constexpr int NW = CG * KW;
    constexpr size_t lds = 2 * (size_t)(KIX + KIZ) * NW * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    constexpr int OCC = (2 * lds <= 160 * 1024 && NW <= 4) ? 2 : 1;
    if (count < 1 || count > ERF_W1D_MAX) return ERF_EINVAL;
    W1dTable tab{};
    int ntiles_min = 0x7FFFFFFF;
    for (int i = 0; i < count; ++i) {
        W1dParams& p = tab.p[i];
        p.x = jobs[i].x; p.dz = jobs[i].dz; p.d = jobs[i].dilation;
        const int rc = prep_w1d<C, CG, KW, CPW, M, KIX, KIZ>(p, N, H, W, jobs[i].axis);
        if (rc) return rc;
        p.partial = jobs[i].workspace;
        ntiles_min = p.ntiles < ntiles_min ? p.ntiles : ntiles_min;
    }
    for (int i = count; i < ERF_W1D_MAX; ++i) tab.p[i] = tab.p[0];
    // one layer alone: 128 workgroups per unit of occupancy (more only adds partial tiles); several layers share one
    // workgroup per CU (measured at 128 / 256 / 512 / 768 / 1024 in all: 256 is the fastest for every channel count)
    static const int want = erf_debug_env("ERF_WGRAD1D_WGS", 128);
    static const int want_many = erf_debug_env("ERF_WGRAD1D_WGS_MANY", 256);
    const size_t tile_bytes = (size_t)C * 3 * C * sizeof(float);
    long long grid = count == 1 ? (long long)want * OCC : (long long)want_many / count;
    if (grid > ntiles_min) grid = ntiles_min;
    if ((size_t)grid * KW * tile_bytes > workspace_bytes) grid = (long long)(workspace_bytes / (KW * tile_bytes));
    if (grid < 1) return ERF_EUNSUPPORTED;
    auto kern = wgrad1d_kernel<C, CG, KW, CPW, M, KIX, KIZ, OCC>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3((int)grid, count), dim3(NW * 64), lds, stream, tab);
    int rc = erf_check_launch();
    if (rc) return rc;
    for (int i = 0; i < count; ++i) {
        if (descs) wgrad_reduce_defer(&descs[i]);
        rc = wgrad_reduce_launch(jobs[i].workspace, (int)grid * KW, C, 3 * C, C, maps[i], jobs[i].dw, stream);
        if (descs) wgrad_reduce_defer(nullptr);
        if (rc) return rc;
    }
    return ERF_OK;
}

int wgrad1d_many_launch(const W1dJob* jobs, int count, int dtype, int N, int H, int W, int C, const PackParams* maps,
                        size_t workspace_bytes, WgradReduceDesc* descs, hipStream_t stream) {
    if (dtype != 1) return ERF_EUNSUPPORTED;
    static const int off = erf_debug_env("ERF_WGRAD1D_OFF", 0);
    if (off) return ERF_EUNSUPPORTED;
    if (C == 128) return launch_w1d<128, 8, 1, 3, 128, 6, 4>(jobs, count, N, H, W, maps, workspace_bytes, descs, stream);
    if (C == 64) return launch_w1d<64, 4, 1, 3, 128, 5, 4>(jobs, count, N, H, W, maps, workspace_bytes, descs, stream);
    if (C == 16) return launch_w1d<16, 1, 4, 3, 512, 5, 4>(jobs, count, N, H, W, maps, workspace_bytes, descs, stream);
    return ERF_EUNSUPPORTED;
}

int wgrad1d_launch(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis, int dilation,
                   const PackParams& map, float* workspace, size_t workspace_bytes, hipStream_t stream) {
    const W1dJob job{x, dz, dw, workspace, axis, dilation};
    return wgrad1d_many_launch(&job, 1, dtype, N, H, W, C, &map, workspace_bytes, nullptr, stream);
}
