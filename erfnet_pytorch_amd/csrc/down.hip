// DownsamplerBlock (bf16): y = act(scale * cat[conv3x3 s2 p1 (x), maxpool2x2 (x)] + shift), one launch.
//
// Tile: R output rows x 16 output columns of one image.  The (2R+1) x 33 input pixels it needs are DMA-ed
// into LDS (buffer_load ... lds; out-of-image pixels are lanes with an out-of-range offset and read zeros),
// de-interleaved by column parity on the way: every LDS row holds [16 even columns | 17 odd columns | pad]
// (48 slots), so that the 16 lanes of an MFMA B-fragment -- 16 consecutive OUTPUT columns, i.e. input columns
// two apart -- read 16 consecutive slots for each of the three horizontal taps (conflict-free with the usual
// XOR swizzle of the 16-byte chunk index).  The global gather kernel this replaces fetched every input line
// several times over (16-byte reads 256 bytes apart).
//  * GEMM: K = 9 taps x Cin, weights (packed A fragments, [ct][kstep][lane]) in registers for the launch;
//    per-lane LDS offsets for every K-step are launch-invariant, the output row is an immediate.
//  * max-pool: same LDS tile (rows 2r+1, 2r+2; even slot c and odd slot c+1), so the input is read once.
//  * epilogue: conv and pool halves are staged as [pixel][Cout] bf16 and stored as whole 16-byte chunks.
#include "common.cuh"
#include "misc.cuh"
#include "conv1d_tile.cuh"

typedef bf16raw T;

struct DownParams {
    const T* x;
    T* y;
    const void* w;
    const float* scale;
    const float* shift;
    int N, Hin, Win, relu;
    int CA, CB, ntiles;          // tiles along output rows / columns
    FastDiv fCB, fCA;
    uint32_t xbytes, ybytes, pad;
};

template <int CIN, int COUT, int WM, int WN, int R, int KI, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC)
void down_kernel(const DownParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CC = COUT - CIN;                 // convolution channels
    constexpr int CH = CIN / 8;
    constexpr int PIXB = CIN * 2;
    constexpr int OPIXB = COUT * 2;
    constexpr int OCH = COUT / 8;
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int CT = (CC + 15) / 16;
    constexpr int CTW = CT / WN;
    constexpr int PTW = R / WM;                     // pixel tile = one output row of 16 columns
    constexpr int KS = (9 * CH + 3) / 4;
    constexpr int RP = 48;                          // LDS slots per input row
    constexpr int NROW = 2 * R + 1;
    constexpr int M = R * 16;
    constexpr int NS = (M * OCH) / NT;
    constexpr int NP = (M * CH + NT - 1) / NT;      // pool items per thread
    static_assert(CT % WN == 0 && R % WM == 0 && (M * OCH) % NT == 0, "tile split");
    static_assert(NROW * RP * CH <= KI * NW * 64, "input tile fits the DMA plan");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STG = KI * NW * 1024;             // staging [M][COUT] behind the (padded) input tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int W = p.Win, Ho = p.Hin / 2, Wo = p.Win / 2;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, p.xbytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.ybytes);

    i32x4 wreg[CTW][KS];
    {
        const i32x4* wp = (const i32x4*)p.w;
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[c][ks] = wp[((wn * CTW + c) * KS + ks) * 64 + lane];
    }
    f32x4 sc[CTW], sh[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int co = (wn * CTW + c) * 16 + q * 4;          // < CC rounded up to 16: scale/shift have COUT >= that many entries
        sc[c] = p.scale ? *(const f32x4*)(p.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
        sh[c] = p.shift ? *(const f32x4*)(p.shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- launch-invariant addressing
    // DMA slot -> (input row u, slot s in the row, chunk); input column relative to the tile's first column 2*ow0 - 1
    uint32_t in_rel[KI];
    int in_uw[KI];                         // (u << 8) | wrel, or u = 255 for pad slots
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int u = pix / RP, s = pix % RP;
        const int wrel = s < 16 ? 2 * s + 1 : 2 * (s - 16);
        const bool real = u < NROW && s <= 32;
        in_rel[k] = (uint32_t)(u * W + wrel) * PIXB + swz<PIXB>(pix, slot) * 16;
        in_uw[k] = ((real ? u : 255) << 8) | wrel;
    }
    // B-fragment offsets per K-step: tap (kh, kw) and chunk folded in; output row r adds the immediate 2*r*RP*PIXB
    int lbk[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int gi = ks * 4 + q;
        const int tap = gi / CH < 8 ? gi / CH : 8, chunk = gi % CH;
        const int kh = tap / 3, kw = tap % 3;
        const int s = kw == 1 ? r16 : (kw == 0 ? 16 + r16 : 17 + r16);      // input column 2*ow - 1 + kw
        const int pix = (2 * wm * PTW + kh) * RP + s;
        lbk[ks] = pix * PIXB + swz<PIXB>(pix, chunk) * 16;
    }
    int stg_w[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int chunk = (wn * CTW + c) * 2 + (q >> 1);                     // 16-byte chunk of the conv half
        stg_w[c] = STG + ((wm * PTW) * 16 + r16) * OPIXB + swz<OPIXB>(r16, chunk) * 16 + (q & 1) * 8;       // rows are 16-aligned: key = r16
    }
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tie_reg(wreg[c][ks]);
        asm volatile("" : "+v"(sc[c]), "+v"(sh[c]), "+v"(stg_w[c]));
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(in_rel[k]), "+v"(in_uw[k]));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(lbk[ks]));
    const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();

    // scale / shift of the pooled channels this thread writes: the same for every tile (loading them inside the tile loop put
    // a memory round trip between the matrix phase and the stores of every tile)
    // (in registers only where they are there: the 64 -> 128 instance holds 72 weight registers and would spill; it keeps
    // the 2 x CIN values in LDS behind the staging area instead)
    constexpr bool HOIST = CIN <= 16;
    constexpr int AFF = STG + R * 16 * COUT * 2;
    if constexpr (!HOIST) {
        for (int i = tid; i < CIN; i += NT) {
            ((float*)(smem + AFF))[i] = p.scale ? p.scale[CC + i] : 1.f;
            ((float*)(smem + AFF))[CIN + i] = p.shift ? p.shift[CC + i] : 0.f;
        }
        ERF_LDS_BARRIER();
    }
    float pool_s[HOIST ? NP : 1][8], pool_t[HOIST ? NP : 1][8];
#pragma unroll
    for (int it = 0; it < (HOIST ? NP : 0); ++it) {
        const int ch0 = CC + ((tid + it * NT) % CH) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            pool_s[it][e] = p.scale ? p.scale[ch0 + e] : 1.f;
            pool_t[it][e] = p.shift ? p.shift[ch0 + e] : 0.f;
        }
    }
    // Tile decode + input DMA as one step: a tile's input is requested as soon as the previous tile's has been consumed
    // (before that tile's read-back and stores), not at the top of its own iteration.
    struct TileGeo { int n, oh0, ow0; };
    auto request = [&](int tile) {
        uint32_t t = (uint32_t)tile;
        uint32_t nq = fast_div(t, p.fCB);
        const int cb = (int)(t - nq * p.fCB.div); t = nq;
        nq = fast_div(t, p.fCA);
        const int ca = (int)(t - nq * p.fCA.div);
        const int n = (int)nq;
        const int oh0 = ca * R, ow0 = cb * 16;
        // input origin (row 2*oh0 - 1, column 2*ow0 - 1); the descriptor starts (W + 1) pixels before x
        const uint32_t in_soff = (uint32_t)((n * p.Hin + 2 * oh0) * W + 2 * ow0) * PIXB;
        const int ulo = oh0 == 0 ? 1 : 0;
        int uhi = p.Hin - (2 * oh0 - 1);
        uhi = uhi < NROW ? uhi : NROW;
        const int wlo = ow0 == 0 ? 1 : 0;
        const int whi = W - (2 * ow0 - 1);
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int u = in_uw[k] >> 8, wr = in_uw[k] & 0xFF;
            const bool ok = (unsigned)(u - ulo) < (unsigned)(uhi - ulo) && wr >= wlo && wr < whi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (k * NW + wave) * 1024), 16,
                                                     ok ? in_rel[k] : ERF_OOB, in_soff, 0, 0);
        }
        return TileGeo{n, oh0, ow0};
    };
    TileGeo geo = {0, 0, 0};
    if ((int)blockIdx.x < p.ntiles) geo = request(blockIdx.x);
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int n = geo.n, oh0 = geo.oh0, ow0 = geo.ow0;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

        // ---- MFMA: rows of 16 output pixels x this wave's channel tiles
        f32x4 acc[PTW][CTW];
        static_for<0, PTW>([&](auto pti) {
            constexpr int pt = decltype(pti)::value;
            i32x4 bb[KS];
            static_for<0, KS>([&](auto ksi) { bb[decltype(ksi)::value] = lds_read_b128<pt * 2 * RP * PIXB>(lbk[decltype(ksi)::value]); });
            wait_lgkm<0>();
            static_for<0, KS>([&](auto ksi) { tie_reg(bb[decltype(ksi)::value]); });
            static_for<0, KS>([&](auto ksi) {
                constexpr int ks = decltype(ksi)::value;
                i32x4 b = bb[ks];
                if constexpr ((9 * CH) % 4 != 0) {
                    if (ks * 4 + q >= 9 * CH) b = i32x4{0, 0, 0, 0};
                }
#pragma unroll
                for (int c = 0; c < CTW; ++c)
                    acc[pt][c] = mma_step<T>(wreg[c][ks], b, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
            });
        });

        // ---- conv half -> staging (rows beyond CC are zero weights and are not written)
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                if ((wn * CTW + c) * 16 + q * 4 >= CC) continue;
                f32x4 v = acc[pt][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[c][e] + sh[c][e], floor_v);
                *(i32x2*)(smem + stg_w[c] + pt * 16 * OPIXB) = pack_bf16x4(v);
            }
        // ---- pooled half -> staging: max over input rows 2r+1, 2r+2 (tile rows), columns even slot c / odd slot c+1
#pragma unroll
        for (int it = 0; it < NP; ++it) {
            const int id = tid + it * NT;
            if (id < M * CH) {
                const int m = id / CH, chunk = id % CH;
                const int r = m >> 4, c = m & 15;
                float best[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) best[e] = -__builtin_huge_valf();
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int pix = (2 * r + 1 + a) * RP + (b == 0 ? c : 17 + c);
                        float f[8];
                        ChunkOps<T>::unpack(*(const i32x4*)(smem + pix * PIXB + swz<PIXB>(pix, chunk) * 16), f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], f[e]);
                    }
                if constexpr (HOIST) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e] * pool_s[it][e] + pool_t[it][e], floor_v);
                } else {
                    const float* const as_ = (const float*)(smem + AFF) + chunk * 8;      // staged once per launch, see above
#pragma unroll
                    for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e] * as_[e] + as_[CIN + e], floor_v);
                }
                *(i32x4*)(smem + STG + m * OPIXB + swz<OPIXB>(m, CC / 8 + chunk) * 16) = ChunkOps<T>::pack(best);
            }
        }
        ERF_LDS_BARRIER();
        if (tile + (int)gridDim.x < p.ntiles) geo = request(tile + gridDim.x);      // the input tile has been consumed

        // ---- coalesced stores of whole output pixels
        i32x4 outv[NS];
        uint32_t ooff[NS];
        const uint32_t out_soff = (uint32_t)((n * Ho + oh0) * Wo + ow0) * OPIXB;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int cid = tid + s * NT;
            const int m = cid / OCH, chunk = cid % OCH;
            const int r = m >> 4, c = m & 15;
            const bool ok = oh0 + r < Ho && ow0 + c < Wo;
            ooff[s] = ok ? (uint32_t)(r * Wo + c) * OPIXB + chunk * 16 : ERF_OOB;
            outv[s] = *(const i32x4*)(smem + STG + m * OPIXB + swz<OPIXB>(m, chunk) * 16);
        }
        wait_lgkm<0>();
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[s]), rs_y, ooff[s], out_soff, 0);
        // nothing may rewrite the store data registers right behind the stores; the barrier also frees the tile and the staging
        asm volatile("s_nop 7\n\ts_barrier" ::: "memory");
    }
#endif
}

template <int CIN, int COUT, int WM, int WN, int R, int KI>
static int launch_down(DownParams p, hipStream_t stream) {
    constexpr int NW = WM * WN;
    constexpr size_t lds = (size_t)KI * NW * 1024 + (size_t)R * 16 * COUT * 2 + (CIN <= 16 ? 0 : 2 * CIN * sizeof(float));
    static_assert(lds <= 160 * 1024, "LDS budget");
    constexpr int OCC = 2 * lds <= 160 * 1024 ? 2 : 1;
    const int Ho = p.Hin / 2, Wo = p.Win / 2;
    p.CA = (Ho + R - 1) / R;
    p.CB = (Wo + 15) / 16;
    const long long nt = (long long)p.N * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA);
    const long long xb = (long long)p.N * p.Hin * p.Win * CIN * 2, yb = (long long)p.N * Ho * Wo * COUT * 2;
    const long long pad = (long long)(p.Win + 1) * CIN * 2;
    if (xb + (long long)(2 * R + 4) * p.Win * CIN * 2 >= 0x7FFFFFFFLL || yb >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.xbytes = (uint32_t)xb; p.ybytes = (uint32_t)yb; p.pad = (uint32_t)pad;
    int grid = 256 * OCC;
    if (grid > p.ntiles) grid = p.ntiles;
    auto kern = down_kernel<CIN, COUT, WM, WN, R, KI, OCC>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
    return erf_check_launch();
}

int down_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin,
                int Win, int Cin, int Cout, int relu, hipStream_t stream) {
    if (dtype != 1 || ((Hin | Win) & 1)) return ERF_EUNSUPPORTED;
    DownParams p{};
    p.x = (const T*)x; p.y = (T*)y; p.w = packed; p.scale = scale; p.shift = shift;
    p.N = N; p.Hin = Hin; p.Win = Win; p.relu = relu;
    if (Cin == 64 && Cout == 128) return launch_down<64, 128, 1, 4, 4, 14>(p, stream);
    if (Cin == 16 && Cout == 64) return launch_down<16, 64, 4, 1, 8, 7>(p, stream);
    return ERF_EUNSUPPORTED;
}
