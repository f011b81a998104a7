// UpsamplerBlock (bf16): y = act(scale * ConvTranspose2d(3x3, stride 2, pad 1, output_pad 1)(x) + shift), one launch.
//
// Output pixel (2i + a, 2j + b) only sees the input pixels (i + di, j + dj), di <= a, dj <= b: four parity
// phases with 1 / 2 / 2 / 4 taps (the packed weights are laid out per phase, see erf_upsampler_pack).
// Tile: R x 16 input pixels -> 2R x 32 output pixels.
//  * the (R+1) x 17 input pixels go to LDS by DMA (buffer_load ... lds; pixels past the image edge are
//    lanes with an out-of-range offset and read zeros, which is exactly their contribution);
//  * each B fragment (16 consecutive input columns, one of the four (di, dj) positions) is read ONCE and
//    fed to every phase that uses it (9 MFMAs per 4 reads), the four phases' accumulators live together;
//    all weights (9 taps x Cin per output-channel tile) stay in registers for the launch;
//  * the four phases interleave into a [2R][32][Cout] staging tile that leaves as whole 16-byte chunks of
//    contiguous output rows.  The gather kernel this replaces wrote 8-byte pieces 2 pixels apart.
#include "common.cuh"
#include "misc.cuh"
#include "conv1d_tile.cuh"

typedef bf16raw T;

struct UpParams {
    const T* x;
    T* y;
    const void* w;
    const float* scale;
    const float* shift;
    int N, Hin, Win, relu;
    int CA, CB, ntiles;
    FastDiv fCB, fCA;
    uint32_t xbytes, ybytes;
};

// phase ph = 2a + b; along one axis parity 0 has the tap list [d = 0], parity 1 has [d = 1, d = 0]
__host__ __device__ constexpr int up_ntaps(int ph) { return ((ph >> 1) + 1) * ((ph & 1) + 1); }
__host__ __device__ constexpr int up_tap_base(int ph) { return ph == 0 ? 0 : ph == 1 ? 1 : ph == 2 ? 3 : 5; }    // in taps
__host__ __device__ constexpr int up_axis_index(int parity, int d) { return parity == 0 ? (d == 0 ? 0 : -1) : (d == 1 ? 0 : 1); }

template <int CIN, int COUT, int WM, int WN, int R, int KI, int OCC>
__global__ __launch_bounds__(WM * WN * 64, OCC)
void up_kernel(const UpParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CH = CIN / 8;
    constexpr int PIXB = CIN * 2;
    constexpr int OPIXB = COUT * 2;
    constexpr int OCH = COUT / 8;
    constexpr int NW = WM * WN, NT = NW * 64;
    constexpr int CT = COUT / 16;
    static_assert(CT % WN == 0 && CT / WN == 1, "one output-channel tile per wave");
    constexpr int PTW = R / WM;
    constexpr int KST = CH / 4;                      // K-steps per tap
    constexpr int NCL = CH >= 16 ? 4 : KST;          // chunk classes (CH <= 16 here)
    constexpr int RP = 32;                           // LDS slots per input row (17 used)
    constexpr int STG = KI * NW * 1024;
    constexpr int NS = (2 * R * 32 * OCH) / NT;
    static_assert(CH >= 4 && CH <= 16 && (2 * R * 32 * OCH) % NT == 0 && R % WM == 0, "layout");
    static_assert((R + 1) * RP * CH <= KI * NW * 64, "input tile fits the DMA plan");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int Wout = 2 * p.Win;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.xbytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.ybytes);

    // weights of this wave's output-channel tile, all phases: wreg[tap_base(ph) * KST + ks_in_phase]
    i32x4 wreg[9 * KST];
    {
        const i32x4* wp = (const i32x4*)p.w;
        int chunk_off = 0;
#pragma unroll
        for (int ph = 0; ph < 4; ++ph) {
            const int ksp = up_ntaps(ph) * KST;
#pragma unroll
            for (int ks = 0; ks < ksp; ++ks) wreg[up_tap_base(ph) * KST + ks] = wp[chunk_off + (wn * ksp + ks) * 64 + lane];
            chunk_off += CT * ksp * 64;
        }
    }
    const int co = wn * 16 + q * 4;
    f32x4 sc = p.scale ? *(const f32x4*)(p.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
    f32x4 sh = p.shift ? *(const f32x4*)(p.shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- launch-invariant addressing
    uint32_t in_rel[KI];
    int in_us[KI];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int u = pix / RP, s = pix % RP;
        in_rel[k] = (uint32_t)(u * p.Win + s) * PIXB + swz<PIXB>(pix, slot) * 16;
        in_us[k] = ((u <= R && s <= 16 ? u : 255) << 8) | s;
    }
    int lb[2][NCL];                                  // [dj][chunk class]; row (r + di) is an immediate
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int cl = 0; cl < NCL; ++cl) {
            const int pix = (wm * PTW) * RP + r16 + dj;
            lb[dj][cl] = pix * PIXB + swz<PIXB>(pix, cl * 4 + q) * 16;
        }
    int stg_w[2];                                    // [b]: staging write offset of output column 2*r16 + b, output row 2*wm*PTW
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int so = (2 * wm * PTW) * 32 + 2 * r16 + b;
        stg_w[b] = STG + so * OPIXB + swz<OPIXB>(so, wn * 2 + (q >> 1)) * 16 + (q & 1) * 8;
    }
    int rb_lds[NS], out_rc[NS];
    uint32_t out_rel[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cid = tid + s * NT;
        const int so = cid / OCH, chunk = cid % OCH;
        const int orow = so / 32, ocol = so % 32;
        rb_lds[s] = STG + so * OPIXB + swz<OPIXB>(so, chunk) * 16;
        out_rc[s] = (orow << 8) | ocol;
        out_rel[s] = (uint32_t)(orow * Wout + ocol) * OPIXB + chunk * 16;
    }
#pragma unroll
    for (int i = 0; i < 9 * KST; ++i) tie_reg(wreg[i]);
    asm volatile("" : "+v"(sc), "+v"(sh), "+v"(stg_w[0]), "+v"(stg_w[1]));
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(in_rel[k]), "+v"(in_us[k]));
#pragma unroll
    for (int dj = 0; dj < 2; ++dj)
#pragma unroll
        for (int cl = 0; cl < NCL; ++cl) asm volatile("" : "+v"(lb[dj][cl]));
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(rb_lds[s]), "+v"(out_rc[s]), "+v"(out_rel[s]));
    const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();

    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        uint32_t t = (uint32_t)tile;
        uint32_t nq = fast_div(t, p.fCB);
        const int cb = (int)(t - nq * p.fCB.div); t = nq;
        nq = fast_div(t, p.fCA);
        const int ca = (int)(t - nq * p.fCA.div);
        const int n = (int)nq;
        const int i0 = ca * R, j0 = cb * 16;
        const uint32_t in_soff = (uint32_t)((n * p.Hin + i0) * p.Win + j0) * PIXB;
        int uhi = p.Hin - i0; uhi = uhi < R + 1 ? uhi : R + 1;
        int shi = p.Win - j0; shi = shi < 17 ? shi : 17;

#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int u = in_us[k] >> 8, s = in_us[k] & 0xFF;
            const bool ok = u < uhi && s < shi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (k * NW + wave) * 1024), 16,
                                                     ok ? in_rel[k] : ERF_OOB, in_soff, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

        static_for<0, PTW>([&](auto pti) {
            constexpr int pt = decltype(pti)::value;
            f32x4 acc[4];
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) acc[ph] = f32x4{0.f, 0.f, 0.f, 0.f};
            // the 2 x KST fragments of input row pt + di (positions dj = 0, 1), one input row at a time
            static_for<0, 2>([&](auto dii) {
                constexpr int di = decltype(dii)::value;
                i32x4 bb[2 * KST];
                static_for<0, 2 * KST>([&](auto fi) {
                    constexpr int f = decltype(fi)::value;
                    constexpr int dj = f / KST, kc = f % KST;
                    bb[f] = lds_read_b128<(pt + di) * RP * PIXB>(lb[dj][kc % NCL]);
                });
                wait_lgkm<0>();
                static_for<0, 2 * KST>([&](auto fi) { tie_reg(bb[decltype(fi)::value]); });
                static_for<0, 2 * KST>([&](auto fi) {
                    constexpr int f = decltype(fi)::value;
                    constexpr int dj = f / KST, kc = f % KST;
                    static_for<0, 4>([&](auto phi) {
                        constexpr int ph = decltype(phi)::value;
                        constexpr int ia = up_axis_index(ph >> 1, di), ib = up_axis_index(ph & 1, dj);
                        if constexpr (ia >= 0 && ib >= 0) {
                            constexpr int tapi = ia * ((ph & 1) + 1) + ib;
                            acc[ph] = mma_step<T>(wreg[(up_tap_base(ph) + tapi) * KST + kc], bb[f], acc[ph]);
                        }
                    });
                });
            });
            // four output pixels of every base pixel -> staging
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                f32x4 v = acc[ph];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sc[e] + sh[e], floor_v);
                *(i32x2*)(smem + stg_w[ph & 1] + (2 * pt + (ph >> 1)) * 32 * OPIXB) = pack_bf16x4(v);
            }
        });
        ERF_LDS_BARRIER();

        i32x4 outv[NS];
        uint32_t ooff[NS];
        const uint32_t out_soff = (uint32_t)((n * 2 * p.Hin + 2 * i0) * Wout + 2 * j0) * OPIXB;
        const int rhi = 2 * (p.Hin - i0), chi = 2 * (p.Win - j0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int orow = out_rc[s] >> 8, ocol = out_rc[s] & 0xFF;
            ooff[s] = (orow < rhi && ocol < chi) ? out_rel[s] : ERF_OOB;
            outv[s] = *(const i32x4*)(smem + rb_lds[s]);
        }
        wait_lgkm<0>();
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[s]), rs_y, ooff[s], out_soff, 0);
        // nothing may rewrite the store data registers right behind the stores; the barrier also frees tile and staging
        asm volatile("s_nop 7\n\ts_barrier" ::: "memory");
    }
#endif
}

template <int CIN, int COUT, int WM, int WN, int R, int KI>
static int launch_up(UpParams p, hipStream_t stream) {
    constexpr int NW = WM * WN;
    constexpr size_t lds = (size_t)KI * NW * 1024 + (size_t)2 * R * 32 * COUT * 2;
    static_assert(lds <= 160 * 1024, "LDS budget");
    constexpr int OCC = 2 * lds <= 160 * 1024 ? 2 : 1;
    p.CA = (p.Hin + R - 1) / R;
    p.CB = (p.Win + 15) / 16;
    const long long nt = (long long)p.N * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA);
    const long long xb = (long long)p.N * p.Hin * p.Win * CIN * 2, yb = 4LL * p.N * p.Hin * p.Win * COUT * 2;
    if (xb + (long long)(R + 2) * p.Win * CIN * 2 >= 0x7FFFFFFFLL || yb >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.xbytes = (uint32_t)xb; p.ybytes = (uint32_t)yb;
    int grid = 256 * OCC;
    if (grid > p.ntiles) grid = p.ntiles;
    auto kern = up_kernel<CIN, COUT, WM, WN, R, KI, OCC>;
    if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);
    return erf_check_launch();
}

int up_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
              int Cin, int Cout, int relu, hipStream_t stream) {
    if (dtype != 1) return ERF_EUNSUPPORTED;
    UpParams p{};
    p.x = (const T*)x; p.y = (T*)y; p.w = packed; p.scale = scale; p.shift = shift;
    p.N = N; p.Hin = Hin; p.Win = Win; p.relu = relu;
    if (Cin == 128 && Cout == 64) return launch_up<128, 64, 1, 4, 4, 10>(p, stream);
    if (Cin == 64 && Cout == 16) return launch_up<64, 16, 4, 1, 8, 9>(p, stream);
    return ERF_EUNSUPPORTED;
}
