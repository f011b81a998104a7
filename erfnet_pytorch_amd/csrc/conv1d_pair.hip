// Fused pair of factorized convolutions (bf16, inference):
//
//     y = act( sB * conv1x3_d( relu( sA * conv3x1_d(x) + tA ) ) + tB (+ residual) )
//
// i.e. one half of a non_bottleneck_1d block in a single launch: the intermediate feature map
// never leaves the CU.
//
// Tile geometry.  Both convolutions have the same dilation d, so the pixels with
// (h mod d, w mod d) = (rho_h, rho_w) form an independent undilated 2-D problem (a "lattice").
// A tile is R consecutive lattice rows x TC consecutive lattice columns of one lattice:
//   * input  (R+2) x TC pixels, DMA-ed into LDS (double buffered, zeros outside the image);
//   * GEMM 1 (3x1, taps = LDS rows i, i+1, i+2) produces the R x TC intermediate, written to LDS
//     as bf16 in rows of TC+2 slots whose first and last slot stay zero (the 1x3 zero padding);
//     intermediate pixels right of the image are forced to zero for the same reason;
//   * GEMM 2 (1x3, taps = slots j, j+1, j+2) produces R x TC outputs, of which columns 1..TC-2
//     are complete; column 0 is complete when the tile starts at the image's left edge and column
//     TC-1 when it is the lattice's last column.  Tiles therefore advance by TC-2 columns;
//   * epilogue through LDS with coalesced 16-byte stores, as in the single-convolution kernel.
// Both weight sets stay in registers for the whole launch.
//
// Instruction budget.  Per tile a wave issues 2 x PTW x CTW x KS MFMAs (96 for C = 64 / 128); everything
// else has to stay well below that, so all per-lane addressing is computed once per launch:
// global accesses are raw-buffer instructions taking (launch-invariant per-lane offset) + (per-tile
// scalar offset), out-of-image lanes get an offset beyond the buffer (reads return zeros, stores are
// dropped), and the tile index is decoded with multiply-high "fast division" on the scalar unit.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"
#include <stdlib.h>

typedef bf16raw T;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct PairParams {
    const T* x;
    T* y;
    const void* wA;
    const void* wB;
    const float *scaleA, *shiftA, *scaleB, *shiftB;
    const T* res;
    int d, relu;
    int N, H, W;
    int CA, CB;      // tiles per lattice along rows / columns
    int ntiles;
    FastDiv fCB, fCA, fD;
    uint32_t bytes;  // N*H*W*C*2
    uint32_t pad;    // the input descriptor starts `pad` bytes before x so that the halo row above the image has an offset >= 0
    int dbg;         // diagnostic bit mask (ERF_PAIR_DBG): 1 no image reads, 2 no stores, 4 no MFMA phases
};

// Per-tile scalars (all wave-uniform).
struct PairGeo {
    uint32_t in_soff, out_soff;
    int ulo, uhi;        // valid input rows u
    int jhi;             // valid input / intermediate columns j < jhi
    int ihi;             // valid output rows i < ihi
    int jlo, jho;        // valid output columns jlo <= j < jho
};

template <int C, int R, int TC>
__device__ __forceinline__ PairGeo pair_geo(const PairParams& p, int tile, int tile_end) {
    constexpr int PIXB = C * 2;
    PairGeo g;
    uint32_t t = (uint32_t)tile;
    uint32_t nq = fast_div(t, p.fCB);
    const int tw = (int)(t - nq * p.fCB.div); t = nq;
    nq = fast_div(t, p.fCA);
    const int th = (int)(t - nq * p.fCA.div); t = nq;
    nq = fast_div(t, p.fD);
    const int rho_w = (int)(t - nq * p.fD.div); t = nq;
    nq = fast_div(t, p.fD);
    const int rho_h = (int)(t - nq * p.fD.div);
    const int n = (int)nq;
    const int nh = (int)fast_div((uint32_t)(p.H - rho_h + p.d - 1), p.fD);      // lattice rows / columns of this residue class
    const int nw = (int)fast_div((uint32_t)(p.W - rho_w + p.d - 1), p.fD);
    const int hh0 = th * R - 1, c0 = tw * (TC - 2);
    const int h1 = rho_h + (hh0 + 1) * p.d, w0 = rho_w + c0 * p.d;
    g.out_soff = (uint32_t)((n * p.H + h1) * p.W + w0) * PIXB;
    g.in_soff = g.out_soff;                                       // + pad - one lattice row (pad == d*W*PIXB)
    g.ulo = hh0 < 0 ? 1 : 0;
    int uhi = nh - hh0;
    uhi = tile < tile_end ? (uhi < R + 2 ? uhi : R + 2) : 0;
    g.uhi = uhi > g.ulo ? uhi : g.ulo;                            // empty range when the tile lies below the image
    const int rem = nw - c0;
    g.jhi = rem < TC ? rem : TC;
    const int ihi = nh - hh0 - 1;
    g.ihi = ihi < R ? ihi : R;
    g.jlo = tw == 0 ? 0 : 1;
    g.jho = rem <= TC ? rem : TC - 1;
    return g;
}

// The buffer-descriptor type only exists in the device pass, so everything that touches it is
// compiled there alone (the host pass needs just the kernel's signature for the launch stub).
#if defined(__HIP_DEVICE_COMPILE__)
// Input tile -> LDS: KI 16-byte slots per lane; masked slots read zeros (offset beyond the buffer).
template <int KI, int NW>
__device__ __forceinline__ void pair_dma(__amdgpu_buffer_rsrc_t rs, char* buf, const uint32_t (&in_rel)[KI], const int (&in_uj)[KI],
                                         const PairGeo& g, int wave) {
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int u = in_uj[k] >> 16, j = in_uj[k] & 0xFFFF;
        const bool ok = (unsigned)(u - g.ulo) < (unsigned)(g.uhi - g.ulo) && j < g.jhi;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (k * NW + wave) * 1024), 16,
                                                 ok ? in_rel[k] : ERF_OOB, g.in_soff, 0, 0);
    }
}

#endif

struct ImmIn {
    template <int PIXB, int TC> static constexpr int of(int pt, int tap) { return (pt * 16 + tap * TC) * PIXB; }
    template <int NCL> static constexpr int li(int tap, int cl) { return cl; }
};
struct ImmMid {
    template <int PIXB, int TC> static constexpr int of(int pt, int tap) {
        return ((pt / (TC / 16)) * (TC + 2) + (pt % (TC / 16)) * 16) * PIXB;
    }
    template <int NCL> static constexpr int li(int tap, int cl) { return tap * NCL + cl; }
};

// One GEMM of the pair: acc[pt][c] += W[c][ks] x B(pt, ks); B fragments come from LDS at
// lb[class] + immediate through hand-placed reads in groups of GS, group g+1 in flight while group g
// feeds the MFMAs (LDS reads return in order, so "at most GS outstanding" means group g has arrived).
//
// `epi(pt)` consumes the accumulators of pixel tile pt.  It is called for the tiles finished by group g - 1 right
// after group g's matrix instructions are issued, so its VALU / LDS-write work runs under those instructions
// instead of after the whole phase; the tiles finished by the last group follow the loop.
template <int C, int TC, int PTW, int CTW, typename IMM, typename EPI>
__device__ __forceinline__ void mfma_phase(const int* lb, const i32x4 (&wreg)[CTW][(3 * (C / 8) + 3) / 4], f32x4 (&acc)[PTW][CTW], int q,
                                           EPI&& epi) {
    constexpr int CH = C / 8;
    constexpr int PIXB = C * 2;
    constexpr int KS = (3 * CH + 3) / 4;
    constexpr int NCL = CH >= 4 ? (CH < 16 ? CH : 16) / 4 : 1;
    constexpr int GS = (KS % 6 == 0) ? 6 : (KS % 4 == 0 ? 4 : KS);
    constexpr int NG = PTW * KS / GS;
    static_assert((PTW * KS) % GS == 0, "read groups");
    i32x4 bb[2][GS];
    auto issue = [&](auto gi, i32x4* dst) {
        static_for<0, GS>([&](auto e) {
            constexpr int r = decltype(gi)::value * GS + decltype(e)::value;
            constexpr int pt = r / KS, ks = r % KS;
            constexpr int tap = (ks * 4) / CH;
            // CH < 4: the tap differs per lane and is folded into lb[ks] together with the chunk
            constexpr int li = CH >= 4 ? IMM::template li<NCL>(tap, ks % NCL) : ks;
            constexpr int imm = IMM::template of<PIXB, TC>(pt, CH >= 4 ? tap : 0) + (CH > 16 ? (((ks * 4) % CH) & ~15) * 16 : 0);
            static_assert(imm < 65536, "ds offset field");
            dst[decltype(e)::value] = lds_read_b128<imm>(lb[li]);
        });
    };
    issue(std::integral_constant<int, 0>{}, bb[0]);
    static_for<0, NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        i32x4* cb = bb[g & 1];
        if constexpr (g + 1 < NG) {
            issue(std::integral_constant<int, g + 1>{}, bb[(g + 1) & 1]);
            wait_lgkm<GS>();
        } else {
            wait_lgkm<0>();
        }
        static_for<0, GS>([&](auto e) { tie_reg(cb[decltype(e)::value]); });
        static_for<0, GS>([&](auto e) {
            constexpr int r = g * GS + decltype(e)::value;
            constexpr int pt = r / KS, ks = r % KS;
            i32x4 b = cb[decltype(e)::value];
            if constexpr ((3 * CH) % 4 != 0) {
                if (ks * 4 + q >= 3 * CH) b = i32x4{0, 0, 0, 0};
            }
#pragma unroll
            for (int c = 0; c < CTW; ++c)            // the first K-step starts from a literal zero: no accumulator clearing
                acc[pt][c] = mma_step<T>(wreg[c][ks], b, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
        });
        constexpr int done_lo = g == 0 ? 0 : ((g - 1) * GS) / KS;       // tiles complete before group g - 1 started
        constexpr int done_hi = (g * GS) / KS;                           // ... before this group started
        static_for<done_lo, done_hi>([&](auto pt) { epi(pt); });
    });
    static_for<((NG - 1) * GS) / KS, PTW>([&](auto pt) { epi(pt); });
}


// GEMM 1 (the 3x1 convolution over the input tile) with every pixel fragment read ONCE.  A tap shifts the pixel index by
// TC, a whole number S of 16-pixel tiles, so fragment (pixel tile pt, tap t, chunk kc) IS fragment (LDS tile pt + t S, kc):
// walking the LDS tiles j = 0 .. PTW + 2S - 1 and feeding each fragment to the (up to) three accumulators pt = j - t S it
// belongs to needs (PTW + 2S) / (3 PTW) of the reads of mfma_phase -- 10 instead of 24 per chunk at PTW = 8 -- and LDS reads
// are what bounds this phase.  Every accumulator still receives its K-steps in the order tap 0, 1, 2 / chunks ascending, so
// the sums are bit-identical to mfma_phase's.
template <int C, int TC, int PTW, int CTW, typename EPI>
__device__ __forceinline__ void mfma_phase_shared(const int* lb, const i32x4 (&wreg)[CTW][(3 * (C / 8) + 3) / 4], f32x4 (&acc)[PTW][CTW],
                                                  EPI&& epi) {
    constexpr int CH = C / 8;
    constexpr int PIXB = C * 2;
    constexpr int KPT = CH / 4;                        // K-steps per tap
    constexpr int NCL = (CH < 16 ? CH : 16) / 4;
    constexpr int S = TC / 16;
    constexpr int NJ = PTW + 2 * S;
    constexpr int NR = NJ * KPT;                       // fragment reads of the phase
    constexpr int GS = NR % 6 == 0 ? 6 : NR % 5 == 0 ? 5 : NR % 4 == 0 ? 4 : NR % 3 == 0 ? 3 : 2;
    constexpr int NG = NR / GS;
    static_assert(CH >= 4 && CH % 4 == 0 && TC % 16 == 0 && NR % GS == 0, "shared-fragment phase");
    i32x4 bb[2][GS];
    auto issue = [&](auto gi, i32x4* dst) {
        static_for<0, GS>([&](auto e) {
            constexpr int r = decltype(gi)::value * GS + decltype(e)::value;
            constexpr int j = r / KPT, kc = r % KPT;
            constexpr int imm = j * 16 * PIXB + (CH > 16 ? ((kc * 4) & ~15) * 16 : 0);
            static_assert(imm < 65536, "ds offset field");
            dst[decltype(e)::value] = lds_read_b128<imm>(lb[kc % NCL]);
        });
    };
    // pixel tiles whose last K-step (fragment j = pt + 2S, last chunk) lies before read r
    auto done_before = [](int r) constexpr { const int n = r / KPT - 2 * S; return n < 0 ? 0 : (n > PTW ? PTW : n); };
    issue(std::integral_constant<int, 0>{}, bb[0]);
    static_for<0, NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        i32x4* cb = bb[g & 1];
        if constexpr (g + 1 < NG) {
            issue(std::integral_constant<int, g + 1>{}, bb[(g + 1) & 1]);
            wait_lgkm<GS>();
        } else {
            wait_lgkm<0>();
        }
        static_for<0, GS>([&](auto e) { tie_reg(cb[decltype(e)::value]); });
        static_for<0, GS>([&](auto e) {
            constexpr int r = g * GS + decltype(e)::value;
            constexpr int j = r / KPT, kc = r % KPT;
            const i32x4 b = cb[decltype(e)::value];
            static_for<0, 3>([&](auto ti) {
                constexpr int tap = decltype(ti)::value;
                constexpr int pt = j - tap * S;
                if constexpr (pt >= 0 && pt < PTW) {
                    constexpr int ks = tap * KPT + kc;
#pragma unroll
                    for (int c = 0; c < CTW; ++c)
                        acc[pt][c] = mma_step<T>(wreg[c][ks], b, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
                }
            });
        });
        constexpr int done_lo = g == 0 ? 0 : done_before((g - 1) * GS);
        constexpr int done_hi = done_before(g * GS);
        static_for<done_lo, done_hi>([&](auto pt) { epi(pt); });
    });
    static_for<done_before((NG - 1) * GS), PTW>([&](auto pt) { epi(pt); });
}


template <int C, int WM, int WN, int M, int TC, int KI, int OCC, bool POST>
__global__ __launch_bounds__(WM * WN * 64, OCC)
void conv1d_pair_kernel(const PairParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int EPC = 8;
    constexpr int CH = C / EPC;
    constexpr int PIXB = C * 2;
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int CT = C / 16;
    constexpr int CTW = CT / WN;
    constexpr int PTW = (M / 16) / WM;
    constexpr int KS = (3 * CH + 3) / 4;
    constexpr int R = M / TC;
    constexpr int RS = TC + 2;
    constexpr int TB = TC / 16;
    constexpr int BUFB = KI * NW * 1024;
    constexpr int MIDB = R * RS * PIXB;
    constexpr int NS = (M * CH) / NT;
    constexpr int NCL = CH >= 4 ? (CH < 16 ? CH : 16) / 4 : 1;
    constexpr int NLB1 = CH >= 4 ? NCL : KS;
    constexpr int NLB2 = CH >= 4 ? 3 * NCL : KS;
    static_assert(CT % WN == 0 && (M / 16) % WM == 0 && M % TC == 0, "tile split");
    static_assert((M * CH) % NT == 0 && M * PIXB <= BUFB, "epilogue staging");
    static_assert((R + 2) * TC * CH <= KI * NW * 64, "input tile fits the DMA plan");
    static_assert(WM == 1 || TB == 1, "wave row split needs 16-column tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MID = 2 * BUFB;                      // byte offset of the intermediate rows

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;
#ifdef ERF_DEBUG
    const uint64_t tentry = __builtin_amdgcn_s_memtime();
#endif

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, (p.dbg & 1) ? 0u : p.bytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, (p.dbg & 2) ? 0u : p.bytes);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res ? p.res : p.x, p.bytes);

    i32x4 wA[CTW][KS], wB[CTW][KS];
    f32x4 sA[CTW], tA[CTW], sB[CTW], tB[CTW];
    {
        const i32x4* pa = (const i32x4*)p.wA;
        const i32x4* pb = (const i32x4*)p.wB;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                wA[c][ks] = pa[((wn * CTW + c) * KS + ks) * 64 + lane];
                wB[c][ks] = pb[((wn * CTW + c) * KS + ks) * 64 + lane];
            }
            const int co = (wn * CTW + c) * 16 + q * 4;
            sA[c] = p.scaleA ? *(const f32x4*)(p.scaleA + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            tA[c] = p.shiftA ? *(const f32x4*)(p.shiftA + co) : f32x4{0.f, 0.f, 0.f, 0.f};
            sB[c] = p.scaleB ? *(const f32x4*)(p.scaleB + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            tB[c] = p.shiftB ? *(const f32x4*)(p.shiftB + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- launch-invariant per-lane addressing
    uint32_t in_rel[KI];                 // DMA source offset relative to the tile's input origin
    int in_uj[KI];                       // (u << 16) | j of that slot; u = 255 for slots past the tile
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int u = pix / TC, j = pix % TC;
        in_rel[k] = (uint32_t)((u * p.W + j) * p.d) * PIXB + swz<PIXB>(pix, slot) * 16;
        in_uj[k] = ((u < R + 2 ? u : 255) << 16) | j;
    }
    int lb1[NLB1], lb2[NLB2];            // B-fragment read offsets: GEMM 1 (without the buffer offset), GEMM 2
    if constexpr (CH >= 4) {
#pragma unroll
        for (int cl = 0; cl < NCL; ++cl) lb1[cl] = ((wm * PTW) * 16 + r16) * PIXB + swz<PIXB>(r16, cl * 4 + q) * 16;
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int cl = 0; cl < NCL; ++cl)
                lb2[tap * NCL + cl] = MID + (wm * PTW) * RS * PIXB + (r16 + tap) * PIXB + swz<PIXB>(r16 + tap, cl * 4 + q) * 16;
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int gi = ks * 4 + q;
            const int tap = gi / CH < 2 ? gi / CH : 2;
            lb1[ks] = ((wm * PTW) * 16 + tap * TC + r16) * PIXB + swz<PIXB>(r16, gi % CH) * 16;
            lb2[ks] = MID + (wm * PTW) * RS * PIXB + (r16 + tap) * PIXB + swz<PIXB>(r16 + tap, gi % CH) * 16;
        }
    }
    int mid_w[CTW], stg_w[CTW];          // accumulator write offsets: intermediate rows / staging (pixel tile 0 of this wave)
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        const int chunk = (wn * CTW + c) * 2 + (q >> 1);
        const int s = r16 + 1;
        mid_w[c] = MID + ((wm * PTW / TB) * RS + s) * PIXB + swz<PIXB>(s, chunk) * 16 + (q & 1) * 8;
        stg_w[c] = ((wm * PTW) * 16 + r16) * PIXB + swz<PIXB>(r16, chunk) * 16 + (q & 1) * 8;
    }
    int rb_lds[NS], out_ij[NS];          // read-back: LDS offset, (i << 16) | j, global offset relative to the tile's output origin
    uint32_t out_rel[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cid = tid + s * NT;
        const int m = cid / CH, chunk = cid % CH;
        const int i = m / TC, j = m % TC;
        rb_lds[s] = m * PIXB + swz<PIXB>(m, chunk) * 16;
        out_ij[s] = (i << 16) | j;
        out_rel[s] = (uint32_t)((i * p.W + j) * p.d) * PIXB + chunk * 16;
    }

    // Pin everything loaded or computed above: a wait for it placed at its first use inside the loop
    // would drain the DMA prefetch, and rematerialising it would cost the VALU work it is here to save.
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            tie_reg(wA[c][ks]);
            tie_reg(wB[c][ks]);
        }
        asm volatile("" : "+v"(sA[c]), "+v"(tA[c]), "+v"(sB[c]), "+v"(tB[c]), "+v"(mid_w[c]), "+v"(stg_w[c]));
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(in_rel[k]), "+v"(in_uj[k]));
#pragma unroll
    for (int i = 0; i < NLB1; ++i) asm volatile("" : "+v"(lb1[i]));
#pragma unroll
    for (int i = 0; i < NLB2; ++i) asm volatile("" : "+v"(lb2[i]));
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(rb_lds[s]), "+v"(out_ij[s]), "+v"(out_rel[s]));

    // intermediate rows start as zeros; only slots 1..TC of each row are ever written
    for (int o = tid * 16; o < MIDB; o += NT * 16) *(i32x4*)(smem + MID + o) = i32x4{0, 0, 0, 0};

    const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();      // max(v, floor_v): ReLU or identity without a select
    // Workgroups are dealt to the 8 XCDs round-robin.  Give every XCD one contiguous eighth of the tile
    // sequence, so that tiles sharing halo rows / columns run on the same XCD at about the same time and
    // the second fetch of a halo pixel is an L2 hit instead of another trip to HBM.
    int tile = blockIdx.x, tile_end = p.ntiles, G = gridDim.x;
    if ((G & 7) == 0) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3);
        tile_end = (xcd + 1) * per < p.ntiles ? (xcd + 1) * per : p.ntiles;
        G >>= 3;
    }
    PairGeo g = pair_geo<C, R, TC>(p, tile, tile_end);
    PairGeo g1 = pair_geo<C, R, TC>(p, tile + G, tile_end);
    pair_dma<KI, NW>(rs_x, smem, in_rel, in_uj, g, wave);
    pair_dma<KI, NW>(rs_x, smem + BUFB, in_rel, in_uj, g1, wave);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");
    ERF_LDS_BARRIER();

#ifdef ERF_DEBUG
    // diagnostic build only (ERF_PAIR_DBG & 256): shader cycles per section of the tile loop, summed over this workgroup's
    // tiles, written over the head of y (scripts/pair_phase_time.py reads them)
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t tprev = __builtin_amdgcn_s_memtime();
    const uint64_t tstart = tprev;
#define PAIR_STAMP(i) { const uint64_t tn = __builtin_amdgcn_s_memtime(); st[i] += tn - tprev; tprev = tn; }
#else
#define PAIR_STAMP(i)
#endif
    int cur = 0;
    for (; tile < tile_end; tile += G, cur ^= 1) {
        const int boff = cur * BUFB;

        // ---------------- GEMM 1: 3x1 over the input tile
        int lb[NLB1];
#pragma unroll
        for (int i = 0; i < NLB1; ++i) lb[i] = lb1[i] + boff;
        f32x4 acc[PTW][CTW];
        if (p.dbg & 4) {
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[pt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // intermediate = relu(sA * acc + tA) -> LDS rows (zero right of the image)
        auto ep1 = [&](auto pti) {
            constexpr int pt = decltype(pti)::value;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                f32x4 v = acc[pt][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e] * sA[c][e] + tA[c][e], 0.f);
                *(i32x2*)(smem + mid_w[c] + ((pt / TB) * RS + (pt % TB) * 16) * PIXB) = pack_bf16x4(v);
            }
        };
        if (p.dbg & 4) static_for<0, PTW>(ep1);
        else if constexpr (CH >= 4 && TB * 16 == TC && (TB == 1 || WM == 1)) mfma_phase_shared<C, TC, PTW, CTW>(lb, wA, acc, ep1);
        else mfma_phase<C, TC, PTW, CTW, ImmIn>(lb, wA, acc, q, ep1);
        if (g.jhi < TC) {                             // wave-uniform and rare: tiles at the right image edge
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
                if ((pt % TB) * 16 + r16 >= g.jhi) {  // WM == 1 whenever TB > 1
#pragma unroll
                    for (int c = 0; c < CTW; ++c) *(i32x2*)(smem + mid_w[c] + ((pt / TB) * RS + (pt % TB) * 16) * PIXB) = i32x2{0, 0};
                }
            }
        }
        PAIR_STAMP(0)
        ERF_LDS_BARRIER();                 // intermediate complete; input tile no longer read
        PAIR_STAMP(1)

        // ---------------- GEMM 2: 1x3 over the intermediate rows
        // accumulators -> LDS as [pixel][C] bf16 (in the consumed input buffer)
        auto ep2 = [&](auto pti) {
            constexpr int pt = decltype(pti)::value;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                f32x4 v = acc[pt][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * sB[c][e] + tB[c][e];
                    if constexpr (!POST) v[e] = fmaxf(v[e], floor_v);
                }
                *(i32x2*)(smem + boff + stg_w[c] + pt * 16 * PIXB) = pack_bf16x4(v);
            }
        };
        if (p.dbg & 4) static_for<0, PTW>(ep2);
        else mfma_phase<C, TC, PTW, CTW, ImmMid>(lb2, wB, acc, q, ep2);
        PAIR_STAMP(2)
        ERF_LDS_BARRIER();                 // staging complete; intermediate no longer read
        PAIR_STAMP(3)

        // ---------------- coalesced read-back (+ residual, ReLU)
        i32x4 outv[NS];
        uint32_t ooff[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = out_ij[s] >> 16, j = out_ij[s] & 0xFFFF;
            const bool ok = i < g.ihi && j >= g.jlo && j < g.jho;
            ooff[s] = ok ? out_rel[s] : ERF_OOB;
            outv[s] = *(const i32x4*)(smem + boff + rb_lds[s]);
        }
        if constexpr (POST) {
            u32x4 rv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) rv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[s], g.out_soff, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float f[8], r[8];
                ChunkOps<T>::unpack(outv[s], f);
                ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, rv[s]), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] + r[e], floor_v);
                outv[s] = ChunkOps<T>::pack(f);
            }
        }
        PAIR_STAMP(4)
        ERF_LDS_BARRIER();                 // staging consumed: the buffer may take tile k+2
        PAIR_STAMP(5)

        const uint32_t out_soff = g.out_soff;
        g = g1;
        g1 = pair_geo<C, R, TC>(p, tile + 2 * G, tile_end);
        pair_dma<KI, NW>(rs_x, smem + boff, in_rel, in_uj, g1, wave);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");      // this wave's share of tile k+1 has landed
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[s]), rs_y, ooff[s], out_soff, 0);
        // the data registers of a 16-byte buffer store must not be rewritten in the cycles right after its issue
        // (the compiler does not guard stores with an SGPR offset): pause before anything else runs
        PAIR_STAMP(6)
        asm volatile("s_nop 7\n\ts_barrier" ::: "memory");             // ... and everyone else's
        PAIR_STAMP(7)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup
#ifdef ERF_DEBUG
    if ((p.dbg & 256) && (tid & 63) == 0) {
        uint64_t* o = (uint64_t*)p.y + (blockIdx.x * NW + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = __builtin_amdgcn_s_memtime() - tstart;
        o[9] = tstart - tentry;                                          // prologue: weights, addressing, first two tiles
    }
#endif
#endif
}

// ---- Producer / consumer form for 128 channels -------------------------------------------------------------------------
// In conv1d_pair_kernel every wave of the one workgroup a CU holds owns 16 output channels of BOTH convolutions and all
// waves walk the phases of a tile in lockstep: every 1 KB pixel-fragment read feeds a single matrix instruction, and
// nothing runs under the epilogue / read-back / store half of a tile.  Here the eight waves split by ROLE:
//   waves 0-3 (producers) hold the 3x1 weights, 32 output channels each: DMA of tile s+1, GEMM 1 of tile s -> intermediate
//                         rows in LDS (two buffers);
//   waves 4-7 (consumers) hold the 1x3 weights, 32 output channels each: GEMM 2 of tile s-1 from the other intermediate
//                         buffer, scale/shift (+ residual) and the stores straight from the accumulators (a lane owns
//                         4 consecutive channels of a pixel: 8-byte pieces, 64 contiguous bytes per pixel and wave).
// A SIMD runs one wave of each role, so one role's matrix instructions cover the other's reads, epilogue and stores;
// every fragment read feeds two matrix instructions; a stage has ONE workgroup barrier (the hand-over of the intermediate
// and of the freshly landed input tile).  The sums are those of conv1d_pair_kernel, K-step for K-step: bit-identical.
template <int C, int R, int TC, bool POST>
__global__ __launch_bounds__(512, 1)
void conv1d_pair_pc_kernel(const PairParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int CH = C / 8;
    constexpr int PIXB = C * 2;
    constexpr int NP = 4;                              // waves per role
    constexpr int CTW = (C / 16) / NP;
    constexpr int M = R * TC, PTW = M / 16;
    constexpr int KS = (3 * CH + 3) / 4;
    constexpr int RS = TC + 2;
    constexpr int INB = (R + 2) * TC * PIXB;           // one input buffer
    constexpr int KIP = INB / (NP * 1024);             // DMA slots per producer lane
    constexpr int MIDB = R * RS * PIXB;                // one intermediate buffer
    constexpr int NCL = (CH < 16 ? CH : 16) / 4;
    static_assert(CH >= 16 && CH % 4 == 0 && TC == 16 && (C / 16) % NP == 0 && KIP * NP * 1024 == INB, "producer/consumer tiling");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int MID = 2 * INB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave / NP, wn = wave % NP;
    const int q = lane >> 4, r16 = lane & 15;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, p.bytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res ? p.res : p.x, p.bytes);

    i32x4 w[CTW][KS];
    f32x4 sc[CTW], sh[CTW];
    {
        const i32x4* pw = (const i32x4*)(role ? p.wB : p.wA);
        const float* ps = role ? p.scaleB : p.scaleA;
        const float* pt = role ? p.shiftB : p.shiftA;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[c][ks] = pw[((wn * CTW + c) * KS + ks) * 64 + lane];
            const int co = (wn * CTW + c) * 16 + q * 4;
            sc[c] = ps ? *(const f32x4*)(ps + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[c] = pt ? *(const f32x4*)(pt + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tie_reg(w[c][ks]);
        asm volatile("" : "+v"(sc[c]), "+v"(sh[c]));
    }

    // both intermediate buffers start as zeros; only slots 1..TC of each row are ever written
    for (int o = tid * 16; o < 2 * MIDB; o += 512 * 16) *(i32x4*)(smem + MID + o) = i32x4{0, 0, 0, 0};

    int tile = blockIdx.x, tile_end = p.ntiles, G = gridDim.x;
    if ((G & 7) == 0) {                                // one contiguous eighth of the tile sequence per XCD (see above)
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3);
        tile_end = (xcd + 1) * per < p.ntiles ? (xcd + 1) * per : p.ntiles;
        G >>= 3;
    }
    const int floor_pk = p.relu ? 0 : (int)0x80008000;             // max_bf16x2(o, floor_pk): ReLU or identity without a select
#ifdef ERF_DEBUG
    // diagnostic build only (ERF_PAIR_DBG & 256): shader cycles per section of a stage, summed over this wave's stages,
    // written over the head of y (scripts/pair_pc_phase_time.py reads them)
    uint64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint64_t tentry = __builtin_amdgcn_s_memtime();
    uint64_t tprev = tentry;
#define PC_STAMP(i) { const uint64_t tn = __builtin_amdgcn_s_memtime(); st[i] += tn - tprev; tprev = tn; }
#else
#define PC_STAMP(i)
#endif

    if (role == 0) {
        // ------------------------------------------------------------------ producers
        uint32_t in_rel[KIP];
        int in_uj[KIP];
#pragma unroll
        for (int k = 0; k < KIP; ++k) {
            const int id = (k * NP + wn) * 64 + lane;
            const int pix = id / CH, slot = id % CH;
            const int u = pix / TC, j = pix % TC;
            in_rel[k] = (uint32_t)((u * p.W + j) * p.d) * PIXB + swz<PIXB>(pix, slot) * 16;
            in_uj[k] = (u << 16) | j;
        }
        int lb1[NCL], mid_w[CTW];
#pragma unroll
        for (int cl = 0; cl < NCL; ++cl) lb1[cl] = r16 * PIXB + swz<PIXB>(r16, cl * 4 + q) * 16;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
            const int chunk = (wn * CTW + c) * 2 + (q >> 1);
            mid_w[c] = MID + (r16 + 1) * PIXB + swz<PIXB>(r16 + 1, chunk) * 16 + (q & 1) * 8;
        }
#pragma unroll
        for (int k = 0; k < KIP; ++k) asm volatile("" : "+v"(in_rel[k]), "+v"(in_uj[k]));
#pragma unroll
        for (int i = 0; i < NCL; ++i) asm volatile("" : "+v"(lb1[i]));
#pragma unroll
        for (int c = 0; c < CTW; ++c) asm volatile("" : "+v"(mid_w[c]));

        PairGeo g = pair_geo<C, R, TC>(p, tile, tile_end);
        pair_dma<KIP, NP>(rs_x, smem, in_rel, in_uj, g, wn);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ERF_LDS_BARRIER();                             // tile 0 has landed, the intermediate buffers are cleared
        PC_STAMP(7)
        for (int cur = 0;; tile += G, cur ^= 1) {
            const bool active = tile < tile_end;
            if (active) {
                const PairGeo gn = pair_geo<C, R, TC>(p, tile + G, tile_end);
                pair_dma<KIP, NP>(rs_x, smem + (cur ^ 1) * INB, in_rel, in_uj, gn, wn);
                PC_STAMP(0)
                int lb[NCL];
#pragma unroll
                for (int i = 0; i < NCL; ++i) lb[i] = lb1[i] + cur * INB;
                const int moff = cur * MIDB;
                f32x4 acc[PTW][CTW];
                auto ep1 = [&](auto pti) {
                    constexpr int pt = decltype(pti)::value;
#pragma unroll
                    for (int c = 0; c < CTW; ++c) {
                        f32x4 v = acc[pt][c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[c][e] + sh[c][e];
                        // ReLU on the rounded pairs (one instruction per two values; the same bits as before the rounding)
                        const int o0 = relu_bf16x2(pack_bf16x2(v[0], v[1])), o1 = relu_bf16x2(pack_bf16x2(v[2], v[3]));
                        *(i32x2*)(smem + moff + mid_w[c] + pt * RS * PIXB) = i32x2{o0, o1};
                    }
                };
                mfma_phase_shared<C, TC, PTW, CTW>(lb, w, acc, ep1);
                if (g.jhi < TC) {                      // intermediate pixels right of the image are the 1x3 zero padding
                    if (r16 >= g.jhi) {
#pragma unroll
                        for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
                            for (int c = 0; c < CTW; ++c) *(i32x2*)(smem + moff + mid_w[c] + pt * RS * PIXB) = i32x2{0, 0};
                    }
                }
                g = gn;
                PC_STAMP(1)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's share of the next tile has landed
                PC_STAMP(2)
            }
            ERF_LDS_BARRIER();
            PC_STAMP(3)
            if (!active) break;
        }
    } else {
        // ------------------------------------------------------------------ consumers
        int lb2[3 * NCL];
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int cl = 0; cl < NCL; ++cl) lb2[tap * NCL + cl] = MID + (r16 + tap) * PIXB + swz<PIXB>(r16 + tap, cl * 4 + q) * 16;
        uint32_t orel[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) orel[c] = (uint32_t)(r16 * p.d) * PIXB + ((wn * CTW + c) * 16 + q * 4) * 2;
#pragma unroll
        for (int i = 0; i < 3 * NCL; ++i) asm volatile("" : "+v"(lb2[i]));
#pragma unroll
        for (int c = 0; c < CTW; ++c) asm volatile("" : "+v"(orel[c]));
        const uint32_t rowb = (uint32_t)(p.W * p.d) * PIXB;           // one lattice row down

        ERF_LDS_BARRIER();
        PC_STAMP(7)
        const int tile0 = tile;
        for (int cur = 0;; tile += G, cur ^= 1) {
            const bool active = tile < tile_end;
            if (tile != tile0) {                       // GEMM 2 of the tile the producers finished in the previous stage
                const PairGeo g = pair_geo<C, R, TC>(p, tile - G, tile_end);
                uint32_t vo[CTW];
                const bool colok = r16 >= g.jlo && r16 < g.jho;
#pragma unroll
                for (int c = 0; c < CTW; ++c) vo[c] = colok ? orel[c] : ERF_OOB;
                i32x2 rv[PTW][CTW];
                if constexpr (POST) {
#pragma unroll
                    for (int pt = 0; pt < PTW; ++pt)
#pragma unroll
                        for (int c = 0; c < CTW; ++c)
                            rv[pt][c] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_r, pt < g.ihi ? vo[c] : ERF_OOB,
                                                                                                       g.out_soff + pt * rowb, 0));
                }
                PC_STAMP(0)
                int lb[3 * NCL];
#pragma unroll
                for (int i = 0; i < 3 * NCL; ++i) lb[i] = lb2[i] + (cur ^ 1) * MIDB;
                f32x4 acc[PTW][CTW];
                auto ep2 = [&](auto pti) {
                    constexpr int pt = decltype(pti)::value;
                    if constexpr (POST && pt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the residual pieces
#pragma unroll
                    for (int c = 0; c < CTW; ++c) {
                        f32x4 v = acc[pt][c];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[c][e] + sh[c][e];
                        int o0 = pack_bf16x2(v[0], v[1]), o1 = pack_bf16x2(v[2], v[3]);
                        if constexpr (POST) {          // same rounding points as the staged epilogue: bf16, + residual, bf16
                            f32x4 f, r;
                            f[0] = __uint_as_float(((uint32_t)o0) << 16); f[1] = __uint_as_float(((uint32_t)o0) & 0xFFFF0000u);
                            f[2] = __uint_as_float(((uint32_t)o1) << 16); f[3] = __uint_as_float(((uint32_t)o1) & 0xFFFF0000u);
                            const i32x2 rr = rv[pt][c];
                            const int r0 = rr[0], r1 = rr[1];
                            r[0] = __uint_as_float(((uint32_t)r0) << 16); r[1] = __uint_as_float(((uint32_t)r0) & 0xFFFF0000u);
                            r[2] = __uint_as_float(((uint32_t)r1) << 16); r[3] = __uint_as_float(((uint32_t)r1) & 0xFFFF0000u);
                            o0 = pack_bf16x2(f[0] + r[0], f[1] + r[1]);
                            o1 = pack_bf16x2(f[2] + r[2], f[3] + r[3]);
                        }
                        // the activation on the rounded pairs: one instruction per two values, the same bits as max(v, floor) before the rounding
                        o0 = max_bf16x2(o0, floor_pk);
                        o1 = max_bf16x2(o1, floor_pk);
                        __builtin_amdgcn_raw_buffer_store_b64(u32x2{(uint32_t)o0, (uint32_t)o1}, rs_y, pt < g.ihi ? vo[c] : ERF_OOB,
                                                              g.out_soff + pt * rowb, 0);
                    }
                    // the data registers of a buffer store with an SGPR offset must not be rewritten right after its issue
                    asm volatile("s_nop 4" ::: "memory");
                };
                mfma_phase<C, TC, PTW, CTW, ImmMid>(lb, w, acc, q, ep2);
                PC_STAMP(1)
            }
            ERF_LDS_BARRIER();
            PC_STAMP(3)
            if (!active) break;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup
#ifdef ERF_DEBUG
    if ((p.dbg & 256) && lane == 0) {
        uint64_t* o = (uint64_t*)p.y + (blockIdx.x * 8 + wave) * 10;
        for (int i = 0; i < 8; ++i) o[i] = st[i];
        o[8] = __builtin_amdgcn_s_memtime() - tentry;
        o[9] = 0;
    }
#endif
#undef PC_STAMP
#endif
}


// ---- Whole-row form for sparse lattices (128 channels, bf16) -------------------------------------------------------------
// With a large dilation on a small map (d = 16 on 64 x 128) a residue class of the 2-D lattice has 4 x 8 points and the
// lattice tiles above are mostly padding.  Here only the ROWS are split into residue classes (rows rho, rho + d, ...: at most
// 4 of them, at most 128 pixels wide) and a class is resident in LDS as a whole (4 x 32 KB):
//   * the 3x1 taps of output row t are the class rows t - 1, t, t + 1 (rows outside the image are skipped, not read);
//   * along a row the 1x3 taps lie d = 16 pixels apart, i.e. exactly one 16-pixel matrix tile: intermediate tile j feeds
//     the output tiles j + 1, j, j - 1 (taps 0, 1, 2), so every intermediate fragment is read once for three matrix
//     instructions per channel tile, and the zero padding left and right of the row is a skipped instruction.
// Roles as in conv1d_pair_pc_kernel: waves 0-3 run GEMM 1 of row t + 1 while waves 4-7 run GEMM 2 / epilogue / stores of row t.
// LDS holds ONE intermediate row (4 x 32 + 32 KB = 160 KB), so the producers keep their accumulators until the consumers
// have finished reading it: two barriers per row.  K-steps per accumulator in the order of the single launches (taps
// ascending, chunks ascending; a skipped tap would have added zeros): bit-identical sums.
struct RowsParams {
    const T* x;
    T* y;
    const void* wA;
    const void* wB;
    const float *scaleA, *shiftA, *scaleB, *shiftB;
    const T* res;
    int d, relu;
    int N, H, W;
    int nclass;          // N * d row classes
    FastDiv fD;
    uint32_t bytes;      // N*H*W*256
};

#if defined(__HIP_DEVICE_COMPILE__)
// GEMM 1 of one output row: taps TLO..THI, lbt[tap][kc] = fragment address of pixel tile 0 in that tap's class row.
template <int TLO, int THI>
__device__ __forceinline__ void rows_gemm1(const int (&lbt)[3][4], const i32x4 (&w)[2][12], f32x4 (&acc)[8][2]) {
    constexpr int NTAP = THI - TLO + 1, NR = 8 * NTAP * 4;                // reads in the order (pixel tile, tap, chunk)
    constexpr int GS = NR % 6 == 0 ? 6 : 4, NG = NR / GS;
    i32x4 bb[2][GS];
    auto issue = [&](auto gi, i32x4* dst) {
        static_for<0, GS>([&](auto e) {
            constexpr int r = decltype(gi)::value * GS + decltype(e)::value;
            constexpr int pt = r / (NTAP * 4), tap = TLO + (r / 4) % NTAP, kc = r % 4;
            dst[decltype(e)::value] = lds_read_b128<pt * 4096>(lbt[tap][kc]);
        });
    };
    issue(std::integral_constant<int, 0>{}, bb[0]);
    static_for<0, NG>([&](auto gi) {
        constexpr int g = decltype(gi)::value;
        i32x4* cb = bb[g & 1];
        if constexpr (g + 1 < NG) {
            issue(std::integral_constant<int, g + 1>{}, bb[(g + 1) & 1]);
            wait_lgkm<GS>();
        } else {
            wait_lgkm<0>();
        }
        static_for<0, GS>([&](auto e) { tie_reg(cb[decltype(e)::value]); });
        static_for<0, GS>([&](auto e) {
            constexpr int r = g * GS + decltype(e)::value;
            constexpr int pt = r / (NTAP * 4), tap = TLO + (r / 4) % NTAP, kc = r % 4, ks = tap * 4 + kc;
            const i32x4 b = cb[decltype(e)::value];
#pragma unroll
            for (int c = 0; c < 2; ++c)
                acc[pt][c] = mma_step<T>(w[c][ks], b, (tap == TLO && kc == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
        });
    });
}

// GEMM 2 of one output row from the intermediate row at lb[kc] (pixel tile 0): group j reads intermediate tile j once and
// feeds output tiles j + 1 (tap 0), j (tap 1), j - 1 (tap 2).  epi(pt) runs one group after tile pt received its last K-step.
template <typename EPI>
__device__ __forceinline__ void rows_gemm2(const int (&lb)[4], const i32x4 (&w)[2][12], f32x4 (&acc)[8][2], EPI&& epi) {
    constexpr int GS = 4, NG = 8;
    i32x4 bb[2][GS];
    auto issue = [&](auto gi, i32x4* dst) {
        constexpr int j = decltype(gi)::value;
        static_for<0, GS>([&](auto e) { dst[decltype(e)::value] = lds_read_b128<j * 4096>(lb[decltype(e)::value]); });
    };
    issue(std::integral_constant<int, 0>{}, bb[0]);
    static_for<0, NG>([&](auto gi) {
        constexpr int j = decltype(gi)::value;
        i32x4* cb = bb[j & 1];
        if constexpr (j + 1 < NG) {
            issue(std::integral_constant<int, j + 1>{}, bb[(j + 1) & 1]);
            wait_lgkm<GS>();
        } else {
            wait_lgkm<0>();
        }
        static_for<0, GS>([&](auto e) { tie_reg(cb[decltype(e)::value]); });
        static_for<0, GS>([&](auto e) {
            constexpr int kc = decltype(e)::value;
            const i32x4 b = cb[kc];
            static_for<0, 3>([&](auto ti) {
                constexpr int tap = decltype(ti)::value, pt = j + 1 - tap;
                if constexpr (pt >= 0 && pt < 8) {
                    constexpr int ks = tap * 4 + kc;
                    constexpr bool first = kc == 0 && (pt == 0 ? tap == 1 : tap == 0);       // tile 0 has no tap 0: left of the row
#pragma unroll
                    for (int c = 0; c < 2; ++c) acc[pt][c] = mma_step<T>(w[c][ks], b, first ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
                }
            });
        });
        if constexpr (j >= 2) epi(std::integral_constant<int, j - 2>{});      // tile j - 2 was completed by group j - 1
    });
    epi(std::integral_constant<int, 6>{});
    epi(std::integral_constant<int, 7>{});
}
#endif

template <bool POST>
__global__ __launch_bounds__(512, 1)
void conv1d_pair_rows_kernel(const RowsParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PIXB = 256, NP = 4, CTW = 2, KS = 12;
    constexpr int ROWB = 128 * PIXB;                   // one class row in LDS
    constexpr int MID = 4 * ROWB;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave / NP, wn = wave % NP;
    const int q = lane >> 4, r16 = lane & 15;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res ? p.res : p.x, p.bytes);

    i32x4 w[CTW][KS];
    f32x4 sc[CTW], sh[CTW];
    {
        const i32x4* pw = (const i32x4*)(role ? p.wB : p.wA);
        const float* ps = role ? p.scaleB : p.scaleA;
        const float* pt = role ? p.shiftB : p.shiftA;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[c][ks] = pw[((wn * CTW + c) * KS + ks) * 64 + lane];
            const int co = (wn * CTW + c) * 16 + q * 4;
            sc[c] = ps ? *(const f32x4*)(ps + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[c] = pt ? *(const f32x4*)(pt + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tie_reg(w[c][ks]);
        asm volatile("" : "+v"(sc[c]), "+v"(sh[c]));
    }
    const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();
    const uint32_t rowstep = (uint32_t)(p.d * p.W) * PIXB;               // one class row down

    // The class loop is written once per role: with the role branch inside it the compiler keeps both roles' loop
    // invariants live in registers at once, which does not fit.
    if (role == 0) {
        // -------------------------------------------------------------------- producers
        for (int cls = blockIdx.x; cls < p.nclass; cls += gridDim.x) {
            const uint32_t n = fast_div((uint32_t)cls, p.fD);
            const int rho = cls - (int)n * p.d;
            const int nh = rho < p.H ? (int)fast_div((uint32_t)(p.H - rho + p.d - 1), p.fD) : 0;   // rows of this class (host: at most 4)
            const uint32_t base = (uint32_t)(((int)n * p.H + rho) * p.W) * PIXB;                 // row 0 of the class
            // DMA of the whole class, row by row: instruction k moves pixels (k & 7) * 16 .. + 15 of row k >> 3
            uint32_t in_rel[8];
            int wc = p.W;                              // opaque per class: keeps these eight out of the registers live across the rows
            asm volatile("" : "+s"(wc));
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int j = k * 16 + wn * 4 + q;                       // lane = (pixel q of this wave's four, 16-byte slot r16)
                in_rel[k] = j < wc ? (uint32_t)j * PIXB + swz<PIXB>(j, r16) * 16 : ERF_OOB;
            }
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int r = k >> 3;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (k * NP + wn) * 1024), 16,
                                                         r < nh ? in_rel[k & 7] : ERF_OOB, base + r * rowstep, 0, 0);
            }
            int lb1[4], mid_w[CTW];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) lb1[kc] = r16 * PIXB + swz<PIXB>(r16, kc * 4 + q) * 16;
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                const int chunk = (wn * CTW + c) * 2 + (q >> 1);
                mid_w[c] = MID + r16 * PIXB + swz<PIXB>(r16, chunk) * 16 + (q & 1) * 8;
            }
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");           // this wave's share of rows 0 and 1
            ERF_LDS_BARRIER();
            for (int t = 0; t <= nh; ++t) {
                f32x4 acc[8][CTW];
                if (t < nh) {
                    int lbt[3][4];
#pragma unroll
                    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
                        for (int kc = 0; kc < 4; ++kc) lbt[tap][kc] = lb1[kc] + (t - 1 + tap) * ROWB;
                    if (t == 0) rows_gemm1<1, 2>(lbt, w, acc);           // no row above the image
                    else if (t == 3) rows_gemm1<0, 1>(lbt, w, acc);      // LDS holds four rows; a fifth does not exist
                    else rows_gemm1<0, 2>(lbt, w, acc);                  // a row below the image was filled with zeros by the DMA
                    // row t + 2 is needed next: every workgroup streams its class in at launch start, so the rows land
                    // one after the other and the last one is not worth waiting for after the first row's matrix work
                    if (t == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                ERF_LDS_BARRIER();                                       // the consumers have read the intermediate row
                if (t < nh) {
                    int wl = p.W;
                    asm volatile("" : "+s"(wl));
#pragma unroll
                    for (int pt = 0; pt < 8; ++pt) {
                        const bool inside = r16 < wl - pt * 16;          // pixels right of the image are the 1x3 zero padding
#pragma unroll
                        for (int c = 0; c < CTW; ++c) {
                            f32x4 v = acc[pt][c];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e] * sc[c][e] + sh[c][e], 0.f) : 0.f;
                            *(i32x2*)(smem + mid_w[c] + pt * 4096) = pack_bf16x4(v);
                        }
                    }
                }
                ERF_LDS_BARRIER();                                       // intermediate row t complete
            }
        }
    } else {
        // -------------------------------------------------------------------- consumers
        for (int cls = blockIdx.x; cls < p.nclass; cls += gridDim.x) {
            const uint32_t n = fast_div((uint32_t)cls, p.fD);
            const int rho = cls - (int)n * p.d;
            const int nh = rho < p.H ? (int)fast_div((uint32_t)(p.H - rho + p.d - 1), p.fD) : 0;
            const uint32_t base = (uint32_t)(((int)n * p.H + rho) * p.W) * PIXB;
            int lb2[4];
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) lb2[kc] = MID + r16 * PIXB + swz<PIXB>(r16, kc * 4 + q) * 16;
            uint32_t orel[CTW];
#pragma unroll
            for (int c = 0; c < CTW; ++c) orel[c] = (uint32_t)r16 * PIXB + ((wn * CTW + c) * 16 + q * 4) * 2;
            ERF_LDS_BARRIER();
            for (int t = 0; t <= nh; ++t) {
                if (t >= 1) {
                    const uint32_t rowoff = base + (uint32_t)(t - 1) * rowstep;
                    int wl = p.W;                      // opaque per row: 16 loop-invariant masked offsets would not fit the registers
                    asm volatile("" : "+s"(wl));
                    i32x2 rv[8][CTW];
                    if constexpr (POST) {
#pragma unroll
                        for (int pt = 0; pt < 8; ++pt)
#pragma unroll
                            for (int c = 0; c < CTW; ++c)
                                rv[pt][c] = __builtin_bit_cast(i32x2, __builtin_amdgcn_raw_buffer_load_b64(
                                    rs_r, r16 < wl - pt * 16 ? orel[c] : ERF_OOB, rowoff + pt * 4096, 0));
                    }
                    f32x4 acc[8][CTW];
                    auto ep2 = [&](auto pti) {
                        constexpr int pt = decltype(pti)::value;
                        if constexpr (POST && pt == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the residual pieces
                        const bool inside = r16 < wl - pt * 16;
#pragma unroll
                        for (int c = 0; c < CTW; ++c) {
                            f32x4 v = acc[pt][c];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                v[e] = v[e] * sc[c][e] + sh[c][e];
                                if constexpr (!POST) v[e] = fmaxf(v[e], floor_v);
                            }
                            i32x2 o = pack_bf16x4(v);
                            if constexpr (POST) {      // rounding points of the staged epilogue: bf16, + residual, bf16
                                f32x4 f, r;
                                f[0] = __uint_as_float(((uint32_t)o[0]) << 16); f[1] = __uint_as_float(((uint32_t)o[0]) & 0xFFFF0000u);
                                f[2] = __uint_as_float(((uint32_t)o[1]) << 16); f[3] = __uint_as_float(((uint32_t)o[1]) & 0xFFFF0000u);
                                const i32x2 rr = rv[pt][c];
                                r[0] = __uint_as_float(((uint32_t)rr[0]) << 16); r[1] = __uint_as_float(((uint32_t)rr[0]) & 0xFFFF0000u);
                                r[2] = __uint_as_float(((uint32_t)rr[1]) << 16); r[3] = __uint_as_float(((uint32_t)rr[1]) & 0xFFFF0000u);
#pragma unroll
                                for (int e = 0; e < 4; ++e) f[e] = fmaxf(f[e] + r[e], floor_v);
                                o = pack_bf16x4(f);
                            }
                            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs_y, inside ? orel[c] : ERF_OOB,
                                                                  rowoff + pt * 4096, 0);
                        }
                        // the data registers of a buffer store with an SGPR offset must not be rewritten right after its issue
                        asm volatile("s_nop 4" ::: "memory");
                    };
                    rows_gemm2(lb2, w, acc, ep2);
                }
                ERF_LDS_BARRIER();                                       // the producers' two barriers of this row
                ERF_LDS_BARRIER();
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup
#endif
}

// Whole-row form: 128 channels, dilation 16 (a 1x3 tap = one 16-pixel tile), rows at most 128 pixels wide, at most 4 rows
// per residue class.
static bool pair_rows_fits(int H, int W, int d) { return d == 16 && W >= 1 && W <= 128 && (H + d - 1) / d <= 4; }

static int launch_pair_rows(const PairParams& q, hipStream_t stream) {
    RowsParams p{};
    p.x = q.x; p.y = q.y; p.wA = q.wA; p.wB = q.wB; p.scaleA = q.scaleA; p.shiftA = q.shiftA; p.scaleB = q.scaleB; p.shiftB = q.shiftB;
    p.res = q.res; p.d = q.d; p.relu = q.relu; p.N = q.N; p.H = q.H; p.W = q.W;
    if (!pair_rows_fits(p.H, p.W, p.d)) return ERF_EUNSUPPORTED;
    const long long bytes = (long long)p.N * p.H * p.W * 256, ncls = (long long)p.N * p.d;
    if (bytes >= 0x7FFFFFFFLL || ncls > 0x1FFFFFFF) return ERF_EUNSUPPORTED;      // 32-bit offsets, bit 31 marks masked lanes
    p.bytes = (uint32_t)bytes; p.nclass = (int)ncls; p.fD = make_fastdiv(p.d);
    constexpr int lds = 5 * 128 * 256;
    const int grid = 256 < p.nclass ? 256 : p.nclass;
    if (p.res) {
        auto kern = conv1d_pair_rows_kernel<true>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return ERF_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    } else {
        auto kern = conv1d_pair_rows_kernel<false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return ERF_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    }
    return erf_check_launch();
}

template <int C, int R, int TC>
static int launch_pair_pc(PairParams p, hipStream_t stream) {
    constexpr int PIXB = C * 2;
    constexpr size_t lds = 2 * (size_t)(R + 2) * TC * PIXB + 2 * (size_t)R * (TC + 2) * PIXB;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int nh = (p.H + p.d - 1) / p.d, nw = (p.W + p.d - 1) / p.d;
    p.CA = (nh + R - 1) / R;
    p.CB = nw <= 2 ? 1 : (nw - 2 + TC - 3) / (TC - 2);
    const long long nt = (long long)p.N * p.d * p.d * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA); p.fD = make_fastdiv(p.d);
    const long long bytes = (long long)p.N * p.H * p.W * PIXB, pad = (long long)p.d * p.W * PIXB;
    if (bytes + pad + (long long)(R + 2) * pad >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.bytes = (uint32_t)bytes; p.pad = (uint32_t)pad;
    const int grid = 256 < p.ntiles ? 256 : p.ntiles;
    if (p.res) {
        auto kern = conv1d_pair_pc_kernel<C, R, TC, true>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ERF_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    } else {
        auto kern = conv1d_pair_pc_kernel<C, R, TC, false>;
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return ERF_ELAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(512), lds, stream, p);
    }
    return erf_check_launch();
}

template <int C, int WM, int WN, int M, int TC, int KI, int OCC, bool POST>
static void launch_kernel(size_t lds, int grid, hipStream_t stream, const PairParams& p) {
    auto kern = conv1d_pair_kernel<C, WM, WN, M, TC, KI, OCC, POST>;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WM * WN * 64), lds, stream, p);
}

template <int C, int WM, int WN, int M, int TC, int KI>
static int launch_pair(PairParams p, hipStream_t stream) {
    constexpr int NW = WM * WN;
    constexpr int R = M / TC;
    constexpr int PIXB = C * 2;
    constexpr size_t lds = 2 * (size_t)KI * NW * 1024 + (size_t)R * (TC + 2) * PIXB;
    static_assert(lds <= 160 * 1024, "LDS budget");
    constexpr int OCC = (lds * 2 <= 160 * 1024 && NW <= 4) ? 2 : 1;
    const int nh = (p.H + p.d - 1) / p.d, nw = (p.W + p.d - 1) / p.d;      // lattice points per residue (max)
    p.CA = (nh + R - 1) / R;
    p.CB = nw <= 2 ? 1 : (nw - 2 + TC - 3) / (TC - 2);
    const long long nt = (long long)p.N * p.d * p.d * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA); p.fD = make_fastdiv(p.d);
    const long long bytes = (long long)p.N * p.H * p.W * PIXB, pad = (long long)p.d * p.W * PIXB;
    // 32-bit offsets: whole tensor plus one tile's reach below 2 GiB (bit 31 marks masked lanes)
    if (bytes + pad + (long long)(R + 2) * pad >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.bytes = (uint32_t)bytes; p.pad = (uint32_t)pad;
    int grid = 256 * OCC;
    if (grid > p.ntiles) grid = p.ntiles;
    if (p.res) launch_kernel<C, WM, WN, M, TC, KI, OCC, true>(lds, grid, stream, p);
    else launch_kernel<C, WM, WN, M, TC, KI, OCC, false>(lds, grid, stream, p);
    return erf_check_launch();
}

// Lattice-tile cost model: useful outputs / computed outputs for a TC-column tiling.
static double pair_utilization(int H, int W, int d, int R, int TC) {
    const int nh = (H + d - 1) / d, nw = (W + d - 1) / d;
    const int CA = (nh + R - 1) / R;
    const int CB = nw <= 2 ? 1 : (nw - 2 + TC - 3) / (TC - 2);
    return (double)nh * nw / ((double)CA * R * CB * TC);
}

int conv1d_pair_launch(const void* x, void* y, const void* packedA, const void* packedB, const float* scaleA, const float* shiftA,
                       const float* scaleB, const float* shiftB, const void* residual, int dtype, int N, int H, int W, int C,
                       int dilation, int relu, hipStream_t stream) {
    if (dtype != 1) return ERF_EUNSUPPORTED;
    if ((long long)N * H * W >= (1LL << 31)) return ERF_EUNSUPPORTED;
    PairParams p{};
    p.x = (const T*)x; p.y = (T*)y; p.wA = packedA; p.wB = packedB;
    p.scaleA = scaleA; p.shiftA = shiftA; p.scaleB = scaleB; p.shiftB = shiftB; p.res = (const T*)residual;
    p.d = dilation; p.relu = relu; p.N = N; p.H = H; p.W = W;
    static const int dbg = erf_debug_env("ERF_PAIR_DBG", 0);
    p.dbg = dbg;
    // Sparse lattices (few points per residue class) waste most of a tile: leave those to two single launches.
    if (C == 128) {
        const double u16 = pair_utilization(H, W, dilation, 8, 16), u32 = pair_utilization(H, W, dilation, 4, 32);
        if (u16 < 0.6 && u32 < 0.6) return pair_rows_fits(H, W, dilation) ? launch_pair_rows(p, stream) : ERF_EUNSUPPORTED;
        if (u32 > u16 * 1.1) return launch_pair<128, 1, 8, 128, 32, 6>(p, stream);
        static const int pc_off = erf_debug_env("ERF_PAIR_PC_OFF", 0);
        if (!pc_off) return launch_pair_pc<128, 8, 16>(p, stream);
        return launch_pair<128, 1, 8, 128, 16, 5>(p, stream);
    }
    if (C == 64) {
        if (pair_utilization(H, W, dilation, 8, 16) < 0.6) return ERF_EUNSUPPORTED;
        return launch_pair<64, 2, 2, 128, 16, 5>(p, stream);
    }
    if (C == 16) {
        if (pair_utilization(H, W, dilation, 32, 16) < 0.6) return ERF_EUNSUPPORTED;
        return launch_pair<16, 4, 1, 512, 16, 5>(p, stream);
    }
    return ERF_EUNSUPPORTED;
}
