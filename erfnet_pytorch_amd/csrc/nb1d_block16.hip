// A whole non_bottleneck_1d block in one launch for the 16-channel maps of the decoder (bf16 inference, dilation 1;
// train/erfnet.py:58-74 with both BatchNorms in eval mode folded into scale / shift):
//
//   a1 = relu(conv3x1(x) + t1);  a2 = relu(s2 * conv1x3(a1) + t2);  a3 = relu(conv3x1(a2) + t3);
//   y  = relu(s4 * conv1x3(a3) + t4 + x)
//
// At 16 channels a pixel is 32 bytes and a convolution is two matrix instructions per 16 pixels: the block is pure memory
// traffic, and run as two conv1d_pair launches it moves five maps (x, mid, mid, x again, y).  Here x is read once (its
// second use, the residual, is requested together with the tile and is an L2 hit) and y written once; a1, a2, a3 live in
// LDS only.
//
// Shape of the kernel.  One workgroup of four waves per tile of 8 rows x 64 columns, no tile loop: three workgroups fit a
// CU (49.6 KB of LDS each, <= 168 registers per lane), so one tile's load, another's arithmetic and a third's stores overlap
// without any software pipelining.  Everything is laid out in one frame of 12 rows x 68 columns (origin 2 rows / 2 columns
// above-left of the tile) with a row pitch of 68 pixels, so that every tap of every convolution is a constant shift of the
// linear pixel index: +0/+68/+136 for a 3x1 tap, +0/+1/+2 for a 1x3 tap reading a map stored one slot late.  Regions, as
// (frame rows) x 68 columns of consecutive pixels:
//   x  rows 0..11 (816 px)  -> a1 rows 1..10 (680) -> a2 rows 1..10 (columns 1..66 valid) -> a3 rows 2..9 (544)
//   -> y rows 2..9, columns 2..65.
// A shift that wraps around a row end only reaches frame columns that no later stage uses.  Intermediate pixels outside
// the image are forced to zero (they are the next convolution's zero padding).  Two LDS buffers: X holds x, then a2, then
// the staged output; A holds a1, then a3.
//
// Measured on MI355X (16 x 256 x 512 x 16): 44.6 us per launch against 37 + 33 us for the two conv1d_pair launches of the same
// block, bit-identical results.  What bounds it is LDS bandwidth, not HBM (134 MB in 44.6 us is 3 TB/s): per tile the four
// waves issue 320 fragment reads of 1 KB and 160 stores of 512 bytes (two-way bank conflicts: a lane's 8 bytes sit 32 bytes
// apart) -- about 4 k LDS cycles per tile and CU, against 1.3 k cycles of matrix instructions and 2.4 k of vector work.  A
// persistent variant (two workgroups per CU, the next frame DMA-ed during the current tile) measured the same 45 us, which
// is how the load latency was ruled out.
//
// K = 3 taps x 16 channels = 48 is padded to two K-steps of 32; lane (pixel r, quarter q) of K-step ks supplies the eight
// channels (gi & 1) * 8 .. + 7 of tap gi >> 1, gi = 4 ks + q, and zeros (read from a zeroed LDS slot) for gi >= 6 -- the fragment order of the packed
// weights (erf_conv1d_pack, mode 0) and of conv1d_pair, so the sums are bit-identical to the two-launch path.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"

typedef bf16raw T;

struct NbBlock16Params {
    const T* x;
    T* y;
    const void* w[4];
    const float* tab;        // [4 convolutions][scale | shift][16] fp32
    int N, H, W;
    int CA, CB;              // tiles per image along rows / columns
    FastDiv fCA, fCB;
    uint32_t bytes;
};

namespace nb16 {
constexpr int C = 16, PIXB = 32, KS = 2;
constexpr int R = 8, TW = 64, PITCH = TW + 4, XROWS = R + 4;
constexpr int NP1 = (R + 2) * PITCH;                 // 680 pixels of a1 / a2
constexpr int NP3 = R * PITCH;                       // 544 pixels of a3 / staged output
constexpr int NPT1 = (NP1 + 15) / 16;                // 43 pixel tiles
constexpr int NPT3 = NP3 / 16;                       // 34
constexpr int NW = 4, NT = NW * 64;
constexpr int KI = (XROWS * PITCH * 2 + NT - 1) / NT;       // 7 DMA slots of 16 bytes per lane
constexpr int XSLOTS = KI * NT / 2;                  // 896 pixel slots: the DMA plan, and >= 16 NPT1 + 2 PITCH
constexpr int XBYTES = XSLOTS * PIXB;                // 28672
constexpr int ASLOTS = NPT1 * 16 + 4;                // stored one slot late, read up to two slots past the last tile
constexpr int ABYTES = ASLOTS * PIXB;                // 22144
constexpr int ZSLOT = XBYTES + ABYTES;               // 16 bytes of zeros
constexpr int LDS_BYTES = ZSLOT + 16;                // 50832
static_assert(NP3 % 16 == 0 && XSLOTS >= NPT1 * 16 + 2 * PITCH && 3 * LDS_BYTES <= 160 * 1024, "frame geometry");
static_assert(((NPT1 + NW - 1) / NW) * NW * 16 * PIXB < 65536, "ds offset field");
}  // namespace nb16

#if defined(__HIP_DEVICE_COMPILE__)
// One convolution over NPT pixel tiles of 16 consecutive pixels, pixel tile it * 4 + wave in step `it` (fully unrolled: the
// LDS offsets of a step are immediates).  rd0 / rd1: this lane's fragment addresses for K-steps 0 / 1 at pixel tile `wave`;
// wr: address of its 8 output bytes (channels 4 q .. 4 q + 3 of pixel r16).  MASK: pixels outside the image become zero;
// m0 = this lane's pixel index in pixel tile `wave` (< 64, so row 0 of the region), (hb, wb) = image coordinates of the
// region's first pixel.
template <int NPT, bool RELU, bool MASK>
__device__ __forceinline__ void conv_phase16(char* smem, int rd0, int rd1, int st1, int wr, const i32x4& w0, const i32x4& w1, const f32x4& sc,
                                             const f32x4& sh, int wave, int m0, int hb, int wb, int H, int W) {
    using namespace nb16;
    constexpr int NIT = (NPT + NW - 1) / NW;
    constexpr int STEPB = NW * 16 * PIXB;
    constexpr int PF = 4;                                                  // fragments are requested PF steps ahead
    // The compiler keeps LDS reads and writes in program order (it cannot tell the two buffers apart), so the reads of step
    // it + PF are written in front of the store of step it; steps are taken in pairs so that two independent accumulation
    // chains alternate on the matrix pipe.
    const bool short_last = NIT * NW > NPT && (NIT - 1) * NW + wave >= NPT;      // wave-uniform: this wave has no last step
    const f32x2 sc01{sc[0], sc[1]}, sc23{sc[2], sc[3]}, sh01{sh[0], sh[1]}, sh23{sh[2], sh[3]};
    i32x4 b0[NIT], b1[NIT];
    auto fetch = [&](auto iti) {
        constexpr int it = decltype(iti)::value;
        b0[it] = *(const i32x4*)(smem + rd0 + it * STEPB);                 // (a short wave's last fetch reads a valid address; unused)
        b1[it] = *(const i32x4*)(smem + rd1);                              // K-step 1: tap 2 in lanes 0..31, the zero slot in the rest
        rd1 += st1;                                                        // (their stride is 0)
    };
    auto finish = [&](auto iti, const f32x4& acc) {
        constexpr int it = decltype(iti)::value;
        const f32x2 lo = f32x2{acc[0], acc[1]} * sc01 + sh01, hi = f32x2{acc[2], acc[3]} * sc23 + sh23;
        int o0 = pack_bf16x2(lo[0], lo[1]), o1 = pack_bf16x2(hi[0], hi[1]);
        if constexpr (RELU) {
            // bf16 is sign-magnitude: max with 0 as 16-bit integers is ReLU on the rounded values (rounding keeps the sign).
            // (Scalars on purpose: __builtin_bit_cast of a vector ELEMENT reads element 0 with this compiler.)
            o0 = relu_bf16x2(o0);
            o1 = relu_bf16x2(o1);
        }
        i32x2 o{o0, o1};
        if constexpr (MASK) {
            constexpr int a = (it * NW * 16) / PITCH, b = (it * NW * 16) % PITCH;      // this step is a rows + b pixels further on
            int c = m0 + b;
            const bool wrap = c >= PITCH;
            c = wrap ? c - PITCH : c;
            const int r = a + (wrap ? 1 : 0);
            const bool ok = (unsigned)(hb + r) < (unsigned)H && (unsigned)(wb + c) < (unsigned)W;
            o = ok ? o : i32x2{0, 0};
        }
        if (it == NIT - 1 && short_last) return;
        *(i32x2*)(smem + wr + it * STEPB) = o;
    };
    static_for<0, (PF < NIT ? PF : NIT)>(fetch);
    static_for<0, (NIT + 1) / 2>([&](auto pi) {
        constexpr int i0 = decltype(pi)::value * 2, i1 = i0 + 1;
        if constexpr (i0 + PF < NIT) fetch(std::integral_constant<int, i0 + PF>{});
        if constexpr (i1 + PF < NIT) fetch(std::integral_constant<int, i1 + PF>{});
        f32x4 acc0 = mma_step<T>(w0, b0[i0], f32x4{0.f, 0.f, 0.f, 0.f}), acc1;
        if constexpr (i1 < NIT) acc1 = mma_step<T>(w0, b0[i1], f32x4{0.f, 0.f, 0.f, 0.f});
        acc0 = mma_step<T>(w1, b1[i0], acc0);
        if constexpr (i1 < NIT) acc1 = mma_step<T>(w1, b1[i1], acc1);
        finish(std::integral_constant<int, i0>{}, acc0);
        if constexpr (i1 < NIT) finish(std::integral_constant<int, i1>{}, acc1);
    });
}
#endif

__global__ __launch_bounds__(256, 3)
void nb1d_block16_kernel(const NbBlock16Params p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace nb16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;

    // Workgroups are dealt to the 8 XCDs round-robin: give every XCD one contiguous eighth of the tile sequence (tile rows
    // fastest), so that the tiles sharing halo rows run on the same XCD at about the same time and share its L2.
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int t2 = fast_div(tile, p.fCA), trow = tile - t2 * p.CA;
    const int n = fast_div(t2, p.fCB), tcol = t2 - n * p.CB;
    const int h0 = trow * R, w0 = tcol * TW;
    const int H = p.H, W = p.W;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.x, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.bytes);

    // ---- the x frame: straight into LDS, zeros outside the image
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int L = id >> 1;
        const int row = L / PITCH, col = L - row * PITCH;
        const int gh = h0 - 2 + row, gw = w0 - 2 + col;
        const bool ok = (unsigned)gh < (unsigned)H && (unsigned)gw < (unsigned)W && row < XROWS;
        const uint32_t off = (uint32_t)(((n * H + gh) * W + gw) * PIXB + swz<PIXB>(L, id & 1) * 16);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + (k * NW + wave) * 1024), 16,
                                                 ok ? off : ERF_OOB, 0, 0, 0);
    }

    // ---- weights, scale / shift (this lane's channels 4 q .. 4 q + 3)
    i32x4 wf[4][KS];
    f32x4 sc[4], sh[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[s][ks] = ((const i32x4*)p.w[s])[ks * 64 + lane];
        sc[s] = *(const f32x4*)(p.tab + (s * 2 + 0) * C + q * 4);
        sh[s] = *(const f32x4*)(p.tab + (s * 2 + 1) * C + q * 4);
    }

    // ---- output chunks of this lane (4 of the tile's 1024) and their residual values
    constexpr int NS = R * TW * 2 / NT;
    uint32_t ooff[NS];
    int rb_lds[NS];
    u32x4 rv[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cid = tid + s * NT;
        const int o = cid >> 1, chunk = cid & 1;
        const int i = o / TW, j = o % TW;
        const int m3 = i * PITCH + j + 2;
        rb_lds[s] = m3 * PIXB + swz<PIXB>(m3, chunk) * 16;
        const bool ok = h0 + i < H && w0 + j < W;
        ooff[s] = ok ? (uint32_t)(((n * H + h0 + i) * W + w0 + j) * PIXB + chunk * 16) : ERF_OOB;
        rv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ooff[s], 0, 0);
    }

    // ---- per-lane LDS addresses (pixel tile `wave`)
    const int m0 = wave * 16 + r16;
    int rdX[KS], rdA[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int gi = ks * 4 + q;
        const int tap = gi / 2 < 2 ? gi / 2 : 2, ch = gi & 1;
        const int sx = m0 + tap * PITCH, sa = m0 + tap;
        rdX[ks] = gi < 6 ? sx * PIXB + swz<PIXB>(sx, ch) * 16 : ZSLOT;
        rdA[ks] = gi < 6 ? XBYTES + sa * PIXB + swz<PIXB>(sa, ch) * 16 : ZSLOT;
    }
    const int st1 = q < 2 ? NW * 16 * PIXB : 0;            // K-step 1 address stride per step
    if (tid == 0) *(i32x4*)(smem + ZSLOT) = i32x4{0, 0, 0, 0};
    const int wrA = XBYTES + (m0 + 1) * PIXB + swz<PIXB>(m0 + 1, q >> 1) * 16 + (q & 1) * 8;
    const int wrX = m0 * PIXB + swz<PIXB>(m0, q >> 1) * 16 + (q & 1) * 8;

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // Tiles whose frame lies inside the image need no zero padding of the intermediates.
    const bool border = h0 < 2 || h0 + R + 2 > H || w0 < 2 || w0 + TW + 2 > W;
    auto chain = [&](auto masked) {
        constexpr bool MASK = decltype(masked)::value;
        conv_phase16<NPT1, true, MASK>(smem, rdX[0], rdX[1], st1, wrA, wf[0][0], wf[0][1], sc[0], sh[0], wave, m0, h0 - 1, w0 - 2, H, W);
        __syncthreads();                   // a1 complete; x no longer read
        conv_phase16<NPT1, true, MASK>(smem, rdA[0], rdA[1], st1, wrX, wf[1][0], wf[1][1], sc[1], sh[1], wave, m0, h0 - 1, w0 - 2, H, W);
        __syncthreads();                   // a2 complete; a1 no longer read
        conv_phase16<NPT3, true, MASK>(smem, rdX[0], rdX[1], st1, wrA, wf[2][0], wf[2][1], sc[2], sh[2], wave, m0, h0, w0 - 2, H, W);
        __syncthreads();                   // a3 complete; a2 no longer read
        conv_phase16<NPT3, false, false>(smem, rdA[0], rdA[1], st1, wrX, wf[3][0], wf[3][1], sc[3], sh[3], wave, m0, 0, 0, H, W);
        __syncthreads();                   // staged output complete
    };
    if (border) chain(std::true_type{});
    else chain(std::false_type{});

    // ---- coalesced read-back, residual, ReLU, store
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const i32x4 ov = *(const i32x4*)(smem + rb_lds[s]);
        u32x4 out;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const uint32_t a = (uint32_t)ov[d], r = rv[s][d];
            const f32x2 sum = f32x2{__uint_as_float(a << 16), __uint_as_float(a & 0xFFFF0000u)} +
                              f32x2{__uint_as_float(r << 16), __uint_as_float(r & 0xFFFF0000u)};
            const int pk = pack_bf16x2(sum[0], sum[1]);
            out[d] = (uint32_t)relu_bf16x2(pk);
        }
        __builtin_amdgcn_raw_buffer_store_b128(out, rs_y, ooff[s], 0, 0);
    }
#endif
}

int nb1d_block16_launch(const void* x, void* y, const void* const* packed4, const float* tab, int dtype, int N, int H, int W, int C,
                        int dilation, hipStream_t stream) {
    using namespace nb16;
    if (dtype != 1 || C != nb16::C || dilation != 1) return ERF_EUNSUPPORTED;
    if (H < 1 || W < 1 || N < 1) return ERF_EINVAL;
    NbBlock16Params p{};
    p.x = (const T*)x; p.y = (T*)y;
    for (int i = 0; i < 4; ++i) p.w[i] = packed4[i];
    p.tab = tab; p.N = N; p.H = H; p.W = W;
    p.CA = (H + R - 1) / R;
    p.CB = (W + TW - 1) / TW;
    const long long nt = (long long)N * p.CA * p.CB;
    const long long bytes = (long long)N * H * W * PIXB;
    // 32-bit byte offsets with bit 31 marking masked lanes; the frame reaches two rows / columns past the image
    if (nt > 0x3FFFFFFF || bytes + (long long)(2 * W + 4) * PIXB >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.fCA = make_fastdiv(p.CA); p.fCB = make_fastdiv(p.CB);
    p.bytes = (uint32_t)bytes;
    hipLaunchKernelGGL(nb1d_block16_kernel, dim3((unsigned)nt), dim3(NT), LDS_BYTES, stream, p);
    return erf_check_launch();
}
