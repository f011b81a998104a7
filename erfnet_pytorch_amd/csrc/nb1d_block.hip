// A whole non_bottleneck_1d block in one launch (bf16 inference, 64 channels, dilation 1; train/erfnet.py:58-74 with both
// BatchNorms in eval mode folded into scale / shift):
//
//   a1 = relu(conv3x1(x) + t1);  a2 = relu(s2 * conv1x3(a1) + t2);  a3 = relu(conv3x1(a2) + t3);
//   y  = relu(s4 * conv1x3(a3) + t4 + x)
//
// x is read from HBM once (its second use, the residual, is an L2 hit a few microseconds later) and y written once; the
// three intermediate maps live in LDS only.
//
// Shape of the kernel.  One workgroup of FOUR waves per CU, one wave per SIMD, each wave owning ALL 64 output channels of
// its pixels.  That is what lets the weights of all four convolutions stay in registers for the whole launch: 4 sets x 4
// channel tiles x 6 K-steps x 16 bytes = 384 registers per lane, which a wave that has its SIMD's register file to itself
// can hold (256 VGPRs + 256 AGPRs; matrix instructions take their weight operand from either file).  A pixel fragment read
// from LDS then feeds four matrix instructions (one per channel tile) instead of one or two, which takes LDS read bandwidth
// off the critical path.
//
// Tile = 8 rows x 32 columns of output.  Everything is laid out in one frame of 12 rows x 36 columns (origin 2 rows / 2
// columns above-left of the tile) with a row pitch of 36 pixels, so that every tap of every convolution is a constant shift
// of the linear pixel index: +0/+36/+72 for a 3x1 tap, -1/0/+1 for a 1x3 tap (a shift that wraps around a row end only
// reaches frame columns that no later stage uses).  Regions, as (frame rows) x 36 columns of consecutive pixels:
//   x  rows 0..11 (432 px)  -> a1 rows 1..10 (360) -> a2 rows 1..10 (columns 1..34 valid) -> a3 rows 2..9 (288)
//   -> y rows 2..9, columns 2..33.
// Intermediate pixels outside the image are forced to zero (they are the next convolution's zero padding).
// LDS: two x buffers (the next tile is DMA-ed while this one is computed), one buffer for a1 / a3; a2 and the output
// staging reuse the current x buffer once the first convolution has consumed it.
//
// Measured on MI355X (16 x 128 x 256 x 64, in-kernel cycle stamps of a -DERF_DEBUG build, 2.1 GHz under this load): 121 us per
// launch against 2 x 40 us for the same block as two conv1d_pair launches -- bit-identical results, but SLOWER, so the
// module path keeps the pair kernels and this entry point is opt-in.  Per tile (8 per CU): 24.3 k cycles = 3.2 k skeleton
// (DMA issue, residual, read-back, stores, five barriers) + 8.4 k for the 528 matrix instructions at their 16-cycle issue
// rate + 5 k because the bare matrix loop runs at 25 rather than 16 cycles per instruction (phase prologues / tails and the
// waits of a lone wave) + 5.7 k for the epilogue slices and refills: a wave alone on its SIMD can hide about two vector
// instructions behind a 16x16x32 matrix instruction, and the epilogue needs 3.5.  On top come 11 us of launch prologue
// (every wave loads all 96 weight fragments) and the 1.28 x recomputation of halo pixels.  The two-launch path wins because
// eight waves per CU hide each other's epilogues.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"

typedef bf16raw T;

struct NbBlockParams {
    const T* x;
    T* y;
    const void* w[4];
    const float* tab;        // [4 convolutions][scale | shift][64] fp32
    int N, H, W;
    int CA, CB, ntiles;      // tiles per image along rows / columns
    FastDiv fCB, fCA;
    uint32_t bytes, pad;
    int dbg;                 // -DERF_DEBUG builds only (ERF_BLOCK_DBG): 1 no DMA, 2 no matrix phases, 4 no residual / stores, 8 one phase only
};

namespace nbb {
constexpr int C = 64, CH = 8, PIXB = 128, KS = 6, CT = 4;
constexpr int R = 8, TW = 32, PITCH = 36, XROWS = 12, XPIX = XROWS * PITCH;          // 432
constexpr int XBYTES = XPIX * PIXB;                                                  // 55296
constexpr int APIX = 23 * 16 + 2;                                                    // 23 pixel tiles + a pad slot at each end
constexpr int ABYTES = APIX * PIXB;                                                  // 47360
constexpr int XB0 = 0, AB = 2 * XBYTES, TAB = AB + ABYTES;
constexpr int TRASH = TAB + 4 * 2 * C * 4;                                           // 512 bytes nobody reads
constexpr int LDS_BYTES = TRASH + 512;                                               // 160512
constexpr int KI = 14;                                                               // 14 x 256 16-byte slots >= 432 x 8
constexpr int NW = 4;
}  // namespace nbb

#define NBB_F(SET, c, ks) ((SET) * 24 + (c) * KS + (ks))
#define NBB_AW(SET, c, ks) (NBB_F(SET, c, ks) < NAW)
#if defined(__HIP_DEVICE_COMPILE__)
// The matrix instruction as inline assembly, so that the register FILE of the weight operand is chosen here and not by the
// register allocator (left to itself it parks weights in accumulation registers and copies them back to vector registers in
// front of every use).  AW: the weight fragment lives in an accumulation register ("a") or a vector register ("v").
// ZERO: first K-step, the sum starts from the inline constant 0.
// Hazards the compiler cannot see inside asm: the pixel operand comes from an LDS read retired by s_waitcnt (no vector-ALU
// producer); an accumulator is read by vector instructions only after at least four later matrix instructions were issued
// (64+ cycles), or behind the explicit pause at the end of a phase.
template <bool AW, bool ZERO>
__device__ __forceinline__ void mfma_w(f32x4& acc, const i32x4& wfrag, const i32x4& b) {
    // pixels are the A operand (rows of the result), weights the B operand (columns): a lane ends up with ONE output channel
    // per channel tile and four pixels, so its scale / shift values are the same for every pixel tile of a phase
    if constexpr (AW) {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, 0" : "=&v"(acc) : "a"(wfrag), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+v"(acc) : "a"(wfrag), "v"(b));
    } else {
        if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, 0" : "=&v"(acc) : "v"(wfrag), "v"(b));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %1, %0" : "+v"(acc) : "v"(wfrag), "v"(b));
    }
}

// One convolution of the chain over NPT pixel tiles (16 consecutive pixels each): pixel tile r * 4 + wave in round r.
// in_lb[tap * 2 + cl]: per-lane LDS byte address of the pixel fragment of pixel tile `wave` (chunk class cl), tap `tap`.
// wf: flat weight array (fragment SET * 24 + c * 6 + ks; the first NAW fragments live in accumulation registers); column
// i of channel tile c is output channel 4 i + c, so that a lane (i = lane & 15, q = lane >> 4) holds, for each of its four
// pixels 4 q + j, the four ADJACENT channels 4 i .. 4 i + 3 (one per channel tile): 8 bytes of bf16 per pixel, and the 16
// lanes of a pixel write its 128 bytes in one piece.
// tq: per-lane LDS address of the scale / shift table entries of channels 4 i .. 4 i + 3 (read once per phase).
//
// Every LDS access of a phase is issued by hand so that the waits can be counted: with one wave per SIMD there is nobody
// to hide an LDS round trip behind, and a compiler-placed `lgkmcnt(0)` in front of each epilogue piece cost more than the
// matrix work itself.  The epilogue of a pixel tile (scale, shift, ReLU, pack, store) runs one round late, one of the
// lane's four pixels per K-step, cut into slices that sit between the matrix instructions of that step.  Per K-step ks:
//
//   wait (fragment ks has arrived) | matrix 0 | slice: fma, ReLU | matrix 1 | slice: pack, border mask, address
//   | matrix 2 | matrix 3 | refill of the fragment consumed one step ago | store of pixel ks (steps 0..3)
//
// LDS operations retire in order, so a wait names the number of operations issued after the youngest one it needs.
// Never issue a read whose result the code does not use afterwards: the compiler reuses the destination registers at once
// and the data, arriving later, overwrites whatever it put there.
//
// A read inside the rounds names its destination as an in/out operand, which pins the data to the registers the variable
// already lives in.  With a plain output the allocator is free to hand it registers that died a moment ago -- those of the
// fragment the matrix instructions just issued are still reading -- and the hardware does not interlock LDS data arriving
// in a register that an instruction in flight has yet to read.
//
// The rounds are a real loop (two rounds per trip, one per accumulator set): unrolled, the four convolutions come to ~60 KB.
template <int OFF>
__device__ __forceinline__ f32x4 lds_read_f4(int addr) {
    f32x4 r;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
    return r;
}
template <int OFF, typename V>
__device__ __forceinline__ void lds_read_into(V& dst, int addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(dst) : "v"(addr), "n"(OFF) : "memory");
}

// Where the outputs of a phase go and how they are masked.
struct PhaseOut {
    int abase[4];            // per-lane LDS byte address of the 8 bytes of pixel 4 q + j of pixel tile `wave` of round 0
    int trash;               // per-lane address inside the trash area (stores of pixel tiles past the region's end)
    int pix0;                // frame-linear index (row pitch 36) of this lane's pixel j = 0 in round 0
    int gh, gw, H, W;        // image coordinates of region row 0 / column 0, image size (border tiles only)
    bool border;
};

template <int SET, int NAW, bool FIRST, bool RELU>
__device__ __forceinline__ void chain_round(int r, int (&lb)[6], i32x4 (&bb)[nbb::KS], f32x4 (&acc)[nbb::CT], f32x4 (&prev)[nbb::CT],
                                            const f32x4& sc, const f32x4& sh, int (&keep)[4], const i32x4 (&wf)[96],
                                            const PhaseOut& po) {
    using namespace nbb;
    const int roff = (r - 1) * 8192;                          // the pixel tile finished in this round is the previous round's
    static_for<0, KS>([&](auto k) {
        constexpr int ks = decltype(k)::value;
        constexpr bool PIECE = !FIRST && ks < 4;              // this step carries the epilogue of pixel ks
        // operations in issue order -- first round: the six initial reads, then one refill per step from step 1 on; later
        // rounds: a refill per step and a store behind it in steps 0..3.  At least four operations are younger than the
        // refill a step depends on.
        constexpr int WAIT = FIRST ? (ks == 0 ? 5 : 4) : 4;
        wait_lgkm<WAIT>();
        tie_reg(bb[ks]);
        f32x4 v;
        i32x2 pk;
        int addr;
        mfma_w<NBB_AW(SET, 0, ks), ks == 0>(acc[0], wf[NBB_F(SET, 0, ks)], bb[ks]);
        if constexpr (PIECE) {
            asm volatile("" : "+v"(prev[0]), "+v"(prev[1]), "+v"(prev[2]), "+v"(prev[3]));
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                v[c] = __builtin_fmaf(prev[c][ks], sc[c], sh[c]);
                if (RELU) v[c] = fmaxf(v[c], 0.f);
            }
        }
        mfma_w<NBB_AW(SET, 1, ks), ks == 0>(acc[1], wf[NBB_F(SET, 1, ks)], bb[ks]);
        if constexpr (PIECE) {
            asm volatile("" : "+v"(v));
            pk = pack_bf16x4(v);
            pk[0] &= keep[ks]; pk[1] &= keep[ks];              // all ones except on pixels of a border tile outside the image
            addr = po.abase[ks] + roff;
        }
        mfma_w<NBB_AW(SET, 2, ks), ks == 0>(acc[2], wf[NBB_F(SET, 2, ks)], bb[ks]);
        if constexpr (PIECE) asm volatile("" : "+v"(pk), "+v"(addr));
        mfma_w<NBB_AW(SET, 3, ks), ks == 0>(acc[3], wf[NBB_F(SET, 3, ks)], bb[ks]);
        // Refill the registers of the fragment consumed ONE STEP AGO: the four matrix instructions that read it were issued a
        // whole step (64 cycles of matrix pipe) before the four just issued, so they have certainly taken their operands by
        // the time the LDS data comes back.  Each fragment address then moves on one round (64 pixels).
        if constexpr (ks == 0) {
            if constexpr (!FIRST) lds_read_into<0>(bb[KS - 1], lb[KS - 1]);           // this round's last fragment
            asm volatile("v_add_u32 %0, 0x2000, %0" : "+v"(lb[KS - 1]));
        } else {
            lds_read_into<8192>(bb[ks - 1], lb[ks - 1]);                               // next round's
            asm volatile("v_add_u32 %0, 0x2000, %0" : "+v"(lb[ks - 1]));
        }
        if constexpr (PIECE) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pk) : "memory");
        // steps 4 and 5 carry no epilogue piece: step 4 prepares the border masks of THIS round's pixels (used next round)
        if constexpr (ks == 4) {
            if (po.border) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int pidx = po.pix0 + j + r * 64;
                    const int row = (pidx * 1821) >> 16, col = pidx - row * PITCH;
                    keep[j] = ((unsigned)(po.gh + row) < (unsigned)po.H && (unsigned)(po.gw + col) < (unsigned)po.W) ? -1 : 0;
                }
            }
        }
    });
}

template <int NPT, int SET, int NAW, bool RELU>
__device__ __forceinline__ void chain_phase(int wave, const int (&in_lb)[6], const i32x4 (&wf)[96], int tq, const PhaseOut& po) {
    using namespace nbb;
    constexpr int ROUNDS = (NPT + NW - 1) / NW;
    static_assert(ROUNDS >= 3, "round 0 and the two-round loop assume three rounds or more");
    i32x4 bb[KS];
    f32x4 acc0[CT], acc1[CT];
    int lb[6], keep[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int i = 0; i < 6; ++i) lb[i] = in_lb[i];
    const f32x4 sc = lds_read_f4<SET * 512>(tq), sh = lds_read_f4<SET * 512 + 256>(tq);
    static_for<0, KS>([&](auto k) {
        constexpr int ks = decltype(k)::value;
        bb[ks] = lds_read_b128<0>(lb[ks]);
    });
    // round 0, then pairs of rounds in a real loop, then an odd one; round r accumulates into set r & 1 and finishes the pixel
    // tile of round r - 1 (which always exists: only the last round has waves without a pixel tile)
    chain_round<SET, NAW, true, RELU>(0, lb, bb, acc0, acc1, sc, sh, keep, wf, po);
#pragma nounroll
    for (int r = 1; r + 1 < ROUNDS; r += 2) {
        chain_round<SET, NAW, false, RELU>(r, lb, bb, acc1, acc0, sc, sh, keep, wf, po);
        chain_round<SET, NAW, false, RELU>(r + 1, lb, bb, acc0, acc1, sc, sh, keep, wf, po);
    }
    if constexpr (!(ROUNDS & 1)) chain_round<SET, NAW, false, RELU>(ROUNDS - 1, lb, bb, acc1, acc0, sc, sh, keep, wf, po);
    // Tail: the last pixel tile's epilogue.  The pause lets the last matrix instructions write their accumulators.
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    // The last round refilled its fragments for a round that does not exist (reads past the region: harmless).  Their data
    // has arrived now; naming the registers here keeps the compiler from reusing them while it was still on its way.
#pragma unroll
    for (int i = 0; i < KS; ++i) asm volatile("" ::"v"(bb[i]));
    const bool live = (ROUNDS - 1) * NW + wave < NPT;
    f32x4 (&last)[CT] = (ROUNDS & 1) ? acc0 : acc1;
    asm volatile("" : "+v"(last[0]), "+v"(last[1]), "+v"(last[2]), "+v"(last[3]));
    static_for<0, 4>([&](auto ji) {
        constexpr int j = decltype(ji)::value;
        f32x4 v;
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            v[c] = __builtin_fmaf(last[c][j], sc[c], sh[c]);
            if (RELU) v[c] = fmaxf(v[c], 0.f);
        }
        i32x2 pk = pack_bf16x4(v);
        pk[0] &= keep[j]; pk[1] &= keep[j];
        const int addr = live ? po.abase[j] + (ROUNDS - 1) * 8192 : po.trash;
        asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pk) : "memory");
    });
}
#endif

__global__ __launch_bounds__(256, 1)
void nb1d_block_kernel(const NbBlockParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace nbb;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4, r16 = lane & 15;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, p.bytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.x, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.bytes);

    // ---- weights of the four convolutions: registers for the whole launch.  Fragment f = set * 24 + c * 6 + ks; the first NAW
    // (64 x 4 = all 256 accumulation registers) are pinned there, the other 32 fragments stay in vector registers.
    constexpr int NAW = 64;
    i32x4 wf[96];
#pragma unroll
    for (int f = 0; f < 96; ++f) {
        // column i of channel tile c = output channel 4 i + c = row 4 (i & 3) + c of packed channel tile i >> 2
        const int c = (f % 24) / KS, ks = f % KS;
        wf[f] = ((const i32x4*)p.w[f / 24])[((r16 >> 2) * KS + ks) * 64 + q * 16 + 4 * (r16 & 3) + c];
    }
#pragma unroll
    for (int f = 0; f < 96; ++f) {
        if (f < NAW) asm volatile("" : "+a"(wf[f]));
        else asm volatile("" : "+v"(wf[f]));
    }
    // scale / shift table -> LDS (read back per channel tile in the epilogues)
    for (int i = tid; i < 4 * 2 * C; i += 256) *(float*)(smem + TAB + i * 4) = p.tab[i];
    // pad slots of the a1 / a3 buffer (read by the first / last 1x3 taps, never written)
    if (tid < 16) {
        *(i32x4*)(smem + AB + (tid & 7) * 16) = i32x4{0, 0, 0, 0};
    }

    // ---- launch-invariant per-lane addressing
    // B fragments: pixel slot = 16 * (4 * round + wave) + r16 + shift; its swizzle depends on (r16 + shift) & 7 only
    int lbH[6], lbW[6];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int cl = 0; cl < 2; ++cl) {
            const int sh = wave * 16 + r16 + tap * PITCH;          // 3x1: input slot = output index + tap * pitch
            lbH[tap * 2 + cl] = sh * PIXB + (((cl * 4 + q) ^ (sh & 7)) << 4);
            const int sw = wave * 16 + r16 + tap;                  // 1x3: input stored one slot late, slot = output index + tap
            lbW[tap * 2 + cl] = sw * PIXB + (((cl * 4 + q) ^ (sw & 7)) << 4);
        }
    // accumulator stores: lane holds channels 4 r16 .. + 3 of pixels 4 q + j -> 8 bytes: half r16 & 1 of chunk r16 >> 1
    const int slot0 = wave * 16 + 4 * q;                            // + j + 64 * round (+ 1 for the late-by-one buffers)
    const int tab_q = TAB + r16 * 16;                               // + conv * 512 (+ 256 for shift)
#pragma unroll
    for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(lbH[i]), "+v"(lbW[i]));

    // ---- tiles: every XCD walks one contiguous eighth of the sequence (neighbouring tiles share halo rows in its L2)
    int tile = blockIdx.x, tile_end = p.ntiles, G = gridDim.x;
    if ((G & 7) == 0) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3);
        tile_end = (xcd + 1) * per < p.ntiles ? (xcd + 1) * per : p.ntiles;
        G >>= 3;
    }
    struct Geo { int n, h0, w0; uint32_t soff; bool live; };
    auto geo = [&](int t) {
        Geo g;
        g.live = t < tile_end;
        uint32_t u = g.live ? (uint32_t)t : 0u;
        uint32_t nq = fast_div(u, p.fCB);
        const int tw = (int)(u - nq * p.fCB.div); u = nq;
        nq = fast_div(u, p.fCA);
        const int th = (int)(u - nq * p.fCA.div);
        g.n = (int)nq; g.h0 = th * R; g.w0 = tw * TW;
        g.soff = (uint32_t)((g.n * p.H + g.h0) * p.W + g.w0) * PIXB;           // byte offset of the tile's first output pixel
        return g;
    };
    // x tile -> LDS: slot id = k * 256 + tid -> frame pixel id / 8, chunk id % 8; pixels outside the image read zeros
    auto dma = [&](const Geo& g, int boff) {
        int t8 = tid >> 3, chunk = tid & 7;                  // opaque: the per-instruction offsets are recomputed, not kept (14 registers)
        asm volatile("" : "+v"(t8), "+v"(chunk));
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            if (k == KI - 1 && wave >= 2) break;           // 432 x 8 slots = 13.5 x 256: the last instruction is waves 0 and 1 only
            const int pix = k * 32 + t8;
            const int row = (pix * 1821) >> 16, col = pix - row * PITCH;
            const int h = g.h0 - 2 + row, wc = g.w0 - 2 + col;
            const bool ok = g.live && pix < XPIX && (unsigned)h < (unsigned)p.H && (unsigned)wc < (unsigned)p.W;
            const uint32_t rel = (uint32_t)(row * p.W + col) * PIXB + (uint32_t)((chunk ^ (pix & 7)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + boff + (k * NW + wave) * 1024),
                                                     16, ok ? rel : ERF_OOB, g.soff, 0, 0);
        }
    };

    Geo g = geo(tile), g1 = geo(tile + G);
    dma(g, XB0);
    dma(g1, XBYTES);
    if (wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI - 1) : "memory");
    __syncthreads();

#ifdef ERF_DEBUG
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    int cur = 0;
    for (; tile < tile_end; tile += G, cur ^= 1) {
        const int xb = cur * XBYTES;
        const bool border = g.h0 < 2 || g.h0 + R + 2 > p.H || g.w0 < 2 || g.w0 + TW + 2 > p.W;
        const int gh = g.h0 - 2, gw = g.w0 - 2;
        // Everything the epilogues derive from these three is tile-invariant; left visible as such, the compiler hoists ~90
        // store addresses out of the tile loop and spills them.  Opaque copies keep the (cheap) arithmetic inside the loop.
        int s0 = slot0, tq = tab_q, ii = r16;
        asm volatile("" : "+v"(s0), "+v"(tq), "+v"(ii));

        // where a phase's outputs go: slot = 16 * (4 * round + wave) + 4 q + j (+ 1 in the late-by-one buffer), pixel-major, swizzled
        auto phase_out = [&](int obase, int late, int row0) {
            PhaseOut po;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int slot = s0 + j + late;
                po.abase[j] = obase + slot * PIXB + (ii & 1) * 8 + (((ii >> 1) ^ (slot & 7)) << 4);
            }
            po.trash = TRASH + lane * 8;
            po.pix0 = s0;
            po.gh = gh + row0; po.gw = gw; po.H = p.H; po.W = p.W; po.border = border;
            return po;
        };

        int lb[6];
        // ---- conv 1 (3x1): x -> a1 (frame rows 1..10), 23 pixel tiles, stored one slot late in the a buffer
#pragma unroll
        for (int i = 0; i < 6; ++i) lb[i] = lbH[i] + xb;
        if (!(p.dbg & 2)) chain_phase<23, 0, NAW, true>(wave, lb, wf, tq, phase_out(AB, 1, 1));
        ERF_LDS_BARRIER();
        // ---- conv 2 (1x3): a1 -> a2 (frame rows 1..10) in the consumed x buffer
#pragma unroll
        for (int i = 0; i < 6; ++i) lb[i] = lbW[i] + AB;
        if (!(p.dbg & 10)) chain_phase<23, 1, NAW, true>(wave, lb, wf, tq, phase_out(xb, 0, 1));
        ERF_LDS_BARRIER();
        // ---- conv 3 (3x1): a2 -> a3 (frame rows 2..9), 18 pixel tiles, one slot late in the a buffer
#pragma unroll
        for (int i = 0; i < 6; ++i) lb[i] = lbH[i] + xb;
        if (!(p.dbg & 10)) chain_phase<18, 2, NAW, true>(wave, lb, wf, tq, phase_out(AB, 1, 2));
        ERF_LDS_BARRIER();
        // ---- conv 4 (1x3): a3 -> s4 * conv + t4 (bf16, frame rows 2..9) staged in the x buffer
#pragma unroll
        for (int i = 0; i < 6; ++i) lb[i] = lbW[i] + AB;
        if (!(p.dbg & 10)) chain_phase<18, 3, NAW, false>(wave, lb, wf, tq, phase_out(xb, 0, 2));
        // the residual: 8 rows x 32 columns x 8 chunks = 8 chunks per lane (row s, column tid >> 3, chunk tid & 7); requested
        // before the barrier so that part of the L2 round trip passes while the other waves finish
        int j = tid >> 3, chunk = tid & 7;
        asm volatile("" : "+v"(j), "+v"(chunk));
        uint32_t ooff[R];
        u32x4 rv[R];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const bool ok = g.h0 + s < p.H && g.w0 + j < p.W && !(p.dbg & 4);
            ooff[s] = ok ? (uint32_t)(s * p.W + j) * PIXB + chunk * 16 : ERF_OOB;
        }
#pragma unroll
        for (int s = 0; s < R; ++s) rv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[s], g.soff, 0);
        ERF_LDS_BARRIER();

        // ---- read-back: + residual, ReLU, coalesced 16-byte stores
        i32x4 outv[R];
#pragma unroll
        for (int s = 0; s < R; ++s) {
            const int slot = s * PITCH + j + 2;
            outv[s] = *(const i32x4*)(smem + xb + slot * PIXB + ((chunk ^ (slot & 7)) << 4));
        }
#pragma unroll
        for (int s = 0; s < R; ++s) {
            float f[8], rr[8];
            ChunkOps<T>::unpack(outv[s], f);
            ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, rv[s]), rr);
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e] + rr[e], 0.f);
            outv[s] = ChunkOps<T>::pack(f);
        }
        // (pinned here: left to the scheduler, the sums sink below the requests for tile k + 2, and their wait for the
        // residual -- memory operations retire in order -- becomes a wait for that whole tile)
#pragma unroll
        for (int s = 0; s < R; ++s) asm volatile("" : "+v"(outv[s]));
        ERF_LDS_BARRIER();                 // staging consumed by every wave: the buffer may take tile k + 2

        const uint32_t out_soff = g.soff;
        g = g1;
        g1 = geo(tile + 2 * G);
        if (!(p.dbg & 1)) dma(g1, xb);
#pragma unroll
        for (int s = 0; s < R; ++s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[s]), rs_y, ooff[s], out_soff, 0);
        // a 16-byte buffer store with an SGPR offset must not have its data registers rewritten right behind it
        asm volatile("s_nop 7" ::: "memory");
        // tile k + 1 was requested a whole tile ago and every wave has waited for its share (the wait for the residual, which
        // is younger, retires it) before the barrier above: the next iteration may read it, and write the a buffer, at once
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef ERF_DEBUG
    if ((p.dbg & 256) && tid == 0) {       // diagnostic build only: shader cycles and 100 MHz ticks of this workgroup's tile loop
        uint64_t* o = (uint64_t*)p.y + blockIdx.x * 2;
        o[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        o[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
#endif
}

int nb1d_block_launch(const void* x, void* y, const void* const* packed4, const float* tab, int dtype, int N, int H, int W, int C,
                      int dilation, hipStream_t stream) {
    using namespace nbb;
    if (dtype != 1 || C != nbb::C || dilation != 1) return ERF_EUNSUPPORTED;
    if (H < 1 || W < 1) return ERF_EINVAL;
    NbBlockParams p{};
    p.x = (const T*)x; p.y = (T*)y;
    for (int i = 0; i < 4; ++i) p.w[i] = packed4[i];
    p.tab = tab; p.N = N; p.H = H; p.W = W;
    p.CA = (H + R - 1) / R;
    p.CB = (W + TW - 1) / TW;
    const long long nt = (long long)N * p.CA * p.CB;
    const long long bytes = (long long)N * H * W * PIXB, pad = ((long long)2 * W + 2) * PIXB;
    if (nt > 0x1FFFFFFF || bytes + pad + (long long)(XROWS + 1) * W * PIXB >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA);
    p.bytes = (uint32_t)bytes; p.pad = (uint32_t)pad;
    p.dbg = erf_debug_env("ERF_BLOCK_DBG", 0);
    int grid = 256;
    if (grid > p.ntiles) grid = p.ntiles;
    if (hipFuncSetAttribute((const void*)nb1d_block_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(nb1d_block_kernel, dim3(grid), dim3(256), LDS_BYTES, stream, p);
    return erf_check_launch();
}
