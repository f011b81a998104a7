// A whole non_bottleneck_1d block in one launch for 64 channels, as a STREAM of image rows through four wave roles
// (bf16 inference, dilation 1; train/erfnet.py:58-74 with both BatchNorms in eval mode folded into scale / shift):
//
//   a1 = relu(conv3x1(x) + t1);  a2 = relu(s2 * conv1x3(a1) + t2);  a3 = relu(conv3x1(a2) + t3);
//   y  = relu(s4 * conv1x3(a3) + t4 + x)
//
// Run as two conv1d_pair launches the block moves five maps through HBM (x, a2 | a2, x again, y) and its waves walk two
// matrix phases and four barriers per tile in lockstep.  Here x is read once and y written once, a1 / a2 / a3 live in LDS
// only, and the eight waves of the one workgroup a CU holds are split by ROLE instead of by output channel:
//
//   waves 0,1: conv3x1_1   waves 2,3: conv1x3_1   waves 4,5: conv1x3_2   waves 6,7: conv3x1_2
//   (waves w and w + 4 share a SIMD: the pairing puts the role with the most vector work beside the one with the least)
//
// each wave keeping ALL 64 output channels of its convolution in registers (96 registers of weights) for two of the four
// 16-pixel matrix tiles of a row, so every 1 KB pixel-fragment read feeds four matrix instructions.
//
// The unit of work is one row of a column STRIP: a frame of 64 columns (local column c = image column - c0 + 2) of which
// 60 are output columns (c in [2, 62)); the two columns either side are the halo the two 1x3 convolutions need and are
// recomputed by the neighbouring strip.  When an image's last strip has at most 28 output columns it is a NARROW strip: a
// frame of 32 columns, one matrix tile per wave instead of two (W = 256: four wide strips and a narrow one).  A workgroup marches down its strip one row per step, all roles working at once on
// different rows -- the vertical halo of the two 3x1 convolutions costs nothing, their three taps are simply the three
// youngest rows of a ring buffer:
//
//   step t:  DMA       x row t  requested -> X[t % 9]; it is first read D + 1 = 2 steps later
//            conv3x1_1 a1 row u = t-2-D  from X[u-1], X[u], X[u+1]       -> A1[u % 2]
//            conv1x3_1 a2 row u = t-3-D  from A1[u % 2]                   -> A2[u % 4]   (zeros for rows outside the image)
//            conv3x1_2 a3 row u = t-5-D  from A2[u-1], A2[u], A2[u+1]     -> A3[u % 2]
//            conv1x3_2 y' row u = t-6-D  from A3[u % 2]                   -> Y[u % 2]    (bf16(s4 * acc + t4), staged)
//            finish    y  row u = t-7-D  = relu(Y[u % 2] + X[u % 9])      -> HBM as whole 16-byte chunks (x stays in its ring
//                                            row until then: x is read from HBM once)
//   one workgroup barrier per step.  The x requests are issued by the four 3x1 waves (2 KB each), the rows are finished by the
//   four 1x3 waves (16 pixels each) at the start of their step: the 3x1 waves' matrix instructions start at once and cover that
//   vector work, the 1x3 waves' matrix instructions then cover the 3x1 waves' epilogues (every wave finishing 8 pixels measured
//   5 % slower: then all eight delay their matrix instructions together).
//
// Every row of every ring has the same layout: 66 pixel slots of 128 bytes (slot = local column + 1; slots 0 and 65 stay
// zero and only ever feed the frame's outermost columns, which no output column depends on), the eight 16-byte chunks of a
// pixel XOR-swizzled by the slot index so that a fragment read (16 consecutive pixels, one chunk per quarter wave) is free
// of bank conflicts.  19 rows = 157 KB of LDS, plus the table of the workgroup's rows.
//
// A workgroup's rows are a contiguous run of the linearised (image, strip, row) space (the runs are cut by the host so that
// every workgroup has about the same number of steps), cut where it crosses into the next strip; each piece ("chunk") of RC output rows takes RC + 4 steps (rows r0-2 .. r0+RC+1 are loaded, a1 / a2 are formed
// on RC + 2 of them), and the roles simply keep stepping from one chunk into the next, so the pipeline fills and drains
// once per workgroup, not once per chunk.  Intermediate values outside the image are forced to zero where the next
// convolution pads with zeros: a1 and a3 in columns outside the image, a2 in rows outside the image; x outside the image
// reads as zero through the buffer descriptor.
//
// Sums: every accumulator receives its six K-steps in the order of conv1d_pair_kernel (taps ascending, channel chunks
// ascending), with the same matrix instruction, and the epilogues round where that kernel's do -- the output is
// bit-identical to the two-launch path.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"

typedef bf16raw T;

#ifndef NBS_MIN_ROWS
#define NBS_MIN_ROWS 40          // fewest rows of a workgroup's run (see the launcher)
#endif

namespace nbs {
constexpr int C = 64, PIXB = 128, KS = 6, CT = 4;
constexpr int FW = 64, TW = 60;                     // frame / output columns of a strip
constexpr int ROWB = (FW + 2) * PIXB;               // 8448 bytes per ring row
constexpr int NX = 9, NA1 = 2, NA2 = 4, NA3 = 2, NY = 2;
constexpr int OFF_X = 0;
constexpr int OFF_A1 = OFF_X + NX * ROWB;
constexpr int OFF_A2 = OFF_A1 + NA1 * ROWB;
constexpr int OFF_A3 = OFF_A2 + NA2 * ROWB;
constexpr int OFF_Y = OFF_A3 + NA3 * ROWB;
constexpr int OFF_TAB = OFF_Y + NY * ROWB;          // 152064: the slot table, 16 bytes per slot
constexpr int MAX_SLOTS = 200;
constexpr int LDS_BYTES = OFF_TAB + MAX_SLOTS * 16; // 19 rows + the table = 163712
constexpr int D = 1;                                // steps between the request of an x row and its first use
constexpr int LAG1 = 2 + D, LAG2 = 3 + D, LAG3 = 5 + D, LAG4 = 6 + D, LAGY = 7 + D;
static_assert(NX == LAGY + 1, "an x row lives from its request to its use as the residual of the finished row");
constexpr int PAD = 2 * PIXB;                       // the descriptors start two pixels before the tensors (column c0 - 2 of strip 0)
constexpr int NTHREADS = 512;
constexpr int FWN = 32, TWN = 28;                   // frame / output columns of a narrow strip
constexpr int MAX_WGS = 256;
static_assert(LDS_BYTES <= 160 * 1024 && LDS_BYTES % 16 == 0, "ring buffers fit the CU's LDS");
}  // namespace nbs

struct NbStreamParams {
    const T* x;
    T* y;
    const void* w[4];
    const float* tab;        // [4 convolutions][scale | shift][64] fp32
    int N, H, W, S;          // S = strips per image
    int narrow_last;         // the last strip of an image is a narrow one (32-column frame, 28 output columns)
    uint32_t bytes;          // N*H*W*128
    int l0[nbs::MAX_WGS + 1];     // workgroup b runs rows [l0[b], l0[b + 1]) of the linearised (image, strip, row) space
    int dbg;                 // -DERF_DEBUG builds only (ERF_BLOCK_DBG): 1 no matrix instructions, 2 no epilogue, 4 no row finishing,
                             // 8 no x requests, 16 no fragment reads, 256 cycle stamps
};

#ifdef NBS_ABLATE
// ablation builds (-DNBS_ABLATE=mask, same bits as ERF_BLOCK_DBG): parts of a step removed at COMPILE time, so that what is left
// runs exactly as in the product build (the run-time switches of the debug build cost a step more than they remove)
#define NBS_DBG(bit) ((NBS_ABLATE) & (bit))
#define NBS_STAMP(i)
#elif defined(ERF_DEBUG)
// cycle stamps of a diagnostic build: [workgroup][wave][section], read back by erf_debug_block64s_stamps; nothing else reads them
static __device__ unsigned long long g_nbs_stamps[256 * 8 * 8];
#define NBS_DBG(bit) (p.dbg & (bit))
#define NBS_STAMP(i) if (p.dbg & 256) { const unsigned long long tn = __builtin_amdgcn_s_memtime(); st[i] += tn - tprev; tprev = tn; }
#else
#define NBS_DBG(bit) 0
#define NBS_STAMP(i)
#endif

#if defined(__HIP_DEVICE_COMPILE__)
namespace nbs {

// What a role needs to know about the row ("slot") it works on in a step; wave-uniform.
struct Slot {
    int soff;    // byte offset of (image, row, column c0) from the tensor = of column c0 - 2 from the descriptor base
    int c0;      // first output column of the strip
    int fl;      // 1: slot exists, 2: a1 / a2 are formed on this row, 4: a3 / y are formed on this row, 8: row inside the image,
                 // 16: narrow strip
    __device__ __forceinline__ int fw() const { return (fl & 16) ? FWN : FW; }
    __device__ __forceinline__ int tw() const { return (fl & 16) ? TWN : TW; }
};

template <int ROLE>
__device__ __forceinline__ void role_loop(const NbStreamParams& p, char* smem, const int wave, const int lane) {
    constexpr bool VERT = ROLE == 0 || ROLE == 2;
    constexpr int LAG = ROLE == 0 ? LAG1 : ROLE == 1 ? LAG2 : ROLE == 2 ? LAG3 : LAG4;
    const int half = wave & 1;
    const int q = lane >> 4, r16 = lane & 15;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - PAD, p.bytes + PAD);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc((char*)p.y - PAD, p.bytes + PAD);

    // ---- this role's weights (all 64 output channels) and epilogue constants
    i32x4 w[CT][KS];
    f32x4 sc[CT], sh[CT];
    {
        const i32x4* pw = (const i32x4*)p.w[ROLE];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) w[ct][ks] = pw[(ct * KS + ks) * 64 + lane];
            sc[ct] = *(const f32x4*)(p.tab + (2 * ROLE) * C + ct * 16 + q * 4);
            sh[ct] = *(const f32x4*)(p.tab + (2 * ROLE + 1) * C + ct * 16 + q * 4);
        }
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tie_reg(w[ct][ks]);
        asm volatile("" : "+v"(sc[ct]), "+v"(sh[ct]));
    }

    // ---- launch-invariant per-lane addressing (byte offsets inside a ring row)
    const int p0 = 1 + 32 * half + r16;                  // slot of this lane's pixel in matrix tile 0 of this wave (tile 1: + 16 slots)
    constexpr int NFO = VERT ? 2 : 6;
    int fo[NFO];                                          // fragment reads: VERT [chunk half kc], else [tap][kc]
    if constexpr (VERT) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) fo[kc] = p0 * PIXB + (((4 * kc + q) ^ (p0 & 7)) * 16);
    } else {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                const int pp = p0 + tap - 1;
                fo[tap * 2 + kc] = pp * PIXB + (((4 * kc + q) ^ (pp & 7)) * 16);
            }
    }
    int wo[CT];                                           // accumulator stores: channels 16 ct + 4 q .. + 3 of pixel p0
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) wo[ct] = p0 * PIXB + (((ct * 2 + (q >> 1)) ^ (p0 & 7)) * 16) + (q & 1) * 8;
    // this wave's share of a row's memory traffic (3x1 waves: the x request; 1x3 waves: finishing the output): local columns
    // 16 duty .. 16 duty + 15, as two wave-wide accesses of 8 pixels, one 16-byte chunk per lane
    const int duty = 2 * (ROLE >> 1) + half;
    const int pc = 16 * duty + (lane >> 3);              // + 8 for the second access; (pc + 8 + 1) & 7 is the same
    const int kphys = lane & 7;                           // physical chunk position in the slot; it holds logical chunk klog
    const uint32_t gvoff = (uint32_t)(pc * PIXB + (kphys ^ ((pc + 1) & 7)) * 16);     // global offset relative to (row, column c0 - 2)
    const int lfin = (pc + 1) * PIXB + kphys * 16;        // LDS offset of that chunk inside a ring row; second: + 8 slots
#pragma unroll
    for (int i = 0; i < NFO; ++i) asm volatile("" : "+v"(fo[i]));
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) asm volatile("" : "+v"(wo[ct]));

    // ---- this workgroup's run of rows: nslots rows ("slots"), described by the table the prologue left in LDS
    const int nslots = *(const int*)(smem + OFF_TAB + 12);            // entry 0 carries the count in its fourth word
    const int nsteps = nslots + LAGY;
    // slot e as three scalars; entries outside [0, nslots) are "nothing to do".  The 16-byte read is wave-uniform (a broadcast).
    auto slot_load = [&](int e) -> i32x4 {
        return (unsigned)e < (unsigned)nslots ? *(const i32x4*)(smem + OFF_TAB + e * 16) : i32x4{0, 0, 0, 0};
    };
    auto slot_of = [&](const i32x4& v) -> Slot {
        return Slot{__builtin_amdgcn_readfirstlane(v[0]), __builtin_amdgcn_readfirstlane(v[1]), __builtin_amdgcn_readfirstlane(v[2])};
    };
    // the two slots a wave looks at in a step are read one step ahead, so their LDS latency lies under the barrier
    constexpr int LAG_B = VERT ? 0 : LAGY;                            // 3x1 waves: the row to request; 1x3 waves: the row to finish
    i32x4 nxt_me = slot_load(0 - LAG), nxt_b = slot_load(0 - LAG_B);
    int xr = 0;                                                       // t mod NX

#ifdef ERF_DEBUG
    unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    for (int t = 0; t < nsteps; ++t) {
        const Slot me = slot_of(nxt_me), sb = slot_of(nxt_b);
        nxt_me = slot_load(t + 1 - LAG);
        nxt_b = slot_load(t + 1 - LAG_B);
        auto xrow = [&](int back) { const int r = xr - back; return OFF_X + (r < 0 ? r + NX : r) * ROWB; };   // ring row of x row t - back
        // ---- 1x3 waves: finish row t - LAGY: y = relu(y' + x), 16 pixels as whole chunks to HBM; x is still in its ring row.  This
        //      runs under the 3x1 waves' matrix instructions, which start at once.
        if constexpr (!VERT) {
            if ((sb.fl & 4) && !NBS_DBG(4)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const i32x4 a = *(const i32x4*)(smem + OFF_Y + ((t - LAGY) & (NY - 1)) * ROWB + lfin + j * 8 * PIXB);
                    const i32x4 r = *(const i32x4*)(smem + xrow(LAGY) + lfin + j * 8 * PIXB);
                    float fy[8], fr[8];
                    ChunkOps<T>::unpack(a, fy);
                    ChunkOps<T>::unpack(r, fr);
                    int oo[4];                                         // scalars: bit casts of vector ELEMENTS misread with this compiler
#pragma unroll
                    for (int e = 0; e < 4; ++e)                        // ReLU on the rounded pairs: same bits as before the rounding
                        oo[e] = relu_bf16x2(pack_bf16x2(fy[2 * e] + fr[2 * e], fy[2 * e + 1] + fr[2 * e + 1]));
                    const i32x4 o{oo[0], oo[1], oo[2], oo[3]};
                    const bool ok = pc + 8 * j >= 2 && pc + 8 * j < 2 + sb.tw() && sb.c0 - 2 + pc + 8 * j < p.W;
                    // non-temporal: the output is not read again by this launch
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rs_y, ok ? gvoff + j * 8 * PIXB : ERF_OOB,
                                                           (uint32_t)sb.soff, 2);
                }
                // a 16-byte buffer store with a scalar offset reads its data registers for some cycles after its issue, longer when
                // the memory pipeline is backed up: nothing may rewrite them meanwhile
                asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            }
        }
        // ---- 3x1 waves: request x row t (always two operations, masked when there is nothing to fetch: the wait below counts them)
        if constexpr (VERT) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = (sb.fl & 8) && pc + 8 * j < sb.fw() && (unsigned)(sb.c0 - 2 + pc + 8 * j) < (unsigned)p.W && !NBS_DBG(8);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    rs_x, (__attribute__((address_space(3))) void*)(smem + xrow(0) + PIXB + (2 * duty + j) * 1024), 16,
                    ok ? gvoff + j * 8 * PIXB : ERF_OOB, (uint32_t)sb.soff, 0, 0);
            }
        }

        NBS_STAMP(0)
        // ---- this role's convolution on row t - LAG
        constexpr int need = ROLE < 2 ? 2 : 4;
        // A narrow strip has one matrix tile per wave instead of two: tile `half` of the frame, i.e. this lane's usual slot minus
        // 16 half slots (the swizzle key, slot & 7, is the same).  Two copies of the code, so that the wide one keeps its registers.
        auto conv = [&](auto narrow_c) {
            constexpr bool NARROW = decltype(narrow_c)::value;
            const int shift = NARROW ? half * 16 * PIXB : 0;
            char* dst;
            if constexpr (ROLE == 0) dst = smem + OFF_A1 + ((t - LAG1) & (NA1 - 1)) * ROWB - shift;
            else if constexpr (ROLE == 1) dst = smem + OFF_A2 + ((t - LAG2) & (NA2 - 1)) * ROWB - shift;
            else if constexpr (ROLE == 2) dst = smem + OFF_A3 + ((t - LAG3) & (NA3 - 1)) * ROWB - shift;
            else dst = smem + OFF_Y + ((t - LAG4) & (NY - 1)) * ROWB - shift;
            if (ROLE == 1 && !(me.fl & 8)) {
                // a2 rows outside the image are the zero padding of conv3x1_2
#pragma unroll
                for (int pt = 0; pt < (NARROW ? 1 : 2); ++pt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) *(i32x2*)(dst + wo[ct] + pt * 16 * PIXB) = i32x2{0, 0};
                return;
            }
            // fragment addresses of this step
            int fa[6];
            if constexpr (VERT) {
#pragma unroll
                for (int tap = 0; tap < 3; ++tap) {
                    const int rb = (ROLE == 0 ? xrow(LAG + 1 - tap) : OFF_A2 + ((t - LAG - 1 + tap) & (NA2 - 1)) * ROWB) - shift;
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) fa[tap * 2 + kc] = fo[kc] + rb;
                }
            } else {
                const int rb = (ROLE == 1 ? OFF_A1 + ((t - LAG2) & (NA1 - 1)) * ROWB : OFF_A3 + ((t - LAG4) & (NA3 - 1)) * ROWB) - shift;
#pragma unroll
                for (int i = 0; i < 6; ++i) fa[i] = fo[i] + rb;
            }
            i32x4 b0[KS];
            [[maybe_unused]] i32x4 b1[KS];
            if (NBS_DBG(16)) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b0[ks] = b1[ks] = i32x4{fa[ks], 0, 0, 0};
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) b0[ks] = lds_read_b128<0>(fa[ks]);
                if constexpr (!NARROW) {
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) b1[ks] = lds_read_b128<16 * PIXB>(fa[ks]);
                }
            }
            f32x4 acc[NARROW ? 1 : 2][CT];
            auto epi = [&](auto pti) {
                constexpr int pt = decltype(pti)::value;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    f32x4 v = acc[pt][ct];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[ct][e] + sh[ct][e];
                    // scalars, not vector elements: a bit cast of a vector ELEMENT misreads with this compiler
                    int o0 = pack_bf16x2(v[0], v[1]), o1 = pack_bf16x2(v[2], v[3]);
                    if constexpr (ROLE != 3) {         // ReLU on the rounded pairs: one instruction per two values, same bits
                        o0 = relu_bf16x2(o0);
                        o1 = relu_bf16x2(o1);
                    }
                    *(i32x2*)(dst + wo[ct] + pt * 16 * PIXB) = i32x2{o0, o1};
                }
            };
            wait_lgkm<NARROW ? 0 : KS>();
            NBS_STAMP(1)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) tie_reg(b0[ks]);
            if (NBS_DBG(1)) {
#pragma unroll
                for (int pt = 0; pt < (NARROW ? 1 : 2); ++pt)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[pt][ct] = __builtin_bit_cast(f32x4, b0[ct]);
                wait_lgkm<0>();
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[0][ct] = mma_step<T>(w[ct][ks], b0[ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[0][ct]);
                if constexpr (!NARROW) {
                    wait_lgkm<0>();
                    NBS_STAMP(2)
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) tie_reg(b1[ks]);
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[1][ct] = mma_step<T>(w[ct][ks], b1[ks], ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[1][ct]);
                }
            }
            if (!NBS_DBG(2)) {
                epi(std::integral_constant<int, 0>{});
                if constexpr (!NARROW) epi(std::integral_constant<int, 1>{});
            }
            if constexpr (VERT) {
                // columns of the frame outside the image (first / last strip only; wave-uniform and rare): a1 / a3 are zero
                // there -- the next convolution's zero padding.  The zeros follow the values through the same, ordered, queue.
                if (me.c0 < 2 || me.c0 - 2 + (NARROW ? FWN : FW) > p.W) {
                    const int cbase = me.c0 - 2 + (NARROW ? 16 : 32) * half + r16;
#pragma unroll
                    for (int pt = 0; pt < (NARROW ? 1 : 2); ++pt) {
                        if ((unsigned)(cbase + 16 * pt) >= (unsigned)p.W) {
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct) *(i32x2*)(dst + wo[ct] + pt * 16 * PIXB) = i32x2{0, 0};
                        }
                    }
                }
            }
            NBS_STAMP(3)
        };
        if (me.fl & need) {
            if (me.fl & 16) conv(std::true_type{});
            else conv(std::false_type{});
        }
        NBS_STAMP(4)
        if constexpr (!VERT) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            // the x row requested D steps ago has landed (two requests per step, in order, and nothing else in this wave's queue)
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * D) : "memory");
        }
        NBS_STAMP(5)
        asm volatile("s_barrier" ::: "memory");
        xr = xr + 1 == NX ? 0 : xr + 1;
        NBS_STAMP(6)
    }
#ifdef ERF_DEBUG
    if (NBS_DBG(256) && lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_nbs_stamps + (blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 7; ++i) o[i] = st[i];
        o[7] = __builtin_amdgcn_s_memtime() - tstart;
    }
#endif
}

}  // namespace nbs
#endif

__global__ __launch_bounds__(nbs::NTHREADS, 1)
void nb1d_block64s_kernel(const NbStreamParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace nbs;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int o = tid * 16; o < OFF_TAB; o += NTHREADS * 16) *(i32x4*)(smem + o) = i32x4{0, 0, 0, 0};
    // The slot table: this workgroup's rows are the run [L0, L1) of the linearised (image, strip, row) space, cut into chunks
    // where it crosses into the next strip; a chunk of rc output rows starting at row r0 is rc + 4 slots (rows r0-2 .. r0+rc+1).
    {
        const int L0 = p.l0[blockIdx.x], L1 = p.l0[blockIdx.x + 1];
        int nslots = 0;
        if (L0 < L1) {
            const int cid0 = L0 / p.H, r00 = L0 - cid0 * p.H;
            const int rc0 = p.H - r00 < L1 - L0 ? p.H - r00 : L1 - L0;
            nslots = (L1 - L0) + 4 * ((L1 - 1) / p.H - cid0 + 1);
            for (int e = tid; e < nslots; e += NTHREADS) {
                int k = 0, ci = e, r0 = r00, rc = rc0;
                if (e >= rc0 + 4) {
                    const int e2 = e - (rc0 + 4);
                    k = 1 + e2 / (p.H + 4);
                    ci = e2 - (k - 1) * (p.H + 4);
                    r0 = 0;
                    const int left = (L1 - L0) - rc0 - (k - 1) * p.H;
                    rc = p.H < left ? p.H : left;
                }
                const int cid = cid0 + k, n = cid / p.S, st = cid - n * p.S;
                const int h = r0 - 2 + ci, c0 = st * TW;
                const int fl = 1 | ((ci >= 1 && ci < rc + 3) ? 2 : 0) | ((ci >= 2 && ci < rc + 2) ? 4 : 0) | ((h >= 0 && h < p.H) ? 8 : 0) |
                               ((p.narrow_last && st == p.S - 1) ? 16 : 0);
                *(i32x4*)(smem + OFF_TAB + e * 16) = i32x4{((n * p.H + h) * p.W + c0) * PIXB, c0, fl, e == 0 ? nslots : 0};
            }
        } else if (tid == 0) {
            *(i32x4*)(smem + OFF_TAB) = i32x4{0, 0, 0, 0};
        }
    }
    ERF_LDS_BARRIER();
    switch (wave >> 1) {
        case 0: role_loop<0>(p, smem, wave, lane); break;
        case 1: role_loop<1>(p, smem, wave, lane); break;
        case 2: role_loop<3>(p, smem, wave, lane); break;
        default: role_loop<2>(p, smem, wave, lane); break;
    }
#endif
}

int nb1d_block64s_launch(const void* x, void* y, const void* const* packed4, const float* tab, int dtype, int N, int H, int W, int C,
                         int dilation, hipStream_t stream) {
    using namespace nbs;
    if (dtype != 1 || C != nbs::C || dilation != 1) return ERF_EUNSUPPORTED;
    if (N < 1 || H < 1 || W < 1) return ERF_EINVAL;
    NbStreamParams p{};
    p.x = (const T*)x; p.y = (T*)y;
    for (int i = 0; i < 4; ++i) p.w[i] = packed4[i];
    p.tab = tab; p.N = N; p.H = H; p.W = W;
    // strips: 60 output columns each; a remainder of at most 28 columns is a narrow strip (half the matrix work per row)
    const int wide = W / TW, rest = W - wide * TW;
    p.S = wide + (rest > 0);
    p.narrow_last = rest > 0 && rest <= TWN && erf_debug_env("ERF_NBS_NARROW", 1);         // (the switch exists in debug builds only)
    const long long rows = (long long)N * p.S * H;
    // byte offsets are 32-bit, and a row two above / below an image may be addressed (masked) before it is clipped
    const long long bytes = (long long)N * H * W * PIXB;
    if (rows > 0x3FFFFFFF || bytes + (long long)(3 * W + FW) * PIXB >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;
    p.bytes = (uint32_t)bytes;
    // Partition of the linearised (image, strip, row) space into at most MAX_WGS runs of about equal COST: a row of a wide strip
    // costs one step, a row of a narrow strip NARROW_COST of one (measured: a step is mostly fixed cost -- slot decode, LDS
    // latency, the barrier -- so half the matrix work saves only ~15 %; sweeping the constant over 0.5 .. 1.0 put the best
    // batch time at 0.8 .. 1.0), every chunk (a run's part inside one strip) four more rows of halo, every run the pipeline's
    // fill and drain.  A workgroup's pipeline takes LAGY steps to fill and drain whatever the length of its run: below 40 rows
    // per workgroup it is better to use fewer CUs (and leave the rest to whatever else runs on the device).
    if (rows > (long long)MAX_WGS * (MAX_SLOTS - 8)) return ERF_EUNSUPPORTED;     // more rows than 256 slot tables hold
    const double NARROW_COST = erf_debug_env("ERF_NBS_NARROW_COST", 850) * 1e-3;
    int want = MAX_WGS;
    if (rows < (long long)MAX_WGS * NBS_MIN_ROWS) want = (int)((rows + NBS_MIN_ROWS - 1) / NBS_MIN_ROWS);
    const long long ncols = (long long)N * p.S;
    auto col_cost = [&](long long cid) { return (p.narrow_last && cid % p.S == p.S - 1) ? NARROW_COST : 1.0; };
    double total = 0;
    for (long long c = 0; c < ncols; ++c) total += col_cost(c) * (H + 4);
    // the partition depends on the shape only: the last one is remembered per thread (a replayed graph never comes here)
    static thread_local struct { int N, H, W, grid; int l0[MAX_WGS + 1]; } memo = {0, 0, 0, 0, {0}};
    int grid = 0;
    if (memo.N == N && memo.H == H && memo.W == W) {
        grid = memo.grid;
        for (int b = 0; b <= MAX_WGS; ++b) p.l0[b] = memo.l0[b];
    } else {
    int tries = 0;
    for (double target = total / want;; target *= 1.02) {
        if (++tries > 1000) return ERF_EUNSUPPORTED;
        // greedy: fill a run up to `target`, never beyond MAX_SLOTS table entries
        grid = 0;
        long long row = 0;
        bool ok = true;
        while (row < rows) {
            if (grid == MAX_WGS) { ok = false; break; }
            p.l0[grid++] = (int)row;
            double cost = 0;
            int slots = 0;
            while (row < rows) {
                const long long cid = row / H;
                const int left = (int)((cid + 1) * H - row);                    // rows to the end of this strip
                const double w = col_cost(cid);
                int take = (int)((target - cost - 4 * w) / w);
                if (take > left) take = left;
                if (take > MAX_SLOTS - slots - 4) take = MAX_SLOTS - slots - 4;
                if (take < 1) {
                    if (cost == 0) take = 1; else break;                        // a run holds at least one row
                }
                row += take; cost += (take + 4) * w; slots += take + 4;
                if (take < left) break;                                         // the run is full
            }
        }
        if (ok && grid <= want) break;                                  // never more runs than asked for: a run holds a whole CU
    }
    for (int b = grid; b <= MAX_WGS; ++b) p.l0[b] = (int)rows;
    memo.N = N; memo.H = H; memo.W = W; memo.grid = grid;
    for (int b = 0; b <= MAX_WGS; ++b) memo.l0[b] = p.l0[b];
    }
    p.dbg = erf_debug_env("ERF_BLOCK_DBG", 0);
    if (hipFuncSetAttribute((const void*)nb1d_block64s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess)
        return ERF_ELAUNCH;
    hipLaunchKernelGGL(nb1d_block64s_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, stream, p);
    return erf_check_launch();
}

#ifdef ERF_DEBUG
// diagnostic builds only: the cycle stamps of the last launch, [256 workgroups][8 waves][8 sections] (section 7 = whole loop)
extern "C" int erf_debug_block64s_stamps(unsigned long long* host_out) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_nbs_stamps), sizeof(unsigned long long) * 256 * 8 * 8) == hipSuccess ? 0 : -3;
}
#endif
