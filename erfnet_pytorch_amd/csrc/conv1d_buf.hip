// Factorized 3x1 / 1x3 (dilated) convolution C -> C, forward and data-gradient form -- the kernel of
// the training step and of the inference layers the fused pair kernel does not take.
//
//     y = act( (scale * conv(x) + shift) * [gate > 0] + residual )        (+ per-channel sum / sum of squares of y)
//
// Same lattice tiling as described in conv1d.hip (Tm lattice positions spaced d apart x Tn positions of
// the other axis, halo 2 for every dilation, one staged tile serving all three taps), with:
//  * the input tile DMA-ed (buffer_load ... lds, 16 B per lane) into a double buffer: tile k+2 is
//    requested as soon as tile k's buffer is free; out-of-image pixels are lanes whose offset lies
//    beyond the buffer descriptor, which read zeros (and whose stores are dropped);
//  * accumulators written back through LDS as [pixel][C] in the output type and stored as whole
//    16-byte chunks, 64 consecutive per wave instruction; gate / residual read the same way;
//  * all per-lane addressing computed once per launch (registers) + one scalar offset per tile decoded
//    with multiply-high division: with 2 waves per SIMD the instruction count per tile is the budget;
//  * workgroups of one XCD walk one contiguous eighth of the tile sequence (halo re-reads hit its L2);
//  * STATS: every thread owns one 16-byte channel chunk in the read-back phase, so it can keep running
//    sums of y and y^2 for its 4 / 8 channels in registers over the whole launch; they meet in LDS at
//    the end and leave as one fp64 partial row per workgroup -- the BatchNorm statistics (forward) or
//    the bias gradient (backward) of the tensor just written, without another pass over it.
#include "common.cuh"
#include "conv1d.cuh"
#include "conv1d_tile.cuh"
#include <stdlib.h>

struct BufParams {
    const void* x;
    void* y;
    const void* w;
    const float* scale;
    const float* shift;
    const void* res;
    const void* gate;
    int gate_bits;          // gate is one byte per 16-byte chunk of the tensor (bit e = [value e > 0]) instead of the tensor
    double* stats;          // [grid][2C] partial rows, or null: (sum, sum of squares) of y, or the two sums below
    // BatchNorm-backward sums of y taken as the gradient at a BatchNorm's output (null bz: off):
    //   row = (sum y*m, sum y*m*xhat), xhat = (bz - bmean) * binvstd, m = bmask[image][channel] (null: 1)
    const void* bz;
    const float *bmean, *binvstd, *bmask;
    int gate_after;         // gate and residual both given: 0: y = gate(conv) + res, 1: y = gate(conv + res)
    int d, relu;
    int A, Bn;
    long long img_stride;
    int stride_u, stride_v;
    int Tm, Tn, log2Tn;
    int CA, CB, ntiles;
    FastDiv fCB, fCA, fD;
    uint32_t bytes, pad;
};

struct BufGeo {
    uint32_t soff;
    int ulo, uhi, vhi;
    int img;                // image index of the tile (axis H), 0 when rows of all images are merged (axis W)
};

template <int PIXB>
__device__ __forceinline__ BufGeo buf_geo(const BufParams& p, int tile, int tile_end) {
    BufGeo g;
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
    // offset of tile pixel (u = 1, v = 0) = first output pixel; the x descriptor starts one lattice step earlier
    g.soff = (uint32_t)(img * (int)p.img_stride + (a_first + p.d) * (p.stride_u / p.d) + b0 * p.stride_v) * PIXB;
    g.ulo = a_first < 0 ? 1 : 0;
    int uhi = (int)fast_div((uint32_t)(p.A - a_first + p.d - 1), p.fD);
    if (uhi > p.Tm + 2) uhi = p.Tm + 2;
    if (tile >= tile_end) uhi = 0;
    g.uhi = uhi > g.ulo ? uhi : g.ulo;
    const int vhi = p.Bn - b0;
    g.vhi = vhi < p.Tn ? vhi : p.Tn;
    g.img = img;
    return g;
}

// STATS: 0 none, 1 (sum, sum of squares) of the output, 2 BatchNorm-backward sums of the output
template <typename T, int C, int WM, int WN, int M, int KI, int OCC, bool POST, int STATS>
__global__ __launch_bounds__(WM * WN * 64, OCC)
void conv1d_buf_kernel(const BufParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int EPC = Elem<T>::EPC;
    constexpr int CH = C / EPC;
    constexpr int PIXB = C * (int)sizeof(T);
    constexpr int NW = WM * WN;
    constexpr int NT = NW * 64;
    constexpr int CT = C / 16;
    constexpr int CTW = CT / WN;
    constexpr int PTW = (M / 16) / WM;
    constexpr int KS = (3 * CH + 3) / 4;
    constexpr int BUFB = KI * NW * 1024;
    constexpr int NS = (M * CH) / NT;
    constexpr int NCL = CH >= 4 ? (CH < 16 ? CH : 16) / 4 : 1;
    constexpr int NLB = CH >= 4 ? 3 * NCL : KS;
    constexpr int GS = (KS % 6 == 0) ? 6 : (KS % 4 == 0 ? 4 : KS);
    constexpr int NG = PTW * KS / GS;
    constexpr bool ASMREAD = sizeof(T) == 2;
    static_assert(CT % WN == 0 && (M / 16) % WM == 0, "tile split");
    static_assert((M * CH) % NT == 0 && M * PIXB <= BUFB && NT % CH == 0, "epilogue staging");
    static_assert((PTW * KS) % GS == 0, "read groups");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int q = lane >> 4, r16 = lane & 15;
    const int Tn = p.Tn;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc((const char*)p.x - p.pad, p.bytes + p.pad);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(p.y, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.res ? p.res : p.x, p.bytes);
    const __amdgpu_buffer_rsrc_t rs_g = make_rsrc(p.gate ? p.gate : p.x, p.gate && p.gate_bits ? p.bytes >> 4 : p.bytes);
    const __amdgpu_buffer_rsrc_t rs_z = make_rsrc(p.bz ? p.bz : p.x, p.bytes);

    i32x4 wreg[CTW][KS];
    f32x4 sc[CTW], sh[CTW];
    {
        const i32x4* wp = (const i32x4*)p.w;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) wreg[c][ks] = wp[((wn * CTW + c) * KS + ks) * 64 + lane];
            const int co = (wn * CTW + c) * 16 + q * 4;
            sc[c] = p.scale ? *(const f32x4*)(p.scale + co) : f32x4{1.f, 1.f, 1.f, 1.f};
            sh[c] = p.shift ? *(const f32x4*)(p.shift + co) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- launch-invariant per-lane addressing
    uint32_t in_rel[KI];
    int in_uv[KI];
#pragma unroll
    for (int k = 0; k < KI; ++k) {
        const int id = (k * NW + wave) * 64 + lane;
        const int pix = id / CH, slot = id % CH;
        const int u = pix >> p.log2Tn, v = pix & (Tn - 1);
        in_rel[k] = (uint32_t)(u * p.stride_u + v * p.stride_v) * PIXB + swz<PIXB>(pix, slot) * 16;
        in_uv[k] = ((u < p.Tm + 2 ? u : 255) << 16) | v;
    }
    int lb0[NLB];
    if constexpr (CH >= 4) {
#pragma unroll
        for (int tap = 0; tap < 3; ++tap)
#pragma unroll
            for (int cl = 0; cl < NCL; ++cl)
                lb0[tap * NCL + cl] = ((wm * PTW) * 16 + tap * Tn + r16) * PIXB + swz<PIXB>(r16, cl * 4 + q) * 16;
    } else {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int gi = ks * 4 + q;
            const int tap = gi / CH < 2 ? gi / CH : 2;
            lb0[ks] = ((wm * PTW) * 16 + tap * Tn + r16) * PIXB + swz<PIXB>(r16, gi % CH) * 16;
        }
    }
    int stg_w[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
        if constexpr (sizeof(T) == 2) {
            const int chunk = (wn * CTW + c) * 2 + (q >> 1);
            stg_w[c] = ((wm * PTW) * 16 + r16) * PIXB + swz<PIXB>(r16, chunk) * 16 + (q & 1) * 8;
        } else {
            const int chunk = (wn * CTW + c) * 4 + q;
            stg_w[c] = ((wm * PTW) * 16 + r16) * PIXB + swz<PIXB>(r16, chunk) * 16;
        }
    }
    int rb_lds[NS], out_iv[NS];
    uint32_t out_rel[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int cid = tid + s * NT;
        const int m = cid / CH, chunk = cid % CH;
        const int i = m >> p.log2Tn, v = m & (Tn - 1);
        rb_lds[s] = m * PIXB + swz<PIXB>(m, chunk) * 16;
        out_iv[s] = (i << 16) | v;
        out_rel[s] = (uint32_t)(i * p.stride_u + v * p.stride_v) * PIXB + chunk * 16;
    }
    // pin: no wait for the weights inside the loop (it would drain the DMA prefetch), no rematerialisation
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) tie_reg(wreg[c][ks]);
        asm volatile("" : "+v"(sc[c]), "+v"(sh[c]), "+v"(stg_w[c]));
    }
#pragma unroll
    for (int k = 0; k < KI; ++k) asm volatile("" : "+v"(in_rel[k]), "+v"(in_uv[k]));
#pragma unroll
    for (int i = 0; i < NLB; ++i) asm volatile("" : "+v"(lb0[i]));
#pragma unroll
    for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(rb_lds[s]), "+v"(out_iv[s]), "+v"(out_rel[s]));

    auto dma = [&](const BufGeo& g, int boff) {
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int u = in_uv[k] >> 16, v = in_uv[k] & 0xFFFF;
            const bool ok = (unsigned)(u - g.ulo) < (unsigned)(g.uhi - g.ulo) && v < g.vhi;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(smem + boff + (k * NW + wave) * 1024),
                                                     16, ok ? in_rel[k] : ERF_OOB, g.soff, 0, 0);
        }
    };

    float ssum[EPC], ssq[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) { ssum[e] = 0.f; ssq[e] = 0.f; }
    float bmu[EPC], bis[EPC];            // STATS == 2: statistics of this thread's channel chunk (tid % CH)
    if constexpr (STATS == 2) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            bmu[e] = p.bmean[(tid % CH) * EPC + e];
            bis[e] = p.binvstd[(tid % CH) * EPC + e];
        }
    }
    const float floor_v = p.relu ? 0.f : -__builtin_huge_valf();

    int tile = blockIdx.x, tile_end = p.ntiles, G = gridDim.x;
    if ((G & 7) == 0) {
        const int per = (p.ntiles + 7) >> 3, xcd = blockIdx.x & 7;
        tile = xcd * per + (blockIdx.x >> 3);
        tile_end = (xcd + 1) * per < p.ntiles ? (xcd + 1) * per : p.ntiles;
        G >>= 3;
    }
    BufGeo g = buf_geo<PIXB>(p, tile, tile_end);
    BufGeo g1 = buf_geo<PIXB>(p, tile + G, tile_end);
    dma(g, 0);
    dma(g1, BUFB);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");
    asm volatile("s_barrier" ::: "memory");

    int cur = 0;
    for (; tile < tile_end; tile += G, cur ^= 1) {
        const int boff = cur * BUFB;
        int lb[NLB];
#pragma unroll
        for (int i = 0; i < NLB; ++i) lb[i] = lb0[i] + boff;

        f32x4 acc[PTW][CTW];
        if constexpr (ASMREAD) {
            // hand-placed B-fragment reads in groups of GS: group g+1 in flight while group g feeds the MFMAs
            i32x4 bb[2][GS];
            auto issue = [&](auto gi, i32x4* dst) {
                static_for<0, GS>([&](auto e) {
                    constexpr int r = decltype(gi)::value * GS + decltype(e)::value;
                    constexpr int pt = r / KS, ks = r % KS;
                    constexpr int tap = (ks * 4) / CH;
                    constexpr int li = CH >= 4 ? tap * NCL + (ks % NCL) : ks;
                    constexpr int imm = pt * 16 * PIXB + (CH > 16 ? (((ks * 4) % CH) & ~15) * 16 : 0);
                    static_assert(imm < 65536, "ds offset field");
                    dst[decltype(e)::value] = lds_read_b128<imm>(lb[li]);
                });
            };
            issue(std::integral_constant<int, 0>{}, bb[0]);
            static_for<0, NG>([&](auto gi) {
                constexpr int gg = decltype(gi)::value;
                i32x4* cb = bb[gg & 1];
                if constexpr (gg + 1 < NG) {
                    issue(std::integral_constant<int, gg + 1>{}, bb[(gg + 1) & 1]);
                    wait_lgkm<GS>();
                } else {
                    wait_lgkm<0>();
                }
                static_for<0, GS>([&](auto e) { tie_reg(cb[decltype(e)::value]); });
                static_for<0, GS>([&](auto e) {
                    constexpr int r = gg * GS + decltype(e)::value;
                    constexpr int pt = r / KS, ks = r % KS;
                    i32x4 b = cb[decltype(e)::value];
                    if constexpr ((3 * CH) % 4 != 0) {
                        if (ks * 4 + q >= 3 * CH) b = i32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int c = 0; c < CTW; ++c)
                        acc[pt][c] = mma_step<T>(wreg[c][ks], b, ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pt][c]);
                });
            });
        } else {
            // fp32 is MFMA-bound (4 instructions per K-step): compiler-scheduled reads are enough
#pragma unroll
            for (int pt = 0; pt < PTW; ++pt) {
#pragma unroll
                for (int c = 0; c < CTW; ++c) acc[pt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const int tap = (ks * 4) / CH;
                    const int hi = CH > 16 ? (((ks * 4) % CH) & ~15) * 16 : 0;
                    const int off = (CH >= 4 ? lb[tap * NCL + (ks % NCL)] : lb[ks]) + pt * 16 * PIXB + hi;
                    i32x4 b = *(const i32x4*)(smem + off);
                    if constexpr ((3 * CH) % 4 != 0) {
                        if (ks * 4 + q >= 3 * CH) b = i32x4{0, 0, 0, 0};
                    }
#pragma unroll
                    for (int c = 0; c < CTW; ++c) acc[pt][c] = mma_step<T>(wreg[c][ks], b, acc[pt][c]);
                }
            }
        }
        ERF_LDS_BARRIER();                 // every wave is done reading the input tile

        // ---- accumulators -> LDS as [pixel][C] in the output type
#pragma unroll
        for (int pt = 0; pt < PTW; ++pt) {
#pragma unroll
            for (int c = 0; c < CTW; ++c) {
                f32x4 v = acc[pt][c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = v[e] * sc[c][e] + sh[c][e];
                    if constexpr (!POST) v[e] = fmaxf(v[e], floor_v);
                }
                if constexpr (sizeof(T) == 2) *(i32x2*)(smem + boff + stg_w[c] + pt * 16 * PIXB) = pack_bf16x4(v);
                else *(f32x4*)(smem + boff + stg_w[c] + pt * 16 * PIXB) = v;
            }
        }
        ERF_LDS_BARRIER();

        // ---- coalesced read-back (+ gate, residual, ReLU, statistics)
        i32x4 outv[NS];
        uint32_t ooff[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int i = out_iv[s] >> 16, v = out_iv[s] & 0xFFFF;
            const bool ok = i < g.uhi - 1 && v < g.vhi;
            ooff[s] = ok ? out_rel[s] : ERF_OOB;
            outv[s] = *(const i32x4*)(smem + boff + rb_lds[s]);
        }
        if constexpr (POST) {
            bool gated_by_bits = false;
            if constexpr (sizeof(T) == 2) if (p.gate && p.gate_bits) {
                gated_by_bits = true;
                // the gate as one byte per chunk (bit e = [value e > 0]): a sixteenth of the bytes of the tensor it stands for
                uint32_t gb[NS];
                u32x4 rv[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s)
                    gb[s] = __builtin_amdgcn_raw_buffer_load_b8(rs_g, ooff[s] == ERF_OOB ? ERF_OOB : ooff[s] >> 4, g.soff >> 4, 0);
                if (p.res) {
#pragma unroll
                    for (int s = 0; s < NS; ++s) rv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[s], g.soff, 0);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float f[EPC];
                    ChunkOps<T>::unpack(outv[s], f);
                    if (p.res) {
                        float r[EPC];
                        ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, rv[s]), r);
                        if (p.gate_after) {
#pragma unroll
                            for (int e = 0; e < EPC; ++e) f[e] = (gb[s] >> e) & 1u ? f[e] + r[e] : 0.f;
                        } else {
#pragma unroll
                            for (int e = 0; e < EPC; ++e) f[e] = ((gb[s] >> e) & 1u ? f[e] : 0.f) + r[e];
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = (gb[s] >> e) & 1u ? f[e] : 0.f;
                    }
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] = fmaxf(f[e], floor_v);
                    outv[s] = ChunkOps<T>::pack(f);
                }
            }
            if (!gated_by_bits) {
            // one operand in the common case (gate: ReLU backward; residual: skip connection), both allowed
            u32x4 pv[NS];
            const bool both = p.gate && p.res;
#pragma unroll
            for (int s = 0; s < NS; ++s) pv[s] = __builtin_amdgcn_raw_buffer_load_b128(p.gate ? rs_g : rs_r, ooff[s], g.soff, 0);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float f[EPC], t[EPC];
                ChunkOps<T>::unpack(outv[s], f);
                ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, pv[s]), t);
                if (p.gate) {
                    if (both && p.gate_after) {
                        float r[EPC];
                        ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[s], g.soff, 0)), r);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = t[e] > 0.f ? f[e] + r[e] : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < EPC; ++e) f[e] = t[e] > 0.f ? f[e] : 0.f;
                        if (both) {
                            ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[s], g.soff, 0)), t);
#pragma unroll
                            for (int e = 0; e < EPC; ++e) f[e] += t[e];
                        }
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < EPC; ++e) f[e] += t[e];
                }
#pragma unroll
                for (int e = 0; e < EPC; ++e) f[e] = fmaxf(f[e], floor_v);
                outv[s] = ChunkOps<T>::pack(f);
            }
            }
        }
        if constexpr (STATS == 1) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float f[EPC];
                ChunkOps<T>::unpack(outv[s], f);
                const float live = ooff[s] == ERF_OOB ? 0.f : 1.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float v = f[e] * live;
                    ssum[e] += v;
                    ssq[e] += v * v;
                }
            }
        }
        if constexpr (STATS == 2) {
            // the stored (rounded) output is what the BatchNorm backward reads back as its incoming gradient
            u32x4 zv[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) zv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_z, ooff[s], g.soff, 0);
            float mk[EPC];
#pragma unroll
            for (int e = 0; e < EPC; ++e) mk[e] = p.bmask ? p.bmask[g.img * C + (tid % CH) * EPC + e] : 1.f;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                float f[EPC], z[EPC];
                ChunkOps<T>::unpack(outv[s], f);
                ChunkOps<T>::unpack(__builtin_bit_cast(i32x4, zv[s]), z);       // masked lanes read zeros and f is gated by `live`
                const float live = ooff[s] == ERF_OOB ? 0.f : 1.f;
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const float v = f[e] * live * mk[e];
                    ssum[e] += v;
                    ssq[e] += v * ((z[e] - bmu[e]) * bis[e]);
                }
            }
        }
        if constexpr (STATS != 0) {        // keep the statistics arithmetic above the stores (see the note below)
#pragma unroll
            for (int e = 0; e < EPC; ++e) asm volatile("" : "+v"(ssum[e]), "+v"(ssq[e])::"memory");
        }
        ERF_LDS_BARRIER();                 // staging consumed: the buffer may take tile k+2

        const uint32_t out_soff = g.soff;
        g = g1;
        g1 = buf_geo<PIXB>(p, tile + 2 * G, tile_end);
        dma(g1, boff);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KI) : "memory");      // this wave's share of tile k+1 has landed
#pragma unroll
        for (int s = 0; s < NS; ++s) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, outv[s]), rs_y, ooff[s], out_soff, 0);
        // A 16-byte buffer store with an SGPR offset still reads its data registers for a few cycles after issue, and
        // the compiler does not guard that case: a VALU write to them scheduled right behind the store corrupted the
        // last lanes of the stored data (observed on gfx950).  Nothing may follow the stores but this pause.
        asm volatile("s_nop 7\n\ts_barrier" ::: "memory");             // ... and everyone else's share of tile k+1 has landed
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // no DMA may outlive the workgroup

    if constexpr (STATS != 0) {
        // thread `tid` owns chunk tid % CH: the NT / CH owners of a chunk meet in LDS, summed in double in a fixed order
        __syncthreads();
        float* red = (float*)smem;
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            red[tid * 2 * EPC + e] = ssum[e];
            red[tid * 2 * EPC + EPC + e] = ssq[e];
        }
        __syncthreads();
        if (tid < 2 * C) {
            const int kind = tid / C, ch = tid % C;
            const int chunk = ch / EPC, e = ch % EPC;
            double a = 0.0;
            for (int k = 0; k < NT / CH; ++k) a += (double)red[(chunk + k * CH) * 2 * EPC + kind * EPC + e];
            p.stats[(long long)blockIdx.x * 2 * C + kind * C + ch] = a;
        }
    }
#endif
}

static inline int ilog2b(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

template <typename T, int C, int WM, int WN, int M, int KI>
static int launch_buf(BufParams p, int N, int H, int W, int axis, int* groups, hipStream_t stream) {
    constexpr int CH = C / Elem<T>::EPC;
    constexpr int PIXB = C * (int)sizeof(T);
    constexpr int NW = WM * WN;
    constexpr int OCC = NW <= 4 ? 2 : 1;
    constexpr size_t lds = 2 * (size_t)KI * NW * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    const int d = p.d;
    if (axis == 0) {
        p.A = H; p.Bn = W; p.img_stride = (long long)H * W; p.stride_u = d * W; p.stride_v = 1;
    } else {
        p.A = W; p.Bn = N * H; p.img_stride = 0; p.stride_u = d; p.stride_v = W;
    }
    const int nimg = axis == 0 ? N : 1;
    const int npts = (p.A + d - 1) / d;
    const int tm_max = axis == 0 ? 8 : 32;
    int Tm = 2;
    while (Tm * 2 <= npts && Tm * 2 <= tm_max && Tm * 2 <= M / 16) Tm *= 2;
    const int Tn = M / Tm;
    p.Tm = Tm; p.Tn = Tn; p.log2Tn = ilog2b(Tn);
    if ((Tm + 2) * Tn * CH > KI * NW * 64) return ERF_EUNSUPPORTED;
    p.CA = (npts + Tm - 1) / Tm;
    p.CB = (p.Bn + Tn - 1) / Tn;
    const long long nt = (long long)nimg * d * p.CA * p.CB;
    if (nt > 0x1FFFFFFF) return ERF_EUNSUPPORTED;
    p.ntiles = (int)nt;
    p.fCB = make_fastdiv(p.CB); p.fCA = make_fastdiv(p.CA); p.fD = make_fastdiv(d);
    const long long bytes = (long long)N * H * W * PIXB, pad = (long long)p.stride_u * PIXB;
    if (bytes + (long long)(Tm + 3) * pad >= 0x7FFFFFFFLL) return ERF_EUNSUPPORTED;      // 32-bit offsets, bit 31 marks masked lanes
    p.bytes = (uint32_t)bytes; p.pad = (uint32_t)pad;
    int wg_per_cu = (int)(160 * 1024 / lds);
    if (wg_per_cu > OCC) wg_per_cu = OCC;
    int grid = 256 * wg_per_cu;
    if (grid > p.ntiles) grid = p.ntiles;
    if (groups) *groups = grid;
    const bool post = p.res || p.gate;
#define ERF_LAUNCH_BUF(POSTV, STATSV)                                                                               \
    {                                                                                                               \
        auto kern = conv1d_buf_kernel<T, C, WM, WN, M, KI, OCC, POSTV, STATSV>;                                     \
        if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
            return ERF_ELAUNCH;                                                                                     \
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, p);                                        \
    }
    if (p.bz) {
        if (!p.stats || !post || !p.bmean || !p.binvstd) return ERF_EINVAL;
        if (p.bmask && axis != 0) return ERF_EUNSUPPORTED;       // the mask is per image: tiles must not mix images
        ERF_LAUNCH_BUF(true, 2)
    } else if (p.stats) { if (post) ERF_LAUNCH_BUF(true, 1) else ERF_LAUNCH_BUF(false, 1) }
    else { if (post) ERF_LAUNCH_BUF(true, 0) else ERF_LAUNCH_BUF(false, 0) }
#undef ERF_LAUNCH_BUF
    return erf_check_launch();
}

// Tiles of an M-pixel tiling (the count launch_buf arrives at), to compare tilings before one is launched.
static long long buf_tiles(int N, int H, int W, int axis, int d, int M) {
    const int A = axis == 0 ? H : W, Bn = axis == 0 ? W : N * H, nimg = axis == 0 ? N : 1;
    const int npts = (A + d - 1) / d, tm_max = axis == 0 ? 8 : 32;
    int Tm = 2;
    while (Tm * 2 <= npts && Tm * 2 <= tm_max && Tm * 2 <= M / 16) Tm *= 2;
    const int Tn = M / Tm;
    return (long long)nimg * d * ((npts + Tm - 1) / Tm) * ((Bn + Tn - 1) / Tn);
}

// Share of the last round of a 256-workgroup launch that has work: 384 tiles keep half the CUs idle in round two.
static double buf_fill(long long nt) { return (double)nt / (double)(((nt + 255) / 256) * 256); }

// Largest number of partial statistic rows a launch can produce (size of the `stats` buffer in rows of 2C doubles).
int conv1d_buf_max_groups() { return 512; }

int conv1d_buf_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                      const void* gate, double* stats, int* groups, int dtype, int N, int H, int W, int C, int axis, int dilation,
                      int relu, hipStream_t stream, const Conv1dBnTail* tail, int gate_bits) {
    static const int off = erf_debug_env("ERF_CONV1D_BUF_OFF", 0);
    if (off) return ERF_EUNSUPPORTED;
    BufParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift; p.res = residual; p.gate = gate; p.stats = stats;
    p.d = dilation; p.relu = relu; p.gate_bits = gate && gate_bits;
    if (p.gate_bits && dtype != 1) return ERF_EUNSUPPORTED;       // the byte-per-chunk gate is read by the bf16 instances only
    if (tail) {
        p.bz = tail->z; p.bmean = tail->mean; p.binvstd = tail->invstd; p.bmask = tail->mask; p.gate_after = tail->gate_after;
    }
    if ((long long)N * H * W >= (1LL << 31)) return ERF_EUNSUPPORTED;
    int rc = ERF_EUNSUPPORTED;
    if (dtype == 1) {
        if (C == 128) {
            // small maps: 64-pixel tiles when they fill the rounds of the one-workgroup-per-CU grid markedly better
            // (batch 6 at 64x128: 768 tiles = 3 full rounds instead of 384 = 1.5 rounds that cost 2)
            const long long t128 = buf_tiles(N, H, W, axis, dilation, 128), t64 = buf_tiles(N, H, W, axis, dilation, 64);
            if (t128 < 1024 && buf_fill(t64) * 0.9 > buf_fill(t128))
                rc = launch_buf<bf16raw, 128, 1, 8, 64, 3>(p, N, H, W, axis, groups, stream);
            if (rc == ERF_EUNSUPPORTED) rc = launch_buf<bf16raw, 128, 1, 8, 128, 5>(p, N, H, W, axis, groups, stream);
            if (rc == ERF_EUNSUPPORTED) rc = launch_buf<bf16raw, 128, 1, 8, 128, 6>(p, N, H, W, axis, groups, stream);
        } else if (C == 64) {
            rc = launch_buf<bf16raw, 64, 2, 2, 128, 5>(p, N, H, W, axis, groups, stream);
            if (rc == ERF_EUNSUPPORTED) rc = launch_buf<bf16raw, 64, 2, 2, 128, 6>(p, N, H, W, axis, groups, stream);
        } else if (C == 16) {
            rc = launch_buf<bf16raw, 16, 4, 1, 512, 5>(p, N, H, W, axis, groups, stream);
        }
    } else {
        if (C == 128) {
            const long long t128 = buf_tiles(N, H, W, axis, dilation, 128), t64 = buf_tiles(N, H, W, axis, dilation, 64);
            if (t128 < 1024 && buf_fill(t64) * 0.9 > buf_fill(t128))
                rc = launch_buf<float, 128, 1, 8, 64, 6>(p, N, H, W, axis, groups, stream);
            if (rc == ERF_EUNSUPPORTED) rc = launch_buf<float, 128, 1, 8, 128, 10>(p, N, H, W, axis, groups, stream);
        } else if (C == 64) {
            rc = launch_buf<float, 64, 1, 4, 128, 10>(p, N, H, W, axis, groups, stream);
            if (rc == ERF_EUNSUPPORTED) rc = launch_buf<float, 64, 1, 4, 128, 12>(p, N, H, W, axis, groups, stream);
        } else if (C == 16) {
            rc = launch_buf<float, 16, 4, 1, 512, 10>(p, N, H, W, axis, groups, stream);
        }
    }
    return rc;
}
