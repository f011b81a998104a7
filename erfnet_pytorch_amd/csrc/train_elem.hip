// Bandwidth-bound kernels of the training step: BatchNorm batch statistics / apply / backward,
// Dropout2d mask, max-pool backward, weighted cross-entropy forward/backward, fused Adam/SGD.
// All reductions are two-stage (per-workgroup partials in a caller-provided workspace, then a
// one-block finalize in double precision): no float atomics, so results are bit-reproducible.
#include "common.cuh"
#include "train.cuh"
#include "conv1d_tile.cuh"


// --------------------------------------------------------------------------------------------
// Per-channel partial sums over an NHWC tensor.  MODE 0: (sum x, sum x^2).
// MODE 1: BatchNorm backward sums (sum dy, sum dy * xhat) with dy = g * [act > 0] * mask[n][c].
// Thread layout: C/4 channel groups x (256 / (C/4)) pixel lanes; 4 channels per thread.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void chan_reduce_kernel(const T* __restrict__ a, const T* __restrict__ gate,
                                                          const float* __restrict__ mask, const T* __restrict__ z,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          long long P, int C, long long HW, float* __restrict__ partials) {
    extern __shared__ double red[];     // [256][8]
    const int C4 = C >> 2;
    const int cg = threadIdx.x % C4, pl = threadIdx.x / C4;
    const int PL = 256 / C4;            // pixel lanes per block (C4 <= 256 and divides 256 for C in {16,64,128})
    const int c = cg * 4;
    double s0[4] = {0, 0, 0, 0}, s1[4] = {0, 0, 0, 0};      // fp64: E[x^2]-E[x]^2 must survive 39 stacked BatchNorms
    f32x4 mu = {0, 0, 0, 0}, is = {1, 1, 1, 1};
    if (MODE == 1) { mu = *(const f32x4*)(mean + c); is = *(const f32x4*)(invstd + c); }
    if (pl < PL) {
        // U pixels per trip: all their loads are issued before the first use, which is what keeps enough bytes in
        // flight with only a few waves per CU (the reduction is one pass over up to three tensors)
        constexpr int U = 4;
        const long long step = (long long)gridDim.x * PL;
        for (long long p0 = (long long)blockIdx.x * PL + pl; p0 < P; p0 += U * step) {
            f32x4 v[U], gt[U], zz[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long p = p0 + u * step;
                const bool live = p < P;
                const long long pp = live ? p : p0;
                v[u] = load4<T>(a + pp * C + c);
                if (!live) v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (MODE == 1) {
                    gt[u] = gate ? load4<T>(gate + pp * C + c) : f32x4{1.f, 1.f, 1.f, 1.f};
                    zz[u] = load4<T>(z + pp * C + c);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (MODE == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { s0[e] += (double)v[u][e]; s1[e] += (double)v[u][e] * (double)v[u][e]; }
                } else {
                    const long long p = p0 + u * step < P ? p0 + u * step : p0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[u][e] = gt[u][e] > 0.f ? v[u][e] : 0.f;
                    if (mask) {
                        const f32x4 mk = *(const f32x4*)(mask + (p / HW) * C + c);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[u][e] *= mk[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s0[e] += (double)v[u][e];
                        s1[e] += (double)(v[u][e] * ((zz[u][e] - mu[e]) * is[e]));
                    }
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[threadIdx.x * 8 + e] = s0[e]; red[threadIdx.x * 8 + 4 + e] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < C4) {
        double t0[4] = {0, 0, 0, 0}, t1[4] = {0, 0, 0, 0};
        for (int l = 0; l < PL; ++l)
#pragma unroll
            for (int e = 0; e < 4; ++e) { t0[e] += red[(l * C4 + cg) * 8 + e]; t1[e] += red[(l * C4 + cg) * 8 + 4 + e]; }
        double* out = (double*)partials + (long long)blockIdx.x * 2 * C;
#pragma unroll
        for (int e = 0; e < 4; ++e) { out[c + e] = t0[e]; out[C + c + e] = t1[e]; }
    }
}

static int red_groups(long long P, int C) {
    const int PL = 256 / (C / 4);
    long long g = (P + (long long)PL * 8 - 1) / ((long long)PL * 8);
    if (g > RED_MAX_GROUPS) g = RED_MAX_GROUPS;
    if (g < 1) g = 1;
    return (int)g;
}

// fp64 partials [RED_MAX_GROUPS][2C] followed by a 4C-float block for per-channel coefficients
size_t bn_workspace_bytes(int C) { return (size_t)RED_MAX_GROUPS * 2 * C * sizeof(double) + 4 * C * sizeof(float); }
float* bn_workspace_coef(void* workspace, int C) { return (float*)((double*)workspace + (size_t)RED_MAX_GROUPS * 2 * C); }


// Sum the fp64 partials [groups][2C] per channel.  One block owns FIN_CB consecutive channels (a single block
// for all channels is limited by what one CU can pull from memory); thread (part, cl) adds groups part,
// part + FIN_PARTS, ... of channel c0 + cl, the parts meet in LDS in a fixed order.  On return threads
// tid < FIN_CB hold the two column sums of channel blockIdx.x * FIN_CB + tid (when that is < C).
#define FIN_THREADS 256
#define FIN_CB 8
#define FIN_PARTS (FIN_THREADS / FIN_CB)
static inline int fin_blocks(int C) { return (C + FIN_CB - 1) / FIN_CB; }
__device__ __forceinline__ void reduce_partials(const double* __restrict__ partials, int groups, int C, double& s, double& ss) {
    __shared__ double sh[2 * FIN_THREADS];
    const int cl = threadIdx.x % FIN_CB, part = threadIdx.x / FIN_CB;
    const int c = blockIdx.x * FIN_CB + cl;
    double a = 0, b = 0;
    if (c < C) {
        // eight independent chains keep the loads in flight; the order of the additions is fixed
        double a8[8] = {0, 0, 0, 0, 0, 0, 0, 0}, b8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int g = part;
        for (; g + 7 * FIN_PARTS < groups; g += 8 * FIN_PARTS) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a8[u] += partials[(long long)(g + u * FIN_PARTS) * 2 * C + c];
                b8[u] += partials[(long long)(g + u * FIN_PARTS) * 2 * C + C + c];
            }
        }
        for (; g < groups; g += FIN_PARTS) {
            a8[0] += partials[(long long)g * 2 * C + c];
            b8[0] += partials[(long long)g * 2 * C + C + c];
        }
        a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        b = ((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7]));
    }
    sh[threadIdx.x] = a;
    sh[FIN_THREADS + threadIdx.x] = b;
    __syncthreads();
    s = 0; ss = 0;
    if (threadIdx.x < FIN_CB)
        for (int k = 0; k < FIN_PARTS; ++k) { s += sh[k * FIN_CB + threadIdx.x]; ss += sh[FIN_THREADS + k * FIN_CB + threadIdx.x]; }
}

__global__ void channel_sum_finalize_kernel(const double* __restrict__ partials, int groups, int C, float* __restrict__ sums) {
    double s, ss;
    reduce_partials(partials, groups, C, s, ss);
    const int c = blockIdx.x * FIN_CB + threadIdx.x;
    if (threadIdx.x < FIN_CB && c < C) sums[c] = (float)s;
}

// Many channel sums in one launch: blockIdx.y = entry, blockIdx.x = channel block of that entry (entries with fewer
// channels leave their extra blocks idle).  The table travels as a kernel argument.
__global__ void channel_sum_finalize_many_kernel(const ChannelSumTable t) {
    const ChannelSumEntry& e = t.e[blockIdx.y];
    if ((int)blockIdx.x * FIN_CB >= e.C) return;
    double s, ss;
    reduce_partials(e.partials, e.groups, e.C, s, ss);
    const int c = blockIdx.x * FIN_CB + threadIdx.x;
    if (threadIdx.x < FIN_CB && c < e.C) e.sums[c] = (float)s;
}

int channel_sum_finalize_many_launch(const ChannelSumTable& t, hipStream_t stream) {
    if (t.count <= 0 || t.count > ERF_CHANNEL_SUM_MAX) return ERF_EINVAL;
    int maxc = 0;
    for (int i = 0; i < t.count; ++i) maxc = t.e[i].C > maxc ? t.e[i].C : maxc;
    hipLaunchKernelGGL(channel_sum_finalize_many_kernel, dim3(fin_blocks(maxc), t.count), dim3(FIN_THREADS), 0, stream, t);
    return erf_check_launch();
}

int channel_sum_finalize_launch(const float* partials, int groups, int C, float* sums, hipStream_t stream) {
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(fin_blocks(C)), dim3(FIN_THREADS), 0, stream, (const double*)partials, groups, C, sums);
    return erf_check_launch();
}

static bool chan_ok(int C) { return C > 0 && (C & 3) == 0 && C <= 512; }

int bn_stats_launch(const void* z, int dtype, long long P, int C, float* partials, int* groups, hipStream_t stream) {
    if (!chan_ok(C)) return ERF_EUNSUPPORTED;
    const int g = red_groups(P, C);
    *groups = g;
    if (dtype == 1)
        hipLaunchKernelGGL((chan_reduce_kernel<bf16raw, 0>), dim3(g), dim3(256), 256 * 8 * sizeof(double), stream,
                           (const bf16raw*)z, (const bf16raw*)nullptr, (const float*)nullptr, (const bf16raw*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, P, C, 1LL, partials);
    else
        hipLaunchKernelGGL((chan_reduce_kernel<float, 0>), dim3(g), dim3(256), 256 * 8 * sizeof(double), stream,
                           (const float*)z, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, P, C, 1LL, partials);
    return erf_check_launch();
}

// mean / invstd / folded affine / running statistics from the partial sums (one block, double).
__global__ void bn_finalize_kernel(const double* __restrict__ partials, int groups, int C, double count,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* __restrict__ running_mean, float* __restrict__ running_var, float momentum,
                                   float eps, float* __restrict__ mean, float* __restrict__ invstd,
                                   float* __restrict__ scale, float* __restrict__ shift) {
    double s, ss;
    reduce_partials(partials, groups, C, s, ss);
    const int c = blockIdx.x * FIN_CB + threadIdx.x;
    if (threadIdx.x < FIN_CB && c < C) {
        const double m = s / count;
        double var = ss / count - m * m;
        if (var < 0) var = 0;
        const double is = 1.0 / sqrt(var + (double)eps);
        mean[c] = (float)m;
        invstd[c] = (float)is;
        const double sc = (double)gamma[c] * is;
        scale[c] = (float)sc;
        shift[c] = (float)((double)beta[c] - m * sc);
        if (running_mean) {
            const double unbiased = count > 1 ? var * count / (count - 1) : var;
            running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * m);
            running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * unbiased);
        }
    }
}

int bn_finalize_launch(const float* partials, int groups, int C, long long count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t stream) {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(fin_blocks(C)), dim3(FIN_THREADS), 0, stream, (const double*)partials, groups, C, (double)count, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, invstd, scale, shift);
    return erf_check_launch();
}

// y = act( ((z - center) * scale + shift) * mask[n][c] + residual ); center may be null (= 0)
template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ z, T* __restrict__ y,
                                                         const float* __restrict__ center,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         const T* __restrict__ res, const float* __restrict__ mask,
                                                         long long total4, int C, long long HW, int relu) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long p = i / C4;
        f32x4 v = load4<T>(z + i * 4);
        const f32x4 sc = *(const f32x4*)(scale + c), sh = *(const f32x4*)(shift + c);
        if (center) {
            const f32x4 ce = *(const f32x4*)(center + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] -= ce[e];
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * sc[e] + sh[e];
        if (mask) {
            const f32x4 mk = *(const f32x4*)(mask + (p / HW) * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mk[e];
        }
        if (res) {
            const f32x4 r = load4<T>(res + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r[e];
        }
        if (relu)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
        store4<T>(y + i * 4, v);
    }
}

// The same on 16-byte chunks (8 bf16 / 4 fp32 channels) with 32-bit index arithmetic: the form every launch of the
// training step takes (C a multiple of the chunk, fewer than 2^31 chunks); the kernel above stays for other shapes.
template <typename T>
__global__ __launch_bounds__(256) void affine_act_vec_kernel(const i32x4* __restrict__ z, i32x4* __restrict__ y,
                                                             const float* __restrict__ center, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const i32x4* __restrict__ res,
                                                             const float* __restrict__ mask, uint32_t total, int C, FastDiv fCv,
                                                             FastDiv fHW, int relu, unsigned char* __restrict__ gate_out) {
    constexpr int E = ChunkOps<T>::N;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t p = fast_div(i, fCv);
        const int c = (int)(i - p * fCv.div) * E;
        const i32x4 zr = z[i];
        i32x4 rr = {0, 0, 0, 0};
        if (res) rr = res[i];
        float v[E], a[E], b[E];
        ChunkOps<T>::unpack(zr, v);
#pragma unroll
        for (int e = 0; e < E; e += 4) {
            *(f32x4*)(a + e) = *(const f32x4*)(scale + c + e);
            *(f32x4*)(b + e) = *(const f32x4*)(shift + c + e);
        }
        if (center) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] -= center[c + e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = v[e] * a[e] + b[e];
        if (mask) {
            const float* mk = mask + (size_t)fast_div(p, fHW) * C + c;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] *= mk[e];
        }
        if (res) {
            float r[E];
            ChunkOps<T>::unpack(rr, r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const i32x4 packed = ChunkOps<T>::pack(v);
        y[i] = packed;
        if (gate_out) {
            // one byte per 16-byte chunk: bit e = [stored value e > 0], what a ReLU backward needs of this tensor -- taken from
            // the ROUNDED values, so that it equals the test on the tensor itself
            float w[E];
            ChunkOps<T>::unpack(packed, w);
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < E; ++e) bits |= (w[e] > 0.f ? 1u : 0u) << e;
            gate_out[i] = (unsigned char)bits;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_vec_kernel(const i32x4* __restrict__ g, const i32x4* __restrict__ gate,
                                                               const float* __restrict__ mask, const i32x4* __restrict__ z,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ coef, i32x4* __restrict__ dz,
                                                               i32x4* __restrict__ g_out, uint32_t total, int C, FastDiv fCv,
                                                               FastDiv fHW) {
    constexpr int E = ChunkOps<T>::N;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t p = fast_div(i, fCv);
        const int c = (int)(i - p * fCv.div) * E;
        const i32x4 gr = g[i], zr = z[i];
        i32x4 tr = {0, 0, 0, 0};
        if (gate) tr = gate[i];
        float v[E], zz[E];
        ChunkOps<T>::unpack(gr, v);
        ChunkOps<T>::unpack(zr, zz);
        if (gate) {
            float gt[E];
            ChunkOps<T>::unpack(tr, gt);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = gt[e] > 0.f ? v[e] : 0.f;
        }
        if (g_out) g_out[i] = ChunkOps<T>::pack(v);
        if (mask) {
            const float* mk = mask + (size_t)fast_div(p, fHW) * C + c;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] *= mk[e];
        }
        float o[E];
#pragma unroll
        for (int e = 0; e < E; ++e)
            o[e] = coef[c + e] * (v[e] - coef[C + c + e] - (zz[e] - mean[c + e]) * invstd[c + e] * coef[2 * C + c + e]);
        dz[i] = ChunkOps<T>::pack(o);
    }
}

template <typename T>
static bool vec_shape_ok(long long P, int C, long long HW) {
    constexpr int E = ChunkOps<T>::N;
    return C % E == 0 && P * (C / E) < 0x7FFFFFFFLL && HW > 0 && HW < 0x7FFFFFFFLL;
}

static unsigned ew_grid(long long total) {
    long long g = (total + 255) / 256;
    if (g > 8192) g = 8192;
    return (unsigned)(g < 1 ? 1 : g);
}

int bn_apply_launch(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* res, const float* mask,
                    int dtype, long long P, int C, long long HW, int relu, hipStream_t stream, unsigned char* gate_out) {
    if (C & 3) return ERF_EUNSUPPORTED;
    if (dtype == 1 ? vec_shape_ok<bf16raw>(P, C, HW) : vec_shape_ok<float>(P, C, HW)) {
        const int cv = C / (dtype == 1 ? 8 : 4);
        const long long total = P * cv;
        const FastDiv fCv = make_fastdiv(cv), fHW = make_fastdiv((int)HW);
        if (dtype == 1)
            hipLaunchKernelGGL(affine_act_vec_kernel<bf16raw>, dim3(ew_grid(total)), dim3(256), 0, stream, (const i32x4*)z, (i32x4*)y,
                               center, scale, shift, (const i32x4*)res, mask, (uint32_t)total, C, fCv, fHW, relu, gate_out);
        else
            hipLaunchKernelGGL(affine_act_vec_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, stream, (const i32x4*)z, (i32x4*)y,
                               center, scale, shift, (const i32x4*)res, mask, (uint32_t)total, C, fCv, fHW, relu, gate_out);
        return erf_check_launch();
    }
    if (gate_out) return ERF_EUNSUPPORTED;            // gate bytes are per 16-byte chunk: the chunked kernel only
    const long long total4 = P * (C / 4);
    if (dtype == 1)
        hipLaunchKernelGGL(affine_act_kernel<bf16raw>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const bf16raw*)z,
                           (bf16raw*)y, center, scale, shift, (const bf16raw*)res, mask, total4, C, HW, relu);
    else
        hipLaunchKernelGGL(affine_act_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const float*)z, (float*)y,
                           center, scale, shift, (const float*)res, mask, total4, C, HW, relu);
    return erf_check_launch();
}

int bn_bwd_reduce_launch(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                         const float* invstd, int dtype, long long P, int C, long long HW, float* partials, int* groups,
                         hipStream_t stream) {
    if (!chan_ok(C)) return ERF_EUNSUPPORTED;
    const int gr = red_groups(P, C);
    *groups = gr;
    if (dtype == 1)
        hipLaunchKernelGGL((chan_reduce_kernel<bf16raw, 1>), dim3(gr), dim3(256), 256 * 8 * sizeof(double), stream,
                           (const bf16raw*)g, (const bf16raw*)gate, mask, (const bf16raw*)z, mean, invstd, P, C, HW, partials);
    else
        hipLaunchKernelGGL((chan_reduce_kernel<float, 1>), dim3(gr), dim3(256), 256 * 8 * sizeof(double), stream,
                           (const float*)g, (const float*)gate, mask, (const float*)z, mean, invstd, P, C, HW, partials);
    return erf_check_launch();
}

// dgamma, dbeta and the three per-channel coefficients of dz = a * (dy - b - xhat * c)
__global__ void bn_bwd_finalize_kernel(const double* __restrict__ partials, int groups, int C, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ invstd, int batch_stats,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
    double s, sx;
    reduce_partials(partials, groups, C, s, sx);
    const int c = blockIdx.x * FIN_CB + threadIdx.x;
    if (threadIdx.x < FIN_CB && c < C) {
        dbeta[c] = (float)s;
        dgamma[c] = (float)sx;
        coef[c] = gamma[c] * invstd[c];
        coef[C + c] = batch_stats ? (float)(s / count) : 0.f;
        coef[2 * C + c] = batch_stats ? (float)(sx / count) : 0.f;
    }
}

int bn_bwd_finalize_launch(const float* partials, int groups, int C, long long count, const float* gamma,
                           const float* invstd, int batch_stats, float* dgamma, float* dbeta, float* coef, hipStream_t stream) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(fin_blocks(C)), dim3(FIN_THREADS), 0, stream, (const double*)partials, groups, C, (double)count, gamma,
                       invstd, batch_stats, dgamma, dbeta, coef);
    return erf_check_launch();
}

// dy = g * [gate > 0] * mask;  dz = a * (dy - b - xhat * c);  optionally g_out = g * [gate > 0]
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ g, const T* __restrict__ gate,
                                                           const float* __restrict__ mask, const T* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ coef, T* __restrict__ dz,
                                                           T* __restrict__ g_out, long long total4, int C, long long HW) {
    const int C4 = C >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4) * 4;
        const long long p = i / C4;
        f32x4 v = load4<T>(g + i * 4);
        if (gate) {
            const f32x4 gt = load4<T>(gate + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gt[e] > 0.f ? v[e] : 0.f;
        }
        if (g_out) store4<T>(g_out + i * 4, v);
        if (mask) {
            const f32x4 mk = *(const f32x4*)(mask + (p / HW) * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= mk[e];
        }
        const f32x4 zz = load4<T>(z + i * 4);
        const f32x4 mu = *(const f32x4*)(mean + c), is = *(const f32x4*)(invstd + c);
        const f32x4 a = *(const f32x4*)(coef + c), b = *(const f32x4*)(coef + C + c), cc = *(const f32x4*)(coef + 2 * C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * (v[e] - b[e] - (zz[e] - mu[e]) * is[e] * cc[e]);
        store4<T>(dz + i * 4, o);
    }
}

int bn_bwd_apply_launch(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                        const float* invstd, const float* coef, void* dz, void* g_out, int dtype, long long P, int C,
                        long long HW, hipStream_t stream) {
    if (C & 3) return ERF_EUNSUPPORTED;
    if (dtype == 1 ? vec_shape_ok<bf16raw>(P, C, HW) : vec_shape_ok<float>(P, C, HW)) {
        const int cv = C / (dtype == 1 ? 8 : 4);
        const long long total = P * cv;
        const FastDiv fCv = make_fastdiv(cv), fHW = make_fastdiv((int)HW);
        if (dtype == 1)
            hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<bf16raw>, dim3(ew_grid(total)), dim3(256), 0, stream, (const i32x4*)g,
                               (const i32x4*)gate, mask, (const i32x4*)z, mean, invstd, coef, (i32x4*)dz, (i32x4*)g_out,
                               (uint32_t)total, C, fCv, fHW);
        else
            hipLaunchKernelGGL(bn_bwd_apply_vec_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, stream, (const i32x4*)g,
                               (const i32x4*)gate, mask, (const i32x4*)z, mean, invstd, coef, (i32x4*)dz, (i32x4*)g_out,
                               (uint32_t)total, C, fCv, fHW);
        return erf_check_launch();
    }
    const long long total4 = P * (C / 4);
    if (dtype == 1)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16raw>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const bf16raw*)g,
                           (const bf16raw*)gate, mask, (const bf16raw*)z, mean, invstd, coef, (bf16raw*)dz, (bf16raw*)g_out,
                           total4, C, HW);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const float*)g,
                           (const float*)gate, mask, (const float*)z, mean, invstd, coef, (float*)dz, (float*)g_out, total4,
                           C, HW);
    return erf_check_launch();
}

// out = g * [gate > 0]  (ReLU backward on its own, for the conv -> ReLU -> conv links)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ g, const T* __restrict__ gate, T* __restrict__ out,
                                                       long long total4) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        f32x4 v = load4<T>(g + i * 4);
        const f32x4 gt = load4<T>(gate + i * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gt[e] > 0.f ? v[e] : 0.f;
        store4<T>(out + i * 4, v);
    }
}

int relu_bwd_launch(const void* g, const void* gate, void* out, int dtype, long long total, hipStream_t stream) {
    if (total & 3) return ERF_EUNSUPPORTED;
    const long long total4 = total / 4;
    if (dtype == 1)
        hipLaunchKernelGGL(relu_bwd_kernel<bf16raw>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const bf16raw*)g,
                           (const bf16raw*)gate, (bf16raw*)out, total4);
    else
        hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(ew_grid(total4)), dim3(256), 0, stream, (const float*)g,
                           (const float*)gate, (float*)out, total4);
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// Dropout2d mask: mask[n][c] = Bernoulli(1 - p) / (1 - p), counter-based hash of (seed, index).
__global__ void dropout_mask_kernel(float* __restrict__ mask, int count, float p, unsigned long long seed) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f);        // [0,1)
    mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

// The masks of all Dropout2d layers of a forward pass in one launch: segment s covers mask[begin[s] .. begin[s + 1]) with its
// own probability and seed; an element's value is what dropout_mask_kernel gives for (seed[s], index within the segment).
struct DropoutTable { int n; int begin[ERF_DROPOUT_MAX + 1]; float p[ERF_DROPOUT_MAX]; unsigned long long seed[ERF_DROPOUT_MAX]; };

__global__ void dropout_mask_many_kernel(float* __restrict__ mask, const DropoutTable t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.begin[t.n]) return;
    int s = 0;
    while (s + 1 < t.n && i >= t.begin[s + 1]) ++s;
    const float p = t.p[s];
    unsigned long long x = t.seed[s] + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i - t.begin[s] + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
    mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

int dropout_mask_many_launch(float* mask, int segments, const int* counts, const float* p, const unsigned long long* seeds,
                             hipStream_t stream) {
    if (segments < 1 || segments > ERF_DROPOUT_MAX) return ERF_EINVAL;
    DropoutTable t{};
    t.n = segments;
    long long total = 0;
    for (int s = 0; s < segments; ++s) {
        if (counts[s] <= 0 || p[s] < 0.f || p[s] >= 1.f) return ERF_EINVAL;
        t.begin[s] = (int)total; t.p[s] = p[s]; t.seed[s] = seeds[s];
        total += counts[s];
        if (total > 0x7FFFFFFF) return ERF_EINVAL;
    }
    t.begin[segments] = (int)total;
    hipLaunchKernelGGL(dropout_mask_many_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mask, t);
    return erf_check_launch();
}

// The same masks with the seeds kept on the device (state[0] = base, state[1] = number of masks drawn so far): segment s
// uses seed = base + (state[1] + s + 1) * K, then the count moves on by the number of segments.  Nothing in the launch
// depends on host values that change from step to step, so a captured graph can replay it.
struct DropoutTableC { int n; int begin[ERF_DROPOUT_MAX + 1]; float p[ERF_DROPOUT_MAX]; };

__global__ void dropout_mask_many_counter_kernel(float* __restrict__ mask, const DropoutTableC t,
                                                 const unsigned long long* __restrict__ state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= t.begin[t.n]) return;
    int s = 0;
    while (s + 1 < t.n && i >= t.begin[s + 1]) ++s;
    const float p = t.p[s];
    const unsigned long long seed = state[0] + (state[1] + (unsigned long long)(s + 1)) * 0xD1B54A32D192ED03ULL;
    unsigned long long x = seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(i - t.begin[s] + 1);
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ULL;
    x ^= x >> 27; x *= 0x94D049BB133111EBULL;
    x ^= x >> 31;
    const float u = (float)(x >> 40) * (1.0f / 16777216.0f);
    mask[i] = u >= p ? 1.0f / (1.0f - p) : 0.f;
}

__global__ void counter_add_kernel(unsigned long long* state, unsigned long long v) { state[1] += v; }

int dropout_mask_many_counter_launch(float* mask, int segments, const int* counts, const float* p, unsigned long long* state,
                                     hipStream_t stream) {
    if (segments < 1 || segments > ERF_DROPOUT_MAX) return ERF_EINVAL;
    DropoutTableC t{};
    t.n = segments;
    long long total = 0;
    for (int s = 0; s < segments; ++s) {
        if (counts[s] <= 0 || p[s] < 0.f || p[s] >= 1.f) return ERF_EINVAL;
        t.begin[s] = (int)total; t.p[s] = p[s];
        total += counts[s];
        if (total > 0x7FFFFFFF) return ERF_EINVAL;
    }
    t.begin[segments] = (int)total;
    hipLaunchKernelGGL(dropout_mask_many_counter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, mask, t, state);
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, stream, state, (unsigned long long)segments);
    return erf_check_launch();
}

int dropout_mask_launch(float* mask, int count, float p, unsigned long long seed, hipStream_t stream) {
    if (count <= 0 || p < 0.f || p >= 1.f) return ERF_EINVAL;
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, mask, count, p, seed);
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// Max-pool 2x2 backward, accumulated into dx: dx[argmax of window] += dz[.., zoff + c].
// Ties go to the first maximum in row-major window order (torch's max_pool2d_with_indices).
template <typename T>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dz, T* __restrict__ dx,
                                                       int N, int Hin, int Win, int Cin, int ldz, int zoff, int accumulate) {
    const int Ho = Hin >> 1, Wo = Win >> 1, C4 = Cin >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C4) * 4;
        const long long pix = idx / C4;
        const int wo = (int)(pix % Wo);
        const long long rest = pix / Wo;
        const int ho = (int)(rest % Ho);
        const int n = (int)(rest / Ho);
        const long long o00 = (((long long)n * Hin + 2 * ho) * Win + 2 * wo) * Cin + c;
        const long long offs[4] = {o00, o00 + Cin, o00 + (long long)Win * Cin, o00 + (long long)Win * Cin + Cin};
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = load4<T>(x + offs[k]);
        const f32x4 g = load4<T>(dz + pix * ldz + zoff + c);
        f32x4 o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int best = 0;
            float bv = v[0][e];
#pragma unroll
            for (int k = 1; k < 4; ++k)
                if (v[k][e] > bv) { bv = v[k][e]; best = k; }
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k][e] = best == k ? g[e] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (accumulate) {
                const f32x4 old = load4<T>(dx + offs[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[k][e] += old[e];
            }
            store4<T>(dx + offs[k], o[k]);
        }
    }
}

int pool_bwd_launch(const void* x, const void* dz, void* dx, int dtype, int N, int Hin, int Win, int Cin, int ldz, int zoff,
                    int accumulate, hipStream_t stream) {
    if (((Hin | Win) & 1) || (Cin & 3) || (ldz & 3) || (zoff & 3)) return ERF_EINVAL;
    const long long total = (long long)N * (Hin / 2) * (Win / 2) * (Cin / 4);
    if (dtype == 1)
        hipLaunchKernelGGL(pool_bwd_kernel<bf16raw>, dim3(ew_grid(total)), dim3(256), 0, stream, (const bf16raw*)x,
                           (const bf16raw*)dz, (bf16raw*)dx, N, Hin, Win, Cin, ldz, zoff, accumulate);
    else
        hipLaunchKernelGGL(pool_bwd_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, stream, (const float*)x,
                           (const float*)dz, (float*)dx, N, Hin, Win, Cin, ldz, zoff, accumulate);
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// Weighted cross-entropy over NCHW logits (one thread per pixel, class planes read coalesced).
// fwd: partial (sum w[t]*nll, sum w[t]) per block -> finalize writes {num, den, num/den}.
// bwd: dlogits[c] = coefficient * w[t] * (softmax[c] - [c == t]), coefficient read from device memory.
// NC > 0: the class count is a compile-time constant and a pixel's logits are read once into registers
// (NC independent loads in flight per thread); NC == 0: any class count, three passes over the planes.
template <typename T, int BWD, int NC>
__global__ __launch_bounds__(256) void ce_kernel(const T* __restrict__ z, const long long* __restrict__ target,
                                                 const float* __restrict__ weight, int C, long long HW, long long total,
                                                 float* __restrict__ partials, const float* __restrict__ coef,
                                                 T* __restrict__ dz) {
    __shared__ float rs[256], rw[256];
    float acc_l = 0.f, acc_w = 0.f;
    const float k = BWD ? coef[0] : 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, hw = idx % HW;
        const T* p = z + n * C * HW + hw;
        const long long t = target[idx];
        // torch's CrossEntropyLoss raises on a target outside [0, C) other than its ignore_index (-100); a kernel cannot raise
        // without a host round trip, so such a pixel gets a NaN weight: the loss and the gradients turn NaN at once instead of
        // training on silently with a different denominator
        const bool inrange = t >= 0 && t < C;
        const bool valid = inrange || t != -100;
        const float w = inrange ? (weight ? weight[t] : 1.f) : (t == -100 ? 0.f : __builtin_nanf(""));
        if constexpr (NC > 0) {
            float v[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = Elem<T>::ld(p + c * HW);
            float m = v[0];
#pragma unroll
            for (int c = 1; c < NC; ++c) m = fmaxf(m, v[c]);
            float se = 0.f, zt = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                zt = (c == t) ? v[c] : zt;
                v[c] = __expf(v[c] - m);
                se += v[c];
            }
            if (!BWD) {
                if (valid) {
                    acc_l += w * (m + __logf(se) - zt);
                    acc_w += w;
                }
            } else {
                const float inv = 1.f / se;
                T* q = dz + n * C * HW + hw;
#pragma unroll
                for (int c = 0; c < NC; ++c) Elem<T>::st(q + c * HW, k * w * (v[c] * inv - (c == t ? 1.f : 0.f)));
            }
            continue;
        }
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, Elem<T>::ld(p + c * HW));
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += __expf(Elem<T>::ld(p + c * HW) - m);
        if (!BWD) {
            if (valid) {
                const float zt = inrange ? Elem<T>::ld(p + t * HW) : 0.f;
                acc_l += w * (m + __logf(se) - zt);
                acc_w += w;
            }
        } else {
            const float inv = 1.f / se;
            T* q = dz + n * C * HW + hw;
            for (int c = 0; c < C; ++c) {
                const float sm = __expf(Elem<T>::ld(p + c * HW) - m) * inv;
                Elem<T>::st(q + c * HW, k * w * (sm - (c == t ? 1.f : 0.f)));
            }
        }
    }
    if (!BWD) {
        rs[threadIdx.x] = acc_l; rw[threadIdx.x] = acc_w;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) { rs[threadIdx.x] += rs[threadIdx.x + s]; rw[threadIdx.x] += rw[threadIdx.x + s]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) { partials[2 * blockIdx.x] = rs[0]; partials[2 * blockIdx.x + 1] = rw[0]; }
    }
}

__global__ __launch_bounds__(1024) void ce_finalize_kernel(const float* __restrict__ partials, int groups, float* __restrict__ out) {
    // 1024 threads: thread t sums groups t, t + 1024, ... in double; a fixed-order butterfly per wave, then the 16 waves
    // meet in LDS in wave order
    __shared__ double sh[16][2];
    double a = 0, b = 0;
    for (int g = threadIdx.x; g < groups; g += 1024) {
        const float2 v = *(const float2*)(partials + 2 * g);
        a += v.x; b += v.y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if ((threadIdx.x & 63) == 0) { sh[threadIdx.x >> 6][0] = a; sh[threadIdx.x >> 6][1] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = 0; b = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { a += sh[w][0]; b += sh[w][1]; }
        out[0] = (float)a; out[1] = (float)b; out[2] = (float)(a / b);
    }
}

int ce_partials_finalize_launch(const float* partials, long long groups, float* out3, hipStream_t stream) {
    if (groups <= 0 || groups > 0x7FFFFFFFLL) return ERF_EINVAL;
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, (int)groups, out3);
    return erf_check_launch();
}

size_t ce_workspace_bytes() { return (size_t)RED_MAX_GROUPS * 2 * sizeof(float); }

int ce_fwd_launch(const void* z, int dtype, const long long* target, const float* weight, int N, int C, long long HW,
                  float* partials, float* out, hipStream_t stream) {
    const long long total = (long long)N * HW;
    long long g = (total + 256 * 4 - 1) / (256 * 4);
    if (g > RED_MAX_GROUPS) g = RED_MAX_GROUPS;
    if (g < 1) g = 1;
#define CE_FWD(TT, NCV)                                                                                                    \
    hipLaunchKernelGGL((ce_kernel<TT, 0, NCV>), dim3((unsigned)g), dim3(256), 0, stream, (const TT*)z, target, weight, C, HW, \
                       total, partials, (const float*)nullptr, (TT*)nullptr)
    if (dtype == 1) { if (C == 20) CE_FWD(bf16raw, 20); else if (C == 6) CE_FWD(bf16raw, 6); else CE_FWD(bf16raw, 0); }
    else { if (C == 20) CE_FWD(float, 20); else if (C == 6) CE_FWD(float, 6); else CE_FWD(float, 0); }
#undef CE_FWD
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(1024), 0, stream, partials, (int)g, out);
    return erf_check_launch();
}

int ce_bwd_launch(const void* z, int dtype, const long long* target, const float* weight, int N, int C, long long HW,
                  const float* coef, void* dz, hipStream_t stream) {
    const long long total = (long long)N * HW;
#define CE_BWD(TT, NCV)                                                                                                      \
    hipLaunchKernelGGL((ce_kernel<TT, 1, NCV>), dim3(ew_grid(total)), dim3(256), 0, stream, (const TT*)z, target, weight, C, HW, \
                       total, (float*)nullptr, coef, (TT*)dz)
    if (dtype == 1) { if (C == 20) CE_BWD(bf16raw, 20); else if (C == 6) CE_BWD(bf16raw, 6); else CE_BWD(bf16raw, 0); }
    else { if (C == 20) CE_BWD(float, 20); else if (C == 6) CE_BWD(float, 6); else CE_BWD(float, 0); }
#undef CE_BWD
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// Fused optimiser steps over a flat fp32 arena (4 elements per thread).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float b1, float omb1,
                                                   float b2, float omb2, float eps, float wd, float bc2_sqrt, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;                 // coupled L2 decay (torch.optim.Adam)
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

int adam_launch(float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2, double eps,
                double wd, int step, double gscale, hipStream_t stream) {
    if (n <= 0 || step < 1) return ERF_EINVAL;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2s = sqrt(1.0 - pow(b2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, m, v, n, (float)(lr / bc1), (float)b1,
                       (float)(1.0 - b1), (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, (float)bc2s, (float)gscale);
    return erf_check_launch();
}

// Adam with the step count and the learning rate in device memory.  adam_advance_kernel (one thread) moves the count on and
// leaves the two step-dependent factors in hyper[0..1]; adam_hyper_kernel reads them.  The arithmetic per element is
// adam_kernel's.
__global__ void adam_advance_kernel(int* step, const double* lr, float* hyper, double b1, double b2) {
    const int s = *step + 1;
    *step = s;
    hyper[0] = (float)(*lr / (1.0 - pow(b1, (double)s)));
    hyper[1] = (float)sqrt(1.0 - pow(b2, (double)s));
}

__global__ __launch_bounds__(256) void adam_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                         float* __restrict__ v, long long n, const float* __restrict__ hyper, float b1,
                                                         float omb1, float b2, float omb2, float eps, float wd, float gscale) {
    const float step_size = hyper[0], bc2_sqrt = hyper[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float mi = b1 * m[i] + omb1 * gi;
        const float vi = b2 * v[i] + omb2 * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

int adam_advance_launch(int* step, const double* lr, float* hyper, double b1, double b2, hipStream_t stream) {
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, stream, step, lr, hyper, b1, b2);
    return erf_check_launch();
}

int adam_hyper_launch(float* p, const float* g, float* m, float* v, long long n, const float* hyper, double b1, double b2,
                      double eps, double wd, double gscale, hipStream_t stream) {
    if (n <= 0) return ERF_EINVAL;
    hipLaunchKernelGGL(adam_hyper_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, m, v, n, hyper, (float)b1, (float)(1.0 - b1),
                       (float)b2, (float)(1.0 - b2), (float)eps, (float)wd, (float)gscale);
    return erf_check_launch();
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                  long long n, float lr, float momentum, float wd, int first, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float bi = first ? gi : momentum * buf[i] + gi;
        buf[i] = bi;
        p[i] = pi - lr * bi;
    }
}

int sgd_launch(float* p, const float* g, float* buf, long long n, double lr, double momentum, double wd, int first,
               double gscale, hipStream_t stream) {
    if (n <= 0) return ERF_EINVAL;
    hipLaunchKernelGGL(sgd_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, buf, n, (float)lr, (float)momentum, (float)wd,
                       first, (float)gscale);
    return erf_check_launch();
}

// SGD with the learning rate and the weight decay in device memory (hyper[0], hyper[1]): a captured step replays with
// whatever values the schedule has stored there.  The arithmetic per element is sgd_kernel's.
__global__ __launch_bounds__(256) void sgd_hyper_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                        long long n, const float* __restrict__ hyper, float momentum, int first,
                                                        float gscale) {
    const float lr = hyper[0], wd = hyper[1];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] * gscale + wd * pi;
        const float bi = first ? gi : momentum * buf[i] + gi;
        buf[i] = bi;
        p[i] = pi - lr * bi;
    }
}

int sgd_hyper_launch(float* p, const float* g, float* buf, long long n, const float* hyper, double momentum, int first,
                     double gscale, hipStream_t stream) {
    if (n <= 0) return ERF_EINVAL;
    hipLaunchKernelGGL(sgd_hyper_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, p, g, buf, n, hyper, (float)momentum, first,
                       (float)gscale);
    return erf_check_launch();
}

// --------------------------------------------------------------------------------------------
// dlogits NCHW [N,nc,2H,2W] -> "patch" NHWC [N,H,W,ld] (channel = 4*class + phase, phase = 2a+b
// the position inside the 2x2 output patch; channels >= 4*nc zero), the layout output_conv's
// backward GEMMs take.  `ncp` = ld / 4.
// S = 1 is the degenerate patch: channel = class (Encoder.output_conv, the 1x1 head).
template <typename TS> __device__ __forceinline__ void ld_pair(const TS* p, float& a, float& b);
template <> __device__ __forceinline__ void ld_pair<float>(const float* p, float& a, float& b) {
    const f32x2 v = *(const f32x2*)p;
    a = v[0]; b = v[1];
}
template <> __device__ __forceinline__ void ld_pair<bf16raw>(const bf16raw* p, float& a, float& b) {
    const uint32_t v = *(const uint32_t*)p;
    a = __builtin_bit_cast(float, v << 16); b = __builtin_bit_cast(float, v & 0xFFFF0000u);
}

template <typename TS, typename T, int S>
__global__ __launch_bounds__(256) void patches_kernel(const TS* __restrict__ dl, T* __restrict__ out, int N, int H, int W,
                                                      int nc, int ncp, int ld) {
    // one block = 64 consecutive pixels of the [N,H,W] grid; tile[px][v] in LDS (odd pitch: the transposing
    // accesses below hit distinct banks)
    extern __shared__ float tile[];                 // [64][ld + 1]
    const long long P = (long long)N * H * W;
    const long long pix0 = (long long)blockIdx.x * 64;
    const int ldt = ld + 1;
    // read: a thread keeps its pixel (lane -> consecutive columns: full-line reads of the planes) and walks over
    // (class, patch row); both columns of the patch row come with one 8-byte load
    const int px = threadIdx.x & 63;
    const long long pix = pix0 + px;
    const bool live = pix < P;
    const int j = (int)(pix % W);
    const long long rest = pix / W;
    const int i = (int)(rest % H);
    const long long n = rest / H;
    const long long plane = (long long)(S * H) * (S * W);
    const TS* src = dl + n * nc * plane + (long long)(S * i) * (S * W) + S * j;
    const int items = ld / S;                       // (class, patch row) pairs incl. the zero padding
    for (int it = threadIdx.x >> 6; it < items; it += 4) {
        const int cls = S == 2 ? it >> 1 : it, a = S == 2 ? it & 1 : 0;
        float v0 = 0.f, v1 = 0.f;
        if (live && cls < nc) {
            if (S == 2) ld_pair<TS>(src + cls * plane + (long long)a * (S * W), v0, v1);
            else v0 = Elem<TS>::ld(src + cls * plane);
        }
        tile[px * ldt + it * S] = v0;
        if (S == 2) tile[px * ldt + it * S + 1] = v1;
    }
    __syncthreads();
    // write: the 64 pixels x ld channels are one contiguous run of the NHWC tensor; 4 channels per thread
    const int ld4 = ld >> 2;
    for (int id = threadIdx.x; id < 64 * ld4; id += 256) {
        const int p4 = id / ld4, v = (id % ld4) * 4;
        if (pix0 + p4 < P) {
            const float* t = tile + p4 * ldt + v;
            store4<T>(out + (pix0 + p4) * ld + v, f32x4{t[0], t[1], t[2], t[3]});
        }
    }
}

int patches_launch(const void* dl, int src_dtype, void* out, int dtype, int N, int H, int W, int nc, int ncp, int ld,
                   hipStream_t stream, int patch_edge) {
    const long long blocks = ((long long)N * H * W + 63) / 64;
    if (blocks > 0x7FFFFFFFLL || (patch_edge != 1 && patch_edge != 2)) return ERF_EINVAL;
    const size_t lds = (size_t)64 * (ld + 1) * sizeof(float);
#define PK2(TS, T, S) hipLaunchKernelGGL((patches_kernel<TS, T, S>), dim3((unsigned)blocks), dim3(256), lds, stream, (const TS*)dl, (T*)out, N, H, W, nc, ncp, ld)
#define PK(TS, T) do { if (patch_edge == 2) PK2(TS, T, 2); else PK2(TS, T, 1); } while (0)
    if (src_dtype == 0 && dtype == 0) PK(float, float);
    else if (src_dtype == 0 && dtype == 1) PK(float, bf16raw);
    else if (src_dtype == 1 && dtype == 1) PK(bf16raw, bf16raw);
    else if (src_dtype == 1 && dtype == 0) PK(bf16raw, float);
    else return ERF_EUNSUPPORTED;
#undef PK
#undef PK2
    return erf_check_launch();
}

// image NCHW [N,3,H,W] -> NHWC [N,H,W,cpad] (channels >= 3 zero), operand of the stem's weight gradient
// uint8 sources (LAY 0: planar, 1: interleaved [N,H,W,3]) become value / 255 (correctly rounded fp32 division), as in stem_kernel
template <typename TS, typename T, int LAY = 0>
__global__ __launch_bounds__(256) void image_pad_kernel(const TS* __restrict__ x, T* __restrict__ out, long long N, long long HW,
                                                        int cpad) {
    const long long total = N * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long n = idx / HW, hw = idx % HW;
        for (int c = 0; c < cpad; ++c) {
            float v = 0.f;
            if (c < 3) {
                if constexpr (sizeof(TS) == 1) v = (float)x[LAY ? idx * 3 + c : (n * 3 + c) * HW + hw] / 255.0f;
                else v = Elem<TS>::ld(x + (n * 3 + c) * HW + hw);
            }
            Elem<T>::st(out + idx * cpad + c, v);
        }
    }
}

int image_pad_launch(const void* x, int src_dtype, void* out, int dtype, int N, long long HW, int cpad, hipStream_t stream) {
    const long long total = (long long)N * HW;
#define IP(TS, T) hipLaunchKernelGGL((image_pad_kernel<TS, T>), dim3(ew_grid(total)), dim3(256), 0, stream, (const TS*)x, (T*)out, (long long)N, HW, cpad)
    if (src_dtype == 0 && dtype == 0) IP(float, float);
    else if (src_dtype == 0 && dtype == 1) IP(float, bf16raw);
    else if (src_dtype == 1 && dtype == 1) IP(bf16raw, bf16raw);
    else if (src_dtype == 1 && dtype == 0) IP(bf16raw, float);
#define IP8(T, LAY) hipLaunchKernelGGL((image_pad_kernel<uint8_t, T, LAY>), dim3(ew_grid(total)), dim3(256), 0, stream, (const uint8_t*)x, (T*)out, (long long)N, HW, cpad)
    else if (src_dtype == 2 && dtype == 0) IP8(float, 0);
    else if (src_dtype == 2 && dtype == 1) IP8(bf16raw, 0);
    else if (src_dtype == 3 && dtype == 0) IP8(float, 1);
    else if (src_dtype == 3 && dtype == 1) IP8(bf16raw, 1);
#undef IP8
    else return ERF_EUNSUPPORTED;
#undef IP
    return erf_check_launch();
}
