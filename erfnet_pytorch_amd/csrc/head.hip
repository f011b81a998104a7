// Classifier head of the ImageNet pre-training variant (train/erfnet_imagenet.py:97-120 of the reference):
//   features -> MaxPool2d(2, stride 2) -> AvgPool2d(14, 1, 0) -> view(N, C) -> Linear(C, K)
// on an NHWC feature map of exactly 28 x 28 pixels (the only size for which the reference's view(N, 128)
// is valid).  All four kernels are tiny (N*C = 4096 outputs, K = 1000) and bandwidth/latency-bound.
#include "common.cuh"

// feat[n][c] = mean over the H/2 x W/2 pooled positions of max over each 2x2 window (fp32 output)
template <typename T>
__global__ __launch_bounds__(256) void pool_mean_fwd_kernel(const T* __restrict__ x, float* __restrict__ feat, int N, int H, int W,
                                                            int C) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * C) return;
    const int c = idx % C, n = idx / C;
    const T* p = x + (long long)n * H * W * C + c;
    float s = 0.f;
    for (int i = 0; i < H / 2; ++i)
        for (int j = 0; j < W / 2; ++j) {
            const T* q = p + ((long long)(2 * i) * W + 2 * j) * C;
            const float a = Elem<T>::ld(q), b = Elem<T>::ld(q + C), cc = Elem<T>::ld(q + (long long)W * C), d = Elem<T>::ld(q + (long long)W * C + C);
            s += fmaxf(fmaxf(a, b), fmaxf(cc, d));
        }
    feat[idx] = s / (float)((H / 2) * (W / 2));
}

// feat[n][c] = mean over all H x W positions (fp32 output)
template <typename T>
__global__ __launch_bounds__(1024) void spatial_mean_kernel(const T* __restrict__ x, float* __restrict__ feat, int N, long long HW, int C) {
    // one block per image; the block's threads form 1024 / C slices of the pixel range x C channels (rows of C channels
    // are read coalesced), and the slices of a channel meet in LDS in slice order: the sum's order is fixed
    __shared__ float part[1024];
    const int n = blockIdx.x;
    const int nsl = C <= 1024 ? 1024 / C : 1;
    const int sl = threadIdx.x / C;
    const long long per = (HW + nsl - 1) / nsl;
    for (int c0 = 0; c0 < C; c0 += 1024) {
        const int c = c0 + (C <= 1024 ? (int)threadIdx.x % C : (int)threadIdx.x);
        float acc = 0.f;
        if (sl < nsl && c < C) {
            const T* p = x + (long long)n * HW * C + c;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            long long i = sl * per;
            const long long i1 = i + per < HW ? i + per : HW;
            for (; i + 3 < i1; i += 4) {
                s0 += Elem<T>::ld(p + i * C); s1 += Elem<T>::ld(p + (i + 1) * C);
                s2 += Elem<T>::ld(p + (i + 2) * C); s3 += Elem<T>::ld(p + (i + 3) * C);
            }
            for (; i < i1; ++i) s0 += Elem<T>::ld(p + i * C);
            acc = (s0 + s1) + (s2 + s3);
        }
        if (nsl == 1) {
            if (c < C) feat[n * C + c] = acc / (float)HW;
            continue;
        }
        part[threadIdx.x] = acc;
        __syncthreads();
        if ((int)threadIdx.x < C) {
            float s = 0.f;
            for (int k = 0; k < nsl; ++k) s += part[k * C + threadIdx.x];
            feat[n * C + threadIdx.x] = s / (float)HW;
        }
        __syncthreads();
    }
}

int spatial_mean_launch(const void* x, int dtype, float* feat, int N, long long HW, int C, hipStream_t stream) {
    if (dtype == 1) hipLaunchKernelGGL(spatial_mean_kernel<bf16raw>, dim3(N), dim3(1024), 0, stream, (const bf16raw*)x, feat, N, HW, C);
    else hipLaunchKernelGGL(spatial_mean_kernel<float>, dim3(N), dim3(1024), 0, stream, (const float*)x, feat, N, HW, C);
    return erf_check_launch();
}

// out[p][0:Ca] = a[p][:], out[p][Ca:Ca+Cb] = b[p][:] for NHWC maps, in 16-byte chunks (the decoders' skip concatenation)
__global__ __launch_bounds__(256) void concat_channels_kernel(const i32x4* __restrict__ a, const i32x4* __restrict__ b,
                                                              i32x4* __restrict__ out, long long pixels, int cha, int chb) {
    const int ch = cha + chb;
    const long long total = pixels * ch;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / ch;
        const int k = (int)(i - p * ch);
        out[i] = k < cha ? a[p * cha + k] : b[p * chb + (k - cha)];
    }
}

int concat_channels_launch(const void* a, int Ca, const void* b, int Cb, void* out, int dtype, long long pixels, hipStream_t stream) {
    const int epc = dtype == 1 ? 8 : 4;
    if (Ca % epc || Cb % epc) return ERF_EUNSUPPORTED;
    const long long total = pixels * ((Ca + Cb) / epc);
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(concat_channels_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const i32x4*)a, (const i32x4*)b,
                       (i32x4*)out, pixels, Ca / epc, Cb / epc);
    return erf_check_launch();
}

// dx = dfeat[n][c] / (#windows) at the first maximum of each 2x2 window (row-major scan, as torch), 0 elsewhere
template <typename T>
__global__ __launch_bounds__(256) void pool_mean_bwd_kernel(const T* __restrict__ x, const float* __restrict__ dfeat, T* __restrict__ dx,
                                                            int N, int H, int W, int C) {
    const long long total = (long long)N * (H / 2) * (W / 2) * C;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % C);
    long long r = idx / C;
    const int j = (int)(r % (W / 2)); r /= (W / 2);
    const int i = (int)(r % (H / 2));
    const int n = (int)(r / (H / 2));
    const long long base = (((long long)n * H + 2 * i) * W + 2 * j) * C + c;
    const long long off[4] = {0, C, (long long)W * C, (long long)W * C + C};
    float best = Elem<T>::ld(x + base);
    int arg = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) {
        const float v = Elem<T>::ld(x + base + off[k]);
        if (v > best) { best = v; arg = k; }
    }
    const float g = dfeat[n * C + c] / (float)((H / 2) * (W / 2));
#pragma unroll
    for (int k = 0; k < 4; ++k) Elem<T>::st(dx + base + off[k], k == arg ? g : 0.f);
}

// y[n][k] = b[k] + sum_c f[n][c] * w[k][c]
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ b,
                                                         float* __restrict__ y, int N, int C, int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N * K) return;
    const int k = idx % K, n = idx / K;
    float s = b ? b[k] : 0.f;
    for (int c = 0; c < C; ++c) s += f[n * C + c] * w[(long long)k * C + c];
    y[idx] = s;
}

// one launch: df[n][c] = sum_k dy[n][k] w[k][c];  dw[k][c] = sum_n dy[n][k] f[n][c];  db[k] = sum_n dy[n][k]
__global__ __launch_bounds__(256) void linear_bwd_kernel(const float* __restrict__ f, const float* __restrict__ w, const float* __restrict__ dy,
                                                         float* __restrict__ df, float* __restrict__ dw, float* __restrict__ db, int N, int C,
                                                         int K) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int n_df = N * C, n_dw = K * C;
    if (idx < n_df) {
        const int c = idx % C, n = idx / C;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += dy[n * K + k] * w[(long long)k * C + c];
        df[idx] = s;
    } else if (idx < n_df + n_dw) {
        const int e = idx - n_df, c = e % C, k = e / C;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[n * K + k] * f[n * C + c];
        dw[e] = s;
    } else if (idx < n_df + n_dw + K) {
        const int k = idx - n_df - n_dw;
        float s = 0.f;
        for (int n = 0; n < N; ++n) s += dy[n * K + k];
        db[k] = s;
    }
}

int pool_mean_fwd_launch(const void* x, int dtype, float* feat, int N, int H, int W, int C, hipStream_t stream) {
    const int total = N * C;
    if (dtype == 1)
        hipLaunchKernelGGL(pool_mean_fwd_kernel<bf16raw>, dim3((total + 255) / 256), dim3(256), 0, stream, (const bf16raw*)x, feat, N, H, W, C);
    else
        hipLaunchKernelGGL(pool_mean_fwd_kernel<float>, dim3((total + 255) / 256), dim3(256), 0, stream, (const float*)x, feat, N, H, W, C);
    return erf_check_launch();
}

int pool_mean_bwd_launch(const void* x, const float* dfeat, void* dx, int dtype, int N, int H, int W, int C, hipStream_t stream) {
    const long long total = (long long)N * (H / 2) * (W / 2) * C;
    const unsigned grid = (unsigned)((total + 255) / 256);
    if (dtype == 1)
        hipLaunchKernelGGL(pool_mean_bwd_kernel<bf16raw>, dim3(grid), dim3(256), 0, stream, (const bf16raw*)x, dfeat, (bf16raw*)dx, N, H, W, C);
    else
        hipLaunchKernelGGL(pool_mean_bwd_kernel<float>, dim3(grid), dim3(256), 0, stream, (const float*)x, dfeat, (float*)dx, N, H, W, C);
    return erf_check_launch();
}

int linear_fwd_launch(const float* f, const float* w, const float* b, float* y, int N, int C, int K, hipStream_t stream) {
    hipLaunchKernelGGL(linear_fwd_kernel, dim3((N * K + 255) / 256), dim3(256), 0, stream, f, w, b, y, N, C, K);
    return erf_check_launch();
}

int linear_bwd_launch(const float* f, const float* w, const float* dy, float* df, float* dw, float* db, int N, int C, int K,
                      hipStream_t stream) {
    const int total = N * C + K * C + K;
    hipLaunchKernelGGL(linear_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, f, w, dy, df, dw, db, N, C, K);
    return erf_check_launch();
}
