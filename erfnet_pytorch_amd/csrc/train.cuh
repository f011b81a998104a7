#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "pack.cuh"
#include "gconv.cuh"

// upper bound on the per-workgroup partial rows of every two-stage reduction (workspaces are sized for it)
#define RED_MAX_GROUPS 512
size_t bn_workspace_bytes(int C);
float* bn_workspace_coef(void* workspace, int C);
int channel_sum_finalize_launch(const float* partials, int groups, int C, float* sums, hipStream_t stream);
#define ERF_CHANNEL_SUM_MAX 48
struct ChannelSumEntry { const double* partials; float* sums; int groups, C; };
struct ChannelSumTable { ChannelSumEntry e[ERF_CHANNEL_SUM_MAX]; int count; };
int channel_sum_finalize_many_launch(const ChannelSumTable& t, hipStream_t stream);
int bn_stats_launch(const void* z, int dtype, long long P, int C, float* partials, int* groups, hipStream_t stream);
int bn_finalize_launch(const float* partials, int groups, int C, long long count, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                       float* scale, float* shift, hipStream_t stream);
int bn_apply_launch(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* res, const float* mask,
                    int dtype, long long P, int C, long long HW, int relu, hipStream_t stream, unsigned char* gate_out = nullptr);
int bn_bwd_reduce_launch(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                         const float* invstd, int dtype, long long P, int C, long long HW, float* partials, int* groups,
                         hipStream_t stream);
int bn_bwd_finalize_launch(const float* partials, int groups, int C, long long count, const float* gamma,
                           const float* invstd, int batch_stats, float* dgamma, float* dbeta, float* coef, hipStream_t stream);
int bn_bwd_apply_launch(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                        const float* invstd, const float* coef, void* dz, void* g_out, int dtype, long long P, int C,
                        long long HW, hipStream_t stream);
int relu_bwd_launch(const void* g, const void* gate, void* out, int dtype, long long total, hipStream_t stream);
int dropout_mask_launch(float* mask, int count, float p, unsigned long long seed, hipStream_t stream);
#define ERF_DROPOUT_MAX 32
int dropout_mask_many_launch(float* mask, int segments, const int* counts, const float* p, const unsigned long long* seeds,
                             hipStream_t stream);
int pool_bwd_launch(const void* x, const void* dz, void* dx, int dtype, int N, int Hin, int Win, int Cin, int ldz, int zoff,
                    int accumulate, hipStream_t stream);
size_t ce_workspace_bytes();
int ce_fwd_launch(const void* z, int dtype, const long long* target, const float* weight, int N, int C, long long HW,
                  float* partials, float* out, hipStream_t stream);
int ce_bwd_launch(const void* z, int dtype, const long long* target, const float* weight, int N, int C, long long HW,
                  const float* coef, void* dz, hipStream_t stream);
int dropout_mask_many_counter_launch(float* mask, int segments, const int* counts, const float* p, unsigned long long* state,
                                     hipStream_t stream);
int adam_advance_launch(int* step, const double* lr, float* hyper, double b1, double b2, hipStream_t stream);
int adam_hyper_launch(float* p, const float* g, float* m, float* v, long long n, const float* hyper, double b1, double b2,
                      double eps, double wd, double gscale, hipStream_t stream);
int adam_launch(float* p, const float* g, float* m, float* v, long long n, double lr, double b1, double b2, double eps,
                double wd, int step, double gscale, hipStream_t stream);
int sgd_launch(float* p, const float* g, float* buf, long long n, double lr, double momentum, double wd, int first,
               double gscale, hipStream_t stream);
int sgd_hyper_launch(float* p, const float* g, float* buf, long long n, const float* hyper, double momentum, int first,
                     double gscale, hipStream_t stream);
int patches_launch(const void* dl, int src_dtype, void* out, int dtype, int N, int H, int W, int nc, int ncp, int ld,
                   hipStream_t stream, int patch_edge = 2);
int image_pad_launch(const void* x, int src_dtype, void* out, int dtype, int N, long long HW, int cpad, hipStream_t stream);

// ---- weight gradients (wgrad.hip): dW[row][tap][ci] = sum_pixels dz[out pixel][row] * x[in pixel(tap)][ci]
// Geometry is described like the forward gather conv (GConvParams: x = layer input, phases/taps),
// `dz` is the gradient at the layer output (NHWC, channel stride lddz, first channel dzoff, `rows`
// channels).  Partials go to `workspace`; the reduction writes fp32 gradients in torch layout
// through the same index map the weight packer reads with (PackParams).
size_t wgrad_workspace_bytes(int rows, int cin, int max_taps_per_phase, long long base_pixels);
int wgrad_launch(const GConvParams& geo, const void* dz, int lddz, int dzoff, int rows, int cin, const PackParams* maps,
                 float* dw, float* workspace, size_t workspace_bytes, int dtype, hipStream_t stream);

// Sum of `nwg` fp32 partial tiles [rows_pad][ncols_pad] (column = tap*cin + ci), scattered into torch weight layout.
struct WgradReduceDesc { const float* partial; float* dw; int nwg, rows_pad, ncols_pad, cin; PackParams m; };
void wgrad_reduce_defer(WgradReduceDesc* slot);
void wgrad_reduce_record_begin(WgradReduceDesc* table, int capacity);
int wgrad_reduce_record_end();
int wgrad_reduce_many_launch(const void* table_host, int count, hipStream_t stream);
int wgrad_reduce_launch(const float* partial, int nwg, int rows_pad, int ncols_pad, int cin, const PackParams& m, float* dw,
                        hipStream_t stream);
// Factorized-convolution weight gradient on lattice tiles (bf16).  ERF_EUNSUPPORTED: use wgrad_launch.
int wgrad1d_launch(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis, int dilation,
                   const PackParams& map, float* workspace, size_t workspace_bytes, hipStream_t stream);
// Up to 4 layers of one shape [N, H, W, C] in one launch (each with its own operands, axis, dilation and workspace of
// workspace_bytes); descs != nullptr: the per-layer reductions are not launched, their arguments are written there.
struct W1dJob { const void* x; const void* dz; float* dw; float* workspace; int axis, dilation; };
int wgrad1d_many_launch(const W1dJob* jobs, int count, int dtype, int N, int H, int W, int C, const PackParams* maps,
                        size_t workspace_bytes, WgradReduceDesc* descs, hipStream_t stream);

// ---- classifier head of the ImageNet variant (head.hip)
int concat_channels_launch(const void* a, int Ca, const void* b, int Cb, void* out, int dtype, long long pixels, hipStream_t stream);
int spatial_mean_launch(const void* x, int dtype, float* feat, int N, long long HW, int C, hipStream_t stream);
int pool_mean_fwd_launch(const void* x, int dtype, float* feat, int N, int H, int W, int C, hipStream_t stream);
int pool_mean_bwd_launch(const void* x, const float* dfeat, void* dx, int dtype, int N, int H, int W, int C, hipStream_t stream);
int linear_fwd_launch(const float* f, const float* w, const float* b, float* y, int N, int C, int K, hipStream_t stream);
int linear_bwd_launch(const float* f, const float* w, const float* dy, float* df, float* dw, float* db, int N, int C, int K,
                      hipStream_t stream);
// (sum w*nll, sum w) partial pairs -> out3 = (num, den, num / den), fixed order
int ce_partials_finalize_launch(const float* partials, long long groups, float* out3, hipStream_t stream);
// Decoder.output_conv fused with the weighted cross entropy (head_loss.hip)
size_t head_ce_workspace_bytes(int N, int Hin, int Win);
int head_ce_fwd_launch(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                       float* out3, float* workspace, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream);
int head_ce_bwd_launch(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                       const float* coef, void* patches, int ld, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream);
