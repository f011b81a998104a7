// extern "C" training entry points of libermi355.so (declared in include/erfnet_mi355.h):
// BatchNorm statistics / backward, data gradients, weight gradients, loss and optimiser.
#include "../../include/erfnet_mi355.h"
#include "common.cuh"
#include "gconv.cuh"
#include "pack.cuh"
#include "misc.cuh"
#include "conv1d.cuh"
#include "train.cuh"

static inline bool dtype_ok(int d) { return d == ERF_F32 || d == ERF_BF16; }
static inline int ct_of(int cout) { return (cout + 15) / 16; }
static const int NO_PHASES = 1 << 30;

static int up_axis_taps(int parity, int th[2], int d[2]) {
    if (parity == 0) { th[0] = 1; d[0] = 0; return 1; }
    th[0] = 0; d[0] = 1; th[1] = 2; d[1] = 0; return 2;
}

// Phase/tap table of a transposed 3x3 stride-2 convolution (padding 1, output_padding 1) written
// as a gather over the INPUT grid: fills geo.ph[0..3]; `wtap[ph][t]` = th*3+tw of tap t.
static void convT3x3_phases(GConvParams& p, int wtap[4][4]) {
    p.nphase = 4;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            GConvPhase& ph = p.ph[a * 2 + b];
            int th[2], dh[2], tw[2], dw[2];
            const int nh = up_axis_taps(a, th, dh), nw = up_axis_taps(b, tw, dw);
            ph.oh = a; ph.ow = b; ph.ntaps = nh * nw;
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) {
                    ph.dh[i * nw + j] = dh[i]; ph.dw[i * nw + j] = dw[j];
                    wtap[a * 2 + b][i * nw + j] = th[i] * 3 + tw[j];
                }
        }
}

extern "C" {

// ------------------------------------------------------------------ BatchNorm2d (training)
size_t erf_bn_workspace_bytes(int C) { return bn_workspace_bytes(C); }

int erf_bn_stats_finalize(const void* z, int dtype, long long pixels, int C, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps, float* mean, float* invstd,
                          float* scale, float* shift, void* workspace, void* stream) {
    if (!z || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace || !dtype_ok(dtype) || pixels <= 0)
        return ERF_EINVAL;
    int groups = 0;
    int rc = bn_stats_launch(z, dtype, pixels, C, (float*)workspace, &groups, (hipStream_t)stream);
    if (rc) return rc;
    return bn_finalize_launch((const float*)workspace, groups, C, pixels, gamma, beta, running_mean, running_var, momentum,
                              eps, mean, invstd, scale, shift, (hipStream_t)stream);
}

int erf_bn_apply(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* residual,
                 const float* mask, int dtype, int N, long long HW, int C, int relu, void* stream) {
    if (!z || !y || !scale || !shift || !dtype_ok(dtype) || N <= 0 || HW <= 0) return ERF_EINVAL;
    return bn_apply_launch(z, y, center, scale, shift, residual, mask, dtype, (long long)N * HW, C, HW, relu, (hipStream_t)stream);
}

// erf_bn_apply that also writes the ReLU gate of its output as one byte per 16-byte chunk (bit e = [y value e > 0]; bf16: 8
// channels per byte, fp32: 4), for the data-gradient launches that would otherwise read y back for its signs only
int erf_bn_apply_gates(const void* z, void* y, const float* center, const float* scale, const float* shift, const void* residual,
                       const float* mask, int dtype, int N, long long HW, int C, int relu, void* gate_out, void* stream) {
    if (!z || !y || !scale || !shift || !gate_out || !dtype_ok(dtype) || N <= 0 || HW <= 0) return ERF_EINVAL;
    return bn_apply_launch(z, y, center, scale, shift, residual, mask, dtype, (long long)N * HW, C, HW, relu, (hipStream_t)stream,
                           (unsigned char*)gate_out);
}

int erf_bn_backward(const void* g, const void* gate, const float* mask, const void* z, const float* mean,
                    const float* invstd, const float* gamma, int batch_stats, void* dz, void* g_out, float* dgamma,
                    float* dbeta, int dtype, int N, long long HW, int C, void* workspace, void* stream) {
    if (!g || !z || !mean || !invstd || !gamma || !dz || !dgamma || !dbeta || !workspace || !dtype_ok(dtype) || N <= 0 || HW <= 0)
        return ERF_EINVAL;
    const long long P = (long long)N * HW;
    float* partials = (float*)workspace;
    float* coef = bn_workspace_coef(workspace, C);
    int groups = 0;
    int rc = bn_bwd_reduce_launch(g, gate, mask, z, mean, invstd, dtype, P, C, HW, partials, &groups, (hipStream_t)stream);
    if (rc) return rc;
    rc = bn_bwd_finalize_launch(partials, groups, C, P, gamma, invstd, batch_stats, dgamma, dbeta, coef, (hipStream_t)stream);
    if (rc) return rc;
    return bn_bwd_apply_launch(g, gate, mask, z, mean, invstd, coef, dz, g_out, dtype, P, C, HW, (hipStream_t)stream);
}

int erf_relu_backward(const void* g, const void* gate, void* out, int dtype, long long count, void* stream) {
    if (!g || !gate || !out || !dtype_ok(dtype) || count <= 0) return ERF_EINVAL;
    return relu_bwd_launch(g, gate, out, dtype, count, (hipStream_t)stream);
}

int erf_channel_sum(const void* z, int dtype, long long pixels, int C, float* sums, void* workspace, void* stream) {
    // sums[c] = sum over pixels of z[p][c]  (bias gradients); reuses the BatchNorm reduction
    if (!z || !sums || !workspace || !dtype_ok(dtype) || pixels <= 0) return ERF_EINVAL;
    int groups = 0;
    int rc = bn_stats_launch(z, dtype, pixels, C, (float*)workspace, &groups, (hipStream_t)stream);
    if (rc) return rc;
    return channel_sum_finalize_launch((const float*)workspace, groups, C, sums, (hipStream_t)stream);
}

// z = conv1d(x) + bias with the BatchNorm batch statistics of z taken inside the convolution's epilogue
int erf_conv1d_factorized_bn(const void* x, void* z, const void* packed, const float* bias, int dtype, int N, int H, int W, int C,
                             int axis, int dilation, const float* gamma, const float* beta, float* running_mean,
                             float* running_var, float momentum, float eps, float* mean, float* invstd, float* scale,
                             float* shift, void* workspace, void* stream) {
    if (!x || !z || !packed || !gamma || !beta || !mean || !invstd || !scale || !shift || !workspace || !dtype_ok(dtype) ||
        N <= 0 || H <= 0 || W <= 0 || dilation < 1 || (axis != ERF_AXIS_H && axis != ERF_AXIS_W))
        return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    int groups = 0;
    int rc = conv1d_buf_launch(x, z, packed, nullptr, bias, nullptr, nullptr, (double*)workspace, &groups, dtype, N, H, W, C, axis,
                               dilation, 0, (hipStream_t)stream);
    if (rc) return rc;
    return bn_finalize_launch((const float*)workspace, groups, C, (long long)N * H * W, gamma, beta, running_mean, running_var,
                              momentum, eps, mean, invstd, scale, shift, (hipStream_t)stream);
}

// dx = conv1d_dgrad(dy) * [gate > 0] + residual, and dbias[c] = sum over pixels of dx[p][c] from the same launch
int erf_conv1d_dgrad_bias(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual, int dtype,
                          int N, int H, int W, int C, int axis, int dilation, float* dbias, void* workspace, void* stream) {
    if (!dy || !dx || !packed_t || !dbias || !workspace || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1)
        return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    int groups = 0;
    int rc = conv1d_buf_launch(dy, dx, packed_t, nullptr, nullptr, residual, gate, (double*)workspace, &groups, dtype, N, H, W, C,
                               axis, dilation, 0, (hipStream_t)stream);
    if (rc) return rc;
    return channel_sum_finalize_launch((const float*)workspace, groups, C, dbias, (hipStream_t)stream);
}

// dx = gate(conv1d_dgrad(dy) + residual), then the backward pass of the BatchNorm whose output gradient dx is:
// its two reductions ride in the convolution's epilogue, so only the apply pass reads dx again.
static int dgrad_bn_backward(const void* dy, void* dx, const void* packed_t, const void* gate, int gate_bits, const void* residual,
                             int dtype, int N, int H, int W, int C, int axis, int dilation, const void* bn_z,
                             const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_mask,
                             int batch_stats, void* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
    if (!dy || !dx || !packed_t || !gate || !bn_z || !bn_mean || !bn_invstd || !bn_gamma || !dz || !dgamma || !dbeta ||
        !workspace || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1)
        return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    Conv1dBnTail tail{bn_z, bn_mean, bn_invstd, bn_mask, 1};
    int groups = 0;
    int rc = conv1d_buf_launch(dy, dx, packed_t, nullptr, nullptr, residual, gate, (double*)workspace, &groups, dtype, N, H, W, C,
                               axis, dilation, 0, (hipStream_t)stream, &tail, gate_bits);
    if (rc) return rc;
    const long long P = (long long)N * H * W;
    float* coef = bn_workspace_coef(workspace, C);
    rc = bn_bwd_finalize_launch((const float*)workspace, groups, C, P, bn_gamma, bn_invstd, batch_stats, dgamma, dbeta, coef,
                                (hipStream_t)stream);
    if (rc) return rc;
    return bn_bwd_apply_launch(dx, nullptr, bn_mask, bn_z, bn_mean, bn_invstd, coef, dz, nullptr, dtype, P, C, (long long)H * W,
                               (hipStream_t)stream);
}

int erf_conv1d_dgrad_bn_backward(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual,
                                 int dtype, int N, int H, int W, int C, int axis, int dilation, const void* bn_z,
                                 const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_mask,
                                 int batch_stats, void* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
    return dgrad_bn_backward(dy, dx, packed_t, gate, 0, residual, dtype, N, H, W, C, axis, dilation, bn_z, bn_mean, bn_invstd, bn_gamma,
                             bn_mask, batch_stats, dz, dgamma, dbeta, workspace, stream);
}

// The same with the gate given as erf_bn_apply_gates wrote it: one byte per 16-byte chunk of the gating tensor
int erf_conv1d_dgrad_bn_backward_bits(const void* dy, void* dx, const void* packed_t, const void* gate_bits, const void* residual,
                                      int dtype, int N, int H, int W, int C, int axis, int dilation, const void* bn_z,
                                      const float* bn_mean, const float* bn_invstd, const float* bn_gamma, const float* bn_mask,
                                      int batch_stats, void* dz, float* dgamma, float* dbeta, void* workspace, void* stream) {
    return dgrad_bn_backward(dy, dx, packed_t, gate_bits, 1, residual, dtype, N, H, W, C, axis, dilation, bn_z, bn_mean, bn_invstd,
                             bn_gamma, bn_mask, batch_stats, dz, dgamma, dbeta, workspace, stream);
}

// erf_conv1d_dgrad_bias without its last step: the partial rows stay in `workspace` (the return value is their number,
// > 0) and erf_channel_sum_finalize_many later turns the partial rows of many such calls into their sums in one launch.
int erf_conv1d_dgrad_bias_partial(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual,
                                  int dtype, int N, int H, int W, int C, int axis, int dilation, void* workspace, void* stream) {
    if (!dy || !dx || !packed_t || !workspace || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1) return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    int groups = 0;
    const int rc = conv1d_buf_launch(dy, dx, packed_t, nullptr, nullptr, residual, gate, (double*)workspace, &groups, dtype, N, H, W,
                                     C, axis, dilation, 0, (hipStream_t)stream);
    return rc ? rc : groups;
}

// table: count rows of {workspace pointer, number of partial rows, C, destination pointer (C floats)} in HOST memory
int erf_channel_sum_finalize_many(const long long* table, int count, void* stream) {
    if (!table || count <= 0) return ERF_EINVAL;
    for (int base = 0; base < count; base += ERF_CHANNEL_SUM_MAX) {
        ChannelSumTable t{};
        t.count = count - base < ERF_CHANNEL_SUM_MAX ? count - base : ERF_CHANNEL_SUM_MAX;
        for (int i = 0; i < t.count; ++i) {
            const long long* r = table + 4 * (long long)(base + i);
            if (!r[0] || !r[3] || r[1] <= 0 || r[2] <= 0) return ERF_EINVAL;
            t.e[i].partials = (const double*)r[0]; t.e[i].groups = (int)r[1]; t.e[i].C = (int)r[2]; t.e[i].sums = (float*)r[3];
        }
        const int rc = channel_sum_finalize_many_launch(t, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ERF_OK;
}

int erf_dropout2d_mask(float* mask, int count, float p, unsigned long long seed, void* stream) {
    if (!mask) return ERF_EINVAL;
    return dropout_mask_launch(mask, count, p, seed, (hipStream_t)stream);
}

int erf_dropout2d_mask_many(float* mask, int segments, const int* counts, const float* p, const unsigned long long* seeds,
                            void* stream) {
    if (!mask || !counts || !p || !seeds) return ERF_EINVAL;
    return dropout_mask_many_launch(mask, segments, counts, p, seeds, (hipStream_t)stream);
}

int erf_dropout2d_mask_many_counter(float* mask, int segments, const int* counts, const float* p, unsigned long long* state,
                                    void* stream) {
    if (!mask || !counts || !p || !state) return ERF_EINVAL;
    return dropout_mask_many_counter_launch(mask, segments, counts, p, state, (hipStream_t)stream);
}

int erf_maxpool2x2_backward(const void* x, const void* dz, void* dx, int dtype, int N, int Hin, int Win, int Cin, int ldz,
                            int zoff, int accumulate, void* stream) {
    if (!x || !dz || !dx || !dtype_ok(dtype) || N <= 0) return ERF_EINVAL;
    return pool_bwd_launch(x, dz, dx, dtype, N, Hin, Win, Cin, ldz, zoff, accumulate, (hipStream_t)stream);
}

// ------------------------------------------------------------------ output head fused with the loss
size_t erf_output_conv_ce_workspace_bytes(int N, int Hin, int Win) {
    if (N <= 0 || Hin <= 0 || Win <= 0) return 0;
    return head_ce_workspace_bytes(N, Hin, Win);
}

int erf_output_conv_ce_fwd(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                           float* out3, void* workspace, int dtype, int N, int Hin, int Win, int nc, void* stream) {
    if (!x || !packed || !bias || !target || !out3 || !workspace || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0)
        return ERF_EINVAL;
    return head_ce_fwd_launch(x, packed, bias, target, weight, out3, (float*)workspace, dtype, N, Hin, Win, nc, (hipStream_t)stream);
}

int erf_output_conv_ce_bwd(const void* x, const void* packed, const float* bias, const long long* target, const float* weight,
                           const float* coef, void* patches, int dtype, int N, int Hin, int Win, int nc, void* stream) {
    if (!x || !packed || !bias || !target || !coef || !patches || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0)
        return ERF_EINVAL;
    return head_ce_bwd_launch(x, packed, bias, target, weight, coef, patches, erf_output_conv_patch_channels(nc), dtype, N, Hin, Win,
                              nc, (hipStream_t)stream);
}

// ------------------------------------------------------------------ loss
size_t erf_ce_workspace_bytes(void) { return ce_workspace_bytes(); }

int erf_weighted_ce_fwd(const void* logits, int dtype, const long long* target, const float* weight, int N, int C,
                        long long HW, float* out3, void* workspace, void* stream) {
    if (!logits || !target || !out3 || !workspace || !dtype_ok(dtype) || N <= 0 || C <= 0 || HW <= 0) return ERF_EINVAL;
    return ce_fwd_launch(logits, dtype, target, weight, N, C, HW, (float*)workspace, out3, (hipStream_t)stream);
}

int erf_weighted_ce_bwd(const void* logits, int dtype, const long long* target, const float* weight, int N, int C,
                        long long HW, const float* coef, void* dlogits, void* stream) {
    if (!logits || !target || !coef || !dlogits || !dtype_ok(dtype) || N <= 0 || C <= 0 || HW <= 0) return ERF_EINVAL;
    return ce_bwd_launch(logits, dtype, target, weight, N, C, HW, coef, dlogits, (hipStream_t)stream);
}

// ------------------------------------------------------------------ optimiser
int erf_adam_fused_arena(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, double lr,
                         double beta1, double beta2, double eps, double weight_decay, int step, double grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq) return ERF_EINVAL;
    return adam_launch(params, grads, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale,
                       (hipStream_t)stream);
}

int erf_adam_advance(int* step, const double* lr, float* hyper, double beta1, double beta2, void* stream) {
    if (!step || !lr || !hyper) return ERF_EINVAL;
    return adam_advance_launch(step, lr, hyper, beta1, beta2, (hipStream_t)stream);
}

int erf_adam_fused_arena_hyper(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, const float* hyper,
                               double beta1, double beta2, double eps, double weight_decay, double grad_scale, void* stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !hyper) return ERF_EINVAL;
    return adam_hyper_launch(params, grads, exp_avg, exp_avg_sq, n, hyper, beta1, beta2, eps, weight_decay, grad_scale,
                             (hipStream_t)stream);
}

int erf_sgd_fused_arena(float* params, const float* grads, float* momentum_buf, long long n, double lr, double momentum,
                        double weight_decay, int first_step, double grad_scale, void* stream) {
    if (!params || !grads || !momentum_buf) return ERF_EINVAL;
    return sgd_launch(params, grads, momentum_buf, n, lr, momentum, weight_decay, first_step, grad_scale, (hipStream_t)stream);
}

int erf_sgd_fused_arena_hyper(float* params, const float* grads, float* momentum_buf, long long n, const float* hyper,
                              double momentum, int first_step, double grad_scale, void* stream) {
    if (!params || !grads || !momentum_buf || !hyper) return ERF_EINVAL;
    return sgd_hyper_launch(params, grads, momentum_buf, n, hyper, momentum, first_step, grad_scale, (hipStream_t)stream);
}

// ------------------------------------------------------------------ data gradients
int erf_conv1d_dgrad(const void* dy, void* dx, const void* packed_t, const void* gate, const void* residual, int dtype,
                     int N, int H, int W, int C, int axis, int dilation, void* stream) {
    if (!dy || !dx || !packed_t || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1) return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    int rc = conv1d_launch(dy, dx, packed_t, nullptr, nullptr, residual, gate, dtype, N, H, W, C, axis, dilation, 0,
                           (hipStream_t)stream);
    if (rc != ERF_EUNSUPPORTED) return rc;
    GConvParams p{};
    p.x = dy; p.y = dx; p.w = packed_t; p.res = residual; p.gate = gate;
    p.N = N; p.Hin = H; p.Win = W; p.ldx = C; p.Hb = H; p.Wb = W; p.Hout = H; p.Wout = W; p.ldy = C;
    p.Cout = C; p.Si = 1; p.So = 1; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.ntaps = 3; ph.ksteps = gconv_ksteps(dtype, C, 3);
    for (int t = 0; t < 3; ++t) {
        ph.dh[t] = axis == ERF_AXIS_H ? (t - 1) * dilation : 0;
        ph.dw[t] = axis == ERF_AXIS_W ? (t - 1) * dilation : 0;
    }
    return gconv_launch(p, dtype, C, (hipStream_t)stream);
}

// DownsamplerBlock conv part: dx = convT3x3s2(dz[..., :Cout-Cin]); the pool part is added by
// erf_maxpool2x2_backward(accumulate = 1).
size_t erf_downsampler_dgrad_packed_bytes(int dtype, int Cin, int Cout) { return erf_upsampler_packed_bytes(dtype, Cout - Cin, Cin); }

int erf_downsampler_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || !((Cin == 16 && Cout == 64) || (Cin == 64 && Cout == 128))) return ERF_EINVAL;
    const int Cc = Cout - Cin;
    GConvParams geo{};
    int wtap[4][4];
    convT3x3_phases(geo, wtap);
    long long off = 0;
    for (int ph = 0; ph < 4; ++ph) {
        PackParams p{};
        p.src = w; p.dst = (char*)packed + off * 16; p.cin = Cc; p.cout_virtual = Cin; p.ncp = NO_PHASES; p.nc_valid = Cin;
        p.s_co = 9; p.s_ci = 9LL * Cin; p.ntaps = geo.ph[ph].ntaps;          // Conv2d weight [Cc][Cin][3][3]: row = ci, k = co
        for (int t = 0; t < p.ntaps; ++t) p.toff[t] = wtap[ph][t];
        p.ct = ct_of(Cin); p.ksteps = gconv_ksteps(dtype, Cc, p.ntaps);
        int rc = pack_launch(p, dtype, (hipStream_t)stream);
        if (rc) return rc;
        off += gconv_phase_chunks(p.ct, p.ksteps);
    }
    return ERF_OK;
}

int erf_downsampler_dgrad(const void* dz, void* dx, const void* packed_t, int dtype, int N, int Hin, int Win, int Cin,
                          int Cout, void* stream) {
    if (!dz || !dx || !packed_t || !dtype_ok(dtype) || N <= 0 || ((Hin | Win) & 1)) return ERF_EINVAL;
    if (!((Cin == 16 && Cout == 64) || (Cin == 64 && Cout == 128))) return ERF_EUNSUPPORTED;
    const int Cc = Cout - Cin;
    GConvParams p{};
    int wtap[4][4];
    convT3x3_phases(p, wtap);
    p.x = dz; p.y = dx; p.w = packed_t;
    p.N = N; p.Hin = Hin / 2; p.Win = Win / 2; p.ldx = Cout; p.xoff = 0; p.Hb = Hin / 2; p.Wb = Win / 2;
    p.Hout = Hin; p.Wout = Win; p.ldy = Cin; p.Cout = Cin; p.Si = 1; p.So = 2;
    long long off = 0;
    for (int ph = 0; ph < 4; ++ph) {
        p.ph[ph].ksteps = gconv_ksteps(dtype, Cc, p.ph[ph].ntaps);
        p.ph[ph].w_chunk_off = (int)off;
        off += gconv_phase_chunks(ct_of(Cin), p.ph[ph].ksteps);
    }
    return gconv_launch(p, dtype, Cc, (hipStream_t)stream);
}

// UpsamplerBlock: dx[i][j][ci] = sum_{th,tw,co} dz[2i-1+th][2j-1+tw][co] * W[ci][co][th][tw]
size_t erf_upsampler_dgrad_packed_bytes(int dtype, int Cin, int Cout) {
    if (!dtype_ok(dtype)) return 0;
    return (size_t)gconv_phase_chunks(ct_of(Cin), gconv_ksteps(dtype, Cout, 9)) * 16;
}

int erf_upsampler_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || !((Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 16))) return ERF_EINVAL;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = Cout; p.cout_virtual = Cin; p.ncp = NO_PHASES; p.nc_valid = Cin;
    p.s_co = 9LL * Cout; p.s_ci = 9; p.ntaps = 9;                          // ConvTranspose2d weight [Cin][Cout][3][3]
    for (int t = 0; t < 9; ++t) p.toff[t] = t;
    p.ct = ct_of(Cin); p.ksteps = gconv_ksteps(dtype, Cout, 9);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_upsampler_dgrad(const void* dz, void* dx, const void* packed_t, const void* gate, int dtype, int N, int Hin, int Win,
                        int Cin, int Cout, void* stream) {
    if (!dz || !dx || !packed_t || !dtype_ok(dtype) || N <= 0) return ERF_EINVAL;
    if (!((Cin == 128 && Cout == 64) || (Cin == 64 && Cout == 16))) return ERF_EUNSUPPORTED;
    GConvParams p{};
    p.x = dz; p.y = dx; p.w = packed_t; p.gate = gate;
    p.N = N; p.Hin = 2 * Hin; p.Win = 2 * Win; p.ldx = Cout; p.Hb = Hin; p.Wb = Win; p.Hout = Hin; p.Wout = Win;
    p.ldy = Cin; p.Cout = Cin; p.Si = 2; p.So = 1; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.ntaps = 9; ph.ksteps = gconv_ksteps(dtype, Cout, 9);
    for (int t = 0; t < 9; ++t) { ph.dh[t] = t / 3 - 1; ph.dw[t] = t % 3 - 1; }
    return gconv_launch(p, dtype, Cout, (hipStream_t)stream);
}

// Decoder.output_conv backward.  dlogits NCHW -> patch layout [N,H,W,ldp] (channel 4*class + phase),
// ldp = 4*nc rounded up to 16; then dx = patches x W^T (1x1), dW = x^T x patches.
int erf_output_conv_patch_channels(int nc) { return (4 * nc + 15) / 16 * 16; }

int erf_output_conv_patches(const void* dlogits, int src_dtype, void* patches, int dtype, int N, int Hin, int Win, int nc,
                            void* stream) {
    if (!dlogits || !patches || !dtype_ok(dtype) || !dtype_ok(src_dtype) || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ld = erf_output_conv_patch_channels(nc);
    return patches_launch(dlogits, src_dtype, patches, dtype, N, Hin, Win, nc, ld / 4, ld, (hipStream_t)stream);
}

size_t erf_output_conv_dgrad_packed_bytes(int dtype, int nc) {
    if (!dtype_ok(dtype) || nc < 1 || nc > 32) return 0;
    return (size_t)gconv_phase_chunks(1, gconv_ksteps(dtype, erf_output_conv_patch_channels(nc), 1)) * 16;
}

int erf_output_conv_dgrad_pack(const float* w, void* packed, int dtype, int nc, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ld = erf_output_conv_patch_channels(nc);
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = ld; p.cin_valid = 4 * nc; p.cout_virtual = 16; p.ncp = NO_PHASES; p.nc_valid = 16;
    p.s_co = 4LL * nc; p.s_ci = 1; p.ntaps = 1; p.toff[0] = 0;             // weight [16][nc][2][2] = [ci][v]
    p.ct = 1; p.ksteps = gconv_ksteps(dtype, ld, 1);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_output_conv_dgrad(const void* patches, void* dx, const void* packed_t, const void* gate, int dtype, int N, int Hin,
                          int Win, int nc, void* stream) {
    if (!patches || !dx || !packed_t || !dtype_ok(dtype) || N <= 0 || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ld = erf_output_conv_patch_channels(nc);
    GConvParams p{};
    p.x = patches; p.y = dx; p.w = packed_t; p.gate = gate;
    p.N = N; p.Hin = Hin; p.Win = Win; p.ldx = ld; p.Hb = Hin; p.Wb = Win; p.Hout = Hin; p.Wout = Win; p.ldy = 16; p.Cout = 16;
    p.Si = 1; p.So = 1; p.nphase = 1;
    p.ph[0].ntaps = 1; p.ph[0].ksteps = gconv_ksteps(dtype, ld, 1);
    return gconv_launch(p, dtype, ld, (hipStream_t)stream);
}

// ---- Encoder.output_conv (1x1, 128 -> nc): gradient of the encoder-only training pass
int erf_pointwise_grad_channels(int nc) { return (nc + 15) / 16 * 16; }

int erf_pointwise_grad_nhwc(const void* dlogits, int src_dtype, void* g, int dtype, int N, int H, int W, int nc, void* stream) {
    if (!dlogits || !g || !dtype_ok(dtype) || !dtype_ok(src_dtype) || N <= 0 || H <= 0 || W <= 0 || nc < 1 || nc > 32)
        return ERF_EINVAL;
    const int ld = erf_pointwise_grad_channels(nc);
    return patches_launch(dlogits, src_dtype, g, dtype, N, H, W, nc, ld, ld, (hipStream_t)stream, 1);
}

size_t erf_pointwise_dgrad_packed_bytes(int dtype, int Cin, int nc) {
    if (!dtype_ok(dtype) || Cin != 128 || nc < 1 || nc > 32) return 0;
    return (size_t)gconv_phase_chunks(Cin / 16, gconv_ksteps(dtype, erf_pointwise_grad_channels(nc), 1)) * 16;
}

int erf_pointwise_dgrad_pack(const float* w, void* packed, int dtype, int Cin, int nc, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || Cin != 128 || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ld = erf_pointwise_grad_channels(nc);
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = ld; p.cin_valid = nc; p.cout_virtual = Cin; p.ncp = NO_PHASES; p.nc_valid = Cin;
    p.s_co = 1; p.s_ci = Cin; p.ntaps = 1; p.toff[0] = 0;                 // weight [nc][Cin]: rows of the transposed GEMM = Cin
    p.ct = Cin / 16; p.ksteps = gconv_ksteps(dtype, ld, 1);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_pointwise_dgrad(const void* g, void* dx, const void* packed_t, int dtype, int N, int H, int W, int Cin, int nc,
                        void* stream) {
    if (!g || !dx || !packed_t || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || nc < 1 || nc > 32) return ERF_EINVAL;
    if (Cin != 128) return ERF_EUNSUPPORTED;
    const int ld = erf_pointwise_grad_channels(nc);
    GConvParams p{};
    p.x = g; p.y = dx; p.w = packed_t;
    p.N = N; p.Hin = H; p.Win = W; p.ldx = ld; p.Hb = H; p.Wb = W; p.Hout = H; p.Wout = W; p.ldy = Cin; p.Cout = Cin;
    p.Si = 1; p.So = 1; p.nphase = 1;
    p.ph[0].ntaps = 1; p.ph[0].ksteps = gconv_ksteps(dtype, ld, 1);
    return gconv_launch(p, dtype, ld, (hipStream_t)stream);
}

int erf_pointwise_wgrad(const void* x, const void* g, float* dw, int dtype, int N, int H, int W, int Cin, int nc,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !g || !dw || !workspace || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || nc < 1 || nc > 32) return ERF_EINVAL;
    if (Cin != 128) return ERF_EUNSUPPORTED;
    const int ld = erf_pointwise_grad_channels(nc);
    GConvParams geo{};
    geo.x = x; geo.N = N; geo.Hin = H; geo.Win = W; geo.ldx = Cin; geo.Hb = H; geo.Wb = W; geo.Hout = H; geo.Wout = W;
    geo.Si = 1; geo.So = 1; geo.nphase = 1; geo.ph[0].ntaps = 1;
    PackParams m{};
    m.cin = Cin; m.cout_virtual = nc; m.ncp = NO_PHASES; m.nc_valid = nc; m.s_co = Cin; m.s_ci = 1; m.ntaps = 1;
    return wgrad_launch(geo, g, ld, 0, nc, Cin, &m, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------ weight gradients
size_t erf_wgrad_workspace_bytes(int rows, int cin, int taps_per_phase, long long base_pixels) {
    return wgrad_workspace_bytes(rows, cin, taps_per_phase, base_pixels);
}

int erf_conv1d_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis, int dilation,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dz || !dw || !workspace || !dtype_ok(dtype) || N <= 0 || dilation < 1) return ERF_EINVAL;
    GConvParams g{};
    g.x = x; g.N = N; g.Hin = H; g.Win = W; g.ldx = C; g.Hb = H; g.Wb = W; g.Hout = H; g.Wout = W; g.Si = 1; g.So = 1; g.nphase = 1;
    g.ph[0].ntaps = 3;
    for (int t = 0; t < 3; ++t) {
        g.ph[0].dh[t] = axis == ERF_AXIS_H ? (t - 1) * dilation : 0;
        g.ph[0].dw[t] = axis == ERF_AXIS_W ? (t - 1) * dilation : 0;
    }
    PackParams m{};
    m.cin = C; m.cout_virtual = C; m.ncp = NO_PHASES; m.nc_valid = C; m.s_co = 3LL * C; m.s_ci = 3; m.ntaps = 3;
    m.toff[0] = 0; m.toff[1] = 1; m.toff[2] = 2;
    const int rc = wgrad1d_launch(x, dz, dw, dtype, N, H, W, C, axis, dilation, m, (float*)workspace, workspace_bytes, (hipStream_t)stream);
    if (rc != ERF_EUNSUPPORTED) return rc;
    return wgrad_launch(g, dz, C, 0, C, C, &m, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

// erf_conv1d_wgrad without its last step: the per-workgroup partial tiles stay in `workspace` (which must stay untouched)
// and `desc` (host memory, erf_wgrad_reduce_desc_bytes()) receives the arguments of the pending reduction.
size_t erf_wgrad_reduce_desc_bytes(void) { return sizeof(WgradReduceDesc); }

int erf_conv1d_wgrad_partial(const void* x, const void* dz, float* dw, int dtype, int N, int H, int W, int C, int axis,
                             int dilation, void* workspace, size_t workspace_bytes, void* desc, void* stream) {
    if (!desc) return ERF_EINVAL;
    WgradReduceDesc* d = (WgradReduceDesc*)desc;
    d->partial = nullptr;
    wgrad_reduce_defer(d);
    const int rc = erf_conv1d_wgrad(x, dz, dw, dtype, N, H, W, C, axis, dilation, workspace, workspace_bytes, stream);
    wgrad_reduce_defer(nullptr);
    if (rc) return rc;
    return d->partial ? ERF_OK : ERF_ELAUNCH;           // the contraction must have reached (and deferred) its reduction
}

// The weight gradients of up to 4 factorized convolutions on tensors of one shape (the four of a non_bottleneck_1d block)
// as ONE contraction launch; like erf_conv1d_wgrad_partial the reductions are left pending in descs[count].
int erf_conv1d_wgrad_partial_many(const erf_wgrad_job* jobs, int count, int dtype, int N, int H, int W, int C,
                                  size_t workspace_bytes, void* descs, void* stream) {
    static_assert(sizeof(erf_wgrad_job) == sizeof(W1dJob), "erf_wgrad_job is W1dJob");
    if (!jobs || !descs || count < 1 || count > 4 || !dtype_ok(dtype) || N <= 0) return ERF_EINVAL;
    PackParams maps[4];
    W1dJob w[4];
    for (int i = 0; i < count; ++i) {
        if (!jobs[i].x || !jobs[i].dz || !jobs[i].dw || !jobs[i].workspace || jobs[i].dilation < 1 ||
            (jobs[i].axis != ERF_AXIS_H && jobs[i].axis != ERF_AXIS_W))
            return ERF_EINVAL;
        PackParams m{};
        m.cin = C; m.cout_virtual = C; m.ncp = NO_PHASES; m.nc_valid = C; m.s_co = 3LL * C; m.s_ci = 3; m.ntaps = 3;
        m.toff[0] = 0; m.toff[1] = 1; m.toff[2] = 2;
        maps[i] = m;
        w[i] = W1dJob{jobs[i].x, jobs[i].dz, jobs[i].dw, (float*)jobs[i].workspace, jobs[i].axis, jobs[i].dilation};
        ((WgradReduceDesc*)descs)[i].partial = nullptr;
    }
    const int rc = wgrad1d_many_launch(w, count, dtype, N, H, W, C, maps, workspace_bytes, (WgradReduceDesc*)descs, (hipStream_t)stream);
    if (rc) return rc;
    for (int i = 0; i < count; ++i)
        if (!((WgradReduceDesc*)descs)[i].partial) return ERF_ELAUNCH;
    return ERF_OK;
}

// Between begin and end every erf_*_wgrad call of the calling thread leaves its reductions in `descs` (host memory,
// capacity * erf_wgrad_reduce_desc_bytes()) instead of launching them; end returns how many were recorded.
int erf_wgrad_reduce_record_begin(void* descs, int capacity) {
    if (!descs || capacity < 1) return ERF_EINVAL;
    wgrad_reduce_record_begin((WgradReduceDesc*)descs, capacity);
    return ERF_OK;
}
int erf_wgrad_reduce_record_end(void) { return wgrad_reduce_record_end(); }

int erf_wgrad_reduce_many(const void* table, int count, void* stream) {
    return wgrad_reduce_many_launch(table, count, (hipStream_t)stream);
}

int erf_downsampler_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int Hin, int Win, int Cin, int Cout,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dz || !dw || !workspace || !dtype_ok(dtype) || N <= 0 || ((Hin | Win) & 1)) return ERF_EINVAL;
    const int Cc = Cout - Cin;
    GConvParams g{};
    g.x = x; g.N = N; g.Hin = Hin; g.Win = Win; g.ldx = Cin; g.Hb = Hin / 2; g.Wb = Win / 2; g.Hout = Hin / 2; g.Wout = Win / 2;
    g.Si = 2; g.So = 1; g.nphase = 1;
    g.ph[0].ntaps = 9;
    for (int t = 0; t < 9; ++t) { g.ph[0].dh[t] = t / 3 - 1; g.ph[0].dw[t] = t % 3 - 1; }
    PackParams m{};
    m.cin = Cin; m.cout_virtual = Cc; m.ncp = NO_PHASES; m.nc_valid = Cc; m.s_co = 9LL * Cin; m.s_ci = 9; m.ntaps = 9;
    for (int t = 0; t < 9; ++t) m.toff[t] = t;
    return wgrad_launch(g, dz, Cout, 0, Cc, Cin, &m, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

// Stem (3 -> 16): x_pad is the image as NHWC with the channels zero-padded to `cpad` (8 for bf16, 4 for
// fp32; erf_image_to_nhwc_padded), dz the [N,Hin/2,Win/2,16] gradient of the block's pre-BN output.
int erf_image_to_nhwc_padded(const void* x_nchw, int x_dtype, void* out, int dtype, int N, int H, int W, void* stream) {
    if (!x_nchw || !out || !dtype_ok(dtype) || x_dtype < 0 || x_dtype > ERF_U8_NHWC || N <= 0) return ERF_EINVAL;
    return image_pad_launch(x_nchw, x_dtype, out, dtype, N, (long long)H * W, dtype == ERF_BF16 ? 8 : 4, (hipStream_t)stream);
}

int erf_stem_wgrad(const void* x_pad, const void* dz, float* dw, int dtype, int N, int Hin, int Win, void* workspace,
                   size_t workspace_bytes, void* stream) {
    if (!x_pad || !dz || !dw || !workspace || !dtype_ok(dtype) || N <= 0 || ((Hin | Win) & 1)) return ERF_EINVAL;
    const int cpad = dtype == ERF_BF16 ? 8 : 4;
    GConvParams g{};
    g.x = x_pad; g.N = N; g.Hin = Hin; g.Win = Win; g.ldx = cpad; g.Hb = Hin / 2; g.Wb = Win / 2; g.Hout = Hin / 2; g.Wout = Win / 2;
    g.Si = 2; g.So = 1; g.nphase = 1;
    g.ph[0].ntaps = 9;
    for (int t = 0; t < 9; ++t) { g.ph[0].dh[t] = t / 3 - 1; g.ph[0].dw[t] = t % 3 - 1; }
    PackParams m{};
    m.cin = cpad; m.cin_valid = 3; m.cout_virtual = 13; m.ncp = NO_PHASES; m.nc_valid = 13; m.s_co = 27; m.s_ci = 9; m.ntaps = 9;
    for (int t = 0; t < 9; ++t) m.toff[t] = t;
    return wgrad_launch(g, dz, 16, 0, 13, cpad, &m, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

int erf_upsampler_wgrad(const void* x, const void* dz, float* dw, int dtype, int N, int Hin, int Win, int Cin, int Cout,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dz || !dw || !workspace || !dtype_ok(dtype) || N <= 0) return ERF_EINVAL;
    GConvParams g{};
    int wtap[4][4];
    convT3x3_phases(g, wtap);
    g.x = x; g.N = N; g.Hin = Hin; g.Win = Win; g.ldx = Cin; g.Hb = Hin; g.Wb = Win; g.Hout = 2 * Hin; g.Wout = 2 * Win;
    g.Si = 1; g.So = 2;
    PackParams maps[4];
    for (int ph = 0; ph < 4; ++ph) {
        PackParams m{};
        m.cin = Cin; m.cout_virtual = Cout; m.ncp = NO_PHASES; m.nc_valid = Cout; m.s_co = 9; m.s_ci = 9LL * Cout;
        m.ntaps = g.ph[ph].ntaps;
        for (int t = 0; t < m.ntaps; ++t) m.toff[t] = wtap[ph][t];
        maps[ph] = m;
    }
    return wgrad_launch(g, dz, Cout, 0, Cout, Cin, maps, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

int erf_output_conv_wgrad(const void* x, const void* patches, float* dw, int dtype, int N, int Hin, int Win, int nc,
                          void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !patches || !dw || !workspace || !dtype_ok(dtype) || N <= 0 || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ld = erf_output_conv_patch_channels(nc);
    GConvParams g{};
    g.x = x; g.N = N; g.Hin = Hin; g.Win = Win; g.ldx = 16; g.Hb = Hin; g.Wb = Win; g.Hout = Hin; g.Wout = Win; g.Si = 1; g.So = 1;
    g.nphase = 1; g.ph[0].ntaps = 1;
    PackParams m{};
    m.cin = 16; m.cout_virtual = 4 * nc; m.ncp = NO_PHASES; m.nc_valid = 4 * nc; m.s_co = 1; m.s_ci = 4LL * nc; m.ntaps = 1;
    return wgrad_launch(g, patches, ld, 0, 4 * nc, 16, &m, dw, (float*)workspace, workspace_bytes, dtype, (hipStream_t)stream);
}

// ---- classifier head of the ImageNet pre-training variant
int erf_maxpool2_avgpool_fwd(const void* x, int dtype, float* feat, int N, int H, int W, int C, void* stream) {
    if (!x || !feat || !dtype_ok(dtype) || N <= 0 || C <= 0 || H < 2 || W < 2 || ((H | W) & 1)) return ERF_EINVAL;
    return pool_mean_fwd_launch(x, dtype, feat, N, H, W, C, (hipStream_t)stream);
}

int erf_maxpool2_avgpool_bwd(const void* x, const float* dfeat, void* dx, int dtype, int N, int H, int W, int C, void* stream) {
    if (!x || !dfeat || !dx || !dtype_ok(dtype) || N <= 0 || C <= 0 || H < 2 || W < 2 || ((H | W) & 1)) return ERF_EINVAL;
    return pool_mean_bwd_launch(x, dfeat, dx, dtype, N, H, W, C, (hipStream_t)stream);
}

int erf_linear_fwd(const float* f, const float* w, const float* b, float* y, int N, int C, int K, void* stream) {
    if (!f || !w || !y || N <= 0 || C <= 0 || K <= 0) return ERF_EINVAL;
    return linear_fwd_launch(f, w, b, y, N, C, K, (hipStream_t)stream);
}

int erf_linear_bwd(const float* f, const float* w, const float* dy, float* df, float* dw, float* db, int N, int C, int K,
                   void* stream) {
    if (!f || !w || !dy || !df || !dw || !db || N <= 0 || C <= 0 || K <= 0) return ERF_EINVAL;
    return linear_bwd_launch(f, w, dy, df, dw, db, N, C, K, (hipStream_t)stream);
}

}  // extern "C"
