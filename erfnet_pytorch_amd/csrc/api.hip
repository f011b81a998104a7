// extern "C" surface of libermi355.so (declared in include/erfnet_mi355.h): argument checks,
// tap/phase tables for each layer kind, and dispatch to the kernels.
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

extern "C" {

int erf_version(void) { return 100; }

// ------------------------------------------------------------------ factorized conv, C -> C
size_t erf_conv1d_packed_bytes(int dtype, int C) {
    if (!dtype_ok(dtype) || C <= 0) return 0;
    return (size_t)gconv_phase_chunks(ct_of(C), gconv_ksteps(dtype, C, 3)) * 16;
}

int erf_conv1d_pack(const float* w, void* packed, int dtype, int C, int mode, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || (C != 16 && C != 64 && C != 128) || (mode != 0 && mode != 1)) return ERF_EINVAL;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = C; p.cout_virtual = C; p.ncp = NO_PHASES; p.nc_valid = C;
    p.s_phase = 0; p.ntaps = 3;
    if (mode == 0) { p.s_co = 3LL * C; p.s_ci = 3; p.toff[0] = 0; p.toff[1] = 1; p.toff[2] = 2; }
    else           { p.s_co = 3; p.s_ci = 3LL * C; p.toff[0] = 2; p.toff[1] = 1; p.toff[2] = 0; }
    p.ct = ct_of(C); p.ksteps = gconv_ksteps(dtype, C, 3);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_conv1d_pack_many(const long long* table, int count, int max_c, int dtype, void* stream) {
    if (!table || count <= 0 || !dtype_ok(dtype) || (max_c != 16 && max_c != 64 && max_c != 128)) return ERF_EINVAL;
    return pack1d_many_launch(table, count, max_c, dtype, (hipStream_t)stream);
}

int erf_conv1d_factorized(const void* x, void* y, const void* packed, const float* scale, const float* shift,
                          const void* residual, int dtype, int N, int H, int W, int C, int axis, int dilation, int relu,
                          void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1) return ERF_EINVAL;
    if (C != 16 && C != 64 && C != 128) return ERF_EUNSUPPORTED;
    if (axis != ERF_AXIS_H && axis != ERF_AXIS_W) return ERF_EINVAL;
    int rc = conv1d_launch(x, y, packed, scale, shift, residual, nullptr, dtype, N, H, W, C, axis, dilation, relu, (hipStream_t)stream);
    if (rc != ERF_EUNSUPPORTED) return rc;
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift; p.res = residual;
    p.N = N; p.Hin = H; p.Win = W; p.ldx = C; p.xoff = 0; p.Hb = H; p.Wb = W; p.Hout = H; p.Wout = W; p.ldy = C; p.yoff = 0;
    p.Cout = C; p.Si = 1; p.So = 1; p.relu = relu; p.out_mode = 0; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.oh = ph.ow = 0; ph.ntaps = 3; ph.w_chunk_off = 0; ph.ksteps = gconv_ksteps(dtype, C, 3);
    for (int t = 0; t < 3; ++t) {
        ph.dh[t] = axis == ERF_AXIS_H ? (t - 1) * dilation : 0;
        ph.dw[t] = axis == ERF_AXIS_W ? (t - 1) * dilation : 0;
    }
    return gconv_launch(p, dtype, C, (hipStream_t)stream);
}

int erf_conv1d_pair(const void* x, void* y, const void* packed_a, const void* packed_b, const float* scale_a, const float* shift_a,
                    const float* scale_b, const float* shift_b, const void* residual, int dtype, int N, int H, int W, int C,
                    int dilation, int relu, void* stream) {
    if (!x || !y || !packed_a || !packed_b || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || dilation < 1) return ERF_EINVAL;
    if (x == y) return ERF_EINVAL;
    return conv1d_pair_launch(x, y, packed_a, packed_b, scale_a, shift_a, scale_b, shift_b, residual, dtype, N, H, W, C, dilation,
                              relu, (hipStream_t)stream);
}

int erf_nb1d_block_infer(const void* x, void* y, const void* packed1, const void* packed2, const void* packed3, const void* packed4,
                         const float* table, int dtype, int N, int H, int W, int C, int dilation, void* stream) {
    if (!x || !y || !packed1 || !packed2 || !packed3 || !packed4 || !table || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 ||
        dilation < 1 || x == y)
        return ERF_EINVAL;
    const void* const packed[4] = {packed1, packed2, packed3, packed4};
    if (C == 16) return nb1d_block16_launch(x, y, packed, table, dtype, N, H, W, C, dilation, (hipStream_t)stream);
    return nb1d_block64s_launch(x, y, packed, table, dtype, N, H, W, C, dilation, (hipStream_t)stream);
}

// ------------------------------------------------------------------ DownsamplerBlock
size_t erf_downsampler_packed_bytes(int dtype, int Cin, int Cout) {
    if (!dtype_ok(dtype) || Cin <= 0 || Cout <= Cin) return 0;
    return (size_t)gconv_phase_chunks(ct_of(Cout - Cin), gconv_ksteps(dtype, Cin, 9)) * 16;
}

int erf_downsampler_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || (Cin != 16 && Cin != 64) || Cout <= Cin) return ERF_EINVAL;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = Cin; p.cout_virtual = Cout - Cin; p.ncp = NO_PHASES; p.nc_valid = Cout - Cin;
    p.s_co = 9LL * Cin; p.s_ci = 9; p.s_phase = 0; p.ntaps = 9;
    for (int t = 0; t < 9; ++t) p.toff[t] = t;
    p.ct = ct_of(Cout - Cin); p.ksteps = gconv_ksteps(dtype, Cin, 9);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_downsampler(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N,
                    int Hin, int Win, int Cin, int Cout, int relu, void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0) return ERF_EINVAL;
    if ((Hin | Win) & 1) return ERF_EINVAL;      // the reference's cat() needs even H, W as well
    if (!((Cin == 16 && Cout == 64) || (Cin == 64 && Cout == 128))) return ERF_EUNSUPPORTED;
    {
        const int rc = down_launch(x, y, packed, scale, shift, dtype, N, Hin, Win, Cin, Cout, relu, (hipStream_t)stream);
        if (rc != ERF_EUNSUPPORTED) return rc;
    }
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift; p.res = nullptr;
    p.N = N; p.Hin = Hin; p.Win = Win; p.ldx = Cin; p.xoff = 0; p.Hb = Hin / 2; p.Wb = Win / 2; p.Hout = Hin / 2; p.Wout = Win / 2;
    p.ldy = Cout; p.yoff = 0; p.Cout = Cout - Cin; p.Si = 2; p.So = 1; p.relu = relu; p.out_mode = 0; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.oh = ph.ow = 0; ph.ntaps = 9; ph.w_chunk_off = 0; ph.ksteps = gconv_ksteps(dtype, Cin, 9);
    for (int t = 0; t < 9; ++t) { ph.dh[t] = t / 3 - 1; ph.dw[t] = t % 3 - 1; }
    int rc = gconv_launch(p, dtype, Cin, (hipStream_t)stream);
    if (rc) return rc;
    return pool_affine_launch(x, y, scale, shift, dtype, N, Hin, Win, Cin, Cout, Cout - Cin, relu, (hipStream_t)stream);
}

int erf_stem_downsampler(const void* x_nchw, int x_dtype, void* y, const float* w, const float* scale, const float* shift,
                         int dtype, int N, int Hin, int Win, int relu, void* stream) {
    if (!x_nchw || !y || !w || !dtype_ok(dtype) || x_dtype < 0 || x_dtype > ERF_U8_NHWC || N <= 0 || Hin <= 0 || Win <= 0) return ERF_EINVAL;
    return stem_launch(x_nchw, x_dtype, y, w, scale, shift, dtype, N, Hin, Win, relu, (hipStream_t)stream);
}

// ------------------------------------------------------------------ UpsamplerBlock
// Output parity phase (a, b): rows use kernel taps th with (a + 1 - th) even, reading input row
// i + (a + 1 - th) / 2; same for columns.  a = 0: th = 1 (dh 0).  a = 1: th = 0 (dh +1), th = 2 (dh 0).
static int up_axis_taps(int parity, int th[2], int d[2]) {
    if (parity == 0) { th[0] = 1; d[0] = 0; return 1; }
    th[0] = 0; d[0] = 1; th[1] = 2; d[1] = 0; return 2;
}

size_t erf_upsampler_packed_bytes(int dtype, int Cin, int Cout) {
    if (!dtype_ok(dtype) || Cin <= 0 || Cout <= 0) return 0;
    long long chunks = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) chunks += gconv_phase_chunks(ct_of(Cout), gconv_ksteps(dtype, Cin, (a + 1) * (b + 1)));
    return (size_t)chunks * 16;
}

int erf_upsampler_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || !((Cin == 128 && (Cout == 64 || Cout == 16)) || (Cin == 64 && Cout == 16))) return ERF_EINVAL;
    long long off = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            int th[2], dh[2], tw[2], dw[2];
            const int nh = up_axis_taps(a, th, dh), nw = up_axis_taps(b, tw, dw);
            PackParams p{};
            p.src = w; p.dst = (char*)packed + off * 16; p.cin = Cin; p.cout_virtual = Cout; p.ncp = NO_PHASES; p.nc_valid = Cout;
            p.s_co = 9; p.s_ci = 9LL * Cout; p.s_phase = 0; p.ntaps = nh * nw;      // ConvTranspose2d weight [Cin][Cout][3][3]
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) p.toff[i * nw + j] = th[i] * 3 + tw[j];
            p.ct = ct_of(Cout); p.ksteps = gconv_ksteps(dtype, Cin, p.ntaps);
            int rc = pack_launch(p, dtype, (hipStream_t)stream);
            if (rc) return rc;
            off += gconv_phase_chunks(p.ct, p.ksteps);
        }
    return ERF_OK;
}

int erf_upsampler_convT3x3s2(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype,
                             int N, int Hin, int Win, int Cin, int Cout, int relu, void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0) return ERF_EINVAL;
    if (!((Cin == 128 && (Cout == 64 || Cout == 16)) || (Cin == 64 && Cout == 16))) return ERF_EUNSUPPORTED;   // 128 -> 16: multi_models.py:127
    {
        const int rc = up_launch(x, y, packed, scale, shift, dtype, N, Hin, Win, Cin, Cout, relu, (hipStream_t)stream);
        if (rc != ERF_EUNSUPPORTED) return rc;
    }
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift; p.res = nullptr;
    p.N = N; p.Hin = Hin; p.Win = Win; p.ldx = Cin; p.xoff = 0; p.Hb = Hin; p.Wb = Win; p.Hout = 2 * Hin; p.Wout = 2 * Win;
    p.ldy = Cout; p.yoff = 0; p.Cout = Cout; p.Si = 1; p.So = 2; p.relu = relu; p.out_mode = 0; p.nphase = 4;
    long long off = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            GConvPhase& ph = p.ph[a * 2 + b];
            int th[2], dh[2], tw[2], dw[2];
            const int nh = up_axis_taps(a, th, dh), nw = up_axis_taps(b, tw, dw);
            ph.oh = a; ph.ow = b; ph.ntaps = nh * nw;
            for (int i = 0; i < nh; ++i)
                for (int j = 0; j < nw; ++j) { ph.dh[i * nw + j] = dh[i]; ph.dw[i * nw + j] = dw[j]; }
            ph.ksteps = gconv_ksteps(dtype, Cin, ph.ntaps);
            ph.w_chunk_off = (int)off;
            off += gconv_phase_chunks(ct_of(Cout), ph.ksteps);
        }
    return gconv_launch(p, dtype, Cin, (hipStream_t)stream);
}

// ------------------------------------------------------------------ Decoder.output_conv
size_t erf_output_conv_packed_bytes(int dtype, int nc) {
    if (!dtype_ok(dtype) || nc < 1 || nc > 32) return 0;
    const int ncp = (nc + 3) & ~3;
    return (size_t)gconv_phase_chunks(ct_of(4 * ncp), gconv_ksteps(dtype, 16, 1)) * 16;
}

int erf_output_conv_pack(const float* w, void* packed, int dtype, int nc, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || nc < 1 || nc > 32) return ERF_EINVAL;
    const int ncp = (nc + 3) & ~3;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = 16; p.cout_virtual = 4 * ncp; p.ncp = ncp; p.nc_valid = nc;
    p.s_co = 4; p.s_phase = 1; p.s_ci = 4LL * nc; p.ntaps = 1; p.toff[0] = 0;       // weight [16][nc][2][2]
    p.ct = ct_of(4 * ncp); p.ksteps = gconv_ksteps(dtype, 16, 1);
    p.interleave4 = 1;          // tile c, row 4*phase + e  <->  class 4c + e: a lane of the MFMA result owns one phase
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_output_convT2x2s2(const void* x, const void* packed, const float* bias, void* logits, int logits_f32, void* labels,
                          int label_bytes, int dtype, int N, int Hin, int Win, int nc, void* stream) {
    if (!x || !packed || !bias || (!logits && !labels) || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0) return ERF_EINVAL;
    return outconv_launch(x, packed, bias, logits, logits_f32, labels, label_bytes, dtype, N, Hin, Win, nc, (hipStream_t)stream);
}

// ------------------------------------------------------------------ Encoder.output_conv (1x1)
size_t erf_pointwise_packed_bytes(int dtype, int Cin, int nc) {
    if (!dtype_ok(dtype) || Cin <= 0 || nc < 1 || nc > 32) return 0;
    return (size_t)gconv_phase_chunks(ct_of(nc), gconv_ksteps(dtype, Cin, 1)) * 16;
}

int erf_pointwise_pack(const float* w, void* packed, int dtype, int Cin, int nc, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || Cin != 128 || nc < 1 || nc > 32) return ERF_EINVAL;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = Cin; p.cout_virtual = nc; p.ncp = NO_PHASES; p.nc_valid = nc;
    p.s_co = Cin; p.s_ci = 1; p.s_phase = 0; p.ntaps = 1; p.toff[0] = 0;
    p.ct = ct_of(nc); p.ksteps = gconv_ksteps(dtype, Cin, 1);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_pointwise_conv(const void* x, void* y, const void* packed, const float* bias, int out_f32, int dtype, int N, int H,
                       int W, int Cin, int nc, void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0) return ERF_EINVAL;
    if (Cin != 128 || nc < 1 || nc > 32) return ERF_EUNSUPPORTED;
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = nullptr; p.shift = bias; p.res = nullptr;
    p.N = N; p.Hin = H; p.Win = W; p.ldx = Cin; p.xoff = 0; p.Hb = H; p.Wb = W; p.Hout = H; p.Wout = W;
    p.ldy = nc; p.yoff = 0; p.Cout = nc; p.Si = 1; p.So = 1; p.relu = 0; p.out_mode = out_f32 ? 2 : 1; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.oh = ph.ow = 0; ph.ntaps = 1; ph.dh[0] = ph.dw[0] = 0; ph.w_chunk_off = 0; ph.ksteps = gconv_ksteps(dtype, Cin, 1);
    return gconv_launch(p, dtype, Cin, (hipStream_t)stream);
}

// ------------------------------------------------------------------ Multi_models (train/multi_models.py), inference
// The multi-task variant reuses the blocks above plus four generic pieces on the gather-form convolution.
static bool conv3x3_shape_ok(int Cin, int Cout) {
    return (Cin == 16 && (Cout == 16 || Cout == 64)) || (Cin == 64 && Cout == 128) || (Cin == 128 && Cout == 64);
}

size_t erf_conv3x3_packed_bytes(int dtype, int Cin, int Cout) {
    if (!dtype_ok(dtype) || !conv3x3_shape_ok(Cin, Cout)) return 0;
    return (size_t)gconv_phase_chunks(ct_of(Cout), gconv_ksteps(dtype, Cin, 9)) * 16;
}

int erf_conv3x3_pack(const float* w, void* packed, int dtype, int Cin, int Cin_valid, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || !conv3x3_shape_ok(Cin, Cout) || Cin_valid < 1 || Cin_valid > Cin) return ERF_EINVAL;
    PackParams p{};
    p.src = w; p.dst = packed; p.cin = Cin; p.cin_valid = Cin_valid == Cin ? 0 : Cin_valid; p.cout_virtual = Cout; p.ncp = NO_PHASES;
    p.nc_valid = Cout; p.s_co = 9LL * Cin_valid; p.s_ci = 9; p.s_phase = 0; p.ntaps = 9;      // weight [Cout][Cin_valid][3][3]
    for (int t = 0; t < 9; ++t) p.toff[t] = t;
    p.ct = ct_of(Cout); p.ksteps = gconv_ksteps(dtype, Cin, 9);
    return pack_launch(p, dtype, (hipStream_t)stream);
}

int erf_conv3x3(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin,
                int Win, int Cin, int Cout, int stride, int relu, void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0 || (stride != 1 && stride != 2)) return ERF_EINVAL;
    if (!conv3x3_shape_ok(Cin, Cout)) return ERF_EUNSUPPORTED;
    const int Ho = (Hin + stride - 1) / stride, Wo = (Win + stride - 1) / stride;      // kernel 3, padding 1
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.scale = scale; p.shift = shift;
    p.N = N; p.Hin = Hin; p.Win = Win; p.ldx = Cin; p.Hb = Ho; p.Wb = Wo; p.Hout = Ho; p.Wout = Wo;
    p.ldy = Cout; p.Cout = Cout; p.Si = stride; p.So = 1; p.relu = relu; p.out_mode = 0; p.nphase = 1;
    GConvPhase& ph = p.ph[0];
    ph.ntaps = 9; ph.ksteps = gconv_ksteps(dtype, Cin, 9);
    for (int t = 0; t < 9; ++t) { ph.dh[t] = t / 3 - 1; ph.dw[t] = t % 3 - 1; }
    return gconv_launch(p, dtype, Cin, (hipStream_t)stream);
}

int erf_maxpool2x2_affine(const void* x, void* y, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
                          int C, int relu, void* stream) {
    if (!x || !y || !dtype_ok(dtype) || N <= 0 || Hin < 2 || Win < 2 || ((Hin | Win) & 1) || C <= 0) return ERF_EINVAL;
    return pool_affine_launch(x, y, scale, shift, dtype, N, Hin, Win, C, C, 0, relu, (hipStream_t)stream);
}

size_t erf_convT2x2s2_packed_bytes(int dtype, int Cin, int Cout) {
    if (!dtype_ok(dtype) || Cin != 32 || Cout < 1 || Cout > 16) return 0;
    return (size_t)4 * gconv_phase_chunks(1, gconv_ksteps(dtype, Cin, 1)) * 16;
}

int erf_convT2x2s2_pack(const float* w, void* packed, int dtype, int Cin, int Cout, void* stream) {
    if (!w || !packed || !dtype_ok(dtype) || Cin != 32 || Cout < 1 || Cout > 16) return ERF_EINVAL;
    const long long per = gconv_phase_chunks(1, gconv_ksteps(dtype, Cin, 1));
    for (int ph = 0; ph < 4; ++ph) {
        PackParams p{};
        p.src = w; p.dst = (char*)packed + ph * per * 16; p.cin = Cin; p.cout_virtual = Cout; p.ncp = NO_PHASES; p.nc_valid = Cout;
        p.s_co = 4; p.s_ci = 4LL * Cout; p.ntaps = 1; p.toff[0] = ph;               // ConvTranspose2d weight [Cin][Cout][2][2]
        p.ct = 1; p.ksteps = gconv_ksteps(dtype, Cin, 1);
        const int rc = pack_launch(p, dtype, (hipStream_t)stream);
        if (rc) return rc;
    }
    return ERF_OK;
}

int erf_convT2x2s2(const void* x, void* y, const void* packed, const float* bias, int out_f32, int dtype, int N, int Hin, int Win,
                   int Cin, int Cout, void* stream) {
    if (!x || !y || !packed || !dtype_ok(dtype) || N <= 0 || Hin <= 0 || Win <= 0) return ERF_EINVAL;
    if (Cin != 32 || Cout < 1 || Cout > 16) return ERF_EUNSUPPORTED;
    GConvParams p{};
    p.x = x; p.y = y; p.w = packed; p.shift = bias;
    p.N = N; p.Hin = Hin; p.Win = Win; p.ldx = Cin; p.Hb = Hin; p.Wb = Win; p.Hout = 2 * Hin; p.Wout = 2 * Win;
    p.ldy = Cout; p.Cout = Cout; p.Si = 1; p.So = 2; p.out_mode = out_f32 ? 2 : 1; p.nphase = 4;
    const long long per = gconv_phase_chunks(1, gconv_ksteps(dtype, Cin, 1));
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            GConvPhase& ph = p.ph[a * 2 + b];
            ph.oh = a; ph.ow = b; ph.ntaps = 1; ph.dh[0] = ph.dw[0] = 0;
            ph.ksteps = gconv_ksteps(dtype, Cin, 1);
            ph.w_chunk_off = (int)((a * 2 + b) * per);
        }
    return gconv_launch(p, dtype, Cin, (hipStream_t)stream);
}

int erf_spatial_mean(const void* x, int dtype, float* feat, int N, int H, int W, int C, void* stream) {
    if (!x || !feat || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || C <= 0) return ERF_EINVAL;
    return spatial_mean_launch(x, dtype, feat, N, (long long)H * W, C, (hipStream_t)stream);
}

int erf_concat_channels(const void* a, int Ca, const void* b, int Cb, void* out, int dtype, int N, int H, int W, void* stream) {
    if (!a || !b || !out || !dtype_ok(dtype) || N <= 0 || H <= 0 || W <= 0 || Ca <= 0 || Cb <= 0) return ERF_EINVAL;
    return concat_channels_launch(a, Ca, b, Cb, out, dtype, (long long)N * H * W, (hipStream_t)stream);
}

int erf_image_to_nhwc_channels(const void* x_nchw, int x_dtype, void* out, int dtype, int N, int H, int W, int cpad, void* stream) {
    if (!x_nchw || !out || !dtype_ok(dtype) || !dtype_ok(x_dtype) || N <= 0 || cpad < 3) return ERF_EINVAL;
    return image_pad_launch(x_nchw, x_dtype, out, dtype, N, (long long)H * W, cpad, (hipStream_t)stream);
}

// ------------------------------------------------------------------ batched weight re-packing
size_t erf_pack_desc_bytes(void) { return sizeof(PackParams); }

int erf_pack_record_begin(void* host_table, int capacity) { return pack_record_begin(host_table, capacity); }

int erf_pack_record_end(long long* max_chunks, int* dtype) { return pack_record_end(max_chunks, dtype); }

int erf_pack_replay(const void* table_dev, int count, long long max_chunks, int dtype, void* stream) {
    return pack_replay_launch(table_dev, count, max_chunks, dtype, (hipStream_t)stream);
}

// ------------------------------------------------------------------ layout glue and metrics
int erf_nchw_to_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || !dtype_ok(src_dtype) || !dtype_ok(dst_dtype) || N <= 0 || C <= 0 || H <= 0 || W <= 0) return ERF_EINVAL;
    return transpose_launch(src, src_dtype, dst, dst_dtype, N, C, (long long)H * W, (hipStream_t)stream);
}

int erf_nhwc_to_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int N, int C, int H, int W, void* stream) {
    if (!src || !dst || !dtype_ok(src_dtype) || !dtype_ok(dst_dtype) || N <= 0 || C <= 0 || H <= 0 || W <= 0) return ERF_EINVAL;
    return transpose_launch(src, src_dtype, dst, dst_dtype, N, (long long)H * W, C, (hipStream_t)stream);
}

int erf_argmax_c(const void* logits, int dtype, void* labels, int label_bytes, int N, int C, long long HW, void* stream) {
    if (!logits || !labels || !dtype_ok(dtype) || N <= 0 || C <= 0 || HW <= 0) return ERF_EINVAL;
    return argmax_launch(logits, dtype, labels, label_bytes, N, C, HW, (hipStream_t)stream);
}

int erf_iou_hist_accumulate(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count, int nc,
                            long long* hist, void* stream) {
    if (!pred || !label || !hist) return ERF_EINVAL;
    return iou_hist_launch(pred, pred_bytes, label, label_bytes, count, nc, -1, 0, hist, (hipStream_t)stream);
}

int erf_iou_hist_accumulate_relabel(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count, int nc,
                                    int olabel, int nlabel, long long* hist, void* stream) {
    if (!pred || !label || !hist || olabel < 0) return ERF_EINVAL;
    return iou_hist_launch(pred, pred_bytes, label, label_bytes, count, nc, olabel, nlabel, hist, (hipStream_t)stream);
}

int erf_labels_from_u8(const void* src, long long* dst, long long count, int olabel, int nlabel, void* stream) {
    if (!src || !dst || count < 0) return ERF_EINVAL;
    if ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) return ERF_EINVAL;
    return labels_from_u8_launch(src, dst, count, olabel, nlabel, (hipStream_t)stream);
}

size_t erf_iou_planes_workspace_bytes(int nc) { return iou_planes_workspace_bytes(nc); }

int erf_iou_planes_accumulate(const void* x, int x_kind, const void* y, int y_kind, int N, int nc, long long HW, int ignore_index,
                              double* sums, void* workspace, void* stream) {
    if (!x || !y || !sums || !workspace || N <= 0 || HW <= 0) return ERF_EINVAL;
    return iou_planes_launch(x, x_kind, y, y_kind, N, nc, HW, ignore_index, sums, (double*)workspace, (hipStream_t)stream);
}

}  // extern "C"
