#pragma once
#include <hip/hip_runtime.h>

// LDS-tiled factorized 3x1 / 1x3 convolution (the hot kernel).  Returns ERF_EUNSUPPORTED for
// shapes it does not take; the caller then uses the gather-form kernel.
int conv1d_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                  const void* gate, int dtype, int N, int H, int W, int C, int axis, int dilation, int relu, hipStream_t stream);

// Fused conv3x1 -> bias -> ReLU -> conv1x3 -> scale/shift (+ residual) -> ReLU (bf16 only).  Returns
// ERF_EUNSUPPORTED for shapes it does not take; the caller then issues two conv1d launches.
int conv1d_pair_launch(const void* x, void* y, const void* packedA, const void* packedB, const float* scaleA, const float* shiftA,
                       const float* scaleB, const float* shiftB, const void* residual, int dtype, int N, int H, int W, int C,
                       int dilation, int relu, hipStream_t stream);

// The output y is the gradient arriving at the output of a BatchNorm layer (y is a data gradient): take that layer's
// two backward sums in the epilogue.  z: the layer's pre-normalisation input (same shape and type as y), mean / invstd
// [C], mask: Dropout2d multipliers [N][C] applied after the BatchNorm (or null).  gate_after: with both a gate and a
// residual, y = gate(conv + residual) instead of gate(conv) + residual.
struct Conv1dBnTail {
    const void* z;
    const float *mean, *invstd, *mask;
    int gate_after;
};

// Same convolution on the DMA / buffer-addressed kernel (conv1d_buf.hip).  `stats` (optional): [groups][2C] fp64 partial
// rows, *groups = rows written (<= conv1d_buf_max_groups()): per-channel (sum, sum of squares) of the stored output, or,
// with `tail`, (sum y*m, sum y*m*xhat) -- the partial rows bn_bwd_finalize_launch takes.
int conv1d_buf_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                      const void* gate, double* stats, int* groups, int dtype, int N, int H, int W, int C, int axis, int dilation,
                      int relu, hipStream_t stream, const Conv1dBnTail* tail = nullptr, int gate_bits = 0);
int conv1d_buf_max_groups();

// A whole non_bottleneck_1d block (four factorized convolutions, folded BatchNorms, residual) in one launch, bf16, dilation 1.
// packed4: the four mode-0 packed weights; tab: [4][scale | shift][C] fp32 in device memory.  ERF_EUNSUPPORTED for every other
// shape (nothing launched).
// C = 16: one workgroup per tile, three workgroups per CU (nb1d_block16.hip).
int nb1d_block16_launch(const void* x, void* y, const void* const* packed4, const float* tab, int dtype, int N, int H, int W, int C,
                        int dilation, hipStream_t stream);
// C = 64: a stream of rows through four wave roles (nb1d_block64s.hip).
int nb1d_block64s_launch(const void* x, void* y, const void* const* packed4, const float* tab, int dtype, int N, int H, int W, int C,
                         int dilation, hipStream_t stream);
