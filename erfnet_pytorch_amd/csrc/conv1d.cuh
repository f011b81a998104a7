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

// Same convolution on the DMA / buffer-addressed kernel (conv1d_buf.hip).  `stats` (optional): [groups][2C] fp64 partial
// rows (per-channel sum and sum of squares of the stored output), *groups = rows written (<= conv1d_buf_max_groups()).
int conv1d_buf_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                      const void* gate, double* stats, int* groups, int dtype, int N, int H, int W, int C, int axis, int dilation,
                      int relu, hipStream_t stream);
int conv1d_buf_max_groups();
