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
