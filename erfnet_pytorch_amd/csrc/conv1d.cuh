#pragma once
#include <hip/hip_runtime.h>

// LDS-tiled factorized 3x1 / 1x3 convolution (the hot kernel).  Returns ERF_EUNSUPPORTED for
// shapes it does not take; the caller then uses the gather-form kernel.
int conv1d_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                  const void* gate, int dtype, int N, int H, int W, int C, int axis, int dilation, int relu, hipStream_t stream);
