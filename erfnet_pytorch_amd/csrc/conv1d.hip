#include "common.cuh"
#include "conv1d.cuh"

int conv1d_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, const void* residual,
                  int dtype, int N, int H, int W, int C, int axis, int dilation, int relu, hipStream_t stream) {
    (void)x; (void)y; (void)packed; (void)scale; (void)shift; (void)residual; (void)dtype; (void)N; (void)H; (void)W;
    (void)C; (void)axis; (void)dilation; (void)relu; (void)stream;
    return ERF_EUNSUPPORTED;
}
