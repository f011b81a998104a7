#pragma once
#include <hip/hip_runtime.h>

int stem_launch(const void* x, int x_dtype, void* y, const float* w, const float* scale, const float* shift, int dtype,
                int N, int Hin, int Win, int relu, hipStream_t stream);
// Whole DownsamplerBlock in one launch (down.hip, bf16); ERF_EUNSUPPORTED: use the gather conv + pool_affine pair.
int down_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin,
                int Win, int Cin, int Cout, int relu, hipStream_t stream);
// Whole UpsamplerBlock in one launch (up.hip, bf16); ERF_EUNSUPPORTED: use the gather conv.
int up_launch(const void* x, void* y, const void* packed, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
              int Cin, int Cout, int relu, hipStream_t stream);
int pool_affine_launch(const void* x, void* y, const float* scale, const float* shift, int dtype, int N, int Hin, int Win,
                       int Cin, int ldy, int yoff, int relu, hipStream_t stream);
int transpose_launch(const void* src, int sdt, void* dst, int ddt, int N, long long R, long long Cc, hipStream_t stream);
int argmax_launch(const void* z, int dtype, void* lab, int label_bytes, int N, int C, long long HW, hipStream_t stream);
int iou_hist_launch(const void* pred, int pred_bytes, const void* label, int label_bytes, long long count, int nc,
                    int olabel, int nlabel, long long* hist, hipStream_t stream);
int labels_from_u8_launch(const void* src, long long* dst, long long count, int olabel, int nlabel, hipStream_t stream);
size_t iou_planes_workspace_bytes(int nc);
int iou_planes_launch(const void* x, int x_kind, const void* y, int y_kind, int N, int nc, long long HW, int ignore, double* sums,
                      double* workspace, hipStream_t stream);
int outconv_launch(const void* x, const void* packed, const float* bias, void* logits, int logits_f32, void* labels,
                   int label_bytes, int dtype, int N, int Hin, int Win, int nc, hipStream_t stream);
