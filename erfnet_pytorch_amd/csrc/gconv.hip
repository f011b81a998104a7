// Instantiations and host launcher of the gather-form MFMA convolution (gconv.cuh).
#include "gconv.cuh"

template <typename T, int CIN, int CT, int PT>
static int launch_one(const GConvParams& p, hipStream_t stream) {
    const long long P = (long long)p.N * p.Hb * p.Wb;
    const long long per_block = 4LL * 16 * PT;
    const long long gx = (P + per_block - 1) / per_block;
    if (gx <= 0 || gx > 0x7FFFFFFFLL) return ERF_EINVAL;
    hipLaunchKernelGGL((gconv_kernel<T, CIN, CT, PT>), dim3((unsigned)gx, p.nphase), dim3(256), 0, stream, p);
    return erf_check_launch();
}

#define GC_CASE(CIN, CT, PT)                                                      \
    if (cin == CIN && ct == CT)                                                   \
        return dtype == 1 ? launch_one<bf16raw, CIN, CT, PT>(p, stream) : launch_one<float, CIN, CT, PT>(p, stream);

int gconv_launch(const GConvParams& p, int dtype, int cin, hipStream_t stream) {
    const int ct = (p.Cout + 15) / 16;
    if (p.nphase < 1 || p.nphase > GC_MAX_PHASES) return ERF_EINVAL;
    GC_CASE(16, 1, 4)
    GC_CASE(16, 2, 4)
    GC_CASE(16, 3, 4)
    GC_CASE(16, 4, 4)
    GC_CASE(16, 8, 2)
    GC_CASE(32, 1, 4)
    GC_CASE(32, 8, 2)
    GC_CASE(48, 1, 4)
    GC_CASE(64, 1, 4)
    GC_CASE(80, 1, 4)
    GC_CASE(96, 1, 4)
    GC_CASE(112, 1, 4)
    GC_CASE(64, 4, 4)
    GC_CASE(64, 8, 2)
    GC_CASE(128, 1, 4)
    GC_CASE(128, 2, 4)
    GC_CASE(128, 4, 4)
    GC_CASE(128, 8, 2)
    return ERF_EUNSUPPORTED;
}
