// Weight repacking: fp32 parameters in torch layout -> MFMA A-operand fragments.
//
// Packed layout of one phase: [ct][kstep][lane 0..63][16 bytes].  Lane (i = lane & 15,
// q = lane >> 4) of (ct, kstep) holds, for output channel co = 16*ct + i, the 16-byte
// chunk g = 4*kstep + q of that channel's K vector, K = (tap, ci) with ci fastest:
// tap = g / CH, ci = (g % CH) * EPC + e, CH = Cin / EPC, EPC = 8 (bf16) or 4 (fp32).
// Rows co >= Cout and chunks g >= ntaps * CH are zero.
#include "common.cuh"
#include "pack.cuh"

template <typename T>
__global__ void pack_kernel(const PackParams p) {
    constexpr int EPC = Elem<T>::EPC;
    const int CH = p.cin / EPC;
    const long long total = (long long)p.ct * p.ksteps * 64;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long long rest = idx >> 6;
    const int ks = (int)(rest % p.ksteps);
    const int ct = (int)(rest / p.ksteps);
    const int co = ct * 16 + (lane & 15);
    const int g = ks * 4 + (lane >> 4);
    const int tap = g / CH, chunk = g % CH;
    int phase = co / p.ncp, cls = co % p.ncp;
    if (p.interleave4) { phase = (co & 15) >> 2; cls = (co >> 4) * 4 + (co & 3); }
    const bool ok = tap < p.ntaps && cls < p.nc_valid && co < p.cout_virtual;
    T out[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        float v = 0.f;
        if (ok) {
            const int ci = chunk * EPC + e;
            if (p.cin_valid == 0 || ci < p.cin_valid)
                v = p.src[(long long)cls * p.s_co + (long long)phase * p.s_phase + (long long)ci * p.s_ci + p.toff[tap]];
        }
        out[e] = Elem<T>::from_f32(v);
    }
    *((i32x4*)p.dst + idx) = *(const i32x4*)out;
}

// Many factorized-convolution weights in one launch.  table[4*e ..]: {fp32 weight pointer, packed pointer, C, mode}
// (mode 0 forward, 1 data-gradient), blockIdx.y = entry; same fragment layout as pack_kernel with ntaps = 3.
template <typename T>
__global__ void pack1d_many_kernel(const long long* __restrict__ table) {
    constexpr int EPC = Elem<T>::EPC;
    const long long* ent = table + 4 * (long long)blockIdx.y;
    const float* src = (const float*)ent[0];
    i32x4* dst = (i32x4*)ent[1];
    const int C = (int)ent[2], mode = (int)ent[3];
    const int CH = C / EPC, ksteps = (3 * CH + 3) / 4;
    const int total = (C / 16) * ksteps * 64;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = idx & 63, rest = idx >> 6;
    const int ks = rest % ksteps, ct = rest / ksteps;
    const int co = ct * 16 + (lane & 15);
    const int g = ks * 4 + (lane >> 4);
    const int tap = g / CH, chunk = g % CH;
    const long long s_co = mode == 0 ? 3LL * C : 3, s_ci = mode == 0 ? 3 : 3LL * C;
    const int toff = mode == 0 ? tap : 2 - tap;
    T out[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        float v = 0.f;
        if (tap < 3) v = src[co * s_co + (chunk * EPC + e) * s_ci + toff];
        out[e] = Elem<T>::from_f32(v);
    }
    dst[idx] = *(const i32x4*)out;
}

int pack1d_many_launch(const long long* table, int count, int max_c, int dtype, hipStream_t stream) {
    const int epc = dtype == 1 ? 8 : 4;
    const int total = (max_c / 16) * ((3 * (max_c / epc) + 3) / 4) * 64;
    const dim3 grid((total + 255) / 256, count);
    if (dtype == 1) hipLaunchKernelGGL(pack1d_many_kernel<bf16raw>, grid, dim3(256), 0, stream, table);
    else hipLaunchKernelGGL(pack1d_many_kernel<float>, grid, dim3(256), 0, stream, table);
    return erf_check_launch();
}

// The same re-pack for a whole table of descriptors (device memory), blockIdx.y = entry.
template <typename T>
__global__ void pack_many_kernel(const PackParams* __restrict__ table) {
    constexpr int EPC = Elem<T>::EPC;
    const PackParams& p = table[blockIdx.y];
    const int CH = p.cin / EPC;
    const long long total = (long long)p.ct * p.ksteps * 64;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int lane = (int)(idx & 63);
    const long long rest = idx >> 6;
    const int ks = (int)(rest % p.ksteps);
    const int ct = (int)(rest / p.ksteps);
    const int co = ct * 16 + (lane & 15);
    const int g = ks * 4 + (lane >> 4);
    const int tap = g / CH, chunk = g % CH;
    int phase = co / p.ncp, cls = co % p.ncp;
    if (p.interleave4) { phase = (co & 15) >> 2; cls = (co >> 4) * 4 + (co & 3); }
    const bool ok = tap < p.ntaps && cls < p.nc_valid && co < p.cout_virtual;
    T out[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        float v = 0.f;
        if (ok) {
            const int ci = chunk * EPC + e;
            if (p.cin_valid == 0 || ci < p.cin_valid)
                v = p.src[(long long)cls * p.s_co + (long long)phase * p.s_phase + (long long)ci * p.s_ci + p.toff[tap]];
        }
        out[e] = Elem<T>::from_f32(v);
    }
    *((i32x4*)p.dst + idx) = *(const i32x4*)out;
}

// Recording: between pack_record_begin and pack_record_end every pack_launch of the calling thread appends its
// descriptor to a host table instead of launching; the table (copied to the device by the caller) is replayed by
// pack_replay_launch in one launch.  Pointers inside the descriptors are the ones passed while recording.
static thread_local PackParams* t_rec = nullptr;
static thread_local int t_rec_cap = 0, t_rec_n = 0, t_rec_dtype = -1, t_rec_err = 0;
static thread_local long long t_rec_max = 0;

int pack_record_begin(void* host_table, int capacity) {
    if (!host_table || capacity <= 0 || t_rec) return ERF_EINVAL;
    t_rec = (PackParams*)host_table; t_rec_cap = capacity; t_rec_n = 0; t_rec_dtype = -1; t_rec_err = 0; t_rec_max = 0;
    return ERF_OK;
}

int pack_record_end(long long* max_chunks, int* dtype) {
    if (!t_rec) return ERF_EINVAL;
    t_rec = nullptr;
    if (t_rec_err) return t_rec_err;
    if (max_chunks) *max_chunks = t_rec_max;
    if (dtype) *dtype = t_rec_dtype;
    return t_rec_n;
}

int pack_replay_launch(const void* table_dev, int count, long long max_chunks, int dtype, hipStream_t stream) {
    if (!table_dev || count <= 0 || max_chunks <= 0) return ERF_EINVAL;
    const dim3 grid((unsigned)((max_chunks + 255) / 256), (unsigned)count);
    if (dtype == 1) hipLaunchKernelGGL(pack_many_kernel<bf16raw>, grid, dim3(256), 0, stream, (const PackParams*)table_dev);
    else hipLaunchKernelGGL(pack_many_kernel<float>, grid, dim3(256), 0, stream, (const PackParams*)table_dev);
    return erf_check_launch();
}

int pack_launch(const PackParams& p, int dtype, hipStream_t stream) {
    const long long total = (long long)p.ct * p.ksteps * 64;
    if (total <= 0) return ERF_EINVAL;
    if (t_rec) {
        if (t_rec_n >= t_rec_cap || (t_rec_dtype >= 0 && t_rec_dtype != dtype)) { t_rec_err = ERF_EINVAL; return ERF_EINVAL; }
        t_rec[t_rec_n++] = p;
        t_rec_dtype = dtype;
        if (total > t_rec_max) t_rec_max = total;
        return ERF_OK;
    }
    const int block = 256;
    const int grid = (int)((total + block - 1) / block);
    if (dtype == 1) hipLaunchKernelGGL(pack_kernel<bf16raw>, dim3(grid), dim3(block), 0, stream, p);
    else hipLaunchKernelGGL(pack_kernel<float>, dim3(grid), dim3(block), 0, stream, p);
    return erf_check_launch();
}
