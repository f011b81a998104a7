#pragma once
#include <hip/hip_runtime.h>

struct PackParams {
    const float* src;
    void* dst;
    int cin;
    int cin_valid;          // input channels >= cin_valid are zero / not written (0 = all valid)
    int cout_virtual;       // rows that exist (others are zero)
    int ncp;                // rows per weight "phase" (virtual channel = phase * ncp + cls)
    int nc_valid;           // cls < nc_valid are real output channels
    long long s_co, s_ci, s_phase;   // element strides in src
    int ntaps;
    int toff[9];            // element offset of each tap in src
    int ct, ksteps;
    int interleave4;        // rows ordered [class group of 4][phase 0..3][class in group] (16 per tile) instead of phase-major
};

int pack_launch(const PackParams& p, int dtype, hipStream_t stream);
int pack_record_begin(void* host_table, int capacity);
int pack_record_end(long long* max_chunks, int* dtype);
int pack_replay_launch(const void* table_dev, int count, long long max_chunks, int dtype, hipStream_t stream);
int pack1d_many_launch(const long long* table, int count, int max_c, int dtype, hipStream_t stream);
