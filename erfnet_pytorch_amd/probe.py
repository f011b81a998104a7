"""Measured peaks of the GPU this process runs on: the denominators beside the nominal ones in bench.py's roofline objects.

The reference times whole forward passes only (eval/eval_forwardTime.py:40-59); pricing a kernel against the hardware
needs to know what the hardware sustains, which `erf_probe_mfma` / `erf_probe_copy` (csrc/probe.hip) measure:
a matrix-instruction-only loop on every SIMD, and a 16-bytes-per-lane streaming copy.
"""
from __future__ import annotations

import torch

from ._lib import call, query


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _time_ms(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


SIMDS_PER_CU = 4
BF16_FLOP_PER_SIMD_CYCLE = 1024          # v_mfma_f32_16x16x32_bf16: 16384 flop in 16 cycles
F32_FLOP_PER_SIMD_CYCLE = 64             # v_mfma_f32_16x16x4_f32: 2048 flop in 32 cycles


def mfma_peak(dtype_code: int, dev, workgroups: int = 0, iters: int = 20000, reps: int = 3):
    """(sustained TFLOP/s, the shader clock in GHz that rate implies at the instruction's issue cost, the tick rate in GHz of
    s_memtime during the loop -- from the 100 MHz s_memrealtime beside it).  dtype_code 0 fp32, 1 bf16, 2 bf16 with operands
    that differ from lane to lane and register to register.  Default: sixteen workgroups of 8 waves per CU; a launch lasts a
    few milliseconds."""
    cu = torch.cuda.get_device_properties(dev).multi_processor_count
    workgroups = workgroups or 16 * cu
    sink = torch.zeros(4, device=dev)
    flop = query("erf_probe_mfma_flop", dtype_code, workgroups, iters)
    ms = _time_ms(lambda: call("erf_probe_mfma", dtype_code, workgroups, iters, sink.data_ptr(), _stream()), reps)
    rate = flop / (ms * 1e-3)
    per_cycle = cu * SIMDS_PER_CU * (BF16_FLOP_PER_SIMD_CYCLE if dtype_code != 0 else F32_FLOP_PER_SIMD_CYCLE)
    ticks = sink.cpu()
    tick_ghz = float(ticks[1] / ticks[2]) * 0.1 if float(ticks[2]) > 0 else 0.0
    return rate / 1e12, rate / per_cycle / 1e9, tick_ghz


def copy_peak(dev, nbytes: int = 1 << 30, reps: int = 5):
    """Streaming copy rate in GB/s counting bytes read + bytes written (a 1 GiB buffer is far beyond the 256 MB
    last-level cache).  The best of a few grid sizes."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    src.fill_(1)
    best = 0.0
    for wgs in (1024, 2048, 4096, 8192, 16384):
        ms = _time_ms(lambda: call("erf_probe_copy", src.data_ptr(), dst.data_ptr(), nbytes, wgs, _stream()), reps)
        best = max(best, 2.0 * nbytes / (ms * 1e-3) / 1e9)
    return best


def measured_peaks(dev):
    bf16, ghz_b, tick_b = mfma_peak(1, dev)
    var, ghz_v, tick_v = mfma_peak(2, dev)
    f32, ghz_f, _ = mfma_peak(0, dev, iters=10000)
    return {"mfma_bf16_tflops": bf16, "mfma_bf16_implied_clock_ghz": ghz_b,
            "mfma_bf16_varied_operands_tflops": var, "mfma_bf16_varied_operands_implied_clock_ghz": ghz_v,
            "mfma_f32_tflops": f32, "mfma_f32_implied_clock_ghz": ghz_f,
            "s_memtime_ghz": {"constant_operands": tick_b, "varied_operands": tick_v},
            "hbm_copy_gbs": copy_peak(dev),
            "how": "erf_probe_mfma: MFMA-only loop, 8 chains x 2 waves per SIMD on every CU (constant operands; varied: mantissas from a "
                   "hash, different in every lane and register); s_memtime_ghz: tick rate of s_memtime in that loop against the 100 MHz "
                   "s_memrealtime; erf_probe_copy: 1 GiB, 16 B per lane, read + write bytes"}
