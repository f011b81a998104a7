"""bench.py's one-line JSON contract, on small shapes: the keys the driver reads, the objects this repository adds
(roofline, roofline_block, cpu_baseline, inference_one_chain, training, fp32) at N = 1, and at N = 2 (rehearsal: both ranks on the one GPU, gloo)
the training step as the headline with the inference leg, the process group's world size and the per-bucket all-reduce
times beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BASE_KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
             "dtype", "data", "config"}


def _last_json(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_every_leg():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--train-steps", "2", "--train-warmup", "3",
           "--height", "64", "--width", "128", "--batch", "2", "--train-batch", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert BASE_KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "bf16" and d["unit"] == "img/s"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["world_size"] == 1
    assert "inference" in d["config"]["workload"] and d["config"]["hip_graph"] is True
    assert d["config"]["streams"] == 2                      # the batch as two chains of launches; the one-chain leg beside it
    one = d["inference_one_chain"]
    assert one["config"]["streams"] == 1 and one["value"] > 0 and one["dtype"] == "bf16"
    assert abs(d["value"] - 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf and rf["avg_launch_us"] > 0
    rb = d["roofline_block"]
    assert rb["launches_per_block"] in (1, 2) and rb["algorithmic_bytes_per_block"] * (1.0 if rb["launches_per_block"] == 1 else 2.5) == rb["launch_bytes_per_block"]
    assert 0 < rb["frac"] <= rf["frac"] * (1 + 1e-9)
    assert rf["traffic_from_profile"] is None or rf["traffic_from_profile"].startswith("profiles/")
    mp = d["measured_peak"]
    assert mp["mfma_bf16_tflops"] > 100 and mp["mfma_f32_tflops"] > 10 and mp["hbm_copy_gbs"] > 500
    assert abs(rf["frac_of_measured"] - rf["achieved"] / mp["hbm_copy_gbs"]) < 1e-9 and "frac_of_measured" in rb
    assert abs(d["mfma_frac"] - d["model_tflops"] / 2500.0) < 1e-9 and d["mfma_frac_of_measured"] > d["mfma_frac"]
    h2d = d["inference_with_h2d"]
    assert h2d["value"] > 0 and h2d["h2d_bytes_per_step"] == 2 * 4 * 64 * 128 and "uint8" in h2d["config"]["workload"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    tr = d["training"]
    assert tr["dtype"] == "bf16" and tr["config"]["hip_graph"] is True and tr["steps"] == 2 and tr["value"] > 0
    assert tr["config"]["final_loss"] == tr["config"]["final_loss"]                 # finite
    fp = d["fp32"]
    assert fp["inference"]["dtype"] == "fp32" and fp["training"]["dtype"] == "fp32" and fp["training"]["value"] > 0


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


BENCH_ARGS_2 = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--infer-steps", "2", "--infer-warmup", "1", "--height", "64",
                "--width", "128", "--train-batch", "2", "--batch", "2"]


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_two_rank_line_reports_training_as_the_headline(tmp_path, launcher):
    """`python bench.py --gpus 2` starts its two ranks itself (no launcher environment); under torch.distributed.run it joins
    the ranks it is given.  Either way: n_gpus == world_size == 2 and the gradient buckets travel under the backward pass."""
    env = dict(os.environ, ERF_BENCH_REHEARSAL="1", ERF_BENCH_REHEARSAL_LIMIT="200")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + BENCH_ARGS_2
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + BENCH_ARGS_2
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert BASE_KEYS <= set(d)
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert "training step" in d["config"]["workload"] and d["config"]["global_batch"] == 4
    assert d["config"]["hip_graph"] == "segments"             # five graphs, the collectives between their replays
    assert abs(d["value"] - 2 * 2 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    assert "REHEARSAL" in d["data"]
    inf = d["inference"]
    assert "inference" in inf["config"]["workload"] and inf["steps"] == 2 and inf["value"] > 0
    ar = d["allreduce"]
    assert ar["ranks"] == 2 and len(ar["buckets"]) == 3 and all(b["ms"] > 0 and b["bytes"] == 4 * b["elements"] for b in ar["buckets"])
    assert sum(b["elements"] for b in ar["buckets"]) <= 2_066_836
    assert ar["overlapped_with_backward"] is True and "exposed_ms" in ar and ar["step_ms_without_collectives"] > 0
    assert "cpu_baseline" not in d


def test_gpus_flag_must_match_the_process_group():
    """A launcher environment that disagrees with --gpus is an error, not a silent one-rank run."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)
