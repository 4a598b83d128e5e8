"""bench.py's multi-rank plumbing, end to end on CPU: `python bench.py --gpus 2 --cpu-double ...` re-executes itself under
torch.distributed.run (127.0.0.1), the ranks rendezvous over gloo, run the step on the torch double of the C ABI (reduced-width
models), gather the frames, pass the output guards and rank 0 prints ONE JSON line with the driver's contract fields -- so the first
multi-GPU run of the driver cannot fail on launcher plumbing (the reference's live multi-GPU path: inference_cli.py:1127-1288).
Also: the refusals (test mode without its environment switch, WORLD_SIZE / --gpus disagreement)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")
CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def _run(args, env_extra=None, timeout=900):
    env = dict(os.environ, SVR_BENCH_ALLOW_CPU_DOUBLE="1", OMP_NUM_THREADS="2", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _line(proc):
    lines = [l for l in proc.stdout.splitlines() if l.startswith("{")]
    assert proc.returncode == 0 and len(lines) == 1, (proc.returncode, proc.stdout[-2000:], proc.stderr[-3000:])
    return json.loads(lines[0])


@pytest.mark.slow
def test_two_ranks_weak_scaling_line():
    res = _line(_run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "cfg3", "--cpu-double"]))
    for k in CONTRACT:
        assert k in res, k
    assert res["n_gpus"] == 2 and res["steps"] == 1 and res["warmup"] == 1 and res["scaling"] == "weak" and res["test_mode"] is True
    assert res["config"]["parallelism"] == "dp2" and res["higher_is_better"] is True and res["vs_baseline"] is None
    g = res["output_guard"]
    assert g["finite"] and g["deterministic"] is True
    # two ranks, each its own 4-frame share (5 padded frames, 4 kept), frames / max-over-ranks time
    assert abs(res["value"] - 2 * 4 * res["steps"] / (res["ms_per_step"] * 1e-3 * res["steps"])) < 1e-6 * res["value"]
    p = res["predicted_s"]
    assert p["per_step"] > 0 and "a priori" in p["model"] and p["measured_per_step"] > 0 and "measured_minus_predicted_s" in p
    r = res["ranks"]                                       # per-rank min / median / max of the step and its phases
    assert r["n"] == 2 and set(r["stats"]) >= {"step_ms", "encode_ms", "dit_ms", "decode_ms", "gather_ms"}
    assert r["stats"]["step_ms"]["min"] <= r["stats"]["step_ms"]["median"] <= r["stats"]["step_ms"]["max"]
    assert r["stats"]["step_ms"]["argmax_rank"] in (0, 1)
    assert "cpu_baseline" not in res                       # a 1-GPU line item


@pytest.mark.slow
def test_two_ranks_sharded_clip_line():
    res = _line(_run(["--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "cfg4", "--cpu-double"]))
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["test_mode"] is True
    assert res["output_guard"]["finite"] and res["output_guard"]["deterministic"] is True and "predicted_s" in res


def test_refusals():
    # the test mode is refused without its switch: the product path has no CPU fallback
    p = _run(["--cpu-double", "--steps", "1", "--warmup", "0"], env_extra={"SVR_BENCH_ALLOW_CPU_DOUBLE": "0"}, timeout=120)
    assert p.returncode == 2 and not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # a launcher's WORLD_SIZE and --gpus must agree (checked before the rendezvous)
    p = _run(["--gpus", "2", "--cpu-double"], env_extra={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"}, timeout=120)
    assert p.returncode == 2 and "must agree" in p.stderr
    # without GPUs and outside the test mode --gpus N refuses to report an N-GPU number
    p = _run(["--gpus", "2"], timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr
