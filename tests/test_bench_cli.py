"""bench.py's launch contract, checked without a GPU: `--gpus N` must never silently report a 1-GPU number
(VERDICT round 1, weak item 10)."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def test_multi_gpu_request_without_gpus_fails_loudly():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode != 0 and "refusing to report" in r.stderr and r.stdout.strip() == ""


def test_world_size_and_gpus_must_agree():
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                                                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode != 0 and r.stdout.strip() == ""


def test_every_args_attribute_bench_reads_is_a_declared_flag():
    """bench.py cannot be executed without a GPU, so a flag that main() reads but the parser does not declare (round 4 lost two
    bench runs of a GPU call to exactly that) is caught statically: every ``args.<name>`` in the source is an ``add_argument`` dest."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    declared, used = set(), set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            flag = next(a.value for a in node.args if isinstance(a, ast.Constant) and str(a.value).startswith("--"))
            declared.add(flag[2:].replace("-", "_"))
        if isinstance(node, ast.Attribute) and isinstance(node.value, ast.Name) and node.value.id == "args":
            used.add(node.attr)
    assert used and used <= declared, sorted(used - declared)
