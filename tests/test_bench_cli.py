"""CPU: the launcher logic of bench.py that needs no GPU -- `--gpus N` without a launcher must either spawn N ranks
or refuse loudly; it used to be parsed and ignored.  (The 2-rank run itself is tests/test_gpu_bench_multirank.py.)"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *flags):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_n_refuses_a_node_without_n_gpus():
    import torch

    if torch.cuda.device_count() >= 2:
        return   # (a multi-GPU node: the command would really run; covered by the GPU suite)
    r = _run({}, "--gpus", "2", "--steps", "2")
    assert r.returncode != 0 and "--gpus 2" in (r.stderr + r.stdout) and "GPU(s)" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])


def test_world_size_must_match_gpus():
    r = _run({"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "4", "--steps", "2")
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout), (r.returncode, r.stderr[-400:])
    r = _run({}, "--gpus", "0")
    assert r.returncode != 0 and ">= 1" in (r.stderr + r.stdout)
