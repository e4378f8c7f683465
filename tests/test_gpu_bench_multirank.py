"""GPU: `python bench.py --gpus 2` launches its two ranks itself and runs the whole N > 1 path of the benchmark -- shards, the
staged exchange, the no-gather regions, the comparison of the three exchange modes -- end to end (the flag used to
be parsed and ignored, and no test executed bench.py with more than one rank).  Two processes share the one GPU of the driver's
box (`--single-device`), gloo carries the collectives; what the links do needs the 8-GPU node, what the command does does not.
Reference: autoencode.py:9-16 (one process per GPU under DDP), SURVEY.md section 8e."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], cwd=ROOT, env=env, capture_output=True,
                       text=True, timeout=timeout)
    return r


def test_bench_gpus_2_spawns_two_ranks_and_reports_the_exchange():
    r = _run("--gpus", "2", "--single-device", "--dist-backend", "gloo", "--steps", "4", "--warmup", "2", "--repeats", "1",
             "--settle", "2", "--prereplay", "2")
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines            # ONE JSON line, from rank 0
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["batch_per_gpu"] == 32
    assert res["config"]["parallelism"].startswith("dp2+allgather")
    ex = res["exchange"]
    assert ex["world_size"] == 2 and ex["bytes_in_per_rank_per_step"] == 32 * 47761 * 4
    assert ex["ms_per_step_no_gather"] > 0 and ex["value_no_gather"] > 0
    # whole-job value = both ranks' samples over the max-over-ranks region time
    assert abs(res["value"] - 2 * 32 * 47761 / (res["ms_per_step"] * 1e-3)) / res["value"] < 1e-6
    modes = ex["modes"]
    assert set(modes) == {"rccl_all_gather_per_step", "rccl_all_gather_staged_1", "peer_store"}, modes
    for name, m in modes.items():
        assert "error" not in m and m["ms_per_step"] > 0 and m["value"] > 0, (name, m)
    assert len(ex["flagged_utterances_per_rank"]) == 2


def test_bench_refuses_a_rank_count_it_cannot_have():
    r = _run("--gpus", "3", "--steps", "2", timeout=120)   # the driver's box has one GPU
    assert r.returncode != 0 and "GPU(s)" in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
    env_world = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2"], cwd=ROOT, env=env_world,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
