"""CPU: the C restatement (oracle/golf_oracle.c) against the pinned numpy oracle, the timed CPU-port baseline,
and the multi-rank sharding/gather path under gloo (world_size 2)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

from conftest import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_oracle_matches_numpy_oracle():
    from golf_amd.synthetic import make_inputs
    from oracle import cpu_baseline as CB
    from oracle import golf_oracle as O

    inp = make_inputs(B=2, T=9600)
    ex, gain, a = inp["noise"], inp["gain"], inp["a"]
    ref = O.ltv_allpole_ss_forward(ex.numpy(), gain.numpy(), a.numpy(), 240)
    y64 = CB.ltv_ss_c(ex.double(), gain.double(), a.double(), 240).numpy()
    assert rel_err(y64, ref)[0] < 1e-12
    y32 = CB.ltv_ss_c(ex, gain, a, 240).numpy()
    assert rel_err(y32, ref)[0] < 1e-4  # the fp32 reference path itself
    gy = torch.randn(2, ref.shape[1], dtype=torch.float64)
    g = CB.ltv_ss_bwd_c(gy, torch.from_numpy(ref), ex, gain, a, 240)
    r = O.ltv_allpole_ss_backward(gy.numpy(), ex.numpy(), gain.numpy(), a.numpy(), 240)
    assert rel_err(g[0].numpy(), r[0][:, : ref.shape[1]])[0] < 1e-11
    assert rel_err(g[1].numpy(), r[1])[0] < 1e-11
    assert rel_err(g[2].numpy(), r[2])[0] < 1e-11
    # per-row LTI filter vs numpy oracle
    x = np.random.default_rng(0).normal(0, 1, (5, 300))
    aa = a[0, :5].double().numpy()
    out = np.empty_like(x)
    import ctypes

    CB.lib().golf_oracle_lfilter_rows_f64(ctypes.c_void_p(x.ctypes.data), ctypes.c_void_p(aa.ctypes.data),
                                          ctypes.c_void_p(out.ctypes.data), 5, 300, 22)
    assert rel_err(out, O.lfilter_allpole(x, aa))[0] < 1e-12


def test_cpu_port_reproduces_reference_op_structure():
    """The timed baseline (reference op sequence, fp32) agrees with the float64 oracle to fp32-path accuracy on a
    short clip (long clips drift through the reference's fp32 cumsum)."""
    from golf_amd.synth import DownsampledIndexedGlottalFlowTable
    from golf_amd.synthetic import make_inputs
    from oracle import cpu_baseline as CB
    from oracle import golf_oracle as O

    inp = make_inputs(B=2, T=2400)
    osc = DownsampledIndexedGlottalFlowTable(hop_rate=10, in_channels=8, oversampling=4, equal_energy=True, lf_v2=True,
                                             points=2048)
    y = CB.golf_ss_synth_cpu(inp, osc.table, osc.decimater.kernel).numpy()
    _, ref = O.source_filter_ss(inp["phase"].numpy(), 1, inp["wsel"].numpy(), inp["w_hop"], osc.table.numpy(),
                                inp["noise"].numpy(), inp["gain"].numpy(), inp["a"].numpy(), 240, 4, True,
                                osc.decimater.taps.numpy())
    assert y.shape == ref.shape
    assert rel_err(y, ref)[0] < 2e-3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist

    from golf_amd.dist import synth_sharded
    from oracle import cpu_baseline as CB

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from golf_amd.synthetic import make_inputs

    inp = make_inputs(B=total, T=2400)
    # the CPU oracle stands in for the HIP kernels here: what is under test is sharding + gather
    fn = lambda loc: CB.ltv_ss_c(loc["noise"], loc["gain"], loc["a"], 240)
    y = synth_sharded(fn, inp)
    # pipelined (double-buffered) gather, as bench.py / a serving loop uses it
    from golf_amd.dist import gather_audio_async, shard_inputs

    loc = fn(shard_inputs(inp, rank, world))
    bufs = [torch.empty(world * loc.shape[0], loc.shape[1]) for _ in range(2)]
    pending = []
    for k in range(3):
        while len(pending) > 1:
            pending.pop(0)[0].wait()
        src = loc + k
        pending.append((gather_audio_async(src, bufs[k & 1]), src))
    while pending:
        pending.pop(0)[0].wait()
    assert torch.equal(bufs[0][:total], y + 2) and torch.equal(bufs[1][:total], y + 1)
    if rank == 0:
        q.put(y.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [4, 5])  # 5: ragged tail, padded then trimmed
def test_sharded_synthesis_gloo_world2(total):
    import torch.multiprocessing as mp

    from golf_amd.synthetic import make_inputs
    from oracle import cpu_baseline as CB

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    inp = make_inputs(B=total, T=2400)
    ref = CB.ltv_ss_c(inp["noise"], inp["gain"], inp["a"], 240).numpy()
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got, ref)  # sharding must not change a single bit


def test_shard_helpers():
    from golf_amd.dist import gather_audio, shard_bounds, shard_inputs

    assert shard_bounds(256, 3, 8) == (96, 128, 32)
    assert shard_bounds(5, 1, 2) == (3, 5, 3)
    d = {"x": torch.arange(10).view(5, 2), "hop": 240}
    s0, s1 = shard_inputs(d, 0, 2), shard_inputs(d, 1, 2)
    assert s0["x"].shape == (3, 2) and s1["x"].shape == (3, 2) and s1["hop"] == 240
    assert torch.equal(s1["x"][2], d["x"][4])  # padded with the last utterance
    y = torch.ones(3, 4)
    assert gather_audio(y, total=2).shape == (2, 4)  # no process group: identity + trim


def _staged_worker(rank, world, port, q):
    import torch.distributed as dist
    from golf_amd.dist import StagedGather

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, T, every, steps = 3, 17, 2, 5
    sg = StagedGather(rows, T, every, device="cpu")
    seen = []
    for k in range(steps):
        y = torch.full((rows, T), float(100 * rank + k))
        h = sg.push(y)
        if h is not None:                                     # a full group went out: steps k-1, k
            h.wait()
            seen.append(sg.result(1 - sg.cur).clone())        # the pair just sent
    sg.flush()                                                # the half-filled third group (step 4): 1 step is sent
    assert sg.last_fill == 1
    part = sg.result(1 - sg.cur, steps=1)                     # (world, 1, rows, T)
    seen.append(torch.cat([part, torch.full_like(part, -1.0)], 1).clone())
    if rank == 0:
        q.put((torch.stack(seen).numpy(), sg.groups_sent, sg.bytes_per_collective))
    dist.barrier()
    dist.destroy_process_group()


def test_staged_gather_gloo_world2():
    """K steps' audio per collective (the N > 1 mitigation for the per-link xGMI bound): every rank sees every rank's
    output of every step, grouped, double buffered, with a ragged last group."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_staged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    seen, groups, nbytes = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert groups == 3 and nbytes == 2 * 3 * 17 * 4
    assert seen.shape == (3, 2, 2, 3, 17)                     # (group, rank, step in group, rows, T)
    for grp in range(3):
        for r in range(2):
            for j in range(2):
                k = 2 * grp + j
                if k < 5:
                    assert np.all(seen[grp, r, j] == 100 * r + k), (grp, r, j)


def test_staged_gather_single_process():
    from golf_amd.dist import StagedGather

    sg = StagedGather(2, 5, 3, device="cpu", world=1)
    for k in range(3):
        sg.push(torch.full((2, 5), float(k)))
    sg.flush()
    assert torch.equal(sg.result(0)[0, :, 0, 0], torch.tensor([0.0, 1.0, 2.0]))
