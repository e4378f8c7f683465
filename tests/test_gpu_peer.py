"""GPU: the peer-store exchange (golf_amd.dist.PeerStoreGather on golf_peer_*) with TWO processes sharing ONE GPU --
what a 1-GPU box can check: allocation / IPC export / mapping in another process, the slot ring with acknowledgements,
producer loops that never read, and the bounded wait.  Link behaviour needs a multi-GPU node and is not covered."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

ROWS, T = 3, 1001


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _block(rank, step, device):
    base = torch.arange(ROWS * T, dtype=torch.float32, device=device).view(ROWS, T) * 1e-3
    return base + (100.0 * rank + step)


def _worker(rank, world, port, scenario, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from golf_amd.dist import PeerStoreGather

        dev = torch.device("cuda:0")
        torch.cuda.set_device(dev)
        if scenario == "consume":          # every step is waited for, compared and released; 7 steps through 2 slots
            g = PeerStoreGather(ROWS, T, depth=2, device=dev)
            for k in range(7):
                g.push(_block(rank, k, dev))
                got = g.wait().clone()
                g.release()
                torch.cuda.synchronize()
                for r in range(world):
                    assert torch.equal(got[r], _block(r, k, dev)), (rank, k, r)
            g.flush()
            g.close()
        elif scenario == "produce":        # a producer loop that never looks: push consumes the oldest step itself
            g = PeerStoreGather(ROWS, T, depth=3, device=dev)
            n = 11
            for k in range(n):
                g.push(_block(rank, k, dev))
            assert g.pushed - g.released <= 3
            g.flush()                      # everything pushed anywhere has arrived and been released
            torch.cuda.synchronize()
            dist.barrier()
            last = g.recv.clone()
            for k in range(n - 3, n):
                for r in range(world):
                    assert torch.equal(last[k % 3, r], _block(r, k, dev)), (rank, k, r)
            g.close()
        elif scenario == "timeout":        # rank 1 never pushes: rank 0's wait gives up and reports who was missing
            g = PeerStoreGather(ROWS, T, depth=2, device=dev, timeout_s=0.3)
            if rank == 0:
                g.push(_block(rank, 0, dev))
                g.wait()
                try:
                    g.check()
                    raise AssertionError("expected a timeout")
                except RuntimeError as e:
                    assert "rank 1" in str(e)
            g.close()
        q.put((rank, "ok"))
    except BaseException as e:  # noqa: BLE001 -- reported to the parent, which fails the test
        q.put((rank, f"{type(e).__name__}: {e}"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["consume", "produce", "timeout"])
def test_peer_store_two_processes_one_gpu(scenario):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in procs:
            r, msg = q.get(timeout=120)
            results[r] = msg
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert results == {0: "ok", 1: "ok"}, results
