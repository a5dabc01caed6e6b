"""world_size-2 `gloo` test of the N>1 path on CPU: agent sharding by global id and the
optional all-gather of trajectory histories (the only collective; nothing on the step path)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ratinabox_amd import parallel


def test_shard_ranges_cover_and_align():
    for n, w in ((4096, 8), (32768, 8), (10, 2), (7, 4), (65536, 3)):
        got = [parallel.shard_range(n, r, w) for r in range(w)]
        assert got[0][0] == 0 and sum(c for _, c in got) == n
        for (a0, c), (b0, d) in zip(got[:-1], got[1:]):
            assert a0 + c == b0 or d == 0  # contiguous; empty trailing shards only keep the alignment
        assert all(a0 % 4 == 0 for a0, _ in got)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_total, T = 10, 3
    a0, n = parallel.shard_range(n_total, rank, world)
    p = parallel.sharded_agent_params(n_total, dt=0.01)
    assert p["n_agents"] == n and p["agent_id0"] == a0
    Bp = (n + 3) // 4 * 4
    hist = torch.zeros((T, 8, Bp))
    for t in range(T):
        for r in range(8):
            hist[t, r, :n] = torch.arange(a0, a0 + n) + 100 * r + 1000 * t  # value encodes (t, row, global id)
    full = parallel.all_gather_trajectory(hist, n)
    assert full.shape == (T, 8, n_total)
    expect = torch.arange(n_total)[None, None, :] + 100 * torch.arange(8)[None, :, None] + 1000 * torch.arange(T)[:, None, None]
    assert torch.equal(full, expect.to(full.dtype))
    dist.barrier()
    np.save(os.path.join(out_dir, f"ok_{rank}.npy"), np.array([1]))
    dist.destroy_process_group()


def test_all_gather_trajectory_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert all(os.path.exists(tmp_path / f"ok_{r}.npy") for r in range(2))
