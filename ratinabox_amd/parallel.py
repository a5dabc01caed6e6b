"""Agent-parallel sharding across the GPUs of one node.

Agents never interact on the accelerated path, so the step path has NO collective:
rank r owns a contiguous range of global agent ids, replicates the (tiny) wall and
cell tables, and keys the in-kernel Philox streams by GLOBAL agent id, which makes
every agent's trajectory, rates and spikes independent of the world size.  The only
communication offered is an optional all-gather of trajectory histories over
RCCL/xGMI (`torch.distributed`, backend "nccl" on ROCm; "gloo" on CPU for tests) —
firing-rate histories are deliberately kept sharded (BASELINE cfg 4 writes 67 MB per
step per GPU)."""
import torch


def shard_range(n_agents_total, rank, world_size):
    """(agent_id0, n_local) of rank `rank`: contiguous, sizes multiples of 4 (the kernels'
    agent-axis granule) except possibly the last rank, covering [0, n_agents_total)."""
    assert 0 <= rank < world_size
    per = -(-n_agents_total // world_size)  # ceil
    per = (per + 3) // 4 * 4
    a0 = min(rank * per, (n_agents_total + 3) // 4 * 4)  # empty trailing shards keep an aligned id
    a1 = min(a0 + per, n_agents_total)
    return a0, max(0, a1 - a0)


def sharded_agent_params(n_agents_total, rank=None, world_size=None, **params):
    """Params dict for `Agent(...)` holding this rank's shard of `n_agents_total` agents."""
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    a0, n = shard_range(n_agents_total, rank, world_size)
    if n == 0:
        raise ValueError(f"rank {rank} of {world_size} gets no agents out of {n_agents_total} (shards are multiples "
                         "of 4 agents): use fewer ranks or leave this rank out of the run")
    return dict(params, n_agents=n, agent_id0=a0)


def all_gather_trajectory(hist, n_local, group=None, single_rank_shortcut=True):
    """Concatenate per-rank trajectory histories `[T, 8, B_local_padded]` along the agent
    axis -> `[T, 8, sum(n_local)]` on every rank.  Off the step path: call it once per
    run / chunk.  Shards may differ in size (last rank), so rows are gathered at the
    largest padded width and trimmed.  A group of one rank needs no collective and returns its own rows
    (`single_rank_shortcut=False` runs the collectives anyway: the one-GPU test of the RCCL code path)."""
    import torch.distributed as dist
    if not dist.is_initialized() or (single_rank_shortcut and dist.get_world_size(group) == 1):
        return hist[..., :n_local].contiguous()
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=hist.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([n_local], dtype=torch.int64, device=hist.device), group=group)
    sizes = [int(s.item()) for s in sizes]
    width = (max(sizes) + 3) // 4 * 4
    mine = torch.zeros((*hist.shape[:-1], width), dtype=hist.dtype, device=hist.device)
    mine[..., :n_local] = hist[..., :n_local]
    if hist.is_cuda:
        out = torch.empty((world, *mine.shape), dtype=hist.dtype, device=hist.device)
        dist.all_gather_into_tensor(out, mine, group=group)  # one RCCL all-gather over xGMI
        parts = [out[r][..., :sizes[r]] for r in range(world)]
    else:
        bufs = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(bufs, mine, group=group)
        parts = [bufs[r][..., :sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=-1)
