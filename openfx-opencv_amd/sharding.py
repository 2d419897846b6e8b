"""Frame-pair sharding across the GPUs of one node.

Every render() / frame pair is independent (VectorGenerator.cpp:523-640 keeps no cross-frame state), so pair i goes
to rank i mod world and nothing is exchanged on the data path; the only collective is the max-reduce of the
wall-clock time that bench.py reports.  Kept free of GPU calls so the logic is covered by gloo tests on CPU.
"""


def pairs_for_rank(n_pairs, rank, world):
    """indices of the frame pairs rank `rank` of `world` processes (pair i -> rank i mod world)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world %r/%r" % (rank, world))
    return list(range(rank, n_pairs, world))


def seed_for_pair(pair_index, base=1234):
    """synthetic-input seed of a frame pair (BASELINE config 5: seeds 1234 ... 1297 for 64 pairs)."""
    return base + pair_index


def reduce_elapsed_max(elapsed, dist=None, device=None):
    """max over ranks of the per-rank elapsed seconds (identity when not distributed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch
    t = torch.tensor([float(elapsed)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reduce_count_sum(count, dist=None, device=None):
    """sum over ranks of the units (frame pairs) each rank processed."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return int(count)
    import torch
    t = torch.tensor([int(count)], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
