"""The N > 1 path of bench.py on CPU: two gloo processes shard frame pairs and reduce time / counts."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from openfx_opencv_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.pairs_for_rank(n_pairs, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    elapsed = sharding.reduce_elapsed_max(1.0 + rank * 0.5, dist)          # rank 1 is the slow one
    total = sharding.reduce_count_sum(len(mine), dist)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, gathered, elapsed, total))


@pytest.mark.parametrize("n_pairs", [64, 7])
def test_two_ranks_partition_pairs_and_reduce(n_pairs):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, mine0, g0, e0, t0), (r1, mine1, g1, e1, t1) = res
    assert sorted(mine0 + mine1) == list(range(n_pairs)) and not set(mine0) & set(mine1)     # every pair exactly once
    assert mine0 == list(range(0, n_pairs, 2)) and mine1 == list(range(1, n_pairs, 2))      # pair i -> rank i mod 2
    assert g0 == g1 == [mine0, mine1]
    assert e0 == e1 == 1.5 and t0 == t1 == n_pairs                                            # max time, summed units


def test_sharding_helpers_single_process():
    sys.path.insert(0, ROOT)
    from openfx_opencv_amd import sharding
    assert sharding.pairs_for_rank(64, 3, 8) == [3, 11, 19, 27, 35, 43, 51, 59]               # BASELINE config 5: 8 pairs per GPU
    assert sum(len(sharding.pairs_for_rank(64, r, 8)) for r in range(8)) == 64
    assert sharding.pairs_for_rank(0, 0, 1) == [] and sharding.pairs_for_rank(3, 2, 4) == [2]
    assert [sharding.seed_for_pair(i) for i in (0, 63)] == [1234, 1297]
    assert sharding.reduce_elapsed_max(0.25) == 0.25 and sharding.reduce_count_sum(5) == 5
    with pytest.raises(ValueError):
        sharding.pairs_for_rank(4, 2, 2)
