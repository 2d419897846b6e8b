"""bench.py as its own launcher (`--gpus N` without torch.distributed.run): N ranks are spawned, rendezvous over 127.0.0.1, shard the pairs and
reduce time / counts, and the line reports the n_gpus that ran.  BENCH_DRY_RUN=1 keeps the GPU out of it (the timed work is a sleep)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, expect_ok=True):
    env = dict(os.environ, BENCH_DRY_RUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=300)
    if expect_ok:
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        return json.loads(lines[0])
    return r


def test_gpus_flag_spawns_that_many_ranks():
    line = _run(["--gpus", "2"])
    assert line["n_gpus"] == 2 and line["pairs"] == 16            # 8 pairs per rank (weak scaling)
    assert line["seeds_rank0"] == [1234 + i for i in range(0, 16, 2)]   # pair i -> rank i mod world


def test_config_5_is_the_4k_batch_of_eight():
    line = _run(["--gpus", "2", "--config", "5"])
    assert line["size"] == "3840x2160" and line["batch"] == 8 and line["n_gpus"] == 2


def test_single_rank_and_launcher_mismatch():
    assert _run([])["n_gpus"] == 1
    r = _run(["--gpus", "4"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"}, expect_ok=False)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
