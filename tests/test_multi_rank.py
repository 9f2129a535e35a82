"""CPU, world_size 2 (gloo): the N>1 path bench.py uses -- owner-computes shard
ranges (apm_shard_range via sharding.rank_shard) + one all-reduce of the partial
counts.  The per-shard scan is stood in for by the oracle restricted to the rank's
own range of window starts (no GPU here); the result must equal the reference's
whole-text counts, including at seams where the reference's own DB_OVER_RANKS
over-counts (src/database_over_ranks.c:339-343)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case_names, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import importlib
    import helpers as HH
    sharding = importlib.import_module(HH.PKG_NAME + ".sharding")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ok = True
        for name in case_names:
            c = next(c for c in HH.golden()["cases"] if c["name"] == name)
            text = HH.case_text(c)
            n, k = len(text), c["k"]
            m_max = max(len(p) for p in c["patterns"])
            ob, oe, lo, hi = sharding.rank_shard(n, k, m_max, rank, world)
            # the rank only needs text[lo:hi]; windows past hi would be a halo bug
            local = text[lo:hi]
            part = []
            for p in c["patterns"]:
                # same call shape as apm_count_shard_device: shard bytes + global coordinates
                r = 0
                m = len(p)
                for j in range(ob, min(oe, max(0, n - k))):
                    size = min(m, n - j)
                    assert j - lo + size <= len(local), "halo too short"
                    r += HH.window_distance(p[:size], local[j - lo:j - lo + size]) <= k
                part.append(r)
            t = torch.tensor(part, dtype=torch.int64)
            sharding.allreduce_counts(t)
            ok = ok and (t.tolist() == c["counts"])
        ret[rank] = ok
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_counts_allreduce_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    names = ["chrY_k3", "A16_k0", "m_gt_n_k3", "rand_abc_nl_1", "newline_k1"]
    mp.spawn(_worker, args=(world, _free_port(), names, ret), nprocs=world, join=True)
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_rank_shard_covers_text_once():
    import importlib
    sharding = importlib.import_module(H.PKG_NAME + ".sharding")
    for n, k, m_max, world in [(1 << 20, 0, 32, 8), (1000, 5, 50, 3), (40, 0, 64, 4), (3, 5, 4, 2)]:
        prev = 0
        for r in range(world):
            ob, oe, lo, hi = sharding.rank_shard(n, k, m_max, r, world)
            assert ob == prev and lo == ob and hi <= n
            if oe > ob:
                assert hi == min(n, oe + m_max - 1)
            prev = oe
        assert prev == max(0, n - k)
