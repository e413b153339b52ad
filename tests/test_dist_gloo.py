"""world_size-2 gloo test of the cross-rank arg-max used by the sharded loop-closure path."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_chains, out_dir):
    import torch
    import torch.distributed as dist
    from yag_slam_amd import dist as ymdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # every rank scores its shard with the same deterministic stand-in for the matcher
        rng = np.random.default_rng(5)
        resp = rng.uniform(0.1, 0.9, size=n_chains)
        if n_chains > 4:
            resp[[3, n_chains - 2]] = 0.95  # a tie across two different ranks
        lo, hi = ymdist.shard_range(n_chains, rank, world)
        rec = torch.full((ymdist.RECORD,), -1.0, dtype=torch.float64)
        if hi > lo:
            j = lo + int(np.argmax(resp[lo:hi]))
            rec = torch.tensor([resp[j], float(j), 1.0 + j, 2.0 + j, 0.1, 0.01, 0.02, 0.03], dtype=torch.float64)
        win, allrec = ymdist.all_gather_best(rec)
        np.save(os.path.join(out_dir, "win%d.npy" % rank), win.numpy())
        np.save(os.path.join(out_dir, "all%d.npy" % rank), allrec.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [9, 1])
def test_all_gather_best_two_ranks(tmp_path, n_chains):
    import torch.multiprocessing as mp
    from yag_slam_amd import dist as ymdist
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n_chains, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "win0.npy"), np.load(tmp_path / "win1.npy")
    assert np.array_equal(w0, w1)  # every rank agrees on the winner
    allrec = np.load(tmp_path / "all0.npy")
    assert ymdist.pick_best(allrec) >= 0
    assert np.array_equal(allrec[ymdist.pick_best(allrec)], w0)
    if n_chains == 9:
        assert w0[0] == 0.95 and w0[1] == 3.0  # tie -> lowest global chain id
    else:
        assert w0[1] == 0.0  # rank 1 had an empty shard
