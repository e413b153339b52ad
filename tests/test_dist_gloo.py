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


# ---- ShardedLoopMatcher itself, two ranks over gloo, with a stand-in for the GPU matcher --------------------------
class _StubBatch(object):
    """what MatchBatch offers ShardedLoopMatcher: run_async writes the shard's best record through the raw pointer
    (as argbest_kernel does on the device), wait returns per-chain results"""

    def __init__(self, responses):
        self.responses = list(responses)
        self.pending = {}

    def run_async(self, penalty, do_fine, slot, chain_id_base=0, dev_best_out=None):
        import ctypes as C
        j = int(np.argmax(self.responses))
        rec = [self.responses[j], float(chain_id_base + j), 10.0 + chain_id_base + j, 20.0, 0.5, 0.01, 0.02, 0.03]
        if dev_best_out:
            C.memmove(dev_best_out, (C.c_double * 8)(*rec), 64)
        self.pending[slot] = (chain_id_base, j)

    def wait(self, slot=0, per_chain=True):
        from collections import namedtuple
        R = namedtuple("R", "response")
        base, j = self.pending.pop(slot)
        per = [R(r) for r in self.responses]
        return per, per[j], j


class _StubMatcher(object):
    def __init__(self, all_responses):
        self.all_responses = all_responses
        self.made = []

    def make_batch(self, query, chains):
        b = _StubBatch([self.all_responses[c] for c in chains])  # a "chain" is its global index here
        self.made.append(b)
        return b

    def set_stream(self, stream):
        raise AssertionError("stream=False must keep the matcher's stream")


def _sharded_worker(rank, world, port, n_chains, out_dir):
    import torch
    import torch.distributed as dist
    from yag_slam_amd import dist as ymdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(11)
        resp = rng.uniform(0.1, 0.9, size=n_chains).tolist()
        if n_chains > 4:
            resp[2] = resp[n_chains - 1] = 0.97  # tie across the two shards
        chains = list(range(n_chains))
        rec = torch.zeros(ymdist.RECORD, dtype=torch.float64)
        # (1) constructed from the full chain list
        sh = ymdist.ShardedLoopMatcher(_StubMatcher(resp), "query", chains, rank, world, stream=False)
        assert (sh.lo, sh.hi) == ymdist.shard_range(n_chains, rank, world)
        win, allrec, per = sh.match(rec, False, False, slot=rank)
        assert (per is None) == (sh.hi == sh.lo)
        if per is not None:
            assert [p.response for p in per] == resp[sh.lo:sh.hi]
        # (2) constructed from this rank's shard only
        lo, hi = ymdist.shard_range(n_chains, rank, world)
        sh2 = ymdist.ShardedLoopMatcher.from_local_shard(_StubMatcher(resp), "query", chains[lo:hi], lo, n_chains, rank,
                                                         world, stream=False)
        rec2 = torch.zeros(ymdist.RECORD, dtype=torch.float64)
        sh2.run_async(rec2, False, False, slot=5)
        win2, _ = sh2.reduce(rec2)
        assert torch.equal(win, win2)
        np.save(os.path.join(out_dir, "swin%d.npy" % rank), win.numpy())
        np.save(os.path.join(out_dir, "sall%d.npy" % rank), allrec.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_chains", [11, 1])
def test_sharded_loop_matcher_two_ranks(tmp_path, n_chains):
    import torch.multiprocessing as mp
    from yag_slam_amd import dist as ymdist
    mp.spawn(_sharded_worker, args=(2, _free_port(), n_chains, str(tmp_path)), nprocs=2, join=True)
    w0, w1 = np.load(tmp_path / "swin0.npy"), np.load(tmp_path / "swin1.npy")
    assert np.array_equal(w0, w1)
    allrec = np.load(tmp_path / "sall0.npy")
    if n_chains == 11:
        assert w0[0] == 0.97 and w0[1] == 2.0 and w0[2] == 12.0   # tie -> lowest global chain id, payload follows
        assert allrec[1, 1] == 10.0                                # rank 1's best carries its global id
    else:
        assert w0[1] == 0.0 and allrec[1, 1] == -1.0               # rank 1's shard is empty
    with pytest.raises(ValueError):
        ymdist.ShardedLoopMatcher.from_local_shard(_StubMatcher([0.1] * 4), "q", [0, 1, 2], 0, 4, 0, 2, stream=False)


# ---- AngleSplitMatcher (one match split by coarse angle), two ranks over gloo, stand-in matcher on CPU tensors -----
class _StubSliceMatcher(object):
    """slice_begin fills this rank's angles of the response volume and its own per-cell maxima through the raw
    pointers, like the device kernels; slice_finish reads the whole volume back"""
    NX, NY, NT = 5, 4, 7

    @staticmethod
    def value(k, c):
        return ((k * 37 + c * 11) % 23) / 23.0

    def coarse_dims(self):
        return (self.NX, self.NY, self.NT)

    def slice_begin(self, query, base, penalty, do_fine, k0, k1, dev_resp, dev_probs):
        import ctypes as C
        nxy = self.NX * self.NY
        self.ptrs = (dev_resp, dev_probs)
        probs = [0.0] * nxy
        for k in range(k0, k1):
            row = [self.value(k, c) for c in range(nxy)]
            C.memmove(dev_resp + 8 * k * nxy, (C.c_double * nxy)(*row), 8 * nxy)
            probs = [max(a, b) for a, b in zip(probs, row)]
        C.memmove(dev_probs, (C.c_double * nxy)(*probs), 8 * nxy)

    def slice_finish(self):
        import ctypes as C
        nxy = self.NX * self.NY
        resp = np.ctypeslib.as_array((C.c_double * (self.NT * nxy)).from_address(self.ptrs[0])).copy()
        probs = np.ctypeslib.as_array((C.c_double * nxy).from_address(self.ptrs[1])).copy()
        return resp, probs


def _angle_worker(rank, world, port, out_dir):
    import torch.distributed as dist
    from yag_slam_amd import dist as ymdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sp = ymdist.AngleSplitMatcher(_StubSliceMatcher(), rank, world, stream=False)
        assert (sp.k0, sp.k1) == ((0, 4) if rank == 0 else (4, 7))
        resp, probs = sp.match_scan("q", ["b"], True, True)
        np.save(os.path.join(out_dir, "aresp%d.npy" % rank), resp)
        np.save(os.path.join(out_dir, "aprobs%d.npy" % rank), probs)
    finally:
        dist.destroy_process_group()


def test_angle_split_matcher_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_angle_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    S = _StubSliceMatcher
    nxy = S.NX * S.NY
    want = np.array([[S.value(k, c) for c in range(nxy)] for k in range(S.NT)])
    for r in (0, 1):
        resp, probs = np.load(tmp_path / ("aresp%d.npy" % r)), np.load(tmp_path / ("aprobs%d.npy" % r))
        assert np.array_equal(resp.reshape(S.NT, nxy), want)        # both ranks hold the whole volume
        assert np.array_equal(probs, want.max(axis=0))              # and the maxima over ALL angles
