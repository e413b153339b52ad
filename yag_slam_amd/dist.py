"""The matcher across the GPUs of one node (one process per GPU, torch.distributed; "nccl" is RCCL on ROCm).

Two splits: loop-closure BATCHES shard by chain (ShardedLoopMatcher, below: no data-path
collective, one all-gather of a 64-byte record), and ONE huge match shards by coarse angle (AngleSplitMatcher, at the
end: the grid is rasterised on every rank, the response volume is all-gathered).

Loop-closure batches:

The reference closes loops by matching one query scan against candidate chains one after another
(/root/reference/yag_slam/graph_slam.py:217-254).  Chains are independent problems, so they shard
across ranks with no data-path exchange; the only collective is one RCCL all-gather of a 64-byte
"best of my shard" record per rank (backend "nccl" is RCCL on ROCm; "gloo" in the CPU tests),
followed by a deterministic local arg-max (highest response, ties to the lowest global chain id).
"""
import numpy as np

RECORD = 8  # doubles: response, global chain id, x, y, heading, cov_xx, cov_yy, cov_tt


def shard_range(n_chains, rank, world):
    """Contiguous block of chain indices [lo, hi) owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_chains), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pick_best(records):
    """records: (world, RECORD) array-like -> winning row index (max response, then min chain id).
    Rows with a negative chain id (empty shard) never win."""
    r = np.asarray(records, dtype=np.float64).reshape(-1, RECORD)
    best = -1
    for i in range(r.shape[0]):
        if r[i, 1] < 0:
            continue
        if best < 0 or r[i, 0] > r[best, 0] or (r[i, 0] == r[best, 0] and r[i, 1] < r[best, 1]):
            best = i
    return best


def _staged_through_host(tensor, group=None):
    """True when the collective has to run on a host copy: a device tensor under a backend that moves host memory ("gloo").
    That is how two ranks share ONE GPU in the tests (RCCL refuses two ranks on one device); on a node with a GPU per rank the
    backend is "nccl" (= RCCL) and the tensors never leave the device."""
    import torch.distributed as dist
    return tensor.is_cuda and dist.get_backend(group) == "gloo"


def all_gather_best(local_record, group=None):
    """local_record: torch tensor (RECORD,) float64 on the rank's device (cuda for nccl, cpu for gloo).
    Returns (winner_record_tensor, gathered (world, RECORD) tensor)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):  # a single process: its record is the winner
        out = local_record.contiguous().view(1, RECORD)
        return out[0], out
    if _staged_through_host(local_record, group):
        local_record = local_record.cpu()  # (waits for the current stream: the arg-best kernel has written the record)
    world = dist.get_world_size(group)
    flat = torch.empty(world * RECORD, dtype=torch.float64, device=local_record.device)
    dist.all_gather_into_tensor(flat, local_record.contiguous().view(-1), group=group)
    out = flat.view(world, RECORD)
    resp, cid = out[:, 0], out[:, 1]
    valid = cid >= 0
    # lexicographic arg-max on the device: response descending, then chain id ascending
    top = torch.where(valid, resp, torch.full_like(resp, -1.0)).max()
    cand = torch.where(valid & (resp == top), cid, torch.full_like(cid, float("inf")))
    win = torch.argmin(cand)
    return out[win], out


class ShardedLoopMatcher(object):
    """Each rank owns a ScanMatcher on its GPU and a contiguous shard of the candidate chains.

    The matcher is moved onto a torch stream (`stream`, default: torch's current stream) and the collective is issued
    under that stream: RCCL orders itself against it, so the record the arg-best kernel writes is complete before the
    all-gather reads it (the matcher's own stream is non-blocking and would not be ordered with anything torch does).
    `stream=False` keeps the matcher's stream (CPU tests with a stand-in matcher)."""

    def __init__(self, matcher, query, chains, rank, world, stream=None):
        self.matcher = matcher
        self.rank, self.world = rank, world
        self.n_chains = len(chains)
        self.lo, self.hi = shard_range(len(chains), rank, world)
        self.torch_stream = None
        if stream is not False:
            import torch
            if stream is None:
                stream = torch.cuda.current_stream()
                if stream.cuda_stream == 0:
                    # torch's default stream is the null stream: its handle (0) is how ym_set_stream spells "the matcher's
                    # own stream".  Work on a side stream instead; reduce() makes the caller's stream wait for it.
                    stream = torch.cuda.Stream()
            self.torch_stream = stream
            matcher.set_stream(stream.cuda_stream)
        self.batch = matcher.make_batch(query, chains[self.lo:self.hi]) if self.hi > self.lo else None

    @classmethod
    def from_local_shard(cls, matcher, query, local_chains, lo, n_chains, rank, world, stream=None):
        """The same, for a rank that only holds ITS chains: local_chains = chains[lo:hi] of the n_chains candidates,
        with (lo, hi) = shard_range(n_chains, rank, world)."""
        lo_, hi_ = shard_range(n_chains, rank, world)
        if lo != lo_ or lo + len(local_chains) != hi_:
            raise ValueError("shard [%d, %d) is not rank %d's block [%d, %d)" % (lo, lo + len(local_chains), rank, lo_, hi_))
        self = cls(matcher, query, [], rank, world, stream)
        self.n_chains, self.lo, self.hi = n_chains, lo_, hi_
        self.batch = matcher.make_batch(query, local_chains) if local_chains else None
        return self

    def _on_stream(self):
        import contextlib
        if self.torch_stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.torch_stream)

    def run_async(self, record, penalty=False, do_fine=False, slot=0):
        """Enqueue the local shard; `record` (torch float64[RECORD] on this GPU) receives the shard's best.
        The record is the one written BEFORE Karto's response expansion (see include/yagmatch.h); `match` below
        is the form that is exact in that case too."""
        if self.batch is None:
            with self._on_stream():
                record.fill_(-1.0)
            return
        self.batch.run_async(penalty, do_fine, slot, chain_id_base=self.lo, dev_best_out=record.data_ptr())

    def reduce(self, record, group=None):
        """all-gather of the ranks' records + arg-max, ordered after this rank's arg-best kernel"""
        with self._on_stream():
            out = all_gather_best(record, group)
        if self.torch_stream is not None:
            import torch
            torch.cuda.current_stream().wait_stream(self.torch_stream)
        return out

    def match(self, record, penalty=False, do_fine=False, slot=0, group=None):
        """The whole loop-closure step: local shard, wait (so that a response expansion has rewritten the record),
        cross-rank arg-max.  Returns (winner record, gathered records, local per-chain results or None)."""
        self.run_async(record, penalty, do_fine, slot)
        per = None
        if self.batch is not None:
            per, _, _ = self.batch.wait(slot)
        win, allrec = self.reduce(record, group)
        return win, allrec, per


class AngleSplitMatcher(object):
    """ONE match over `world` GPUs (BASELINE configs[4]: 201 x 201 x 46 coarse hypotheses): every rank rasterises the same
    grid and scores its block of coarse angles; the response slices are all-gathered in place (RCCL), the per-(x, y)
    maxima (Karto's search-space probability grid) reduced with MAX; arg-max, tie mean, covariances and the fine pass then
    run on the whole volume on every rank.  Integer sums and fp64 responses do not depend on who computed them, and
    the finish stage is the single-GPU one, so every rank returns the bits a single-GPU match_scan returns.

    The exchange is one all-gather of nt_pad * ny * nx doubles (15 MB for configs[4]) + one all-reduce of ny * nx doubles
    (0.3 MB) per match; on xGMI that costs about as much as the 1/world of the correlate it saves on that config, so this
    is for lattices larger than one GPU wants to score alone, not a latency win on configs[4] itself."""

    def __init__(self, matcher, rank, world, group=None, stream=None):
        import torch
        self.matcher, self.rank, self.world, self.group = matcher, rank, world, group
        nx, ny, nt = matcher.coarse_dims()
        self.nxy, self.nt = nx * ny, nt
        self.per = (nt + world - 1) // world                   # angles per rank; the last ranks may get fewer, or none
        self.k0 = min(nt, rank * self.per)
        self.k1 = min(nt, self.k0 + self.per)
        if stream is None:
            stream = torch.cuda.current_stream()
            if stream.cuda_stream == 0:                        # see ShardedLoopMatcher
                stream = torch.cuda.Stream()
        self.torch_stream = stream
        if stream is not False:
            matcher.set_stream(stream.cuda_stream)
        dev = "cpu" if stream is False else "cuda"
        self.resp = torch.zeros(world * self.per * self.nxy, dtype=torch.float64, device=dev)
        self.probs = torch.zeros(self.nxy, dtype=torch.float64, device=dev)

    def match_scan(self, query, base_scans, penalty=True, do_fine=False):
        import contextlib
        import torch
        import torch.distributed as dist
        self.matcher.slice_begin(query, base_scans, penalty, do_fine, self.k0, self.k1, self.resp.data_ptr(), self.probs.data_ptr())
        ctx = contextlib.nullcontext() if self.torch_stream is False else torch.cuda.stream(self.torch_stream)
        with ctx:
            if self.world > 1:
                chunk = self.per * self.nxy
                mine = self.resp[self.rank * chunk:(self.rank + 1) * chunk]
                if _staged_through_host(self.resp, self.group):  # (two ranks on one GPU over gloo: the tests)
                    whole, probs = torch.empty(self.resp.shape, dtype=torch.float64), self.probs.cpu()
                    dist.all_gather_into_tensor(whole, mine.cpu(), group=self.group)
                    dist.all_reduce(probs, op=dist.ReduceOp.MAX, group=self.group)
                    self.resp.copy_(whole)
                    self.probs.copy_(probs)
                else:
                    dist.all_gather_into_tensor(self.resp, mine, group=self.group)       # in place: slice r lands at r * chunk
                    dist.all_reduce(self.probs, op=dist.ReduceOp.MAX, group=self.group)
        return self.matcher.slice_finish()
