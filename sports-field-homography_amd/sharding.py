"""Frame sharding over the GPUs of one node.

``predict`` has no cross-frame operation (eval-mode BatchNorm uses running statistics), so a
batch shards by frame with NO data-path collective: rank r of W processes frames
[r*n, (r+1)*n).  The only exchange is the gather of the per-frame results that the caller
needs on every rank / on rank 0: theta (9 floats) and consist_score (1 float) = 40 B per
frame.  One process per GPU; backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU
tests.  The payload is latency-bound (5 KB for 128 frames), so a single all_gather per batch
is used - ``all_gather_into_tensor`` on ONE (world * n, 10) receive buffer: with the list form RCCL gathers into a
flat staging buffer and then copies every rank's rows out with a launch per rank; on the fully connected xGMI
mesh the collective itself resolves in one hop.

``force_collective`` (every exchange entry point): a world of ONE rank normally takes a local copy instead of the
collective; with the switch the real ``all_gather_into_tensor`` / ``all_reduce`` runs anyway.  That is how the N > 1 code
is put on RCCL on a one-GPU box (``tests/test_gpu_rccl.py``, ``bench.py --gpus 1 --force-collective``): same calls, same
side stream, same events as at N = 8 - only the peer count differs.
"""
import os

import torch
import torch.distributed as dist

FORCE_COLLECTIVE = os.environ.get("SFH_FORCE_COLLECTIVE", "") not in ("", "0")


def _use_collective(world, force):
    """the collective runs when there is a peer, or when a one-rank world was told to exercise it anyway"""
    if world > 1 and not dist.is_initialized():
        raise RuntimeError(f"{world} ranks but torch.distributed is not initialised: call dist.init_process_group "
                           "(backend 'nccl' = RCCL on GPUs) before the sharded entry points")
    return dist.is_initialized() and (world > 1 or force or FORCE_COLLECTIVE)


def _all_gather_flat(buf, rows, group=None):
    """one all_gather_into_tensor on the flat receive buffer; list form where a backend lacks the flat one"""
    try:
        dist.all_gather_into_tensor(buf, rows, group=group)
    except (RuntimeError, NotImplementedError) as e:
        if "allgather_base" not in str(e) and "all_gather_into_tensor" not in str(e) and "not support" not in str(e):
            raise
        n = rows.shape[0]
        dist.all_gather([buf[r * n:(r + 1) * n] for r in range(buf.shape[0] // max(n, 1))], rows, group=group)


def shard_range(n_frames, rank, world):
    """Contiguous, balanced split of n_frames; the first (n_frames % world) ranks get one more."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_results(theta, consist_score=None):
    """(n,1,3,3) [+ (n,)] -> (n,10) float32 rows [theta(9), score]."""
    n = theta.shape[0]
    out = torch.zeros((n, 10), dtype=torch.float32, device=theta.device)
    out[:, :9] = theta.reshape(n, 9)
    if consist_score is not None:
        out[:, 9] = consist_score
    return out


def gather_results(theta, consist_score=None, group=None, n_max=None, force_collective=False):
    """All ranks receive (theta_all (N,1,3,3), score_all (N,)) in frame order.

    Ragged shards (n differs across ranks) are padded to ``n_max`` rows; pass the per-rank
    frame counts via ``counts`` implicit in shard_range when known, otherwise they are
    exchanged first (one tiny all_gather of an int)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rows = pack_results(theta, consist_score)
    if not _use_collective(world, force_collective):
        return theta, rows[:, 9].clone()
    n = rows.shape[0]
    cnt = torch.tensor([n], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    if n < n_max:
        rows = torch.cat([rows, rows.new_zeros((n_max - n, 10))], 0)
    buf = rows.new_empty((world * n_max, 10))
    _all_gather_flat(buf, rows.contiguous(), group=group)
    if all(c == n_max for c in counts):
        allrows = buf
    else:
        allrows = torch.cat([buf[r * n_max:r * n_max + c] for r, c in enumerate(counts)], 0)
    return allrows[:, :9].reshape(-1, 1, 3, 3), allrows[:, 9].clone()


class ResultGather:
    """The exchange step of the sharded path, off the compute stream: every step's rows [theta(9), score] are
    all-gathered on a SIDE HIP stream while the next batch's kernels already run on the caller's stream (40 B per
    frame: the collective is pure latency, which this hides).  A ring of `depth` slots (rows + receive buffers);
    a slot is reused only after an event says its previous gather has finished.  On CPU tensors (gloo tests) the
    same calls run synchronously.

        g = ResultGather(world, frames_per_rank, device)
        slot = g.submit(out["theta"], out.get("consist_score"))     # returns at once
        theta_all, score_all = g.result(slot)                        # orders the caller's stream behind the gather
    """

    def __init__(self, world, n, device, depth=2, group=None, force_collective=False):
        device = torch.device(device)
        self.world, self.n, self.group = world, n, group
        self.collective = _use_collective(world, force_collective)
        self.collectives_run = 0      # all_gather_into_tensor calls issued (0 on the one-rank shortcut)
        self.cuda = device.type == "cuda"
        self.side = torch.cuda.Stream(device) if self.cuda else None
        # one flat receive buffer per slot: rank r's rows land at [r * n, (r + 1) * n) - a single collective, no copy-out
        self.slots = [{"rows": torch.zeros((n, 10), dtype=torch.float32, device=device),
                       "buf": torch.empty((world * n, 10), dtype=torch.float32, device=device),
                       "done": None} for _ in range(depth)]
        self.k = 0

    def submit(self, theta, consist_score=None):
        slot = self.slots[self.k % len(self.slots)]
        self.k += 1
        n = theta.shape[0]
        if n != self.n:
            raise ValueError(f"ResultGather was sized for {self.n} frames per rank, got {n}")
        if self.cuda and slot["done"] is not None:
            torch.cuda.current_stream().wait_event(slot["done"])   # the gather that used this slot last has read its rows
        slot["rows"][:, :9] = theta.reshape(n, 9)
        if consist_score is not None:
            slot["rows"][:, 9] = consist_score
        if not self.collective:       # one rank, nothing to exchange (world > 1 without a process group raised in __init__)
            slot["buf"].copy_(slot["rows"])
            return slot
        self.collectives_run += 1
        if not self.cuda:
            _all_gather_flat(slot["buf"], slot["rows"], group=self.group)
            return slot
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(self.side):
            self.side.wait_event(ready)
            _all_gather_flat(slot["buf"], slot["rows"], group=self.group)
            slot["done"] = torch.cuda.Event()
            slot["done"].record()
        return slot

    def result(self, slot):
        """(theta_all (world*n,1,3,3), score_all (world*n,)) in rank order"""
        if self.cuda and slot["done"] is not None:
            torch.cuda.current_stream().wait_event(slot["done"])
        rows = slot["buf"]
        return rows[:, :9].reshape(-1, 1, 3, 3).clone(), rows[:, 9].clone()    # (copies: the slot is reused two steps later)


def predict_sharded(net, frames, consistency=True, group=None, force_collective=False):
    """Run ``net.predict`` on this rank's shard of ``frames`` (a tensor holding the WHOLE batch
    or a callable rank_range -> shard tensor) and gather theta/consist_score from all ranks.
    Per-frame masks (logits, warp_mask) stay on the rank that computed them."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if callable(frames):
        x, (s, e) = frames(rank, world)
    else:
        s, e = shard_range(frames.shape[0], rank, world)
        x = frames[s:e]
    out = net.predict(x, consistency=consistency) if e > s else {}
    if e > s:
        theta = out["theta"]
        score = out.get("consist_score")
    else:  # empty shard (more ranks than frames)
        dev = frames.device if not callable(frames) else x.device
        theta = torch.zeros((0, 1, 3, 3), device=dev)
        score = torch.zeros((0,), device=dev) if consistency else None
    theta_all, score_all = gather_results(theta, score, group=group, force_collective=force_collective)
    out["theta_all"], out["consist_score_all"] = theta_all, score_all
    out["shard"] = (s, e)
    return out


# ------------------------------------------------------------------------ data-parallel training
def flat_views(shapes, device, dtype=torch.float32):
    """One flat buffer + one view per shape (gradients of all parameters back to back): the whole
    gradient exchange of a step is then a single collective over 4*sum(numel) bytes (209 MB for the
    default model) instead of 182 small ones - on xGMI's point-to-point links the ring all-reduce is
    per-link bandwidth bound, so one large message is the efficient shape."""
    sizes = [int(torch.Size(s).numel()) for s in shapes]
    flat = torch.zeros(sum(sizes), dtype=dtype, device=device)
    views, off = [], 0
    for s, n in zip(shapes, sizes):
        views.append(flat[off:off + n].view(s))
        off += n
    return flat, views


def world_size(group=None):
    """ranks of the initialised process group, 1 without one"""
    return dist.get_world_size(group) if dist.is_initialized() else 1


def allreduce_gradients(flat, group=None, force_collective=False):
    """Sum the flat gradient buffer over the ranks (RCCL all-reduce over xGMI on GPUs, gloo in the CPU
    tests); returns the factor 1/world that turns the sum into the data-parallel mean - the optimizer
    kernel applies it while reading the gradient (before clip_grad_value_, like DistributedDataParallel's
    averaged gradients)."""
    if not dist.is_initialized():
        return 1.0
    world = dist.get_world_size(group)
    if not _use_collective(world, force_collective):
        return 1.0
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return 1.0 / world
