"""Multi-GPU evaluation: one process per GPU, contiguous independent shards, scores gathered to
the search host's rank.

The reference has no multi-device code (every placement is '/GPU:0', chestimator.py:33); boards
are independent in the inference graph (batch-norm in inference mode,
ann_creation_helper.py:244), so the only exchange step is returning the scores.  With
``torch.distributed`` backend "nccl" that gather runs over RCCL/xGMI; the same code runs on
"gloo" for CPU tests.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n boards for this rank (sizes differ by at most one)."""
    base, extra = divmod(int(n), int(world))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_wire(piece_filters, occupied_bbs, rank, world):
    """This rank's contiguous shard of a wire-format batch as a stand-alone wire-format pair.

    The nibble stream is cut at sum-of-popcount boundaries; a shard that begins on
    an odd nibble is shifted by four bits so that its first code sits in a high nibble again, and a
    shard with an odd number of codes gets the trailing zero nibble own_concat would have given it
    (numba_negamax_zero_window.py:58,65-66).  (The one-process multi-GPU callable does not re-pack:
    its decode kernel takes the start nibble, bf_multi_eval_wire_host.)"""
    pf = np.ascontiguousarray(piece_filters, np.uint8)
    occ = np.ascontiguousarray(occupied_bbs).view(np.uint64)
    lo, hi = shard_range(occ.size, rank, world)
    counts = np.array([bin(int(m)).count("1") for m in occ], dtype=np.int64)
    s_lo, s_hi = int(counts[:lo].sum()), int(counts[:hi].sum())
    nibbles = np.empty(2 * pf.size, np.uint8)
    nibbles[0::2] = pf >> 4
    nibbles[1::2] = pf & 15
    mine = nibbles[s_lo:s_hi]
    if mine.size % 2:
        mine = np.r_[mine, np.uint8(0)]
    return ((mine[0::2] << 4) | mine[1::2]).astype(np.uint8), occ[lo:hi].copy()


def gather_scores(local_scores, n_total, dst=0, group=None):
    """Gather every rank's shard of scores to ``dst`` in shard order.

    Returns the full float32[n_total] tensor on ``dst`` and None elsewhere.  Shards may be ragged
    (n_total not divisible by the world size): they are padded to the largest shard for the
    collective and trimmed afterwards.
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [hi - lo for lo, hi in (shard_range(n_total, r, world) for r in range(world))]
    if local_scores.numel() != sizes[rank]:
        raise ValueError("rank %d holds %d scores, its shard has %d" % (rank, local_scores.numel(), sizes[rank]))
    width = max(sizes)
    send = local_scores
    if send.numel() != width:
        send = torch.zeros(width, dtype=local_scores.dtype, device=local_scores.device)
        send[:local_scores.numel()] = local_scores
    if rank == dst:
        parts = [torch.empty(width, dtype=send.dtype, device=send.device) for _ in range(world)]
        dist.gather(send, parts, dst=dst, group=group)
        return torch.cat([p[:s] for p, s in zip(parts, sizes)])
    dist.gather(send, None, dst=dst, group=group)
    return None


def evaluate_sharded(eval_local, records, dst=0, group=None):
    """Evaluate ``records`` (any indexable of boards, same on every rank) cooperatively.

    ``eval_local(shard) -> float32 tensor`` scores this rank's shard (on the GPU:
    ``lambda r: torch.ops.batchfirst.eval_structs(ctx.handle, r)``).
    """
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(len(records), rank, world)
    local = eval_local(records[lo:hi])
    if isinstance(local, np.ndarray):
        local = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float32))
    return gather_scores(local, len(records), dst=dst, group=group)
