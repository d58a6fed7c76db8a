"""World-size-2 (and 3, ragged) runs of the shard + gather path on the gloo backend."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from batch_first_amd import distributed as D, positions as P
from oracle import encoding as E


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _material(structs):
    # the reference's fake evaluator (code_testing.py:523-532) stands in for the GPU
    return E.bf_piece_sum_eval(*E.struct_array_to_ann_inputs(structs)).astype(np.float32)


def _worker(rank, world, port, n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        structs = P.test_set(n, seed=3)
        full = D.evaluate_sharded(_material, structs, dst=0)
        if rank == 0:
            np.save(out_path, full.numpy())
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 64), (3, 50), (2, 1)])
def test_sharded_evaluation_equals_single_rank(tmp_path, world, n):
    out = str(tmp_path / "scores.npy")
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    got = np.load(out)
    exp = _material(P.test_set(n, seed=3))
    assert np.array_equal(got, exp)


def _wire_worker(rank, world, port, n, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nib, occ = E.struct_array_to_ann_inputs(P.test_set(n, seed=5))   # every rank holds the full feed arrays
        pf_shard, occ_shard = D.shard_wire(nib, occ, rank, world)
        local = torch.from_numpy(E.bf_piece_sum_eval(pf_shard, occ_shard).astype(np.float32))
        full = D.gather_scores(local, n, dst=0)
        if rank == 0:
            np.save(out_path, full.numpy())
    finally:
        dist.destroy_process_group()


def _odd_split_sizes(world):
    """Batch sizes whose shards start on an odd nibble for some rank > 0."""
    sizes = []
    for n in range(world, 80):
        _, occ = E.struct_array_to_ann_inputs(P.test_set(n, seed=5))
        counts = np.array([bin(int(m)).count("1") for m in occ])
        starts = [int(counts[:D.shard_range(n, r, world)[0]].sum()) for r in range(1, world)]
        if any(s % 2 for s in starts):
            sizes.append(n)
    return sizes


@pytest.mark.parametrize("world", [2, 3])
def test_wire_format_shards_including_an_odd_nibble_split(tmp_path, world):
    """The real feed arrays (nibble stream + masks) sharded over the ranks: a shard that begins on an
    odd nibble must decode to the same boards (the reference's wire-format consumer, code_testing.py:
    505-532, scores them)."""
    n = _odd_split_sizes(world)[0]
    out = str(tmp_path / "scores.npy")
    mp.spawn(_wire_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    nib, occ = E.struct_array_to_ann_inputs(P.test_set(n, seed=5))
    assert np.array_equal(np.load(out), E.bf_piece_sum_eval(nib, occ).astype(np.float32))


def test_shard_wire_round_trip_and_native_split():
    """Every rank's shard decodes to exactly its boards' planes, for even and odd starts, and the native
    splitter of the one-process multi-GPU call (bf_host_split_wire) cuts at the same places."""
    import ctypes
    from batch_first_amd import _lib
    lib = _lib.load()
    structs = P.test_set(97, seed=9)
    nib, occ = E.struct_array_to_ann_inputs(structs)
    planes = E.decode_planes_fast(nib, occ)
    counts = np.array([bin(int(m)).count("1") for m in occ])
    odd_starts = 0
    for world in (1, 2, 3, 5, 8):
        board_lo = np.zeros(world + 1, np.uint64)
        nibble_lo = np.zeros(world + 1, np.uint64)
        assert lib.bf_host_split_wire(occ.ctypes.data, occ.size, world, board_lo.ctypes.data, nibble_lo.ctypes.data) == 0
        assert int(board_lo[-1]) == occ.size and int(nibble_lo[-1]) == int(counts.sum())
        for r in range(world):
            lo, hi = D.shard_range(occ.size, r, world)
            assert (int(board_lo[r]), int(board_lo[r + 1])) == (lo, hi)
            assert int(nibble_lo[r]) == int(counts[:lo].sum())
            odd_starts += int(nibble_lo[r]) % 2
            pf_s, occ_s = D.shard_wire(nib, occ, r, world)
            assert pf_s.size == (int(counts[lo:hi].sum()) + 1) // 2
            assert np.array_equal(E.decode_planes_fast(pf_s, occ_s), planes[lo:hi])
    assert odd_starts > 3


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 65536, 2 ** 24 + 5):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
