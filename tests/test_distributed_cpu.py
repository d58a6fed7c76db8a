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


def test_shard_ranges_cover_everything():
    for n in (0, 1, 7, 64, 65536, 2 ** 24 + 5):
        for world in (1, 2, 3, 8):
            spans = [D.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
