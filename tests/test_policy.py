"""The move-scoring network: move encoding tables, oracle consistency (CPU) and GPU parity.

Scores are PARITY UNPINNED by the reference (no weights, no golden scores; TensorFlow absent): the
checker is oracle.network.policy_logits_torch, a restatement of
batch_first/anns/move_evaluation_ann.py:10-73.  Bar on the GPU: |g - r| <= 1e-4 max(|r|, rms(r)).
"""
import os
import threading

import numpy as np
import pytest

from batch_first_amd import _lib, moves as M, positions as P, weights as W
from oracle import encoding as E, network as N

HERE = os.path.dirname(os.path.abspath(__file__))
REL_TOL = 1e-4


@pytest.fixture(scope="module")
def pgolden():
    with np.load(os.path.join(HERE, "golden", "policy_256.npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="module")
def pweights():
    return W.make_policy_weights(seed=4321, bn="perturbed")


def test_move_filter_table_known_answers():
    t = M.MOVE_FILTER_LOOKUP
    assert t.shape == (64, 64, 6) and t.dtype == np.uint8
    # queen-move planes: direction blocks of 7 distances (batch_first/__init__.py:60-69)
    assert t[12, 28, 0] == 1            # e2e4: (file 0, rank +2) -> j-1
    assert t[0, 7, 0] == 13 + 7         # a1h1: (file +7, rank 0) -> j+13
    assert t[63, 0, 0] == 48 + 7        # h8a1: (-7, -7) -> j+48
    assert t[6, 21, 0] == 58            # g1f3: knight (-1, +2)
    assert t[57, 42, 0] == 57           # b8c6: knight (+1, -2)
    assert t[62, 45, 0] == 59           # g8f6: knight (-1, -2)
    assert t[1, 18, 0] == 56            # b1c3: (+1, +2)
    # promotions: queen (5) keeps the queen-move plane, N/B/R (2,3,4) get 64 + 3*(1+file_diff) + {0,1,2}
    assert list(t[52, 60, 2:5]) == [67, 68, 69] and t[52, 60, 5] == 0
    assert list(t[52, 59, 2:5]) == [64, 65, 66]
    assert list(t[52, 61, 2:5]) == [70, 71, 72]
    assert t.max() == 72
    # a black move is mirrored into the mover's frame first (numba_negamax_zero_window.py:512-515)
    frm, flt, per = M.move_info_for([0], [1], [[52, 36, 0]])      # e7e5 as black
    assert frm[0] == 12 and flt[0] == 1 and per[0] == 1
    assert np.array_equal(M.SQUARES_180[M.SQUARES_180], np.arange(64))


def test_policy_oracle_float32_vs_float64_and_golden(pgolden, pweights, golden):
    assert W.blob_sha256(W.to_policy_blob(pweights)).encode() == pgolden["weights_sha256"].tobytes()
    planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[:64]
    l32, x9 = N.policy_logits_torch(planes, pweights)
    l64, _ = N.policy_logits_torch(planes, pweights, double=True)
    assert l32.shape == (64, 73, 64) and x9.shape == (64, 32, 8, 8)
    assert np.max(np.abs(l32 - l64)) <= 1e-5 * np.abs(l64).max()
    per = pgolden["moves_per_board"][:64]
    m = int(per.sum())
    got = N.policy_gather(l32, pgolden["from_squares"][:m], pgolden["move_filters"][:m], per)
    assert np.array_equal(got.astype(np.float32), pgolden["scores_f32"][:m])
    # a single-piece closed form: the logit of (filter f, square s) is W_logit[:, f] . x9[:, s]
    b, f, s = 3, 17, 42
    assert abs(l64[b, f, s] - float(pweights["convs"][17][0, 0, :, f].astype(np.float64) @ x9[b].reshape(32, 64)[:, s])) < 1e-4


def test_policy_pack_rejects_wrong_sizes():
    lib = _lib.load()
    blob = W.to_policy_blob(W.make_policy_weights())
    out = np.zeros(lib.bf_packed_floats(), np.float32)
    assert lib.bf_pack_policy_weights(blob.ctypes.data, blob.size, out.ctypes.data, out.size) == 0
    assert lib.bf_pack_policy_weights(blob.ctypes.data, blob.size - 1, out.ctypes.data, out.size) == _lib.BF_E_ARG
    # the tail of the image holds the logit kernel as [filter][channel]
    head = out[-2 * 64 * 64 * 4:][:73 * 32].reshape(73, 32)
    assert np.array_equal(head, W.make_policy_weights()["convs"][17][0, 0].T)


# ---- GPU ------------------------------------------------------------------------------------------

def _ok(got, ref):
    rms = float(np.sqrt(np.mean(np.asarray(ref, np.float64) ** 2)))
    e = np.abs(got - ref) / np.maximum(np.abs(ref), rms)
    return float(e.max())


@pytest.fixture(scope="module")
def both(pweights):
    from batch_first_amd import chestimator
    bp, start, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"),
                                                            policy_weights=pweights)
    yield bp, start
    closer()


@pytest.mark.gpu
def test_policy_callable_matches_golden(both, pgolden, golden):
    bp, start = both
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)[:256]
    nib, occ = E.struct_array_to_ann_inputs(structs)
    finish = start(nib, occ)
    got = finish([pgolden["from_squares"], pgolden["move_filters"], pgolden["moves_per_board"]])
    assert got.dtype == np.float32 and got.shape == pgolden["scores_f32"].shape
    w32, w64 = _ok(got, pgolden["scores_f32"]), _ok(got, pgolden["scores_f64"])
    print("policy worst error / rms-floored magnitude: %.2e vs f32 oracle, %.2e vs f64 oracle" % (w32, w64))
    assert w32 <= REL_TOL and w64 <= REL_TOL
    # same again: deterministic, and the inputs as wider integer types
    finish2 = start(nib.astype(np.int32), occ.view(np.int64))
    got2 = finish2([pgolden["from_squares"].astype(np.int64), pgolden["move_filters"].astype(np.int32),
                    pgolden["moves_per_board"].astype(np.int64)])
    assert np.array_equal(got.view(np.uint32), got2.view(np.uint32))


@pytest.mark.gpu
def test_policy_edge_cases_and_errors(both, pgolden, golden):
    bp, start = both
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    nib1, occ1 = E.struct_array_to_ann_inputs(structs[:1])
    # one board, the root-node case (numba_negamax_zero_window.py:737-755)
    one = start(nib1, occ1)([np.array([12, 6], np.uint8), np.array([1, 58], np.uint8), np.array([2], np.uint8)])
    full = start(*E.struct_array_to_ann_inputs(structs[:3]))(
        [np.array([12, 6, 0], np.uint8), np.array([1, 58, 5], np.uint8), np.array([2, 0, 1], np.uint8)])
    assert np.array_equal(one, full[:2])
    # no moves at all
    assert start(nib1, occ1)([np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(1, np.uint8)]).shape == (0,)
    with pytest.raises(ValueError):      # moves_per_board does not add up
        start(nib1, occ1)([np.array([1], np.uint8), np.array([1], np.uint8), np.array([2], np.uint8)])
    with pytest.raises(ValueError):      # filter out of range
        start(nib1, occ1)([np.array([1], np.uint8), np.array([73], np.uint8), np.array([1], np.uint8)])
    with pytest.raises(ValueError):      # board count differs from the started batch
        start(nib1, occ1)([np.array([1], np.uint8), np.array([1], np.uint8), np.array([1, 0], np.uint8)])
    with pytest.raises(ValueError):      # malformed wire input
        start(nib1[:-1], occ1)
    f = start(nib1, occ1)
    f([np.array([1], np.uint8), np.array([1], np.uint8), np.array([1], np.uint8)])
    with pytest.raises(ValueError):      # a closure is good for one call (partial_run handles are, too)
        f([np.array([1], np.uint8), np.array([1], np.uint8), np.array([1], np.uint8)])


@pytest.mark.gpu
def test_policy_overlaps_value_call(both, pgolden, golden):
    """The search runs both networks from two threads at once (numba_negamax_zero_window.py:178,199)."""
    bp, start = both
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = E.struct_array_to_ann_inputs(structs[:256])
    value_ref = bp(golden["piece_filters"], golden["occupied_bbs"])
    out = {}

    def value_worker():
        for _ in range(10):
            out["v"] = bp(golden["piece_filters"], golden["occupied_bbs"])

    t = threading.Thread(target=value_worker)
    t.start()
    for _ in range(10):
        out["p"] = start(nib, occ)([pgolden["from_squares"], pgolden["move_filters"], pgolden["moves_per_board"]])
    t.join()
    assert np.array_equal(out["v"], value_ref)
    assert _ok(out["p"], pgolden["scores_f32"]) <= REL_TOL


@pytest.mark.gpu
def test_policy_device_ops(pweights, pgolden, golden):
    import torch
    from batch_first_amd import ops
    ctx = ops.DeviceContext(W.make_weights(seed=1234, bn="perturbed"), device=0)
    try:
        with pytest.raises(ValueError):
            torch.ops.batchfirst.policy_features_structs(ctx.handle, torch.from_numpy(golden["records"][:4].copy()).cuda())
        ctx.set_policy(pweights)
        recs = torch.from_numpy(golden["records"][:256].copy()).cuda()
        x9 = torch.ops.batchfirst.policy_features_structs(ctx.handle, recs)
        planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[:256]
        _, x9_ref = N.policy_logits_torch(planes, pweights)
        assert np.max(np.abs(x9.cpu().numpy() - x9_ref)) < 2e-5
        nib, occ = E.struct_array_to_ann_inputs(np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)[:256])
        x9w = torch.ops.batchfirst.policy_features_wire(ctx.handle, torch.from_numpy(nib).cuda(),
                                                        torch.from_numpy(occ.view(np.int64)).cuda())
        assert torch.equal(x9, x9w)
        dev = lambda a: torch.from_numpy(a.copy()).cuda()
        s = torch.ops.batchfirst.policy_scores(ctx.handle, x9, dev(pgolden["from_squares"]), dev(pgolden["move_filters"]),
                                               dev(pgolden["moves_per_board"]))
        ctx.check_policy()
        assert _ok(s.cpu().numpy(), pgolden["scores_f32"]) <= REL_TOL
    finally:
        ctx.close()


@pytest.mark.gpu
def test_policy_bf16_precision(pweights, pgolden, golden):
    """BF16 context: the move-scoring network runs on the bf16 kernel (no reference counterpart;
    bar as for the value network's bf16 precision, tests/test_gpu_bf16.py): scores within
    4e-2 * max(|r|, rms) of the float32 oracle (measured worst 2.4e-2 on the sibling groups below, 1.6e-2 here),
    and the moves of a board keep their order wherever float32 separates them by more than 4x that tolerance.
    Parity unpinned: the oracle is this repository's restatement of the graph."""
    from batch_first_amd import chestimator
    bp, start, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"), dtype="bf16",
                                                            policy_weights=pweights)
    try:
        structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)[:256]
        nib, occ = E.struct_array_to_ann_inputs(structs)
        got = start(nib, occ)([pgolden["from_squares"], pgolden["move_filters"], pgolden["moves_per_board"]])
        again = start(nib, occ)([pgolden["from_squares"], pgolden["move_filters"], pgolden["moves_per_board"]])
    finally:
        closer()
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    ref = pgolden["scores_f32"]
    worst = _ok(got, ref)
    print("bf16 policy: worst error / rms-floored magnitude %.2e" % worst)
    assert worst <= 4e-2
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    gap = 4 * 4e-2 * rms
    pos, checked = 0, 0
    for k in pgolden["moves_per_board"].astype(np.int64):   # int64: a uint8 running sum wraps at 256
        k = int(k)
        r, g = ref[pos:pos + k].astype(np.float64), got[pos:pos + k].astype(np.float64)
        assert len(r) == k
        sep = (r[:, None] - r[None, :]) > gap
        assert np.all((g[:, None] - g[None, :])[sep] > 0)
        checked += int(sep.sum())
        pos += k
    assert pos == len(ref)
    assert checked > 1000


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "f32s", "f32s3", "bf16"])
def test_policy_orders_the_legal_moves_of_real_parents(pweights, dtype):
    """The move-scoring network on what the search feeds it: the 128 parents of
    tests/golden/siblings_128.npz with ALL their legal moves as requests in the mover's frame
    (numba_negamax_zero_window.py:485-527).  float32 and f32s (BF_PRECISION_F32_SPLIT): scores within REL_TOL of the
    oracle.  bf16 (no
    reference counterpart): within 4e-2 * max(|r|, rms), and the moves of a parent keep their order and
    their best move wherever float32 separates them by more than 4x that tolerance (x rms)."""
    import os
    from test_gpu_bf16 import ordering_report
    from batch_first_amd import chestimator
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "siblings_128.npz")) as z:
        sib = {k: z[k] for k in z.files}
    bp, start, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"), dtype=dtype,
                                                            policy_weights=pweights)
    try:
        structs = np.frombuffer(sib["parent_records"].tobytes(), dtype=P.COMPACT_DTYPE)
        nib, occ = E.struct_array_to_ann_inputs(structs)
        got = start(nib, occ)([sib["from_squares"], sib["move_filters"], sib["group_sizes"]])
    finally:
        closer()
    ref = sib["policy_scores_f32"]
    worst = _ok(got, ref)
    rms = float(np.sqrt(np.mean(ref.astype(np.float64) ** 2)))
    tol = 4e-2 if dtype == "bf16" else REL_TOL
    print("%s policy on real move lists: worst error / rms-floored magnitude %.2e" % (dtype, worst))
    assert worst <= tol
    rep = ordering_report(got, ref, sib["group_sizes"], 4 * tol * rms)
    print("policy ordering: %d best moves and %d pairs kept above the gap; %d of %d pairs differ below it"
          % (rep["top_checked"], rep["pairs_checked"], rep["flipped_below_gap"], rep["pairs"]))
    assert rep["groups"] == 128 and rep["pairs_checked"] > (1000 if dtype == "bf16" else 40000)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_policy_over_several_devices_equals_one_device(pweights, pgolden, golden, dtype):
    """MultiDevicePredictor.start_move_prediction: stage 1 of the move-scoring network on every device's shard
    of the boards, each device's moves to its own second stage (bf_multi_policy_*).  On a one-GPU box three
    contexts on device 0 stand behind the handle, so the board split (odd-nibble shard starts included), the
    split of the move lists at board boundaries and the gather run either way.  Scores are bit-equal to one
    device's for every batch size; the native search takes the handle through the same function pointers."""
    import torch
    from batch_first_amd import chestimator, search
    w = W.to_blob(W.make_weights(seed=1234, bn="perturbed"))
    single, start1, close1 = chestimator.get_inference_functions(w, dtype=dtype, policy_weights=pweights)
    count = torch.cuda.device_count()
    devices = list(range(count)) if count > 1 else [0, 0, 0]
    multi = chestimator.MultiDevicePredictor(w, devices, dtype=dtype, min_shard=32, allow_repeats=count == 1)
    try:
        assert not multi.has_policy
        multi.set_policy(pweights)
        assert multi.has_policy
        structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)[:256]
        per = pgolden["moves_per_board"].astype(np.int64)
        ends = np.concatenate([[0], np.cumsum(per)])
        for n in (256, 255, 100, 97, 33, 1):
            nib, occ = E.struct_array_to_ann_inputs(structs[:n])
            info = [pgolden["from_squares"][:ends[n]], pgolden["move_filters"][:ends[n]], pgolden["moves_per_board"][:n]]
            want = start1(nib, occ)(info)
            got = multi.start_move_prediction(nib, occ)(info)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), n
        # no moves for some shard, and errors
        nib, occ = E.struct_array_to_ann_inputs(structs[:96])
        per0 = pgolden["moves_per_board"][:96].copy()
        per0[32:64] = 0
        keep = np.concatenate([np.arange(ends[0], ends[32]), np.arange(ends[64], ends[96])])
        info0 = [pgolden["from_squares"][keep], pgolden["move_filters"][keep], per0]
        assert np.array_equal(multi.start_move_prediction(nib, occ)(info0), start1(nib, occ)(info0))
        with pytest.raises(ValueError):
            multi.start_move_prediction(nib[:-1], occ)
        f = multi.start_move_prediction(nib, occ)
        with pytest.raises(ValueError):
            f([info0[0], info0[1], per0[:-1]])
        with pytest.raises(ValueError):   # the failed call abandoned the prediction
            f(info0)
        # the search: same value, move and counters with the move-scoring network on one device or on all
        fen = "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"
        multi_small = chestimator.MultiDevicePredictor(w, devices, dtype=dtype, min_shard=8, allow_repeats=count == 1)
        try:
            multi_small.set_policy(pweights)
            results = []
            for pred in (single, multi_small):
                s = search.Search.on_gpu(pred, use_policy=True, max_batch_size=512, threads=2)
                value, move = s.mtd_f(fen, 3)
                st = s.stats()
                results.append((value, tuple(move), st["iterations"], st["nodes_out"], st["boards_evaluated"], st["moves_scored"]))
                s.close()
            assert results[0] == results[1]
        finally:
            multi_small.close()
    finally:
        multi.close()
        close1()


@pytest.mark.gpu
def test_measured_min_shard_is_a_crossover():
    """measure_min_shard: t(2s) > t(s) + the second shard's enqueue at the returned s, and not at the sizes below."""
    from batch_first_amd import chestimator
    w = W.to_blob(W.make_weights(seed=1234, bn="perturbed"))
    bp = chestimator.BoardPredictor(w, dtype="bf16")
    try:
        s, t = chestimator.measure_min_shard(bp, sizes=(128, 256, 512, 1024, 2048), reps=5)
    finally:
        bp.close()
    print("bf16 one-device call times (us):", {k: round(v, 1) for k, v in sorted(t.items())}, "-> min_shard", s)
    assert s in (128, 256, 512, 1024, 2048)
    assert s == 2048 or t[2 * s] > t[s] + 10.0
