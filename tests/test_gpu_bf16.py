"""GPU tests of the BF16 precision (BASELINE configs[2]) through the C ABI.

The reference evaluates in FP32 only, so there is no reference result for this arithmetic
("parity unpinned"): the checker is oracle.network.forward_bf16_model, a CPU restatement of what
include/batchfirst.h documents for BF_PRECISION_BF16 (folded kernels and activations x1..x9 rounded
to bf16, float32 sums).  Bars, stated here as north_star asks:

* layer by layer, each layer of the model applied to the GPU's own previous layer: x1 and x10
  agree to float32 rounding; of x3..x9 fewer than 0.1 % of the elements differ (the two sides sum
  in a different float32 order, which now and then flips a bf16 rounding) and none by more than
  one bf16 step of the layer's largest value;
* end to end against that model: |g - m| <= 3e-2 * max(|m|, rms(m)) -- a flipped rounding grows
  through the six 3x3 layers, so the model is no tighter a witness than the FP32 oracle here
  (measured: worst 1.7e-2, mean 1e-3);
* against the FP32 oracle (golden scores): |g - r| <= FP32_TOL * max(|r|, rms(r)), FP32_TOL = 4e-2
  (measured worst: 2.9e-2 on the 1024 golden positions, 3.3e-2 on the 3734 sibling positions, mean
  6e-3; the CPU model itself sits at the same distance; round 1 allowed 5e-2);
* move ordering on REAL sibling groups (tests/golden/siblings_128.npz: 128 parents from random legal
  playouts, every legal move of each, 20-40 children a group -- what create_child_structs,
  numba_negamax_zero_window.py:315-347, hands the evaluator): the arg-max is the FP32 arg-max
  whenever FP32 separates its top two by more than ORDER_GAP = 4 x FP32_TOL x rms,
  and every pair that FP32 separates by more than that keeps its order.  Because both scores of a
  pair move by at most the tolerance, the same must already hold at HALF that gap, which is asserted
  too.  Below the gap bf16 does reorder siblings (with random-init weights a group spreads over only
  0.18 rms; the CPU model flips 1.2 % of all sibling pairs and 4 of the 128 arg-maxima): the test
  prints those counts, north_star's "identical move ordering" holds above the stated gap only.
"""
import numpy as np
import pytest

from batch_first_amd import positions as P, weights as W
from oracle import encoding as E, network as N

pytestmark = pytest.mark.gpu

MODEL_TOL = 3e-2
FP32_TOL = 4e-2
ORDER_GAP = 4 * FP32_TOL      # in units of the fp32 scores' rms


def _rel(got, ref):
    rms = float(np.sqrt(np.mean(np.asarray(ref, np.float64) ** 2)))
    return np.abs(got - ref) / np.maximum(np.abs(ref), rms), rms


@pytest.fixture(scope="module")
def weights():
    return W.make_weights(seed=1234, bn="perturbed")


@pytest.fixture(scope="module")
def ctx16(weights):
    import torch
    assert torch.cuda.is_available()
    from batch_first_amd import ops
    c = ops.DeviceContext(weights, device=0, dtype="bf16")
    yield c
    c.close()


@pytest.fixture(scope="module")
def predictor16(weights):
    from batch_first_amd import chestimator
    bp, _, closer = chestimator.get_inference_functions(weights, dtype="bf16")
    assert bp.dtype == "bf16"
    yield bp
    closer()


@pytest.fixture(scope="module")
def model_scores(golden, weights):
    planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])
    return N.forward_bf16_model(planes, weights)


def test_bf16_scores_match_model_and_fp32_oracle(predictor16, golden, model_scores):
    got = predictor16(golden["piece_filters"], golden["occupied_bbs"])
    assert got.dtype == np.float32 and got.shape == (1024,) and np.isfinite(got).all()
    e_model, _ = _rel(got, model_scores)
    e_fp32, rms = _rel(got, golden["scores_f32_perturbed"])
    print("bf16 kernel: worst vs bf16 model %.2e, worst vs fp32 oracle %.2e (rms score %.3f)"
          % (e_model.max(), e_fp32.max(), rms))
    assert e_model.max() <= MODEL_TOL
    assert e_fp32.max() <= FP32_TOL


def test_bf16_layers_follow_model(ctx16, golden, weights):
    import torch
    from batch_first_amd import ops
    n = 129   # odd: the last workgroup carries a padding board
    mbc = golden["mailbox"][:n]
    mb = (mbc[:, 0::2] | (mbc[:, 1::2] << 4)).astype(np.uint8)
    scores, dbg = ops.eval_mailbox_debug(ctx16.handle, torch.from_numpy(mb).cuda())
    dbg = dbg.cpu().numpy()
    planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[:n]
    ref, inter = N.forward_bf16_model(planes, weights, return_intermediates=True)
    pos, prev = 0, None
    for name, ch in [("x1", 236), ("x3", 128), ("x4", 32), ("x5", 32), ("x6", 32), ("x7", 32), ("x8", 32), ("x9", 32)]:
        got = dbg[:, pos:pos + ch * 64].reshape(n, ch, 8, 8)
        pos += ch * 64
        if prev is None:
            # stage 1 sums at most 9 taps of exact one-hot products: no rounding flips
            assert np.max(np.abs(got - inter["x1"])) <= 2.0 ** -9 * np.abs(inter["x1"]).max()
            assert np.mean(got != inter["x1"]) < 1e-4
        else:
            want = N.bf16_model_layer(prev, weights, name)
            diff = np.abs(got - want)
            assert np.mean(diff > 0) < 1e-3, (name, float(np.mean(diff > 0)))
            assert diff.max() <= 2.0 ** -7 * max(1.0, float(np.abs(want).max())), (name, float(diff.max()))
        prev = got
    want10 = N.bf16_model_layer(prev, weights, "x10").reshape(n, 16)
    assert np.max(np.abs(dbg[:, pos:pos + 16] - want10)) < 1e-5
    e, _ = _rel(scores.cpu().numpy(), ref)
    assert e.max() <= MODEL_TOL


def test_bf16_entry_points_agree_and_are_deterministic(ctx16, predictor16, golden):
    import torch
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    recs = torch.from_numpy(golden["records"].copy()).cuda()
    pf = torch.from_numpy(golden["piece_filters"].copy()).cuda()
    occ = torch.from_numpy(golden["occupied_bbs"].view(np.int64).copy()).cuda()
    a = torch.ops.batchfirst.eval_wire(ctx16.handle, pf, occ).cpu().numpy()
    b = torch.ops.batchfirst.eval_structs(ctx16.handle, recs).cpu().numpy()
    c = predictor16(golden["piece_filters"], golden["occupied_bbs"])
    d = predictor16.eval_structs(structs)
    a2 = torch.ops.batchfirst.eval_wire(ctx16.handle, pf, occ).cpu().numpy()
    for other in (b, c, d, a2):
        assert np.array_equal(a.view(np.uint32), other.view(np.uint32))
    # a board's score does not depend on its neighbours or its place in the batch: N = 1, 2, 3, 5
    for lo, hi in [(0, 1), (7, 9), (100, 103), (1019, 1024)]:
        part = predictor16.eval_structs(structs[lo:hi])
        assert np.array_equal(part.view(np.uint32), a[lo:hi].view(np.uint32)), (lo, hi)
    assert a[3] == a[4]   # mirrored reference FENs (code_testing.py:36-37)


def test_records_decoded_inside_the_kernel_equal_the_two_kernel_form(weights, golden, monkeypatch):
    """A bf16 context hands 4-byte-aligned records straight to the evaluation kernel (one launch a call); 806-byte
    reference records (unaligned) and a context created under BF_T16_TWO_KERNELS=1 go through k_to_mailbox first.
    All three give the same bits, for whole batches, odd sizes and through the host entry point."""
    import torch
    from batch_first_amd import ops
    monkeypatch.setenv("BF_T16_TWO_KERNELS", "1")
    two = ops.DeviceContext(weights, device=0, dtype="bf16")
    monkeypatch.delenv("BF_T16_TWO_KERNELS")
    one = ops.DeviceContext(weights, device=0, dtype="bf16")
    try:
        structs806 = P.test_set(1024, seed=0)
        assert structs806.dtype.itemsize == 806
        compact = P.to_compact(structs806)
        r88 = ops.records_to_tensor(compact, torch.device("cuda", 0))
        r806 = ops.records_to_tensor(structs806, torch.device("cuda", 0))
        assert r88.shape[1] == 88 and r806.shape[1] == 806
        want = torch.ops.batchfirst.eval_structs(two.handle, r88).cpu().numpy()
        for n in (1024, 1023, 5, 1):
            a = torch.ops.batchfirst.eval_structs(one.handle, r88[:n].contiguous()).cpu().numpy()
            b = torch.ops.batchfirst.eval_structs(one.handle, r806[:n].contiguous()).cpu().numpy()
            assert np.array_equal(a.view(np.uint32), want[:n].view(np.uint32)), n
            assert np.array_equal(b.view(np.uint32), want[:n].view(np.uint32)), n
        # a view that starts 2 bytes into a record row is not 4-byte aligned: falls back, same scores
        shifted = torch.zeros(1024 * 88 + 2, dtype=torch.uint8, device="cuda")
        shifted[2:] = r88.reshape(-1)
        c = torch.ops.batchfirst.eval_structs(one.handle, shifted[2:].view(1024, 88)).cpu().numpy()
        assert np.array_equal(c.view(np.uint32), want.view(np.uint32))
    finally:
        one.close()
        two.close()


def test_small_calls_on_one_board_per_workgroup_equal_the_dense_kernel(weights, golden):
    """Calls of up to bf_set_split_boards() boards run k_eval_t16s (one board per workgroup, a column tile per
    wave); larger ones k_eval_t16 (one board per wave).  Same bits from both, for scores and for the move-scoring
    network's features, from 4-bit mailboxes (wire format, 806-byte records) and from aligned records; sizes
    around the number of compute units, odd sizes, one board."""
    import torch
    from batch_first_amd import _lib, encoder, ops
    lib = _lib.load()
    ctx = ops.DeviceContext(weights, device=0, dtype="bf16")
    ctx.set_policy(W.make_policy_weights())
    before = lib.bf_set_split_boards(0)
    try:
        assert before == _lib.BF_SPLIT_BOARDS_DEFAULT
        structs806 = P.test_set(1024, seed=0)
        r88 = ops.records_to_tensor(P.to_compact(structs806), torch.device("cuda", 0))
        r806 = ops.records_to_tensor(structs806, torch.device("cuda", 0))
        sizes = (1, 2, 3, 146, 255, 256, 257, 511, 777, 1024)

        def run(n):
            idx = (np.arange(n) * 5 + 1) % 1024
            a88 = r88[torch.from_numpy(idx).cuda()].contiguous()
            a806 = r806[torch.from_numpy(idx).cuda()].contiguous()
            sel = structs806[idx]
            pf, occ = encoder.struct_array_to_ann_inputs(sel, sel[:0], np.ones(n, bool), np.zeros(0, bool), n)
            out = [torch.ops.batchfirst.eval_structs(ctx.handle, a88).cpu().numpy(),
                   torch.ops.batchfirst.eval_structs(ctx.handle, a806).cpu().numpy(),
                   torch.ops.batchfirst.policy_features_structs(ctx.handle, a88).cpu().numpy(),
                   torch.ops.batchfirst.policy_features_structs(ctx.handle, a806).cpu().numpy()]
            out.append(torch.ops.batchfirst.eval_wire(ctx.handle, torch.from_numpy(pf).cuda(),
                                                      torch.from_numpy(occ.view(np.int64)).cuda()).cpu().numpy())
            return out

        dense = {n: run(n) for n in sizes}
        assert lib.bf_set_split_boards(1 << 20) == 0
        for n in sizes:
            for k, (want, got) in enumerate(zip(dense[n], run(n))):
                assert got.shape == want.shape
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (n, k)
        # the two forms of one call agree with each other too (records decoded in the kernel / through mailboxes)
        assert np.array_equal(dense[257][0].view(np.uint32), dense[257][1].view(np.uint32))
        # the host entry points at the default switch-over: 146 boards split, 1024 dense.  The reference's callable
        # (wire format in, NumPy out) and its move-scoring closure decode the wire format inside the small-call
        # kernel (one launch): same bits as the mailbox route, malformed input still refused.
        lib.bf_set_split_boards(_lib.BF_SPLIT_BOARDS_DEFAULT)
        compact = P.to_compact(structs806)
        from batch_first_amd import chestimator
        pred, start_moves, closer = chestimator.get_inference_functions(weights, dtype="bf16",
                                                                        policy_weights=W.make_policy_weights())
        try:
            small = pred.eval_structs(compact[:146])
            whole = pred.eval_structs(compact)
            assert np.array_equal(small.view(np.uint32), whole[:146].view(np.uint32))

            def wire_runs(n):
                idx = (np.arange(n) * 5 + 1) % 1024
                sel = structs806[idx]
                pf, occ = encoder.struct_array_to_ann_inputs(sel, sel[:0], np.ones(n, bool), np.zeros(0, bool), n)
                per = np.full(n, 3, np.uint8)
                frm = np.tile(np.array([0, 9, 63], np.uint8), n)
                flt = np.tile(np.array([0, 17, 72], np.uint8), n)
                return pf, occ, pred(pf, occ), start_moves(pf, occ)([frm, flt, per])

            for n in (1, 2, 33, 146, 256, 257, 511):
                lib.bf_set_split_boards(0)
                _, _, want_s, want_m = wire_runs(n)
                lib.bf_set_split_boards(_lib.BF_SPLIT_BOARDS_DEFAULT)
                pf, occ, got_s, got_m = wire_runs(n)
                assert np.array_equal(got_s.view(np.uint32), want_s.view(np.uint32)), n
                assert np.array_equal(got_m.view(np.uint32), want_m.view(np.uint32)), n
                assert np.array_equal(got_s.view(np.uint32), dense[n][0].view(np.uint32)) if n in dense else True
            bad = pf.copy()
            bad[len(bad) // 2] = 0xFF                          # code 15 is not a piece code
            with pytest.raises(ValueError):
                pred(bad, occ)
            with pytest.raises(ValueError):
                pred(pf[:-1], occ)                             # stream shorter than the masks say
            with pytest.raises(ValueError):
                start_moves(bad, occ)([np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros(len(occ), np.uint8)])
            again = pred(pf, occ)                              # the flag does not stick to the next call
            assert np.array_equal(again.view(np.uint32), got_s.view(np.uint32))
        finally:
            closer()
    finally:
        lib.bf_set_split_boards(before)
        ctx.close()


@pytest.fixture(scope="module")
def siblings():
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "siblings_128.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def ordering_report(got, ref, group_sizes, gap):
    """Per sibling group: arg-max and pair order of `got` against `ref` wherever `ref` separates by more
    than `gap`.  Returns counts; raises on a violation."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    pos = 0
    out = dict(groups=0, top_checked=0, pairs_checked=0, pairs=0, flipped_below_gap=0, argmax_moved=0)
    for k in group_sizes:
        k = int(k)
        r, s = ref[pos:pos + k], got[pos:pos + k]
        pos += k
        order = np.argsort(-r)
        if r[order[0]] - r[order[1]] > gap:
            assert int(np.argmax(s)) == int(order[0]), ("arg-max moved", out["groups"])
            out["top_checked"] += 1
        out["argmax_moved"] += int(np.argmax(s) != order[0])
        dr, ds = r[:, None] - r[None, :], s[:, None] - s[None, :]
        sep = dr > gap
        assert np.all(ds[sep] > 0), ("pair reordered above the gap", out["groups"])
        out["pairs_checked"] += int(sep.sum())
        iu = np.triu_indices(k, 1)
        out["pairs"] += len(iu[0])
        out["flipped_below_gap"] += int(np.sum(np.sign(dr[iu]) != np.sign(ds[iu])))
        out["groups"] += 1
    assert pos == len(ref)
    return out


def test_bf16_move_ordering_on_real_sibling_groups(predictor16, siblings):
    structs = np.frombuffer(siblings["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    got = predictor16.eval_structs(structs)
    again = predictor16(siblings["piece_filters"], siblings["occupied_bbs"])
    assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    ref = siblings["scores_f32"]
    e, rms = _rel(got, ref)
    print("siblings: worst vs fp32 oracle %.2e, mean %.2e (rms score %.3f)" % (e.max(), e.mean(), rms))
    assert e.max() <= FP32_TOL
    sizes = siblings["group_sizes"]
    assert len(sizes) == 128 and sizes.min() >= 20 and sizes.max() <= 40
    rep = ordering_report(got, ref, sizes, ORDER_GAP * rms)
    half = ordering_report(got, ref, sizes, 0.5 * ORDER_GAP * rms)
    print("ordering above %.3f rms: %d arg-maxima, %d pairs kept; above half the gap: %d arg-maxima, %d pairs kept; "
          "below: %d of %d pairs and %d of 128 arg-maxima differ from fp32"
          % (ORDER_GAP, rep["top_checked"], rep["pairs_checked"], half["top_checked"], half["pairs_checked"],
             rep["flipped_below_gap"], rep["pairs"], rep["argmax_moved"]))
    assert rep["groups"] == 128 and rep["pairs_checked"] > 10000 and half["pairs_checked"] > rep["pairs_checked"]
    assert rep["flipped_below_gap"] < 0.03 * rep["pairs"]


@pytest.mark.parametrize("dtype", ["f32s", "f32s3"])
def test_f32_split_move_ordering_on_real_sibling_groups(weights, siblings, dtype):
    """BF_PRECISION_F32_SPLIT on the same sibling groups (the ordering the bf16
    precision loses, bought back): scores inside the float32 bound, every pair and every arg-max that float32
    separates by more than four times that bound kept, and over ALL pairs the line asked of a repaired bf16 kernel:
    at least 126 of 128 arg-maxima and under 0.5 % flipped pairs (measured: see the printed line)."""
    from batch_first_amd import chestimator
    bp, _, closer = chestimator.get_inference_functions(weights, dtype=dtype)
    try:
        assert bp.dtype == dtype
        structs = np.frombuffer(siblings["records"].tobytes(), dtype=P.COMPACT_DTYPE)
        got = bp.eval_structs(structs)
        again = bp(siblings["piece_filters"], siblings["occupied_bbs"])
        assert np.array_equal(got.view(np.uint32), again.view(np.uint32))
    finally:
        closer()
    ref = siblings["scores_f32"]
    e, rms = _rel(got, ref)
    sizes = siblings["group_sizes"]
    rep = ordering_report(got, ref, sizes, 4e-4 * rms)
    print(dtype + " siblings: worst vs fp32 oracle %.2e (bound 1e-4); above 4e-4 rms: %d arg-maxima, %d pairs kept; "
          "over all pairs %d of %d flipped, %d of 128 arg-maxima moved"
          % (e.max(), rep["top_checked"], rep["pairs_checked"], rep["flipped_below_gap"], rep["pairs"], rep["argmax_moved"]))
    assert e.max() <= 1e-4
    assert rep["groups"] == 128 and rep["pairs_checked"] > 50000
    assert rep["argmax_moved"] <= 2 and rep["flipped_below_gap"] < 0.005 * rep["pairs"]


def test_bf16_batch_65536_properties(ctx16, golden, model_scores):
    import torch
    recs = torch.from_numpy(np.tile(golden["records"], (64, 1))).cuda()
    s = torch.ops.batchfirst.eval_structs(ctx16.handle, recs)
    torch.cuda.synchronize()
    s = s.cpu().numpy().reshape(64, 1024)
    assert np.isfinite(s).all()
    assert np.all(s.view(np.uint32) == s.view(np.uint32)[0:1])
    e, _ = _rel(s[63], model_scores)
    assert e.max() <= MODEL_TOL


def test_bf16_ragged_batches_on_persistent_workgroups(ctx16, golden):
    """The bf16 kernel runs one persistent workgroup per compute unit over the board quads: batches that
    give the workgroups unequal numbers of quads, a partial last quad, or fewer quads than units must
    score every board exactly as the 1024-board batch does (size-independent property; the oracle
    comparison of those 1024 scores is test_bf16_scores_match_model_and_fp32_oracle)."""
    import torch
    base = torch.ops.batchfirst.eval_structs(ctx16.handle, torch.from_numpy(golden["records"].copy()).cuda()).cpu().numpy()
    for n in (1, 3, 5, 1023, 1025, 2051, 4099, 9001):
        idx = (np.arange(n) * 7 + 3) % 1024
        recs = torch.from_numpy(golden["records"][idx].copy()).cuda()
        got = torch.ops.batchfirst.eval_structs(ctx16.handle, recs).cpu().numpy()
        assert got.shape == (n,)
        assert np.array_equal(got.view(np.uint32), base[idx].view(np.uint32)), n
