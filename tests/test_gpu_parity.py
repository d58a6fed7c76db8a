"""GPU parity tests: the HIP path (through the C ABI) against the oracle and the golden fixtures.

Bars: encoding bit-exact; fp32 scores |g - r| <= 1e-4 * max(|r|, rms(r)) (north_star: 1e-4
relative; rms floor as in SURVEY.md 8(d) so near-zero scores are not held to an impossible bar).
"""
import threading

import numpy as np
import pytest

from batch_first_amd import positions as P, weights as W
from oracle import encoding as E, network as N

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4


def _score_ok(got, ref):
    rms = float(np.sqrt(np.mean(np.asarray(ref, np.float64) ** 2)))
    bound = REL_TOL * np.maximum(np.abs(ref), rms)
    return np.all(np.abs(got - ref) <= bound), float(np.max(np.abs(got - ref) / bound))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


@pytest.fixture(scope="module")
def ctx(torch_mod):
    from batch_first_amd import ops
    c = ops.DeviceContext(W.make_weights(seed=1234, bn="perturbed"), device=0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def predictor():
    from batch_first_amd import chestimator
    bp, move_fn, closer = chestimator.get_inference_functions(W.to_blob(W.make_weights(seed=1234, bn="perturbed")))
    with pytest.raises(NotImplementedError):
        move_fn()
    yield bp
    closer()


def _dev_wire(torch, golden, lo=0, hi=None):
    occ = golden["occupied_bbs"][lo:hi]
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)[lo:hi]
    nib, occ2 = E.struct_array_to_ann_inputs(structs)
    assert np.array_equal(occ, occ2)
    return (torch.from_numpy(nib.copy()).cuda(), torch.from_numpy(occ.view(np.int64).copy()).cuda())


# ---- encoding: bit-exact -----------------------------------------------------------------

def test_planes_from_wire_bit_exact(torch_mod, ctx, golden):
    torch = torch_mod
    pf, occ = _dev_wire(torch, golden)
    planes = torch.ops.batchfirst.encode_planes_wire(ctx.handle, pf, occ)
    ctx.check()
    ref = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])
    assert planes.dtype == torch.float32 and tuple(planes.shape) == (1024, 15, 8, 8)
    assert np.array_equal(planes.cpu().numpy().view(np.uint32), ref.view(np.uint32))


@pytest.mark.parametrize("full_stride", [False, True])
def test_planes_and_mailbox_from_records_bit_exact(torch_mod, ctx, golden, full_stride):
    torch = torch_mod
    from batch_first_amd import ops
    if full_stride:
        structs = P.test_set(1024, seed=0)  # 806-byte reference records: unaligned fields
        assert structs.dtype.itemsize == 806
        recs = ops.records_to_tensor(structs, "cuda")
    else:
        recs = torch.from_numpy(golden["records"].copy()).cuda()
    planes = torch.ops.batchfirst.encode_planes_structs(ctx.handle, recs)
    ref = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])
    assert np.array_equal(planes.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    mb = torch.ops.batchfirst.encode_mailbox_structs(ctx.handle, recs).cpu().numpy()
    codes = np.stack([mb & 15, mb >> 4], axis=-1).reshape(len(mb), 64)
    assert np.array_equal(codes, golden["mailbox"])


def test_mailbox_from_wire_and_uneven_splits(torch_mod, ctx, golden):
    torch = torch_mod
    # code_testing.py:242-250: the same boards fed in uneven pieces, including a single board
    edges = [0, 1, 2, 3, 10, 11, 64, 65, 257, 1023, 1024]
    out = []
    for a, b in zip(edges[:-1], edges[1:]):
        pf, occ = _dev_wire(torch, golden, a, b)
        mb = torch.ops.batchfirst.encode_mailbox_wire(ctx.handle, pf, occ)
        out.append(mb.cpu().numpy())
    ctx.check()
    mb = np.concatenate(out)
    codes = np.stack([mb & 15, mb >> 4], axis=-1).reshape(len(mb), 64)
    assert np.array_equal(codes, golden["mailbox"])


def test_wire_more_boards_than_one_scan_block(torch_mod, ctx, golden):
    torch = torch_mod
    structs = P.random_structs(5000, seed=11)  # the reference's test batch size (code_testing.py:763)
    nib, occ = E.struct_array_to_ann_inputs(structs)
    planes = torch.ops.batchfirst.encode_planes_wire(
        ctx.handle, torch.from_numpy(nib).cuda(), torch.from_numpy(occ.view(np.int64)).cuda())
    ctx.check()
    assert np.array_equal(planes.cpu().numpy(), E.decode_planes_fast(nib, occ))


def test_empty_inputs(torch_mod, ctx, predictor):
    torch = torch_mod
    pf = torch.zeros(0, dtype=torch.uint8, device="cuda")
    occ = torch.zeros(0, dtype=torch.int64, device="cuda")
    assert torch.ops.batchfirst.eval_wire(ctx.handle, pf, occ).numel() == 0
    assert predictor(np.zeros(0, np.uint8), np.zeros(0, np.uint64)).shape == (0,)
    # a board with no relevant squares at all is legal input for the decoder
    s = predictor(np.zeros(0, np.uint8), np.zeros(3, np.uint64))
    assert s.shape == (3,) and np.all(s == s[0]) and np.isfinite(s[0])


def test_malformed_wire_is_reported(torch_mod, ctx, predictor, golden):
    torch = torch_mod
    nib, occ = golden["piece_filters"], golden["occupied_bbs"]
    with pytest.raises(ValueError):
        predictor(nib[:-3], occ)                      # stream too short for the masks
    with pytest.raises(ValueError):
        predictor(np.r_[nib, np.zeros(5, np.uint8)], occ)
    bad = nib.copy()
    bad[10] = 0xFF                                    # code 15 is not a piece code
    with pytest.raises(ValueError):
        predictor(bad, occ)
    # device entry point: sticky flag, no out-of-bounds read
    pf = torch.from_numpy(nib[:100].copy()).cuda()
    oc = torch.from_numpy(occ.view(np.int64).copy()).cuda()
    torch.ops.batchfirst.encode_mailbox_wire(ctx.handle, pf, oc)
    with pytest.raises(ValueError):
        ctx.check()
    ctx.check()  # flag is cleared by the report
    # the evaluator still works afterwards
    assert np.isfinite(predictor(nib, occ)).all()


# ---- scores ----------------------------------------------------------------------------------

@pytest.mark.parametrize("bn", ["perturbed", "fresh"])
def test_scores_match_golden(bn, golden):
    from batch_first_amd import chestimator
    bp, _, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn=bn))
    try:
        got = bp(golden["piece_filters"], golden["occupied_bbs"])
    finally:
        closer()
    assert got.dtype == np.float32 and got.shape == (1024,)
    ok, worst = _score_ok(got, golden["scores_f32_" + bn])
    ok64, worst64 = _score_ok(got, golden["scores_f64_" + bn])
    print("worst error / bound vs f32 oracle %.3f, vs f64 oracle %.3f" % (worst, worst64))
    assert ok and ok64


def test_intermediates_match_oracle(torch_mod, ctx, golden):
    """Layer-by-layer check through the debug entry point (locates a fault, should one appear)."""
    torch = torch_mod
    from batch_first_amd import ops
    n = 32
    mbc = golden["mailbox"][:n]
    mb = (mbc[:, 0::2] | (mbc[:, 1::2] << 4)).astype(np.uint8)
    scores, dbg = ops.eval_mailbox_debug(ctx.handle, torch.from_numpy(mb).cuda())
    dbg = dbg.cpu().numpy()
    planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[:n]
    _, inter = N.forward_numpy64(planes, W.make_weights(seed=1234, bn="perturbed"), True)
    pos = 0
    for name, ch in [("x1", 236), ("x3", 128), ("x4", 32), ("x5", 32), ("x6", 32), ("x7", 32), ("x8", 32), ("x9", 32)]:
        got = dbg[:, pos:pos + ch * 64].reshape(n, ch, 8, 8)
        err = np.max(np.abs(got - inter[name]))
        assert err < 2e-5, (name, err)
        pos += ch * 64
    assert np.max(np.abs(dbg[:, pos:pos + 16] - inter["x10"].reshape(n, 16))) < 2e-5
    ok, _ = _score_ok(scores.cpu().numpy(), golden["scores_f64_perturbed"][:n])
    assert ok


def test_all_entry_points_agree_bitwise(torch_mod, ctx, predictor, golden):
    torch = torch_mod
    pf, occ = _dev_wire(torch, golden)
    recs = torch.from_numpy(golden["records"].copy()).cuda()
    a = torch.ops.batchfirst.eval_wire(ctx.handle, pf, occ).cpu().numpy()
    b = torch.ops.batchfirst.eval_structs(ctx.handle, recs).cpu().numpy()
    mb = torch.ops.batchfirst.encode_mailbox_structs(ctx.handle, recs)
    c = torch.ops.batchfirst.eval_mailbox(ctx.handle, mb).cpu().numpy()
    d = predictor(golden["piece_filters"], golden["occupied_bbs"])
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    e = predictor.eval_structs(structs)
    for other in (b, c, d, e):
        assert np.array_equal(a.view(np.uint32), other.view(np.uint32))
    # run-to-run determinism (scores are compared against float32 separators by the search,
    # numba_negamax_zero_window.py:414)
    assert np.array_equal(a.view(np.uint32), torch.ops.batchfirst.eval_wire(ctx.handle, pf, occ).cpu().numpy().view(np.uint32))


def test_callable_contract(predictor, golden):
    nib, occ = golden["piece_filters"], golden["occupied_bbs"]
    n1 = E.popcount(occ[0])
    one = predictor(nib[:(n1 + 1) // 2].copy(), occ[:1])        # N = 1
    assert one.shape == (1,)
    full = predictor(nib, occ)
    assert one[0] == full[0]
    # inputs as the TF placeholders declare them: int32 bytes, int64 bit patterns
    alt = predictor(nib.astype(np.int32), occ.view(np.int64))
    assert np.array_equal(alt, full)
    # a strided view, as compress_square_array returns (numba_negamax_zero_window.py:19)
    strided = np.repeat(nib, 2)[::2]
    assert not strided.flags["C_CONTIGUOUS"] or strided.base is not None
    assert np.array_equal(predictor(strided, occ), full)
    # engine.py:260 subtracts a float32 zero shift from the result
    shifted = full - np.float32(0.25)
    assert shifted.dtype == np.float32
    # mirrored reference FENs evaluate identically (code_testing.py:36-37)
    assert full[3] == full[4]


def test_start_board_evaluations_and_masks(predictor, golden):
    from batch_first_amd import encoder
    structs = P.test_set(1024, seed=0)
    mask = np.arange(1024) % 3 != 0
    t, scores = encoder.start_board_evaluations(structs, mask, predictor)
    t.join()
    full = predictor(golden["piece_filters"], golden["occupied_bbs"])
    assert np.array_equal(scores, full[mask])
    # through the wire format and a plain callable wrapper (what engine.py:260 builds)
    t2, scores2 = encoder.start_board_evaluations(structs, mask, lambda *a: predictor(*a) - np.float32(0))
    t2.join()
    assert np.array_equal(scores2, scores)


def test_concurrent_callers(predictor, golden):
    """The search calls the evaluator from worker threads (numba_negamax_zero_window.py:178,199)."""
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    full = predictor(golden["piece_filters"], golden["occupied_bbs"])
    results, errors = {}, []

    def work(k, lo, hi):
        try:
            for _ in range(5):
                results[k] = predictor(*E.struct_array_to_ann_inputs(structs[lo:hi]))
        except Exception as exc:  # pragma: no cover
            errors.append(exc)

    spans = [(0, 1024), (5, 6), (100, 613), (900, 1024)]
    threads = [threading.Thread(target=work, args=(k, lo, hi)) for k, (lo, hi) in enumerate(spans)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for k, (lo, hi) in enumerate(spans):
        assert np.array_equal(results[k], full[lo:hi])


# ---- full size (BASELINE config 2): properties that do not need the oracle ------------------------

def test_batch_65536_properties(torch_mod, ctx, golden):
    torch = torch_mod
    base = golden["records"]                                   # 1024 records
    recs = torch.from_numpy(np.tile(base, (64, 1))).cuda()    # 65536 records
    scores = torch.ops.batchfirst.eval_structs(ctx.handle, recs)
    torch.cuda.synchronize()
    s = scores.cpu().numpy().reshape(64, 1024)
    assert np.isfinite(s).all()
    # every copy of a position gets bit-identical scores wherever it sits in the batch
    assert np.all(s.view(np.uint32) == s.view(np.uint32)[0:1])
    ok, _ = _score_ok(s[63], golden["scores_f32_perturbed"])
    assert ok
    # planes at full size: exactly one plane per relevant square, none elsewhere
    planes = torch.ops.batchfirst.encode_planes_structs(ctx.handle, recs)
    per_square = planes.sum(dim=1).reshape(65536, 64)
    occ = torch.from_numpy(np.tile(golden["occupied_bbs"], 64).view(np.int64).copy()).cuda()
    bits = ((occ[:, None] >> torch.arange(64, device="cuda")[None, :]) & 1).to(torch.float32)
    assert torch.equal(per_square, bits)
    assert float(planes.sum()) == float(bits.sum())


# ---- BASELINE configs[0]: 1024 random legal positions ------------------------------------------------

def test_scores_on_random_legal_positions(predictor):
    """Positions from random legal playouts (tests/golden/make_golden_legal.py): records straight to
    the GPU, wire format through the callable, both against the oracle's scores."""
    import os
    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "legal_1024.npz")) as z:
        g = {k: z[k] for k in z.files}
    structs = np.frombuffer(g["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    a = predictor.eval_structs(structs)
    b = predictor(g["piece_filters"], g["occupied_bbs"])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ok32, w32 = _score_ok(a, g["scores_f32"])
    ok64, w64 = _score_ok(a, g["scores_f64"])
    print("legal positions: worst error / bound %.3f vs f32 oracle, %.3f vs f64 oracle" % (w32, w64))
    assert ok32 and ok64


def test_two_streams_on_one_context_do_not_share_scratch_in_flight(torch_mod, ctx, golden):
    """Device entry points only enqueue, and calls on one context may come on different streams:
    the context's mailbox / offset scratch is handed from one call to the next by an event
    (api.hip, Lane::scratch_done).  A long call on stream A followed at once by a short call on
    stream B, which rewrites the first mailboxes: both must give the scores of a serial run."""
    torch = torch_mod
    big = torch.from_numpy(np.tile(golden["records"], (64, 1))).cuda()                   # 65 536 boards, ~16 ms
    small = torch.from_numpy(np.ascontiguousarray(golden["records"][::-1][:512])).cuda()  # other boards, same slots
    pf, occ = _dev_wire(torch, golden)
    want_big = torch.ops.batchfirst.eval_structs(ctx.handle, big).cpu().numpy()
    want_small = torch.ops.batchfirst.eval_structs(ctx.handle, small).cpu().numpy()
    want_wire = torch.ops.batchfirst.eval_wire(ctx.handle, pf, occ).cpu().numpy()
    torch.cuda.synchronize()
    sa, sb, sc = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(sa):
            got_big = torch.ops.batchfirst.eval_structs(ctx.handle, big)
        with torch.cuda.stream(sb):
            got_small = torch.ops.batchfirst.eval_structs(ctx.handle, small)
        with torch.cuda.stream(sc):
            got_wire = torch.ops.batchfirst.eval_wire(ctx.handle, pf, occ)
        torch.cuda.synchronize()
        assert np.array_equal(got_big.cpu().numpy().view(np.uint32), want_big.view(np.uint32))
        assert np.array_equal(got_small.cpu().numpy().view(np.uint32), want_small.view(np.uint32))
        assert np.array_equal(got_wire.cpu().numpy().view(np.uint32), want_wire.view(np.uint32))
    ctx.check()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_multi_device_callable_equals_single_device(torch_mod, golden, dtype):
    """get_inference_functions(..., devices=[...]): one callable over every visible GPU (one context and
    one stream per device, wire format split at popcount boundaries, odd-nibble shard starts handled by
    the decode kernel).  On a one-GPU box the same code path runs with two contexts on device 0, so the
    splitting, the start-nibble decode and the gather are exercised either way; scores must be bit-equal
    to the single-device callable's for every split."""
    from batch_first_amd import chestimator, search
    w = W.to_blob(W.make_weights(seed=1234, bn="perturbed"))
    single, _, close_single = chestimator.get_inference_functions(w, dtype=dtype)
    count = torch_mod.cuda.device_count()
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    counts = np.array([bin(int(m)).count("1") for m in golden["occupied_bbs"]])
    try:
        want = single(golden["piece_filters"], golden["occupied_bbs"])
        if count > 1:
            multi, _, close_multi = chestimator.get_inference_functions(w, dtype=dtype, devices=list(range(count)), min_shard=64)
        else:
            # three contexts (streams) on the one device behind the same handle
            multi = chestimator.MultiDevicePredictor(w, [0, 0, 0], dtype=dtype, min_shard=64, allow_repeats=True)
            close_multi = multi.close
        try:
            odd = 0
            for n in (1024, 1023, 777, 500, 193, 130, 64, 5, 1):
                nib, occ = E.struct_array_to_ann_inputs(structs[:n])
                got = multi(nib, occ)
                assert np.array_equal(got.view(np.uint32), want[:n].view(np.uint32)), n
                got2 = multi.eval_structs(structs[:n])
                assert np.array_equal(got2.view(np.uint32), want[:n].view(np.uint32)), n
                parts_n = min(len(multi.parts), max(1, n // 64))
                odd += sum(int(counts[:D_shard(n, r, parts_n)].sum()) % 2 for r in range(1, parts_n))
            assert odd > 0, "no shard in this test started on an odd nibble"
            mask = np.arange(1024) % 3 != 0
            assert np.array_equal(multi.eval_structs(structs, mask).view(np.uint32), want[mask].view(np.uint32))
            with pytest.raises(ValueError):
                multi(golden["piece_filters"][:-1], golden["occupied_bbs"])
            # the native search takes the multi-device evaluator through the same function-pointer signature
            s = search.Search.on_gpu(multi, use_policy=False, max_batch_size=2048)
            s1 = search.Search.on_gpu(single, use_policy=False, max_batch_size=2048)
            fen = "r3k2r/p1ppqpb1/bn2pnp1/3PN3/1p2P3/2N2Q1p/PPPBBPPP/R3K2R w KQkq - 0 1"
            assert s.mtd_f(fen, 2)[0] == s1.mtd_f(fen, 2)[0]
            s.close()
            s1.close()
        finally:
            close_multi()
    finally:
        close_single()


def D_shard(n, r, parts):
    from batch_first_amd.distributed import shard_range
    return shard_range(n, r, parts)[0]


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_evaluate_sharded_with_the_real_evaluator_world_size_one(torch_mod, golden, dtype):
    """batch_first_amd.distributed.evaluate_sharded (what bench.py's ranks do: this rank's contiguous shard through
    the HIP evaluator, scores gathered to rank 0) in a world of one process on the GPU box: the shard is everything,
    the gather is the identity, the scores are the single-context ones bit for bit.  (World sizes 2 and 3, ragged
    shards and odd-nibble wire shards run on gloo with a host evaluator in tests/test_distributed_cpu.py.)"""
    import socket
    import torch.distributed as dist
    from batch_first_amd import distributed as D, ops
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    ctx = ops.DeviceContext(W.make_weights(seed=1234, bn="perturbed"), device=0, dtype=dtype)
    try:
        recs = torch_mod.from_numpy(golden["records"].copy()).cuda()
        want = torch_mod.ops.batchfirst.eval_structs(ctx.handle, recs).cpu()
        got = D.evaluate_sharded(lambda r: torch_mod.ops.batchfirst.eval_structs(ctx.handle, r).cpu(), recs, dst=0)
        assert got.shape == (1024,) and torch_mod.equal(got, want)
        ragged = D.evaluate_sharded(lambda r: torch_mod.ops.batchfirst.eval_structs(ctx.handle, r).cpu(), recs[:777], dst=0)
        assert torch_mod.equal(ragged, want[:777])
    finally:
        ctx.close()
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype,bar,x3_bar", [("f32s", 0.8, 3e-4), ("f32s3", 0.15, 2e-5)])
def test_f32_split_precision_scores_within_the_fp32_bound(torch_mod, golden, dtype, bar, x3_bar):
    """BF_PRECISION_F32_SPLIT ("f32s") and BF_PRECISION_F32_SPLIT3 ("f32s3"): the fp32 kernel's structure with the 1x1
    pair and the 3x3 layers as sums of bf16 x bf16 products on the bf16 MFMA (no reference counterpart; precisions of
    their own beside F32 and BF16).  f32s: two terms a side in the pair, three in the 3x3 layers; bar: inside the fp32
    bar, |g - r| <= 1e-4 max(|r|, rms) against the float32 and float64 oracles on both golden weight sets (measured
    worst 0.18 of it with the perturbed batch-norm statistics, 0.62 with the fresh ones; F32 measures 0.03 and 0.08).
    f32s3: three terms everywhere; as close to the oracle as F32 itself: measured worst 0.04 / 0.11 of the bar (0.10
    against the float64 oracle; VERDICT round 2, item 3 asked for 0.1), bar 0.15.  Records, wire format and mailboxes give the same bits; the result does not depend on the batch
    or a board's place in it; the activations in front of and behind the pair agree with the oracle's."""
    from batch_first_amd import chestimator, ops
    structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    for bn in ("perturbed", "fresh"):
        w = W.make_weights(seed=1234, bn=bn)
        bp, _, closer = chestimator.get_inference_functions(W.to_blob(w), dtype=dtype)
        try:
            assert bp.dtype == dtype
            got = bp(golden["piece_filters"], golden["occupied_bbs"])
            got2 = bp.eval_structs(structs)
            part = bp.eval_structs(structs[100:357])
        finally:
            closer()
        assert np.array_equal(got.view(np.uint32), got2.view(np.uint32))
        assert np.array_equal(part.view(np.uint32), got[100:357].view(np.uint32))
        for ref in (golden["scores_f32_" + bn], golden["scores_f64_" + bn].astype(np.float32)):
            ok, worst = _score_ok(got, ref)
            print("%s, batch-norm %s: worst error / bound %.3f" % (dtype, bn, worst))
            assert ok and worst <= bar
    # layer by layer against the oracle's activations (debug build of the same kernel)
    w = W.make_weights(seed=1234, bn="perturbed")
    ctx = ops.DeviceContext(w, device=0, dtype=dtype)
    try:
        n = 32
        mbc = golden["mailbox"][:n]
        mb = (mbc[:, 0::2] | (mbc[:, 1::2] << 4)).astype(np.uint8)
        scores, dbg = ops.eval_mailbox_debug(ctx.handle, torch_mod.from_numpy(mb).cuda())
        dbg = dbg.cpu().numpy()
        planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[:n]
        _, inter = N.forward_numpy64(planes, w, True)
        x1 = dbg[:, :236 * 64].reshape(n, 236, 8, 8)
        x3 = dbg[:, 236 * 64:(236 + 128) * 64].reshape(n, 128, 8, 8)
        assert np.max(np.abs(x1 - inter["x1"])) < 2e-5                 # stage 1 is F32's
        e3 = float(np.max(np.abs(x3 - inter["x3"])))
        print("%s: x3 worst absolute error %.2e (largest |x3| %.2f)" % (dtype, e3, float(np.abs(inter["x3"]).max())))
        assert e3 < x3_bar                                             # two layers of 16-bit (f32s) / 24-bit operands
    finally:
        ctx.close()


@pytest.mark.parametrize("dtype", ["f32", "f32s", "f32s3"])
def test_fp32_small_calls_run_another_kernel_with_the_same_bits(torch_mod, golden, dtype):
    """A float32 context runs calls of up to 256 boards on k_eval_f32q (one board per workgroup, eight waves, 16 x 16
    tiles for the 3x3 layers: eval_f32q.hip) and larger ones on k_eval_f32.  Both fp32 MFMA shapes are fmaf chains in k
    order (tools/mfma_shape_bits.hip) and the small kernel keeps every layer's order of k, so a board's score and the
    move-scoring network's features must not depend on the size of the call it travels in: bit equality over call
    sizes around every switch, for mailbox, record and wire input, value network and move-scoring features
    (as test_gpu_bf16.py does for k_eval_t16s).  The split precisions (f32s, f32s3) take the same
    route with their bf16-term products: the same 32 x 32 x 16 MFMAs per output tile in the same order, the hidden layer
    exchanged as float32 and split by its readers."""
    from batch_first_amd import ops
    w = W.make_weights(seed=1234, bn="perturbed")
    ctx = ops.DeviceContext(w, device=0, dtype=dtype)
    try:
        ctx.set_policy(W.make_policy_weights(seed=4321))
        recs = torch_mod.from_numpy(golden["records"].copy()).cuda()
        mbc = golden["mailbox"]
        mb = torch_mod.from_numpy((mbc[:, 0::2] | (mbc[:, 1::2] << 4)).astype(np.uint8)).cuda()
        big = torch_mod.ops.batchfirst.eval_structs(ctx.handle, recs)              # 1 024 boards: k_eval_f32
        big_mb = torch_mod.ops.batchfirst.eval_mailbox(ctx.handle, mb)
        big_x9 = torch_mod.ops.batchfirst.policy_features_structs(ctx.handle, recs)
        assert torch_mod.equal(big, big_mb)
        ok, worst = _score_ok(big.cpu().numpy(), golden["scores_f32_perturbed"])
        assert ok
        for n, lo in ((1, 0), (1, 777), (2, 5), (63, 100), (64, 200), (200, 300), (255, 512), (256, 768), (257, 3), (300, 700)):
            part = torch_mod.ops.batchfirst.eval_structs(ctx.handle, recs[lo:lo + n].contiguous())
            assert torch_mod.equal(part, big[lo:lo + n]), ("records", n, lo)
            part = torch_mod.ops.batchfirst.eval_mailbox(ctx.handle, mb[lo:lo + n].contiguous())
            assert torch_mod.equal(part, big[lo:lo + n]), ("mailbox", n, lo)
            x9 = torch_mod.ops.batchfirst.policy_features_structs(ctx.handle, recs[lo:lo + n].contiguous())
            assert torch_mod.equal(x9, big_x9[lo:lo + n]), ("features", n, lo)
    finally:
        ctx.close()
    # the reference's callable (wire format, host arrays) at a small and a large size
    from batch_first_amd import chestimator
    bp, _, closer = chestimator.get_inference_functions(W.to_blob(w), dtype=dtype)
    try:
        structs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
        whole = bp(golden["piece_filters"], golden["occupied_bbs"])
        assert np.array_equal(whole.view(np.uint32), big.cpu().numpy().view(np.uint32))
        nib, occ = E.struct_array_to_ann_inputs(structs[40:190])
        small = bp(nib, occ)
        assert np.array_equal(small.view(np.uint32), whole[40:190].view(np.uint32))
        small = bp.eval_structs(structs[900:1000])
        assert np.array_equal(small.view(np.uint32), whole[900:1000].view(np.uint32))
    finally:
        closer()
