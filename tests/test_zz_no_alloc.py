"""The allocation-free entry point bf_eval_structs_noalloc (include/batchfirst.h) against
bf_eval_structs, through the C ABI, bit for bit: two calls in a row, the second with a smaller
input (the static scratch keeps the first call's mailboxes), in both precisions, plus the size
limit.  Collected last on purpose (tests/test_zz_*): the round's end runs pytest -x."""
import ctypes

import numpy as np
import pytest

from batch_first_amd import _lib, weights as W

pytestmark = pytest.mark.gpu


def _eval(lib, fn, ctx, recs, torch):
    out = torch.full((recs.shape[0],), float("nan"), dtype=torch.float32, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(fn(ctx, recs.data_ptr(), recs.shape[1], recs.shape[0], out.data_ptr(), stream))
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_noalloc_entry_point_matches_eval_structs(golden, dtype):
    import torch
    assert torch.cuda.is_available()
    lib = _lib.load()
    blob = W.to_blob(W.make_weights(seed=1234, bn="perturbed"))
    ctx = ctypes.c_void_p()
    _lib.check(lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, _lib.precision_code(dtype)))
    try:
        first = torch.from_numpy(np.tile(golden["records"], (3, 1))[:2500].copy()).cuda()      # 2500 boards
        second = torch.from_numpy(np.ascontiguousarray(golden["records"][::-1][:777])).cuda()   # 777 other boards
        for recs in (first, second):      # two calls in a row, the second smaller
            want = _eval(lib, lib.bf_eval_structs, ctx, recs, torch)
            got = _eval(lib, lib.bf_eval_structs_noalloc, ctx, recs, torch)
            assert np.isfinite(want).all()
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), dtype
        # and again in the other order of sizes, without a reference call in between
        a = _eval(lib, lib.bf_eval_structs_noalloc, ctx, second, torch)
        b = _eval(lib, lib.bf_eval_structs_noalloc, ctx, first, torch)
        c = _eval(lib, lib.bf_eval_structs_noalloc, ctx, second[:1], torch)
        assert np.array_equal(a.view(np.uint32), _eval(lib, lib.bf_eval_structs, ctx, second, torch).view(np.uint32))
        assert np.array_equal(b.view(np.uint32), _eval(lib, lib.bf_eval_structs, ctx, first, torch).view(np.uint32))
        assert c[0] == a[0]
        # empty input is fine, a batch beyond the static scratch is refused before anything is enqueued
        out = torch.zeros(1, device="cuda")
        assert lib.bf_eval_structs_noalloc(ctx, first.data_ptr(), first.shape[1], 0, out.data_ptr(), None) == 0
        rc = lib.bf_eval_structs_noalloc(ctx, first.data_ptr(), first.shape[1], _lib.BF_NOALLOC_MAX_BOARDS + 1,
                                         out.data_ptr(), None)
        assert rc == _lib.BF_E_ARG and b"static scratch" in lib.bf_last_error()
    finally:
        lib.bf_ctx_destroy(ctx)


def test_noalloc_at_its_maximum_size(golden):
    """BF_NOALLOC_MAX_BOARDS boards in one call: copies of a position score identically wherever
    they sit in the batch (size-independent property), equal to bf_eval_structs."""
    import torch
    lib = _lib.load()
    blob = W.to_blob(W.make_weights(seed=1234, bn="perturbed"))
    ctx = ctypes.c_void_p()
    _lib.check(lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, _lib.BF_PRECISION_BF16))
    try:
        reps = _lib.BF_NOALLOC_MAX_BOARDS // 1024
        recs = torch.from_numpy(np.tile(golden["records"], (reps, 1))).cuda()
        got = _eval(lib, lib.bf_eval_structs_noalloc, ctx, recs, torch).reshape(reps, 1024)
        want = _eval(lib, lib.bf_eval_structs, ctx, recs[:1024], torch)
        assert np.all(got.view(np.uint32) == want.view(np.uint32)[None, :])
    finally:
        lib.bf_ctx_destroy(ctx)
