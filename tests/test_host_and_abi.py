"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the native host
encoder equals the oracle, weight packing feeds the kernel model correctly."""
import ctypes
import os
import re

import numpy as np
import pytest

from batch_first_amd import _lib, encoder, positions as P, weights as W
from oracle import encoding as E, network as N
import kernel_sim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "batchfirst.h")).read()
    declared = set(re.findall(r"\b(bf_[a-z0-9_]+)\s*\(", header))
    declared.discard("bf_ctx")
    assert len(declared) >= 17
    for name in declared:
        assert hasattr(lib, name), name
        assert name in _lib.SIGNATURES, name
    assert lib.bf_version().startswith(b"batchfirst-hip")


def test_split_boards_knob_is_process_wide_and_returns_the_previous_value():
    lib = _lib.load()
    before = lib.bf_set_split_boards(7)
    try:
        assert lib.bf_set_split_boards(0) == 7
        assert lib.bf_set_split_boards(1 << 20) == 0
    finally:
        assert lib.bf_set_split_boards(before) == 1 << 20
    assert before == _lib.BF_SPLIT_BOARDS_DEFAULT or "BF_SPLIT_BOARDS" in __import__("os").environ


def test_pack_rejects_wrong_sizes():
    lib = _lib.load()
    blob = np.zeros(10, np.float32)
    out = np.zeros(lib.bf_packed_floats(), np.float32)
    rc = lib.bf_pack_weights(blob.ctypes.data, blob.size, out.ctypes.data, out.size)
    assert rc == _lib.BF_E_ARG
    assert b"expected" in lib.bf_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)


def test_host_encoder_equals_oracle(golden):
    recs = np.frombuffer(golden["records"].tobytes(), dtype=P.COMPACT_DTYPE)
    nib, occ = encoder.struct_array_to_ann_inputs(recs, recs[:0], np.ones(len(recs), bool), np.zeros(0, bool), len(recs))
    assert np.array_equal(nib, golden["piece_filters"])
    assert np.array_equal(occ, golden["occupied_bbs"])


def test_host_encoder_full_stride_masks_and_two_arrays():
    structs = P.test_set(60)            # 806-byte records, unaligned fields
    a, b = structs[:35], structs[35:]
    ma = np.arange(35) % 3 != 1
    mb = np.arange(25) % 2 == 1
    total = int(ma.sum() + mb.sum())
    got = encoder.struct_array_to_ann_inputs(a, b, ma, mb, total)
    exp = E.struct_array_to_ann_inputs(a, b, ma, mb, total)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1])
    got1 = encoder.struct_array_to_ann_inputs(a, a[:0], ma, np.zeros(0, bool), int(ma.sum()))
    exp1 = E.struct_array_to_ann_inputs(a, None, ma)
    assert np.array_equal(got1[0], exp1[0]) and np.array_equal(got1[1], exp1[1])
    empty = encoder.struct_array_to_ann_inputs(a[:0], a[:0], np.zeros(0, bool), np.zeros(0, bool), 0)
    assert empty[0].size == 0 and empty[1].size == 0


def test_start_board_evaluations_with_plain_callable():
    # any callable with the wire signature plugs in (code_testing.py:523-532)
    structs = P.test_set(30)
    mask = np.arange(30) % 4 != 0
    t, scores = encoder.start_board_evaluations(structs, mask, E.bf_piece_sum_eval)
    t.join()
    exp = E.bf_piece_sum_eval(*E.struct_array_to_ann_inputs(structs[mask]))
    assert scores.dtype == np.float32 and np.array_equal(scores, exp.astype(np.float32))


@pytest.mark.parametrize("dtype,split,tol", [("f32", 0, 1e-5), ("f32s", 1, 5e-5), ("f32s3", 2, 1e-5)])
def test_packed_image_drives_kernel_model_to_oracle_scores(golden, dtype, split, tol):
    """Walks the fragment stream the way the HIP kernel does (tests/kernel_sim.py), for each of the three precisions
    that share the kernel: the float32 image, and the images whose 1x1 pair and 3x3 layers are bf16 terms
    (BF_PRECISION_F32_SPLIT, BF_PRECISION_F32_SPLIT3; the kernel's operand splits and product selection are modelled,
    its float32 sums are float64 here)."""
    lib = _lib.load()
    w = W.make_weights(seed=1234, bn="perturbed")
    blob = W.to_blob(w)
    packed = np.zeros(lib.bf_packed_floats(), np.float32)
    _lib.check(lib.bf_pack_weights_ex(blob.ctypes.data, blob.size, packed.ctypes.data, packed.size, _lib.precision_code(dtype)))
    if split == 0:
        again = np.zeros_like(packed)
        _lib.check(lib.bf_pack_weights(blob.ctypes.data, blob.size, again.ctypes.data, again.size))
        assert np.array_equal(packed.view(np.uint32), again.view(np.uint32))
    for b in (1, 7, 500):
        score, inter = kernel_sim.eval_board(packed, golden["mailbox"][b], split)
        ref = golden["scores_f64_perturbed"][b]
        assert abs(score - ref) <= tol * max(abs(ref), 0.2), (b, score, ref)
    assert lib.bf_pack_weights_ex(blob.ctypes.data, blob.size, packed.ctypes.data, packed.size, _lib.BF_PRECISION_BF16) == _lib.BF_E_ARG


def test_weights_blob_round_trip():
    w = W.make_weights(seed=3, bn="perturbed")
    w2 = W.from_blob(W.to_blob(w))
    for a, b in zip(w["convs"], w2["convs"]):
        assert np.array_equal(a, b)
    for x, y in zip(w["bn"], w2["bn"]):
        for a, b in zip(x, y):
            assert np.array_equal(a, b)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from batch_first_amd import chestimator
    with pytest.raises((RuntimeError, ValueError)):
        chestimator.get_inference_functions(W.to_blob(W.make_weights()))


def test_t16_image_drives_kernel_model_to_bf16_oracle(golden):
    """The image of the 16x16x32 bf16 kernel (bf_pack_weights_t16) walked lane by lane as eval_t16.hip walks
    it (tests/kernel_sim_t16.py): every bf16 activation equals oracle.network.forward_bf16_model's, i.e. the
    fragment order, the k permutations between layers, the tap pairing and tile skipping of stage 1 and the
    plane channel order are what the packer and the kernel agree on."""
    import kernel_sim_t16
    lib = _lib.load()
    w = W.make_weights(seed=1234, bn="perturbed")
    blob = W.to_blob(w)
    image = np.zeros(lib.bf_packed_t16_bytes(), np.uint8)
    _lib.check(lib.bf_pack_weights_t16(blob.ctypes.data, blob.size, image.ctypes.data, image.size))
    idx = [3, 777]
    scores, inters, mfmas = kernel_sim_t16.eval_boards(image, golden["mailbox"][idx])
    planes = E.decode_planes_fast(golden["piece_filters"], golden["occupied_bbs"])[idx]
    ref, ri = N.forward_bf16_model(planes, w, return_intermediates=True, double=True)
    for name in ("x1", "x3", "x4", "x5", "x9"):
        got = np.stack([i[name] for i in inters])
        assert np.array_equal(got, ri[name].reshape(got.shape)), name   # bf16 values: exact
    assert np.max(np.abs(scores - ref)) <= 1e-6
    assert mfmas == 2468    # 39 488 matrix-pipe cycles a board: DESIGN.md 3.2
    assert lib.bf_pack_weights_t16(blob.ctypes.data, blob.size, image.ctypes.data, image.size - 1) == _lib.BF_E_ARG
