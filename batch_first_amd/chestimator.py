"""The inference boundary: the callables the search is handed.

Mirrors batch_first/chestimator.py:51-74: ``get_inference_functions`` returns the triple
``(board_predictor, start_move_prediction, closer_fn)``.  ``board_predictor`` keeps the
reference's positional signature ``(piece_filters, occupied_bbs) -> float32[N]``
(chestimator.py:38, caller batch_first/numba_negamax_zero_window.py:191-197) and runs the
hand-written gfx950 kernels through the C ABI; the ctypes call releases the GIL for its whole
duration, which is the overlap the search's worker threads rely on (:199-200).

There is no CPU fallback: without the built HIP library or without a GPU this raises.
"""
import ctypes
import os
import threading

import numpy as np

from . import _lib, weights as _weights


def _as_blob(weights):
    if isinstance(weights, (str, os.PathLike)):
        path = os.fspath(weights)
        if path.endswith(".npz"):
            with np.load(path) as z:
                blob = z["blob"]
        else:
            blob = np.load(path)
    elif isinstance(weights, dict):
        blob = _weights.to_blob(weights)
    else:
        blob = np.asarray(weights)
    blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
    if blob.size != _weights.blob_size():
        raise ValueError("weight blob has %d floats, expected %d" % (blob.size, _weights.blob_size()))
    return blob


class BoardPredictor:
    """Callable evaluator bound to one GPU context."""

    def __init__(self, weights, device=0, dtype="f32"):
        self._lib = _lib.load()
        blob = _as_blob(weights)
        ctx = ctypes.c_void_p()
        _lib.check(self._lib.bf_ctx_create_ex(ctypes.byref(ctx), int(device), blob.ctypes.data, blob.size,
                                              _lib.precision_code(dtype)))
        self._ctx = ctx
        self.dtype = "bf16" if _lib.precision_code(dtype) == _lib.BF_PRECISION_BF16 else "f32"
        self._closed = False
        self._close_lock = threading.Lock()
        self.device = int(device)

    # -- the reference's callable ---------------------------------------------------------
    def __call__(self, piece_filters, occupied_bbs):
        ctx = self._handle()
        # to_compress[::2] in compress_square_array (numba_negamax_zero_window.py:19) is a strided view
        pf = np.ascontiguousarray(piece_filters)
        if pf.dtype != np.uint8:
            # the TF placeholder is int32 "given as an array of uint8s" (ann_creation_helper.py:50)
            if pf.size and (pf.min() < 0 or pf.max() > 255):
                raise ValueError("piece_filters values must be bytes (0..255)")
            pf = pf.astype(np.uint8)
        occ = np.ascontiguousarray(occupied_bbs)
        if occ.dtype == np.int64:
            occ = occ.view(np.uint64)  # placeholder is int64 fed with uint64 bits (:51)
        elif occ.dtype != np.uint64:
            occ = occ.astype(np.uint64)
        if pf.ndim != 1 or occ.ndim != 1:
            raise ValueError("piece_filters and occupied_bbs must be one-dimensional")
        n = occ.size
        scores = np.empty(n, dtype=np.float32)
        if n == 0:
            if pf.size:
                raise ValueError("piece_filters given without boards")
            return scores
        _lib.check(self._lib.bf_eval_wire_host(ctx, pf.ctypes.data, pf.size, occ.ctypes.data, n, scores.ctypes.data))
        return scores

    # -- the same scores straight from node records (no host encoder) ----------------------
    def eval_structs(self, struct_array, to_score_mask=None):
        ctx = self._handle()
        recs = np.ascontiguousarray(struct_array)
        if recs.ndim != 1 or recs.dtype.itemsize < 82:
            raise ValueError("struct_array must be a 1-d array of node records (>= 82 bytes each)")
        n = recs.size
        mask_ptr = None
        count = n
        if to_score_mask is not None:
            mask = np.ascontiguousarray(to_score_mask, dtype=np.bool_).view(np.uint8)
            if mask.shape != (n,):
                raise ValueError("to_score_mask must have one entry per record")
            mask_ptr = mask.ctypes.data
            count = int(np.count_nonzero(mask))
        scores = np.empty(count, dtype=np.float32)
        if count == 0:
            return scores
        done = ctypes.c_size_t(0)
        _lib.check(self._lib.bf_eval_structs_host(ctx, recs.ctypes.data, recs.dtype.itemsize, mask_ptr, n,
                                                  scores.ctypes.data, ctypes.byref(done)))
        if done.value != count:
            raise _lib.BatchFirstError("evaluated %d boards, expected %d" % (done.value, count))
        return scores

    def _handle(self):
        if self._closed:
            raise RuntimeError("the evaluator has been closed")
        return self._ctx

    def close(self):
        with self._close_lock:
            if not self._closed:
                self._closed = True
                self._lib.bf_ctx_destroy(self._ctx)
                self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_inference_functions(weights, session_gpu_memory=.4, device=0, dtype="f32"):
    """(board_predictor, start_move_prediction, closer_fn), as batch_first/chestimator.py:51-74.

    ``weights`` replaces ``graphdef_filename``: a ``.npy``/``.npz`` file holding the canonical
    blob (weights.to_blob), the blob itself, or a weights dict.  ``session_gpu_memory`` is
    accepted for call compatibility and ignored (TensorFlow's allocator fraction has no
    counterpart here).  ``dtype``: "f32" (the parity path) or "bf16" (bf16 operands, float32 sums;
    include/batchfirst.h, BF_PRECISION_BF16).
    """
    del session_gpu_memory
    board_predictor = BoardPredictor(weights, device=device, dtype=dtype)

    def start_move_prediction(*board_inputs):
        # The policy network (move_evaluation_ann.py) is outside this build's scope.
        raise NotImplementedError("the move-scoring network is not part of this build")

    def closer_fn():
        board_predictor.close()

    return board_predictor, start_move_prediction, closer_fn
