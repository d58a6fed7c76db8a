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

from . import _lib, graphdef as _graphdef, weights as _weights


def _as_policy_blob(weights):
    if isinstance(weights, (str, os.PathLike)):
        if _graphdef.is_graphdef_path(weights):
            blob = _weights.to_policy_blob(_graphdef.policy_weights(weights))
        else:
            blob = np.load(os.fspath(weights))
    elif isinstance(weights, dict):
        blob = _weights.to_policy_blob(weights)
    else:
        blob = np.asarray(weights)
    blob = np.ascontiguousarray(blob, dtype=np.float32).ravel()
    if blob.size != _weights.policy_blob_size():
        raise ValueError("policy weight blob has %d floats, expected %d" % (blob.size, _weights.policy_blob_size()))
    return blob


def _wire_arrays(piece_filters, occupied_bbs):
    """The feed arrays as raw bytes / bit patterns (the TF placeholders are int32 / int64 fed with
    uint8 / uint64, ann_creation_helper.py:50-51; compress_square_array returns a strided view, :19)."""
    pf = np.ascontiguousarray(piece_filters)
    if pf.dtype != np.uint8:
        if pf.size and (pf.min() < 0 or pf.max() > 255):
            raise ValueError("piece_filters values must be bytes (0..255)")
        pf = pf.astype(np.uint8)
    occ = np.ascontiguousarray(occupied_bbs)
    if occ.dtype == np.int64:
        occ = occ.view(np.uint64)
    elif occ.dtype != np.uint64:
        occ = occ.astype(np.uint64)
    if pf.ndim != 1 or occ.ndim != 1:
        raise ValueError("piece_filters and occupied_bbs must be one-dimensional")
    return pf, occ


def _as_blob(weights):
    if isinstance(weights, (str, os.PathLike)):
        path = os.fspath(weights)
        if _graphdef.is_graphdef_path(path):
            blob = _weights.to_blob(_graphdef.value_weights(path))
        elif path.endswith(".npz"):
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


def _finish_closure(finish_fn, handle_fn):
    """The closure start_move_prediction returns (chestimator.py:44-46) over a C finish entry point."""

    def finish(move_info):
        frm, flt, per = (np.ascontiguousarray(a) for a in move_info)
        for name, a in (("from_squares", frm), ("move_filters", flt), ("moves_per_board", per)):
            if a.dtype != np.uint8:   # uint8 placeholders, move_evaluation_ann.py:244-246
                if a.size and (a.min() < 0 or a.max() > 255):
                    raise ValueError("%s values must fit uint8" % name)
        frm, flt, per = frm.astype(np.uint8, copy=False), flt.astype(np.uint8, copy=False), per.astype(np.uint8, copy=False)
        if frm.shape != flt.shape or frm.ndim != 1 or per.ndim != 1:
            raise ValueError("from_squares and move_filters must be one-dimensional and equally long")
        scores = np.empty(frm.size, dtype=np.float32)
        _lib.check(finish_fn(handle_fn(), frm.ctypes.data, flt.ctypes.data, per.ctypes.data, per.size, frm.size,
                             scores.ctypes.data))
        return scores

    return finish


# min_shard when none is given: the batch size from which two devices finish a call sooner than one, read off the
# one-device latency sweep (profiles/r2/sweep_r2g.json): a device call costs t(n); cutting 2n
# boards in two pays when t(2n) > t(n) + the second shard's enqueue (about 10 us).  bf16: t(512) = 40, t(1024) = 47,
# t(2048) = 75 us -> from 1 024 boards a device; fp32: t(256) = 158, t(512) = 166, t(1024) = 281 us -> from 512.
# measure_min_shard() repeats the measurement on the box at hand.
DEFAULT_MIN_SHARD = {"bf16": 1024, "f32": 512, "f32s": 512, "f32s3": 512}


def measure_min_shard(predictor, sizes=(128, 256, 512, 1024, 2048, 4096), reps=9, enqueue_us=10.0):
    """The smallest shard size s of ``sizes`` with t(2s) > t(s) + enqueue_us on ``predictor``'s device: from 2s boards
    on, a second device shortens the call.  t = median wall time of ``eval_structs`` on synthetic records (host
    pointers in, host scores out: what a shard of the multi-device call costs).  Returns (min_shard, {n: us})."""
    import time
    from . import positions
    recs = positions.to_compact(positions.random_structs(2 * max(sizes), seed=7))
    t = {}
    for n in sorted(set(list(sizes) + [2 * s for s in sizes])):
        predictor.eval_structs(recs[:n])
        samples = []
        for _ in range(reps):
            t0 = time.perf_counter()
            predictor.eval_structs(recs[:n])
            samples.append(time.perf_counter() - t0)
        t[n] = 1e6 * float(np.median(samples))
    for s in sorted(sizes):
        if t[2 * s] > t[s] + enqueue_us:
            return s, t
    return max(sizes), t


class BoardPredictor:
    """Callable evaluator bound to one GPU context."""

    def __init__(self, weights, device=0, dtype="f32"):
        self._lib = _lib.load()
        blob = _as_blob(weights)
        ctx = ctypes.c_void_p()
        _lib.check(self._lib.bf_ctx_create_ex(ctypes.byref(ctx), int(device), blob.ctypes.data, blob.size,
                                              _lib.precision_code(dtype)))
        self._ctx = ctx
        self.dtype = {_lib.BF_PRECISION_F32: "f32", _lib.BF_PRECISION_BF16: "bf16",
                      _lib.BF_PRECISION_F32_SPLIT: "f32s", _lib.BF_PRECISION_F32_SPLIT3: "f32s3"}[_lib.precision_code(dtype)]
        self._closed = False
        self.has_policy = False
        self._close_lock = threading.Lock()
        self.device = int(device)

    # -- the reference's callable ---------------------------------------------------------
    def __call__(self, piece_filters, occupied_bbs):
        ctx = self._handle()
        pf, occ = _wire_arrays(piece_filters, occupied_bbs)
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

    # -- the move-scoring network ------------------------------------------------------------
    def set_policy(self, policy_weights):
        blob = _as_policy_blob(policy_weights)
        _lib.check(self._lib.bf_ctx_set_policy(self._handle(), blob.ctypes.data, blob.size))
        self.has_policy = True

    def start_move_prediction(self, piece_filters, occupied_bbs):
        """chestimator.py:40-46: start stage 1 on the boards, return the closure that takes
        ``move_info = [from_squares, move_filters, moves_per_board]`` and returns float32 scores."""
        ctx = self._handle()
        pf, occ = _wire_arrays(piece_filters, occupied_bbs)
        n = occ.size
        _lib.check(self._lib.bf_policy_start_wire_host(ctx, pf.ctypes.data, pf.size, occ.ctypes.data, n))
        return _finish_closure(self._lib.bf_policy_finish_host, self._handle)

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


class MultiDevicePredictor:
    """The same callable over several GPUs of the node: one :class:`BoardPredictor` (one bf_ctx, one
    stream) per device behind a ``bf_multi`` handle.  Every batch is cut into contiguous shards, the
    wire format at sum-of-popcount boundaries (a shard may begin on an odd nibble); all
    shards are enqueued from the calling thread before any is waited for, and one ``float32[N]`` comes
    back in board order.  Scores are bit-identical to a single device's: a board's score does not
    depend on its place in a batch.  The move-scoring network runs on every device the same way
    (``bf_multi_policy_*``: stage 1 on each device's shard, each device's moves to its second stage).
    ``min_shard``: a device is used only if every device in use gets that many boards; None = the
    crossover read off the one-device sweep (DEFAULT_MIN_SHARD), "measure" = measure_min_shard() on
    the first device at construction."""

    def __init__(self, weights, devices, dtype="f32", min_shard=None, allow_repeats=False):
        devices = [int(d) for d in devices]
        # allow_repeats: several contexts (streams) on one GPU -- how a one-GPU box runs this path in tests
        if not devices or (len(set(devices)) != len(devices) and not allow_repeats):
            raise ValueError("devices must be a non-empty list of distinct GPU indices")
        self._lib = _lib.load()
        blob = _as_blob(weights)
        self.parts = []
        try:
            for d in devices:
                self.parts.append(BoardPredictor(blob, device=d, dtype=dtype))
            arr = (ctypes.c_void_p * len(self.parts))(*[p._ctx for p in self.parts])
            handle = ctypes.c_void_p()
            self.min_shard_times = None
            if min_shard is None:
                min_shard = DEFAULT_MIN_SHARD[self.parts[0].dtype]
            elif min_shard == "measure":
                min_shard, self.min_shard_times = measure_min_shard(self.parts[0])
            self.min_shard = int(min_shard)
            _lib.check(self._lib.bf_multi_create(ctypes.byref(handle), arr, len(self.parts), self.min_shard))
        except Exception:
            for p in self.parts:
                p.close()
            raise
        self._multi = handle
        self.devices = devices
        self.device = devices[0]
        self.dtype = self.parts[0].dtype
        self._closed = False
        self._close_lock = threading.Lock()

    @property
    def has_policy(self):
        return all(p.has_policy for p in self.parts)

    def _multi_handle(self):
        if self._closed:
            raise RuntimeError("the evaluator has been closed")
        return self._multi

    def _handle(self):
        """The first device's context."""
        self._multi_handle()
        return self.parts[0]._handle()

    def __call__(self, piece_filters, occupied_bbs):
        m = self._multi_handle()
        pf, occ = _wire_arrays(piece_filters, occupied_bbs)
        scores = np.empty(occ.size, dtype=np.float32)
        if occ.size == 0:
            if pf.size:
                raise ValueError("piece_filters given without boards")
            return scores
        _lib.check(self._lib.bf_multi_eval_wire_host(m, pf.ctypes.data, pf.size, occ.ctypes.data, occ.size, scores.ctypes.data))
        return scores

    def eval_structs(self, struct_array, to_score_mask=None):
        m = self._multi_handle()
        recs = np.ascontiguousarray(struct_array)
        if recs.ndim != 1 or recs.dtype.itemsize < 82:
            raise ValueError("struct_array must be a 1-d array of node records (>= 82 bytes each)")
        n, mask_ptr, count = recs.size, None, recs.size
        if to_score_mask is not None:
            mask = np.ascontiguousarray(to_score_mask, dtype=np.bool_).view(np.uint8)
            if mask.shape != (n,):
                raise ValueError("to_score_mask must have one entry per record")
            mask_ptr, count = mask.ctypes.data, int(np.count_nonzero(mask))
        scores = np.empty(count, dtype=np.float32)
        if count == 0:
            return scores
        done = ctypes.c_size_t(0)
        _lib.check(self._lib.bf_multi_eval_structs_host(m, recs.ctypes.data, recs.dtype.itemsize, mask_ptr, n,
                                                        scores.ctypes.data, ctypes.byref(done)))
        if done.value != count:
            raise _lib.BatchFirstError("evaluated %d boards, expected %d" % (done.value, count))
        return scores

    def set_policy(self, policy_weights):
        blob = _as_policy_blob(policy_weights)
        for p in self.parts:
            p.set_policy(blob)

    def start_move_prediction(self, piece_filters, occupied_bbs):
        m = self._multi_handle()
        pf, occ = _wire_arrays(piece_filters, occupied_bbs)
        _lib.check(self._lib.bf_multi_policy_start_wire_host(m, pf.ctypes.data, pf.size, occ.ctypes.data, occ.size))
        return _finish_closure(self._lib.bf_multi_policy_finish_host, self._multi_handle)

    def close(self):
        with self._close_lock:
            if not self._closed:
                self._closed = True
                self._lib.bf_multi_destroy(self._multi)
                self._multi = None
                for p in self.parts:
                    p.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_inference_functions(weights, session_gpu_memory=.4, device=0, dtype="f32", policy_weights=None, devices=None,
                            min_shard=None):
    """(board_predictor, start_move_prediction, closer_fn), as batch_first/chestimator.py:51-74.

    ``weights`` stands where ``graphdef_filename`` does: a frozen GraphDef file (``.pb`` binary or
    ``.pbtxt`` text, read without TensorFlow by :mod:`batch_first_amd.graphdef`; when it also holds
    a ``policy_network`` that one is loaded too, as the reference's single file does), a
    ``.npy``/``.npz`` file holding the canonical blob (weights.to_blob), the blob itself, or a
    weights dict.  ``session_gpu_memory`` is accepted for call compatibility and ignored
    (TensorFlow's allocator fraction has no counterpart here).  ``dtype``: "f32" (the parity path)
    or "bf16" (bf16 operands, float32 sums; include/batchfirst.h, BF_PRECISION_BF16), for both
    networks.  ``policy_weights``: the move-scoring network's blob (weights.to_policy_blob), file
    or dict.  ``devices``: a list of GPU indices turns ``board_predictor`` into one callable over all
    of them (:class:`MultiDevicePredictor`: contiguous shards, one stream per device, scores gathered
    to the caller, for both networks; batches under ``min_shard`` boards per device use fewer devices: None =
    the measured crossover of the one-device sweep, "measure" = measure it now).
    """
    del session_gpu_memory
    if isinstance(weights, (str, os.PathLike)) and _graphdef.is_graphdef_path(weights):
        weights, graph_policy = _graphdef.blobs_from_graphdef(weights)
        if policy_weights is None:
            policy_weights = graph_policy
    if devices is not None:
        board_predictor = MultiDevicePredictor(weights, devices, dtype=dtype, min_shard=min_shard)
    else:
        board_predictor = BoardPredictor(weights, device=device, dtype=dtype)
    if policy_weights is not None:
        board_predictor.set_policy(policy_weights)

    def start_move_prediction(*board_inputs):
        if not board_predictor.has_policy:
            # the reference's frozen graph holds both networks; here the policy blob is optional
            raise NotImplementedError("no move-scoring network was loaded (policy_weights=...)")
        return board_predictor.start_move_prediction(*board_inputs)

    def closer_fn():
        board_predictor.close()

    return board_predictor, start_move_prediction, closer_fn
