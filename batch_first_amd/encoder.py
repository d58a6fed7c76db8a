"""Host-side callers of the evaluation path, with the reference's names and signatures.

* :func:`struct_array_to_ann_inputs` -- batch_first/numba_negamax_zero_window.py:71-105,
  backed by the native host encoder (bf_host_encode_wire).
* :func:`start_board_evaluations` -- batch_first/numba_negamax_zero_window.py:183-202: spawns
  the worker thread and returns ``(thread, evaluation_scores)``.  When the evaluator can ingest
  node records itself (``eval_structs``) the host encoder is skipped altogether; any other
  callable with the wire signature (for example the reference's material-count fake,
  code_testing.py:523-532) still works.
"""
import ctypes
import threading

import numpy as np

from . import _lib


def struct_array_to_ann_inputs(child_structs, not_child_structs, to_score_child_mask,
                               to_score_not_child_mask, total_num_to_score):
    """Node records -> (uint8 nibble stream, uint64 relevance masks), children first."""
    lib = _lib.load()
    parts = []
    for structs, mask in ((child_structs, to_score_child_mask), (not_child_structs, to_score_not_child_mask)):
        structs = np.ascontiguousarray(structs)
        if structs.size == 0:
            continue
        if structs.dtype.itemsize < 82:
            raise ValueError("node records must be at least 82 bytes")
        mask = np.ascontiguousarray(mask, dtype=np.bool_)
        if mask.shape != structs.shape:
            raise ValueError("mask and struct array lengths differ")
        parts.append((structs, mask.view(np.uint8)))
    total = int(total_num_to_score)
    occupied_bbs = np.empty(total, dtype=np.uint64)
    # own_concat sizes the stream from the squares actually found (:58); 33 codes a board is the
    # reference's own bound (ann_creation_helper.py:41), 64 is the physical one
    codes_cap = 64 * total // 2 + 1
    piece_filters = np.zeros(codes_cap, dtype=np.uint8)
    if len(parts) == 2:
        # the stream is one run over both arrays, so hand the C encoder one gathered array
        rec_bytes = 82
        sel = [s.view(np.uint8).reshape(len(s), -1)[m.astype(bool), :rec_bytes] for s, m in parts]
        gathered = np.ascontiguousarray(np.concatenate(sel, axis=0))
        structs_ptr, stride, mask_ptr, n = gathered.ctypes.data, rec_bytes, None, len(gathered)
        keep = gathered  # noqa: F841  (keeps the buffer alive across the call)
    elif len(parts) == 1:
        s, m = parts[0]
        structs_ptr, stride, mask_ptr, n = s.ctypes.data, s.dtype.itemsize, m.ctypes.data, len(s)
    else:
        if total:
            raise ValueError("total_num_to_score does not match the masks")
        return piece_filters[:0], occupied_bbs
    n_bytes, n_scored = ctypes.c_size_t(0), ctypes.c_size_t(0)
    if int(sum(int(np.count_nonzero(m)) for _, m in parts)) != total:
        raise ValueError("total_num_to_score does not match the masks")
    _lib.check(lib.bf_host_encode_wire(structs_ptr, stride, mask_ptr, n, piece_filters.ctypes.data,
                                       occupied_bbs.ctypes.data, ctypes.byref(n_bytes), ctypes.byref(n_scored)))
    return piece_filters[:n_bytes.value].copy(), occupied_bbs


def start_board_evaluations(struct_array, to_score_mask, board_eval_fn):
    """Start the evaluation of the selected records on a worker thread."""
    to_score_mask = np.asarray(to_score_mask, dtype=np.bool_)
    num_to_score = int(np.sum(to_score_mask))
    evaluation_scores = np.empty(num_to_score, dtype=np.float32)

    def evaluate_and_set():
        if hasattr(board_eval_fn, "eval_structs"):
            evaluation_scores[:] = board_eval_fn.eval_structs(struct_array, to_score_mask)
        else:
            evaluation_scores[:] = board_eval_fn(
                *struct_array_to_ann_inputs(struct_array, struct_array[:0], to_score_mask,
                                            np.array([], dtype=np.bool_), num_to_score))

    t = threading.Thread(target=evaluate_and_set)
    t.start()
    return t, evaluation_scores
