"""ctypes binding of libbatchfirst_search.so: the native search driver that feeds the evaluators.

Mirrors what the reference's engine wires together (batch_first/engine.py:211-288,
numba_negamax_zero_window.py:696-870): ``Search(board_eval, move_eval, max_batch_size, ...)`` then
``zero_window`` / ``mtd_f`` / ``iterative_deepening``.  The evaluators are C function pointers; for
the GPU path they are the HIP library's own entry points (bf_eval_structs_host,
bf_policy_start_structs_host, bf_policy_finish_host), so no Python runs inside the search loop.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# BATCHFIRST_SEARCH_LIB: another build of the search library (A/B timing on one box)
LIB_PATH = os.environ.get("BATCHFIRST_SEARCH_LIB") or os.path.join(_HERE, "libbatchfirst_search.so")

_c = ctypes
BOARD_EVAL_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                             _c.POINTER(_c.c_size_t))
MOVE_START_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t)
MOVE_FINISH_FN = _c.CFUNCTYPE(_c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_size_t,
                              _c.c_void_p)


class Config(_c.Structure):
    _fields_ = [("max_batch_size", _c.c_size_t), ("tt_bits", _c.c_int), ("bin_edges", _c.POINTER(_c.c_float)),
                ("n_bin_edges", _c.c_size_t), ("bin_zero_shift", _c.c_float), ("eval_zero_shift", _c.c_float),
                ("eval_user", _c.c_void_p), ("board_eval", _c.c_void_p), ("move_user", _c.c_void_p),
                ("move_start", _c.c_void_p), ("move_finish", _c.c_void_p), ("max_nodes", _c.c_size_t),
                ("threads", _c.c_int)]


class Stats(_c.Structure):
    _fields_ = [("iterations", _c.c_uint64), ("nodes_out", _c.c_uint64), ("nodes_in", _c.c_uint64),
                ("boards_evaluated", _c.c_uint64), ("moves_scored", _c.c_uint64), ("max_batch", _c.c_uint64),
                ("seconds", _c.c_double), ("eval_wait_seconds", _c.c_double), ("eval_calls", _c.c_uint64),
                ("policy_calls", _c.c_uint64), ("expand_seconds", _c.c_double), ("settle_seconds", _c.c_double),
                ("policy_seconds", _c.c_double), ("value_wait_seconds", _c.c_double), ("update_seconds", _c.c_double),
                ("compact_seconds", _c.c_double), ("select_seconds", _c.c_double), ("arena_peak", _c.c_uint64),
                ("update_busy_seconds", _c.c_double), ("select_insert_seconds", _c.c_double),
                ("select_pop_seconds", _c.c_double), ("select_live_seconds", _c.c_double), ("compact_copy_seconds", _c.c_double),
                ("wide_iterations", _c.c_uint64),
                ("overlapped_updates", _c.c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with make -C batch_first_amd/csrc/search" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        lib.bfs_last_error.restype = _c.c_char_p
        for name in ("bfs_perft", "bfs_legal_moves", "bfs_push", "bfs_record_from_fen", "bfs_random_playout_records",
                     "bfs_negamax", "bfs_search_create", "bfs_zero_window", "bfs_mtd_f", "bfs_iterative_deepening"):
            getattr(lib, name).restype = _c.c_int
        lib.bfs_perft.argtypes = [_c.c_char_p, _c.c_int, _c.POINTER(_c.c_uint64)]
        lib.bfs_legal_moves.argtypes = [_c.c_char_p, _c.c_void_p, _c.POINTER(_c.c_int)]
        lib.bfs_legal_moves_record.restype = _c.c_int
        lib.bfs_legal_moves_record.argtypes = [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_int)]
        lib.bfs_push.argtypes = [_c.c_char_p, _c.c_void_p, _c.c_char_p, _c.c_size_t]
        for name in ("bfs_is_legal", "bfs_push_record", "bfs_hash_walk"):
            getattr(lib, name).restype = _c.c_int
        lib.bfs_is_legal.argtypes = [_c.c_char_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p]
        lib.bfs_push_record.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_void_p]
        lib.bfs_hash_walk.argtypes = [_c.c_char_p, _c.c_int, _c.c_int, _c.c_uint64, _c.POINTER(_c.c_uint64),
                                      _c.POINTER(_c.c_uint64)]
        lib.bfs_material_eval.restype = _c.c_int
        lib.bfs_material_eval.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_size_t, _c.c_void_p,
                                          _c.POINTER(_c.c_size_t)]
        lib.bfs_record_from_fen.argtypes = [_c.c_char_p, _c.c_void_p]
        lib.bfs_random_playout_records.argtypes = [_c.c_char_p, _c.c_int, _c.c_uint64, _c.c_void_p, _c.POINTER(_c.c_int)]
        lib.bfs_negamax.argtypes = [_c.c_char_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_float)]
        lib.bfs_search_create.argtypes = [_c.POINTER(_c.c_void_p), _c.POINTER(Config)]
        lib.bfs_search_destroy.argtypes = [_c.c_void_p]
        lib.bfs_search_destroy.restype = None
        lib.bfs_search_threads.argtypes = [_c.c_void_p]
        lib.bfs_search_threads.restype = _c.c_int
        lib.bfs_search_clear_tt.argtypes = [_c.c_void_p]
        lib.bfs_search_clear_tt.restype = None
        lib.bfs_search_stats.argtypes = [_c.c_void_p, _c.POINTER(Stats), _c.c_int]
        lib.bfs_search_stats.restype = None
        lib.bfs_zero_window.argtypes = [_c.c_void_p, _c.c_char_p, _c.c_int, _c.c_float, _c.POINTER(_c.c_float)]
        lib.bfs_mtd_f.argtypes = [_c.c_void_p, _c.c_char_p, _c.c_int, _c.c_float, _c.c_float, _c.POINTER(_c.c_float),
                                  _c.c_void_p]
        lib.bfs_iterative_deepening.argtypes = [_c.c_void_p, _c.c_char_p, _c.POINTER(_c.c_int), _c.c_int,
                                                _c.POINTER(_c.c_float), _c.c_void_p]
        _lib = lib
    return _lib


def _check(rc):
    if rc:
        raise RuntimeError(load().bfs_last_error().decode("utf-8", "replace"))


def perft(fen, depth):
    out = _c.c_uint64(0)
    _check(load().bfs_perft(fen.encode(), int(depth), _c.byref(out)))
    return out.value


def legal_moves(fen):
    buf = np.zeros((100, 3), np.uint8)
    n = _c.c_int(0)
    _check(load().bfs_legal_moves(fen.encode(), buf.ctypes.data, _c.byref(n)))
    return buf[:n.value].copy()


def legal_moves_of_record(record):
    """Legal moves of one node record (any stride >= 82 bytes)."""
    rec = np.ascontiguousarray(record).view(np.uint8).ravel()
    buf = np.zeros((100, 3), np.uint8)
    n = _c.c_int(0)
    _check(load().bfs_legal_moves_record(rec.ctypes.data, buf.ctypes.data, _c.byref(n)))
    return buf[:n.value].copy()


def push(fen, move):
    mv = np.asarray(move, np.uint8)
    out = _c.create_string_buffer(128)
    _check(load().bfs_push(fen.encode(), mv.ctypes.data, out, 128))
    return out.value.decode()


def is_legal(fen, moves):
    """(by_generator, direct): legality of each (from, to, promotion) triple, decided by membership in the
    generated legal moves and from the move alone (code_testing.py:266-293)."""
    mv = np.ascontiguousarray(moves, np.uint8).reshape(-1, 3)
    a, b = np.zeros(len(mv), np.uint8), np.zeros(len(mv), np.uint8)
    _check(load().bfs_is_legal(fen.encode(), mv.ctypes.data, len(mv), a.ctypes.data, b.ctypes.data))
    return a.astype(bool), b.astype(bool)


def push_record(record, move):
    """The 88-byte child record of a node record after a legal move."""
    rec = np.ascontiguousarray(record).view(np.uint8).ravel()
    mv = np.ascontiguousarray(move, np.uint8)
    out = np.zeros(88, np.uint8)
    _check(load().bfs_push_record(rec.ctypes.data, mv.ctypes.data, out.ctypes.data))
    return out


def hash_walk(fen, lines=1000, plies=20, seed=1):
    """(positions checked, mismatches) of incremental against recomputed hashes (code_testing.py:86-136)."""
    n, bad = _c.c_uint64(0), _c.c_uint64(0)
    _check(load().bfs_hash_walk(fen.encode(), int(lines), int(plies), int(seed), _c.byref(n), _c.byref(bad)))
    return n.value, bad.value


def record_from_fen(fen):
    rec = np.zeros(88, np.uint8)
    _check(load().bfs_record_from_fen(fen.encode(), rec.ctypes.data))
    return rec


def random_playout_records(fen, plies, seed):
    recs = np.zeros((plies, 88), np.uint8)
    n = _c.c_int(0)
    _check(load().bfs_random_playout_records(fen.encode(), int(plies), int(seed), recs.ctypes.data, _c.byref(n)))
    return recs[:n.value].copy()


def material_eval_pointer():
    """The reference's fake evaluator (code_testing.py:516-532) as a native function pointer."""
    return _c.cast(load().bfs_material_eval, _c.c_void_p).value


def negamax(fen, depth, eval_pointer=None, eval_user=None):
    out = _c.c_float(0)
    _check(load().bfs_negamax(fen.encode(), int(depth), eval_pointer or material_eval_pointer(), eval_user, _c.byref(out)))
    return out.value


class Search:
    """One search instance: transposition table, open-node bins, evaluator hooks."""

    def __init__(self, board_eval=None, eval_user=None, move_start=None, move_finish=None, move_user=None,
                 max_batch_size=5000, tt_bits=22, bin_edges=None, bin_zero_shift=0.0, eval_zero_shift=0.0,
                 max_nodes=0, threads=0):
        """max_nodes: limit of the node arena (0 = 2^22 nodes, 3.5 GB); a search that would pass it fails
        with an error instead of growing.  threads: host threads for the per-node loops of a batch (0 = the job's cores
        minus four, at most 16);
        results do not depend on it."""
        self._lib = load()
        if bin_edges is None:
            bin_edges = np.linspace(-4000, 4000, 1000)   # code_testing.py:602
        self._bins = np.ascontiguousarray(bin_edges, np.float32)
        self._keep = []

        def ptr(fn, proto):
            if fn is None:
                return None
            if isinstance(fn, int):
                return fn
            cb = proto(fn)          # a Python callable: keep the thunk alive
            self._keep.append(cb)
            return _c.cast(cb, _c.c_void_p).value

        cfg = Config(int(max_batch_size), int(tt_bits), self._bins.ctypes.data_as(_c.POINTER(_c.c_float)), self._bins.size,
                     float(bin_zero_shift), float(eval_zero_shift), eval_user,
                     ptr(board_eval, BOARD_EVAL_FN) or material_eval_pointer(), move_user,
                     ptr(move_start, MOVE_START_FN), ptr(move_finish, MOVE_FINISH_FN), int(max_nodes), int(threads))
        self._s = _c.c_void_p()
        _check(self._lib.bfs_search_create(_c.byref(self._s), _c.byref(cfg)))

    @classmethod
    def on_gpu(cls, predictor, use_policy=True, **kw):
        """Search driven by a batch_first_amd.chestimator.BoardPredictor: the HIP library's host
        entry points are called straight from the native loop."""
        from . import _lib as hip
        lib = hip.load()
        ctx = predictor._handle()
        if hasattr(predictor, "_multi_handle"):   # chestimator.MultiDevicePredictor: same signature, handle as user
            args = dict(board_eval=_c.cast(lib.bf_multi_eval_structs_host, _c.c_void_p).value,
                        eval_user=predictor._multi_handle())
        else:
            args = dict(board_eval=_c.cast(lib.bf_eval_structs_host, _c.c_void_p).value, eval_user=ctx)
        if use_policy and predictor.has_policy:
            if hasattr(predictor, "_multi_handle"):
                args.update(move_start=_c.cast(lib.bf_multi_policy_start_structs_host, _c.c_void_p).value,
                            move_finish=_c.cast(lib.bf_multi_policy_finish_host, _c.c_void_p).value,
                            move_user=predictor._multi_handle())
            else:
                args.update(move_start=_c.cast(lib.bf_policy_start_structs_host, _c.c_void_p).value,
                            move_finish=_c.cast(lib.bf_policy_finish_host, _c.c_void_p).value, move_user=ctx)
        args.update(kw)
        s = cls(**args)
        s._keep.append(predictor)
        return s

    def zero_window(self, fen, depth, separator):
        out = _c.c_float(0)
        _check(self._lib.bfs_zero_window(self._s, fen.encode(), int(depth), float(separator), _c.byref(out)))
        return out.value

    def mtd_f(self, fen, depth, first_guess=0.0, guess_increment=0.05):
        out = _c.c_float(0)
        mv = np.zeros(3, np.uint8)
        _check(self._lib.bfs_mtd_f(self._s, fen.encode(), int(depth), float(first_guess), float(guess_increment),
                                   _c.byref(out), mv.ctypes.data))
        return out.value, mv

    def iterative_deepening(self, fen, depths):
        d = (_c.c_int * len(depths))(*[int(x) for x in depths])
        out = _c.c_float(0)
        mv = np.zeros(3, np.uint8)
        _check(self._lib.bfs_iterative_deepening(self._s, fen.encode(), d, len(depths), _c.byref(out), mv.ctypes.data))
        return out.value, mv

    def stats(self, reset=False):
        st = Stats()
        self._lib.bfs_search_stats(self._s, _c.byref(st), 1 if reset else 0)
        return st.as_dict()

    @property
    def threads(self):
        return self._lib.bfs_search_threads(self._s)

    def clear_tt(self):
        self._lib.bfs_search_clear_tt(self._s)

    def close(self):
        if self._s:
            self._lib.bfs_search_destroy(self._s)
            self._s = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
