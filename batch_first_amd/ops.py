"""PyTorch-ROCm custom ops over the C ABI: device tensors in, device tensors out, on torch's
current stream.  PyTorch is plumbing here (allocator, streams, torch.distributed); every op body
is one call into libbatchfirst_hip.so.

    torch.ops.batchfirst.eval_wire(ctx, piece_filters u8[B], occupied_bbs i64[N]) -> f32[N]
    torch.ops.batchfirst.eval_structs(ctx, records u8[N, stride])                 -> f32[N]
    torch.ops.batchfirst.eval_mailbox(ctx, mailbox u8[N, 32])                      -> f32[N]
    torch.ops.batchfirst.encode_planes_wire(ctx, piece_filters, occupied_bbs)      -> f32[N,15,8,8]
    torch.ops.batchfirst.encode_planes_structs(ctx, records)                       -> f32[N,15,8,8]
    torch.ops.batchfirst.encode_mailbox_wire / encode_mailbox_structs              -> u8[N,32]

``ctx`` is the integer handle of a :class:`DeviceContext`.  occupied_bbs travels as int64
because torch has no first-class uint64; only the bits matter (the reference's placeholder is
int64 too, ann_creation_helper.py:51).
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .chestimator import _as_blob, _as_policy_blob

_contexts = {}


class DeviceContext:
    """Owns a bf_ctx on one GPU; ``handle`` is what the ops take."""

    def __init__(self, weights, device=0, dtype="f32"):
        self._lib = _lib.load()
        blob = _as_blob(weights)
        ctx = ctypes.c_void_p()
        _lib.check(self._lib.bf_ctx_create_ex(ctypes.byref(ctx), int(device), blob.ctypes.data, blob.size,
                                              _lib.precision_code(dtype)))
        self._ctx = ctx
        self.device = torch.device("cuda", int(device))
        self.handle = int(ctx.value)
        _contexts[self.handle] = self

    def set_policy(self, policy_weights):
        blob = _as_policy_blob(policy_weights)
        _lib.check(self._lib.bf_ctx_set_policy(self._ctx, blob.ctypes.data, blob.size))

    def check_policy(self):
        _lib.check(self._lib.bf_policy_check(self._ctx, _stream(self.device)))

    def check(self):
        """Synchronise the current stream and raise ValueError if a wire input was malformed."""
        _lib.check(self._lib.bf_ctx_check(self._ctx, _stream(self.device)))

    def close(self):
        if self._ctx is not None:
            _contexts.pop(self.handle, None)
            self._lib.bf_ctx_destroy(self._ctx)
            self._ctx = None


def _stream(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _ctx(handle):
    try:
        return _contexts[int(handle)]
    except KeyError:
        raise ValueError("unknown batchfirst context handle %r" % (handle,))


def _dev(t, dtype, name, c=None):
    if not t.is_cuda:
        raise ValueError("%s must be a device tensor" % name)
    if t.dtype != dtype:
        raise ValueError("%s must have dtype %s" % (name, dtype))
    if c is not None and t.device != c.device:
        # the kernels run on the context's GPU: a pointer and a stream from another one must not reach them
        raise ValueError("%s is on %s, but the context lives on %s" % (name, t.device, c.device))
    return t.contiguous()


def _records(t, c):
    recs = _dev(t, torch.uint8, "records", c)
    if recs.dim() != 2 or recs.shape[1] < 82:
        raise ValueError("records must be uint8[N, stride] with stride >= 82")
    return recs


@torch.library.custom_op("batchfirst::eval_wire", mutates_args=())
def eval_wire(ctx: int, piece_filters: torch.Tensor, occupied_bbs: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    pf = _dev(piece_filters, torch.uint8, "piece_filters", c)
    occ = _dev(occupied_bbs, torch.int64, "occupied_bbs", c)
    out = torch.empty(occ.numel(), dtype=torch.float32, device=occ.device)
    _lib.check(c._lib.bf_eval_wire(c._ctx, pf.data_ptr(), pf.numel(), occ.data_ptr(), occ.numel(), out.data_ptr(),
                                   _stream(occ.device)))
    return out


@torch.library.custom_op("batchfirst::eval_structs", mutates_args=())
def eval_structs(ctx: int, records: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    recs = _records(records, c)
    out = torch.empty(recs.shape[0], dtype=torch.float32, device=recs.device)
    _lib.check(c._lib.bf_eval_structs(c._ctx, recs.data_ptr(), recs.shape[1], recs.shape[0], out.data_ptr(),
                                      _stream(recs.device)))
    return out


@torch.library.custom_op("batchfirst::eval_mailbox", mutates_args=())
def eval_mailbox(ctx: int, mailbox: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    mb = _dev(mailbox, torch.uint8, "mailbox", c)
    if mb.dim() != 2 or mb.shape[1] != 32:
        raise ValueError("mailbox must be uint8[N, 32]")
    out = torch.empty(mb.shape[0], dtype=torch.float32, device=mb.device)
    _lib.check(c._lib.bf_eval_mailbox(c._ctx, mb.data_ptr(), mb.shape[0], out.data_ptr(), _stream(mb.device)))
    return out


@torch.library.custom_op("batchfirst::encode_planes_wire", mutates_args=())
def encode_planes_wire(ctx: int, piece_filters: torch.Tensor, occupied_bbs: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    pf = _dev(piece_filters, torch.uint8, "piece_filters", c)
    occ = _dev(occupied_bbs, torch.int64, "occupied_bbs", c)
    out = torch.empty((occ.numel(), 15, 8, 8), dtype=torch.float32, device=occ.device)
    _lib.check(c._lib.bf_encode_planes_wire(c._ctx, pf.data_ptr(), pf.numel(), occ.data_ptr(), occ.numel(),
                                            out.data_ptr(), _stream(occ.device)))
    return out


@torch.library.custom_op("batchfirst::encode_planes_structs", mutates_args=())
def encode_planes_structs(ctx: int, records: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    recs = _records(records, c)
    out = torch.empty((recs.shape[0], 15, 8, 8), dtype=torch.float32, device=recs.device)
    _lib.check(c._lib.bf_encode_planes_structs(c._ctx, recs.data_ptr(), recs.shape[1], recs.shape[0],
                                               out.data_ptr(), _stream(recs.device)))
    return out


@torch.library.custom_op("batchfirst::encode_mailbox_wire", mutates_args=())
def encode_mailbox_wire(ctx: int, piece_filters: torch.Tensor, occupied_bbs: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    pf = _dev(piece_filters, torch.uint8, "piece_filters", c)
    occ = _dev(occupied_bbs, torch.int64, "occupied_bbs", c)
    out = torch.empty((occ.numel(), 32), dtype=torch.uint8, device=occ.device)
    _lib.check(c._lib.bf_encode_mailbox_wire(c._ctx, pf.data_ptr(), pf.numel(), occ.data_ptr(), occ.numel(),
                                             out.data_ptr(), _stream(occ.device)))
    return out


@torch.library.custom_op("batchfirst::encode_mailbox_structs", mutates_args=())
def encode_mailbox_structs(ctx: int, records: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    recs = _records(records, c)
    out = torch.empty((recs.shape[0], 32), dtype=torch.uint8, device=recs.device)
    _lib.check(c._lib.bf_encode_mailbox_structs(c._ctx, recs.data_ptr(), recs.shape[1], recs.shape[0],
                                                out.data_ptr(), _stream(recs.device)))
    return out


@torch.library.custom_op("batchfirst::policy_features_wire", mutates_args=())
def policy_features_wire(ctx: int, piece_filters: torch.Tensor, occupied_bbs: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    pf = _dev(piece_filters, torch.uint8, "piece_filters", c)
    occ = _dev(occupied_bbs, torch.int64, "occupied_bbs", c)
    out = torch.empty((occ.numel(), 32, 8, 8), dtype=torch.float32, device=occ.device)
    _lib.check(c._lib.bf_policy_features_wire(c._ctx, pf.data_ptr(), pf.numel(), occ.data_ptr(), occ.numel(),
                                              out.data_ptr(), _stream(occ.device)))
    return out


@torch.library.custom_op("batchfirst::policy_features_structs", mutates_args=())
def policy_features_structs(ctx: int, records: torch.Tensor) -> torch.Tensor:
    c = _ctx(ctx)
    recs = _records(records, c)
    out = torch.empty((recs.shape[0], 32, 8, 8), dtype=torch.float32, device=recs.device)
    _lib.check(c._lib.bf_policy_features_structs(c._ctx, recs.data_ptr(), recs.shape[1], recs.shape[0],
                                                 out.data_ptr(), _stream(recs.device)))
    return out


@torch.library.custom_op("batchfirst::policy_scores", mutates_args=())
def policy_scores(ctx: int, features: torch.Tensor, from_squares: torch.Tensor, move_filters: torch.Tensor,
                  moves_per_board: torch.Tensor) -> torch.Tensor:
    """Stage 2: one score per move; the three move tensors are uint8 device tensors."""
    c = _ctx(ctx)
    x9 = _dev(features, torch.float32, "features", c)
    frm = _dev(from_squares, torch.uint8, "from_squares", c)
    flt = _dev(move_filters, torch.uint8, "move_filters", c)
    per = _dev(moves_per_board, torch.uint8, "moves_per_board", c)
    n = x9.shape[0]
    if per.numel() != n or frm.numel() != flt.numel():
        raise ValueError("moves_per_board needs one entry per board, from_squares and move_filters one per move")
    # np.repeat of the board index, as numpy_style_repeat_1d_creator does (ann_creation_helper.py:488-510)
    board_index = torch.repeat_interleave(torch.arange(n, device=x9.device, dtype=torch.int32), per.to(torch.int64))
    if board_index.numel() != frm.numel():
        raise ValueError("moves_per_board sums to %d, but %d moves were given" % (board_index.numel(), frm.numel()))
    out = torch.empty(frm.numel(), dtype=torch.float32, device=x9.device)
    _lib.check(c._lib.bf_policy_scores(c._ctx, x9.data_ptr(), n, board_index.data_ptr(), frm.data_ptr(), flt.data_ptr(),
                                       frm.numel(), out.data_ptr(), _stream(x9.device)))
    return out


def eval_mailbox_debug(ctx, mailbox):
    """Scores plus every intermediate activation (testing aid; not a registered op)."""
    c = _ctx(ctx)
    mb = _dev(mailbox, torch.uint8, "mailbox", c)
    n = mb.shape[0]
    per = c._lib.bf_debug_floats_per_board()
    scores = torch.empty(n, dtype=torch.float32, device=mb.device)
    dbg = torch.zeros((n, per), dtype=torch.float32, device=mb.device)
    _lib.check(c._lib.bf_eval_mailbox_debug(c._ctx, mb.data_ptr(), n, scores.data_ptr(), dbg.data_ptr(),
                                            _stream(mb.device)))
    return scores, dbg


def records_to_tensor(structs, device):
    """View a NumPy array of node records as a uint8[N, stride] device tensor."""
    raw = np.ascontiguousarray(structs).view(np.uint8).reshape(len(structs), -1)
    return torch.from_numpy(raw.copy()).to(device)
