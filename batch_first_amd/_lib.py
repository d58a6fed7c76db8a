"""ctypes binding of libbatchfirst_hip.so (the C ABI in include/batchfirst.h).

The library is the product: if it is missing this module raises, it never falls back to a
CPU path.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C batch_first_amd/csrc``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BATCHFIRST_LIB: another build of the same library (A/B timing and diagnostics, tools/build_variant.sh)
LIB_PATH = os.environ.get("BATCHFIRST_LIB") or os.path.join(_HERE, "libbatchfirst_hip.so")

BF_OK, BF_E_ARG, BF_E_HIP, BF_E_FORMAT, BF_E_NOGPU = 0, -1, -2, -3, -4
BF_PRECISION_F32, BF_PRECISION_BF16, BF_PRECISION_F32_SPLIT, BF_PRECISION_F32_SPLIT3 = 0, 1, 2, 3
BF_NOALLOC_MAX_BOARDS = 262144
BF_SPLIT_BOARDS_DEFAULT = 512
PRECISIONS = {"f32": BF_PRECISION_F32, "fp32": BF_PRECISION_F32, "float32": BF_PRECISION_F32,
              "bf16": BF_PRECISION_BF16, "bfloat16": BF_PRECISION_BF16,
              "f32s": BF_PRECISION_F32_SPLIT, "f32_split": BF_PRECISION_F32_SPLIT,
              "f32s3": BF_PRECISION_F32_SPLIT3, "f32_split3": BF_PRECISION_F32_SPLIT3}


def precision_code(dtype):
    try:
        return PRECISIONS[str(dtype).replace("torch.", "")]
    except KeyError:
        raise ValueError("dtype must be 'f32', 'bf16', 'f32s' or 'f32s3', got %r" % (dtype,))

_c = ctypes
_vp, _sz, _int = _c.c_void_p, _c.c_size_t, _c.c_int

# name -> (restype, argtypes); must list every symbol include/batchfirst.h declares
SIGNATURES = {
    "bf_last_error": (_c.c_char_p, []),
    "bf_version": (_c.c_char_p, []),
    "bf_ctx_create": (_int, [_c.POINTER(_vp), _int, _vp, _sz]),
    "bf_ctx_create_ex": (_int, [_c.POINTER(_vp), _int, _vp, _sz, _int]),
    "bf_ctx_precision": (_int, [_vp]),
    "bf_set_split_boards": (_sz, [_sz]),
    "bf_ctx_destroy": (None, [_vp]),
    "bf_eval_wire": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "bf_eval_structs": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "bf_eval_structs_noalloc": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "bf_encode_planes_wire": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "bf_encode_planes_structs": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "bf_encode_mailbox_wire": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "bf_encode_mailbox_structs": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "bf_eval_mailbox": (_int, [_vp, _vp, _sz, _vp, _vp]),
    "bf_ctx_check": (_int, [_vp, _vp]),
    "bf_eval_wire_host": (_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "bf_eval_structs_host": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _c.POINTER(_sz)]),
    "bf_multi_create": (_int, [_c.POINTER(_vp), _c.POINTER(_vp), _sz, _sz]),
    "bf_multi_destroy": (None, [_vp]),
    "bf_multi_devices": (_sz, [_vp]),
    "bf_multi_eval_wire_host": (_int, [_vp, _vp, _sz, _vp, _sz, _vp]),
    "bf_multi_eval_structs_host": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _c.POINTER(_sz)]),
    "bf_multi_set_min_shard": (_sz, [_vp, _sz]),
    "bf_multi_policy_start_wire_host": (_int, [_vp, _vp, _sz, _vp, _sz]),
    "bf_multi_policy_start_structs_host": (_int, [_vp, _vp, _sz, _vp, _sz]),
    "bf_multi_policy_finish_host": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "bf_host_split_wire": (_int, [_vp, _sz, _sz, _vp, _vp]),
    "bf_host_encode_wire": (_int, [_vp, _sz, _vp, _sz, _vp, _vp, _c.POINTER(_sz), _c.POINTER(_sz)]),
    "bf_packed_floats": (_sz, []),
    "bf_pack_weights": (_int, [_vp, _sz, _vp, _sz]),
    "bf_pack_weights_ex": (_int, [_vp, _sz, _vp, _sz, _int]),
    "bf_pack_policy_weights": (_int, [_vp, _sz, _vp, _sz]),
    "bf_ctx_set_policy": (_int, [_vp, _vp, _sz]),
    "bf_policy_features_wire": (_int, [_vp, _vp, _sz, _vp, _sz, _vp, _vp]),
    "bf_policy_features_structs": (_int, [_vp, _vp, _sz, _sz, _vp, _vp]),
    "bf_policy_scores": (_int, [_vp, _vp, _sz, _vp, _vp, _vp, _sz, _vp, _vp]),
    "bf_policy_check": (_int, [_vp, _vp]),
    "bf_policy_start_wire_host": (_int, [_vp, _vp, _sz, _vp, _sz]),
    "bf_policy_start_structs_host": (_int, [_vp, _vp, _sz, _vp, _sz]),
    "bf_policy_finish_host": (_int, [_vp, _vp, _vp, _vp, _sz, _sz, _vp]),
    "bf_packed_t16_bytes": (_sz, []),
    "bf_pack_weights_t16": (_int, [_vp, _sz, _vp, _sz]),
    # debugging aids, not part of the public header
    "bf_eval_mailbox_debug": (_int, [_vp, _vp, _sz, _vp, _vp, _vp]),
    "bf_debug_floats_per_board": (_sz, []),
}

_lib = None


class BatchFirstError(RuntimeError):
    pass


def load():
    """Load the shared library once; raise if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: the HIP extension has not been built (run __graft_entry__.build()); "
            "there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    """Translate a C return code into the exception the boundary documents."""
    if rc == BF_OK:
        return
    msg = load().bf_last_error().decode("utf-8", "replace")
    if rc in (BF_E_ARG, BF_E_FORMAT):
        raise ValueError(msg)
    raise BatchFirstError(msg)
