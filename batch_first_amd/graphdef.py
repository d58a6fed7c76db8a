"""Weight import from a TensorFlow frozen GraphDef, without TensorFlow.

The reference's engine is handed a GraphDef file holding both networks with their variables frozen
into ``Const`` nodes (batch_first/chestimator.py:14-21: ``.pb`` = binary protobuf, anything else =
protobuf text format; written by ann_creation_helper.py:138-146, :164-168, :189).  This module reads
either encoding with a small protobuf wire / text parser of its own, picks the kernels and
batch-norm statistics out by the scope names the reference's graph builders give them, and returns
the weights dicts / canonical blobs the C ABI takes (weights.to_blob, weights.to_policy_blob):

    value_network/input_module/3x3_adjacency_filter/kernel            ann_creation_helper.py:206-218
    value_network/input_module/input_KxK_with_AxB_dilation/kernel     :221-235
    value_network/input_module/batch_normalization/{moving_mean,moving_variance,beta}   :240-246
    value_network/module_2/path_1/layer_I/kernel  (I = 1..9)          :291-310, evaluation_ann.py:32,268
    value_network/module_2/path_1/batch_normalization[_J]/...         :312-317 (J-th default-named layer in the scope)
    value_network/logit_layer/kernel                                  evaluation_ann.py:35-44
    policy_network/...                                                the same with 8 tower layers and a
                                                                      [1,1,32,73] logit kernel, move_evaluation_ann.py:18-45

Kernels are HWIO whatever the graph's data format (tf.layers.conv2d).  Only float32 ``Const`` tensors
are read (``tensor_content`` or ``float_val``).  A graph whose constants were folded into TensorRT
engine nodes (ann_creation_helper.py:171-189) no longer holds them as tensors and is rejected.

Parity unpinned: the reference ships no GraphDef (README.md:89), so the names above follow the
graph-building code and TensorFlow 1.10's naming rules, not a file.  The tests check the parser against
the protobuf library's own encoder on messages built from the published schema field numbers.

A writer (:func:`write_graphdef`) is included so weights made here can be handed to TensorFlow-side
tools, and for the tests.
"""
import os
import re
import struct

import numpy as np

from . import spec, weights as _weights

DT_FLOAT = 1

# field numbers of tensorflow/core/framework/{graph,node_def,attr_value,tensor,tensor_shape}.proto
_GRAPH_NODE = 1
_NODE_NAME, _NODE_OP, _NODE_INPUT, _NODE_ATTR = 1, 2, 3, 5
_MAP_KEY, _MAP_VALUE = 1, 2
_ATTR_TYPE, _ATTR_TENSOR = 6, 8
_TENSOR_DTYPE, _TENSOR_SHAPE, _TENSOR_CONTENT, _TENSOR_FLOAT_VAL = 1, 2, 4, 5
_SHAPE_DIM, _DIM_SIZE = 2, 1


class GraphDefError(ValueError):
    pass


# ---- binary wire format ------------------------------------------------------------------------

def _varint(buf, pos):
    result = 0
    shift = 0
    while True:
        if pos >= len(buf):
            raise GraphDefError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise GraphDefError("varint too long")


def _fields(buf):
    """Yield (field number, wire type, value) over one message; value is an int for varint /
    fixed types and a memoryview for length-delimited ones."""
    pos = 0
    end = len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            value, pos = _varint(buf, pos)
        elif wire == 1:
            if pos + 8 > end:
                raise GraphDefError("truncated fixed64")
            value = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wire == 2:
            size, pos = _varint(buf, pos)
            if pos + size > end:
                raise GraphDefError("length-delimited field runs past the end of its message")
            value = buf[pos:pos + size]
            pos += size
        elif wire == 5:
            if pos + 4 > end:
                raise GraphDefError("truncated fixed32")
            value = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise GraphDefError("unsupported wire type %d" % wire)
        yield field, wire, value


def _tensor_from_wire(buf):
    dtype = 0
    dims = []
    content = None
    float_val = []
    for field, wire, value in _fields(buf):
        if field == _TENSOR_DTYPE and wire == 0:
            dtype = value
        elif field == _TENSOR_SHAPE and wire == 2:
            for f2, w2, v2 in _fields(value):
                if f2 == _SHAPE_DIM and w2 == 2:
                    size = 0
                    for f3, w3, v3 in _fields(v2):
                        if f3 == _DIM_SIZE and w3 == 0:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    dims.append(size)
        elif field == _TENSOR_CONTENT and wire == 2:
            content = bytes(value)
        elif field == _TENSOR_FLOAT_VAL:
            if wire == 2:      # packed
                float_val.append(np.frombuffer(bytes(value), "<f4"))
            elif wire == 5:
                float_val.append(np.array([value], "<u4").view("<f4"))
    return dtype, dims, content, (np.concatenate(float_val) if float_val else None)


def _parse_binary(data):
    consts = {}
    ops = {}
    buf = memoryview(data)
    for field, wire, node in _fields(buf):
        if field != _GRAPH_NODE or wire != 2:
            continue
        name = op = None
        tensor = None
        for f2, w2, v2 in _fields(node):
            if f2 == _NODE_NAME and w2 == 2:
                name = bytes(v2).decode("utf-8")
            elif f2 == _NODE_OP and w2 == 2:
                op = bytes(v2).decode("utf-8")
            elif f2 == _NODE_ATTR and w2 == 2:
                key = None
                val = None
                for f3, w3, v3 in _fields(v2):
                    if f3 == _MAP_KEY and w3 == 2:
                        key = bytes(v3)
                    elif f3 == _MAP_VALUE and w3 == 2:
                        val = v3
                if key == b"value" and val is not None:
                    for f4, w4, v4 in _fields(val):
                        if f4 == _ATTR_TENSOR and w4 == 2:
                            tensor = v4
        if name is None:
            continue
        ops[name] = op
        if op == "Const" and tensor is not None:
            consts[name] = _tensor_from_wire(tensor)
    return ops, consts


# ---- protobuf text format ----------------------------------------------------------------------

_TOKEN = re.compile(r"""\s*(?:(\#[^\n]*)|([{}<>:;,\[\]])|("(?:[^"\\\n]|\\.)*"|'(?:[^'\\\n]|\\.)*')|([^\s{}<>:;,\[\]"'#]+))""")
_SIMPLE_ESCAPES = {"n": 10, "r": 13, "t": 9, "\\": 92, "'": 39, '"': 34, "a": 7, "b": 8, "f": 12, "v": 11, "?": 63}


def _unescape(body):
    """C-style escaped text (as protobuf's text format writes bytes fields) -> bytes."""
    out = bytearray()
    i = 0
    n = len(body)
    while i < n:
        ch = body[i]
        if ch != "\\":
            out += ch.encode("utf-8")
            i += 1
            continue
        i += 1
        if i >= n:
            raise GraphDefError("dangling backslash in a string")
        ch = body[i]
        if ch in "01234567":
            j = i
            while j < n and j < i + 3 and body[j] in "01234567":
                j += 1
            out.append(int(body[i:j], 8) & 0xFF)
            i = j
        elif ch in "xX":
            j = i + 1
            while j < n and j < i + 3 and body[j] in "0123456789abcdefABCDEF":
                j += 1
            if j == i + 1:
                raise GraphDefError("bad \\x escape")
            out.append(int(body[i + 1:j], 16))
            i = j
        elif ch in _SIMPLE_ESCAPES:
            out.append(_SIMPLE_ESCAPES[ch])
            i += 1
        else:
            raise GraphDefError("unknown escape \\%s" % ch)
    return bytes(out)


def _tokens(text):
    pos = 0
    n = len(text)
    while True:
        m = _TOKEN.match(text, pos)
        if not m:
            if text[pos:].strip():
                raise GraphDefError("cannot tokenise text-format GraphDef at offset %d" % pos)
            return
        pos = m.end()
        if m.group(1) is not None:
            continue
        if m.group(2) is not None:
            yield "p", m.group(2)
        elif m.group(3) is not None:
            yield "s", m.group(3)[1:-1]
        else:
            yield "w", m.group(4)
        if pos >= n:
            return


def _parse_text_message(tok, closer):
    """Generic text-format message -> {field: [values]}; a value is a dict (sub-message), a
    ("s", escaped text) pair or a bare word."""
    msg = {}
    while True:
        kind, val = next(tok, (None, None))
        if kind is None:
            if closer is None:
                return msg
            raise GraphDefError("text-format GraphDef ends inside a message")
        if kind == "p" and val == closer:
            return msg
        if kind == "p" and val in ";,":
            continue
        if kind != "w":
            raise GraphDefError("expected a field name, found %r" % (val,))
        field = val
        kind, val = next(tok, (None, None))
        if kind == "p" and val == ":":
            kind, val = next(tok, (None, None))
        if kind == "p" and val in "{<":
            msg.setdefault(field, []).append(_parse_text_message(tok, "}" if val == "{" else ">"))
        elif kind == "p" and val == "[":
            items = msg.setdefault(field, [])
            while True:
                kind, val = next(tok, (None, None))
                if kind is None:
                    raise GraphDefError("unterminated list")
                if kind == "p" and val == "]":
                    break
                if kind == "p" and val == ",":
                    continue
                items.append(("s", val) if kind == "s" else val)
        elif kind == "s":
            # adjacent string literals concatenate
            msg.setdefault(field, []).append(("s", val))
        elif kind == "w":
            msg.setdefault(field, []).append(val)
        else:
            raise GraphDefError("field %s has no value" % field)


def _text_str(v):
    return _unescape(v[1]).decode("utf-8") if isinstance(v, tuple) else str(v)


def _text_float(word):
    w = word.rstrip("fF") if not word.lower().endswith(("inf", "nan")) else word
    return float(w)


def _parse_text(text):
    consts = {}
    ops = {}
    graph = _parse_text_message(_tokens(text), None)
    for node in graph.get("node", []):
        if "name" not in node:
            continue
        name = _text_str(node["name"][0])
        op = _text_str(node["op"][0]) if "op" in node else None
        ops[name] = op
        if op != "Const":
            continue
        for attr in node.get("attr", []):
            if "key" not in attr or _text_str(attr["key"][0]) != "value":
                continue
            for value in attr.get("value", []):
                for tensor in value.get("tensor", []):
                    dt = tensor.get("dtype", ["DT_INVALID"])[0]
                    dtype = DT_FLOAT if dt in ("DT_FLOAT", "1") else -1
                    dims = []
                    for shape in tensor.get("tensor_shape", []):
                        for dim in shape.get("dim", []):
                            dims.append(int(dim.get("size", ["0"])[0]))
                    content = None
                    if "tensor_content" in tensor:
                        content = b"".join(_unescape(v[1]) for v in tensor["tensor_content"])
                    float_val = None
                    if "float_val" in tensor:
                        float_val = np.array([_text_float(w) for w in tensor["float_val"]], np.float32)
                    consts[name] = (dtype, dims, content, float_val)
    return ops, consts


# ---- tensors -> weights --------------------------------------------------------------------------

def _materialise(name, entry):
    dtype, dims, content, float_val = entry
    if dtype != DT_FLOAT:
        raise GraphDefError("%s: dtype %s, expected DT_FLOAT" % (name, dtype))
    if any(d < 0 for d in dims):
        raise GraphDefError("%s: unknown dimension in a constant" % name)
    count = int(np.prod(dims)) if dims else 1
    if content is not None and len(content):
        if len(content) != 4 * count:
            raise GraphDefError("%s: tensor_content holds %d bytes, shape %s needs %d" % (name, len(content), dims, 4 * count))
        arr = np.frombuffer(content, "<f4").astype(np.float32)
    elif float_val is not None and float_val.size:
        if float_val.size == count:
            arr = float_val.astype(np.float32)
        elif float_val.size < count:
            # TensorProto semantics: the last value given repeats to fill the shape
            arr = np.concatenate([float_val, np.full(count - float_val.size, float_val[-1], np.float32)])
        else:
            raise GraphDefError("%s: %d float_val entries for shape %s" % (name, float_val.size, dims))
    elif count == 0:
        arr = np.zeros(0, np.float32)
    else:
        arr = np.zeros(count, np.float32)   # an all-default TensorProto is all zeros
    return arr.reshape(dims)


class GraphDefTensors:
    """The float32 constants of a GraphDef by node name, and every node's op."""

    def __init__(self, ops, consts):
        self.ops = ops
        self._consts = consts

    def names(self):
        return sorted(self._consts)

    def __contains__(self, name):
        return name in self._consts

    def get(self, name):
        if name not in self._consts:
            raise GraphDefError("the graph has no constant named %r" % name)
        return _materialise(name, self._consts[name])


def read_graphdef(path):
    """Read a GraphDef file the way the reference chooses the decoder (chestimator.py:14-21):
    ``.pb`` is binary, anything else text format."""
    path = os.fspath(path)
    if path.endswith(".pb"):
        with open(path, "rb") as fh:
            ops, consts = _parse_binary(fh.read())
    else:
        with open(path, "r") as fh:
            ops, consts = _parse_text(fh.read())
    if not ops:
        raise GraphDefError("%s holds no graph nodes" % path)
    return GraphDefTensors(ops, consts)


def _input_conv_names():
    """Layer names of the nine input convolutions, in concat order (ann_creation_helper.py:218-234)."""
    names = ["3x3_adjacency_filter"]
    dilations = [(d, d) for d in range(2, 8)] + [(2, 4), (4, 2)]
    sizes = [3] * 6 + [2] * 2
    for (a, b), k in zip(dilations, sizes):
        names.append("input_%dx%d_with_%dx%d_dilation" % (k, k, a, b))
    return names


def _bn_scopes(tensors, scope):
    """Default-named batch-norm layers directly under ``scope``, in creation order:
    batch_normalization, batch_normalization_1, ... (tf.variable_scope default_name rule)."""
    pat = re.compile(re.escape(scope) + r"/batch_normalization(?:_(\d+))?/moving_mean$")
    found = []
    for name in tensors.names():
        m = pat.match(name)
        if m:
            found.append((int(m.group(1) or 0), name[:-len("/moving_mean")]))
    return [s for _, s in sorted(found)]


def _bn(tensors, bn_scope, channels):
    out = []
    for leaf in ("moving_mean", "moving_variance", "beta"):
        a = tensors.get(bn_scope + "/" + leaf)
        if a.shape != (channels,):
            raise GraphDefError("%s/%s has shape %s, expected (%d,)" % (bn_scope, leaf, a.shape, channels))
        out.append(np.ascontiguousarray(a, np.float32))
    if "%s/gamma" % bn_scope in tensors:
        raise GraphDefError("%s has a gamma: the reference's layers are scale=False (ann_creation_helper.py:243,314)" % bn_scope)
    return tuple(out)


def _kernel(tensors, layer_scope, shape):
    if "%s/bias" % layer_scope in tensors:
        raise GraphDefError("%s has a bias: the reference's convolutions are use_bias=False" % layer_scope)
    a = tensors.get(layer_scope + "/kernel")
    if tuple(a.shape) != tuple(shape):
        raise GraphDefError("%s/kernel has shape %s, expected HWIO %s" % (layer_scope, tuple(a.shape), tuple(shape)))
    return np.ascontiguousarray(a, np.float32)


def _network(tensors, scope, shapes, chans, what):
    if any(op == "TRTEngineOp" for op in tensors.ops.values()) and not any(n.startswith(scope + "/") for n in tensors.names()):
        raise GraphDefError("the %s network's constants are inside TensorRT engine nodes; import the graph "
                            "as written before save_trt_graphdef" % what)
    n_tower = len(shapes) - 10
    convs = []
    for name, shape in zip(_input_conv_names(), shapes[:9]):
        convs.append(_kernel(tensors, "%s/input_module/%s" % (scope, name), shape))
    in_bn = _bn_scopes(tensors, scope + "/input_module")
    if len(in_bn) != 1:
        raise GraphDefError("expected one batch-norm layer under %s/input_module, found %d" % (scope, len(in_bn)))
    bns = [_bn(tensors, in_bn[0], chans[0])]
    path = scope + "/module_2/path_1"          # num_previous_modules=1, one module, one path
    tower_bn = _bn_scopes(tensors, path)
    if len(tower_bn) != n_tower:
        raise GraphDefError("expected %d batch-norm layers under %s, found %d" % (n_tower, path, len(tower_bn)))
    for i in range(n_tower):
        convs.append(_kernel(tensors, "%s/layer_%d" % (path, i + 1), shapes[9 + i]))
        bns.append(_bn(tensors, tower_bn[i], chans[1 + i]))
    convs.append(_kernel(tensors, scope + "/logit_layer", shapes[-1]))
    return {"convs": convs, "bn": bns}


def value_weights(source, scope="value_network"):
    """Weights dict of the value network (weights.to_blob takes it) from a GraphDef file or a
    :class:`GraphDefTensors`."""
    tensors = source if isinstance(source, GraphDefTensors) else read_graphdef(source)
    return _network(tensors, scope, spec.layer_shapes(), spec.bn_channels(), "value")


def policy_weights(source, scope="policy_network"):
    """Weights dict of the move-scoring network (weights.to_policy_blob takes it)."""
    tensors = source if isinstance(source, GraphDefTensors) else read_graphdef(source)
    return _network(tensors, scope, spec.policy_layer_shapes(), spec.policy_bn_channels(), "move-scoring")


def has_network(source, scope):
    tensors = source if isinstance(source, GraphDefTensors) else read_graphdef(source)
    return any(n.startswith(scope + "/") for n in tensors.names())


def is_graphdef_path(path):
    return os.fspath(path).endswith((".pb", ".pbtxt", ".pbtext", ".graphdef"))


# ---- writer ----------------------------------------------------------------------------------------

def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _enc_len(field, payload):
    return _enc_varint(field << 3 | 2) + _enc_varint(len(payload)) + payload


def _enc_int(field, v):
    return _enc_varint(field << 3) + _enc_varint(v)


def _named_tensors(weights, scope, n_tower):
    convs, bns = weights["convs"], weights["bn"]
    out = []
    for name, k in zip(_input_conv_names(), convs[:9]):
        out.append(("%s/input_module/%s/kernel" % (scope, name), k))
    for leaf, a in zip(("moving_mean", "moving_variance", "beta"), bns[0]):
        out.append(("%s/input_module/batch_normalization/%s" % (scope, leaf), a))
    for i in range(n_tower):
        out.append(("%s/module_2/path_1/layer_%d/kernel" % (scope, i + 1), convs[9 + i]))
        bn_scope = "%s/module_2/path_1/batch_normalization%s" % (scope, "" if i == 0 else "_%d" % i)
        for leaf, a in zip(("moving_mean", "moving_variance", "beta"), bns[1 + i]):
            out.append(("%s/%s" % (bn_scope, leaf), a))
    out.append(("%s/logit_layer/kernel" % scope, convs[-1]))
    return out


def _escape_bytes(data):
    """protobuf text format's CEscape: printable ASCII as is, the rest as three-digit octal."""
    out = []
    for b in data:
        if b == 10:
            out.append("\\n")
        elif b == 13:
            out.append("\\r")
        elif b == 9:
            out.append("\\t")
        elif b == 34:
            out.append('\\"')
        elif b == 39:
            out.append("\\'")
        elif b == 92:
            out.append("\\\\")
        elif 32 <= b < 127:
            out.append(chr(b))
        else:
            out.append("\\%03o" % b)
    return "".join(out)


def write_graphdef(path, value=None, policy=None):
    """Write a frozen-graph-shaped GraphDef holding the given networks' variables as ``Const`` nodes
    (plus the ``<name>/read`` Identity node TensorFlow's freezing leaves beside each).  ``.pb`` ->
    binary, otherwise text format."""
    tensors = []
    if value is not None:
        tensors += _named_tensors(value, "value_network", 9)
    if policy is not None:
        tensors += _named_tensors(policy, "policy_network", 8)
    path = os.fspath(path)
    if path.endswith(".pb"):
        blob = bytearray()
        for name, arr in tensors:
            arr = np.ascontiguousarray(arr, "<f4")
            shape = b"".join(_enc_len(_SHAPE_DIM, _enc_int(_DIM_SIZE, d)) for d in arr.shape)
            tensor = _enc_int(_TENSOR_DTYPE, DT_FLOAT) + _enc_len(_TENSOR_SHAPE, shape) + _enc_len(_TENSOR_CONTENT, arr.tobytes())
            attr_dtype = _enc_len(_MAP_KEY, b"dtype") + _enc_len(_MAP_VALUE, _enc_int(_ATTR_TYPE, DT_FLOAT))
            attr_value = _enc_len(_MAP_KEY, b"value") + _enc_len(_MAP_VALUE, _enc_len(_ATTR_TENSOR, tensor))
            node = (_enc_len(_NODE_NAME, name.encode()) + _enc_len(_NODE_OP, b"Const")
                    + _enc_len(_NODE_ATTR, attr_dtype) + _enc_len(_NODE_ATTR, attr_value))
            blob += _enc_len(_GRAPH_NODE, node)
            attr_t = _enc_len(_MAP_KEY, b"T") + _enc_len(_MAP_VALUE, _enc_int(_ATTR_TYPE, DT_FLOAT))
            read = (_enc_len(_NODE_NAME, (name + "/read").encode()) + _enc_len(_NODE_OP, b"Identity")
                    + _enc_len(_NODE_INPUT, name.encode()) + _enc_len(_NODE_ATTR, attr_t))
            blob += _enc_len(_GRAPH_NODE, read)
        with open(path, "wb") as fh:
            fh.write(bytes(blob))
    else:
        with open(path, "w") as fh:
            for name, arr in tensors:
                arr = np.ascontiguousarray(arr, "<f4")
                fh.write('node {\n  name: "%s"\n  op: "Const"\n' % name)
                fh.write('  attr {\n    key: "dtype"\n    value {\n      type: DT_FLOAT\n    }\n  }\n')
                fh.write('  attr {\n    key: "value"\n    value {\n      tensor {\n        dtype: DT_FLOAT\n        tensor_shape {\n')
                for d in arr.shape:
                    fh.write("          dim {\n            size: %d\n          }\n" % d)
                fh.write('        }\n        tensor_content: "%s"\n      }\n    }\n  }\n}\n' % _escape_bytes(arr.tobytes()))
                fh.write('node {\n  name: "%s/read"\n  op: "Identity"\n  input: "%s"\n' % (name, name))
                fh.write('  attr {\n    key: "T"\n    value {\n      type: DT_FLOAT\n    }\n  }\n}\n')


def blobs_from_graphdef(path):
    """(value blob, policy blob or None) from one GraphDef file, as get_inference_functions needs."""
    tensors = read_graphdef(path)
    value_blob = _weights.to_blob(value_weights(tensors))
    policy_blob = _weights.to_policy_blob(policy_weights(tensors)) if has_network(tensors, "policy_network") else None
    return value_blob, policy_blob
