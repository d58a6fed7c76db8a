"""GraphDef weight import (batch_first_amd/graphdef.py) against the protobuf library's own encoders.

The reference ships no GraphDef, so the messages are built here from the published schema's field
numbers (tensorflow/core/framework/*.proto) with google.protobuf's dynamic descriptors, serialised by
the library in both encodings the reference reads (chestimator.py:14-21), and read back by the
module's own parsers.  The module's writer is checked the other way round: the library parses it.
"""
import numpy as np
import pytest

from batch_first_amd import graphdef as G, weights as W

pb = pytest.importorskip("google.protobuf")
from google.protobuf import descriptor_pb2, descriptor_pool, message_factory, text_format  # noqa: E402

F = descriptor_pb2.FieldDescriptorProto


def _schema():
    fd = descriptor_pb2.FileDescriptorProto(name="bf_test_graph.proto", package="bftest", syntax="proto3")

    def msg(name, parent=None):
        m = (parent.nested_type if parent is not None else fd.message_type).add()
        m.name = name
        return m

    def field(m, name, number, ftype, label=F.LABEL_OPTIONAL, type_name=None, oneof=None):
        f = m.field.add(name=name, number=number, type=ftype, label=label)
        if type_name:
            f.type_name = type_name
        if oneof is not None:
            f.oneof_index = oneof
        return f

    shape = msg("TensorShapeProto")
    dim = msg("Dim", shape)
    field(dim, "size", 1, F.TYPE_INT64)
    field(dim, "name", 2, F.TYPE_STRING)
    field(shape, "dim", 2, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bftest.TensorShapeProto.Dim")
    field(shape, "unknown_rank", 3, F.TYPE_BOOL)

    dt = fd.enum_type.add(name="DataType")
    for name, num in (("DT_INVALID", 0), ("DT_FLOAT", 1), ("DT_DOUBLE", 2), ("DT_INT32", 3), ("DT_INT64", 9)):
        dt.value.add(name=name, number=num)

    tensor = msg("TensorProto")
    field(tensor, "dtype", 1, F.TYPE_ENUM, type_name=".bftest.DataType")
    field(tensor, "tensor_shape", 2, F.TYPE_MESSAGE, type_name=".bftest.TensorShapeProto")
    field(tensor, "version_number", 3, F.TYPE_INT32)
    field(tensor, "tensor_content", 4, F.TYPE_BYTES)
    field(tensor, "float_val", 5, F.TYPE_FLOAT, F.LABEL_REPEATED)
    field(tensor, "int_val", 7, F.TYPE_INT32, F.LABEL_REPEATED)

    attr = msg("AttrValue")
    attr.oneof_decl.add(name="value")
    field(attr, "s", 2, F.TYPE_BYTES, oneof=0)
    field(attr, "i", 3, F.TYPE_INT64, oneof=0)
    field(attr, "f", 4, F.TYPE_FLOAT, oneof=0)
    field(attr, "b", 5, F.TYPE_BOOL, oneof=0)
    field(attr, "type", 6, F.TYPE_ENUM, type_name=".bftest.DataType", oneof=0)
    field(attr, "shape", 7, F.TYPE_MESSAGE, type_name=".bftest.TensorShapeProto", oneof=0)
    field(attr, "tensor", 8, F.TYPE_MESSAGE, type_name=".bftest.TensorProto", oneof=0)

    node = msg("NodeDef")
    entry = msg("AttrEntry", node)
    entry.options.map_entry = True
    field(entry, "key", 1, F.TYPE_STRING)
    field(entry, "value", 2, F.TYPE_MESSAGE, type_name=".bftest.AttrValue")
    field(node, "name", 1, F.TYPE_STRING)
    field(node, "op", 2, F.TYPE_STRING)
    field(node, "input", 3, F.TYPE_STRING, F.LABEL_REPEATED)
    field(node, "device", 4, F.TYPE_STRING)
    field(node, "attr", 5, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bftest.NodeDef.AttrEntry")

    versions = msg("VersionDef")
    field(versions, "producer", 1, F.TYPE_INT32)
    graph = msg("GraphDef")
    field(graph, "node", 1, F.TYPE_MESSAGE, F.LABEL_REPEATED, ".bftest.NodeDef")
    field(graph, "version", 3, F.TYPE_INT32)
    field(graph, "versions", 4, F.TYPE_MESSAGE, type_name=".bftest.VersionDef")

    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    return message_factory.GetMessageClass(pool.FindMessageTypeByName("bftest.GraphDef"))


GraphDef = _schema()


def _library_graph(value, policy, float_val_for=()):
    """A frozen-graph lookalike built with the protobuf library: placeholders and ops around the
    constants, const tensors as tensor_content or (for the names given) float_val."""
    g = GraphDef()
    g.versions.producer = 26
    ph = g.node.add(name="input_parser/piece_filters", op="Placeholder")
    ph.attr["dtype"].type = 3
    ph.attr["shape"].shape.unknown_rank = True
    misc = g.node.add(name="value_network/some_int_const", op="Const")
    misc.attr["dtype"].type = 3
    misc.attr["value"].tensor.dtype = 3
    misc.attr["value"].tensor.int_val.append(7)
    named = G._named_tensors(value, "value_network", 9) + (G._named_tensors(policy, "policy_network", 8) if policy else [])
    for name, arr in named:
        n = g.node.add(name=name, op="Const", device="/device:GPU:0")
        n.attr["dtype"].type = 1
        t = n.attr["value"].tensor
        t.dtype = 1
        for d in arr.shape:
            t.tensor_shape.dim.add(size=d)
        if name in float_val_for:
            t.float_val.extend(float(x) for x in arr.ravel())
        else:
            t.tensor_content = np.ascontiguousarray(arr, "<f4").tobytes()
        r = g.node.add(name=name + "/read", op="Identity")
        r.input.append(name)
        r.attr["T"].type = 1
        r.attr["_class"].s = b"loc:@" + name.encode()
    sq = g.node.add(name="value_network/Squeeze", op="Squeeze")
    sq.input.append("value_network/logit_layer/Conv2D")
    return g


def _same(a, b):
    assert len(a["convs"]) == len(b["convs"]) and len(a["bn"]) == len(b["bn"])
    for x, y in zip(a["convs"], b["convs"]):
        assert x.shape == y.shape and np.array_equal(x, y)
    for p, q in zip(a["bn"], b["bn"]):
        for x, y in zip(p, q):
            assert np.array_equal(x, y)


@pytest.fixture(scope="module")
def nets():
    return W.make_weights(seed=99, bn="perturbed"), W.make_policy_weights(seed=98, bn="perturbed")


def test_reads_library_binary_and_text(tmp_path, nets):
    value, policy = nets
    fv = ("value_network/input_module/batch_normalization/beta", "value_network/logit_layer/kernel")
    g = _library_graph(value, policy, float_val_for=fv)
    (tmp_path / "g.pb").write_bytes(g.SerializeToString())
    (tmp_path / "g.pbtxt").write_text(text_format.MessageToString(g))
    for name in ("g.pb", "g.pbtxt"):
        t = G.read_graphdef(tmp_path / name)
        assert t.ops["value_network/Squeeze"] == "Squeeze"
        assert "value_network/some_int_const" in t          # present, but not a float tensor
        with pytest.raises(G.GraphDefError):
            t.get("value_network/some_int_const")
        _same(G.value_weights(t), value)
        _same(G.policy_weights(t), policy)
        vb, pbl = G.blobs_from_graphdef(tmp_path / name)
        assert np.array_equal(vb, W.to_blob(value)) and np.array_equal(pbl, W.to_policy_blob(policy))


def test_library_reads_our_writer(tmp_path, nets):
    value, policy = nets
    G.write_graphdef(tmp_path / "w.pb", value=value, policy=policy)
    G.write_graphdef(tmp_path / "w.pbtxt", value=value, policy=policy)
    a = GraphDef()
    a.ParseFromString((tmp_path / "w.pb").read_bytes())
    b = text_format.Parse((tmp_path / "w.pbtxt").read_text(), GraphDef())
    assert a == b
    consts = {n.name: n for n in a.node if n.op == "Const"}
    assert len(consts) == 9 + 3 + 9 * 4 + 1 + 9 + 3 + 8 * 4 + 1
    k = consts["policy_network/logit_layer/kernel"].attr["value"].tensor
    assert [d.size for d in k.tensor_shape.dim] == [1, 1, 32, 73]
    assert np.array_equal(np.frombuffer(k.tensor_content, "<f4").reshape(1, 1, 32, 73), policy["convs"][-1])
    # and our reader reads our writer
    for name in ("w.pb", "w.pbtxt"):
        _same(G.value_weights(tmp_path / name), value)
        _same(G.policy_weights(tmp_path / name), policy)


def test_value_only_graph_and_errors(tmp_path, nets):
    value, _ = nets
    G.write_graphdef(tmp_path / "v.pb", value=value)
    vb, pbl = G.blobs_from_graphdef(tmp_path / "v.pb")
    assert pbl is None and vb.size == W.blob_size()
    with pytest.raises(G.GraphDefError):
        G.policy_weights(tmp_path / "v.pb")
    # a kernel of the wrong shape is refused, naming the tensor
    bad = {"convs": [c.copy() for c in value["convs"]], "bn": value["bn"]}
    g = _library_graph(bad, None)
    for n in g.node:
        if n.name == "value_network/module_2/path_1/layer_3/kernel":
            del n.attr["value"].tensor.tensor_shape.dim[:]
            n.attr["value"].tensor.tensor_shape.dim.add(size=128 * 32 * 9)
    (tmp_path / "bad.pb").write_bytes(g.SerializeToString())
    with pytest.raises(G.GraphDefError, match="layer_3/kernel"):
        G.value_weights(tmp_path / "bad.pb")
    # truncated file
    data = (tmp_path / "v.pb").read_bytes()
    (tmp_path / "cut.pb").write_bytes(data[: len(data) // 2])
    with pytest.raises(G.GraphDefError):
        G.read_graphdef(tmp_path / "cut.pb")
    # a gamma (scale=True) or a bias means it is not the reference's architecture
    g = _library_graph(value, None)
    n = g.node.add(name="value_network/input_module/batch_normalization/gamma", op="Const")
    n.attr["value"].tensor.dtype = 1
    n.attr["value"].tensor.tensor_shape.dim.add(size=236)
    (tmp_path / "gamma.pb").write_bytes(g.SerializeToString())
    with pytest.raises(G.GraphDefError, match="gamma"):
        G.value_weights(tmp_path / "gamma.pb")


def test_text_escapes_and_float_val_fill():
    assert G._unescape(r"\000\377a\n\\\"\x41") == b"\x00\xffa\n\\\"A"
    assert G._escape_bytes(b"\x00\xffa\n\\\"A'") == "\\000\\377a\\n\\\\\\\"A\\'"
    rng = np.random.default_rng(0)
    raw = rng.integers(0, 256, 4096, dtype=np.uint8).tobytes()
    assert G._unescape(G._escape_bytes(raw)) == raw
    text = '''
    node { name: "c" op: "Const"
      attr { key: "value" value { tensor { dtype: DT_FLOAT tensor_shape { dim { size: 2 } dim { size: 3 } } float_val: 0.5 } } } }
    node { name: "d" op: "Const"   # scalar without a shape
      attr { key: "value" value { tensor { dtype: DT_FLOAT tensor_shape { } float_val: -1.5e-3 } } } }
    '''
    ops, consts = G._parse_text(text)
    t = G.GraphDefTensors(ops, consts)
    assert np.array_equal(t.get("c"), np.full((2, 3), 0.5, np.float32))
    assert t.get("d").shape == () and t.get("d") == np.float32(-1.5e-3)


@pytest.mark.gpu
def test_factory_takes_a_graphdef_file(tmp_path, nets):
    """get_inference_functions(graphdef_filename) as the reference is called (chestimator.py:51):
    one file, both networks; scores identical to loading the same weights as dicts."""
    from batch_first_amd import chestimator, encoder, positions
    value, policy = nets
    G.write_graphdef(tmp_path / "both.pbtxt", value=value, policy=policy)
    structs = positions.to_compact(positions.random_structs(300, seed=5))
    nib, occ = encoder.struct_array_to_ann_inputs(structs, structs[:0], np.ones(300, bool), np.zeros(0, bool), 300)
    bp_a, start_a, close_a = chestimator.get_inference_functions(str(tmp_path / "both.pbtxt"))
    bp_b, start_b, close_b = chestimator.get_inference_functions(value, policy_weights=policy)
    try:
        assert np.array_equal(bp_a(nib, occ), bp_b(nib, occ))
        info = [np.array([12, 6], np.uint8), np.array([0, 57], np.uint8), np.array([1, 1] + [0] * 298, np.uint8)]
        assert np.array_equal(start_a(nib, occ)(info), start_b(nib, occ)(info))
    finally:
        close_a()
        close_b()
