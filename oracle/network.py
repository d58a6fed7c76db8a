"""ORACLE (test infrastructure, not product): CPU restatement of the reference's value network.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

PARITY UNPINNED for scores: the reference ships no weights (README.md:89) and no golden scores,
and its arithmetic lives in TensorFlow 1.10 (README.md:74), which is not installable here.  What
is restated is the graph the reference *builds* --
batch_first/anns/ann_creation_helper.py:192-252 (nine bias-free input convs, concat, BN, ReLU),
:256-331 (tower layers: conv, BN(scale=False), ReLU), batch_first/anns/evaluation_ann.py:17-46
(value net, 1x1 logit, squeeze) with the spec at :22 and :268 -- under TensorFlow's published
semantics: cross-correlation, HWIO kernels, SAME = pad (k-1)*d in total with floor(/2) first,
VALID 8x8 -> 1x1, inference batch-norm y = (x - mean) / sqrt(var + 1e-3) + beta (no gamma).
Two independent implementations (torch-CPU float32 with explicit zero padding; NumPy float64
direct sums over taps) are cross-checked in tests/ and against closed-form single-piece cases.
"""
import numpy as np

BN_EPS = 1e-3  # tf.layers.batch_normalization default epsilon

# (kernel, (dilation_h, dilation_w)) of the nine input convs, ann_creation_helper.py:211-236
INPUT_CONV_GEOMETRY = [(3, (1, 1))] + [(3, (d, d)) for d in range(2, 8)] + [(2, (2, 4)), (2, (4, 2))]
# tower spec, evaluation_ann.py:268: list => SAME, tuple => VALID
CONVOLUTIONAL_MODULES = [[[[512, 1], [128, 1]] + 6 * [[32, 3]] + [(16, 8)]]]


def _same_pads(k, d):
    total = (k - 1) * d
    before = total // 2
    return before, total - before


def forward_torch(planes, weights, return_intermediates=False, num_threads=None):
    """float32 forward on the CPU with torch.  ``planes``: [N,15,8,8] float32 (NCHW, as the
    reference's DATA_FORMAT, evaluation_ann.py:254).  Returns float32[N]."""
    import torch
    import torch.nn.functional as F

    if num_threads is not None:
        torch.set_num_threads(num_threads)
    convs, bns = weights["convs"], weights["bn"]
    inter = {}

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32))

    def conv(x, w_hwio, dil=(1, 1), padding="same"):
        w = t(np.transpose(w_hwio, (3, 2, 0, 1)))  # HWIO -> OIHW; conv2d is cross-correlation
        k = w_hwio.shape[0]
        if padding == "same":
            pt, pb = _same_pads(k, dil[0])
            pl, pr = _same_pads(k, dil[1])
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, w, dilation=dil)

    def bn_relu(x, bn):
        mean, var, beta = (t(a).view(1, -1, 1, 1) for a in bn)
        return torch.relu((x - mean) / torch.sqrt(var + BN_EPS) + beta)

    with torch.no_grad():
        x0 = t(planes)
        paths = [conv(x0, convs[i], dil) for i, (_, dil) in enumerate(INPUT_CONV_GEOMETRY)]
        x = bn_relu(torch.cat(paths, dim=1), bns[0])
        inter["x1"] = x
        li = 9
        for section in CONVOLUTIONAL_MODULES[0][0]:
            padding = "valid" if isinstance(section, tuple) else "same"
            x = bn_relu(conv(x, convs[li], (1, 1), padding), bns[li - 8])
            inter["x%d" % (li - 7)] = x
            li += 1
        logits = conv(x, convs[18], (1, 1), "valid")
        scores = logits.reshape(-1)
    if return_intermediates:
        return scores.numpy(), {k: v.numpy() for k, v in inter.items()}
    return scores.numpy()


def forward_numpy64(planes, weights, return_intermediates=False):
    """float64 forward with explicit sums over taps (no conv library)."""
    convs, bns = weights["convs"], weights["bn"]
    inter = {}

    def conv(x, w_hwio, dil=(1, 1), padding="same"):
        n, cin, h, wd = x.shape
        kh, kw, _, cout = w_hwio.shape
        w = w_hwio.astype(np.float64)
        if padding == "valid":
            oh, ow = h - (kh - 1) * dil[0], wd - (kw - 1) * dil[1]
            out = np.zeros((n, cout, oh, ow))
            for i in range(kh):
                for j in range(kw):
                    patch = x[:, :, i * dil[0]:i * dil[0] + oh, j * dil[1]:j * dil[1] + ow]
                    out += np.einsum("nchw,co->nohw", patch, w[i, j])
            return out
        pt, _ = _same_pads(kh, dil[0])
        pl, _ = _same_pads(kw, dil[1])
        out = np.zeros((n, cout, h, wd))
        for i in range(kh):
            for j in range(kw):
                dr, df = i * dil[0] - pt, j * dil[1] - pl
                r0, r1 = max(0, -dr), min(h, h - dr)
                f0, f1 = max(0, -df), min(wd, wd - df)
                if r0 >= r1 or f0 >= f1:
                    continue
                patch = x[:, :, r0 + dr:r1 + dr, f0 + df:f1 + df]
                out[:, :, r0:r1, f0:f1] += np.einsum("nchw,co->nohw", patch, w[i, j])
        return out

    def bn_relu(x, bn):
        mean, var, beta = (a.astype(np.float64).reshape(1, -1, 1, 1) for a in bn)
        return np.maximum((x - mean) / np.sqrt(var + BN_EPS) + beta, 0.0)

    x0 = np.asarray(planes, np.float64)
    paths = [conv(x0, convs[i], dil) for i, (_, dil) in enumerate(INPUT_CONV_GEOMETRY)]
    x = bn_relu(np.concatenate(paths, axis=1), bns[0])
    inter["x1"] = x
    li = 9
    for section in CONVOLUTIONAL_MODULES[0][0]:
        padding = "valid" if isinstance(section, tuple) else "same"
        x = bn_relu(conv(x, convs[li], (1, 1), padding), bns[li - 8])
        inter["x%d" % (li - 7)] = x
        li += 1
    scores = conv(x, convs[18], (1, 1), "valid").reshape(-1)
    if return_intermediates:
        return scores, inter
    return scores


def bf16_round(a):
    """float32 -> nearest bf16 (ties to even), returned as float32.  Finite inputs."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def fold_batch_norm(weights):
    """(kernels, shifts) with inference batch-norm folded in float32, the way the GPU library's
    packers do it: W' = W * inv, shift = beta - mean * inv, inv = 1 / sqrt(var + 1e-3).
    Kernels: the 9 input convs, the 9 tower layers, the logit kernel (untouched).
    Shifts: one per batch-norm layer (input module, 9 tower layers)."""
    convs, bns = weights["convs"], weights["bn"]
    inv, shift = [], []
    for mean, var, beta in bns:
        s = (np.float32(1.0) / np.sqrt(var.astype(np.float32) + np.float32(BN_EPS))).astype(np.float32)
        inv.append(s)
        shift.append((beta.astype(np.float32) - mean.astype(np.float32) * s).astype(np.float32))
    kernels = []
    base = 0
    for i in range(9):
        c = convs[i].shape[3]
        kernels.append((convs[i].astype(np.float32) * inv[0][base:base + c]).astype(np.float32))
        base += c
    for j in range(9):
        kernels.append((convs[9 + j].astype(np.float32) * inv[1 + j]).astype(np.float32))
    kernels.append(convs[18].astype(np.float32))
    return kernels, shift


def forward_bf16_model(planes, weights, return_intermediates=False, double=False):
    """CPU model of the library's BF16 precision (include/batchfirst.h, BF_PRECISION_BF16; NO
    reference counterpart -- the reference evaluates in FP32 only): folded kernels rounded to
    bf16, every activation x1..x9 rounded to bf16 after its ReLU, float32 (or float64 with
    ``double``) sums, exact float32 shifts, x10 and the logit unrounded."""
    import torch
    import torch.nn.functional as F

    kernels, shift = fold_batch_norm(weights)
    dt = torch.float64 if double else torch.float32
    inter = {}

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dt)

    def conv(x, w_hwio, dil=(1, 1), padding="same"):
        w = t(np.transpose(bf16_round(w_hwio), (3, 2, 0, 1)))
        k = w_hwio.shape[0]
        if padding == "same":
            pt, pb = _same_pads(k, dil[0])
            pl, pr = _same_pads(k, dil[1])
            x = F.pad(x, (pl, pr, pt, pb))
        return F.conv2d(x, w, dilation=dil)

    def act(x, sh, round_out=True):
        y = torch.relu(x + t(sh).view(1, -1, 1, 1))
        if round_out:
            y = t(bf16_round(y.to(torch.float32).numpy()))
        return y

    with torch.no_grad():
        x0 = t(planes)
        paths = [conv(x0, kernels[i], dil) for i, (_, dil) in enumerate(INPUT_CONV_GEOMETRY)]
        x = act(torch.cat(paths, dim=1), shift[0])
        inter["x1"] = x
        li = 9
        for section in CONVOLUTIONAL_MODULES[0][0]:
            padding = "valid" if isinstance(section, tuple) else "same"
            x = act(conv(x, kernels[li], (1, 1), padding), shift[li - 8], round_out=(padding == "same"))
            inter["x%d" % (li - 7)] = x
            li += 1
        w_logit = t(np.transpose(kernels[18], (3, 2, 0, 1)))
        scores = F.conv2d(x, w_logit).reshape(-1)
    if return_intermediates:
        return scores.to(torch.float32).numpy(), {k: v.to(torch.float32).numpy() for k, v in inter.items()}
    return scores.to(torch.float32).numpy()


def bf16_model_layer(x_in, weights, layer, double=False):
    """One layer of forward_bf16_model from a given (bf16-valued) input: ``layer`` = "x3" maps x1
    -> x3 through the fused 1x1 pair (x2 rounded in between), "x4".."x9" map the previous 3x3
    input, "x10" maps x9 -> x10 (unrounded).  Lets a test follow the GPU layer by layer from the
    GPU's own inputs, so that one flipped rounding does not grow through later layers."""
    import torch
    import torch.nn.functional as F

    kernels, shift = fold_batch_norm(weights)
    dt = torch.float64 if double else torch.float32

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dt)

    def step(x, li, padding, round_out):
        w = t(np.transpose(bf16_round(kernels[li]), (3, 2, 0, 1)))
        k = kernels[li].shape[0]
        if padding == "same":
            p = _same_pads(k, 1)
            x = F.pad(x, (p[0], p[1], p[0], p[1]))
        y = torch.relu(F.conv2d(x, w) + t(shift[li - 8]).view(1, -1, 1, 1))
        if round_out:
            y = t(bf16_round(y.to(torch.float32).numpy()))
        return y

    with torch.no_grad():
        x = t(x_in)
        if layer == "x3":
            x = step(step(x, 9, "same", True), 10, "same", True)
        elif layer == "x10":
            x = step(x, 17, "valid", False)
        else:
            x = step(x, 7 + int(layer[1:]), "same", True)
    return x.to(torch.float32).numpy()


# ---- the move-scoring ("policy") network ---------------------------------------------------
# batch_first/anns/move_evaluation_ann.py:10-73: input convs (:18-23), tower
# CONVOLUTIONAL_MODULES = [[[[512,1],[128,1]] + 6*[[32,3]]]] (:291), bias-free 1x1 logit conv to 73
# filters (:35-44, :280), reshape to [N, 73, 64] (NCHW, :54-57) and gather_nd at
# (board, move_filter, move_from_square) (:68-73) with board = np.repeat(arange, moves_per_board)
# (:59-61).  PARITY UNPINNED for scores, as for the value network.
POLICY_MODULES = [[[[512, 1], [128, 1]] + 6 * [[32, 3]]]]


def policy_logits_torch(planes, weights, double=False):
    """All 73 x 64 logits per board, [N, 73, 64] ("policy_network/Reshape", chestimator.py:59)."""
    import torch
    import torch.nn.functional as F

    convs, bns = weights["convs"], weights["bn"]
    dt = torch.float64 if double else torch.float32

    def t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dt)

    def conv(x, w_hwio, dil=(1, 1)):
        w = t(np.transpose(w_hwio, (3, 2, 0, 1)))
        k = w_hwio.shape[0]
        pt, pb = _same_pads(k, dil[0])
        pl, pr = _same_pads(k, dil[1])
        return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w, dilation=dil)

    def bn_relu(x, bn):
        mean, var, beta = (t(a).view(1, -1, 1, 1) for a in bn)
        return torch.relu((x - mean) / torch.sqrt(var + BN_EPS) + beta)

    with torch.no_grad():
        x0 = t(planes)
        x = bn_relu(torch.cat([conv(x0, convs[i], dil) for i, (_, dil) in enumerate(INPUT_CONV_GEOMETRY)], dim=1), bns[0])
        for j in range(8):
            x = bn_relu(conv(x, convs[9 + j]), bns[1 + j])
        logits = conv(x, convs[17])
    n = logits.shape[0]
    return logits.reshape(n, 73, 64).numpy(), x.numpy()


def policy_gather(logits, from_squares, move_filters, moves_per_board):
    """requested_move_scores: one score per move (move_evaluation_ann.py:59-73)."""
    board = np.repeat(np.arange(len(moves_per_board)), np.asarray(moves_per_board, np.int64))
    return logits[board, np.asarray(move_filters, np.int64), np.asarray(from_squares, np.int64)]
