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
