"""Random-init weights for the value network and the canonical flat blob the C-ABI takes.

No trained weights ship with the reference (README.md:89), so parity is checked on
random-init weights drawn the way the reference initialises them:
``tf.contrib.layers.variance_scaling_initializer()`` defaults
(batch_first/anns/evaluation_ann.py:252) = truncated normal (|z| <= 2) with
std ``sqrt(1.3 * 2 / fan_in)``, fan_in = kh*kw*cin.  Batch-norm tensors are either
"fresh" (mean 0, var 1, beta 0 -- what an untrained tf.layers.batch_normalization holds)
or "perturbed" so that a wrong mean/var/beta wiring shows up in the scores.

Canonical blob (float32, little-endian, concatenated in this order; ``include/batchfirst.h``
documents the same layout for C callers):

    for each of the 9 input convs:   kernel HWIO [k][k][15][C_i]
    input batch-norm:                mean[236], var[236], beta[236]
    for each of the 9 tower layers:  kernel HWIO [k][k][Cin][Cout], mean[Cout], var[Cout], beta[Cout]
    logit kernel:                    HWIO [1][1][16][1]
"""
import hashlib

import numpy as np

from . import spec


def _truncated_normal(rng, shape, std):
    n = int(np.prod(shape))
    out = np.empty(n, np.float64)
    filled = 0
    while filled < n:
        z = rng.standard_normal(n - filled + 64)
        z = z[np.abs(z) <= 2.0]
        take = min(len(z), n - filled)
        out[filled:filled + take] = z[:take]
        filled += take
    return (out * std).astype(np.float32).reshape(shape)


def make_weights(seed=1234, bn="perturbed"):
    """Return ``{"convs": [HWIO arrays x19], "bn": [(mean, var, beta) x10]}``, all float32."""
    rng = np.random.Generator(np.random.PCG64(seed))
    convs = []
    for shape in spec.layer_shapes():
        kh, kw, cin, _ = shape
        std = np.sqrt(1.3 * 2.0 / (kh * kw * cin))
        convs.append(_truncated_normal(rng, shape, std))
    bns = []
    for c in spec.bn_channels():
        if bn == "fresh":
            mean = np.zeros(c, np.float32)
            var = np.ones(c, np.float32)
            beta = np.zeros(c, np.float32)
        elif bn == "perturbed":
            mean = (rng.standard_normal(c) * 0.1).astype(np.float32)
            var = rng.uniform(0.5, 1.5, c).astype(np.float32)
            beta = (rng.standard_normal(c) * 0.1).astype(np.float32)
        else:
            raise ValueError("bn must be 'fresh' or 'perturbed'")
        bns.append((mean, var, beta))
    return {"convs": convs, "bn": bns}


def to_blob(weights):
    """Flatten a weights dict into the canonical float32 blob."""
    convs, bns = weights["convs"], weights["bn"]
    parts = [np.ascontiguousarray(w, np.float32).ravel() for w in convs[:9]]
    parts += [np.ascontiguousarray(a, np.float32) for a in bns[0]]
    for j in range(9):
        parts.append(np.ascontiguousarray(convs[9 + j], np.float32).ravel())
        parts += [np.ascontiguousarray(a, np.float32) for a in bns[1 + j]]
    parts.append(np.ascontiguousarray(convs[18], np.float32).ravel())
    return np.concatenate(parts)


def blob_size():
    n = sum(int(np.prod(s)) for s in spec.layer_shapes())
    n += 3 * sum(spec.bn_channels())
    return n


def from_blob(blob):
    """Inverse of :func:`to_blob`."""
    blob = np.asarray(blob, np.float32)
    if blob.size != blob_size():
        raise ValueError("weight blob has %d floats, expected %d" % (blob.size, blob_size()))
    shapes = spec.layer_shapes()
    chans = spec.bn_channels()
    pos = 0

    def take(shape):
        nonlocal pos
        n = int(np.prod(shape))
        a = blob[pos:pos + n].reshape(shape).copy()
        pos += n
        return a

    convs = [None] * 19
    bns = [None] * 10
    for i in range(9):
        convs[i] = take(shapes[i])
    bns[0] = tuple(take((chans[0],)) for _ in range(3))
    for j in range(9):
        convs[9 + j] = take(shapes[9 + j])
        bns[1 + j] = tuple(take((chans[1 + j],)) for _ in range(3))
    convs[18] = take(shapes[18])
    return {"convs": convs, "bn": bns}


def blob_sha256(blob):
    return hashlib.sha256(np.ascontiguousarray(blob, np.float32).tobytes()).hexdigest()


# ---- the move-scoring ("policy") network: same recipe, its own shapes ------------------------

def make_policy_weights(seed=4321, bn="perturbed"):
    """``{"convs": [HWIO x18], "bn": [(mean, var, beta) x9]}`` for the policy network
    (batch_first/anns/move_evaluation_ann.py:18-45, :280, :291)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    convs = []
    for shape in spec.policy_layer_shapes():
        kh, kw, cin, _ = shape
        convs.append(_truncated_normal(rng, shape, np.sqrt(1.3 * 2.0 / (kh * kw * cin))))
    bns = []
    for c in spec.policy_bn_channels():
        if bn == "fresh":
            bns.append((np.zeros(c, np.float32), np.ones(c, np.float32), np.zeros(c, np.float32)))
        elif bn == "perturbed":
            bns.append(((rng.standard_normal(c) * 0.1).astype(np.float32), rng.uniform(0.5, 1.5, c).astype(np.float32),
                        (rng.standard_normal(c) * 0.1).astype(np.float32)))
        else:
            raise ValueError("bn must be 'fresh' or 'perturbed'")
    return {"convs": convs, "bn": bns}


def policy_blob_size():
    return sum(int(np.prod(s)) for s in spec.policy_layer_shapes()) + 3 * sum(spec.policy_bn_channels())


def to_policy_blob(weights):
    """Canonical float32 blob of the policy network (include/batchfirst.h, BF_POLICY_WEIGHT_FLOATS)."""
    convs, bns = weights["convs"], weights["bn"]
    parts = [np.ascontiguousarray(w, np.float32).ravel() for w in convs[:9]]
    parts += [np.ascontiguousarray(a, np.float32) for a in bns[0]]
    for j in range(8):
        parts.append(np.ascontiguousarray(convs[9 + j], np.float32).ravel())
        parts += [np.ascontiguousarray(a, np.float32) for a in bns[1 + j]]
    parts.append(np.ascontiguousarray(convs[17], np.float32).ravel())
    blob = np.concatenate(parts)
    assert blob.size == policy_blob_size()
    return blob
