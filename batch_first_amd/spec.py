"""Architecture of the Batch-First value network, as data.

The numbers are the ones the reference hard-codes:

* nine bias-free input convolutions on the 15-plane one-hot board
  (reference: batch_first/anns/ann_creation_helper.py:192-252, filter counts from
  batch_first/anns/evaluation_ann.py:22),
* one tower path ``[[512,1],[128,1]] + 6*[[32,3]] + [(16,8)]``
  (reference: batch_first/anns/evaluation_ann.py:268, grammar at
  batch_first/anns/ann_creation_helper.py:263-283: list => SAME, tuple => VALID),
* a bias-free 1x1 logit conv 16 -> 1 and a squeeze
  (reference: batch_first/anns/evaluation_ann.py:35-46).

Every conv is followed by inference batch-norm with ``scale=False`` (eps 1e-3, the
tf.layers default) and ReLU, except the logit layer which has neither.
"""

NUM_PLANES = 15
BOARD = 8
SQUARES = 64
BN_EPS = 1e-3

# (kernel size, (dilation rows, dilation cols), out channels) in concat order.
INPUT_CONVS = (
    (3, (1, 1), 64),
    (3, (2, 2), 32),
    (3, (3, 3), 20),
    (3, (4, 4), 20),
    (3, (5, 5), 18),
    (3, (6, 6), 18),
    (3, (7, 7), 16),
    (2, (2, 4), 24),
    (2, (4, 2), 24),
)
INPUT_CHANNELS = sum(c for _, _, c in INPUT_CONVS)  # 236

# (out channels, kernel size, padding) for the tower, in order.
TOWER = (
    (512, 1, "same"),
    (128, 1, "same"),
    (32, 3, "same"),
    (32, 3, "same"),
    (32, 3, "same"),
    (32, 3, "same"),
    (32, 3, "same"),
    (32, 3, "same"),
    (16, 8, "valid"),
)
LOGIT_IN = 16


def same_pad_before(k, d):
    """TF SAME padding for a dilated stride-1 conv pads (k-1)*d in total, floor(/2) first."""
    return ((k - 1) * d) // 2


def tap_offsets(k, dil):
    """(row offset, col offset) of every kernel tap, in HW (row-major) kernel order."""
    dh, dw = dil
    ph, pw = same_pad_before(k, dh), same_pad_before(k, dw)
    return [(i * dh - ph, j * dw - pw) for i in range(k) for j in range(k)]


def layer_shapes():
    """HWIO kernel shapes of every conv in evaluation order: 9 input convs, 9 tower convs, logit."""
    shapes = [(k, k, NUM_PLANES, c) for k, _, c in INPUT_CONVS]
    cin = INPUT_CHANNELS
    for cout, k, _ in TOWER:
        shapes.append((k, k, cin, cout))
        cin = cout
    shapes.append((1, 1, LOGIT_IN, 1))
    return shapes


def bn_channels():
    """Channel count of each of the 10 batch-norm layers (input module, then 9 tower layers)."""
    return [INPUT_CHANNELS] + [c for c, _, _ in TOWER]


def macs_per_board(in_bounds=True):
    """Dense multiply-accumulates per board. ``in_bounds`` drops taps that fall on zero padding."""
    total = 0
    for k, dil, c in INPUT_CONVS:
        for dr, df in tap_offsets(k, dil):
            if in_bounds:
                n = sum(1 for r in range(8) for f in range(8) if 0 <= r + dr < 8 and 0 <= f + df < 8)
            else:
                n = 64
            total += n * NUM_PLANES * c
    cin = INPUT_CHANNELS
    for cout, k, pad in TOWER:
        if pad == "valid":
            total += k * k * cin * cout
        else:
            for dr, df in tap_offsets(k, (1, 1)):
                if in_bounds:
                    n = sum(1 for r in range(8) for f in range(8) if 0 <= r + dr < 8 and 0 <= f + df < 8)
                else:
                    n = 64
                total += n * cin * cout
        cin = cout
    total += LOGIT_IN
    return total


FLOPS_PER_BOARD = 2 * macs_per_board(True)  # 34 976 560, the figure SURVEY.md 8(d) fixes


# ---- the move-scoring ("policy") network --------------------------------------------------
# batch_first/anns/move_evaluation_ann.py: same input convolutions (:18-23), tower without the
# 8x8 layer (CONVOLUTIONAL_MODULES, :291), 73 logit filters (:280).
POLICY_TOWER = TOWER[:8]
POLICY_LOGIT_IN = 32
POLICY_FILTERS = 73


def policy_layer_shapes():
    """HWIO kernel shapes: 9 input convs, 8 tower convs, logit."""
    shapes = [(k, k, NUM_PLANES, c) for k, _, c in INPUT_CONVS]
    cin = INPUT_CHANNELS
    for cout, k, _ in POLICY_TOWER:
        shapes.append((k, k, cin, cout))
        cin = cout
    shapes.append((1, 1, POLICY_LOGIT_IN, POLICY_FILTERS))
    return shapes


def policy_bn_channels():
    return [INPUT_CHANNELS] + [c for c, _, _ in POLICY_TOWER]
