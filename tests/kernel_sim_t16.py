"""Lane-level NumPy model of k_eval_t16 (batch_first_amd/csrc/eval_t16.hip).

Walks the 16x16x32 weight image (bf_pack_weights_t16) exactly as one wave does -- same group order
(stage 1 steps, kPairOrder, 3x3 layers), same lane maps of v_mfma_f32_16x16x32_bf16, same packing of
accumulator registers into the next layer's B operand, same activation-plane channel order -- so the
fragment arrangement can be checked against oracle.network.forward_bf16_model on a CPU.
Test infrastructure only.  Constants restate batch_first_amd/csrc/bf_layout_t16.h.
"""
import numpy as np

from kernel_sim import bf16_bits_to_float


def bf16r(x):
    """float -> nearest bf16 (ties to even) as float64."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32).astype(np.float64)


def pack_relu(a):
    """ReLU then bf16 of accumulator values kept in float32 on the GPU."""
    return bf16r(np.maximum(a, 0).astype(np.float32))

INPUT_CONVS = [(3, 1, 1, 64, 0), (3, 2, 2, 32, 64), (3, 3, 3, 20, 96), (3, 4, 4, 20, 116), (3, 5, 5, 18, 136),
               (3, 6, 6, 18, 154), (3, 7, 7, 16, 172), (2, 2, 4, 24, 188), (2, 4, 2, 24, 212)]
ROW_TILES = [4, 2, 2, 2, 2, 2, 1, 2, 2]
X1_REG_COUNT = [16, 8, 6, 6, 6, 6, 4, 6, 6]
X1_REG_BASE = [0, 16, 24, 30, 36, 42, 48, 52, 58]
STEPS3 = [(0, 1), (6, 7), (2, 8), (3, 4), (5, -1)]
STEPS2 = [(0, 1), (2, 3)]
STREAM_GROUPS, WALK_GROUPS = 629, 640
B_X3, B_X4, B_X10, B_LOGIT, BIAS_FLOATS = 0, 128, 320, 336, 352
OFF_BIAS = WALK_GROUPS * 1024
OFF_HEAD = OFF_BIAS + ((BIAS_FLOATS * 4 + 255) & ~255)

LANE = np.arange(64)
C16 = LANE & 15
Q = LANE >> 4


def pair_order():
    order = [(0, tile, s) for tile in range(2) for s in range(8)]
    for h in range(15):
        order += [(0, 2 * h + 2, s) for s in range(8)]
        for s in range(8):
            order += [(0, 2 * h + 3, s), (1, h, s)]
    order += [(1, 15, s) for s in range(8)]
    assert len(order) == 384
    return order


def mfma16(a8, b8, acc):
    """acc[4][64] += A(16x32) @ B(32x16): lane (c, q) holds A[row c][k = 8q + j], B[k = 8q + j][col c];
    D[row = 4q + r][col = c] in register r."""
    A = np.zeros((16, 32))
    B = np.zeros((32, 16))
    for q in range(4):
        A[:, 8 * q:8 * q + 8] = a8[16 * q:16 * q + 16]
        B[8 * q:8 * q + 8, :] = b8[16 * q:16 * q + 16].T
    D = A @ B
    for r in range(4):
        acc[r] += D[4 * Q + r, C16]


def tap_offsets(k, dh, dw, tap):
    return (tap // k) * dh - ((k - 1) * dh) // 2, (tap % k) * dw - ((k - 1) * dw) // 2


def tap_reaches(dr, t):
    return 0 <= 2 * t + dr <= 7 or 0 <= 2 * t + 1 + dr <= 7


class Stream:
    def __init__(self, image):
        self.groups = image[:WALK_GROUPS * 1024].view(np.uint16).reshape(-1, 64, 8)
        self.g = 0

    def take(self):
        a = bf16_bits_to_float(self.groups[self.g])
        self.g += 1
        return a


def eval_boards(image, mailboxes):
    """image: uint8 weight image; mailboxes: uint8[B][64] codes (15 = empty) of the boards of one
    workgroup.  Returns (scores[B], [intermediates per board]); also counts the MFMAs issued."""
    image = np.asarray(image, np.uint8)
    bias = image[OFF_BIAS:OFF_BIAS + BIAS_FLOATS * 4].view(np.float32).astype(np.float64)
    head = bf16_bits_to_float(image[OFF_HEAD:OFF_HEAD + 64 * 64 * 16].view(np.uint16).reshape(64, 64, 8))
    nb = len(mailboxes)
    inters, x9_planes, mfmas = [], [], 0
    order = pair_order()
    for mailbox64 in mailboxes:
        inter = {}
        oh = np.zeros((22, 22, 16))
        oh[:, :, 15] = 1.0
        for s in range(64):
            c = int(mailbox64[s])
            if c != 15:
                oh[(s >> 3) + 7, (s & 7) + 7, c] = 1.0
        ws = Stream(image)
        rank = lambda t: 2 * t + (C16 >> 3)
        file = C16 & 7
        # ---- stage 1 ----
        X = np.zeros((4, 64, 64))      # x1 registers per column tile (bf16 values)
        x1 = np.zeros((236, 64))
        for ci, (k, dh, dw, ch, base) in enumerate(INPUT_CONVS):
            steps = STEPS3 if k == 3 else STEPS2
            acc = np.zeros((ROW_TILES[ci], 4, 4, 64))   # [row tile][column tile][r][lane]
            for (ta, tb) in steps:
                frags = [ws.take() for _ in range(ROW_TILES[ci])]
                for t in range(4):
                    offs = [tap_offsets(k, dh, dw, tap) if tap >= 0 else None for tap in (ta, tb)]
                    if k == 3 and not any(o is not None and tap_reaches(o[0], t) for o in offs):
                        continue
                    b8 = np.zeros((64, 8))
                    for l in range(64):
                        o = offs[Q[l] >> 1] or offs[0]
                        entry = oh[rank(t)[l] + 7 + o[0], file[l] + 7 + o[1]]
                        b8[l] = entry[8 * (Q[l] & 1):8 * (Q[l] & 1) + 8]
                    for tile in range(ROW_TILES[ci]):
                        mfma16(frags[tile], b8, acc[tile, t])
                        mfmas += 1
            for t in range(4):
                for lr in range(X1_REG_COUNT[ci]):
                    tile, r = lr // 4, lr % 4
                    v = pack_relu(acc[tile, t, r])
                    X[t, X1_REG_BASE[ci] + lr] = v
                    local = 16 * tile + 4 * r + Q
                    ok = local < ch
                    x1[(base + local)[ok], (16 * t + C16)[ok]] = v[ok]
        inter["x1"] = x1
        # ---- the 1x1 pair ----
        acc3 = np.zeros((8, 4, 4, 64))
        for o in range(8):
            for r in range(4):
                acc3[o, :, r] = bias[B_X3 + 16 * o + 4 * Q + r][None, :]
        acc2, H = {}, {}
        for kind, a, b in order:
            w = ws.take()
            if kind == 0:
                if b == 0:
                    acc2[a] = np.zeros((4, 4, 64))
                for t in range(4):
                    regs = [X[t, 8 * b + j] for j in range(8)]
                    mfma16(w, np.stack(regs, axis=1), acc2[a][t])
                    mfmas += 1
            else:
                if a not in H:
                    H[a] = [np.stack([pack_relu(acc2[2 * a + (j >> 2)][t][j & 3]) for j in range(8)], axis=1) for t in range(4)]
                for t in range(4):
                    mfma16(w, H[a][t], acc3[b, t])
                    mfmas += 1
        x3 = np.zeros((128, 64))
        for o in range(8):
            for t in range(4):
                for r in range(4):
                    x3[16 * o + 4 * r + Q, 16 * t + C16] = pack_relu(acc3[o, t, r])
        inter["x3"] = x3

        # ---- 3x3 layers through activation planes [pos][k], k-slot (q, j) = channel 16(j>>2) + 4(j&3) + q ----
        def to_plane(tiles2):     # tiles2[row tile][t][r][lane] (float32 accumulators) -> [10][10][32] in plane order
            p = np.zeros((10, 10, 32))
            for t in range(4):
                for j in range(8):
                    p[rank(t) + 1, file + 1, 8 * Q + j] = pack_relu(tiles2[j >> 2][t][j & 3])
            return p

        def conv_chunk(plane, acc):   # acc[row tile][t][r][lane]
            nonlocal mfmas
            for tap in range(9):
                dr, df = tap // 3 - 1, tap % 3 - 1
                frags = [ws.take(), ws.take()]
                for t in range(4):
                    b8 = np.zeros((64, 8))
                    for l in range(64):
                        b8[l] = plane[rank(t)[l] + 1 + dr, file[l] + 1 + df, 8 * Q[l]:8 * Q[l] + 8]
                    for tile in range(2):
                        mfma16(frags[tile], b8, acc[tile][t])
                        mfmas += 1

        def seeded(layer):
            a = np.zeros((2, 4, 4, 64))
            for tile in range(2):
                for r in range(4):
                    a[tile, :, r] = bias[B_X4 + 32 * layer + 16 * tile + 4 * Q + r][None, :]
            return a

        acc = seeded(0)
        for chunk in range(4):
            conv_chunk(to_plane([acc3[2 * chunk], acc3[2 * chunk + 1]]), acc)
        for layer in range(5):
            x = np.zeros((32, 64))
            for tile in range(2):
                for t in range(4):
                    for r in range(4):
                        x[16 * tile + 4 * r + Q, 16 * t + C16] = pack_relu(acc[tile, t, r])
            inter["x%d" % (4 + layer)] = x
            plane = to_plane(acc)
            acc = seeded(layer + 1)
            conv_chunk(plane, acc)
        x9 = np.zeros((32, 64))
        for tile in range(2):
            for t in range(4):
                for r in range(4):
                    x9[16 * tile + 4 * r + Q, 16 * t + C16] = pack_relu(acc[tile, t, r])
        inter["x9"] = x9
        assert ws.g == STREAM_GROUPS
        inters.append(inter)
        x9_planes.append(to_plane(acc))

    # head: out[board][co] = sum over squares of x9[board][sq][k] * W[sq][k][co], k in plane order
    D = np.zeros((nb, 16))
    for sqk in range(64):
        Wk = np.zeros((32, 16))
        for l in range(64):
            Wk[8 * (l >> 4):8 * (l >> 4) + 8, l & 15] = head[sqk, l]
        for b in range(nb):
            D[b] += x9_planes[b][(sqk >> 3) + 1, (sqk & 7) + 1] @ Wk
    scores = np.zeros(nb)
    for b in range(nb):
        y = np.maximum(D[b] + bias[B_X10:B_X10 + 16], 0)
        inters[b]["x10"] = y
        scores[b] = float((y * bias[B_LOGIT:B_LOGIT + 16]).sum())
    return scores, inters, mfmas // nb
