"""Lane-level NumPy model of k_eval_bf16 (batch_first_amd/csrc/eval_bf16.hip).

Walks the bf16 weight image (bf_pack_weights_bf16) exactly as one workgroup does -- same group
order, same lane maps for v_mfma_f32_32x32x16_bf16 and v_mfma_f32_16x16x32_bf16, same packing of
accumulator registers into the next layer's B operand -- so the fragment arrangement can be
checked against oracle.network.forward_bf16_model on a CPU.  Test infrastructure only.
"""
import numpy as np

from kernel_sim import HL, LANE, LN, bf16_bits_to_float, mfma_bf16, rowmap

INPUT_CONVS = [(3, 1, 1, 64, 0, 2), (3, 2, 2, 32, 64, 1), (3, 3, 3, 20, 96, 1), (3, 4, 4, 20, 116, 1),
               (3, 5, 5, 18, 136, 1), (3, 6, 6, 18, 154, 1), (3, 7, 7, 16, 172, 1), (2, 2, 4, 24, 188, 1),
               (2, 4, 2, 24, 212, 1)]
X1_REG_BASE = [0, 32, 48, 58, 68, 78, 88, 96, 108]
X1_REG_COUNT = [32, 16, 10, 10, 10, 10, 8, 12, 12]
STREAM_GROUPS, STREAM_PAD = 626, 14
B_X3, B_X4, B_X10, B_LOGIT, BIAS_FLOATS = 0, 128, 320, 336, 352
OFF_BIAS = (STREAM_GROUPS + STREAM_PAD) * 1024
OFF_HEAD = OFF_BIAS + ((BIAS_FLOATS * 4 + 255) & ~255)


def _gemm_order():
    """bf_layout_bf16.h:kGemmOrder -- (kind, hidden tile, index) of the 16 x 24 groups of the 1x1 pair."""
    seq = [0, 1, 2, 3, 100, 4, 101, 5, 102, 6, 103, 7, 8, 104, 9, 105, 10, 106, 11, 107, 12, 13, 14]
    order = [(0, 0, s) for s in range(15)]
    for u in range(15):
        order += [(0, u + 1, e) if e < 100 else (1, u, e - 100) for e in seq]
    order += [(1, 15, i) for i in range(8)]
    order += [(2, 0, 0)] * (16 * 24 - len(order))
    assert len(order) == 16 * 24
    return order


GEMM_ORDER = _gemm_order()


def bf16r(x):
    """float -> nearest bf16 (ties to even) as float64."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    u = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return u.view(np.float32).astype(np.float64)


def pack_relu(a):
    """ReLU then bf16 of accumulator values kept in float32 on the GPU."""
    return bf16r(np.maximum(a, 0).astype(np.float32))


class Stream16:
    def __init__(self, image):
        self.groups = image[:(STREAM_GROUPS + STREAM_PAD) * 1024].view(np.uint16).reshape(-1, 64, 8)
        self.g = 0

    def take(self):
        a = bf16_bits_to_float(self.groups[self.g])
        self.g += 1
        return a


def b_frag(dwords4):
    """[4][64] packed values given as element arrays: returns [64][8] with element j = dword j>>1, half j&1."""
    return dwords4


def eval_boards(image, mailboxes):
    """image: uint8 bf16 weight image; mailboxes: uint8[B][64] codes (15 = empty), B boards that
    share one workgroup (the head runs across them).  Returns (scores[B], [intermediates per board])."""
    image = np.asarray(image, np.uint8)
    bias = image[OFF_BIAS:OFF_BIAS + BIAS_FLOATS * 4].view(np.float32).astype(np.float64)
    head = bf16_bits_to_float(image[OFF_HEAD:OFF_HEAD + 64 * 64 * 16].view(np.uint16).reshape(64, 64, 8))
    nb = len(mailboxes)
    inters, x9_planes = [], []
    for mailbox64 in mailboxes:
        inter = {}
        # one-hot table [22][22][16]: plane 15 = 1.0 everywhere
        oh = np.zeros((22, 22, 16))
        oh[:, :, 15] = 1.0
        for s in range(64):
            c = int(mailbox64[s])
            if c != 15:
                oh[(s >> 3) + 7, (s & 7) + 7, c] = 1.0
        streams = [Stream16(image), Stream16(image)]
        x1 = np.zeros((236, 64))
        acc3_h = []
        for half in range(2):
            ws = streams[half]
            sq = 32 * half + LN
            rank, file = sq >> 3, sq & 7
            P = np.zeros((120, 64))  # x1 registers (bf16 values), register r of lane l
            for ci, (k, dh, dw, ch, base, tiles) in enumerate(INPUT_CONVS):
                ph, pw = ((k - 1) * dh) // 2, ((k - 1) * dw) // 2
                acc = np.zeros((tiles, 16, 64))
                for tap in range(k * k):
                    dr, df = (tap // k) * dh - ph, (tap % k) * dw - pw
                    entry = oh[rank + 7 + dr, file + 7 + df]          # [64][16]
                    b8 = np.where(HL[:, None] == 1, entry[:, 8:], entry[:, :8])
                    for tile in range(tiles):
                        mfma_bf16(ws.take(), b8, acc[tile])
                for lr in range(X1_REG_COUNT[ci]):
                    v = pack_relu(acc[lr // 16, lr % 16])
                    P[X1_REG_BASE[ci] + lr] = v
                    c_local = 2 * lr + HL
                    ok = c_local < ch
                    x1[(base + c_local)[ok], sq[ok]] = v[ok]
            assert np.all(P[68 + 9] == 1.0) and np.all(P[78 + 9] == 1.0), "ones registers"
            acc3 = np.zeros((4, 16, 64))
            for c3 in range(4):
                for t in range(16):
                    acc3[c3, t] = bias[B_X3 + 32 * c3 + rowmap(t, HL)]
            acc2, H = {}, {}
            for kind, tile, idx in GEMM_ORDER:      # the stream order of the 1x1 pair (kGemmOrder)
                w = ws.take()
                if kind == 0:
                    if idx == 0:
                        acc2[tile] = np.zeros((16, 64))
                    mfma_bf16(w, P[8 * idx:8 * idx + 8].T, acc2[tile])
                elif kind == 1:
                    if tile not in H:
                        H[tile] = pack_relu(acc2[tile])
                    s, c3 = idx >> 2, idx & 3
                    mfma_bf16(w, H[tile][8 * s:8 * s + 8].T, acc3[c3])
            acc3_h.append(acc3)
        inter["x1"] = x1
        x3 = np.zeros((128, 64))
        for half in range(2):
            for c3 in range(4):
                for t in range(16):
                    x3[32 * c3 + rowmap(t, HL), 32 * half + LN] = pack_relu(acc3_h[half][c3, t])
        inter["x3"] = x3

        def conv_chunk(ws, plane, acc, half):
            sq = 32 * half + LN
            rank, file = sq >> 3, sq & 7
            for tap in range(9):
                dr, df = tap // 3 - 1, tap % 3 - 1
                for s in range(2):
                    # lane (sq, h) element j = channel 16s + 8h + j at the shifted position
                    ch = 16 * s + 8 * HL[:, None] + np.arange(8)[None, :]
                    b8 = plane[ch, (rank + 1 + dr)[:, None], (file + 1 + df)[:, None]]
                    mfma_bf16(ws.take(), b8, acc)

        def to_plane(x32):
            p = np.zeros((32, 10, 10))
            p[:, 1:9, 1:9] = x32.reshape(32, 8, 8)
            return p

        accs = [np.zeros((16, 64)), np.zeros((16, 64))]
        for half in range(2):
            for t in range(16):
                accs[half][t] = bias[B_X4 + rowmap(t, HL)]
        for q in range(4):
            plane = to_plane(x3[32 * q:32 * q + 32])
            for half in range(2):
                conv_chunk(streams[half], plane, accs[half], half)
        for layer in range(5):
            x = np.zeros((32, 64))
            for half in range(2):
                for t in range(16):
                    x[rowmap(t, HL), 32 * half + LN] = pack_relu(accs[half][t])
            inter["x%d" % (4 + layer)] = x
            plane = to_plane(x)
            for half in range(2):
                for t in range(16):
                    accs[half][t] = bias[B_X4 + 32 * (layer + 1) + rowmap(t, HL)]
                conv_chunk(streams[half], plane, accs[half], half)
        x9 = np.zeros((32, 64))
        for half in range(2):
            for t in range(16):
                x9[rowmap(t, HL), 32 * half + LN] = pack_relu(accs[half][t])
        inter["x9"] = x9
        assert streams[0].g == STREAM_GROUPS and streams[1].g == STREAM_GROUPS
        inters.append(inter)
        x9_planes.append(x9)

    # head: v_mfma_f32_16x16x32_bf16, A[row = l&15][k = 8(l>>4)+j], B[k][col = l&15], D: col = l&15,
    # row = 4(l>>4) + reg; one k-step per square, columns = boards
    col, kc = LANE & 15, LANE >> 4
    D = np.zeros((16, 16))
    for sqk in range(64):
        A = np.zeros((16, 32))
        B = np.zeros((32, 16))
        for l in range(64):
            A[l & 15, 8 * (l >> 4):8 * (l >> 4) + 8] = head[sqk, l]
            bb = col[l] if col[l] < nb else 0
            B[8 * kc[l]:8 * kc[l] + 8, col[l]] = x9_planes[bb][8 * kc[l]:8 * kc[l] + 8, sqk]
        D += A @ B
    scores = np.zeros(nb)
    for b in range(nb):
        y = np.maximum(D[:, b] + bias[B_X10:B_X10 + 16], 0)
        inters[b]["x10"] = y
        scores[b] = float((y * bias[B_LOGIT:B_LOGIT + 16]).sum())
    return scores, inters
