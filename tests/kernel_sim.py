"""Lane-level NumPy model of k_eval_f32 (batch_first_amd/csrc/eval_f32.hip).

It walks the packed weight image (bf_pack_weights / bf_pack_weights_ex) exactly as one workgroup does -- same
k-step order, same register/lane maps for v_mfma_f32_32x32x2_f32 and v_mfma_f32_32x32x16_bf16 -- so the fragment
arrangement and the chaining of accumulator tiles can be checked against the oracle on a CPU,
without a GPU.  split = 0: float32; 1: BF_PRECISION_F32_SPLIT (two bf16 terms a side in the 1x1 pair, three in the
3x3 layers); 2: BF_PRECISION_F32_SPLIT3 (three everywhere).  Sums are float64 here, so what is modelled of the split
precisions is the operand terms and which of their products are taken, not the kernel's float32 rounding.
Test infrastructure only.
"""
import numpy as np

INPUT_CONVS = [(3, 1, 1, 64, 0, 2), (3, 2, 2, 32, 64, 1), (3, 3, 3, 20, 96, 1), (3, 4, 4, 20, 116, 1),
               (3, 5, 5, 18, 136, 1), (3, 6, 6, 18, 154, 1), (3, 7, 7, 16, 172, 1), (2, 2, 4, 24, 188, 1),
               (2, 4, 2, 24, 212, 1)]
STREAM_GROUPS = {0: 1300, 1: 1462, 2: 1830}   # bf_layout.h: kStreamGroups, kStreamGroupsSplit, kStreamGroupsSplit3
STREAM_REGION = 1830 + 32   # groups the image reserves: the longest stream (bf_layout.h: kStreamGroupsSplit3) + padding
BIAS_FLOATS = 1104
B_X1, B_X2, B_X3, B_X4, B_X10, B_LOGIT = 0, 236, 748, 876, 1068, 1084

LANE = np.arange(64)
HL = LANE >> 5
LN = LANE & 31


def rowmap(t, hl):
    return (t & 3) + 8 * (t >> 2) + 4 * hl


def mfma_bf16(a8, b8, acc):
    """acc[16][64] += A(32x16) @ B(16x32), v_mfma_f32_32x32x16_bf16 lane maps:
    lane (r, h) holds A[row r][k = 8h + j] and B[k = 8h + j][col r], j = 0..7."""
    A = np.zeros((32, 16))
    B = np.zeros((16, 32))
    for h in range(2):
        A[:, 8 * h:8 * h + 8] = a8[32 * h:32 * h + 32]
        B[8 * h:8 * h + 8, :] = b8[32 * h:32 * h + 32].T
    D = A @ B
    for t in range(16):
        acc[t] += D[rowmap(t, HL), LN]


def bf16_bits_to_float(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def mfma(a, b, acc):
    """acc[16][64] += A(32x2) @ B(2x32) with the gfx950 32x32x2 f32 lane maps."""
    A = a.reshape(2, 32).T.astype(np.float64)   # A[i][k] = a[lane: ln=i, hl=k]
    B = b.reshape(2, 32).astype(np.float64)     # B[k][j] = b[lane: hl=k, ln=j]
    D = A @ B
    for t in range(16):
        acc[t] += D[rowmap(t, HL), LN]


def bf16_terms(x, n):
    """x (float64 holding float32 values) as n bf16 terms, each round-to-nearest-even of what is left (split2 / split3
    in the kernel, split_bf16x3 in pack.cpp).  Returns n float64 arrays."""
    r = np.asarray(x, np.float64).astype(np.float32)
    out = []
    for _ in range(n):
        u = r.view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
        t = (u.astype(np.uint32) << 16).view(np.float32)
        out.append(t.astype(np.float64))
        r = (r - t).astype(np.float32)
    return out


def split_products(w_terms, x_terms, acc, fragment):
    """acc += sum of the products the kernel takes: all of them for two terms a side, those with i + j <= 2 for
    three.  fragment(term) -> b8[64 lanes][8]."""
    for i, w in enumerate(w_terms):
        for j in range(len(x_terms)):
            if len(w_terms) == 2 or i + j <= 2:
                mfma_bf16(w, fragment(x_terms[j]), acc)


class Stream:
    def __init__(self, packed):
        self.data = packed[:STREAM_REGION * 256].reshape(-1, 64, 4)
        self.kstep = 0

    def next_group_bf16(self):
        """One whole group as 8 bf16 per lane (stage 1)."""
        assert self.kstep % 4 == 0
        g = self.kstep // 4
        self.kstep += 4
        raw = np.ascontiguousarray(self.data[g]).view(np.uint16).reshape(64, 8)
        return bf16_bits_to_float(raw)

    def next(self):
        g, j = divmod(self.kstep, 4)
        self.kstep += 1
        return self.data[g, :, j].astype(np.float64)

    def skip(self, n):
        self.kstep += n


def eval_board(packed, mailbox64, split=0):
    """mailbox64: uint8[64] codes (15 = empty) in the mover's frame.  Returns (score, intermediates)."""
    packed = np.asarray(packed, np.float32)
    bias = packed[STREAM_REGION * 256:][:BIAS_FLOATS].astype(np.float64)
    head = packed[STREAM_REGION * 256 + BIAS_FLOATS:].reshape(2, 64, 64, 4).astype(np.float64)
    mb = np.full((22, 22), 255, np.int64)
    mb[7:15, 7:15] = np.where(mailbox64 == 15, 255, mailbox64).reshape(8, 8)

    inter = {}
    x1_halves, x3_halves = [], []
    streams = [Stream(packed), Stream(packed)]
    # ---- stage 1 + 2 per half (waves are independent here)
    acc3_h = []
    for half in range(2):
        ws = streams[half]
        sq = 32 * half + LN
        rank, file = sq >> 3, sq & 7
        x1 = np.zeros((118, 64))
        for (k, dh, dw, ch, base, tiles) in INPUT_CONVS:
            ph, pw = ((k - 1) * dh) // 2, ((k - 1) * dw) // 2
            acc = np.zeros((tiles, 16, 64))
            for tile in range(tiles):
                for t in range(16):
                    c = 32 * tile + 2 * t
                    if c < ch:
                        acc[tile, t] = np.where(HL == 1, bias[B_X1 + base + c + 1], bias[B_X1 + base + c])
            for tap in range(k * k):
                dr, df = (tap // k) * dh - ph, (tap % k) * dw - pw
                cc = mb[rank + 7 + dr, file + 7 + df] - 8 * HL
                b8 = (cc[:, None] == np.arange(8)[None, :]).astype(np.float64)
                for tile in range(tiles):
                    for _split in range(3):
                        mfma_bf16(ws.next_group_bf16(), b8, acc[tile])
            for tile in range(tiles):
                for t in range(16):
                    if 32 * tile + 2 * t < ch:
                        x1[base // 2 + 16 * tile + t] = np.maximum(acc[tile, t], 0)
        x1_halves.append(x1)
        acc3 = np.zeros((4, 16, 64))
        for c3 in range(4):
            for t in range(16):
                acc3[c3, t] = bias[B_X3 + 32 * c3 + rowmap(t, HL)]
        pair_terms = {0: 0, 1: 2, 2: 3}[split]
        if split:
            xp = np.zeros((120, 64))
            xp[:118] = x1
            xp[118] = np.where(HL == 0, 1.0, 0.0)        # the all-ones slot that carries the hidden layer's shift
            xt = bf16_terms(xp, pair_terms)
        for c2 in range(16 if split else 0):
            acc2 = np.zeros((16, 64))
            for s in range(15):          # k-step s: registers 8s .. 8s+7 of the lane
                wt = [ws.next_group_bf16() for _ in range(pair_terms)]
                split_products(wt, xt, acc2, lambda x: x[8 * s:8 * s + 8].T)
            ht = bf16_terms(np.maximum(acc2, 0), pair_terms)
            for c3 in range(4):
                for v in range(2):
                    wt = [ws.next_group_bf16() for _ in range(pair_terms)]
                    split_products(wt, ht, acc3[c3], lambda x: x[8 * v:8 * v + 8].T)
        for c2 in range(0 if split else 16):
            acc2 = np.zeros((16, 64))
            for r in range(118):
                mfma(ws.next(), x1[r], acc2)
            mfma(ws.next(), np.ones(64), acc2)  # k-step 118: the batch-norm shift, B = 1
            ws.skip(1)
            acc2 = np.maximum(acc2, 0)
            for c3 in range(4):
                for t in range(16):
                    mfma(ws.next(), acc2[t], acc3[c3])
        acc3_h.append(acc3)

    def gather(regs_by_half, nch, chan_of):
        out = np.zeros((nch, 64))
        for half in range(2):
            regs = regs_by_half[half]
            for r in range(regs.shape[0]):
                out[chan_of(r, HL), 32 * half + LN] = regs[r]
        return out

    inter["x1"] = gather(x1_halves, 236, lambda r, hl: 2 * r + hl)
    x3 = np.zeros((128, 64))
    for half in range(2):
        for c3 in range(4):
            for t in range(16):
                x3[32 * c3 + rowmap(t, HL), 32 * half + LN] = np.maximum(acc3_h[half][c3, t], 0)
    inter["x3"] = x3

    # ---- 3x3 layers: LDS plane shared by both halves
    def conv_chunk(ws, xplane, acc, half):
        sq = 32 * half + LN
        rank, file = sq >> 3, sq & 7
        if split:   # three terms a side; k-step s, element (h, j) = channel rowmap(8 h + j, s)
            xt = bf16_terms(xplane, 3)
            for tap in range(9):
                dr, df = tap // 3 - 1, tap % 3 - 1
                for s in range(2):
                    wt = [ws.next_group_bf16() for _ in range(3)]
                    chan = rowmap(8 * HL[:, None] + np.arange(8)[None, :], s)      # [64 lanes][8]
                    split_products(wt, xt, acc, lambda x: x[chan, (rank + 1 + dr)[:, None], (file + 1 + df)[:, None]])
            return
        for tap in range(9):
            dr, df = tap // 3 - 1, tap % 3 - 1
            for s in range(16):
                b = xplane[2 * s + HL, rank + 1 + dr, file + 1 + df]
                mfma(ws.next(), b, acc)

    def to_plane(x32):  # [32][64] -> zero-haloed [32][10][10]
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
                x[rowmap(t, HL), 32 * half + LN] = np.maximum(accs[half][t], 0)
        inter["x%d" % (4 + layer)] = x
        plane = to_plane(x)
        for half in range(2):
            for t in range(16):
                accs[half][t] = bias[B_X4 + 32 * (layer + 1) + rowmap(t, HL)]
            conv_chunk(streams[half], plane, accs[half], half)
    x9 = np.zeros((32, 64))
    part = np.zeros((128, 16))
    for half in range(2):
        for t in range(16):
            v = np.maximum(accs[half][t], 0)
            x9[rowmap(t, HL), 32 * half + LN] = v
            for cq in range(4):
                part[64 * half + LANE, 4 * cq:4 * cq + 4] += v[:, None] * head[half, t * 4 + cq]
    inter["x9"] = x9
    s = part.sum(axis=0)
    y = np.maximum(s + bias[B_X10:B_X10 + 16], 0)
    inter["x10"] = y
    assert streams[0].kstep == streams[1].kstep == STREAM_GROUPS[split] * 4, (streams[0].kstep, streams[1].kstep)
    return float((y * bias[B_LOGIT:B_LOGIT + 16]).sum()), inter
