"""Layer-by-layer comparison of the bf16 kernel with the CPU model (diagnostic)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batch_first_amd import ops, weights as W
from oracle import encoding as E, network as N
g = np.load(os.path.join(ROOT, "tests/golden/positions_1024.npz"))
w = W.make_weights(seed=1234, bn="perturbed")
ctx = ops.DeviceContext(w, device=0, dtype="bf16")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mbc = g["mailbox"][:n]
mb = (mbc[:, 0::2] | (mbc[:, 1::2] << 4)).astype(np.uint8)
scores, dbg = ops.eval_mailbox_debug(ctx.handle, torch.from_numpy(mb).cuda())
dbg = dbg.cpu().numpy()
planes = E.decode_planes_fast(g["piece_filters"], g["occupied_bbs"])[:n]
ref, inter = N.forward_bf16_model(planes, w, return_intermediates=True)
pos = 0
for name, ch in [("x1", 236), ("x3", 128), ("x4", 32), ("x5", 32), ("x6", 32), ("x7", 32), ("x8", 32), ("x9", 32)]:
    got = dbg[:, pos:pos + ch * 64].reshape(n, ch, 8, 8)
    diff = np.abs(got - inter[name])
    bad = diff > 2.0 ** -6 * np.maximum(np.abs(inter[name]), 2.0 ** -6)
    per_board = bad.reshape(n, -1).sum(1)
    print(name, "max diff %.4f" % diff.max(), "bad elems even boards", int(per_board[0::2].sum()), "odd boards", int(per_board[1::2].sum()))
    if bad.any():
        b, c, r, f = np.argwhere(bad)[0]
        print("   first bad: board %d ch %d rank %d file %d got %.5f want %.5f" % (b, c, r, f, got[b, c, r, f], inter[name][b, c, r, f]))
        bb = np.argwhere(bad)
        print("   bad by rank", np.bincount(bb[:, 2], minlength=8), "by file", np.bincount(bb[:, 3], minlength=8), "by ch%8", np.bincount(bb[:, 1] % 8, minlength=8))
    pos += ch * 64
print("x10 max diff", np.abs(dbg[:, pos:pos + 16] - inter["x10"].reshape(n, 16)).max())
print("scores max diff", np.abs(scores.cpu().numpy() - ref).max())
print("---- teacher-forced: each layer from the GPU's own previous layer")
pos = 0
layers = [("x1", 236), ("x3", 128), ("x4", 32), ("x5", 32), ("x6", 32), ("x7", 32), ("x8", 32), ("x9", 32)]
prev = None
for name, ch in layers:
    got = dbg[:, pos:pos + ch * 64].reshape(n, ch, 8, 8)
    pos += ch * 64
    if prev is not None:
        want = N.bf16_model_layer(prev, w, name)
        diff = np.abs(got - want)
        ulp = 2.0 ** -8 * np.maximum(np.abs(want), 2.0 ** -126)
        print(name, "mismatched %.5f%%" % (100 * np.mean(diff > 0)), "max diff %.5f" % diff.max(), "max |x| %.3f" % np.abs(want).max(),
              "max diff/ulp %.2f" % (diff / np.maximum(ulp, 1e-9))[diff > 0].max() if (diff > 0).any() else "exact")
    prev = got
want10 = N.bf16_model_layer(prev, w, "x10").reshape(n, 16)
print("x10 max diff", np.abs(dbg[:, pos:pos + 16] - want10).max())
if n == 1024:
    got = scores.cpu().numpy()
    for nm, r in [("bf16 model", ref), ("fp32 oracle", g["scores_f32_perturbed"])]:
        rms = np.sqrt(np.mean(r.astype(np.float64) ** 2))
        e = np.abs(got - r) / np.maximum(np.abs(r), rms)
        print("scores vs %s: rms %.4f worst %.3e mean %.3e  rms abs err %.3e" % (nm, rms, e.max(), e.mean(), np.sqrt(np.mean((got - r) ** 2))))
    m = ref; r = g["scores_f32_perturbed"]; rms = np.sqrt(np.mean(r.astype(np.float64) ** 2))
    e = np.abs(m - r) / np.maximum(np.abs(r), rms)
    print("bf16 model vs fp32 oracle: worst %.3e mean %.3e" % (e.max(), e.mean()))
