"""Per-wave stage stamps of a BF_STAMPS build of eval_t16h.hip (diagnostic only).
    BF_T16_KERNEL=h BF_T16H_STAMPS=1 python tools/stamps_t16h.py build/variants/libbf_x.so 65536"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batch_first_amd import _lib, positions, weights
path, n = sys.argv[1], int(sys.argv[2])
lib = ctypes.CDLL(os.path.join(ROOT, path))
for fn, (res, argt) in _lib.SIGNATURES.items():
    getattr(lib, fn).restype = res; getattr(lib, fn).argtypes = argt
blob = weights.to_blob(weights.make_weights())
ctx = ctypes.c_void_p(); assert lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, 1) == 0
structs = positions.to_compact(positions.random_structs(n, seed=5))
recs = torch.from_numpy(np.ascontiguousarray(structs).view(np.uint8).reshape(n, -1).copy()).cuda()
mb = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.bf_encode_mailbox_structs(ctx, recs.data_ptr(), recs.shape[1], n, mb.data_ptr(), st)
scores = torch.zeros(n, device="cuda")
dbg = torch.zeros((4096, 64), device="cuda")
for _ in range(3):
    assert lib.bf_eval_mailbox_debug(ctx, mb.data_ptr(), n, scores.data_ptr(), dbg.data_ptr(), st) == 0
torch.cuda.synchronize()
waves = 8
blocks = min((n + 3) // 4, torch.cuda.get_device_properties(0).multi_processor_count)
raw = dbg.view(-1)[: blocks * waves * 16 * 2].cpu().numpy().view(np.int64).reshape(blocks, waves, 16)
M = (1 << 32) - 1
t = raw[:, :, :7]
d = np.diff(t, axis=2) & M
names = ["image", "stage1", "pair", "conv4", "conv5-9", "head"]
# matrix-pipe cycles of ONE SIMD for the stage (its two waves together)
ideal = [0, 332 * 16, 384 * 4 * 16, 288 * 16, 360 * 16, 16 * 16]
life = (t[:, :, 6] - t[:, :, 0]) & M
print("blocks", blocks, "board-quad walk (one SIMD: two waves, one board) median %.0f cycles; MFMA cycles %d" % (np.median(life), sum(ideal)))
for i, nm in enumerate(names):
    print("%-8s median %9.0f  p10 %9.0f  p90 %9.0f   MFMA cycles of the SIMD %7d" % (nm, np.median(d[:, :, i]), np.percentile(d[:, :, i], 10), np.percentile(d[:, :, i], 90), ideal[i]))
