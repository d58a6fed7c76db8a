"""Reads per-wave stage stamps from a BF_STAMPS build (diagnostic only)."""
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
dtype = os.environ.get('STAMPS_DTYPE', 'f32')   # f32 or f32s
ctx = ctypes.c_void_p(); assert lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, _lib.precision_code(dtype)) == 0
structs = positions.to_compact(positions.random_structs(n, seed=5))
recs = torch.from_numpy(np.ascontiguousarray(structs).view(np.uint8).reshape(n, -1).copy()).cuda()
mb = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.bf_encode_mailbox_structs(ctx, recs.data_ptr(), recs.shape[1], n, mb.data_ptr(), st)
scores = torch.zeros(n, device="cuda")
dbg = torch.zeros((n, lib.bf_debug_floats_per_board()), device="cuda")
for _ in range(2):
    lib.bf_eval_mailbox_debug(ctx, mb.data_ptr(), n, scores.data_ptr(), dbg.data_ptr(), st)
torch.cuda.synchronize()
raw = dbg.view(-1)[: (n // 2) * 4 * 16 * 2].cpu().numpy().view(np.int64).reshape(n // 2, 4, 16)
t = raw[:, :, :7]
d = np.diff(t, axis=2)
names = ["setup", "stage1", "gemm", "conv4", "conv5-9", "head"]
ideal = [0, 240 * 32, 16 * 182 * 64, 576 * 64, 720 * 64, 0] if dtype == "f32" else [0, 240 * 32, 16 * 92 * 32, 4 * 108 * 32, 5 * 108 * 32, 0]
print("blocks", n // 2, "wave life median", np.median(t[:, :, 6] - t[:, :, 0]))
for i, nm in enumerate(names):
    print("%-8s median %9.0f  p10 %9.0f  p90 %9.0f   own MFMA cycles %7d" % (nm, np.median(d[:, :, i]), np.percentile(d[:, :, i], 10), np.percentile(d[:, :, i], 90), ideal[i]))
print("GEMM1 one tile (119 MFMA = 7616 cyc):", np.median(raw[:, :, 7]), " ReLU+GEMM2 (64 MFMA = 4096 cyc):", np.median(raw[:, :, 8]))
span = t[:, :, 6].max() - t[:, :, 0].min()
print("kernel span (cycles)", span)
