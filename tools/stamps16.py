"""Reads per-wave stage stamps from a BF_STAMPS build of the round-1 bf16 kernel k_eval_bf16 (diagnostic only; that kernel
was removed in round 3 -- kept as the record of how the stamps under profiles/r1 were taken; tools/stamps_t16.py is its
successor for k_eval_t16).
    python tools/stamps16.py build/variants/libbf_x.so 65536 [boards per workgroup]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batch_first_amd import _lib, positions, weights
path, n = sys.argv[1], int(sys.argv[2])
bpb = int(sys.argv[3]) if len(sys.argv) > 3 else 2
lib = ctypes.CDLL(os.path.join(ROOT, path))
for fn, (res, argt) in _lib.SIGNATURES.items():
    getattr(lib, fn).restype = res; getattr(lib, fn).argtypes = argt
blob = weights.to_blob(weights.make_weights())
if os.environ.get("BF_ZERO_WEIGHTS"):   # diagnostic: data-dependent power (operand toggling) against the same instruction stream
    blob = np.zeros_like(blob)
ctx = ctypes.c_void_p(); assert lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, 1) == 0
structs = positions.to_compact(positions.random_structs(n, seed=5))
recs = torch.from_numpy(np.ascontiguousarray(structs).view(np.uint8).reshape(n, -1).copy()).cuda()
mb = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
lib.bf_encode_mailbox_structs(ctx, recs.data_ptr(), recs.shape[1], n, mb.data_ptr(), st)
scores = torch.zeros(n, device="cuda")
dbg = torch.zeros((n, lib.bf_debug_floats_per_board()), device="cuda")
for _ in range(3):
    lib.bf_eval_mailbox_debug(ctx, mb.data_ptr(), n, scores.data_ptr(), dbg.data_ptr(), st)
torch.cuda.synchronize()
waves = bpb
# persistent workgroups: one per compute unit, stamps are those of each workgroup's last board quad
blocks = min((n + bpb - 1) // bpb, torch.cuda.get_device_properties(0).multi_processor_count)
raw = dbg.view(-1)[: blocks * waves * 16 * 2].cpu().numpy().view(np.int64).reshape(blocks, waves, 16)
M = (1 << 32) - 1   # the kernel keeps the low 32 bits of each stamp
t = raw[:, :, :7]
d = np.diff(t, axis=2) & M
names = ["setup", "stage1", "gemm", "conv4", "conv5-9", "head"]
ideal = [0, 160 * 32, 16 * 46 * 32, 144 * 32, 180 * 32, 0]
life = (t[:, :, 6] - t[:, :, 0]) & M
print("blocks", blocks, "wave life median %.0f  own MFMA cycles %d" % (np.median(life), sum(ideal)))
for i, nm in enumerate(names):
    print("%-8s median %9.0f  p10 %9.0f  p90 %9.0f   own MFMA cycles %7d" % (nm, np.median(d[:, :, i]), np.percentile(d[:, :, i], 10), np.percentile(d[:, :, i], 90), ideal[i]))
step = (raw[:, :, 9] - raw[:, :, 8]) & M
print("one step of the 1x1 pair (23 groups, 46 MFMAs = 1472 cycles): median %.0f p10 %.0f p90 %.0f" % (np.median(step), np.percentile(step, 10), np.percentile(step, 90)))
rt0, rt1 = raw[:, :, 7], raw[:, :, 14]
rt_life = ((rt1 - rt0) & M).astype(np.float64)
ok = rt_life > 0
print("stamp ticks per 100 MHz tick: median %.2f  => stamp clock %.3f GHz" % (np.median(life[ok] / rt_life[ok]), np.median(life[ok] / rt_life[ok]) * 0.1))
