import ctypes, os, sys
import numpy as np
ROOT = "/root/repo" if os.path.exists("/root/repo") else os.getcwd()
sys.path.insert(0, os.getcwd())
import torch
from batch_first_amd import _lib, positions, weights
path, n = sys.argv[1], int(sys.argv[2])
lib = ctypes.CDLL(os.path.join(os.getcwd(), path))
for fn, (res, argt) in _lib.SIGNATURES.items():
    if hasattr(lib, fn):
        getattr(lib, fn).restype = res; getattr(lib, fn).argtypes = argt
blob = weights.to_blob(weights.make_weights(seed=1234, bn="perturbed"))
ctx = ctypes.c_void_p(); assert lib.bf_ctx_create_ex(ctypes.byref(ctx), 0, blob.ctypes.data, blob.size, 1) == 0
structs = positions.to_compact(positions.random_structs(n, seed=5))
recs = torch.from_numpy(np.ascontiguousarray(structs).view(np.uint8).reshape(n, -1).copy()).cuda()
mb = torch.empty((n, 32), dtype=torch.uint8, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
assert lib.bf_encode_mailbox_structs(ctx, recs.data_ptr(), recs.shape[1], n, mb.data_ptr(), st) == 0
scores = torch.zeros(n, device="cuda")
rc = lib.bf_eval_mailbox(ctx, mb.data_ptr(), n, scores.data_ptr(), st)
torch.cuda.synchronize()
print(path, n, "rc", rc, "ok; scores[:4]", scores[:4].cpu().numpy(), flush=True)
