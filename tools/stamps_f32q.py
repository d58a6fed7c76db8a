"""Per-phase stamps of k_eval_f32q (diagnostic):
    tools/build_variant.sh qstamps eval_f32q -DBFQ_STAMPS && BATCHFIRST_LIB=build/variants/libbf_qstamps.so python tools/stamps_f32q.py [n]
Block 0 leaves eight s_memtime stamps per wave behind the last board's features (MODE 2 = the move-scoring network's
first stage, which is the value kernel without its head)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from batch_first_amd import _lib, ops, positions, weights
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
DT = sys.argv[2] if len(sys.argv) > 2 else "f32"
ctx = ops.DeviceContext(weights.make_weights(seed=1234, bn="perturbed"), device=0, dtype=DT)
ctx.set_policy(weights.make_policy_weights(seed=4321))
recs = ops.records_to_tensor(positions.to_compact(positions.random_structs(n, seed=7)), torch.device("cuda", 0))
x9 = torch.zeros((n + 1, 2048), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    _lib.check(ctx._lib.bf_policy_features_structs(ctx._ctx, recs.data_ptr(), recs.shape[1], n, x9.data_ptr(), st))
torch.cuda.synchronize()
s = x9[n].cpu().numpy().view(np.int64)[:64].reshape(8, 8)
names = ["set-up", "stage 1", "236->512 (4 tiles)", "barrier", "512->128", "zero + x3 + barriers", "128->32 (4 chunks)", "five 32->32 + store"]
d = np.diff(s, axis=1)
print("n =", n, " wave life (ticks of s_memtime):", (s[:, 7] - s[:, 0]).tolist())
for i in range(7):
    print("%-24s per wave: %s" % (names[i + 1] if i else "set-up .. stage 1", d[:, i].tolist()))
ctx.close()
