import sys, os, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from batch_first_amd import ops, positions, weights, chestimator, encoder
w = weights.make_weights(seed=1234, bn="perturbed")
DT = sys.argv[1] if len(sys.argv) > 1 else "f32"
ctx = ops.DeviceContext(w, device=0, dtype=DT)
bp, _, closer = chestimator.get_inference_functions(w, device=0, dtype=DT)
dev = torch.device("cuda", 0)
structs_all = positions.to_compact(positions.random_structs(1024, seed=7))
for n in (1, 16, 64, 128, 146, 200, 256, 257, 384, 512, 1024):
    recs = ops.records_to_tensor(structs_all[:n], dev)
    for _ in range(5): torch.ops.batchfirst.eval_structs(ctx.handle, recs)
    torch.cuda.synchronize()
    ts = []
    for _ in range(30):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.ops.batchfirst.eval_structs(ctx.handle, recs); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    s = structs_all[:n]
    for _ in range(3): bp.eval_structs(s)
    t0 = time.perf_counter()
    for _ in range(100): bp.eval_structs(s)
    host = (time.perf_counter() - t0) / 100 * 1e6
    print(DT + " N=%5d  device %7.1f us   host call (records) %7.1f us" % (n, float(np.median(ts)), host), flush=True)
closer(); ctx.close()
