"""Headline benchmark: board evaluations per second at batch 65536 per GPU (BASELINE.json).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch: 65 536 packed node records (88-byte
stride, resident in HBM) -> record ingest kernel -> fused fp32 evaluation kernel -> 65 536
scores on the device.  With N > 1 every rank evaluates its own 65 536 boards (weak scaling,
boards are independent) and the scores are gathered to rank 0 over RCCL inside the step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 65536
SIDE_WARMUP = 12
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 matrix peak (not the 2:1-sparsity figure)
PEAK_HBM_GBS = 8000.0


# HBM traffic of the dominant kernel per 65 536-board launch, from the rocprofv3 PMC passes kept
# under profiles/ (FETCH_SIZE / WRITE_SIZE corrected as MI355X_MICROARCH.md prescribes; see
# tools/pmc_summary.py).  Counters cannot be read from inside this process, so the figure is the
# committed measurement of the same kernel and batch, not a live one.
# Round 3: the passes are taken with the kernel ALONE in the process (tools/kernel_only.py, tools/pmc_alone.sh: no
# other kernel's dirty lines in the figure) and the file records a hash of the kernel's sources; the bench line says
# whether that hash is the one of the sources this tree holds ("traffic_matches_sources").
TRAFFIC_PROFILES = {"f32": "profiles/r3/pmc_alone_f32.json", "bf16": "profiles/r3/pmc_alone_bf16.json"}
BF16_KERNEL = "k_eval_t16"


def profiled_traffic(dtype):
    """(bytes per launch, file, does the file belong to this tree's kernel sources)"""
    path = os.path.join(ROOT, TRAFFIC_PROFILES[dtype])
    try:
        with open(path) as fh:
            d = json.load(fh)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_only import source_sha16
        m = d["hbm_bytes_per_launch_median"]
        return float(m["read"] + m["write"]), TRAFFIC_PROFILES[dtype], d.get("source_sha16") == source_sha16(dtype)
    except (OSError, KeyError, ValueError, ImportError):
        return None, None, None


def cpu_baseline(budget_s=12.0):
    """The oracle's CPU path (C encoder, single thread as the reference's serial @njit loop, then
    the torch-CPU fp32 restatement of the graph on all host cores) on a bounded sample."""
    import torch
    from batch_first_amd import positions, weights
    from oracle import encoding, network

    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "c", "liboracle_encoder.so"))
    lib.oracle_struct_array_to_ann_inputs.restype = ctypes.c_size_t
    lib.oracle_struct_array_to_ann_inputs.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                      ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p,
                                                      ctypes.c_void_p]
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)))
    torch.set_num_threads(cores)
    w = weights.make_weights(seed=1234, bn="perturbed")
    chunk = 1024
    structs = positions.to_compact(positions.random_structs(chunk, seed=99))
    raw = np.ascontiguousarray(structs).view(np.uint8)
    nib = np.zeros(33 * chunk, np.uint8)
    occ = np.zeros(chunk, np.uint64)

    def one_chunk():
        nbytes = lib.oracle_struct_array_to_ann_inputs(raw.ctypes.data, structs.dtype.itemsize, None, chunk,
                                                       nib.ctypes.data, occ.ctypes.data, None)
        planes = encoding.decode_planes_fast(nib[:nbytes], occ)
        return network.forward_torch(planes, w)

    one_chunk()  # warm-up (thread pools, oneDNN primitives)
    done, t0 = 0, time.perf_counter()
    while True:
        one_chunk()
        done += chunk
        dt = time.perf_counter() - t0
        if dt >= budget_s or done >= 256 * chunk:
            break
    cpu_model = ""
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    cpu_model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": done / dt, "unit": "boards/s", "cores": cores, "kind": "port",
            "sample": "%d boards in chunks of %d: oracle C encoder (1 thread) + torch-CPU fp32 graph (%d threads), %.1f s"
                      % (done, chunk, cores, dt),
            "label": "CPU restatement of the reference graph (the reference's TensorFlow + Numba path cannot run here)",
            "cpu_model": cpu_model, "torch": torch.__version__}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help=argparse.SUPPRESS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="batch65536", choices=["batch65536", "shard2p24"],
                    help="batch65536 = BASELINE configs[1], 65536 boards per GPU (weak scaling, the headline); "
                         "shard2p24 = BASELINE configs[3], 2^24 boards in total sharded over the GPUs (strong scaling)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = BASELINE configs[1] (the headline); bf16 = the configs[2] kernel at the same batch")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from batch_first_amd import ops, positions, spec, weights

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run" % (args.gpus, world))
    # BF_BENCH_BACKEND=gloo is a rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share
    # devices, the score gather goes through host memory); the driver's runs use nccl (= RCCL), one GPU a rank
    backend = os.environ.get("BF_BENCH_BACKEND", "nccl")
    rehearsal = backend != "nccl"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group(backend)
        else:
            dist.init_process_group("nccl", device_id=dev)
    coll_dev = torch.device("cpu") if rehearsal else dev

    shard = args.workload == "shard2p24"
    if shard:
        from batch_first_amd.distributed import shard_range
        lo, hi = shard_range(1 << 24, rank, world)
        n = hi - lo
    else:
        n = args.batch
    w = weights.make_weights(seed=1234, bn="perturbed")
    if shard:
        # 2^24 synthetic boards: a seeded 65536-board block per rank, repeated (the generator is a host loop)
        block = ops.records_to_tensor(positions.to_compact(positions.random_structs(65536, seed=1000 + rank)), dev)
        recs = block.repeat((n + 65535) // 65536, 1)[:n].contiguous()
    else:
        recs = ops.records_to_tensor(positions.to_compact(positions.random_structs(n, seed=1000 + rank)), dev)
    gathered = [torch.empty(n, dtype=torch.float32, device=coll_dev) for _ in range(world)] if (world > 1 and rank == 0) else None

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    side_warmup = args.warmup if shard else max(args.warmup, SIDE_WARMUP)   # (a 2^24-board step is long enough as it is)

    def measure(dtype):
        """The timed region of the contract (W warm-up steps, exactly K steps between fences, MAX over
        ranks) and then the dominant kernel alone, for one precision.  Returns (elapsed seconds of the K
        steps, kernel ms per 65 536-board launch, the context).  The kernel figure is that of the kernel the timed
        step launches: for bf16 the record-in form of k_eval_t16 (one launch a step; the mailbox-in form, which a
        round-2 line quoted, is given beside it), for fp32 k_eval_f32 (the step's other launch is the 2 MB record ->
        mailbox kernel)."""
        ctx = ops.DeviceContext(w, device=dev_index, dtype=dtype)

        def step():
            scores = torch.ops.batchfirst.eval_structs(ctx.handle, recs)
            if world > 1:
                dist.gather(scores.cpu() if rehearsal else scores, gathered, dst=0)  # RCCL: scores back to the search host's rank
            return scores

        # The side blocks (a precision other than --dtype) follow tens of milliseconds of host work (packing and
        # uploading their weights) during which the GPU idles and drops its clock; their timed region is short (20 steps
        # of 1.7 ms for bf16), so three warm-up steps left one run in ten timing the clock's ramp (profiles/r3/README.md).
        # They get at least SIDE_WARMUP untimed steps, and say so in their "warmup" field.  The headline keeps W.
        for _ in range(args.warmup if dtype == args.dtype else side_warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            scores = step()
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert bool(torch.isfinite(scores).all())
        # dominant kernel alone, HIP events on the stream it is launched on, quoted at 65 536 boards
        krecs = recs[:65536].contiguous() if shard else recs
        mailbox = torch.ops.batchfirst.encode_mailbox_structs(ctx.handle, krecs)
        reps = max(3, min(args.steps, 20))
        # untimed burst first: after the host-side checks above the GPU has idled and its clock ramps up over the
        # first launches (single launches with a sync between them measure 2-5 % slower than a running burst)
        for _ in range(reps):
            torch.ops.batchfirst.eval_mailbox(ctx.handle, mailbox)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            torch.ops.batchfirst.eval_mailbox(ctx.handle, mailbox)
        ev1.record()
        torch.cuda.synchronize()
        mailbox_ms = ev0.elapsed_time(ev1) / reps
        if dtype != "bf16":
            return elapsed, mailbox_ms, None, ctx
        for _ in range(reps):
            torch.ops.batchfirst.eval_structs(ctx.handle, krecs)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(reps):
            torch.ops.batchfirst.eval_structs(ctx.handle, krecs)
        ev1.record()
        torch.cuda.synchronize()
        return elapsed, ev0.elapsed_time(ev1) / reps, mailbox_ms, ctx

    bf16 = args.dtype == "bf16"
    peak_tflops = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
    elapsed, kernel_ms, kernel_ms_mailbox_in, ctx = measure(args.dtype)
    total_boards = (1 << 24) if shard else world * n
    n_kernel = 65536 if shard else n
    achieved_tflops = spec.FLOPS_PER_BOARD * n_kernel / (kernel_ms * 1e-3) / 1e12
    # BASELINE configs[2] beside the fp32 headline, in the same run and under the same contract: the bf16
    # kernel on the same records (the headline fields above stay those of --dtype)
    other = None
    if not bf16:
        o_elapsed, o_kernel_ms, o_mailbox_ms, o_ctx = measure("bf16")
        o_tflops = spec.FLOPS_PER_BOARD * n_kernel / (o_kernel_ms * 1e-3) / 1e12
        o_traffic, o_src, o_match = profiled_traffic("bf16")
        other = {"value": total_boards * args.steps / o_elapsed, "unit": "boards/s", "steps": args.steps, "warmup": side_warmup,
                 "ms_per_step": 1e3 * o_elapsed / args.steps, "dtype": "bf16",
                 "config": {"workload": "BASELINE configs[2] kernel (bf16 operands, fp32 sums) on the headline's records and batch"},
                 "roofline": {"bound": "mfma", "kernel": BF16_KERNEL + "<4> (node records in: the kernel the timed step launches)",
                              "achieved": o_tflops, "peak": PEAK_BF16_MFMA_TFLOPS,
                              "unit": "TFLOP/s", "frac": o_tflops / PEAK_BF16_MFMA_TFLOPS, "kernel_ms": o_kernel_ms,
                              "kernel_ms_mailbox_in": o_mailbox_ms,
                              "traffic": o_traffic, "traffic_source": o_src, "traffic_matches_sources": o_match,
                              "flops_per_board": spec.FLOPS_PER_BOARD}}
        o_ctx.close()
    # BF_PRECISION_F32_SPLIT beside them (round 3): the fp32 kernel's structure with the 1x1 pair and the 3x3 layers
    # as sums of bf16 x bf16 products.  No fraction of the fp32 peak (it executes no fp32 MFMA); the block gives
    # boards/s, the kernel time beside k_eval_f32's, and the bf16 MFMA work the kernel EXECUTES per board (2 waves x
    # (240 + 16 x 92 + 9 x 108) v_mfma_f32_32x32x16_bf16 of 32 768 FLOP: 5.0 x the algorithmic FLOPs, that is the
    # price of the terms) against the bf16 peak
    def split_block(dtype, code, mfma_per_wave, what):
        s_elapsed, s_kernel_ms, _, s_ctx = measure(dtype)
        mfma_per_board = 2 * mfma_per_wave
        executed = mfma_per_board * 32768
        s_tflops = executed * n_kernel / (s_kernel_ms * 1e-3) / 1e12
        block = {"value": total_boards * args.steps / s_elapsed, "unit": "boards/s", "steps": args.steps, "warmup": side_warmup,
                 "ms_per_step": 1e3 * s_elapsed / args.steps, "dtype": dtype,
                 "config": {"workload": code + " on the headline's records and batch: " + what + ", fp32 sums, stage 1 and "
                                        "head as the fp32 kernel (scores within the fp32 kernel's 1e-4 bound, "
                                        "tests/test_gpu_parity.py)"},
                 "kernel": "k_eval_f32<0, %d>" % (1 if dtype == "f32s" else 2), "kernel_ms": s_kernel_ms, "kernel_ms_f32": kernel_ms,
                 "speedup_over_f32_kernel": kernel_ms / s_kernel_ms,
                 "executed_bf16": {"mfma_per_board": mfma_per_board, "flops_per_board": executed, "achieved": s_tflops,
                                   "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s",
                                   "share_of_bf16_peak": s_tflops / PEAK_BF16_MFMA_TFLOPS,
                                   "algorithmic_flops_per_board": spec.FLOPS_PER_BOARD,
                                   "note": "executed bf16 MFMA work (all terms and products), not algorithmic FLOPs: "
                                           "a utilisation figure for this kernel, not comparable with the roofline "
                                           "fractions above"}}
        s_ctx.close()
        return block

    split = split3 = None
    if not bf16:
        split = split_block("f32s", "BF_PRECISION_F32_SPLIT", 240 + 16 * 92 + 9 * 108,
                            "236->512->128 pair as (hi+mid) x (hi+mid) bf16, six 3x3 layers as (hi+mid+lo) x (hi+mid+lo) "
                            "bf16 (six products)")
        split3 = split_block("f32s3", "BF_PRECISION_F32_SPLIT3", 240 + 16 * 138 + 9 * 108,
                             "236->512->128 pair and six 3x3 layers as (hi+mid+lo) x (hi+mid+lo) bf16 (six products): every "
                             "matrix product at 24 significant bits, scores as close to the oracle as the fp32 kernel's")
    if shard:
        recs = recs[:65536].contiguous()
        n = 65536
    reps = max(3, min(args.steps, 20))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    # who ran: what torch.distributed and every rank's device report (a SCALE run shows its ranks here)
    prop = torch.cuda.get_device_properties(dev)
    me = {"rank": rank, "local_rank": local_rank, "device_index": dev.index, "name": prop.name,
          "gcnArchName": getattr(prop, "gcnArchName", ""), "cus": prop.multi_processor_count}
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
        dist_info = {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": ranks}
    else:
        dist_info = {"world_size": 1, "backend": None, "ranks": [me]}

    # the encoding kernel alone against HBM (records -> float planes, 88 B in + 3840 B out per board)
    # straight through the C ABI into one preallocated output: a 45 us kernel behind a Python custom op
    # would be timed together with the op's allocation and dispatch gaps
    planes = torch.empty((n, 15, 8, 8), dtype=torch.float32, device=dev)
    cur = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def encode_once():
        rc = ctx._lib.bf_encode_planes_structs(ctx._ctx, recs.data_ptr(), recs.shape[1], n, planes.data_ptr(), cur)
        if rc != 0:
            raise RuntimeError("bf_encode_planes_structs failed: %d" % rc)

    encode_once()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(reps):
        encode_once()
    ev1.record()
    torch.cuda.synchronize()
    enc_ms = ev0.elapsed_time(ev1) / reps
    enc_gbs = n * (88 + 3840) / (enc_ms * 1e-3) / 1e9
    del planes

    traffic, traffic_src, traffic_match = profiled_traffic(args.dtype)
    if rank == 0:
        value = total_boards * args.steps / elapsed
        out = {
            "metric": "board evals/sec, 2^24 boards sharded" if shard else "board evals/sec at batch 65536",
            "value": value,
            "unit": "boards/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if shard else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[3] (2^24 boards sharded, one launch per shard), network as " if shard else "")
                                   + "BASELINE configs[1]: value CNN (9 dilated input convs + tower + logit), "
                                   "random init, batch %d per GPU, %s, from packed node records resident in HBM "
                                   "(fused ingest + conv + head)" % (n, "bf16 operands / fp32 sums (configs[2] kernel)" if bf16 else "fp32"),
                       "per_gpu_batch": ((1 << 24) // world) if shard else n,
                       "global_batch": total_boards,
                       "parallelism": "independent shards per GPU" + (", RCCL gather of scores to rank 0" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": (BF16_KERNEL + "<4> (node records in)") if bf16 else "k_eval_f32", "achieved": achieved_tflops,
                         "peak": peak_tflops, "unit": "TFLOP/s", "frac": achieved_tflops / peak_tflops,
                         "traffic": traffic, "traffic_unit": "bytes per launch (HBM, PMC, kernel alone in the process)",
                         "traffic_source": traffic_src, "traffic_matches_sources": traffic_match,
                         "algorithmic_bytes": n * ((88 if bf16 else 32) + 4), "kernel_ms": kernel_ms,
                         **({"kernel_ms_mailbox_in": kernel_ms_mailbox_in} if bf16 else {}),
                         "flops_per_board": spec.FLOPS_PER_BOARD},
            "encode_roofline": {"bound": "hbm", "kernel": "k_to_planes<StructSrc>", "achieved": enc_gbs,
                                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": enc_gbs / PEAK_HBM_GBS,
                                "kernel_ms": enc_ms, "bytes_per_board": 88 + 3840},
        }
        out["distributed"] = dist_info
        if other is not None:
            out["bf16"] = other
        if split is not None:
            out["f32s"] = split
            out["f32s3"] = split3
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
