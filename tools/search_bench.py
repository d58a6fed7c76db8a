"""BASELINE configs[4]: end-to-end k-best-first zero-window negamax at fixed depth driving the GPU
evaluators; reports search nodes per second (nodes taken from the open list and expanded per second
of search wall time, the figure numba_negamax_zero_window.py:865 prints).

    python tools/search_bench.py [--depth 4] [--batch 2048] [--dtype f32|bf16] [--no-policy] [--out FILE]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--batch", type=int, default=2048)   # playing_chess.py:108
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--no-policy", action="store_true")
    ap.add_argument("--cpu-material", action="store_true", help="no GPU: the reference's material-count fake (host-side ceiling)")
    ap.add_argument("--threads", type=int, default=0, help="host threads of the search (0 = the library default: cores - 4, at most 16)")
    ap.add_argument("--max-nodes", type=int, default=1 << 24, help="arena limit in nodes (0.8 KB each)")
    ap.add_argument("--fens", type=int, default=7, help="use the first N of the 7 test positions")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from batch_first_amd import search as S, weights as W
    from tests_fens import FENS

    if args.cpu_material:
        make = lambda: S.Search(max_batch_size=args.batch, tt_bits=22, threads=args.threads, max_nodes=args.max_nodes)
        closer = lambda: None
    else:
        from batch_first_amd import chestimator
        bp, _, closer = chestimator.get_inference_functions(W.make_weights(seed=1234, bn="perturbed"), dtype=args.dtype,
                                                            policy_weights=W.make_policy_weights(seed=4321))
        make = lambda: S.Search.on_gpu(bp, use_policy=not args.no_policy, max_batch_size=args.batch, tt_bits=22, threads=args.threads, max_nodes=args.max_nodes)
    rows = []
    stages = ["expand_seconds", "settle_seconds", "policy_seconds", "value_wait_seconds", "update_seconds",
              "compact_seconds", "select_seconds"]
    total = {"nodes_out": 0, "boards_evaluated": 0, "moves_scored": 0, "seconds": 0.0, "eval_wait_seconds": 0.0, "iterations": 0,
             "eval_calls": 0, "policy_calls": 0, "update_busy_seconds": 0.0, **{k: 0.0 for k in stages}}
    threads = 0
    for fen in FENS[:args.fens]:
        s = make()
        threads = s.threads
        t0 = time.perf_counter()
        value, move = s.iterative_deepening(fen, list(range(1, args.depth + 1)))
        wall = time.perf_counter() - t0
        st = s.stats()
        s.close()
        rows.append({"fen": fen, "value": value, "move": [int(x) for x in move], "wall_s": wall, **st})
        for k in total:
            total[k] += st[k]
        print("%-72s value %9.4f  %8d nodes  %7d evals  largest batch %5d  %.3f s  %9.0f nodes/s"
              % (fen[:72], value, st["nodes_out"], st["boards_evaluated"], st["max_batch"], st["seconds"],
                 st["nodes_out"] / max(st["seconds"], 1e-9)), flush=True)
    closer()
    out = {"config": "BASELINE configs[4]: iterative deepening mtd(f) to depth %d on the reference's 7 test positions, "
                     "max batch %d, %s" % (args.depth, args.batch,
                                          "material-count fake on the host" if args.cpu_material else
                                          "value network %s on the GPU%s" % (args.dtype, "" if args.no_policy else " + move-scoring network")),
           "nodes_per_s": total["nodes_out"] / total["seconds"], "evals_per_s": total["boards_evaluated"] / total["seconds"],
           "eval_wait_fraction": total["eval_wait_seconds"] / total["seconds"],
           "mean_eval_batch": total["boards_evaluated"] / max(total["eval_calls"], 1),
           "max_eval_batch": max(r["max_batch"] for r in rows),
           "mean_nodes_per_iteration": total["nodes_out"] / max(total["iterations"], 1),
           "host_threads": threads,
           "arena_peak_nodes": max(r["arena_peak"] for r in rows),
           "stage_share": {k[:-8]: total[k] / total["seconds"] for k in stages},
           "update_busy_share": total["update_busy_seconds"] / total["seconds"],
           **total, "positions": rows}
    print(json.dumps({k: v for k, v in out.items() if k != "positions"}))
    if args.out:
        os.makedirs(os.path.dirname(os.path.join(ROOT, args.out)), exist_ok=True)
        with open(os.path.join(ROOT, args.out), "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
