"""Static issue model of a one-wave-per-SIMD MFMA kernel (diagnostic only): walks the kernel's ISA
(hipcc -S output) with the issue costs measured by tools/issue_rate.hip and reports where the filler
instructions after an MFMA outlast the MFMA's pipe time, i.e. where the matrix pipe idles for issue
reasons alone (memory waits and barriers are not modelled).
    python tools/gap_model.py build/eval_bf16.s <kernel symbol substring> [top N]"""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l and ":" in l and not l.startswith("\t"))
body = []
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"): break
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
    body.append(t.split(";")[0].strip())
def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m: return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)$", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()
t = 0.0; pipe_free = 0.0; idle_total = 0.0; gaps = []; last_valu_dst = set(); fill = []; n_mfma = 0; mfma_cycles = 0
for idx, ins in enumerate(body):
    op = ins.split()[0]
    ops = [o.strip() for o in ins[len(op):].split(",")]
    if op.startswith("v_mfma"):
        dur = 16 if "16x16x32" in op else 32
        startt = max(t, pipe_free)
        idle = max(0.0, t - pipe_free) if n_mfma else 0.0
        if idle > 0: gaps.append((idle, idx, list(fill)))
        idle_total += idle
        pipe_free = startt + dur; t = startt + 4; n_mfma += 1; mfma_cycles += dur; fill = []
        last_valu_dst = set(); continue
    if op.startswith("v_"):
        src = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
        cost = 8.0 if (src & last_valu_dst) else 4.3
        if op.startswith("v_accvgpr"): cost = 8.0
        last_valu_dst = regs(ops[0]) if ops else set()
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"): cost = 4.0
    elif op.startswith("s_barrier"): cost = 8.0
    elif op.startswith("s_"): cost = 8.0
    elif op.startswith("ds_") or op.startswith("global_") or op.startswith("buffer_"): cost = 4.3
    else: cost = 4.3
    t += cost; fill.append(ins)
print("MFMAs %d, own pipe cycles %d, modelled issue-only idle %.0f cycles (%.1f %%)" % (n_mfma, mfma_cycles, idle_total, 100 * idle_total / mfma_cycles))
gaps.sort(key=lambda g: -g[0])
for idle, idx, f in gaps[:top]:
    print("--- idle %.0f before instruction %d, %d fillers:" % (idle, idx, len(f)))
    for x in f[:40]: print("      " + x)
hist = {}
for idle, idx, f in gaps: hist[int(idle // 8) * 8] = hist.get(int(idle // 8) * 8, 0) + 1
print("gap histogram (idle cycles bucket: count):", sorted(hist.items()))
