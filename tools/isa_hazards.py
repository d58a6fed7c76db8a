#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc --save-temps, *.s) for the hazards hipcc does
not handle around inline asm (cdna_hip_programming.md section 5.7, items 2 and 4):

  H1  a VALU instruction writes an SGPR (v_readlane_b32, v_readfirstlane_b32, v_cmp*, carry-outs)
      and a vector-memory instruction INSIDE an inline-asm block reads that SGPR as base /
      descriptor / offset fewer than 5 wait states later.  The compiler pads this pair when both
      instructions are its own; an asm block is opaque to its hazard recogniser.  This is what an
      SGPR restored from a spill lane (v_readlane_b32) right in front of an asm
      `global_load_lds_dwordx4 v, s[a:b]` looks like: the load issues with the pair's old value.
  H2  M0 written by compiler code (outside ;;#ASMSTART/;;#ASMEND) in a kernel whose asm blocks read
      M0 in a later statement than the one that set it.
  H3  an SALU write of M0 followed by an LDS-DMA (`global_load_lds_*`, `buffer_load ... lds`)
      without one wait state between.

Usage: isa_hazards.py file.s [--kernel SUBSTR] [--verbose]   (exit status 1 when a hazard is found)
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|flat_|scratch_)")
SREG = re.compile(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b")
VALU_SGPR_WRITERS = ("v_readlane_b32", "v_readfirstlane_b32")


def sregs(text):
    out = set()
    if re.search(r"\bvcc\b", text):
        out.update((106, 107))
    if re.search(r"\bvcc_lo\b", text):
        out.add(106)
    if re.search(r"\bvcc_hi\b", text):
        out.add(107)
    for m in SREG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def valu_sgpr_dst(ins):
    """SGPRs written by a VALU instruction, empty for anything else."""
    op = ins.split()[0]
    if op in VALU_SGPR_WRITERS:
        return sregs(ins.split(",")[0])
    if op.startswith("v_cmp"):
        return sregs(ins.split(",")[0]) if "_e64" in op else {106, 107}
    if op.startswith(("v_add_co", "v_sub_co", "v_addc_co", "v_subb_co", "v_div_scale", "v_mad_u64", "v_mad_i64")):
        parts = ins.split(",")
        return sregs(parts[1]) if len(parts) > 1 else set()
    return set()


def wait_states(ins):
    m = re.match(r"\s*s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def scan(path, kernel_filter=None, verbose=False):
    findings = []
    kernel = None
    in_asm = False
    window = []          # (instruction text, wait states it provides, inside asm?)
    m0_compiler_writes = []
    asm_m0_reads_without_set = 0
    stats = {}
    for lineno, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].rstrip() if not raw.lstrip().startswith(";;#") else raw.strip()
        if raw.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            asm_sets_m0 = False
            continue
        if raw.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        s = line.strip()
        if not s or s.startswith((".", "//")):
            m = re.match(r"\s*\.type\s+(\S+),@function", raw)
            if m:
                kernel = m.group(1)
                window = []
                stats[kernel] = {"asm_vmem": 0, "valu_sgpr_writes": 0}
            continue
        if s.endswith(":"):
            window = []      # a label: predecessors unknown, the compiler pads across edges itself
            continue
        if kernel is None or (kernel_filter and kernel_filter not in kernel):
            continue
        op = s.split()[0]
        # H2 / H3: M0
        writes_m0 = bool(re.match(r"s_\w+\s+m0\s*,", s))
        if writes_m0 and not in_asm:
            m0_compiler_writes.append((kernel, lineno, s))
        if in_asm and writes_m0:
            asm_sets_m0 = True
        is_dma = op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in s)
        if is_dma:
            prev = window[-1][0] if window else ""
            if re.match(r"s_\w+\s+m0\s*,", prev.strip()):
                findings.append(("H3", kernel, lineno, s, prev.strip()))
            if in_asm and not asm_sets_m0:
                asm_m0_reads_without_set += 1
        # H1
        if in_asm and VMEM.match(s):
            stats[kernel]["asm_vmem"] += 1
            need = sregs(s)
            dist = 0
            for prev, ws, prev_asm in reversed(window):
                if dist >= 5:
                    break
                hit = valu_sgpr_dst(prev.strip()) & need
                if hit:
                    findings.append(("H1", kernel, lineno, s, "%s  (%d wait states before, needs 5; s%s)" % (
                        prev.strip(), dist, sorted(hit))))
                    break
                dist += ws
        if valu_sgpr_dst(s):
            stats[kernel]["valu_sgpr_writes"] += 1
        window.append((s, wait_states(s), in_asm))
        if len(window) > 16:
            window.pop(0)
    if asm_m0_reads_without_set:
        for k, ln, s in m0_compiler_writes:
            if not kernel_filter or kernel_filter in k:
                findings.append(("H2", k, ln, s, "compiler writes M0 in a kernel whose asm LDS-DMA relies on an earlier statement's M0"))
    if verbose:
        for k, st in stats.items():
            if not kernel_filter or kernel_filter in k:
                print("%s: %d vector-memory instructions in asm blocks, %d VALU writes of SGPRs" % (k, st["asm_vmem"], st["valu_sgpr_writes"]))
        print("LDS-DMA statements that rely on an earlier statement's M0: %d; compiler writes of M0: %d" % (
            asm_m0_reads_without_set, len(m0_compiler_writes)))
    return findings


def main(argv):
    if len(argv) < 2:
        print(__doc__)
        return 2
    kf = argv[argv.index("--kernel") + 1] if "--kernel" in argv else None
    findings = scan(argv[1], kf, "--verbose" in argv)
    for kind, kernel, lineno, ins, why in findings:
        print("%s %s line %d: %s\n      <- %s" % (kind, kernel, lineno, ins, why))
    print("%d hazard(s) in %s" % (len(findings), argv[1]))
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
