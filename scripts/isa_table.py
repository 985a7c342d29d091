#!/usr/bin/env python3
"""Per-phase instruction table of a gfx950 kernel from the compiler's assembly (no GPU needed).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -gline-tables-only -save-temps -c X.hip
    python scripts/isa_table.py X-hip-amdgcn-amd-amdhsa-gfx950.s <kernel-substring> [--phases phases.json] [--blocks]

Every instruction is attributed to the innermost source location the compiler recorded for it (`.loc file line`), the
locations are bucketed into phases (file / line ranges), and the instructions are classed by issue pipe:
VALU (of which packed, fp64, transcendental, 32-bit integer multiply), SALU, LDS, VMEM, SMEM, other (waitcnt, branch).
`--blocks` also lists every basic block (label, instruction count, the back edges), which is how the loop body is found:
phases can then be restricted to the label range of the loop with `--from LABEL --to LABEL`.
"""
import argparse
import collections
import json
import re
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
QUARTER = ("v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32", "v_mad_u64_u32", "v_mad_i64_i32")


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop", "s_branch", "s_cbranch", "s_barrier", "s_endpgm", "s_setprio", "s_sleep")):
        return "other"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def sub_class(op):
    out = []
    if op.startswith("v_pk_"):
        out.append("pk")
    if "_f64" in op:
        out.append("f64")
    if op.startswith(TRANS):
        out.append("trans")
    if op.startswith(QUARTER):
        out.append("imul")
    if op.startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr")):
        out.append("mov")
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        out.append("lane")
    if op.startswith(("v_cndmask", "v_cmp")):
        out.append("sel")
    if "dpp" in op or op.startswith(("v_permlane", "ds_bpermute", "ds_permute", "ds_swizzle")):
        out.append("xlane")
    return out


def parse(path, kernel, by_root=False):
    files = {}
    cur = None
    inside = False
    loc = (None, 0)
    root = None
    label = "<entry>"
    insts = []                       # (label, op, file, line, text)
    fre = re.compile(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"')
    lre = re.compile(r"\s*\.loc\s+(\d+)\s+(\d+)")
    with open(path) as fh:
        for raw in fh:
            m = fre.match(raw)
            if m:
                files[int(m.group(1))] = m.group(3)
                continue
            comment = raw.split(";", 1)[1] if ";" in raw else ""
            s = raw.rstrip("\n").split(";")[0].rstrip()
            if not inside:
                if s.endswith(":") and not s.startswith((".", "\t", " ")) and kernel in s:
                    inside = True
                    cur = s[:-1]
                continue
            if s.startswith(".Lfunc_end"):
                break
            m = lre.match(s)
            if m:
                loc = (int(m.group(1)), int(m.group(2)))
                root = None
                if by_root:                      # outermost frame of the inlined-at chain: "a.h:10:3 @[ b.hip:200:7 @[ c.hip:50:1 ] ]"
                    chain = re.findall(r"([^\s\[\]@]+):(\d+):\d+", comment)
                    if chain:
                        root = (chain[-1][0].split("/")[-1], int(chain[-1][1]))
                continue
            t = s.strip()
            if not t or t.startswith((";", "//")):
                continue
            if t.endswith(":") and not t.startswith("\t."):
                label = t[:-1]
                continue
            if t.startswith("."):
                continue
            op = t.split()[0]
            if not re.match(r"^[a-z][a-z0-9_]+$", op):
                continue
            if by_root and root:
                insts.append((label, op, root[0], root[1], t))
            else:
                insts.append((label, op, files.get(loc[0], "?"), loc[1], t))
    return cur, insts


def phase_of(phases, f, line):
    for ph in phases:
        if ph["file"] != f:
            continue
        for lo, hi in ph["lines"]:
            if lo <= line <= hi:
                return ph["name"]
    return "%s (other lines)" % f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--phases")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--from", dest="lo")
    ap.add_argument("--to", dest="hi")
    ap.add_argument("--lines", action="store_true", help="per source line instead of per phase")
    ap.add_argument("--by-root", action="store_true", help="attribute an instruction to the outermost frame of its inlined-at chain (the kernel body's line) instead of the innermost location")
    ap.add_argument("--loop", action="store_true", help="restrict to the largest loop of the kernel (the longest back-edge span)")
    a = ap.parse_args()
    name, insts = parse(a.asm, a.kernel, a.by_root)
    if not insts:
        sys.exit("kernel not found")
    print("kernel:", name, "| instructions:", len(insts))
    if a.blocks:
        order = []
        count = collections.Counter()
        for lab, op, *_ in insts:
            if lab not in count:
                order.append(lab)
            count[lab] += 1
        pos = {lab: i for i, lab in enumerate(order)}
        for lab, op, f, line, text in insts:
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = text.split()[-1]
                if tgt in pos and pos[tgt] <= pos[lab]:
                    span = sum(count[x] for x in order[pos[tgt]:pos[lab] + 1])
                    print("  back edge %-12s -> %-12s  loop span %5d instructions" % (lab, tgt, span))
        return
    if a.loop:
        order = []
        count = collections.Counter()
        for lab, *_ in insts:
            if lab not in count:
                order.append(lab)
            count[lab] += 1
        pos = {lab: i for i, lab in enumerate(order)}
        best = (0, None, None)
        for lab, op, f, line, text in insts:
            if op.startswith(("s_cbranch", "s_branch")):
                tgt = text.split()[-1]
                if tgt in pos and pos[tgt] <= pos[lab]:
                    span = sum(count[x] for x in order[pos[tgt]:pos[lab] + 1])
                    if span > best[0]:
                        best = (span, tgt, lab)
        a.lo, a.hi = best[1], best[2]
    if a.lo or a.hi:
        order = []
        for lab, *_ in insts:
            if lab not in order:
                order.append(lab)
        i0 = order.index(a.lo) if a.lo else 0
        i1 = order.index(a.hi) if a.hi else len(order) - 1
        keep = set(order[i0:i1 + 1])
        insts = [x for x in insts if x[0] in keep]
        print("restricted to labels %s..%s: %d instructions" % (order[i0], order[i1], len(insts)))
    phases = json.load(open(a.phases)) if a.phases else []
    tab = collections.OrderedDict()
    for lab, op, f, line, text in insts:
        key = "%s:%d" % (f, line) if a.lines else phase_of(phases, f, line)
        row = tab.setdefault(key, collections.Counter())
        row[classify(op)] += 1
        for sc in sub_class(op):
            row[sc] += 1
    cols = ["VALU", "pk", "f64", "trans", "imul", "mov", "sel", "lane", "xlane", "SALU", "LDS", "VMEM", "SMEM", "other"]
    names = [ph["name"] for ph in phases if ph["name"] in tab] + [k for k in tab if k not in {ph["name"] for ph in phases}]
    w = max(len(k) for k in names) + 2
    print("%-*s" % (w, "phase") + "".join("%7s" % c for c in cols))
    tot = collections.Counter()
    for k in names:
        print("%-*s" % (w, k) + "".join("%7d" % tab[k][c] for c in cols))
        tot.update(tab[k])
    print("%-*s" % (w, "TOTAL") + "".join("%7d" % tot[c] for c in cols))


if __name__ == "__main__":
    main()
