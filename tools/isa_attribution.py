"""Static attribution of a kernel's gfx950 instructions to source regions.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -gline-tables-only -I include -S --cuda-device-only \\
          -o /tmp/kernels_flow.s mav-detection_amd/csrc/kernels_flow.hip
    python tools/isa_attribution.py /tmp/kernels_flow.s _Z16k_blur_iter_fastILi6ELb1ELb1EE

Every instruction between the kernel's label and its s_endpgm is assigned to the source line of the last `.loc` directive in front of
it (the innermost inlined frame, as the assembler prints it) and counted per region of kernels_flow.hip and per class: scalar ALU / moves
(s_*), s_waitcnt, scalar branches, s_load, vector ALU (v_*), vector memory (global_ / buffer_ / flat_), LDS (ds_*).  Static counts; in
this kernel every loop body runs at most once per thread, so for an interior tile they are also what a wave executes, minus the edge
branch that is listed separately."""
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
REGIONS = [                                   # (first line, last line, name) in kernels_flow.hip -- checked against the source below
    ("tile_of_block", "XCD-aware tile order (tile_of_block)"),
    ("entry_r0", "entry: R0 of the thread's four update pixels requested"),
    ("col_interior", "column pass, interior tile (float2 columns, 28 clamped row addresses)"),
    ("col_edge", "column pass, tiles on the left / right image edge (scalar columns)"),
    ("row_solve", "row pass from LDS + 2x2 solve + flow store / park"),
    ("update", "update phase: gather R1 (gather_issue), UpdateMatrices (update_finish), M' stores"),
]


def region_lines(src):
    """line ranges of the regions, found by their marker comments / function names so that edits do not silently shift them"""
    L = open(src).read().split("\n")
    def find(pat, start=0):
        for i in range(start, len(L)):
            if pat in L[i]:
                return i + 1
        raise SystemExit(f"marker not found: {pat}")
    k = find("void k_blur_iter_fast(")
    r = {}
    r["tile_of_block"] = (find("bool tile_of_block("), find("#ifdef MAV_STAMPS") - 1)
    r["entry_r0"] = (find("// entry: R0 of this thread's four phase-C pixels", k), find("if (x0 >= M_T && x0 + FT_X + M_T <= w)", k) - 1)
    r["col_interior"] = (find("if (x0 >= M_T && x0 + FT_X + M_T <= w)", k), find("// tiles touching the left/right image edge", k) - 1)
    r["col_edge"] = (find("// tiles touching the left/right image edge", k), find("STAMP(ts1);", k) - 1)
    r["row_solve"] = (find("STAMP(ts1);", k), find("STAMP(ts3);", k))
    r["update"] = (find("STAMP(ts3);", k) + 1, find("STAMP_ADD(0, ts0, ts1)", k))
    r["gather_issue"] = (find("void gather_issue("), find("void update_finish(") - 1)
    r["update_finish"] = (find("void update_finish("), find("struct __attribute__((packed, aligned(4))) F2U") - 1)
    r["solve_px"] = (find("void solve_px("), find("float2 upsample_flow(") - 1)
    r["clampi"] = (find("int clampi("), find("int clampi("))
    return r, k


def classify(op):
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_nop", "s_sleep")):
        return "s_branch/barrier/nop"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "s_load"
    if op.startswith("s_"):
        return "s_alu/mov"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "v_alu"
    return "other"


import os
SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mav-detection_amd", "csrc", "kernels_flow.hip")
R, kline = region_lines(SRC)
fold = {"gather_issue": "update", "update_finish": "update", "solve_px": "row_solve"}
counts, classes = {}, ["s_alu/mov", "s_waitcnt", "s_branch/barrier/nop", "s_load", "v_alu", "vmem", "lds", "other"]
files = {}
inside, line, fileno = False, 0, 0
clamp_in = {}
for raw in open(path):
    t = raw.strip()
    m = re.match(r"\.file\s+(\d+)\s+\"([^\"]*)\"(?:\s+\"([^\"]*)\")?", t)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2))
    if t.startswith(sym) and t.split(":")[0].startswith(sym) and ":" in t:
        inside = True
        continue
    if not inside:
        continue
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        fileno, line = int(m.group(1)), int(m.group(2))
        continue
    if t.startswith(".Lfunc_end"):
        break
    if not t or t.startswith((".", ";")) or t.endswith(":"):
        continue
    op = t.split()[0]
    if not re.match(r"[a-z]", op):
        continue
    reg = "other"
    if files.get(fileno, "").endswith("kernels_flow.hip"):
        for name, (a, b) in R.items():
            if a <= line <= b:
                reg = name
        if reg == "clampi":
            reg = "col_interior(clampi)"
        reg = fold.get(reg, reg)
    else:
        reg = "hip headers (min / floorf / fmaf / shuffles)"
    counts.setdefault(reg, dict.fromkeys(classes, 0))[classify(op)] += 1

print(f"{sym}: static instruction counts by source region ({os.path.basename(SRC)}, kernel at line {kline})")
print(f"  {'region':62s} " + " ".join(f"{c:>10s}" for c in classes))
tot = dict.fromkeys(classes, 0)
names = dict(REGIONS)
for reg in list(names) + [r for r in counts if r not in names]:
    if reg not in counts:
        continue
    c = counts[reg]
    print(f"  {names.get(reg, reg)[:62]:62s} " + " ".join(f"{c[k]:10d}" for k in classes))
    for k in classes:
        tot[k] += c[k]
print(f"  {'total':62s} " + " ".join(f"{tot[k]:10d}" for k in classes))
