#!/usr/bin/env python
"""Issue model of the DESCENT scans of gls_kernel's headline instantiation (TSP100 x 1024: gls_kernel<TriDGlobalP,false,2,false,4>)
from its disassembly: executed instructions of ONE wavefront's share of a scan, per scan kind, for bench.py's critical_path.descent.

    cd gnngls_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DGLS_DEV_ONLY_HEADLINE \
        -DGLS_ISA_MARKS --cuda-device-only -S gls_kernels.hip -o /tmp/gls_headline.s
    python scripts/isa_descent_model.py /tmp/gls_headline.s        -> profiles/r06_isa/descent_model.json, descent_regions.txt,
                                                                      headline_descent_scans.s (the marked regions)

The marks (ISA_MARK in gls_descent_scans.h / gls_kernels.hip; comment lines, the shipped library has none) cut the scans into
regions; a region's instruction count is static, how often a scan executes it follows from n (loop trip counts) and, where the
trip count depends on the data, from the counting pass of the bench (stated per entry).  Everything refers to the copy of
local_search that the outer iterations run (the second one in the kernel: the start descent of algorithms.py:142 is inlined
separately).

Model: a wavefront issues at most one instruction per ~4.4 cycles when it has its SIMD to itself (profiles/r05_isa_latency_probe.txt:
any instruction class); in the descent all four wavefronts of a workgroup scan and the SIMD they sit on is shared with one
wavefront of each of the three other resident workgroups, which are in their own descent 57 % of the time and otherwise have one
wavefront of four active: a vector instruction occupies the SIMD's ALU for 4 cycles (wave64 on 16 lanes), so with W wavefronts
issuing vector work at once each gets one slot per 4 W cycles.  issue_model_cycles = instructions x CPI with CPI = 4.4 x (1 + 3 x
0.68) / 2 = 6.7: the single-wavefront issue interval stretched by the expected number of co-resident wavefronts that compete for
the same issue ports, half of whose instructions (the scalar, LDS and branch ones) issue beside a vector instruction of another
wavefront.  The table states the single-wavefront floor (x 4.4) next to it.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from isa_blocks import classify  # noqa: E402

SINGLE_WAVE_CPI = 4.4
ACTIVE_OTHERS = 3 * (0.57 + 0.43 / 4)            # expected co-resident wavefronts of other workgroups that are issuing
CPI = SINGLE_WAVE_CPI * (1 + ACTIVE_OTHERS) / 2


def load(path):
    lines = open(path).read().splitlines()
    marks = [(i, l.split()[2]) for i, l in enumerate(lines) if l.strip().startswith("; GLSMARK")]
    return lines, marks


def count(lines, a, b):
    """instructions by class in lines[a:b] (labels, comments, directives skipped)"""
    cnt = {}
    for l in lines[a:b]:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or re.match(r"^[.\w$]+:", t):
            continue
        c = classify(t.split()[0])
        cnt[c] = cnt.get(c, 0) + 1
    return cnt


def total(c):
    return sum(c.values())


def main():
    path = sys.argv[1]
    n = 100
    lines, marks = load(path)
    names = [m for _, m in marks]
    # second copy of the descent (the one the outer iterations run): last occurrence of each region start
    def last(name, before=None):
        idx = [i for i, m in marks if m == name and (before is None or i < before)]
        return idx[-1]

    def nxt(i):
        return min(j for j, _ in marks if j > i)

    out_regions = []

    def region(label, a, b, times, note):
        c = count(lines, a, b)
        out_regions.append({"region": label, "lines": [a + 1, b], "instructions": total(c), "by_class": c, "times_per_scan": times, "note": note})
        return total(c) * times, {k: v * times for k, v in c.items()}

    def add(dst, src):
        for k, v in src.items():
            dst[k] = dst.get(k, 0) + v

    model = {}
    # ---- relocate_a2a, every row: scan_relocate_a2a_lean<2, true, 6, ..., QT = true> ----
    b0 = last("reloc_lean_begin")
    e0 = [i for i, m in marks if m == "reloc_lean_end" and i > b0][0]
    inner = [(i, m) for i, m in marks if b0 < i < e0]
    g = [i for i, m in inner if m == "reloc_lean_group"]
    s1 = [i for i, m in inner if m == "reloc_lean_single"]
    cands = [i for i, m in inner if m == "reloc_lean_candidates"]
    # wavefront share at n = 100: two row blocks x two wavefronts, 50 target edges each: 8 groups of six + 2 single steps (the wavefront
    # whose range crosses position 63 runs 2 + 6 groups and 1 + 1 single steps: the same count)
    tot, cls = 0, {}
    t_, c_ = region("relocate_full: prologue (lane tour, row constants, first distance)", b0, g[0], 1, "once per scan"); tot += t_; add(cls, c_)
    t_, c_ = region("relocate_full: group of six target edges, loads + arithmetic + group filter", g[0], cands[0], 8, "8 groups per wavefront at n = 100"); tot += t_; add(cls, c_)
    cand_end = nxt(cands[0])
    t_, c_ = region("relocate_full: per-step tests behind the group filter", cands[0], cand_end, 0.0,
                    "entered only when the group's minimum beats the lane's best: not in the model (data dependent; the first relocate scan of a descent enters it often)")
    t_, c_ = region("relocate_full: single step (range remainders)", s1[0], [c for c in cands if c > s1[0]][0], 2, "2 per wavefront at n = 100"); tot += t_; add(cls, c_)
    t_, c_ = region("relocate_full: flags (row minimum against the quiet threshold, one LDS atomic per flagged row)", e0, e0 + 12, 1, "once per scan (upper bound: 12 lines)"); tot += t_; add(cls, c_)
    model["relocate_full"] = {"instructions_executed": tot, "by_class": cls, "evaluations_per_scan": (n - 2) * (n - 2),
                              "instructions_per_evaluation": 4 * tot / ((n - 2) * (n - 2))}
    # ---- relocate_a2a, flagged rows: scan_relocate_a2a_quiet ----
    rb = last("quiet_refresh_begin"); re_ = last("quiet_refresh_end"); rl = last("quiet_row_loop")
    rr = last("quiet_row_rare"); rend = last("quiet_scan_end")
    ROWS = 2.4                                           # flagged rows per wavefront and scan with the bench's guide (stamps: 9.6 per workgroup)
    tot, cls = 0, {}
    t_, c_ = region("relocate_flagged: refresh (row constants of the lane's node, 5 pending tour edges for every node)", rb, re_, 1, "once per scan"); tot += t_; add(cls, c_)
    t_, c_ = region("relocate_flagged: target-edge registers of the lanes", re_, rl, 1, "once per scan with flagged rows"); tot += t_; add(cls, c_)
    # (the compiler rotates the loop: the tail mark sits in front of the body; [row_loop, row_rare) is tail + body of one row)
    t_, c_ = region("relocate_flagged: one flagged row, two passes of 64 target edges (common path incl. loop control)", rl, rr, ROWS, f"{ROWS} rows per wavefront (counting pass: flagged rows per scan / 4)"); tot += t_; add(cls, c_)
    t_, c_ = region("relocate_flagged: candidate path of a row", rr, rend, 0.0, "only for rows with a delta below the lane's best: not in the model")
    model["relocate_flagged"] = {"instructions_executed": tot, "by_class": cls, "evaluations_per_scan": ROWS * 4 * n + 5 * (n - 1),
                                 "instructions_per_evaluation": 4 * tot / (ROWS * 4 * n + 5 * (n - 1)), "flagged_rows_per_wavefront": ROWS}
    # ---- two_opt_a2a, pruned: scan_two_opt_a2a_pruned ----
    tb = last("twoopt_pruned_begin"); tp = last("twoopt_pruned_pass"); tl = last("twoopt_pruned_level")
    tc = [i for i, m in marks if m == "twoopt_pruned_candidate" and i > tl][:2]; to = last("twoopt_pruned_overflow")
    tend = nxt(to)
    PASSES, LEVELS = 3.25, 1.3                           # 792 row tasks on 256 threads: wavefront 0 has a fourth pass; second level of 16 entries for ~30 % of the passes
    tot, cls = 0, {}
    t_, c_ = region("two_opt: pass prologue (row of the lane's node: position, neighbours, edge lengths, threshold)", tp, tl, PASSES, f"{PASSES} passes per wavefront (3 or 4)"); tot += t_; add(cls, c_)
    t_, c_ = region("two_opt: level of 16 list entries without its candidate regions", tl, tc[0], PASSES * LEVELS, f"{LEVELS} levels per pass"); tot += t_; add(cls, c_)
    t_, c_ = region("two_opt: candidate evaluation, first entry of a lane", tc[0], tc[1], PASSES * LEVELS, "entered whenever a lane of the wavefront has a candidate: almost every level"); tot += t_; add(cls, c_)
    t_, c_ = region("two_opt: candidate evaluation, second entry of a lane + level tail", tc[1], to, PASSES * LEVELS, ""); tot += t_; add(cls, c_)
    t_, c_ = region("two_opt: overflow rows (all 32 entries inside the threshold)", to, tend, 0.0, "0.23 rows per wavefront and scan: not in the model")
    model["two_opt"] = {"instructions_executed": tot, "by_class": cls, "evaluations_per_scan": 1000.0,
                        "instructions_per_evaluation": 4 * tot / 1000.0,
                        "note": "~1,000 of the 4,753 moves of a scan are evaluated (counting pass: prune_ratio); the instructions are list walking, not evaluations"}
    # ---- workgroup arg-min and move application ----
    ab = last("argmin_lds_begin"); ae = last("argmin_lds_end")
    c = count(lines, ab, ae)
    model["argmin"] = {"instructions_executed": total(c), "by_class": c}
    pb = last("descent_apply_begin"); pe = last("descent_apply_end")
    c = count(lines, pb, pe)
    model["apply"] = {"instructions_static": total(c), "by_class": c, "note": "static count of the region incl. both operators' paths; a thread moves one or two tour positions"}
    for k in ("relocate_full", "relocate_flagged", "two_opt", "argmin"):
        model[k]["single_wavefront_floor_cycles"] = model[k]["instructions_executed"] * SINGLE_WAVE_CPI
        model[k]["issue_model_cycles"] = model[k]["instructions_executed"] * CPI
    model["cycles_per_instruction"] = CPI
    model["single_wavefront_cycles_per_instruction"] = SINGLE_WAVE_CPI
    model["source"] = ("profiles/r06_isa/descent_model.json: scripts/isa_descent_model.py on the -DGLS_ISA_MARKS disassembly of the headline "
                       "instantiation (profiles/r06_isa/headline_descent_scans.s)")
    os.makedirs(os.path.join(ROOT, "profiles", "r06_isa"), exist_ok=True)
    json.dump({f"tsp{n}": model}, open(os.path.join(ROOT, "profiles", "r06_isa", "descent_model.json"), "w"), indent=1)
    with open(os.path.join(ROOT, "profiles", "r06_isa", "descent_regions.txt"), "w") as f:
        for r in out_regions:
            f.write(f"{r['region']}\n    lines {r['lines'][0]}-{r['lines'][1]} of the excerpt source, {r['instructions']} instructions x {r['times_per_scan']} per scan"
                    f"  ({' '.join(f'{k}={v}' for k, v in sorted(r['by_class'].items()))})  {r['note']}\n")
        f.write("\n")
        for k in ("two_opt", "relocate_full", "relocate_flagged", "argmin"):
            m = model[k]
            f.write(f"{k}: {m['instructions_executed']:.0f} instructions per wavefront and scan; single-wavefront floor {m['single_wavefront_floor_cycles']:.0f} cycles, "
                    f"issue model (x {CPI:.2f}) {m['issue_model_cycles']:.0f} cycles"
                    + (f"; {m['instructions_per_evaluation']:.2f} instructions per evaluation (all four wavefronts)" if "instructions_per_evaluation" in m else "") + "\n")
    # the marked regions themselves
    lo = min(tb, rb, b0, ab, pb); hi = max(tend, rend, e0 + 12, ae, pe)
    with open(os.path.join(ROOT, "profiles", "r06_isa", "headline_descent_scans.s"), "w") as f:
        f.write("\n".join(lines[lo:hi]) + "\n")
    print(open(os.path.join(ROOT, "profiles", "r06_isa", "descent_regions.txt")).read())


if __name__ == "__main__":
    main()
