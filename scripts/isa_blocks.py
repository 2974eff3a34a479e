#!/usr/bin/env python
"""Basic blocks of a disassembly excerpt with instruction counts by class (helper for profiles/r05_isa/: the dependent-chain /
issue table of the penalty step).  usage: python scripts/isa_blocks.py pert.s"""
import re
import sys

CLASSES = [("dpp", r"_dpp\b"), ("div", r"v_div_(scale|fmas|fixup)_f64|v_rcp_f64"), ("readlane", r"v_readlane|v_readfirstlane|v_writelane"),
           ("cndmask_vcc", r"v_cndmask_b32_e32"), ("cndmask_sgpr", r"v_cndmask_b32_e64"), ("vmem", r"^(buffer_|global_|scratch_)"),
           ("lds", r"^ds_"), ("smem", r"^s_load|^s_memrealtime|^s_memtime"), ("branch", r"^s_cbranch|^s_branch"),
           ("waitcnt", r"^s_waitcnt"), ("nop", r"^s_nop"), ("valu", r"^v_"), ("salu", r"^s_")]


def classify(op):
    for name, pat in CLASSES:
        if re.search(pat, op):
            return name
    return "other"


def main():
    blocks, cur = [], {"label": "entry", "ins": [], "marks": []}
    for line in open(sys.argv[1]):
        t = line.strip()
        if not t:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append(cur)
            cur = {"label": m.group(1), "ins": [], "marks": []}
            continue
        if t.startswith("; GLSMARK"):
            cur["marks"].append(t.split()[2])
            continue
        if t.startswith(";") or t.startswith("."):
            continue
        cur["ins"].append(t.split(";")[0].strip())
    blocks.append(cur)
    for b in blocks:
        cnt = {}
        for i in b["ins"]:
            c = classify(i.split()[0])
            cnt[c] = cnt.get(c, 0) + 1
        br = [i for i in b["ins"] if i.startswith("s_cbranch") or i.startswith("s_branch")]
        print(f"{b['label']:12s} n={len(b['ins']):4d} {' '.join(f'{k}={v}' for k, v in sorted(cnt.items())):70s} marks={','.join(b['marks'])} -> {'; '.join(br)}")


if __name__ == "__main__":
    main()
