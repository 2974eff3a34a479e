#!/usr/bin/env python
"""Issue / dependent-chain model of ONE penalty step of gls_kernel's serial perturbation phase (algorithms.py:150-185) from the
committed disassembly of the headline instantiation gls_kernel<TriDGlobalP,false,2,false,4,false,false> and the per-instruction
constants measured by scripts/isa_probe/latency_probe.hip (profiles/r05_isa_latency_probe.txt).

    python scripts/isa_critical_path.py            -> profiles/r05_isa/critical_path.json + the tables on stdout

The disassembly (profiles/r05_isa/headline_edge_form_perturbation_phase.s: `s_setprio 3` .. `s_setprio 0` of a -DGLS_ISA_MARKS
-DGLS_DEV_ONLY_HEADLINE build; the marks are comment lines) is cut into basic blocks; PATH says how often a penalty step of the
bench's workload (TSP100 x 1024, regret_pred guide: 4 one-to-all scans, 1.2 accepted moves per step, the first scan accepts in
93 % of the steps) executes each block and which share of its instructions lies on the executed side of its internal branches.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from isa_blocks import classify  # noqa: E402

# cycles per instruction of ONE wavefront alone on its SIMD (latency_probe, first table); fp64 and integer VALU cost the same
COST = {"valu": 4.3, "salu": 4.6, "cndmask_sgpr": 4.3, "cndmask_vcc": 13.3, "readlane": 4.5, "dpp": 10.5, "div": 4.9, "nop": 1.5,
        "waitcnt": 1.0, "branch": 16.0, "lds": 4.5, "vmem": 4.5, "smem": 4.6, "other": 4.5}
L2_ROUND_TRIP, LDS_ROUND_TRIP = 270.0, 64.0       # buffer_load_dword L2 hit / ds_read round trip, idle chip

# block label -> (executions per penalty step, executed share of the block's instructions, what it is).  The four scans of a step
# are unrolled (one block each, with its acceptance code); "#slow" = the part of a scan block behind "some lane has a negative
# delta" (np.isclose, per-lane strict <, DPP arg-min without its tie path: ~56 of the ~80 instructions of that side)
PATH = {
    ".LBB5_564": (1.0, 1.0, "step latch"), ".LBB5_515": (1.0, 1.0, "loop top"),
    ".LBB5_516": (1.0, 0.58, "arg-max: two fp64 divisions, key image, DPP min over the high words (the tie path, 42 % of the block, is rare)"),
    ".LBB5_519": (1.0, 1.0, "arg-max: position of the single winner (s_ff1 + v_readlane)"),
    ".LBB5_521": (1.0, 0.646, "penalise (counter + 1 by its owner lane, one buffer store) + two_opt_o2a scan of endpoint 0 + acceptance, fast side"),
    ".LBB5_521#slow": (0.93, 0.248, "acceptance of scan 0, slow side (the first scan accepts in 93 % of the steps)"),
    ".LBB5_526": (0.93, 1.0, "arg-min: key of the single winner"),
    ".LBB5_528": (0.93, 1.0, "move after scan 0: scalar move parameters, new (u, v) from the old tour in LDS + DPP neighbour, tour written, loads issued"),
    ".LBB5_529": (0.07, 1.0, "scan 0 without a move"),
    ".LBB5_531": (1.0, 0.626, "relocate_o2a scan of endpoint 0 (2 slots x 2 guided values + G[a,c] of a uniform pair) + acceptance, fast side"),
    ".LBB5_531#slow": (0.06, 0.262, "acceptance of scan 1, slow side"), ".LBB5_535": (0.06, 1.0, "arg-min"), ".LBB5_537": (0.06, 1.0, "move after scan 1"),
    ".LBB5_539": (1.0, 1.0, "endpoint 1: index"),
    ".LBB5_542": (1.0, 0.59, "two_opt_o2a scan of endpoint 1 + acceptance, fast side"),
    ".LBB5_542#slow": (0.15, 0.287, "acceptance of scan 2, slow side"), ".LBB5_546": (0.15, 1.0, "arg-min"), ".LBB5_548": (0.15, 1.0, "move after scan 2"),
    ".LBB5_550": (1.0, 0.631, "relocate_o2a scan of endpoint 1 + acceptance, fast side"),
    ".LBB5_550#slow": (0.06, 0.258, "acceptance of scan 3, slow side"), ".LBB5_555": (0.06, 1.0, "arg-min"), ".LBB5_557": (0.06, 1.0, "move after scan 3"),
    ".LBB5_559": (1.0, 1.0, "step tail"), ".LBB5_561": (1.0, 1.0, "step tail"),
}
# exposed memory round trips per step (not covered by independent instructions of the same wavefront): per scan the LAST of its
# loads is issued ~25 instructions (110 cycles) before the wait; per move one LDS round trip (the old tour's nodes)
EXPOSED = [("counter / distance loads of a scan, L2 hit, last load issued ~110 cycles before its wait", 4.0, L2_ROUND_TRIP - 110.0),
           ("old-tour read of a move (LDS)", 1.2, LDS_ROUND_TRIP)]
# the dependent chain of a step by itself (infinite issue width): instructions ON the chain at their dependent latency
CHAIN = [
    ("arg-max", 1.0, [("v_cvt + v_add (1 + count)", 2 * 4.3), ("fp64 division", 68.5), ("slot select", 4.3), ("key image: 5 VALU", 5 * 4.3),
                      ("no-edge select", 4.3), ("DPP min, 6 steps", 6 * 12.0), ("v_readlane + hazard", 8.0), ("v_cmp_eq e64", 4.5),
                      ("tie test: 4 SALU + branch", 4 * 4.6 + 16.0), ("s_ff1 + v_readlane", 10.0)]),
    ("penalise", 1.0, [("2 v_readlane + s_cselect", 14.0), ("packed offset: 5 SALU", 5 * 4.6), ("v_mov, select, buffer_store issue", 13.0)]),
    ("one-to-all scan", 4.0, [("a = t[i]: v_readlane + s_cselect", 9.0), ("row offset of a: 3 SALU", 3 * 4.6), ("pair offset: v_cmp, v_add, select", 3 * 4.3),
                              ("buffer_load_dword, L2 hit", L2_ROUND_TRIP), ("G = D + k P: cvt, mul, add", 3 * 4.3), ("delta: 3 fp64 adds", 3 * 4.3),
                              ("v_cmp_lt, s_and, s_or", 4.5 + 2 * 4.6), ("s_cmp + branch", 4.6 + 16.0)]),
    ("acceptance, slow side", 1.2, [("isclose: mul, add, cmp", 3 * 4.3), ("2 s_and + select", 2 * 4.6 + 4.3), ("any-candidate test + branch", 4.5 + 16.0),
                                   ("order-preserving image: 6 VALU", 6 * 4.3), ("DPP min, 6 steps", 6 * 12.0), ("single-winner test + 2 v_readlane", 4 * 4.6 + 16.0 + 9.0)]),
    ("move", 1.2, [("move parameters: 8 SALU", 8 * 4.6), ("source position: sub, cmp, mad, 2 selects", 5 * 4.3), ("ds_read_u8 round trip", LDS_ROUND_TRIP),
                   ("row offset + pair offset of the new edge: 5 VALU", 5 * 4.3)]),
    ("loop control", 1.0, [("step latch + scan loop latches", 50.0)]),
]


def main():
    src = os.path.join(ROOT, "profiles", "r05_isa", "headline_edge_form_perturbation_phase.s")
    blocks, cur = {}, None
    for line in open(src):
        t = line.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if cur is None or not t or t.startswith(";") or t.startswith("."):
            continue
        blocks[cur].append(t.split(";")[0].strip().split()[0])
    rows, total_ins, total_cyc = [], 0.0, 0.0
    for key, (mult, share, what) in PATH.items():
        label = key.split("#")[0]
        ins = blocks[label]
        cnt = {}
        for op in ins:
            c = classify(op)
            cnt[c] = cnt.get(c, 0) + 1
        n_exec = len(ins) * share * mult
        cyc = sum(COST[c] * v for c, v in cnt.items()) * share * mult
        rows.append({"block": key, "what": what, "static_instructions": len(ins), "executed_share": share, "per_step": mult,
                     "instructions_per_step": n_exec, "issue_cycles_per_step": cyc, "classes": cnt})
        total_ins += n_exec
        total_cyc += cyc
    exposed = sum(m * c for _, m, c in EXPOSED)
    chain = sum(m * sum(c for _, c in items) for _, m, items in CHAIN)
    out = {"tsp100": {
        "instantiation": "gls_kernel<TriDGlobalP,false,2,false,4,false,false> (TSP100 x 1024, compact store, two register slots per lane)",
        "instructions_per_step": total_ins, "issue_cycles_per_step": total_cyc, "exposed_memory_cycles_per_step": exposed,
        "issue_model_cycles_per_step": total_cyc + exposed, "chain_floor_cycles_per_step": chain,
        "measured_cycles_per_step_product_kernel_r05c": 4883,
        "blocks": rows, "exposed": [{"what": w, "per_step": m, "cycles_each": c} for w, m, c in EXPOSED],
        "chain": [{"stage": s, "per_step": m, "cycles_each": sum(c for _, c in items), "items": [{"what": w, "cycles": c} for w, c in items]}
                  for s, m, items in CHAIN],
        "constants": {"cost_per_instruction_class": COST, "l2_round_trip": L2_ROUND_TRIP, "lds_round_trip": LDS_ROUND_TRIP,
                      "source": "profiles/r05_isa_latency_probe.txt (one wavefront per SIMD)"},
        "source": "scripts/isa_critical_path.py on profiles/r05_isa/headline_edge_form_perturbation_phase.s"}}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r05_isa", "critical_path.json"), "w"), indent=1)
    print(f"{'block':16s} {'x/step':>6s} {'instr/step':>10s} {'cycles/step':>11s}  what")
    for r in rows:
        print(f"{r['block']:16s} {r['per_step']:6.1f} {r['instructions_per_step']:10.1f} {r['issue_cycles_per_step']:11.0f}  {r['what']}")
    print(f"{'sum':16s} {'':6s} {total_ins:10.1f} {total_cyc:11.0f}  + exposed memory round trips {exposed:.0f} = issue model {total_cyc + exposed:.0f} cycles per step")
    print()
    for s, m, items in CHAIN:
        print(f"chain: {s:24s} x{m:3.1f}  {sum(c for _, c in items):6.0f} cycles each: " + ", ".join(f"{w} {c:.0f}" for w, c in items))
    print(f"dependent-chain floor {chain:.0f} cycles per step")


if __name__ == "__main__":
    main()
