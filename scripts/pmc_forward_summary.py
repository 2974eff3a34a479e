#!/usr/bin/env python
"""Summarise scripts/pmc_forward.sh: per forward kernel the matrix-pipe occupancy (SQ_VALU_MFMA_BUSY_CYCLES against the
kernel's own SIMD cycles), vector-ALU and LDS busy fractions and HBM-side bytes per line-graph node.

    python scripts/pmc_forward_summary.py profiles/r03_pmc_forward [rows_per_launch]
SQ cycle counters tick once per 4 shader cycles and are summed over the device; a kernel's SIMD-cycle budget is
1024 SIMDs x duration x clock (clock = GRBM_GUI_ACTIVE / 8 XCDs / duration)."""
import csv
import json
import os
import sys
from collections import defaultdict


def read(path):
    rows = defaultdict(lambda: defaultdict(float))
    span = defaultdict(float)
    launches = defaultdict(int)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = next(n for n in ("ffn_fused_bf16x3_kernel", "ffn_pack_bf16x3_kernel", "ffn_fused_kernel", "gat_rows_kernel", "gemm_f32_kernel") if n in r["Kernel_Name"])
            rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
                span[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
                launches[k] += 1
    return rows, span, launches


def main():
    d = sys.argv[1]
    rows_per_launch = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 512 * 4950.0
    out = {}
    mf, t_m, _ = read(os.path.join(d, "mfma_counter_collection.csv"))
    va, t_v, _ = read(os.path.join(d, "valu_counter_collection.csv"))
    fe, t_f, n_f = read(os.path.join(d, "fetch_counter_collection.csv"))
    wr, t_w, n_w = read(os.path.join(d, "write_counter_collection.csv"))
    for k in mf:
        clock = mf[k]["GRBM_GUI_ACTIVE"] / 8 / t_m[k]
        simd_cycles = 1024 * t_m[k] * clock
        clock_v = va[k]["GRBM_GUI_ACTIVE"] / 8 / t_v[k]
        out[k] = {
            # (this counter is in shader cycles, not in the 4-cycle ticks of the SQ_ACTIVE_* / SQ_BUSY_* counters: with x4 the
            # fused FFN would be 3.06 of its own SIMD cycles; as it is, 0.765 -- its algorithmic 0.755 of the fp32 MFMA peak)
            "mfma_busy_frac": mf[k]["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
            "wave_wait_frac": mf[k]["SQ_WAIT_ANY"] / mf[k]["SQ_WAVE_CYCLES"],
            "valu_busy_frac": va[k]["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * t_v[k] * clock_v),
            "lds_busy_frac": va[k]["SQ_LDS_IDX_ACTIVE"] / (256 * t_v[k] * clock_v),
            "lds_bank_conflict_frac": va[k]["SQ_LDS_BANK_CONFLICT"] / max(va[k]["SQ_LDS_IDX_ACTIVE"], 1.0),
            "valu_insts_per_mfma": va[k]["SQ_INSTS_VALU"] / max(va[k]["SQ_INSTS_MFMA"], 1.0),
            "hbm_bytes_per_row": (fe[k]["FETCH_SIZE"] / n_f[k] + wr[k]["WRITE_SIZE"] / n_w[k]) * 1024 / rows_per_launch,
            "clock_ghz": clock / 1e9,
        }
        out[k]["avg_launch_ms"] = t_f[k] / max(n_f[k], 1) * 1e3
    out["source"] = f"{d}/*_counter_collection.csv (scripts/pmc_forward.sh: predict_regret, 512 TSP100 instances)"
    print(json.dumps(out, indent=1))
    if "--write" in sys.argv:
        # merge into profiles/traffic_<round>.json under the names bench.py uses (round from the directory name: r04_pmc_forward)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        rel = os.path.relpath(d, root)
        rnd = os.path.basename(os.path.normpath(d)).split("_")[0][:3]
        path = os.path.join(root, "profiles", f"traffic_{rnd}.json")
        table = json.load(open(path)) if os.path.isfile(path) else {}
        for name, kern in (("ffn_fused", "ffn_fused_kernel"), ("ffn_fused", "ffn_fused_bf16x3_kernel"), ("gemm_fc", "gemm_f32_kernel"), ("gat_aggregate", "gat_rows_kernel")):
            if kern in out:
                table[name] = dict(out[kern], source=f"{rel}/*_counter_collection.csv (scripts/pmc_forward.sh: predict_regret, 512 TSP100 instances)")
        json.dump(table, open(path, "w"), indent=1)
        print("updated", path, file=sys.stderr)


if __name__ == "__main__":
    main()
