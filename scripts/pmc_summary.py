#!/usr/bin/env python
"""Summarise the PMC passes of scripts/pmc_gls.sh into the `gls_kernel` entry of profiles/traffic_r02.json (the file
bench.py reads for roofline.pmc / roofline.traffic).

    python scripts/pmc_summary.py profiles/r02c_pmc [--write]

Counters (per launch of gls_kernel; one launch = 1024 TSP100 instances for ~2 s):
  FETCH_SIZE, WRITE_SIZE     memory-side traffic in KiB.  The gfx950 x2 correction of FETCH_SIZE is NOT applied here: this
                             kernel's reads are narrow scattered 4-byte penalty loads served from L2 (the figure is an upper
                             bound of HBM traffic either way: < 1 % of 8 TB/s, with or without the factor)
  GRBM_GUI_ACTIVE / 8        shader clock x time (the counter is summed over the 8 XCDs)
  SQ_ACTIVE_INST_VALU x 4    cycles the vector ALUs are busy (SQ cycle counters tick once per 4 cycles), against
                             1024 SIMDs x time x clock
  SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT   LDS pipe active cycles (against 256 CUs x time x clock) and conflict replays
  SQ_WAIT_ANY / SQ_WAVE_CYCLES   fraction of their resident cycles the waves spend waiting
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read(path):
    out, span = {}, None
    with open(path) as f:
        for row in csv.DictReader(f):
            if "gls_kernel" not in row["Kernel_Name"]:
                continue
            out[row["Counter_Name"]] = out.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            span = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9
    return out, span


def main():
    d = sys.argv[1]
    fetch, t_f = read(os.path.join(d, "fetch_counter_collection.csv"))
    write, t_w = read(os.path.join(d, "write_counter_collection.csv"))
    lds, t_l = read(os.path.join(d, "lds_counter_collection.csv"))
    iss, t_i = read(os.path.join(d, "issue_counter_collection.csv"))
    wl_path = os.path.join(d, "workload.json")            # written by scripts/pmc_gls.sh since round 4; before: TSP100 x 1024, noise
    workload = json.load(open(wl_path)) if os.path.isfile(wl_path) else None
    instances, simds, cus, xcds = (workload["instances"] if workload else 1024), 1024, 256, 8
    clock = lds["GRBM_GUI_ACTIVE"] / xcds / t_l                      # the counter is summed over the 8 XCDs
    fetch_b = fetch["FETCH_SIZE"] * 1024                              # KiB as reported (see the module docstring)
    write_b = write["WRITE_SIZE"] * 1024
    hbm_per_inst_s = (fetch_b / t_f + write_b / t_w) / instances
    e = {
        "hbm_bytes_per_instance_second": hbm_per_inst_s,
        "hbm_gbs": hbm_per_inst_s * instances / 1e9,
        "hbm_gbs_fetch_doubled": (2 * fetch_b / t_f + write_b / t_w) / 1e9,      # upper bound: the guide's x2 applied anyway
        "lds_busy_frac": lds["SQ_LDS_IDX_ACTIVE"] / (cus * t_l * clock),
        "lds_bank_conflict_frac": lds["SQ_LDS_BANK_CONFLICT"] / lds["SQ_LDS_IDX_ACTIVE"],
        "valu_busy_frac": iss["SQ_ACTIVE_INST_VALU"] * 4 / (simds * t_i * clock),     # SQ cycle counters tick once per 4 cycles
        "valu_insts_per_s": iss["SQ_INSTS_VALU"] / t_i,
        "salu_insts_per_s": iss["SQ_INSTS_SALU"] / t_i,
        "lds_insts_per_s": lds["SQ_INSTS_LDS"] / t_l,
        "vmem_rd_insts_per_s": iss["SQ_INSTS_VMEM_RD"] / t_i,
        "wave_wait_frac": lds["SQ_WAIT_ANY"] / lds["SQ_WAVE_CYCLES"],
        "clock_ghz": clock / 1e9,
        "source": f"{os.path.relpath(d, ROOT)}/*_counter_collection.csv: rocprofv3 --kernel-trace --pmc <counters> -- python3 "
                  + ("scripts/probe_gls.py %d %d 2.0 0 %s" % (workload["n"], workload["instances"], workload["guide"]) if workload
                     else "scripts/probe_gls.py 100 1024 2.0 0 noise") +
                  " (4 separate passes: FETCH_SIZE; WRITE_SIZE; SQ LDS set; SQ issue "
                  "set), scripts/pmc_gls.sh, summarised by scripts/pmc_summary.py",
    }
    if workload:
        e["workload"] = {k: workload[k] for k in ("n", "instances", "guide")}
        # measured LDS-pipe fraction by instruction count: LDS wave-instructions issued per second x 2 array cycles (ds_read_b64 /
        # b32, conflict-free: MI355X_MICROARCH.md LDS table) against the CUs' LDS cycles -- the conflict-free floor of lds_busy_frac
        e["lds_insts_floor_frac"] = lds["SQ_INSTS_LDS"] * 2 / (cus * t_l * clock)
    print(json.dumps(e, indent=1))
    if "--write" in sys.argv:
        # profiles/traffic_<round>.json, the round taken from the directory name (r03_pmc -> traffic_r03.json)
        rnd = os.path.basename(os.path.normpath(d)).split("_")[0][:3]
        p = os.path.join(ROOT, "profiles", f"traffic_{rnd}.json")
        t = json.load(open(p)) if os.path.isfile(p) else {}
        # the headline workload (TSP100 x 1024) is the `gls_kernel` entry; other workloads get a key of their own
        key = "gls_kernel" if not workload or (workload["n"], workload["instances"]) == (100, 1024) else \
            "gls_kernel@tsp%dx%d" % (workload["n"], workload["instances"])
        t[key] = e
        json.dump(t, open(p, "w"), indent=1)
        print("updated", p)


if __name__ == "__main__":
    main()
