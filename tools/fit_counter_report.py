#!/usr/bin/env python3
"""Condense a rocprofv3 counter pass of tools/critic_fit_probe.py: per k_critic_fit* kernel, the LAST `--last` dispatches
(the steady state of the closed loop): duration, waves, VALU instructions per wave, issue-slot occupancy, wait fractions.

    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU \
              SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d gpurun_out/prof_fit -o f -- \
              python3 tools/critic_fit_probe.py quadratic
    python tools/fit_counter_report.py gpurun_out/prof_fit
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
last = int(sys.argv[2]) if len(sys.argv) > 2 else 15
f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)[0]
rows = defaultdict(lambda: defaultdict(dict))  # kernel -> dispatch id -> counter -> value
meta = {}
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "k_critic_fit" not in k:
        continue
    did = int(r["Dispatch_Id"])
    rows[k][did][r["Counter_Name"]] = rows[k][did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    meta[(k, did)] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("VGPR_Count"), r.get("LDS_Block_Size"),
                      r.get("Grid_Size"), r.get("Workgroup_Size"))
out = {}
for k, disp in rows.items():
    ids = sorted(disp)[-last:]
    n = len(ids)
    acc = defaultdict(float)
    dur = 0.0
    for i in ids:
        for c, v in disp[i].items():
            acc[c] += v / n
        dur += meta[(k, i)][0] / n
    e = {"dispatches": n, "avg_us": dur / 1e3, "vgpr": meta[(k, ids[-1])][1], "lds": meta[(k, ids[-1])][2],
         "grid": meta[(k, ids[-1])][3], "wg": meta[(k, ids[-1])][4]}
    e.update({c: v for c, v in acc.items()})
    if acc.get("SQ_WAVES"):
        e["valu_per_wave"] = acc["SQ_INSTS_VALU"] / acc["SQ_WAVES"]
    if acc.get("SQ_INSTS_VALU"):
        e["issue_slot_occupancy"] = acc["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * dur * 1e-9)
    if acc.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            if c in acc:
                e[c + "_over_WAVE_CYCLES"] = acc[c] / acc["SQ_WAVE_CYCLES"]
    out[k.split("(")[0][:90]] = e
print(json.dumps(out, indent=1, sort_keys=True))
