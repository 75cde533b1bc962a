#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small files kept under profiles/.

    python tools/prof_summary.py --round r01 --kt gpurun_out/prof_kt --fetch gpurun_out/prof_fetch \
        --write gpurun_out/prof_write --key k_actor_streamed_B65536_K256_N10_f32

Writes profiles/<round>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, kernel names shortened),
profiles/<round>_pmc.json (per-kernel FETCH_SIZE / WRITE_SIZE per launch, separate --pmc passes) and
updates profiles/pmc_traffic.json, which bench.py reads for `roofline.traffic`.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB;
FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read (128-B requests tallied
at 64 B), so reads = FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact for streaming stores.
"""
import argparse
import csv
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"\(.*$", "", name)  # drop the argument list
    name = name.replace("void ", "")
    if len(name) > 110:
        name = name[:107] + "..."
    return name


def one(pattern):
    g = glob.glob(pattern, recursive=True)
    return g[0] if g else None


def counters(d, counter):
    f = one(os.path.join(d, "**", "*_counter_collection.csv"))
    out = {}
    if not f:
        return out
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        e = out.setdefault(k, {"n": 0, "sum": 0.0, "dur_ns": 0, "vgpr": r.get("VGPR_Count"), "sgpr": r.get("SGPR_Count"),
                               "grid": r.get("Grid_Size"), "wg": r.get("Workgroup_Size")})
        e["n"] += 1
        e["sum"] += float(r["Counter_Value"])
        e["dur_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", required=True)
    ap.add_argument("--kt")
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--key", help="bench.py workload key for pmc_traffic.json (k_actor entry)")
    ap.add_argument("--valu", help="rocprofv3 --pmc SQ_* pass of tools/valu_probe.py (directory)")
    ap.add_argument("--valu-units", help="the JSON line tools/valu_probe.py printed in that pass (file)")
    ap.add_argument("--kernel", default="rcg::k_actor")
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    tag = f"_{a.tag}" if a.tag else ""

    if a.kt:
        f = one(os.path.join(a.kt, "**", "*_kernel_stats.csv"))
        rows = list(csv.DictReader(open(f)))
        dst = os.path.join(ROOT, "profiles", f"{a.round}{tag}_kernel_stats.csv")
        with open(dst, "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                            r["MinNs"], r["MaxNs"], r["StdDev"]])
        print("wrote", dst)

    pmc = {}
    if a.fetch:
        for k, e in counters(a.fetch, "FETCH_SIZE").items():
            pmc.setdefault(k, {})
            pmc[k].update(launches_fetch_pass=e["n"], FETCH_SIZE_KiB_per_launch=e["sum"] / e["n"],
                          read_bytes_per_launch=e["sum"] / e["n"] * 1024 * 2,
                          avg_ns_fetch_pass=e["dur_ns"] / e["n"], vgpr=e["vgpr"], sgpr=e["sgpr"], grid=e["grid"], wg=e["wg"])
    if a.write:
        for k, e in counters(a.write, "WRITE_SIZE").items():
            pmc.setdefault(k, {})
            pmc[k].update(launches_write_pass=e["n"], WRITE_SIZE_KiB_per_launch=e["sum"] / e["n"],
                          write_bytes_per_launch=e["sum"] / e["n"] * 1024, avg_ns_write_pass=e["dur_ns"] / e["n"])
    if pmc:
        for k, e in pmc.items():
            e["hbm_bytes_per_launch"] = e.get("read_bytes_per_launch", 0.0) + e.get("write_bytes_per_launch", 0.0)
        pmc = {k: v for k, v in pmc.items() if k.startswith("rcg::")}
        dst = os.path.join(ROOT, "profiles", f"{a.round}{tag}_pmc.json")
        json.dump({"corrections": "bytes = KiB*1024; reads doubled (gfx950 FETCH_SIZE counts 128-B requests at 64 B)",
                   "kernels": pmc}, open(dst, "w"), indent=1, sort_keys=True)
        print("wrote", dst)
        if a.key:
            tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            t = json.load(open(tf)) if os.path.exists(tf) else {}
            hit = [v for k, v in pmc.items() if k.startswith(a.kernel)]
            if hit:
                t[a.key] = {"hbm_bytes_per_launch": hit[0]["hbm_bytes_per_launch"], "round": a.round,
                            "read_bytes": hit[0].get("read_bytes_per_launch"), "write_bytes": hit[0].get("write_bytes_per_launch")}
                json.dump(t, open(tf, "w"), indent=1, sort_keys=True)
                print("updated", tf, a.key, t[a.key])


    if a.valu:
        valu_summary(a)


VALU_PEAK_SURVEY = 7.9e13  # lane-instr/s, SURVEY.md 8d (157.3 TFLOP/s FMA / 2: counts packed issue)
N_SIMD, CLK = 256 * 4, 2.4e9
SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_VALU_TRANS_F32", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES",
               "SQ_ACTIVE_INST_VALU", "SQ_WAIT_INST_ANY", "SQ_INSTS_SALU")


def valu_summary(a):
    """profiles/<round>_valu_pmc.json: per kernel, per launch - wave-level VALU instructions (SQ_INSTS_VALU), waves,
    duration inside the counter pass; derived: lane-instructions/s = SQ_INSTS_VALU x 64 / duration against the SURVEY's
    7.9e13 peak, and the issue-slot occupancy (4 cycles per wave64 instruction, 8 for v_sin/v_cos/v_rcp...) of the
    1024 SIMDs at 2.4 GHz.  profiles/valu_instr.json: VALU instructions per _actor_cost evaluation for bench.py."""
    per = {}
    for c in SQ_COUNTERS:
        for k, e in counters(a.valu, c).items():
            if not k.startswith("rcg::"):
                continue
            d = per.setdefault(k, {"launches": e["n"], "avg_ns_pmc_pass": e["dur_ns"] / e["n"], "vgpr": e["vgpr"],
                                   "grid": e["grid"], "wg": e["wg"]})
            d[c + "_per_launch"] = e["sum"] / e["n"]
    for k, d in per.items():
        iv, it = d.get("SQ_INSTS_VALU_per_launch"), d.get("SQ_INSTS_VALU_TRANS_F32_per_launch", 0.0)
        if not iv:
            continue
        sec = d["avg_ns_pmc_pass"] * 1e-9
        d["lane_instr_per_s"] = iv * 64 / sec
        d["frac_of_7.9e13_lane_instr_per_s"] = iv * 64 / sec / VALU_PEAK_SURVEY
        d["issue_slot_occupancy"] = (iv * 4 + it * 4) / (N_SIMD * CLK * sec)  # transcendental: 8 cycles = 4 extra
        d["valu_instr_per_wave"] = iv / d["SQ_WAVES_per_launch"] if d.get("SQ_WAVES_per_launch") else None
    units = {}
    if a.valu_units and os.path.exists(a.valu_units):
        for line in open(a.valu_units):
            if line.startswith("{"):
                units = json.loads(line).get("units_per_launch", {})
    dst = os.path.join(ROOT, "profiles", f"{a.round}_valu_pmc.json")
    json.dump({"note": "SQ counters are summed over the 8 XCDs; instructions are wave-level (x64 lanes); durations are "
                       "those of the counter pass (profiled clocks run a few % lower than un-profiled ones)",
               "kernels": per, "units_per_launch": units}, open(dst, "w"), indent=1, sort_keys=True)
    print("wrote", dst)
    # instructions per evaluation for the generated-grid rollout of the bench workload (k_actor, Sys3WRobot, f32, GENERIC=0)
    vi = os.path.join(ROOT, "profiles", "valu_instr.json")
    t = json.load(open(vi)) if os.path.exists(vi) else {}
    def gen_kernel(sysname, tgt):
        """The kernel a generated-grid tick of this system runs (round 4: the fused k_ticks_pk for the robots' preset
        weights, else the hand-packed k_actor instance, else the general one): first prefix that has launches."""
        for pre in (f"rcg::k_ticks_pk<rcg::{sysname}>", f"rcg::k_actor<rcg::{sysname}, float, false, {tgt}, false, true>",
                    f"rcg::k_actor<rcg::{sysname}, float, false, {tgt}, false"):
            hh = [(k, d) for k, d in per.items() if k.startswith(pre) and d.get("SQ_INSTS_VALU_per_launch")]
            if hh:
                return hh
        return []

    # the mixed pool of configs[4] (valu_probe.py pool): one generated-grid kernel per system type, K = 256, N = 15
    for sysname, tgt in (("Sys3WRobot", "false"), ("Sys3WRobotNI", "false"), ("Sys2Tank", "true")):
        key = {"Sys3WRobot": "3wrobot", "Sys3WRobotNI": "3wrobotNI", "Sys2Tank": "2tank"}[sysname]
        uu = units.get(f"k_actor_generated_{key}_N15_f32_C5")
        hh = gen_kernel(sysname, tgt)
        if uu and hh:
            k, d = hh[0]
            t[f"k_actor_generated_{key}_N15_f32_C5"] = {
                "valu_instr_per_eval": d["SQ_INSTS_VALU_per_launch"] * 64 / uu["evals"], "round": a.round, "kernel": k,
                "launches": d["launches"]}
    # configs[2] generated (2tank, N = 20): MPC on the specialised instance, RQL / SQL on the generic one - priced only from a
    # pass in which the instance ran ONE of them (valu_probe.py c3rql / c3sql), recognised by the launch count
    launches_each = 0
    if a.valu_units and os.path.exists(a.valu_units):
        for line in open(a.valu_units):
            if line.startswith("{"):
                launches_each = json.loads(line).get("launches_each", 0)
    for mode, gen in (("MPC", "false"), ("RQL", "true"), ("SQL", "true")):
        key = f"k_actor_generated_2tank_N20_{mode}_f32"
        uu = units.get(key)
        hh = [(k, d) for k, d in per.items() if k.startswith(f"rcg::k_actor<rcg::Sys2Tank, float, {gen}, true, false")]
        if uu and hh and hh[0][1]["launches"] == launches_each:
            k, d = hh[0]
            t[key] = {"valu_instr_per_eval": d["SQ_INSTS_VALU_per_launch"] * 64 / uu["evals"], "round": a.round, "kernel": k,
                      "launches": d["launches"]}
    hit = [(k, d) for k, d in gen_kernel("Sys3WRobot", "false") if d["launches"] == launches_each] if launches_each else []
    u = units.get("k_actor_generated_3wrobot_N10_f32")
    if t:
        json.dump(t, open(vi, "w"), indent=1, sort_keys=True)
    if hit and u:
        k, d = hit[0]
        # the probe launches this instance for the C2 shape AND (N = 15) for the mixed pool: use the per-wave figure of
        # the C2 grid (one wave = 64 evaluations) only when the launch counts say the kernel ran the C2 shape alone
        t["k_actor_generated_3wrobot_N10_f32"] = {"valu_instr_per_eval": d["SQ_INSTS_VALU_per_launch"] * 64 / u["evals"],
                                                  "round": a.round, "kernel": k, "launches": d["launches"]}
        json.dump(t, open(vi, "w"), indent=1, sort_keys=True)
        print("updated", vi, t["k_actor_generated_3wrobot_N10_f32"])


if __name__ == "__main__":
    main()
