#!/usr/bin/env python3
"""tools/fold_mid.py TAG OUT KERNEL VOICES [FRAMES] -- fold the passes of tools/profile_mid.sh (gpurun_out/prof_<TAG>/: a kernel trace
and three SQ counter passes of one bench configuration) into profiles/<OUT>_kernel_stats.csv and profiles/<OUT>_pmc_summary.json:
per-dispatch counter means of KERNEL over the timed launches, and the shares of a wave's lifetime they imply
(SQ_ACTIVE_INST_ANY + SQ_WAIT_ANY + SQ_WAIT_INST_ANY ~ SQ_WAVE_CYCLES; /opt/skills/guides/MI355X_MICROARCH.md, rocprofv3 PMC slots).
Runs here, after gpurun merged gpurun_out/ back."""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, out, kernel, voices = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    frames = int(sys.argv[5]) if len(sys.argv) > 5 else 512
    d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = [f for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True) if kernel in open(f).read()]
    if stats:
        shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{out}_kernel_stats.csv"))
    # the timed dispatches of the trace, one by one (the --stats average also covers the recipe's warm-up launches)
    durs = []
    for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if kernel in r["Kernel_Name"]]
        durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows][-50:] or durs
    vals = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "sq*", "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(dict)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for k in sorted(per)[-5:]:
            for c, v in per[k].items():
                vals[c].append(v)
    avg = {c: sum(v) / len(v) for c, v in vals.items()}
    summ = {"kernel": kernel, "voices": voices, "frames_per_launch": frames,
            "command": f"tools/profile_mid.sh {tag} (rocprofv3 --kernel-trace --stats, then three --pmc passes of SQ counters, each its own run of "
                       "bench.py --steps 5 --warmup 5 --no-cpu --no-extra --time-every 1)",
            "counters_mean_per_dispatch": avg}
    if durs:
        summ["kernel_ns_last50"] = {"mean": sum(durs) / len(durs), "min": min(durs), "n": len(durs)}
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        summ["share_of_wave_lifetime"] = {k: avg[c] / wc for k, c in (("instruction_active", "SQ_ACTIVE_INST_ANY"), ("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"),
                                                                       ("issue_stalled", "SQ_WAIT_INST_ANY"), ("valu_active", "SQ_ACTIVE_INST_VALU"),
                                                                       ("lds_active", "SQ_ACTIVE_INST_LDS")) if c in avg}
    if "SQ_WAVES" in avg:
        w = avg["SQ_WAVES"]
        summ["waves"] = w
        summ["per_wave_frame"] = {k: avg[c] / w / frames for k, c in (("valu_insts", "SQ_INSTS_VALU"), ("salu_insts", "SQ_INSTS_SALU"), ("lds_insts", "SQ_INSTS_LDS")) if c in avg}
    if "SQ_LDS_IDX_ACTIVE" in avg and "SQ_LDS_BANK_CONFLICT" in avg:
        summ["lds_bank_conflict_share_of_lds_cycles"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
    if "GRBM_GUI_ACTIVE" in avg:
        cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
        summ["kernel_cycles_per_xcd"] = cyc
        if "SQ_ACTIVE_INST_VALU" in avg:
            summ["valu_busy_fraction"] = avg["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
        if "SQ_LDS_IDX_ACTIVE" in avg:
            summ["lds_busy_fraction"] = avg["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)
    json.dump(summ, open(os.path.join(ROOT, "profiles", f"{out}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in summ.items() if k != "counters_mean_per_dispatch"}, indent=1))


if __name__ == "__main__":
    main()
