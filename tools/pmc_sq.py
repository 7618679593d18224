#!/usr/bin/env python3
"""tools/pmc_sq.py DIR KERNEL_SUBSTRING -- per-dispatch averages of every counter found in the rocprofv3 --pmc passes under
DIR (sq*/ sub-directories written by tools/profile_mid.sh), for the dispatches of one kernel, last dispatches only (the
warm-up launches run other paths), plus the derived per-wave shares."""
import csv, glob, sys, collections
d, kname = sys.argv[1], sys.argv[2]
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 5
vals = collections.defaultdict(list)
for f in glob.glob(d + "/sq*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if kname in r["Kernel_Name"]:
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    for k in sorted(per)[skip:]:
        for c, v in per[k].items():
            vals[c].append(v)
avg = {c: sum(v) / len(v) for c, v in vals.items()}
for c in sorted(avg):
    print(f"{c:28s} {avg[c]:16.1f}   (n={len(vals[c])})")
wc = avg.get("SQ_WAVE_CYCLES")
if wc:
    for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if c in avg:
            print(f"{c}/SQ_WAVE_CYCLES = {avg[c] / wc:.3f}")
if "SQ_LDS_IDX_ACTIVE" in avg and "SQ_LDS_BANK_CONFLICT" in avg:
    print(f"LDS bank conflict share of active cycles = {avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']:.3f}")
if "SQ_WAVES" in avg and "SQ_INSTS_VALU" in avg:
    print(f"VALU instructions per wave = {avg['SQ_INSTS_VALU'] / avg['SQ_WAVES']:.1f}")
