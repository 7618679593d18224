#!/usr/bin/env python3
"""Fold rocprofv3 --pmc CSV output (one directory per pass) into one JSON summary for profiles/.

usage: pmc_summary.py --kernel sk_render_fast2_kernel --voices N --frames F [--last 5] [--note ...] DIR [DIR ...]

Means are taken over the last `--last` dispatches of the named kernel in every pass (the timed, steady-state
launches of `bench.py --steps 5 --warmup 5`).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950
FETCH_SIZE under-counts by 2x (/opt/skills/guides/MI355X_MICROARCH.md), so the read figure is doubled here.
"""
import argparse
import collections
import csv
import glob
import json
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--voices", type=int, required=True)
    ap.add_argument("--frames", type=int, required=True)
    ap.add_argument("--voices-per-wave", type=int, default=128)
    ap.add_argument("--last", type=int, default=5)
    ap.add_argument("--note", default="")
    ap.add_argument("--command", default="")
    ap.add_argument("--algorithmic-bytes-per-voice-sample", type=float, default=0.0)
    ap.add_argument("dirs", nargs="+")
    a = ap.parse_args()
    means = {}
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            per = collections.defaultdict(dict)
            for r in csv.DictReader(open(f)):
                if a.kernel in r["Kernel_Name"]:
                    per[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
            ids = sorted(per)[-a.last:]
            for c in per[ids[-1]] if ids else []:
                means[c] = sum(per[i][c] for i in ids) / len(ids)
    out = {"kernel": a.kernel, "voices": a.voices, "frames_per_launch": a.frames, "command": a.command,
           "note": a.note, "dispatches_averaged": a.last, "counters_mean_per_dispatch": means}
    waves = a.voices / a.voices_per_wave
    wf = waves * a.frames
    if "SQ_INSTS_VALU" in means:
        out["per_wave_frame"] = {"voice_samples": a.voices_per_wave, "valu_insts": means["SQ_INSTS_VALU"] / wf,
                                 "salu_insts": means.get("SQ_INSTS_SALU", 0) / wf,
                                 "lds_insts": means.get("SQ_INSTS_LDS", 0) / wf}
    if "GRBM_GUI_ACTIVE" in means:
        cyc = means["GRBM_GUI_ACTIVE"] / 8.0            # the counter sums the 8 XCDs
        out["kernel_cycles_per_xcd"] = cyc
        if "SQ_ACTIVE_INST_VALU" in means:
            # 4 cycles per wave64 instruction on a 16-lane SIMD, 1024 SIMDs
            out["valu_busy_fraction"] = means["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * 1024.0)
        if "TA_BUSY_avr" in means:
            out["ta_busy_fraction"] = means["TA_BUSY_avr"] / cyc
        if "TCP_TCC_READ_REQ_sum" in means:
            out["l2_read_requests_per_voice_sample"] = means["TCP_TCC_READ_REQ_sum"] / (a.voices * a.frames)
    if "FETCH_SIZE" in means or "WRITE_SIZE" in means:
        rd = means.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0
        wr = means.get("WRITE_SIZE", 0.0) * 1024.0
        out["hbm_bytes_per_launch"] = {"read_corrected": rd, "write": wr, "total": rd + wr}
        if a.algorithmic_bytes_per_voice_sample:
            out["hbm_bytes_per_launch"]["algorithmic_bytes_per_launch"] = \
                a.algorithmic_bytes_per_voice_sample * a.voices * a.frames
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
