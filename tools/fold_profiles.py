#!/usr/bin/env python3
"""tools/fold_profiles.py TAG [OUT_PREFIX] -- fold the rocprofv3 passes of tools/profile_round.sh (gpurun_out/prof_<TAG>_<workload>/)
into profiles/<OUT_PREFIX>_<workload>_{kernel_stats.csv,pmc_summary.json} and refresh profiles/pmc_traffic.json (what bench.py
reports as roofline.traffic / c4.l2_requests).  Runs here, after gpurun merged gpurun_out/ back."""
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# key -> (kernel, voices, voices per wave, gather bytes per voice-sample, frames per launch, directory suffix under gpurun_out/prof_<TAG>_)
W = {"c1": ("sk_render_fast_kernel", 4096, 64, 0.0, 512, "c1"), "c2": ("sk_render_fast_kernel", 65536, 64, 0.0, 512, "c2"),
     "c3": ("sk_render_fast2_kernel", 1048576, 128, 0.0, 512, "c3"), "c4": ("sk_render_fast_kernel", 262144, 64, 8.0, 512, "c4"),
     # config 3 at the other block lengths bench.py reports (low_latency, long_block): tools/profile_round.sh c3 <TAG>_c3f64 --frames 64
     "c3@64": ("sk_render_fast2_kernel", 1048576, 128, 0.0, 64, "c3f64"), "c3@4800": ("sk_render_fast2_kernel", 1048576, 128, 0.0, 4800, "c3f4800"),
     # the shards of config 3 (strong_scaling_proxy): tools/profile_round.sh c2 <TAG>_shard17 --voices 131072
     "shard17": ("sk_render_fast_kernel", 131072, 64, 0.0, 512, "shard17"), "shard18": ("sk_render_fast2_kernel", 262144, 128, 0.0, 512, "shard18"),
     "shard19": ("sk_render_fast2_kernel", 524288, 128, 0.0, 512, "shard19")}


def main():
    tag = sys.argv[1]
    prefix = sys.argv[2] if len(sys.argv) > 2 else tag
    traffic_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    traffic = json.load(open(traffic_path))
    ta = {}
    ta_file = os.path.join(ROOT, "profiles", "r03_ta_rate.txt")
    if os.path.exists(ta_file):
        import re
        for line in open(ta_file):
            m = re.match(r"^(.*?)\s+([\d.]+) ms\s+([\d.e+]+) lane-gathers/s", line)
            if m:
                ta[m.group(1).strip()] = float(m.group(3))
    for w, (kernel, voices, vpw, gbytes, frames, suffix) in W.items():
        d = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{suffix}")
        if not os.path.isdir(d):
            continue
        stats = glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True)
        stats = [f for f in stats if "sk_render" in open(f).read()]      # (bench.py's child processes -- tools/issue_rate -- are traced too)
        if stats:
            shutil.copy(stats[0], os.path.join(ROOT, "profiles", f"{prefix}_{suffix}_kernel_stats.csv"))
        B = gbytes + 292.0 / frames
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--kernel", kernel, "--voices", str(voices),
                              "--frames", str(frames), "--voices-per-wave", str(vpw), "--algorithmic-bytes-per-voice-sample", str(B),
                              "--command", f"tools/profile_round.sh {w} {tag} (rocprofv3 --pmc ... -- python3 bench.py --workload {w} --steps 5 "
                                           "--warmup 5 --no-cpu --no-extra --time-every 1; separate passes for FETCH_SIZE, WRITE_SIZE, SQ and memory counters)",
                              "--note", f"{prefix}: {w} on {kernel}; one dispatch = one block (render + in-kernel mix-down + master volume)",
                              os.path.join(d, "fetch"), os.path.join(d, "write"), os.path.join(d, "sq"), os.path.join(d, "mem")],
                             capture_output=True, text=True, check=True)
        summ = json.loads(out.stdout)
        json.dump(summ, open(os.path.join(ROOT, "profiles", f"{prefix}_{suffix}_pmc_summary.json"), "w"), indent=1)
        e = {"frames_per_launch": frames, "voices": voices, "hbm_bytes_per_launch": summ["hbm_bytes_per_launch"]["total"],
             "algorithmic_bytes_per_launch": summ["hbm_bytes_per_launch"]["algorithmic_bytes_per_launch"],
             "source": f"profiles/{prefix}_{suffix}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH doubled)"}
        valu = summ["counters_mean_per_dispatch"].get("SQ_INSTS_VALU")
        if valu:      # bench.py: roofline_valu (wave-level VALU instructions per launch; VALU-busy fraction of the same pass)
            e["valu_insts_per_launch"] = valu
            e["valu_busy_fraction"] = summ.get("valu_busy_fraction")
            e["valu_source"] = f"profiles/{prefix}_{suffix}_pmc_summary.json (rocprofv3 --pmc SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, own pass)"
        if w == "c4":
            req = summ["counters_mean_per_dispatch"].get("TCP_TCC_READ_REQ_sum")
            if req:
                e["l2_read_requests_per_launch"] = req
            key = "5 x dwordx4 every 8th iter"            # the table-window refill pattern of the kernel
            if key in ta:
                # the tool counts lane-iterations; that mode issues 5 dwordx4 lane-requests per 8 of them
                e["l2_request_peak_per_s"] = ta[key] * 5.0 / 8.0
                e["l2_request_peak_source"] = ("tools/ta_rate.hip on the same box, mode '5 x dwordx4 every 8th iter' (the window refill pattern: "
                                               f"{ta[key]:.3e} lane-iterations/s x 5/8 lane-requests each), profiles/r03_ta_rate.txt")
        traffic[w] = e
        print(w, json.dumps({k: summ.get(k) for k in ("per_wave_frame", "valu_busy_fraction", "ta_busy_fraction", "kernel_cycles_per_xcd")}),
              "hbm MB", summ["hbm_bytes_per_launch"]["total"] / 1e6)
    json.dump(traffic, open(traffic_path, "w"), indent=1)


if __name__ == "__main__":
    main()
