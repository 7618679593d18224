#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X render path on BASELINE's synthetic voice banks.

Metric (BASELINE.json): voice-samples/s = voices x frames rendered / wall seconds, whole job.
One "step" = one pass of the hot path over the whole bank: ONE launch that renders F frames of every voice, adds the
per-workgroup partial mixes up and applies the master volume (N = 1); for N > 1 GPUs: render + mix-down on every GPU,
one RCCL reduce(sum) of float[F][2] onto rank 0, master volume there.  State and tables are resident in HBM before the
timed region starts; the output frames stay in HBM (a real-time host would copy 8*F bytes).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4] [--frames F]

Default workload: BASELINE config 3 -- ONE bank of 2^20 voices (mixed notamy LUTs + biquad + ADSR), F = 512 frames
per launch (the reference's callback size, skred.h:12), 48 kHz.  With N > 1 (the driver launches one rank per GPU
through torch.distributed.run) that bank is SPLIT over the GPUs -- the literal config 3, "scaling": "strong" -- and the
line also carries `weak_scaling` (a 2^20-voice bank per GPU) for comparison.  With N = 1 the line also carries the
other BASELINE workloads (`c1`, `c2`, `c4`), two other block lengths (`low_latency`, `long_block`), the recipe from its
first frame (`envelopes_in_motion`), a bank under control traffic (`live_control`), the fixed-point path and the CPU
baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec"
STATE_READ, STATE_WRITE = 240, 52   # bytes per voice per launch, SURVEY §8(d)

WORKLOADS = {
    #        recipe  voices   interp  table bytes gathered from HBM-resident tables per voice-sample
    "c1": ("c1", 4096, 0, 0.0),
    "c2": ("c2", 65536, 0, 0.0),
    "c3": ("c2", 1048576, 0, 0.0),
    "c4": ("c4", 262144, 1, 8.0),     # 2 taps x 4 B (float tables), SURVEY §8(d)
}
DESCR = {
    "c1": "C1: 4096 voices, sine LUT + ADSR + amp smoother, 48 kHz, fp32",
    "c2": "C2: 65536 voices, mixed notamy sine/triangle/impulse LUTs + per-voice biquad + ADSR, 48 kHz, fp32",
    "c3": "C3 bank: 1048576 voices, mixed notamy sine/triangle/impulse LUTs + per-voice biquad + ADSR + amp smoother, 48 kHz, fp32",
    "c4": "C4: 262144 PCM voices (pcm_map geometry, synthetic samples), linear interpolation, 48 kHz, fp32",
}
KERNELS = {0: "sk_render_kernel", 1: "sk_render_fast_kernel", 2: "sk_render_mod_kernel", 3: "sk_render_fast2_kernel"}
RECIPE_WARMUP_S = 0.11     # attack + decay of the recipe's ADSR: after that every voice sits in its sustain stage


def cpu_baseline(recipe, interp, seconds_per_leg=8.0):
    """The oracle (oracle/cpu_ref.c, bit-pinned to the reference) timed on this box's host cores,
    compiled with the reference's own flags (-O3 -march=native, reference Makefile:23-28).  It is the
    CHECKER being timed as a baseline, never the thing measured above."""
    from oracle import cpuref
    from skred_amd import banks
    # a 1-GPU box grants a CPU share of 16 cores whatever os.cpu_count() says: use at most that many threads
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("SKRED_CPU_THREADS", "16"))))
    n1 = 4096
    bank, tables, g = banks.RECIPES[recipe](n1)
    cpuref.lib(fast=True)
    t = time.perf_counter()
    cpuref.render(bank.copy(), g.copy(), tables, 64, interp, fast=True)
    probe = max(time.perf_counter() - t, 1e-4)
    rate = n1 * 64 / probe
    frames = int(max(512, min(48000, seconds_per_leg * rate / n1)))
    t = time.perf_counter()
    cpuref.render(bank.copy(), g.copy(), tables, frames, interp, fast=True)
    one = n1 * frames / (time.perf_counter() - t)
    nm = n1 * cores
    bank_m, tables_m, g_m = banks.RECIPES[recipe](nm)
    t = time.perf_counter()
    cpuref.render_mt(bank_m, g_m, tables_m, frames, cores, interp, fast=True)
    many = nm * frames / (time.perf_counter() - t)
    ref = reference_rate()
    return {"value": many, "unit": "voice-samples/s", "cores": cores, "kind": "port",
            "value_1thread": one, "reference_1thread": ref,
            "sample": f"oracle/cpu_ref.c -O3 -march=native on the same recipe: {n1} voices x {frames} frames on 1 thread "
                      f"(the reference is single-threaded), {nm} voices x {frames} frames on {cores} threads (static voice partition)"}


def reference_rate(seconds=3.0):
    """The UNMODIFIED reference synth() (oracle/_ref/libskred_ref.so, compiled from /root/reference by
    oracle/Makefile with -O2 -ffp-contract=off; the .so travels to the GPU box, the sources do not) on its
    own maximum of 64 voices: sine/triangle/square tables + biquad + ADSR, one thread, 512-frame callbacks.
    Runs in a child process (the library prints and keeps static state).  None when the .so is absent."""
    so = os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so")
    if not os.path.exists(so):
        return None
    code = r"""
import ctypes, time, sys
import numpy as np
L = ctypes.CDLL(sys.argv[1]); L.ref_boot()
for v in range(64):
    f = 55.0 * 2 ** (v / 12.0)
    L.ref_wire(("v%d w%d f%.3f a1 p%.2f J%d K%.1f Q1.2 t0.01,0.1,0.7,0.2 l1" % (v, [0, 4, 1][v % 3], f, (v % 9) / 4.0 - 1.0, 1 + v % 4, 300.0 + 40 * v)).encode())
out = np.zeros((512, 2), np.float32)
p = out.ctypes.data_as(ctypes.c_void_p)
for _ in range(200): L.ref_callback(p, 512)
n, t0 = 0, time.perf_counter()
while time.perf_counter() - t0 < float(sys.argv[2]):
    for _ in range(100): L.ref_callback(p, 512)
    n += 100
print("RATE", 64 * 512 * n / (time.perf_counter() - t0))
"""
    import subprocess
    try:
        out = subprocess.run([sys.executable, "-c", code, so, str(seconds)], capture_output=True, text=True, timeout=60)
        rate = [float(l.split()[1]) for l in out.stdout.splitlines() if l.startswith("RATE ")]
        return {"value": rate[-1], "unit": "voice-samples/s", "cores": 1,
                "what": "unmodified reference synth()+seq() via its own synth_callback, 64 voices (VOICE_MAX), "
                        "oracle/_ref/libskred_ref.so (-O2 -ffp-contract=off)"} if rate else None
    except Exception:
        return None


def pmc_entry(workload, frames=None):
    """What the committed rocprofv3 PMC passes measured for this workload shape (profiles/pmc_traffic.json: HBM bytes
    per launch, L2 read requests per launch, VALU instructions per launch); bench.py itself cannot collect PMC counters.
    Block lengths other than 512 frames have entries of their own (`c3@64`, `c3@4800`), shards of config 3 too (`shard17` ...)."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if frames is not None and f"{workload}@{frames}" in tab:
            return tab[f"{workload}@{frames}"]
        return tab[workload]
    except Exception:
        return None


EVENT_PAIR_MS = 0.0        # what an EMPTY hipEvent pair on the render stream reads (calibrated once per run: main())
ISSUE_PEAK = None          # tools/issue_rate on this box: VALU instructions a SIMD issues per second (main())


def issue_peak():
    """tools/issue_rate (built by __graft_entry__.build() from tools/issue_rate.hip) on THIS box: the rate at which a SIMD with
    four resident waves issues plain fp32 VALU instructions (v_add_f32, independent chains), and packed ones.  None when the
    binary is absent."""
    exe = os.path.join(ROOT, "tools", "issue_rate")
    if not os.path.exists(exe):
        return None
    import subprocess
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
        for line in out.splitlines():
            if line.startswith("PEAK "):
                f = line.split()
                return {"plain_per_simd": float(f[2]), "packed_per_simd": float(f[4]), "simds": int(f[6]), "clock_mhz": float(f[8]),
                        "source": "tools/issue_rate.hip on this box: v_add_f32 / v_pk_add_f32, four independent chains, 4 waves per SIMD"}
    except Exception:
        pass
    return None


def valu_roofline(workload, voices, frames, k_mean):
    """Second roofline of the line: the resource that actually binds the LDS-table banks.  achieved = VALU (wave) instructions
    per launch, from the committed rocprofv3 PMC pass of this workload (SQ_INSTS_VALU, profiles/pmc_traffic.json), / this run's
    kernel time; peak = SIMDs x the plain-fp32 issue rate tools/issue_rate measures on this box.  A packed instruction
    (v_pk_*_f32: two voices' worth) occupies the VALU about 1.6x as long as a plain one, so a kernel that is half packed
    cannot reach 1.0; `valu_busy_fraction` is the PMC's own SQ_ACTIVE_INST_VALU / cycles figure for the same pass."""
    pm = pmc_entry(workload, frames)
    if not pm or not pm.get("valu_insts_per_launch") or pm.get("frames_per_launch") != frames or pm.get("voices") != voices or not ISSUE_PEAK:
        return None
    insts = pm["valu_insts_per_launch"]
    peak = ISSUE_PEAK["plain_per_simd"] * ISSUE_PEAK["simds"]
    ach = insts / (k_mean * 1e-3)
    return {"bound": "valu", "achieved": ach / 1e9, "peak": peak / 1e9, "unit": "G wave-instructions/s", "frac": ach / peak,
            "valu_insts_per_launch": insts, "valu_insts_per_voice_sample": insts / (voices * frames),
            "valu_busy_fraction_pmc": pm.get("valu_busy_fraction"), "packed_issue_per_simd": ISSUE_PEAK["packed_per_simd"],
            "peak_source": ISSUE_PEAK["source"], "insts_source": pm.get("valu_source", pm.get("source", ""))}


def roofline(workload, voices, frames, gather_bytes, k_mean, k_min, k_cnt, kernel):
    B = gather_bytes + (STATE_READ + STATE_WRITE) / frames       # algorithmic bytes / voice-sample, SURVEY §8(d)
    launch_bytes = B * voices * frames
    k_raw = k_mean
    k_mean = max(k_mean - EVENT_PAIR_MS, 1e-6)                   # the event pair's own share of the bracket (calibrated per run)
    k_min = max(k_min - EVENT_PAIR_MS, 1e-6)
    achieved = launch_bytes / (k_mean * 1e-3)
    pm = pmc_entry(workload, frames)
    traffic = pm["hbm_bytes_per_launch"] if pm and pm.get("frames_per_launch") == frames and pm.get("voices") == voices else None
    out = {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
           "frac": achieved / HBM_PEAK, "traffic": traffic,
           "traffic_rate": None if traffic is None else traffic / (k_mean * 1e-3) / 1e9,   # measured HBM GB/s of this kernel
           "traffic_frac": None if traffic is None else traffic / (k_mean * 1e-3) / HBM_PEAK,
           "kernel": kernel, "frames_per_launch": frames,
           "kernel_ms_mean": k_mean, "kernel_ms_min": k_min, "launches_timed": k_cnt,
           "kernel_ms_mean_raw": k_raw, "event_pair_ms": EVENT_PAIR_MS,
           "algorithmic_bytes_per_voice_sample": B, "algorithmic_bytes_per_launch": launch_bytes,
           "kernel_voice_samples_per_s": voices * frames / (k_mean * 1e-3)}
    if traffic is not None and pm.get("l2_read_requests_per_launch") and pm.get("l2_request_peak_per_s"):
        # the resource that binds a bank whose tables live in L2: line requests from the CUs' texture-address units
        req = pm["l2_read_requests_per_launch"]
        out["l2_requests"] = {"bound": "l2_read_requests", "achieved": req / (k_mean * 1e-3), "peak": pm["l2_request_peak_per_s"],
                              "unit": "requests/s", "frac": req / (k_mean * 1e-3) / pm["l2_request_peak_per_s"],
                              "requests_per_launch": req, "peak_source": pm.get("l2_request_peak_source", ""),
                              "units_note": "achieved counts CACHE-LINE requests that reached L2 (TCP_TCC_READ_REQ), peak counts LANE requests "
                                            "of the microbenchmark: not the same unit -- see gather_rate for the like-for-like figure"}
        # like for like: one lane-iteration of tools/ta_rate.hip's refill mode = one voice-frame of the kernel (every 8th iteration
        # a lane fetches its 5 x 16 bytes): the kernel's voice-frames per second against the microbenchmark's lane-iterations per second
        lane_peak = pm["l2_request_peak_per_s"] * 8.0 / 5.0
        out["gather_rate"] = {"bound": "texture-address / L1 request rate of the window refill pattern", "achieved": voices * frames / (k_mean * 1e-3),
                              "peak": lane_peak, "unit": "lane-iterations/s (= voice-frames/s)", "frac": voices * frames / (k_mean * 1e-3) / lane_peak,
                              "peak_source": pm.get("l2_request_peak_source", "")}
    return out


def main():
    # stdout carries exactly ONE line, the JSON result: libraries that chat on fd 1 (RCCL prints a version banner
    # when the first communicator is created) are sent to stderr for the whole run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=512, help="frames per launch (reference callback size, skred.h:12)")
    ap.add_argument("--voices", type=int, default=0, help="override the bank's voice count")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1 -- strong (default): ONE bank of the workload size split over the GPUs (BASELINE config 3 as written); "
                         "weak: a bank of that size on every GPU.  The default run reports both (`weak_scaling`)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="N = 1: only the main line (no other workloads / block lengths / legs)")
    ap.add_argument("--no-low-latency", action="store_true", help="(older name of --no-extra)")
    ap.add_argument("--no-fixed-point", action="store_true", help="skip the fixed-point leg")
    ap.add_argument("--no-scaling-proxy", action="store_true", help="skip the strong_scaling_proxy leg (shard blocks of 2^17 .. 2^19 voices)")
    ap.add_argument("--no-patches", action="store_true", help="skip the shipped_patches leg (the reference's .sk patches tiled over the bank)")
    ap.add_argument("--no-event-calibration", action="store_true", help="report bracketed kernel times as read (no empty-pair correction)")
    ap.add_argument("--no-recipe-warmup", action="store_true",
                    help="skip the recipe's own 0.11 s of untimed rendering (to time launches with envelopes still ramping)")
    ap.add_argument("--time-every", type=int, default=0,
                    help="bracket the render kernel of every n-th launch with HIP events (kernel duration for the roofline); "
                         "0 = choose so that at least 8 launches of the timed region are bracketed")
    ap.add_argument("--spinup-ms", type=float, default=40.0,
                    help="untimed rendering before the W warm-up steps until this much wall time has passed: a GPU taken from idle "
                         "needs ~25 ms of work before clocks and caches settle (20 steps right after first touch time ~10 %% slow)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="with one process: run the N > 1 code path (the shard's RCCL reduce, with one rank) on a single-GPU box")
    ap.add_argument("--serial-collective", action="store_true",
                    help="N > 1: render -> reduce -> master one after the other (skred_shard_render_mix); default: both forms are "
                         "calibrated on 40 untimed blocks and the faster one is timed")
    ap.add_argument("--pipelined-collective", action="store_true",
                    help="N > 1: the collective of block k beside the render of block k + 1 (skred_shard_render_mix_pipelined)")
    ap.add_argument("--fast2-min-voices", type=int, default=-1,
                    help="override the bank size from which the two-voices-per-lane kernel is used (-1: library default)")
    a = ap.parse_args()
    if a.no_low_latency:
        a.no_extra = True

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import numpy as np
    import torch
    import torch.distributed as dist

    from skred_amd import banks, device
    from skred_amd.sharded import Shard

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.rehearse_dist       # the N>1 code path (collective, barrier, max-over-ranks timing)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=dev)

    stream = torch.cuda.current_stream().cuda_stream
    every = a.time_every if a.time_every > 0 else min(8, max(1, a.steps // 10))

    # what an empty event pair on the render stream reads: taken out of every bracketed kernel time (the pair's own
    # few microseconds made kernel_ms_mean exceed ms_per_step on the small banks)
    global EVENT_PAIR_MS, ISSUE_PEAK
    if not a.no_event_calibration:
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
        x = torch.zeros(1 << 20, device=dev)
        for _ in range(3):
            x.add_(1.0)                                   # (the stream is busy, as it is around a bracketed launch)
            for e0, e1 in pairs:
                e0.record(); e1.record()
            torch.cuda.synchronize()
        EVENT_PAIR_MS = float(np.median([e0.elapsed_time(e1) for e0, e1 in pairs]))
    ISSUE_PEAK = issue_peak() if rank == 0 else None

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def recipe_warmup(render_block, frames):
        """BASELINE.md §4 / SURVEY §8(d): "render 1 s after 0.1 s warm-up".  The recipe's note-ons are staggered over the
        last second and attack+decay last 0.11 s, so those first frames are rendered untimed: the timed region then
        starts with every voice in its sustain stage, as the recipe intends (`envelopes_in_motion` times the rest)."""
        done = 0
        while done < int(RECIPE_WARMUP_S * 48000):
            render_block(frames)
            done += frames
        return done

    def spinup(render_block, frames):
        """untimed blocks until --spinup-ms of wall time have passed (reported as config.spinup_blocks)"""
        n, t0 = 0, time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < a.spinup_ms:
            for _ in range(8):
                render_block(frames)
            torch.cuda.synchronize()
            n += 8
        return n

    def timed(render_block, db, frames, steps, warmup, timing_every):
        """warm-up launches, then EXACTLY `steps` launches between two fences; max over ranks."""
        db.kernel_timing(timing_every)
        for _ in range(warmup):
            render_block(frames)
        fence()
        db.timing_reset()
        t0 = time.perf_counter()
        for _ in range(steps):
            render_block(frames)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        k_mean, k_min, k_cnt = db.timing_summary()
        return dt, k_mean, k_min, k_cnt

    # ------------------------------------------------------------------ the main line
    recipe, bank_voices, interp, gather_bytes = WORKLOADS[a.workload]
    if a.voices:
        bank_voices = a.voices
    F = a.frames

    def sharded_leg(total, scaling):
        """One bank of `total` voices split over the ranks (strong) or one bank of `total / world`... see caller."""
        sh = Shard(total, rank, world, local)
        if scaling == "strong" or world == 1:
            whole, tables, g = banks.RECIPES[recipe](total)          # the one global bank, generated on every rank
            sh.bank.set_tables(tables)
            sh.upload(whole)
            del whole
        else:                                                        # weak: every rank holds its own bank (seed + rank)
            mine, tables, g = banks.RECIPES[recipe](sh.n_local, seed=banks.SEED + rank)
            sh.bank.set_tables(tables)
            sh.bank.upload(mine)
            del mine
        sh.bank.set_globals(g)
        if a.fast2_min_voices >= 0:
            sh.bank.fast2_min_voices(a.fast2_min_voices)
        if world > 1 or a.rehearse_dist:
            ids = [Shard.rccl_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            sh.init_rccl(ids[0])
            if a.rehearse_dist and world == 1:
                sh.set_reduce(None, always_reduce=True)
        outs = [torch.zeros(F, 2, device=dev, dtype=torch.float32) for _ in range(2)]
        out = outs[0]
        kblk = [0]

        def block(frames):
            # the pipelined form: the collective of block k runs beside the render of block k + 1 (two output buffers in turn)
            if a.serial_collective:
                sh.render_mix(frames, out.data_ptr(), 2, interp, stream)
            else:
                sh.render_mix_pipelined(frames, outs[kblk[0] & 1].data_ptr(), 2, interp, stream)
                kblk[0] += 1
        warm = 0 if a.no_recipe_warmup else recipe_warmup(block, F)
        spun = spinup(block, F)
        fence()
        calib = None
        if not a.serial_collective and not a.pipelined_collective:
            # which form of the block is faster HERE (it depends on what the collective costs on this node): 40 untimed blocks each
            calib = {}
            for form in ("serial", "pipelined"):
                a.serial_collective = form == "serial"
                dtc, _, _, _ = timed(block, sh.bank, F, 40, 8, 0)
                calib[form + "_ms"] = dtc / 40 * 1e3
            a.serial_collective = calib["serial_ms"] <= calib["pipelined_ms"]
        dt, k_mean, k_min, k_cnt = timed(block, sh.bank, F, a.steps, a.warmup, every)
        form_used = "serial" if a.serial_collective else "pipelined"
        if calib is not None:
            a.serial_collective = False
        sh.flush(stream)
        torch.cuda.synchronize()
        finite = bool(all(torch.isfinite(o).all().item() for o in outs)) if rank == 0 else True
        res = {"dt": dt, "k": (k_mean, k_min, k_cnt), "kernel": KERNELS.get(sh.bank.last_kernel(), "?"), "finite": finite,
               "warm": warm, "spun": spun, "n_local": sh.n_local, "form": form_used, "calib": calib}
        sh.close()
        return res

    res = None
    if use_dist:
        total = bank_voices if a.scaling == "strong" else bank_voices * world
        m = sharded_leg(total, a.scaling)
        if rank == 0:
            value = total * F * a.steps / m["dt"]
            rl = roofline(a.workload, m["n_local"], F, gather_bytes, *m["k"], m["kernel"])
            res = {
                "metric": "voice-samples/s", "value": value, "unit": "voice-samples/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": m["dt"] / a.steps * 1e3, "higher_is_better": True,
                "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": DESCR[a.workload] + (f" [voices overridden: {bank_voices}]" if a.voices else "") +
                                       (f"; split over {world} GPUs" if a.scaling == "strong" else f"; one such bank per GPU ({world} GPUs)"),
                           "voices_total": total, "voices_per_gpu": m["n_local"], "frames_per_launch": F, "sample_rate": 48000,
                           "interp": "linear" if interp else "truncate",
                           "parallelism": f"voices block-partitioned over {world} GPU(s); one RCCL reduce(sum) of float[F][2] per launch "
                                          + ("(skred_shard_render_mix: render + mix-down on every GPU -> ncclReduce -> master volume on rank 0)" if m["form"] == "serial" else
                                             "(skred_shard_render_mix_pipelined: render + mix-down on every GPU; ncclReduce + master volume of block k on a "
                                             "second stream beside the render of block k + 1)"),
                           "collective_form": m["form"], "collective_form_calibration": m["calib"],
                           "seed": "0x5EED", "recipe_warmup_frames": m["warm"], "spinup_blocks": m["spun"]},
                "realtime_factor_48k": value / (total * 48000.0), "output_finite": m["finite"], "roofline": rl,
            }
        if world > 1 and not a.no_extra:
            other = "weak" if a.scaling == "strong" else "strong"
            total2 = bank_voices * world if other == "weak" else bank_voices
            m2 = sharded_leg(total2, other)
            if rank == 0:
                v2 = total2 * F * a.steps / m2["dt"]
                res[other + "_scaling"] = {"value": v2, "unit": "voice-samples/s", "scaling": other, "voices_total": total2,
                                           "voices_per_gpu": m2["n_local"], "ms_per_step": m2["dt"] / a.steps * 1e3,
                                           "kernel": m2["kernel"], "kernel_ms_mean": m2["k"][0], "launches_timed": m2["k"][2]}
        if rank == 0:
            os.write(result_fd, (json.dumps(res) + "\n").encode())
        dist.destroy_process_group()
        return

    # ------------------------------------------------------------------ one GPU
    def single_leg(wl, frames, steps, warmup, timing_every, voices=0, warm_recipe=True):
        rec, n, itp, gbytes = WORKLOADS[wl]
        n = voices or n
        bank, tables, g = banks.RECIPES[rec](n)
        db = device.DeviceBank(n, local)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        if a.fast2_min_voices >= 0:
            db.fast2_min_voices(a.fast2_min_voices)
        out = torch.zeros(frames, 2, device=dev, dtype=torch.float32)

        def block(fr):
            db.render_mix(fr, out.data_ptr(), 2, 0, itp, stream)
        warm = recipe_warmup(block, frames) if warm_recipe else 0
        spun = spinup(block, frames)
        fence()
        dt, k_mean, k_min, k_cnt = timed(block, db, frames, steps, warmup, timing_every)
        r = roofline(wl, n, frames, gbytes, k_mean, k_min, k_cnt, KERNELS.get(db.last_kernel(), "?"))
        r["value"] = n * frames * steps / dt
        r["ms_per_step"] = dt / steps * 1e3
        r["realtime_factor_48k"] = r["value"] / (n * 48000.0)
        r["voices"] = n
        finite = bool(torch.isfinite(out).all().item())
        return r, dt, (warm, spun), finite, (db, bank, tables, g, out)

    rl, dt, warm, finite, keep = single_leg(a.workload, F, a.steps, a.warmup, every, bank_voices, not a.no_recipe_warmup)
    db, bank, tables, g, out = keep
    value = bank_voices * F * a.steps / dt
    rl.pop("value"); rl.pop("ms_per_step"); rl.pop("realtime_factor_48k"); rl.pop("voices")
    vr = valu_roofline(a.workload, bank_voices, F, rl["kernel_ms_mean"])
    rl["note"] = ("LUTs are LDS-resident and the recurrences live in registers, so compulsory HBM traffic is one state sweep per "
                  "launch: at F=512 the kernel is bound by fp32 VALU issue (profiles/), not by HBM; the HBM fraction grows as F "
                  "shrinks (low_latency).  PCM banks (c4) gather from an L2-resident pool: see c4.l2_requests")
    res = {
        "metric": "voice-samples/s", "value": value, "unit": "voice-samples/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
        "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": DESCR[a.workload] + (f" [voices overridden: {bank_voices}]" if a.voices else ""),
                   "voices_total": bank_voices, "voices_per_gpu": bank_voices, "frames_per_launch": F, "sample_rate": 48000,
                   "interp": "linear" if interp else "truncate",
                   "parallelism": "one GPU: one launch per block (render + mix-down + master volume in the render kernel)",
                   "seed": "0x5EED", "recipe_warmup_frames": warm[0], "spinup_blocks": warm[1]},
        "realtime_factor_48k": value / (bank_voices * 48000.0), "output_finite": finite, "roofline": rl,
    }
    if vr:
        res["roofline_valu"] = vr
    if not a.no_extra:
        # ---- the same bank with LINEAR interpolation (north_star's lookup mode; the reference truncates): two LDS taps
        if not interp:
            def block_lin(frames):
                db.render_mix(frames, out.data_ptr(), 2, 0, 1, stream)
            steps_l = max(60, a.steps)                    # (another instantiation of the kernel: its own warm-up)
            dtl, kml, knl, kcl = timed(block_lin, db, F, steps_l, 20, min(8, max(1, steps_l // 10)))
            rlin = roofline(a.workload, bank_voices, F, gather_bytes, kml, knl, kcl, KERNELS.get(db.last_kernel(), "?"))
            rlin.pop("traffic", None); rlin.pop("traffic_rate", None); rlin.pop("traffic_frac", None)   # (the PMC pass was the truncating run)
            rlin["value"] = bank_voices * F * steps_l / dtl
            rlin["ms_per_step"] = dtl / steps_l * 1e3
            rlin["interp"] = "linear"
            res[a.workload + "_linear"] = rlin
        # ---- other block lengths on the same bank (state keeps running): F = 64 (1.33 ms callbacks: the per-launch state
        # sweep dominates and the kernel approaches the HBM roofline) and F = 4800 (100 ms blocks), SURVEY §8(d)
        for key, fr, steps, warmup, te in (("low_latency", 64, max(50, a.steps), 20, min(8, max(1, max(50, a.steps) // 10))),
                                           ("long_block", 4800, max(12, a.steps // 10), 2, 2)):
            if fr == F:
                continue
            o2 = torch.zeros(fr, 2, device=dev, dtype=torch.float32)

            def block2(frames, _o=o2):
                db.render_mix(frames, _o.data_ptr(), 2, 0, interp, stream)
            dt2, km, kn, kc = timed(block2, db, fr, steps, warmup, te)
            r2 = roofline(a.workload, bank_voices, fr, gather_bytes, km, kn, kc, KERNELS.get(db.last_kernel(), "?"))
            r2["value"] = bank_voices * fr * steps / dt2
            r2["ms_per_step"] = dt2 / steps * 1e3
            r2["realtime_factor_48k"] = r2["value"] / (bank_voices * 48000.0)
            res[key] = r2

        # ---- the recipe from its FIRST frame: ~11 % of the voices are in attack / decay at t0 and reach sustain over the
        # first 0.11 s; these are the launches the main line renders untimed (sk_render_env2_kernel takes the slices with a
        # moving envelope).  Fresh state per repetition; the blocks of the first 0.117 s are timed.
        n_blocks = -(-int(RECIPE_WARMUP_S * 48000) // F)
        reps = []
        for _ in range(3):
            db.upload(bank)
            db.set_globals(g)
            db.kernel_timing(0)
            fence()
            t0 = time.perf_counter()
            for _ in range(n_blocks):
                db.render_mix(F, out.data_ptr(), 2, 0, interp, stream)
            fence()
            reps.append((time.perf_counter() - t0) / n_blocks)
        reps.sort()
        res["envelopes_in_motion"] = {"value": bank_voices * F / reps[1], "unit": "voice-samples/s", "ms_per_step": reps[1] * 1e3,
                                      "ms_per_step_min": reps[0] * 1e3, "blocks_timed": n_blocks, "repetitions": 3,
                                      "what": f"the first {n_blocks} blocks of the recipe (note-ons staggered over the last second, "
                                              "attack + decay 0.11 s): launches with envelope ramps in flight, no recipe warm-up"}

        # ---- control traffic: every block 0.05 % of the voices get a note-off (stamped on the device) or a note-on with
        # new parameters, through skred_bank_update on the render stream
        for _ in range(12):                                    # back to the all-sustain state
            db.render_mix(F, out.data_ptr(), 2, 0, interp, stream)
        rng = np.random.default_rng(1)
        D = device
        for key, share in (("live_control", 0.0005), ("live_control_sparse", 0.0001)):
            k_ev = max(2, int(bank_voices * share))

            def control_block(frames):
                vs = rng.choice(bank_voices, k_ev, replace=False).astype(np.int32)
                db.update(bank, vs[:k_ev // 2], D.STAMP_RELEASE, stream)
                db.update(bank, vs[k_ev // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, stream)
                db.render_mix(frames, out.data_ptr(), 2, 0, interp, stream)
            steps_c = max(60, a.steps)
            dtc, kmc, knc, kcc = timed(control_block, db, F, steps_c, 30, 0)
            res[key] = {"value": bank_voices * F * steps_c / dtc, "unit": "voice-samples/s", "ms_per_step": dtc / steps_c * 1e3,
                        "voices_touched_per_block": k_ev, "note_events_per_s": k_ev * steps_c / dtc,
                        "motion_list_rendered": "in place (sk_gain_kernel + the steady kernel's in-place instantiation)" if db.last_in_place()
                                                else "by the envelope kernel beside the steady kernel",
                        "what": f"{share * 100:g} % of the voices per block: half note-offs (SKRED_STAMP_RELEASE), half note-ons with new "
                                "parameters (SKRED_STAMP_TRIGGER | DIRTY_PHASE | DIRTY_PARAMS) via skred_bank_update, then the block"}
            for _ in range(24):                                # the notes of this leg come to rest
                db.render_mix(F, out.data_ptr(), 2, 0, interp, stream)
    db.close()
    del bank

    if not a.no_extra:
        # ---- the other BASELINE workloads, each on its own bank at BASELINE's size, F = 512
        for wl in ("c1", "c2", "c3", "c4"):
            if wl == a.workload or (wl == "c3" and a.workload == "c3"):
                continue
            steps_w = max(100, a.steps)
            r, _, _, fin, kp = single_leg(wl, 512, steps_w, 20, min(8, max(1, steps_w // 10)))
            # the same bank from its recipe's first frame (envelopes in motion): fresh state per repetition, first 0.117 s
            db_w, bank_w, _, g_w, out_w = kp
            n_blocks_w = -(-int(RECIPE_WARMUP_S * 48000) // 512)
            reps_w = []
            for _ in range(3):
                db_w.upload(bank_w)
                db_w.set_globals(g_w)
                db_w.kernel_timing(0)
                fence()
                t0w = time.perf_counter()
                for _ in range(n_blocks_w):
                    db_w.render_mix(512, out_w.data_ptr(), 2, 0, WORKLOADS[wl][2], stream)
                fence()
                reps_w.append((time.perf_counter() - t0w) / n_blocks_w)
            reps_w.sort()
            r["envelopes_in_motion"] = {"ms_per_step": reps_w[1] * 1e3, "value": r["voices"] * 512 / reps_w[1], "unit": "voice-samples/s",
                                        "blocks_timed": n_blocks_w, "repetitions": 3}
            if not WORKLOADS[wl][2]:                      # ... and with linear interpolation (north_star's lookup mode)
                db_w.upload(bank_w)
                db_w.set_globals(g_w)

                def block_wl(fr, _db=db_w, _o=out_w):
                    _db.render_mix(fr, _o.data_ptr(), 2, 0, 1, stream)
                recipe_warmup(block_wl, 512)
                dtl, kml, knl, kcl = timed(block_wl, db_w, 512, steps_w, 20, min(8, max(1, steps_w // 10)))
                rlin = roofline(wl, r["voices"], 512, WORKLOADS[wl][3], kml, knl, kcl, KERNELS.get(db_w.last_kernel(), "?"))
                rlin.pop("traffic", None); rlin.pop("traffic_rate", None); rlin.pop("traffic_frac", None)
                rlin["value"] = r["voices"] * 512 * steps_w / dtl
                rlin["ms_per_step"] = dtl / steps_w * 1e3
                rlin["interp"] = "linear"
                res[wl + "_linear"] = rlin
            db_w.close()
            r["workload"] = DESCR[wl]
            r["output_finite"] = fin
            vr_w = valu_roofline(wl, r["voices"], 512, r["kernel_ms_mean"])
            if vr_w:
                r["roofline_valu"] = vr_w
            res[wl] = r

    # ---- strong scaling, as far as ONE GPU can show it: the block of the shard BASELINE config 3 leaves each of 8 / 4 / 2 GPUs
    # (2^17 / 2^18 / 2^19 voices), as a fused single-GPU block, through the N > 1 sequence with a one-rank RCCL reduce, and through
    # the pipelined form of that sequence.  What an N-GPU run adds to these numbers is the collective's own latency (serial form) or
    # nothing until it exceeds the render (pipelined form).
    if not a.no_extra and a.workload == "c3" and not a.no_scaling_proxy:
        proxy = {"what": "per-GPU shard of BASELINE config 3 on this ONE GPU: ms per 512-frame block.  fused: render + mix-down + master in one "
                         "launch; serial / pipelined: the N > 1 sequence skred_shard_render_mix(_pipelined) -- sum-only render, master kernel, "
                         "and for the pipelined form the two-stream event chain -- without a collective; rccl_1rank_*: the same with the "
                         "library's own RCCL communicator of ONE rank in the sequence (a one-rank ncclReduce launches nothing on the stream it "
                         "was first used on, but costs ~70 us of stream time on the pipelined form's second stream: not what an 8-rank reduce "
                         "does; the 8-GPU run itself calibrates both forms and times the faster one)", "shards": {}}
        steps_p = max(100, a.steps)
        for gpus, n_sh in ((8, 1 << 17), (4, 1 << 18), (2, 1 << 19)):
            ent = {"voices": n_sh}
            try:
                whole, tables_p, g_p = banks.RECIPES["c2"](n_sh)
                for form in ("fused", "serial", "pipelined", "rccl_1rank_serial", "rccl_1rank_pipelined"):
                    sh = Shard(n_sh, 0, 1, local)
                    sh.bank.set_tables(tables_p)
                    sh.upload(whole)
                    sh.bank.set_globals(g_p)
                    o2 = [torch.zeros(512, 2, device=dev, dtype=torch.float32) for _ in range(2)]
                    kk = [0]
                    if form.startswith("rccl"):
                        sh.init_rccl(Shard.rccl_unique_id())
                        sh.set_reduce(None, always_reduce=True)

                    def blk(frames, _sh=sh, _o=o2, _kk=kk, _form=form):
                        if _form == "fused":
                            _sh.bank.render_mix(frames, _o[0].data_ptr(), 2, 0, 0, stream)
                        elif _form.endswith("serial"):
                            _sh.render_mix(frames, _o[0].data_ptr(), 2, 0, stream)
                        else:
                            _sh.render_mix_pipelined(frames, _o[_kk[0] & 1].data_ptr(), 2, 0, stream)
                            _kk[0] += 1
                    recipe_warmup(blk, 512)
                    spinup(blk, 512)
                    dtp, _, _, _ = timed(blk, sh.bank, 512, steps_p, 20, 0)
                    ent[form + "_ms"] = dtp / steps_p * 1e3
                    if form == "fused":
                        # the shard's own kernel time (event pairs around every 8th launch, a run of its own: a pair costs the stream a
                        # few microseconds) and what it means against the VALU issue peak (instructions per launch: profiles/)
                        _, kmp, knp, kcp = timed(blk, sh.bank, 512, steps_p, 10, 8)
                        ent["kernel"] = KERNELS.get(sh.bank.last_kernel(), "?")
                        ent["kernel_ms"] = max(kmp - EVENT_PAIR_MS, 1e-6)
                        ent["kernel_ms_min"] = max(knp - EVENT_PAIR_MS, 1e-6)
                        ent["launches_timed"] = kcp
                        vr_s = valu_roofline(f"shard{n_sh.bit_length() - 1}", n_sh, 512, ent["kernel_ms"])
                        if vr_s:
                            ent["roofline_valu"] = vr_s
                    sh.close()
                del whole
            except Exception as ex:       # (a box without RCCL: the fused number stands alone)
                ent["error"] = repr(ex)[:200]
            proxy["shards"][f"1/{gpus}"] = ent
        one = res["ms_per_step"]
        for key, ent in proxy["shards"].items():
            gpus = int(key.split("/")[1])
            for form in ("serial", "pipelined"):
                if form + "_ms" in ent:
                    ent[form + "_speedup_if_collective_were_free"] = one / ent[form + "_ms"]
        res["strong_scaling_proxy"] = proxy

    # ---- the reference's SHIPPED PATCHES tiled over a bank of the headline size (banks.bank_patch: the voice state the unmodified
    # reference holds after loading the patch, repeated; routings of synth.c:548-558,584-587,597-602 -- a modulator shared by three
    # carriers, frequency + amplitude + pan modulation, chains, sample & hold, a modulator BELOW its carrier).  Parity of exactly
    # these banks: tests/test_patch_banks.py, tests/test_fm_skew.py.
    if not a.no_extra and a.workload == "c3" and not a.no_patches:
        pat = {"what": "ms per 512-frame block of a 2^20-voice bank made by tiling one of the reference's .sk patches; kernel 1: one voice per "
                       "lane (previous-frame modulation: source lanes a block ahead of their readers, samples through an LDS ring -- "
                       "SKRED_OPT_FM_SKEW), 2: the modulated kernel (same-frame dependencies); pack: lanes per 64-voice group when the "
                       "bank is sparse (SKRED_OPT_PACK)", "banks": {}}
        for pname in ("3sk", "1sk", "7sk", "37sk", "18sk"):
            try:
                pb, pt, pg = banks.bank_patch(pname, bank_voices)
                pdb = device.DeviceBank(bank_voices, local)
                pdb.set_tables(pt); pdb.upload(pb); pdb.set_globals(pg); pdb.kernel_timing(0)
                pout = torch.zeros(512, 2, device=dev, dtype=torch.float32)

                def pblk(frames, _db=pdb, _o=pout):
                    _db.render_mix(frames, _o.data_ptr(), 2, 0, 0, stream)
                for _ in range(12):
                    pblk(512)
                steps_q = max(20, a.steps)
                dtq, _, _, _ = timed(pblk, pdb, 512, steps_q, 5, 0)
                pat["banks"][pname] = {"ms_per_step": dtq / steps_q * 1e3, "value": bank_voices * 512 * steps_q / dtq, "unit": "voice-samples/s",
                                       "kernel": int(pdb.last_kernel()), "pack": int(pdb.last_pack()),
                                       "output_finite": bool(torch.isfinite(pout).all().item())}
                pdb.close()
                del pb
            except Exception as ex:
                pat["banks"][pname] = {"error": repr(ex)[:200]}
        res["shipped_patches"] = pat

    # ---- the fixed-point LUT path (include/skred_amd_fxpt.h; integer mix, exact)
    if not a.no_extra and not a.no_fixed_point and a.workload != "c4":
        from skred_amd import fxbank
        fb, fpool, fcount0 = fxbank.bank_fx(bank_voices)
        fdb = fxbank.DeviceFxBank(bank_voices, local)
        fdb.set_tables(fpool)
        fdb.upload(fb)
        fdb.set_sample_count(fcount0)
        fmix = torch.zeros(F, 2, device=dev, dtype=torch.int64)
        for _ in range(12 + a.warmup):                      # 12 x 512 frames: the recipe's last note-on reaches sustain
            fdb.render_mix(F, fmix.data_ptr(), 1, 0, stream)
        fence()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fdb.render_mix(F, fmix.data_ptr(), 1, 0, stream)   # ONE launch per block: render + int64 mix-down + integer master stage
        fence()
        fdt = time.perf_counter() - t0
        res["fixed_point"] = {"value": bank_voices * F * a.steps / fdt, "unit": "voice-samples/s", "dtype": fxbank.DTYPE_NOTE,
                              "frames_per_launch": F, "ms_per_step": fdt / a.steps * 1e3, "kernel": "sk_fx_render_kernel", "launches_per_block": 1,
                              "kernel_ms_last": fdb.last_render_ms(), "mix_nonzero": bool((fmix != 0).any().item()),
                              "workload": fxbank.WORKLOAD_NOTE}
        fdb.close()
    if not a.no_cpu:
        res["cpu_baseline"] = cpu_baseline(recipe, interp)
    os.write(result_fd, (json.dumps(res) + "\n").encode())


if __name__ == "__main__":
    main()
