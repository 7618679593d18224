#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X render path on BASELINE's synthetic voice banks.

Metric (BASELINE.json): voice-samples/s = voices x frames rendered / wall seconds, whole job.
One "step" = one pass of the hot path over the whole bank: one render launch of F frames for
every voice (+ the partial-mix reduction, + for N>1 GPUs the RCCL sum of the per-GPU partial
mixes, + the master-volume stage on rank 0).  State and tables are resident in HBM before the
timed region starts; the output frames stay in HBM (a real-time host would copy 8*F bytes).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c1|c2|c3|c4] [--frames F]

Default workload: the BASELINE config-3 bank -- 2^20 voices (mixed notamy LUTs + biquad + ADSR) per
GPU, F = 512 frames per launch (the reference's callback size, skred.h:12), 48 kHz.  N>1 is
launched by the driver through torch.distributed.run, one rank per GPU; every GPU holds a bank of
that size (weak scaling), voices never cross GPUs, the only collective is one reduce of float[F][2].
`--scaling strong` splits ONE 2^20-voice bank over the GPUs instead (the literal config 3).
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


def pmc_traffic(workload, voices, frames):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json), when
    one exists for exactly this workload shape; bench.py itself cannot collect PMC counters."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[workload]
        if d["frames_per_launch"] == frames and d["voices"] == voices:
            return d["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


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
    ap.add_argument("--voices", type=int, default=0, help="override the per-bank voice count")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="weak (default): every GPU gets a whole bank of the workload size; strong: one bank is split over the GPUs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-fixed-point", action="store_true", help="skip the fixed-point leg (N=1 only)")
    ap.add_argument("--no-low-latency", action="store_true", help="skip the secondary F=64 measurement")
    ap.add_argument("--no-recipe-warmup", action="store_true",
                    help="skip the recipe's own 0.11 s of untimed rendering (to time launches with envelopes still ramping)")
    ap.add_argument("--time-every", type=int, default=8,
                    help="bracket the render kernels of every n-th launch with HIP events (kernel duration for the roofline)")
    ap.add_argument("--tail-overlap", action="store_true",
                    help="one GPU: run every block's reduction + master stage on the bank's internal stream so that the next "
                         "block's render overlaps them (SKRED_OPT_OVERLAP_TAIL).  Off by default: measured at these very settings "
                         "it is worth -4..+0.5 % on the four workloads (DESIGN.md, 'Per-block launch count')")
    ap.add_argument("--no-tail-overlap", action="store_true", help="(the default; kept for the scripts under tools/)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="with one process: still create a (1-rank) process group and run the N>1 code path through it "
                         "(exercises the RCCL calls on a single-GPU box)")
    ap.add_argument("--overlap", action="store_true",
                    help="N>1: double-buffered step, the reduce of block k (RCCL's stream) overlaps the render of block k+1 "
                         "(ShardedRender.step_overlapped).  Default: the sequential step render -> reduce -> master; measured "
                         "through a one-rank RCCL group the overlapped form costs 17 us more per block than it hides "
                         "(0.293 vs 0.276 ms: the collective's kernel dispatched in turns with the render), DESIGN.md section 5")
    ap.add_argument("--no-overlap", action="store_true", help="(the default; kept for older command lines)")
    ap.add_argument("--backend", default=os.environ.get("SKRED_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default).  gloo is only for rehearsing the N>1 code path on a box with fewer "
                         "GPUs than ranks (ranks then share devices and the partial mix is reduced through host memory)")
    ap.add_argument("--fast2-min-voices", type=int, default=-1,
                    help="override the bank size from which the two-voices-per-lane kernel is used (-1: library default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        a.gpus = world

    import torch
    import torch.distributed as dist

    from skred_amd import banks, device
    from skred_amd.sharded import ShardedRender

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    ndev = torch.cuda.device_count()
    if a.backend == "nccl" and local >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but only {ndev} GPU(s) visible")
    local = local % ndev                       # gloo rehearsal: ranks may share a device
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or a.rehearse_dist       # the N>1 code path (collectives, barrier, max-over-ranks timing)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    recipe, bank_voices, interp, gather_bytes = WORKLOADS[a.workload]
    if a.voices:
        bank_voices = a.voices
    total = bank_voices * world if a.scaling == "weak" else bank_voices
    sh = ShardedRender(total, rank, world, always_reduce=a.rehearse_dist)
    F = a.frames

    # seeded banks: weak scaling gives every rank its own bank of the workload size (seed + rank);
    # strong scaling generates the one global bank on every rank and keeps this rank's block
    if a.scaling == "weak":
        shard, tables, g = banks.RECIPES[recipe](bank_voices, seed=banks.SEED + rank)
    else:
        full, tables, g = banks.RECIPES[recipe](total)
        shard = full.take(slice(sh.lo, sh.hi)) if world > 1 else full
        del full
    assert shard.n == sh.n_local
    db = device.DeviceBank(shard.n, local)
    db.set_tables(tables)
    db.upload(shard)
    db.set_globals(g)
    if a.fast2_min_voices >= 0:
        db.fast2_min_voices(a.fast2_min_voices)
    # the render kernels of every 8th launch are bracketed by an event pair (roofline.kernel_ms_*): a pair costs ~6 us
    # of stream time, so bracketing every launch would tax the very throughput being measured
    db.kernel_timing(max(1, min(a.time_every, a.steps)))
    if world == 1 and not a.rehearse_dist and a.tail_overlap and not a.no_tail_overlap:
        db.overlap_tail(True)          # block k's reduction + master overlap block k+1's render (all inside the timed region)

    stream = torch.cuda.current_stream().cuda_stream

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(frames, steps, warmup):
        partial = torch.zeros(frames, 2, device=dev, dtype=torch.float32)
        out = torch.zeros(frames, 2, device=dev, dtype=torch.float32)

        def render_partial(p):
            db.render(frames, p.data_ptr(), 0, interp, stream)

        def master(p, o):
            db.master(p.data_ptr(), frames, o.data_ptr(), 2, stream)

        if use_dist and a.backend == "gloo":     # rehearsal only: reduce through host memory
            host = torch.zeros(frames, 2, dtype=torch.float32)
            dev_render = render_partial

            def render_partial(p):                 # noqa: F811
                dev_render(partial)
                torch.cuda.synchronize()
                p.copy_(partial)

            dev_master = master

            def master(p, o):                      # noqa: F811
                partial.copy_(p)
                dev_master(partial, o)

            red = host
        else:
            red = partial
        if use_dist and a.overlap and not a.no_overlap:
            # double-buffered: the reduce of block k (RCCL, its own stream) overlaps the render of block k+1;
            # every block is still rendered, reduced and mastered inside the region it is counted in (drain)
            sh.begin([red, torch.zeros_like(red)])

            def run(n):
                for _ in range(n):
                    sh.step_overlapped(render_partial, master, out)
                sh.drain(master, out)                       # leaves the pair of buffers in place
        elif use_dist:
            def run(n):
                for _ in range(n):
                    sh.step(render_partial, master, red, out)   # render -> RCCL reduce -> master on rank 0
        else:
            def run(n):                                         # one GPU: render + (last reduction stage fused with) master
                for _ in range(n):
                    db.render_mix(frames, out.data_ptr(), 2, 0, interp, stream)
                db.wait_mix(stream)                             # the last block's tail joins the stream before the fence
        run(warmup)
        fence()
        db.timing_reset()
        t0 = time.perf_counter()
        run(steps)
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        k_mean, k_min, k_cnt = db.timing_summary()
        finite = bool(torch.isfinite(out).all().item()) if rank == 0 else True
        return dt, k_mean, k_min, k_cnt, finite

    def roofline(frames, k_mean, k_min, k_cnt):
        B = gather_bytes + (STATE_READ + STATE_WRITE) / frames       # algorithmic bytes / voice-sample
        launch_bytes = B * shard.n * frames
        achieved = launch_bytes / (k_mean * 1e-3)
        traffic = pmc_traffic(a.workload, shard.n, frames)      # HBM bytes per launch from the PMC passes (profiles/)
        return {"bound": "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": achieved / HBM_PEAK, "traffic": traffic,
                "traffic_rate": None if traffic is None else traffic / (k_mean * 1e-3) / 1e9,   # measured HBM GB/s of this kernel
                "traffic_frac": None if traffic is None else traffic / (k_mean * 1e-3) / HBM_PEAK,
                "kernel": KERNELS.get(db.last_kernel(), "?"), "frames_per_launch": frames,
                "kernel_ms_mean": k_mean, "kernel_ms_min": k_min, "launches_timed": k_cnt,
                "algorithmic_bytes_per_voice_sample": B, "algorithmic_bytes_per_launch": launch_bytes,
                "kernel_voice_samples_per_s": shard.n * frames / (k_mean * 1e-3)}

    # BASELINE.md §4 / SURVEY §8(d): "render 1 s after 0.1 s warm-up".  The recipe's note-ons are staggered over
    # the last second and attack+decay last 0.11 s, so those first 5280 frames are rendered here, untimed and
    # independent of --warmup: the timed region then starts with every voice in its sustain stage, as the
    # recipe intends (launches with voices still in attack/decay are ~3.5x slower: DESIGN.md §4).
    recipe_warmup_frames = 0
    if not a.no_recipe_warmup:
        scratch = torch.zeros(F, 2, device=dev, dtype=torch.float32)
        while recipe_warmup_frames < int(0.11 * 48000):
            db.render(F, scratch.data_ptr(), 0, interp, stream)
            recipe_warmup_frames += F
        fence()
    dt, k_mean, k_min, k_cnt, finite = timed(F, a.steps, a.warmup)

    res = None
    if rank == 0:
        value = total * F * a.steps / dt
        rl = roofline(F, k_mean, k_min, k_cnt)
        rl["note"] = ("LUTs are LDS-resident and the recurrences live in registers, so compulsory HBM traffic is one "
                      "state sweep per launch: at F=512 the kernel is bound by fp32 VALU issue (72% VALU-busy, "
                      "profiles/r01_v9_c3_pmc_summary.json), not by HBM; the HBM fraction grows as F shrinks (low_latency). "
                      "PCM banks (c4) gather from an L2-resident pool: profiles/r01_v9_c4_pmc_summary.json")
        where = ""
        if world > 1:
            where = f"; one such bank per GPU ({world} GPUs)" if a.scaling == "weak" else f"; split over {world} GPUs"
        res = {
            "metric": "voice-samples/s", "value": value, "unit": "voice-samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": a.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": DESCR[a.workload] + (f" [voices overridden: {bank_voices} per bank]" if a.voices else "") + where,
                       "voices_total": total, "voices_per_gpu": shard.n,
                       "frames_per_launch": F, "sample_rate": 48000, "interp": "linear" if interp else "truncate",
                       "parallelism": f"voices block-partitioned over {world} GPU(s)" + ("; one RCCL reduce(sum) of float[F][2] per launch" + (", overlapped with the next launch's render" if a.overlap and not a.no_overlap else "") if world > 1 else ""),
                       "seed": "0x5EED", "recipe_warmup_frames": recipe_warmup_frames},
            "realtime_factor_48k": value / (total * 48000.0),
            "output_finite": finite,
            "roofline": rl,
        }
    # secondary operating point: short callbacks (F = 64 frames = 1.33 ms at 48 kHz), where the per-launch
    # state sweep dominates and the kernel approaches the HBM roofline (single GPU only)
    if world == 1 and not a.no_low_latency and F != 64:
        k2 = max(50, a.steps)
        dt2, km2, kn2, kc2, _ = timed(64, k2, 20)
        ll = roofline(64, km2, kn2, kc2)
        ll["value"] = total * 64 * k2 / dt2
        ll["realtime_factor_48k"] = ll["value"] / (total * 48000.0)
        res["low_latency"] = ll
    # SURVEY 8(d) also asks for F = 4800 (100 ms blocks): the state sweep is amortised further, B = G + 292/4800
    if world == 1 and not a.no_low_latency and F != 4800:
        k3 = max(10, a.steps // 10)
        dt3, km3, kn3, kc3, _ = timed(4800, k3, 2)
        lb = roofline(4800, km3, kn3, kc3)
        lb["value"] = total * 4800 * k3 / dt3
        lb["realtime_factor_48k"] = lb["value"] / (total * 48000.0)
        res["long_block"] = lb
    # the fixed-point LUT path (include/skred_amd_fxpt.h; integer mix, exact): same voice count, int16 LUT pyramids in LDS,
    # linear interpolation, ADSR + smoother, no biquad (section 6 of DESIGN.md)
    if world == 1 and not a.no_low_latency and not a.no_fixed_point and a.workload != "c4":
        from skred_amd import fxbank
        fb, fpool, fcount0 = fxbank.bank_fx(bank_voices)
        fdb = fxbank.DeviceFxBank(bank_voices, local)
        fdb.set_tables(fpool)
        fdb.upload(fb)
        fdb.set_sample_count(fcount0)
        fmix = torch.zeros(F, 2, device=dev, dtype=torch.int64)
        for _ in range(12 + a.warmup):                      # 12 x 512 frames: the recipe's last note-on reaches sustain
            fdb.render(F, fmix.data_ptr(), 1, 0, stream)
        fence()
        t0 = time.perf_counter()
        kms = []
        for _ in range(a.steps):
            fdb.render(F, fmix.data_ptr(), 1, 0, stream)
        fence()
        fdt = time.perf_counter() - t0
        kms.append(fdb.last_render_ms())
        res["fixed_point"] = {"value": bank_voices * F * a.steps / fdt, "unit": "voice-samples/s", "dtype": "q15/u32/i64",
                              "frames_per_launch": F, "ms_per_step": fdt / a.steps * 1e3, "kernel": "sk_fx_render_kernel",
                              "kernel_ms_last": kms[-1], "mix_nonzero": bool((fmix != 0).any().item()),
                              "workload": "fixed-point analogue of the C2 recipe without the biquad (fxbank.bank_fx), linear interpolation"}
        fdb.close()
    if rank == 0:
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = cpu_baseline(recipe, interp)
        os.write(result_fd, (json.dumps(res) + "\n").encode())
    db.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
