"""Where the envelope kernel sits in time relative to the steady kernel: reads a rocprofv3 kernel trace (csv) of `tools/ab.py live`
and prints, for a sample of blocks, start / end of collect, envelope and steady kernels relative to the steady kernel's start."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(path)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ev.sort()
def short(n):
    for k in ("sk_gain", "sk_render_fast2", "sk_render_env2", "sk_collect_scan", "sk_collect_expand", "sk_update", "sk_stamp", "sk_classify"):
        if k in n: return k[3:]
    return n[:24]
steady = [i for i, e in enumerate(ev) if "sk_render_fast2" in e[2]]
print(f"{len(ev)} dispatches, {len(steady)} steady kernels")
want = int(sys.argv[2]) if len(sys.argv) > 2 else 12
step = max(1, len(steady) // want)
for si in steady[::step]:
    s0, s1, _ = ev[si]
    line = [f"steady 0 .. {(s1 - s0) / 1e3:6.1f}"]
    for j in range(max(0, si - 8), min(len(ev), si + 8)):
        if j == si: continue
        a, b, n = ev[j]
        if b < s0 - 60000 or a > s1 + 200000: continue
        line.append(f"{short(n)} {(a - s0) / 1e3:6.1f} .. {(b - s0) / 1e3:6.1f}")
    nxt = ev[steady[steady.index(si) + 1]][0] if steady.index(si) + 1 < len(steady) else None
    if nxt: line.append(f"next steady at {(nxt - s0) / 1e3:6.1f}")
    print(" | ".join(line))
