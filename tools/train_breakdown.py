"""Per-kernel breakdown of the LAST training step in a rocprofv3 kernel trace of `bench.py --train`.

usage: python tools/train_breakdown.py <kernel_trace.csv> [--timeline]
"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = [i for i, r in enumerate(rows) if "fftconv1_kernel<true>" in r["Kernel_Name"] or "gfx_fftconv_pipe_t1" in r["Kernel_Name"]][-1]
sub = rows[first - 12:]
t0 = int(sub[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in sub)
agg = collections.defaultdict(lambda: [0, 0])


def short(n):
    return n.split("(")[0][-60:] if "at::native" not in n else "T:" + n[n.find("native::") + 8:][:70]


prev = t0
for r in sub:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[short(r["Kernel_Name"])][0] += d
    agg[short(r["Kernel_Name"])][1] += 1
    if "--timeline" in sys.argv and (d > 150000 or int(r["Start_Timestamp"]) - prev > 100000):
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e6:8.2f} {d / 1e6:7.3f} gap {(int(r['Start_Timestamp']) - prev) / 1e3:7.1f}us "
              f"{short(r['Kernel_Name'])}")
    prev = max(prev, int(r["End_Timestamp"]))
busy = sum(v[0] for v in agg.values())
print(f"window {(t1 - t0) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, {len(sub)} kernels")
for n, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"{v[0] / 1e6:8.2f} ms {v[1]:4d}  {n}")
