"""Collapsed timeline of the last training step in a rocprofv3 kernel trace:  python tools/step_timeline.py <dir with *kernel_trace.csv>
(kernels shorter than 0.1 ms are folded into runs of "small")."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else "gfx_fftconv_pipe_t1_o8"
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"] == anchor]
seg = rows[idx[-2]:idx[-1]]
t0 = int(seg[0]["Start_Timestamp"])
out, prev = [], None
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    name = re.sub(r"void |at::native::|\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
    k = name if d > 100000 else "small"
    if k == "small" and prev == "small":
        out[-1][2] += 1
        out[-1][3] += d
        out[-1][4] = (int(r["End_Timestamp"]) - t0) / 1e6
    else:
        out.append([(int(r["Start_Timestamp"]) - t0) / 1e6, k, 1, d, (int(r["End_Timestamp"]) - t0) / 1e6])
    prev = k
print(f"step {(int(rows[idx[-1]]['Start_Timestamp']) - t0) / 1e6:.2f} ms, {len(seg)} kernels, busy "
      f"{sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e6:.2f} ms")
for o in out:
    print(f"{o[0]:8.2f}-{o[4]:8.2f} {o[1]:72s} n={o[2]:3d} busy={o[3] / 1e6:.3f}")
