#!/usr/bin/env python3
"""Distil gpurun_out/profiles_raw (see tools/collect_profiles.sh) into profiles/r1/.

HBM bytes per launch = FETCH_SIZE (KB, x2 on gfx950: the counter tallies 128-byte requests as 64 bytes --
MI355X_MICROARCH.md, HBM / rocprofv3 section) + WRITE_SIZE (KB), averaged over the launches of a kernel.
Template instances of one kernel (fftconv1_kernel<true>/<false>) are pooled under the bare name; the hand-scheduled
persistent kernels keep their own names (gfx_fftconv_pipe_t1_o8, ...), which is what bench.py's live timing hook
keys on (gfx_fftconv_last_kernel).
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "gpurun_out", "profiles_raw")
DST = os.path.join(ROOT, "profiles", os.environ.get("GRAFX_ROUND", "r6"))


def bare(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"<.*$", "", name)
    return name


def counter(dirname, cname):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(RAW, dirname, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                agg[bare(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    # bench.py renders one single graph before the warm-up (kernel loading); those launches move ~1/256 of the
    # bytes of a real one and are dropped so that the averages describe the measured workload
    for k, v in agg.items():
        top = max(v) if v else 0.0
        agg[k] = [x for x in v if x >= 0.02 * top]
    return agg


def main():
    os.makedirs(DST, exist_ok=True)
    bench = json.loads(open(os.path.join(RAW, "bench.json")).read().strip().splitlines()[-1])
    shutil.copy(os.path.join(RAW, "bench.json"), os.path.join(DST, f"bench_{os.environ.get('GRAFX_ROUND', 'r6')}.json"))
    shutil.copy(os.path.join(RAW, "bench_under_rocprof.json"), os.path.join(DST, "bench_under_rocprof.json"))
    stats = glob.glob(os.path.join(RAW, "trace", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(stats[0], os.path.join(DST, "rocprofv3_kernel_stats.csv"))
    def traffic(suffix=""):
        fetch, write = counter("pmc_fetch" + suffix, "FETCH_SIZE"), counter("pmc_write" + suffix, "WRITE_SIZE")
        out = {}
        for k in sorted(set(fetch) | set(write)):
            if not k.startswith(("gfx::", "gfx_")):
                continue
            f = sum(fetch[k]) / max(len(fetch[k]), 1)
            w = sum(write[k]) / max(len(write[k]), 1)
            out[k] = {
                "launches": len(fetch[k]),
                "FETCH_SIZE_KB_avg_per_launch": f,
                "WRITE_SIZE_KB_avg_per_launch": w,
                "hbm_read_bytes_per_launch_corrected": 2 * f * 1024,
                "hbm_write_bytes_per_launch": w * 1024,
                "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024,
            }
        return out

    kernels = traffic()
    # round 5: the same two passes for the console with long compressor poles / the ballistics smoother, and for the
    # ballistics recursion alone (tools/ballistics_bench.py: 9216 x 131072 rows) -- launches of different sizes are averaged
    # per kernel name, so read these next to the kernel statistics of the same runs
    for suffix in ("_longpole", "_ballistics", "_ballistics_rows", "_train"):
        if glob.glob(os.path.join(RAW, "pmc_fetch" + suffix, "**", "*counter_collection.csv"), recursive=True):
            json.dump({"note": "FETCH_SIZE doubled per MI355X_MICROARCH.md; averages over the launches of the run, the single-graph "
                               "warm-up launches dropped", "kernels": traffic(suffix)},
                      open(os.path.join(DST, f"pmc_hbm_traffic{suffix}.json"), "w"), indent=1)
    cfg = bench["config"]
    rec = {
        "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline",
        "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B); sanity check: "
                "dyn_oneshot_mix_kernel reads 4*R*C*L bytes and writes 4*(R + 5*B)*C*L",
        "config": {"batch": cfg["batch_per_gpu"], "audio_len": cfg["audio_len"], "fsm_fir_len": cfg["fsm_fir_len"],
                   "iir_len": cfg["iir_len"], "ir_len": cfg["ir_len"]},
        "kernels": kernels,
    }
    json.dump(rec, open(os.path.join(DST, "pmc_hbm_traffic.json"), "w"), indent=1)
    # the line's roofline, recomputed from the kernel trace of the same command: the dominant kernel's launches of the
    # timed region are its last launches_per_step * steps ones (nothing runs after it with --no-train --no-secondary
    # --no-sustained --no-cpu-baseline)
    under = json.loads(open(os.path.join(RAW, "bench_under_rocprof.json")).read().strip().splitlines()[-1])
    roof = under["roofline"]
    dom = roof["kernel"]
    rows = []
    for f in glob.glob(os.path.join(RAW, "trace", "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if bare(r["Kernel_Name"]) in (dom, "gfx::" + dom.split("<")[0]):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    n = roof["launches_per_step"] * under["steps"]
    last = rows[-n:]
    avg_ms = sum(d for _, d, _ in last) / len(last) / 1e6
    rec2 = {"what": f"launches of {dom} in the timed region of bench_under_rocprof.json, from the rocprofv3 kernel trace",
            "launches": len(last), "avg_launch_ms_trace": avg_ms, "avg_launch_ms_line": roof["avg_launch_ms"],
            "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
            "frac_trace": roof["algorithmic_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9 / roof["peak"], "frac_line": roof["frac"],
            "kernels": sorted({k for _, _, k in last})}
    json.dump(rec2, open(os.path.join(DST, "roofline_from_trace.json"), "w"), indent=1)
    print("roofline frac: line", round(roof["frac"], 4), "trace", round(rec2["frac_trace"], 4))
    k = kernels.get(dom) or kernels.get("gfx::" + dom.split("<")[0])
    print("bench:", bench["ms_per_step"], "ms/step", bench["value"], bench["unit"])
    print(dom, "HBM bytes/launch:", k and k["hbm_bytes_per_launch"], "algorithmic", roof["algorithmic_bytes_per_launch"])
    for name in ("gfx::dyn_oneshot_mix_kernel", "gfx::dyn_oneshot_kernel"):
        d = kernels.get(name)
        print(name, "HBM bytes/launch:", d and (d["hbm_read_bytes_per_launch_corrected"], d["hbm_write_bytes_per_launch"]))


if __name__ == "__main__":
    sys.exit(main())
