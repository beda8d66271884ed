#!/usr/bin/env python3
"""Turn the raw outputs of scratch/collect_profiles.sh (run on the GPU box, merged back under gpurun_out/) into the
per-round files of this directory.  Usage: python profiles/make_profiles.py r01

gpurun_out/final_bench.json         <- python bench.py
gpurun_out/final_bench_sync.json    <- python bench.py --sync --steps 100 --warmup 10 --no-cpu-baseline --reint 0
gpurun_out/final_stats/             <- rocprofv3 --kernel-trace --stats ... bench.py --steps 100 --warmup 10 --no-cpu-baseline --reint 0
gpurun_out/final_fetch|final_write/ <- rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace ... bench.py --steps 30 --warmup 5
gpurun_out/final_stress.json        <- python denseslam-global-consistency-h_amd/harness/stress.py 64
gpurun_out/final_side_bench.json    <- python denseslam-global-consistency-h_amd/harness/side_bench.py 50
gpurun_out/final_quality.json       <- python denseslam-global-consistency-h_amd/harness/quality.py 40
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
HERE = os.path.join(ROOT, "profiles")


def newest(pattern):
    files = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    if not files:
        sys.exit(f"missing {pattern} under gpurun_out/")
    return files[-1]


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    sys.exit(f"no JSON line in {path}")


def counter_mean(dirname, counter, skip):
    rows = [r for r in csv.DictReader(open(newest(f"{dirname}/*/*counter_collection.csv")))
            if r["Counter_Name"] == counter and "k_integrate" in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows][skip:]
    return sum(vals) / len(vals), len(vals)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    bench = last_json_line(os.path.join(OUT, "final_bench.json"))
    json.dump(bench, open(os.path.join(HERE, f"{tag}_bench_default.json"), "w"), indent=1)
    json.dump(last_json_line(os.path.join(OUT, "final_bench_sync.json")),
              open(os.path.join(HERE, f"{tag}_bench_sync.json"), "w"), indent=1)
    shutil.copy(newest("final_stats/*/*kernel_stats.csv"), os.path.join(HERE, f"{tag}_bench_steps100_kernel_stats.csv"))
    json.dump(last_json_line(os.path.join(OUT, "final_stress.json")),
              open(os.path.join(HERE, f"{tag}_stress_integrate.json"), "w"), indent=1)
    hostio = os.path.join(OUT, "final_bench_hostio.json")
    if os.path.exists(hostio):
        json.dump(last_json_line(hostio), open(os.path.join(HERE, f"{tag}_bench_hostio.json"), "w"), indent=1)
    side = os.path.join(OUT, "final_side_bench.json")
    if os.path.exists(side):
        json.dump(last_json_line(side), open(os.path.join(HERE, f"{tag}_side_bench.json"), "w"), indent=1)
    maint = os.path.join(OUT, "final_maintenance.json")
    if os.path.exists(maint):
        json.dump(last_json_line(maint), open(os.path.join(HERE, f"{tag}_maintenance.json"), "w"), indent=1)
    quality = os.path.join(OUT, "final_quality.json")
    if os.path.exists(quality):
        json.dump(last_json_line(quality), open(os.path.join(HERE, f"{tag}_quality.json"), "w"), indent=1)

    # HBM traffic of k_integrate from the two PMC passes (MI355X_MICROARCH.md, "HBM / rocprofv3"): counters are in KiB;
    # on gfx950 FETCH_SIZE counts wide (16 B per lane) streaming reads at half their size -> doubled; WRITE_SIZE is exact.
    warm = 5
    run = last_json_line(os.path.join(OUT, "final_fetch.log"))
    fetch, n = counter_mean("final_fetch", "FETCH_SIZE", warm)
    write, n2 = counter_mean("final_write", "WRITE_SIZE", warm)
    vis = run["config"]["visible_blocks_per_frame"]
    traffic = (2.0 * fetch + write) * 1024.0
    algo = 8212.0 * vis + 8.0 * 640 * 480
    json.dump({
        "kernel": "k_integrate<false,true>",
        "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py "
                   "--steps 30 --warmup 5 --no-cpu-baseline --reint 0 (two separate passes; the 30 timed launches)",
        "launches": min(n, n2),
        "FETCH_SIZE_KiB_mean": fetch,
        "WRITE_SIZE_KiB_mean": write,
        "visible_blocks_per_launch": vis,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads -> doubled "
                      "(MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "traffic_bytes_per_launch": traffic,
        "traffic_bytes_per_visible_block": traffic / vis,
        "algorithmic_bytes_per_launch": algo,
    }, open(os.path.join(HERE, f"{tag}_integrate_pmc.json"), "w"), indent=1)
    print(f"{tag}: {bench['value']:.0f} frames/s, roofline frac {bench['roofline']['frac']:.3f}, "
          f"traffic/algorithmic {traffic / algo:.3f}")


if __name__ == "__main__":
    main()
