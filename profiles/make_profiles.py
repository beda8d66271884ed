#!/usr/bin/env python3
"""Turn the raw outputs of profiles/collect_profiles.sh (run on the GPU box, merged back under gpurun_out/) into the
per-round files of this directory.  Usage: python profiles/make_profiles.py r03

gpurun_out/final_bench.json              <- python bench.py                      (the driver's command)
gpurun_out/final_bench_sync.json         <- python bench.py --mode sync ...
gpurun_out/final_stats[_device]/         <- rocprofv3 --kernel-trace --stats ... bench.py [--mode device]
gpurun_out/final_fetch|final_write/      <- rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace ... bench.py --mode device
gpurun_out/final_sq/                     <- rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
                                            SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace ... bench.py --mode device
gpurun_out/final_integrate_waves.bin     <- DSLAM_DBG_INTEGRATE dump of k_integrate (per wave: 12 timestamps, clocks, placement)
gpurun_out/final_wavetime.bin            <- DSLAM_DBG_WAVETIME dump of k_render (per wave: cycles, march length, ...)
gpurun_out/final_sweep.bin               <- DSLAM_DBG_SWEEP dump of k_alloc_sweep (per tile: 8 timestamps)
gpurun_out/final_{stress,pipeline,side_bench,quality,maintenance,shard_emulation}.json
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
HERE = os.path.join(ROOT, "profiles")


def newest(pattern, required=True):
    files = sorted(glob.glob(os.path.join(OUT, pattern)), key=os.path.getmtime)
    if not files:
        if required:
            sys.exit(f"missing {pattern} under gpurun_out/")
        return None
    return files[-1]


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    sys.exit(f"no JSON line in {path}")


def counter_rows(dirname, counter, kernel):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(newest(f"{dirname}/*/*counter_collection.csv")))
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]]


def copy_json(src, dst):
    p = os.path.join(OUT, src)
    if os.path.exists(p) and os.path.getsize(p) > 2:
        json.dump(last_json_line(p), open(os.path.join(HERE, dst), "w"), indent=1)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    bench = last_json_line(os.path.join(OUT, "final_bench.json"))
    json.dump(bench, open(os.path.join(HERE, f"{tag}_bench_default.json"), "w"), indent=1)
    copy_json("final_bench_sync.json", f"{tag}_bench_sync.json")
    shutil.copy(newest("final_stats/*/*kernel_stats.csv"), os.path.join(HERE, f"{tag}_bench_steps100_kernel_stats.csv"))
    dev = newest("final_stats_device/*/*kernel_stats.csv", required=False)
    if dev:
        shutil.copy(dev, os.path.join(HERE, f"{tag}_bench_device_steps100_kernel_stats.csv"))
    for src, dst in (("final_stress.json", "stress_integrate"), ("final_pipeline.json", "pipeline_breakdown"),
                     ("final_side_bench.json", "side_bench"), ("final_maintenance.json", "maintenance"),
                     ("final_quality.json", "quality"), ("final_shard_emulation.json", "shard_emulation"),
                     ("final_mirror.json", "mirror_bench")):
        copy_json(src, f"{tag}_{dst}.json")
    pv = os.path.join(OUT, "push_variants.json")   # (profiles/experiments/push_variants.sh: an indented JSON list)
    if os.path.exists(pv):
        rows = json.load(open(pv))
        json.dump({"what": "k_integrate<false,true,true> on the S-stress map (V = 262144), kernel time by packet-attached events, per build "
                           "of integrate.hip with -DDSLAM_PUSH_VARIANT (an experiment macro of round 4, since removed): 0 = the round-3 product "
                           "(ring bit as a device-scope atomic from the gathering lane + last_seen store), 1 = no push, 2 = ring bit only, "
                           "3 = last_seen only, 4 = load-OR-store instead of the atomic, 5 / 6 = 0 / 4 behind the group's block stores; "
                           "process_frame_us = launched by ProcessFrame (push on), integrate_into_scene_us = no push in any variant",
                   "runs": rows}, open(os.path.join(HERE, f"{tag}_push_variants.json"), "w"), indent=1)

    # HBM traffic of k_integrate from the two PMC passes (MI355X_MICROARCH.md, "HBM / rocprofv3"): counters are in KiB;
    # on gfx950 FETCH_SIZE counts wide (16 B per lane) streaming reads at half their size -> doubled; WRITE_SIZE is exact.
    warm = 5
    run = last_json_line(os.path.join(OUT, "final_fetch.log"))
    fetch = counter_rows("final_fetch", "FETCH_SIZE", "k_integrate")[warm:]
    write = counter_rows("final_write", "WRITE_SIZE", "k_integrate")[warm:]
    vis = run["config"]["visible_blocks_per_frame"]
    fetch_m, write_m = sum(fetch) / len(fetch), sum(write) / len(write)
    traffic = (2.0 * fetch_m + write_m) * 1024.0
    algo = 8212.0 * vis + 8.0 * 640 * 480
    pmc = {
        "kernel": "k_integrate<false,true>",
        "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 bench.py --steps 30 "
                   "--warmup 5 --no-cpu-baseline --no-stress --no-extra-rates --reint 0 --mode device (two separate passes; the 30 timed launches)",
        "launches": min(len(fetch), len(write)), "FETCH_SIZE_KiB_mean": fetch_m, "WRITE_SIZE_KiB_mean": write_m,
        "visible_blocks_per_launch": vis,
        "correction": "gfx950: FETCH_SIZE reports 1/2 of wide (16 B/lane) streaming reads -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact",
        "traffic_bytes_per_launch": traffic, "traffic_bytes_per_visible_block": traffic / vis,
        "algorithmic_bytes_per_launch": algo,
    }
    # instruction counters: VALU instructions per wavefront of the launch = per voxel block (one block per wave at this V)
    if newest("final_sq/*/*counter_collection.csv", required=False):
        valu = counter_rows("final_sq", "SQ_INSTS_VALU", "k_integrate")[warm:]
        waves = counter_rows("final_sq", "SQ_WAVES", "k_integrate")[warm:]
        if valu and waves:
            pmc["SQ_INSTS_VALU_mean"] = sum(valu) / len(valu)
            pmc["SQ_WAVES_mean"] = sum(waves) / len(waves)
            pmc["valu_instructions_per_visible_block"] = (sum(valu) / len(valu)) / vis
            pmc["sq_command"] = "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU --kernel-trace ... (same bench command, its own pass)"
        # where the waves' time goes (quad-cycle counters, MI355X_MICROARCH.md "rocprofv3 PMC slots"): parked on
        # s_waitcnt / barrier, stalled at issue, issuing -- for the kernels of the frame
        breakdown = {}
        for kern in ("k_integrate", "k_render", "k_alloc_sweep", "k_mark", "k_bits_select", "k_fill_range_tiles"):
            c = {n: counter_rows("final_sq", n, kern)[warm:] for n in
                 ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU",
                  "SQ_INSTS_SALU", "SQ_WAVES")}
            if not c["SQ_WAVE_CYCLES"]:
                continue
            m = {n: sum(v) / len(v) for n, v in c.items() if v}
            wc = m["SQ_WAVE_CYCLES"]
            breakdown[kern] = {"waves": m.get("SQ_WAVES"), "valu_instructions_per_wave": m["SQ_INSTS_VALU"] / m["SQ_WAVES"],
                               "salu_instructions_per_wave": m["SQ_INSTS_SALU"] / m["SQ_WAVES"],
                               "wave_lifetime_cycles_mean": 4.0 * wc / m["SQ_WAVES"],
                               "frac_parked_on_waitcnt_or_barrier": m["SQ_WAIT_ANY"] / wc,
                               "frac_stalled_at_issue": m["SQ_WAIT_INST_ANY"] / wc, "frac_issuing": m["SQ_ACTIVE_INST_ANY"] / wc,
                               "valu_busy_quad_cycles_per_simd": m["SQ_ACTIVE_INST_VALU"] / 1024.0}
        if breakdown:
            pmc["wave_time_breakdown"] = breakdown
    json.dump(pmc, open(os.path.join(HERE, f"{tag}_integrate_pmc.json"), "w"), indent=1)

    # per-wave dump of the ray march: 6 u64 per single-wave workgroup {cycles, longest march of its lanes, iterations in
    # which a lane read a straddling cell, their cycles, setup cycles, refinement cycles}
    wt = os.path.join(OUT, "final_wavetime.bin")
    if os.path.exists(wt):
        d = np.fromfile(wt, dtype=np.uint64).reshape(-1, 6).astype(np.float64)
        d = d[d[:, 0] > 0]
        us = d[:, 0] / 2400.0
        json.dump({
            "kernel": "k_render<1,false,DIAG>", "source": "DSLAM_DBG_WAVETIME dump of the 30th launch (shader clock, 2.4 ticks per ns)",
            "waves": int(len(d)), "simds_on_chip": 1024,
            "march_steps_longest_lane": {"mean": float(d[:, 1].mean()), "p50": float(np.median(d[:, 1])), "p99": float(np.percentile(d[:, 1], 99)), "max": float(d[:, 1].max())},
            "wave_lifetime_us": {"mean": float(us.mean()), "p99": float(np.percentile(us, 99)), "max": float(us.max())},
            "sum_of_lifetimes_over_simds_us": float(us.sum() / 1024.0),
            "iterations_with_a_straddling_cell_frac": float(d[:, 2].sum() / max(1.0, d[:, 1].sum())),
            "cycles_per_plain_iteration": float((d[:, 0] - d[:, 3] - d[:, 4] - d[:, 5]).sum() / max(1.0, (d[:, 1] - d[:, 2]).sum())),
            "cycles_per_straddling_iteration": float(d[:, 3].sum() / max(1.0, d[:, 2].sum())),
        }, open(os.path.join(HERE, f"{tag}_render_wave_dump.json"), "w"), indent=1)
    # per-wave timeline of the fusion kernel: 16 u64 per wave (see IntegrateParams::dbg_waves), s_memrealtime = 10 ns ticks
    iw = os.path.join(OUT, "final_integrate_waves.bin")
    if os.path.exists(iw):
        a = np.fromfile(iw, dtype=np.uint64).reshape(-1, 16).astype(np.int64)
        act = a[:, 11] > 0
        w = a[act]
        t0 = a[a[:, 0] > 0, 0].min()
        names = ["entry", "table_ready", "list_length_known", "entries_gathered", "h0_chunk0_updated", "h0_chunk1_updated",
                 "h0_colour_done", "h0_stores_issued", "h1_chunk0_updated", "h1_chunk1_updated", "h1_colour_done", "h1_stores_issued"]
        us = lambda x: (x - t0) * 0.01
        pct = lambda t: {k: round(float(v), 2) for k, v in zip(("min", "p10", "median", "p90", "max"),
                                                                 [t.min(), np.percentile(t, 10), np.median(t), np.percentile(t, 90), t.max()])}
        d = np.diff(w[:, :12], axis=1) * 0.01
        life = (w[:, 11] - w[:, 0]) * 0.01
        xcc = (w[:, 15] >> 32) & 0xf
        ends = us(w[:, 11])
        json.dump({
            "kernel": "k_integrate<false,true,PLAIN,DIAG>", "source": "DSLAM_DBG_INTEGRATE dump of the 60th fusion launch of the bench loop (s_memrealtime, 10 ns ticks; us since the first wave's entry)",
            "waves_with_a_block": int(act.sum()), "waves_without": int(((a[:, 0] > 0) & ~act).sum()),
            "reached_at_us": {n: pct(us(w[:, k])) for k, n in enumerate(names)},
            "phase_median_us": {names[k + 1]: round(float(np.median(d[:, k])), 2) for k in range(11)},
            "wave_lifetime_us": pct(life),
            "shader_clock_GHz_median": round(float(np.median((w[:, 13] - w[:, 12]) / (life * 1e3))), 3),
            "entry_median_us_per_xcc": {int(x): round(float(np.median(us(w[xcc == x, 0]))), 2) for x in np.unique(xcc)},
            "waves_still_running_at_us": {str(t): int(((us(w[:, 0]) <= t) & (ends > t)).sum()) for t in range(0, int(ends.max()) + 2, 2)},
        }, open(os.path.join(HERE, f"{tag}_integrate_wave_timeline.json"), "w"), indent=1)
    sw = os.path.join(OUT, "final_sweep.bin")
    if os.path.exists(sw):
        subprocess.run([sys.executable, os.path.join(HERE, "experiments", "sweep_timeline3.py"), sw,
                        os.path.join(HERE, f"{tag}_sweep_timeline.json")], stdout=subprocess.DEVNULL, check=True)
    sel = os.path.join(OUT, "final_select.bin")
    if os.path.exists(sel):
        subprocess.run([sys.executable, os.path.join(HERE, "experiments", "select_timeline.py"), sel,
                        os.path.join(HERE, f"{tag}_select_timeline.json")], stdout=subprocess.DEVNULL, check=True)
    mk = os.path.join(OUT, "final_mark.bin")
    if os.path.exists(mk):
        d = np.fromfile(mk, dtype=np.uint64).reshape(-1, 4)
        n_pix = 1200   # 640 x 480 / 256
        re_, pix = d[:len(d) - n_pix].astype(np.int64), d[len(d) - n_pix:]
        t = pix.astype(np.int64)
        t3 = (pix[:, 3] & np.uint64((1 << 56) - 1)).astype(np.int64)
        a, b, c = (t[:, 2] - t[:, 0]) / 2400.0, (t3 - t[:, 2]) / 2400.0, (t[:, 1] - t3) / 2400.0
        rl = (re_[:, 1] - re_[:, 0]) / 2400.0
        json.dump({"kernel": "k_mark", "source": "DSLAM_DBG_MARK dump of the 60th pass of the bench loop: 4 stamps of the first wave of every workgroup "
                   "(s_memtime, taken as 2.4 GHz; counters of different CUs are not aligned: only differences inside a workgroup)",
                   "retest_workgroups": int(len(re_)), "retest_life_us": {"mean": round(float(rl.mean()), 2), "max": round(float(rl.max()), 2)},
                   "pixel_workgroups": n_pix,
                   "pixel_first_wave_us": {"entry_to_depth_pixel": round(float(a.mean()), 2), "walk": round(float(b.mean()), 2), "tail": round(float(c.mean()), 2),
                                           "total_mean": round(float((a + b + c).mean()), 2), "total_p99": round(float(np.percentile(a + b + c, 99)), 2),
                                           "total_max": round(float((a + b + c).max()), 2)}},
                  open(os.path.join(HERE, f"{tag}_mark_timeline.json"), "w"), indent=1)
    rq = newest("final_reint_sq/*/*counter_collection.csv", required=False)
    if rq:
        import collections
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(rq)):
            k = r["Kernel_Name"].split("(")[0]
            if "k_reintegrate_blocks" in k or "k_batch_" in k:
                acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
        out = {"command": "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU "
                          "--kernel-trace -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-stress --no-extra-rates --mode device "
                          "(the re-integration leg: 32 keyframes of the 110-keyframe map)", "kernels": {}}
        for k, c in acc.items():
            n = len(disp[k])
            m = {kk: v / n for kk, v in c.items()}
            wc = m.get("SQ_WAVE_CYCLES", 0.0)
            if not wc:
                continue
            out["kernels"][k] = {"launches": n, "waves": m["SQ_WAVES"], "valu_instructions": m["SQ_INSTS_VALU"], "salu_instructions": m["SQ_INSTS_SALU"],
                                 "frac_of_wave_time_with_a_valu_instruction_active": round(m["SQ_ACTIVE_INST_VALU"] / wc, 3),
                                 "frac_issuing": round(m["SQ_ACTIVE_INST_ANY"] / wc, 3), "frac_stalled_at_issue": round(m["SQ_WAIT_INST_ANY"] / wc, 3),
                                 "frac_parked_on_waitcnt_or_barrier": round(m["SQ_WAIT_ANY"] / wc, 3)}
        # HBM-side traffic of the same launches (two separate single-counter passes) and the batch's own counts
        rf = newest("final_reint_fetch/*/*counter_collection.csv", required=False)
        rw = newest("final_reint_write/*/*counter_collection.csv", required=False)
        ri = (bench.get("reintegration") or {}).get("block_major", {})
        if rf and rw:
            def per_kernel(path, name):
                acc, disp = collections.defaultdict(float), collections.defaultdict(set)
                for r in csv.DictReader(open(path)):
                    k = r["Kernel_Name"].split("(")[0]
                    if r["Counter_Name"] == name and ("k_reintegrate_blocks" in k or "k_batch_" in k):
                        acc[k] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
                return {k: v / len(disp[k]) for k, v in acc.items()}
            fe, wr = per_kernel(rf, "FETCH_SIZE"), per_kernel(rw, "WRITE_SIZE")
            for k in out["kernels"]:
                if k in fe and k in wr:
                    out["kernels"][k].update({"FETCH_SIZE_KiB": fe[k], "WRITE_SIZE_KiB": wr[k]})
            out["traffic_command"] = "the same bench command under rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE (one counter per pass)"
            blk = [k for k in out["kernels"] if "k_reintegrate_blocks" in k]
            if blk and ri.get("blocks_touched"):
                kb = out["kernels"][blk[0]]
                nb, nops = ri["blocks_touched"], ri["block_operations"]
                alg = nb * 8192.0
                # FETCH_SIZE halves wide (16 B per lane) reads on gfx950 (MI355X_MICROARCH.md): the block loads are wide, the image
                # gathers are dwords -- both bounds are given
                lo = (kb["FETCH_SIZE_KiB"] + kb["WRITE_SIZE_KiB"]) * 1024.0
                hi = (2.0 * kb["FETCH_SIZE_KiB"] + kb["WRITE_SIZE_KiB"]) * 1024.0
                kb.update({"blocks_touched": nb, "block_operations": nops, "valu_instructions_per_block_operation": kb["valu_instructions"] / nops,
                           "algorithmic_bytes_blocks_once": alg, "algorithmic_bytes_of_the_per_keyframe_loop": nops * 8212.0,
                           "traffic_bytes_lower_bound": lo, "traffic_bytes_if_every_read_were_wide": hi,
                           "write_bytes_per_touched_block": kb["WRITE_SIZE_KiB"] * 1024.0 / nb, "fetch_bytes_per_touched_block_lower_bound": kb["FETCH_SIZE_KiB"] * 1024.0 / nb,
                           "note": "writes = the changed 16-byte chunks of every touched block, once (<= 4 KiB per block); fetches = the 4 KiB of every block "
                                   "once PLUS the depth / colour gathers of every operation: 32 keyframes' images (78 MB) are in flight at once "
                                   "and miss the 4 MiB L2s, where a per-keyframe launch gathers from ONE 2.4 MB image pair -- the L2-miss traffic of "
                                   "the gathers (served by the Infinity Cache) is what FETCH_SIZE shows above the blocks' own 4 KiB"})
        json.dump(out, open(os.path.join(HERE, f"{tag}_reintegration_pmc.json"), "w"), indent=1)
    ms_src = os.path.join(OUT, "final_memory_sensitivity.json")
    if os.path.exists(ms_src) and os.path.getsize(ms_src) > 2:
        json.dump(json.load(open(ms_src)), open(os.path.join(HERE, f"{tag}_memory_shape_sensitivity.json"), "w"), indent=1)
    rs = newest("final_reint_stats/*/*kernel_stats.csv", required=False)
    if rs:
        shutil.copy(rs, os.path.join(HERE, f"{tag}_reintegration_kernel_stats.csv"))
    # the maintenance path per kernel: the stats file as it is, and -- from the trace of the same run -- the longest launches
    # of the streaming kernels, which are the calls on the S-stress map (1 GiB of voxels, 262144 blocks): GB/s against the peak
    ms = newest("final_maint_stats/*/*kernel_stats.csv", required=False)
    mt = newest("final_maint_stats/*/*kernel_trace.csv", required=False)
    if ms and mt:
        shutil.copy(ms, os.path.join(HERE, f"{tag}_maintenance_kernel_stats.csv"))
        import collections
        durs = collections.defaultdict(list)
        for r in csv.DictReader(open(mt)):
            durs[r["Kernel_Name"].split("(")[0]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
        gib = 262144 * 4096
        table = {}
        for kern, nbytes, what in (("dslam::k_decay_blocks", gib, "read-only sweep of every block (4 KiB read per block)"),
                                   ("dslam::k_release_and_leaders", gib, "release of every block (4 KiB reset per block)"),
                                   ("dslam::k_fill_voxels", gib, "ResetScene's fill"),
                                   ("void dslam::k_integrate<true, true, false, false>", 2 * gib + 262144 * 20, "de-integration of every block"),
                                   ("void dslam::k_integrate<false, true, true, false>", 2 * gib + 262144 * 20, "fusion into every block")):
            v = sorted(durs.get(kern, []), reverse=True)
            if not v:
                continue
            # (k_decay_blocks: its longest launch is the sweep that also resets every voxel -- 2 GiB; the read-only ones follow)
            pick = v[1] if kern == "dslam::k_decay_blocks" and len(v) > 1 else v[0]
            table[kern] = {"what": what, "launch_us": round(pick, 1), "algorithmic_bytes": nbytes,
                           "GBps": round(nbytes / pick / 1e3, 1), "frac_of_8TBps": round(nbytes / pick / 1e3 / 8000.0, 3),
                           "longest_launches_us": [round(x, 1) for x in v[:5]]}
        mj = os.path.join(HERE, f"{tag}_maintenance.json")
        if os.path.exists(mj):
            d = json.load(open(mj))
            d["kernels_on_the_s_stress_map"] = table
            d["per_kernel_summary"] = f"profiles/{tag}_maintenance_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same script)"
            json.dump(d, open(mj, "w"), indent=1)
    print(f"{tag}: {bench['value']:.0f} frames/s, roofline frac {bench['roofline']['frac']:.3f}, "
          f"traffic/algorithmic {traffic / algo:.3f}")


if __name__ == "__main__":
    main()
