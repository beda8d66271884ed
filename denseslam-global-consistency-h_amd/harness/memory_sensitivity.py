"""How much do the shape figures of the four memory curves (harness/memory_curves.py, tests/test_memory_shape.py) move with
the two knobs that were set from the reference's own logs -- the stereo-noise level of the synthetic input and max_age --
and how much with a DECISION of DESIGN.md section 5?  (if the knobs move the asserted ratios more than
the decision does, the shape pin cannot discriminate between decisions, and the documentation has to say so.)

  noise    in {0.25, 0.5, 1.0} px disparity noise (0.3 % gross mismatches throughout)
  max_age  in {0.6, 0.81, 1.0} x the origin run's fill time e0 (the reference: 259 / 320 = 0.81)
  decision Decay(forceAll = true) gated by `last_seen <= newest - minAge` (this build; what InfiniTamDriver::Decay passes)
           against (a) the same sweep WITHOUT the gate (min_decay_age 0: every block is swept on every call) and (b) DynSLAM's
           aged-list mode (forceAll = false: each visible list is decayed once, when it is minAge lists old), both at
           noise 0.5 px, max_age 0.81 e0

    python denseslam-global-consistency-h_amd/harness/memory_sensitivity.py [keyframes] > profiles/r03_memory_shape_sensitivity.json
"""
import json
import os
import sys


def main():
    sys.path.insert(0, ".")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from dslam_amd.harness import memory_curves as mc
    from dslam_amd.harness import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    noises, factors = (0.25, 0.5, 1.0), (0.6, 0.81, 1.0)
    frames = {}
    for noise in noises:  # (before this process touches the GPU: the worker pool forks)
        frames[noise] = mc.generate(synth.s_street(640, 480, stereo_noise_px=noise, outlier_frac=0.003), n, 16)
    eng = pkg.open_engine(0)
    wl = synth.s_street(640, 480)
    rows = []

    def four_ratios(curves, exhausted):
        sh = mc.shape_metrics({k: mc.to_units(v) for k, v in curves.items()}, exhausted)
        after = sh.get("window_decay_over_window_after_window_full") or {}
        return {"decay_over_origin_fill_time": sh["decay_over_origin_fill_time"], "window_decay_over_window_median": after.get("median"),
                "window_decay_peak_over_pool": sh["window_decay_peak_over_pool"], "window_peak_over_pool": sh["window_peak_over_pool"],
                "origin_fills_pool_at": sh["origin_fills_pool_at"], "decay_fills_pool_at": sh["decay_fills_pool_at"]}

    for noise in noises:
        base = {}
        for min_age, force_all, label in ((30, True, "gated full sweep (this build)"),) + (
                ((0, True, "full sweep without the last_seen gate"), (30, False, "aged-list mode (forceAll = false)")) if noise == 0.5 else ()):
            params = pkg.SceneParams(history_words=8, **wl.scene_kwargs)
            un = {}
            for name in ("memory", "memory_decay"):   # the un-windowed runs do not depend on max_age
                vd, sw = mc.MODES[name]
                un[name] = mc.run_mode(eng, pkg, wl, params, frames[noise], vd, min_age, 3, sw, 0, force_all=force_all)
            e0 = un["memory"][1] or len(un["memory"][0])
            for f in factors if (min_age == 30 and force_all) else (0.81,):
                max_age = max(1, int(round(f * e0)))
                params = pkg.SceneParams(history_words=(max_age + 64) // 64 + 1, **wl.scene_kwargs)
                curves = {k_: v[0] for k_, v in un.items()}
                exhausted = {k_: v[1] for k_, v in un.items()}
                for name in ("memory_slide_window", "memory_decay_slide_window"):
                    vd, sw = mc.MODES[name]
                    curves[name], exhausted[name] = mc.run_mode(eng, pkg, wl, params, frames[noise], vd, min_age, 3, sw, max_age, force_all=force_all)
                row = {"noise_px": noise, "max_age_over_e0": f, "max_age": max_age, "decay": label}
                row.update(four_ratios(curves, exhausted))
                rows.append(row)
                print(json.dumps(row), file=sys.stderr, flush=True)
    ref = json.load(open(os.path.join("tests", "golden", "reference_memory_shape.json")))
    print(json.dumps({"what": __doc__.strip().split("\n\n")[0], "keyframes": n, "rows": rows, "reference_shape": ref}, indent=1))


if __name__ == "__main__":
    main()
