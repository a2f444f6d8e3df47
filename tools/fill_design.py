"""Fills the measured numbers of tools/DESIGN.tpl (section 5) from the round's bench lines under profiles/ -> DESIGN.md.
usage: python tools/fill_design.py [r06]"""
import json
import os
import sys

R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def load(name):
    p = os.path.join(R, "profiles", name)
    if not os.path.exists(p):
        return None
    text = open(p).read()
    try:                                   # a full record (bench.py's detail file, indented) ...
        return json.loads(text)
    except ValueError:
        pass
    for line in text.split("\n"):          # ... or a one-line record behind other output
        if line.startswith("{"):
            return json.loads(line)
    return None


def main(tag="r06"):
    t = open(os.path.join(R, "tools", "DESIGN.tpl")).read()
    d = load(f"{tag}_bench_cfg2.json")
    drv = load(f"{tag}_bench_cfg2_driver_flags.json")
    c3, c5, dep = load(f"{tag}_bench_cfg3.json"), load(f"{tag}_bench_cfg5_n1.json"), load(f"{tag}_bench_cfg2_depth.json")
    g2, g5 = load(f"{tag}_bench_cfg2_gpus2_gloo.json"), load(f"{tag}_bench_cfg5_gpus2_gloo.json")
    f0 = lambda v: "n/a" if v is None else f"{v:,.0f}".replace(",", " ")
    rep = {"@CFG2@": f0(d["value"]), "@CFG2MS@": f"{d['ms_per_step']:.3f}", "@STEADY@": f0(d["steady_state"]["value"]),
           "@DRV@": f0(drv and drv["value"]), "@DRVSTEADY@": f0(drv and drv["steady_state"] and drv["steady_state"]["value"]),
           "@DEPTH@": f0(dep and dep["value"]), "@CFG3@": f0(c3 and c3["value"]),
           "@MULTI@": f0(d["multi_clip"]["value"]), "@MULTIX@": f"{d['multi_clip']['vs_single_clip']:.2f}",
           "@MULTIF@": f"{d['multi_clip']['roofline']['frac']:.3f}", "@CFG5@": f0(c5 and c5["value"]),
           "@G2@": f0(g2 and g2["value"]), "@G5@": f0(g5 and g5["value"]),
           "@CPU@": f"{d['cpu_baseline']['value']:.2f}", "@CPUOFF@": f"{d['cpu_baseline']['value_logging_off']:.2f}",
           "@RATIO@": f0(d["steady_state"]["value"] / d["cpu_baseline"]["value"]),
           "@WHOLE@": f"{100 * d['steady_state']['roofline']['whole_iteration']['frac']:.1f} %"}
    po = load(f"{tag}_bench_poseinit.json")
    loops = (po or {}).get("config", {}).get("pose_steps_per_s_by_loop", {})
    rep.update({"@POSE@": f0(po and po["value"]), "@POSEMS@": "n/a" if not po else f"{po['ms_per_step']:.2f}",
                "@POSEFIT@": "n/a" if not po else f"{po['seconds_per_fit']:.3f}",
                "@POSELOOPS@": ", ".join(f"{k} {f0(v)}" for k, v in loops.items()) or "n/a",
                "@POSECPU@": f0(po and po["cpu_baseline"]["value"])})
    e2e, mixed, fr = d.get("end_to_end"), load(f"{tag}_bench_mixed_shard.json"), load(f"{tag}_freerun_cfg2_400.json")
    rep.update({"@MIXED@": f0(mixed and mixed["its_per_s"]),
                "@E2E@": "n/a" if not e2e else f"{e2e['clips_per_s_end_to_end']:.1f}",
                "@E2ERES@": "n/a" if not e2e else f"{e2e['repeated_shape']['clips_per_s']:.1f}",
                "@E2ESETUP@": "n/a" if not e2e else f"{100 * e2e['repeated_shape']['setup_fraction_of_fit']:.1f} %",
                "@E2ESPLIT@": "n/a" if not e2e else ", ".join(f"{k} {v:.3f} s" for k, v in e2e["split_s"].items()),
                "@FREEEQ@": "n/a" if not fr else str(fr["object_params_bit_equal_all_steps"]),
                "@FREEALL@": "n/a" if not fr else str(fr.get("all_params_bit_equal_all_steps")),
                "@FREELOSS@": "n/a" if not fr else f"{fr['max_rel_loss']:.1e}",
                "@FREEVO@": "n/a" if not fr else f"{fr['final_vertex_diff_mm']['object']:.1e}",
                "@FREEVH@": "n/a" if not fr else f"{fr['final_vertex_diff_mm']['hand']:.1e}"})
    pk = ((po or {}).get("roofline") or {}).get("kernels", {})
    for key, name in (("PISWP", "k_bwd_sweep"), ("PIRAS", "k_raster_fwd"), ("PILIN", "k_bwd_lines")):
        k = pk.get(name)
        rep[f"@{key}@"] = "n/a" if not k else f"{k['avg_launch_us']:.0f}"
        rep[f"@{key}G@"] = "n/a" if not k else f"{k['achieved_GBps']:.0f}"
        rep[f"@{key}F@"] = "n/a" if not k else f"{k['achieved_GBps'] / 8000:.2f}"
    ks = d["steady_state"]["roofline"]["kernels"]
    for key, name in (("RAS", "k_raster_fwd"), ("SWP", "k_bwd_sweep"), ("LIN", "k_bwd_lines")):
        k = ks[name]
        rep[f"@{key}@"] = f"{k['avg_launch_us']:.1f}"
        rep[f"@{key}G@"] = f"{k['achieved_GBps']:.0f}"
        rep[f"@{key}F@"] = f"**{k['achieved_GBps'] / 8000:.3f}**"
        rep[f"@{key}T@"] = f"{k['traffic_bytes'] / 1e6:.1f}" if k.get("traffic_bytes") else "n/a"
        rep[f"@{key}V@"] = f"{k['valu_wave_instr'] / 1e6:.1f} M" if k.get("valu_wave_instr") else "n/a"
        rep[f"@{key}VF@"] = f"{k['valu_frac']:.2f} / {k.get('valu_frac_4cyc', 0.0):.2f}" if k.get("valu_frac") else "n/a"
    st = d["steady_state"]["roofline"].get("raster_stage_8d")
    rep["@STAGEF@"] = "n/a" if not st else f"{st['frac']:.3f} = {st['algorithmic_bytes'] / 1e6:.1f} MB in {st['kernels_us']:.1f} µs"
    for a, b in rep.items():
        t = t.replace(a, b)
    import re
    left = re.findall(r"@[A-Z0-9]+@", t)
    assert not left, left
    open(os.path.join(R, "DESIGN.md"), "w").write(t)
    print("DESIGN.md written,", len(t.split("\n")), "lines")


if __name__ == "__main__":
    main(*sys.argv[1:])
