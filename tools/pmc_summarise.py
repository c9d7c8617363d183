#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/gpu_profile.sh: pmc_summary.json (+ derived figures) and hbm_traffic.json.
usage: python tools/pmc_summarise.py <output dir of gpu_profile.sh>"""
import sys
O = sys.argv[1]
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob(O + "/pmc_*")):
    if not os.path.isdir(d): continue
    # ONE launch per pass: the first trace kernel dispatch (a pass whose command launches more than once -- warm-ups, extra legs -- is
    # reduced to its first launch, so that every counter is "per launch")
    c = collections.defaultdict(float); dur = []; first = None
    rows = [row for f in glob.glob(d + "/*counter_collection.csv") for row in csv.DictReader(open(f)) if "trace_" in row["Kernel_Name"]]
    if rows:
        first = min(int(row["Dispatch_Id"]) for row in rows)
        for row in rows:
            if int(row["Dispatch_Id"]) == first: c[row["Counter_Name"]] += float(row["Counter_Value"])
    for f in glob.glob(d + "/*kernel_trace.csv"):
        ks = sorted(((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) for row in csv.DictReader(open(f)) if "trace_" in row["Kernel_Name"]))
        if ks: dur.append(ks[0][1])
    out[os.path.basename(d)] = {"counters": dict(c), "launches": len(dur), "kernel_ns": dur, "launches_in_pass": len(set(row["Dispatch_Id"] for row in rows))}
# derived figures of the headline kernel: per wave-segment = per 64 ray segments (segments from the bench line of the same workload)
try:
    # 1920 x 1080 x 1000 spp: the segments the kernel counted in that pass (the bench line of the pass: it depends on the numerics mode --
    # 4.119 per sample in the reference's evaluation order, 3.944 in the contract form of rounds 1-4)
    def segs_of(tag):
        line = [ln for ln in open(O + "/pmc_%s.log" % tag) if ln.startswith("{")][-1]
        d = json.loads(line)
        return d["segments_per_sample"] * 1920 * 1080 * 1000, d.get("numerics")
    segs, numerics = segs_of("f32_sqA")
    a, m = out["pmc_f32_sqA"]["counters"], out["pmc_f32_mfma"]["counters"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    out["derived_f32"] = {"numerics": numerics, "segments": segs, "valu_per_wave_segment": a["SQ_INSTS_VALU"] / (segs / 64), "mfma_per_wave_segment": m["SQ_INSTS_MFMA"] / (segs / 64),
                          "clock_GHz": cyc / out["pmc_f32_sqA"]["kernel_ns"][0], "mfma_busy_frac_of_simd_cycles": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                          "valu_busy_frac_at_2_cycles": a["SQ_INSTS_VALU"] * 2 / (1024 * cyc), "wait_any_frac_of_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]}
    wc = a["SQ_WAVE_CYCLES"]
    out["derived_f32"].update({"wait_inst_any_frac_of_wave_cycles": a["SQ_WAIT_INST_ANY"] / wc, "active_inst_any_frac_of_wave_cycles": a["SQ_ACTIVE_INST_ANY"] / wc,
                               "issue_busy": (m["SQ_VALU_MFMA_BUSY_CYCLES"] + 2 * a["SQ_INSTS_VALU"]) / (1024 * cyc)})
    out["derived_f32"].update({"salu_per_wave_segment": a["SQ_INSTS_SALU"] / (segs / 64), "lds_per_wave_segment": out["pmc_f32_sqB"]["counters"]["SQ_INSTS_LDS"] / (segs / 64)})
    ca, cm = out.get("pmc_f32_cull_sqA"), out.get("pmc_f32_cull_mfma")
    if ca and cm:
        ca, cm, ccyc = ca["counters"], cm["counters"], ca["counters"]["GRBM_GUI_ACTIVE"] / 8
        out["derived_f32_cull"] = {"valu_per_wave_segment": ca["SQ_INSTS_VALU"] / (segs / 64), "salu_per_wave_segment": ca["SQ_INSTS_SALU"] / (segs / 64),
                                   "mfma_per_wave_segment": cm["SQ_INSTS_MFMA"] / (segs / 64), "clock_GHz": ccyc / out["pmc_f32_cull_sqA"]["kernel_ns"][0],
                                   "mfma_busy_frac_of_simd_cycles": cm["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * ccyc), "valu_busy_frac_at_2_cycles": ca["SQ_INSTS_VALU"] * 2 / (1024 * ccyc)}
except Exception as e:
    out.setdefault("derived_f32", {})["error"] = str(e)
json.dump(out, open(O + "/pmc_summary.json", "w"), indent=1)
for k, v in out.items(): print(k, v if k.startswith("derived") else (v["launches"], {a: round(b, 3) for a, b in v["counters"].items()}, [round(x / 1e6, 2) for x in v["kernel_ns"]]))
# HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: on gfx950 it reports
# half of the bytes of wide reads); WRITE_SIZE is calibrated on this box (tools/ubench_write_size.hip: 1.00 x for whole-line stores, 32-byte sectors)
tr = {}
for key, tag in (("f32_1920x1080_1000spp_d50_plain", "f32"), ("f32_1920x1080_1000spp_d50_cull", "f32_cull"), ("f64_3840x2160_1000spp_d50_plain", "f64")):
    f, w = out.get("pmc_%s_fetch" % tag), out.get("pmc_%s_write" % tag)
    if f and w and f["launches"] and w["launches"]:
        fb = f["counters"].get("FETCH_SIZE", 0) / f["launches"] * 1024
        wb = w["counters"].get("WRITE_SIZE", 0) / w["launches"] * 1024
        tr[key] = {"hbm_bytes_per_launch": int(2 * fb + wb), "fetch_size_bytes_raw": int(fb), "write_size_bytes": int(wb),
                   "source": "rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE / --pmc GRBM_GUI_ACTIVE WRITE_SIZE, one launch each (tools/gpu_profile.sh)"}
json.dump(tr, open(O + "/hbm_traffic.json", "w"), indent=1)
print(json.dumps(tr, indent=1))
