#!/bin/bash
# The round's whole measurement set in ONE call on the GPU box (same box, same binary): what the driver runs at round end
# (smoke, pytest -m gpu, bench) plus the rocprofv3 summaries that are committed under profiles/.
#   usage: tools/gpu_profile.sh [tag=round4] [notests]
# Every rocprofv3 pass is its own run: --kernel-trace --stats only, or --pmc only with --kernel-trace (FETCH_SIZE and WRITE_SIZE
# do not fit one pass; MI355X_MICROARCH.md, HBM section).  Output: gpurun_out/<tag>/ -- kernel_stats CSVs, pmc_summary.json,
# hbm_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE per launch, KiB -> bytes), bench_*.json.  Copy what is to be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; TAG=${1:-round4}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
[ "$2" = "notests" ] || timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B32="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
B64="python $R/bench.py --dtype f64 --width 3840 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32 -o t -- $B32 > $O/trace_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_cull -o t -- $B32 --group-cull > $O/trace_f32_cull.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_valu -o t -- $B32 --scan-valu > $O/trace_f32_valu.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_pool -o t -- $B32 --ray-pool > $O/trace_f32_pool.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f64 -o t -- $B64 > $O/trace_f64.log 2>&1
for f in $O/trace_*/*kernel_stats.csv; do echo "== $f"; cat $f; done
pmc() { tag=$1; shift; cmd=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$tag -o p -- $cmd > $O/pmc_$tag.log 2>&1; }
P32="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
pmc f32_fetch "$P32" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f32_write "$P32" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f32_cull_fetch "$P32 --group-cull" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f32_cull_write "$P32 --group-cull" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f64_fetch "$B64" GRBM_GUI_ACTIVE FETCH_SIZE
pmc f64_write "$B64" GRBM_GUI_ACTIVE WRITE_SIZE
pmc f32_sqA "$P32" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_sqB "$P32" SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT
pmc f32_mfma "$P32" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
pmc f32_valu_sqA "$P32 --scan-valu" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_pool_sqA "$P32 --ray-pool" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f32_pool_sqB "$P32 --ray-pool" SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT
pmc f32_pool_mfma "$P32 --ray-pool" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pmc f64_sqA "$B64" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f64_mfma "$B64" SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$O/pmc_*")):
    if not os.path.isdir(d): continue
    c = collections.defaultdict(float); dur = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "trace_" in row["Kernel_Name"]: c[row["Counter_Name"]] += float(row["Counter_Value"])
    for f in glob.glob(d + "/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            if "trace_" in row["Kernel_Name"]: dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out[os.path.basename(d)] = {"counters": dict(c), "launches": len(dur), "kernel_ns": dur}
# derived figures of the headline kernel: per wave-segment = per 64 ray segments (segments from the bench line of the same workload)
try:
    segs = 8177903451.0        # 1920 x 1080 x 1000 spp: the segments the kernel counts (rtw_stats_t.segments; 3.9438 per sample)
    a, m = out["pmc_f32_sqA"]["counters"], out["pmc_f32_mfma"]["counters"]
    cyc = a["GRBM_GUI_ACTIVE"] / 8
    out["derived_f32"] = {"valu_per_wave_segment": a["SQ_INSTS_VALU"] / (segs / 64), "mfma_per_wave_segment": m["SQ_INSTS_MFMA"] / (segs / 64),
                          "clock_GHz": cyc / out["pmc_f32_sqA"]["kernel_ns"][0], "mfma_busy_frac_of_simd_cycles": m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc),
                          "valu_busy_frac_at_2_cycles": a["SQ_INSTS_VALU"] * 2 / (1024 * cyc), "wait_any_frac_of_wave_cycles": a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"]}
    wc = a["SQ_WAVE_CYCLES"]
    out["derived_f32"].update({"wait_inst_any_frac_of_wave_cycles": a["SQ_WAIT_INST_ANY"] / wc, "active_inst_any_frac_of_wave_cycles": a["SQ_ACTIVE_INST_ANY"] / wc,
                               "issue_busy": (m["SQ_VALU_MFMA_BUSY_CYCLES"] + 2 * a["SQ_INSTS_VALU"]) / (1024 * cyc)})
    pa, pm_ = out["pmc_f32_pool_sqA"]["counters"], out["pmc_f32_pool_mfma"]["counters"]
    pb = out["pmc_f32_pool_sqB"]["counters"]
    pcyc = pa["GRBM_GUI_ACTIVE"] / 8
    out["derived_f32_pool"] = {"valu_per_wave_segment": pa["SQ_INSTS_VALU"] / (segs / 64), "salu_per_wave_segment": pa["SQ_INSTS_SALU"] / (segs / 64),
                               "lds_per_wave_segment": pb["SQ_INSTS_LDS"] / (segs / 64), "mfma_per_wave_segment": pm_["SQ_INSTS_MFMA"] / (segs / 64),
                               "clock_GHz": pcyc / out["pmc_f32_pool_sqA"]["kernel_ns"][0], "mfma_busy_frac_of_simd_cycles": pm_["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * pcyc),
                               "valu_busy_frac_at_2_cycles": pa["SQ_INSTS_VALU"] * 2 / (1024 * pcyc), "issue_busy": (pm_["SQ_VALU_MFMA_BUSY_CYCLES"] + 2 * pa["SQ_INSTS_VALU"]) / (1024 * pcyc),
                               "wait_any_frac_of_wave_cycles": pa["SQ_WAIT_ANY"] / pa["SQ_WAVE_CYCLES"], "wait_inst_any_frac_of_wave_cycles": pa["SQ_WAIT_INST_ANY"] / pa["SQ_WAVE_CYCLES"],
                               "lds_bank_conflict_frac": pm_["SQ_LDS_BANK_CONFLICT"] / max(pm_["SQ_LDS_IDX_ACTIVE"], 1),
                               "salu_per_wave_segment_lane_loop": a["SQ_INSTS_SALU"] / (segs / 64), "lds_per_wave_segment_lane_loop": out["pmc_f32_sqB"]["counters"]["SQ_INSTS_LDS"] / (segs / 64)}
except Exception as e:
    out.setdefault("derived_f32", {})["error"] = str(e)
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
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
json.dump(tr, open("$O/hbm_traffic.json", "w"), indent=1)
print(json.dumps(tr, indent=1))
PY
# the bench lines read the HBM traffic of THIS run (box-local copy; the merged gpurun_out/<tag>/hbm_traffic.json is what gets
# committed as profiles/r04_hbm_traffic.json (likewise r04_pmc_summary.json: `roofline.issue_busy`) -- `traffic_static` in the line says that the figure is not measured by bench.py itself)
cp $O/hbm_traffic.json $R/profiles/r04_hbm_traffic.json; cp $O/pmc_summary.json $R/profiles/r04_pmc_summary.json
cd $R
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; cut -c1-400 $O/bench_f32.json
python bench.py --dtype f64 --width 3840 --steps 2 --warmup 1 --no-extras > $O/bench_f64_4k.json 2> $O/bench_f64_4k.err; cut -c1-300 $O/bench_f64_4k.json
python bench.py --emulate-shard-of 8 --steps 3 --no-cpu-baseline > $O/bench_f32_shard8.json 2>/dev/null; cut -c1-300 $O/bench_f32_shard8.json
RTW_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $O/bench_f32_2ranks_one_device.json 2>/dev/null; cut -c1-300 $O/bench_f32_2ranks_one_device.json
python tools/kernel_resources.py > $O/kernel_resources.txt 2>&1; tail -12 $O/kernel_resources.txt
# calibration of WRITE_SIZE / HW_REG_XCC_ID and the price of each phase (built on the box)
mkdir -p $R/build
/opt/rocm/bin/hipcc -O2 -std=c++17 --offload-arch=gfx950 tools/ubench_write_size.hip -o build/ubench_write_size 2>/dev/null
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/ws -o w -- $R/build/ubench_write_size > $O/ws.log 2>&1)
python3 - > $O/ubench_write_size.txt <<PY
import csv, glob
print(open("$O/ws.log").read().split("tool finalization")[-1].split("\n", 1)[-1] if "tool finalization" in open("$O/ws.log").read() else open("$O/ws.log").read())
for f in glob.glob("$O/ws/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "WRITE_SIZE" and ("k_" in r["Kernel_Name"]): print("%-24s WRITE_SIZE %10.1f KiB = %.2f x the 24.9 MB written" % (r["Kernel_Name"][:24], float(r["Counter_Value"]), float(r["Counter_Value"]) / 24300))
PY
cat $O/ubench_write_size.txt
bash tools/gpu_probe_phases.sh base mfma eval extract resolve reject noskip rejcap3 rejcap2 cmp 2>&1 | grep -v amdgpu.ids > $O/probe_phases.txt; cat $O/probe_phases.txt
# the ray-pool kernel's own stage profile (batches, fill, wave-cycles per stage) next to the lane loop's phase profile
(RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f32 1920 1000 50 pool 1; RTW_PHASE_PROFILE=1 python tools/gpu_quick.py f32 1920 1000 50 plain 1) 2>&1 | grep -E "profile\]|kernel" > $O/pool_stage_profile.txt; cat $O/pool_stage_profile.txt
