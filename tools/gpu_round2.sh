#!/bin/bash
# Round-2 measurement set: what the driver runs at round end (smoke, pytest -m gpu, bench) plus the
# rocprofv3 summaries that go to profiles/.  Each rocprofv3 pass is its own run (kernel-trace/stats
# only, or --pmc only with --kernel-trace).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/round2; rm -rf $O; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
B32="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
B64="python $R/bench.py --dtype f64 --width 3840 --steps 1 --warmup 0 --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32 -o t -- $B32 > $O/trace_f32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_cull -o t -- $B32 --group-cull > $O/trace_f32_cull.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_f32_valu -o t -- $B32 --scan-valu > $O/trace_f32_valu.log 2>&1
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
pmc f64_sqA "$B64" GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU
pmc f64_sqB "$B64" SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_FLAT
python3 - <<PY
import csv, glob, collections, json, os
out = {}
for d in sorted(glob.glob("$O/pmc_*")):
    if not os.path.isdir(d): continue
    c = collections.defaultdict(float); dur = []
    for f in glob.glob(d + "/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"]: c[row["Counter_Name"]] += float(row["Counter_Value"])
    for f in glob.glob(d + "/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            if "trace_kernel" in row["Kernel_Name"]: dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out[os.path.basename(d)] = {"counters": dict(c), "launches": len(dur), "kernel_ns": dur}
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
for k, v in out.items(): print(k, v["launches"], {a: round(b, 3) for a, b in v["counters"].items()}, [round(x / 1e6, 2) for x in v["kernel_ns"]])
# HBM bytes per launch: FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section:
# on gfx950 it reports half of the bytes of wide reads; WRITE_SIZE is taken as is -- uncalibrated)
tr = {}
for key, tag in (("f32_1920x1080_1000spp_d50_plain", "f32"), ("f32_1920x1080_1000spp_d50_cull", "f32_cull"), ("f64_3840x2160_1000spp_d50_plain", "f64")):
    f, w = out.get("pmc_%s_fetch" % tag), out.get("pmc_%s_write" % tag)
    if f and w and f["launches"] and w["launches"]:
        fb = f["counters"].get("FETCH_SIZE", 0) / f["launches"] * 1024
        wb = w["counters"].get("WRITE_SIZE", 0) / w["launches"] * 1024
        tr[key] = {"hbm_bytes_per_launch": int(2 * fb + wb), "fetch_size_bytes_raw": int(fb), "write_size_bytes": int(wb),
                   "source": "rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE / --pmc GRBM_GUI_ACTIVE WRITE_SIZE, one launch each (tools/gpu_round2.sh)"}
json.dump(tr, open("$O/hbm_traffic.json", "w"), indent=1)
print(json.dumps(tr, indent=1))
PY
# the bench lines read the HBM traffic of THIS run (box-local copy of the profile; the merged gpurun_out/round2/hbm_traffic.json
# is what gets committed as profiles/r02_hbm_traffic.json)
cp $O/hbm_traffic.json $R/profiles/r02_hbm_traffic.json
cd $R
python bench.py > $O/bench_f32.json 2> $O/bench_f32.err; cut -c1-400 $O/bench_f32.json
python bench.py --dtype f64 --width 3840 --steps 2 --warmup 1 > $O/bench_f64_4k.json 2> $O/bench_f64_4k.err; cut -c1-400 $O/bench_f64_4k.json
python bench.py --emulate-shard-of 8 --steps 3 --no-cpu-baseline > $O/bench_f32_shard8.json 2>/dev/null; cut -c1-300 $O/bench_f32_shard8.json
