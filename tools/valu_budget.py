#!/usr/bin/env python3
"""VALU budget of the headline trace kernel: static VALU instructions per phase from the ISA x how often the phase runs
per wave-iteration (the event counters of the phase-profile build), against the measured SQ_INSTS_VALU per wave-iteration.

  python tools/valu_budget.py [counts.txt] [measured VALU per wave-iteration]

* ISA: rtw_launch.hip compiled with the product's flags + -gline-tables-only (line tables do not change the code), -save-temps.  Every
  instruction of trace_kernel<float, lds-scene, matrix pipe, numerics fixed> carries the source line of its last `.loc`; a BASIC BLOCK belongs to
  the phase of the first line it shows in rtw_kernels.hpp / rtw_scan_mfma.hpp (the phases are line ranges there, found by their anchor
  comments); blocks that show only inlined helper code (rtw_path.hpp: generator, reject_trial, normalize ...) inherit the phase of the block
  before them in layout order.  Exec-masked code counts in full (a VALU instruction issues whatever its mask); blocks behind an
  s_cbranch_execz are counted as executed -- the budget is an upper estimate where a whole wave skips a branch.
* counts.txt: the `[rtw phase counts]` line of a run with RTW_ENABLE_TEST_AIDS=1 RTW_PHASE_PROFILE=1 (tools/gpu_valu_budget.sh writes it)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raytracingweekend.jl_amd", "csrc")
OUT = os.path.join(ROOT, "build", "asm_g")
KERNEL = "_ZN3rtw12trace_kernelIfLb0ELb1ELb0ELb1ELi0EEE"
KERNEL_CULL = "_ZN3rtw12trace_kernelIfLb0ELb1ELb1ELb1ELi0EEE"       # `--cull`: the group-cull instance

# phase = (file, anchor substring of the line where it starts); a phase runs to the next anchor of the same file.  `mult`: counter name(s).
PHASES = [
    ("rtw_kernels.hpp", "template <typename T, bool PROFILE, bool LDS_SCENE, bool CULL, bool MFMA", "kernel prologue (once per wave)", "0"),
    ("rtw_kernels.hpp", "// ---- (S) closest hit over the whole sphere list", "S   scan: call site", "1"),
    ("rtw_kernels.hpp", "// ---- (H1) a miss ends the sample", "H1  miss: sky, (lane, channel) tasks", "H1_executed"),
    ("rtw_kernels.hpp", "for (unsigned t0 = 0; t0 < n3; t0 += 64u)", "H1  miss: exact accumulation, per round of 64 tasks", "H1_rounds"),
    ("rtw_kernels.hpp", "has_ray = false;", "H1  (end)", "1"),
    ("rtw_kernels.hpp", "// ---- (A) lanes whose chunk is done retire it", "A   retire + item hand-out", "A_executed"),
    ("rtw_kernels.hpp", "store_job<T>(P, S, lane, out);", "A   store of a finished job", "jobs_stored"),
    ("rtw_kernels.hpp", "// Items for the lanes that need one", "A   item hand-out (rounds)", "A_executed"),
    ("rtw_kernels.hpp", "if (pool_next >= pool_end) {", "A   ticket / slot / open job / stream set-up of a batch", "batches_set_up"),
    ("rtw_kernels.hpp", "// hand out items of the wave's batch", "A   item hand-out", "A_executed"),
    ("rtw_kernels.hpp", "// ---- (H2) a hit starts the scatter", "H2  hit: material fetch, hit record, scatter_begin", "H2_executed"),
    ("rtw_kernels.hpp", "// ---- (B) start the next sample", "B   new sample: jitter, pixel coordinates", "B_executed"),
    ("rtw_kernels.hpp", "// ---- (R) ONE rejection loop", "R   rejection loop: set-up", "R_executed"),
    ("rtw_kernels.hpp", "            while (pending) {", "R   rejection loop: one trial", "reject_trials"),
    ("rtw_kernels.hpp", "// ---- (F) finish the scatter / the camera ray", "F   scatter_finish, camera ray, normalize", "1"),
    ("rtw_kernels.hpp", "    if (PROFILE && lane == 0) {", "kernel epilogue (once per wave)", "0"),
    ("rtw_scan_mfma.hpp", "__device__ __forceinline__ void test_singles(", "S   pass 2: exact test, per round of 64 candidates", "exact_test_rounds", "v_sqrt_f32"),
    ("rtw_scan_mfma.hpp", "__device__ __forceinline__ void resolve_pairs_impl(", "S   pass 2: explode entries into candidates", "explode_iterations", "v_mbcnt_hi"),
    ("rtw_scan_mfma.hpp", "// ---- ray features (binary32) ----", "S   prologue: ray features, f16 splits, operand words", "1"),
    ("rtw_scan_mfma.hpp", "[[maybe_unused]] auto block_sets = [&]", "S   group cull: clip of the ray + table vote (once per group of 32 blocks)", "1"),
    ("rtw_scan_mfma.hpp", "// (measured: running the first group's vote and the first operand fetch HERE", "S   prologue: ray features, f16 splits, operand words", "1"),
    ("rtw_scan_mfma.hpp", "// The rest of the in-lane class (group cull", "S   group cull: discriminant of the in-lane class, list entries", "1"),
    ("rtw_scan_mfma.hpp", "// ---- the result cells, initialised with", "S   prologue: in-lane test of the huge spheres", "1"),
    ("rtw_scan_mfma.hpp", "// Wave priority: low inside the block loop", "S   block loop: control, MFMA issue, AND pre-check", "blocks"),
    ("rtw_scan_mfma.hpp", "for (int r = r0; r < r0 + GS; ++r) mask = __builtin_amdgcn_alignbit", "S   block loop: sign collection (16 v_alignbit)", "sign_collections", "v_alignbit_b32/16"),
    ("rtw_scan_mfma.hpp", "any_cand = true;", "S   block loop: control, MFMA issue, AND pre-check", "blocks"),
    ("rtw_scan_mfma.hpp", "// the lanes with a candidate in this block record", "S   block loop: record the block's entries", "blocks_recording"),
    ("rtw_scan_mfma.hpp", "if (use_prio) __builtin_amdgcn_s_setprio(sizeof(T) == 4 ? 1 : 0);", "S   epilogue: final pass 2 call, result cells", "1"),
]


def build_isa():
    os.makedirs(OUT, exist_ok=True)
    m = re.search(r"^FLAGS\s*\?=\s*(.*)$", open(os.path.join(CSRC, "Makefile")).read(), re.M)
    fl = m.group(1).replace("$(EXTRA)", "").replace("$(ARCH)", "gfx950").split()
    cmd = ["/opt/rocm/bin/hipcc"] + fl + ["-gline-tables-only", "-save-temps", "-c", os.path.join(CSRC, "rtw_launch.hip"), "-o", "rtw_launch.o"]
    subprocess.run(cmd, cwd=OUT, check=True, capture_output=True)
    return os.path.join(OUT, "rtw_launch-hip-amdgcn-amd-amdhsa-gfx950.s")


def phase_table():
    tab = {}
    for fname in sorted({p[0] for p in PHASES}):
        lines = open(os.path.join(CSRC, fname)).read().split("\n")
        starts = []
        for f, anchor, name, mult in [p[:4] for p in PHASES]:
            if f != fname:
                continue
            hits = [i + 1 for i, ln in enumerate(lines) if anchor in ln]
            if not hits:
                raise SystemExit(f"anchor not found in {fname}: {anchor!r}")
            starts.append((hits[0] if "any_cand = true;" not in anchor else hits[0] + 1, name, mult))
        tab[fname] = sorted(starts)
    return tab


def phase_of(tab, fname, line):
    st = tab.get(fname)
    if not st:
        return None
    cur = None
    for l0, name, mult in st:
        if line >= l0:
            cur = (name, mult)
    return cur


def main():
    counts = {"1": 1.0, "0": 0.0}
    measured = None
    global KERNEL
    if "--cull" in sys.argv:
        sys.argv.remove("--cull")
        KERNEL = KERNEL_CULL
    for a in sys.argv[1:]:
        if os.path.exists(a):
            txt = open(a).read()
            for k, v in re.findall(r"(\w+)=([0-9.eE+-]+)", txt):
                counts[k] = float(v)
        else:
            measured = float(a)
    s = open(build_isa()).read()
    files = {int(n): os.path.basename(nm) for n, nm in re.findall(r'^\s*\.file\s+(\d+)\s+(?:"[^"]*"\s+)?"([^"]+)"', s, re.M)}
    start = s.index("\n" + KERNEL)
    body = s[start:s.index(".Lfunc_end", start)]
    tab = phase_table()
    # helpers defined in front of the phases' line ranges (udiv_magic, fx_accumulate_channel, store_job, claim_job, open_job; split_f16, lane_get)
    # are looked through: an instruction belongs to the innermost frame of its inlined-at chain that lies INSIDE a phase range
    first_line = {f: st[0][0] for f, st in tab.items()}
    agg, cur_phase, n_blocks = {}, ("kernel prologue (once per wave)", "0"), 0
    slow_ops = ("v_alignbit", "v_cmp", "v_readlane", "v_writelane", "v_readfirstlane", "v_lshlrev", "v_mul_lo", "v_mul_hi", "v_div", "v_max", "v_min", "v_cvt", "v_ffb", "v_bfe",
                "v_mbcnt", "v_lshl_add", "v_add3", "v_sqrt", "v_rcp", "v_rsq", "v_pk_")
    for ln in body.split("\n"):
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            n_blocks += 1
            continue
        if t.startswith(".loc"):
            frames = re.findall(r"([\w./+-]+):(\d+):\d+", t.split(";", 1)[1]) if ";" in t else []
            for fpath, line in frames:                       # innermost -> outermost
                f = os.path.basename(fpath)
                if f in tab and int(line) >= first_line[f]:
                    ph = phase_of(tab, f, int(line))
                    if ph:
                        cur_phase = ph
                        break
            continue
        if not t or t[0] in ";." or not t.startswith("v_") or t.startswith("v_mfma"):
            continue
        a = agg.setdefault(cur_phase[0], {"mult": cur_phase[1], "valu": 0, "slow": 0, "ops": {}})
        a["valu"] += 1
        a["ops"][t.split()[0]] = a["ops"].get(t.split()[0], 0) + 1
        if t.startswith(slow_ops) or "_f64" in t:
            a["slow"] += 1
    blocks = [None] * n_blocks
    total_static = sum(a["valu"] for a in agg.values())
    print(f"VALU budget of {KERNEL}... (static: {total_static} VALU instructions in {len(blocks)} basic blocks)")
    print(f"{'phase':66s} {'static':>6s} {'slow':>5s} {'x per wave-iteration':>28s} {'= VALU':>8s}")
    tot = 0.0
    order = [p[2] for p in PHASES]
    sig = {p[2]: p[4] for p in PHASES if len(p) > 4}        # phases inlined several times: copies = occurrences of a signature instruction
    for name in sorted(agg, key=lambda n: order.index(n) if n in order else 99):
        key = name
        a = agg[name]
        mult = counts.get(a["mult"])
        copies = 1
        if name in sig:
            op, _, per = sig[name].partition("/")
            copies = max(1, round(sum(v for k, v in a["ops"].items() if k.startswith(op)) / float(per or 1)))
        dyn = a["valu"] / copies * mult if mult is not None else float("nan")
        name = name + (f" [{copies} inlined copies]" if copies > 1 else "")
        if mult is not None:
            tot += dyn
        print(f"{name:66s} {a['valu'] // copies:6d} {a['slow']:5d} {a['mult']:>18s} = {mult if mult is not None else float('nan'):7.3f} {dyn:8.1f}")
    print(f"{'sum':66s} {'':6s} {'':5s} {'':28s} {tot:8.1f}" + (f"   measured SQ_INSTS_VALU per wave-iteration: {measured:.0f} ({100 * tot / measured:.0f} % accounted)" if measured else ""))


if __name__ == "__main__":
    main()
