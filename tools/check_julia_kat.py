#!/usr/bin/env python3
"""Tier T2 in one command: diff the output of tools/julia_kat.jl (run wherever Julia + the reference
exist) against the oracle, item by item, and say which [UNVERIFIED] assumptions of DESIGN.md section 3
hold.  If every RNG item passes, `--accept` renames tests/golden/rng_provisional.npz -> rng_pinned.npz.

    julia --project=/path/to/RayTracingWeekend.jl -t1 tools/julia_kat.jl > julia_kat.txt
    python tools/check_julia_kat.py julia_kat.txt [--accept]

    python tools/check_julia_kat.py --self-test     # no Julia here: write the file the ORACLE predicts,
                                                    # then check it (exercises the parser; all items pass)
    python tools/check_julia_kat.py --self-test --numerics contract    # ... as a Julia whose hit(::Sphere) evaluates the discriminant in
                                                    # another of the oracle's numerics modes would print it: the checker must NAME that mode

Round 5: the `adv` records (4096 rays leaving computed hit points on the r = 1000 ground sphere) decide WHICH evaluation order of
src/hit.jl:16-18 the Julia build emits -- the oracle's `reference` (default of the library), `reference_fma`, `reference_fma2` or
`contract`; the summary names the mode(s) that reproduce every record and says what to change if it is not the default.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)
import rtw_amd as R          # noqa: E402
import rtw_oracle as O       # noqa: E402

TS = {"Float32": np.float32, "Float64": np.float64}
CAM_ORDER = O.CAM_FIELDS
HIT_RAYS = [((13, 2, 3), (-13, -2.2, -3.1)), ((0, 0, 0), (0.1, -0.05, -1)), ((0.3, 0.1, -0.6), (-0.2, 0.4, -1)), ((4, 1.5, 2), (-1, -0.4, -0.55))]
HIT_SPHERES = [((0, -1000, -1), 1000), ((0, 0, -1), 0.5), ((0, 0, -1), -0.4), ((0, 1, 0), 1)]


M64 = (1 << 64) - 1


def _sm_mix(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def seed_expansion_counter(seed):
    """(x, y) = the first two outputs of a SplitMix64 generator seeded with `seed`: the state is a Weyl counter stepped by the
    golden gamma, each output mixes the counter (Vigna's splitmix64.c; what oracle/rtw_oracle.c rng_seed_int restates)."""
    s1 = (seed + 0x9E3779B97F4A7C15) & M64
    s2 = (s1 + 0x9E3779B97F4A7C15) & M64
    return _sm_mix(s1), _sm_mix(s2)


def seed_expansion_output_fed(seed):
    """(x, y) when a stateless one-step function `splitmix64(x) -> output` is chained: each OUTPUT is the next call's input
    (x = f(seed), y = f(x)) -- the other plausible reading of RandomNumbers.jl's `init_seed(seed, UInt64, 2)`."""
    f = lambda v: _sm_mix((v + 0x9E3779B97F4A7C15) & M64)
    x = f(seed & M64)
    return x, f(x)


def after_one_step(x, y):
    """the xoroshiro128+ (55, 14, 36) state after one discarded output"""
    rotl = lambda v, k: ((v << k) | (v >> (64 - k))) & M64
    s1 = x ^ y
    return rotl(x, 55) ^ s1 ^ ((s1 << 14) & M64), rotl(s1, 36)


def classify_seed_expansion(seed, state):
    """Which known expansion produces `state` = (x, y) right after Xoroshiro128Plus(seed)?  None if none does."""
    state = (int(state[0]), int(state[1]))
    for name, fn in (("counter-stepped SplitMix64 (oracle/rtw_oracle.c as restated)", seed_expansion_counter),
                     ("output-fed SplitMix64 (x = f(seed), y = f(x)): change rng_seed_int in oracle/rtw_oracle.c and "
                      "raytracingweekend.jl_amd/rng.py accordingly", seed_expansion_output_fed)):
        xy = fn(seed)
        if state == after_one_step(*xy):
            return name
        if state == xy:
            return name + ", WITHOUT the discarded first output"
    return None


def fmt(x, T):
    return ("%.9g" if T is np.float32 else "%.17g") % float(x)


def normalize(v):
    T = v.dtype.type
    inv = T(1) / np.sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2])
    return np.array([inv * v[0], inv * v[1], inv * v[2]], T)


def cameras(T):
    return (("t_default_cam", R.t_default_cam(elem_type=T)), ("t_cam1", R.t_cam1(elem_type=T)), ("t_cam2", R.t_cam2(elem_type=T)))


NUMERICS_MODES = ("reference", "reference_fma", "reference_fma2", "contract")


def adv_inputs(T, n=4096):
    """The self-test's own adversarial rays (julia_kat.jl 4c prints ITS inputs -- sphere, origin, direction --; the checker never has to
    reproduce a recipe): first half rays leaving computed hit points on the r = 1000 ground sphere, then near-grazing rays at spheres of
    radius 0.05 ... 0.45 and rays leaving computed hit points on such spheres.  -> [(c, r, o, d) | None]"""
    rng = np.random.default_rng(20260929)
    gc, gr = np.array([0, -1000, -1], T), T(1000)
    out = []
    with O.numerics("reference"):
        for k in range(n // 2):
            tgt = np.array([rng.uniform(-11, 11), 0, rng.uniform(-11, 11)], T)
            d = normalize((tgt - np.array([13, 2, 3], T)).astype(T))
            h = O.hit_sphere(gc, gr, np.array([13, 2, 3], T), d, T(1e-4), np.inf, T)
            if h is None:
                out.append(None); continue
            u = rng.uniform(-1, 1, 3).astype(T)
            nrm = np.asarray(h["n"], T)
            d2 = normalize(np.array([nrm[0] + T(0.98) * u[0], nrm[1] + T(0.98) * u[1], nrm[2] + T(0.98) * u[2]], T))
            out.append((gc, gr, np.asarray(h["p"], T), d2))
    for k in range(n // 2, n):
        c = np.array([rng.uniform(-11, 11), 0.2, rng.uniform(-11, 11)], T)
        o = np.array([rng.uniform(-12, 12), rng.uniform(0.05, 3.05), rng.uniform(-12, 12)], T)
        w = rng.uniform(-1, 1, 3).astype(T)
        rr = T(rng.uniform(0.05, 0.45))
        d1 = normalize(((c - o) + T(0.95) * rr * w).astype(T))
        if k < 3 * n // 4:
            out.append((c, rr, o, d1))
            continue
        with O.numerics("reference"):                               # last quarter: rays LEAVING the computed hit point on that small sphere
            h = O.hit_sphere(c, rr, o, d1, T(1e-4), np.inf, T)
        if h is None:
            out.append(None); continue
        u = rng.uniform(-1, 1, 3).astype(T)
        nrm = np.asarray(h["n"], T)
        out.append((c, rr, np.asarray(h["p"], T), normalize(np.array([nrm[0] + T(0.98) * u[0], nrm[1] + T(0.98) * u[1], nrm[2] + T(0.98) * u[2]], T))))
    return out


def adv_answer(c, r, o, d, T, mode):
    with O.numerics(mode):
        h = O.hit_sphere(c, r, o, d, T(1e-4), np.inf, T)
    return "miss" if h is None else fmt(h["t"], T)


def predicted_lines(outdir, numerics="reference"):
    """What julia_kat.jl prints if every assumption of the oracle is right (and hit(::Sphere) evaluates in `numerics`)."""
    prev = O.set_numerics(numerics)
    try:
        return _predicted_lines(outdir, numerics)
    finally:
        O.set_numerics(prev)


def _predicted_lines(outdir, numerics):
    out = ["julia_version: (oracle self-test) nthreads: 1"]
    for seed in (1, 2):
        st = O.rng_seed(seed)
        out.append("rng_state seed=%d: %016x %016x" % (seed, int(st[0]), int(st[1])))
        out.append("rng_u64 seed=%d: " % seed + " ".join("%016x" % O.rng_next(st) for _ in range(16)))
        st = O.rng_seed(seed); out.append("rng_f32 seed=%d: " % seed + " ".join(fmt(O.rng_float(st, np.float32), np.float32) for _ in range(16)))
        st = O.rng_seed(seed); out.append("rng_f64 seed=%d: " % seed + " ".join(fmt(O.rng_float(st, np.float64), np.float64) for _ in range(16)))
    for name, T in TS.items():
        flat = O.scene_random_spheres(1, T)
        out.append(f"scene {name} n: {flat['n']}")
        for i in range(flat["n"]):
            out.append(f"sphere {name} {i}: " + " ".join(fmt(flat[k][i], T) for k in ("cx", "cy", "cz", "r")) + f" {int(flat['kind'][i])} " +
                       " ".join(fmt(flat[k][i], T) for k in ("ar", "ag", "ab", "param")))
        for cname, cam in cameras(T):
            v = [x for k in CAM_ORDER for x in np.asarray(getattr(cam, k), T)] + [cam.lens_radius]
            out.append(f"camera {cname} {name}: " + " ".join(fmt(x, T) for x in v))
        v = np.array([0.3, -0.7, 0.2], T)
        out.append(f"normalize {name}: " + " ".join(fmt(x, T) for x in normalize(v)))
        w = np.array([0.1, 0.2, 0.3], T)
        out.append(f"dot {name}: " + fmt((v[0] * w[0] + v[1] * w[1]) + v[2] * w[2], T))
        from rtw_amd.structs import _tand
        out.append(f"tand {name}: " + " ".join(fmt(x, T) for x in (_tand(T(10), T), _tand(T(45), T), _tand(T(20) / T(2), T))))
        for ri, (o, d) in enumerate(HIT_RAYS, 1):
            o = np.array(o, T); d = normalize(np.array(d, T))
            for si, (c, r) in enumerate(HIT_SPHERES, 1):
                c = np.array(c, T)
                h = O.hit_sphere(c, T(r), o, d, T(1e-4), np.inf, T)
                head = " ".join(fmt(x, T) for x in (*o, *d, *c, T(r)))
                tail = "miss" if h is None else " ".join(fmt(x, T) for x in (h["t"], *h["p"], *h["n"])) + " " + ("1" if h["front"] else "0")
                out.append(f"hit {name} {ri} {si}: {head} -> {tail}")
        # the tmin self-intersection regime on the ground sphere (julia_kat.jl, 4b)
        gc, gr = np.array([0, -1000, -1], T), T(1000)
        us = [np.array(u, T) for u in ((0.6, 0.1, -0.7), (-0.3, -0.9, 0.2), (0, -0.999, 0.02), (0.5, 0.5, 0.5))]
        for k in range(16):
            d = normalize(np.array([T(-13) + T(0.37) * T(k), T(-2.2) - T(0.03) * T(k), T(-3.1) + T(0.21) * T(k)], T))
            h = O.hit_sphere(gc, gr, np.array([13, 2, 3], T), d, T(1e-4), np.inf, T)
            if h is None:
                out.append(f"selfhit {name} {k}: primary miss"); continue
            outs = []
            for u in us:
                n = np.asarray(h["n"], T)
                h2 = O.hit_sphere(gc, gr, np.asarray(h["p"], T), normalize(np.array([n[0] + u[0], n[1] + u[1], n[2] + u[2]], T)), T(1e-4), np.inf, T)
                outs.append("miss" if h2 is None else fmt(h2["t"], T))
            out.append(f"selfhit {name} {k}: " + " ".join(fmt(x, T) for x in (h["t"], *h["p"])) + " -> " + " ".join(outs))
        for k, inp in enumerate(adv_inputs(T)):
            if inp is None:
                out.append(f"adv {name} {k}: primary miss"); continue
            c, r, o, d2 = inp
            out.append(f"adv {name} {k}: " + " ".join(fmt(x, T) for x in (*c, r, *o, *d2)) + " -> " + adv_answer(c, r, o, d2, T, numerics))
        img, _ = O.render(R.flatten_scene(R.scene_2_spheres(elem_type=T), T), R.t_default_cam(elem_type=T), 96, 54, 16, T=T, max_depth=16,
                          rng_mode=O.REF_SERIAL, ref_threads=1, product_order=O.PRODUCT_REFERENCE, numerics=numerics)
        np.ascontiguousarray(img.transpose(1, 0, 2)).astype(T).tofile(os.path.join(outdir, f"julia_render_2spheres_96x54_16spp_{name}.bin"))
        out.append(f"render {name} mean: " + fmt(img.astype(T).mean(dtype=np.float64), T))
    return out


def classify_numerics(got):
    """Which numerics mode(s) of the oracle reproduce EVERY `adv` record of a dump?  -> {T name: {mode: (matching, total)}}"""
    res = {}
    for name, T in TS.items():
        recs = [(k, v) for k, v in got.items() if k.startswith(f"adv {name} ") and "->" in v]
        if not recs:
            continue
        counts = {m: 0 for m in NUMERICS_MODES}
        for _, v in recs:
            lhs, rhs = v.split("->")
            x = [T(t) for t in lhs.split()]
            c, r, p, d = np.array(x[:3], T), x[3], np.array(x[4:7], T), np.array(x[7:10], T)
            rhs = rhs.strip()
            for m in NUMERICS_MODES:
                a = adv_answer(c, r, p, d, T, m)
                if a == rhs or (a != "miss" and rhs != "miss" and T(a) == T(rhs)):
                    counts[m] += 1
        res[name] = {m: (c, len(recs)) for m, c in counts.items()}
    return res


def check(path):
    base = os.path.dirname(os.path.abspath(path))
    got = {}
    for line in open(path):
        if ":" in line:
            k, v = line.split(":", 1)
            got[k.strip()] = v.strip()
    # round 5: first find out in which numerics mode this Julia evaluates hit(::Sphere); everything downstream (hit / selfhit records,
    # the -t1 render) is then predicted in that mode
    cls = classify_numerics(got)
    mode = "reference"
    verdicts = []
    for name, counts in cls.items():
        full = [m for m, (c, n) in counts.items() if c == n]
        print(f"numerics of hit(::Sphere{{{name}}}) -- adversarial rays reproduced: " + ", ".join(f"{m} {c}/{n}" for m, (c, n) in counts.items()))
        verdicts.append((name, full))
    f32 = dict(verdicts).get("Float32")
    if f32 is not None:
        if len(f32) == 1:
            mode = f32[0]
            print(f"  -> this Julia evaluates src/hit.jl:16-18 as the oracle's `{mode}` mode" + (" (the library's default)" if mode == "reference" else
                  ": ORACLE-ONLY since ABI 4 (the library dropped it in round 6 because no compiler was known to emit it): restore RTW_FLAG_NUMERICS_REFERENCE_FMA from git history" if mode == "reference_fma" else
                  f": make it the default (include/rtw_hip.h RTW_FLAG_NUMERICS_*; `numerics` of render()), or pass numerics={mode!r}"
                 ))
        elif not f32:
            print("  -> NO numerics mode of the oracle reproduces every record: inspect julia_hit_sphere_Float32.ll (fmuladd? contract flags? a reassociated dot?)")
        else:
            mode = f32[0]
            print(f"  -> not decided by these rays ({', '.join(f32)} all reproduce them)")
    for name in TS:
        ll = os.path.join(base, f"julia_hit_sphere_{name}.ll")
        if os.path.exists(ll):
            txt = open(ll).read()
            print(f"  {os.path.basename(ll)}: fmul {txt.count('fmul')}, fadd {txt.count('fadd')}, fsub {txt.count('fsub')}, llvm.fmuladd {txt.count('llvm.fmuladd')}, "
                  f"llvm.fma {txt.count('llvm.fma.')}, `contract`/`fast` flags {txt.count(' contract ') + txt.count(' fast ')}, llvm.powi {txt.count('llvm.powi')}")
    want = {}
    for line in predicted_lines("/tmp", mode):
        k, v = line.split(":", 1)
        want[k.strip()] = v.strip()
    items = [(f"numerics mode of hit(::Sphere) identified ({mode})", f32 is None or len(f32) == 1, "src/hit.jl:16-18", [], [])]

    def item(name, keys, why, numeric_T=None):
        ks = [k for k in want if any(k.startswith(p) for p in keys)]
        missing = [k for k in ks if k not in got]
        bad = []
        for k in ks:
            if k in got and got[k] != want[k]:
                if numeric_T is not None:        # compare as numbers of type T (print formats may differ in trailing digits)
                    try:
                        a = [numeric_T(x) if x not in ("miss", "->") else x for x in got[k].split()]
                        b = [numeric_T(x) if x not in ("miss", "->") else x for x in want[k].split()]
                        if a == b:
                            continue
                    except ValueError:
                        pass
                bad.append(k)
        ok = not missing and not bad
        items.append((name, ok, why, missing, bad))
    item("RNG seed expansion (SplitMix64 x2 + one discarded output)", ["rng_state"], "src/init.jl:9, src/rand.jl:2")
    item("RNG stream (xoroshiro128+ 55/14/36)", ["rng_u64"], "RandomNumbers.jl 1.5.3")
    item("rand(rng, Float32): low 23 bits", ["rng_f32"], "src/rand.jl:12", np.float32)
    item("rand(rng, Float64): low 52 bits", ["rng_f64"], "src/rand.jl:12", np.float64)
    for name, T in TS.items():
        item(f"scene_random_spheres {name} (draw order, literals)", [f"scene {name}", f"sphere {name}"], "src/scenes.jl:49-84", T)
        item(f"default_camera presets {name}", [f"camera t_default_cam {name}", f"camera t_cam1 {name}", f"camera t_cam2 {name}"], "src/camera.jl:18-41", T)
        item(f"StaticArrays normalize/dot {name}", [f"normalize {name}", f"dot {name}"], "inv(norm(v)) * v; (x1y1 + x2y2) + x3y3", T)
        item(f"tand {name}", [f"tand {name}"], "src/camera.jl:23", T)
        item(f"hit(::Sphere) {name} (@fastmath contraction of the discriminant)", [f"hit {name}"], "src/hit.jl:12-35", T)
        item(f"tmin self-intersection on the r = 1000 ground sphere {name}", [f"selfhit {name}"], "src/ray_color.jl:19, src/hit.jl:19-29", T)
        item(f"adversarial ground-sphere rays {name} (in the identified mode)", [f"adv {name}"], "src/hit.jl:16-18", T)
    # a failing seed expansion: say WHICH expansion Julia uses (two are plausible; the package source is not in the reference tree)
    for seed in (1, 2):
        k = f"rng_state seed={seed}"
        if k in got and got[k] != want.get(k):
            try:
                st = tuple(int(v, 16) for v in got[k].split())
                print(f"seed expansion, seed {seed}: Julia's state matches -> {classify_seed_expansion(seed, st) or 'NEITHER known expansion: inspect RandomNumbers.jl src/common.jl init_seed'}")
            except ValueError:
                pass
    print(f"{'item':72s} result")
    for name, ok, why, missing, bad in items:
        print(f"{name:72s} {'PASS' if ok else 'FAIL'}   [{why}]" + ("" if ok else f"  missing {missing[:3]} differing {bad[:3]}"))
    # the -t1 render against REF_SERIAL
    for name, T in TS.items():
        f = os.path.join(base, f"julia_render_2spheres_96x54_16spp_{name}.bin")
        if not os.path.exists(f):
            print(f"render {name}: {f} not found -- run julia_kat.jl in the directory of the text file")
            items.append((f"render {name}", False, "", [f], []))
            continue
        jl = np.fromfile(f, dtype=T).reshape(96, 54, 3).transpose(1, 0, 2)
        ref, _ = O.render(R.flatten_scene(R.scene_2_spheres(elem_type=T), T), R.t_default_cam(elem_type=T), 96, 54, 16, T=T, max_depth=16,
                          rng_mode=O.REF_SERIAL, ref_threads=1, product_order=O.PRODUCT_REFERENCE, numerics=mode)
        same = np.array_equal(jl, ref)
        d = np.abs(jl.astype(np.float64) - ref.astype(np.float64))
        print(f"{'render(scene_2_spheres, default cam, 96, 16), julia -t1 vs oracle REF_SERIAL ' + name:72s} "
              f"{'PASS (bit-identical)' if same else 'FAIL'}   max abs diff {d.max():.3g}, mean {d.mean():.3g}, channels differing {int((d > 0).sum())} of {d.size}")
        items.append((f"render {name}", same, "", [], []))
    return items


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    if sys.argv[1] == "--self-test":
        rest = [a for a in sys.argv[2:] if not a.startswith("--")]
        numerics = sys.argv[sys.argv.index("--numerics") + 1] if "--numerics" in sys.argv else "reference"
        rest = [a for a in rest if a != numerics]
        d = rest[0] if rest else "/tmp/julia_kat_selftest"
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "julia_kat.txt")
        open(path, "w").write("\n".join(predicted_lines(d, numerics)) + "\n")
        items = check(path)
    else:
        items = check(sys.argv[1])
    ok = all(i[1] for i in items)
    rng_ok = all(i[1] for i in items if i[0].startswith(("RNG", "rand(")))
    print("ALL PASS: the oracle is pinned against the real reference (tier T2)" if ok else "some items FAIL: see above; fix oracle/ and re-run")
    if "--accept" in sys.argv and rng_ok:
        src, dst = (os.path.join(ROOT, "tests", "golden", n) for n in ("rng_provisional.npz", "rng_pinned.npz"))
        if os.path.exists(src):
            os.rename(src, dst)
            print(f"renamed {src} -> {dst}; update tests/test_oracle_golden.py to load rng_pinned.npz")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
