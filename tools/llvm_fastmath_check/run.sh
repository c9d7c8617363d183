#!/bin/bash
# Which of the numerics modes (include/rtw_hip.h RTW_FLAG_NUMERICS_*) does an LLVM x86 back end with FMA produce for the arithmetic of
# hit(::Sphere{Float32}) (/root/reference/src/hit.jl:13-18)?  No Julia here, but LLVM is (ROCm's clang, x86-64 target): hit_sphere_julia_ir.ll
# holds a hand-written reconstruction of the IR Julia's front end emits -- StaticArrays' dot as fmul / fadd WITHOUT fast-math flags (a callee
# that @fastmath does not rewrite), the syntactic `-` as `fsub fast`, `x^2` as pow_fast(x, Val(2)) = a ccall of llvm.powi.f32 (no flags) --
# and a second function where the squares are `fmul fast`.  Compiled for znver2 (the reference's Ryzen 3700X):
#   disc_julia   -> vmulss / vaddss / vsubss only: NO fma anywhere (the two `fsub fast` are merely re-associated: hb^2 + (r^2 - oc.oc), same bits)
#                   = numerics mode `reference`
#   disc_fastsq  -> vfnmadd231ss (c = oc.oc - r*r fused) + vfmsub231ss (disc = hb*hb - c fused) = numerics mode `reference_fma2`
#   root_julia   -> vsqrtss (IEEE), no vrsqrtss estimate, no division
# i.e. `reference_fma` (only the last step fused) is not something this LLVM produces from either IR; `contract` would need @fastmath to
# reach inside dot.  The real answer is tools/julia_kat.jl on a Julia box; this narrows what to expect.
cd "$(dirname "$0")"
CLANG=${CLANG:-/opt/rocm/lib/llvm/bin/clang}
$CLANG -O2 -S -march=znver2 hit_sphere_julia_ir.ll -o /tmp/hit_sphere_julia_ir.s 2>/dev/null || exit 1
awk '/^disc_/ {f=$1} /vfm|vfnm/ {n[f]++} /^disc_/ {n[$1]+=0} END {for (k in n) print k, "fused multiply-adds:", n[k]}' /tmp/hit_sphere_julia_ir.s | sort
grep -E "^disc_|vfm|vfnm" /tmp/hit_sphere_julia_ir.s
# the fast square root of line 20 stays the IEEE instruction (no reciprocal-square-root estimate for a scalar sqrt on x86), the division by a = 1 folds away
awk '/^root_julia:/ {f=1} f && /vsqrtss|vrsqrt|vdivss/ {print "root_julia:", $1} f && /retq/ {f=0}' /tmp/hit_sphere_julia_ir.s
