; A reconstruction of what Julia 1.6/1.7 emits for the arithmetic of hit(::Sphere{Float32}) (src/hit.jl:13-18):
;   oc = r.origin - s.center            fsub (StaticArrays, no flags)
;   half_b = oc . r.dir                 StaticArrays dot: fmul/fadd WITHOUT fast-math flags (a callee; @fastmath rewrites syntax only)
;   c = oc.oc - s.radius^2              sub_fast -> fsub fast;  ^2 -> pow_fast(x, Val(2)) -> llvm.powi.f32 (a ccall: no flags)
;   discriminant = half_b^2 - a*c       fsub fast; llvm.powi
define float @disc_julia(float %ox, float %oy, float %oz, float %cx, float %cy, float %cz, float %dx, float %dy, float %dz, float %r) {
  %ocx = fsub float %ox, %cx
  %ocy = fsub float %oy, %cy
  %ocz = fsub float %oz, %cz
  %m1 = fmul float %ocx, %dx
  %m2 = fmul float %ocy, %dy
  %a1 = fadd float %m1, %m2
  %m3 = fmul float %ocz, %dz
  %hb = fadd float %a1, %m3
  %q1 = fmul float %ocx, %ocx
  %q2 = fmul float %ocy, %ocy
  %b1 = fadd float %q1, %q2
  %q3 = fmul float %ocz, %ocz
  %ococ = fadd float %b1, %q3
  %r2 = call float @llvm.powi.f32.i32(float %r, i32 2)
  %c = fsub fast float %ococ, %r2
  %hb2 = call float @llvm.powi.f32.i32(float %hb, i32 2)
  %disc = fsub fast float %hb2, %c
  ret float %disc
}
; the same with the squares carrying fast-math flags (if pow_fast lowered to `fmul fast`)
define float @disc_fastsq(float %ox, float %oy, float %oz, float %cx, float %cy, float %cz, float %dx, float %dy, float %dz, float %r) {
  %ocx = fsub float %ox, %cx
  %ocy = fsub float %oy, %cy
  %ocz = fsub float %oz, %cz
  %m1 = fmul float %ocx, %dx
  %m2 = fmul float %ocy, %dy
  %a1 = fadd float %m1, %m2
  %m3 = fmul float %ocz, %dz
  %hb = fadd float %a1, %m3
  %q1 = fmul float %ocx, %ocx
  %q2 = fmul float %ocy, %ocy
  %b1 = fadd float %q1, %q2
  %q3 = fmul float %ocz, %ocz
  %ococ = fadd float %b1, %q3
  %r2 = fmul fast float %r, %r
  %c = fsub fast float %ococ, %r2
  %hb2 = fmul fast float %hb, %hb
  %disc = fsub fast float %hb2, %c
  ret float %disc
}
; src/hit.jl:20-23: sqrtd = sqrt_fast(discriminant) -> `call fast float @llvm.sqrt.f32`; root = (-half_b - sqrtd) / a with a = 1
define float @root_julia(float %hb, float %disc) {
  %s = call fast float @llvm.sqrt.f32(float %disc)
  %n = fneg fast float %hb
  %r = fsub fast float %n, %s
  %q = fdiv fast float %r, 1.0
  ret float %q
}
declare float @llvm.sqrt.f32(float)
declare float @llvm.powi.f32.i32(float, i32)
