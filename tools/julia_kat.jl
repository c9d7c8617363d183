# julia_kat.jl -- run on ANY box that has Julia + the reference package to pin the oracle's
# unverified third-party restatements (DESIGN.md section 3, tier T2).  This image has no julia.
#
#   julia --project=/path/to/RayTracingWeekend.jl -t1 tools/julia_kat.jl > julia_kat.txt
#   python tools/check_julia_kat.py julia_kat.txt          # prints PASS/FAIL per [UNVERIFIED] item
#   julia --project=/path/to/RayTracingWeekend.jl -t 16 tools/julia_bench.jl 16 > julia_bench.json     # the reference's CPU path timed (BASELINE.md B1)
#                                                          # + the first execution of julia/RTWeekendHIP.jl where the library and a GPU exist
#
# Output = one record per line, `key: values...`; floats are printed with %.9g (Float32) / %.17g
# (Float64), which round-trips exactly.  The two images go to julia_render_2spheres_96x54_16spp_<T>.bin
# next to the text file (raw column-major RGB{T}, what check_julia_kat.py reads back).
#   1. Xoroshiro128Plus(seed): state after construction, first 16 UInt64 / Float32 / Float64
#   2. scene_random_spheres(elem_type=T) after reseed!()  (SoA dump)
#   3. default_camera presets (22 scalars)
#   4. StaticArrays normalize / dot, Base tand, and hit(::Sphere) on fixed rays (@fastmath contraction)
#   4b. the tmin self-intersection regime on the r = 1000 ground sphere (src/ray_color.jl:19: tmin = T(1e-4) is the size of a binary32
#       ulp there -- it is what makes Float32 take 3.94 segments per sample against 2.71 in Float64): rays that LEAVE a computed hit
#       point on the ground sphere, steep to grazing, and whether hit(ground, ...) finds the far root again
#   4c. WHICH evaluation of src/hit.jl:16-18 this Julia build emits (round 5): 4096 rays that leave computed hit points on the r = 1000
#       ground sphere (inputs printed, so the checker evaluates exactly these) -- the oracle's numerics modes `reference` (StaticArrays'
#       un-fused dot, one rounding per operation), `reference_fma` (disc = fma(half_b, half_b, -c)), `reference_fma2` (... and
#       c = fma(-r, r, oc.oc)) and `contract` (three FMA chains) disagree on a few per cent of them -- plus the LLVM IR and the native code of
#       hit(::Sphere{Float32}, ...) in julia_hit_sphere_Float32.ll / .s (look for fmuladd / contract flags / vfmadd)
#   5. render(scene_2_spheres, default cam, 96, 16) with ONE thread (the reference's own smoke
#      render, test/runtests.jl:194) -> compare with the oracle's REF_SERIAL, ref_threads = 1
using RayTracingWeekend, StaticArrays, RandomNumbers.Xorshifts, Printf, InteractiveUtils

fmt(x::Float32) = @sprintf("%.9g", x)
fmt(x::Float64) = @sprintf("%.17g", x)
fmt(x::Integer) = string(x)
vals(xs) = join((fmt(x) for x in xs), " ")

println("julia_version: ", VERSION, " nthreads: ", Threads.nthreads())
for seed in (1, 2)
    r = Xoroshiro128Plus(seed)
    @printf("rng_state seed=%d: %016x %016x\n", seed, r.x, r.y)
    println("rng_u64 seed=$seed: ", join([@sprintf("%016x", rand(r, UInt64)) for _ in 1:16], " "))
    r = Xoroshiro128Plus(seed); println("rng_f32 seed=$seed: ", vals([rand(r, Float32) for _ in 1:16]))
    r = Xoroshiro128Plus(seed); println("rng_f64 seed=$seed: ", vals([rand(r, Float64) for _ in 1:16]))
end

kindof(::Lambertian) = 0
kindof(::Metal) = 1
kindof(::Dielectric) = 2
for T in (Float32, Float64)
    reseed!()
    s = scene_random_spheres(elem_type=T)
    println("scene $T n: ", length(s))
    for (i, h) in enumerate(s)
        m = h.mat
        alb = m isa Dielectric ? SVector{3,T}(1, 1, 1) : m.albedo
        par = m isa Metal ? m.fuzz : (m isa Dielectric ? m.ir : zero(T))
        println("sphere $T $(i-1): ", vals((h.center..., h.radius)), " ", kindof(m), " ", vals((alb..., par)))
    end
    for (name, cam) in (("t_default_cam", default_camera(SA{T}[0, 0, 0])),
                        ("t_cam1", default_camera([13, 2, 3], [0, 0, 0], [0, 1, 0], 20, 16 / 9, 0.1, 10.0; elem_type=T)),
                        ("t_cam2", default_camera([3, 3, 2], [0, 0, -1], [0, 1, 0], 20, 16 / 9, 2.0, sqrt(27.0); elem_type=T)))
        println("camera $name $T: ", vals((cam.origin..., cam.lower_left_corner..., cam.horizontal..., cam.vertical...,
                                           cam.u..., cam.v..., cam.w..., cam.lens_radius)))
    end
    v = SA{T}[0.3, -0.7, 0.2]
    println("normalize $T: ", vals(normalize(v)))
    println("dot $T: ", fmt(v ⋅ SA{T}[0.1, 0.2, 0.3]))
    println("tand $T: ", vals((tand(T(10)), tand(T(45)), tand(T(20) / 2))))
    # hit(::Sphere) on fixed rays: pins what @fastmath contracts in src/hit.jl:13-29
    rays = [(SA{T}[13, 2, 3], normalize(SA{T}[-13, -2.2, -3.1])), (SA{T}[0, 0, 0], normalize(SA{T}[0.1, -0.05, -1])),
            (SA{T}[0.3, 0.1, -0.6], normalize(SA{T}[-0.2, 0.4, -1])), (SA{T}[4, 1.5, 2], normalize(SA{T}[-1, -0.4, -0.55]))]
    sph = [Sphere(SA{T}[0, -1000, -1], T(1000), Lambertian(SA{T}[0.5, 0.5, 0.5])), Sphere(SA{T}[0, 0, -1], T(0.5), Lambertian(SA{T}[0.5, 0.5, 0.5])),
           Sphere(SA{T}[0, 0, -1], T(-0.4), Dielectric(T(1.5))), Sphere(SA{T}[0, 1, 0], T(1), Dielectric(T(1.5)))]
    for (ri, (o, d)) in enumerate(rays), (si, sp) in enumerate(sph)
        rec = RayTracingWeekend.hit(sp, RayTracingWeekend.Ray(o, d), T(1e-4), typemax(T))
        println("hit $T $ri $si: ", vals((o..., d..., sp.center..., sp.radius)), " -> ",
                rec === nothing ? "miss" : vals((rec.t, rec.p..., rec.n⃗..., rec.front_face ? 1 : 0)))
    end
    ground = Sphere(SA{T}[0, -1000, -1], T(1000), Lambertian(SA{T}[0.5, 0.5, 0.5]))
    us = (SA{T}[0.6, 0.1, -0.7], SA{T}[-0.3, -0.9, 0.2], SA{T}[0, -0.999, 0.02], SA{T}[0.5, 0.5, 0.5])
    for k in 0:15
        d = normalize(SA{T}[-13 + T(0.37) * k, T(-2.2) - T(0.03) * k, T(-3.1) + T(0.21) * k])
        rec = RayTracingWeekend.hit(ground, RayTracingWeekend.Ray(SA{T}[13, 2, 3], d), T(1e-4), typemax(T))
        rec === nothing && (println("selfhit $T $k: primary miss"); continue)
        outs = String[]
        for u in us
            rec2 = RayTracingWeekend.hit(ground, RayTracingWeekend.Ray(rec.p, normalize(rec.n⃗ + u)), T(1e-4), typemax(T))
            push!(outs, rec2 === nothing ? "miss" : fmt(rec2.t))
        end
        println("selfhit $T $k: ", vals((rec.t, rec.p...)), " -> ", join(outs, " "))
    end
    # 4c: adversarial rays for the evaluation order of the discriminant.  A 64-bit LCG (wrapping UInt64 arithmetic) picks points and
    # directions; EVERY value that enters hit() is printed (sphere, origin, direction): the checker does not have to reproduce this recipe.
    # Records 0 - 2047: rays leaving computed hit points on the r = 1000 ground sphere (the consequential regime: contract vs the rest);
    # 2048 - 3071: near-grazing rays at spheres of radius 0.05 ... 0.45; 3072 - 4095: rays leaving computed hit points on such spheres (r^2 is
    # inexact and c ~ 0 there: separates reference / reference_fma / reference_fma2).
    let st = UInt64(12345)
        nextu() = (st = st * 0x5851f42d4c957f2d + 0x14057b7ef767814f; T((st >> 40) % 0x100000) / T(0x100000))       # 20 bits: exact in Float32
        adv(k, sp, o, d) = begin
            rec2 = RayTracingWeekend.hit(sp, RayTracingWeekend.Ray(o, d), T(1e-4), typemax(T))
            println("adv $T $k: ", vals((sp.center..., sp.radius, o..., d...)), " -> ", rec2 === nothing ? "miss" : fmt(rec2.t))
        end
        for k in 0:2047
            tgt = SA{T}[T(-11) + T(22) * nextu(), T(0), T(-11) + T(22) * nextu()]
            rec = RayTracingWeekend.hit(ground, RayTracingWeekend.Ray(SA{T}[13, 2, 3], normalize(tgt - SA{T}[13, 2, 3])), T(1e-4), typemax(T))
            u = SA{T}[T(2) * nextu() - T(1), T(2) * nextu() - T(1), T(2) * nextu() - T(1)]
            rec === nothing && (println("adv $T $k: primary miss"); continue)
            adv(k, ground, rec.p, normalize(rec.n⃗ + T(0.98) * u))
        end
        for k in 2048:4095
            c = SA{T}[T(-11) + T(22) * nextu(), T(0.2), T(-11) + T(22) * nextu()]
            rr = T(0.05) + T(0.4) * nextu()                           # (a radius whose square is not exact: separates c = oc.oc - r^2 from fma(-r, r, oc.oc))
            sp = Sphere(c, rr, Lambertian(SA{T}[0.5, 0.5, 0.5]))
            o = SA{T}[T(-12) + T(24) * nextu(), T(0.05) + T(3) * nextu(), T(-12) + T(24) * nextu()]
            w = SA{T}[T(2) * nextu() - T(1), T(2) * nextu() - T(1), T(2) * nextu() - T(1)]
            d1 = normalize((c - o) + T(0.95) * rr * w)                # aimed at the sphere, missing its centre by up to its radius
            if k < 3072
                adv(k, sp, o, d1)
            else                                                       # 3072 - 4095: rays LEAVING the computed hit point on that small sphere (c = oc.oc - r^2 is ~ 0:
                rec = RayTracingWeekend.hit(sp, RayTracingWeekend.Ray(o, d1), T(1e-4), typemax(T))      #  the rounding of r^2 decides: reference_fma vs reference_fma2)
                rec === nothing && (println("adv $T $k: primary miss"); continue)
                u = SA{T}[T(2) * nextu() - T(1), T(2) * nextu() - T(1), T(2) * nextu() - T(1)]
                adv(k, sp, rec.p, normalize(rec.n⃗ + T(0.98) * u))
            end
        end
    end
    open("julia_hit_sphere_$(T).ll", "w") do io
        code_llvm(io, RayTracingWeekend.hit, (Sphere{T}, RayTracingWeekend.Ray{T}, T, T); debuginfo=:none)
    end
    open("julia_hit_sphere_$(T).s", "w") do io
        code_native(io, RayTracingWeekend.hit, (Sphere{T}, RayTracingWeekend.Ray{T}, T, T); debuginfo=:none)
    end
    Threads.nthreads() == 1 || @warn "run with -t1: the image depends on the thread count (SURVEY F6)"
    img = render(scene_2_spheres(elem_type=T), default_camera(SA{T}[0, 0, 0]), 96, 16)
    open("julia_render_2spheres_96x54_16spp_$(T).bin", "w") do io
        write(io, reinterpret(T, vec(img)))
    end
    println("render $T mean: ", fmt(sum(reinterpret(T, vec(img))) / T(3 * length(img))))
end
