# julia_kat.jl -- run on ANY box that has Julia + the reference package to pin the oracle's
# unverified third-party restatements (DESIGN.md section 3, tier T2).  This image has no julia.
#
#   julia --project=/path/to/RayTracingWeekend.jl -t1 tools/julia_kat.jl > julia_kat.txt
#
# then compare with tests/golden/rng_provisional.npz and the oracle's REF_SERIAL render:
#   1. Xoroshiro128Plus(seed) state after construction, first 16 UInt64 / Float32 / Float64
#   2. scene_random_spheres(elem_type=T) after reseed!()  (SoA dump)
#   3. default_camera presets (22 scalars)
#   4. render(scene_2_spheres, default cam, 96, 16) with ONE thread (the reference's own smoke
#      render, test/runtests.jl:194) -> compare with oracle REF_SERIAL, ref_threads = 1
using RayTracingWeekend, StaticArrays, RandomNumbers.Xorshifts, Printf

for seed in (1, 2)
    r = Xoroshiro128Plus(seed)
    @printf("rng seed=%d state=(%016x,%016x)\n", seed, r.x, r.y)
    println("  u64: ", join([@sprintf("%016x", rand(r, UInt64)) for _ in 1:16], " "))
    r = Xoroshiro128Plus(seed); println("  f32: ", join([@sprintf("%.9g", rand(r, Float32)) for _ in 1:16], " "))
    r = Xoroshiro128Plus(seed); println("  f64: ", join([@sprintf("%.17g", rand(r, Float64)) for _ in 1:16], " "))
end

for T in (Float32, Float64)
    reseed!()
    s = scene_random_spheres(elem_type=T)
    println("scene_random_spheres $T n=", length(s))
    for (i, h) in enumerate(s)
        m = h.mat
        @printf("  %d c=(%.9g,%.9g,%.9g) r=%.9g %s", i, h.center..., h.radius, nameof(typeof(m)))
        m isa Lambertian && @printf(" albedo=(%.9g,%.9g,%.9g)", m.albedo...)
        m isa Metal && @printf(" albedo=(%.9g,%.9g,%.9g) fuzz=%.9g", m.albedo..., m.fuzz)
        m isa Dielectric && @printf(" ir=%.9g", m.ir)
        println()
    end
    for (name, cam) in (("t_default_cam", default_camera(SA{T}[0, 0, 0])),
                        ("t_cam1", default_camera([13, 2, 3], [0, 0, 0], [0, 1, 0], 20, 16 / 9, 0.1, 10.0; elem_type=T)))
        println("camera $name $T: ", cam)
    end
    # normalize / dot conventions of StaticArrays
    v = SA{T}[0.3, -0.7, 0.2]
    println("normalize $T: ", normalize(v), "  dot: ", v ⋅ SA{T}[0.1, 0.2, 0.3], " tand(10): ", tand(T(10)))
    Threads.nthreads() == 1 || @warn "run with -t1: the image depends on the thread count (SURVEY F6)"
    img = render(scene_2_spheres(elem_type=T), default_camera(SA{T}[0, 0, 0]), 96, 16)
    open("julia_render_2spheres_96x54_16spp_$(T).bin", "w") do io
        write(io, reinterpret(T, vec(img)))
    end
    println("render $T mean=", sum(reinterpret(T, vec(img))) / (3 * length(img)))
end
