# julia_bench.jl -- the reference's own CPU path timed beside the GPU (BASELINE.md section 2, baseline B1) and, when librtw_hip.so and an
# MI355X are there, the FIRST EXECUTION of the drop-in shim julia/RTWeekendHIP.jl.  This image has no julia: run on any box that has it.
#
#   julia --project=/path/to/RayTracingWeekend.jl -t 16 tools/julia_bench.jl [spp=16] [width=1920] > julia_bench.json
#
# Prints ONE JSON line: `cpu_baseline` in the shape bench.py uses (kind "reference": the real src/render.jl:8-44 with Threads.@threads over
# rows, src/render.jl:23), per element type, scaled from `spp` samples (per-sample work is i.i.d.: cost is linear in spp), depth 16 (the
# reference's only depth, src/ray_color.jl:14); and -- if the shim loads -- the same call through RTWeekendHIP.render with the frame compared
# statistically (the reference's per-thread serial RNG is not reproducible by a parallel device: SURVEY F6; mean abs diff of the gamma image).
using RayTracingWeekend, StaticArrays, Printf
spp = length(ARGS) >= 1 ? parse(Int, ARGS[1]) : 16
W = length(ARGS) >= 2 ? parse(Int, ARGS[2]) : 1920
H = W ÷ (16 // 9)
t_cam1(T) = default_camera([13, 2, 3], [0, 0, 0], [0, 1, 0], 20, 16 / 9, 0.1, 10.0; elem_type=T)      # src/proto/proto.jl:19
legs = String[]
hip = nothing
try
    include(joinpath(@__DIR__, "..", "julia", "RTWeekendHIP.jl"))
    global hip = RTWeekendHIP
catch e
    @warn "RTWeekendHIP not loaded (no librtw_hip.so / no GPU?): CPU legs only" exception = e
end
for T in (Float32, Float64)
    reseed!()                                                      # src/proto/proto.jl:198-199
    scene = scene_random_spheres(elem_type=T)
    cam = t_cam1(T)
    render(scene, cam, W, 1)                                       # warm-up / compilation
    dt = @elapsed img = render(scene, cam, W, spp)
    ms = W * H * spp / dt / 1e6
    push!(legs, @sprintf("{\"value\": %.4f, \"unit\": \"Msamples/s\", \"cores\": %d, \"kind\": \"reference\", \"dtype\": \"%s\", \"sample\": \"RayTracingWeekend.render(scene_random_spheres, t_cam1, %d, %d), depth 16, julia %s -t %d, %.2f s\"}",
                         ms, Threads.nthreads(), T, W, spp, VERSION, Threads.nthreads(), dt))
    if hip !== nothing
        try
            hip.render(scene, cam, W, 1)                           # context, scene upload
            dg = @elapsed gimg = hip.render(scene, cam, W, spp)
            a = reinterpret(T, vec(img)); b = reinterpret(T, vec(gimg))
            push!(legs, @sprintf("{\"value\": %.2f, \"unit\": \"Msamples/s\", \"kind\": \"RTWeekendHIP.render (first execution of the shim)\", \"dtype\": \"%s\", \"seconds\": %.4f, \"mean_abs_diff_vs_reference\": %.5f, \"image_mean\": [%.5f, %.5f]}",
                                 W * H * spp / dg / 1e6, T, dg, sum(abs.(a .- b)) / length(a), sum(a) / length(a), sum(b) / length(b)))
        catch e
            push!(legs, "{\"error\": \"RTWeekendHIP.render failed: $(replace(string(e), '"' => '\''))\"}")
        end
    end
end
println("{\"julia_bench\": [", join(legs, ", "), "]}")
