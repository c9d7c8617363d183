# RTWeekendHIP.jl -- the reference-side binding a maintainer of claforte/RayTracingWeekend.jl
# would add to route `render` through librtw_hip.so (C ABI: include/rtw_hip.h).
#
# NOT EXECUTED in this repository's CI: the build image has no `julia`.  It is kept small on
# purpose -- everything testable lives behind the C ABI (tests/ call exactly these entry points
# through ctypes).  See INTEGRATION.md.
#
# Usage (inside the reference package, after `include("RTWeekendHIP.jl")`):
#     using .RTWeekendHIP
#     img = RTWeekendHIP.render(scene_random_spheres(elem_type=Float32), t_cam1, 1920, 1000)
# Same positional signature and return type as RayTracingWeekend.render (src/render.jl:8-44):
# Matrix{RGB{T}} of size (image_width ÷ 16//9, image_width), gamma-2 applied, unclamped.
module RTWeekendHIP

using Images: RGB
using StaticArrays
using ..RayTracingWeekend: Sphere, Lambertian, Metal, Dielectric, Camera, HittableList, Hittable

const LIB = get(ENV, "RTW_HIP_LIB", joinpath(@__DIR__, "..", "raytracingweekend.jl_amd", "lib", "librtw_hip.so"))

# rtw_scene_f32 / rtw_scene_f64 (include/rtw_hip.h): SoA view of a HittableList of Sphere{T}
struct CScene{T}
    n::Int32
    cx::Ptr{T}; cy::Ptr{T}; cz::Ptr{T}; r::Ptr{T}
    kind::Ptr{Int32}
    ar::Ptr{T}; ag::Ptr{T}; ab::Ptr{T}
    param::Ptr{T}
end

# rtw_camera_*: the 22 scalars of Camera{T} in the field order of src/camera.jl:2-9
struct CCamera{T}
    origin::NTuple{3,T}; lower_left_corner::NTuple{3,T}; horizontal::NTuple{3,T}; vertical::NTuple{3,T}
    u::NTuple{3,T}; v::NTuple{3,T}; w::NTuple{3,T}
    lens_radius::T
end
CCamera(c::Camera{T}) where T = CCamera{T}(Tuple(c.origin), Tuple(c.lower_left_corner), Tuple(c.horizontal),
                                           Tuple(c.vertical), Tuple(c.u), Tuple(c.v), Tuple(c.w), c.lens_radius)

# rtw_params (ABI version 3: same layout as version 2; new flag bits)
struct CParams
    width::Int32; height::Int32; spp::Int32; max_depth::Int32
    seed::UInt64
    n_chunks::Int32; shard_index::Int32; shard_count::Int32; device::Int32; gamma::Int32; flags::Int32
    n_devices::Int32; job_pixels::Int32
    device_ids::Ptr{Int32}
end

function __init__()
    v = ccall((:rtw_abi_version, LIB), Cint, ())
    v == 4 || error("librtw_hip.so has ABI version $v; this shim binds version 4 (include/rtw_hip.h)")
end

matkind(::Lambertian) = Int32(0)
matkind(::Metal) = Int32(1)
matkind(::Dielectric) = Int32(2)
matkind(m) = throw(ArgumentError("unsupported material $(typeof(m)) on the HIP path"))
albedo(m::Lambertian{T}) where T = m.albedo
albedo(m::Metal{T}) where T = m.albedo
albedo(::Dielectric{T}) where T = SVector{3,T}(1, 1, 1)
matparam(::Lambertian{T}) where T = zero(T)
matparam(m::Metal{T}) where T = m.fuzz
matparam(m::Dielectric{T}) where T = m.ir

last_error() = unsafe_string(ccall((:rtw_last_error, LIB), Cstring, ()))

"""
    render(scene, cam, image_width=400, n_samples=1; depth=16, seed=1, n_chunks=0, device=-1, devices=nothing, numerics=:reference, group_cull=false, scan_valu=false, ray_pool=false, rccl_reduce=false)

Drop-in for `RayTracingWeekend.render` (src/render.jl:8-44) on MI355X.  Keyword extras only.
`depth=16` is the reference's hard-wired `ray_color` default (src/ray_color.jl:14).
`devices=:all` uses every visible GPU, `devices=[0, 1, 2]` the listed ones (the 8x8 tiles are dealt
round-robin to the devices inside the library; the image is identical for any device list).
`numerics` selects the deciding arithmetic of `hit(::Sphere)` (src/hit.jl:16-18): `:reference` (default) = the reference's own order -- StaticArrays' un-fused
`dot`, one rounding per written operation --, `:reference_fma2` = the same with both squares contracted, `disc = fma(half_b, half_b, -c)` and `c = fma(-r, r, oc⋅oc)`, `:contract` = three FMA chains
(RTW_FLAG_NUMERICS_*; in Float32 the choice moves the image mean by 0.003 and the work by 4 %: `tools/julia_kat.jl` tells which one this Julia build emits).
`group_cull=true` selects the opt-in culling scan (RTW_FLAG_GROUP_CULL), `scan_valu=true` the all-VALU form of either
scan (RTW_FLAG_SCAN_VALU, for A/B measurements): same image bit for bit in every mode; `ray_pool=true` (RTW_FLAG_RAY_POOL) needs a `make POOL=1` build of the library.
`rccl_reduce=true` (with `devices`): the shards are put together by one ncclReduce inside the library (RTW_FLAG_RCCL_REDUCE) instead of peer copies.
"""
function render(scene::HittableList, cam::Camera{T}, image_width=400, n_samples=1;
                depth=16, seed=1, n_chunks=0, device=-1, devices=nothing, numerics=:reference, group_cull=false, scan_valu=false, ray_pool=false, rccl_reduce=false) where T <: Union{Float32,Float64}
    numerics in (:reference, :contract, :reference_fma2) || throw(ArgumentError("numerics must be :reference, :contract or :reference_fma2"))
    nflags = numerics === :contract ? 32 : numerics === :reference_fma2 ? 128 : 0         # RTW_FLAG_NUMERICS_CONTRACT / _REFERENCE_FMA2
    image_height = image_width ÷ (16//9)                       # src/render.jl:11-12
    n = length(scene)
    cx = Vector{T}(undef, n); cy = similar(cx); cz = similar(cx); r = similar(cx)
    ar = similar(cx); ag = similar(cx); ab = similar(cx); param = similar(cx)
    kind = Vector{Int32}(undef, n)
    for (i, h) in enumerate(scene)
        h isa Sphere{T} || throw(ArgumentError("scene[$i] is $(typeof(h)); the HIP path takes Sphere{$T} only"))
        cx[i], cy[i], cz[i] = h.center
        r[i] = h.radius
        kind[i] = matkind(h.mat)
        ar[i], ag[i], ab[i] = albedo(h.mat)
        param[i] = matparam(h.mat)
    end
    img = Matrix{RGB{T}}(undef, image_height, image_width)      # column-major H x W, 3 x T per pixel
    ccam = Ref(CCamera(cam))
    ids = devices isa AbstractVector ? Int32.(devices) : Int32[]
    n_devices = devices === :all ? -1 : (length(ids) > 1 ? length(ids) : 0)
    length(ids) == 1 && (device = ids[1])
    rc = GC.@preserve cx cy cz r kind ar ag ab param img ids begin
        params = Ref(CParams(image_width, image_height, n_samples, depth, seed, n_chunks, 0, 1, device, 1, (group_cull ? 1 : 0) | (scan_valu ? 4 : 0) | (ray_pool ? 8 : 0) | (rccl_reduce ? 16 : 0) | nflags,
                             n_devices, 0, length(ids) > 1 ? pointer(ids) : Ptr{Int32}(C_NULL)))
        cscene = Ref(CScene{T}(n, pointer(cx), pointer(cy), pointer(cz), pointer(r), pointer(kind),
                               pointer(ar), pointer(ag), pointer(ab), pointer(param)))
        if T === Float32
            ccall((:rtw_render_f32, LIB), Cint, (Ref{CScene{Float32}}, Ref{CCamera{Float32}}, Ref{CParams}, Ptr{Float32}),
                  cscene, ccam, params, pointer(reinterpret(Float32, vec(img))))
        else
            ccall((:rtw_render_f64, LIB), Cint, (Ref{CScene{Float64}}, Ref{CCamera{Float64}}, Ref{CParams}, Ptr{Float64}),
                  cscene, ccam, params, pointer(reinterpret(Float64, vec(img))))
        end
    end
    rc == 0 || error("librtw_hip: error $rc: $(last_error())")
    img
end

end # module
