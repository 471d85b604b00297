# SpectralKernelsHIP.jl -- thin `ccall` shim over libspectralkernels_hip.so.
#
# Drop-in for the kernel_values path of SpectralKernels.jl: same constructor keywords as
# `AdaptiveKernelConfig` (src/adaptive.jl:24-27) and the same `kernel_values(cfg, xs; k0,
# param_derivative, verbose) -> (vals, errs)` contract (src/adaptive.jl:95-108).  All numerical
# work happens behind the C ABI declared in include/spectralkernels_hip.h; this file only
# marshals arguments.  (Julia is not installed in the build container, so this shim is
# exercised through its line-by-line ctypes mirror spectralkernels.jl_amd/api.py.)
module SpectralKernelsHIP

export AdaptiveKernelConfig, kernel_values, kernel_values_multi, kernel_values_pairs, build_dense_cov_matrix, compute_k0, MaternSDF, ExponentialSDF, GaussianSDF

const LIB = get(ENV, "SPECTRALKERNELS_HIP_LIB",
                joinpath(@__DIR__, "..", "libspectralkernels_hip.so"))

# catalogue SDFs evaluated on the device (sk_sdf_kind); any other callable goes through the
# batched host callback
struct MaternSDF;      phi::Float64; rho::Float64; nu::Float64; d::Float64; end
struct ExponentialSDF; a::Float64; end
struct GaussianSDF;    a::Float64; end
(s::MaternSDF)(w)      = s.phi*(s.rho^2 + w^2)^(-s.nu - s.d/2)
(s::ExponentialSDF)(w) = exp(-s.a*abs(w))
(s::GaussianSDF)(w)    = exp(-(w/s.a)^2)

# mirror of sk_config_args (field order and types must match the header)
struct SkConfigArgs
  sdf_kind::Cint
  sdf_params::NTuple{8,Cdouble}
  f::Ptr{Cvoid};  f_ctx::Ptr{Cvoid}
  df::Ptr{Cvoid}; df_ctx::Ptr{Cvoid}
  dim::Cint
  alpha::Cdouble
  tol::Cdouble
  derivative::Cint
  logw::Cint
  criteria::Cint
  has_tail::Cint
  tail::Cdouble
  m::Int64
  k::Int64
  device::Cint
  nufft_width::Cint
  hankel_min_lags::Cint
end

# void (*sk_sdf_batch_fn)(const double* w, double* out, size_t n, void* ctx)
function _sdf_batch(w::Ptr{Cdouble}, out::Ptr{Cdouble}, n::Csize_t, ctx::Ptr{Cvoid})::Cvoid
  f = unsafe_pointer_to_objref(ctx)[]
  @inbounds for i in 1:n
    unsafe_store!(out, Float64(f(unsafe_load(w, i))), i)
  end
  nothing
end

mutable struct AdaptiveKernelConfig{S}
  f::S
  handle::Ptr{Cvoid}
  fref::Tuple{Base.RefValue{Any},Base.RefValue{Any}}  # keeps the closures (f, df) alive for the callbacks
  dim::Int64; alpha::Float64; tol::Float64; derivative::Bool; logw::Bool
  convergence_criteria::Symbol; tail::Union{Float64,Nothing}; quadspec::Tuple{Int64,Int64}
end

_last_error() = unsafe_string(ccall((:sk_last_error, LIB), Cstring, ()))
_check(code) = (code == 0 || code == 100) ? code : error(_last_error())

_kind(::MaternSDF) = (Cint(1), s -> (s.phi, s.rho, s.nu, s.d, 0.0, 0.0, 0.0, 0.0))
_kind(::ExponentialSDF) = (Cint(2), s -> (s.a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_kind(::GaussianSDF) = (Cint(3), s -> (s.a, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0))
_kind(::Any) = (Cint(0), s -> ntuple(_ -> 0.0, 8))

function AdaptiveKernelConfig(f; df=nothing, dim=1, alpha=0.0, tol::Float64=1e-8, derivative=false,
                              logw=false, convergence_criteria::Symbol=:both, tail=nothing,
                              quadspec::Tuple{Int64,Int64}=(2^12, 2^4), device=-1)
  convergence_criteria in (:panel, :tails, :both) ||
    error("Argument convergence_criteria must be one of :panel, :tails, :both.")
  (kind, prm) = _kind(f)
  fref = Ref{Any}(f)
  cb   = kind == 0 ? @cfunction(_sdf_batch, Cvoid, (Ptr{Cdouble}, Ptr{Cdouble}, Csize_t, Ptr{Cvoid})) : C_NULL
  ctx  = kind == 0 ? pointer_from_objref(fref) : C_NULL
  # df: only read for callback SDFs with logw=true (src/quadrature.jl:194); catalogue SDFs differentiate on the device
  dfref = Ref{Any}(df)
  dcb   = (kind == 0 && !isnothing(df)) ? cb : C_NULL
  dctx  = (kind == 0 && !isnothing(df)) ? pointer_from_objref(dfref) : C_NULL
  crit = convergence_criteria == :both ? 0 : convergence_criteria == :panel ? 1 : 2
  args = SkConfigArgs(kind, prm(f), cb, ctx, dcb, dctx, dim, alpha, tol, derivative, logw, crit,
                      isnothing(tail) ? 0 : 1, isnothing(tail) ? 0.0 : Float64(tail),
                      quadspec[1], quadspec[2], device, 0, 0)
  h = Ref{Ptr{Cvoid}}(C_NULL)
  code = _check(ccall((:sk_config_create, LIB), Cint, (Ref{Ptr{Cvoid}}, Ref{SkConfigArgs}), h, args))
  code == 100 && @warn(_last_error())                       # src/adaptive.jl:38
  m = Ref{Int64}(0); k = Ref{Int64}(0)
  ccall((:sk_config_info, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Ptr{Cdouble}, Ref{Int64}, Ref{Int64}, Ptr{Cint}),
        h[], C_NULL, C_NULL, m, k, C_NULL)
  cfg = AdaptiveKernelConfig(f, h[], (fref, dfref), Int64(dim), Float64(alpha), tol, Bool(derivative), Bool(logw),
                             convergence_criteria, isnothing(tail) ? nothing : Float64(tail), (m[], k[]))
  finalizer(c -> ccall((:sk_config_destroy, LIB), Cvoid, (Ptr{Cvoid},), c.handle), cfg)
  cfg
end

function compute_k0(cfg::AdaptiveKernelConfig)
  k0 = Ref{Cdouble}(0.0)
  _check(ccall((:sk_compute_k0, LIB), Cint, (Ptr{Cvoid}, Ref{Cdouble}), cfg.handle, k0))
  k0[]
end

function kernel_values(cfg::AdaptiveKernelConfig, xs::AbstractVector{Float64};
                       k0=nothing, param_derivative=false, verbose=false)
  x    = Vector{Float64}(xs)
  vals = Vector{Float64}(undef, length(x))
  errs = Vector{Float64}(undef, length(x))
  GC.@preserve cfg begin
    _check(ccall((:sk_kernel_values, LIB), Cint,
                 (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Cdouble, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
                 cfg.handle, x, length(x), isnothing(k0) ? 0.0 : Float64(k0), isnothing(k0) ? 0 : 1,
                 param_derivative ? 1 : 0, vals, errs))
  end
  (vals, errs)
end

# Several configs at one lag set -- kernel_sdf_derivatives / kernel_warping_gradients /
# kernel_singularity_derivative (src/derivatives.jl:51-81) in one call: one upload, one device sort + dedup.
# Returns a length(cfgs) x length(xs) matrix of values and one of error estimates.
function kernel_values_multi(cfgs::AbstractVector{<:AdaptiveKernelConfig}, xs::AbstractVector{Float64};
                             k0s, param_derivative=fill(false, length(cfgs)))
  x    = Vector{Float64}(xs)
  nc   = length(cfgs)
  hs   = Ptr{Cvoid}[c.handle for c in cfgs]
  ks   = Float64[k for k in (k0s isa Number ? fill(Float64(k0s), nc) : k0s)]
  pd   = Cint[p ? 1 : 0 for p in param_derivative]
  vals = Matrix{Float64}(undef, length(x), nc)                  # column c = config c (row-major nc x n on the C side)
  errs = Matrix{Float64}(undef, length(x), nc)
  GC.@preserve cfgs begin
    _check(ccall((:sk_kernel_values_multi, LIB), Cint,
                 (Ptr{Ptr{Cvoid}}, Csize_t, Ptr{Cdouble}, Csize_t, Ptr{Cdouble}, Ptr{Cint}, Ptr{Cdouble}, Ptr{Cdouble}),
                 hs, nc, x, length(x), ks, pd, vals, errs))
  end
  (permutedims(vals), permutedims(errs))
end

# Lags of point pairs formed on the device -- what gen_kernel_setup / gen_kernel do with
# `norm(warp_pts[j] - warp_pts[k])` (src/model.jl:64-65,76).  pts: vector of points (numbers or static vectors),
# pairs: vector of 1-based index tuples as in sm.kernel_index_pairs.
function kernel_values_pairs(cfg::AdaptiveKernelConfig, pts::AbstractVector, pairs::AbstractVector{<:Tuple{Integer,Integer}};
                             k0=nothing, param_derivative=false)
  pdim = length(first(pts))
  p    = Matrix{Float64}(undef, pdim, length(pts))            # column-major pdim x npts == row-major npts x pdim
  for (j, pt) in enumerate(pts), d in 1:pdim
    p[d, j] = pt[d]
  end
  ix   = Matrix{Int64}(undef, 2, length(pairs))
  for (e, (j, k)) in enumerate(pairs)
    ix[1, e] = j - 1; ix[2, e] = k - 1
  end
  vals = Vector{Float64}(undef, length(pairs))
  errs = Vector{Float64}(undef, length(pairs))
  GC.@preserve cfg begin
    _check(ccall((:sk_kernel_values_pairs, LIB), Cint,
                 (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Cint, Ptr{Int64}, Csize_t, Cdouble, Cint, Cint, Ptr{Cdouble}, Ptr{Cdouble}),
                 cfg.handle, p, length(pts), pdim, ix, length(pairs), isnothing(k0) ? 0.0 : Float64(k0),
                 isnothing(k0) ? 0 : 1, param_derivative ? 1 : 0, vals, errs))
  end
  (vals, errs)
end

# build_dense_cov_matrix(cfg, pts) -- src/utils.jl:41-64
function build_dense_cov_matrix(cfg::AdaptiveKernelConfig, pts::AbstractVector)
  pdim = length(first(pts))
  p    = Matrix{Float64}(undef, pdim, length(pts))
  for (j, pt) in enumerate(pts), d in 1:pdim
    p[d, j] = pt[d]
  end
  out = Matrix{Float64}(undef, length(pts), length(pts))       # symmetric: layout does not matter
  GC.@preserve cfg begin
    _check(ccall((:sk_dense_cov_matrix, LIB), Cint, (Ptr{Cvoid}, Ptr{Cdouble}, Csize_t, Cint, Ptr{Cdouble}),
                 cfg.handle, p, length(pts), pdim, out))
  end
  out
end

end # module
