# PMCHip.jl -- ccall glue between ParticlesMC.jl and libpmc_hip.so (include/pmc.h).
#
# This file is the code of INTEGRATION.md sections 1-4 in one place, so that a ParticlesMC maintainer can
# `include("PMCHip.jl")` it next to the built library.  It was written against include/pmc.h and the reference's own
# types; it could NOT be executed in the build image (no `julia`, no Arianna / StaticArrays: SURVEY F2, F3), so treat
# it as reviewed-but-untested glue.  The tested twin of every call sequence in here is the Python host in
# particlesmc_amd/ (same call order, same argument meaning), which the GPU test-suite exercises.
#
#   using ParticlesMC, Arianna, StaticArrays
#   include("PMCHip.jl"); using .PMCHip
#   chains = load_chains("config.xyz", args=Dict("temperature" => [1.0], "model" => ["LJ"], "list_type" => "HIPLinkedList"))
#
module PMCHip
using StaticArrays
import Random
import ParticlesMC, Arianna
const LIB = joinpath(@__DIR__, "libpmc_hip.so")

struct PairModel            # pmc_pair_model
    kind::Int32; n::Int32
    eps::Float64; eps4::Float64; sigma::Float64; sigma2::Float64
    rcut::Float64; rcut2::Float64; shift::Float64
    ndiv2_i::Int64; ndiv2_f::Float64
    C0::Float64; C2_s2::Float64; C4_s4::Float64
end
# the constants are the fields ParticlesMC's constructors already computed (src/models.jl:110-119, 64-70, 150-160)
PairModel(m::ParticlesMC.LennardJones)       = PairModel(0, 0, m.ϵ, m.ϵ4, m.σ, m.σ2, m.rcut, m.rcut2, m.shift, 0, 0.0, 0.0, 0.0, 0.0)
PairModel(m::ParticlesMC.SoftSpheres)        = PairModel(m.ndiv2 isa Int ? 1 : 2, m.n, m.ϵ, 0.0, m.σ, m.σ2, m.rcut, m.rcut2, m.shift,
                                                         m.ndiv2 isa Int ? m.ndiv2 : 0, m.ndiv2 isa Int ? 0.0 : m.ndiv2, 0.0, 0.0, 0.0)
PairModel(m::ParticlesMC.SmoothLennardJones) = PairModel(3, 0, m.ϵ, m.ϵ4, m.σ, m.σ2, m.rcut, m.rcut2, 0.0, 0, 0.0, m.C0, m.C2_σ2, m.C4_σ4)

struct Params               # pmc_params
    N::Int64; d::Int32; nspecies::Int32
    box::NTuple{3,Float64}; cell::NTuple{3,Float64}; ncells::NTuple{3,Int64}
    temperature::Float64; density::Float64
    models::Ptr{PairModel}
    device::Int32; chain::UInt32; seed::UInt64
end

struct CMove                # pmc_move
    kind::Int32; species::NTuple{2,Int32}; probability::Float64; sigma::Float64
end

struct Stats                # pmc_stats
    attempted::NTuple{8,Int64}; accepted::NTuple{8,Int64}
    energy::Float64; energy_fix::Int64
    kernel_ms::Float64; sweep_kernel_ms::Float64; sweep_kernel_launches::Int64
    pair_tests::Int64; pairs_in_range::Int64; null_moves::Int64
    dense_evals::Int64; staged_candidates::Int64; written_back::Int64
end

check(rc, h=C_NULL) = rc == 0 || error(unsafe_string(ccall((:pmc_last_error, LIB), Cstring, (Ptr{Cvoid},), h)))
pad3(v, fill) = ntuple(a -> a <= length(v) ? v[a] : fill, 3)

mutable struct HIPLinkedList{T<:AbstractFloat,d} <: ParticlesMC.NeighbourList
    box::SVector{d,T}          # cell size, as in LinkedList (src/neighbours.jl:226)
    ncells::NTuple{d,Int}
    rcut::T
    N::Int
    handle::Ptr{Cvoid}
    device::Int32; seed::UInt64; chain::UInt32
end

function HIPLinkedList(box, rcut, N; list_parameters=nothing)
    cell = box ./ fld.(box, rcut)                                   # src/neighbours.jl:237 -- computed in Julia,
    ncells = ntuple(a -> Int(box[a] / cell[a]), length(box))        # :238            so membership is bit-exact
    lp = something(list_parameters, Dict())
    HIPLinkedList(cell, ncells, rcut, N, C_NULL, Int32(get(lp, "device", 0)), UInt64(get(lp, "seed", 0)), UInt32(get(lp, "chain", 0)))
end

function ParticlesMC.build_neighbour_list!(system::ParticlesMC.Particles, L::HIPLinkedList)
    if L.handle == C_NULL
        models = [PairModel(system.model_matrix[i, j]) for j in axes(system.model_matrix, 2), i in axes(system.model_matrix, 1)] # row-major [i][j]
        GC.@preserve models begin
            p = Params(system.N, system.d, size(system.model_matrix, 1), pad3(system.box, 1.0), pad3(L.box, 1.0), pad3(L.ncells, 1),
                       system.temperature, system.density, pointer(models), L.device, L.chain, L.seed)
            h = Ref{Ptr{Cvoid}}(C_NULL)
            check(ccall((:pmc_create, LIB), Cint, (Ref{Params}, Ref{Ptr{Cvoid}}), p, h))
            L.handle = h[]
            finalizer(l -> ccall((:pmc_destroy, LIB), Cint, (Ptr{Cvoid},), l.handle), L)
        end
    end
    pos = reinterpret(Float64, system.position)                     # Vector{SVector{d,Float64}} is already AoS N×d
    sp = Int32.(system.species)
    GC.@preserve pos sp begin
        check(ccall((:pmc_upload, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}), L.handle, pos, sp), L.handle)
    end
    check(ccall((:pmc_build_cells, LIB), Cint, (Ptr{Cvoid},), L.handle), L.handle)
    # NOT here: system.energy[1].  System() calls this function at src/atoms.jl:50 while `energy` is still
    # Vector{T}(undef, 1) (:44); the value only exists after :51-56.  The running energy is handed over on every use
    # instead (handover! below): the Julia-side system.energy[1] is the authority between calls, exactly as in the
    # reference, whose wrapper (src/moves.jl:11-20) and revert_action! (:76-81, :201-207) keep it on the host.
    return nothing
end

# system.energy[1] -> the handle's running energy.  One host-side store in the library (no device work); called at the top
# of every entry point whose result depends on the running energy.
handover!(system) = check(ccall((:pmc_set_energy, LIB), Cint, (Ptr{Cvoid}, Float64), system.neighbour_list.handle, system.energy[1]),
                          system.neighbour_list.handle)

# `neighbour_list(system, i)` (src/neighbours.jl:371-377): the ids the reference's LinkedIterator (:316-365) yields for
# particle i, materialised -- i itself included, as in the reference.  Compatibility path for code that iterates the list
# itself (e.g. the generic compute_energy_particle of src/atoms.jl:81-88); the overloads below never go through it.
function (L::HIPLinkedList)(system::ParticlesMC.Particles, i::Int)
    n = Ref{Int64}(0)
    out = Vector{Int64}(undef, 1024)
    rc = ccall((:pmc_neighbours_of, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ref{Int64}), L.handle, i, out, length(out), n)
    if rc != 0 && n[] > length(out)                                  # a denser neighbourhood than the first guess
        resize!(out, n[])
        rc = ccall((:pmc_neighbours_of, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Int64}, Int64, Ref{Int64}), L.handle, i, out, length(out), n)
    end
    check(rc, L.handle)
    return resize!(out, n[])
end

function Base.getproperty(L::HIPLinkedList, s::Symbol)              # neighbour_list.cs (src/neighbours.jl:49-51)
    if s === :cs
        cs = Vector{Int}(undef, getfield(L, :N)); check(ccall((:pmc_get_cells, LIB), Cint, (Ptr{Cvoid}, Ptr{Int64}), getfield(L, :handle), cs)); return cs
    end
    return getfield(L, s)
end

function ParticlesMC.compute_energy_particle(system::ParticlesMC.Atoms, i, L::HIPLinkedList)   # src/atoms.jl:81-88
    e = Ref{Float64}(0.0)
    check(ccall((:pmc_energy_particle, LIB), Cint, (Ptr{Cvoid}, Int64, Ref{Float64}), L.handle, i, e), L.handle)
    return e[]
end

function ParticlesMC.perform_action!(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}, action::ParticlesMC.Displacement) where {D,V}
    e1, e2, de = Ref(0.0), Ref(0.0), Ref(0.0)                        # replaces src/moves.jl:57-67; the wrapper (:11-20) then adds δe to
    handover!(system)                                                # system.energy[1] on the host, the library does the same on its copy
    δ = collect(Float64, action.δ)
    check(ccall((:pmc_displacement_perform, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                system.neighbour_list.handle, action.i, δ, e1, e2, de), system.neighbour_list.handle)
    action.δe = de[]
    return e1[], e2[]
end
# revert_action! -> pmc_displacement_revert(h, action.i, action.δ (already inverted), action.δe)      src/moves.jl:76-81
# DiscreteSwap   -> pmc_swap_perform(h, i, j, e1, e2, de) / pmc_swap_revert(h, i, j, δe)               src/moves.jl:187-207

# ASSUMPTIONS about Arianna (not in the reference tree, never executed here; INTEGRATION.md section 4): the supertype
# `Arianna.Algorithm`, the per-tick hook `Arianna.make_step!(simulation, algorithm)` and the fields of `Arianna.Move`.
struct HIPMetropolis <: Arianna.Algorithm
    pool::Vector{CMove}; mode::Int32            # 0 = sequential (reference chain), 1 = checkerboard (throughput)
    sweeps_per_step::Int64                      # sweeps per scheduler tick: one library call (one sync, one counter fold) per tick,
end                                             # so coarse schedulers should tick in blocks (INTEGRATION.md section 4)
# default: ONE sweep of N attempted moves per scheduler tick, as the reference's Metropolis entry (sweepstep = length(chains[1]),
# src/ParticlesMC.jl:246) -- callbacks and schedulers fire at the reference's cadence.  Coarser ticks are an explicit opt-in:
# HIPMetropolis(pool, 1; sweeps_per_step = 4m) lets a pass of m sweeps per grid shift (pmc_set_rounds) stay inside one call
# (a pass never spans calls: include/pmc.h), at the price of a tick that is 4m sweeps long.
HIPMetropolis(pool, mode; sweeps_per_step=1) = HIPMetropolis(pool, mode, sweeps_per_step)

# how mode = 0 runs: 0 the library's choice, 1 one wavefront, 2 speculative windows, 3 the chain's state in LDS (include/pmc.h)
set_sequential_engine!(system, engine, window=0) =
    check(ccall((:pmc_set_sequential_engine, LIB), Cint, (Ptr{Cvoid}, Int32, Int32), system.neighbour_list.handle, engine, window),
          system.neighbour_list.handle)

# checkpoint / restart of a mode = 1 chain: the periodic images travel with the positions (include/pmc.h: pmc_get_images)
function get_images(system)
    img = Vector{Int32}(undef, system.N * system.d)
    check(ccall((:pmc_get_images, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), system.neighbour_list.handle, img), system.neighbour_list.handle)
    return img
end
set_images!(system, img::Vector{Int32}) =
    check(ccall((:pmc_set_images, LIB), Cint, (Ptr{Cvoid}, Ptr{Int32}), system.neighbour_list.handle, img), system.neighbour_list.handle)
cmove(m::Arianna.Move{<:ParticlesMC.Displacement}) = CMove(0, (0, 0), m.weight, m.parameters.σ)
cmove(m::Arianna.Move{<:ParticlesMC.DiscreteSwap}) = CMove(1, Int32.(m.action.species), m.weight, 0.0)

function Arianna.make_step!(sim::Arianna.Simulation, alg::HIPMetropolis)          # one scheduler tick = alg.sweeps_per_step sweeps (default 1)
    for system in sim.chains                                                      # parallel=true: Threads.@threads, one handle per chain
        st = Ref{Stats}()
        handover!(system)
        check(ccall((:pmc_sweep, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{CMove}, Int32, Int64, Ref{Stats}),
                    system.neighbour_list.handle, alg.mode, alg.pool, length(alg.pool), alg.sweeps_per_step, st), system.neighbour_list.handle)
        system.energy[1] = st[].energy                                             # energy callback, src/utils.jl:51-53
        # st[].attempted / .accepted feed StoreAcceptance
    end
end

# ---- the remaining per-move verbs (src/moves.jl:76-90, 187-214) ---------------------------------------------------

function Arianna.revert_action!(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}, action::ParticlesMC.Displacement) where {D,V}
    h = system.neighbour_list.handle                                # action.δ was inverted by invert_action! (src/moves.jl:88-90)
    δ = collect(Float64, action.δ)
    handover!(system)
    check(ccall((:pmc_displacement_revert, LIB), Cint, (Ptr{Cvoid}, Int64, Ptr{Float64}, Float64), h, action.i, δ, action.δe), h)
    system.energy[1] -= action.δe                                   # src/moves.jl:79 (the library subtracts it from its copy too)
    return nothing
end

function ParticlesMC.perform_action!(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}, action::ParticlesMC.DiscreteSwap) where {D,V}
    h = system.neighbour_list.handle
    e1, e2, de = Ref(0.0), Ref(0.0), Ref(0.0)
    handover!(system)
    check(ccall((:pmc_swap_perform, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Ref{Float64}, Ref{Float64}, Ref{Float64}),
                h, action.i, action.j, e1, e2, de), h)
    action.δe = de[]
    return e1[], e2[]
end

function Arianna.revert_action!(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}, action::ParticlesMC.DiscreteSwap) where {D,V}
    h = system.neighbour_list.handle                                # called with the inverted action (src/moves.jl:212-214)
    handover!(system)
    check(ccall((:pmc_swap_revert, LIB), Cint, (Ptr{Cvoid}, Int64, Int64, Float64), h, action.i, action.j, action.δe), h)
    system.energy[1] -= action.δe                                   # src/moves.jl:206
    return nothing
end

# ---- keeping the Julia-side arrays current for IO and callbacks (src/IO/IO.jl:365-372, src/utils.jl:51-53) ---------

function refresh!(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}) where {D,V}
    h = system.neighbour_list.handle
    pos = reinterpret(Float64, system.position)
    sp = Vector{Int32}(undef, system.N)
    GC.@preserve pos sp begin
        check(ccall((:pmc_download, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int32}), h, pos, sp), h)
    end
    system.species .= sp
    e = Ref(0.0)
    check(ccall((:pmc_get_energy, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), h, e), h)
    system.energy[1] = e[]
    return system
end

# Temperature.  `Atoms` is an immutable struct (src/atoms.jl:18): `system.temperature = T` does not exist, in the reference or
# here.  A tempering run changes temperatures, so the CURRENT temperature of a HIP-backed chain lives on its handle
# (system.temperature stays the value the chain was created with) and the two target-density functions of
# src/utils.jl:4-10 read it from there.
set_temperature!(system, T) = check(ccall((:pmc_set_temperature, LIB), Cint, (Ptr{Cvoid}, Float64), system.neighbour_list.handle, T),
                                    system.neighbour_list.handle)
function temperature(system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}) where {D,V}
    T = Ref(0.0)
    check(ccall((:pmc_get_temperature, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), system.neighbour_list.handle, T), system.neighbour_list.handle)
    return T[]
end
Arianna.unnormalised_log_target_density(e, system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}) where {D,V} = -e / temperature(system)
Arianna.delta_log_target_density(e1, e2, system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}) where {D,V} = -(e2 - e1) ./ temperature(system)

# ---- the reference's other list types on the device's reference-order engine --------------------------------------
# Construct HIPLinkedList with the geometry of that list (pmc_cell_list_geometry / pmc_verlet_list_geometry) and call
# this right after pmc_create: 0 EmptyList, 1 LinkedList, 2 CellList, 3 VerletList (dr = list_parameters["dr"]).
set_list_type!(L::HIPLinkedList, list_type::Integer, dr::Real=0.0) =
    check(ccall((:pmc_set_list_type, LIB), Cint, (Ptr{Cvoid}, Int32, Float64), L.handle, list_type, dr), L.handle)

# ---- all the chains of a run (parallel = true) in shared launches: pmc_sweep_many ------------------------------------
# Chains of one shape (same N, box, models) share every kernel launch; anything else runs as one stream per handle.

function sweep_many!(systems, pool::Vector{CMove}, nsweeps::Integer; mode::Integer=1)
    foreach(handover!, systems)
    hs = Ptr{Cvoid}[s.neighbour_list.handle for s in systems]
    stats = Vector{Stats}(undef, length(systems))
    GC.@preserve hs stats pool begin
        check(ccall((:pmc_sweep_many, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int32, Int32, Ptr{CMove}, Int32, Int64, Ptr{Stats}),
                    hs, length(hs), mode, pool, length(pool), nsweeps, stats), hs[1])
    end
    for (k, s) in enumerate(systems)
        s.energy[1] = stats[k].energy
    end
    return stats
end

# ---- parallel tempering (the reference has none: README.md:33) -------------------------------------------------------
# label[r] = index (0-based) of the temperature replica r holds; the library reads the handles' running energies,
# decides, and sets every handle's temperature.  Replicas of one process (one GPU, or one handle per device):
function replica_exchange!(systems, ladder::Vector{Float64}, seed::Integer, round::Integer, label::Vector{Int32})
    foreach(handover!, systems)
    hs = Ptr{Cvoid}[s.neighbour_list.handle for s in systems]
    nacc = Ref{Int32}(0)
    GC.@preserve hs ladder label begin
        check(ccall((:pmc_replica_exchange, LIB), Cint, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Float64}, UInt64, UInt64, Ptr{Int32}, Ref{Int32}),
                    hs, length(hs), ladder, seed, round, label, nacc), hs[1])
    end
    # (the library has set every handle's temperature to ladder[label[r] + 1]: temperature(systems[r]) returns it)
    return nacc[]
end
# One process per GPU: `comm` from pmc_rccl_comm_create (rank 0 calls pmc_rccl_unique_id and ships the 128 bytes, e.g. with
# MPI.Bcast or a file); the round is ONE ncclAllGather of a Float64 per rank over RCCL / xGMI.
function replica_exchange_rccl!(system, comm::Ptr{Cvoid}, rank::Integer, world::Integer, ladder::Vector{Float64}, seed::Integer,
                                round::Integer, label::Vector{Int32})
    h = system.neighbour_list.handle
    nacc = Ref{Int32}(0)
    handover!(system)
    GC.@preserve ladder label begin
        check(ccall((:pmc_replica_exchange_rccl, LIB), Cint,
                    (Ptr{Cvoid}, Ptr{Cvoid}, Int32, Int32, Ptr{Float64}, UInt64, UInt64, Ptr{Int32}, Ref{Int32}),
                    h, comm, rank, world, ladder, seed, round, label, nacc), h)
    end
    return nacc[]                                                   # temperature(system) == ladder[label[rank + 1] + 1] from here on
end

# ---- EnergyBias on the device (src/moves.jl:246-280): nothing of size N crosses the bus ------------------------------
function Arianna.sample_action!(action::ParticlesMC.DiscreteSwap, ::ParticlesMC.EnergyBias, parameters,
                                system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}, rng) where {D,V}
    i, j, lq = Ref{Int64}(0), Ref{Int64}(0), Ref(0.0)
    h = system.neighbour_list.handle
    check(ccall((:pmc_energy_bias_sample, LIB), Cint,
                (Ptr{Cvoid}, Int32, Int32, Float64, Float64, Float64, Float64, Ref{Int64}, Ref{Int64}, Ref{Float64}),
                h, action.species[1], action.species[2], parameters.θ₁, parameters.θ₂, rand(rng), rand(rng), i, j, lq), h)
    action.i, action.j = i[], j[]
    return nothing
end
function Arianna.log_proposal_density(action::ParticlesMC.DiscreteSwap, ::ParticlesMC.EnergyBias, parameters,
                                      system::ParticlesMC.Atoms{D,V,<:HIPLinkedList}) where {D,V}
    lq = Ref(0.0)
    h = system.neighbour_list.handle
    check(ccall((:pmc_energy_bias_logq, LIB), Cint, (Ptr{Cvoid}, Int32, Int32, Float64, Float64, Int64, Int64, Ref{Float64}),
                h, action.species[1], action.species[2], parameters.θ₁, parameters.θ₂, action.i, action.j, lq), h)
    return lq[]
end

# ---- the device's sequential random stream as an AbstractRNG (the Julia twin of particlesmc_amd.SequentialStream) ----
# Step k of the device's reference-order chain (pmc_sweep, mode 0) consumes one pair of Philox blocks (R, Q) =
# pmc_sequential_words(seed, chain, k):  move index from R[1] against the pool's cumulative weights, particle
# i = mulhi(R[2], N) + 1, ln u from (R[3], R[4]), displacement = sigma * BoxMuller(Q[1], Q[2]), BoxMuller(Q[3], Q[4])
# (Julia 1-based indices into the NTuples below).  With the sample_action! / accept overloads underneath, Arianna's own
# mc_step! driven by this RNG proposes and decides exactly what the device does, step for step: that is how the real
# reference can be compared with the device move by move once a Julia binary is at hand.
mutable struct SequentialWords <: Random.AbstractRNG
    seed::UInt64
    chain::UInt32
    step::UInt64
    R::NTuple{4,UInt32}
    Q::NTuple{4,UInt32}
    cursor::Int                     # next of the 8 words rand(rng, UInt32) hands out (generic consumers)
end

function load!(rng::SequentialWords)
    R = Ref{NTuple{4,UInt32}}((0, 0, 0, 0)); Q = Ref{NTuple{4,UInt32}}((0, 0, 0, 0))
    check(ccall((:pmc_sequential_words, LIB), Cint, (UInt64, UInt32, UInt64, Ref{NTuple{4,UInt32}}, Ref{NTuple{4,UInt32}}),
                rng.seed, rng.chain, rng.step, R, Q))
    rng.R, rng.Q, rng.cursor = R[], Q[], 1
    return rng
end
SequentialWords(seed::Integer, chain::Integer=0, step::Integer=0) =
    load!(SequentialWords(UInt64(seed), UInt32(chain), UInt64(step), (0, 0, 0, 0), (0, 0, 0, 0), 1))
advance!(rng::SequentialWords) = (rng.step += 1; load!(rng))

# generic AbstractRNG interface: the eight words of the current step in order, then the next step's
function Random.rand(rng::SequentialWords, ::Random.SamplerType{UInt32})
    rng.cursor > 8 && advance!(rng)
    w = rng.cursor <= 4 ? rng.R[rng.cursor] : rng.Q[rng.cursor - 4]
    rng.cursor += 1
    return w
end
Random.rand(rng::SequentialWords, ::Random.SamplerType{UInt64}) = (UInt64(rand(rng, UInt32)) << 32) | UInt64(rand(rng, UInt32))
Random.rng_native_52(::SequentialWords) = UInt64

mulhi(w::UInt32, n::Integer) = Int((UInt64(w) * UInt64(n)) >> 32)
function device_normals(rng::SequentialWords)
    z = ntuple(_ -> Ref(0.0), 4)
    ccall((:pmc_box_muller, LIB), Cint, (UInt32, UInt32, Ref{Float64}, Ref{Float64}), rng.Q[1], rng.Q[2], z[1], z[2])
    ccall((:pmc_box_muller, LIB), Cint, (UInt32, UInt32, Ref{Float64}, Ref{Float64}), rng.Q[3], rng.Q[4], z[3], z[4])
    return map(getindex, z)
end
function device_log_uniform(rng::SequentialWords)
    v = Ref(0.0)
    ccall((:pmc_log_uniform, LIB), Cint, (UInt32, UInt32, Ref{Float64}), rng.R[3], rng.R[4], v)
    return v[]
end

# SimpleGaussian with the device's draws (replaces src/moves.jl:114-123 when rng isa SequentialWords)
function Arianna.sample_action!(action::ParticlesMC.Displacement, ::ParticlesMC.SimpleGaussian, parameters, system::ParticlesMC.Particles,
                                rng::SequentialWords)
    action.i = mulhi(rng.R[2], length(system)) + 1
    z = device_normals(rng)
    action.δ = SVector{system.d,Float64}(ntuple(a -> parameters.σ * z[a], system.d))
    return nothing
end
# DoubleUniform with the device's draws (replaces src/moves.jl:235-241)
function Arianna.sample_action!(action::ParticlesMC.DiscreteSwap, ::ParticlesMC.DoubleUniform, parameters, system::ParticlesMC.Particles,
                                rng::SequentialWords)
    ids1 = system.species_list.sp_ids[action.species[1]]; ids2 = system.species_list.sp_ids[action.species[2]]
    action.i = ids1[mulhi(rng.R[2], length(ids1)) + 1]
    action.j = ids2[mulhi(rng.Q[1], length(ids2)) + 1]
    return nothing
end

end # module PMCHip
