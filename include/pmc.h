/*
 * pmc.h -- C ABI of libpmc_hip.so: the MI355X (gfx950) implementation of the
 * ParticlesMC pair-energy + Metropolis hot path.
 *
 * The reference (Julia) has no FFI; these are the entry points a `ccall` shim
 * for this path binds (INTEGRATION.md shows the Julia side).  Each function
 * cites the reference interface it replaces (paths relative to the reference
 * tree).  Conventions:
 *   - particle ids and species are 1-based at the ABI (reference convention);
 *   - positions are AoS N*d doubles, never wrapped (src/moves.jl:46-48);
 *   - the caller owns every host buffer; the library owns device memory behind
 *     the handle and retains no host pointer past a call;
 *   - every function returns 0 or a negative pmc_status; nothing throws or
 *     exits; pmc_last_error() gives the text;
 *   - one handle = one chain on one HIP device with one stream; a handle is not
 *     thread-safe, distinct handles are (Arianna's chain-level parallelism).
 */
#ifndef PMC_H
#define PMC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PMC_OK = 0,
    PMC_ERR_BADARG = -1,
    PMC_ERR_HIP = -2,
    PMC_ERR_NONFINITE = -3,   /* "Initial configuration has infinite or NaN energy." src/atoms.jl:53-55 */
    PMC_ERR_UNSUPPORTED = -4,
    PMC_ERR_STATE = -5
} pmc_status;

/* src/models.jl: LennardJones 99-123, SoftSpheres 52-74, SmoothLennardJones 137-167 */
typedef enum {
    PMC_MODEL_LJ = 0,
    PMC_MODEL_SOFT_INT = 1,    /* even n: (sigma2/r2)^(n/2) with an integer power */
    PMC_MODEL_SOFT_FLOAT = 2,  /* odd n: ndiv2_f = n/2 = k + 1/2, evaluated as sqrt((sigma2/r2)^n) -- see DESIGN.md section 3 */
    PMC_MODEL_SMOOTH_LJ = 3
} pmc_model_kind;

/* One entry of system.model_matrix (src/atoms.jl:25); all derived constants are
 * computed by the HOST exactly as the reference constructors do. */
typedef struct {
    int32_t kind;
    int32_t n;
    double eps, eps4, sigma, sigma2, rcut, rcut2, shift;
    int64_t ndiv2_i;
    double ndiv2_f;
    double C0, C2_s2, C4_s4;
} pmc_pair_model;

/* Atoms + LinkedList geometry (src/atoms.jl:18-30,45; src/neighbours.jl:236-238).
 * box, cell and ncells are computed by the host so that cell membership is
 * bit-exact with the reference. */
typedef struct {
    int64_t N;
    int32_t d;                 /* 2 or 3 */
    int32_t nspecies;
    double box[3];
    double cell[3];
    int64_t ncells[3];
    double temperature;
    double density;
    const pmc_pair_model *models;   /* nspecies*nspecies, row-major [s_i-1][s_j-1] */
    int32_t device;
    uint32_t chain;
    uint64_t seed;
} pmc_params;

typedef enum { PMC_MOVE_DISPLACEMENT = 0, PMC_MOVE_SWAP = 1 } pmc_move_kind;

/* Arianna Move(action, policy, parameters, probability): Displacement +
 * SimpleGaussian (src/moves.jl:34-38,105-123), DiscreteSwap + DoubleUniform
 * (src/moves.jl:137-149,226-241). */
typedef struct {
    int32_t kind;
    int32_t species[2];
    double probability;
    double sigma;
} pmc_move;

#define PMC_MAX_MOVES 8

typedef enum {
    PMC_SEQUENTIAL = 0,    /* the reference's one-particle-per-step chain (src/moves.jl:57-81) */
    PMC_CHECKERBOARD = 1   /* many independent proposals per sweep (DESIGN.md "schedule"): an even grid of >= 4 cells of width >= rcut per
                              side, 2^d colour passes; or -- a box of 2 ... 4 cutoffs per side in every dimension with N <= 1024 -- ONE cell
                              spanning the periodic box: the plain Metropolis chain, its N proposal slots evaluated eight at a time.
                              Other boxes: PMC_ERR_UNSUPPORTED (PMC_SEQUENTIAL runs them) */
} pmc_mode;

typedef struct {
    int64_t attempted[PMC_MAX_MOVES];
    int64_t accepted[PMC_MAX_MOVES];
    double energy;        /* system.energy[1] after the call */
    int64_t energy_fix;   /* same, in units of 2^-32 (checkerboard mode; exact) */
    double kernel_ms;     /* HIP-event time of the sweep kernels on the handle's stream */
    double sweep_kernel_ms; /* of which: the dE/Metropolis kernel alone */
    int64_t sweep_kernel_launches; /* its launches in the timed sweeps: one per sweep (all colours fused), 2^d with PMC_NO_FUSED=1 */
    /* work counters of the throughput engine, counted by the kernel (0 in PMC_SEQUENTIAL mode): */
    int64_t pair_tests;       /* (slot, candidate, old|new position) distance tests: a WORK counter -- it depends on the LDS staging
                                 capacity a launch was sized with (learnt from earlier sweeps), the results do not */
    int64_t pairs_in_range;   /* pair energies evaluated (r2 <= rcut2, src/atoms.jl:72) */
    int64_t null_moves;       /* attempted slots that evaluated no dE: Displacements that would leave their cell, swaps in a cell lacking a species */
    int64_t dense_evals;      /* FP64 distance evaluations: (survivor of the FP32 pre-filter, old | new position); counted by the
                                 single-species kernels with one wavefront per cell (the large-system path) and by the
                                 big-neighbourhood team kernels (boxes of 4-5 cutoffs per side), 0 otherwise */
    int64_t staged_candidates;/* candidates the cell visits loaded from global memory (all 3^d cells of every visit) */
    int64_t written_back;     /* particles the cell visits wrote back (every particle of a cell with an accepted move) */
} pmc_stats;

typedef struct pmc_system pmc_system;

/* -- lifetime ------------------------------------------------------------- */
int pmc_create(const pmc_params *p, pmc_system **out);          /* Atoms(...) src/atoms.jl:40-49 */
int pmc_destroy(pmc_system *h);
const char *pmc_last_error(const pmc_system *h);                /* h may be NULL */
int pmc_device_count(int *n);

/* -- host-side constructors (pure host arithmetic; usable without a GPU) ----
 * A Julia caller already owns these numbers (it passes its own structs); hosts
 * without the reference's constructors use these restatements. */
int pmc_model_lennard_jones(double eps, double sigma, double rcut, int32_t shift_potential,
                            pmc_pair_model *out);                 /* LennardJones(...) src/models.jl:110-119 */
int pmc_model_soft_spheres(double eps, double sigma, int32_t n, double rcut,
                           pmc_pair_model *out);                  /* SoftSpheres(...) src/models.jl:64-70 */
int pmc_model_smooth_lennard_jones(double eps, double sigma, double rcut,
                                   pmc_pair_model *out);          /* SmoothLennardJones(...) src/models.jl:150-160 */
int pmc_potential(double r2, const pmc_pair_model *m, double *u); /* potential(r2, model) src/models.jl:72,121,162 */
int pmc_box_from_density(int64_t N, double density, int32_t d, double *box);   /* src/atoms.jl:45 */
int pmc_linked_list_geometry(int32_t d, const double *box, double rcut,
                             double *cell, int64_t *ncells);      /* LinkedList(box, rcut, N) src/neighbours.jl:237-238 */
int pmc_cell_list_geometry(int32_t d, const double *box, double rcut,
                           double *cell, int64_t *ncells);        /* CellList(box, rcut, N) src/neighbours.jl:126-133 */
int pmc_verlet_list_geometry(int32_t d, const double *box, double rcut, double dr,
                             double *cell, int64_t *ncells);      /* VerletList(box, rcut, N; dr) src/neighbours.jl:405-406 */

/* -- state ---------------------------------------------------------------- */
int pmc_upload(pmc_system *h, const double *pos, const int32_t *species);   /* system.position / .species */
int pmc_download(pmc_system *h, double *pos, int32_t *species);
/* Checkpoint and restart of a PMC_CHECKERBOARD chain.  Positions are never wrapped (src/moves.jl:46-48); the throughput
 * engine therefore carries a persistent integer periodic image per particle and dimension, derived as floor(x / L) whenever it
 * takes a state over (after pmc_upload, per-move verbs, PMC_SEQUENTIAL sweeps) and frozen while it runs (DESIGN.md 2.2
 * "images").  A plain download + upload re-derives them, which changes the rounding of later moves of particles that have
 * left the box: to continue a chain bit for bit store pmc_get_images with the positions (and pmc_get_counters), and after
 * pmc_upload call pmc_set_images (and pmc_set_counters, pmc_set_energy).  img: N*d entries, |k| <= 511. */
int pmc_get_images(pmc_system *h, int32_t *img);
int pmc_set_images(pmc_system *h, const int32_t *img);
int pmc_set_temperature(pmc_system *h, double T);               /* system.temperature */
int pmc_get_temperature(pmc_system *h, double *T);
int pmc_set_energy(pmc_system *h, double e);                    /* system.energy[1]; not taken (and 0 returned) after a throughput
                                                                   sweep that failed part-way: the positions have moved, the
                                                                   library recomputes the energy before its next use */
int pmc_get_energy(pmc_system *h, double *e);
int pmc_set_counters(pmc_system *h, uint64_t step, uint64_t sweep);  /* RNG stream position */
int pmc_get_counters(pmc_system *h, uint64_t *step, uint64_t *sweep);

/* -- the reference's other list types (reference-order engine only; the throughput engine's results do not
 *    depend on the list).  Call after pmc_create (whose cell / ncells then hold THAT list's geometry) and before
 *    the first build.  list_type: 0 EmptyList (src/neighbours.jl:13-44), 1 LinkedList (default, :224-377),
 *    2 CellList (:61-217), 3 VerletList with list_parameters["dr"] (:387-543).  Per-particle energies are summed
 *    in the order that list's iterator yields; moves maintain it exactly as update_neighbour_list! does.  After
 *    throughput sweeps the list is rebuilt from the current state (build_neighbour_list!). */
int pmc_set_list_type(pmc_system *h, int32_t list_type, double dr);
int pmc_get_cell_members(pmc_system *h, int64_t c, int64_t *out, int64_t cap, int64_t *n);      /* CellList/VerletList .cells[c] */
int pmc_get_verlet_neighbours(pmc_system *h, int64_t i, int64_t *out, int64_t cap, int64_t *n); /* VerletList .neighbours[i] */

/* -- LinkedList (reference-order engine) ---------------------------------- */
int pmc_build_cells(pmc_system *h);                             /* build_neighbour_list! src/neighbours.jl:251-270 */
int pmc_get_cells(pmc_system *h, int64_t *cs);                  /* neighbour_list.cs, 1-based */
int pmc_get_linked_list(pmc_system *h, int64_t *head, int64_t *list); /* .head (-1 = empty), .list */
int pmc_get_neighbour_cells(pmc_system *h, int64_t c, int64_t *out, int32_t *n); /* .neighbour_cells[c] src/neighbours.jl:94-111 */
int pmc_neighbours_of(pmc_system *h, int64_t i, int64_t *out, int64_t cap, int64_t *n);  /* collect(L(system, i)): (L::LinkedList)(system, i) + LinkedIterator, src/neighbours.jl:316-377 (includes i; n = needed size) */
int pmc_energy_particle(pmc_system *h, int64_t i, double *e);   /* compute_energy_particle src/atoms.jl:81-88 */
int pmc_energy_particles(pmc_system *h, double *e);             /* [compute_energy_particle(i) for i] src/atoms.jl:51 */
int pmc_init_energy(pmc_system *h, double *e);                  /* build list + sum(local)/2 + Inf/NaN guard, src/atoms.jl:50-56 */

int pmc_species_count(pmc_system *h, int32_t sp, int64_t *n);    /* length(species_list.sp_ids[sp]) src/utils.jl:31-34 */
int pmc_get_species_ids(pmc_system *h, int32_t sp, int64_t *ids);/* species_list.sp_ids[sp], current order */
/* Hand-over rule: the sequential chain and the swap verbs maintain sp_ids exactly as src/moves.jl:170-185 does
 * (swap-and-overwrite). PMC_CHECKERBOARD sweeps swap cell-locally and do not; after them the next consumer sees
 * SpeciesList(species) of the current state (src/utils.jl:36-49: ids ascending per species), just as the linked
 * list it sees is build_neighbour_list! of the current positions. */

/* -- per-move verbs (batch of one; reference semantics incl. system.energy) */
int pmc_displacement_perform(pmc_system *h, int64_t i, const double *delta,
                             double *e1, double *e2, double *de);   /* src/moves.jl:11-20,57-67 */
int pmc_displacement_revert(pmc_system *h, int64_t i, const double *minus_delta, double de); /* src/moves.jl:76-81 */
int pmc_swap_perform(pmc_system *h, int64_t i, int64_t j,
                     double *e1, double *e2, double *de);           /* src/moves.jl:11-20,187-194 */
int pmc_swap_revert(pmc_system *h, int64_t i, int64_t j, double de);/* src/moves.jl:201-207 */

/* -- EnergyBias policy on the device (src/moves.jl:246-280) ----------------- */
/* sample_action!(::DiscreteSwap, ::EnergyBias, ...): i from species sp1 with probability exp(theta1 e_i) / sum, j from sp2
 * with exp(theta2 e_j) / sum, over the DEVICE-resident per-particle energies (reference-order compute_energy_particle of
 * every particle; nothing of size N crosses the bus).  u1, u2: the caller's uniforms in [0, 1) for the two
 * rand(rng, Categorical(w)) picks (inverse CDF in sp_ids order).  Also returns log q(i, j) (moves.jl:254-263). */
int pmc_energy_bias_sample(pmc_system *h, int32_t sp1, int32_t sp2, double theta1, double theta2, double u1, double u2,
                           int64_t *i, int64_t *j, double *logq);
/* log_proposal_density(::DiscreteSwap, ::EnergyBias, ...) of a given pair in the CURRENT state (called again after
 * invert_action! for the backward move). */
int pmc_energy_bias_logq(pmc_system *h, int32_t sp1, int32_t sp2, double theta1, double theta2, int64_t i, int64_t j,
                         double *logq);
int pmc_exp(double x, double *y);                                  /* the fixed-operation-sequence exp of the weights */

/* -- throughput engine ----------------------------------------------------- */
int pmc_total_energy(pmc_system *h, double *E, int64_t *E_fix);  /* sum(local)/2 src/atoms.jl:51-52, exact fixed-point */
int pmc_energy_particles_exact(pmc_system *h, double *e);         /* all compute_energy_particle(i) at once (src/ParticlesMC.jl:104-113), exact sums */
int pmc_delta_e(pmc_system *h, int64_t n, const int64_t *i, const double *delta,
                double *e1, double *e2);                          /* independent batch of Displacement dE */
int pmc_grid(pmc_system *h, int64_t *n_mc, double *w);            /* checkerboard grid geometry */

/* Schedule parameter m of the throughput engine ("rounds per cell visit"; no reference counterpart: the
 * reference chain is sequential).  The sweeps of a call are taken in passes of min(m, sweeps left) sweeps: a pass draws ONE
 * grid shift, bins once, and every cell visit runs that many rounds of n_c proposal slots against the neighbourhood it staged
 * once.  m = 1 (default): the grid is re-shifted every N attempted moves.  1 <= m <= PMC_MAX_ROUNDS.  A pass never spans
 * calls, so the chain depends on how sweeps are grouped into calls when m > 1 (oracle: orc_cb_sweeps_rounds). */
#define PMC_MAX_ROUNDS 16
int pmc_set_rounds(pmc_system *h, int32_t rounds);
int pmc_get_rounds(pmc_system *h, int32_t *rounds);

/* mc_sweep!(system, pool, rng; mc_steps = nsweeps * N) -- the Arianna entry
 * the reference reaches from run!(Simulation) (src/ParticlesMC.jl:246,294-297;
 * benchmark/particles_benchmarks.jl:29). */
int pmc_sweep(pmc_system *h, int32_t mode, const pmc_move *pool, int32_t npool,
              int64_t nsweeps, pmc_stats *out);

/* How PMC_SEQUENTIAL sweeps run (the results are the same chain bit for bit: positions, head / list / cs, accept counts,
 * system.energy[1]).  engine 1: one wavefront takes the steps of src/moves.jl:57-81 literally, one after the other.  engine 2:
 * speculative windows -- the random part of a step is a function of the step counter, a step only reads and writes the
 * neighbour cells of its particle (src/neighbours.jl:94-111,316-377), so a window of pending steps is evaluated at once and
 * every step that no earlier pending step can have influenced commits.  engine 3: the whole state of a small
 * system (positions, species, cell rows, SpeciesList: up to 160 KB) lives in LDS and one wavefront walks the chain with no
 * global-memory traffic; pmc_sweep_many runs any number of such chains of one shape in one launch (the reference's own
 * workloads: benchmark/particles_benchmarks.jl:10-28, N = 216; test/ss142d.jl:8-21, 100 chains of N = 8).  engine 0 (default):
 * 3 for LinkedList / CellList systems of at most 7 cells per side whose state fits 64 KB of LDS and for EmptyList systems of up
 * to ~17 000 particles (beyond 4096 in 3-D their positions stay in global memory, everything else in LDS), else 2 for
 * LinkedList / CellList; 4 for VerletList systems (src/neighbours.jl:387-543; the reference's CI system, test/runtests.jl:41-91)
 * whose positions, positions_at_last_update, cell vectors and neighbour lists fit the LDS: one workgroup per chain, the next
 * twelve steps behind the one being resolved evaluated by other wavefronts; 5 for EmptyList systems of a few hundred particles
 * under pools of Displacements only (the reference's published benchmark, benchmark/particles_benchmarks.jl:10-28): as 4 with six or twelve evaluators, and an
 * evaluation that earlier steps have overtaken is repaired -- the pair terms of the particles stored in between replaced in its
 * packed list, the ordered sums run again -- instead of discarded; 1 for larger VerletList systems, for still larger EmptyList
 * systems and for swap moves between a species and itself.  window: pending steps per round of engine 2
 * (0: chosen from the cell count; at most 4096); with engine 5 the wavefronts per chain, 8 or 16 (0: sixteen while every chain
 * of the call has a CU to itself, eight -- two chains to a CU -- beyond). */
int pmc_set_sequential_engine(pmc_system *h, int32_t engine, int32_t window);
/* diagnostics of the last PMC_SEQUENTIAL call that ran engine 2: out[0] rounds, out[1] step evaluations, out[2] window,
 * out[3] steps of that call folded into system.energy (= nsweeps * N when it ended well) */
int pmc_sequential_info(pmc_system *h, int64_t *out);
/* diagnostics of the last PMC_SEQUENTIAL call that ran engine 4 (VerletList state in LDS): out[0] batches, out[1] steps that
 * wavefront 0 ran alone on the current state (list rebuilds of update_neighbour_list!, src/neighbours.jl:494-543, swaps, steps
 * whose inputs an earlier step of the batch had touched), out[2] list rebuilds, out[3] steps of the call the engine ran (the
 * rest, if a list outgrew its LDS row, ran on engine 1).  After a call that ran engine 5: out[0] polls of the resolver for a record that
 * was not ready, out[1] steps evaluated again from scratch (their own particle had been stored in between), out[2] steps whose
 * evaluation was repaired, out[3] steps */
int pmc_verlet_info(pmc_system *h, int64_t *out);

/* Fault injection for the library's own tests (no reference counterpart; never needed by a host).  what =
 * PMC_INJECT_STARVED_LAUNCH: the next fused checkerboard launch of this handle is reported as starved although it ran to the
 * end -- the path a lost completion flag takes: PMC_ERR_STATE, the chain invalid, the handle switched to the single
 * dispatch-order-independent ticket counter.  One shot; a handle property, so nothing in the environment can trigger it. */
#define PMC_INJECT_STARVED_LAUNCH 1
int pmc_debug_inject(pmc_system *h, int32_t what);

/* The same sweep split in two so that several handles (chains with parallel=true,
 * src/ParticlesMC.jl:164,246; replicas of a tempering ladder) run concurrently on one GPU: begin
 * enqueues on the handle's stream and returns, end waits and collects.  No other call on the
 * handle between the two. */
int pmc_sweep_begin(pmc_system *h, int32_t mode, const pmc_move *pool, int32_t npool, int64_t nsweeps);
int pmc_sweep_end(pmc_system *h, pmc_stats *out);

/* All the chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) or all the replicas of a tempering ladder in
 * shared launches: when the n handles have one shape (same device, N, box, models, grid; N < 5e5, <= 32768 cells) every kernel of a sweep
 * is launched ONCE for all of them -- the binning kernels take the replica from the second grid dimension, the fused
 * sweep kernel from its ticket -- instead of n times on n streams, and the launch takes as many wavefronts per cell as the
 * cells of ALL its chains call for (many small chains fill the chip like one large system).  The result of every chain is
 * bit-identical to pmc_sweep on it alone.  Other cases (different shapes, PMC_SEQUENTIAL, large systems) run as pmc_sweep_begin on every
 * handle followed by pmc_sweep_end on every handle.  out: n entries or NULL. */
int pmc_sweep_many(pmc_system **h, int32_t n, int32_t mode, const pmc_move *pool, int32_t npool, int64_t nsweeps,
                   pmc_stats *out);

/* -- parallel tempering (no reference counterpart: README.md:33 "soon") ----- */
/* Deterministic swap decisions for one exchange round over n replicas:
 * label[r] = index of the temperature replica r currently holds; pairs of
 * adjacent temperatures (2k + parity, 2k + parity + 1); accept with
 * min(1, exp((beta_a - beta_b) (E_a - E_b))).  Every rank calls this with the
 * same all-gathered energies and gets the same answer. */
int pmc_pt_decide(uint64_t seed, uint64_t round, int32_t n, const double *energy,
                  const double *temperature_ladder, int32_t *label, int32_t *n_accepted);

/* One exchange round over n replicas that are handles of THIS process (one GPU, or one handle per device of the
 * node): reads every handle's running energy, takes the pmc_pt_decide decisions and sets every handle's temperature to
 * ladder[label[r]].  label (in/out, n entries) is the permutation "replica r holds temperature label[r]". */
int pmc_replica_exchange(pmc_system **h, int32_t n, const double *temperature_ladder, uint64_t seed, uint64_t round,
                         int32_t *label, int32_t *n_accepted);

/* The same round for one-process-per-GPU hosts: the only collective of the path, ncclAllGather of ONE double per
 * rank over RCCL / xGMI, on the handle's stream.  `comm` is an ncclComm_t (from the host's own RCCL binding, or from
 * pmc_rccl_comm_create); librccl.so is opened on first use.  label has `world` entries and must be identical on
 * every rank; every rank returns the same label and n_accepted. */
int pmc_replica_exchange_rccl(pmc_system *h, void *nccl_comm, int32_t rank, int32_t world, const double *temperature_ladder,
                              uint64_t seed, uint64_t round, int32_t *label, int32_t *n_accepted);
int pmc_rccl_unique_id(void *id128);                                   /* ncclGetUniqueId: rank 0 calls it and ships the 128 bytes to the others */
int pmc_rccl_comm_create(const void *id128, int32_t rank, int32_t world, int32_t device, void **nccl_comm);  /* ncclCommInitRank */
int pmc_rccl_comm_destroy(void *nccl_comm);

/* -- the schedule's random streams, host side (DESIGN.md "random streams") -- */
int pmc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
int pmc_box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1);
int pmc_log_uniform(uint32_t hi, uint32_t lo, double *lnu);       /* log of the 53-bit uniform in (0,1) */
/* words of sequential step `step`: R = block TAG_SEQ0, Q = block TAG_SEQ1 */
int pmc_sequential_words(uint64_t seed, uint32_t chain, uint64_t step, uint32_t R[4], uint32_t Q[4]);

/* -- debug / test hooks ----------------------------------------------------- */
int pmc_debug_rng(pmc_system *h, int64_t n, const uint32_t *w1, const uint32_t *w2,
                  double *z0, double *z1, double *lnu);           /* device Box-Muller / log, for bit-parity tests */
int pmc_debug_div(pmc_system *h, int64_t n, const double *num, const double *den,
                  double *q_ieee, double *q_fast);                 /* guard-free device division vs IEEE, for bit-parity tests */
const char *pmc_version(void);

#ifdef __cplusplus
}
#endif
#endif
