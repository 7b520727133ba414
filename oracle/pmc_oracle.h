/*
 * pmc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, IEEE double, no fast-math, no FP contraction) of the
 * ParticlesMC hot path: LinkedList neighbour cells -> per-particle pair energy ->
 * Displacement / DiscreteSwap dE -> Metropolis accept.  Every function cites the
 * reference file:line it follows (paths relative to the reference tree).
 *
 * Nothing under particlesmc_amd/ (the product) may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * Parity pin: the reference's own known answer (test/runtests.jl:36-38,
 * energy/N = -2.676832 +- 1e-6 for test/config_0.* under JBB, EmptyList and
 * LinkedList) is reproduced by tests/test_oracle_golden.py.  The Monte-Carlo
 * driver (Arianna.jl) and Julia's RNG are NOT in the reference tree, so the
 * accept sequence is pinned only against this file's own restatement of the
 * Arianna call contract ("parity unpinned" for that layer).
 */
#ifndef PMC_ORACLE_H
#define PMC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- models (src/models.jl) ---------------------------------------------- */
enum { ORC_LJ = 0, ORC_SOFT_INT = 1, ORC_SOFT_FLOAT = 2, ORC_SMOOTH_LJ = 3 };

typedef struct {
    int32_t kind;
    int32_t n;           /* SoftSpheres exponent (models.jl:56) */
    double eps, eps4, sigma, sigma2, rcut, rcut2, shift;
    int64_t ndiv2_i;     /* n / 2 when n even  (models.jl:66) */
    double ndiv2_f;      /* n / 2.0 when n odd (models.jl:66) */
    double C0, C2_s2, C4_s4; /* SmoothLennardJones (models.jl:150-157) */
} orc_model;

double orc_lennard_jones(double r2, double eps4, double sigma2);
double orc_pow_int(double x, int64_t n);
void orc_model_lj(orc_model *m, double eps, double sigma, double rcut, int shift_potential);
void orc_model_soft(orc_model *m, double eps, double sigma, int n, double rcut);
void orc_model_smooth_lj(orc_model *m, double eps, double sigma, double rcut);
double orc_potential(double r2, const orc_model *m);

/* ---- Julia Base semantics used by the path -------------------------------- */
double orc_jl_fld(double x, double y);
int64_t orc_jl_fld_i(int64_t a, int64_t b);

/* ---- system (src/atoms.jl, src/neighbours.jl, src/utils.jl) --------------- */
enum { ORC_EMPTYLIST = 0, ORC_LINKEDLIST = 1, ORC_CELLLIST = 2, ORC_VERLETLIST = 3 };
typedef struct orc_system orc_system;

/* System(...) atoms.jl:40-58.  species are 1-based (reference convention).
 * Returns NULL and sets *err: 1 = non-finite initial energy (atoms.jl:53-55),
 * 2 = bad arguments / degenerate cell grid. */
orc_system *orc_system_create(int64_t N, int32_t d, const double *pos, const int32_t *species,
                              double density, double temperature, int32_t nspecies,
                              const orc_model *models, int32_t list_type, int32_t *err);
/* the same with the list parameter of VerletList (list_parameters["dr"], neighbours.jl:401-404) */
orc_system *orc_system_create_list(int64_t N, int32_t d, const double *pos, const int32_t *species,
                                   double density, double temperature, int32_t nspecies,
                                   const orc_model *models, int32_t list_type, double dr, int32_t *err);
int64_t orc_system_cell_members(const orc_system *s, int64_t c, int64_t *out, int64_t cap); /* CellList/VerletList cells[c] */
void orc_system_destroy(orc_system *s);
double orc_system_energy(const orc_system *s);            /* system.energy[1] */
void orc_system_set_energy(orc_system *s, double e);
void orc_system_box(const orc_system *s, double *box);
void orc_system_cell_geometry(const orc_system *s, double *cell, int64_t *ncells);
void orc_system_get_cells(const orc_system *s, int64_t *cs);                 /* 1-based */
void orc_system_get_linked_list(const orc_system *s, int64_t *head, int64_t *list);
int64_t orc_system_neighbour_cells(const orc_system *s, int64_t c, int64_t *out); /* c 1-based */
void orc_system_get_positions(const orc_system *s, double *pos);
void orc_system_get_species(const orc_system *s, int32_t *species);
void orc_system_set_temperature(orc_system *s, double T);
double orc_energy_particle(const orc_system *s, int64_t i);   /* atoms.jl:81-88; i 1-based */
void orc_energy_particles(const orc_system *s, double *e);    /* all i */
double orc_total_energy(const orc_system *s);                 /* atoms.jl:51-52 */
int64_t orc_neighbours_of(const orc_system *s, int64_t i, int64_t *out, int64_t cap);

/* ---- moves (src/moves.jl) ------------------------------------------------- */
void orc_displacement_perform(orc_system *s, int64_t i, const double *delta,
                              double *e1, double *e2, double *de);
void orc_displacement_revert(orc_system *s, int64_t i, const double *minus_delta, double de);
void orc_swap_perform(orc_system *s, int64_t i, int64_t j, int32_t sp1, int32_t sp2,
                      double *e1, double *e2, double *de);
void orc_swap_revert(orc_system *s, int64_t i, int64_t j, int32_t sp1, int32_t sp2, double de);
int64_t orc_species_count(const orc_system *s, int32_t sp);
void orc_species_ids(const orc_system *s, int32_t sp, int64_t *ids);
void orc_species_heads(const orc_system *s, int64_t *heads);

/* ---- RNG + deterministic math (the schedule's proposal generator) --------- */
void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);
double orc_det_log(double x);
double orc_det_exp(double x);
int orc_energy_bias(const orc_system *s, int32_t sp1, int32_t sp2, double theta1, double theta2, double u1, double u2,
                    int64_t gi, int64_t gj, int64_t *i_out, int64_t *j_out, double *logq);   /* moves.jl:246-280 */
void orc_det_sincos2pi(uint32_t w, double *s, double *c);
void orc_box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1);
double orc_uniform53(uint32_t hi, uint32_t lo);

enum { ORC_MOVE_DISPLACEMENT = 0, ORC_MOVE_SWAP = 1 };
typedef struct {
    int32_t kind;
    double probability;
    double sigma;            /* Displacement / SimpleGaussian */
    int32_t species[2];      /* DiscreteSwap / DoubleUniform */
} orc_move;

typedef struct {
    int64_t attempted[8];
    int64_t accepted[8];
} orc_stats;

/* Sequential Metropolis: restates the Arianna mc_step! call contract
 * (sample_action! -> perform_action! -> delta_log_target_density ->
 * accept or revert_action!) with the Philox stream defined in DESIGN.md.
 * step0 = index of the first step (global step counter). */
void orc_mc_steps_sequential(orc_system *s, const orc_move *pool, int32_t npool,
                             uint64_t seed, uint32_t chain, uint64_t step0, int64_t nsteps,
                             orc_stats *stats);

/* ---- parallel (checkerboard) schedule restatement -------------------------- */
typedef struct orc_cb orc_cb;
orc_cb *orc_cb_create(int64_t N, int32_t d, const double *pos, const int32_t *species,
                      const double *box, double temperature, int32_t nspecies,
                      const orc_model *models, uint64_t seed, uint32_t chain, int32_t *err);
void orc_cb_destroy(orc_cb *c);
void orc_cb_grid(const orc_cb *c, int64_t *n_mc, double *w);
void orc_cb_sweeps(orc_cb *c, const orc_move *pool, int32_t npool, uint64_t sweep0,
                   int64_t nsweeps, orc_stats *stats);
void orc_cb_sweeps_rounds(orc_cb *c, const orc_move *pool, int32_t npool, uint64_t sweep0,
                          int64_t nsweeps, int32_t rounds, orc_stats *stats);   /* rounds per cell visit (m) */
void orc_cb_get_positions(const orc_cb *c, double *pos);
void orc_cb_get_species(const orc_cb *c, int32_t *species);
int64_t orc_cb_energy_fix(const orc_cb *c);         /* running energy, units of 2^-32 */
int64_t orc_cb_total_energy_fix(const orc_cb *c);   /* recomputed, units of 2^-32 */
void orc_cb_set_temperature(orc_cb *c, double T);
/* quantised per-particle energy against an arbitrary position (for delta_e parity) */
double orc_cb_energy_at(const orc_cb *c, int64_t i, const double *x, int32_t sp);

#ifdef __cplusplus
}
#endif
#endif
