/*
 * pmc_oracle.c -- TEST INFRASTRUCTURE ONLY (see pmc_oracle.h).
 *
 * Plain-C restatement of the ParticlesMC LinkedList + pair-energy + Metropolis
 * path.  Compile with -O2 -ffp-contract=off and WITHOUT -ffast-math: every
 * floating-point operation below is meant to be one IEEE-754 double operation,
 * in the order the Julia reference performs it.
 *
 * Julia Base semantics that are not in the reference tree (float fld, integer
 * fld, round-half-even, Float64^Int) are restated from the published Base
 * definitions and flagged "Base:" below; they could not be re-verified against a
 * Julia binary in the build container.
 */
#include "pmc_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================== */
/* Julia Base semantics                                                      */
/* ======================================================================== */

/* Base: mod(x::Float64, y::Float64) -- result has the sign of y. */
static double jl_mod(double x, double y)
{
    double r = fmod(x, y);
    if (r == 0.0)
        return copysign(r, y);
    if ((r > 0.0) != (y > 0.0))
        return r + y;
    return r;
}

/* Base: fld(x, y) = div(x, y, RoundDown) = round((x - mod(x, y)) / y)
 * (base/div.jl), round = ties-to-even.  Used by get_cell, neighbours.jl:178-180,
 * and by the LinkedList constructor, neighbours.jl:237. */
double orc_jl_fld(double x, double y)
{
    return rint((x - jl_mod(x, y)) / y);
}

/* Base: fld(a::Int, b::Int) floors toward -inf.  utils.jl:12 on cell tuples. */
int64_t orc_jl_fld_i(int64_t a, int64_t b)
{
    int64_t q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0)))
        q -= 1;
    return q;
}

/* Base: x::Float64 ^ n::Int -> pow_body (base/math.jl), compensated power by
 * squaring.  models.jl:28 `(sigma2 / r2)^(ndiv2)` with Int ndiv2 lands here.
 * Restated from memory of Julia 1.9-1.11; any faithful pow agrees to <= 1 ulp. */
double orc_pow_int(double x, int64_t n)
{
    double y = 1.0, xnlo = 0.0, ynlo = 0.0, err, hi;
    if (n == 3)
        return x * x * x;
    if (n < 0) {
        double rx = 1.0 / x;
        if (n == -2)
            return rx * rx;
        if (isfinite(x))
            xnlo = -fma(x, rx, -1.0) * rx;
        x = rx;
        n = -n;
    }
    while (n > 1) {
        if (n & 1) {
            err = fma(y, xnlo, x * ynlo);
            hi = x * y;
            ynlo = fma(x, y, -hi);
            y = hi;
            ynlo += err;
        }
        err = x * 2.0 * xnlo;
        hi = x * x;
        xnlo = fma(x, x, -hi);
        x = hi;
        xnlo += err;
        n >>= 1;
    }
    err = fma(y, xnlo, x * ynlo);
    if (isfinite(x) && isfinite(err))
        return fma(x, y, err);
    return x * y;
}

/* ======================================================================== */
/* models.jl                                                                 */
/* ======================================================================== */

/* models.jl:30-34 */
double orc_lennard_jones(double r2, double eps4, double sigma2)
{
    double x = sigma2 * (1.0 / r2);
    double x3 = x * x * x;
    return eps4 * (x3 * x3 - x3);
}

/* models.jl:28 inverse_power(r2, eps, sigma2, ndiv2) */
static double inverse_power(double r2, double eps, double sigma2, const orc_model *m)
{
    double q = sigma2 / r2;
    if (m->kind == ORC_SOFT_INT)
        return eps * orc_pow_int(q, m->ndiv2_i);
    /* Float64^Float64 is Julia's own table-driven pow (< 1 ulp), which cannot be restated here without its
     * tables; n odd => ndiv2 = k + 1/2, evaluated as sqrt(q^n) with the compensated integer power (<= 0.75 ulp
     * from the true value). PARITY UNPINNED at the ulp level for this branch. */
    return eps * sqrt(orc_pow_int(q, (int64_t)(2.0 * m->ndiv2_f)));
}

/* models.jl:110-119 */
void orc_model_lj(orc_model *m, double eps, double sigma, double rcut, int shift_potential)
{
    memset(m, 0, sizeof(*m));
    m->kind = ORC_LJ;
    m->eps = eps;
    m->eps4 = 4.0 * eps;
    m->sigma = sigma;
    m->sigma2 = sigma * sigma;
    m->rcut = rcut;
    m->rcut2 = rcut * rcut;
    m->shift = shift_potential ? orc_lennard_jones(m->rcut2, m->eps4, m->sigma2) : 0.0;
}

/* models.jl:64-70 */
void orc_model_soft(orc_model *m, double eps, double sigma, int n, double rcut)
{
    memset(m, 0, sizeof(*m));
    m->n = n;
    m->eps = eps;
    m->sigma = sigma;
    m->sigma2 = sigma * sigma;
    m->rcut = rcut;
    m->rcut2 = rcut * rcut;
    if (n & 1) {
        m->kind = ORC_SOFT_FLOAT;
        m->ndiv2_f = (double)n / 2.0;
    } else {
        m->kind = ORC_SOFT_INT;
        m->ndiv2_i = n / 2;
    }
    m->shift = inverse_power(m->rcut2, eps, m->sigma2, m);
}

/* models.jl:150-160 */
void orc_model_smooth_lj(orc_model *m, double eps, double sigma, double rcut)
{
    const double C0 = 0.04049023795, C2 = -0.00970155098, C4 = 0.00062012616;
    memset(m, 0, sizeof(*m));
    m->kind = ORC_SMOOTH_LJ;
    m->eps = eps;
    m->eps4 = 4.0 * eps;
    m->sigma = sigma;
    m->sigma2 = sigma * sigma;
    m->C0 = C0;
    m->C2_s2 = C2 / m->sigma2;
    m->C4_s4 = C4 / (m->sigma2 * m->sigma2);
    m->rcut = rcut;
    m->rcut2 = rcut * rcut;
}

/* potential(r2, model): models.jl:72-74 (SoftSpheres), 121-123 (LennardJones),
 * 162-167 (SmoothLennardJones; muladd -> fma) */
double orc_potential(double r2, const orc_model *m)
{
    switch (m->kind) {
    case ORC_LJ:
        return orc_lennard_jones(r2, m->eps4, m->sigma2) - m->shift;
    case ORC_SOFT_INT:
    case ORC_SOFT_FLOAT:
        return inverse_power(r2, m->eps, m->sigma2, m) - m->shift;
    default: {
        double lj = orc_lennard_jones(r2, m->eps4, m->sigma2);
        double sh = m->eps4 * (m->C0 + r2 * fma(r2, m->C4_s4, m->C2_s2));
        return lj + sh;
    }
    }
}

/* ======================================================================== */
/* utils.jl                                                                  */
/* ======================================================================== */

/* utils.jl:15-18 vector_1D */
static inline double vector_1d(double c1, double c2, double L)
{
    double dx = c1 - c2;
    return dx - rint(dx / L) * L;
}

/* utils.jl:24-28: sum(abs2, dx) over an SVector = left fold */
static inline double nearest_image_distance_squared(const double *xi, const double *xj,
                                                    const double *box, int d)
{
    double v = vector_1d(xi[0], xj[0], box[0]);
    double r2 = v * v;
    for (int a = 1; a < d; a++) {
        v = vector_1d(xi[a], xj[a], box[a]);
        r2 = r2 + v * v;
    }
    return r2;
}

/* ======================================================================== */
/* system: atoms.jl + neighbours.jl (LinkedList, EmptyList) + SpeciesList    */
/* ======================================================================== */

struct orc_system {
    int64_t N;
    int32_t d;
    double *pos;       /* AoS N*d, never wrapped (moves.jl:46-48) */
    int32_t *species;  /* 1-based */
    double density, temperature;
    double energy;     /* system.energy[1] */
    double box[3];
    int32_t nspecies;
    orc_model *models; /* [si-1][sj-1] row-major */
    int32_t list_type;
    /* LinkedList, neighbours.jl:224-232 */
    double cell[3];
    int64_t ncells[3];
    int64_t ncell_total;
    int64_t *head, *list, *cs;
    int64_t *nbr;      /* ncell_total * nbr_cap, 1-based scalar indices */
    int32_t *nbr_cnt;
    int32_t nbr_cap;
    /* SpeciesList, utils.jl:31-49 */
    int64_t **sp_ids;
    int64_t *sp_n;
    int64_t *sp_heads;
    /* CellList / VerletList, neighbours.jl:61-217, 387-543: per-cell vectors of ids */
    int64_t **cells;
    int64_t *cells_n, *cells_cap;
    /* VerletList: per-particle vectors of neighbour ids, positions at the last update */
    double maxcut, dr, dr2o4, rcutoff2;
    int64_t **vnb;
    int64_t *vnb_n, *vnb_cap;
    double *last;
};

static void vec_push(int64_t **v, int64_t *n, int64_t *cap, int64_t x)   /* push! */
{
    if (*n == *cap) {
        *cap = *cap ? 2 * *cap : 8;
        *v = (int64_t *)realloc(*v, sizeof(int64_t) * (size_t)*cap);
    }
    (*v)[(*n)++] = x;
}

static void vec_remove_all(int64_t *v, int64_t *n, int64_t x)   /* filter!(y -> y != x, v) == deleteat!(v, v .== x) */
{
    int64_t w = 0;
    for (int64_t r = 0; r < *n; r++)
        if (v[r] != x) v[w++] = v[r];
    *n = w;
}

/* cell_index, neighbours.jl:79-88 (fold_back utils.jl:12; last dim fastest) */
static int64_t cell_index(const int64_t *ncells, const int64_t *mc, int d)
{
    int64_t c = 1, stride = 1;
    for (int a = d - 1; a >= 0; a--) {
        int64_t f = mc[a] - orc_jl_fld_i(mc[a], ncells[a]) * ncells[a];
        c += f * stride;
        stride *= ncells[a];
    }
    return c;
}

/* get_cell, neighbours.jl:178-180 */
static void get_cell(const double *x, const double *cell, int d, int64_t *mc)
{
    for (int a = 0; a < d; a++)
        mc[a] = (int64_t)orc_jl_fld(x[a], cell[a]);
}

static int64_t position_cell_index(const orc_system *s, const double *x)
{
    int64_t mc[3];
    get_cell(x, s->cell, s->d, mc);
    return cell_index(s->ncells, mc, s->d);
}

/* build_neighbour_cells, neighbours.jl:94-111: offsets iterate first dim
 * fastest (Iterators.product), unique! keeps first occurrences. */
static void build_neighbour_cells(orc_system *s)
{
    int d = s->d;
    int cap = (d == 3) ? 27 : (d == 2 ? 9 : 3);
    s->nbr_cap = cap;
    s->nbr = (int64_t *)malloc(sizeof(int64_t) * (size_t)s->ncell_total * cap);
    s->nbr_cnt = (int32_t *)calloc((size_t)s->ncell_total, sizeof(int32_t));
    int64_t mc[3] = {1, 1, 1};
    for (int64_t it = 0; it < s->ncell_total; it++) {
        /* mc runs over product(1:n1, 1:n2, ...) first dim fastest */
        int64_t c = cell_index(s->ncells, mc, d);
        int64_t *out = s->nbr + (size_t)(c - 1) * cap;
        int cnt = 0;
        int noff = cap;
        for (int o = 0; o < noff; o++) {
            int64_t mc2[3];
            int r = o;
            for (int a = 0; a < d; a++) {
                mc2[a] = mc[a] + (r % 3) - 1;
                r /= 3;
            }
            int64_t c2 = cell_index(s->ncells, mc2, d);
            int dup = 0;
            for (int k = 0; k < cnt; k++)
                if (out[k] == c2) { dup = 1; break; }
            if (!dup)
                out[cnt++] = c2;
        }
        s->nbr_cnt[c - 1] = cnt;
        for (int a = 0; a < d; a++) {
            if (++mc[a] <= s->ncells[a]) break;
            mc[a] = 1;
        }
    }
}

/* build_neighbour_list!(::LinkedList), neighbours.jl:251-270 */
static void build_linked_list(orc_system *s)
{
    for (int64_t c = 0; c < s->ncell_total; c++)
        s->head[c] = -1;
    for (int64_t i = 1; i <= s->N; i++) {
        int64_t c = position_cell_index(s, s->pos + (size_t)(i - 1) * s->d);
        s->list[i - 1] = s->head[c - 1];
        s->head[c - 1] = i;
        s->cs[i - 1] = c;
    }
}

/* build_neighbour_list!(::CellList), neighbours.jl:145-156 (cells are not emptied: called once by System) */
static void build_cell_vectors(orc_system *s)
{
    for (int64_t i = 1; i <= s->N; i++) {
        int64_t c = position_cell_index(s, s->pos + (size_t)(i - 1) * s->d);
        s->cs[i - 1] = c;
        vec_push(&s->cells[c - 1], &s->cells_n[c - 1], &s->cells_cap[c - 1], i);
    }
}

/* update_neighbour_list!(i, c, c2, ::CellList) 162-170 / update_cell_list!(::VerletList) 468-476 */
static void update_cell_vectors(orc_system *s, int64_t i)
{
    int64_t c = s->cs[i - 1];
    int64_t c2 = position_cell_index(s, s->pos + (size_t)(i - 1) * s->d);
    if (c == c2)
        return;
    vec_remove_all(s->cells[c - 1], &s->cells_n[c - 1], i);
    vec_push(&s->cells[c2 - 1], &s->cells_n[c2 - 1], &s->cells_cap[c2 - 1], i);
    s->cs[i - 1] = c2;
}

/* build_neighbour_list!(::VerletList), neighbours.jl:429-462 */
static void build_verlet(orc_system *s)
{
    build_cell_vectors(s);
    for (int64_t i = 1; i <= s->N; i++) {
        const double *xi = s->pos + (size_t)(i - 1) * s->d;
        int64_t c = position_cell_index(s, xi);
        const int64_t *nb = s->nbr + (size_t)(c - 1) * s->nbr_cap;
        for (int k = 0; k < s->nbr_cnt[c - 1]; k++) {
            int64_t c2 = nb[k];
            for (int64_t q = 0; q < s->cells_n[c2 - 1]; q++) {
                int64_t j = s->cells[c2 - 1][q];
                if (j <= i)
                    continue;
                double r2 = nearest_image_distance_squared(xi, s->pos + (size_t)(j - 1) * s->d, s->box, s->d);
                if (r2 < s->rcutoff2) {
                    vec_push(&s->vnb[i - 1], &s->vnb_n[i - 1], &s->vnb_cap[i - 1], j);
                    vec_push(&s->vnb[j - 1], &s->vnb_n[j - 1], &s->vnb_cap[j - 1], i);
                }
            }
        }
        memcpy(s->last + (size_t)(i - 1) * s->d, xi, sizeof(double) * (size_t)s->d);
    }
}

/* update_neighbour_list!(::VerletList), neighbours.jl:494-543 */
static void update_verlet(orc_system *s, int64_t i)
{
    const double *xi = s->pos + (size_t)(i - 1) * s->d;
    const double *li = s->last + (size_t)(i - 1) * s->d;
    double d2 = 0.0;                       /* sum(abs2, dx) */
    for (int a = 0; a < s->d; a++) {
        double dx = xi[a] - li[a];
        d2 = (a == 0) ? dx * dx : d2 + dx * dx;
    }
    if (d2 < s->dr2o4)
        return;
    for (int64_t q = 0; q < s->vnb_n[i - 1]; q++) {
        int64_t j = s->vnb[i - 1][q];
        vec_remove_all(s->vnb[j - 1], &s->vnb_n[j - 1], i);
    }
    s->vnb_n[i - 1] = 0;
    update_cell_vectors(s, i);
    int64_t c = position_cell_index(s, xi);
    const int64_t *nb = s->nbr + (size_t)(c - 1) * s->nbr_cap;
    for (int k = 0; k < s->nbr_cnt[c - 1]; k++) {
        int64_t c2 = nb[k];
        for (int64_t q = 0; q < s->cells_n[c2 - 1]; q++) {
            int64_t j = s->cells[c2 - 1][q];
            if (i == j)
                continue;
            double r2 = nearest_image_distance_squared(xi, s->pos + (size_t)(j - 1) * s->d, s->box, s->d);
            if (r2 < s->rcutoff2) {
                vec_push(&s->vnb[i - 1], &s->vnb_n[i - 1], &s->vnb_cap[i - 1], j);
                vec_push(&s->vnb[j - 1], &s->vnb_n[j - 1], &s->vnb_cap[j - 1], i);
            }
        }
    }
    memcpy(s->last + (size_t)(i - 1) * s->d, xi, sizeof(double) * (size_t)s->d);
}

/* update_neighbour_list!, neighbours.jl:276-302; old_new_cell 306-312 */
static void update_linked_list(orc_system *s, int64_t i)
{
    if (s->list_type == ORC_CELLLIST) { update_cell_vectors(s, i); return; }
    if (s->list_type == ORC_VERLETLIST) { update_verlet(s, i); return; }
    if (s->list_type != ORC_LINKEDLIST)
        return;
    int64_t c = s->cs[i - 1];
    int64_t c2 = position_cell_index(s, s->pos + (size_t)(i - 1) * s->d);
    if (c == c2)
        return;
    if (s->head[c - 1] == i) {
        s->head[c - 1] = s->list[i - 1];
    } else {
        int64_t j = s->head[c - 1];
        while (s->list[j - 1] != i)
            j = s->list[j - 1];
        s->list[j - 1] = s->list[i - 1];
    }
    s->list[i - 1] = s->head[c2 - 1];
    s->head[c2 - 1] = i;
    s->cs[i - 1] = c2;
}

static inline const orc_model *get_model(const orc_system *s, int64_t i, int64_t j)
{
    /* ParticlesMC.jl:55 */
    return s->models + (size_t)(s->species[i - 1] - 1) * s->nspecies + (s->species[j - 1] - 1);
}

/* compute_energy_ij, atoms.jl:66-75 */
static inline double energy_ij(const orc_system *s, int64_t i, int64_t j, const double *xi)
{
    if (i == j)
        return 0.0;
    const orc_model *m = get_model(s, i, j);
    double r2 = nearest_image_distance_squared(xi, s->pos + (size_t)(j - 1) * s->d, s->box, s->d);
    if (r2 > m->rcut2)
        return 0.0;
    return orc_potential(r2, m);
}

/* compute_energy_particle, atoms.jl:81-88, with the LinkedIterator order of
 * neighbours.jl:332-377 (cell from the CURRENT position, not cs[i]) or the
 * EmptyList order 1:N of neighbours.jl:42-44. */
double orc_energy_particle(const orc_system *s, int64_t i)
{
    double e = 0.0;
    const double *xi = s->pos + (size_t)(i - 1) * s->d;
    if (s->list_type == ORC_EMPTYLIST) {
        for (int64_t j = 1; j <= s->N; j++)
            e += energy_ij(s, i, j, xi);
        return e;
    }
    if (s->list_type == ORC_VERLETLIST) {       /* (verlet_list)(system, i), neighbours.jl:422-424 */
        for (int64_t q = 0; q < s->vnb_n[i - 1]; q++)
            e += energy_ij(s, i, s->vnb[i - 1][q], xi);
        return e;
    }
    int64_t c = position_cell_index(s, xi);
    const int64_t *nb = s->nbr + (size_t)(c - 1) * s->nbr_cap;
    int cnt = s->nbr_cnt[c - 1];
    if (s->list_type == ORC_CELLLIST) {         /* (cell_list)(system, i), neighbours.jl:209-217 */
        for (int k = 0; k < cnt; k++)
            for (int64_t q = 0; q < s->cells_n[nb[k] - 1]; q++)
                e += energy_ij(s, i, s->cells[nb[k] - 1][q], xi);
        return e;
    }
    for (int k = 0; k < cnt; k++) {
        int64_t j = s->head[nb[k] - 1];
        while (j != -1) {
            e += energy_ij(s, i, j, xi);
            j = s->list[j - 1];
        }
    }
    return e;
}

int64_t orc_neighbours_of(const orc_system *s, int64_t i, int64_t *out, int64_t cap)
{
    int64_t n = 0;
    const double *xi = s->pos + (size_t)(i - 1) * s->d;
    if (s->list_type == ORC_EMPTYLIST) {
        for (int64_t j = 1; j <= s->N && n < cap; j++)
            out[n++] = j;
        return n;
    }
    if (s->list_type == ORC_VERLETLIST) {
        for (int64_t q = 0; q < s->vnb_n[i - 1] && n < cap; q++)
            out[n++] = s->vnb[i - 1][q];
        return n;
    }
    int64_t c = position_cell_index(s, xi);
    const int64_t *nb = s->nbr + (size_t)(c - 1) * s->nbr_cap;
    if (s->list_type == ORC_CELLLIST) {
        for (int k = 0; k < s->nbr_cnt[c - 1]; k++)
            for (int64_t q = 0; q < s->cells_n[nb[k] - 1]; q++)
                if (n < cap)
                    out[n++] = s->cells[nb[k] - 1][q];
        return n;
    }
    for (int k = 0; k < s->nbr_cnt[c - 1]; k++)
        for (int64_t j = s->head[nb[k] - 1]; j != -1; j = s->list[j - 1])
            if (n < cap)
                out[n++] = j;
    return n;
}

void orc_energy_particles(const orc_system *s, double *e)
{
    for (int64_t i = 1; i <= s->N; i++)
        e[i - 1] = orc_energy_particle(s, i);
}

/* Base: sum(::Vector{Float64}) is pairwise with 1024-element leaves
 * (base/reduce.jl mapreduce_impl); the leaf loop is @simd, so its association
 * is compiler-chosen in Julia -- a sequential leaf is used here (difference
 * <= a few ulp of the total; the tolerance on totals is 1e-6 relative). */
static double pairwise_sum(const double *v, int64_t lo, int64_t hi)
{
    if (hi - lo < 1024) {
        double a = v[lo];
        for (int64_t k = lo + 1; k <= hi; k++)
            a += v[k];
        return a;
    }
    int64_t mid = lo + ((hi - lo) >> 1);
    return pairwise_sum(v, lo, mid) + pairwise_sum(v, mid + 1, hi);
}

/* atoms.jl:51-52 */
double orc_total_energy(const orc_system *s)
{
    double *e = (double *)malloc(sizeof(double) * (size_t)s->N);
    orc_energy_particles(s, e);
    double t = pairwise_sum(e, 0, s->N - 1) / 2.0;
    free(e);
    return t;
}

/* SpeciesList(species), utils.jl:36-49 (assumes species are 1..S) */
static void build_species_list(orc_system *s)
{
    s->sp_ids = (int64_t **)calloc((size_t)s->nspecies, sizeof(int64_t *));
    s->sp_n = (int64_t *)calloc((size_t)s->nspecies, sizeof(int64_t));
    s->sp_heads = (int64_t *)calloc((size_t)s->N, sizeof(int64_t));
    for (int64_t i = 0; i < s->N; i++)
        s->sp_n[s->species[i] - 1]++;
    for (int k = 0; k < s->nspecies; k++) {
        s->sp_ids[k] = (int64_t *)malloc(sizeof(int64_t) * (size_t)(s->sp_n[k] > 0 ? s->sp_n[k] : 1));
        s->sp_n[k] = 0;
    }
    for (int64_t i = 0; i < s->N; i++) {
        int k = s->species[i] - 1;
        s->sp_ids[k][s->sp_n[k]] = i + 1;
        s->sp_n[k]++;
        s->sp_heads[i] = s->sp_n[k]; /* 1-based slot in its list */
    }
}

orc_system *orc_system_create(int64_t N, int32_t d, const double *pos, const int32_t *species,
                              double density, double temperature, int32_t nspecies,
                              const orc_model *models, int32_t list_type, int32_t *err)
{
    return orc_system_create_list(N, d, pos, species, density, temperature, nspecies, models, list_type, 0.0, err);
}

orc_system *orc_system_create_list(int64_t N, int32_t d, const double *pos, const int32_t *species,
                                   double density, double temperature, int32_t nspecies,
                                   const orc_model *models, int32_t list_type, double dr, int32_t *err)
{
    *err = 0;
    if (N < 1 || d < 1 || d > 3 || nspecies < 1) { *err = 2; return NULL; }
    orc_system *s = (orc_system *)calloc(1, sizeof(*s));
    s->N = N;
    s->d = d;
    s->density = density;
    s->temperature = temperature;
    s->nspecies = nspecies;
    s->list_type = list_type;
    s->pos = (double *)malloc(sizeof(double) * (size_t)N * d);
    memcpy(s->pos, pos, sizeof(double) * (size_t)N * d);
    s->species = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    memcpy(s->species, species, sizeof(int32_t) * (size_t)N);
    s->models = (orc_model *)malloc(sizeof(orc_model) * (size_t)nspecies * nspecies);
    memcpy(s->models, models, sizeof(orc_model) * (size_t)nspecies * nspecies);
    for (int64_t i = 0; i < N; i++)
        if (species[i] < 1 || species[i] > nspecies) { *err = 2; orc_system_destroy(s); return NULL; }
    /* atoms.jl:45: box = (N / density)^(1 / d) (libm pow; Julia's ^ may differ by 1 ulp, F7) */
    double b = pow((double)N / density, 1.0 / (double)d);
    for (int a = 0; a < d; a++)
        s->box[a] = b;
    /* atoms.jl:46 */
    double maxcut = 0.0;
    for (int k = 0; k < nspecies * nspecies; k++)
        if (models[k].rcut > maxcut) maxcut = models[k].rcut;
    if (list_type == ORC_LINKEDLIST) {
        /* LinkedList(box, rcut, N), neighbours.jl:236-244 */
        s->ncell_total = 1;
        for (int a = 0; a < d; a++) {
            s->cell[a] = s->box[a] / orc_jl_fld(s->box[a], maxcut);
            double q = s->box[a] / s->cell[a];
            if (!(q >= 1.0) || q != rint(q)) { *err = 2; orc_system_destroy(s); return NULL; } /* Int() InexactError / ncells = 0 */
            s->ncells[a] = (int64_t)q;
            s->ncell_total *= s->ncells[a];
        }
        s->head = (int64_t *)malloc(sizeof(int64_t) * (size_t)s->ncell_total);
        s->list = (int64_t *)calloc((size_t)N, sizeof(int64_t));
        s->cs = (int64_t *)calloc((size_t)N, sizeof(int64_t));
        build_neighbour_cells(s);
    }
    s->maxcut = maxcut;
    if (list_type == ORC_CELLLIST || list_type == ORC_VERLETLIST) {
        s->ncell_total = 1;
        if (list_type == ORC_VERLETLIST) {
            if (!(dr > 0.0)) { *err = 2; orc_system_destroy(s); return NULL; }   /* "Verlet list must have dr as a parameter" */
            s->dr = dr;
            s->dr2o4 = (dr * dr) / 4;                           /* (dr^2)/4, neighbours.jl:415 */
            s->rcutoff2 = (maxcut + dr) * (maxcut + dr);        /* (rcut + dr)^2 */
        }
        for (int a = 0; a < d; a++) {
            if (list_type == ORC_CELLLIST) {
                /* CellList(box, rcut, N), neighbours.jl:126-133: box / floor(Int, box / rcut); max(1, floor(Int, box / cell)) */
                double nf = floor(s->box[a] / maxcut);
                if (!(nf >= 1.0)) { *err = 2; orc_system_destroy(s); return NULL; }
                s->cell[a] = s->box[a] / (double)(int64_t)nf;
                double q = floor(s->box[a] / s->cell[a]);
                s->ncells[a] = q >= 1.0 ? (int64_t)q : 1;
            } else {
                /* VerletList(box, rcut, N), neighbours.jl:405-406 */
                s->cell[a] = s->box[a] / orc_jl_fld(s->box[a], maxcut + dr);
                double q = s->box[a] / s->cell[a];
                if (!(q >= 1.0) || q != rint(q)) { *err = 2; orc_system_destroy(s); return NULL; }
                s->ncells[a] = (int64_t)q;
            }
            s->ncell_total *= s->ncells[a];
        }
        s->cs = (int64_t *)calloc((size_t)N, sizeof(int64_t));
        s->cells = (int64_t **)calloc((size_t)s->ncell_total, sizeof(int64_t *));
        s->cells_n = (int64_t *)calloc((size_t)s->ncell_total, sizeof(int64_t));
        s->cells_cap = (int64_t *)calloc((size_t)s->ncell_total, sizeof(int64_t));
        build_neighbour_cells(s);
        if (list_type == ORC_VERLETLIST) {
            s->vnb = (int64_t **)calloc((size_t)N, sizeof(int64_t *));
            s->vnb_n = (int64_t *)calloc((size_t)N, sizeof(int64_t));
            s->vnb_cap = (int64_t *)calloc((size_t)N, sizeof(int64_t));
            s->last = (double *)calloc((size_t)N * d, sizeof(double));
        }
    }
    build_species_list(s);           /* atoms.jl:48 */
    if (list_type == ORC_LINKEDLIST)
        build_linked_list(s);        /* atoms.jl:50 */
    if (list_type == ORC_CELLLIST)
        build_cell_vectors(s);
    if (list_type == ORC_VERLETLIST)
        build_verlet(s);
    double e = orc_total_energy(s);  /* atoms.jl:51-52 */
    if (isinf(e) || isnan(e)) {      /* atoms.jl:53-55 */
        *err = 1;
        orc_system_destroy(s);
        return NULL;
    }
    s->energy = e;
    return s;
}

void orc_system_destroy(orc_system *s)
{
    if (!s) return;
    free(s->pos); free(s->species); free(s->models);
    free(s->head); free(s->list); free(s->cs); free(s->nbr); free(s->nbr_cnt);
    if (s->sp_ids)
        for (int k = 0; k < s->nspecies; k++) free(s->sp_ids[k]);
    free(s->sp_ids); free(s->sp_n); free(s->sp_heads);
    if (s->cells)
        for (int64_t c = 0; c < s->ncell_total; c++) free(s->cells[c]);
    free(s->cells); free(s->cells_n); free(s->cells_cap);
    if (s->vnb)
        for (int64_t i = 0; i < s->N; i++) free(s->vnb[i]);
    free(s->vnb); free(s->vnb_n); free(s->vnb_cap); free(s->last);
    free(s);
}

double orc_system_energy(const orc_system *s) { return s->energy; }
void orc_system_set_energy(orc_system *s, double e) { s->energy = e; }
void orc_system_set_temperature(orc_system *s, double T) { s->temperature = T; }
void orc_system_box(const orc_system *s, double *box) { for (int a = 0; a < s->d; a++) box[a] = s->box[a]; }
void orc_system_cell_geometry(const orc_system *s, double *cell, int64_t *ncells)
{
    for (int a = 0; a < s->d; a++) { cell[a] = s->cell[a]; ncells[a] = s->ncells[a]; }
}
void orc_system_get_cells(const orc_system *s, int64_t *cs) { memcpy(cs, s->cs, sizeof(int64_t) * (size_t)s->N); }
int64_t orc_system_cell_members(const orc_system *s, int64_t c, int64_t *out, int64_t cap)   /* cells[c], c 1-based */
{
    if (!s->cells) return -1;
    int64_t n = s->cells_n[c - 1];
    for (int64_t q = 0; q < n && q < cap; q++) out[q] = s->cells[c - 1][q];
    return n;
}
void orc_system_get_linked_list(const orc_system *s, int64_t *head, int64_t *list)
{
    memcpy(head, s->head, sizeof(int64_t) * (size_t)s->ncell_total);
    memcpy(list, s->list, sizeof(int64_t) * (size_t)s->N);
}
int64_t orc_system_neighbour_cells(const orc_system *s, int64_t c, int64_t *out)
{
    int cnt = s->nbr_cnt[c - 1];
    memcpy(out, s->nbr + (size_t)(c - 1) * s->nbr_cap, sizeof(int64_t) * (size_t)cnt);
    return cnt;
}
void orc_system_get_positions(const orc_system *s, double *pos) { memcpy(pos, s->pos, sizeof(double) * (size_t)s->N * s->d); }
void orc_system_get_species(const orc_system *s, int32_t *sp) { memcpy(sp, s->species, sizeof(int32_t) * (size_t)s->N); }
int64_t orc_species_count(const orc_system *s, int32_t sp) { return s->sp_n[sp - 1]; }
void orc_species_ids(const orc_system *s, int32_t sp, int64_t *ids) { memcpy(ids, s->sp_ids[sp - 1], sizeof(int64_t) * (size_t)s->sp_n[sp - 1]); }


/* EnergyBias (moves.jl:246-280).  sample: w = exp.(theta .* compute_energy_particle(system, sp_ids[species])),
 * w ./= sum(w) (Base pairwise sum), rand(rng, Categorical(w)) as the inverse-CDF walk `cp <= u` over the caller's
 * uniform; log density: theta1 e_i + theta2 e_j - log(sum(w1)) - log(sum(w2)).  gi / gj > 0: density of a given pair.
 * exp and log are the fixed-operation-sequence versions, so the device reproduces every bit. */
int orc_energy_bias(const orc_system *s, int32_t sp1, int32_t sp2, double theta1, double theta2, double u1, double u2,
                    int64_t gi, int64_t gj, int64_t *i_out, int64_t *j_out, double *logq)
{
    const int32_t sp[2] = {sp1, sp2};
    const double theta[2] = {theta1, theta2}, u[2] = {u1, u2};
    const int64_t given[2] = {gi, gj};
    int64_t pick[2] = {0, 0};
    double numer = 0.0, lse[2] = {0.0, 0.0};
    for (int side = 0; side < 2; side++) {
        const int64_t n = s->sp_n[sp[side] - 1];
        const int64_t *ids = s->sp_ids[sp[side] - 1];
        if (n <= 0) return 1;
        double *w = (double *)malloc(sizeof(double) * (size_t)n);
        for (int64_t k = 0; k < n; k++) w[k] = orc_det_exp(theta[side] * orc_energy_particle(s, ids[k]));
        const double S = pairwise_sum(w, 0, n - 1);
        if (given[side] > 0) {
            pick[side] = given[side];
        } else {
            for (int64_t k = 0; k < n; k++) w[k] = w[k] / S;
            int64_t i = 0;
            double cp = w[0];
            while (cp <= u[side] && i < n - 1) { i++; cp += w[i]; }
            pick[side] = ids[i];
        }
        lse[side] = orc_det_log(S);
        numer += theta[side] * orc_energy_particle(s, pick[side]);
        free(w);
    }
    if (i_out) *i_out = pick[0];
    if (j_out) *j_out = pick[1];
    if (logq) *logq = numer - lse[0] - lse[1];
    return 0;
}
void orc_species_heads(const orc_system *s, int64_t *h) { memcpy(h, s->sp_heads, sizeof(int64_t) * (size_t)s->N); }

/* ======================================================================== */
/* moves.jl                                                                  */
/* ======================================================================== */

/* perform_action!(::Displacement) moves.jl:57-67 wrapped by
 * Arianna.perform_action! moves.jl:11-20 */
void orc_displacement_perform(orc_system *s, int64_t i, const double *delta,
                              double *e1, double *e2, double *de)
{
    double a = orc_energy_particle(s, i);
    double *x = s->pos + (size_t)(i - 1) * s->d;
    for (int k = 0; k < s->d; k++)
        x[k] = x[k] + delta[k];          /* update_position! moves.jl:46-48 */
    update_linked_list(s, i);
    double b = orc_energy_particle(s, i);
    if (isinf(a) || isinf(b)) {
        *de = 0.0;
    } else {
        *de = b - a;
        s->energy += *de;
    }
    *e1 = a;
    *e2 = b;
}

/* revert_action!(::Displacement) moves.jl:76-81; caller passes the inverted
 * delta (invert_action! moves.jl:88-90) */
void orc_displacement_revert(orc_system *s, int64_t i, const double *minus_delta, double de)
{
    double *x = s->pos + (size_t)(i - 1) * s->d;
    for (int k = 0; k < s->d; k++)
        x[k] = x[k] + minus_delta[k];
    s->energy -= de;
    update_linked_list(s, i);
}

/* update_species_list!, moves.jl:175-179 */
static void update_species_list(orc_system *s, int32_t sp1, int32_t sp2, int64_t i, int64_t j)
{
    s->sp_ids[sp1 - 1][s->sp_heads[i - 1] - 1] = j;
    s->sp_ids[sp2 - 1][s->sp_heads[j - 1] - 1] = i;
    int64_t t = s->sp_heads[i - 1];
    s->sp_heads[i - 1] = s->sp_heads[j - 1];
    s->sp_heads[j - 1] = t;
}

/* perform_action!(::DiscreteSwap) moves.jl:187-194 + swap_particle_species!
 * 159-167, wrapped by moves.jl:11-20 */
void orc_swap_perform(orc_system *s, int64_t i, int64_t j, int32_t sp1, int32_t sp2,
                      double *e1, double *e2, double *de)
{
    int32_t spi = s->species[i - 1], spj = s->species[j - 1];
    double e1i = orc_energy_particle(s, i);
    double e1j = orc_energy_particle(s, j);
    s->species[i - 1] = spj;
    s->species[j - 1] = spi;
    double e2i = orc_energy_particle(s, i);
    double e2j = orc_energy_particle(s, j);
    double a = e1i + e1j, b = e2i + e2j;
    update_species_list(s, sp1, sp2, i, j);
    if (isinf(a) || isinf(b)) {
        *de = 0.0;
    } else {
        *de = b - a;
        s->energy += *de;
    }
    *e1 = a;
    *e2 = b;
}

/* revert_action!(::DiscreteSwap) moves.jl:201-207; called with the inverted
 * action (i, j exchanged by invert_action!, moves.jl:212-214) */
void orc_swap_revert(orc_system *s, int64_t i, int64_t j, int32_t sp1, int32_t sp2, double de)
{
    int32_t spi = s->species[i - 1], spj = s->species[j - 1];
    s->species[j - 1] = spi;
    s->species[i - 1] = spj;
    update_species_list(s, sp1, sp2, i, j);
    s->energy -= de;
}

/* ======================================================================== */
/* Philox4x32-10 (Salmon et al., SC'11) and deterministic log / sincos        */
/* ======================================================================== */

void orc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Natural log for x in (0, 1], normal numbers only, in a fixed sequence of
 * IEEE operations (fdlibm-style argument reduction and minimax polynomial), so
 * that a CPU and a GPU produce the same bits. */
double orc_det_log(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    uint64_t b;
    memcpy(&b, &x, 8);
    int64_t k = (int64_t)((b >> 52) & 0x7ff) - 1023;
    uint64_t mant = b & 0x000fffffffffffffull;
    if (mant >= 0x6a09e667f3bcdull) { /* m >= sqrt(2): use m / 2 */
        k += 1;
        b = mant | 0x3fe0000000000000ull;
    } else {
        b = mant | 0x3ff0000000000000ull;
    }
    double m;
    memcpy(&m, &b, 8);
    double f = m - 1.0;
    double dk = (double)k;
    double s = f / (2.0 + f);
    double z = s * s;
    double w = z * z;
    double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
    double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
    double R = t2 + t1;
    double hfsq = 0.5 * f * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp(x) in a fixed sequence of IEEE operations (fdlibm e_exp.c), finite |x| <= 700: the weights of the EnergyBias
 * policy must come out with the same bits on a CPU and a GPU (pmc_device.h:det_exp). */
double orc_det_exp(double x)
{
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        invln2 = 1.44269504088896338700e+00,
                        P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                        P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
    if (x > 700.0) x = 700.0;
    if (x < -700.0) x = -700.0;
    double hi = x, lo = 0.0;
    int64_t k = 0;
    if (x > 0.34657359027997264 || x < -0.34657359027997264) {
        k = (int64_t)(invln2 * x + (x < 0.0 ? -0.5 : 0.5));
        double t = (double)k;
        hi = x - t * ln2_hi;
        lo = t * ln2_lo;
        x = hi - lo;
    }
    double t = x * x;
    double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
    if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
    double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
    uint64_t b;
    memcpy(&b, &y, 8);
    b += (uint64_t)(k << 52);
    memcpy(&y, &b, 8);
    return y;
}

/* sin and cos of 2*pi*w/2^32: the octant comes from the top 3 bits, the
 * remainder is reduced exactly, polynomial kernels on [0, pi/4]. */
void orc_det_sincos2pi(uint32_t w, double *sn, double *cs)
{
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                        S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                        S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10,
                        C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                        C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                        C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11,
                        PIO4 = 7.85398163397448278999e-01;
    uint32_t oct = w >> 29;
    uint32_t rem = w & 0x1fffffffu;
    /* odd octants measure the angle back from the next multiple of pi/4 */
    uint32_t r = (oct & 1) ? (0x20000000u - rem) : rem;
    double y = ((double)r * (1.0 / 536870912.0)) * PIO4; /* in [0, pi/4] */
    double z = y * y;
    double ps = y + y * (z * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6))))));
    double pc = 1.0 - (0.5 * z - z * (z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))));
    double s, c;
    switch (oct) {
    case 0: s = ps;  c = pc;  break;
    case 1: s = pc;  c = ps;  break;
    case 2: s = pc;  c = -ps; break;
    case 3: s = ps;  c = -pc; break;
    case 4: s = -ps; c = -pc; break;
    case 5: s = -pc; c = -ps; break;
    case 6: s = -pc; c = ps;  break;
    default: s = -ps; c = pc; break;
    }
    *sn = s;
    *cs = c;
}

/* Box-Muller: u1 = (w1 + 1) / 2^32 in (0, 1], angle = 2 pi w2 / 2^32 */
void orc_box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1)
{
    double u1 = ((double)w1 + 1.0) * (1.0 / 4294967296.0);
    double r = sqrt(-2.0 * orc_det_log(u1));
    double s, c;
    orc_det_sincos2pi(w2, &s, &c);
    *z0 = r * c;
    *z1 = r * s;
}

/* 53-bit uniform in (0, 1) */
double orc_uniform53(uint32_t hi, uint32_t lo)
{
    double a = (double)(hi >> 5), b = (double)(lo >> 6);
    return ((a * 67108864.0 + b) + 0.5) * (1.0 / 9007199254740992.0);
}

static inline uint32_t mulhi32(uint32_t a, uint32_t b)
{
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
}

enum { TAG_SEQ0 = 1, TAG_SEQ1 = 2, TAG_SHIFT = 3, TAG_SLOT0 = 4, TAG_SLOT1 = 5 };

static int select_move(const orc_move *pool, int npool, uint32_t w)
{
    if (npool == 1)
        return 0;
    double tot = 0.0;
    for (int m = 0; m < npool; m++)
        tot += pool[m].probability;
    double u = (double)w * (1.0 / 4294967296.0) * tot;
    double acc = 0.0;
    for (int m = 0; m < npool; m++) {
        acc += pool[m].probability;
        if (u < acc)
            return m;
    }
    return npool - 1;
}

/* ======================================================================== */
/* Sequential Metropolis driver (Arianna mc_step! contract)       */
/* ======================================================================== */

void orc_mc_steps_sequential(orc_system *s, const orc_move *pool, int32_t npool,
                             uint64_t seed, uint32_t chain, uint64_t step0, int64_t nsteps,
                             orc_stats *stats)
{
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) + chain};
    for (int64_t t = 0; t < nsteps; t++) {
        uint64_t g = step0 + (uint64_t)t;
        uint32_t c0[4] = {(uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ0};
        uint32_t c1[4] = {(uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ1};
        uint32_t R[4], Q[4];
        orc_philox4x32(c0, key, R);
        orc_philox4x32(c1, key, Q);
        int m = select_move(pool, npool, R[0]);
        double lnu = orc_det_log(orc_uniform53(R[2], R[3]));
        double e1, e2, de;
        stats->attempted[m]++;
        if (pool[m].kind == ORC_MOVE_DISPLACEMENT) {
            /* sample_action!, moves.jl:120-123 */
            int64_t i = 1 + (int64_t)mulhi32(R[1], (uint32_t)s->N);
            double z[4], delta[3], mdelta[3];
            orc_box_muller(Q[0], Q[1], &z[0], &z[1]);
            orc_box_muller(Q[2], Q[3], &z[2], &z[3]);
            for (int a = 0; a < s->d; a++) {
                delta[a] = pool[m].sigma * z[a];
                mdelta[a] = -delta[a];
            }
            orc_displacement_perform(s, i, delta, &e1, &e2, &de);
            /* delta_log_target_density utils.jl:8-10; the forward and backward
             * log_proposal_density (moves.jl:110-112) are bitwise equal and cancel */
            double arg = -(e2 - e1) / s->temperature;
            if (lnu < arg)
                stats->accepted[m]++;
            else
                orc_displacement_revert(s, i, mdelta, de);
        } else {
            /* sample_action!, moves.jl:238-241 */
            int32_t sa = pool[m].species[0], sb = pool[m].species[1];
            int64_t na = s->sp_n[sa - 1], nb = s->sp_n[sb - 1];
            if (na == 0 || nb == 0)
                continue;
            int64_t i = s->sp_ids[sa - 1][mulhi32(R[1], (uint32_t)na)];
            int64_t j = s->sp_ids[sb - 1][mulhi32(Q[0], (uint32_t)nb)];
            orc_swap_perform(s, i, j, sa, sb, &e1, &e2, &de);
            double arg = -(e2 - e1) / s->temperature;
            if (lnu < arg)
                stats->accepted[m]++;
            else
                orc_swap_revert(s, j, i, sa, sb, de); /* inverted action: i <-> j */
        }
    }
}

/* ======================================================================== */
/* Parallel (checkerboard) schedule restatement -- see DESIGN.md "schedule"   */
/*                                                                            */
/* Same pair arithmetic as above (vector_1D / potential), but                 */
/*  - per-pair energies are clamped to UMAX and rounded to multiples of 2^-32 */
/*    so that every sum is exact and therefore independent of summation order */
/*  - one sweep = random grid shift, binning into an even grid of cells of    */
/*    width >= rcut, 2^d colour passes; each cell of the active colour does   */
/*    n_c proposal slots on its own particles; moves leaving the cell are     */
/*    rejected.                                                               */
/* ======================================================================== */

#define CB_UMAX 1024.0
#define CB_CQ 1572864.0           /* 1.5 * 2^20: rounds to multiples of 2^-32 */
#define CB_FIX 4294967296.0       /* 2^32 */
#define ORC_CB_SINGLE_MAX 1024    /* particles a single-cell box may hold (the staging capacity of one cell visit) */

struct orc_cb {
    int64_t N;
    int32_t d;
    double *pos;
    int32_t *species;
    double box[3];
    double temperature;
    int32_t nspecies;
    orc_model *models;
    uint32_t key[2];
    int64_t n_mc[3];
    double w[3];
    int64_t ncell;
    int64_t *cell_start; /* ncell + 1 */
    int64_t *cell_ids;   /* N, 0-based ids, ascending inside a cell */
    int64_t *cell_of;    /* N */
    int64_t e_fix;
    /* Periodic images (DESIGN.md 2.2 "images"): positions are never wrapped (src/moves.jl:46-48), so every particle carries a
     * PERSISTENT integer image k per dimension, fixed when the state is handed over (orc_cb_create = pmc_upload) as
     * floor(x / L) (|k| <= 511, else 0); all distances of the schedule are taken between the folded coordinates
     * xf = x - k L (k = 0: xf = x), recomputed from the positions at every binning. */
    int32_t *img;        /* N * d */
    double *xf;          /* N * d, folded copy of the current pass */
};

#define CB_IMG_MAX 511
static inline double cb_fold(double x, int32_t k, double L) { return k == 0 ? x : x - (double)k * L; }
static inline double cb_unfold(double xf, int32_t k, double L) { return k == 0 ? xf : xf + (double)k * L; }
/* the folded coordinate a move to t = xf + delta is evaluated and stored at: it survives the round trip through the
 * unwrapped position array exactly (k = 0: t itself, i.e. the reference's x + delta) */
static inline double cb_settle(double t, int32_t k, double L)
{
    if (k == 0) return t;
    double kl = (double)k * L;
    return (t + kl) - kl;
}

static inline double cb_quant(double u)
{
    double c = fmin(u, CB_UMAX);
    return (c + CB_CQ) - CB_CQ;
}

static inline double cb_pair(const orc_cb *c, const double *xi, int32_t spi, int64_t j)
{
    const orc_model *m = c->models + (size_t)(spi - 1) * c->nspecies + (c->species[j] - 1);
    double r2 = nearest_image_distance_squared(xi, c->xf + (size_t)j * c->d, c->box, c->d);   /* xi: a FOLDED coordinate */
    if (r2 > m->rcut2)
        return 0.0;
    return cb_quant(orc_potential(r2, m));
}

static void cb_bin(orc_cb *c, const double *shift)
{
    int d = c->d;
    memset(c->cell_start, 0, sizeof(int64_t) * (size_t)(c->ncell + 1));
    for (int64_t i = 0; i < c->N; i++) {
        int64_t lin = 0;
        for (int a = 0; a < d; a++) {
            c->xf[(size_t)i * d + a] = cb_fold(c->pos[(size_t)i * d + a], c->img[(size_t)i * d + a], c->box[a]);
            double f = floor((c->xf[(size_t)i * d + a] - shift[a]) / c->w[a]);
            int64_t q = (int64_t)f;
            int64_t r = q % c->n_mc[a];
            if (r < 0) r += c->n_mc[a];
            lin = lin * c->n_mc[a] + r;
        }
        c->cell_of[i] = lin;
        c->cell_start[lin + 1]++;
    }
    for (int64_t k = 0; k < c->ncell; k++)
        c->cell_start[k + 1] += c->cell_start[k];
    int64_t *fill = (int64_t *)calloc((size_t)c->ncell, sizeof(int64_t));
    for (int64_t i = 0; i < c->N; i++) {
        int64_t lin = c->cell_of[i];
        c->cell_ids[c->cell_start[lin] + fill[lin]++] = i;
    }
    free(fill);
}

/* ids of all particles in the 3^d cells around cell `lin` (any order) */
static int64_t cb_gather(const orc_cb *c, int64_t lin, int64_t *out)
{
    int d = c->d;
    int64_t cc[3] = {0, 0, 0}, r = lin;
    for (int a = d - 1; a >= 0; a--) { cc[a] = r % c->n_mc[a]; r /= c->n_mc[a]; }
    int nn = (d == 3) ? 27 : (d == 2 ? 9 : 3);
    int64_t n = 0;
    for (int o = 0; o < nn; o++) {
        int rr = o, dup = 0;
        int64_t l2 = 0;
        for (int a = 0; a < d; a++) {
            int off = (rr % 3) - 1;
            int64_t q = cc[a] + off;
            rr /= 3;
            if (c->n_mc[a] == 1 && off != 0) dup = 1;      /* single-cell box: the cell is its own neighbour once, not 3^d times */
            if (q < 0) q += c->n_mc[a];
            if (q >= c->n_mc[a]) q -= c->n_mc[a];
            l2 = l2 * c->n_mc[a] + q;
        }
        if (dup) continue;
        for (int64_t k = c->cell_start[l2]; k < c->cell_start[l2 + 1]; k++)
            out[n++] = c->cell_ids[k];
    }
    return n;
}

static double cb_energy_excl(const orc_cb *c, const double *x, int32_t sp, const int64_t *cand,
                             int64_t ncand, int64_t ex1, int64_t ex2)
{
    double e = 0.0;
    for (int64_t k = 0; k < ncand; k++) {
        int64_t j = cand[k];
        if (j == ex1 || j == ex2) continue;
        e += cb_pair(c, x, sp, j);
    }
    return e;
}

orc_cb *orc_cb_create(int64_t N, int32_t d, const double *pos, const int32_t *species,
                      const double *box, double temperature, int32_t nspecies,
                      const orc_model *models, uint64_t seed, uint32_t chain, int32_t *err)
{
    *err = 0;
    orc_cb *c = (orc_cb *)calloc(1, sizeof(*c));
    c->N = N; c->d = d; c->temperature = temperature; c->nspecies = nspecies;
    c->pos = (double *)malloc(sizeof(double) * (size_t)N * d);
    memcpy(c->pos, pos, sizeof(double) * (size_t)N * d);
    c->species = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    memcpy(c->species, species, sizeof(int32_t) * (size_t)N);
    c->models = (orc_model *)malloc(sizeof(orc_model) * (size_t)nspecies * nspecies);
    memcpy(c->models, models, sizeof(orc_model) * (size_t)nspecies * nspecies);
    c->key[0] = (uint32_t)seed;
    c->key[1] = (uint32_t)(seed >> 32) + chain;
    double maxcut = 0.0;
    for (int k = 0; k < nspecies * nspecies; k++)
        if (models[k].rcut > maxcut) maxcut = models[k].rcut;
    /* widest even grid whose cells are wider than rcut by a safety margin */
    double need = maxcut * (1.0 + 1.0 / 1048576.0);
    c->ncell = 1;
    int nsmall = 0;
    for (int a = 0; a < d; a++) {
        c->box[a] = box[a];
        int64_t n = (int64_t)floor(box[a] / need);
        n -= (n & 1);
        while (n >= 4 && box[a] / (double)n < need) n -= 2;
        if (n < 4) { nsmall++; n = 1; }
        c->n_mc[a] = n;
        c->w[a] = box[a] / (double)n;
        c->ncell *= n;
    }
    /* SINGLE-CELL BOXES (round 5).  A box of fewer than 4 cutoffs per side cannot be two-coloured periodically; if it is small
     * in EVERY dimension, still at least 2 (1 + 2^-10) cutoffs wide (the two roundings of the minimum image -- and the device's FP32
     * pre-filter -- then only differ for pairs beyond the cutoff) and holds at most ORC_CB_SINGLE_MAX particles, the whole periodic box is ONE cell: one colour, one visit per
     * pass, its N proposal slots in order against all N particles, no move ever "leaves the cell" -- i.e. the plain Metropolis
     * chain of src/moves.jl:57-81 with a uniformly picked particle (and the reference's global DoubleUniform for swaps), keyed
     * (cell 0, slot t, sweep g).  The reference's own validation example is such a box (N = 1000, rcut = 4 sigma,
     * examples/lj-mixture/run-validation.py:26-34,155-166).  Mixed boxes (small in some dimensions only) stay refused. */
    if (nsmall > 0) {
        int ok = (nsmall == d) && N <= ORC_CB_SINGLE_MAX;
        for (int a = 0; a < d; a++) if (box[a] < 2.0 * maxcut * (1.0 + 1.0 / 1024.0)) ok = 0;
        if (!ok) { *err = 2; orc_cb_destroy(c); return NULL; }
    }
    c->cell_start = (int64_t *)calloc((size_t)c->ncell + 1, sizeof(int64_t));
    c->cell_ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    c->cell_of = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    c->img = (int32_t *)malloc(sizeof(int32_t) * (size_t)N * d);
    c->xf = (double *)malloc(sizeof(double) * (size_t)N * d);
    for (int64_t i = 0; i < N; i++)
        for (int a = 0; a < d; a++) {
            double f = floor(pos[(size_t)i * d + a] / box[a]);
            c->img[(size_t)i * d + a] = (fabs(f) <= (double)CB_IMG_MAX) ? (int32_t)f : 0;
        }
    c->e_fix = orc_cb_total_energy_fix(c);
    return c;
}

void orc_cb_destroy(orc_cb *c)
{
    if (!c) return;
    free(c->pos); free(c->species); free(c->models);
    free(c->cell_start); free(c->cell_ids); free(c->cell_of);
    free(c->img); free(c->xf);
    free(c);
}

void orc_cb_grid(const orc_cb *c, int64_t *n_mc, double *w)
{
    for (int a = 0; a < c->d; a++) { n_mc[a] = c->n_mc[a]; w[a] = c->w[a]; }
}
void orc_cb_get_positions(const orc_cb *c, double *pos) { memcpy(pos, c->pos, sizeof(double) * (size_t)c->N * c->d); }
void orc_cb_get_species(const orc_cb *c, int32_t *sp) { memcpy(sp, c->species, sizeof(int32_t) * (size_t)c->N); }
int64_t orc_cb_energy_fix(const orc_cb *c) { return c->e_fix; }
void orc_cb_set_temperature(orc_cb *c, double T) { c->temperature = T; }

int64_t orc_cb_total_energy_fix(const orc_cb *cc)
{
    orc_cb *c = (orc_cb *)cc;
    double shift[3] = {0.0, 0.0, 0.0};
    cb_bin(c, shift);
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)c->N);
    int64_t tot = 0;
    for (int64_t lin = 0; lin < c->ncell; lin++) {
        if (c->cell_start[lin + 1] == c->cell_start[lin]) continue;
        int64_t nc = cb_gather(c, lin, cand);
        for (int64_t k = c->cell_start[lin]; k < c->cell_start[lin + 1]; k++) {
            int64_t i = c->cell_ids[k];
            double e = cb_energy_excl(c, c->xf + (size_t)i * c->d, c->species[i], cand, nc, i, -1);
            tot += (int64_t)(e * CB_FIX);
        }
    }
    free(cand);
    return tot / 2;
}

double orc_cb_energy_at(const orc_cb *cc, int64_t i, const double *x, int32_t sp)
{
    /* brute force over all particles: exact sums do not depend on the set of
     * zero-contributing (out-of-range) candidates.  x is unwrapped, in the frame of particle i: folded with its image. */
    orc_cb *c = (orc_cb *)cc;
    double xq[3] = {0, 0, 0};
    for (int64_t j = 0; j < c->N; j++)
        for (int a = 0; a < c->d; a++)
            c->xf[(size_t)j * c->d + a] = cb_fold(c->pos[(size_t)j * c->d + a], c->img[(size_t)j * c->d + a], c->box[a]);
    for (int a = 0; a < c->d; a++) xq[a] = cb_fold(x[a], c->img[(size_t)(i - 1) * c->d + a], c->box[a]);
    double e = 0.0;
    for (int64_t j = 0; j < c->N; j++)
        if (j != i - 1)
            e += cb_pair(c, xq, sp, j);
    return e;
}

void orc_cb_sweeps(orc_cb *c, const orc_move *pool, int32_t npool, uint64_t sweep0,
                   int64_t nsweeps, orc_stats *stats)
{
    orc_cb_sweeps_rounds(c, pool, npool, sweep0, nsweeps, 1, stats);
}

/* The schedule with `rounds` m >= 1 ("rounds per visit"): the sweeps of a call are taken in PASSES of
 * k = min(m, sweeps left) sweeps.  A pass that starts at sweep counter g draws ONE grid shift (keyed by g), bins once, and
 * every cell visit of the pass runs k n_c proposal slots t = 0 .. k n_c - 1 (keyed (cell, t, g)) against the same staged
 * neighbourhood; the counter then advances by k.  m = 1 is the schedule of orc_cb_sweeps.  Every slot is an attempted
 * move: a pass is k N attempted moves. */
void orc_cb_sweeps_rounds(orc_cb *c, const orc_move *pool, int32_t npool, uint64_t sweep0,
                          int64_t nsweeps, int32_t rounds, orc_stats *stats)
{
    int d = c->d;
    int64_t *cand = (int64_t *)malloc(sizeof(int64_t) * (size_t)c->N);
    if (rounds < 1) rounds = 1;
    for (int64_t sw = 0, kr = 0; sw < nsweeps; sw += kr) {
        kr = nsweeps - sw < rounds ? nsweeps - sw : rounds;
        uint64_t g = sweep0 + (uint64_t)sw;
        uint32_t s_lo = (uint32_t)g, s_hi = (uint32_t)(g >> 32) & 0xffffffu;
        uint32_t cs[4] = {0, 0, s_lo, s_hi | ((uint32_t)TAG_SHIFT << 24)}, W[4];
        orc_philox4x32(cs, c->key, W);
        double shift[3] = {0, 0, 0};
        for (int a = 0; a < d; a++)
            shift[a] = ((double)W[a] * (1.0 / 4294967296.0)) * c->w[a];
        cb_bin(c, shift);
        int ncol = 1 << d;
        for (int col = 0; col < ncol; col++) {
            for (int64_t lin = 0; lin < c->ncell; lin++) {
                int64_t r = lin, par = 0;
                for (int a = d - 1; a >= 0; a--) {
                    par |= ((r % c->n_mc[a]) & 1) << a;
                    r /= c->n_mc[a];
                }
                if (par != col) continue;
                int64_t k0 = c->cell_start[lin];
                int64_t nc = c->cell_start[lin + 1] - k0;
                if (nc == 0) continue;
                int64_t ncand = cb_gather(c, lin, cand);
                for (int64_t t = 0; t < kr * nc; t++) {
                    uint32_t c0[4] = {(uint32_t)lin, (uint32_t)t, s_lo, s_hi | ((uint32_t)TAG_SLOT0 << 24)};
                    uint32_t c1[4] = {(uint32_t)lin, (uint32_t)t, s_lo, s_hi | ((uint32_t)TAG_SLOT1 << 24)};
                    uint32_t R[4], Q[4];
                    orc_philox4x32(c0, c->key, R);
                    orc_philox4x32(c1, c->key, Q);
                    int m = select_move(pool, npool, R[0]);
                    double lnu = orc_det_log(orc_uniform53(R[2], R[3]));
                    stats->attempted[m]++;
                    if (pool[m].kind == ORC_MOVE_DISPLACEMENT) {
                        int64_t i = c->cell_ids[k0 + mulhi32(R[1], (uint32_t)nc)];
                        double z[4], xn[3];
                        double *xo = c->xf + (size_t)i * d;            /* folded coordinates */
                        const int32_t *ki = c->img + (size_t)i * d;
                        orc_box_muller(Q[0], Q[1], &z[0], &z[1]);
                        orc_box_muller(Q[2], Q[3], &z[2], &z[3]);
                        int inside = 1;
                        for (int a = 0; a < d; a++) {
                            xn[a] = cb_settle(xo[a] + pool[m].sigma * z[a], ki[a], c->box[a]);
                            if (c->n_mc[a] > 1 && floor((xn[a] - shift[a]) / c->w[a]) != floor((xo[a] - shift[a]) / c->w[a]))
                                inside = 0;                             /* (a single-cell box has no cell to leave) */
                        }
                        if (!inside) continue;
                        double e1 = cb_energy_excl(c, xo, c->species[i], cand, ncand, i, -1);
                        double e2 = cb_energy_excl(c, xn, c->species[i], cand, ncand, i, -1);
                        double de = e2 - e1;
                        if (c->temperature * lnu < -de) {   /* accept iff T ln u < -dE (DESIGN.md "schedule") */
                            for (int a = 0; a < d; a++) {
                                xo[a] = xn[a];
                                c->pos[(size_t)i * d + a] = cb_unfold(xn[a], ki[a], c->box[a]);   /* k = 0: x + delta (src/moves.jl:46-48) */
                            }
                            c->e_fix += (int64_t)(de * CB_FIX);
                            stats->accepted[m]++;
                        }
                    } else {
                        int32_t sa = pool[m].species[0], sb = pool[m].species[1];
                        int64_t na = 0, nb = 0;
                        for (int64_t k = 0; k < nc; k++) {
                            int32_t sp = c->species[c->cell_ids[k0 + k]];
                            na += (sp == sa);
                            nb += (sp == sb);
                        }
                        if (na == 0 || nb == 0 || sa == sb) continue;
                        int64_t ka = mulhi32(R[1], (uint32_t)na), kb = mulhi32(Q[0], (uint32_t)nb);
                        int64_t i = -1, j = -1;
                        for (int64_t k = 0; k < nc; k++) {
                            int64_t id = c->cell_ids[k0 + k];
                            if (c->species[id] == sa && ka-- == 0) i = id;
                            if (c->species[id] == sb && kb-- == 0) j = id;
                        }
                        const double *xi = c->xf + (size_t)i * d, *xj = c->xf + (size_t)j * d;
                        double e1 = cb_energy_excl(c, xi, sa, cand, ncand, i, j) +
                                    cb_energy_excl(c, xj, sb, cand, ncand, i, j);
                        double e2 = cb_energy_excl(c, xi, sb, cand, ncand, i, j) +
                                    cb_energy_excl(c, xj, sa, cand, ncand, i, j);
                        double de = e2 - e1;
                        if (c->temperature * lnu < -de) {   /* accept iff T ln u < -dE (DESIGN.md "schedule") */
                            c->species[i] = sb;
                            c->species[j] = sa;
                            c->e_fix += (int64_t)(de * CB_FIX);
                            stats->accepted[m]++;
                        }
                    }
                }
            }
        }
    }
    free(cand);
}
