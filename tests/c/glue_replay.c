/* glue_replay.c -- the call sequences of julia/PMCHip.jl, replayed in C against libpmc_hip.so.
 *
 * Julia is not available in the build image, so the ccall glue cannot run there.  What CAN be checked without Julia:
 *  (1) that the isbits structs PMCHip.jl declares (PairModel, Params, CMove, Stats) have exactly the layout of
 *      include/pmc.h  -- `glue_replay --layout` prints size / offset of every field as the C compiler lays them out;
 *      tests/test_julia_glue.py derives the same numbers from the field list in the .jl file (Julia lays isbits structs
 *      out like C) and from the ctypes mirror, and compares all three;
 *  (2) that every call sequence the glue performs is accepted by the library and returns what the reference semantics
 *      require -- `glue_replay <input> <output>` performs them in the glue's order with the glue's argument shapes
 *      (1-based ids, AoS positions, Ref-style out parameters) and writes the results for comparison with the oracle.
 * Each block names the PMCHip.jl function it replays.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pmc.h"

#define CHECK(call)                                                                            \
    do {                                                                                       \
        int rc_ = (call);                                                                      \
        if (rc_ != 0) {                                                                        \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, pmc_last_error(h));                  \
            return 2;                                                                          \
        }                                                                                      \
    } while (0)

static void layout(void)
{
#define F(S, f) printf(#S "." #f " %zu %zu\n", offsetof(S, f), sizeof(((S *)0)->f))
    printf("pmc_pair_model.sizeof %zu\n", sizeof(pmc_pair_model));
    F(pmc_pair_model, kind); F(pmc_pair_model, n); F(pmc_pair_model, eps); F(pmc_pair_model, eps4); F(pmc_pair_model, sigma);
    F(pmc_pair_model, sigma2); F(pmc_pair_model, rcut); F(pmc_pair_model, rcut2); F(pmc_pair_model, shift);
    F(pmc_pair_model, ndiv2_i); F(pmc_pair_model, ndiv2_f); F(pmc_pair_model, C0); F(pmc_pair_model, C2_s2); F(pmc_pair_model, C4_s4);
    printf("pmc_params.sizeof %zu\n", sizeof(pmc_params));
    F(pmc_params, N); F(pmc_params, d); F(pmc_params, nspecies); F(pmc_params, box); F(pmc_params, cell); F(pmc_params, ncells);
    F(pmc_params, temperature); F(pmc_params, density); F(pmc_params, models); F(pmc_params, device); F(pmc_params, chain); F(pmc_params, seed);
    printf("pmc_move.sizeof %zu\n", sizeof(pmc_move));
    F(pmc_move, kind); F(pmc_move, species); F(pmc_move, probability); F(pmc_move, sigma);
    printf("pmc_stats.sizeof %zu\n", sizeof(pmc_stats));
    F(pmc_stats, attempted); F(pmc_stats, accepted); F(pmc_stats, energy); F(pmc_stats, energy_fix); F(pmc_stats, kernel_ms);
    F(pmc_stats, sweep_kernel_ms); F(pmc_stats, sweep_kernel_launches); F(pmc_stats, pair_tests); F(pmc_stats, pairs_in_range);
    F(pmc_stats, null_moves); F(pmc_stats, dense_evals); F(pmc_stats, staged_candidates); F(pmc_stats, written_back);
#undef F
}

/* sum(::Vector{Float64}) as the oracle restates it (oracle/pmc_oracle.c pairwise_sum: blocks of 1024, sequential leaves) */
static double pairwise_sum(const double *v, int64_t lo, int64_t hi)
{
    if (hi - lo < 1024) {
        double a = v[lo];
        for (int64_t k = lo + 1; k <= hi; k++) a += v[k];
        return a;
    }
    const int64_t mid = lo + ((hi - lo) >> 1);
    return pairwise_sum(v, lo, mid) + pairwise_sum(v, mid + 1, hi);
}

int main(int argc, char **argv)
{
    pmc_system *h = NULL;
    if (argc == 2 && strcmp(argv[1], "--layout") == 0) { layout(); return 0; }
    if (argc != 3) { fprintf(stderr, "usage: glue_replay --layout | glue_replay <input.bin> <output.txt>\n"); return 1; }

    /* input: int64 N, int32 d, int32 S, double rho, double T, double rcut, N*d doubles, N int32 species */
    FILE *fi = fopen(argv[1], "rb");
    if (!fi) { perror(argv[1]); return 1; }
    int64_t N; int32_t d, S; double rho, T, rcut;
    if (fread(&N, 8, 1, fi) != 1 || fread(&d, 4, 1, fi) != 1 || fread(&S, 4, 1, fi) != 1 || fread(&rho, 8, 1, fi) != 1 ||
        fread(&T, 8, 1, fi) != 1 || fread(&rcut, 8, 1, fi) != 1) return 1;
    double *pos = malloc((size_t)N * d * sizeof(double));
    int32_t *sp = malloc((size_t)N * sizeof(int32_t));
    if (fread(pos, sizeof(double), (size_t)N * d, fi) != (size_t)N * d || fread(sp, 4, (size_t)N, fi) != (size_t)N) return 1;
    fclose(fi);
    FILE *fo = fopen(argv[2], "w");
    if (!fo) { perror(argv[2]); return 1; }

    /* ---- HIPLinkedList(box, rcut, N) + build_neighbour_list!: Params by reference, models row-major [i][j] ---------- */
    pmc_pair_model *models = calloc((size_t)S * S, sizeof *models);
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++)
            CHECK(pmc_model_lennard_jones(1.0 + 0.25 * (i != j), 1.0 - 0.1 * (i + j) / 2.0, rcut, 1, &models[i * S + j]));
    pmc_params p;
    memset(&p, 0, sizeof p);
    p.N = N; p.d = d; p.nspecies = S; p.temperature = T; p.density = rho; p.models = models;
    p.device = 0; p.chain = 3; p.seed = 77;
    for (int a = 0; a < 3; a++) { p.box[a] = 1.0; p.cell[a] = 1.0; p.ncells[a] = 1; }      /* pad3(v, fill) */
    double box1;
    CHECK(pmc_box_from_density(N, rho, d, &box1));
    for (int a = 0; a < d; a++) p.box[a] = box1;
    CHECK(pmc_linked_list_geometry(d, p.box, rcut, p.cell, p.ncells));
    /* System(...) in the reference's own order (src/atoms.jl:44-56):
     *   :44 energy = Vector{T}(undef, 1)                          -- NOTHING to hand over yet
     *   :47 list_type(box, maxcut, N)                             -- HIPLinkedList ctor (geometry above)
     *   :50 build_neighbour_list!(system)                         -- create + upload + build; no pmc_set_energy here
     *   :51 [compute_energy_particle(system, i, nl) for i]        -- N calls through the glue's overload
     *   :52 sum(local_energy) / 2                                 -- on the host (Julia's pairwise sum)
     *   :56 system.energy[1] = energy                             -- a plain store in a Julia Vector: the glue never sees it
     * The library's running energy is garbage-in-waiting until the first handover!() (top of every entry point below). */
    CHECK(pmc_create(&p, &h));
    CHECK(pmc_upload(h, pos, sp));
    CHECK(pmc_build_cells(h));
    double *local = malloc((size_t)N * sizeof(double));
    for (int64_t i = 1; i <= N; i++) CHECK(pmc_energy_particle(h, i, &local[i - 1]));
    double system_energy = pairwise_sum(local, 0, N - 1) / 2.0;     /* the Julia-side system.energy[1] from here on */
    free(local);
    fprintf(fo, "energy0 %.17g\n", system_energy);
#define HANDOVER() CHECK(pmc_set_energy(h, system_energy))          /* PMCHip.handover!(system) */

    /* ---- getproperty(L, :cs) ------------------------------------------------------------------------------------ */
    int64_t *cs = malloc((size_t)N * sizeof(int64_t));
    CHECK(pmc_get_cells(h, cs));
    fprintf(fo, "cs");
    for (int64_t i = 0; i < N; i++) fprintf(fo, " %lld", (long long)cs[i]);
    fprintf(fo, "\n");

    /* ---- (L::HIPLinkedList)(system, i): the LinkedIterator's ids, first with the glue's 1024-entry buffer ------------ */
    for (int64_t i = 1; i <= N; i += N / 3 + 1) {
        int64_t nb[1024], nn = 0;
        CHECK(pmc_neighbours_of(h, i, nb, 1024, &nn));
        fprintf(fo, "neighbours %lld", (long long)i);
        for (int64_t k = 0; k < nn; k++) fprintf(fo, " %lld", (long long)nb[k]);
        fprintf(fo, "\n");
    }
    {   /* a buffer that is too small: status != 0, n = the needed size, and the retry with that size succeeds */
        int64_t tiny[4], nn = 0;
        if (pmc_neighbours_of(h, 1, tiny, 4, &nn) == 0 || nn <= 4) { fprintf(stderr, "pmc_neighbours_of accepted a short buffer\n"); return 2; }
        int64_t *big = malloc((size_t)nn * sizeof(int64_t)), n2 = 0;
        CHECK(pmc_neighbours_of(h, 1, big, nn, &n2));
        fprintf(fo, "neighbours_retry %lld %lld\n", (long long)nn, (long long)n2);
        free(big);
    }

    /* ---- compute_energy_particle(system, i, L): 1-based id, Ref{Float64} out ---------------------------------------- */
    for (int64_t i = 1; i <= N; i += N / 7 + 1) {
        double e = 0.0;
        CHECK(pmc_energy_particle(h, i, &e));
        fprintf(fo, "e_particle %lld %.17g\n", (long long)i, e);
    }

    /* ---- perform_action!(Displacement) / revert_action!: delta as a d-vector, (e1, e2) returned, de stored ---------- */
    {
        const int64_t i = 5;
        double delta[3] = {0.031, -0.017, 0.044}, e1 = 0, e2 = 0, de = 0, en = 0;
        HANDOVER();                                                 /* the FIRST use of the running energy after System() */
        CHECK(pmc_displacement_perform(h, i, delta, &e1, &e2, &de));
        system_energy += de;                                        /* the reference's wrapper, on the host (src/moves.jl:11-20) */
        CHECK(pmc_get_energy(h, &en));
        fprintf(fo, "disp_perform %.17g %.17g %.17g %.17g %.17g\n", e1, e2, de, en, system_energy);
        double minus[3] = {-delta[0], -delta[1], -delta[2]};       /* invert_action! negated it (src/moves.jl:88-90) */
        HANDOVER();
        CHECK(pmc_displacement_revert(h, i, minus, de));
        system_energy -= de;                                        /* src/moves.jl:79, in the glue's revert_action! */
        CHECK(pmc_get_energy(h, &en));
        fprintf(fo, "disp_revert %.17g %.17g\n", en, system_energy);
    }

    /* ---- perform_action!(DiscreteSwap) / revert_action! ------------------------------------------------------------- */
    if (S > 1) {
        int64_t i = 0, j = 0;
        for (int64_t k = 0; k < N && (!i || !j); k++) { if (sp[k] == 1 && !i) i = k + 1; if (sp[k] == 2 && !j) j = k + 1; }
        double e1 = 0, e2 = 0, de = 0, en = 0;
        HANDOVER();
        CHECK(pmc_swap_perform(h, i, j, &e1, &e2, &de));
        system_energy += de;
        CHECK(pmc_get_energy(h, &en));
        fprintf(fo, "swap_perform %lld %lld %.17g %.17g %.17g %.17g %.17g\n", (long long)i, (long long)j, e1, e2, de, en, system_energy);
        HANDOVER();
        CHECK(pmc_swap_revert(h, j, i, de));                       /* the inverted action swaps the roles of i and j */
        system_energy -= de;                                        /* src/moves.jl:206 */
        CHECK(pmc_get_energy(h, &en));
        fprintf(fo, "swap_revert %.17g %.17g\n", en, system_energy);
    }

    /* ---- make_step!(sim, HIPMetropolis): one scheduler tick = pmc_sweep(h, mode, pool, npool, 1, Ref{Stats}) --------- */
    pmc_move pool[2];
    memset(pool, 0, sizeof pool);
    pool[0].kind = PMC_MOVE_DISPLACEMENT; pool[0].probability = S > 1 ? 0.8 : 1.0; pool[0].sigma = 0.05;
    pool[1].kind = PMC_MOVE_SWAP; pool[1].species[0] = 1; pool[1].species[1] = 2; pool[1].probability = 0.2;
    const int npool = S > 1 ? 2 : 1;
    for (int mode = 0; mode <= 1; mode++) {                         /* 0 = the reference's sequential chain, 1 = checkerboard */
        pmc_stats st;
        memset(&st, 0, sizeof st);
        HANDOVER();
        CHECK(pmc_sweep(h, mode, pool, npool, 1, &st));
        system_energy = st.energy;                                  /* make_step!: system.energy[1] = st[].energy */
        fprintf(fo, "sweep %d %lld %lld %lld %lld %.17g %lld\n", mode, (long long)st.attempted[0], (long long)st.accepted[0],
                (long long)st.attempted[1], (long long)st.accepted[1], st.energy, (long long)st.energy_fix);
    }

    /* ---- sweep_many!: begin on every handle, then end on every handle (here: one handle, two rounds) ----------------- */
    for (int round = 0; round < 2; round++) {
        pmc_stats st;
        HANDOVER();
        CHECK(pmc_sweep_begin(h, PMC_CHECKERBOARD, pool, npool, 2));
        CHECK(pmc_sweep_end(h, &st));
        system_energy = st.energy;
        fprintf(fo, "sweep_many %lld %lld %.17g %lld\n", (long long)st.attempted[0], (long long)st.accepted[0], st.energy, (long long)st.energy_fix);
    }

    /* ---- refresh!: positions straight into the Vector{SVector{d,Float64}} buffer, species as Int32, energy ------------ */
    double en = 0;
    CHECK(pmc_download(h, pos, sp));
    CHECK(pmc_get_energy(h, &en));
    fprintf(fo, "refresh_energy %.17g\n", en);
    fprintf(fo, "pos");
    for (int64_t k = 0; k < N * d; k++) fprintf(fo, " %.17g", pos[k]);
    fprintf(fo, "\nspecies");
    for (int64_t k = 0; k < N; k++) fprintf(fo, " %d", sp[k]);
    fprintf(fo, "\n");

    /* ---- set_temperature! -------------------------------------------------------------------------------------------- */
    CHECK(pmc_set_temperature(h, 1.5 * T));
    double T2 = 0;
    CHECK(pmc_get_temperature(h, &T2));
    fprintf(fo, "temperature %.17g\n", T2);

    /* ---- EnergyBias overloads: sample_action! / log_proposal_density on the device ------------------------------------- */
    if (S > 1) {
        int64_t bi = 0, bj = 0;
        double lq = 0, lq2 = 0;
        CHECK(pmc_energy_bias_sample(h, 1, 2, 0.3, -0.2, 0.25, 0.75, &bi, &bj, &lq));
        CHECK(pmc_energy_bias_logq(h, 1, 2, 0.3, -0.2, bi, bj, &lq2));
        fprintf(fo, "energy_bias %lld %lld %.17g %.17g\n", (long long)bi, (long long)bj, lq, lq2);
    }

    /* ---- sweep_many! and replica_exchange!: a second chain of the same shape, both through the shared-launch entry ------ */
    {
        pmc_system *h2 = NULL;
        pmc_params p2 = p;
        p2.chain = 4;
        p2.temperature = 1.05 * T;
        CHECK(pmc_create(&p2, &h2));
        CHECK(pmc_upload(h2, pos, sp));       /* pos / sp: the first chain's current state (refresh! above) */
        double e2 = 0;
        CHECK(pmc_init_energy(h2, &e2));      /* (the second chain's System(): the same steps as above, in one call) */
        CHECK(pmc_set_energy(h2, e2));        /* handover!(systems[2]) */
        HANDOVER();                           /* handover!(systems[1]) */
        CHECK(pmc_set_temperature(h, T));
        pmc_system *hs[2] = {h, h2};
        pmc_stats st2[2];
        CHECK(pmc_sweep_many(hs, 2, PMC_CHECKERBOARD, pool, npool, 2, st2));
        for (int r = 0; r < 2; r++)
            fprintf(fo, "many %d %lld %lld %lld\n", r, (long long)st2[r].attempted[0], (long long)st2[r].accepted[0], (long long)st2[r].energy_fix);
        double ladder[2] = {T, 1.05 * T}, Ta = 0, Tb = 0;
        int32_t label[2] = {0, 1}, nacc = 0;
        CHECK(pmc_replica_exchange(hs, 2, ladder, 99, 0, label, &nacc));
        CHECK(pmc_get_temperature(h, &Ta));
        CHECK(pmc_get_temperature(h2, &Tb));
        fprintf(fo, "exchange %d %d %d %.17g %.17g\n", label[0], label[1], nacc, Ta, Tb);
        CHECK(pmc_destroy(h2));
    }

    /* ---- the AbstractRNG wrapper (PMCHip.SequentialWords): the words of sequential step k, host side ------------------ */
    for (uint64_t step = 0; step < 3; step++) {
        uint32_t R[4], Q[4];
        CHECK(pmc_sequential_words(77, 3, step, R, Q));
        fprintf(fo, "words %llu %u %u %u %u %u %u %u %u\n", (unsigned long long)step, R[0], R[1], R[2], R[3], Q[0], Q[1], Q[2], Q[3]);
    }
    CHECK(pmc_destroy(h));
    h = NULL;
    fclose(fo);
    free(pos); free(sp); free(cs); free(models);
    return 0;
}
