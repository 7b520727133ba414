// pmc_seq_host.h -- host side of the fast reference-order engine (pmc_seq.h); included by pmc_hip.hip after the handle.
#pragma once

static int seq_rows_cap(const pmc_system *h)
{
    const double mean = (double)h->N / (double)h->ncell_ref;
    return std::max(h->rows_min_cap, std::max(32, (int)(3.0 * mean) + 32));   // (rows_min_cap: a row overflowed mid-chain before)
}

// A cell row overflowed in the middle of a chain (err bit 0): the reference's vectors simply grow, the device rows are fixed for
// the length of a call, so the call fails -- but the next one gets rows of twice the capacity (seq_rows_cap honours
// rows_min_cap; CellList / VerletList rows are re-allocated by pmc_build_cells, which doubles until the fullest cell fits and
// then once more).
static void seq_rows_overflowed(pmc_system *h, int err)
{
    if (!(err & 1)) return;
    h->rows_min_cap = std::max(h->rows_min_cap, 2 * std::max(h->cl_cap, 16));
    if (h->list_type == 1 && h->d_cl_items) { (void)hipFree(h->d_cl_items); h->d_cl_items = nullptr; h->cl_cap = 0; }
}

// the cell rows of a LinkedList: allocated on the first fast sequential call, rebuilt by every pmc_build_cells
static int seq_build_rows(pmc_system *h)
{
    const int tot = std::max(h->N, h->ncell_ref);
    for (int attempt = 0; attempt < 6; attempt++) {
        HIPCHK(h, hipMemsetAsync(h->d_ref_err, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_ref_cellvec, dim3(nblk(tot, 256)), dim3(256), 0, h->stream, h->N, h->d_cell_of, h->d_start_ref,
                           h->d_ids_ref, h->ncell_ref, h->d_cl_items, h->cl_cap, h->d_ccount, h->d_ref_err);
        HIPCHK(h, hipGetLastError());
        int e = 0;
        HIPCHK(h, hipMemcpyAsync(&e, h->d_ref_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (!(e & 1)) return PMC_OK;
        // a cell holds more particles than a row: twice the capacity and again (k_ref_cellvec clamped ccount, k_ref_links
        // must restore it)
        (void)hipFree(h->d_cl_items);
        h->d_cl_items = nullptr;
        h->cl_cap *= 2;
        if ((double)h->ncell_ref * h->cl_cap * sizeof(int) > 64e9) break;
        HIPCHK(h, hipMalloc(&h->d_cl_items, (size_t)h->ncell_ref * h->cl_cap * sizeof(int)));
        hipLaunchKernelGGL(k_ref_links, dim3(nblk(tot, 256)), dim3(256), 0, h->stream, h->d_cell_of, h->N, h->d_start_ref,
                           h->d_ids_ref, h->ncell_ref, h->d_head, h->d_list, h->d_ccount);
    }
    return fail(h, PMC_ERR_UNSUPPORTED, "sequential engine: a cell holds more particles than the row capacity allows");
}

// rows for a LinkedList that is already built (and whose chains moves may have reordered): from head / list, not from positions
static int seq_alloc_rows(pmc_system *h, int cap)
{
    if (!h->d_ref_err) {
        HIPCHK(h, hipMalloc(&h->d_ref_err, 16));
        HIPCHK(h, hipMemset(h->d_ref_err, 0, 16));
    }
    for (int attempt = 0; attempt < 6; attempt++) {
        if (h->d_cl_items) (void)hipFree(h->d_cl_items);
        h->d_cl_items = nullptr;
        h->cl_cap = cap;
        HIPCHK(h, hipMalloc(&h->d_cl_items, (size_t)h->ncell_ref * cap * sizeof(int)));
        if (!h->cells_built) return PMC_OK;            // the next pmc_build_cells fills the rows
        HIPCHK(h, hipMemsetAsync(h->d_ref_err, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_sq_rows_from_chains, dim3(nblk(h->ncell_ref, 128)), dim3(128), 0, h->stream, h->ncell_ref, h->d_head,
                           h->d_list, h->d_cl_items, cap, h->d_ccount, h->d_ref_err);
        HIPCHK(h, hipGetLastError());
        int e = 0;
        HIPCHK(h, hipMemcpyAsync(&e, h->d_ref_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (!(e & 1)) return PMC_OK;
        cap *= 2;
        if ((double)h->ncell_ref * cap * sizeof(int) > 64e9) break;
    }
    return fail(h, PMC_ERR_UNSUPPORTED, "sequential engine: a cell holds more particles than the row capacity allows");
}

static bool seq_pool_ok(const pmc_move *pool, int npool)
{
    for (int m = 0; m < npool; m++)
        if (pool[m].kind == PMC_MOVE_SWAP && pool[m].species[0] == pool[m].species[1]) return false;   // i == j picks: the literal loop
    return true;
}

static bool seq_fast_applicable(const pmc_system *h, const pmc_move *pool, int npool)
{
    if (h->seq_engine == 1) return false;
    if (h->list_type != 1 && h->list_type != 2) return false;      // EmptyList: every step reads everything; VerletList: own update rule
    if (!seq_pool_ok(pool, npool)) return false;
    if ((double)h->ncell_ref * seq_rows_cap(h) * sizeof(int) > 32e9) return false;   // dilute giants: rows would not fit
    return true;
}

static int seq_ensure(pmc_system *h, int K)
{
    if (!h->d_ref_err) {
        HIPCHK(h, hipMalloc(&h->d_ref_err, 16));
        HIPCHK(h, hipMemset(h->d_ref_err, 0, 16));
    }
    if (h->list_type == 1 && !h->d_cl_items) {
        const int rc = seq_alloc_rows(h, seq_rows_cap(h));
        if (rc) return rc;
    }
    if (!h->d_sq_ctl) {
        HIPCHK(h, hipMalloc(&h->d_sq_cp, (size_t)2 * SQ_KW * sizeof(unsigned long long)));
        HIPCHK(h, hipMalloc(&h->d_sq_cellhead, (size_t)h->ncell_ref * sizeof(int)));
        HIPCHK(h, hipHostMalloc(&h->h_sq_ctl, 2 * sizeof(SqCtl)));      // [0]: the device's block read back; [1]: words on their way to it
        HIPCHK(h, hipMalloc(&h->d_sq_ctl, sizeof(SqCtl)));
    }
    // what scales with the window: slot records, the conflict matrix's rows, the random parts (128 windows' worth)
    if (K > h->sq_cap_K) {
        void *old[] = {h->d_sq_slots[0], h->d_sq_slots[1], h->d_sq_gen, h->d_sq_M, h->d_sq_moveto, h->d_sq_cellnext};
        for (void *q : old) if (q) (void)hipFree(q);
        h->d_sq_slots[0] = h->d_sq_slots[1] = nullptr; h->d_sq_gen = nullptr; h->d_sq_M = nullptr; h->d_sq_moveto = nullptr; h->d_sq_cellnext = nullptr;
        h->sq_cap_K = 0;
        // (512 windows' worth for small windows: a batch between two visits of the host is then ~20 ms of rounds, and the
        // copies, folds and launches per chain and batch -- a third of the time for 128 chains -- shrink with it)
        int gen = 8192;
        while (gen < 512 * K && gen < SQ_GEN_MAX) gen *= 2;
        HIPCHK(h, hipMalloc(&h->d_sq_slots[0], (size_t)K * sizeof(SqSlot)));
        HIPCHK(h, hipMalloc(&h->d_sq_slots[1], (size_t)K * sizeof(SqSlot)));
        HIPCHK(h, hipMalloc(&h->d_sq_gen, (size_t)gen * sizeof(SqGen)));
        HIPCHK(h, hipMalloc(&h->d_sq_M, (size_t)K * SQ_KW * sizeof(unsigned long long)));
        HIPCHK(h, hipMalloc(&h->d_sq_moveto, (size_t)K * 4 * sizeof(int)));
        HIPCHK(h, hipMalloc(&h->d_sq_cellnext, (size_t)K * 2 * sizeof(int)));
        h->sq_cap_K = K;
        h->sq_gen = gen;
    }
    // the log ring: up to 2048 windows' worth of steps (four batches) between two folds (16 B per step, device + pinned host)
    int ring = 16384;
    long long ringx = 2048;                                              // (four batches of 512 rounds: see the pipelined order below)
    if (const char *e = getenv("PMC_SEQ_RINGX")) ringx = atoll(e);       // (A/B of the batch length)
    while (ring < ringx * K && ring < SQ_RING_MAX) ring *= 2;
    if (ring > h->sq_ring) {
        if (h->d_sq_log) (void)hipFree(h->d_sq_log);
        if (h->h_sq_log) (void)hipHostFree(h->h_sq_log);
        h->d_sq_log = nullptr; h->h_sq_log = nullptr; h->sq_ring = 0;
        HIPCHK(h, hipMalloc(&h->d_sq_log, (size_t)ring * sizeof(SqLog)));
        HIPCHK(h, hipHostMalloc(&h->h_sq_log, (size_t)ring * sizeof(SqLog)));
        h->sq_ring = ring;
    }
    return PMC_OK;
}

static int seq_window(const pmc_system *h)
{
    int K = h->seq_window;
    if (K <= 0) {
        // Round 5 (distance-based conflicts, a round of a small window as one launch): small systems keep the windows that run
        // as one launch per round -- 64 below ~400 cells (N = 4096: 19.8 steps per 24.6 us round; 128 commits no more), 128
        // below ~1500 -- and from there on the window is what the admission can fill, ~16 cells' worth of neighbourhoods,
        // with the commit / admission / preparation as launches of their own.  Measured (3-D LJ liquid, 10^6 moves/s, window
        // 128 / 256 / 512 / 1024): N = 32768 3.4 / 3.2 / 3.5 / 3.6; N = 110592 - / 5.4 / 8.7 / 8.9; N = 10^6: 24.2 at 2048,
        // 23.1 at 4096.  (Round 4, cell-rule conflicts: N = 10^6 1.45 / 1.58 / 1.51e7 with 1024 / 2048 / 4096; N = 4.1e6
        // 1.73 / 2.49 / 2.65e7; N = 8e6 1.74 / 2.61 / 2.98e7 -- hence 4096 from ~4e6 particles on.)
        long long nn = 1;
        for (int a = 0; a < h->D; a++) nn *= 3;
        const long long nbh = h->ncell_ref / nn;             // disjoint neighbourhoods in the box
        if (nbh < 15) K = 64;
        else if (nbh < 56) K = 128;
        else K = (2 * nbh >= 16384) ? SQ_KMAX : (int)std::min<long long>(16 * nbh, 2048);
        if (const char *e = getenv("PMC_SEQ_WINDOW")) K = std::min(atoi(e), SQ_KMAX);
    }
    int P = 64;                                             // a power of two (the fill kernel's hash table has 4 K entries)
    while (P < K && P < SQ_KMAX) P *= 2;
    return P;
}

// Arianna.perform_action! / revert_action! keep system.energy[1] on the host (src/moves.jl:11-20,79,206): `+= de` for
// every step with finite energies, `-= de` again when the step is rejected, in step order.  The device logs (de, accepted)
// per step; this folds the contiguous finished prefix.  Returns the new prefix.
static unsigned long long seq_fold(const SqLog *log, unsigned long long mask, unsigned long long from, unsigned long long to,
                                   double *energy, unsigned long long *att, unsigned long long *acc)
{
    double E = *energy;
    unsigned long long g = from;
    for (; g < to; g++) {
        const SqLog &r = log[g & mask];
        if (!(r.flag & SQL_DONE) || r.tag != (uint32_t)g) break;
        const int m = (r.flag >> 4) & 7;
        att[m]++;
        if (r.flag & SQL_NULL) continue;
        if (r.flag & SQL_ACCEPT) acc[m]++;
        if (r.flag & SQL_FINITE) {
            E += r.de;
            if (!(r.flag & SQL_ACCEPT)) E -= r.de;
        }
    }
    *energy = E;
    return g;
}

// mc_sweep!(system, pool, rng; mc_steps = nsweeps N) as the reference runs it -- one particle per step, every step seeing
// the state all earlier steps left (src/moves.jl:11-20,57-81,187-207) -- in speculative windows (pmc_seq.h).  One call is a
// sequence of BATCHES of rounds.  The chains of a pmc_sweep_many call form a GROUP: they share every launch of a round (the
// kernels take the chain from blockIdx.y and their arguments from a device table), on the first chain's stream; a chain that
// has finished its steps idles in the launches that are left.  pmc_sweep is a group of one.
struct SeqRun {
    pmc_system *h = nullptr;
    SqArgs s;
    SqCtl c0, c;
    unsigned long long ring = 0, gen_hi = 0, next_known = 0, prefix = 0;
    unsigned long long att[8] = {0}, acc[8] = {0};
    double energy = 0.0, per_round = 0.0;
    long long guard = 0, nsteps = 0;
    int64_t nsweeps = 0;
    bool done = false, failed = false;
};

struct SeqGroup {
    std::vector<SeqRun> runs;
    pmc_system *lead = nullptr;       // its stream carries the group's launches, its events time them
    SqArgs *d_tab = nullptr;          // device table of the chains' arguments
    int n = 0, K = 0;
    size_t lds = 0, lds_fill = 0;
    const void *k_eval = nullptr;     // k_sq_round_a for the group's dimension and model kind
    const void *k_fused = nullptr;    // k_sq_round_fused likewise
    int nw = SQ_EW;                   // wavefronts per evaluated step (PMC_SEQ_EW=1: one, the A/B partner)
    bool fused = false;               // windows of <= SQ_FUSED_KMAX: a round is one launch (evaluate; the last workgroup commits, admits, prepares)
    bool primed = false;              // ... after the first window of the call has been admitted and prepared (one k_sq_tail)
    long long rounds_last = 0, rounds_folded = 0;     // rounds of the batch enqueued last / of the batch being folded
    hipGraphExec_t graph = nullptr;   // SQ_GRAPH_ROUNDS rounds, captured once per call (a group of one caches it on the handle)
    bool own_graph = false;
    ~SeqGroup()
    {
        if (d_tab && n > 1) (void)hipFree(d_tab);
        if (graph && own_graph) (void)hipGraphExecDestroy(graph);
    }
};

#define SQ_GRAPH_ROUNDS 32

static void seq_launch_round(const SeqGroup &g)
{
    pmc_system *h = g.lead;
    const int K = g.K, n = g.n;
    const SqArgs *tab = g.d_tab;
    void *kargs[1] = {(void *)&tab};
    if (g.fused) {
        (void)hipLaunchKernel(g.k_fused, dim3(K, n), dim3(64 * g.nw), kargs, std::max(g.lds, g.lds_fill), h->stream);
        return;
    }
    if (h->D == 3) {
        hipLaunchKernelGGL(k_sq_fill<3>, dim3(1, n), dim3(SQ_FILL_THREADS), g.lds_fill, h->stream, g.d_tab);
        hipLaunchKernelGGL(k_sq_prep<3>, dim3((2 * K + 255) / 256, n), dim3(256), 0, h->stream, g.d_tab);
        (void)hipLaunchKernel(g.k_eval, dim3(K, n), dim3(64 * g.nw), kargs, g.lds, h->stream);
        hipLaunchKernelGGL(k_sq_commit<3>, dim3(K, n), dim3(64), 0, h->stream, g.d_tab);
    } else {
        hipLaunchKernelGGL(k_sq_fill<2>, dim3(1, n), dim3(SQ_FILL_THREADS), g.lds_fill, h->stream, g.d_tab);
        hipLaunchKernelGGL(k_sq_prep<2>, dim3((2 * K + 255) / 256, n), dim3(256), 0, h->stream, g.d_tab);
        (void)hipLaunchKernel(g.k_eval, dim3(K, n), dim3(64 * g.nw), kargs, g.lds, h->stream);
        hipLaunchKernelGGL(k_sq_commit<2>, dim3(K, n), dim3(64), 0, h->stream, g.d_tab);
    }
}

// SQ_GRAPH_ROUNDS rounds as one graph launch: the arguments of a round never change within a call (which window buffer is
// current and how far the random parts reach live in the device's control blocks), and four kernels per round enqueued one
// by one cost the host more than the device needs to run them
static int seq_graph(SeqGroup &g)
{
    pmc_system *h = g.lead;
    if (g.graph) return PMC_OK;
    const int graph_key = (g.fused ? -g.K : g.K) * 16 + g.nw;
    if (g.n == 1 && h->sq_graph && h->sq_graph_K == graph_key && memcmp(&h->sq_graph_args, &g.runs[0].s, sizeof(SqArgs)) == 0) {
        g.graph = h->sq_graph;
        return PMC_OK;
    }
    hipGraph_t gr = nullptr;
    hipGraphExec_t ex = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    for (int q = 0; q < SQ_GRAPH_ROUNDS; q++) seq_launch_round(g);
    HIPCHK(h, hipStreamEndCapture(h->stream, &gr));
    const hipError_t e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    (void)hipGraphDestroy(gr);
    if (e != hipSuccess) return fail(h, PMC_ERR_HIP, "hipGraphInstantiate failed");
    g.graph = ex;
    if (g.n == 1) {                                          // kept with the handle: the next call with the same arguments reuses it
        if (h->sq_graph) (void)hipGraphExecDestroy(h->sq_graph);
        h->sq_graph = ex;
        h->sq_graph_args = g.runs[0].s;
        h->sq_graph_K = graph_key;
    } else {
        g.own_graph = true;
    }
    return PMC_OK;
}

static int seq_run_begin(SeqRun &r, pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, int K, hipStream_t stream)
{
    r.h = h;
    int rc = seq_ensure(h, K);
    if (rc) return rc;
    if (!h->cells_built) {
        rc = pmc_build_cells(h);
        if (rc) return rc;
    }
    if (int rs = refresh_stale_energy(h)) return rs;
    r.nsweeps = nsweeps;
    r.nsteps = nsweeps * (int64_t)h->N;                    // sweepstep = N, src/ParticlesMC.jl:246
    r.ring = (unsigned long long)h->sq_ring;
    SqArgs &s = r.s;
    memset(&s, 0, sizeof s);
    s.a = ref_args(h);
    for (int q = 0; q < 3; q++) { s.invbox[q] = 1.0 / s.a.box[q]; s.invcell[q] = (q < h->D) ? 1.0 / s.a.cell[q] : 1.0; }
    s.conf_r2 = 0.0;                                       // PMC_SEQ_CELLRULE=1: the cell rule for every pair of steps (the A/B partner)
    if (!getenv("PMC_SEQ_CELLRULE")) {
        for (const pmc_pair_model &pm : h->models) s.conf_r2 = std::max(s.conf_r2, pm.rcut2);
        s.conf_r2 *= 1.0 + 0x1p-20;
    }
    s.slots[0] = h->d_sq_slots[0];
    s.slots[1] = h->d_sq_slots[1];
    s.gen = h->d_sq_gen;
    s.M = h->d_sq_M;
    s.cellhead = h->d_sq_cellhead;
    s.cellnext = h->d_sq_cellnext;
    s.cp = h->d_sq_cp;
    s.log = h->d_sq_log;
    s.ring_mask = r.ring - 1;
    s.gen_mask = (unsigned long long)h->sq_gen - 1;
    s.moveto = h->d_sq_moveto;
    s.cand = h->d_sq_moveto + h->sq_cap_K;
    s.ctl = h->d_sq_ctl;
    s.pool = make_pool(pool, npool);
    s.K = K;
    s.reverse = (h->list_type == 1) ? 1 : 0;
    s.ids_cap = h->nn * h->cl_cap;
    s.arith_nbr = 1;
    s.pk_ok = 1;
    for (int q = 0; q < h->D; q++) {
        if (h->p.ncells[q] < 3) s.arith_nbr = 0;
        if (h->p.ncells[q] > 65535) s.pk_ok = 0;
    }
    if (getenv("PMC_SEQ_NBRTABLE")) s.arith_nbr = 0;
    if (const char *e = getenv("PMC_SEQ_DBG")) s.dbg = atoi(e);
    memset(&r.c0, 0, sizeof r.c0);
    r.c0.next = r.c0.prefix = h->step_ctr;
    r.c0.gend = h->step_ctr + (unsigned long long)r.nsteps;
    r.c0.trunc = INT_MAX;
    r.c0.gen_hi = r.c0.next;
    PMC_TIME(r.c0.t_first = ~0ull;)
    *h->h_sq_ctl = r.c0;
    HIPCHK(h, hipStreamSynchronize(h->stream));            // earlier work on the chain's own stream
    HIPCHK(h, hipMemcpyAsync(h->d_sq_ctl, h->h_sq_ctl, sizeof(SqCtl), hipMemcpyHostToDevice, stream));
    HIPCHK(h, hipMemsetAsync(h->d_sq_log, 0, (size_t)r.ring * sizeof(SqLog), stream));
    HIPCHK(h, hipMemsetAsync(h->d_sq_cellhead, 0xff, (size_t)h->ncell_ref * sizeof(int), stream));   // (clean after every round; cheap insurance per call)
    HIPCHK(h, hipStreamSynchronize(stream));               // the pinned block is reused for the read-backs below
    r.gen_hi = r.next_known = r.c0.next;
    r.prefix = r.c0.prefix;
    r.energy = h->energy;
    r.per_round = 0.5 * K;                                  // steps a round finishes: learnt from the batches
    r.c = r.c0;
    r.done = r.prefix >= r.c0.gend;
    return PMC_OK;
}

static int seq_group_begin(SeqGroup &g, pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps)
{
    g.n = n;
    g.lead = hs[0];
    g.runs.resize((size_t)n);
    pmc_system *m = g.lead;
    int rc = set_device(m);
    if (rc) return rc;
    g.K = 0;
    for (int k = 0; k < n; k++) g.K = std::max(g.K, seq_window(hs[k]));
    for (int k = 0; k < n; k++) {
        rc = seq_run_begin(g.runs[k], hs[k], pool, npool, nsweeps, g.K, m->stream);
        if (rc) return rc;
    }
    int ids_cap = 0;
    for (int k = 0; k < n; k++) ids_cap = std::max(ids_cap, g.runs[k].s.ids_cap);
    for (int k = 0; k < n; k++) g.runs[k].s.ids_cap = ids_cap;     // one LDS layout for the launch
    g.nw = SQ_EW;
    if (const char *e = getenv("PMC_SEQ_EW")) g.nw = (atoi(e) == 1) ? 1 : SQ_EW;
    g.lds = sq_eval_lds_bytes(ids_cap, g.nw);                        // ids + the term buffers
    if (g.lds > 160 * 1024 - 256) return fail(m, PMC_ERR_UNSUPPORTED, "sequential engine: neighbourhood rows exceed the LDS");
    {
        // the evaluation kernel of a group whose chains all have pair models of ONE kind has that potential compiled in (PMC_SEQ_ANYKIND=1: the general kernel, for A/B)
        int ukind = -1;
        for (int k = 0; k < n; k++)
            for (const pmc_pair_model &pm : hs[k]->models) {
                if (ukind == -1) ukind = (int)pm.kind;
                else if ((int)pm.kind != ukind) ukind = 4;
            }
        if (ukind < 0 || ukind > 3 || getenv("PMC_SEQ_ANYKIND")) ukind = 4;
#define SQ_PICK5(KER, DD, WW) (ukind == 0 ? (const void *)KER<DD, 0, WW> : ukind == 1 ? (const void *)KER<DD, 1, WW> : \
                               ukind == 2 ? (const void *)KER<DD, 2, WW> : ukind == 3 ? (const void *)KER<DD, 3, WW> : (const void *)KER<DD, 4, WW>)
        if (g.nw == 1) {
            g.k_eval = (m->D == 3) ? SQ_PICK5(k_sq_round_a, 3, 1) : SQ_PICK5(k_sq_round_a, 2, 1);
            g.k_fused = (m->D == 3) ? SQ_PICK5(k_sq_round_fused, 3, 1) : SQ_PICK5(k_sq_round_fused, 2, 1);
        } else {
            g.k_eval = (m->D == 3) ? SQ_PICK5(k_sq_round_a, 3, SQ_EW) : SQ_PICK5(k_sq_round_a, 2, SQ_EW);
            g.k_fused = (m->D == 3) ? SQ_PICK5(k_sq_round_fused, 3, SQ_EW) : SQ_PICK5(k_sq_round_fused, 2, SQ_EW);
        }
#undef SQ_PICK5
    }
    // PMC_SEQ_NOFUSE=1: four launches per round whatever the window (the A/B partner; what windows beyond SQ_FUSED_KMAX run)
    g.fused = g.K <= SQ_FUSED_KMAX && !getenv("PMC_SEQ_NOFUSE");
    if (g.lds > 48 * 1024) {
        HIPCHK(m, hipFuncSetAttribute(g.k_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
    }
    g.lds_fill = (size_t)8 * g.K * sizeof(int);            // keys[4 K] + vals[4 K]
    if (g.fused && std::max(g.lds, g.lds_fill) > 48 * 1024)
        HIPCHK(m, hipFuncSetAttribute(g.k_fused, hipFuncAttributeMaxDynamicSharedMemorySize, (int)std::max(g.lds, g.lds_fill)));
    if (g.lds_fill > 48 * 1024) {
        if (m->D == 3) HIPCHK(m, hipFuncSetAttribute((const void *)k_sq_fill<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_fill));
        else HIPCHK(m, hipFuncSetAttribute((const void *)k_sq_fill<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_fill));
        if (m->D == 3) HIPCHK(m, hipFuncSetAttribute((const void *)k_sq_tail<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_fill));
        else HIPCHK(m, hipFuncSetAttribute((const void *)k_sq_tail<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_fill));
    }
    // the table of arguments: a group of one keeps it with the handle
    std::vector<SqArgs> tab((size_t)n);
    for (int k = 0; k < n; k++) tab[k] = g.runs[k].s;
    if (n == 1) {
        if (!m->d_sq_args) HIPCHK(m, hipMalloc(&m->d_sq_args, sizeof(SqArgs)));
        g.d_tab = m->d_sq_args;
    } else {
        HIPCHK(m, hipMalloc(&g.d_tab, (size_t)n * sizeof(SqArgs)));
    }
    HIPCHK(m, hipMemcpy(g.d_tab, tab.data(), (size_t)n * sizeof(SqArgs), hipMemcpyHostToDevice));
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    return PMC_OK;
}

static bool seq_group_done(const SeqGroup &g)
{
    for (const SeqRun &r : g.runs) if (!r.done && !r.failed) return false;
    return true;
}

// one batch of rounds for every chain of the group onto the lead's stream; returns without waiting
static int seq_group_enqueue(SeqGroup &g, bool pipelined = false)
{
    pmc_system *m = g.lead;
    const int K = g.K;
    // (pipelined: the host is still folding the batch before this one while it runs, so the device's idea of the folded prefix
    // lags by two batches: a quarter of the ring per batch keeps the admission front clear of it)
    const unsigned long long ring_div = pipelined ? 4ull : 2ull;
    int rc = set_device(m);
    if (rc) return rc;
    long long rounds = 1;
    for (SeqRun &r : g.runs) {
        if (r.done || r.failed) continue;
        pmc_system *h = r.h;
        // random parts ahead of the admission front (admitted steps have been copied out of the ring)
        const unsigned long long target = std::min<unsigned long long>(r.c0.gend, r.next_known + (unsigned long long)h->sq_gen);
        if (r.gen_hi < target) {
            unsigned long long g0 = r.gen_hi;
            while (g0 < target) {
                const int nn = (int)std::min<unsigned long long>(target - g0, 1u << 20);
                hipLaunchKernelGGL(k_sq_gen, dim3(nblk(nn, 256)), dim3(256), 0, m->stream, h->d_sq_gen, r.s.gen_mask, g0, nn, h->key[0], h->key[1], r.s.pool);
                g0 += (unsigned long long)nn;
            }
            r.gen_hi = target;
            HIPCHK(h, hipMemcpyAsync(&h->d_sq_ctl->gen_hi, &r.gen_hi, sizeof r.gen_hi, hipMemcpyHostToDevice, m->stream));
        }
        // steps still to decide: from the fold's position, or (pipelined: the fold lags) from the admission front plus a window
        const double left = pipelined ? (double)(r.c0.gend - r.next_known) + (double)K : (double)(r.c0.gend - r.prefix);
        long long want = (long long)(left / std::max(r.per_round, 1.0)) + 2;
        want = std::min<long long>(want, std::max<long long>(1, (long long)((r.gen_hi - r.next_known) / (unsigned long long)K)));
        want = std::min<long long>(want, (long long)(r.ring / (ring_div * K)));    // the log ring holds a batch and the window
        rounds = std::max(rounds, want);
    }
    for (SeqRun &r : g.runs)                                 // ... and no chain's ring may overflow in the batch
        if (!r.done && !r.failed) rounds = std::min<long long>(rounds, std::max<long long>(1, (long long)(r.ring / (ring_div * K))));
    g.rounds_last = rounds;
    if (g.fused && !g.primed) {                    // the first window of the call: nothing to commit, admitted and prepared
        g.primed = true;
        if (m->D == 3) hipLaunchKernelGGL(k_sq_tail<3>, dim3(1, g.n), dim3(SQ_TAIL_THREADS), g.lds_fill, m->stream, g.d_tab);
        else hipLaunchKernelGGL(k_sq_tail<2>, dim3(1, g.n), dim3(SQ_TAIL_THREADS), g.lds_fill, m->stream, g.d_tab);
    }
    long long q = 0;
    // PMC_SEQ_NOGRAPH=1: the rounds go launch by launch.  The profile scripts set it: on ROCm 7.2 hipGraphLaunch under
    // `rocprofv3 --kernel-trace` crashes inside the tool's dispatch callback for some of these graphs (windows of 64 and 1024 in
    // runs of >= 10^6 steps; never without the tool).  The library itself never looks for a profiler: what runs under one is
    // what runs without, unless the caller says otherwise.
    if (rounds >= SQ_GRAPH_ROUNDS && !getenv("PMC_SEQ_NOGRAPH")) {
        if (int rg = seq_graph(g)) return rg;
        for (; q + SQ_GRAPH_ROUNDS <= rounds; q += SQ_GRAPH_ROUNDS) HIPCHK(m, hipGraphLaunch(g.graph, m->stream));
    }
    for (; q < rounds; q++) seq_launch_round(g);
    HIPCHK(m, hipGetLastError());
    for (SeqRun &r : g.runs)
        if (!r.done && !r.failed)
            HIPCHK(r.h, hipMemcpyAsync(r.h->h_sq_ctl, r.h->d_sq_ctl, sizeof(SqCtl), hipMemcpyDeviceToHost, m->stream));
    return PMC_OK;
}

// waits for the batch, reads every chain's control block and enqueues the copies of its new log records; `all_admitted`: no
// chain has steps left to admit (what remains is at most a window per chain)
static int seq_group_wait(SeqGroup &g, bool &copies, bool &all_admitted)
{
    pmc_system *m = g.lead;
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(m, hipStreamSynchronize(m->stream));
    int first = PMC_OK;
    copies = false;
    all_admitted = true;
    for (SeqRun &r : g.runs) {
        if (r.done || r.failed) continue;
        pmc_system *h = r.h;
        r.c = *h->h_sq_ctl;
        if (r.c.err) {
            h->cells_built = false;
            h->energy_stale = true;
            h->state_invalid = true;        // positions are partly advanced: no caller-side energy belongs to them (pmc_set_energy)
            seq_rows_overflowed(h, r.c.err);
            r.failed = true;
            const int e = fail(h, PMC_ERR_STATE, (r.c.err & 1) ? "sequential engine: a cell row overflowed" : "sequential engine: a particle was not in its cell row");
            if (!first) first = e;
            continue;
        }
        // the log records of [prefix, next): at most two contiguous pieces of the ring
        if (r.c.next > r.prefix) {
            const unsigned long long lo = r.prefix & r.s.ring_mask, nrec = r.c.next - r.prefix;
            const unsigned long long n1 = std::min<unsigned long long>(nrec, r.ring - lo);
            HIPCHK(h, hipMemcpyAsync(h->h_sq_log + lo, h->d_sq_log + lo, (size_t)n1 * sizeof(SqLog), hipMemcpyDeviceToHost, m->stream));
            if (nrec > n1) HIPCHK(h, hipMemcpyAsync(h->h_sq_log, h->d_sq_log, (size_t)(nrec - n1) * sizeof(SqLog), hipMemcpyDeviceToHost, m->stream));
            copies = true;
        }
        r.next_known = r.c.next;
        if (r.c.next < r.c0.gend) all_admitted = false;
    }
    if (copies) HIPCHK(m, hipEventRecord(m->ev[2], m->stream));
    return first;
}

// folds every chain's copied log into its system.energy and counters (seq_fold) and tells the device how far
static int seq_group_fold(SeqGroup &g, bool copies)
{
    pmc_system *m = g.lead;
    int first = PMC_OK;
    if (copies) HIPCHK(m, hipEventSynchronize(m->ev[2]));
    for (SeqRun &r : g.runs) {
        if (r.done || r.failed) continue;
        pmc_system *h = r.h;
        const unsigned long long before = r.prefix;
        r.prefix = seq_fold(h->h_sq_log, r.s.ring_mask, r.prefix, r.c.next, &r.energy, r.att, r.acc);
        // (from pinned memory: with the next batch already on the stream this copy must not make the host wait for it)
        h->h_sq_ctl[1].prefix = r.prefix;
        HIPCHK(h, hipMemcpyAsync(&h->d_sq_ctl->prefix, &h->h_sq_ctl[1].prefix, sizeof r.prefix, hipMemcpyHostToDevice, m->stream));
        const double done = (double)(r.prefix - before);
        if (done > 0) r.per_round = std::max(1.0, done / (double)g.rounds_folded);
        if (done == 0 && ++r.guard > 8) {
            r.failed = true;
            const int e = fail(h, PMC_ERR_STATE, "sequential engine: no progress");
            if (!first) first = e;
            continue;
        }
        if (done > 0) r.guard = 0;
        r.done = r.prefix >= r.c0.gend;
    }
    return first;
}

static int seq_group_end(SeqGroup &g, pmc_stats *out)
{
    pmc_system *m = g.lead;
    int rc = set_device(m);
    if (rc) return rc;
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    HIPCHK(m, hipEventSynchronize(m->ev[1]));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    for (int k = 0; k < g.n; k++) {
        SeqRun &r = g.runs[k];
        pmc_system *h = r.h;
        h->mc_binned = false;
        h->e_fix_valid = false;
        h->img_valid = false;
        if (r.failed) continue;
        r.c.prefix = r.prefix;
        PMC_TIME({
            unsigned long long t[16];
            if (hipMemcpyFromSymbol(t, HIP_SYMBOL(g_sqt), sizeof t) == hipSuccess && t[9]) {
                const double ne = (double)t[9];
                fprintf(stderr, "evaluation of slot 0, %.0f evaluations; ticks each: ctl+slot %.0f to-sequence %.0f counts+rows %.0f ids->LDS %.0f chains %.0f gather %.0f terms %.0f (wait for the others %.0f chain %.0f) sums' end %.0f decide %.0f\n",
                        ne, t[0] / ne, t[1] / ne, t[2] / ne, t[3] / ne, t[4] / ne, t[5] / ne, t[6] / ne, t[10] / ne, t[11] / ne, t[7] / ne, t[8] / ne);
                memset(t, 0, sizeof t);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sqt), t, sizeof t);
            }
        })
        PMC_TIME(if (r.c.tacc[6]) {
            const double nl = (double)r.c.tacc[6];
            fprintf(stderr, "fused rounds: %.0f launches; ticks per launch: spread %.0f eval %.0f commit %.0f fill %.0f prep %.0f gap %.0f\n", nl,
                    r.c.tacc[0] / nl, r.c.tacc[1] / nl, r.c.tacc[2] / nl, r.c.tacc[3] / nl, r.c.tacc[4] / nl, r.c.tacc[5] / nl);
        })
        h->sq_last = r.c;
        h->sq_last_steps = (r.c.gend == r.c.next) ? (long long)r.nsteps - (long long)(r.c.gend - r.prefix) : 0;   // steps of THIS call folded
        h->energy = r.energy;
        h->step_ctr += (uint64_t)r.nsteps;
        h->sweep_ctr += (uint64_t)r.nsweeps;
        if (out) {
            pmc_stats &o = out[k];
            memset(&o, 0, sizeof o);
            for (int q = 0; q < PMC_MAX_MOVES; q++) { o.attempted[q] = (int64_t)r.att[q]; o.accepted[q] = (int64_t)r.acc[q]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = ms;
            o.sweep_kernel_launches = (int64_t)r.c.rounds;
            o.energy = h->energy;
        }
    }
    return PMC_OK;
}

static int sweep_sequential_fast_many(pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    SeqGroup g;
    int rc = seq_group_begin(g, hs, n, pool, npool, nsweeps);
    if (rc) return rc;
    int first = PMC_OK;
    // Batches of rounds; while steps are left to admit, the next batch is enqueued BEFORE the host copies and folds the log of the
    // one that has just finished (the copies are ordered ahead of it on the stream), so the device does not wait for the fold
    // (large windows only: N = 10^6 gains 3 %; with windows of 64 .. 512 a batch is a few hundred microseconds of copies and folds
    // per several milliseconds of rounds, and the shorter batches the lagging prefix asks for cost more than that -- 32 chains of
    // N = 4096 lost 12 %)
    // (groups of chains as well, now that their ring holds four batches: 32 chains of N = 32768 +7 %, 128 of N = 4096 +3 %)
    const bool pipe = (g.K >= 1024 || g.n > 1 || getenv("PMC_SEQ_PIPE")) && !getenv("PMC_SEQ_NOPIPE");
    rc = seq_group_enqueue(g);
    if (rc) return rc;
    g.rounds_folded = g.rounds_last;
    while (true) {
        bool copies = false, all_admitted = true;
        rc = seq_group_wait(g, copies, all_admitted);
        if (rc && !first) first = rc;
        bool enqueued = false;
        const long long rounds_done = g.rounds_last;
        if (pipe && !all_admitted && !first) {
            rc = seq_group_enqueue(g, true);
            if (rc) return rc;
            enqueued = true;
        }
        g.rounds_folded = rounds_done;
        rc = seq_group_fold(g, copies);
        if (rc && !first) first = rc;
        if (seq_group_done(g)) break;              // (never with a batch in flight: a chain with steps to admit is not done)
        if (!enqueued) {
            rc = seq_group_enqueue(g);
            if (rc) return rc;
        }
    }
    rc = seq_group_end(g, out);
    return first ? first : rc;
}

static int sweep_sequential_fast(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    pmc_system *one[1] = {h};
    return sweep_sequential_fast_many(one, 1, pool, npool, nsweeps, out);
}

// ---- engine 3: chains whose whole state fits the LDS, one wavefront per chain, any number of chains per launch (pmc_seq.h) ----

static size_t seq_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// LDS bytes of k_sq_lds for this system's shape (and the offsets of its arrays)
static size_t seq_lds_layout(const pmc_system *h, int cap, SqLdsArgs *s, int nw = 1)
{
    const bool empty = h->list_type == 0;                       // EmptyList: no cells, no rows; the neighbours are 1:N
    const size_t N = (size_t)h->N, nc = empty ? 1 : (size_t)h->ncell_ref, nn = empty ? 1 : (size_t)h->nn;
    if (empty) cap = 1;
    // an EmptyList system whose positions alone outgrow the LDS keeps them in global memory (everything else -- species, the
    // SpeciesList, the random parts, the ordered sums' segments -- is in LDS): the all-pairs chain of 10^4 particles on four
    // wavefronts instead of the literal loop's one
    const bool pos_global = empty && N * h->D * sizeof(double) > 96 * 1024;
    size_t o = pos_global ? 0 : seq_align(N * h->D * sizeof(double), 16);
    const size_t o_rows = o;      o = seq_align(o + nc * cap * 2, 16);
    const size_t o_ccount = o;    o = seq_align(o + nc * 2, 16);
    const size_t o_cs = o;        o = seq_align(o + N * 2, 16);
    const size_t o_spids = o;     o = seq_align(o + N * 2, 16);
    const size_t o_spheads = o;   o = seq_align(o + N * 2, 16);
    const size_t o_species = o;   o = seq_align(o + N, 16);
    const size_t o_seq = o;       o = seq_align(o + nn * cap * 2, 16);
    const size_t o_gen = o;       o = seq_align(o + (size_t)nw * 64 * sizeof(SqGen), 16);
    const size_t trips = empty ? (N + 63) / 64 : (nn * cap + 63) / 64;
    const size_t gtr = pos_global ? (size_t)SQL_TRGO : trips;    // NW > 1: a segment per trip; positions in global memory: per trip of one go (sql_pass)
    const size_t o_terms = o;     o = seq_align(o + (nw > 1 ? 2 * gtr * (64 + SQ_TERMS_PAD) : 2 * (size_t)(SQ_TERMS + 64 + SQ_TERMS_PAD)) * sizeof(double), 16);
    const size_t o_tmask = o;     if (nw > 1) o = seq_align(o + 2 * gtr * sizeof(int), 16);
    const size_t o_esum = o;      o = seq_align(o + 4 * sizeof(double), 16);
    const size_t o_models = o;    o = seq_align(o + (size_t)h->S * h->S * sizeof(DevModel), 16);
    const size_t o_spoff = o;     o = seq_align(o + 16 * sizeof(int), 16);
    const bool nbr_in = !empty && nc * nn * 4 <= 16384;
    const size_t o_nbr = o;       if (nbr_in) o = seq_align(o + nc * nn * 4, 16);
    const size_t o_nbrcnt = o;    if (nbr_in) o = seq_align(o + nc * 4, 16);
    if (s) {
        s->o_rows = (int)o_rows; s->o_ccount = (int)o_ccount; s->o_cs = (int)o_cs; s->o_spids = (int)o_spids;
        s->o_spheads = (int)o_spheads; s->o_species = (int)o_species; s->o_seq = (int)o_seq; s->o_gen = (int)o_gen;
        s->o_spoff = (int)o_spoff; s->o_nbr = (int)o_nbr; s->o_nbrcnt = (int)o_nbrcnt; s->nbr_in_lds = nbr_in ? 1 : 0;
        s->o_terms = (int)o_terms; s->o_tmask = (int)o_tmask; s->o_esum = (int)o_esum; s->o_models = (int)o_models; s->max_trips = (int)trips;
        s->pos_global = pos_global ? 1 : 0;
    }
    return o;
}

#define SQ_LDS_MAX (160 * 1024 - 512)
#define SQ_LDS_AUTO (64 * 1024)        // auto: chains that leave room for a second workgroup on the CU

static bool seq_lds_fits(const pmc_system *h, size_t limit)
{
    if (h->N > 65535 || h->S > 8) return false;
    if (h->list_type == 3) return false;                        // VerletList: its own update rule (engine 1)
    const int cap = h->d_cl_items ? h->cl_cap : seq_rows_cap(h);
    if (h->list_type == 0) {
        // EmptyList has no other fast engine: the whole LDS, positions in global memory if need be.  With the positions in
        // global memory the four-wavefront instance is the only one: its layout has to fit, a narrower one does not count
        // (N of about 18 900 .. 22 200: the one-wavefront layout would fit and the kernel would overrun it)
        SqLdsArgs probe;
        const size_t l4 = seq_lds_layout(h, cap, &probe, 4);
        if (l4 <= SQ_LDS_MAX) return true;
        if (probe.pos_global) return false;
    }
    return seq_lds_layout(h, cap, nullptr) <= limit;
}

static bool seq_lds_compatible(pmc_system **hs, int n)
{
    const pmc_system *m = hs[0];
    for (int r = 1; r < n; r++) {
        const pmc_system *h = hs[r];
        for (int q = 0; q < r; q++) if (hs[q] == h) return false;
        if (h->p.device != m->p.device || h->D != m->D || h->N != m->N || h->S != m->S || h->list_type != m->list_type ||
            h->ncell_ref != m->ncell_ref || h->nn != m->nn)
            return false;
        for (int a = 0; a < h->D; a++)
            if (h->p.box[a] != m->p.box[a] || h->p.cell[a] != m->p.cell[a] || h->p.ncells[a] != m->p.ncells[a]) return false;
        if (h->models.size() != m->models.size() || memcmp(h->models.data(), m->models.data(), m->models.size() * sizeof(pmc_pair_model)) != 0)
            return false;
    }
    return true;
}

// n chains of one shape, nsweeps sweeps each of the reference's sequential chain, in shared launches of k_sq_lds
static int sweep_sequential_lds(pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    pmc_system *m = hs[0];
    int rc = set_device(m);
    if (rc) return rc;
    // rows of one capacity on every chain (a crowded cell makes a chain's rows grow: then every chain's do)
    int cap = seq_rows_cap(m);
    for (int pass = 0; pass < 4; pass++) {
        for (int r = 0; r < n; r++) if (hs[r]->d_cl_items) cap = std::max(cap, hs[r]->cl_cap);
        bool same = true;
        for (int r = 0; r < n; r++) {
            pmc_system *h = hs[r];
            if (h->list_type == 1 && (!h->d_cl_items || h->cl_cap != cap)) {
                rc = seq_alloc_rows(h, cap);
                if (rc) return rc;
            }
            if (!h->cells_built) { rc = pmc_build_cells(h); if (rc) return rc; }
            if (h->list_type != 0 && h->cl_cap != cap) same = false;
        }
        if (same) break;
    }
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        if (h->list_type != 0 && h->cl_cap != cap) return fail(h, PMC_ERR_STATE, "sequential engine: the chains' cell rows differ in capacity");
        if (int rs = refresh_stale_energy(h)) return rs;
        rc = ensure_species_list(h);
        if (rc) return rc;
    }
    SqLdsArgs s;
    memset(&s, 0, sizeof s);
    s.a = ref_args(m);
    for (int q = 0; q < 3; q++) { s.invbox[q] = 1.0 / s.a.box[q]; s.invcell[q] = (q < m->D && m->list_type != 0) ? 1.0 / s.a.cell[q] : 1.0; }
    s.pool = make_pool(pool, npool);
    s.reverse = (m->list_type == 1) ? 1 : 0;
    s.ncell = m->ncell_ref;
    s.empty = (m->list_type == 0) ? 1 : 0;
    if (const char *e = getenv("PMC_SEQ_DBG")) s.dbg = atoi(e);
    // wavefronts per chain: four split the pair loop and the two ordered sums of a step while the chains alone would leave SIMDs
    // idle; from a thousand chains on, one wavefront per chain fills the chip and wastes nothing on barriers (PMC_SEQ_WAVES: A/B)
    // (measured, N = 216 LJ, 10^7 moves/s for 256 / 512 / 1000 / 2000 chains: four wavefronts 4.8 / 4.9 / 4.8 / 4.8, two
    // 4.3 / 7.3 / 7.2 / 7.3, one 3.2 / 6.1 / 11.6 / 11.9 -- a team's ceiling falls with its width)
    int nw = (n <= 320) ? 4 : (n <= 608 ? 2 : 1);
    if (const char *e = getenv("PMC_SEQ_WAVES")) { nw = atoi(e); if (nw != 4 && nw != 2) nw = 1; }
    SqLdsArgs probe;
    (void)seq_lds_layout(m, cap, &probe, 4);
    if (probe.pos_global) nw = 4;                            // (positions in global memory: the four-wavefront instance is the only one)
    if (nw > 1 && seq_lds_layout(m, cap, nullptr, nw) > SQ_LDS_MAX) {
        // (the positions-in-global-memory kernel deals its trips over four wavefronts and sums on two of them: there is no
        // one-wavefront form of it to fall back to)
        if (probe.pos_global) return fail(m, PMC_ERR_UNSUPPORTED, "sequential engine 3: an EmptyList system of this size does not fit the LDS even with its positions in global memory");
        nw = 1;
    }
    const size_t lds = seq_lds_layout(m, cap, &s, nw);
    if (lds > SQ_LDS_MAX) return fail(m, PMC_ERR_UNSUPPORTED, "sequential engine 3: the chain's state does not fit the LDS");
    // the kernel of a system whose pair models are all of one kind has that potential compiled in (PMC_SEQ_ANYKIND=1: the general kernel, for A/B)
    int ukind = m->models.empty() ? 4 : (int)m->models[0].kind;
    for (const pmc_pair_model &pm : m->models) if ((int)pm.kind != ukind) ukind = 4;
    if (ukind > 3 || getenv("PMC_SEQ_ANYKIND")) ukind = 4;
    const void *kfn = nullptr;
#define SQ_LDS_PICK(DD, WW) (ukind == 0 ? (const void *)k_sq_lds<DD, WW, 0> : ukind == 1 ? (const void *)k_sq_lds<DD, WW, 1> : \
                             ukind == 2 ? (const void *)k_sq_lds<DD, WW, 2> : ukind == 3 ? (const void *)k_sq_lds<DD, WW, 3> : (const void *)k_sq_lds<DD, WW, 4>)
    if (m->D == 3) kfn = (nw == 4) ? SQ_LDS_PICK(3, 4) : (nw == 2) ? SQ_LDS_PICK(3, 2) : SQ_LDS_PICK(3, 1);
    else kfn = (nw == 4) ? SQ_LDS_PICK(2, 4) : (nw == 2) ? SQ_LDS_PICK(2, 2) : SQ_LDS_PICK(2, 1);
#undef SQ_LDS_PICK
    if (s.pos_global) {                                     // positions in global memory: the general-potential instance on four wavefronts
        kfn = (m->D == 3) ? (const void *)k_sq_lds<3, 4, 4, true> : (const void *)k_sq_lds<2, 4, 4, true>;
    }
    HIPCHK(m, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<SqChain> tab((size_t)n);
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        SqChain &c = tab[r];
        memset(&c, 0, sizeof c);
        c.pos = h->d_pos; c.species = h->d_species;
        c.cs = h->d_cs; c.head = h->d_head; c.list = h->d_list; c.ccount = h->d_ccount; c.cl_items = h->d_cl_items;
        c.sp_ids = h->d_sp_ids; c.sp_heads = h->d_sp_heads; c.sp_off = h->d_sp_off;
        c.key0 = h->key[0]; c.key1 = h->key[1];
        c.step0 = h->step_ctr;
        c.energy = h->energy;
        c.temperature = h->p.temperature;
    }
    DevTmp dtab;
    StreamDrain drain(m->stream);       // (tab and dtab are read and written by copies on the stream: drained before they go, error returns included)
    HIPCHK(m, dtab.alloc((size_t)n * sizeof(SqChain)));
    s.chains = dtab.as<SqChain>();
    HIPCHK(m, hipMemcpyAsync(s.chains, tab.data(), (size_t)n * sizeof(SqChain), hipMemcpyHostToDevice, m->stream));
    for (int r = 1; r < n; r++) HIPCHK(hs[r], hipStreamSynchronize(hs[r]->stream));   // earlier work on the chains' own streams
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    const long long total = nsweeps * (int64_t)m->N;          // sweepstep = N, src/ParticlesMC.jl:246
    const long long chunk = 1 << 18;                          // steps per launch and chain (a launch stays well under a second)
    long long launches = 0;
    for (long long done = 0; done < total; done += chunk) {
        s.nsteps = std::min(chunk, total - done);
        {
            void *kargs[1] = {(void *)&s};
            HIPCHK(m, hipLaunchKernel(kfn, dim3(n), dim3(nw * 64), kargs, lds, m->stream));
        }
        HIPCHK(m, hipGetLastError());
        launches++;
    }
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    HIPCHK(m, hipMemcpyAsync(tab.data(), s.chains, (size_t)n * sizeof(SqChain), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    int first = PMC_OK;
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        const SqChain &c = tab[r];
        h->mc_binned = false;
        h->e_fix_valid = false;
        h->img_valid = false;
        if (c.err) {
            h->cells_built = false;
            h->energy_stale = true;
            h->state_invalid = true;        // positions are partly advanced: no caller-side energy belongs to them (pmc_set_energy)
            seq_rows_overflowed(h, c.err);
            if (!first) first = fail(h, PMC_ERR_STATE, (c.err & 1) ? "sequential engine: a cell row overflowed" : "sequential engine: a particle was not in its cell row");
            continue;
        }
        h->energy = c.energy;
        h->step_ctr += (uint64_t)total;
        h->sweep_ctr += (uint64_t)nsweeps;
        if (out) {
            pmc_stats &o = out[r];
            memset(&o, 0, sizeof o);
            for (int q = 0; q < PMC_MAX_MOVES; q++) { o.attempted[q] = (int64_t)c.att[q]; o.accepted[q] = (int64_t)c.acc[q]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = ms;
            o.sweep_kernel_launches = launches;
            o.energy = h->energy;
        }
    }
    return first;
}

// ---- engine 4: VerletList chains with their state in LDS, steps evaluated a batch at a time (pmc_verlet.h) ----------------------

// LDS bytes of k_sq_verlet for this system's shape with rows of `vcap` ids and cell vectors of `ccap` (and the offsets of its arrays):
// pmc_sqv_layout, next to the kernel (pmc_verlet_inst.hip)
static size_t seq_verlet_layout(const pmc_system *h, int vcap, int ccap, int nw, SqvArgs *s)
{
    (void)nw;
    return pmc_sqv_layout(h->N, h->D, h->S, h->ncell_ref, h->nn, vcap, ccap, s);
}

// the capacities engine 4 would run this system with: cell vectors of 1.5 x the mean + 12 ids (at most the device rows' own
// capacity, and one wavefront's 64), Verlet rows as long as the rest of the LDS allows (a multiple of 4, at most the device
// rows' capacity); false if such rows would not even hold the expected list (1.15 x the mean number of particles within rcut + dr, + 3)
static bool seq_verlet_caps(const pmc_system *h, int *vcap_out, int *ccap_out)
{
    if (h->list_type != 3 || h->N > 65535 || h->S > 8 || h->ncell_ref > 65535 || !h->d_vl_nb || h->vl_cap < 4) return false;
    const double mean_c = (double)h->N / (double)h->ncell_ref;
    int ccap = std::min(std::min(h->cl_cap, 64), (int)(1.5 * mean_c) + 12);
    if (const char *e = getenv("PMC_SQV_CCAP")) ccap = std::max(1, std::min(std::min(h->cl_cap, 64), atoi(e)));      // (tests: forces the overflow path)
    double vol = 1.0;
    for (int a = 0; a < h->D; a++) vol *= h->p.box[a];
    const double r = h->maxcut + h->list_dr;
    const double sphere = (h->D == 3) ? 4.18879020478639 * r * r * r : 3.14159265358979 * r * r;
    const double mean_v = sphere * (double)h->N / vol;
    int vcap = std::min(h->vl_cap / 4 * 4, 1024);
    while (vcap >= 4 && seq_verlet_layout(h, vcap, ccap, 0, nullptr) > SQ_LDS_MAX) vcap -= 4;
    if (vcap < 4 || (double)vcap < 1.15 * mean_v + 3.0) return false;
    // rows of up to 4 * SQV_HW ids leave a list in registers (sqv_update): no more than that unless the expected list needs it
    if (vcap > 4 * SQV_HW && 1.5 * mean_v + 4.0 <= 4.0 * SQV_HW) vcap = 4 * SQV_HW;
    if (const char *e = getenv("PMC_SQV_VCAP")) vcap = std::max(4, std::min(vcap, atoi(e) / 4 * 4));                 // (tests: forces the overflow path)
    if (vcap_out) *vcap_out = vcap;
    if (ccap_out) *ccap_out = ccap;
    return true;
}

static bool seq_verlet_fits(const pmc_system *h) { return seq_verlet_caps(h, nullptr, nullptr); }

static bool seq_verlet_compatible(pmc_system **hs, int n)
{
    const pmc_system *m = hs[0];
    for (int r = 1; r < n; r++) {
        const pmc_system *h = hs[r];
        for (int q = 0; q < r; q++) if (hs[q] == h) return false;
        if (h->p.device != m->p.device || h->D != m->D || h->N != m->N || h->S != m->S || h->list_type != m->list_type ||
            h->ncell_ref != m->ncell_ref || h->nn != m->nn || h->list_dr != m->list_dr || h->cl_cap != m->cl_cap || h->vl_cap != m->vl_cap)
            return false;
        for (int a = 0; a < h->D; a++)
            if (h->p.box[a] != m->p.box[a] || h->p.cell[a] != m->p.cell[a] || h->p.ncells[a] != m->p.ncells[a]) return false;
        if (h->models.size() != m->models.size() || memcmp(h->models.data(), m->models.data(), m->models.size() * sizeof(pmc_pair_model)) != 0)
            return false;
    }
    return true;
}

// n VerletList chains of one shape, nsweeps sweeps each of the reference's sequential chain, in shared launches of k_sq_verlet (one
// workgroup per chain).  steps_done[r] (may be null): the steps of the call chain r has behind it when this returns -- fewer than
// nsweeps N only if a row or cell vector outgrew its LDS capacity, in which case the caller runs the rest on the literal loop.
static int sweep_sequential_verlet(pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out, long long *steps_done)
{
    pmc_system *m = hs[0];
    int rc = set_device(m);
    if (rc) return rc;
    bool swaps = false;
    for (int k = 0; k < npool; k++) if (pool[k].kind == PMC_MOVE_SWAP) swaps = true;
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        if (!h->cells_built) { rc = pmc_build_cells(h); if (rc) return rc; }
        if (int rs = refresh_stale_energy(h)) return rs;
        rc = ensure_species_list(h);
        if (rc) return rc;
    }
    SqvArgs s;
    memset(&s, 0, sizeof s);
    int vcap = 0, ccap = 0;
    if (!seq_verlet_caps(m, &vcap, &ccap)) return fail(m, PMC_ERR_UNSUPPORTED, "sequential engine 4: not a VerletList system whose state fits the LDS");
    const size_t lds = seq_verlet_layout(m, vcap, ccap, 0, &s);
    s.a = ref_args(m);
    for (int q = 0; q < 3; q++) { s.invbox[q] = 1.0 / s.a.box[q]; s.invcell[q] = (q < m->D) ? 1.0 / s.a.cell[q] : 1.0; }
    s.pool = make_pool(pool, npool);
    s.ncell = m->ncell_ref;
    int ukind = m->models.empty() ? 4 : (int)m->models[0].kind;
    for (const pmc_pair_model &pm : m->models) if ((int)pm.kind != ukind) ukind = 4;
    if (ukind > 3 || getenv("PMC_SEQ_ANYKIND")) ukind = 4;
    const void *kfn = pmc_sqv_kernel(m->D, ukind);
    HIPCHK(m, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<SqvChain> tab((size_t)n);
    DevTmp dtab, dwork, dlast;
    StreamDrain drain(m->stream);       // (tab and the temporaries are read and written by work on the stream: drained before they go)
    // swaps edit the SpeciesList as they are accepted; a launch that gives up must leave no trace, so the chains work on copies
    // that are copied back by the launch that finishes (k_sq_verlet)
    const size_t N = (size_t)m->N;
    if (swaps) HIPCHK(m, dwork.alloc((size_t)n * 2 * N * sizeof(int)));
    HIPCHK(m, dlast.alloc((size_t)n * N * m->D * sizeof(double)));
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        SqvChain &c = tab[r];
        memset(&c, 0, sizeof c);
        c.pos = h->d_pos; c.species = h->d_species;
        c.cs = h->d_cs; c.ccount = h->d_ccount; c.cl_items = h->d_cl_items;
        c.vl_nb = h->d_vl_nb; c.vl_cnt = h->d_vl_cnt; c.vl_last = h->d_vl_last;
        c.sp_ids = h->d_sp_ids; c.sp_heads = h->d_sp_heads; c.sp_off = h->d_sp_off;
        c.sp_ids_work = swaps ? dwork.as<int>() + (size_t)r * 2 * N : nullptr;
        c.sp_heads_work = swaps ? dwork.as<int>() + (size_t)r * 2 * N + N : nullptr;
        c.last_work = dlast.as<double>() + (size_t)r * N * m->D;
        c.key0 = h->key[0]; c.key1 = h->key[1];
        c.step0 = h->step_ctr;
        c.energy = h->energy;
        c.temperature = h->p.temperature;
    }
    HIPCHK(m, dtab.alloc((size_t)n * sizeof(SqvChain)));
    s.chains = dtab.as<SqvChain>();
    HIPCHK(m, hipMemcpyAsync(s.chains, tab.data(), (size_t)n * sizeof(SqvChain), hipMemcpyHostToDevice, m->stream));
    for (int r = 1; r < n; r++) HIPCHK(hs[r], hipStreamSynchronize(hs[r]->stream));   // earlier work on the chains' own streams
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    const long long total = nsweeps * (int64_t)m->N;          // sweepstep = N, src/ParticlesMC.jl:246
    const long long chunk = 1 << 20;                          // steps per launch and chain (a launch stays well under a second)
    long long launches = 0;
    for (long long done = 0; done < total; done += chunk) {
        s.nsteps = std::min(chunk, total - done);
        void *kargs[1] = {(void *)&s};
        HIPCHK(m, hipLaunchKernel(kfn, dim3(n), dim3(pmc_sqv_threads()), kargs, lds, m->stream));
        HIPCHK(m, hipGetLastError());
        launches++;
    }
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    HIPCHK(m, hipMemcpyAsync(tab.data(), s.chains, (size_t)n * sizeof(SqvChain), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    if (getenv("PMC_SQV_TIMES")) {                          // stamped builds (-DPMC_DEBUG_TIME on pmc_verlet_inst.hip): chain 0's phase clock
        unsigned long long t[16];
        pmc_sqv_times(t);
        fprintf(stderr, "sqv phase clock (resolver, s_memtime ticks): wait %llu  step-overhead %llu | taken %llu  applied %llu  alone-to-update %llu  leave-lists %llu  scan %llu  rest %llu  swaps %llu | own rebuilds %llu  applied %llu  alone %llu | evaluator 0 plain %llu; spins %llu in %llu steps that waited  (%.2f ms; chain 0: %llu batches, %llu alone, %llu rebuilds, %llu applied)\n",
                t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7], t[8], t[9], t[10], t[11], t[12], t[13], t[14], ms, tab[0].n_batches, tab[0].n_serial, tab[0].n_rebuild, tab[0].n_applied);
    }
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        const SqvChain &c = tab[r];
        h->mc_binned = false;
        h->e_fix_valid = false;
        h->img_valid = false;
        // (a launch that gave up changed nothing, and so did every launch behind it: the chain stands where the last good launch left it)
        const long long did = (long long)(c.step0 - h->step_ctr);
        h->energy = c.energy;
        h->step_ctr = c.step0;
        if (did == total) h->sweep_ctr += (uint64_t)nsweeps;
        h->sqv_last[0] = (long long)c.n_batches; h->sqv_last[1] = (long long)c.n_serial; h->sqv_last[2] = (long long)c.n_rebuild; h->sqv_last[3] = did;
        if (steps_done) steps_done[r] = did;
        if (out) {
            pmc_stats &o = out[r];
            memset(&o, 0, sizeof o);
            for (int q = 0; q < PMC_MAX_MOVES; q++) { o.attempted[q] = (int64_t)c.att[q]; o.accepted[q] = (int64_t)c.acc[q]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = ms;
            o.sweep_kernel_launches = launches;
            o.energy = h->energy;
        }
        // (a list that overflowed leaves the step half-applied: a particle missing further on in the SAME step -- the second cell move of a
        // step whose first found the cell full -- is part of that overflow, not an inconsistency of the lists)
        if ((c.err & SQV_ERR_LIST) && !(c.err & (SQV_ERR_ROW | SQV_ERR_CELL | SQV_ERR_LOAD))) {
            static char msg[256];
            snprintf(msg, sizeof msg, "sequential engine 4: a particle was missing from a neighbour's list or from its cell vector (site %d: particle %d, %s %d of %d ids)",
                     c.dbg[0], c.dbg[1], c.dbg[0] == 3 ? "cell" : "neighbour", c.dbg[2], c.dbg[3]);
            return fail(h, PMC_ERR_STATE, msg);
        }
    }
    return PMC_OK;
}

// ---- engine 5: small EmptyList chains in LDS, evaluations repaired term by term (pmc_empty.h) ------------------------------------

static bool seq_empty_pool_ok(const pmc_move *pool, int npool)
{
    for (int m = 0; m < npool; m++) if (pool[m].kind != PMC_MOVE_DISPLACEMENT) return false;     // (a swap re-labels two particles: engine 3)
    return npool > 0;
}

static bool seq_empty_fits(const pmc_system *h)
{
    if (h->list_type != 0 || h->N > 65000 || h->N < 2 || h->S > 8) return false;
    return pmc_sqe_layout(h->N, h->D, h->S, 8, nullptr) <= SQ_LDS_MAX;
}

// wavefronts per chain of engine 5: sixteen (twelve evaluators) while every chain has a CU to itself, eight (two chains to a CU)
// for more chains than that and for systems whose blocks only fit six times; pmc_set_sequential_engine(h, 5, 8 or 16) says which
static int seq_empty_waves(const pmc_system *h, int n)
{
    const bool fits16 = pmc_sqe_layout(h->N, h->D, h->S, 16, nullptr) <= SQ_LDS_MAX;
    if (h->seq_engine == 5 && h->seq_window == 8) return 8;
    if (h->seq_engine == 5 && h->seq_window == 16) return fits16 ? 16 : 8;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->p.device) != hipSuccess || cus <= 0) cus = 256;
    return fits16 && n <= cus ? 16 : 8;
}

// n EmptyList chains of one shape, nsweeps sweeps each of the reference's sequential chain, in shared launches of k_sq_empty (one
// workgroup per chain)
static int sweep_sequential_empty(pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    pmc_system *m = hs[0];
    int rc = set_device(m);
    if (rc) return rc;
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        if (!h->cells_built) { rc = pmc_build_cells(h); if (rc) return rc; }
        if (int rs = refresh_stale_energy(h)) return rs;
    }
    SqeArgs s;
    memset(&s, 0, sizeof s);
    const int nw = seq_empty_waves(m, n);
    const size_t lds = pmc_sqe_layout(m->N, m->D, m->S, nw, &s);
    if (lds > SQ_LDS_MAX) return fail(m, PMC_ERR_UNSUPPORTED, "sequential engine 5: the chain's packed terms do not fit the LDS");
    s.a = ref_args(m);
    for (int q = 0; q < 3; q++) s.invbox[q] = 1.0 / s.a.box[q];
    s.pool = make_pool(pool, npool);
    int ukind = m->models.empty() ? 4 : (int)m->models[0].kind;
    for (const pmc_pair_model &pm : m->models) if ((int)pm.kind != ukind) ukind = 4;
    if (ukind > 3 || getenv("PMC_SEQ_ANYKIND")) ukind = 4;
    const void *kfn = pmc_sqe_kernel(m->D, ukind, nw);
    HIPCHK(m, hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    std::vector<SqeChain> tab((size_t)n);
    DevTmp dtab;
    StreamDrain drain(m->stream);
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        SqeChain &c = tab[r];
        memset(&c, 0, sizeof c);
        c.pos = h->d_pos; c.species = h->d_species;
        c.key0 = h->key[0]; c.key1 = h->key[1];
        c.step0 = h->step_ctr;
        c.energy = h->energy;
        c.temperature = h->p.temperature;
    }
    HIPCHK(m, dtab.alloc((size_t)n * sizeof(SqeChain)));
    s.chains = dtab.as<SqeChain>();
    HIPCHK(m, hipMemcpyAsync(s.chains, tab.data(), (size_t)n * sizeof(SqeChain), hipMemcpyHostToDevice, m->stream));
    for (int r = 1; r < n; r++) HIPCHK(hs[r], hipStreamSynchronize(hs[r]->stream));   // earlier work on the chains' own streams
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    const long long total = nsweeps * (int64_t)m->N;          // sweepstep = N, src/ParticlesMC.jl:246
    const long long chunk = 1 << 20;
    long long launches = 0;
    for (long long done = 0; done < total; done += chunk) {
        s.nsteps = std::min(chunk, total - done);
        void *kargs[1] = {(void *)&s};
        HIPCHK(m, hipLaunchKernel(kfn, dim3(n), dim3(nw * 64), kargs, lds, m->stream));
        HIPCHK(m, hipGetLastError());
        launches++;
    }
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    HIPCHK(m, hipMemcpyAsync(tab.data(), s.chains, (size_t)n * sizeof(SqeChain), hipMemcpyDeviceToHost, m->stream));
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    if (getenv("PMC_SQV_TIMES")) {                          // stamped builds: chain 0's resolver
        unsigned long long t[16];
        pmc_sqv_times(t);
        fprintf(stderr, "sqe phase clock (resolver, s_memtime ticks): record %llu  log %llu  pair terms %llu  inserts %llu  chains %llu  rest %llu  (%.2f ms; chain 0: %llu patched, %llu again, %llu inserted, %llu polls)\n",
                t[0], t[1], t[2], t[3], t[4], t[5], ms, tab[0].n_patched, tab[0].n_again, tab[0].n_inserted, tab[0].n_polls);
    }
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        const SqeChain &c = tab[r];
        h->mc_binned = false;
        h->e_fix_valid = false;
        h->img_valid = false;
        h->energy = c.energy;
        h->step_ctr += (uint64_t)total;
        h->sweep_ctr += (uint64_t)nsweeps;
        h->sqv_last[0] = (long long)c.n_polls; h->sqv_last[1] = (long long)c.n_again; h->sqv_last[2] = (long long)c.n_patched; h->sqv_last[3] = total;
        if (out) {
            pmc_stats &o = out[r];
            memset(&o, 0, sizeof o);
            for (int q = 0; q < PMC_MAX_MOVES; q++) { o.attempted[q] = (int64_t)c.att[q]; o.accepted[q] = (int64_t)c.acc[q]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = ms;
            o.sweep_kernel_launches = launches;
            o.energy = h->energy;
        }
    }
    return PMC_OK;
}
