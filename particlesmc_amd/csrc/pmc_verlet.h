// pmc_verlet.h -- engine 4 of the reference-order chain: VerletList systems (src/neighbours.jl:387-543) with their whole state in
// LDS, steps evaluated a BATCH at a time and resolved in step order.
//
// The reference's only CI system (test/runtests.jl:22-91: config_0, N = 1290, 2-D ternary JBB) runs with a VerletList, dr = 0.2.
// On the literal loop (pmc_ref.h, engine 1) a step of that chain is ~1 us of work behind a dozen dependent global-memory round
// trips: 11 us per step, 12 x slower than one host core.  Two things are different here:
//
//  1. The state -- positions, species, the cell vectors, neighbours[i] for every i -- lives in the LDS of ONE workgroup for the
//     length of a launch (ids as 16-bit words: 1290 rows of 40 ids are 103 KB).  positions_at_last_update is read once per step
//     and written once per rebuild: it stays in global memory (a working copy, L2), its load issued before the sums it hides behind.
//  2. The random part of a step is a pure function of the step counter, so the particle of step t + k is known in advance; and a
//     Verlet row is short (27 ids), so a step reads almost nothing: x_i, last_i, neighbours[i] and the positions / species of its
//     members.  The NW wavefronts of the workgroup each evaluate ONE of the next NW steps against the state at the start of the
//     batch (phase A: e1, e2 in row order, whether the trial position asks for a rebuild); wavefront 0 then walks the batch in
//     step order (phase B).  A step whose inputs nobody has touched since the batch started -- `dirty[i] == 0`: neither i nor any
//     member of neighbours[i] has been moved, re-listed or changed species by an earlier step of the batch -- takes its
//     evaluation as it is: decision, position store, dirty marks for i and its row.  Any other step (and every step whose trial
//     position is dr / 2 from the last update, i.e. rebuilds neighbours[i]; and every swap) is run by wavefront 0 on the current
//     state, exactly as the literal loop runs it.  Lists are symmetric (j in neighbours[i] <=> i in neighbours[j]) at every
//     point of the reference's algorithm, so "i or a member of its row changed" is one mark per member of the changed
//     particle's row.
//
// Same arithmetic, same iterator order (neighbours[i] in vector order: deleteat! keeps it, push! appends), same bits as
// orc_mc_steps_sequential: positions, positions_at_last_update, cells[], neighbours[], accept counts, system.energy.
// A row or cell vector that outgrows its LDS capacity ends the launch WITHOUT writing anything back (SqvChain::err): the host
// continues those steps on the literal loop, whose rows live in global memory.
#pragma once
#include "pmc_seq.h"

#define SQV_GEN 128               // random parts generated at once (threads 0 .. 127)
#define SQV_NW 8                  // wavefronts of a chain's workgroup = steps of a batch
#define SQV_HW 10                 // rows of up to 4 * SQV_HW ids leave a list in registers (sqv_update)

const void *pmc_sqv_kernel(int d, int kind);      // pmc_verlet_inst.hip: k_sq_verlet<d, SQV_NW, kind>
struct SqvArgs;
size_t pmc_sqv_layout(int n, int d, int nspecies, int ncell, int nn, int vcap, int ccap, SqvArgs *s);   // ... the LDS layout of its workgroup
void pmc_sqv_times(unsigned long long *out16);     // ... and the phase clock of a stamped build (-DPMC_DEBUG_TIME; zeros otherwise)

// stamped builds: phase clock of chain 0's wavefront 0 (sums of s_memtime differences): 0 random parts, 1 phase A, 2 phase B,
// 3 steps taken from their evaluation, 4 steps run alone up to the list update, 5 rebuild: leaving the neighbours' lists,
// 6 rebuild: cell vectors + scan + appends, 7 second sums after a rebuild + decision + revert, 8 rebuilds
#ifdef PMC_DEBUG_TIME
extern __device__ unsigned long long g_sqvt[16];
#define SQVT(k) do { if (blockIdx.x == 0 && threadIdx.x < 64) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                     if (wave_lane() == 0) { g_sqvt[k] += now_ - g_sqvt[15]; g_sqvt[15] = now_; } } } while (0)
#define SQVT_COUNT(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sqvt[k] += 1; } while (0)
#else
#define SQVT(k)
#define SQVT_COUNT(k)
#endif

// flags of an evaluated step (SqvSpec): SERIAL: not evaluated (a swap; a rejected step whose inexact revert would rebuild the list),
// REBUILD: the trial position rebuilds the list (e1 stands, e2 does not), OK accepted, FINITE e1 and e2 finite (system.energy is
// updated, src/moves.jl:13-18), CHG: the position to store differs from the current one; bits 8-10: the move
enum { SQV_SERIAL = 1, SQV_REBUILD = 2, SQV_OK = 4, SQV_FINITE = 8, SQV_CHG = 16 };
enum { SQV_ERR_ROW = 1, SQV_ERR_CELL = 2, SQV_ERR_LOAD = 4, SQV_ERR_LIST = 8 };

struct SqvChain {                 // per-chain arguments (device table)
    double *pos;
    int *species;
    int *cs, *ccount, *cl_items;
    int *vl_nb, *vl_cnt;
    double *vl_last;
    int *sp_ids, *sp_heads;
    const int *sp_off;
    int *sp_ids_work, *sp_heads_work;   // pools with swaps: the launch works on these copies of the SpeciesList and copies them back when it finishes
    double *last_work;                  // ... and on this copy of positions_at_last_update (a launch that gives up leaves no trace)
    uint32_t key0, key1;
    unsigned long long step0;     // in: first step of the launch; out: one past the last (unchanged if the launch gave up)
    double energy;                // in: system.energy[1]; out: after the steps
    double temperature;
    unsigned long long att[8], acc[8];                   // out (added to)
    unsigned long long n_serial, n_rebuild, n_batches;   // diagnostics (added to): steps run by wavefront 0 alone, list rebuilds, batches
    int err;                      // SQV_ERR_*: the launch changed nothing
    int pad;
};

struct SqvArgs {
    RefArgs a;                    // geometry, models, neighbour-cell table, dr2o4, rcutoff2, cl_cap / vl_cap (global strides)
    double invbox[3];
    double invcell[3];
    SqvChain *chains;
    MovePool pool;
    long long nsteps;
    int ncell;
    int vcap, ccap;               // LDS capacities of a Verlet row (multiple of 4) and of a cell vector
    int tstride;                  // doubles of one half of a wavefront's terms buffer
    int o_species, o_cs, o_vcnt, o_vrows, o_ccnt, o_crows, o_dirty, o_gen, o_spec, o_models, o_nbr, o_nbrcnt, o_terms, o_ctl;
};

struct __attribute__((aligned(16))) SqvSpec {       // phase A -> phase B
    double e1, de;                // first sum; e2 - e1 (0 unless FINITE)
    double xw[3];                 // the position the step stores: x + d if accepted, (x + d) + (-d) otherwise
    int i, flags;
    double pad;
};

struct SqvView {
    double *pos;
    double *last;                 // positions_at_last_update: the launch's working copy in GLOBAL memory
    unsigned char *species, *dirty;
    unsigned short *cs, *vcnt, *vrows, *ccnt, *crows, *nbr, *nbrcnt;
    SqGen *gen;
    SqvSpec *spec;
    const DevModel *models;
    double *terms;
    int *ctl;                     // [0]: SQV_ERR_* of this launch
};

// a[lane] for lane < 3 without indexing registers by a lane number
__device__ __forceinline__ double sqv_sel3(const double *a, int lane) { return lane == 0 ? a[0] : (lane == 1 ? a[1] : a[2]); }

template <int D>
__device__ __forceinline__ double sqv_dist2(const SqvArgs &s, const double *xi, const double *xj)
{
    double w = sq_vector_1d(xi[0], xj[0], s.a.box[0], s.invbox[0]);
    double r2 = w * w;
#pragma unroll
    for (int q = 1; q < D; q++) {
        w = sq_vector_1d(xi[q], xj[q], s.a.box[q], s.invbox[q]);
        r2 = r2 + w * w;
    }
    return r2;
}

// compute_energy_particle(system, i, ::VerletList) (src/atoms.jl:81-88 over src/neighbours.jl:422-424) at TWO positions at once:
// lanes 0-31 take the row's members 32 at a time at xa with species spa, lanes 32-63 the same members at xb with species spb;
// `other` (a swap partner, or -1) is seen with species sp_other at xb (src/moves.jl:159-167).  Each half packs its non-zero
// terms into its half of `tb` in row order (a term that is 0 leaves a running FP64 sum as it is), then lane 0 and lane 32 run
// the two chains of additions in ONE instruction stream.  Called by a whole wavefront; `tb` is that wavefront's own buffer.
template <int D, int KIND>
__device__ __forceinline__ void sqv_energy2(const SqvArgs &s, const SqvView &v, double *tb, int i, int spa, int spb, int other, int sp_other,
                                            const double *xa, const double *xb, double &ea, double &eb)
{
    const int lane = wave_lane(), half = lane >> 5, idx = lane & 31;
    const int n = v.vcnt[i];
    const unsigned short *row = v.vrows + i * s.vcap;
    const int S = s.a.S;
    double xs[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) xs[q] = half ? xb[q] : xa[q];
    const int spme = half ? spb : spa;
    double *buf = tb + half * s.tstride;
    int cnt = 0;
    for (int base = 0; base < n; base += 32) {
        const int t = base + idx;
        double u = 0.0;
        if (t < n) {
            const int j = row[t];
            double xj[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < D; q++) xj[q] = v.pos[j * D + q];
            int sj = (S > 1) ? (int)v.species[j] : 0;
            if (half && j == other) sj = sp_other;
            const DevModel &m = v.models[spme * S + sj];
            const double r2 = sqv_dist2<D>(s, xs, xj);
            if (!(r2 > m.rcut2)) u = (KIND == 4 ? dev_potential_dyn(r2, m) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2, m));
        }
        const unsigned long long mk = __ballot(u != 0.0);
        const unsigned mine = half ? (unsigned)(mk >> 32) : (unsigned)mk;
        const int rank = __popc(mine & ((1u << idx) - 1u));
        if (u != 0.0) buf[cnt + rank] = u;
        cnt += __popc(mine);
    }
    if (idx < SQ_TERMS_PAD) buf[cnt + idx] = 0.0;                 // (+0.0 leaves the sum as it is: the loop runs in blocks of 8)
    sq_wave_sync();
    const int nmax = max(__builtin_amdgcn_readlane(cnt, 0), __builtin_amdgcn_readlane(cnt, 32));
    double acc = 0.0;
    {
        // the first 24 terms are requested at once (a 2-D list has 20-odd non-zero terms: one round trip, then one chain of
        // additions); the buffer is at least that long (seq_verlet_layout), what lies behind the terms is masked
        double t[3 * SQ_TERMS_PAD];
#pragma unroll
        for (int q = 0; q < 3 * SQ_TERMS_PAD; q++) t[q] = buf[q];
#pragma unroll
        for (int b = 0; b < 3; b++) {
            if (b * SQ_TERMS_PAD < nmax) {
                const bool in = b * SQ_TERMS_PAD < cnt;             // (a block that starts inside the terms ends inside their zero padding)
#pragma unroll
                for (int q = 0; q < SQ_TERMS_PAD; q++) acc += in ? t[b * SQ_TERMS_PAD + q] : 0.0;
            }
        }
    }
    for (int k = 3 * SQ_TERMS_PAD; k < nmax; k += SQ_TERMS_PAD) {
        double t[SQ_TERMS_PAD];
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) t[q] = buf[k + q];
        const bool in = k < cnt;
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) acc += in ? t[q] : 0.0;
    }
    ea = readlane_double(acc, 0);
    eb = readlane_double(acc, 32);
    sq_wave_sync();                                                // (the buffer is free for the wavefront's next call)
}

// every member of neighbours[i], and i: their evaluations of this batch no longer describe the state
__device__ __forceinline__ void sqv_mark_row(const SqvArgs &s, const SqvView &v, int i)
{
    const int lane = wave_lane();
    const int n = v.vcnt[i];
    const unsigned short *row = v.vrows + i * s.vcap;
    for (int q = lane; q < n; q += PMC_WAVE) v.dirty[row[q]] = 1;
    if (lane == 0) v.dirty[i] = 1;
}

// update_neighbour_list!(system, i, ::VerletList), src/neighbours.jl:494-543, on the LDS state; one wavefront.  Nothing unless
// the particle is dr / 2 from its position at its last update; then i leaves every list it is in (order kept), its cell vector
// is updated (filter! + push!), and its own list is rebuilt by scanning the neighbourhood in cell / vector order, i appended to
// every new neighbour's list.  Returns true if the lists were rebuilt.
// `last`: positions_at_last_update[i] (the caller has loaded it; it is x after a rebuild).
template <int D>
__device__ __forceinline__ bool sqv_update(const SqvArgs &s, const SqvView &v, int i, const double *x, double *last)
{
    constexpr int NN = (D == 3) ? 27 : 9;
    const int lane = wave_lane();
    double d2 = 0.0;
#pragma unroll
    for (int q = 0; q < D; q++) {
        const double dx = x[q] - last[q];
        d2 = (q == 0) ? dx * dx : d2 + dx * dx;
    }
    if (d2 < s.a.dr2o4) return false;
    SQVT_COUNT(8);
    const int vcap = s.vcap, ccap = s.ccap;
    int err = 0;
    unsigned short *row_i = v.vrows + i * vcap;
    const int ni = v.vcnt[i];
    // 1. i leaves the lists of its neighbours (deleteat!: the order of what stays is kept).
    if (vcap == 4 * SQV_HW) {
        // Rows of 40 ids: TWO LANES per neighbour, each with one half of that neighbour's row in registers (vcap / 4 words of
        // two ids, plus the word behind them).  A lane finds i in its half, the pair agrees on the position, every word behind it
        // takes the ids one place further on (v_alignbit; the word i sits in is merged under a mask), and the halves go back with
        // the same few wide stores they came with.  No loop over the rows, no dependent round trip per row: the 27 rows of a 2-D
        // list leave in ~1 300 cycles where one row per trip of a loop took 12 500 (LAB.md, round 6).
        constexpr int hw = SQV_HW;                                   // words of a half row
        for (int q0 = 0; q0 < ni; q0 += 32) {
            const int q = q0 + (lane >> 1), hf = lane & 1;
            const bool act = q < ni;
            const int j = act ? (int)row_i[q] : i;
            const int n = act ? (int)v.vcnt[j] : 0;
            unsigned int *rw = reinterpret_cast<unsigned int *>(v.vrows + j * vcap) + hf * hw;   // (rows start on 8-byte boundaries: vcap is a multiple of 4)
            unsigned int w[SQV_HW + 1];
#pragma unroll
            for (int k = 0; k <= SQV_HW; k++) w[k] = rw[k];                      // (the word behind the second half is the next row's first, or the array behind the rows: read, never used below position n)
            const int e0 = 2 * hf * hw;                              // first id of this half
            int p = -1;
#pragma unroll
            for (int k = 0; k < SQV_HW; k++) {
                const int m = e0 + 2 * k;
                if ((int)(w[k] & 0xffffu) == i && m < n) p = m;
                if ((int)(w[k] >> 16) == i && m + 1 < n) p = m + 1;
            }
            p = max(p, __builtin_amdgcn_update_dpp(-1, p, 0xb1, 0xf, 0xf, false));      // quad_perm [1,0,3,2]: the other lane of the pair
            if (act && p < 0) err |= SQV_ERR_LIST;
            if (act && p >= 0) {
#pragma unroll
                for (int k = 0; k < SQV_HW; k++) {
                    const int m = e0 + 2 * k;
                    const unsigned int sh = __builtin_amdgcn_alignbit(w[k + 1], w[k], 16);   // ids m + 1, m + 2
                    const unsigned int keep = (p > m + 1 ? 0xffff0000u : 0u) | (p > m ? 0x0000ffffu : 0u);
                    w[k] = (w[k] & keep) | (sh & ~keep);
                }
#pragma unroll
                for (int k = 0; k < SQV_HW; k++) rw[k] = w[k];
                if (hf == 0) { v.vcnt[j] = (unsigned short)(n - 1); v.dirty[j] = 1; }
            }
        }
    } else {
    // Longer rows (3-D systems): lane q holds neighbour q and the length of its list; then one ROW per trip of the loop, the whole
    // wavefront on it: the entries behind i move up one place
    for (int q0 = 0; q0 < ni; q0 += PMC_WAVE) {
        const int qn = min(PMC_WAVE, ni - q0);
        const int jq = (lane < qn) ? (int)row_i[q0 + lane] : 0;
        const int nq = (lane < qn) ? (int)v.vcnt[jq] : 0;
        for (int q = 0; q < qn; q++) {
            const int j = __builtin_amdgcn_readlane(jq, q);
            const int n = __builtin_amdgcn_readlane(nq, q);
            unsigned short *rj = v.vrows + j * vcap;
            if (n <= PMC_WAVE) {
                const int a = (lane < n) ? (int)rj[lane] : -1;
                const int b = (lane + 1 < n) ? (int)rj[lane + 1] : 0;
                const unsigned long long mk = __ballot(a == i);
                if (!mk) err |= SQV_ERR_LIST;
                const int p = mk ? (int)__builtin_ctzll(mk) : n;
                if (lane >= p && lane + 1 < n) rj[lane] = (unsigned short)b;
            } else {
                int p = -1;
                for (int base = 0; base < n && p < 0; base += PMC_WAVE) {
                    const int k = base + lane;
                    const unsigned long long mk = __ballot(k < n && (int)rj[k] == i);
                    if (mk) p = base + (int)__builtin_ctzll(mk);
                }
                if (p < 0) { err |= SQV_ERR_LIST; p = n; }
                for (int base = p; base < n - 1; base += PMC_WAVE) {
                    const int k = base + lane;
                    unsigned short t = 0;
                    if (k < n - 1) t = rj[k + 1];
                    sq_wave_sync();
                    if (k < n - 1) rj[k] = t;
                    sq_wave_sync();
                }
            }
        }
        if (lane < qn) { v.vcnt[jq] = (unsigned short)(nq - 1); v.dirty[jq] = 1; }
    }
    }
    sq_wave_sync();
    SQVT(5);
    // 2. the cell vectors (update_cell_list!, src/neighbours.jl:468-476: filter! keeps the order, push! appends)
    const int c = v.cs[i];
    const int c2 = sq_cell_of_fast<D>(s.a, s.invcell, x);
    if (c != c2) {
        unsigned short *rc = v.crows + c * ccap;
        const int n = v.ccnt[c];                                     // (<= ccap <= 64 checked by the host)
        const int a = (lane < n) ? (int)rc[lane] : -1;
        const int b = (lane + 1 < n) ? (int)rc[lane + 1] : 0;
        const unsigned long long mk = __ballot(a == i);
        if (!mk) err |= SQV_ERR_LIST;
        const int p = mk ? (int)__builtin_ctzll(mk) : n;
        if (lane >= p && lane + 1 < n) rc[lane] = (unsigned short)b;
        const int n2 = v.ccnt[c2];
        if (n2 >= ccap) err |= SQV_ERR_CELL;
        if (lane == 0) {
            v.ccnt[c] = (unsigned short)(mk ? n - 1 : n);
            if (n2 < ccap) { v.crows[c2 * ccap + n2] = (unsigned short)i; v.ccnt[c2] = (unsigned short)(n2 + 1); }
            v.cs[i] = (unsigned short)c2;
        }
    }
    sq_wave_sync();
    // 3. the new list: the cells around c2 in table order, every cell's vector in order.  Lane l < cnt holds neighbour cell l and
    //    its length; candidate t of the scan finds its cell by counting the cells that end at or before t
    const int cnt = v.nbrcnt[c2];
    const int cc = (lane < cnt) ? (int)v.nbr[c2 * NN + lane] : 0;
    const int len = (lane < cnt) ? (int)v.ccnt[cc] : 0;
    const int incl = sq_scan32(len);                                 // (inclusive; lanes >= cnt repeat the total)
    const int total = __builtin_amdgcn_readlane(incl, 31);
    int fill = 0;
    for (int base = 0; base < total; base += PMC_WAVE) {
        const int t = base + lane;
        int l = 0;
#pragma unroll
        for (int m = 0; m < NN - 1; m++) l += (t >= __builtin_amdgcn_readlane(incl, m)) ? 1 : 0;
        l = min(l, cnt - 1);
        const int cc_l = __shfl(cc, l, PMC_WAVE);
        const int end_l = __shfl(incl, l, PMC_WAVE), len_l = __shfl(len, l, PMC_WAVE);
        bool in = false;
        int j = 0;
        if (t < total) {
            j = v.crows[cc_l * ccap + (t - (end_l - len_l))];
            if (j != i) {
                double xj[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < D; q++) xj[q] = v.pos[j * D + q];
                in = sqv_dist2<D>(s, x, xj) < s.a.rcutoff2;
            }
        }
        const unsigned long long mk = __ballot(in);
        if (in) {
            const int p = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            const int nj = v.vcnt[j];
            if (p < vcap && nj < vcap) {
                row_i[p] = (unsigned short)j;
                v.vrows[j * vcap + nj] = (unsigned short)i;
                v.vcnt[j] = (unsigned short)(nj + 1);
                v.dirty[j] = 1;
            } else {
                err |= SQV_ERR_ROW;
            }
        }
        fill += __popcll(mk);
    }
    if (lane == 0) {
        v.vcnt[i] = (unsigned short)min(fill, vcap);
        v.dirty[i] = 1;
    }
    if (lane < D) v.last[i * D + lane] = sqv_sel3(x, lane);
#pragma unroll
    for (int q = 0; q < D; q++) last[q] = x[q];
    if (__ballot(err != 0)) {
        int e = err;
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) e |= __shfl_xor(e, sft, PMC_WAVE);
        if (lane == 0) v.ctl[0] |= e;
    }
    sq_wave_sync();
    SQVT(6);
    return true;
}

// One Displacement step of the chain on the current LDS state, by one wavefront, as the literal loop runs it (ref_disp_perform /
// ref_disp_revert): e1, x += d, update_neighbour_list!, e2, accept or revert (+ update_neighbour_list! again).  Returns accepted.
// have_e1: the step's evaluation of this batch stands as far as e1 goes (nobody has touched i or its row since) and its trial
// position rebuilds the list: e1 is taken, e2 follows the rebuild.
template <int D, int KIND>
__device__ __forceinline__ bool sqv_disp_serial(const SqvArgs &s, const SqvView &v, double *tb, int i, const SqGen &ge, double T, double &E,
                                                unsigned long long &nreb, bool have_e1, double e1_in)
{
    const int lane = wave_lane();
    double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0}, last[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) last[q] = v.last[i * D + q];            // (global: in flight behind the sums below)
#pragma unroll
    for (int q = 0; q < D; q++) {
        x[q] = v.pos[i * D + q];
        xn[q] = x[q] + ge.d[q];
        xr[q] = xn[q] + (-ge.d[q]);
    }
    const int spi = (s.a.S > 1) ? (int)v.species[i] : 0;
    sqv_mark_row(s, v, i);
    double e1 = e1_in, e2 = 0.0, dummy;
    bool have_e2 = false;
    if (!have_e1) {
        // both sums over the row as it is: right unless the trial position rebuilds the list (then e2 is taken again)
        sqv_energy2<D, KIND>(s, v, tb, i, spi, spi, -1, 0, x, xn, e1, e2);
        have_e2 = true;
    }
    if (lane < D) v.pos[i * D + lane] = sqv_sel3(xn, lane);
    sq_wave_sync();
    SQVT(4);
    if (sqv_update<D>(s, v, i, xn, last)) {
        nreb++;
        have_e2 = false;
    }
    if (!have_e2) sqv_energy2<D, KIND>(s, v, tb, i, spi, spi, -1, 0, xn, xn, e2, dummy);
    double de = 0.0;
    if (!(isinf(e1) || isinf(e2))) { de = e2 - e1; E += de; }            // src/moves.jl:13-18
    const bool ok = ge.lnu < -(e2 - e1) / T;
    if (!ok) {
        E -= de;                                                          // src/moves.jl:79
        if (lane < D) v.pos[i * D + lane] = sqv_sel3(xr, lane);
        sq_wave_sync();
        SQVT(7);
        if (sqv_update<D>(s, v, i, xr, last)) nreb++;
    }
    sq_wave_sync();
    SQVT(7);
    return ok;
}

template <int D, int NW, int KIND>
__global__ __launch_bounds__(NW * 64) void k_sq_verlet(SqvArgs s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sqv_smem[];
    constexpr int NN = (D == 3) ? 27 : 9;
    const RefArgs &a = s.a;
    SqvChain &ch = s.chains[blockIdx.x];
    const int tid = threadIdx.x, nth = NW * 64, lane = wave_lane(), wv = tid >> 6;
    const int N = a.N, S = a.S, vcap = s.vcap, ccap = s.ccap;
    SqvView v;
    v.pos = reinterpret_cast<double *>(sqv_smem);
    v.last = ch.last_work;
    v.species = sqv_smem + s.o_species;
    v.dirty = sqv_smem + s.o_dirty;
    v.cs = reinterpret_cast<unsigned short *>(sqv_smem + s.o_cs);
    v.vcnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_vcnt);
    v.vrows = reinterpret_cast<unsigned short *>(sqv_smem + s.o_vrows);
    v.ccnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_ccnt);
    v.crows = reinterpret_cast<unsigned short *>(sqv_smem + s.o_crows);
    v.nbr = reinterpret_cast<unsigned short *>(sqv_smem + s.o_nbr);
    v.nbrcnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_nbrcnt);
    v.gen = reinterpret_cast<SqGen *>(sqv_smem + s.o_gen);
    v.spec = reinterpret_cast<SqvSpec *>(sqv_smem + s.o_spec);
    v.terms = reinterpret_cast<double *>(sqv_smem + s.o_terms);
    v.ctl = reinterpret_cast<int *>(sqv_smem + s.o_ctl);
    {
        DevModel *lm = reinterpret_cast<DevModel *>(sqv_smem + s.o_models);
        const int *src = reinterpret_cast<const int *>(a.models);
        int *dst = reinterpret_cast<int *>(lm);
        for (int k = tid; k < S * S * (int)(sizeof(DevModel) / sizeof(int)); k += nth) dst[k] = src[k];
        v.models = lm;
    }
    // -- the chain's state into LDS
    if (tid == 0) v.ctl[0] = 0;
    __syncthreads();
    int lerr = 0;
    for (int k = tid; k < s.ncell * NN; k += nth) v.nbr[k] = (unsigned short)a.nbr[(size_t)(k / NN) * a.nnmax + (k % NN)];
    for (int k = tid; k < s.ncell; k += nth) {
        v.nbrcnt[k] = (unsigned short)a.nbrcnt[k];
        const int n = ch.ccount[k];
        v.ccnt[k] = (unsigned short)min(n, ccap);
        if (n > ccap) lerr |= SQV_ERR_LOAD;
    }
    for (int k = tid; k < N * D; k += nth) { v.pos[k] = ch.pos[k]; v.last[k] = ch.vl_last[k]; }     // (v.last: the working copy in global memory)
    for (int k = tid; k < N; k += nth) {
        v.species[k] = (unsigned char)ch.species[k];
        v.cs[k] = (unsigned short)ch.cs[k];
        const int n = ch.vl_cnt[k];
        v.vcnt[k] = (unsigned short)min(n, vcap);
        if (n > vcap) lerr |= SQV_ERR_LOAD;
    }
    __syncthreads();
    for (int k = tid; k < s.ncell * ccap; k += nth) {
        const int c = k / ccap, q = k - c * ccap;
        if (q < (int)v.ccnt[c]) v.crows[k] = (unsigned short)ch.cl_items[(size_t)c * a.cl_cap + q];
    }
    for (int k = tid; k < N * vcap; k += nth) {
        const int i = k / vcap, q = k - i * vcap;
        if (q < (int)v.vcnt[i]) v.vrows[k] = (unsigned short)ch.vl_nb[(size_t)i * a.vl_cap + q];
    }
    if (lerr) atomicOr(v.ctl, lerr);
    if (ch.sp_ids_work) for (int k = tid; k < N; k += nth) { ch.sp_ids_work[k] = ch.sp_ids[k]; ch.sp_heads_work[k] = ch.sp_heads[k]; }
    __syncthreads();
    double E = ch.energy;
    const double T = ch.temperature, T_ = T;
    unsigned long long att[8], acc[8], nser = 0, nreb = 0, nbat = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) { att[q] = 0; acc[q] = 0; }
    double *tb = v.terms + (size_t)wv * 2 * s.tstride;
    const unsigned long long gend = ch.step0 + (unsigned long long)s.nsteps;
    bool dead = v.ctl[0] != 0;
    for (unsigned long long g0 = ch.step0; g0 < gend && !dead; g0 += SQV_GEN) {
        if (tid < SQV_GEN) {
            // the words and operations of k_ref_ops for step g
            const unsigned long long g = g0 + (unsigned long long)tid;
            uint32_t R[4], Q[4];
            philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ0, ch.key0, ch.key1, R);
            philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ1, ch.key0, ch.key1, Q);
            SqGen o;
            o.m = select_move(s.pool, R[0]);
            o.lnu = det_log(uniform53(R[2], R[3]));
            o.r1 = R[1]; o.q0 = Q[0]; o.pad = 0;
            o.d[0] = o.d[1] = o.d[2] = 0.0;
            if (s.pool.kind[o.m] == 0) {
                double z[4];
                box_muller(Q[0], Q[1], &z[0], &z[1]);
                box_muller(Q[2], Q[3], &z[2], &z[3]);
                for (int q = 0; q < 3; q++) o.d[q] = s.pool.sigma[o.m] * z[q];
            }
            v.gen[tid] = o;
        }
        __syncthreads();
        SQVT(0);
        const int ng = (int)min((unsigned long long)SQV_GEN, gend - g0);
        for (int b0 = 0; b0 < ng && !dead; b0 += NW) {
            const int nb = min(NW, ng - b0);
            // ---- phase A: wavefront w evaluates step b0 + w against the state as it is now
            for (int k = tid; k < (N + 3) / 4; k += nth) reinterpret_cast<unsigned int *>(v.dirty)[k] = 0u;
            if (wv < nb) {
                const SqGen ge = v.gen[b0 + wv];
                SqvSpec sp;
                sp.e1 = sp.de = sp.pad = 0.0; sp.xw[0] = sp.xw[1] = sp.xw[2] = 0.0; sp.i = 0; sp.flags = SQV_SERIAL | (ge.m << 8);
                if (s.pool.kind[ge.m] == 0) {
                    const int i = (int)mulhi32(ge.r1, (uint32_t)N);
                    double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0}, last[3] = {0, 0, 0};
#pragma unroll
                    for (int q = 0; q < D; q++) last[q] = v.last[i * D + q];     // (global: in flight behind the sums)
#pragma unroll
                    for (int q = 0; q < D; q++) {
                        x[q] = v.pos[i * D + q];
                        xn[q] = x[q] + ge.d[q];
                        xr[q] = xn[q] + (-ge.d[q]);
                    }
                    const int spi = (S > 1) ? (int)v.species[i] : 0;
                    double e1, e2;
                    sqv_energy2<D, KIND>(s, v, tb, i, spi, spi, -1, 0, x, xn, e1, e2);
                    double d2 = 0.0, d2r = 0.0;
                    bool chg = false;
#pragma unroll
                    for (int q = 0; q < D; q++) {
                        const double dx = xn[q] - last[q], dr_ = xr[q] - last[q];
                        d2 = (q == 0) ? dx * dx : d2 + dx * dx;
                        d2r = (q == 0) ? dr_ * dr_ : d2r + dr_ * dr_;
                        chg |= sq_bits_differ(xr[q], x[q]);
                    }
                    int fl = ge.m << 8;
                    sp.i = i;
                    sp.e1 = e1;
                    if (d2 < a.dr2o4) {
                        // the trial position keeps the list: e1 and e2 are sums over the same row, and the step is decided here
                        const bool fin = !(isinf(e1) || isinf(e2));
                        const bool ok = ge.lnu < -(e2 - e1) / T_;
                        if (fin) { fl |= SQV_FINITE; sp.de = e2 - e1; }
                        if (ok) fl |= SQV_OK | SQV_CHG;
                        else if (chg) fl |= SQV_CHG | ((d2r < a.dr2o4) ? 0 : SQV_SERIAL);   // (an inexact revert that rebuilds the list: run alone)
#pragma unroll
                        for (int q = 0; q < D; q++) sp.xw[q] = ok ? xn[q] : xr[q];
                    } else {
                        fl |= SQV_REBUILD;
                    }
                    sp.flags = fl;
                }
                if (lane == 0) v.spec[wv] = sp;
            }
            __syncthreads();
            SQVT(1);
            // ---- phase B: wavefront 0 takes the steps in order
            if (wv == 0) {
                nbat++;
                for (int k = 0; k < nb; k++) {
                    const SqvSpec sp = v.spec[k];
                    const int m = (sp.flags >> 8) & 7;
                    bool ok = false;
                    if (s.pool.kind[m] == 0) {
                        const int i = sp.i;
                        const bool clean = !v.dirty[i];
                        if (clean && !(sp.flags & (SQV_SERIAL | SQV_REBUILD))) {
                            // nobody has touched i or its row since the batch started: the evaluation is the step
                            ok = (sp.flags & SQV_OK) != 0;
                            if (sp.flags & SQV_FINITE) { E += sp.de; if (!ok) E -= sp.de; }     // src/moves.jl:17,79
                            if (sp.flags & SQV_CHG) {
                                // (the stored position is within dr / 2 of the last update: update_neighbour_list! returns at once)
                                if (lane < D) v.pos[i * D + lane] = sqv_sel3(sp.xw, lane);
                                sqv_mark_row(s, v, i);
                            }
                            SQVT(3);
                        } else {
                            nser++;
                            const SqGen ge = v.gen[b0 + k];
                            ok = sqv_disp_serial<D, KIND>(s, v, tb, i, ge, T, E, nreb, clean && (sp.flags & SQV_REBUILD), sp.e1);
                        }
                    } else {
                        // DiscreteSwap (src/moves.jl:137-207): four sums, species exchange, SpeciesList fix-up; the SpeciesList
                        // stays in global memory (the launch's working copy: this wavefront is its only reader and writer)
                        nser++;
                        const SqGen ge = v.gen[b0 + k];
                        const int sa = s.pool.sp0[m], sb = s.pool.sp1[m];
                        const int oa = ch.sp_off[sa], ob = ch.sp_off[sb];
                        const int na = ch.sp_off[sa + 1] - oa, nbb = ch.sp_off[sb + 1] - ob;
                        if (na > 0 && nbb > 0) {
                            const int i = ch.sp_ids_work[oa + mulhi32(ge.r1, (uint32_t)na)];
                            const int j = ch.sp_ids_work[ob + mulhi32(ge.q0, (uint32_t)nbb)];
                            double xi[3] = {0, 0, 0}, xj[3] = {0, 0, 0};
#pragma unroll
                            for (int q = 0; q < D; q++) { xi[q] = v.pos[i * D + q]; xj[q] = v.pos[j * D + q]; }
                            const int spi = v.species[i], spj = v.species[j];
                            double e1i, e2i, e1j, e2j;
                            sqv_energy2<D, KIND>(s, v, tb, i, spi, spj, j, spi, xi, xi, e1i, e2i);
                            sqv_energy2<D, KIND>(s, v, tb, j, spj, spi, i, spj, xj, xj, e1j, e2j);
                            const double e1 = e1i + e1j, e2 = e2i + e2j;
                            double de = 0.0;
                            if (!(isinf(e1) || isinf(e2))) { de = e2 - e1; E += de; }
                            ok = ge.lnu < -(e2 - e1) / T;
                            if (ok) {
                                sqv_mark_row(s, v, i);
                                sqv_mark_row(s, v, j);
                                if (lane == 0) {
                                    v.species[i] = (unsigned char)spj;
                                    v.species[j] = (unsigned char)spi;
                                    // update_species_list!, src/moves.jl:175-179
                                    const int hi_ = ch.sp_heads_work[i], hj_ = ch.sp_heads_work[j];
                                    ch.sp_ids_work[oa + hi_] = j;
                                    ch.sp_ids_work[ob + hj_] = i;
                                    ch.sp_heads_work[i] = hj_;
                                    ch.sp_heads_work[j] = hi_;
                                }
                                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                                __builtin_amdgcn_wave_barrier();
                            } else {
                                E -= de;                                             // src/moves.jl:206
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < 8; q++) {
                        att[q] += (q == m);
                        acc[q] += (q == m && ok);
                    }
                }
            }
            __syncthreads();
            SQVT(2);
            dead = v.ctl[0] != 0;
        }
    }
    __syncthreads();
    if (dead) {
        // a row or cell vector outgrew the LDS (or the lists were inconsistent): nothing is written back -- the swaps of this launch
        // edited only the working copy of the SpeciesList -- and the host runs these steps on the literal loop
        if (tid == 0) ch.err = v.ctl[0];
        return;
    }
    // -- back to global memory
    for (int k = tid; k < N * D; k += nth) { ch.pos[k] = v.pos[k]; ch.vl_last[k] = v.last[k]; }
    for (int k = tid; k < N; k += nth) {
        ch.species[k] = v.species[k];
        ch.cs[k] = v.cs[k];
        ch.vl_cnt[k] = v.vcnt[k];
    }
    if (ch.sp_ids_work) for (int k = tid; k < N; k += nth) { ch.sp_ids[k] = ch.sp_ids_work[k]; ch.sp_heads[k] = ch.sp_heads_work[k]; }
    for (int k = tid; k < s.ncell; k += nth) ch.ccount[k] = v.ccnt[k];
    for (int k = tid; k < s.ncell * ccap; k += nth) {
        const int c = k / ccap, q = k - c * ccap;
        if (q < (int)v.ccnt[c]) ch.cl_items[(size_t)c * a.cl_cap + q] = v.crows[k];
    }
    for (int k = tid; k < N * vcap; k += nth) {
        const int i = k / vcap, q = k - i * vcap;
        if (q < (int)v.vcnt[i]) ch.vl_nb[(size_t)i * a.vl_cap + q] = v.vrows[k];
    }
    if (tid == 0) {
        ch.energy = E;
        ch.step0 = gend;
        for (int q = 0; q < 8; q++) { ch.att[q] += att[q]; ch.acc[q] += acc[q]; }
        ch.n_serial += nser; ch.n_rebuild += nreb; ch.n_batches += nbat;
    }
}
