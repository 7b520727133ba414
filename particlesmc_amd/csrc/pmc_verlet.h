// pmc_verlet.h -- engine 4 of the reference-order chain: VerletList systems (src/neighbours.jl:387-543) with their whole state in
// LDS, one wavefront RESOLVING the steps in order while twelve others EVALUATE the steps behind them, one APPLIES the list edits
// the resolver hands over, and one evaluates AGAIN the steps whose first evaluation did not stand.
//
// The reference's only CI system (test/runtests.jl:22-91: config_0, N = 1290, 2-D ternary JBB) runs with a VerletList, dr = 0.2.
// On the literal loop (pmc_ref.h, engine 1) a step of that chain is ~1 us of work behind a dozen dependent global-memory round
// trips: 11 us per step, 12 x slower than one host core.  What is different here:
//
//  1. The state -- positions, species, the cell vectors, neighbours[i] for every i -- lives in the LDS of ONE workgroup for the
//     length of a launch (ids as 16-bit words: 1290 rows of 40 ids are 103 KB).  positions_at_last_update is read once per step
//     and written once per rebuild: it stays in global memory (a working copy, L2), its load issued before the sums it hides behind.
//  2. The random part of a step is a pure function of the step counter, so the particle of step t + k is known in advance; and a
//     Verlet row is short (27 ids), so a step reads almost nothing: x_i, last_i, neighbours[i] and the positions / species of its
//     members.  The workgroup is a rolling pipeline of sixteen wavefronts, no barrier inside a launch:
//       * twelve EVALUATORS (evaluator e: the steps t = e mod 12) each begin their step once the resolver is within SQV_LOOK
//         steps of it and work it out against the state as they find it: e1 and e2 in row order, the decision, the position to
//         store; and if the trial position is dr / 2 from the last update, the REBUILT list (the scan of the neighbourhood in
//         cell / vector order, src/neighbours.jl:523-538), e2 over it, and the list of the reverted position if the step is
//         rejected far from where it was re-listed.  They leave a record (SqvSpec) and the list.
//       * wavefront 0, the RESOLVER, walks the steps in order.  A step whose inputs nobody has touched since its evaluation
//         began -- no mark on i newer than the evaluation (the resolver marks i and every member of neighbours[i] with the
//         step number whenever it moves, re-lists or re-labels i: lists are symmetric at every point of the reference's
//         algorithm), and for a rebuilt list no such mark on the cells its scan read -- IS its evaluation: energy update,
//         position store, marks.  A rebuilt list is HANDED OVER: the resolver marks what the edit will touch (i, its old and
//         new neighbours, the cells) and goes on;
//       * wavefront 8, the APPLIER, makes the edits in the order they were handed over (i leaves its old neighbours' rows in
//         registers, is appended to the new ones', moves between cell vectors).  An evaluation can vouch for the lists only up
//         to the step before the oldest edit in the making, and says so in its record; the resolver waits for the applier only
//         before a step it runs on the lists itself;
//       * a step whose evaluation does not stand is evaluated AGAIN on the state as it is by wavefront 12 (the second opinion:
//         same code as an evaluator, into the same record) and taken like any other; swaps are run by the resolver, an inexact
//         revert that re-lists the particle by wavefront 12, exactly as the literal loop runs them;
//       * wavefront 4 generates the random parts, 64 steps at a time.  (0, 4, 8 and 12 share a SIMD: three of them sleep.)
//     An evaluation that raced with the resolver's or the applier's stores may have read anything -- every id, count and cell
//     index it can have read is a valid one (rows and cell vectors are zero-filled at the start, entries are stored before the
//     counts that expose them), and whatever it computed is discarded by the marks.
//
// Same arithmetic, same iterator order (neighbours[i] in vector order: deleteat! keeps it, push! appends), same bits as
// orc_mc_steps_sequential: positions, positions_at_last_update, cells[], neighbours[], accept counts, system.energy.
// A row or cell vector that outgrows its LDS capacity ends the launch WITHOUT writing anything back (SqvChain::err): the host
// continues those steps on the literal loop, whose rows live in global memory.
#pragma once
#include "pmc_seq.h"

#define SQV_GEN 128               // ring of random parts: two blocks of 64 steps
#ifndef SQV_NW
#define SQV_NW 16                 // wavefronts of a chain's workgroup: wavefront 0 resolves, 4 generates, 8 applies the handed-over list edits, 12 evaluates again
#endif
#define SQV_NB (3 * SQV_NW / 4)   // evaluators (wavefronts 1-3, 5-7, 9-11, 13-15): evaluator e takes the steps t = e (mod SQV_NB)
#ifndef SQV_LOOK
#define SQV_LOOK 7                // an evaluator begins step t once the resolver has finished step t - SQV_LOOK - 1 (< SQV_NB: its record is free by then)
#endif
#ifndef SQV_HOLD
#define SQV_HOLD SQV_LOOK         // ... and no later than SQV_HOLD steps behind the oldest rebuild the applier has not finished (<= SQV_NB - 1: the list of the evaluator's last step is the applier's until then; 7 / 9 / 11: 1.007 / 0.984 / 0.963e6 moves/s)
#endif
#define SQV_HW 10                 // rows of 4 * SQV_HW ids leave a list in registers (sqv_leave_lists)

struct SqvArgs;
const void *pmc_sqv_kernel(int d, int kind);      // pmc_verlet_inst.hip: k_sq_verlet<d, kind>
size_t pmc_sqv_layout(int n, int d, int nspecies, int ncell, int nn, int vcap, int ccap, SqvArgs *s);   // ... the LDS layout of its workgroup
int pmc_sqv_threads();                              // ... the threads of that workgroup
void pmc_sqv_times(unsigned long long *out16);     // ... and the phase clock of a stamped build (-DPMC_DEBUG_TIME; zeros otherwise)

// stamped builds: phase clock of chain 0's resolver (sums of s_memtime differences): 0 waiting at the batch barrier, 1 clearing
// the marks, 2 steps taken from their evaluation, 3 rebuilds applied from an evaluator's list, 4 steps run alone up to the list
// update, 5 rebuild: leaving the neighbours' lists, 6 rebuild: cell vectors + scan + appends, 7 rest of a step run alone,
// 8 swaps; counters: 9 rebuilds by the resolver's own scan, 10 applied, 11 steps run alone
#ifdef SQV_MARK
#define SQVM(name) asm volatile("; SQVM " name)
#else
#define SQVM(name)
#endif
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
// REBUILD: the trial position rebuilds the list (e1 stands; e2, the decision and the new list only with LIST), REB2: the reverted
// position rebuilds it again (the list is that of the reverted position), OK accepted, FINITE e1 and e2 finite (system.energy is
// updated, src/moves.jl:13-18), CHG: the position to store differs from the current one, DISP: a Displacement; bits 8-10: the move
enum { SQV_SERIAL = 1, SQV_REBUILD = 2, SQV_OK = 4, SQV_FINITE = 8, SQV_CHG = 16, SQV_LIST = 32, SQV_REB2 = 64, SQV_DISP = 128 };
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
    unsigned long long n_serial, n_rebuild, n_batches;   // diagnostics (added to): steps the resolver ran alone, list rebuilds, polls of the resolver for a record that was not ready
    unsigned long long n_applied;                        // ... rebuilds applied from an evaluator's list
    int err;                      // SQV_ERR_*: the launch changed nothing
    int pad;
    int dbg[4];                   // SQV_ERR_LIST: where (1, 2 a neighbour's list, 3 a cell vector), i, the neighbour or cell, its length
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
    int o_species, o_cs, o_vcnt, o_vrows, o_ccnt, o_crows, o_touch, o_ctouch, o_gen, o_spec, o_lst, o_models, o_nbr, o_nbrcnt, o_terms, o_ctl;
};

struct __attribute__((aligned(16))) SqvSpec {       // evaluator -> resolver
    double e1, de;                // first sum; e2 - e1 (0 unless FINITE)
    double xw[3];                 // the position the step stores: x + d if accepted, (x + d) + (-d) otherwise
    double xl[3];                 // LIST: positions_at_last_update after the step
    int i, flags;
    short c0, c2, c3, nl;         // the cell of i when the step was evaluated; LIST: cell of the trial position, of the reverted position (REB2), ids of the list
    int s0;                       // the steps the resolver had finished when the evaluation began: a mark by any later step voids it
    int seq;                      // step + 1 once the record is complete (stored last, read first)
    int pad0, pad1;
};

struct SqvView {
    double *pos;
    double *last;                 // positions_at_last_update: the launch's working copy in GLOBAL memory
    unsigned char *species;
    unsigned short *touch;        // per particle: the step (its number in the launch + 1, low 16 bits) that last moved, re-listed or re-labelled it or a member of its list
    unsigned short *ctouch;       // per cell: ... that last moved a particle of the cell, put one in, took one out or re-labelled one
    unsigned short stamp;         // the resolver's current step, as the marks store it
    unsigned short *cs, *vcnt, *vrows, *ccnt, *crows, *nbr, *nbrcnt;
    SqGen *gen;
    const DevModel *models;
    int *ctl;                     // [0]: SQV_ERR_* of this launch
};

// LDS words the wavefronts of the pipeline talk through (workgroup scope; LDS operations of one wavefront execute in order)
__device__ __forceinline__ int sqv_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sqv_st(int *p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ int sqv_ld_relaxed(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void sqv_st_relaxed(int *p, int x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// An evaluator's record as the resolver takes it.  What every step needs -- the 16 bytes with the sequence number FIRST (the LDS
// takes a wavefront's operations in order; the evaluator stores that number last), then i / flags / cells, e1 / de and the
// position to store -- is requested while the resolver is busy with the step before, and used only when that step is over.
// `zero` is a 0 the compiler cannot see through: the address looks lane-dependent, so the values stay in vector registers
// (no v_readfirstlane, hence no wait, right behind the request) until sqv_rec_decode asks for them.
typedef unsigned int sqv_u4 __attribute__((ext_vector_type(4)));
typedef unsigned int sqv_u2 __attribute__((ext_vector_type(2)));
struct SqvRaw { sqv_u2 q5; sqv_u4 q4, q0, q1; };
__device__ __forceinline__ void sqv_rec_request(const SqvSpec *rec, int zero, SqvRaw &r)
{
    const sqv_u4 *p = reinterpret_cast<const sqv_u4 *>(reinterpret_cast<const unsigned char *>(rec) + zero);
    r.q5 = *reinterpret_cast<const sqv_u2 *>(p + 5);            // (s0, seq: no more than is used -- a register of a pending request that is reused elsewhere makes the compiler wait for the request)
    asm volatile("" ::: "memory");
    r.q4 = p[4];
    r.q0 = p[0];
    r.q1 = p[1];
}
__device__ __forceinline__ double sqv_dbl(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }
__device__ __forceinline__ void sqv_rec_decode(const SqvRaw &r, SqvSpec &sp)
{
    sp.e1 = sqv_dbl(r.q0.x, r.q0.y); sp.de = sqv_dbl(r.q0.z, r.q0.w);
    sp.xw[0] = sqv_dbl(r.q1.x, r.q1.y); sp.xw[1] = sqv_dbl(r.q1.z, r.q1.w);
    sp.i = __builtin_amdgcn_readfirstlane((int)r.q4.x);
    sp.flags = __builtin_amdgcn_readfirstlane((int)r.q4.y);
    const int cz = __builtin_amdgcn_readfirstlane((int)r.q4.z), cw = __builtin_amdgcn_readfirstlane((int)r.q4.w);
    sp.c0 = (short)(cz & 0xffff); sp.c2 = (short)(cz >> 16); sp.c3 = (short)(cw & 0xffff); sp.nl = (short)(cw >> 16);
    sp.s0 = __builtin_amdgcn_readfirstlane((int)r.q5.x);
    sp.seq = __builtin_amdgcn_readfirstlane((int)r.q5.y);
    sp.pad0 = sp.pad1 = 0;
}

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

// the chain acc += t_k over the `cnt` terms packed at buf (cnt per LANE: lanes may walk different buffers), nmax = the largest
// cnt of the wavefront; eight zeros lie behind every lane's terms
__device__ __forceinline__ double sqv_chain(const double *buf, int cnt, int nmax)
{
    // blocks of eight terms, the next block requested before this one is added (one round trip in flight behind every eight
    // additions); the buffer reaches sixteen entries past the longest list (pmc_sqv_layout, pmc_sqe_layout), what lies behind a
    // lane's terms is masked
    double acc = 0.0;
    double cur[SQ_TERMS_PAD], nxt[SQ_TERMS_PAD];
#pragma unroll
    for (int q = 0; q < SQ_TERMS_PAD; q++) cur[q] = buf[q];
    for (int k = 0; k < nmax; k += SQ_TERMS_PAD) {
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) nxt[q] = buf[k + SQ_TERMS_PAD + q];
        const bool in = k < cnt;                                   // (a block that starts inside the terms ends inside their zero padding)
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) acc += in ? cur[q] : 0.0;
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) cur[q] = nxt[q];
    }
    return acc;
}

// every lane takes the value of the lane above it (wave_shl:1; the top lane takes 0)
__device__ __forceinline__ double sqv_shift_down(double u)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(u), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(u), 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// compute_energy_particle(system, i, ::VerletList) (src/atoms.jl:81-88 over src/neighbours.jl:422-424) at TWO positions at once:
// lanes 0-31 take the row's members 32 at a time at xa with species spa, lanes 32-63 the same members at xb with species spb;
// `other` (a swap partner, or -1) is seen with species sp_other at xb (src/moves.jl:159-167).  The two sums are chains of
// additions in row order held by lane 0 and lane 32: every trip of the inner loop adds the term the lane holds and shifts the
// terms one lane down (two DPP moves), ONE instruction stream for both chains and no LDS round trip (a term that is 0 -- a
// member beyond its pair's cutoff, a lane beyond the row -- leaves the sum as it is).  Called by a whole wavefront.
template <int D, int KIND>
__device__ __forceinline__ void sqv_energy2(const SqvArgs &s, const SqvView &v, int i, int spa, int spb, int other, int sp_other,
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
    double acc = 0.0;
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
        // (always 32 trips, no branch between them: the lanes beyond the row hold zeros)
#pragma unroll
        for (int k = 0; k < 32; k++) {
            acc += u;
            u = sqv_shift_down(u);
        }
    }
    ea = readlane_double(acc, 0);
    eb = readlane_double(acc, 32);
}

// every member of neighbours[i], and i: evaluations that read them no longer describe the state
__device__ __forceinline__ void sqv_mark_row(const SqvArgs &s, const SqvView &v, int i)
{
    const int lane = wave_lane();
    const int n = v.vcnt[i];
    const unsigned short *row = v.vrows + i * s.vcap;
    for (int q = lane; q < n; q += PMC_WAVE) v.touch[row[q]] = v.stamp;
    if (lane == 0) v.touch[i] = v.stamp;
}

// i leaves the lists of its neighbours (deleteat! in src/neighbours.jl:508-511: the order of what stays is kept); every
// neighbour is marked (MARK; the applier's rows were marked by the resolver when it handed the rebuild over).  One wavefront.
// Returns SQV_ERR_LIST if i was missing from one of them.
template <bool MARK = true>
__device__ __forceinline__ int sqv_leave_lists(const SqvArgs &s, const SqvView &v, int i)
{
    const int lane = wave_lane();
    const int vcap = s.vcap;
    int err = 0;
    unsigned short *row_i = v.vrows + i * vcap;
    const int ni = v.vcnt[i];
    if (vcap == 4 * SQV_HW) {
        // Rows of 40 ids: TWO LANES per neighbour, each with one half of that neighbour's row in registers (vcap / 4 words of
        // two ids, plus the word behind them).  A lane finds i in its half, the pair agrees on the position, every word behind it
        // takes the ids one place further on (v_alignbit; the word i sits in is merged under a mask), and the halves go back with
        // the same few stores they came with.  No loop over the rows, no dependent round trip per row: the 27 rows of a 2-D
        // list leave in ~2 000 cycles where one row per trip of a loop took 12 500 (LAB.md, round 6).
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
                if (hf == 0) { v.vcnt[j] = (unsigned short)(n - 1); if (MARK) v.touch[j] = v.stamp; }
            }
        }
    } else {
        // Other capacities (3-D systems): lane q holds neighbour q and the length of its list; then one ROW per trip of the loop,
        // the whole wavefront on it: the entries behind i move up one place
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
                    if (!mk) { err |= SQV_ERR_LIST; if (lane == 0) { v.ctl[20] = 1; v.ctl[21] = i; v.ctl[22] = j; v.ctl[23] = n; } }
                    const int p = mk ? (int)__builtin_ctzll(mk) : n;
                    if (lane >= p && lane + 1 < n) rj[lane] = (unsigned short)b;
                } else {
                    int p = -1;
                    for (int base = 0; base < n && p < 0; base += PMC_WAVE) {
                        const int k = base + lane;
                        const unsigned long long mk = __ballot(k < n && (int)rj[k] == i);
                        if (mk) p = base + (int)__builtin_ctzll(mk);
                    }
                    if (p < 0) { err |= SQV_ERR_LIST; p = n; if (lane == 0) { v.ctl[20] = 2; v.ctl[21] = i; v.ctl[22] = j; v.ctl[23] = n; } }
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
            if (lane < qn) { v.vcnt[jq] = (unsigned short)(nq - 1); if (MARK) v.touch[jq] = v.stamp; }
        }
    }
    sq_wave_sync();
    return err;
}

// update_cell_list!(i, c, c2, ::VerletList), src/neighbours.jl:468-476: filter! keeps the order of cells[c], push! appends to
// cells[c2].  One wavefront; both cells are marked (MARK: see sqv_leave_lists).
template <bool MARK = true>
__device__ __forceinline__ int sqv_cell_move(const SqvArgs &s, const SqvView &v, int i, int c, int c2)
{
    const int lane = wave_lane();
    const int ccap = s.ccap;
    int err = 0;
    unsigned short *rc = v.crows + c * ccap;
    const int n = v.ccnt[c];                                         // (<= ccap <= 64)
    const int a = (lane < n) ? (int)rc[lane] : -1;
    const int b = (lane + 1 < n) ? (int)rc[lane + 1] : 0;
    const unsigned long long mk = __ballot(a == i);
    if (!mk) { err |= SQV_ERR_LIST; if (lane == 0) { v.ctl[20] = 3; v.ctl[21] = i; v.ctl[22] = c; v.ctl[23] = n; } }
    const int p = mk ? (int)__builtin_ctzll(mk) : n;
    if (lane >= p && lane + 1 < n) rc[lane] = (unsigned short)b;
    const int n2 = v.ccnt[c2];
    if (n2 >= ccap) err |= SQV_ERR_CELL;
    if (lane == 0) {
        v.ccnt[c] = (unsigned short)(mk ? n - 1 : n);
        if (n2 < ccap) { v.crows[c2 * ccap + n2] = (unsigned short)i; v.ccnt[c2] = (unsigned short)(n2 + 1); }
        v.cs[i] = (unsigned short)c2;
        if (MARK) { v.ctouch[c] = v.stamp; v.ctouch[c2] = v.stamp; }
    }
    sq_wave_sync();
    return err;
}

// The scan that re-creates neighbours[i] at x (src/neighbours.jl:523-538): the cells around c2 in table order, every cell's
// vector in order; j is listed if it is within rcut + dr.  Lane l < cnt holds neighbour cell l and its length; candidate t of
// the scan finds its cell by counting the cells that end at or before t.  APPLY (the resolver, on the live lists): j goes into
// neighbours[i] and i is appended to neighbours[j].  Otherwise (an evaluator): the ids go to `lst`, and with WITH_E the pair
// terms at x go to `tb` in scan order and their chain is returned in e -- the sum compute_energy_particle takes over the new
// list (a listed particle beyond its pair's cutoff contributes the term 0).  Returns the length of the list.
template <int D, int KIND, bool APPLY, bool WITH_E>
__device__ __forceinline__ int sqv_scan(const SqvArgs &s, const SqvView &v, int i, int spi, const double *x, int c2, unsigned short *lst, double *tb,
                                        double &e, int &err)
{
    constexpr int NN = (D == 3) ? 27 : 9;
    const int lane = wave_lane();
    const int vcap = s.vcap, ccap = s.ccap, S = s.a.S;
    const int cc_any = (int)v.nbr[c2 * NN + min(lane, NN - 1)];          // (with the count: one round trip for the two)
    const int cnt = min((int)v.nbrcnt[c2], NN);
    const int cc = (lane < cnt) ? cc_any : 0;
    const int len = (lane < cnt) ? min((int)v.ccnt[cc], ccap) : 0;
    const int incl = sq_scan32(len);                                 // (inclusive; lanes >= cnt repeat the total)
    const int total = __builtin_amdgcn_readlane(incl, 31);
    unsigned short *row_i = v.vrows + i * vcap;
    const int tmax = 2 * s.tstride - 2 * SQ_TERMS_PAD;
    int fill = 0, tcnt = 0;
    // TWO trips of 64 candidates per turn of the loop, every load of both requested before anything of either is stored (a 2-D
    // neighbourhood is 80 candidates: the second trip's three dependent round trips ride behind the first's)
    for (int base = 0; base < total; base += 2 * PMC_WAVE) {
        int tt[2], jj[2];
        bool in[2];
        double uu[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int t = base + h * PMC_WAVE + lane;
            tt[h] = t;
            int l = 0;
#pragma unroll
            for (int m = 0; m < NN - 1; m++) l += (t >= __builtin_amdgcn_readlane(incl, m)) ? 1 : 0;
            l = min(l, cnt - 1);
            const int cc_l = __shfl(cc, l, PMC_WAVE);
            const int end_l = __shfl(incl, l, PMC_WAVE), len_l = __shfl(len, l, PMC_WAVE);
            const int at = min(max(t - (end_l - len_l), 0), ccap - 1);
            jj[h] = (int)v.crows[cc_l * ccap + at];                 // (a candidate beyond the total reads some valid id and is dropped below)
        }
        double r2[2];
        int sj[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            double xj[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < D; q++) xj[q] = v.pos[jj[h] * D + q];
            sj[h] = (WITH_E && S > 1) ? (int)v.species[jj[h]] : 0;
            r2[h] = sqv_dist2<D>(s, x, xj);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            in[h] = tt[h] < total && jj[h] != i && r2[h] < s.a.rcutoff2;
            uu[h] = 0.0;
            if (WITH_E && in[h]) {
                const DevModel &m = v.models[spi * S + sj[h]];
                if (!(r2[h] > m.rcut2)) uu[h] = (KIND == 4 ? dev_potential_dyn(r2[h], m) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2[h], m));
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int j = jj[h];
            const unsigned long long mk = __ballot(in[h]);
            if (in[h]) {
                const int p = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                if (APPLY) {
                    const int nj = v.vcnt[j];
                    if (p < vcap && nj < vcap) {
                        row_i[p] = (unsigned short)j;
                        v.vrows[j * vcap + nj] = (unsigned short)i;
                        v.vcnt[j] = (unsigned short)(nj + 1);
                        v.touch[j] = v.stamp;
                    } else {
                        err |= SQV_ERR_ROW;
                    }
                } else if (p < vcap) {
                    lst[p] = (unsigned short)j;
                }
            }
            fill += __popcll(mk);
            if (WITH_E) {
                const double u = uu[h];
                const unsigned long long mt = __ballot(u != 0.0);
                const int r = tcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mt >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mt, 0u));
                if (u != 0.0 && r < tmax) tb[r] = u;
                tcnt = min(tcnt + __popcll(mt), tmax);
            }
        }
    }
    if (WITH_E) {
        if (lane < SQ_TERMS_PAD) tb[tcnt + lane] = 0.0;
        sq_wave_sync();
        e = sqv_chain(tb, tcnt, tcnt);                               // (every lane runs the same chain)
        sq_wave_sync();
    }
    return fill;
}

// update_neighbour_list!(system, i, ::VerletList), src/neighbours.jl:494-543, on the LDS state; one wavefront (the resolver).
// Nothing unless the particle is dr / 2 from its position at its last update; then i leaves every list it is in (order kept), its
// cell vector is updated (filter! + push!), and its own list is rebuilt by scanning the neighbourhood in cell / vector order, i
// appended to every new neighbour's list.  Returns true if the lists were rebuilt.
// `last`: positions_at_last_update[i] (the caller has loaded it; it is x after a rebuild).
template <int D>
__device__ __forceinline__ bool sqv_update(const SqvArgs &s, const SqvView &v, int i, const double *x, double *last)
{
    const int lane = wave_lane();
    double d2 = 0.0;
#pragma unroll
    for (int q = 0; q < D; q++) {
        const double dx = x[q] - last[q];
        d2 = (q == 0) ? dx * dx : d2 + dx * dx;
    }
    if (d2 < s.a.dr2o4) return false;
    SQVT_COUNT(9);
    int err = sqv_leave_lists(s, v, i);
    SQVT(5);
    const int c = v.cs[i];
    const int c2 = sq_cell_of_fast<D>(s.a, s.invcell, x);
    if (c != c2) err |= sqv_cell_move(s, v, i, c, c2);
    double dummy = 0.0;
    const int fill = sqv_scan<D, 4, true, false>(s, v, i, 0, x, c2, nullptr, nullptr, dummy, err);
    if (lane == 0) {
        v.vcnt[i] = (unsigned short)min(fill, s.vcap);
        v.touch[i] = v.stamp;
        v.ctouch[c2] = v.stamp;
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

// One Displacement step of the chain on the current LDS state, by the resolver, as the literal loop runs it (ref_disp_perform /
// ref_disp_revert): e1, x += d, update_neighbour_list!, e2, accept or revert (+ update_neighbour_list! again).  Returns accepted.
// have_e1: the step's evaluation stands as far as e1 goes (nobody has touched i or its row since it began) and its trial
// position rebuilds the list: e1 is taken, e2 follows the rebuild.
template <int D, int KIND>
__device__ __forceinline__ bool sqv_disp_serial(const SqvArgs &s, const SqvView &v, double *tb, int i, const SqGen &ge, double T, double &E,
                                                unsigned long long &nreb, bool have_e1, double e1_in)
{
    const int lane = wave_lane();
    double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0}, last[3] = {0, 0, 0};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // (the resolver's own earlier stores to positions_at_last_update)
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
    if (lane == 0) v.ctouch[v.cs[i]] = v.stamp;
    double e1 = e1_in, e2 = 0.0, dummy;
    bool have_e2 = false;
    if (!have_e1) {
        // both sums over the row as it is: right unless the trial position rebuilds the list (then e2 is taken again)
        sqv_energy2<D, KIND>(s, v, i, spi, spi, -1, 0, x, xn, e1, e2);
        have_e2 = true;
    }
    if (lane < D) v.pos[i * D + lane] = sqv_sel3(xn, lane);
    sq_wave_sync();
    SQVT(4);
    if (sqv_update<D>(s, v, i, xn, last)) {
        nreb++;
        have_e2 = false;
    }
    if (!have_e2) sqv_energy2<D, KIND>(s, v, i, spi, spi, -1, 0, xn, xn, e2, dummy);
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

// An evaluator's work on one step: everything the resolver needs to take the step as it is, provided nobody touches its inputs.
// HAVE_E1 (the second opinion on a step whose first evaluation stands as far as e1 goes -- nobody has touched i or its row -- and
// whose trial position rebuilds the list): e1 is taken from it, e2 follows the scan.
template <int D, int KIND, bool HAVE_E1 = false>
__device__ __forceinline__ void sqv_evaluate(const SqvArgs &s, const SqvView &v, double *tb, const SqGen &ge, double T, SqvSpec *out, unsigned short *lst,
                                             int s0, int seq, double e1_in = 0.0)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int N = a.N, S = a.S;
    SqvSpec sp;
    sp.e1 = sp.de = 0.0;
    sp.xw[0] = sp.xw[1] = sp.xw[2] = sp.xl[0] = sp.xl[1] = sp.xl[2] = 0.0;
    sp.i = 0; sp.flags = SQV_SERIAL | (ge.m << 8);
    sp.c0 = sp.c2 = sp.c3 = sp.nl = 0;
    sp.s0 = s0; sp.seq = 0; sp.pad0 = sp.pad1 = 0;
    if (s.pool.kind[ge.m] == 0) {
        const int i = (int)mulhi32(ge.r1, (uint32_t)N);
        double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0}, last[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) last[q] = __hip_atomic_load(&v.last[i * D + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (global, past the L1: in flight behind the sums)
#pragma unroll
        for (int q = 0; q < D; q++) {
            x[q] = v.pos[i * D + q];
            xn[q] = x[q] + ge.d[q];
            xr[q] = xn[q] + (-ge.d[q]);
        }
        const int spi = (S > 1) ? min((int)v.species[i], S - 1) : 0;
        double e1 = e1_in, e2 = 0.0;
        bool sums = true;
        if (HAVE_E1) {
            double dd = 0.0;
#pragma unroll
            for (int q = 0; q < D; q++) { const double dx = xn[q] - last[q]; dd = (q == 0) ? dx * dx : dd + dx * dx; }
            sums = dd < a.dr2o4;                                     // (the list stays after all: both sums over the row)
        }
        if (sums) sqv_energy2<D, KIND>(s, v, i, spi, spi, -1, 0, x, xn, e1, e2);
        double d2 = 0.0, d2r = 0.0, d2b = 0.0;
        bool chg = false;
#pragma unroll
        for (int q = 0; q < D; q++) {
            const double dx = xn[q] - last[q], dr_ = xr[q] - last[q], db = xr[q] - xn[q];
            d2 = (q == 0) ? dx * dx : d2 + dx * dx;
            d2r = (q == 0) ? dr_ * dr_ : d2r + dr_ * dr_;
            d2b = (q == 0) ? db * db : d2b + db * db;
            chg |= sq_bits_differ(xr[q], x[q]);
        }
        int fl = (ge.m << 8) | SQV_DISP;
        sp.i = i;
        sp.e1 = e1;
        sp.c0 = (short)v.cs[i];
        if (d2 < a.dr2o4) {
            // the trial position keeps the list: e1 and e2 are sums over the same row, and the step is decided here
            const bool fin = !(isinf(e1) || isinf(e2));
            const bool ok = ge.lnu < -(e2 - e1) / T;
            if (fin) { fl |= SQV_FINITE; sp.de = e2 - e1; }
            if (ok) fl |= SQV_OK | SQV_CHG;
            else if (chg) fl |= SQV_CHG | ((d2r < a.dr2o4) ? 0 : SQV_SERIAL);   // (an inexact revert that rebuilds the list: run alone)
#pragma unroll
            for (int q = 0; q < D; q++) sp.xw[q] = ok ? xn[q] : xr[q];
        } else {
            // the trial position rebuilds the list: the scan around its cell, e2 over the new list, the decision; a rejected step
            // reverts, and rebuilds again if the reverted position is dr / 2 from the trial position (now the last update)
            fl |= SQV_REBUILD;
            int err = 0;
            const int c2 = min(max(sq_cell_of_fast<D>(a, s.invcell, xn), 0), s.ncell - 1);   // (the clamp only ever bites on a position that raced with the resolver)
            int nl = sqv_scan<D, KIND, false, true>(s, v, i, spi, xn, c2, lst, tb, e2, err);
            const bool fin = !(isinf(e1) || isinf(e2));
            const bool ok = ge.lnu < -(e2 - e1) / T;
            if (fin) { fl |= SQV_FINITE; sp.de = e2 - e1; }
            int c3 = c2;
            if (ok) {
                fl |= SQV_OK;
            } else if (!(d2b < a.dr2o4)) {
                fl |= SQV_REB2;
                c3 = min(max(sq_cell_of_fast<D>(a, s.invcell, xr), 0), s.ncell - 1);
                double dummy = 0.0;
                nl = sqv_scan<D, KIND, false, false>(s, v, i, spi, xr, c3, lst, tb, dummy, err);
            }
            if (nl <= s.vcap) fl |= SQV_LIST | SQV_CHG;              // (a longer list: the resolver runs the step and reports the overflow)
#pragma unroll
            for (int q = 0; q < D; q++) {
                sp.xw[q] = ok ? xn[q] : xr[q];
                sp.xl[q] = (fl & SQV_REB2) ? xr[q] : xn[q];
            }
            sp.c2 = (short)c2; sp.c3 = (short)c3; sp.nl = (short)min(nl, s.vcap);
        }
        sp.flags = fl;
    }
    // the record, then (everything of it and of the list having landed) its sequence number
    if (lane == 0) *out = sp;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) sqv_st(&out->seq, seq);
}

// a mark older than `window` steps (the step number + 1 is `now`): set before the evaluation in question began
__device__ __forceinline__ bool sqv_fresh(int tv, int now, int window) { return ((now - tv) & 0xffff) > window; }

template <int D, int KIND>
__global__ __launch_bounds__(SQV_NW * 64) void k_sq_verlet(SqvArgs s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sqv_smem[];
    constexpr int NN = (D == 3) ? 27 : 9;
    constexpr int NW = SQV_NW, NB = SQV_NB;
    const RefArgs &a = s.a;
    SqvChain &ch = s.chains[blockIdx.x];
    const int tid = threadIdx.x, nth = NW * 64, lane = wave_lane(), wv = tid >> 6;
    const int N = a.N, S = a.S, vcap = s.vcap, ccap = s.ccap;
    SqvView v;
    v.pos = reinterpret_cast<double *>(sqv_smem);
    v.last = ch.last_work;
    v.species = sqv_smem + s.o_species;
    v.touch = reinterpret_cast<unsigned short *>(sqv_smem + s.o_touch);
    v.ctouch = reinterpret_cast<unsigned short *>(sqv_smem + s.o_ctouch);
    v.stamp = 0;
    v.cs = reinterpret_cast<unsigned short *>(sqv_smem + s.o_cs);
    v.vcnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_vcnt);
    v.vrows = reinterpret_cast<unsigned short *>(sqv_smem + s.o_vrows);
    v.ccnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_ccnt);
    v.crows = reinterpret_cast<unsigned short *>(sqv_smem + s.o_crows);
    v.nbr = reinterpret_cast<unsigned short *>(sqv_smem + s.o_nbr);
    v.nbrcnt = reinterpret_cast<unsigned short *>(sqv_smem + s.o_nbrcnt);
    v.gen = reinterpret_cast<SqGen *>(sqv_smem + s.o_gen);
    v.ctl = reinterpret_cast<int *>(sqv_smem + s.o_ctl);
    int *const p_err = v.ctl, *const p_done = v.ctl + 1, *const p_gen = v.ctl + 2, *const p_cnt = v.ctl + 4;   // [4 .. 19]: attempted, accepted per move
    // The APPLIER (wavefront 8): the list edits of a rebuild the resolver takes from an evaluator's list -- i leaves its old
    // neighbours' rows, joins those of the new list, moves between cell vectors -- are handed over as a JOB and applied behind
    // the resolver's back, in the order they were handed over.  The resolver marks everything the job will edit when it hands
    // it over (so every evaluation that has read or will read a row in the making is void, as if the edit had been made on the
    // spot), and waits for the applier only before a step it has to run itself on the lists.  [3]: jobs applied; [24]: jobs
    // handed over; [25]: the resolver is through; [28 .. 31]: the step (+ 1) of job k mod 4; [32 ..]: the jobs, six words each
    constexpr int SQV_JQ = 4;
    int *const p_applied = v.ctl + 3, *const p_posted = v.ctl + 24, *const p_fin = v.ctl + 25, *const p_jstep = v.ctl + 28, *const p_job = v.ctl + 32;
    // The SECOND OPINION (wavefront 12): a step whose evaluation does not stand is evaluated again on the state as it is, by a
    // wavefront that has nothing else to do (and none of the resolver's registers to keep alive); [26]: the step (+ 1) the
    // resolver asks for, [27]: the step (+ 1) whose record is ready
    int *const p_ask = v.ctl + 26, *const p_told = v.ctl + 27;
    static_assert(NW > 12, "wavefront 8 applies the rebuilds, wavefront 12 evaluates again");
    SqvSpec *const spec0 = reinterpret_cast<SqvSpec *>(sqv_smem + s.o_spec);            // [NB]: evaluator e's record
    unsigned short *const lst0 = reinterpret_cast<unsigned short *>(sqv_smem + s.o_lst);   // [NB][vcap]: ... and its list
    const int ev = (wv & 3) ? (wv & 3) - 1 + 3 * (wv >> 2) : -1;          // evaluator 0 .. NB - 1, or none
    double *const tb = reinterpret_cast<double *>(sqv_smem + s.o_terms) + (size_t)max(ev, 0) * 2 * s.tstride;   // (an evaluator's terms of the sum over a rebuilt list)
    double *const tbr = reinterpret_cast<double *>(sqv_smem + s.o_terms);
    {
        DevModel *lm = reinterpret_cast<DevModel *>(sqv_smem + s.o_models);
        const int *src = reinterpret_cast<const int *>(a.models);
        int *dst = reinterpret_cast<int *>(lm);
        for (int k = tid; k < S * S * (int)(sizeof(DevModel) / sizeof(int)); k += nth) dst[k] = src[k];
        v.models = lm;
    }
    // -- the chain's state into LDS (rows, cell vectors and marks zero-filled first: whatever an evaluator reads is a valid id)
    if (tid < 64) v.ctl[tid] = 0;
    for (int k = tid; k < (N * vcap + 1) / 2; k += nth) reinterpret_cast<unsigned int *>(v.vrows)[k] = 0u;
    for (int k = tid; k < (s.ncell * ccap + 1) / 2; k += nth) reinterpret_cast<unsigned int *>(v.crows)[k] = 0u;
    for (int k = tid; k < (N + 1) / 2; k += nth) reinterpret_cast<unsigned int *>(v.touch)[k] = 0u;
    for (int k = tid; k < (s.ncell + 1) / 2; k += nth) reinterpret_cast<unsigned int *>(v.ctouch)[k] = 0u;
    for (int k = tid; k < NB * (int)(sizeof(SqvSpec) / 4); k += nth) reinterpret_cast<unsigned int *>(spec0)[k] = 0u;
    for (int k = tid; k < NB * vcap / 2; k += nth) reinterpret_cast<unsigned int *>(lst0)[k] = 0u;
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
    if (lerr) atomicOr(p_err, lerr);
    if (ch.sp_ids_work) for (int k = tid; k < N; k += nth) { ch.sp_ids_work[k] = ch.sp_ids[k]; ch.sp_heads_work[k] = ch.sp_heads[k]; }
    const int nsteps = (int)s.nsteps;                            // (a launch has at most 2^20 steps: the counters above are ints)
    // the words and operations of k_ref_ops for step t of the launch, into the ring of random parts
    auto generate = [&](int t) {
        const unsigned long long g = ch.step0 + (unsigned long long)t;
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
        v.gen[t & (SQV_GEN - 1)] = o;
    };
    if (tid < SQV_GEN && tid < nsteps) generate(tid);           // blocks 0 and 1 of the random parts
    if (tid == 0) *p_gen = 2;
    __syncthreads();
    const double T = ch.temperature;
    double E = ch.energy;
    unsigned long long nser = 0, nreb = 0, napp = 0, nwait = 0;
    const bool dead0 = *p_err != 0;
    if (dead0) {
        // (a row longer than the LDS holds: nothing runs)
    } else if (wv == 0) {
        // ---- the RESOLVER: the steps in order
        // (the loop bound as a value the compiler cannot re-load from the kernel arguments: under register pressure it then keeps
        // it in a lane of a vector register, where a scalar load and its wait -- for every LDS operation in flight too -- stood
        // at the end of every step)
        int nsteps_r = nsteps;
        asm volatile("" : "+s"(nsteps_r));
        int slot = 0;
        // The record of a step in one round trip, requested while the step before is resolved (the evaluators are SQV_LOOK steps
        // ahead: it is nearly always there, and is asked for again if it is not): sqv_rec_request
        SqvRaw raw;
        int zero;
        asm volatile("v_mov_b32 %0, 0" : "=v"(zero));
        sqv_rec_request(spec0, zero, raw);
        // The loop is two loops: the inner one takes the steps that ARE their evaluation (seven in ten) and nothing else -- a
        // record, one round trip for the stamp of i and its row, the stores -- and leaves at the first step that needs a list
        // edit or a re-evaluation; that step is handled behind it, and the inner loop starts again.
        int t = 0;
        unsigned int cntv = 0;                                   // per-move counters of the launch, one per lane (a launch has < 2^32 steps)
        bool stop = false, fin = false;
        int posted = 0;                                          // jobs handed to the applier
        while (!stop && !fin) {
            const SqvSpec *rec = spec0;
            SqvSpec sp;
            int now = 0, window = 0, m = 0, i = 0;
            bool clean = false, clean_known = false;
            while (true) {
                SQVM("step_begin");
                rec = spec0 + slot;
                sqv_rec_decode(raw, sp);
                SQVM("rec_ready");
                SQVT(1);
                if (sp.seq != t + 1) SQVT_COUNT(14);
                while (sp.seq != t + 1) {
                    nwait++;
                    SQVT_COUNT(13);
                    if (sqv_ld(p_err)) { stop = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                    sqv_rec_request(rec, zero, raw);
                    sqv_rec_decode(raw, sp);
                }
                if (stop) break;
                sqv_rec_request(spec0 + ((slot + 1 == NB) ? 0 : slot + 1), zero, raw);     // the next step's, in flight behind this step
                SQVT(0);
                now = t + 1; window = t - sp.s0;
                v.stamp = (unsigned short)now;
                m = (sp.flags >> 8) & 7;
                clean_known = false;
                if ((sp.flags & (SQV_DISP | SQV_SERIAL | SQV_REBUILD)) != SQV_DISP) break;
                i = sp.i;
                // (one round trip: the mark of i, and its row in case the step stores a position)
                const int rj_ = (int)v.vrows[i * vcap + min(lane, vcap - 1)];      // (every lane: a read under a lane mask is scheduled behind the wait for the two below)
                asm volatile("" ::: "memory");
                const int ti = v.touch[i];
                const int ni_ = v.vcnt[i];
                asm volatile("" : : "v"(rj_), "v"(ti), "v"(ni_));      // (all three requested before the first is waited for)
                clean = sqv_fresh(ti, now, window);
                clean_known = true;
                if (!clean) break;
                // nobody has touched i or its row since the evaluation began: the evaluation is the step
                const bool ok = (sp.flags & SQV_OK) != 0;
                if (sp.flags & SQV_FINITE) { E += sp.de; if (!ok) E -= sp.de; }     // src/moves.jl:17,79
                if (sp.flags & SQV_CHG) {
                    // (the stored position is within dr / 2 of the last update: update_neighbour_list! returns at once)
                    if (D == 3) sp.xw[2] = rec->xw[2];
                    if (lane < D) v.pos[i * D + lane] = sqv_sel3(sp.xw, lane);
                    if (lane < ni_) v.touch[rj_] = v.stamp;
                    for (int q = PMC_WAVE + lane; q < ni_; q += PMC_WAVE) v.touch[v.vrows[i * vcap + q]] = v.stamp;
                    if (lane == 0) { v.touch[i] = v.stamp; v.ctouch[sp.c0] = v.stamp; }
                }
                SQVM("clean_end");
                SQVT(2);
                cntv += (lane == m || (ok && lane == 8 + m)) ? 1u : 0u;     // lane q: attempts of move q; lane 8 + q: its accepted steps
                // the evaluators are told that the step is over: the LDS takes this store behind everything the step stored there
                asm volatile("" ::: "memory");
                if (lane == 0) sqv_st_relaxed(p_done, now);
                t++;
                slot = (slot + 1 == NB) ? 0 : slot + 1;
                if (t >= nsteps_r) { fin = true; break; }
            }
            if (stop || fin) break;
            // ---- step t needs a list edit or a re-evaluation
            const unsigned short *lst = lst0 + slot * vcap;
            bool ok = false;
            bool handed = false;
            if (sp.flags & SQV_DISP) {
                i = sp.i;
                for (int pass = 0; ; pass++) {
                    // Everything the hand-over of a rebuild reads, in TWO round trips (every address is in the record): the mark of
                    // i, its cell, its row and the new list (for the marks), the cells the scans read; then those cells' marks.
                    bool take = false;
                    int c_old = 0, ni = 0, rj = 0, lj = 0, applied_seen = 0;
                    if (sp.flags & SQV_LIST) {
                        const int ti = v.touch[i];
                        const int cnt2 = v.nbrcnt[sp.c2], cnt3 = v.nbrcnt[sp.c3];
                        const int nb2 = v.nbr[sp.c2 * NN + min(lane, NN - 1)], nb3 = v.nbr[sp.c3 * NN + min(lane, NN - 1)];
                        c_old = v.cs[i];
                        ni = v.vcnt[i];
                        rj = v.vrows[i * vcap + min(lane, vcap - 1)];
                        lj = lst[min(lane, vcap - 1)];
                        applied_seen = sqv_ld_relaxed(p_applied);
                        sp.xw[2] = rec->xw[2]; sp.xl[0] = rec->xl[0]; sp.xl[1] = rec->xl[1]; sp.xl[2] = rec->xl[2];      // (the rest of the record: the evaluator leaves it alone until this step is over)
                        asm volatile("" : : "v"(ti), "v"(cnt2), "v"(cnt3), "v"(nb2), "v"(nb3), "v"(c_old), "v"(ni), "v"(rj), "v"(lj), "v"(applied_seen));
                        const int t2 = v.ctouch[nb2], t3 = v.ctouch[nb3];
                        clean = sqv_fresh(ti, now, window);
                        bool bad = !clean;
                        bad |= lane < cnt2 && !sqv_fresh(t2, now, window);
                        bad |= (sp.flags & SQV_REB2) && lane < cnt3 && !sqv_fresh(t3, now, window);
                        take = __ballot(bad) == 0ull;
                    } else if (!clean_known) {
                        clean = sqv_fresh(v.touch[i], now, window);
                    }
                    if (take) {
                        // ... and nobody has touched the cells its scan read: the rebuild is taken from the evaluation's list.
                        // Net effect of update_neighbour_list! at the trial position (and again at the reverted one, REB2): i has
                        // left its old neighbours' lists and is appended to those of the final list; the cell vectors took
                        // every move in turn.  The edits are the applier's; here: the marks they need, the position, the job.
                        // (i's own row and cell are nobody's job in the making: a job that edits them has marked i.)
                        ok = (sp.flags & SQV_OK) != 0;
                        if (sp.flags & SQV_FINITE) { E += sp.de; if (!ok) E -= sp.de; }
                        while (posted - applied_seen >= SQV_JQ) {
                            if (sqv_ld(p_err)) { stop = true; break; }
                            __builtin_amdgcn_s_sleep(1);
                            applied_seen = sqv_ld(p_applied);
                        }
                        if (stop) break;
                        const int c = c_old;
                        const unsigned short *row_i = v.vrows + i * vcap;
                        const int nl = sp.nl;
                        if (lane < ni) v.touch[rj] = v.stamp;
                        if (lane < nl) v.touch[lj] = v.stamp;
                        for (int q = PMC_WAVE + lane; q < ni; q += PMC_WAVE) v.touch[row_i[q]] = v.stamp;
                        for (int q = PMC_WAVE + lane; q < nl; q += PMC_WAVE) v.touch[lst[q]] = v.stamp;
                        if (lane == 0) {
                            v.touch[i] = v.stamp;
                            v.ctouch[c] = v.stamp;
                            v.ctouch[sp.c2] = v.stamp;
                            v.ctouch[sp.c3] = v.stamp;
                            int *job = p_job + 6 * (posted & (SQV_JQ - 1));
                            job[0] = i; job[1] = slot; job[2] = sp.c2; job[3] = sp.c3; job[4] = nl;
                            p_jstep[posted & (SQV_JQ - 1)] = now;
                        }
                        if (lane < D) {
                            v.pos[i * D + lane] = sqv_sel3(sp.xw, lane);
                            v.last[i * D + lane] = sqv_sel3(sp.xl, lane);
                        }
                        posted++;
                        handed = true;
                        nreb += (sp.flags & SQV_REB2) ? 2 : 1;
                        napp++;
                        SQVT_COUNT(10);
                        SQVT(3);
                        break;
                    }
                    // the evaluation does not stand (or the step is one an evaluation cannot settle): the lists as every job so
                    // far leaves them, then ...
                    if (pass == 0) { nser++; SQVT_COUNT(11); }
                    SQVT(7);
                    while (sqv_ld(p_applied) != posted) {
                        if (sqv_ld(p_err)) { stop = true; break; }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (stop) break;
                    SQVT(4);
                    if (pass == 0) {
                        // ... the step is evaluated AGAIN on the state as it stands, exactly as an evaluator would (by wavefront 12,
                        // into the record and the buffers of the step's own evaluator, which is through with them and far from
                        // its next step), and taken as above: a rebuilt list goes to the applier like any other
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (the resolver's own earlier stores to positions_at_last_update)
                        const int how = (clean && (sp.flags & SQV_REBUILD)) ? 3 : 1;       // (3: e1 of the first evaluation stands)
                        if (lane == 0) { *reinterpret_cast<volatile double *>(v.ctl + 56) = sp.e1; sqv_st(p_ask, 4 * now + how); }
                        while (sqv_ld(p_told) != 4 * now + how) __builtin_amdgcn_s_sleep(1);
                        SQVT(5);
                        SqvRaw rawb;
                        sqv_rec_request(rec, zero, rawb);
                        sqv_rec_decode(rawb, sp);
                        window = 0;
                        clean_known = false;
                        if ((sp.flags & (SQV_DISP | SQV_SERIAL | SQV_REBUILD)) == SQV_DISP) {
                            // (the list stays: the evaluation is the step)
                            ok = (sp.flags & SQV_OK) != 0;
                            if (sp.flags & SQV_FINITE) { E += sp.de; if (!ok) E -= sp.de; }
                            if (sp.flags & SQV_CHG) {
                                if (D == 3) sp.xw[2] = rec->xw[2];
                                if (lane < D) v.pos[i * D + lane] = sqv_sel3(sp.xw, lane);
                                sqv_mark_row(s, v, i);
                                if (lane == 0) v.ctouch[sp.c0] = v.stamp;
                            }
                            break;
                        }
                        if (sp.flags & SQV_LIST) continue;
                    }
                    // ... or (an inexact revert that re-lists the particle, a list longer than a row) run as the literal loop runs
                    // it, by wavefront 12 as well
                    if (lane == 0) { *reinterpret_cast<volatile double *>(v.ctl + 56) = E; sqv_st(p_ask, 4 * now + 2); }
                    while (sqv_ld(p_told) != 4 * now + 2) __builtin_amdgcn_s_sleep(1);
                    E = *reinterpret_cast<volatile double *>(v.ctl + 58);
                    ok = v.ctl[60] != 0;
                    break;
                }
                if (stop) break;
            } else {
                // DiscreteSwap (src/moves.jl:137-207): four sums, species exchange, SpeciesList fix-up; the SpeciesList
                // stays in global memory (the launch's working copy: this wavefront is its only reader and writer)
                nser++;
                while (sqv_ld(p_applied) != posted) {
                    if (sqv_ld(p_err)) { stop = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (stop) break;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the resolver's own earlier stores to the SpeciesList)
                const SqGen ge = v.gen[t & (SQV_GEN - 1)];
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
                    sqv_energy2<D, KIND>(s, v, i, spi, spj, j, spi, xi, xi, e1i, e2i);
                    sqv_energy2<D, KIND>(s, v, j, spj, spi, i, spj, xj, xj, e1j, e2j);
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
                            v.ctouch[v.cs[i]] = v.stamp;
                            v.ctouch[v.cs[j]] = v.stamp;
                            // update_species_list!, src/moves.jl:175-179
                            const int hi_ = ch.sp_heads_work[i], hj_ = ch.sp_heads_work[j];
                            ch.sp_ids_work[oa + hi_] = j;
                            ch.sp_ids_work[ob + hj_] = i;
                            ch.sp_heads_work[i] = hj_;
                            ch.sp_heads_work[j] = hi_;
                        }
                    } else {
                        E -= de;                                             // src/moves.jl:206
                    }
                }
                SQVT(8);
            }
            // (a row or cell vector that overflowed in this step ends the launch here: the state is no longer the chain's, and
            // nothing of it is written back)
            if (sqv_ld(p_err)) break;
            SQVM("step_end");
            cntv += (lane == m || (ok && lane == 8 + m)) ? 1u : 0u;
            // the evaluators are told that the step is over: the LDS takes this store behind everything the step stored there.
            // (What it stored in global memory -- positions_at_last_update, the SpeciesList -- may still be on its way: an
            // evaluator that reads a stale positions_at_last_update[i] reads it for a particle this step has marked.)
            asm volatile("" ::: "memory");
            if (lane == 0) {
                if (handed) sqv_st_relaxed(p_posted, posted);       // (behind the job and its marks, before the step counter)
                sqv_st_relaxed(p_done, now);
            }
            t++;
            slot = (slot + 1 == NB) ? 0 : slot + 1;
            if (t >= nsteps_r) fin = true;
        }
        if (lane < 16) p_cnt[lane] = (int)cntv;
        asm volatile("" ::: "memory");
        if (lane == 0) sqv_st(p_fin, 1);
    } else if (wv == 4) {
        // ---- the GENERATOR: block B of the random parts takes the place of block B - 2 once the resolver is through with it
        for (int B = 2; (B << 6) < nsteps; B++) {
            bool stop = false;
            while (true) {
                const sqv_u4 c = *reinterpret_cast<const volatile sqv_u4 *>(v.ctl);
                if (c.x) { stop = true; break; }
                if ((int)c.y >= ((B - 1) << 6)) break;
                __builtin_amdgcn_s_sleep(60);
            }
            if (stop) break;
            const int t = (B << 6) + lane;
            if (t < nsteps) generate(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) sqv_st(p_gen, B + 1);
        }
    } else if (wv == 8) {
        // ---- the APPLIER: the jobs in the order they were handed over
        int applied = 0;
        while (true) {
            const int fin = sqv_ld(p_fin);                      // (read before the count: set behind the last job)
            const int pw = sqv_ld(p_posted);
            if (pw != applied) {
                const int *job = p_job + 6 * (applied & (SQV_JQ - 1));
                const int i = __builtin_amdgcn_readfirstlane(job[0]), jslot = __builtin_amdgcn_readfirstlane(job[1]);
                const int c2 = __builtin_amdgcn_readfirstlane(job[2]), c3 = __builtin_amdgcn_readfirstlane(job[3]), nl = __builtin_amdgcn_readfirstlane(job[4]);
                const unsigned short *lst = lst0 + jslot * vcap;
                int err = sqv_leave_lists<false>(s, v, i);
                const int c = v.cs[i];
                if (c != c2) err |= sqv_cell_move<false>(s, v, i, c, c2);
                if (c3 != c2) err |= sqv_cell_move<false>(s, v, i, c2, c3);
                unsigned short *row_i = v.vrows + i * vcap;
                for (int q = lane; q < nl; q += PMC_WAVE) {
                    const int j = lst[q];
                    const int nj = v.vcnt[j];
                    if (nj < vcap) {
                        v.vrows[j * vcap + nj] = (unsigned short)i;
                        v.vcnt[j] = (unsigned short)(nj + 1);
                    } else {
                        err |= SQV_ERR_ROW;
                    }
                    row_i[q] = (unsigned short)j;
                }
                if (lane == 0) v.vcnt[i] = (unsigned short)nl;
                if (__ballot(err != 0)) {
                    int e = err;
#pragma unroll
                    for (int sft = 32; sft >= 1; sft >>= 1) e |= __shfl_xor(e, sft, PMC_WAVE);
                    if (lane == 0) atomicOr(p_err, e);
                }
                applied++;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) sqv_st(p_applied, applied);
                continue;
            }
            if (fin) break;
            __builtin_amdgcn_s_sleep(1);
        }
    } else if (wv == 12) {
        // ---- the SECOND OPINION: step t on the state the resolver has stopped at
        int told = 0;
        while (true) {
            const int fin = sqv_ld(p_fin);
            const int ask = sqv_ld(p_ask);
            if (ask != told) {
                const int t = (ask >> 2) - 1, slot = t % NB;
                const SqGen ge = v.gen[t & (SQV_GEN - 1)];
                if ((ask & 3) == 1) {
                    sqv_evaluate<D, KIND>(s, v, tbr + (size_t)slot * 2 * s.tstride, ge, T, spec0 + slot, lst0 + slot * vcap, t, t + 1);
                } else if ((ask & 3) == 3) {
                    const double e1 = *reinterpret_cast<volatile double *>(v.ctl + 56);
                    sqv_evaluate<D, KIND, true>(s, v, tbr + (size_t)slot * 2 * s.tstride, ge, T, spec0 + slot, lst0 + slot * vcap, t, t + 1, e1);
                } else {
                    // the Displacement step as the literal loop runs it, on the live lists
                    v.stamp = (unsigned short)(t + 1);
                    double E12 = *reinterpret_cast<volatile double *>(v.ctl + 56);
                    bool ok = false;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (this wavefront's own earlier stores to positions_at_last_update)
                    ok = sqv_disp_serial<D, KIND>(s, v, tb, (int)mulhi32(ge.r1, (uint32_t)N), ge, T, E12, nser, false, 0.0);
                    if (lane == 0) { *reinterpret_cast<volatile double *>(v.ctl + 58) = E12; v.ctl[60] = ok ? 1 : 0; }
                }
                told = ask;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0) sqv_st(p_told, told);
                continue;
            }
            if (fin) break;
            __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) v.ctl[61] = (int)nser;                 // (here: the rebuilds of the steps this wavefront ran as the literal loop does)
        nser = 0;
    } else if (ev >= 0) {
        // ---- an EVALUATOR: every NB-th step, begun once the resolver is within SQV_LOOK steps of it
        // (An evaluator is idle nine tenths of the time.  Twelve of them polling the control words every hundred cycles kept the
        // LDS busy enough to double the cost of every step of the resolver: one 16-byte request per poll, and a sleep that
        // follows the distance to go.)
        bool stop = false;
        for (int t = ev; t < nsteps && !stop; t += NB) {
            int s0 = 0;
            while (true) {
                const sqv_u4 c = *reinterpret_cast<const volatile sqv_u4 *>(v.ctl);      // err, done, random parts, -
                if (c.x) { stop = true; break; }
                s0 = (int)c.y;
                if (s0 >= t - SQV_LOOK && (int)c.z > (t >> 6)) {
                    // jobs the applier has not finished: the evaluation can vouch for the lists only up to the step before the
                    // oldest of them (its step is looked up, and the count read again: had it moved on, the entry may be a later job's)
                    asm volatile("" ::: "memory");
                    const int pw = sqv_ld_relaxed(p_posted);
                    if (pw == (int)c.w) break;
                    asm volatile("" ::: "memory");
                    const int js = sqv_ld_relaxed(p_jstep + ((int)c.w & (SQV_JQ - 1)));
                    asm volatile("" ::: "memory");
                    const int ap2 = sqv_ld_relaxed(p_applied);
                    if (ap2 == (int)c.w) {
                        s0 = min(s0, js - 1);
                        if (s0 >= t - SQV_HOLD) break;              // (and the applier is through with this evaluator's last list)
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                if (t - SQV_LOOK - s0 > 2) __builtin_amdgcn_s_sleep(24); else __builtin_amdgcn_s_sleep(3);
            }
            if (stop) break;
            asm volatile("" ::: "memory");
#ifdef PMC_DEBUG_TIME
            const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
            sqv_evaluate<D, KIND>(s, v, tb, v.gen[t & (SQV_GEN - 1)], T, spec0 + ev, lst0 + ev * vcap, s0, t + 1);
#ifdef PMC_DEBUG_TIME
            if (blockIdx.x == 0 && ev == 0 && lane == 0) {       // evaluator 0's clock: 12 plain evaluations, 13 with a rebuilt list, 14 the latter's number
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                const unsigned long long te1 = __builtin_amdgcn_s_memtime();
                const bool rb = (spec0[ev].flags & SQV_REBUILD) != 0;
                g_sqvt[12] += rb ? 0 : te1 - te0;
            }
#endif
        }
    }
    __syncthreads();
    if (*p_err != 0) {
        // a row or cell vector outgrew the LDS (or the lists were inconsistent): nothing is written back -- the swaps of this launch
        // edited only the working copy of the SpeciesList -- and the host runs these steps on the literal loop
        if (tid == 0) { ch.err = *p_err; for (int q = 0; q < 4; q++) ch.dbg[q] = v.ctl[20 + q]; }
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
        ch.step0 += (unsigned long long)nsteps;
        for (int q = 0; q < 8; q++) { ch.att[q] += (unsigned long long)p_cnt[q]; ch.acc[q] += (unsigned long long)p_cnt[8 + q]; }
        ch.n_serial += nser; ch.n_rebuild += nreb + (unsigned long long)v.ctl[61]; ch.n_batches += nwait; ch.n_applied += napp;
    }
}
