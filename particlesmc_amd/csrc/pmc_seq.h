// pmc_seq.h -- the reference's strictly sequential chain (src/moves.jl:11-20,57-81,187-207), run FAST on gfx950.
//
// The chain of the reference takes one particle per step and every step sees the state all earlier steps left
// (SURVEY F5).  k_ref_ops (pmc_kernels.h) runs it literally, one wavefront, one step after the other: 67 us per step.
// This file keeps every bit of that chain -- positions, head/list/cs, accept decisions, system.energy -- and gets its
// parallelism from two facts:
//
//  1. The random part of a step (move, particle word, displacement, ln u) is a pure function of the step counter
//     (Philox keyed by step), so it is known for any number of steps ahead (k_sq_gen).
//  2. Step s reads only the 3^d neighbour cells of the cells its particle is in before and after the move, and writes
//     only those one or two cells (position, cell membership, list order).  Two steps whose cells are more than one cell
//     apart commute EXACTLY: same e1, e2, same accept decision, same list order, whatever the order they run in.
//
// A WINDOW of K pending steps is therefore evaluated at once, one wavefront per step, against the current state
// (k_sq_round_a: "eval"), a K x K bit matrix records which earlier pending steps a step could interact with
// ("pairs"), and a step COMMITS in this round iff no earlier pending step that could interact with it changes the
// state or is itself held back (k_sq_commit).  Steps that are held back stay in the window and are evaluated again
// next round against the state that then exists; the window is refilled in step order (k_sq_fill).  The oldest pending
// step never has a predecessor, so every round makes progress, and the result is the sequential chain's, bit for bit:
//
//   * e1 / e2 are FP64 sums in LinkedIterator order (src/neighbours.jl:332-377).  Terms that are exactly 0 (the self term,
//     pairs beyond their cutoff: src/atoms.jl:68,72) do not change a running FP64 sum, so only the non-zero terms are
//     added, in iterator order.  For the same reason the place of particle i itself in the iteration is irrelevant, and
//     the e2 of a step that changes cell is evaluated without relinking first.
//   * system.energy[1] is a running FP64 sum over steps in step order (`+= de`, and `-= de` on rejection:
//     src/moves.jl:17,79).  Steps commit out of order, so every step logs (de, accepted) into a ring and one wavefront
//     folds the contiguous finished prefix in step order (sq_walk).
//   * A rejected step is perform + revert: x + d + (-d) is stored (it can differ from x in the last bit), and a trial
//     that left its cell relinks twice, which moves i to the front of its chain (src/neighbours.jl:276-302 twice).  Both
//     are state changes like an accepted move.
//
// Cell membership lives in `cl_items` rows ([cell][k], the CellList vectors of src/neighbours.jl:61): for a LinkedList a
// cell's chain is its row read backwards (push-front = append), so both list types share remove + append, and the
// evaluation reads rows instead of chasing head/list pointers.  head / list / cs / ccount are kept current as well.
//
// No two steps of a window share a particle (k_sq_fill stops admitting at the first repeat), so the particle(s), the
// position before the step and the cells of every pending step are exact whatever the other pending steps do.
#pragma once
#include "pmc_kernels.h"

#define SQ_KMAX 4096                 // window slots (bit-matrix rows of SQ_KMAX / 64 words)
#define SQ_KW (SQ_KMAX / 64)
#define SQ_RING 65536                // per-step log ring (de, flags), folded in step order by sq_walk
#define SQ_GEN 262144                // ring of pre-generated random parts
#define SQ_FILL_THREADS 1024

enum {
    SQF_ACCEPT = 1,      // the evaluation accepts the step
    SQF_CROSSED = 2,     // the trial position is in another cell
    SQF_CHANGED = 4,     // committing the step changes positions, species or list order
    SQF_NULL = 8,        // DiscreteSwap with an empty species list: attempted, nothing else (oracle: `continue`)
    SQF_SWAP = 16,
    SQF_COMMITTED = 32,
    SQF_EXOTIC = 64,     // a reverted position that lands in a third cell: runs alone, as the first step of a round
    SQF_FINITE = 128     // e1 and e2 finite: system.energy is updated (src/moves.jl:13-18)
};

enum { SQL_DONE = 1, SQL_ACCEPT = 2, SQL_FINITE = 4, SQL_NULL = 8 };   // log flags; move index in bits 4..6

struct SqGen {           // random part of one step: a pure function of (key, step)
    double lnu;
    double d[3];
    uint32_t r1, q0;     // particle word(s): Displacement i = mulhi(r1, N); swap picks mulhi(r1, nA), mulhi(q0, nB)
    int m;
    int pad;
};

struct __attribute__((aligned(16))) SqSlot {
    int pk[4];           // the two cells as 16-bit coordinates (cx0|cy0<<16, cz0|cx1<<16, cy1|cz1<<16), pk[3]: 1 = has cells
    unsigned long long g;
    int i, j, m, flags;
    int c0, c1, c2, spi; // cell before, trial cell (swap: cell of j), cell after a revert; species of i
    int spj, pad0, pad1, pad2;
    double x[3], d[3], xn[3], xr[3];
    double lnu, de;
};

struct SqCtl {
    unsigned long long next;      // first step not yet admitted
    unsigned long long gend;      // one past the last step of the call
    unsigned long long prefix;    // steps < prefix are folded into energy / att / acc
    double energy;                // system.energy[1]
    unsigned long long att[8], acc[8];
    int nslots;
    int trunc;                    // slots >= trunc are held back this round (an exotic step waits to be first)
    int exclusive;                // slot 0 is exotic: only it commits
    int err;                      // 1: a cell row overflowed, 2: a particle was not in its row
    unsigned long long rounds, evals, commits;   // diagnostics
};

struct SqArgs {
    RefArgs a;
    double invbox[3];
    SqSlot *slots[2];
    const SqGen *gen;
    unsigned long long *M;        // [K][SQ_KW] conflict bits with earlier slots
    unsigned long long *cp;       // [SQ_KW]: slot changes state or has an earlier potential conflict
    int *owner;                   // [N]: INT_MAX free, -1 owned by a pending slot, else the claiming candidate
    double *log_de;               // [SQ_RING]
    int *log_flag;                // [SQ_RING]
    SqCtl *ctl;
    MovePool pool;
    int K;                        // window slots in use (<= SQ_KMAX, multiple of 64)
    int cur;                      // slots[cur] is the window this kernel works on
    int reverse;                  // 1: LinkedList (chain = row read backwards), 0: CellList (vector order)
    int ids_cap;                  // ints of dynamic LDS of the evaluation (>= 3^d * cl_cap)
    unsigned long long gen_hi;    // steps < gen_hi have their random part in the ring
};

// ---- random parts ------------------------------------------------------------------------------------------------------
// exactly the words and operations of k_ref_ops / orc_mc_steps_sequential for step g
PMC_GLOBAL void k_sq_gen(SqGen *__restrict__ gen, unsigned long long g0, int n, uint32_t key0, uint32_t key1, MovePool pool)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const unsigned long long g = g0 + (unsigned long long)t;
    uint32_t R[4], Q[4];
    philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ0, key0, key1, R);
    philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ1, key0, key1, Q);
    SqGen o;
    o.m = select_move(pool, R[0]);
    o.lnu = det_log(uniform53(R[2], R[3]));
    o.r1 = R[1];
    o.q0 = Q[0];
    o.pad = 0;
    o.d[0] = o.d[1] = o.d[2] = 0.0;
    if (pool.kind[o.m] == 0) {
        double z[4];
        box_muller(Q[0], Q[1], &z[0], &z[1]);
        box_muller(Q[2], Q[3], &z[2], &z[3]);
        for (int q = 0; q < 3; q++) o.d[q] = pool.sigma[o.m] * z[q];
    }
    gen[g % SQ_GEN] = o;
}

// ---- helpers -------------------------------------------------------------------------------------------------------------
template <int D>
__device__ __forceinline__ void sq_cell_coords(const RefArgs &a, int c, int out[3])
{
    out[0] = out[1] = out[2] = 0;
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {          // last dimension fastest (src/neighbours.jl:79-88)
        out[q] = c % a.nc[q];
        c /= a.nc[q];
    }
}

template <int D>
__device__ __forceinline__ void sq_pack_cells(const RefArgs &a, int c0, int c1, int pk[4])
{
    int u[3], v[3];
    sq_cell_coords<D>(a, c0, u);
    sq_cell_coords<D>(a, c1, v);
    pk[0] = u[0] | (u[1] << 16);
    pk[1] = u[2] | (v[0] << 16);
    pk[2] = v[1] | (v[2] << 16);
    pk[3] = 1;
}

// vector_1D (src/utils.jl:15-18): dx - round(dx / L) L.  The quotient comes from a product with 1 / L; it can only round
// to another integer than dx / L does within a few ulps of a half-integer, and there the division itself decides.
__device__ __forceinline__ double sq_vector_1d(double c1, double c2, double L, double invL)
{
    const double dx = c1 - c2;
    const double q = dx * invL;
    double n = rint(q);
    if (fabs(fabs(q - n) - 0.5) <= (fabs(q) + 1.0) * 0x1p-48) n = rint(dx / L);
    return dx - n * L;
}

template <int D>
__device__ __forceinline__ double sq_dist2(const SqArgs &s, const double *xi, const double *xj)
{
    double v = sq_vector_1d(xi[0], xj[0], s.a.box[0], s.invbox[0]);
    double r2 = v * v;
#pragma unroll
    for (int q = 1; q < D; q++) {
        v = sq_vector_1d(xi[q], xj[q], s.a.box[q], s.invbox[q]);
        r2 = r2 + v * v;
    }
    return r2;
}

// acc += u for the lanes whose u is not 0, in lane order (a zero term leaves an FP64 sum as it is)
__device__ __forceinline__ double sq_ordered_add(double acc, double u)
{
    unsigned long long mk = __ballot(u != 0.0);
    while (mk) {
        const int k = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(mk));
        acc += readlane_double(u, k);
        mk &= mk - 1;
    }
    return acc;
}

// The ids the iterator of the list yields for a particle in cell c, in order, into LDS (i itself is among them where the
// current rows have it: its term is 0 wherever it stands).  Returns their number.
__device__ __forceinline__ int sq_sequence(const SqArgs &s, int c, int *lds_ids)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int cnt = a.nbrcnt[c];
    const int cc = (lane < cnt) ? a.nbr[(size_t)c * a.nnmax + lane] : 0;
    const int len = (lane < cnt) ? a.ccount[cc] : 0;
    int incl = len;
#pragma unroll
    for (int sft = 1; sft < 32; sft <<= 1) {
        const int v = __shfl_up(incl, sft, PMC_WAVE);
        if (lane >= sft) incl += v;
    }
    const int off = incl - len;
    const int total = __shfl(incl, 31, PMC_WAVE);
    const int *row = a.cl_items + (size_t)cc * a.cl_cap;
    __syncthreads();                              // the previous pass has read its ids
    for (int k = 0; __any(k < len); k++)
        if (k < len) lds_ids[off + k] = row[s.reverse ? len - 1 - k : k];
    __syncthreads();
    return total;
}

// compute_energy_particle (src/atoms.jl:81-88) of particle i, species spi, at xa (and, when BOTH, at xb over the same
// neighbour sequence) for a particle whose iterator runs over the neighbourhood of cell c
template <int D, bool BOTH>
__device__ __forceinline__ void sq_pass(const SqArgs &s, int c, int i, int spi, const double *xa, const double *xb, int *lds_ids,
                                        double &ea, double &eb)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int total = sq_sequence(s, c, lds_ids);
    double acc_a = 0.0, acc_b = 0.0;
    for (int base = 0; base < total; base += PMC_WAVE) {
        const int n = min(PMC_WAVE, total - base);
        double ua = 0.0, ub = 0.0;
        if (lane < n) {
            const int j = lds_ids[base + lane];
            if (j != i) {                                                    // src/atoms.jl:68
                double xj[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < D; q++) xj[q] = a.pos[(size_t)j * D + q];
                const DevModel &m = a.models[spi * a.S + (a.S > 1 ? a.species[j] : 0)];
                const double r2a = sq_dist2<D>(s, xa, xj);
                if (!(r2a > m.rcut2)) ua = dev_potential_dyn(r2a, m);          // src/atoms.jl:72-73
                if (BOTH) {
                    const double r2b = sq_dist2<D>(s, xb, xj);
                    if (!(r2b > m.rcut2)) ub = dev_potential_dyn(r2b, m);
                }
            }
        }
        acc_a = sq_ordered_add(acc_a, ua);
        if (BOTH) acc_b = sq_ordered_add(acc_b, ub);
    }
    ea = acc_a;
    eb = acc_b;
}

// the two energies of one particle of a DiscreteSwap (src/moves.jl:159-167): before (species sp1) and after the exchange
// (species sp2, and the partner `other`, if it is a neighbour, with the species sp1 it then has)
template <int D>
__device__ __forceinline__ void sq_pass_swap(const SqArgs &s, int c, int i, int sp1, int sp2, int other, const double *xi,
                                             int *lds_ids, double &e_before, double &e_after)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int total = sq_sequence(s, c, lds_ids);
    double acc_a = 0.0, acc_b = 0.0;
    for (int base = 0; base < total; base += PMC_WAVE) {
        const int n = min(PMC_WAVE, total - base);
        double ua = 0.0, ub = 0.0;
        if (lane < n) {
            const int j = lds_ids[base + lane];
            if (j != i) {
                double xj[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < D; q++) xj[q] = a.pos[(size_t)j * D + q];
                const int spj = a.species[j];
                const double r2 = sq_dist2<D>(s, xi, xj);
                const DevModel &m1 = a.models[sp1 * a.S + spj];
                if (!(r2 > m1.rcut2)) ua = dev_potential_dyn(r2, m1);
                const DevModel &m2 = a.models[sp2 * a.S + (j == other ? sp1 : spj)];
                if (!(r2 > m2.rcut2)) ub = dev_potential_dyn(r2, m2);
            }
        }
        acc_a = sq_ordered_add(acc_a, ua);
        acc_b = sq_ordered_add(acc_b, ub);
    }
    e_before = acc_a;
    e_after = acc_b;
}

__device__ __forceinline__ bool sq_bits_differ(double p, double q)
{
    return __double_as_longlong(p) != __double_as_longlong(q);
}

// ---- one step, evaluated against the current state (nothing is written but the slot) --------------------------------------
template <int D>
__device__ __forceinline__ void sq_eval(const SqArgs &s, int w, int *lds_ids)
{
    const RefArgs &a = s.a;
    SqSlot &sl = s.slots[s.cur][w];
    const int lane = wave_lane();
    int flags = sl.flags & (SQF_SWAP | SQF_NULL);
    if (flags & SQF_NULL) return;
    double e1, e2;
    if (!(flags & SQF_SWAP)) {
        const int i = sl.i;
        double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) {
            x[q] = sl.x[q];
            xn[q] = sl.xn[q];                          // x + d, update_position! (src/moves.jl:46-48)
            xr[q] = xn[q] + (-sl.d[q]);                // ... and the reverted position (src/moves.jl:76-90)
        }
        const int c0 = sl.c0, c1 = sl.c1;
        const int c2 = ref_cell_of<D>(a, xr);
        const int spi = a.S > 1 ? a.species[i] : 0;
        if (c1 == c0) {
            sq_pass<D, true>(s, c0, i, spi, x, xn, lds_ids, e1, e2);
        } else {
            double dummy;
            sq_pass<D, false>(s, c0, i, spi, x, x, lds_ids, e1, dummy);
            sq_pass<D, false>(s, c1, i, spi, xn, xn, lds_ids, e2, dummy);
            flags |= SQF_CROSSED;
        }
        const bool ok = sl.lnu < -(e2 - e1) / a.temperature;    // src/utils.jl:8-10 (forward / backward log q cancel)
        bool rejchg = false;
#pragma unroll
        for (int q = 0; q < D; q++) rejchg |= sq_bits_differ(xr[q], x[q]);
        if (ok) flags |= SQF_ACCEPT;
        if (ok || c1 != c0 || rejchg || c2 != c0) flags |= SQF_CHANGED;
        if (!ok && c2 != c0) flags |= SQF_EXOTIC;
        if (lane == 0) {
#pragma unroll
            for (int q = 0; q < D; q++) sl.xr[q] = xr[q];
            sl.c2 = c2;
        }
    } else {
        const int i = sl.i, j = sl.j;
        double xi[3] = {0, 0, 0}, xj[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) { xi[q] = sl.x[q]; xj[q] = sl.xn[q]; }
        const int spi = a.species[i], spj = a.species[j];
        double e1i, e2i, e1j, e2j;
        sq_pass_swap<D>(s, sl.c0, i, spi, spj, j, xi, lds_ids, e1i, e2i);
        sq_pass_swap<D>(s, sl.c1, j, spj, spi, i, xj, lds_ids, e1j, e2j);
        e1 = e1i + e1j;                                 // src/moves.jl:166
        e2 = e2i + e2j;
        const bool ok = sl.lnu < -(e2 - e1) / a.temperature;
        if (ok) flags |= SQF_ACCEPT | SQF_CHANGED;      // a reverted swap restores species and SpeciesList exactly
        if (lane == 0) { sl.spi = spi; sl.spj = spj; }
    }
    double de = 0.0;
    if (!(isinf(e1) || isinf(e2))) {                    // src/moves.jl:13-18
        de = e2 - e1;
        flags |= SQF_FINITE;
    }
    if (lane == 0) {
        sl.de = de;
        sl.flags = flags;
        if (flags & SQF_CHANGED) atomicOr(&s.cp[w >> 6], 1ull << (w & 63));
        if (flags & SQF_EXOTIC) {
            if (w == 0) s.ctl->exclusive = 1;
            else atomicMin(&s.ctl->trunc, w);
        }
    }
}

// ---- which earlier pending steps could a step interact with -----------------------------------------------------------------
// Two steps interact iff a cell one of them writes is a neighbour cell (periodic, +-1 in every dimension) of a cell the other one
// reads from, i.e. iff some cell of one is within one cell of some cell of the other.
__device__ __forceinline__ bool sq_near(int u, int v, int nc)
{
    int dd = u - v;
    dd = dd < 0 ? -dd : dd;
    return min(dd, nc - dd) <= 1;
}

template <int D>
__device__ __forceinline__ void sq_pairs(const SqArgs &s, int wordw, int wordt)
{
    const int lane = wave_lane();
    const int nslots = s.ctl->nslots;
    const SqSlot *slots = s.slots[s.cur];
    const int w = wordw * 64 + lane;
    unsigned long long bits = 0;
    if (wordt <= wordw && wordt * 64 < nslots) {
        int p0 = 0, p1 = 0, p2 = 0, has = 0;
        if (w < nslots) { p0 = slots[w].pk[0]; p1 = slots[w].pk[1]; p2 = slots[w].pk[2]; has = slots[w].pk[3]; }
        const int wx0 = p0 & 0xffff, wy0 = (p0 >> 16) & 0xffff, wz0 = p1 & 0xffff;
        const int wx1 = (p1 >> 16) & 0xffff, wy1 = p2 & 0xffff, wz1 = (p2 >> 16) & 0xffff;
        const int tend = min(64, nslots - wordt * 64);
        for (int tt = 0; tt < tend; tt++) {
            const int t = wordt * 64 + tt;
            const int q0 = slots[t].pk[0], q1 = slots[t].pk[1], q2 = slots[t].pk[2], thas = slots[t].pk[3];   // wave-uniform
            if (!thas) continue;
            const int tx0 = q0 & 0xffff, ty0 = (q0 >> 16) & 0xffff, tz0 = q1 & 0xffff;
            const int tx1 = (q1 >> 16) & 0xffff, ty1 = q2 & 0xffff, tz1 = (q2 >> 16) & 0xffff;
            bool hit;
            if (D == 3) {
                hit = (sq_near(wx0, tx0, s.a.nc[0]) && sq_near(wy0, ty0, s.a.nc[1]) && sq_near(wz0, tz0, s.a.nc[2])) ||
                      (sq_near(wx0, tx1, s.a.nc[0]) && sq_near(wy0, ty1, s.a.nc[1]) && sq_near(wz0, tz1, s.a.nc[2])) ||
                      (sq_near(wx1, tx0, s.a.nc[0]) && sq_near(wy1, ty0, s.a.nc[1]) && sq_near(wz1, tz0, s.a.nc[2])) ||
                      (sq_near(wx1, tx1, s.a.nc[0]) && sq_near(wy1, ty1, s.a.nc[1]) && sq_near(wz1, tz1, s.a.nc[2]));
            } else {
                hit = (sq_near(wx0, tx0, s.a.nc[0]) && sq_near(wy0, ty0, s.a.nc[1])) ||
                      (sq_near(wx0, tx1, s.a.nc[0]) && sq_near(wy0, ty1, s.a.nc[1])) ||
                      (sq_near(wx1, tx0, s.a.nc[0]) && sq_near(wy1, ty0, s.a.nc[1])) ||
                      (sq_near(wx1, tx1, s.a.nc[0]) && sq_near(wy1, ty1, s.a.nc[1]));
            }
            if (hit && has && t < w && w < nslots) bits |= 1ull << tt;
        }
    }
    if (w < s.K) s.M[(size_t)w * SQ_KW + wordt] = bits;
    const unsigned long long any = __ballot(bits != 0);
    if (lane == 0 && any) atomicOr(&s.cp[wordw], any);       // P: the slot has an earlier potential conflict
}

// ---- system.energy[1] and the counters, folded in step order ------------------------------------------------------------------
__device__ __forceinline__ void sq_walk(const SqArgs &s)
{
    SqCtl *ctl = s.ctl;
    const int lane = wave_lane();
    unsigned long long prefix = ctl->prefix;
    const unsigned long long gend = ctl->gend;
    double E = ctl->energy;
    unsigned long long att[8], acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { att[q] = 0; acc[q] = 0; }
    for (;;) {
        const unsigned long long g = prefix + (unsigned long long)lane;
        const int idx = (int)(g % SQ_RING);
        int f = 0;
        double de = 0.0;
        if (g < gend) { f = s.log_flag[idx]; de = s.log_de[idx]; }
        const unsigned long long donemask = __ballot(f & SQL_DONE);
        const int ndone = (donemask == ~0ull) ? 64 : (int)__builtin_ctzll(~donemask);   // the contiguous finished prefix
        if (ndone == 0) break;
        if (lane < ndone) s.log_flag[idx] = 0;
        const unsigned long long in = (ndone == 64) ? ~0ull : ((1ull << ndone) - 1);
#pragma unroll
        for (int q = 0; q < 8; q++) {
            const unsigned long long mq = __ballot(((f >> 4) & 7) == q) & in;
            att[q] += __popcll(mq);
            acc[q] += __popcll(mq & __ballot(f & SQL_ACCEPT));
        }
        unsigned long long ops = __ballot((f & SQL_FINITE) && !(f & SQL_NULL)) & in;
        const unsigned long long rej = __ballot(!(f & SQL_ACCEPT));
        while (ops) {
            const int k = __builtin_amdgcn_readfirstlane((int)__builtin_ctzll(ops));
            const double dk = readlane_double(de, k);
            E += dk;                                   // src/moves.jl:17
            if ((rej >> k) & 1) E -= dk;               // src/moves.jl:79, 206
            ops &= ops - 1;
        }
        prefix += (unsigned long long)ndone;
        if (ndone < 64) break;
    }
    if (lane == 0) {
        ctl->prefix = prefix;
        ctl->energy = E;
        for (int q = 0; q < 8; q++) { ctl->att[q] += att[q]; ctl->acc[q] += acc[q]; }
    }
}

PMC_GLOBAL __launch_bounds__(64) void k_sq_walk(SqArgs s) { sq_walk(s); }

// ---- round, first launch: evaluation of every pending step | pair matrix | energy fold --------------------------------------
template <int D>
__global__ __launch_bounds__(64) void k_sq_round_a(SqArgs s)
{
    extern __shared__ int sq_lds[];
    const int b = blockIdx.x;
    const int kw = s.K / 64;
    if (b < s.K) {
        if (b < s.ctl->nslots) sq_eval<D>(s, b, sq_lds);
    } else if (b < s.K + kw * kw) {
        const int t = b - s.K;
        sq_pairs<D>(s, t / kw, t % kw);
    } else {
        sq_walk(s);
    }
}

// ---- round, second launch: decide and apply ------------------------------------------------------------------------------------
// update_neighbour_list! (src/neighbours.jl:276-302, CellList :162-170) as row operations; whole wavefront
__device__ __forceinline__ void sq_row_remove(const SqArgs &s, int c, int i)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    int *row = a.cl_items + (size_t)c * a.cl_cap;
    const int n = a.ccount[c];
    int p = -1;
    for (int base = 0; base < n; base += PMC_WAVE) {
        const int k = base + lane;
        const unsigned long long mk = __ballot(k < n && row[k] == i);
        if (mk) { p = base + (int)__builtin_ctzll(mk); break; }
    }
    if (p < 0) { if (lane == 0) atomicOr(&s.ctl->err, 2); return; }
    if (s.reverse && lane == 0) {
        // the chain is the row read backwards: row[n-1] is the head, list[row[k]] = row[k-1]
        const int succ = p > 0 ? row[p - 1] : -1;
        if (p == n - 1) a.head[c] = succ;
        else a.list[row[p + 1]] = succ;
    }
    for (int base = p; base < n - 1; base += PMC_WAVE) {
        const int k = base + lane;
        int v = 0;
        if (k < n - 1) v = row[k + 1];
        __syncthreads();
        if (k < n - 1) row[k] = v;
        __syncthreads();
    }
    if (lane == 0) a.ccount[c] = n - 1;
    __threadfence();
    __syncthreads();
}

__device__ __forceinline__ void sq_row_append(const SqArgs &s, int c, int i)
{
    const RefArgs &a = s.a;
    if (wave_lane() == 0) {
        int *row = a.cl_items + (size_t)c * a.cl_cap;
        const int n = a.ccount[c];
        if (n >= a.cl_cap) {
            atomicOr(&s.ctl->err, 1);
        } else {
            if (s.reverse) {
                a.list[i] = n > 0 ? row[n - 1] : -1;
                a.head[c] = i;
            }
            row[n] = i;
            a.ccount[c] = n + 1;
            a.cs[i] = c;
        }
    }
    __threadfence();
    __syncthreads();
}

template <int D>
__global__ __launch_bounds__(64) void k_sq_commit(SqArgs s)
{
    const RefArgs &a = s.a;
    const int w = blockIdx.x;
    const int lane = wave_lane();
    SqCtl *ctl = s.ctl;
    if (w >= ctl->nslots) return;
    SqSlot &sl = s.slots[s.cur][w];
    // held back iff an earlier pending step it could interact with changes the state or is itself uncertain
    const int nw = (w >> 6) + 1;
    unsigned long long hit = 0;
    if (lane < nw) hit = s.M[(size_t)w * SQ_KW + lane] & s.cp[lane];
    bool hold = __any(hit != 0);
    if (w >= ctl->trunc) hold = true;
    if (ctl->exclusive && w != 0) hold = true;
    if (hold) return;
    const int flags = sl.flags;
    const int i = sl.i;
    if (!(flags & SQF_NULL)) {
        if (!(flags & SQF_SWAP)) {
            const int c0 = sl.c0, c1 = sl.c1, c2 = sl.c2;
            if (flags & SQF_ACCEPT) {
                if (lane < D) a.pos[(size_t)i * D + lane] = sl.xn[lane];
                if (c1 != c0) { sq_row_remove(s, c0, i); sq_row_append(s, c1, i); }
            } else if (flags & SQF_CHANGED) {
                if (lane < D) a.pos[(size_t)i * D + lane] = sl.xr[lane];
                if (c1 != c0 && c2 == c0) {
                    // there (unlink, push-front in c1) and back (unlink from the head of c1, push-front in c0): c1 ends as it was
                    sq_row_remove(s, c0, i);
                    sq_row_append(s, c0, i);
                } else {
                    if (c1 != c0) { sq_row_remove(s, c0, i); sq_row_append(s, c1, i); }
                    if (c2 != c1) { sq_row_remove(s, c1, i); sq_row_append(s, c2, i); }
                }
            }
        } else if (flags & SQF_ACCEPT) {
            if (lane == 0) {
                const int j = sl.j;
                a.species[i] = sl.spj;                           // src/moves.jl:162-163
                a.species[j] = sl.spi;
                ref_update_species_list(a, s.pool.sp0[sl.m], s.pool.sp1[sl.m], i, j);
            }
        }
    }
    if (lane == 0) {
        const int idx = (int)(sl.g % SQ_RING);
        s.log_de[idx] = sl.de;
        int f = SQL_DONE | (sl.m << 4);
        if (flags & SQF_ACCEPT) f |= SQL_ACCEPT;
        if (flags & SQF_FINITE) f |= SQL_FINITE;
        if (flags & SQF_NULL) f |= SQL_NULL;
        s.log_flag[idx] = f;
        sl.flags = flags | SQF_COMMITTED;
        if (!(flags & SQF_NULL)) {
            s.owner[i] = INT_MAX;
            if (flags & SQF_SWAP) s.owner[sl.j] = INT_MAX;
        }
    }
}

// ---- round, third launch (first of the next round): drop the committed steps, admit new ones in step order ----------------------
// one workgroup; writes slots[cur ^ 1] from slots[cur]
__device__ __forceinline__ int sq_block_scan_excl(int v, int *lds, int *total)
{
    // exclusive scan over the workgroup (SQ_FILL_THREADS threads), wave scan + scan of the 16 wave totals
    const int lane = wave_lane(), wv = threadIdx.x >> 6;
    int incl = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) {
        const int t = __shfl_up(incl, sft, PMC_WAVE);
        if (lane >= sft) incl += t;
    }
    __syncthreads();
    if (lane == 63) lds[wv] = incl;
    __syncthreads();
    int base = 0, tot = 0;
    for (int q = 0; q < SQ_FILL_THREADS / 64; q++) {
        const int t = lds[q];
        if (q < wv) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

template <int D>
__global__ __launch_bounds__(SQ_FILL_THREADS) void k_sq_fill(SqArgs s)
{
    __shared__ int lds[SQ_FILL_THREADS / 64 + 2];
    const RefArgs &a = s.a;
    SqCtl *ctl = s.ctl;
    const SqSlot *src = s.slots[s.cur];
    SqSlot *dst = s.slots[s.cur ^ 1];
    const int tid = threadIdx.x;
    const int nold = ctl->nslots;
    constexpr int PER = SQ_KMAX / SQ_FILL_THREADS;
    // -- keep the held-back steps, in order
    int nkeep = 0;
    for (int r = 0; r < PER; r++) {
        const int w = r * SQ_FILL_THREADS + tid;
        const bool keep = (w < nold) && !(src[w].flags & SQF_COMMITTED);
        int tot;
        const int at = nkeep + sq_block_scan_excl(keep ? 1 : 0, lds, &tot);
        if (keep) dst[at] = src[w];
        nkeep += tot;
        if (r * SQ_FILL_THREADS + SQ_FILL_THREADS >= nold) break;
    }
    // -- admit new steps up to the first one that shares a particle with a pending step
    const unsigned long long next = ctl->next;
    unsigned long long lim = ctl->gend;
    if (s.gen_hi < lim) lim = s.gen_hi;
    if (ctl->prefix + SQ_RING < lim) lim = ctl->prefix + SQ_RING;
    int avail = s.K - nkeep;
    if (lim <= next) avail = 0;
    else if ((unsigned long long)avail > lim - next) avail = (int)(lim - next);
    int ci[PER], cj[PER], cm[PER], cf[PER];
    for (int r = 0; r < PER; r++) {
        const int q = r * SQ_FILL_THREADS + tid;
        ci[r] = -1; cj[r] = -1; cm[r] = 0; cf[r] = 0;
        if (q < avail) {
            const SqGen &ge = s.gen[(next + (unsigned long long)q) % SQ_GEN];
            const int m = ge.m;
            cm[r] = m;
            if (s.pool.kind[m] == 0) {
                ci[r] = (int)mulhi32(ge.r1, (uint32_t)a.N);                      // src/moves.jl:120-123
            } else {
                const int sa = s.pool.sp0[m], sb = s.pool.sp1[m];
                const int na = a.sp_off[sa + 1] - a.sp_off[sa], nb = a.sp_off[sb + 1] - a.sp_off[sb];
                cf[r] = SQF_SWAP;
                if (na > 0 && nb > 0) {
                    ci[r] = a.sp_ids[a.sp_off[sa] + mulhi32(ge.r1, (uint32_t)na)];   // src/moves.jl:238-241
                    cj[r] = a.sp_ids[a.sp_off[sb] + mulhi32(ge.q0, (uint32_t)nb)];
                } else {
                    cf[r] |= SQF_NULL;
                }
            }
            if (ci[r] >= 0) atomicMin(&s.owner[ci[r]], q);
            if (cj[r] >= 0) atomicMin(&s.owner[cj[r]], q);
        }
    }
    __threadfence();
    __syncthreads();
    int mycut = INT_MAX;
    for (int r = 0; r < PER; r++) {
        const int q = r * SQ_FILL_THREADS + tid;
        if (q < avail) {
            bool lose = false;
            if (ci[r] >= 0 && s.owner[ci[r]] != q) lose = true;
            if (cj[r] >= 0 && (s.owner[cj[r]] != q || cj[r] == ci[r])) lose = true;
            if (lose) mycut = min(mycut, q);
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) mycut = min(mycut, __shfl_xor(mycut, sft, PMC_WAVE));
    __syncthreads();
    if (wave_lane() == 0) lds[threadIdx.x >> 6] = mycut;
    __syncthreads();
    int cut = INT_MAX;
    for (int q = 0; q < SQ_FILL_THREADS / 64; q++) cut = min(cut, lds[q]);
    const int nadm = min(cut, avail);
    __syncthreads();
    for (int r = 0; r < PER; r++) {
        const int q = r * SQ_FILL_THREADS + tid;
        if (q >= avail) continue;
        const int i = ci[r], j = cj[r];
        if (q >= nadm) {                                                        // not admitted: give the claims back
            if (i >= 0 && s.owner[i] == q) s.owner[i] = INT_MAX;
            if (j >= 0 && s.owner[j] == q) s.owner[j] = INT_MAX;
            continue;
        }
        const unsigned long long g = next + (unsigned long long)q;
        const SqGen &ge = s.gen[g % SQ_GEN];
        SqSlot sl;
        sl.g = g;
        sl.i = i; sl.j = j; sl.m = cm[r]; sl.flags = cf[r];
        sl.c0 = sl.c1 = sl.c2 = 0; sl.spi = sl.spj = 0; sl.pad0 = sl.pad1 = sl.pad2 = 0;
        sl.lnu = ge.lnu; sl.de = 0.0;
        for (int k = 0; k < 3; k++) { sl.x[k] = 0; sl.d[k] = 0; sl.xn[k] = 0; sl.xr[k] = 0; }
        sl.pk[0] = sl.pk[1] = sl.pk[2] = sl.pk[3] = 0;
        if (i >= 0) {
            s.owner[i] = -1;
            if (j >= 0) s.owner[j] = -1;
#pragma unroll
            for (int k = 0; k < D; k++) sl.x[k] = a.pos[(size_t)i * D + k];
            if (j < 0) {
#pragma unroll
                for (int k = 0; k < D; k++) { sl.d[k] = ge.d[k]; sl.xn[k] = sl.x[k] + ge.d[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < D; k++) sl.xn[k] = a.pos[(size_t)j * D + k];     // swap: the partner's position
            }
            sl.c0 = ref_cell_of<D>(a, sl.x);
            sl.c1 = ref_cell_of<D>(a, sl.xn);
            sl.c2 = sl.c0;
            sq_pack_cells<D>(a, sl.c0, sl.c1, sl.pk);
        }
        dst[nkeep + q] = sl;
    }
    if (tid < SQ_KW) s.cp[tid] = 0;
    if (tid == 0) {
        ctl->nslots = nkeep + nadm;
        ctl->next = next + (unsigned long long)nadm;
        ctl->trunc = INT_MAX;
        ctl->exclusive = 0;
        ctl->rounds += 1;
        ctl->evals += (unsigned long long)(nkeep + nadm);
        ctl->commits += (unsigned long long)(nold - nkeep);
    }
}

// rows of a LinkedList from the CSR the build leaves (ids ascending inside a cell = the chain read backwards): k_ref_cellvec
