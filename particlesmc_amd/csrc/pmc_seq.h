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
//     apart commute EXACTLY: same e1, e2, same accept decision, same list order, whatever the order they run in -- and
//     so do two Displacements that stay in their cells and whose particles cannot come within a cutoff of each other
//     (round 5, sq_sequence: the term of a particle beyond the cutoff is 0 before and after, the sums skip zeros).
//
// A WINDOW of K pending steps is therefore evaluated at once, SQ_EW wavefronts per step, against the current state
// (sq_eval), a K x K bit matrix records which earlier pending steps a step could interact with (found through a
// table of the pending steps by cell), and a step COMMITS in this round iff no earlier pending step that could interact
// with it changes the state or is itself held back.  Steps that are held back stay in the window and are evaluated again
// next round against the state that then exists; the window is refilled in step order (sq_fill_block).  The oldest pending
// step never has a predecessor, so every round makes progress, and the result is the sequential chain's, bit for bit.
// A round is ONE launch for windows of <= SQ_FUSED_KMAX steps (k_sq_round_fused: the workgroup that finishes its evaluation
// last commits, admits and prepares), four launches beyond (k_sq_fill, k_sq_prep, k_sq_round_a, k_sq_commit).
//
//   * e1 / e2 are FP64 sums in LinkedIterator order (src/neighbours.jl:332-377).  Terms that are exactly 0 (the self term,
//     pairs beyond their cutoff: src/atoms.jl:68,72) do not change a running FP64 sum, so only the non-zero terms are
//     added, in iterator order.  For the same reason the place of particle i itself in the iteration is irrelevant, and
//     the e2 of a step that changes cell is evaluated without relinking first.
//   * system.energy[1] is a running FP64 sum over steps in step order (`+= de`, and `-= de` on rejection:
//     src/moves.jl:17,79) -- host code in the reference too (Arianna.perform_action!, src/moves.jl:11-20).  Steps commit
//     out of order, so every step logs (de, accepted) into a ring by step counter and the HOST folds the contiguous
//     finished prefix in step order between batches of rounds (seq_fold): a serial chain of dependent FP64 additions
//     is the one thing on this path a CPU core does faster than a wavefront.
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
//
// Every kernel of a round reads its arguments from a device table, one entry per chain (chain = blockIdx.y): the chains of a
// `parallel = true` run (src/ParticlesMC.jl:164,246) share every launch of a round (pmc_sweep_many), each with its own window.
#pragma once
#include "pmc_kernels.h"

#define SQ_KMAX 4096                 // window slots (bit-matrix rows of SQ_KMAX / 64 words)
#define SQ_KW (SQ_KMAX / 64)
#define SQ_RING_MAX (1 << 20)        // per-step log ring (de, flags): the host folds it into system.energy in step order
#define SQ_GEN_MAX 262144            // ring of pre-generated random parts: 128 windows' worth, at most this many steps
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

struct SqLog {           // one committed step, as Arianna.perform_action! / revert_action! see it (src/moves.jl:11-20,79,206)
    double de;
    uint32_t tag;        // low word of the step counter: tells a record of this lap of the ring from the last lap's
    int flag;            // SQL_*
};

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
    unsigned long long prefix;    // steps < prefix are folded into system.energy by the host: their ring records are free
    int nslots;
    int trunc;                    // slots >= trunc are held back this round (an exotic step waits to be first)
    int exclusive;                // slot 0 is exotic: only it commits
    int err;                      // 1: a cell row overflowed, 2: a particle was not in its row
    unsigned long long rounds, evals, commits;   // diagnostics
    int f_nold, f_nkeep;          // k_sq_fill -> k_sq_prep: slots of the old window, held-back steps among them
    int cur;                      // slots[cur] is the current window (k_sq_fill builds the next one in slots[cur ^ 1] and flips)
    int pad_cur;
    unsigned long long gen_hi;    // steps < gen_hi have their random part in the ring (the host moves it between batches)
    unsigned long long f_next0;   // ... and the step of the first candidate
    unsigned int arrive;          // k_sq_round_fused: workgroups of the launch that have finished their evaluation (0 between launches)
    int pad_arrive;
#ifdef PMC_DEBUG_TIME
    unsigned long long t_first, t_end, tacc[8];     // stamped builds: phase clock of the fused rounds (s_memtime ticks)
#endif
};

struct SqArgs {
    RefArgs a;
    double invbox[3];
    double invcell[3];            // 1 / a.cell (sq_cell_of_fast)
    double conf_r2;               // two steps that stay in their cells interact only within this squared distance (sq_sequence); 0: the cell rule
    SqSlot *slots[2];
    const SqGen *gen;
    unsigned long long *M;        // [K][SQ_KW] conflict bits with earlier slots
    int *cellhead;                // [cells] pending steps by cell: head of a chain of entries 2 slot + which (-1: none); clean between rounds
    int *cellnext;                // [2 K] the chains
    unsigned long long *cp;       // [SQ_KW]: slot changes state or has an earlier potential conflict
    int *moveto;                  // [K] k_sq_fill -> k_sq_prep: new index of an old slot, -1 if it committed
    int *cand;                    // [K][3] ... and the admitted steps: i, j, flags
    SqLog *log;                   // [ring_mask + 1] what every committed step did to system.energy, by step
    unsigned long long ring_mask;
    unsigned long long gen_mask;  // the ring of random parts holds gen_mask + 1 steps
    SqCtl *ctl;
    MovePool pool;
    int K;                        // window slots in use (<= SQ_KMAX, multiple of 64)
    int reverse;                  // 1: LinkedList (chain = row read backwards), 0: CellList (vector order)
    int ids_cap;                  // ints of dynamic LDS of the evaluation (>= 3^d * cl_cap)
    int arith_nbr;                // every dimension has >= 3 cells: the neighbour cells of a cell follow from its coordinates
    int pk_ok;                    // ... and fewer than 65536: the slots carry their cells' coordinates (SqSlot::pk)
    int dbg;                      // PMC_SEQ_DBG (timing experiments only): 1 no energy fold inside the rounds, 2 no pair matrix, 4 no evaluation
};

// ---- random parts ------------------------------------------------------------------------------------------------------
// exactly the words and operations of k_ref_ops / orc_mc_steps_sequential for step g
PMC_GLOBAL void k_sq_gen(SqGen *__restrict__ gen, unsigned long long gen_mask, unsigned long long g0, int n, uint32_t key0, uint32_t key1, MovePool pool)
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
    gen[g & gen_mask] = o;
}

#ifdef PMC_DEBUG_TIME
__device__ unsigned long long g_sqt[16];      // stamped builds: phase clock of the evaluation of slot 0 (sums of s_memtime differences)
#define SQE(k) do { if (w == 0 && threadIdx.x < 64) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                    if (wave_lane() == 0) { g_sqt[k] += now_ - g_sqt[15]; g_sqt[15] = now_; } } } while (0)
#else
#define SQE(k)
#endif

// ---- helpers -------------------------------------------------------------------------------------------------------------
// what the workgroups of ONE launch hand to each other (k_sq_round_fused: the evaluation's results -> the workgroup that arrives
// last): agent-scope relaxed accesses, as for the hand-off of the fused sweep (pmc_kernels.h, ld_agent / st_agent)
__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// vector_1D (src/utils.jl:15-18): dx - round(dx / L) L.  The quotient comes from a product with 1 / L; it can only round
// to another integer than dx / L does within a few ulps of a half-integer, and there the division itself decides.
__device__ __forceinline__ double sq_vector_1d(double c1, double c2, double L, double invL)
{
    const double dx = c1 - c2;
    const double q = dx * invL;
    double n = rint(q);
    // q is within |q| 2^-51 of dx / L: below 2^20 boxes that is < 2^-31, so the two can only round apart when q is within
    // 1e-4 of a half-integer (a generous band: the division costs nothing when it is not taken)
    if (fabs(q - n) > 0.4999 || fabs(q) > 1048576.0) n = rint(dx / L);
    return dx - n * L;
}

// get_cell + cell_index (src/neighbours.jl:178-180, 79-88) of a position: Julia's float fld is floor(x / cell) computed through
// mod (ref_cell_of: an fmod and two roundings).  Away from a cell edge the plain quotient's floor is the same integer; within
// rounding distance of an edge the reference's own sequence of operations decides.
template <int D>
__device__ __forceinline__ int sq_cell_of(const RefArgs &a, const double *x)
{
    long long c = 0, stride = 1;
    bool edge = false;
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        const double qv = x[q] / a.cell[q];
        const double f = floor(qv);
        const double r = qv - f;
        const double tol = fabs(qv) * 0x1p-48 + 0x1p-60;
        edge = edge || r < tol || 1.0 - r < tol || !(fabs(qv) < 4.0e15);
        const long long mc = (long long)f;
        c += (mc - dev_fld_i(mc, a.nc[q]) * a.nc[q]) * stride;
        stride *= a.nc[q];
    }
    return edge ? ref_cell_of<D>(a, x) : (int)c;
}

// ... with the quotient from a product with 1 / cell and 32-bit cell arithmetic: the product is within two ulps of the division,
// so outside a (wider) band around the cell edges its floor is the same integer; inside the band, and for positions many
// boxes away, sq_cell_of decides.  (The three divisions and three 64-bit floor divisions of sq_cell_of were ~2 500 cycles of
// every evaluation and ~5 000 of every admitted step's preparation.)
template <int D>
__device__ __forceinline__ int sq_cell_of_fast(const RefArgs &a, const double *invcell, const double *x, int *coords = nullptr)
{
    int c = 0, stride = 1;
    bool slow = false;
    int fcoord[3] = {0, 0, 0};
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        const double qv = x[q] * invcell[q];
        const double f = floor(qv);
        const double r = qv - f;
        const double tol = fabs(qv) * 0x1p-44 + 0x1p-60;
        slow = slow || r < tol || 1.0 - r < tol || !(fabs(qv) < 1.0e9);
        const int n = a.nc[q];
        int mc = (int)f;
        if (mc < 0) mc += n;                       // the image next to the box; anything farther takes the general path
        else if (mc >= n) mc -= n;
        slow = slow || mc < 0 || mc >= n;
        fcoord[q] = mc;
        c += mc * stride;
        stride *= n;
    }
    if (slow) {
        c = sq_cell_of<D>(a, x);
        int r = c;
#pragma unroll
        for (int q = D - 1; q >= 0; q--) { fcoord[q] = r % a.nc[q]; r /= a.nc[q]; }
    }
    if (coords) { coords[0] = fcoord[0]; coords[1] = fcoord[1]; coords[2] = fcoord[2]; }
    return c;
}

template <int D>
__device__ __forceinline__ int sq_cell_of_fast(const SqArgs &s, const double *x, int *coords = nullptr)
{
    return sq_cell_of_fast<D>(s.a, s.invcell, x, coords);
}

// inclusive prefix sum over lanes 0 .. 31 (DPP shifts inside the rows of 16, the first row's total added to the second)
__device__ __forceinline__ int sq_scan32(int v)
{
    int x = v;
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);    // row_shr:1
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);    // row_shr:2
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);    // row_shr:4
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);    // row_shr:8
    const int first = __builtin_amdgcn_readlane(x, 15);
    const int lane = wave_lane();
    return x + ((lane >= 16 && lane < 32) ? first : 0);
}

// a barrier among the lanes of ONE wavefront: what they stored has landed before any of them goes on
__device__ __forceinline__ void sq_wave_sync()
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
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

__device__ __forceinline__ int __reduce_max_sync_compat(int v)
{
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) v = max(v, __shfl_xor(v, sft, PMC_WAVE));
    return v;
}

// (Ordered FP64 sums through LDS -- sq_terms_push / sq_terms_sum / sq_terms_sum2 -- are in pmc_ref.h: the literal loop uses them too.)

// the evaluation's dynamic LDS: ids_cap ints of neighbour ids, then the two term buffers
__device__ __forceinline__ double *sq_terms_buffer(const SqArgs &s, int *lds_ids)
{
    return reinterpret_cast<double *>(lds_ids + ((s.ids_cap + 3) & ~3));
}

// The ids the iterator of the list yields for a particle in cell c, in order, into LDS (i itself is among them where the
// current rows have it: its term is 0 wherever it stands).  Returns their number.  Everything that only needs the neighbour
// cells is requested at once -- the populations, the first 64 ids of every row (speculatively: which of them exist is known when
// the populations arrive) and, for step w >= 0, the heads of the cells' chains of pending steps -- one round trip to memory
// instead of three.  Earlier pending steps found in those chains are recorded in row w of the conflict matrix (sq_conflicts).
// NW wavefronts (k_sq_round_a / k_sq_round_fused with SQ_EW): each of them works out the cells and offsets, the rows are dealt
// round-robin, the last wavefront walks the chains of pending steps.
template <int D, int NW = 1>
__device__ __forceinline__ int sq_sequence(const SqArgs &s, int c, int *lds_ids, int w = -1, const double *pa = nullptr, const double *pb = nullptr,
                                           const int *fc = nullptr, int room = 0x7fffffff)
{
    constexpr int NN = (D == 3) ? 27 : 9;
    constexpr int PER = (NN + NW - 1) / NW;
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int wv = (NW == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const bool walker = (NW == 1) || wv == NW - 1;
    // the neighbour cells in the reference's order (src/neighbours.jl:94-111: offsets -1, 0, +1, first dimension fastest).  With
    // three or more cells per side no two of them coincide (`unique!` removes nothing) and they follow from the cell's own
    // coordinates: no table, one memory round trip less; smaller grids read the table (entry and count in one round trip).
    int cnt, cc;
    if (s.arith_nbr) {
        cnt = NN;
        int f[3] = {0, 0, 0}, r = c;
        if (fc) {                                  // (the preparation left the cell's coordinates in the slot: no divisions here)
            f[0] = fc[0]; f[1] = fc[1]; f[2] = fc[2];
        } else {
#pragma unroll
            for (int q = D - 1; q >= 0; q--) { f[q] = r % a.nc[q]; r /= a.nc[q]; }   // last dimension fastest (:79-88)
        }
        int o = lane < NN ? lane : 0, lin = 0;
#pragma unroll
        for (int q = 0; q < D; q++) {
            int g = f[q] + (o % 3) - 1;
            o /= 3;
            g = g < 0 ? g + a.nc[q] : (g >= a.nc[q] ? g - a.nc[q] : g);
            lin = lin * a.nc[q] + g;
        }
        cc = lane < NN ? lin : 0;
    } else {
        const int raw = (lane < NN) ? a.nbr[(size_t)c * a.nnmax + lane] : 0;
        cnt = a.nbrcnt[c];
        cc = (lane < cnt) ? raw : 0;
    }
    SQE(1);                                       // (entry of sq_sequence: the slot has arrived)
    const int len = (lane < cnt) ? a.ccount[cc] : 0;
    int e = (w >= 0 && walker && lane < cnt) ? s.cellhead[cc] : -1;
    const int cl_cap = a.cl_cap, reverse = s.reverse;
    const int lane_in_row = lane < cl_cap ? lane : 0;
    int v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int l = k * NW + wv;
        const int cc_l = __builtin_amdgcn_readlane(cc, l < NN ? l : 0);
        v[k] = a.cl_items[(size_t)cc_l * cl_cap + lane_in_row];
    }
    SQE(2);                                       // populations, chain heads, rows
    const int incl = sq_scan32(len);
    const int off = incl - len;
    const int total = __builtin_amdgcn_readlane(incl, 31);
    const bool longrows = __any(len > PMC_WAVE);
    const bool fits = total <= room;              // (a second sequence behind the first one -- a crossing step's -- may not: nothing is written then)
    __syncthreads();                              // the previous pass has read its ids
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int l = k * NW + wv;
        if (l < NN && fits) {
            const int len_l = __builtin_amdgcn_readlane(len, l);
            const int off_l = __builtin_amdgcn_readlane(off, l);
            if (lane < len_l) lds_ids[off_l + (reverse ? len_l - 1 - lane : lane)] = v[k];
        }
    }
    if (longrows && fits) {                       // rows longer than a wavefront (dense cells, long cutoffs)
        for (int l = wv; l < cnt; l += NW) {
            const int len_l = __shfl(len, l, PMC_WAVE), cc_l = __shfl(cc, l, PMC_WAVE), off_l = __shfl(off, l, PMC_WAVE);
            for (int k = PMC_WAVE + lane; k < len_l; k += PMC_WAVE)
                lds_ids[off_l + (reverse ? len_l - 1 - k : k)] = a.cl_items[(size_t)cc_l * cl_cap + k];
        }
    }
    SQE(3);                                       // ids in LDS
    __syncthreads();
    // (the walk runs on the last wavefront while the others start on their trips; it has the fewest trips)
    if (w >= 0 && walker) {
        // two steps interact iff some cell of one is within one cell of some cell of the other: every pending step is entered
        // under its cell(s) (k_sq_prep); bit t of row w for every EARLIER step t in the chains of this cell's neighbour cells
        //
        // Two Displacements that both stay in their cells (pa, pb: this step's position and trial position) interact only if
        // the moved particle of one can be within a cutoff of the other's particle: whatever the earlier step t turns out to
        // do, its particle is at x_t, at xn_t or (a reverted move, src/moves.jl:76-90) within an ulp of x_t, it stays in
        // its row, and a particle that is beyond the largest cutoff of both pa and pb in every one of those places
        // contributes the term 0 to this step's sums before and after -- the sums skip zeros, and the order of the other terms
        // does not change.  So the bit is set only if one of the four distances is within the cutoff (conf_r2: the largest
        // rcut^2 of the system, widened by 2^-20); steps that cross cells (their row operations) and swaps keep the cell rule.
        // At N = 4096 a pair of pending steps conflicts with probability 3 % instead of 12.5 % (27 of 216 cells).
        unsigned long long *row = s.M + (size_t)w * SQ_KW;   // (zeroed by k_sq_prep)
        const SqSlot *slots = s.slots[s.ctl->cur];
        const double r2c = s.conf_r2;
        bool any = false;
        while (__any(e >= 0)) {
            if (e >= 0) {
                const int t = e >> 1;
                const int nxt = s.cellnext[e];
                if (t < w) {
                    bool hit = true;
                    if (pa != nullptr) {
                        const SqSlot &st = slots[t];
                        if (!(st.flags & SQF_SWAP) && st.c0 == st.c1 && st.pk[3]) {
                            double xt[3] = {0, 0, 0}, xnt[3] = {0, 0, 0};
#pragma unroll
                            for (int q = 0; q < D; q++) { xt[q] = st.x[q]; xnt[q] = st.xn[q]; }
                            hit = !(sq_dist2<D>(s, pa, xt) > r2c) || !(sq_dist2<D>(s, pa, xnt) > r2c) ||
                                  !(sq_dist2<D>(s, pb, xt) > r2c) || !(sq_dist2<D>(s, pb, xnt) > r2c);
                        }
                    }
                    if (hit) { atomicOr(&row[t >> 6], 1ull << (t & 63)); any = true; }
                }
                e = nxt;
            }
        }
        if (__any(any) && lane == 0) {                       // P: the step has an earlier potential conflict
            atomicOr(&s.cp[w >> 6], 1ull << (w & 63));
            atomicOr(&s.cp[SQ_KW + (w >> 6)], 1ull << (w & 63));       // (P alone: a step without it has an empty row -- sq_commit_lanes)
        }
    }
    SQE(4);                                       // chains of pending steps walked
    return fits ? total : -1;
}

#define SQ_CHUNK 8          // trips of 64 candidates whose position loads are in flight together

// compute_energy_particle (src/atoms.jl:81-88) of particle i, species spi, at xa (and, when BOTH, at xb over the same
// neighbour sequence) for a particle whose iterator runs over the neighbourhood of cell c
// (KIND: the one kind of every pair model of the system, compiled in, or 4: read per pair -- as for k_sq_lds below)
template <int D, bool BOTH, int KIND = 4>
__device__ __forceinline__ void sq_pass(const SqArgs &s, int c, int i, int spi, const double *xa, const double *xb, int *lds_ids,
                                        double &ea, double &eb, int w)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const bool fine = BOTH && s.conf_r2 > 0.0;
    const int total = sq_sequence<D>(s, c, lds_ids, w, fine ? xa : nullptr, fine ? xb : nullptr);
    double *bufa = sq_terms_buffer(s, lds_ids), *bufb = bufa + SQ_TERMS_BUF;
    double acc_a = 0.0, acc_b = 0.0;
    int na = 0, nb = 0;
    for (int base = 0; base < total; base += SQ_CHUNK * PMC_WAVE) {
        int j[SQ_CHUNK], sj[SQ_CHUNK];
        double xj[SQ_CHUNK][3];
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            const int t = base + u * PMC_WAVE + lane;
            j[u] = (t < total) ? lds_ids[t] : -1;
        }
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            const int jj = j[u] < 0 ? 0 : j[u];
#pragma unroll
            for (int q = 0; q < 3; q++) xj[u][q] = (q < D) ? a.pos[(size_t)jj * D + q] : 0.0;
            sj[u] = (a.S > 1) ? a.species[jj] : 0;
        }
        SQE(5);                                   // positions of a chunk
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            if (base + u * PMC_WAVE >= total) break;
            double ua = 0.0, ub = 0.0;
            if (j[u] >= 0 && j[u] != i) {                                        // src/atoms.jl:68
                const DevModel &m = a.models[spi * a.S + sj[u]];
                const double r2a = sq_dist2<D>(s, xa, xj[u]);
                if (!(r2a > m.rcut2)) ua = (KIND == 4 ? dev_potential_dyn(r2a, m) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2a, m));          // src/atoms.jl:72-73
                if (BOTH) {
                    const double r2b = sq_dist2<D>(s, xb, xj[u]);
                    if (!(r2b > m.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2b, m) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2b, m));
                }
            }
            if (na > SQ_TERMS || nb > SQ_TERMS) {                // the next 64 terms might not fit: sum down, in order
                sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
                na = nb = 0;
            }
            sq_terms_push(bufa, na, ua);
            if (BOTH) sq_terms_push(bufb, nb, ub);
        }
        SQE(6);                                   // pair terms of a chunk
    }
    sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
    SQE(7);                                       // ordered sums
    ea = acc_a;
    eb = acc_b;
}

// the two energies of one particle of a DiscreteSwap (src/moves.jl:159-167): before (species sp1) and after the exchange
// (species sp2, and the partner `other`, if it is a neighbour, with the species sp1 it then has)
template <int D, int KIND = 4>
__device__ __forceinline__ void sq_pass_swap(const SqArgs &s, int c, int i, int sp1, int sp2, int other, const double *xi,
                                             int *lds_ids, double &e_before, double &e_after, int w)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int total = sq_sequence<D>(s, c, lds_ids, w);
    double *bufa = sq_terms_buffer(s, lds_ids), *bufb = bufa + SQ_TERMS_BUF;
    double acc_a = 0.0, acc_b = 0.0;
    int na = 0, nb = 0;
    for (int base = 0; base < total; base += SQ_CHUNK * PMC_WAVE) {
        int j[SQ_CHUNK], sj[SQ_CHUNK];
        double xj[SQ_CHUNK][3];
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            const int t = base + u * PMC_WAVE + lane;
            j[u] = (t < total) ? lds_ids[t] : -1;
        }
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            const int jj = j[u] < 0 ? 0 : j[u];
#pragma unroll
            for (int q = 0; q < 3; q++) xj[u][q] = (q < D) ? a.pos[(size_t)jj * D + q] : 0.0;
            sj[u] = a.species[jj];
        }
#pragma unroll
        for (int u = 0; u < SQ_CHUNK; u++) {
            if (base + u * PMC_WAVE >= total) break;
            double ua = 0.0, ub = 0.0;
            if (j[u] >= 0 && j[u] != i) {
                const double r2 = sq_dist2<D>(s, xi, xj[u]);
                const DevModel &m1 = a.models[sp1 * a.S + sj[u]];
                if (!(r2 > m1.rcut2)) ua = (KIND == 4 ? dev_potential_dyn(r2, m1) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2, m1));
                const DevModel &m2 = a.models[sp2 * a.S + (j[u] == other ? sp1 : sj[u])];
                if (!(r2 > m2.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2, m2) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2, m2));
            }
            if (na > SQ_TERMS || nb > SQ_TERMS) {
                sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
                na = nb = 0;
            }
            sq_terms_push(bufa, na, ua);
            sq_terms_push(bufb, nb, ub);
        }
    }
    sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
    e_before = acc_a;
    e_after = acc_b;
}

// ---- the same passes by SQ_EW wavefronts ---------------------------------------------------------------------------------------------
// One wavefront per pending step spends ~35 000 cycles on it (stamped build, N = 4096: 13 000 in the eight trips of pair terms,
// 5 000 moving the rows into LDS, 4 500 in the two ordered sums) while the chip is all but empty -- a window is 64 .. 2048
// steps, the device has 1024 SIMDs.  Here SQ_EW wavefronts share a step: the trips are dealt round-robin, every trip packs its
// non-zero terms into its own segment of LDS (followed by eight zeros, so that the sums read whole blocks), and once all
// trips are in, wavefront 0 runs the chain of additions of the first sum and wavefront 1 that of the second, in trip order,
// the next block's reads in flight while this block's additions run.  Same terms, same order, same bits.
#define SQ_EW 4
#define SQ_TRMAX 16                  // trips whose segments the LDS holds at once (1024 candidates; longer sequences: in several goes)
#define SQ_SEG (PMC_WAVE + SQ_TERMS_PAD)

// the evaluation's dynamic LDS with SQ_EW wavefronts: ids, two sets of segments, the two sums, the segments' term counts
__host__ __device__ inline size_t sq_eval_lds_bytes(int ids_cap, int nw)
{
    const size_t ids = (size_t)((ids_cap + 3) & ~3) * sizeof(int);
    return nw == 1 ? ids + 2 * (size_t)SQ_TERMS_BUF * sizeof(double)
                   : ids + (2 * (size_t)SQ_TRMAX * SQ_SEG + 2) * sizeof(double) + 2 * SQ_TRMAX * sizeof(int);
}

__device__ __forceinline__ int sq_terms_push_seg(double *seg, double u)
{
    const unsigned long long mk = __ballot(u != 0.0);
    const int n = __popcll(mk), lane = wave_lane();
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
    if (u != 0.0) seg[rank] = u;
    if (lane < SQ_TERMS_PAD) seg[n + lane] = 0.0;               // (+0.0 leaves a sum as it is: the chain reads blocks of 8)
    return n;
}

// acc += every term of the first nt segments, in order; one wavefront, every lane the same chain.  A segment's first block of
// eight is always added (its terms, then zeros -- an empty segment is eight zeros), the next segment's first block being on its
// way from LDS meanwhile; the rare segment with more than eight terms takes its further blocks one by one.
__device__ __forceinline__ double sq_terms_chain(const double *buf, const int *tcount, int nt, double acc)
{
    const int lane = wave_lane();
    const int cntv = (lane < nt) ? tcount[lane] : 0;            // lane l: terms of segment l
    double cur[SQ_TERMS_PAD];
#pragma unroll
    for (int q = 0; q < SQ_TERMS_PAD; q++) cur[q] = buf[q];
    for (int sg = 0; sg < nt; sg++) {
        const int n = __builtin_amdgcn_readlane(cntv, sg);
        const double *p = buf + sg * SQ_SEG;
        double nxt[SQ_TERMS_PAD];
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) nxt[q] = p[SQ_SEG + q];       // (behind the last segment: other LDS of the evaluation, not used)
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) acc += cur[q];
        for (int k = SQ_TERMS_PAD; k < n; k += SQ_TERMS_PAD) {
            double t[SQ_TERMS_PAD];
#pragma unroll
            for (int q = 0; q < SQ_TERMS_PAD; q++) t[q] = p[k + q];
#pragma unroll
            for (int q = 0; q < SQ_TERMS_PAD; q++) acc += t[q];
        }
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) cur[q] = nxt[q];
    }
    return acc;
}

// the trips of one go of one sequence that fall to this wavefront: positions gathered, pair terms packed into the trips' segments
// TM 0: the term at xa (-> bufa); 1: at xa (-> bufa) and at xb (-> bufb); 2: a DiscreteSwap's particle at xa with species spa
// (-> bufa) and with spb, the partner `other` then having spa (-> bufb) -- sq_pass / sq_pass_swap above, term for term
template <int D, int TM, int KIND, int NW>
__device__ __forceinline__ void sq_mw_trips(const SqArgs &s, const int *ids, int total, int tr0, int ntrip, int wv, int i, int spa, int spb, int other,
                                            const double *xa, const double *xb, const double *L, const double *iL,
                                            double *bufa, double *bufb, int *tca, int *tcb, int w)
{
    constexpr int PERW = SQ_TRMAX / NW;
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int S = a.S;
    const double *__restrict__ pos = a.pos;
    const int *__restrict__ species = a.species;
    const DevModel *rowa = a.models + (size_t)spa * S, *rowb = a.models + (size_t)spb * S;
    int j[PERW], sj[PERW];
    double xj[PERW][3];
#pragma unroll
    for (int u = 0; u < PERW; u++) {
        const int t = (tr0 + wv + u * NW) * PMC_WAVE + lane;
        j[u] = (t < total) ? ids[t] : -1;
    }
#pragma unroll
    for (int u = 0; u < PERW; u++) {
        const int jj = j[u] < 0 ? 0 : j[u];
#pragma unroll
        for (int q = 0; q < 3; q++) xj[u][q] = (q < D) ? pos[(size_t)jj * D + q] : 0.0;
        sj[u] = (S > 1) ? species[jj] : 0;
    }
    SQE(5);                                   // positions of this wavefront's trips
#pragma unroll
    for (int u = 0; u < PERW; u++) {
        const int tr = tr0 + wv + u * NW;
        if (tr >= ntrip) break;
        double ua = 0.0, ub = 0.0;
        if (j[u] >= 0 && j[u] != i) {                                        // src/atoms.jl:68
            double r2a = 0.0, r2b = 0.0;
#pragma unroll
            for (int q = 0; q < D; q++) {
                const double va = sq_vector_1d(xa[q], xj[u][q], L[q], iL[q]);
                r2a = (q == 0) ? va * va : r2a + va * va;
                if (TM == 1) {
                    const double vb = sq_vector_1d(xb[q], xj[u][q], L[q], iL[q]);
                    r2b = (q == 0) ? vb * vb : r2b + vb * vb;
                }
            }
            const DevModel &ma = rowa[sj[u]];
            if (!(r2a > ma.rcut2)) ua = (KIND == 4 ? dev_potential_dyn(r2a, ma) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2a, ma));          // src/atoms.jl:72-73
            if (TM == 1) {
                if (!(r2b > ma.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2b, ma) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2b, ma));
            } else if (TM == 2) {
                const DevModel &mb = rowb[j[u] == other ? spa : sj[u]];
                if (!(r2a > mb.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2a, mb) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2a, mb));
            }
        }
        const int sg = tr - tr0;
        const int ca = sq_terms_push_seg(bufa + sg * SQ_SEG, ua);
        if (lane == 0) tca[sg] = ca;
        if (TM != 0) {
            const int cb = sq_terms_push_seg(bufb + sg * SQ_SEG, ub);
            if (lane == 0) tcb[sg] = cb;
        }
    }
    SQE(6);                                   // pair terms of this wavefront's trips
}

// MODE 0: compute_energy_particle at xa over the sequence of cell c; 1: at xa and at xb over that sequence; 2: a DiscreteSwap's
// particle (see sq_mw_trips); 3: a Displacement that crosses cells -- at xa over the sequence of c and at xb over the sequence of
// cb, the two sequences side by side in LDS, their trips in the same goes, their sums on two wavefronts at once (one after the
// other they made a crossing step 1.7 times as long as any other, and nearly every round has one)
template <int D, int MODE, int KIND, int NW>
__device__ __forceinline__ void sq_pass_mw(const SqArgs &s, int c, int cb, int i, int spa, int spb, int other, const double *xa, const double *xb,
                                           int *lds_ids, double &ea, double &eb, int w, const int *fc = nullptr, const int *fcb = nullptr)
{
    static_assert(NW > 1 && SQ_TRMAX % NW == 0, "the trips of one go are dealt evenly");
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int idsz = (s.ids_cap + 3) & ~3;
    const bool fine = MODE == 1 && s.conf_r2 > 0.0;
    const int total = sq_sequence<D, NW>(s, c, lds_ids, w, fine ? xa : nullptr, fine ? xb : nullptr, fc);
    // (a crossing step's second sequence goes behind the first: the region holds 3^d FULL rows, two neighbourhoods of a liquid
    // fill a fraction of it; if they ever do not fit, the two passes run one after the other)
    int *idsb = lds_ids + ((total + 3) & ~3);
    const int totalb = (MODE == 3) ? sq_sequence<D, NW>(s, cb, idsb, w, nullptr, nullptr, fcb, idsz - ((total + 3) & ~3)) : 0;
    if (MODE == 3 && totalb < 0) {
        double dummy;
        sq_pass_mw<D, 0, KIND, NW>(s, c, c, i, spa, spa, -1, xa, xa, lds_ids, ea, dummy, -1, fc, nullptr);
        sq_pass_mw<D, 0, KIND, NW>(s, cb, cb, i, spa, spa, -1, xb, xb, lds_ids, eb, dummy, -1, fcb, nullptr);
        return;
    }
    double *bufa = reinterpret_cast<double *>(lds_ids + idsz), *bufb = bufa + SQ_TRMAX * SQ_SEG, *esum = bufb + SQ_TRMAX * SQ_SEG;
    int *tcount = reinterpret_cast<int *>(esum + 2);
    double L[3] = {1, 1, 1}, iL[3] = {1, 1, 1};
#pragma unroll
    for (int q = 0; q < D; q++) { L[q] = a.box[q]; iL[q] = s.invbox[q]; }
    const int ntrip = (total + PMC_WAVE - 1) / PMC_WAVE, ntripb = (totalb + PMC_WAVE - 1) / PMC_WAVE;
    const int ngo = max(ntrip, ntripb);
    double acc = 0.0;                                   // wavefront 0: the first sum, wavefront 1: the second
    for (int tr0 = 0; tr0 < ngo; tr0 += SQ_TRMAX) {
        if (MODE == 3) {
            sq_mw_trips<D, 0, KIND, NW>(s, lds_ids, total, tr0, ntrip, wv, i, spa, spa, -1, xa, xa, L, iL, bufa, bufa, tcount, tcount, w);
            sq_mw_trips<D, 0, KIND, NW>(s, idsb, totalb, tr0, ntripb, wv, i, spa, spa, -1, xb, xb, L, iL, bufb, bufb, tcount + SQ_TRMAX, tcount + SQ_TRMAX, w);
        } else {
            sq_mw_trips<D, MODE, KIND, NW>(s, lds_ids, total, tr0, ntrip, wv, i, spa, spb, other, xa, xb, L, iL, bufa, bufb, tcount, tcount + SQ_TRMAX, w);
        }
        __syncthreads();
        SQE(10);                                  // every wavefront's trips
        if (wv == 0) acc = sq_terms_chain(bufa, tcount, max(0, min(SQ_TRMAX, ntrip - tr0)), acc);
        else if (MODE != 0 && wv == 1) acc = sq_terms_chain(bufb, tcount + SQ_TRMAX, max(0, min(SQ_TRMAX, (MODE == 3 ? ntripb : ntrip) - tr0)), acc);
        SQE(11);                                  // this wavefront's chain of additions
        if (tr0 + SQ_TRMAX < ngo) __syncthreads();                               // (the segments are free for the next go)
    }
    if (lane == 0 && wv == 0) esum[0] = acc;
    if (lane == 0 && wv == 1) esum[1] = (MODE != 0) ? acc : 0.0;
    __syncthreads();
    SQE(7);                                       // ordered sums
    ea = esum[0];
    eb = esum[1];
}

template <int D, int MODE, int KIND, int NW>
__device__ __forceinline__ void sq_energy(const SqArgs &s, int c, int i, int spa, int spb, int other, const double *xa, const double *xb,
                                          int *lds_ids, double &ea, double &eb, int w, const int *fc = nullptr)
{
    if constexpr (NW == 1) {
        if constexpr (MODE == 2) sq_pass_swap<D, KIND>(s, c, i, spa, spb, other, xa, lds_ids, ea, eb, w);
        else sq_pass<D, MODE == 1, KIND>(s, c, i, spa, xa, xb, lds_ids, ea, eb, w);
    } else {
        sq_pass_mw<D, MODE, KIND, NW>(s, c, c, i, spa, spb, other, xa, xb, lds_ids, ea, eb, w, fc, nullptr);
    }
}

__device__ __forceinline__ bool sq_bits_differ(double p, double q)
{
    return __double_as_longlong(p) != __double_as_longlong(q);
}

// what an evaluation reads of its slot
struct SqHead {
    int flags, i, j, c0, c1, pk[3];
    double x[3], xn[3], d[3], lnu;
};

template <int D>
__device__ __forceinline__ void sq_slot_head(const SqSlot *p, SqHead &h)
{
    const SqSlot *v = p;
    h.flags = v->flags; h.i = v->i; h.j = v->j; h.c0 = v->c0; h.c1 = v->c1;
    h.pk[0] = v->pk[0]; h.pk[1] = v->pk[1]; h.pk[2] = v->pk[2];
#pragma unroll
    for (int q = 0; q < 3; q++) {
        h.x[q] = (q < D) ? v->x[q] : 0.0;
        h.xn[q] = (q < D) ? v->xn[q] : 0.0;
        h.d[q] = (q < D) ? v->d[q] : 0.0;
    }
    h.lnu = v->lnu;
}

// the coordinates of the two cells of a step (SqSlot::pk, written by sq_prep_thread)
__device__ __forceinline__ void sq_unpack_cells(const int *pk, int *f0, int *f1)
{
    f0[0] = pk[0] & 0xffff; f0[1] = (pk[0] >> 16) & 0xffff; f0[2] = pk[1] & 0xffff;
    f1[0] = (pk[1] >> 16) & 0xffff; f1[1] = pk[2] & 0xffff; f1[2] = (pk[2] >> 16) & 0xffff;
}

// ---- one step, evaluated against the current state (nothing is written but the slot) --------------------------------------
template <int D, int KIND = 4, int NW = 1>
__device__ __forceinline__ void sq_eval(const SqArgs &s, int w, int *lds_ids)
{
    const RefArgs &a = s.a;
    PMC_TIME(if (w == 0 && threadIdx.x == 0) g_sqt[15] = __builtin_amdgcn_s_memtime();)
    // Everything the step needs from its slot is requested at once, ahead of the branches that would otherwise split it into two
    // round trips.  (Requesting BOTH window buffers' slots before the control block says which is current, with volatile loads
    // so that they stay ahead of the branch, was measured: system-scope loads, 10 % slower rounds.)
    SqCtl *ctl = s.ctl;
    SqSlot *const buf0 = s.slots[0], *const buf1 = s.slots[1];
    int nslots = ctl->nslots, cur = ctl->cur;
    asm volatile("" : "+v"(nslots), "+v"(cur));         // (both words of the control block in one round trip)
    cur = __builtin_amdgcn_readfirstlane(cur);
    if (w >= nslots) return;
    SqSlot &sl = (cur ? buf1 : buf0)[w];
    SqHead hd;
    sq_slot_head<D>(&sl, hd);
    asm volatile("" : "+v"(hd.flags), "+v"(hd.i), "+v"(hd.j), "+v"(hd.c0), "+v"(hd.c1), "+v"(hd.pk[0]), "+v"(hd.pk[1]), "+v"(hd.pk[2]),
                      "+v"(hd.x[0]), "+v"(hd.x[1]), "+v"(hd.x[2]), "+v"(hd.xn[0]), "+v"(hd.xn[1]), "+v"(hd.xn[2]),
                      "+v"(hd.d[0]), "+v"(hd.d[1]), "+v"(hd.d[2]), "+v"(hd.lnu));      // (the whole slot in one round trip)
    int flags = hd.flags & (SQF_SWAP | SQF_NULL);
    SQE(0);                                       // control block + slot
    if (flags & SQF_NULL) return;
    const double lnu = hd.lnu;
    int fc0s[3], fc1s[3];
    sq_unpack_cells(hd.pk, fc0s, fc1s);
    const int *fc0 = s.pk_ok ? fc0s : nullptr, *fc1 = s.pk_ok ? fc1s : nullptr;
    double e1, e2;
    if (!(flags & SQF_SWAP)) {
        const int i = hd.i;
        double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) {
            x[q] = hd.x[q];
            xn[q] = hd.xn[q];                          // x + d, update_position! (src/moves.jl:46-48)
            xr[q] = xn[q] + (-hd.d[q]);                // ... and the reverted position (src/moves.jl:76-90)
        }
        const int c0 = hd.c0, c1 = hd.c1;
        bool same = true;                               // (the reverted position differs from x by an ulp at most: its cell is c0 unless ...)
#pragma unroll
        for (int q = 0; q < D; q++) same = same && !sq_bits_differ(xr[q], x[q]);
        const int c2 = same ? c0 : sq_cell_of_fast<D>(s, xr);
        const int spi = a.S > 1 ? a.species[i] : 0;
        if (c1 == c0) {
            sq_energy<D, 1, KIND, NW>(s, c0, i, spi, spi, -1, x, xn, lds_ids, e1, e2, w, fc0);
        } else {
            if constexpr (NW == 1) {
                double dummy;
                sq_energy<D, 0, KIND, NW>(s, c0, i, spi, spi, -1, x, x, lds_ids, e1, dummy, w);
                sq_energy<D, 0, KIND, NW>(s, c1, i, spi, spi, -1, xn, xn, lds_ids, e2, dummy, w);
            } else {
                sq_pass_mw<D, 3, KIND, NW>(s, c0, c1, i, spi, spi, -1, x, xn, lds_ids, e1, e2, w, fc0, fc1);
            }
            flags |= SQF_CROSSED;
        }
        const bool ok = lnu < -(e2 - e1) / a.temperature;       // src/utils.jl:8-10 (forward / backward log q cancel)
        bool rejchg = false;
#pragma unroll
        for (int q = 0; q < D; q++) rejchg |= sq_bits_differ(xr[q], x[q]);
        if (ok) flags |= SQF_ACCEPT;
        if (ok || c1 != c0 || rejchg || c2 != c0) flags |= SQF_CHANGED;
        if (!ok && c2 != c0) flags |= SQF_EXOTIC;
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < D; q++) st_agent(&sl.xr[q], xr[q]);
            st_agent(&sl.c2, c2);
        }
    } else {
        const int i = hd.i, j = hd.j;
        double xi[3] = {0, 0, 0}, xj[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) { xi[q] = hd.x[q]; xj[q] = hd.xn[q]; }
        const int spi = a.species[i], spj = a.species[j];
        double e1i, e2i, e1j, e2j;
        sq_energy<D, 2, KIND, NW>(s, hd.c0, i, spi, spj, j, xi, xi, lds_ids, e1i, e2i, w, fc0);
        sq_energy<D, 2, KIND, NW>(s, hd.c1, j, spj, spi, i, xj, xj, lds_ids, e1j, e2j, w, fc1);
        e1 = e1i + e1j;                                 // src/moves.jl:166
        e2 = e2i + e2j;
        const bool ok = lnu < -(e2 - e1) / a.temperature;
        if (ok) flags |= SQF_ACCEPT | SQF_CHANGED;      // a reverted swap restores species and SpeciesList exactly
        if (threadIdx.x == 0) { st_agent(&sl.spi, spi); st_agent(&sl.spj, spj); }
    }
    double de = 0.0;
    if (!(isinf(e1) || isinf(e2))) {                    // src/moves.jl:13-18
        de = e2 - e1;
        flags |= SQF_FINITE;
    }
    if (threadIdx.x == 0) {                             // (agent-scope stores: in a fused round another workgroup of this launch reads them)
        st_agent(&sl.de, de);
        st_agent(&sl.flags, flags);
        if (flags & SQF_CHANGED) atomicOr(&s.cp[w >> 6], 1ull << (w & 63));
        if (flags & SQF_EXOTIC) {
            if (w == 0) st_agent(&s.ctl->exclusive, 1);
            else atomicMin(&s.ctl->trunc, w);
        }
    }
    SQE(8);                                       // decision, results stored
    PMC_TIME(if (w == 0 && threadIdx.x == 0) g_sqt[9] += 1;)
}

// ---- round, third launch: every pending step finds the earlier steps it could interact with and is evaluated --------------------------------------
// (five waves per SIMD: at 28 KB of LDS a CU then holds five steps, the chip 1280 -- the ~1100 pending steps of a window of 2048
// are evaluated in one go instead of 1024 and a second helping)
template <int D, int KIND = 4, int NW = 1>
__global__ __launch_bounds__(NW * 64, NW == 1 ? 1 : 5) void k_sq_round_a(const SqArgs *__restrict__ tab)
{
    const SqArgs &s = tab[blockIdx.y];
    extern __shared__ int sq_lds[];
    if (!(s.dbg & 4)) sq_eval<D, KIND, NW>(s, (int)blockIdx.x, sq_lds);      // (returns at once for a slot beyond the window's steps)
}

// ---- round, fourth launch: decide and apply ------------------------------------------------------------------------------------
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
        sq_wave_sync();
        if (k < n - 1) row[k] = v;
        sq_wave_sync();
    }
    if (lane == 0) a.ccount[c] = n - 1;
    sq_wave_sync();                                     // (the rows a launch edits are read by this wavefront only: no device-scope fence,
                                                        // no workgroup barrier -- the other wavefronts of a fused round commit other steps)
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
    sq_wave_sync();
}

template <int D>
__device__ __forceinline__ void sq_commit_wave(const SqArgs &s, const int w)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    SqCtl *ctl = s.ctl;
    if (w >= ctl->nslots) return;
    SqSlot &sl = s.slots[ctl->cur][w];
    if (lane == 0 && sl.pk[3]) {                        // the cell table is rebuilt by the next round's k_sq_prep
        s.cellhead[sl.c0] = -1;
        s.cellhead[sl.c1] = -1;
    }
    // held back iff an earlier pending step it could interact with changes the state or is itself uncertain: "changes the state or
    // has any earlier potential conflict" is a superset of "is held back" that needs no recursion.  (Resolving the recursion
    // exactly -- every wavefront running it over a window of <= 256 steps -- was measured in round 4: 17 % more steps per round
    // at N = 32768 for a commit kernel 9 us longer: slower.)
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
        SqLog &lg = s.log[sl.g & s.ring_mask];
        lg.de = sl.de;
        lg.tag = (uint32_t)sl.g;
        int f = SQL_DONE | (sl.m << 4);
        if (flags & SQF_ACCEPT) f |= SQL_ACCEPT;
        if (flags & SQF_FINITE) f |= SQL_FINITE;
        if (flags & SQF_NULL) f |= SQL_NULL;
        lg.flag = f;
        sl.flags = flags | SQF_COMMITTED;
    }
}

template <int D>
__global__ __launch_bounds__(64) void k_sq_commit(const SqArgs *__restrict__ tab)
{
    sq_commit_wave<D>(tab[blockIdx.y], (int)blockIdx.x);
}

// ---- round, first launch: drop the committed steps, admit new ones in step order ------------------------------------------------
// One workgroup; writes slots[cur ^ 1] from slots[cur].  No two pending steps may share a particle: the particles of the
// steps that stay and of the candidates go into a hash table in LDS (value = the smallest claimant, -1 for a pending step);
// admission stops at the first candidate that does not hold all its particles.
template <int NT>
__device__ __forceinline__ int sq_block_scan_excl(int v, int *lds, int *total)
{
    // exclusive scan over the workgroup (NT threads), wave scan + scan of the wave totals
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
    for (int q = 0; q < NT / 64; q++) {
        const int t = lds[q];
        if (q < wv) base += t;
        tot += t;
    }
    *total = tot;
    return base + incl - v;
}

__device__ __forceinline__ unsigned sq_hash(int id) { return (unsigned)id * 2654435761u; }

__device__ __forceinline__ void sq_table_claim(int *keys, int *vals, unsigned mask, int id, int claimant)
{
    unsigned hh = (sq_hash(id) >> 7) & mask;
    for (;;) {
        const int old = atomicCAS(&keys[hh], -1, id);
        if (old == -1 || old == id) { atomicMin(&vals[hh], claimant); return; }
        hh = (hh + 1) & mask;
    }
}

__device__ __forceinline__ int sq_table_owner(const int *keys, const int *vals, unsigned mask, int id)
{
    unsigned hh = (sq_hash(id) >> 7) & mask;
    while (keys[hh] != id) hh = (hh + 1) & mask;
    return vals[hh];
}

// NT threads admit for windows of up to 4 NT slots (k_sq_fill: 1024 threads)
// (AG: the COMMITTED flags were written in this launch -- k_sq_round_fused)
template <int D, int NT, bool AG = false>
__device__ __forceinline__ void sq_fill_block(const SqArgs &s, int *sq_table, int *lds)
{
    const RefArgs &a = s.a;
    SqCtl *ctl = s.ctl;
    const int cur = ctl->cur;
    const SqSlot *src = s.slots[cur];
    const int tid = threadIdx.x;
    const unsigned tmask = (unsigned)(4 * s.K - 1);          // the host keeps K a power of two
    int *keys = sq_table, *vals = sq_table + 4 * s.K;
    const int nold = ctl->nslots;
    const unsigned long long next = ctl->next;
    unsigned long long lim = ctl->gend;
    if (ctl->gen_hi < lim) lim = ctl->gen_hi;
    if (ctl->prefix + s.ring_mask + 1 < lim) lim = ctl->prefix + s.ring_mask + 1;
    const int maxcand = (lim <= next) ? 0 : (int)min((unsigned long long)s.K, lim - next);
    for (int k = tid; k < 4 * s.K; k += NT) { keys[k] = -1; vals[k] = INT_MAX; }
    constexpr int PER = 4;                       // slots per thread: the window is at most 4 NT
    // -- the particle words of every step that could be admitted: loads that do not wait for anything below
    int gm[PER];
    uint32_t gr1[PER], gq0[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int q = r * NT + tid;
        gm[r] = 0; gr1[r] = 0; gq0[r] = 0;
        if (q < maxcand) {
            const SqGen &ge = s.gen[(next + (unsigned long long)q) & s.gen_mask];
            gm[r] = ge.m; gr1[r] = ge.r1; gq0[r] = ge.q0;
        }
    }
    // -- keep the held-back steps, in order
    int nkeep = 0;
    for (int r = 0; r < PER; r++) {
        const int w = r * NT + tid;
        int fl = SQF_COMMITTED, pi = -1, pj = -1;
        if (w < nold) { fl = AG ? ld_agent(&src[w].flags) : src[w].flags; pi = src[w].i; pj = src[w].j; }
        const bool keep = !(fl & SQF_COMMITTED);
        int tot;
        const int at = nkeep + sq_block_scan_excl<NT>(keep ? 1 : 0, lds, &tot);   // (its barriers also cover the table's initialisation)
        if (w < nold) s.moveto[w] = keep ? at : -1;
        if (keep) {
            if (pi >= 0) sq_table_claim(keys, vals, tmask, pi, -1);
            if (pj >= 0) sq_table_claim(keys, vals, tmask, pj, -1);
        }
        nkeep += tot;
        if (r * NT + NT >= nold) break;
    }
    // -- candidates in step order
    const int avail = min(s.K - nkeep, maxcand);
    int ci[PER], cj[PER], cf[PER];
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int q = r * NT + tid;
        ci[r] = -1; cj[r] = -1; cf[r] = 0;
        if (q < avail) {
            const int m = gm[r];
            if (s.pool.kind[m] == 0) {
                ci[r] = (int)mulhi32(gr1[r], (uint32_t)a.N);                      // src/moves.jl:120-123
            } else {
                const int sa = s.pool.sp0[m], sb = s.pool.sp1[m];
                const int na = a.sp_off[sa + 1] - a.sp_off[sa], nb = a.sp_off[sb + 1] - a.sp_off[sb];
                cf[r] = SQF_SWAP;
                if (na > 0 && nb > 0) {
                    ci[r] = a.sp_ids[a.sp_off[sa] + mulhi32(gr1[r], (uint32_t)na)];   // src/moves.jl:238-241
                    cj[r] = a.sp_ids[a.sp_off[sb] + mulhi32(gq0[r], (uint32_t)nb)];
                } else {
                    cf[r] |= SQF_NULL;
                }
            }
        }
    }
    __syncthreads();                                   // the pending steps' claims are in the table
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int q = r * NT + tid;
        if (q < avail) {
            if (ci[r] >= 0) sq_table_claim(keys, vals, tmask, ci[r], q);
            if (cj[r] >= 0) sq_table_claim(keys, vals, tmask, cj[r], q);
        }
    }
    __syncthreads();
    int mycut = INT_MAX;
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int q = r * NT + tid;
        if (q < avail) {
            bool lose = false;
            if (ci[r] >= 0 && sq_table_owner(keys, vals, tmask, ci[r]) != q) lose = true;
            if (cj[r] >= 0 && (cj[r] == ci[r] || sq_table_owner(keys, vals, tmask, cj[r]) != q)) lose = true;
            if (lose) mycut = min(mycut, q);
        }
    }
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) mycut = min(mycut, __shfl_xor(mycut, sft, PMC_WAVE));
    __syncthreads();
    if (wave_lane() == 0) lds[threadIdx.x >> 6] = mycut;
    __syncthreads();
    int cut = INT_MAX;
    for (int q = 0; q < NT / 64; q++) cut = min(cut, lds[q]);
    const int nadm = min(cut, avail);
#pragma unroll
    for (int r = 0; r < PER; r++) {
        const int q = r * NT + tid;
        if (q < nadm) { s.cand[3 * q] = ci[r]; s.cand[3 * q + 1] = cj[r]; s.cand[3 * q + 2] = cf[r]; }
    }
    for (int k = tid; k < 2 * SQ_KW; k += NT) s.cp[k] = 0;
    if (tid == 0) {
        ctl->cur = cur ^ 1;                       // (k_sq_prep fills it from slots[cur])
        ctl->f_nold = nold;
        ctl->f_nkeep = nkeep;
        ctl->f_next0 = next;
        ctl->nslots = nkeep + nadm;
        ctl->next = next + (unsigned long long)nadm;
        ctl->trunc = INT_MAX;
        ctl->exclusive = 0;
        ctl->rounds += 1;
        ctl->evals += (unsigned long long)(nkeep + nadm);
        ctl->commits += (unsigned long long)(nold - nkeep);
    }
}

template <int D>
__global__ __launch_bounds__(SQ_FILL_THREADS) void k_sq_fill(const SqArgs *__restrict__ tab)
{
    extern __shared__ int sq_table[];            // keys[4 K], vals[4 K]
    __shared__ int lds[SQ_FILL_THREADS / 64 + 2];
    static_assert(SQ_KMAX == 4 * SQ_FILL_THREADS, "sq_fill_block admits for windows of up to four slots per thread");
    sq_fill_block<D, SQ_FILL_THREADS>(tab[blockIdx.y], sq_table, lds);
}

// ---- round, second launch: the new window -- held-back steps moved up, admitted steps built (position, trial position, cells) ------
// one thread per old slot and per admitted step; writes slots[cur] from slots[cur ^ 1] (the host has flipped cur)
template <int D>
__device__ __forceinline__ void sq_prep_thread(const SqArgs &s, const int t)
{
    const RefArgs &a = s.a;
    const SqCtl *ctl = s.ctl;
    const int nold = ctl->f_nold, nkeep = ctl->f_nkeep;
    const int nadm = ctl->nslots - nkeep;
    const int cur = ctl->cur;
    SqSlot *dst = s.slots[cur];
    if (t < ctl->nslots) {                              // row t of the conflict matrix: the words that can hold an earlier step
        unsigned long long *row = s.M + (size_t)t * SQ_KW;
        for (int k = 0; k <= (t >> 6); k++) row[k] = 0;
    }
    if (t < nold) {
        const int to = s.moveto[t];
        if (to >= 0) {
            // twelve 16-byte words (a local SqSlot went through scratch memory: two more round trips)
            static_assert(sizeof(SqSlot) == 192 && offsetof(SqSlot, flags) == 36 && offsetof(SqSlot, c0) == 40 && offsetof(SqSlot, pk) == 0, "SqSlot layout");
            const uint4 *src4 = reinterpret_cast<const uint4 *>(&s.slots[cur ^ 1][t]);
            uint4 *dst4 = reinterpret_cast<uint4 *>(&dst[to]);
            uint4 q[12];
#pragma unroll
            for (int k = 0; k < 12; k++) q[k] = src4[k];
            q[2].y &= (unsigned)(SQF_SWAP | SQF_NULL);  // flags (bytes 36..39): what the last evaluation found is void; in a fused round it may not even be visible here
#pragma unroll
            for (int k = 0; k < 12; k++) dst4[k] = q[k];
            const int pk3 = (int)q[0].w, kc0 = (int)q[2].z, kc1 = (int)q[2].w;
            if (pk3) {                                  // entered under its cell(s): sq_conflicts walks these chains
                s.cellnext[2 * to] = atomicExch(&s.cellhead[kc0], 2 * to);
                if (kc1 != kc0) s.cellnext[2 * to + 1] = atomicExch(&s.cellhead[kc1], 2 * to + 1);
            }
        }
        return;
    }
    const int q = t - nold;
    if (q >= nadm) return;
    const unsigned long long g = ctl->f_next0 + (unsigned long long)q;
    const SqGen &ge = s.gen[g & s.gen_mask];
    const int i = s.cand[3 * q], j = s.cand[3 * q + 1];
    // the record is assembled in registers and leaves as twelve 16-byte words (a local SqSlot went through scratch memory)
    const int m = ge.m, fl = s.cand[3 * q + 2];
    const double lnu = ge.lnu;
    double x[3], d[3], xn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        x[k] = (k < D && i >= 0) ? a.pos[(size_t)i * D + k] : 0.0;
        d[k] = ge.d[k];
        xn[k] = (k < D && i >= 0) ? ((j >= 0) ? a.pos[(size_t)j * D + k]         // DiscreteSwap: the partner's position
                                            : x[k] + d[k])                     // update_position! (src/moves.jl:46-48)
                                : 0.0;
    }
    int c0 = 0, c1 = 0, pk[4] = {0, 0, 0, 0};
    if (i >= 0) {                                       // (a null step has no cells and interacts with nothing)
        int f0[3], f1[3];
        c0 = sq_cell_of_fast<D>(s, x, f0);
        c1 = sq_cell_of_fast<D>(s, xn, f1);
        pk[0] = f0[0] | (f0[1] << 16);
        pk[1] = f0[2] | (f1[0] << 16);
        pk[2] = f1[1] | (f1[2] << 16);
        pk[3] = 1;
        const int to = nkeep + q;
        s.cellnext[2 * to] = atomicExch(&s.cellhead[c0], 2 * to);
        if (c1 != c0) s.cellnext[2 * to + 1] = atomicExch(&s.cellhead[c1], 2 * to + 1);
    }
    auto lo = [](double v) { return (unsigned)(__double_as_longlong(v) & 0xffffffffll); };
    auto hi = [](double v) { return (unsigned)((unsigned long long)__double_as_longlong(v) >> 32); };
    uint4 *dst4 = reinterpret_cast<uint4 *>(&dst[nkeep + q]);
    dst4[0] = make_uint4((unsigned)pk[0], (unsigned)pk[1], (unsigned)pk[2], (unsigned)pk[3]);
    dst4[1] = make_uint4((unsigned)(g & 0xffffffffull), (unsigned)(g >> 32), (unsigned)i, (unsigned)j);
    dst4[2] = make_uint4((unsigned)m, (unsigned)fl, (unsigned)c0, (unsigned)c1);
    dst4[3] = make_uint4((unsigned)c0, 0u, 0u, 0u);                                  // c2 = c0, spi
    dst4[4] = make_uint4(0u, 0u, lo(x[0]), hi(x[0]));                                // spj, pad0 | bytes 72..: pad1, pad2?  (see the layout assert)
    static_assert(offsetof(SqSlot, c2) == 48 && offsetof(SqSlot, spj) == 56 && offsetof(SqSlot, x) == 72 && offsetof(SqSlot, d) == 96 &&
                  offsetof(SqSlot, xn) == 120 && offsetof(SqSlot, xr) == 144 && offsetof(SqSlot, lnu) == 168 && offsetof(SqSlot, de) == 176, "SqSlot layout");
    dst4[5] = make_uint4(lo(x[1]), hi(x[1]), lo(x[2]), hi(x[2]));
    dst4[6] = make_uint4(lo(d[0]), hi(d[0]), lo(d[1]), hi(d[1]));
    dst4[7] = make_uint4(lo(d[2]), hi(d[2]), lo(xn[0]), hi(xn[0]));
    dst4[8] = make_uint4(lo(xn[1]), hi(xn[1]), lo(xn[2]), hi(xn[2]));
    dst4[9] = make_uint4(0u, 0u, 0u, 0u);                                            // xr[0], xr[1]
    dst4[10] = make_uint4(0u, 0u, lo(lnu), hi(lnu));                                 // xr[2], lnu
    dst4[11] = make_uint4(0u, 0u, 0u, 0u);                                           // de, padding
}

template <int D>
__global__ __launch_bounds__(256) void k_sq_prep(const SqArgs *__restrict__ tab)
{
    sq_prep_thread<D>(tab[blockIdx.y], (int)(blockIdx.x * blockDim.x + threadIdx.x));
}

// ---- small windows: a round as ONE launch --------------------------------------------------------------------------------------
// A round of the windows is four launches (fill, prep, evaluation, commit), each with a floor of ~4.5 us however little it
// does: with windows of 64 .. 256 steps -- what systems below ~10^5 particles get -- that floor IS the round (N = 4096: 7.6
// steps per 34 us).  Here a launch is the evaluation of the window the last launch prepared, by K workgroups of one wavefront,
// and then -- in the workgroup that finishes LAST (an arrival counter; no workgroup waits for another) -- the commit of every
// step, the admission and the preparation of the next window, one lane per slot.  What crosses workgroups inside the launch
// are the evaluation's results only (slot fields, conflict rows, the cp words, trunc / exclusive): written with agent-scope
// stores and atomics, drained (s_waitcnt) before the arrival, read with agent-scope loads behind it -- the hand-off of the
// fused sweep (pmc_kernels.h, publish_done), with no cache write-back or invalidate.  (Four phases behind grid barriers in one
// persistent launch were measured first: 7-9 us per barrier, slower than the launches -- LAB.md.)  Same chain, bit for bit.
#define SQ_FUSED_KMAX 256

// (AG: the evaluation ran in this launch -- k_sq_round_fused; otherwise in the launch before -- k_sq_tail)
template <int D, int NW, bool AG>
__device__ __forceinline__ void sq_commit_lanes(const SqArgs &s, unsigned long long *cp_lds)
{
    const RefArgs &a = s.a;
    const int lane = wave_lane();
    const int wv = (NW == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    SqCtl *ctl = s.ctl;
    int nslots = ctl->nslots, cur = ctl->cur;
    asm volatile("" : "+v"(nslots), "+v"(cur));
    nslots = __builtin_amdgcn_readfirstlane(nslots);
    cur = __builtin_amdgcn_readfirstlane(cur);
    SqSlot *slots = cur ? s.slots[1] : s.slots[0];
    // a wavefront takes blocks of BL consecutive slots: 16 while that keeps every wavefront of the workgroup busy (the steps that
    // edit rows are whole-wavefront operations, one after the other within a wavefront), else 64
    const int BL = (nslots <= NW * 16) ? 16 : PMC_WAVE;
    // -- everything the first block needs of its slots, in one round trip (and the cp words, trunc, exclusive with it)
    struct Pre { int i, j, c0, c1, c2, pk3, flags, m; unsigned long long g; double xn[3], xr[3], de; } r;
    auto fetch = [&](int w, bool valid, Pre &o) {
        const SqSlot &sl = slots[valid ? w : 0];
        o.i = sl.i; o.j = sl.j; o.c0 = sl.c0; o.c1 = sl.c1; o.pk3 = sl.pk[3]; o.m = sl.m; o.g = sl.g;
        o.flags = AG ? ld_agent(&sl.flags) : sl.flags;
        o.c2 = AG ? ld_agent(&sl.c2) : sl.c2;
        o.de = AG ? ld_agent(&sl.de) : sl.de;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            o.xn[q] = (q < D) ? sl.xn[q] : 0.0;
            o.xr[q] = (q < D) ? (AG ? ld_agent(&sl.xr[q]) : sl.xr[q]) : 0.0;
        }
    };
    int base = wv * BL;
    fetch(base + lane, lane < BL && base + lane < nslots, r);
    const int trunc = AG ? ld_agent(&ctl->trunc) : ctl->trunc, excl = AG ? ld_agent(&ctl->exclusive) : ctl->exclusive;
    for (int k = threadIdx.x; k < (nslots + 63) / 64; k += NW * PMC_WAVE) {
        cp_lds[k] = AG ? ld_agent(&s.cp[k]) : s.cp[k];
        cp_lds[SQ_KW + k] = AG ? ld_agent(&s.cp[SQ_KW + k]) : s.cp[SQ_KW + k];
    }
    __syncthreads();
    for (bool first = true; base < nslots; base += NW * BL, first = false) {
        const int w = base + lane;
        const bool valid = lane < BL && w < nslots;
        if (!first) fetch(w, valid, r);
        const int i = r.i, c0 = r.c0, c1 = r.c1, flags = valid ? r.flags : 0;
        if (valid && r.pk3) {                               // (atomics of the next preparation act on these words)
            st_agent(&s.cellhead[c0], -1);
            st_agent(&s.cellhead[c1], -1);
        }
        // the hold rule of sq_commit_wave; a step without the P bit has an empty row
        unsigned long long hit = 0;
        const int nwords = ((base + BL - 1) >> 6) + 1;      // (the same for every lane of the wavefront)
        const bool has_row = valid && ((cp_lds[SQ_KW + (w >> 6)] >> (w & 63)) & 1ull) != 0;
        if (__any(has_row)) {
            const unsigned long long *row = s.M + (size_t)(valid ? w : 0) * SQ_KW;
            for (int k0 = 0; k0 < nwords; k0 += 8) {
                unsigned long long m[8];
#pragma unroll
                for (int q = 0; q < 8; q++) m[q] = (has_row && k0 + q < nwords) ? (AG ? ld_agent(&row[k0 + q]) : row[k0 + q]) : 0ull;
#pragma unroll
                for (int q = 0; q < 8; q++) if (k0 + q < nwords) hit |= m[q] & cp_lds[k0 + q];
            }
        }
        const bool hold = !valid || hit != 0 || w >= trunc || (excl && w != 0);
        int c2 = c0;
        bool rows = false;
        if (!hold && !(flags & SQF_NULL)) {
            if (!(flags & SQF_SWAP)) {
                if (flags & SQF_ACCEPT) {
#pragma unroll
                    for (int q = 0; q < D; q++) a.pos[(size_t)i * D + q] = r.xn[q];
                    rows = c1 != c0;
                } else if (flags & SQF_CHANGED) {
                    c2 = r.c2;
#pragma unroll
                    for (int q = 0; q < D; q++) a.pos[(size_t)i * D + q] = r.xr[q];
                    rows = (c1 != c0) || (c2 != c1);
                }
            } else if (flags & SQF_ACCEPT) {
                rows = true;
            }
        }
        if (!hold) {                                        // (before the row operations: nothing below reads what they write)
            SqLog &lg = s.log[r.g & s.ring_mask];
            lg.de = r.de;
            lg.tag = (uint32_t)r.g;
            int f = SQL_DONE | (r.m << 4);
            if (flags & SQF_ACCEPT) f |= SQL_ACCEPT;
            if (flags & SQF_FINITE) f |= SQL_FINITE;
            if (flags & SQF_NULL) f |= SQL_NULL;
            lg.flag = f;
            st_agent(&slots[w].flags, flags | SQF_COMMITTED);
        }
        // steps that edit cell rows or the SpeciesList: whole-wavefront operations, one step after the other (they touch
        // disjoint cells / species entries: the order among them is free, as it is among the workgroups of k_sq_commit)
        unsigned long long todo = __ballot(rows);
        while (todo) {
            const int l = (int)__builtin_ctzll(todo);
            todo &= todo - 1;
            const int fl = __shfl(flags, l, PMC_WAVE), ii = __shfl(i, l, PMC_WAVE);
            const int d0 = __shfl(c0, l, PMC_WAVE), d1 = __shfl(c1, l, PMC_WAVE), d2 = __shfl(c2, l, PMC_WAVE);
            if (!(fl & SQF_SWAP)) {
                if (fl & SQF_ACCEPT) {
                    sq_row_remove(s, d0, ii); sq_row_append(s, d1, ii);
                } else if (d1 != d0 && d2 == d0) {
                    sq_row_remove(s, d0, ii); sq_row_append(s, d0, ii);
                } else {
                    if (d1 != d0) { sq_row_remove(s, d0, ii); sq_row_append(s, d1, ii); }
                    if (d2 != d1) { sq_row_remove(s, d1, ii); sq_row_append(s, d2, ii); }
                }
            } else {
                const int jj = __shfl(r.j, l, PMC_WAVE), mm = __shfl(r.m, l, PMC_WAVE);
                if (lane == 0) {
                    const SqSlot &sq = slots[base + l];
                    const int spi = AG ? ld_agent(&sq.spi) : sq.spi, spj = AG ? ld_agent(&sq.spj) : sq.spj;
                    a.species[ii] = spj;                            // src/moves.jl:162-163
                    a.species[jj] = spi;
                    ref_update_species_list(a, s.pool.sp0[mm], s.pool.sp1[mm], ii, jj);
                }
            }
        }
    }
}

template <int D, int KIND = 4, int NW = 1>
__global__ __launch_bounds__(NW * 64) void k_sq_round_fused(const SqArgs *__restrict__ tab)
{
    const SqArgs &s = tab[blockIdx.y];
    extern __shared__ int sq_lds[];              // the evaluation's ids and term buffers; the last workgroup also: the admission's hash table
    __shared__ int fill_lds[NW + 4];
    __shared__ unsigned long long cp_lds[2 * SQ_KW];
    SqCtl *ctl = s.ctl;
    const int b = blockIdx.x;
    PMC_TIME(const unsigned long long t_in = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) atomicMin(&ctl->t_first, t_in);)
    if (!(s.dbg & 4)) sq_eval<D, KIND, NW>(s, b, sq_lds);             // (returns at once for a slot beyond the window's steps)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wavefront's stores and atomics have landed
    __syncthreads();                                                  // ... every wavefront's of the workgroup
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(&ctl->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (old == gridDim.x - 1u) ? 1 : 0;
        if (last) __hip_atomic_store(&ctl->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fill_lds[NW + 3] = last;
    }
    __syncthreads();
    if (!fill_lds[NW + 3]) return;
    asm volatile("" ::: "memory");
    PMC_TIME(const unsigned long long t1 = __builtin_amdgcn_s_memtime();)
    sq_commit_lanes<D, NW, true>(s, cp_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PMC_TIME(const unsigned long long t2 = __builtin_amdgcn_s_memtime();)
    sq_fill_block<D, NW * PMC_WAVE, true>(s, sq_lds, fill_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    PMC_TIME(const unsigned long long t3 = __builtin_amdgcn_s_memtime();)
    for (int t = threadIdx.x; t < 2 * s.K; t += NW * PMC_WAVE) sq_prep_thread<D>(s, t);
    PMC_TIME(asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
             if (threadIdx.x == 0) {
                 const unsigned long long t4 = __builtin_amdgcn_s_memtime(), tf = ld_agent(&ctl->t_first);
                 ctl->tacc[0] += t_in - tf;            // first workgroup of the launch -> entry of the last one to arrive
                 ctl->tacc[1] += t1 - t_in;            // its evaluation
                 ctl->tacc[2] += t2 - t1;              // commit
                 ctl->tacc[3] += t3 - t2;              // admission
                 ctl->tacc[4] += t4 - t3;              // preparation
                 if (ctl->t_end) ctl->tacc[5] += tf - ctl->t_end;     // end of the last launch -> first workgroup of this one
                 ctl->tacc[6] += 1;
                 ctl->t_end = t4;
                 ctl->t_first = ~0ull;
             })
}

// ---- commit, admission and preparation by ONE workgroup ---------------------------------------------------------------------------
// With no pending step (the first launch of a call) this just admits and prepares the first window: the prologue of the fused
// rounds.  (As the second launch of every round of a LARGER window -- k_sq_round_a, then this, instead of k_sq_commit +
// k_sq_fill + k_sq_prep -- it was measured and lost: LAB.md, round 5.  That launch form is gone from the library.)
#define SQ_TAIL_THREADS 1024

template <int D>
__global__ __launch_bounds__(SQ_TAIL_THREADS) void k_sq_tail(const SqArgs *__restrict__ tab)
{
    const SqArgs &s = tab[blockIdx.y];
    extern __shared__ int sq_table[];            // keys[4 K], vals[4 K]
    __shared__ int lds[SQ_TAIL_THREADS / 64 + 2];
    __shared__ unsigned long long cp_lds[2 * SQ_KW];
    static_assert(SQ_KMAX == 4 * SQ_TAIL_THREADS, "sq_fill_block admits for windows of up to four slots per thread");
    sq_commit_lanes<D, SQ_TAIL_THREADS / 64, false>(s, cp_lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sq_fill_block<D, SQ_TAIL_THREADS, false>(s, sq_table, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int t = threadIdx.x; t < 2 * s.K; t += SQ_TAIL_THREADS) sq_prep_thread<D>(s, t);
}

// The rows of a LinkedList whose chains already exist (moves have shaped their order: src/neighbours.jl:276-302): the row of
// a cell is its chain read backwards.  One thread per cell; overflow of a row sets bit 0 of *err.
PMC_GLOBAL void k_sq_rows_from_chains(int ncell, const int *__restrict__ head, const int *__restrict__ list, int *__restrict__ items,
                                      int cap, int *__restrict__ ccount, int *__restrict__ err)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncell) return;
    int n = 0;
    for (int j = head[c]; j != -1; j = list[j]) n++;
    ccount[c] = n;
    if (n > cap) { atomicOr(err, 1); return; }
    int k = n - 1;
    for (int j = head[c]; j != -1; j = list[j]) items[(size_t)c * cap + k--] = j;
}

// rows of a LinkedList from the CSR the build leaves (ids ascending inside a cell = the chain read backwards): k_ref_cellvec

// ================================================================================================================================
// Small systems, many chains: the whole chain state in LDS (engine 3)
// ================================================================================================================================
// The reference's own workloads are small systems run as many chains (benchmark/particles_benchmarks.jl:10-28: N = 216;
// test/ss142d.jl:8-21: N = 8, M = 100 chains; parallel = true, src/ParticlesMC.jl:164,246).  A box of a few cells per side has
// nothing for the windows above to work with -- every step interacts with every other -- but its whole state (positions,
// species, cell rows, SpeciesList) is a few KB: ONE wavefront per chain keeps it in LDS and walks the chain step by step with no
// global-memory traffic at all; a launch runs as many chains as it is given, one per SIMD.  Same arithmetic, same iterator
// order, same bits as above.  The random parts of 64 steps are generated at once, one step per lane.

struct SqChain {             // per-chain arguments (device table)
    double *pos;
    int *species;
    int *cs, *head, *list, *ccount, *cl_items;
    int *sp_ids, *sp_heads;
    const int *sp_off;
    uint32_t key0, key1;
    int pad0, pad1;
    unsigned long long step0;
    double energy;           // in: system.energy[1]; out: after the steps
    double temperature;
    unsigned long long att[8], acc[8];   // out (added to)
    int err;
    int pad2;
};

struct SqLdsArgs {
    RefArgs a;               // geometry, models, neighbour-cell table, cl_cap (pointers to chain state unused: see SqChain)
    double invbox[3];
    double invcell[3];       // 1 / a.cell (sq_cell_of_fast)
    SqChain *chains;
    MovePool pool;
    long long nsteps;
    int reverse;
    int ncell;
    int empty;               // EmptyList (src/neighbours.jl:13-44): every particle is a neighbour, j in 1:N; no cells, no rows
    int pos_global;          // an EmptyList system too large for the LDS: everything but the positions is in LDS, the positions stay in global memory
    // byte offsets of the arrays inside the dynamic LDS block
    int o_rows, o_ccount, o_cs, o_spids, o_spheads, o_species, o_seq, o_gen, o_nbr, o_nbrcnt, o_spoff, o_terms, o_tmask, o_esum, o_models;
    int nbr_in_lds;
    int max_trips;           // trips of 64 candidates a neighbourhood can have (3^d rows of cl_cap ids)
    int dbg;                 // PMC_SEQ_DBG, timing experiments only (results invalid): 1 no cell of the trial position, 2 no ordered sums, 4 no pair terms, 8 no sequence
};

#ifdef SQ_TIMING
#define SQT(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); sqt_acc[k] += now_ - sqt_last; sqt_last = now_; } while (0)
#else
#define SQT(k)
#endif

struct SqLdsView {
    double *pos;                 // positions in LDS ...
    double *gpos;                // ... or, for the PG instances (SqLdsArgs::pos_global), in global memory
    unsigned short *rows, *ccount, *cs, *spids, *spheads, *seq;
    unsigned char *species;
    SqGen *gen;
    const int *nbr, *nbrcnt;
    int *spoff;
    double *terms;               // the non-zero pair terms of the two sums of a pass, packed in iterator order (sq_terms_push)
    int *tcount;                 // NW > 1: [2][trips] terms per trip
    double *esum;                // [4]: results of the ordered sums
    const DevModel *models;      // the S x S pair models, copied into LDS (a pair term's two dependent loads of its model were L1 round trips)
};

// NW wavefronts per chain: team helpers (NW == 1: no barrier instructions beyond the wave's own)
template <int NW> __device__ __forceinline__ int sql_wave() { return NW == 1 ? 0 : (int)(threadIdx.x >> 6); }

// The ids the iterator yields for a particle in cell c, in order, into v.seq; returns their number.  Row l is copied by wavefront
// l mod NW.
template <int D, int NW>
__device__ __forceinline__ int sql_sequence(const SqLdsArgs &s, const SqLdsView &v, int c)
{
    constexpr int NN = (D == 3) ? 27 : 9;
    const int lane = wave_lane(), wv = sql_wave<NW>();
    const int cap = s.a.cl_cap;
    const int cnt = v.nbrcnt[c];
    const int cc = (lane < cnt) ? v.nbr[(size_t)c * s.a.nnmax + lane] : 0;
    const int len = (lane < cnt) ? (int)v.ccount[cc] : 0;
    int incl = len;
#pragma unroll
    for (int sft = 1; sft < 32; sft <<= 1) {
        const int t = __shfl_up(incl, sft, PMC_WAVE);
        if (lane >= sft) incl += t;
    }
    const int off = incl - len;
    const int total = __shfl(incl, 31, PMC_WAVE);
    __syncthreads();                               // the previous pass has read v.seq
#pragma unroll
    for (int l = 0; l < NN; l++) {
        if (NW > 1 && (l % NW) != wv) continue;
        const int len_l = __builtin_amdgcn_readlane(len, l);
        const int cc_l = __builtin_amdgcn_readlane(cc, l);
        const int off_l = __builtin_amdgcn_readlane(off, l);
        for (int k = lane; k < len_l; k += PMC_WAVE)
            v.seq[off_l + (s.reverse ? len_l - 1 - k : k)] = v.rows[cc_l * cap + k];
    }
    __syncthreads();
    return total;
}

template <int D>
__device__ __forceinline__ double sql_dist2(const SqLdsArgs &s, const double *xi, const double *xj)
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

// compute_energy_particle at xa (and xb when BOTH) over the neighbourhood of cell c; species spa at xa, spb at xb; `other`
// (a swap partner, or -1) is seen with species sp_other at xb (src/moves.jl:159-167).  The trips of 64 candidates are dealt to
// the NW wavefronts; the two sums are then taken in iterator order by wavefront 0 (ea) and wavefront 1 (eb).
// KIND: the kind every pair model of the system has (the potential is then compiled in: the kernel of a Lennard-Jones or an
// integer-power soft-sphere system carries no other potential), or 4: read from the model, pair by pair.
#define SQL_TRGO 16              // NW > 1: trips whose packed terms the LDS holds at once (a pass of more trips runs in several goes)

// (PG: the positions are in global memory -- a compile-time choice, so that the LDS instances keep their ds_read accesses)
template <int D, bool BOTH, int NW, int KIND = 4, bool PG = false>
__device__ __forceinline__ void sql_pass(const SqLdsArgs &s, const SqLdsView &v, int c, int i, int spa, int spb, int other, int sp_other,
                                         const double *xa, const double *xb, double &ea, double &eb)
{
    const int lane = wave_lane(), wv = sql_wave<NW>();
    const double *const P = PG ? v.gpos : v.pos;
    const int S = s.a.S;
    const int total = s.empty ? s.a.N : ((s.dbg & 8) ? 216 : sql_sequence<D, NW>(s, v, c));
    if (s.empty && NW > 1) __syncthreads();                        // (the barrier sql_sequence ends with: the previous pass has read its terms)
    const int ntrip = (total + PMC_WAVE - 1) / PMC_WAVE;
    constexpr int SEG = PMC_WAVE + SQ_TERMS_PAD;                  // NW > 1: one segment of packed terms per trip and sum
    constexpr int BUF = SQ_TERMS + PMC_WAVE + SQ_TERMS_PAD;      // NW == 1: one buffer per sum, summed down when nearly full
    double acc_a = 0.0, acc_b = 0.0;
    // one candidate's two terms
    auto terms = [&](int t, int j, const double *xj, int sj, double &ua, double &ub) {
        ua = 0.0; ub = 0.0;
        if (t < total && !(s.dbg & 4) && j != i) {
            const DevModel &ma = v.models[spa * S + sj];
            const double r2a = sql_dist2<D>(s, xa, xj);
            if (!(r2a > ma.rcut2)) ua = (KIND == 4 ? dev_potential_dyn(r2a, ma) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2a, ma));
            if (BOTH) {
                const DevModel &mb = v.models[spb * S + (j == other ? sp_other : sj)];
                const double r2b = sql_dist2<D>(s, xb, xj);
                if (!(r2b > mb.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2b, mb) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2b, mb));
            }
        }
    };
    if constexpr (NW == 1) {
        int na = 0, nb = 0;
        double *bufa = v.terms, *bufb = v.terms + BUF;
        for (int tr = 0; tr < ntrip; tr++) {
            const int t = tr * PMC_WAVE + lane;
            const int j = (t < total) ? (s.empty ? t : (int)v.seq[t]) : 0;
            double xj[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < D; q++) xj[q] = P[j * D + q];
            const int sj = (S > 1) ? (int)v.species[j] : 0;
            double ua, ub;
            terms(t, j, xj, sj, ua, ub);
            if (s.dbg & 2) continue;
            if (na > SQ_TERMS || nb > SQ_TERMS) {                // the next 64 terms might not fit: sum down, in order
                sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
                na = nb = 0;
            }
            sq_terms_push(bufa, na, ua);
            if (BOTH) sq_terms_push(bufb, nb, ub);
        }
        if (!(s.dbg & 2)) sq_terms_sum2(bufa, na, acc_a, bufb, nb, acc_b);
    } else {
        // the trips in goes of SQL_TRGO, dealt round-robin; a wavefront's trips of a go have their loads in flight together (the
        // positions of a large EmptyList system come from global memory: SqLdsArgs::pos_global); every trip packs its non-zero
        // terms into its segment; wavefront 0 then adds the first sum's segments in order, wavefront 1 the second's
        static_assert(SQL_TRGO % NW == 0, "the trips of a go are dealt evenly");
        constexpr int PER = SQL_TRGO / NW;
        const int gtr = PG ? SQL_TRGO : s.max_trips;              // segments in LDS (seq_lds_layout)
        double *bufa = v.terms, *bufb = v.terms + gtr * SEG;
        double acc = 0.0;
        if constexpr (!PG) {
            // state in LDS: one trip after the other per wavefront, a segment for every trip the neighbourhood can have
            // (measured: the goes below, compiled into these instances, cost chains of N = 216 9-17 %)
            for (int tr = wv; tr < ntrip; tr += NW) {
                const int t = tr * PMC_WAVE + lane;
                double ua = 0.0, ub = 0.0;
                if (t < total && !(s.dbg & 4)) {
                    const int j1 = s.empty ? t : (int)v.seq[t];
                    if (j1 != i) {
                        double x1[3] = {0, 0, 0};
#pragma unroll
                        for (int q = 0; q < D; q++) x1[q] = P[j1 * D + q];
                        const int s1 = (S > 1) ? (int)v.species[j1] : 0;
                        const DevModel &ma = v.models[spa * S + s1];
                        const double r2a = sql_dist2<D>(s, xa, x1);
                        if (!(r2a > ma.rcut2)) ua = (KIND == 4 ? dev_potential_dyn(r2a, ma) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2a, ma));
                        if (BOTH) {
                            const DevModel &mb = v.models[spb * S + (j1 == other ? sp_other : s1)];
                            const double r2b = sql_dist2<D>(s, xb, x1);
                            if (!(r2b > mb.rcut2)) ub = (KIND == 4 ? dev_potential_dyn(r2b, mb) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2b, mb));
                        }
                    }
                }
                const int ca = sq_terms_push_seg(bufa + tr * SEG, ua);
                int cb = 0;
                if (BOTH) cb = sq_terms_push_seg(bufb + tr * SEG, ub);
                if (lane == 0) { v.tcount[tr] = ca; v.tcount[gtr + tr] = cb; }
            }
            __syncthreads();
            if ((wv == 0 || (BOTH && wv == 1)) && !(s.dbg & 2)) {
                double *seg = wv == 0 ? bufa : bufb;
                if (ntrip <= PMC_WAVE) {                              // one straight chain, the next segment's block in flight (sq_terms_chain)
                    acc = sq_terms_chain(seg, v.tcount + wv * gtr, ntrip, acc);
                } else {
                    for (int tr = 0; tr < ntrip; tr++) acc = sq_terms_sum(seg + tr * SEG, v.tcount[wv * gtr + tr], acc);
                }
            }
        } else
        for (int tr0 = 0; tr0 < ntrip; tr0 += SQL_TRGO) {
            int j[PER], sj[PER];
            double xj[PER][3];
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int t = (tr0 + wv + u * NW) * PMC_WAVE + lane;
                j[u] = 0;
                if (tr0 + wv + u * NW < ntrip) j[u] = (t < total) ? (s.empty ? t : (int)v.seq[t]) : 0;
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                xj[u][0] = xj[u][1] = xj[u][2] = 0.0;
                sj[u] = 0;
                if (tr0 + wv + u * NW < ntrip) {                    // (a wavefront without a u-th trip in this go loads nothing for it)
#pragma unroll
                    for (int q = 0; q < D; q++) xj[u][q] = P[j[u] * D + q];
                    if (S > 1) sj[u] = (int)v.species[j[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < PER; u++) {
                const int tr = tr0 + wv + u * NW;
                if (tr >= ntrip) break;
                double ua, ub;
                terms(tr * PMC_WAVE + lane, j[u], xj[u], sj[u], ua, ub);
                const int sg = tr - tr0;
                const int ca = sq_terms_push_seg(bufa + sg * SEG, ua);  // (eight zeros behind the terms: the chain reads whole blocks)
                int cb = 0;
                if (BOTH) cb = sq_terms_push_seg(bufb + sg * SEG, ub);
                if (lane == 0) { v.tcount[sg] = ca; v.tcount[gtr + sg] = cb; }
            }
            __syncthreads();
            const int nt = min(SQL_TRGO, ntrip - tr0);
            if (!(s.dbg & 2)) {
                if (wv == 0) acc = sq_terms_chain(bufa, v.tcount, nt, acc);
                else if (BOTH && wv == 1) acc = sq_terms_chain(bufb, v.tcount + gtr, nt, acc);
            }
            if (tr0 + SQL_TRGO < ntrip) __syncthreads();            // (the segments are free for the next go)
        }
        if (lane == 0 && wv == 0) v.esum[0] = acc;
        if (lane == 0 && wv == 1) v.esum[1] = BOTH ? acc : 0.0;
        __syncthreads();
        acc_a = v.esum[0];
        if (BOTH) acc_b = v.esum[1];
    }
    ea = acc_a;
    eb = acc_b;
}

// update_neighbour_list!: i leaves the row of `from` (order kept) and is appended to the row of `to`.  Wavefront 0 works, every
// wavefront of the chain takes part in the barriers.
template <int NW>
__device__ __forceinline__ void sql_row_move(const SqLdsArgs &s, const SqLdsView &v, int from, int to, int i)
{
    const int lane = wave_lane();
    const bool w0 = sql_wave<NW>() == 0;
    const int cap = s.a.cl_cap;
    unsigned short *row = v.rows + from * cap;
    const int n = v.ccount[from];
    int p = -1;
    for (int base = 0; base < n; base += PMC_WAVE) {        // (every wavefront searches: p stays uniform over the chain's threads)
        const int k = base + lane;
        const unsigned long long mk = __ballot(k < n && row[k] == i);
        if (mk) { p = base + (int)__builtin_ctzll(mk); break; }
    }
    __syncthreads();
    if (p >= 0) {
        for (int base = p; base < n - 1; base += PMC_WAVE) {
            const int k = base + lane;
            unsigned short t = 0;
            if (w0 && k < n - 1) t = row[k + 1];
            __syncthreads();
            if (w0 && k < n - 1) row[k] = t;
            __syncthreads();
        }
    }
    if (w0 && lane == 0) {
        const int nfrom = (p >= 0) ? n - 1 : n;
        v.ccount[from] = (unsigned short)nfrom;
        const int n2 = (from == to) ? nfrom : (int)v.ccount[to];
        if (n2 < cap) {
            v.rows[to * cap + n2] = (unsigned short)i;
            v.ccount[to] = (unsigned short)(n2 + 1);
        } else {
            s.chains[blockIdx.x].err |= 1;          // a cell outgrew its row: the host reports it and rebuilds
        }
        if (p < 0) s.chains[blockIdx.x].err |= 2;
        v.cs[i] = (unsigned short)to;
    }
    __syncthreads();
}

template <int D, int NW, int KIND = 4, bool PG = false>
__global__ __launch_bounds__(NW * 64) void k_sq_lds(SqLdsArgs s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sql_smem[];
    const RefArgs &a = s.a;
    SqChain &ch = s.chains[blockIdx.x];
    const int tid = threadIdx.x, nth = NW * 64;
    const int N = a.N, S = a.S, cap = a.cl_cap;
    SqLdsView v;
    v.pos = reinterpret_cast<double *>(sql_smem);
    v.gpos = ch.pos;
    double *const P = PG ? v.gpos : v.pos;
    v.rows = reinterpret_cast<unsigned short *>(sql_smem + s.o_rows);
    v.ccount = reinterpret_cast<unsigned short *>(sql_smem + s.o_ccount);
    v.cs = reinterpret_cast<unsigned short *>(sql_smem + s.o_cs);
    v.spids = reinterpret_cast<unsigned short *>(sql_smem + s.o_spids);
    v.spheads = reinterpret_cast<unsigned short *>(sql_smem + s.o_spheads);
    v.species = sql_smem + s.o_species;
    v.seq = reinterpret_cast<unsigned short *>(sql_smem + s.o_seq);
    v.gen = reinterpret_cast<SqGen *>(sql_smem + s.o_gen);
    v.spoff = reinterpret_cast<int *>(sql_smem + s.o_spoff);
    v.terms = reinterpret_cast<double *>(sql_smem + s.o_terms);
    v.tcount = reinterpret_cast<int *>(sql_smem + s.o_tmask);
    v.esum = reinterpret_cast<double *>(sql_smem + s.o_esum);
    {
        DevModel *lm = reinterpret_cast<DevModel *>(sql_smem + s.o_models);
        static_assert(sizeof(DevModel) % sizeof(int) == 0, "DevModel is copied word by word");
        const int *src = reinterpret_cast<const int *>(a.models);
        int *dst = reinterpret_cast<int *>(lm);
        for (int k = tid; k < S * S * (int)(sizeof(DevModel) / sizeof(int)); k += nth) dst[k] = src[k];
        v.models = lm;
    }
    if (s.nbr_in_lds) {
        int *lnbr = reinterpret_cast<int *>(sql_smem + s.o_nbr), *lcnt = reinterpret_cast<int *>(sql_smem + s.o_nbrcnt);
        for (int k = tid; k < s.ncell * a.nnmax; k += nth) lnbr[k] = a.nbr[k];
        for (int k = tid; k < s.ncell; k += nth) lcnt[k] = a.nbrcnt[k];
        v.nbr = lnbr; v.nbrcnt = lcnt;
    } else {
        v.nbr = a.nbr; v.nbrcnt = a.nbrcnt;
    }
    // -- the chain's state into LDS
    if (!PG) for (int k = tid; k < N * D; k += nth) v.pos[k] = ch.pos[k];
    for (int k = tid; k < N; k += nth) {
        v.species[k] = (unsigned char)ch.species[k];
        v.cs[k] = (unsigned short)ch.cs[k];
        v.spids[k] = (unsigned short)ch.sp_ids[k];
        v.spheads[k] = (unsigned short)ch.sp_heads[k];
    }
    for (int k = tid; k <= S; k += nth) v.spoff[k] = ch.sp_off[k];
    for (int c = 0; c < (s.empty ? 0 : s.ncell); c++) {
        const int n = ch.ccount[c];
        if (tid == 0) v.ccount[c] = (unsigned short)n;
        for (int k = tid; k < n; k += nth) v.rows[c * cap + k] = (unsigned short)ch.cl_items[(size_t)c * cap + k];
    }
    __syncthreads();
    double E = ch.energy;
    const double T = ch.temperature;
    unsigned long long att[8], acc[8];
#pragma unroll
    for (int q = 0; q < 8; q++) { att[q] = 0; acc[q] = 0; }
    const unsigned long long gend = ch.step0 + (unsigned long long)s.nsteps;
    constexpr int NGEN = NW * 64;                           // random parts generated at once, one step per thread
    for (unsigned long long g0 = ch.step0; g0 < gend; g0 += NGEN) {
        {
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
            __syncthreads();
            v.gen[tid] = o;
            __syncthreads();
        }
        const int nt = (int)min((unsigned long long)NGEN, gend - g0);
        for (int t = 0; t < nt; t++) {
            const SqGen ge = v.gen[t];
            const int m = ge.m;
            bool ok = false;
            if (s.pool.kind[m] == 0) {
                const int i = (int)mulhi32(ge.r1, (uint32_t)N);
                double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0};
#pragma unroll
                for (int q = 0; q < D; q++) {
                    x[q] = P[i * D + q];
                    xn[q] = x[q] + ge.d[q];
                    xr[q] = xn[q] + (-ge.d[q]);
                }
                const int c0 = s.empty ? 0 : (int)v.cs[i];    // the list is current: cs[i] is the cell of x_i
                const int c1 = (s.empty || (s.dbg & 1)) ? c0 : sq_cell_of_fast<D>(a, s.invcell, xn);
                const int spi = (S > 1) ? (int)v.species[i] : 0;
                double e1, e2, dummy;
                if (c1 == c0) {
                    sql_pass<D, true, NW, KIND, PG>(s, v, c0, i, spi, spi, -1, 0, x, xn, e1, e2);
                } else {
                    sql_pass<D, false, NW, KIND, PG>(s, v, c0, i, spi, spi, -1, 0, x, x, e1, dummy);
                    sql_pass<D, false, NW, KIND, PG>(s, v, c1, i, spi, spi, -1, 0, xn, xn, e2, dummy);
                }
                double de = 0.0;
                if (!(isinf(e1) || isinf(e2))) { de = e2 - e1; E += de; }           // src/moves.jl:13-18
                ok = ge.lnu < -(e2 - e1) / T;
                if (ok) {
                    if (tid < D) P[i * D + tid] = xn[tid];
                    if (c1 != c0) sql_row_move<NW>(s, v, c0, c1, i);
                } else {
                    E -= de;                                                         // src/moves.jl:79
                    bool chg = false;
#pragma unroll
                    for (int q = 0; q < D; q++) chg |= sq_bits_differ(xr[q], x[q]);
                    if (chg && tid < D) P[i * D + tid] = xr[tid];
                    const int c2 = (chg && !s.empty) ? sq_cell_of_fast<D>(a, s.invcell, xr) : c0;
                    if (c1 != c0) sql_row_move<NW>(s, v, c0, c1, i);
                    if (c2 != c1) sql_row_move<NW>(s, v, c1, c2, i);
                }
                __syncthreads();
            } else {
                const int sa = s.pool.sp0[m], sb = s.pool.sp1[m];
                const int na = v.spoff[sa + 1] - v.spoff[sa], nb = v.spoff[sb + 1] - v.spoff[sb];
                if (na > 0 && nb > 0) {
                    const int i = v.spids[v.spoff[sa] + mulhi32(ge.r1, (uint32_t)na)];
                    const int j = v.spids[v.spoff[sb] + mulhi32(ge.q0, (uint32_t)nb)];
                    double xi[3] = {0, 0, 0}, xj[3] = {0, 0, 0};
#pragma unroll
                    for (int q = 0; q < D; q++) { xi[q] = P[i * D + q]; xj[q] = P[j * D + q]; }
                    const int spi = v.species[i], spj = v.species[j];
                    double e1i, e2i, e1j, e2j;
                    sql_pass<D, true, NW, KIND, PG>(s, v, s.empty ? 0 : (int)v.cs[i], i, spi, spj, j, spi, xi, xi, e1i, e2i);
                    sql_pass<D, true, NW, KIND, PG>(s, v, s.empty ? 0 : (int)v.cs[j], j, spj, spi, i, spj, xj, xj, e1j, e2j);
                    const double e1 = e1i + e1j, e2 = e2i + e2j;
                    double de = 0.0;
                    if (!(isinf(e1) || isinf(e2))) { de = e2 - e1; E += de; }
                    ok = ge.lnu < -(e2 - e1) / T;
                    if (ok) {
                        if (tid == 0) {
                            v.species[i] = (unsigned char)spj;
                            v.species[j] = (unsigned char)spi;
                            // update_species_list!, src/moves.jl:175-179
                            v.spids[v.spoff[sa] + v.spheads[i]] = (unsigned short)j;
                            v.spids[v.spoff[sb] + v.spheads[j]] = (unsigned short)i;
                            const unsigned short hh = v.spheads[i];
                            v.spheads[i] = v.spheads[j];
                            v.spheads[j] = hh;
                        }
                    } else {
                        E -= de;                                                     // src/moves.jl:206
                    }
                    __syncthreads();
                }
            }
#pragma unroll
            for (int q = 0; q < 8; q++) {
                att[q] += (q == m);
                acc[q] += (q == m && ok);
            }
        }
    }
    // -- back to global memory: positions, species, SpeciesList, rows, cs, ccount, and head / list rebuilt from the rows
    __syncthreads();
    if (!PG) for (int k = tid; k < N * D; k += nth) ch.pos[k] = v.pos[k];
    for (int k = tid; k < N; k += nth) {
        ch.species[k] = v.species[k];
        ch.cs[k] = v.cs[k];
        ch.sp_ids[k] = v.spids[k];
        ch.sp_heads[k] = v.spheads[k];
    }
    for (int c = 0; c < (s.empty ? 0 : s.ncell); c++) {
        const int n = v.ccount[c];
        if (tid == 0) {
            ch.ccount[c] = n;
            if (s.reverse) ch.head[c] = n > 0 ? (int)v.rows[c * cap + n - 1] : -1;
        }
        for (int k = tid; k < n; k += nth) {
            const int id = v.rows[c * cap + k];
            ch.cl_items[(size_t)c * cap + k] = id;
            if (s.reverse) ch.list[id] = k > 0 ? (int)v.rows[c * cap + k - 1] : -1;   // the chain is the row read backwards
        }
    }
    if (tid == 0) {
        ch.energy = E;
        ch.step0 = gend;                         // the next launch of the call continues here
        for (int q = 0; q < 8; q++) { ch.att[q] += att[q]; ch.acc[q] += acc[q]; }
    }
}
