// pmc_empty.h -- engine 5 of the reference-order chain: small EmptyList systems (src/neighbours.jl:13-44: every particle is a
// neighbour of every other, summed in id order -- the list type of the reference's published benchmark,
// benchmark/particles_benchmarks.jl:10-28: N = 216 BHHP soft spheres) with their state in LDS, one wavefront RESOLVING the steps
// in order while six or twelve others EVALUATE the steps behind them, as engine 4 (pmc_verlet.h) does for VerletList systems.
//
// What is different from engine 4: in an EmptyList system every step reads every particle, so a step that stores a position
// touches the inputs of EVERY evaluation in flight -- an evaluation could hardly ever be taken as it is.  But it can be
// REPAIRED.  e1 and e2 are chains of additions over the non-zero pair terms in id order (a term that is 0 leaves a running FP64
// sum as it is).  An evaluator leaves its packed terms in LDS -- for each of the two sums the ids, the terms, and for every
// particle the place of its term in the packed list -- and the resolver, when the step's turn comes, looks up which particles
// have been stored since the evaluation began (a log of the resolver's own stores), computes for each of them the two pair terms
// against its CURRENT position, puts them where the old ones were (a particle that left the cutoff keeps a place with the term 0;
// one that entered it gets a place inserted in id order), and runs the two chains again: the same terms in the same order as a
// fresh evaluation adds them, hence the same bits, for a few hundred cycles instead of the 2 x N pair terms.  Only a step whose
// OWN particle has been stored in between is evaluated again from scratch (by the resolver, into a buffer of its own).
//
// Pools of Displacements only (a swap changes the species of two particles: every term of both would have to be patched under
// another pair model; such pools run on engine 3).  Same arithmetic, same order, same bits as orc_mc_steps_sequential.
#pragma once
#include "pmc_verlet.h"

#ifndef SQE_LOOK
#define SQE_LOOK 5                // an evaluator begins step t once the resolver has finished step t - SQE_LOOK - 1 (SQE_LOOK < 6 evaluators)
#endif
// Sixteen wavefronts (the resolver, the generator, twelve evaluators) or eight (six evaluators).  The resolver bounds the rate and
// SQE_LOOK the evaluations in flight, so six evaluators serve a chain nearly as well as twelve (1.30 against 1.26 us per step of
// the reference's benchmark system) -- and two such chains share a CU (four wavefronts per SIMD at 128 VGPRs, their blocks in half
// the LDS each): 3.1e8 moves/s from 512 chains on, against 2.0e8 with one chain per CU.  The host picks (seq_empty_waves).
#define SQE_LOG 64                // the resolver's log of stores: a ring (an evaluation is at most 3 NW / 4 steps old)

struct SqeArgs;
const void *pmc_sqe_kernel(int d, int kind, int nw);      // pmc_verlet_inst.hip: k_sq_empty<d, kind, nw> (nw: 8 or 16 wavefronts)
size_t pmc_sqe_layout(int n, int d, int nspecies, int nw, SqeArgs *s);   // ... the LDS layout of its workgroup

struct SqeChain {                 // per-chain arguments (device table)
    double *pos;
    int *species;
    uint32_t key0, key1;
    unsigned long long step0;     // in: first step of the launch; out: one past the last
    double energy;                // in: system.energy[1]; out: after the steps
    double temperature;
    unsigned long long att[8], acc[8];                        // out (added to)
    unsigned long long n_patched, n_again, n_inserted, n_polls;   // diagnostics (added to): steps repaired, evaluated again, terms inserted, resolver polls
};

struct SqeArgs {
    RefArgs a;                    // geometry, models
    double invbox[3];
    SqeChain *chains;
    MovePool pool;
    long long nsteps;
    int kc;                       // capacity of a packed list (N rounded up, + 8)
    int o_species, o_gen, o_spec, o_models, o_log, o_ctl, o_ev, ev_stride;   // byte offsets; an evaluator's block: terms1, terms2 [kc doubles], ids1, ids2 [kc u16], idx1, idx2 [N u16]
};

struct __attribute__((aligned(16))) SqeSpec {       // evaluator -> resolver
    double e1, e2;                // the two sums as evaluated
    int i, c1, c2, pad0;          // the particle; terms in the two packed lists
    int s0;                       // the steps the resolver had finished when the evaluation began
    int seq;                      // step + 1 once the record is complete (stored last, read first)
    int pad1, pad2;
};

struct SqeBlock {                 // one evaluator's (or the resolver's own) packed terms
    double *t1, *t2;              // [kc] the terms, zeros behind them to the end
    double *p1, *p2;              // [kc / 8] the running sum behind every block of eight terms
    unsigned short *id1, *id2, *ix1, *ix2;
};

__device__ __forceinline__ SqeBlock sqe_block(unsigned char *base, const SqeArgs &s, int N)
{
    SqeBlock b;
    b.t1 = reinterpret_cast<double *>(base);
    b.t2 = b.t1 + s.kc;
    b.p1 = b.t2 + s.kc;
    b.p2 = b.p1 + s.kc / 8;
    b.id1 = reinterpret_cast<unsigned short *>(b.p2 + s.kc / 8);
    b.id2 = b.id1 + s.kc;
    b.ix1 = b.id2 + s.kc;
    b.ix2 = b.ix1 + N;
    return b;
}

template <int D>
__device__ __forceinline__ double sqe_dist2(const SqeArgs &s, const double *xi, const double *xj)
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

// Float64 ^ 6 exactly as dev_pow_int(x, 6) computes it (Julia's pow_body: compensated squaring, square-and-multiply), the loop
// over the bits of the exponent written out: the reference's soft spheres are r^-12 (src/models.jl:76-84), and the general
// routine's 64-bit exponent arithmetic and branches are a good part of a pair term
__device__ __forceinline__ double sqe_pow6(double x)
{
    // n = 6: square (y = 1, xnlo = ynlo = 0)
    double hi = x * x;
    double xnlo = __builtin_fma(x, x, -hi);
    double err = x * 2.0 * 0.0;
    xnlo += err;
    x = hi;
    // n = 3: multiply, then square
    err = __builtin_fma(1.0, xnlo, x * 0.0);
    hi = x * 1.0;
    double ynlo = __builtin_fma(x, 1.0, -hi);
    const double y = hi;
    ynlo += err;
    err = x * 2.0 * xnlo;
    hi = x * x;
    xnlo = __builtin_fma(x, x, -hi);
    x = hi;
    xnlo += err;
    // n = 1
    err = __builtin_fma(y, xnlo, x * ynlo);
    if (isfinite(x) && isfinite(err)) return __builtin_fma(x, y, err);
    return x * y;
}

// compute_energy_ij (src/atoms.jl:66-75) of particle i (species spi) at x with particle j as it is in LDS
template <int D, int KIND>
__device__ __forceinline__ double sqe_pair(const SqeArgs &s, const double *pos, const unsigned char *species, const DevModel *models, int spi,
                                           const double *x, int j)
{
    double xj[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) xj[q] = pos[j * D + q];
    const int S = s.a.S;
    const DevModel &m = models[spi * S + ((S > 1) ? (int)species[j] : 0)];
    const double r2 = sqe_dist2<D>(s, x, xj);
    if (r2 > m.rcut2) return 0.0;
    if (KIND == K_SOFT_INT && m.ndiv2_i == 6) return m.eps * sqe_pow6(m.sigma2 / r2) - m.shift;      // dev_potential<K_SOFT_INT>, the power written out
    return KIND == 4 ? dev_potential_dyn(r2, m) : dev_potential<(KIND == 4 ? 0 : KIND)>(r2, m);
}

// the smallest value lanes 0 .. 31 hold (DPP shifts inside the rows of 16, then the two rows' last lanes); in every lane
__device__ __forceinline__ int sqe_min32(int v)
{
    int x = v;
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x111, 0xf, 0xf, false));    // row_shr:1 (a lane without a source keeps its own)
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x112, 0xf, 0xf, false));    // row_shr:2
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xf, false));    // row_shr:4
    x = min(x, __builtin_amdgcn_update_dpp(x, x, 0x118, 0xf, 0xf, false));    // row_shr:8
    return min(__builtin_amdgcn_readlane(x, 15), __builtin_amdgcn_readlane(x, 31));
}

// The two chains of additions over the packed terms: lanes 0-31 walk the first list, lanes 32-63 the second, in blocks of eight
// (zeros lie behind the terms to the end of the buffers: nothing is masked); results in every lane.  RECORD (an evaluator): the
// running sum behind every block goes to p1 / p2.  Otherwise (the resolver, after a repair): each chain starts behind block
// b1 - 1 / b2 - 1 with the sum recorded there -- the terms before the first repaired one add up to what they added up to.
template <bool RECORD>
__device__ __forceinline__ void sqe_chains(const SqeBlock &b, int c1, int c2, int b1, int b2, double &e1, double &e2)
{
    const int lane = wave_lane(), half = lane >> 5;
    const int bs = half ? b2 : b1;                                  // first block of this lane's chain
    const double *buf = (half ? b.t2 : b.t1) + 8 * bs;
    double *pre = half ? b.p2 : b.p1;
    const int nblk = max((c1 + 7) / 8 - b1, (c2 + 7) / 8 - b2);
    double acc = (!RECORD && bs > 0) ? pre[bs - 1] : 0.0;
    double cur[8], nxt[8];
#pragma unroll
    for (int q = 0; q < 8; q++) cur[q] = buf[q];
    for (int k = 0; k < nblk; k++) {
#pragma unroll
        for (int q = 0; q < 8; q++) nxt[q] = buf[8 * (k + 1) + q];
#pragma unroll
        for (int q = 0; q < 8; q++) acc += cur[q];
        if (RECORD && (lane & 31) == 0) pre[bs + k] = acc;
#pragma unroll
        for (int q = 0; q < 8; q++) cur[q] = nxt[q];
    }
    e1 = readlane_double(acc, 0);
    e2 = readlane_double(acc, 32);
}

// Every pair term of particle i at xa and at xb, packed in id order into block b (ids, terms, and for every particle the place of
// its term, 0xffff if it has none); eight zeros behind each list.  Returns the two sums.  Called by a whole wavefront.
template <int D, int KIND>
__device__ __forceinline__ void sqe_evaluate(const SqeArgs &s, const double *pos, const unsigned char *species, const DevModel *models, const SqeBlock &b,
                                             int i, int spi, const double *xa, const double *xb, int &c1, int &c2, double &e1, double &e2)
{
    const int lane = wave_lane();
    const int N = s.a.N;
    int n1 = 0, n2 = 0;
    for (int base = 0; base < N; base += PMC_WAVE) {
        const int j = base + lane;
        double u1 = 0.0, u2 = 0.0;
        if (j < N && j != i) {
            u1 = sqe_pair<D, KIND>(s, pos, species, models, spi, xa, j);
            u2 = sqe_pair<D, KIND>(s, pos, species, models, spi, xb, j);
        }
        const unsigned long long m1 = __ballot(u1 != 0.0), m2 = __ballot(u2 != 0.0);
        const int r1 = n1 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u));
        const int r2 = n2 + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m2, 0u));
        if (u1 != 0.0) { b.t1[r1] = u1; b.id1[r1] = (unsigned short)j; }
        if (u2 != 0.0) { b.t2[r2] = u2; b.id2[r2] = (unsigned short)j; }
        if (j < N) {
            b.ix1[j] = (u1 != 0.0) ? (unsigned short)r1 : (unsigned short)0xffff;
            b.ix2[j] = (u2 != 0.0) ? (unsigned short)r2 : (unsigned short)0xffff;
        }
        n1 += __popcll(m1);
        n2 += __popcll(m2);
    }
    for (int k = n1 + lane; k < s.kc; k += PMC_WAVE) b.t1[k] = 0.0;         // zeros to the end: a chain never looks where its list stops
    for (int k = n2 + lane; k < s.kc; k += PMC_WAVE) b.t2[k] = 0.0;
    sq_wave_sync();
    c1 = n1; c2 = n2;
    sqe_chains<true>(b, n1, n2, 0, 0, e1, e2);
}

// A term for particle p enters a packed list (p has come inside the cutoff since the evaluation): a place in id order -- what lies
// behind it moves up one, and so do the places the particles behind it are noted at.  One wavefront.  Returns the new length.
__device__ __forceinline__ int sqe_insert(double *t, unsigned short *id, unsigned short *ix, int cnt, int p, double u, int &where)
{
    const int lane = wave_lane();
    int pos = 0;
    for (int base = 0; base < cnt; base += PMC_WAVE) {
        const int k = base + lane;
        pos += __popcll(__ballot(k < cnt && (int)id[k] < p));
    }
    // from the back, a trip of 64 at a time: read, then store one place up (the trips do not overlap: every trip reads what it moves)
    for (int hi = cnt; hi > pos; hi -= PMC_WAVE) {
        const int lo = max(pos, hi - PMC_WAVE);
        const int k = lo + lane;
        double tv = 0.0;
        int iv = 0;
        if (k < hi) { tv = t[k]; iv = id[k]; }
        sq_wave_sync();
        if (k < hi) { t[k + 1] = tv; id[k + 1] = (unsigned short)iv; ix[iv] = (unsigned short)(k + 1); }
        sq_wave_sync();
    }
    if (lane == 0) { t[pos] = u; id[pos] = (unsigned short)p; ix[p] = (unsigned short)pos; }
    sq_wave_sync();
    where = pos;
    return cnt + 1;
}

template <int D, int KIND, int NW>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_sq_empty(SqeArgs s)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sqe_smem[];
    constexpr int NB = 3 * NW / 4;
    static_assert(NB > SQE_LOOK && 2 * NB <= 64, "evaluators");
    const RefArgs &a = s.a;
    SqeChain &ch = s.chains[blockIdx.x];
    const int tid = threadIdx.x, nth = NW * 64, lane = wave_lane(), wv = tid >> 6;
    const int N = a.N, S = a.S;
    double *const pos = reinterpret_cast<double *>(sqe_smem);
    unsigned char *const species = sqe_smem + s.o_species;
    SqGen *const gen = reinterpret_cast<SqGen *>(sqe_smem + s.o_gen);
    SqeSpec *const spec0 = reinterpret_cast<SqeSpec *>(sqe_smem + s.o_spec);      // [NB]
    int *const logp = reinterpret_cast<int *>(sqe_smem + s.o_log);                // [SQE_LOG][2]: step, particle
    int *const ctl = reinterpret_cast<int *>(sqe_smem + s.o_ctl);
    int *const p_done = ctl + 1, *const p_gen = ctl + 2;
    const int ev = (wv & 3) ? (wv & 3) - 1 + 3 * (wv >> 2) : -1;                  // evaluator 0 .. NB - 1, or none
    const SqeBlock blk = sqe_block(sqe_smem + s.o_ev + (size_t)(ev >= 0 ? ev : NB) * s.ev_stride, s, N);     // an evaluator's block; block NB is the resolver's own
    const DevModel *models;
    {
        DevModel *lm = reinterpret_cast<DevModel *>(sqe_smem + s.o_models);
        const int *src = reinterpret_cast<const int *>(a.models);
        int *dst = reinterpret_cast<int *>(lm);
        for (int k = tid; k < S * S * (int)(sizeof(DevModel) / sizeof(int)); k += nth) dst[k] = src[k];
        models = lm;
    }
    if (tid < 24) ctl[tid] = 0;
    for (int k = tid; k < NB * (int)(sizeof(SqeSpec) / 4); k += nth) reinterpret_cast<unsigned int *>(spec0)[k] = 0u;
    for (int k = tid; k < N * D; k += nth) pos[k] = ch.pos[k];
    for (int k = tid; k < N; k += nth) species[k] = (unsigned char)ch.species[k];
    const int nsteps = (int)s.nsteps;                            // (a launch has at most 2^20 steps)
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
        double z[4];
        box_muller(Q[0], Q[1], &z[0], &z[1]);
        box_muller(Q[2], Q[3], &z[2], &z[3]);
        for (int q = 0; q < 3; q++) o.d[q] = s.pool.sigma[o.m] * z[q];
        gen[t & (SQV_GEN - 1)] = o;
    };
    if (tid < SQV_GEN && tid < nsteps) generate(tid);           // blocks 0 and 1 of the random parts
    if (tid == 0) *p_gen = 2;
    __syncthreads();
    const double T = ch.temperature;
    double E = ch.energy;
    unsigned long long npat = 0, nagain = 0, nins = 0, nwait = 0;
    unsigned int cntv = 0;                                       // per-move counters of the launch, one per lane
    if (wv == 0) {
        // ---- the RESOLVER: the steps in order
        int nsteps_r = nsteps;
        asm volatile("" : "+s"(nsteps_r));
        int nlog = 0;                                            // stores logged so far (entry k at k mod SQE_LOG)
        int slot = 0;
        for (int t = 0; t < nsteps_r; t++, slot = (slot + 1 == NB) ? 0 : slot + 1) {
            const SqeSpec *rec = spec0 + slot;
            int seq = sqv_ld_relaxed(&rec->seq);
            asm volatile("" ::: "memory");
            SqeSpec sp = *rec;
            while (seq != t + 1) {
                nwait++;
                __builtin_amdgcn_s_sleep(1);
                seq = sqv_ld_relaxed(&rec->seq);
                asm volatile("" ::: "memory");
                sp = *rec;
            }
            SQVT(0);
            const SqGen ge = gen[t & (SQV_GEN - 1)];
            const int m = ge.m, i = sp.i;
            const SqeBlock eb = sqe_block(sqe_smem + s.o_ev + (size_t)slot * s.ev_stride, s, N);
            double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0}, xr[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < D; q++) {
                x[q] = pos[i * D + q];
                xn[q] = x[q] + ge.d[q];
                xr[q] = xn[q] + (-ge.d[q]);
            }
            const int spi = (S > 1) ? (int)species[i] : 0;
            // the stores since the evaluation began: lane l looks at the l-th newest entry of the log
            int lp = -1;
            {
                const int k = nlog - 1 - lane;
                if (k >= 0 && lane < SQE_LOG) {
                    const int st = logp[2 * (k & (SQE_LOG - 1))];
                    if (st >= sp.s0) lp = logp[2 * (k & (SQE_LOG - 1)) + 1];
                }
            }
            const unsigned long long pm = __ballot(lp >= 0);
            SQVT(1);
            double e1 = sp.e1, e2 = sp.e2;
            if (pm) {
                if (__ballot(lp == i)) {
                    // the particle itself has been stored since: its evaluation described another position -- again, from scratch
                    nagain++;
                    int c1, c2;
                    sqe_evaluate<D, KIND>(s, pos, species, models, blk, i, spi, x, xn, c1, c2, e1, e2);
                } else {
                    npat++;
                    // every stored particle's two pair terms against its position as it is now, a lane per (particle, sum): lane
                    // 2 k the k-th newest entry's term of the first sum, lane 2 k + 1 of the second (a particle stored twice gives
                    // the same terms twice)
                    const int lq = __shfl(lp, lane >> 1, PMC_WAVE);
                    const int which = lane & 1;
                    double u = 0.0;
                    int kk = 0xffff;
                    if (lq >= 0) {
                        u = sqe_pair<D, KIND>(s, pos, species, models, spi, which ? xn : x, lq);
                        kk = which ? (int)eb.ix2[lq] : (int)eb.ix1[lq];
                        if (kk != 0xffff) (which ? eb.t2 : eb.t1)[kk] = u;            // (a term that has become 0 keeps its place: it adds nothing)
                    }
                    SQVT(2);
                    int c1 = sp.c1, c2 = sp.c2;
                    // the first place either list changes at: the chains start at its block
                    int k1 = (lq >= 0 && !which && kk != 0xffff) ? kk : 0x7fff, k2 = (lq >= 0 && which && kk != 0xffff) ? kk : 0x7fff;
                    k1 = sqe_min32(k1);                                   // (the entries of the log sit in lanes 0 .. 2 NB - 1)
                    k2 = sqe_min32(k2);
                    // particles that have come inside a cutoff: a place in id order, one at a time (rare)
                    unsigned long long ins = __ballot(lq >= 0 && kk == 0xffff && u != 0.0);
                    sq_wave_sync();
                    while (ins) {
                        const int l = (int)__builtin_ctzll(ins);
                        ins &= ins - 1;
                        const int p = __builtin_amdgcn_readlane(lq, l);
                        const double uu = readlane_double(u, l);
                        int where = 0;
                        if (l & 1) {
                            if (eb.ix2[p] == 0xffff) { c2 = sqe_insert(eb.t2, eb.id2, eb.ix2, c2, p, uu, where); k2 = min(k2, where); nins++; }      // (stored twice: the second finds its place taken, with this very term)
                        } else {
                            if (eb.ix1[p] == 0xffff) { c1 = sqe_insert(eb.t1, eb.id1, eb.ix1, c1, p, uu, where); k1 = min(k1, where); nins++; }
                        }
                    }
                    sq_wave_sync();
                    SQVT(3);
                    // (both chains from the block the earlier change lies in: a chain then never runs past the zeros behind the
                    // LONGER list -- started further on than the other, the shorter one would walk off its buffer's end)
                    const int bst = min(min(k1 >> 3, (c1 + 7) / 8), min(k2 >> 3, (c2 + 7) / 8));
                    sqe_chains<false>(eb, c1, c2, bst, bst, e1, e2);
                    SQVT(4);
                }
            }
            double de = 0.0;
            if (!(isinf(e1) || isinf(e2))) { de = e2 - e1; E += de; }            // src/moves.jl:13-18
            const bool ok = ge.lnu < -(e2 - e1) / T;
            bool stored = ok;
            if (ok) {
                if (lane < D) pos[i * D + lane] = sqv_sel3(xn, lane);
            } else {
                E -= de;                                                          // src/moves.jl:79
                bool chg = false;
#pragma unroll
                for (int q = 0; q < D; q++) chg |= sq_bits_differ(xr[q], x[q]);
                if (chg) { if (lane < D) pos[i * D + lane] = sqv_sel3(xr, lane); stored = true; }
            }
            if (stored) {
                if (lane == 0) { logp[2 * (nlog & (SQE_LOG - 1))] = t; logp[2 * (nlog & (SQE_LOG - 1)) + 1] = i; }
                nlog++;
            }
            cntv += (lane == m || (ok && lane == 8 + m)) ? 1u : 0u;
            asm volatile("" ::: "memory");
            if (lane == 0) sqv_st_relaxed(p_done, t + 1);
            SQVT(5);
        }
        if (lane < 16) ctl[4 + lane] = (int)cntv;
    } else if (wv == 4) {
        // ---- the GENERATOR: block B of the random parts takes the place of block B - 2 once the resolver is through with it
        for (int B = 2; (B << 6) < nsteps; B++) {
            while (sqv_ld(p_done) < ((B - 1) << 6)) __builtin_amdgcn_s_sleep(60);
            const int t = (B << 6) + lane;
            if (t < nsteps) generate(t);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) sqv_st(p_gen, B + 1);
        }
    } else if (ev >= 0) {
        // ---- an EVALUATOR: every NB-th step, begun once the resolver is within SQE_LOOK steps of it
        for (int t = ev; t < nsteps; t += NB) {
            int s0 = 0;
            while (true) {
                const sqv_u4 c = *reinterpret_cast<const volatile sqv_u4 *>(ctl);      // -, done, random parts, -
                s0 = (int)c.y;
                if (s0 >= t - SQE_LOOK && (int)c.z > (t >> 6)) break;
                if (t - SQE_LOOK - s0 > 2) __builtin_amdgcn_s_sleep(24); else __builtin_amdgcn_s_sleep(3);
            }
            asm volatile("" ::: "memory");
            const SqGen ge = gen[t & (SQV_GEN - 1)];
            const int i = (int)mulhi32(ge.r1, (uint32_t)N);
            double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0};
#pragma unroll
            for (int q = 0; q < D; q++) {
                x[q] = pos[i * D + q];
                xn[q] = x[q] + ge.d[q];
            }
            const int spi = (S > 1) ? min((int)species[i], S - 1) : 0;
            SqeSpec sp;
            sp.i = i; sp.s0 = s0; sp.seq = 0; sp.pad0 = sp.pad1 = sp.pad2 = 0;
            sqe_evaluate<D, KIND>(s, pos, species, models, blk, i, spi, x, xn, sp.c1, sp.c2, sp.e1, sp.e2);
            if (lane == 0) spec0[ev] = sp;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) sqv_st(&spec0[ev].seq, t + 1);
        }
    }
    __syncthreads();
    // -- back to global memory
    for (int k = tid; k < N * D; k += nth) ch.pos[k] = pos[k];
    if (tid == 0) {
        ch.energy = E;
        ch.step0 += (unsigned long long)nsteps;
        for (int q = 0; q < 8; q++) { ch.att[q] += (unsigned long long)ctl[4 + q]; ch.acc[q] += (unsigned long long)ctl[12 + q]; }
        ch.n_patched += npat; ch.n_again += nagain; ch.n_inserted += nins; ch.n_polls += nwait;
    }
}
