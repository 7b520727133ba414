// pmc_kernels.h -- HIP kernels of the ParticlesMC hot path for gfx950 (MI355X).
//
// Two engines share the pair arithmetic of pmc_device.h:
//  * reference-order engine: device-resident head/list linked cells exactly as
//    src/neighbours.jl:224-302, per-particle energies summed in the LinkedIterator
//    order (src/neighbours.jl:332-377) -> bit-identical to the sequential Julia path;
//  * throughput engine: cell-sorted SoA + CSR over an even grid, one workgroup of 1, 2 or 4
//    wavefronts per active cell, 3^d neighbour cells staged in LDS, exact (order-free) sums,
//    checkerboard colour passes.
#pragma once
#include "pmc_device.h"
#include <type_traits>

#define PMC_CAP_MAX 1024     // upper bound of the per-wavefront LDS staging capacity (CbArgs::cap)
#define PMC_CAP_BIG 4096     // ... of a TEAM's staging when its neighbourhoods outgrow the register copy of the pre-filter (CbArgs::fcap)
#define PMC_SEGTAB_TEAM 64   // entries of a team's segment tables: compact_candidates counts up to 64 chunks of 64 candidates
#define PMC_CFL 8            // centre-cell particles whose cell coordinates are cached (the staging of the teams; stays_in_cell)
#ifndef PMC_FUSED_STAGE
#define PMC_FUSED_STAGE true  // k_cb_pass applies the box filter while staging (stage_compacted)
#endif
#ifndef PMC_CB_MINWAVES
#define PMC_CB_MINWAVES 4     // waves per SIMD k_cb_pass is compiled for (register budget): one wavefront per cell
#endif
// Teams (NW >= 2 wavefronts per cell) are compiled for THREE wavefronts per SIMD: 168 VGPRs, which all but a handful of their
// 160 instances fit without scratch (at 128 VGPRs they keep 12 ... 128 B per lane).  Teams serve small systems (<= 2400 cells
// per colour), where 12 wavefronts per CU are no constraint once the team width respects it (pmc_create); measured against
// the 128-VGPR build at the end of round 3: -1 ... +5 % either way, row by row.  (The "one instance wrong under scratch"
// incidents of rounds 2-3 that first motivated this were a bug of this file, not of the register budget: DESIGN.md 4.6.
// -DPMC_TEAM_MINWAVES=4 builds the teams for 128 VGPRs; the whole instance matrix is green with it.)
#ifndef PMC_TEAM_MINWAVES
#define PMC_TEAM_MINWAVES 3
#endif
// The kernels that are not templates belong to pmc_hip.hip alone; the translation units that only instantiate k_cb_pass
// (pmc_cb_inst.hip, compiled in parallel) see them as unused statics.
#ifdef PMC_INST_ONLY
#define PMC_GLOBAL static __global__
#else
#define PMC_GLOBAL __global__
#endif
// every k_cb_pass<D, KIND, ...> the host dispatches, for one (D, KIND): X(MULTI, NW, ONE, MW)
#define PMC_CB_VARIANTS_NW(X, NW, MW) X(false, NW, false, MW) X(false, NW, true, MW) X(true, NW, false, MW) X(true, NW, true, MW)
#define PMC_CB_VARIANTS(X)                                                                                   \
    PMC_CB_VARIANTS_NW(X, 1, 3) PMC_CB_VARIANTS_NW(X, 1, 4) PMC_CB_VARIANTS_NW(X, 2, PMC_TEAM_MINWAVES)       \
    PMC_CB_VARIANTS_NW(X, 4, PMC_TEAM_MINWAVES) PMC_CB_VARIANTS_NW(X, 8, PMC_TEAM_MINWAVES)
// ... the big-neighbourhood instances (k_cb_pass<..., BIG = true>: teams of eight, no images): X(MULTI, ONE)
#define PMC_CB_VARIANTS_BIG(X) X(false, false) X(false, true) X(true, false) X(true, true)
// ... and the instances for systems in which some particle carries a periodic image (IMG = true; one register budget per team
// width: PMC_IMG_MW): X(MULTI, NW, ONE).  Systems without images -- every run that starts folded -- run the kernels above, whose slot
// loop does not contain a single instruction of the image path (compiled in, it cost them 1 % through register allocation alone).
#define PMC_CB_VARIANTS_IMG_NW(X, NW) X(false, NW, false) X(false, NW, true) X(true, NW, false) X(true, NW, true)
#define PMC_IMG_MW(NW) ((NW) == 1 ? PMC_CB_MINWAVES : PMC_TEAM_MINWAVES)
#define PMC_CB_VARIANTS_IMG(X) PMC_CB_VARIANTS_IMG_NW(X, 1) PMC_CB_VARIANTS_IMG_NW(X, 2) PMC_CB_VARIANTS_IMG_NW(X, 4) PMC_CB_VARIANTS_IMG_NW(X, 8)
#define PMC_REF_IDS 2048     // neighbourhood size (ids) the ordered-energy LDS buffer holds (8 KB: 20 wavefronts per CU)
#define PMC_MAXNN 27
#define PMC_REC 12           // doubles per slot record: old xyz, new xyz, dE, T ln u, LDS slot i, valid, LDS slot j, move
#define PMC_NSHARD 256       // copies of the counter block: a workgroup adds into copy (ticket or block index) mod 256, k_stats_fold sums
                             // them into copy 0 (55 000 workgroups adding to ONE cache line serialise at ~6 ns per atomic: that alone
                             // was 2.3 ms per sweep at N = 10^6)
#define PMC_NSTATS 64        // counters of one sweep call (CbArgs::stats); [32..47] are used by -DPMC_DEBUG_TIME builds only
#define PMC_STAT_KEPT 48     // [48..63]: cell visits by box-filtered neighbourhood size, bucket b = sizes in (64 (b-1), 64 b]
#define PMC_QCAP 320         // entries of a wavefront's survivor queue (2 B each: the candidate's LDS slot); a pair of filter trips
                             // adds at most 128, a typical slot ends with ~70, and a slot that would overflow takes the direct loop
#ifdef PMC_DEBUG_TIME   // profiling builds only: shader-clock stamps of wavefront 0, summed per phase into stats[32..]
#define PMC_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define PMC_TIME(...) __VA_ARGS__     // code (or a struct member) that only the stamped build contains
#else
#define PMC_T(var)
#define PMC_TIME(...)
#endif

#define PMC_FTRIPS 8         // filter trips whose float coordinates a wavefront keeps in registers: 512 candidates

// ------------------------------------------------------------------------------
// binning: cell index -> CSR (cell_start, ids ascending inside each cell)
// ------------------------------------------------------------------------------

// get_cell + cell_index, src/neighbours.jl:178-180, 79-88 (0-based result)
template <int D>
__global__ void k_ref_cell_index(const double *__restrict__ pos, int N, double c0, double c1, double c2,
                                 int n0, int n1, int n2, int *__restrict__ cell_of)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double cell[3] = {c0, c1, c2};
    const int nc[3] = {n0, n1, n2};
    long long c = 0, stride = 1;
#pragma unroll
    for (int a = D - 1; a >= 0; a--) {
        long long mc = (long long)dev_jl_fld(pos[(size_t)i * D + a], cell[a]);
        long long f = mc - dev_fld_i(mc, nc[a]) * nc[a];
        c += f * stride;
        stride *= nc[a];
    }
    cell_of[i] = (int)c;
}

struct GridGeom {
    int n[3];
    double w[3];
    double shift[3];
    double L[3];               // box lengths (folding of the cell-sorted copy)
};

// ------------------------------------------------------------------------------
// Periodic images of the throughput engine (DESIGN.md 2.2 "images").  Positions are never wrapped (src/moves.jl:46-48), so
// an unwrapped input -- a restart from an unwrapped frame, a long run -- has particles many boxes apart, and every pair
// then needs the minimum image.  Every particle therefore carries a PERSISTENT integer image k (per dimension), fixed at
// pmc_upload as floor(x / L) (|k| <= 511, else 0), and the throughput kernels work on the folded coordinate
//     xf = x - k L          (k = 0: xf = x, bit for bit)
// of the cell-sorted copy.  A move is evaluated, and stored, at a position that survives the round trip through the
// unwrapped array exactly:  xf' = ((xf + delta) + k L) - k L,  x' = xf' + k L  (k = 0: xf' = x' = x + delta, the
// reference's update_position!).  k never changes between uploads, so every energy is a pure function of the state
// (x, k): the running energy still equals a recomputation bit for bit.  Packed: 10 bits per dimension, biased by 512.
// ------------------------------------------------------------------------------
#define PMC_IMG_MAX 511
__host__ __device__ __forceinline__ int img_pack(int kx, int ky, int kz)
{
    return ((kx + 512) & 1023) | (((ky + 512) & 1023) << 10) | (((kz + 512) & 1023) << 20);
}
#define PMC_IMG_ZERO (512 | (512 << 10) | (512 << 20))
__host__ __device__ __forceinline__ int img_get(int packed, int q) { return ((packed >> (10 * q)) & 1023) - 512; }
// xf = x - k L
__host__ __device__ __forceinline__ double img_fold(double x, int k, double L) { return k == 0 ? x : x - (double)k * L; }
// x = xf + k L
__host__ __device__ __forceinline__ double img_unfold(double xf, int k, double L) { return k == 0 ? xf : xf + (double)k * L; }
// the folded coordinate a move to `t` = xf + delta is evaluated and stored at
__host__ __device__ __forceinline__ double img_settle(double t, int k, double L)
{
    if (k == 0) return t;
    const double kl = (double)k * L;
    return (t + kl) - kl;
}

// pmc_upload: k = floor(x / L) per dimension (IEEE division); particles farther than PMC_IMG_MAX boxes stay unfolded
template <int D>
__global__ void k_fold_images(const double *__restrict__ pos, int N, double L0, double L1, double L2, int *__restrict__ img,
                              int *__restrict__ any)       // any: set to 1 if some particle has an image
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double L[3] = {L0, L1, L2};
    int k[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) {
        const double f = floor(pos[(size_t)i * D + q] / L[q]);
        k[q] = (fabs(f) <= (double)PMC_IMG_MAX) ? (int)f : 0;
    }
    const int packed = img_pack(k[0], k[1], k[2]);
    img[i] = packed;
    if (__builtin_amdgcn_ballot_w64(packed != PMC_IMG_ZERO) != 0ull && (threadIdx.x & (PMC_WAVE - 1)) == 0) atomicOr(any, 1);
}

template <int D>
__device__ __forceinline__ double mc_coord(double x, double shift, double w)
{
    return floor((x - shift) / w);
}

// throughput-grid binning, pass A: cell index + arrival ticket inside the cell
template <int D>
__device__ __forceinline__ void mc_bin_a_body(const double *__restrict__ pos, const int *__restrict__ img, int N, const GridGeom &g,
                                              int *__restrict__ cell_of, int *__restrict__ count, int *__restrict__ arrival)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    long long lin = 0;
    const int im = img ? img[i] : PMC_IMG_ZERO;     // (img == nullptr: no particle of this system has an image)
#pragma unroll
    for (int a = 0; a < D; a++) {
        long long q = (long long)mc_coord<D>(img_fold(pos[(size_t)i * D + a], img_get(im, a), g.L[a]), g.shift[a], g.w[a]);
        long long r = q % g.n[a];
        if (r < 0) r += g.n[a];
        lin = lin * g.n[a] + r;
    }
    cell_of[i] = (int)lin;
    // one returning atomic per RUN of consecutive lanes that share a cell (ids are usually spatially coherent:
    // 2-3 particles per run for lattice-ordered ids), the run head adds the run length and hands out the tickets
    const int lane = threadIdx.x & (PMC_WAVE - 1);
    const int c = (int)lin;
    const int prev = __shfl_up(c, 1, PMC_WAVE);
    const bool head = (lane == 0) || (prev != c);
    const unsigned long long hm = __ballot(head);
    const unsigned long long am = __ballot(true);
    const unsigned long long le = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int my_head = 63 - __builtin_clzll(hm & le);
    const unsigned long long after = (lane == 63) ? 0ull : (hm >> (lane + 1));
    const int last_active = 63 - __builtin_clzll(am);
    const int run_end = after ? lane + __builtin_ctzll(after) : last_active;       // last lane of my run (heads only use it)
    int base = 0;
    if (head) base = atomicAdd(&count[c], run_end - lane + 1);
    base = __shfl(base, my_head, PMC_WAVE);
    arrival[i] = base + (lane - my_head);
}

#define PMC_TICKET_LANES 8     // ticket counters of a fused sweep (k_cb_pass), one per XCD
#define PMC_TICKET_STRIDE 16   // ... 128 B apart

// (ticket8: the fused sweep's ticket counters start every sweep at 0 -- this is the first kernel of a sweep)
template <int D>
__global__ void k_mc_bin_a(const double *__restrict__ pos, const int *__restrict__ img, int N, GridGeom g, int *__restrict__ cell_of,
                           int *__restrict__ count, int *__restrict__ arrival, unsigned long long *__restrict__ ticket8)
{
    if (ticket8 && blockIdx.x == 0 && threadIdx.x < PMC_TICKET_LANES) ticket8[threadIdx.x * PMC_TICKET_STRIDE] = 0ull;
    mc_bin_a_body<D>(pos, img, N, g, cell_of, count, arrival);
}

// pass B: place ids by ticket (order inside a cell still depends on atomic arrival)
PMC_GLOBAL void k_mc_bin_b(const int *__restrict__ cell_of, const int *__restrict__ arrival, int N,
                           const int *__restrict__ start, int *__restrict__ tmp_ids)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) tmp_ids[start[cell_of[i]] + arrival[i]] = i;
}

// pass C: rank by id inside the cell (deterministic order) and gather the cell-sorted SoA copy
// optional second job of pass C: the random numbers of proposal slot (cell c, t = rank r), see k_mc_slots
struct SlotOut {
    double *tlnu, *dx, *dy, *dz;
    uint32_t *w1, *w2;
    unsigned char *m;
};
struct MovePool {
    int npool;
    int kind[8];
    int sp0[8], sp1[8];       // 0-based species
    double prob[8];
    double sigma[8];
};
struct SlotGen {
    int enabled;
    int rounds;               // k of this pass: every cell has k n_c proposal slots, stored at index k start[c] + t (t < k n_c)
    uint32_t s_lo, s_hi, key0, key1;
    double temperature;
    MovePool pool;
    SlotOut out;
};
template <int D>
__device__ __forceinline__ void generate_slot(const SlotGen &g, int lin, int t, int k);

template <int D>
__device__ __forceinline__ void mc_bin_c_body(const double *__restrict__ pos, const int *__restrict__ species,
                                              const int *__restrict__ cell_of, int N, const int *__restrict__ start,
                                              const int *__restrict__ tmp_ids, int *__restrict__ ids, double *__restrict__ sx,
                                              double *__restrict__ sy, double *__restrict__ sz, unsigned char *__restrict__ ssp,
                                              const SlotGen &gen, const int *__restrict__ img, int *__restrict__ simg,
                                              double L0, double L1, double L2)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int c = cell_of[i];
    int lo = start[c], hi = start[c + 1];
    int r = 0;
    for (int k = lo; k < hi; k++) r += (tmp_ids[k] < i);
    int dst = lo + r;
    ids[dst] = i;
    const int im = img ? img[i] : PMC_IMG_ZERO;
    if (img) simg[dst] = im;
    sx[dst] = img_fold(pos[(size_t)i * D + 0], img_get(im, 0), L0);
    sy[dst] = img_fold(pos[(size_t)i * D + 1], img_get(im, 1), L1);
    if (D == 3) sz[dst] = img_fold(pos[(size_t)i * D + 2], img_get(im, 2), L2);
    ssp[dst] = (unsigned char)species[i];
    if (gen.enabled) {
        const int nc = hi - lo;
        for (int rr = 0; rr < gen.rounds; rr++) generate_slot<D>(gen, c, rr * nc + r, gen.rounds * lo + rr * nc + r);
    }
}

template <int D>
__global__ void k_mc_bin_c(const double *__restrict__ pos, const int *__restrict__ species,
                           const int *__restrict__ cell_of, int N, const int *__restrict__ start,
                           const int *__restrict__ tmp_ids, int *__restrict__ ids, double *__restrict__ sx,
                           double *__restrict__ sy, double *__restrict__ sz, unsigned char *__restrict__ ssp, SlotGen gen,
                           const int *__restrict__ img, int *__restrict__ simg, double L0, double L1, double L2)
{
    mc_bin_c_body<D>(pos, species, cell_of, N, start, tmp_ids, ids, sx, sy, sz, ssp, gen, img, simg, L0, L1, L2);
}

// ------------------------------------------------------------------------------
// Many replicas in ONE launch (pmc_sweep_many): chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) and the
// replicas of a tempering ladder have the same shape (N, box, grid, models, pool) and differ in their arrays, random-stream
// key, temperature and -- per sweep -- grid shift and sweep index.  The shape travels by value, the per-replica arrays in a
// device table indexed by blockIdx.y (binning) or by the ticket (sweep kernel), the per-sweep values in a second table
// that is uploaded once per call for all its sweeps.
// ------------------------------------------------------------------------------
struct ManyRep {
    const double *pos_ro;      // binning reads
    double *pos;               // sweep kernel writes back
    int *species;
    int *cell_of, *count, *arrival, *start, *tmp_ids, *ids;
    double *sx, *sy, *sz;
    unsigned char *ssp;
    const int *img;            // persistent images, particle-id order
    int *simg;                 // ... cell-sorted copy of the current pass
    SlotOut slots;
    uint32_t key0, key1;
    double temperature;
    unsigned long long *stats;
    unsigned int *done;
};
struct ManyStep {
    double shift[3];
    uint32_t s_lo, s_hi;
    unsigned int epoch;
    int rounds;                // k of this pass (SlotGen::rounds)
};

template <int D>
__global__ void k_many_bin_a(int N, GridGeom g, const ManyRep *__restrict__ reps, const ManyStep *__restrict__ steps,
                             unsigned long long *__restrict__ ticket8)
{
    if (ticket8 && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < PMC_TICKET_LANES) ticket8[threadIdx.x * PMC_TICKET_STRIDE] = 0ull;
    const ManyRep &r = reps[blockIdx.y];
    const ManyStep &st = steps[blockIdx.y];
#pragma unroll
    for (int q = 0; q < 3; q++) g.shift[q] = st.shift[q];
    mc_bin_a_body<D>(r.pos_ro, r.img, N, g, r.cell_of, r.count, r.arrival);
}

// exclusive scan of every replica's cell counts, one block per replica (grids of <= PMC_SCAN_SMALL_MAX cells)
__device__ __forceinline__ int block_inclusive_scan_1024(int v, int *wsum);
PMC_GLOBAL __launch_bounds__(1024) void k_many_scan(int ncell, const ManyRep *__restrict__ reps)
{
    __shared__ int wsum[16];
    __shared__ int last;
    int *count = reps[blockIdx.x].count, *start = reps[blockIdx.x].start;
    int carry = 0;
    for (int base = 0; base < ncell; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < ncell) ? count[i] : 0;
        const int incl = block_inclusive_scan_1024(v, wsum);
        if (i < ncell) {
            start[i] = carry + incl - v;
            count[i] = 0;
        }
        if (threadIdx.x == 1023) last = incl;
        __syncthreads();
        carry += last;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[ncell] = carry;
}

PMC_GLOBAL void k_many_bin_b(int N, const ManyRep *__restrict__ reps)
{
    const ManyRep &r = reps[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) r.tmp_ids[r.start[r.cell_of[i]] + r.arrival[i]] = i;
}

template <int D>
__global__ void k_many_bin_c(int N, MovePool pool, const ManyRep *__restrict__ reps, const ManyStep *__restrict__ steps,
                             double L0, double L1, double L2)
{
    const ManyRep &r = reps[blockIdx.y];
    const ManyStep &st = steps[blockIdx.y];
    SlotGen gen;
    gen.enabled = 1;
    gen.rounds = st.rounds;
    gen.s_lo = st.s_lo; gen.s_hi = st.s_hi; gen.key0 = r.key0; gen.key1 = r.key1;
    gen.temperature = r.temperature;
    gen.pool = pool;
    gen.out = r.slots;
    mc_bin_c_body<D>(r.pos_ro, r.species, r.cell_of, N, r.start, r.tmp_ids, r.ids, r.sx, r.sy, r.sz, r.ssp, gen, r.img, r.simg, L0, L1, L2);
}

// three-launch exclusive scan: tile sums, scan of the tile sums, tile scan + base
#define PMC_SCAN_TILE 1024
__device__ __forceinline__ int block_inclusive_scan_1024(int v, int *wsum)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int x = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        int y = __shfl_up(x, s, 64);
        if (lane >= s) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    if (wave == 0) {
        int w = (lane < 16) ? wsum[lane] : 0;
#pragma unroll
        for (int s = 1; s < 16; s <<= 1) {
            int y = __shfl_up(w, s, 64);
            if (lane >= s) w += y;
        }
        if (lane < 16) wsum[lane] = w;
    }
    __syncthreads();
    int base = (wave == 0) ? 0 : wsum[wave - 1];
    __syncthreads();
    return x + base;
}

PMC_GLOBAL __launch_bounds__(1024) void k_scan_tiles(const int *__restrict__ count, int n, int *__restrict__ tile_sum)
{
    __shared__ int wsum[16];
    int i = blockIdx.x * PMC_SCAN_TILE + threadIdx.x;
    int v = (i < n) ? count[i] : 0;
    int incl = block_inclusive_scan_1024(v, wsum);
    if (threadIdx.x == 1023) tile_sum[blockIdx.x] = incl;
}

PMC_GLOBAL __launch_bounds__(1024) void k_scan_tops(int *__restrict__ tile_sum, int ntiles, int *__restrict__ total)
{
    __shared__ int wsum[16];
    int carry = 0;
    for (int base = 0; base < ntiles; base += 1024) {
        int i = base + threadIdx.x;
        int v = (i < ntiles) ? tile_sum[i] : 0;
        int incl = block_inclusive_scan_1024(v, wsum);
        if (i < ntiles) tile_sum[i] = carry + incl - v;   // exclusive
        __shared__ int last;
        if (threadIdx.x == 1023) last = incl;
        __syncthreads();
        carry += last;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry;
}

PMC_GLOBAL __launch_bounds__(1024) void k_scan_apply(int *__restrict__ count, int n, const int *__restrict__ tile_sum,
                                                     const int *__restrict__ total, int *__restrict__ start, int zero)
{
    __shared__ int wsum[16];
    int i = blockIdx.x * PMC_SCAN_TILE + threadIdx.x;
    int v = (i < n) ? count[i] : 0;
    int incl = block_inclusive_scan_1024(v, wsum);
    if (i < n) start[i] = tile_sum[blockIdx.x] + incl - v;
    if (i == n - 1) start[n] = *total;
    if (zero && i < n) count[i] = 0;        // leaves the counters ready for the next binning pass (no memset launch)
}

// the whole exclusive scan in one launch for small grids: one block walks the counts tile by tile (coalesced),
// carrying the running total; larger grids take the three-launch scan (one block would serialise them)
#define PMC_SCAN_SMALL_MAX 8192
PMC_GLOBAL __launch_bounds__(1024) void k_scan_small(int *__restrict__ count, int n, int *__restrict__ start, int zero)
{
    __shared__ int wsum[16];
    __shared__ int last;
    int carry = 0;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = (i < n) ? count[i] : 0;
        const int incl = block_inclusive_scan_1024(v, wsum);
        if (i < n) {
            start[i] = carry + incl - v;
            if (zero) count[i] = 0;
        }
        if (threadIdx.x == 1023) last = incl;
        __syncthreads();
        carry += last;
        __syncthreads();
    }
    if (threadIdx.x == 0) start[n] = carry;
}


// longest-first ordering of the cells of each colour (counting sort by population, descending):
// the tail of a colour pass then consists of its cheapest cells.
template <int D>
__device__ __forceinline__ int cell_colour(int lin, const int n[3])
{
    int r = lin, col = 0;
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        col |= ((r % n[q]) & 1) << q;
        r /= n[q];
    }
    return col;
}

// histogram of cell populations per colour; block-private LDS bins keep the global atomics few
template <int D>
__global__ __launch_bounds__(256) void k_order_hist(const int *__restrict__ start, int ncell, GridGeom g, int *__restrict__ hist)
{
    __shared__ int lh[8 * 256];
    for (int k = threadIdx.x; k < 8 * 256; k += 256) lh[k] = 0;
    __syncthreads();
    int lin = blockIdx.x * blockDim.x + threadIdx.x;
    if (lin < ncell) {
        int cnt = min(start[lin + 1] - start[lin], 255);
        atomicAdd(&lh[cell_colour<D>(lin, g.n) * 256 + (255 - cnt)], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 8 * 256; k += 256)
        if (lh[k]) atomicAdd(&hist[k], lh[k]);
}

PMC_GLOBAL void k_order_scan(int *__restrict__ hist, int ncol)   // one wave per colour, 256 buckets
{
    int col = blockIdx.x;
    if (col >= ncol) return;
    int lane = threadIdx.x;
    int carry = 0;
    for (int b0 = 0; b0 < 256; b0 += 64) {
        int v = hist[col * 256 + b0 + lane];
        int x = v;
#pragma unroll
        for (int s = 1; s < 64; s <<= 1) {
            int y = __shfl_up(x, s, 64);
            if (lane >= s) x += y;
        }
        hist[col * 256 + b0 + lane] = carry + x - v;
        carry += __shfl(x, 63, 64);
    }
}

// slot = (running offset of the bin, reserved once per block) + (rank inside the block)
template <int D>
__global__ __launch_bounds__(256) void k_order_fill(const int *__restrict__ start, int ncell, GridGeom g, int *__restrict__ hist,
                                                    int nactive, int *__restrict__ order)
{
    __shared__ int lh[8 * 256];
    for (int k = threadIdx.x; k < 8 * 256; k += 256) lh[k] = 0;
    __syncthreads();
    int lin = blockIdx.x * blockDim.x + threadIdx.x;
    int bin = -1, rank = 0, col = 0;
    if (lin < ncell) {
        int cnt = min(start[lin + 1] - start[lin], 255);
        col = cell_colour<D>(lin, g.n);
        bin = col * 256 + (255 - cnt);
        rank = atomicAdd(&lh[bin], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 8 * 256; k += 256)
        if (lh[k]) lh[k] = atomicAdd(&hist[k], lh[k]);     // count -> base of this block's share
    __syncthreads();
    if (bin >= 0) order[col * nactive + lh[bin] + rank] = lin;
}

PMC_GLOBAL void k_count(const int *__restrict__ cell_of, int N, int *__restrict__ count)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) atomicAdd(&count[cell_of[i]], 1);
}

PMC_GLOBAL void k_scatter(const int *__restrict__ cell_of, int N, const int *__restrict__ start,
                          int *__restrict__ fill, int *__restrict__ tmp_ids)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int c = cell_of[i];
    int slot = start[c] + atomicAdd(&fill[c], 1);
    tmp_ids[slot] = i;
}

// rank of each particle inside its cell by id -> deterministic order whatever
// order the atomics of k_scatter ran in
PMC_GLOBAL void k_rank(const int *__restrict__ cell_of, int N, const int *__restrict__ start,
                       const int *__restrict__ tmp_ids, int *__restrict__ ids)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    int c = cell_of[i];
    int lo = start[c], hi = start[c + 1];
    int r = 0;
    for (int k = lo; k < hi; k++) r += (tmp_ids[k] < i);
    ids[lo + r] = i;
}

#include "pmc_ref.h"     // the reference-order engine: linked cells, ordered energies, per-move verbs, EnergyBias

// ------------------------------------------------------------------------------
// throughput engine
// ------------------------------------------------------------------------------

// The random numbers of every proposal slot of one sweep: slot t of cell c is the
// cell-sorted index k = start[c] + t, so one thread per particle (pass C of the binning, which knows its
// cell and rank) generates all N slots with every lane busy; inside k_cb_pass the same arithmetic ran on
// ~18 of 64 lanes, once per wavefront of the team.
template <int D>
__device__ __forceinline__ void generate_slot(const SlotGen &g, int lin, int t, int k)
{
    uint32_t R[4], Q[4];
    philox4x32((uint32_t)lin, (uint32_t)t, g.s_lo, g.s_hi | ((uint32_t)TAG_SLOT0 << 24), g.key0, g.key1, R);
    philox4x32((uint32_t)lin, (uint32_t)t, g.s_lo, g.s_hi | ((uint32_t)TAG_SLOT1 << 24), g.key0, g.key1, Q);
    const int m = (g.pool.npool == 1) ? 0 : select_move(g.pool, R[0]);
    double z[4];
    box_muller(Q[0], Q[1], &z[0], &z[1]);
    box_muller(Q[2], Q[3], &z[2], &z[3]);
    // per-move sigma without a dynamically indexed kernel argument (select chain)
    double sg = g.pool.sigma[0];
#pragma unroll
    for (int q = 1; q < 8; q++) sg = (m == q) ? g.pool.sigma[q] : sg;
    g.out.tlnu[k] = g.temperature * det_log(uniform53(R[2], R[3]));
    g.out.dx[k] = sg * z[0];
    g.out.dy[k] = sg * z[1];
    if (D == 3) g.out.dz[k] = sg * z[2];
    g.out.w1[k] = R[1];
    g.out.w2[k] = Q[0];
    g.out.m[k] = (unsigned char)m;
}

// the same as its own launch, one thread per cell-sorted slot: coalesced stores, which beats the scattered
// stores of the fused form once the slot arrays outgrow the caches (N >~ 5e5)
template <int D>
__global__ __launch_bounds__(256) void k_mc_slots(int N, const int *__restrict__ ids, const int *__restrict__ cell_of,
                                                  const int *__restrict__ start, SlotGen g)
{
    // (grid: rounds * N threads, round slowest: the threads of one cell and round store to consecutive slots)
    const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= (long long)N * g.rounds) return;
    const int rr = (int)(j / N), k = (int)(j - (long long)rr * N);
    const int lin = cell_of[ids[k]];
    const int lo = start[lin], nc = start[lin + 1] - lo;
    const int t = rr * nc + (k - lo);
    generate_slot<D>(g, lin, t, g.rounds * lo + t);
}

struct CbArgs {
    int n[3];
    int half[3];
    double w[3], shift[3], L[3], invL[3];
    double invw[3];            // 1 / w, for estimates only (cell_coord_fast): the reference arithmetic divides by w
    double temperature;
    const int *start;
    const int *ids;
    double *sx, *sy, *sz;
    unsigned char *ssp;
    double *pos;               // AoS, particle-id order (kept current)
    const int *img;            // persistent packed images, particle-id order (img_pack); nullptr: no particle has one
    const int *simg;           // ... of the cell-sorted slots of this pass (nullptr likewise)
    int *species;
    int N, S;
    uint32_t key0, key1, s_lo, s_hi;
    int rounds;                // k of this pass: a cell visit runs k n_c proposal slots (slot t of cell c: index k start[c] + t)
    int colour;
    int nactive;               // cells handled by this launch
    int all_cells;             // 1: every cell (total energy), 0: one colour
    MovePool pool;
    DevModel m0;
    const DevModel *models;
    int cap;                   // candidates staged in LDS per wavefront (multiple of 64)
    int fcap;                  // teams, big neighbourhoods: candidates whose FLOAT copy lives in LDS (0: none; the register copy holds 512)
    int qcap;                  // entries of a wavefront's survivor queue (PMC_QCAP unless fcap > 0)
    double rc2pad;             // (largest cutoff)^2 * (1 + 1e-9): box-distance filter of the staging
    float rc2f;                // (largest cutoff)^2 * (1 + 2^-12), rounded up: threshold of the FP32 pre-filter (conservative)
    int filter;                // 1: drop candidates that are out of range of the whole centre cell
    int fuse_teams;            // 1: teams of NW >= 2 wavefronts take the fused staging too (stage_compacted with a count exchange)
    const int *order;          // per colour: cells sorted by population, descending (may be null)
    unsigned long long *stats; // [0..7] attempted, [8..15] accepted, [16] energy delta (fix), [17] error flags, [18] nonfinite pairs
    double *e_particle;        // optional per-particle energies (id order)
    // proposal slots of this sweep, generated lane-dense by k_mc_slots (index = cell-sorted slot start[cell] + t)
    const double *slot_tlnu, *slot_dx, *slot_dy, *slot_dz;
    const uint32_t *slot_w1, *slot_w2;
    const unsigned char *slot_m;
    // fused sweep: all 2^d colours in ONE launch, ordered by per-cell completion flags
    int fused;                 // 1: grid = 2^d * nactive workgroups, colour taken from the ticket
    unsigned int epoch;        // value a cell's flag takes when its visit of this sweep is complete
    unsigned int *done;        // [ncell] completion flags (agent-scope stores / polls)
    // Workgroup tickets.  ticket_lanes == 8 (default): eight counters, 128 B apart, zeroed by the first kernel of every sweep;
    // workgroup b draws from counter b mod 8 -- its XCD's, workgroups being dealt round-robin to the XCDs -- and ticket t of
    // counter x is visit 8 t + x.  One counter took 11 ns per atomic from the whole chip: 0.62 ms for the 54 872 visits of a
    // sweep at N = 10^6, the floor the kernel was within 30 % of.  ticket_lanes == 1 (PMC_TICKETS=1): that single counter,
    // monotonic across launches, this launch owning [ticket_base, ticket_base + grid).
    unsigned long long *ticket;
    unsigned long long ticket_base;
    int ticket_lanes;
};

// Agent-scope accesses of the data that workgroups of one fused launch hand to each other (the
// cell-sorted coordinates and species): sc1 loads bypass the CU's L1, sc1 stores write through
// (cdna_hip_programming.md Guideline 16, R1).  Outside a fused launch they are merely uncached in L1.
__device__ __forceinline__ double ld_agent(const double *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned char ld_agent(const unsigned char *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(double *p, double v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned char *p, unsigned char v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// bijective remap so that blocks sharing an XCD (b % 8) work on one contiguous slab of cells
// (used by k_total_energy and when the longest-first order is switched off)
__device__ __forceinline__ int xcd_remap(int b, int nb)
{
    int q = nb >> 3, r = nb & 7;
    int xcd = b & 7, loc = b >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

struct WaveNbh {
    double *cx, *cy, *cz;
    double *cfl;   // [D][PMC_CFL]: floor((x - shift) / w) of the first centre-cell particles (have_cfl; not after stage_compacted)
    bool have_cfl;
    double fc[3];  // floor((x - shift) / w) of the first centre-cell particle
    int cap;
    unsigned char *csp;
    int *segoff;   // nseg + 1 prefix offsets of the staged segments
    int *segsrc;   // CSR offset of every segment
    int nseg;
    int ncand;
    int nc;        // centre-cell population: candidates [ctr, ctr + nc)
    int ctr;       // LDS slot of the first centre-cell particle
    int k0;        // CSR offset of the centre cell
    int lin;
    bool nowrap;   // every staged candidate is in the same periodic image as the centre cell
    double *part;  // [8] per-wavefront partial sums (workgroups of several wavefronts)
    // FP32 pre-filter (wave_delta_e_filtered)
    unsigned short *queue;   // [max(NW,1)][PMC_QCAP] survivor queues, one per wavefront
    double bc[3];            // centre of the centre cell (frame of its first particle)
    bool filt;               // the pre-filter is usable for this visit (everything staged, at most 64 PMC_FTRIPS candidates)
    bool fstream;            // ... from the float copy in LDS (fx, fy, fz): a team's neighbourhood beyond the register copy
    float *fx, *fy, *fz;     // [fcap] float images (rel_float) of the staged candidates, kept current by the team's resolution
    int fcap, qcap;
    bool compacted;          // the box filter was applied while staging (stage_compacted)
    int kept;                // candidates that passed the box filter (stage_compacted; a huge value if it could not count them)
    int loaded;              // population of the 3^d cells the staging read from global memory
    int myimg;               // packed image of centre-cell particle `lane` (PMC_IMG_ZERO behind the last one)
    // (stamped builds: stamps inside stage_neighbourhood -- entry, table ready, flags seen, coordinates arrived, done; [5..8] inside stage_compacted)
    PMC_TIME(unsigned long long ts[10];)
    // proposal-slot values of the first chunk, fetched together with the coordinates (latency-bound teams only)
    double p_tlnu, p_dx, p_dy, p_dz;
    uint32_t p_w1, p_w2;
    int p_m;
    double *rec;   // [NW][PMC_REC] records of the speculatively evaluated slots of one batch
    double *cmat;  // [NW][NW] pair corrections c(i,j) of one batch
    int *flag;     // [4] workgroup flags
    int *cnt;      // [16] per-move attempted / accepted counters of this cell visit
    int *ptab;     // [24] the pool's kind / species pair per move (kind[8], sp0[8], sp1[8]): indexed with a run-time move
                   // number, which on a by-value kernel argument makes the compiler copy all of CbArgs to scratch
};

// thread index / stride of the NW wavefronts that share one cell
template <int NW> __device__ __forceinline__ int team_tid() { return NW == 1 ? wave_lane() : (int)threadIdx.x; }
template <int NW> __device__ __forceinline__ int team_wave() { return NW == 1 ? 0 : (int)(threadIdx.x >> 6); }

__device__ __forceinline__ int nbh_find(const WaveNbh &h, int k)
{
    int lo = 0, hi = h.nseg;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (h.segoff[mid] <= k) lo = mid; else hi = mid;
    }
    return h.segsrc[lo] + (k - h.segoff[lo]);
}

// floor((x - shift) / w) for a wave-uniform x: the product with 1/w decides unless it lands within 2^-30 of an integer (its
// error is below 2^-45 for any grid this engine accepts), where the division of the reference arithmetic (mc_coord) does.
template <int D>
__device__ __forceinline__ double cell_coord_fast(double x, double shift, double w, double invw)
{
    const double q = (x - shift) * invw, f = floor(q), r = q - f;
    const bool unsure = !(r > 0x1p-30 && r < 1.0 - 0x1p-30);
    if (__builtin_amdgcn_ballot_w64(unsure) != 0ull) return mc_coord<D>(x, shift, w);
    return f;
}

// Fused staging: the segments of the neighbourhood (runs of 3 cells, or single cells) go from global memory to registers,
// are tested against the centre cell's box there (the test of compact_candidates below, same arithmetic), and only the
// survivors are written to LDS, already compacted.  The staging capacity then only has to hold the ~73 % of a 3^d-cell
// neighbourhood that can be in range of the centre cell, and the LDS round trip of a separate compaction pass is gone.
// lanes 0..nseg-1 hold (seg_start, seg_cnt).  Returns false -- with nothing but LDS scratch modified -- when a segment is
// longer than a wavefront or the survivors do not fit: the caller then stages the old way.
template <int D, bool MULTI, int NW>
__device__ __forceinline__ bool stage_compacted(const CbArgs &a, WaveNbh &h, int seg_start, int seg_cnt, int nseg)
{
    constexpr int NR = (D == 3) ? 9 : 3;
    const int lane = wave_lane(), wv = team_wave<NW>();
    h.kept = 1 << 20;
    if (__builtin_amdgcn_ballot_w64(lane < nseg && seg_cnt > PMC_WAVE) != 0ull) return false;
    const int k0 = h.k0, k1 = h.k0 + h.nc;
    // reference particle (the first of the centre cell): its cell coordinates define the box, its image the no-wrap test
    const double xr = ld_agent(a.sx + k0), yr = ld_agent(a.sy + k0), zr = (D == 3) ? ld_agent(a.sz + k0) : 0.0;
    double bc[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
    bool have_box = false;
    int base = h.nc;           // the centre cell first, the compacted neighbours behind it
    bool wrapped_any = false;
    for (int b = 0; b < nseg; b += NR) {
        double tx[NR], ty[NR], tz[NR];
        unsigned char ts[NR];
        unsigned long long keepm[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int s0 = __builtin_amdgcn_readlane(seg_start, b + r), c0 = __builtin_amdgcn_readlane(seg_cnt, b + r);
            tx[r] = ty[r] = tz[r] = 0.0;
            ts[r] = 0;
            if ((r % NW) == wv && lane < c0) {
                tx[r] = ld_agent(a.sx + s0 + lane);
                ty[r] = ld_agent(a.sy + s0 + lane);
                if (D == 3) tz[r] = ld_agent(a.sz + s0 + lane);
                if (MULTI) ts[r] = ld_agent(a.ssp + s0 + lane);
            }
        }
    PMC_TIME(if (b == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); h.ts[3] = __builtin_amdgcn_s_memtime(); })
        if (!have_box) {
            const double ref[3] = {xr, yr, zr};
#pragma unroll
            for (int q = 0; q < D; q++) {
                h.fc[q] = cell_coord_fast<D>(ref[q], a.shift[q], a.w[q], a.invw[q]);
                bc[q] = a.shift[q] + (h.fc[q] + 0.5) * a.w[q];
                bh[q] = 0.5 * a.w[q] * (1.0 + 1e-9);
            }
            have_box = true;
        }
    PMC_TIME(if (b == 0) h.ts[5] = __builtin_amdgcn_s_memtime();)
        // which candidates are in another periodic image than the reference particle (see the no-wrap test of
        // stage_neighbourhood); one that is not lies in the unfolded 3-cell span, where the minimum image below is the identity
        bool wr[NR];
        bool wrapped = false;
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int c0 = __builtin_amdgcn_readlane(seg_cnt, b + r);
            wr[r] = false;
            if ((r % NW) == wv && lane < c0) {
                wr[r] = rint((xr - tx[r]) * a.invL[0]) != 0.0 || rint((yr - ty[r]) * a.invL[1]) != 0.0 ||
                        ((D == 3) ? (rint((zr - tz[r]) * a.invL[2]) != 0.0) : false);
            }
            wrapped = wrapped || wr[r];
        }
        const bool fold = __builtin_amdgcn_ballot_w64(wrapped) != 0ull;
        wrapped_any = wrapped_any || fold;
    PMC_TIME(if (b == 0) h.ts[6] = __builtin_amdgcn_s_memtime();)
        int cnt[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int s0 = __builtin_amdgcn_readlane(seg_start, b + r), c0 = __builtin_amdgcn_readlane(seg_cnt, b + r);
            bool keep = false;
            if ((r % NW) == wv && lane < c0) {
                double t0 = tx[r] - bc[0], t1 = ty[r] - bc[1], t2 = (D == 3) ? tz[r] - bc[2] : 0.0;
                if (fold) {
                    t0 -= rint(t0 * a.invL[0]) * a.L[0];
                    t1 -= rint(t1 * a.invL[1]) * a.L[1];
                    if (D == 3) t2 -= rint(t2 * a.invL[2]) * a.L[2];
                }
                const double ex = fmax(fabs(t0) - bh[0], 0.0);
                const double ey = fmax(fabs(t1) - bh[1], 0.0);
                double d2 = ex * ex + ey * ey;
                if (D == 3) {
                    const double ez = fmax(fabs(t2) - bh[2], 0.0);
                    d2 += ez * ez;
                }
                const int g = s0 + lane;
                // the centre cell's own particles go first (slots 0 .. nc-1, below): they are not part of the compaction
                keep = !(g >= k0 && g < k1) && (!a.filter || !(d2 > a.rc2pad));
            }
            keepm[r] = __builtin_amdgcn_ballot_w64(keep);
            cnt[r] = __popcll(keepm[r]);
        }
    PMC_TIME(if (b == 0) h.ts[7] = __builtin_amdgcn_s_memtime();)
        if (NW > 1) {
            // every run belongs to one wavefront of the team: exchange the kept counts
#pragma unroll
            for (int r = 0; r < NR; r++)
                if ((r % NW) == wv && lane == 0) h.segoff[r] = cnt[r];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < NR; r++) cnt[r] = h.segoff[r];
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int s0 = __builtin_amdgcn_readlane(seg_start, b + r);
            const unsigned long long m = keepm[r];
            if ((r % NW) == wv) {
                const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int c0 = __builtin_amdgcn_readlane(seg_cnt, b + r);
                const int g = s0 + lane;
                const bool centre = lane < c0 && g >= k0 && g < k1;
                const int pos = centre ? g - k0 : base + below;
                if ((centre || ((m >> lane) & 1ull)) && pos < h.cap) {
                    h.cx[pos] = tx[r];
                    h.cy[pos] = ty[r];
                    if (D == 3) h.cz[pos] = tz[r];
                    if (MULTI) h.csp[pos] = ts[r];
                }
            }
            base += cnt[r];
        }
    }
    PMC_TIME(h.ts[8] = __builtin_amdgcn_s_memtime();)
    if (NW > 1 && wrapped_any && lane == 0) atomicOr(&h.flag[0], 1);
    __syncthreads();
    h.kept = base;
    if (base > h.cap) return false;
    h.nowrap = ((NW > 1) ? (h.flag[0] == 0) : !wrapped_any) && a.n[0] != 1;     // (single-cell box: pairs are up to L apart, see below)
    h.ctr = 0;
    h.ncand = base;
    h.nseg = 1;
    if (lane == 0 && wv == 0) { h.segoff[0] = 0; h.segoff[1] = base; h.segsrc[0] = 0; }
    h.compacted = true;
    return true;
}

// Stage the 3^D neighbour cells of `cell` into LDS.
//  * Fast path (the cell's last-dimension neighbours do not wrap): the 3 cells of one column are
//    contiguous in the cell-sorted arrays, so the neighbourhood is 3^(D-1) runs; every run is
//    loaded with one coalesced load per array and all loads are in flight together.
//  * Generic path: 3^D segments, centre first, located per candidate by binary search.
// wait_colour > 0 (fused sweep): before the first coordinate is loaded the workgroup waits for the completion flags
// of its lower-colour neighbour cells; the first poll is issued at the very top so that it
// travels together with the loads of the CSR offsets instead of after them.
template <int D, bool MULTI, int NW = 1, bool FUSE = false>
__device__ __forceinline__ void stage_neighbourhood(const CbArgs &a, const int cell[3], WaveNbh &h, int wait_colour = 0)
{
    h.compacted = false;
    h.kept = 1 << 20;
    PMC_TIME(h.ts[0] = h.ts[1] = h.ts[2] = h.ts[3] = h.ts[4] = __builtin_amdgcn_s_memtime();)
    constexpr int NN = (D == 3) ? 27 : 9;
    constexpr int NR = (D == 3) ? 9 : 3;
    bool poll_need = false;
    int poll_lin = 0;
    unsigned poll_v = 0;
    if (wait_colour > 0 && team_wave<NW>() == 0) {
        const int pl = wave_lane();
        if (pl < NN) {
            int r = pl, mask = 0;
#pragma unroll
            for (int q = 0; q < D; q++) {
                const int o = (r % 3) - 1;
                r /= 3;
                int cq = cell[q] + o;
                if (cq < 0) cq += a.n[q];
                if (cq >= a.n[q]) cq -= a.n[q];
                poll_lin = poll_lin * a.n[q] + cq;
                if (o != 0) mask |= 1 << q;
            }
            poll_need = (mask != 0) && ((wait_colour ^ mask) < wait_colour);
        }
        poll_v = a.epoch;
        if (poll_need) poll_v = __hip_atomic_load(a.done + poll_lin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    constexpr int LD = D - 1;
    constexpr int TS = NW * PMC_WAVE;             // team stride
    const int lane = wave_lane();
    const int tid = team_tid<NW>(), wv = team_wave<NW>();
    if (NW > 1) {
        if (tid == 0) h.flag[0] = 0;
        __syncthreads();
    }
    const bool runs_shape = (cell[LD] >= 1) && (cell[LD] + 1 < a.n[LD]);
    bool runs_ok = runs_shape;
    int seg_start = 0, seg_cnt = 0;
    int lin_c = 0;
#pragma unroll
    for (int q = 0; q < D; q++) lin_c = lin_c * a.n[q] + cell[q];
    h.k0 = a.start[lin_c];
    h.nc = a.start[lin_c + 1] - h.k0;
    // (requested here so that it travels with the coordinate loads; only visits of particles outside the primary box use it)
    h.myimg = (a.simg && lane < h.nc) ? a.simg[h.k0 + lane] : PMC_IMG_ZERO;     // (a.simg == nullptr: a system without images)
    int run_start = 0, run_cnt = 0;
    bool runs_long = false;    // a run of 3 cells longer than a wavefront: stage_compacted takes the 3^D single cells instead
    if (runs_ok) {
        // the run layout keeps the centre cell mid-buffer: only use it when everything fits in LDS
        int tot = 0;
        if (lane < NR) {
            int lin2 = 0, r = lane;
#pragma unroll
            for (int q = 0; q < LD; q++) {
                int cq = cell[q] + (r % 3) - 1;
                r /= 3;
                if (cq < 0) cq += a.n[q];
                if (cq >= a.n[q]) cq -= a.n[q];
                lin2 = lin2 * a.n[q] + cq;
            }
            lin2 = lin2 * a.n[LD] + (cell[LD] - 1);
            seg_start = a.start[lin2];
            seg_cnt = a.start[lin2 + 3] - seg_start;
            tot = seg_cnt;
        }
#pragma unroll
        for (int sft = 1; sft < 16; sft <<= 1) tot += __shfl_xor(tot, sft, PMC_WAVE);
        tot = __shfl(tot, 0, PMC_WAVE);
        runs_long = FUSE && (NW == 1 || a.fuse_teams != 0) && __builtin_amdgcn_ballot_w64(lane < NR && seg_cnt > PMC_WAVE) != 0ull;
        runs_ok = tot <= h.cap && !runs_long;
        run_start = seg_start;
        run_cnt = seg_cnt;
        if (!runs_ok) { seg_start = 0; seg_cnt = 0; }
    }
    if (runs_ok) {
        h.nseg = NR;
    } else {
        h.nseg = NN;
        if (lane < NN) {
            // lane 0 <-> centre (offset index (NN-1)/2), that lane takes offset 0
            int o = (lane == 0) ? (NN - 1) / 2 : ((lane == (NN - 1) / 2) ? 0 : lane);
            int lin2 = 0, r = o;
            bool dup = false;          // single-cell boxes (n = 1): the cell is its own neighbour once, not 3^d times
#pragma unroll
            for (int q = 0; q < D; q++) {
                const int off = (r % 3) - 1;
                int cq = cell[q] + off;
                r /= 3;
                dup = dup || (a.n[q] == 1 && off != 0);
                if (cq < 0) cq += a.n[q];
                if (cq >= a.n[q]) cq -= a.n[q];
                lin2 = lin2 * a.n[q] + cq;
            }
            seg_start = a.start[lin2];
            seg_cnt = dup ? 0 : a.start[lin2 + 1] - seg_start;
        }
    }
    int incl = seg_cnt;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) {
        int v = __shfl_up(incl, s, PMC_WAVE);
        if (lane >= s) incl += v;
    }
    const int seg_off = incl - seg_cnt;
    h.ncand = __shfl(incl, h.nseg - 1, PMC_WAVE);
    h.loaded = h.ncand;
    // (every wavefront of a team writes the same table.  Not before stage_compacted, which exchanges counts through segoff:
    // a slow wavefront's late write of the table would overwrite a fast one's count)
    const bool try_fused = FUSE && h.nc > 0 && (NW == 1 || a.fuse_teams != 0);
    if (!try_fused) {
        if (lane < h.nseg) {
            h.segoff[lane] = seg_off;
            h.segsrc[lane] = seg_start;
        }
        if (lane == 0) h.segoff[h.nseg] = h.ncand;
    }
    const int lim = min(h.ncand, h.cap);
    PMC_TIME(h.ts[1] = __builtin_amdgcn_s_memtime();)
    if (wait_colour > 0) {
        // a workgroup only ever waits for workgroups that already hold a ticket (see k_cb_pass): no deadlock; a bounded
        // spin turns a lost flag into an error code (stats[17] bit 1) instead of a hang
        if (team_wave<NW>() == 0) {
            unsigned spins = 0;
            while (__ballot(poll_v != a.epoch) != 0ull) {
                if (++spins > (1u << 22)) {
                    if (lane == 0) atomicOr(&a.stats[17], 2ull);
                    break;
                }
                __builtin_amdgcn_s_sleep(8);
                poll_v = a.epoch;
                if (poll_need) poll_v = __hip_atomic_load(a.done + poll_lin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
    }
    PMC_TIME(h.ts[2] = h.ts[3] = __builtin_amdgcn_s_memtime();)
    if (NW >= 4) {
        // (defined on every path: cb_visit reads them whenever NW >= 4)
        h.p_tlnu = h.p_dx = h.p_dy = h.p_dz = 0.0;
        h.p_w1 = h.p_w2 = 0u;
        h.p_m = 0;
    }
    if (NW >= 4 && a.slot_tlnu && !a.all_cells) {
        // small systems: a cell visit is one dependent chain, so the slot values ride along with the coordinate loads
        const int t = lane;
        if (t < h.nc * a.rounds) {
            const int ks = a.rounds * h.k0 + t;
            h.p_m = (int)a.slot_m[ks];
            h.p_tlnu = a.slot_tlnu[ks];
            h.p_w1 = a.slot_w1[ks];
            h.p_w2 = a.slot_w2[ks];
            h.p_dx = a.slot_dx[ks];
            h.p_dy = a.slot_dy[ks];
            if (D == 3) h.p_dz = a.slot_dz[ks];
        }
    }
    if (try_fused) {
        const bool ok = (runs_shape && !runs_long) ? stage_compacted<D, MULTI, NW>(a, h, run_start, run_cnt, NR)
                                                   : stage_compacted<D, MULTI, NW>(a, h, seg_start, seg_cnt, NN);
        if (ok) {
            h.have_cfl = false;        // (h.fc is set; the rare exact 'stays in its cell' test divides for itself)
            __syncthreads();           // the segment table of the compacted layout is visible
    PMC_TIME(h.ts[4] = __builtin_amdgcn_s_memtime();)
            return;
        }
        // (rare) a segment longer than a wavefront, or more survivors than the capacity: the old way, overflow loops and all
        h.nseg = runs_ok ? NR : NN;
        if (lane == 0) h.segoff[h.nseg] = h.ncand;
        if (lane < h.nseg) {
            h.segoff[lane] = seg_off;
            h.segsrc[lane] = seg_start;
        }
        __syncthreads();
    }
    if (runs_ok) {
        // centre cell sits inside the middle run
        h.ctr = __shfl(seg_off, (NR - 1) / 2, PMC_WAVE) + (h.k0 - __shfl(seg_start, (NR - 1) / 2, PMC_WAVE));
        double tx[NR], ty[NR], tz[NR];
        unsigned char ts[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int s0 = __builtin_amdgcn_readlane(seg_start, r), c0 = __builtin_amdgcn_readlane(seg_cnt, r);
            tx[r] = ty[r] = tz[r] = 0.0;
            ts[r] = 0;
            if ((r % NW) == wv && lane < c0) {
                tx[r] = ld_agent(a.sx + s0 + lane);
                ty[r] = ld_agent(a.sy + s0 + lane);
                if (D == 3) tz[r] = ld_agent(a.sz + s0 + lane);
                if (MULTI) ts[r] = ld_agent(a.ssp + s0 + lane);
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            const int c0 = __builtin_amdgcn_readlane(seg_cnt, r), o0 = __builtin_amdgcn_readlane(seg_off, r);
            if ((r % NW) == wv && lane < c0 && o0 + lane < lim) {
                h.cx[o0 + lane] = tx[r];
                h.cy[o0 + lane] = ty[r];
                if (D == 3) h.cz[o0 + lane] = tz[r];
                if (MULTI) h.csp[o0 + lane] = ts[r];
            }
        }
        // runs longer than one wavefront (rare)
        for (int r = 0; r < NR; r++) {
            const int s0 = __builtin_amdgcn_readlane(seg_start, r), c0 = __builtin_amdgcn_readlane(seg_cnt, r);
            const int o0 = __builtin_amdgcn_readlane(seg_off, r);
            for (int l = PMC_WAVE + lane; l < c0; l += PMC_WAVE) {
                if ((r % NW) == wv && o0 + l < lim) {
                    h.cx[o0 + l] = ld_agent(a.sx + s0 + l);
                    h.cy[o0 + l] = ld_agent(a.sy + s0 + l);
                    if (D == 3) h.cz[o0 + l] = ld_agent(a.sz + s0 + l);
                    if (MULTI) h.csp[o0 + l] = ld_agent(a.ssp + s0 + l);
                }
            }
        }
        __syncthreads();
    } else {
        h.ctr = 0;
        __syncthreads();
        for (int k = tid; k < lim; k += TS) {
            int src = nbh_find(h, k);
            h.cx[k] = ld_agent(a.sx + src);
            h.cy[k] = ld_agent(a.sy + src);
            if (D == 3) h.cz[k] = ld_agent(a.sz + src);
            if (MULTI) h.csp[k] = ld_agent(a.ssp + src);
        }
        __syncthreads();
    }
    // No-wrap test: if rint((x_ref - x_j) / L) == 0 in every dimension for every candidate (x_ref = a
    // centre particle), all candidates lie in the unfolded 3-cell span around the centre cell, every
    // centre particle lies in one unfolded cell, and |x_p - x_j| < 2 w <= L/2 for every pair: the
    // minimum image is the identity, dx - rint(dx/L)*L == dx bit for bit.
    {
        bool wrapped = (h.ncand > h.cap) || (h.nc == 0) || (h.ctr + h.nc > h.cap);
        if (!wrapped) {
            const double xr = h.cx[h.ctr], yr = h.cy[h.ctr], zr = (D == 3) ? h.cz[h.ctr] : 0.0;
            for (int k = tid; k < lim; k += TS) {
                bool w0 = rint((xr - h.cx[k]) * a.invL[0]) != 0.0;
                bool w1 = rint((yr - h.cy[k]) * a.invL[1]) != 0.0;
                bool w2 = (D == 3) ? (rint((zr - h.cz[k]) * a.invL[2]) != 0.0) : false;
                wrapped = wrapped || w0 || w1 || w2;
            }
        }
        h.nowrap = (__ballot(wrapped) == 0ull);
        if (NW > 1) {
            if (!h.nowrap && lane == 0) atomicOr(&h.flag[0], 1);
            __syncthreads();
            h.nowrap = (h.flag[0] == 0);
        }
        // (a single-cell box: the candidates are the whole periodic box, two of them can be a box length apart however close
        // each is to the reference particle: the minimum image stays in every pair distance)
        if (a.n[0] == 1) h.nowrap = false;
    }
    if (tid < min(h.nc, PMC_CFL) && h.ctr + tid < h.cap) {
        h.cfl[0 * PMC_CFL + tid] = mc_coord<D>(h.cx[h.ctr + tid], a.shift[0], a.w[0]);
        h.cfl[1 * PMC_CFL + tid] = mc_coord<D>(h.cy[h.ctr + tid], a.shift[1], a.w[1]);
        if (D == 3) h.cfl[2 * PMC_CFL + tid] = mc_coord<D>(h.cz[h.ctr + tid], a.shift[2], a.w[2]);
    }
    __syncthreads();
    h.have_cfl = true;
#pragma unroll
    for (int q = 0; q < 3; q++) h.fc[q] = (q < D && h.nc > 0) ? h.cfl[q * PMC_CFL] : 0.0;
}

// Box filter.  A candidate farther than the largest cutoff from the centre cell's box is out of
// range of every particle that stays inside that cell: it contributes exactly 0 to every slot of
// this visit, and since every sum is exact, dropping it changes no result.  This compacts the
// staged candidates in place (LDS -> registers -> LDS), keeping the centre cell contiguous.
// Preconditions: everything is staged (ncand <= cap) and the centre particles' cell coordinates
// are cached (cfl).  About 30 % of a 27-cell neighbourhood goes away at w ~ rcut.
template <int D, bool MULTI, int NW, bool BIG = false>
__device__ __forceinline__ void compact_candidates(const CbArgs &a, WaveNbh &h)
{
    constexpr int MAXC = 8;                       // chunks of 64 candidates per wavefront held in registers
    const int lane = wave_lane(), wv = team_wave<NW>();
    const int nchunk = (h.ncand + PMC_WAVE - 1) / PMC_WAVE;
    constexpr int SEGN = BIG ? PMC_SEGTAB_TEAM : 32;                   // chunk counts segoff holds
    if (h.compacted || !a.filter || h.nc == 0 || h.ncand > h.cap || nchunk > MAXC * NW || nchunk > SEGN) return;
    // box centre (in the frame of the first centre particle) and padded half width; candidates in
    // another periodic image are brought to the box by the minimum image (|t| <= 2 w <= L/2)
    double bc[3] = {0, 0, 0}, bh[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) {
        const double f = h.fc[q];
        bc[q] = a.shift[q] + (f + 0.5) * a.w[q];
        bh[q] = 0.5 * a.w[q] * (1.0 + 1e-9);
    }
    const bool wrap = !h.nowrap;
    double tx[MAXC], ty[MAXC], tz[MAXC];
    unsigned char ts[MAXC];
    unsigned long long keepm[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        const int chunk = c * NW + wv;
        const int k = chunk * PMC_WAVE + lane;
        tx[c] = ty[c] = tz[c] = 0.0;
        ts[c] = 0;
        bool keep = false;
        if (chunk < nchunk && k < h.ncand) {
            tx[c] = h.cx[k];
            ty[c] = h.cy[k];
            if (D == 3) tz[c] = h.cz[k];
            if (MULTI) ts[c] = h.csp[k];
            double t0 = tx[c] - bc[0], t1 = ty[c] - bc[1], t2 = (D == 3) ? tz[c] - bc[2] : 0.0;
            if (wrap) {
                t0 -= rint(t0 * a.invL[0]) * a.L[0];
                t1 -= rint(t1 * a.invL[1]) * a.L[1];
                if (D == 3) t2 -= rint(t2 * a.invL[2]) * a.L[2];
            }
            const double ex = fmax(fabs(t0) - bh[0], 0.0);
            const double ey = fmax(fabs(t1) - bh[1], 0.0);
            double d2 = ex * ex + ey * ey;
            if (D == 3) {
                const double ez = fmax(fabs(t2) - bh[2], 0.0);
                d2 += ez * ez;
            }
            keep = (k >= h.ctr && k < h.ctr + h.nc) || !(d2 > a.rc2pad);
        }
        keepm[c] = __ballot(keep);
        if (lane == 0 && chunk < SEGN) h.segoff[chunk] = __popcll(keepm[c]);   // kept count of this chunk
    }
    __syncthreads();                              // all reads of the old layout are done; counts published
    int base_of[MAXC];
    {
        int run = 0;
#pragma unroll
        for (int chunk = 0; chunk < MAXC * NW; chunk++) {
            if ((chunk % NW) == wv) base_of[chunk / NW] = run;
            if (chunk < nchunk) run += h.segoff[chunk];
        }
        __syncthreads();                          // counts consumed before segoff is rewritten
        const int total = run;
#pragma unroll
        for (int c = 0; c < MAXC; c++) {
            const int chunk = c * NW + wv;
            const int k = chunk * PMC_WAVE + lane;
            const unsigned long long m = keepm[c];
            if ((m >> lane) & 1ull) {
                const int pos = base_of[c] + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                h.cx[pos] = tx[c];
                h.cy[pos] = ty[c];
                if (D == 3) h.cz[pos] = tz[c];
                if (MULTI) h.csp[pos] = ts[c];
                if (k == h.ctr) h.flag[1] = pos;
            }
        }
        __syncthreads();
        h.ctr = h.flag[1];
        h.ncand = total;
        h.nseg = 1;
        if (lane == 0 && wv == 0) { h.segoff[0] = 0; h.segoff[1] = total; h.segsrc[0] = 0; }
        __syncthreads();
    }
}

template <int D, bool MULTI>
__device__ __forceinline__ void nbh_load(const CbArgs &a, const WaveNbh &h, int k, double &xj, double &yj,
                                         double &zj, int &sj)
{
    if (k < h.cap) {
        xj = h.cx[k];
        yj = h.cy[k];
        zj = (D == 3) ? h.cz[k] : 0.0;
        sj = MULTI ? (int)h.csp[k] : 0;
    } else {
        int src = nbh_find(h, k);
        xj = ld_agent(a.sx + src);
        yj = ld_agent(a.sy + src);
        zj = (D == 3) ? ld_agent(a.sz + src) : 0.0;
        sj = MULTI ? (int)ld_agent(a.ssp + src) : 0;
    }
}

// squared distance when the minimum image is known to be the identity (WaveNbh::nowrap)
template <int D>
__device__ __forceinline__ double dist2_nowrap(double xi, double yi, double zi, double xj, double yj, double zj)
{
    double vx = xi - xj, vy = yi - yj;
    double r2 = vx * vx;
    r2 = r2 + vy * vy;
    if (D == 3) {
        double vz = zi - zj;
        r2 = r2 + vz * vz;
    }
    return r2;
}

template <int D>
__device__ __forceinline__ double dist2_fast(const CbArgs &a, double xi, double yi, double zi, double xj,
                                             double yj, double zj)
{
    double vx = dev_vector_1d_fast(xi, xj, a.L[0], a.invL[0]);
    double vy = dev_vector_1d_fast(yi, yj, a.L[1], a.invL[1]);
    double r2 = vx * vx;
    r2 = r2 + vy * vy;
    if (D == 3) {
        double vz = dev_vector_1d_fast(zi, zj, a.L[2], a.invL[2]);
        r2 = r2 + vz * vz;
    }
    return r2;
}

// v + (v moved by a DPP lane pattern); 64-bit values move as two 32-bit halves
template <int CTRL>
__device__ __forceinline__ double dpp_add(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}

// Sum over the 64 lanes of values that add EXACTLY (multiples of 2^-32, see dev_quant), so
// the association below is free to differ from any other and still give the same bits.
// (The same sum on the matrix core -- v_mfma_f64_16x16x4 with a matrix of ones -- is exact too and was measured 7 % slower:
// its latency sits on the slot's critical path.  LAB.md, round 3.)
__device__ __forceinline__ double wave_sum_exact(double v)
{
    v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);   // row_half_mirror
    v = dpp_add<0x140>(v);   // row_mirror: every lane of a 16-lane row holds the row sum
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3: lane 63 ends with the total (rows a DPP move does
    // not write keep the 0 given as their old value)
    {
        int lo = __double2loint(v), hi = __double2hiint(v);
        int lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x142, 0xA, 0xF, false);
        int hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x142, 0xA, 0xF, false);
        v = v + __hiloint2double(hi2, lo2);
        lo = __double2loint(v); hi = __double2hiint(v);
        lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x143, 0xC, 0xF, false);
        hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x143, 0xC, 0xF, false);
        v = v + __hiloint2double(hi2, lo2);
    }
    return readlane_double(v, 63);
}

// exact sum over all lanes of the NW wavefronts of a team; result in every lane (2 barriers when NW > 1)
template <int NW>
__device__ __forceinline__ double team_sum_exact(const WaveNbh &h, double v)
{
    v = wave_sum_exact(v);
    if (NW == 1) return v;
    if (wave_lane() == 0) h.part[team_wave<NW>()] = v;
    __syncthreads();
    double r = h.part[0];
#pragma unroll
    for (int w = 1; w < NW; w++) r += h.part[w];
    __syncthreads();
    return r;
}


template <int KIND>
__device__ __forceinline__ double pot_kind(double r2, const DevModel &m)
{
    if (KIND == 4) return dev_potential_dyn(r2, m);
    return dev_potential_inrange<(KIND == 4 ? 0 : KIND)>(r2, m);
}

// ------------------------------------------------------------------------------
// FP32 pre-filter + survivor queue + dense FP64 evaluation (one wavefront, one Displacement slot)
//
// Only ~19 % of the (slot, candidate) pairs of a 3^d-cell neighbourhood are inside the cutoff.  Testing all of
// them in FP64 and branching into the potential with a fifth of the lanes live is what made the first
// version of this kernel VALU-bound at 587 wave-instructions per move.  Here every candidate is first tested
// in FP32 against a CONSERVATIVE threshold, the survivors (candidate, old-or-new position) are compacted into
// a per-wavefront queue in LDS (ballot + mbcnt), and only the queue is evaluated in FP64 -- the exact distance,
// the exact `r2 > rcut2` test of src/atoms.jl:72 and the potential -- on dense lanes.
//
// Why results do not change by a single bit: a pair the filter drops has r2 > rcut2 in FP64 (bound below) and
// contributes exactly 0 in the reference arithmetic; a pair it keeps is evaluated by exactly the FP64 operations
// of the direct loop; every term is a multiple of 2^-32 and the sums are exact (dev_quant), so the order in which
// the queue yields them is free.
//
// Error bound of the filter.  Float coordinates are rel = fl32(x - bc [- image]) with |rel| <= R = 1.5 w + |delta|,
// so each carries an error <= 2^-24 R; a difference d = fl32(rel_j - rel_i) is off by <= 2^-24 (2 R + |d|); the
// three squares and two sums add a relative 3 * 2^-24.  For |d| <= rcut this gives
//   |r2_f32 - r2| <= 2^-24 rcut^2 (2 sqrt(3) (2 R + rcut) / rcut + 3) < 2^-24 * 32 rcut^2 = 2^-19 rcut^2
// because w <= 1.5 rcut (n >= 4 cells per side) makes R <= 2.3 rcut.  The threshold is rcut_max^2 (1 + 2^-12):
// 128 times that bound.  (Pairs at |d| ~ L/2, where the two image choices differ, are at >= 2 rcut either way.)
// ------------------------------------------------------------------------------

// float image of a coordinate of the centre cell's neighbourhood
template <bool NOWRAP>
__device__ __forceinline__ float rel_float(double x, double bc, double L, double invL)
{
    double t = x - bc;
    if (!NOWRAP) t -= rint(t * invL) * L;
    return (float)t;
}

// Float coordinates (relative to the centre of the centre cell, folded into the nearest image) of the candidates a
// lane tests: lane l tests candidates l, 64 + l, 128 + l, ... in every slot of the visit, so it keeps exactly those in
// registers -- the filter then reads no LDS at all.  NaN marks the slots behind the last candidate (NaN fails the
// `r2 <= threshold` test whatever the image folding did to it).
struct FRegs {
    float x[PMC_FTRIPS], y[PMC_FTRIPS], z[PMC_FTRIPS];
};

template <int D, bool NOWRAP>
__device__ __forceinline__ void load_fregs(const CbArgs &a, const WaveNbh &h, FRegs &f)
{
    const int lane = wave_lane();
#pragma unroll
    for (int j = 0; j < PMC_FTRIPS; j++) {
        const int k = j * PMC_WAVE + lane;
        f.x[j] = f.y[j] = f.z[j] = __builtin_nanf("");
        if (k < h.ncand) {
            f.x[j] = rel_float<NOWRAP>(h.cx[k], h.bc[0], a.L[0], a.invL[0]);
            f.y[j] = rel_float<NOWRAP>(h.cy[k], h.bc[1], a.L[1], a.invL[1]);
            f.z[j] = (D == 3) ? rel_float<NOWRAP>(h.cz[k], h.bc[2], a.L[2], a.invL[2]) : 0.0f;
        }
    }
}

// after the box filter has compacted the staged candidates: decide whether the pre-filter applies to this visit and
// load every wavefront's register copy
template <int D, int NW = 1, bool BIG = false>
__device__ __forceinline__ void prepare_filter(const CbArgs &a, WaveNbh &h, FRegs &f)
{
#pragma unroll
    for (int q = 0; q < 3; q++) h.bc[q] = (q < D) ? a.shift[q] + (h.fc[q] + 0.5) * a.w[q] : 0.0;
    // (single-cell boxes: differences of two folded float coordinates need the minimum image themselves, which only the streamed
    // filter of the big-neighbourhood instances applies; the other instances take the direct FP64 loop there)
    h.filt = (h.nc > 0) && (h.ncand <= h.cap) && (h.ncand <= PMC_FTRIPS * PMC_WAVE) && a.n[0] != 1;
    if (BIG && !h.filt && h.fcap > 0 && h.nc > 0 && h.ncand <= h.cap && h.ncand <= h.fcap) {
        // A neighbourhood beyond the 512 candidates of the register copy (boxes of 4-5 cutoffs per side at liquid density: 3^d cells
        // of a hundred particles each) that still fits the LDS: the team shares ONE float copy in LDS and every slot's filter
        // streams it, two trips at a time, into the same survivor queue.  Until round 5 such a slot took the direct FP64 loop over
        // every candidate at both positions.
        for (int k = team_tid<NW>(); k < h.ncand; k += NW * PMC_WAVE) {
            if (h.nowrap) {
                h.fx[k] = rel_float<true>(h.cx[k], h.bc[0], a.L[0], a.invL[0]);
                h.fy[k] = rel_float<true>(h.cy[k], h.bc[1], a.L[1], a.invL[1]);
                h.fz[k] = (D == 3) ? rel_float<true>(h.cz[k], h.bc[2], a.L[2], a.invL[2]) : 0.0f;
            } else {
                h.fx[k] = rel_float<false>(h.cx[k], h.bc[0], a.L[0], a.invL[0]);
                h.fy[k] = rel_float<false>(h.cy[k], h.bc[1], a.L[1], a.invL[1]);
                h.fz[k] = (D == 3) ? rel_float<false>(h.cz[k], h.bc[2], a.L[2], a.invL[2]) : 0.0f;
            }
        }
        h.fstream = true;
        __syncthreads();
        return;
    }
    if (!h.filt) return;
    if (h.nowrap) load_fregs<D, true>(a, h, f); else load_fregs<D, false>(a, h, f);
}

// "The proposal stays in its cell": floor((x_new - shift) / w) == floor((x_old - shift) / w) in every dimension.
// Decided from the FP32 coordinate of the new position relative to the cell's centre (the one the pre-filter uses, folded
// into the centre's periodic image): |t| < w/2 - e is inside, |t| > w/2 + e is outside, and only in the band between
// (e = 2^-12 w; the float is within 2^-22 w of the exact offset, the FP64 divisions of the exact test within 2^-50 of
// their real quotients) -- about one slot in 1500 -- the divisions themselves decide.  This replaced exact cell
// boundaries found per visit by stepping ulps through that division (~500 instructions per visit).
template <int D>
__device__ __forceinline__ bool stays_in_cell(const CbArgs &a, const WaveNbh &h, int p, double xo, double yo, double zo, double xn, double yn,
                                              double zn, double my_shift, double my_w)
{
    const int lane = wave_lane();
    float t[3] = {0.0f, 0.0f, 0.0f};
    if (h.nowrap) {
        t[0] = rel_float<true>(xn, h.bc[0], a.L[0], a.invL[0]);
        t[1] = rel_float<true>(yn, h.bc[1], a.L[1], a.invL[1]);
        if (D == 3) t[2] = rel_float<true>(zn, h.bc[2], a.L[2], a.invL[2]);
    } else {
        t[0] = rel_float<false>(xn, h.bc[0], a.L[0], a.invL[0]);
        t[1] = rel_float<false>(yn, h.bc[1], a.L[1], a.invL[1]);
        if (D == 3) t[2] = rel_float<false>(zn, h.bc[2], a.L[2], a.invL[2]);
    }
    // (the fold maps a displacement of about a whole box back onto the cell: such a proposal takes the exact test)
    bool in = h.nowrap || (fabs(xn - xo) < 0.25 * a.L[0] && fabs(yn - yo) < 0.25 * a.L[1] && (D == 2 || fabs(zn - zo) < 0.25 * a.L[2]));
    bool out = false;
    const bool sane = in;
#pragma unroll
    for (int q = 0; q < D; q++) {
        const float hw = (float)(0.5 * a.w[q]), e = hw * (1.0f / 2048.0f);
        in = in && (fabsf(t[q]) < hw - e);
        out = out || (fabsf(t[q]) > hw + e);
    }
    if (__builtin_amdgcn_ballot_w64(in) != 0ull) return true;       // (every lane holds the same values)
    if (__builtin_amdgcn_ballot_w64(out && sane) != 0ull) return false;
    bool bad = false;
    if (lane < D) {
        const double xs_n = (lane == 0) ? xn : ((lane == 1) ? yn : zn);
        const double xs_o = (lane == 0) ? xo : ((lane == 1) ? yo : zo);
        const double fo = (h.have_cfl && p < PMC_CFL) ? h.cfl[lane * PMC_CFL + p] : mc_coord<D>(xs_o, my_shift, my_w);
        bad = mc_coord<D>(xs_n, my_shift, my_w) != fo;
    }
    return __ballot(bad) == 0ull;
}

// an accepted move of the particle in LDS slot pc: the owning lane replaces its float copy
template <int D>
__device__ __forceinline__ void update_fregs(FRegs &f, int pc, float fx, float fy, float fz, bool first_trip_only = false)
{
    const int rel = pc - wave_lane();
    if (first_trip_only) {
        // the fused staging puts the centre cell first (LDS slots 0 .. nc-1, nc <= 64): a moved particle is in trip 0
        const bool mine = (rel == 0);
        f.x[0] = mine ? fx : f.x[0];
        f.y[0] = mine ? fy : f.y[0];
        if (D == 3) f.z[0] = mine ? fz : f.z[0];
        return;
    }
#pragma unroll
    for (int j = 0; j < PMC_FTRIPS; j++) {
        const bool mine = (rel == j * PMC_WAVE);
        f.x[j] = mine ? fx : f.x[j];
        f.y[j] = mine ? fy : f.y[j];
        if (D == 3) f.z[j] = mine ? fz : f.z[j];
    }
}

#ifdef PMC_DEBUG_SLOTLOG   // development builds only (single translation unit): what a team's resolution decided, slot by slot
__device__ double *g_slotlog = nullptr;
#endif

struct PairCount {
    unsigned tests, energies;     // per wavefront: FP32 distance tests, in-range pair energies evaluated
    unsigned dense;               // FP64 distance evaluations: (queue entry, old | new position) of the filtered path, every (candidate, position) of the direct one
    bool count_dense;             // ... counted only by the kernels that have a register for it: one wavefront per cell, one species (the
                                  // benchmark's instance carries it for free; in the others it cost 8-24 B of scratch per lane)
    PMC_TIME(unsigned long long t_filter, t_dense;)
};

// the FP64 evaluation of the queued survivors [0, qn) of one slot; returns this lane's partial sum
// SWAP (DiscreteSwap of the particles in LDS slots pc and pj, species sa <-> sb; `si` carries sa): the two "positions" are the
// two particles, which do not move; a lane adds the pair energy with the species its particle takes and subtracts the one with
// the species it has (the terms of src/moves.jl:189-207 = swap_delta below, candidate by candidate).
template <int D, int KIND, bool MULTI, bool NOWRAP, bool SWAP = false>
__device__ __forceinline__ double queue_flush(const CbArgs &a, const WaveNbh &h, const DevModel *tab, const unsigned short *queue,
                                              int qn, int pc, int si, double xo, double yo, double zo, double xn, double yn, double zn,
                                              PairCount &pcnt, int pj = 0, int sb = 0)
{
    const int lane = wave_lane();
    double acc = 0.0;
    qn = __builtin_amdgcn_readfirstlane(qn);
    pc = __builtin_amdgcn_readfirstlane(pc);
#ifdef PMC_DEBUG_NODENSE   // profiling builds only: price the filter alone
    qn = 0;
#endif
    // An entry is a candidate that passed the filter at the old OR the new position; it is evaluated at both, by two
    // neighbouring lanes: even lanes take the old position (and subtract), odd lanes the new one (and add).  Which of the
    // two a lane takes never changes, so the position is picked once per slot instead of once per trip, and the exact test
    // `r2 > rcut2` below drops the half of a pair that only rode along.
    const bool isnew = (lane & 1) != 0;
    const double xi = isnew ? xn : xo, yi = isnew ? yn : yo, zi = isnew ? zn : zo;
    const int qe = 2 * qn;
    if (pcnt.count_dense) pcnt.dense += (unsigned)qe;
    for (int e0 = 0; e0 < qe; e0 += PMC_WAVE) {
        // lanes behind the last entry re-read it (no branch around the loads) and are masked out of the range test
        const int e = min(e0 + lane, qe - 1);
        const int k = (int)queue[e >> 1];
        const double xj = h.cx[k], yj = h.cy[k], zj = (D == 3) ? h.cz[k] : 0.0;
        const double r2 = NOWRAP ? dist2_nowrap<D>(xi, yi, zi, xj, yj, zj) : dist2_fast<D>(a, xi, yi, zi, xj, yj, zj);
        if (SWAP && MULTI) {
            const int sk = (int)h.csp[k];
            // even lanes: particle pc (species sa = si) becomes sb; odd lanes: particle pj (species sb) becomes sa
            const DevModel &mplus = tab[(isnew ? si : sb) * a.S + sk], &mminus = tab[(isnew ? sb : si) * a.S + sk];
            const bool valid = (e0 + lane < qe) && (k != pc) && (k != pj);
            const bool inp = valid && !(r2 > mplus.rcut2), inm = valid && !(r2 > mminus.rcut2);
            pcnt.energies += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(inp)) + (unsigned)__popcll(__builtin_amdgcn_ballot_w64(inm));
            if (inp) acc += dev_quant(pot_kind<KIND>(r2, mplus));
            if (inm) acc -= dev_quant(pot_kind<KIND>(r2, mminus));
        } else {
            const DevModel &mm = MULTI ? tab[si * a.S + (int)h.csp[k]] : a.m0;
            const bool inr = (e0 + lane < qe) && (k != pc) && !(r2 > mm.rcut2);     // k == pc: the particle itself
            pcnt.energies += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(inr));
            if (inr) {
                const double u = dev_quant(pot_kind<KIND>(r2, mm));
                acc += isnew ? u : -u;
            }
        }
    }
    return acc;
}

// dE = sum_j q(u(new, j)) - q(u(old, j)) of moving the particle in LDS slot pc from (xo,yo,zo) to (xn,yn,zn), by ONE
// wavefront, result in every lane.  f = this wavefront's register copy; fo* / fn* = float images of the old and new
// position (rel_float).  Requires h.filt.  `overflow` is set (and nothing returned) when the survivor queue cannot hold
// the next trip: the caller then takes the direct loop for this slot (dense clusters only).
template <int D, int KIND, bool MULTI, bool SWAP = false, bool STREAM = false>
__device__ __forceinline__ double wave_delta_e_filtered(const CbArgs &a, const WaveNbh &h, const DevModel *tab, unsigned short *queue,
                                                        const FRegs &f, int pc, int si, double xo, double yo, double zo,
                                                        double xn, double yn, double zn, float fxo, float fyo, float fzo,
                                                        float fxn, float fyn, float fzn, PairCount &pcnt, bool &overflow,
                                                        int pj = 0, int sb = 0)
{
    const int lane = wave_lane();
    // wave-uniform values the compiler cannot prove uniform (they come from loads and cross-lane moves): pin them to
    // SGPRs so that the trip guards, the ballot masks and the queue cursor stay on the scalar unit
    const int ncand = __builtin_amdgcn_readfirstlane(h.ncand);
    const float thr = a.rc2f;
    unsigned qn = 0;
    int nrun = ncand;
#ifdef PMC_DEBUG_NOFILTER   // profiling builds only: price the slot prologue / reduction alone
    nrun = 0;
#endif
#ifdef PMC_DEBUG_TRIPS      // profiling builds only: price one filter trip
    nrun = min(nrun, PMC_DEBUG_TRIPS * PMC_WAVE);
#endif
    overflow = false;
    PMC_T(tf_0);
    // The particle itself (distance 0 to its old position) is NOT masked here: it passes the filter like any close
    // neighbour and queue_flush drops it by index -- one compare per dense trip instead of three operations per filter trip.
    // (old, new) pairs in one packed FP32 instruction each: measured equal in time to twelve plain v_*_f32 (a trip
    // holds the VALU for ~40 quad-cycles either way, rocprofv3 SQ_INSTS_VALU), with half the instructions to issue.
    // Trips are processed TWO at a time in straight-line code: a trip is one dependent chain (subtract, multiply, two
    // fused multiply-adds, compare, two lane counts, address, store), and with only 2-3 wavefronts per SIMD nothing
    // else hides its stalls; two trips give the scheduler four independent chains to interleave.
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 px = {fxo, fxn}, py = {fyo, fyn}, pz = {fzo, fzn};
    auto trips2 = [&](auto jc) {
        constexpr int J = decltype(jc)::value;
        f2 r2[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f2 cx = {f.x[J + t], f.x[J + t]}, cy = {f.y[J + t], f.y[J + t]};
            const f2 dx = cx - px, dy = cy - py;
            r2[t] = __builtin_elementwise_fma(dy, dy, dx * dx);
            if (D == 3) {
                const f2 cz = {f.z[J + t], f.z[J + t]};
                const f2 dz = cz - pz;
                r2[t] = __builtin_elementwise_fma(dz, dz, r2[t]);
            }
        }
        // one queue entry per candidate that passes at either position: its LDS slot
        const bool b0 = (r2[0].x <= thr) | (r2[0].y <= thr), b1 = (r2[1].x <= thr) | (r2[1].y <= thr);
        const unsigned long long m0 = __builtin_amdgcn_ballot_w64(b0), m1 = __builtin_amdgcn_ballot_w64(b1);
        // slots behind the last candidate hold NaN and never pass
        const unsigned k0 = (unsigned)lane + (unsigned)(J * PMC_WAVE), k1 = k0 + (unsigned)PMC_WAVE;
        // (the running count goes into the uniform base pointer, not into the lane count: one instruction less per push)
        const unsigned q0 = qn, q1 = q0 + (unsigned)__popcll(m0);
        if (b0) (queue + q0)[__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] = (unsigned short)k0;
        if (b1) (queue + q1)[__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] = (unsigned short)k1;
        qn = q1 + (unsigned)__popcll(m1);
    };
    auto trips = [&](auto jc) { trips2(jc); };
    static_assert(PMC_FTRIPS == 8 && PMC_QCAP >= 2 * 2 * PMC_WAVE, "the first two pairs of trips must fit the queue unchecked");
    if constexpr (STREAM) {
        // the float copy of a big neighbourhood lives in LDS (prepare_filter): the same test, two trips of 64 candidates at a time
        const float nanf_ = __builtin_nanf("");
        // single-cell box: both float coordinates are folded to within L/2 of the cell centre, their difference can be up to L:
        // d -= L rint(d / L) in float.  (A pair inside the cutoff has |d_true| <= rcut <= L/2 (1 - 2^-10): the host only
        // takes boxes of >= 2 (1 + 2^-10) cutoffs as single cells, so its quotient is >= 4.9e-4 from a half-integer, three orders
        // above the float error of d / L, and rint picks the right image; where it might not, the pair is beyond the cutoff.)
        const bool single = (a.n[0] == 1);
        const f2 Lx = {(float)a.L[0], (float)a.L[0]}, Ly = {(float)a.L[1], (float)a.L[1]}, Lz = {(float)a.L[2], (float)a.L[2]};
        const f2 iLx = {(float)a.invL[0], (float)a.invL[0]}, iLy = {(float)a.invL[1], (float)a.invL[1]}, iLz = {(float)a.invL[2], (float)a.invL[2]};
        auto wrap = [](f2 d, f2 L, f2 iL) -> f2 {
            const f2 q = d * iL;
            const f2 n = {__builtin_rintf(q.x), __builtin_rintf(q.y)};
            return __builtin_elementwise_fma(-n, L, d);
        };
        for (int base = 0; base < nrun; base += 2 * PMC_WAVE) {
            if (qn > (unsigned)(h.qcap - 2 * PMC_WAVE)) { overflow = true; break; }
            f2 r2[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int k = base + t * PMC_WAVE + lane;
                const bool live = k < ncand;
                const float vx = live ? h.fx[k] : nanf_, vy = live ? h.fy[k] : nanf_, vz = (D == 3 && live) ? h.fz[k] : 0.0f;
                const f2 cx = {vx, vx}, cy = {vy, vy};
                f2 dx = cx - px, dy = cy - py;
                if (single) { dx = wrap(dx, Lx, iLx); dy = wrap(dy, Ly, iLy); }
                r2[t] = __builtin_elementwise_fma(dy, dy, dx * dx);
                if (D == 3) {
                    const f2 cz = {vz, vz};
                    f2 dz = cz - pz;
                    if (single) dz = wrap(dz, Lz, iLz);
                    r2[t] = __builtin_elementwise_fma(dz, dz, r2[t]);
                }
            }
            const bool b0 = (r2[0].x <= thr) | (r2[0].y <= thr), b1 = (r2[1].x <= thr) | (r2[1].y <= thr);
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(b0), m1 = __builtin_amdgcn_ballot_w64(b1);
            const unsigned k0 = (unsigned)(base + lane), k1 = k0 + (unsigned)PMC_WAVE;
            const unsigned q0 = qn, q1 = q0 + (unsigned)__popcll(m0);
            if (b0) (queue + q0)[__builtin_amdgcn_mbcnt_hi((unsigned)(m0 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m0, 0u))] = (unsigned short)k0;
            if (b1) (queue + q1)[__builtin_amdgcn_mbcnt_hi((unsigned)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m1, 0u))] = (unsigned short)k1;
            qn = q1 + (unsigned)__popcll(m1);
        }
    } else {
    if (nrun > 0) trips(std::integral_constant<int, 0>{});
    if (nrun > 2 * PMC_WAVE) trips(std::integral_constant<int, 2>{});
    if (nrun > 4 * PMC_WAVE) {
        // each further pair adds at most 128 entries
        if (qn > (unsigned)(PMC_QCAP - 2 * PMC_WAVE)) overflow = true; else trips(std::integral_constant<int, 4>{});
    }
    if (nrun > 6 * PMC_WAVE && !overflow) {
        if (qn > (unsigned)(PMC_QCAP - 2 * PMC_WAVE)) overflow = true; else trips(std::integral_constant<int, 6>{});
    }
    }
    if (overflow) return 0.0;
    pcnt.tests += 2u * (unsigned)ncand;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PMC_T(tf_1);
    double acc = h.nowrap ? queue_flush<D, KIND, MULTI, true, SWAP>(a, h, tab, queue, (int)qn, pc, si, xo, yo, zo, xn, yn, zn, pcnt, pj, sb)
                          : queue_flush<D, KIND, MULTI, false, SWAP>(a, h, tab, queue, (int)qn, pc, si, xo, yo, zo, xn, yn, zn, pcnt, pj, sb);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PMC_TIME(
    {
        // the stamp must not be taken before the FP64 work has finished: make it depend on the sum
        const double probe = acc;
        asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 0" :: "v"(probe) : "memory");
        PMC_T(tf_2);
        pcnt.t_filter += tf_1 - tf_0;
        pcnt.t_dense += tf_2 - tf_1;
    }
    )
    return wave_sum_exact(acc);
}

// dE of one Displacement slot by one wavefront: the pre-filter where it applies, the direct loop otherwise
template <int D, int KIND, bool MULTI>
__device__ __forceinline__ double wave_delta_e_direct(const CbArgs &a, const WaveNbh &h, const DevModel *tab, int pc, int si,
                                                      double xo, double yo, double zo, double xn, double yn, double zn,
                                                      PairCount &pcnt);

template <int D, int KIND, bool MULTI, bool TEAM = false>
__device__ __forceinline__ double wave_delta_e(const CbArgs &a, const WaveNbh &h, const DevModel *tab, unsigned short *queue,
                                               const FRegs &f, int pc, int si, double xo, double yo, double zo, double xn,
                                               double yn, double zn, float &fxn, float &fyn, float &fzn, PairCount &pcnt)
{
    if (TEAM && h.fstream) {
        // the filter streams the team's float copy in LDS, which also holds the particle's own image (kept current)
        fzn = 0.0f;
        if (h.nowrap) {
            fxn = rel_float<true>(xn, h.bc[0], a.L[0], a.invL[0]);
            fyn = rel_float<true>(yn, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzn = rel_float<true>(zn, h.bc[2], a.L[2], a.invL[2]);
        } else {
            fxn = rel_float<false>(xn, h.bc[0], a.L[0], a.invL[0]);
            fyn = rel_float<false>(yn, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzn = rel_float<false>(zn, h.bc[2], a.L[2], a.invL[2]);
        }
        const float fxo = h.fx[pc], fyo = h.fy[pc], fzo = (D == 3) ? h.fz[pc] : 0.0f;
        bool overflow;
        const double de = wave_delta_e_filtered<D, KIND, MULTI, false, TEAM>(a, h, tab, queue, f, pc, si, xo, yo, zo, xn, yn, zn,
                                                                             fxo, fyo, fzo, fxn, fyn, fzn, pcnt, overflow);
        if (!overflow) return de;
        return wave_delta_e_direct<D, KIND, MULTI>(a, h, tab, pc, si, xo, yo, zo, xn, yn, zn, pcnt);
    }
    if (h.filt) {
        float fxo, fyo, fzo = 0.0f;
        fzn = 0.0f;
        if (h.nowrap) {
            fxn = rel_float<true>(xn, h.bc[0], a.L[0], a.invL[0]);
            fyn = rel_float<true>(yn, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzn = rel_float<true>(zn, h.bc[2], a.L[2], a.invL[2]);
        } else {
            fxn = rel_float<false>(xn, h.bc[0], a.L[0], a.invL[0]);
            fyn = rel_float<false>(yn, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzn = rel_float<false>(zn, h.bc[2], a.L[2], a.invL[2]);
        }
        if (h.compacted && h.nc <= PMC_WAVE) {
            // centre cell first: the particle's own float copy sits in trip 0 of lane pc (the same bits: load_fregs computed it
            // by rel_float from the same double, update_fregs stores the fxn of an accepted move)
            const int pl = __builtin_amdgcn_readfirstlane(pc);
            fxo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.x[0]), pl));
            fyo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.y[0]), pl));
            if (D == 3) fzo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f.z[0]), pl));
        } else if (h.nowrap) {
            fxo = rel_float<true>(xo, h.bc[0], a.L[0], a.invL[0]);
            fyo = rel_float<true>(yo, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzo = rel_float<true>(zo, h.bc[2], a.L[2], a.invL[2]);
        } else {
            fxo = rel_float<false>(xo, h.bc[0], a.L[0], a.invL[0]);
            fyo = rel_float<false>(yo, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) fzo = rel_float<false>(zo, h.bc[2], a.L[2], a.invL[2]);
        }
        bool overflow;
        const double de = wave_delta_e_filtered<D, KIND, MULTI>(a, h, tab, queue, f, pc, si, xo, yo, zo, xn, yn, zn,
                                                                fxo, fyo, fzo, fxn, fyn, fzn, pcnt, overflow);
        if (!overflow) return de;
    }
    return wave_delta_e_direct<D, KIND, MULTI>(a, h, tab, pc, si, xo, yo, zo, xn, yn, zn, pcnt);
}

// The same dE by the direct loop (every candidate tested in FP64): visits whose neighbourhood does not fit the LDS
// staging capacity (candidates beyond it are read from HBM / L2).
template <int D, int KIND, bool MULTI>
__device__ __forceinline__ double wave_delta_e_direct(const CbArgs &a, const WaveNbh &h, const DevModel *tab, int pc, int si,
                                                      double xo, double yo, double zo, double xn, double yn, double zn,
                                                      PairCount &pcnt)
{
    const int lane = wave_lane();
    const int lim = min(h.ncand, h.cap);
    double dsum = 0.0;
    unsigned nin = 0;
    auto term = [&](int k, double xj, double yj, double zj, int sj, auto nowrap_tag) {
        constexpr bool NOWRAP = decltype(nowrap_tag)::value;
        const DevModel &mm = MULTI ? tab[si * a.S + sj] : a.m0;
        const double r2o = NOWRAP ? dist2_nowrap<D>(xo, yo, zo, xj, yj, zj) : dist2_fast<D>(a, xo, yo, zo, xj, yj, zj);
        const double r2n = NOWRAP ? dist2_nowrap<D>(xn, yn, zn, xj, yj, zj) : dist2_fast<D>(a, xn, yn, zn, xj, yj, zj);
        if (k != pc) {
            if (!(r2n > mm.rcut2)) { dsum += dev_quant(pot_kind<KIND>(r2n, mm)); nin++; }
            if (!(r2o > mm.rcut2)) { dsum -= dev_quant(pot_kind<KIND>(r2o, mm)); nin++; }
        }
    };
    auto sweep_lds = [&](auto nowrap_tag) {
        for (int k = lane; k < lim; k += PMC_WAVE)
            term(k, h.cx[k], h.cy[k], (D == 3) ? h.cz[k] : 0.0, MULTI ? (int)h.csp[k] : 0, nowrap_tag);
    };
    if (h.nowrap) sweep_lds(std::true_type{}); else sweep_lds(std::false_type{});
    for (int k = lim + lane; k < h.ncand; k += PMC_WAVE) {
        double xj, yj, zj;
        int sj;
        nbh_load<D, MULTI>(a, h, k, xj, yj, zj, sj);
        term(k, xj, yj, zj, sj, std::false_type{});
    }
    pcnt.tests += 2u * (unsigned)h.ncand;
    if (pcnt.count_dense) pcnt.dense += 2u * (unsigned)h.ncand;
    unsigned tot = nin;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) tot += __shfl_xor(tot, s, PMC_WAVE);
    pcnt.energies += tot;
    return wave_sum_exact(dsum);
}

// exact per-particle energy of a particle at (xi,yi,zi) with species si against the staged
// neighbourhood, skipping the LDS slots ex1 / ex2; result in every lane
template <int D, int KIND, bool MULTI, int NW = 1>
__device__ __forceinline__ double nbh_energy(const CbArgs &a, const WaveNbh &h, const DevModel *tab,
                                             double xi, double yi, double zi, int si, int ex1, int ex2)
{
    double e = 0.0;
    const int lim = min(h.ncand, h.cap);
    constexpr int TS = NW * PMC_WAVE;
    for (int k = team_tid<NW>(); k < lim; k += TS) {             // staged candidates: LDS
        const DevModel &m = MULTI ? tab[si * a.S + (int)h.csp[k]] : a.m0;
        double r2 = dist2_fast<D>(a, xi, yi, zi, h.cx[k], h.cy[k], (D == 3) ? h.cz[k] : 0.0);
        if (k != ex1 && k != ex2 && !(r2 > m.rcut2)) e += dev_quant(pot_kind<KIND>(r2, m));
    }
    for (int k = lim + team_tid<NW>(); k < h.ncand; k += TS) {   // overflow beyond the LDS capacity: HBM/L2
        double xj, yj, zj;
        int sj;
        nbh_load<D, MULTI>(a, h, k, xj, yj, zj, sj);
        const DevModel &m = MULTI ? tab[si * a.S + sj] : a.m0;
        double r2 = dist2_fast<D>(a, xi, yi, zi, xj, yj, zj);
        if (!(r2 > m.rcut2)) e += dev_quant(pot_kind<KIND>(r2, m));
    }
    return team_sum_exact<NW>(h, e);
}

// dE of exchanging the species of the particles in slots pi (species sa) and pj (species sb):
// [E(x_i, sb) + E(x_j, sa)] - [E(x_i, sa) + E(x_j, sb)] over the neighbourhood without pi, pj, in
// one pass (each distance is computed once and used with both species); every term is exact, so
// this equals the difference of the four separate sums bit for bit.
template <int D, int KIND, bool MULTI, int NW>
__device__ __forceinline__ double swap_delta(const CbArgs &a, const WaveNbh &h, const DevModel *tab, int pi, int pj,
                                             int sa, int sb)
{
    const double xi = h.cx[pi], yi = h.cy[pi], zi = (D == 3) ? h.cz[pi] : 0.0;
    const double xj = h.cx[pj], yj = h.cy[pj], zj = (D == 3) ? h.cz[pj] : 0.0;
    constexpr int TS = NW * PMC_WAVE;
    const int lim = min(h.ncand, h.cap);
    double d = 0.0;
    auto term = [&](int k, double xk, double yk, double zk, int sk) {
        const DevModel &ma = tab[sa * a.S + sk];
        const DevModel &mb = tab[sb * a.S + sk];
        const double r2i = dist2_fast<D>(a, xi, yi, zi, xk, yk, zk);
        const double r2j = dist2_fast<D>(a, xj, yj, zj, xk, yk, zk);
        if (k != pi && k != pj) {
            if (!(r2i > mb.rcut2)) d += dev_quant(pot_kind<KIND>(r2i, mb));
            if (!(r2j > ma.rcut2)) d += dev_quant(pot_kind<KIND>(r2j, ma));
            if (!(r2i > ma.rcut2)) d -= dev_quant(pot_kind<KIND>(r2i, ma));
            if (!(r2j > mb.rcut2)) d -= dev_quant(pot_kind<KIND>(r2j, mb));
        }
    };
    for (int k = team_tid<NW>(); k < lim; k += TS) term(k, h.cx[k], h.cy[k], (D == 3) ? h.cz[k] : 0.0, (int)h.csp[k]);
    for (int k = lim + team_tid<NW>(); k < h.ncand; k += TS) {
        double xk, yk, zk;
        int sk;
        nbh_load<D, MULTI>(a, h, k, xk, yk, zk, sk);
        term(k, xk, yk, zk, sk);
    }
    return team_sum_exact<NW>(h, d);
}

// dE of one DiscreteSwap slot by one wavefront: through the pre-filter and the survivor queue where they apply (the two
// particles play the part of the old and the new position: neither moves, so the register copy holds both), the direct loop
// of swap_delta otherwise.  A swap used to cost 3.4 Displacement slots -- and a team's batch waits for its slowest slot.
template <int D, int KIND, bool MULTI, bool TEAM = false>
__device__ __forceinline__ double wave_swap_delta(const CbArgs &a, const WaveNbh &h, const DevModel *tab, unsigned short *queue,
                                                  const FRegs &f, int pi, int pj, int sa, int sb, PairCount &pcnt)
{
    if (MULTI && TEAM && h.fstream) {
        const double xi = h.cx[pi], yi = h.cy[pi], zi = (D == 3) ? h.cz[pi] : 0.0;
        const double xj = h.cx[pj], yj = h.cy[pj], zj = (D == 3) ? h.cz[pj] : 0.0;
        bool overflow;
        const double de = wave_delta_e_filtered<D, KIND, MULTI, true, TEAM>(a, h, tab, queue, f, pi, sa, xi, yi, zi, xj, yj, zj,
                                                                            h.fx[pi], h.fy[pi], (D == 3) ? h.fz[pi] : 0.0f,
                                                                            h.fx[pj], h.fy[pj], (D == 3) ? h.fz[pj] : 0.0f, pcnt, overflow, pj, sb);
        if (!overflow) return de;
        return swap_delta<D, KIND, MULTI, 1>(a, h, tab, pi, pj, sa, sb);
    }
    if (MULTI && h.filt) {
        const double xi = h.cx[pi], yi = h.cy[pi], zi = (D == 3) ? h.cz[pi] : 0.0;
        const double xj = h.cx[pj], yj = h.cy[pj], zj = (D == 3) ? h.cz[pj] : 0.0;
        float fxi, fyi, fzi = 0.0f, fxj, fyj, fzj = 0.0f;
        if (h.nowrap) {
            fxi = rel_float<true>(xi, h.bc[0], a.L[0], a.invL[0]); fxj = rel_float<true>(xj, h.bc[0], a.L[0], a.invL[0]);
            fyi = rel_float<true>(yi, h.bc[1], a.L[1], a.invL[1]); fyj = rel_float<true>(yj, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) { fzi = rel_float<true>(zi, h.bc[2], a.L[2], a.invL[2]); fzj = rel_float<true>(zj, h.bc[2], a.L[2], a.invL[2]); }
        } else {
            fxi = rel_float<false>(xi, h.bc[0], a.L[0], a.invL[0]); fxj = rel_float<false>(xj, h.bc[0], a.L[0], a.invL[0]);
            fyi = rel_float<false>(yi, h.bc[1], a.L[1], a.invL[1]); fyj = rel_float<false>(yj, h.bc[1], a.L[1], a.invL[1]);
            if (D == 3) { fzi = rel_float<false>(zi, h.bc[2], a.L[2], a.invL[2]); fzj = rel_float<false>(zj, h.bc[2], a.L[2], a.invL[2]); }
        }
        bool overflow;
        const double de = wave_delta_e_filtered<D, KIND, MULTI, true>(a, h, tab, queue, f, pi, sa, xi, yi, zi, xj, yj, zj,
                                                                      fxi, fyi, fzi, fxj, fyj, fzj, pcnt, overflow, pj, sb);
        if (!overflow) return de;
    }
    return swap_delta<D, KIND, MULTI, 1>(a, h, tab, pi, pj, sa, sb);
}

// LDS bytes of one workgroup: coordinates, cached cell coordinates, partial sums, [batch records],
// flags, counters, segment tables, [species].  Kept small on purpose: at N = 10^6 (cap 512) a team of
// two wavefronts needs <= 13 312 B for twelve workgroups (six wavefronts per SIMD) per CU.
__host__ __device__ inline size_t pmc_lds_wave_bytes(int cap, bool multi, int batch_nw, int fcap = 0, int qcap = PMC_QCAP)
{
    // one wavefront per cell (batch_nw == 0): no partial sums, and the segment tables share the survivor queue's bytes -- a visit
    // either reads candidates beyond the LDS capacity through the tables (direct loop, no queue) or queues survivors
    // (everything staged, tables never read): 10 224 B at cap = 384, the 16th workgroup of a CU
    if (batch_nw == 0)
        return (size_t)3 * cap * 8 + 3 * PMC_CFL * 8 + 4 * 4 + 16 * 4 + 24 * 4 + (size_t)PMC_QCAP * 2 + (multi ? cap : 0);
    // (big neighbourhoods, fcap > 0: segment tables of PMC_SEGTAB_TEAM entries, queues of qcap entries, and the float copy)
    const size_t b = (size_t)3 * cap * 8 + 3 * PMC_CFL * 8 + 8 * 8 + (size_t)(2 * PMC_REC * batch_nw + batch_nw * batch_nw) * 8 + 4 * 4 + 16 * 4 +
                     24 * 4 + 2 * (fcap > 0 ? PMC_SEGTAB_TEAM : 32) * 4 + (size_t)batch_nw * qcap * 2 + (multi ? cap : 0);
    return fcap > 0 ? ((b + 15) & ~(size_t)15) + (size_t)3 * fcap * 4 : b;
}

template <int D, bool MULTI, int BATCH = 0, bool BIG = false>   // BATCH: team size of the speculative slot batches (0: none); BIG: see k_cb_pass
__device__ __forceinline__ const DevModel *setup_lds(const CbArgs &a, unsigned char *smem, WaveNbh &h)
{
    h.cap = a.cap;
    h.cx = reinterpret_cast<double *>(smem);
    h.cy = h.cx + h.cap;
    h.cz = h.cy + h.cap;
    h.cfl = h.cz + h.cap;
    h.part = h.cfl + 3 * PMC_CFL;
    h.rec = h.part + (BATCH > 0 ? 8 : 0);
    h.cmat = h.rec + 2 * PMC_REC * BATCH;      // two record buffers, used by alternate batches
    h.flag = reinterpret_cast<int *>(h.cmat + BATCH * BATCH);
    h.cnt = h.flag + 4;
    h.ptab = h.cnt + 16;
    static_assert(PMC_QCAP * 2 >= 2 * 32 * 4, "the segment tables fit into a survivor queue");
    h.qcap = BIG ? a.qcap : PMC_QCAP;
    h.fcap = BIG ? a.fcap : 0;
    constexpr int SEGN = BIG ? PMC_SEGTAB_TEAM : 32;
    if (BATCH > 0) {
        h.segoff = h.ptab + 24;
        h.segsrc = h.segoff + SEGN;
        h.queue = reinterpret_cast<unsigned short *>(h.segsrc + SEGN);
    } else {
        h.segoff = h.ptab + 24;                // (see pmc_lds_wave_bytes: shared with the queue)
        h.segsrc = h.segoff + 32;
        h.queue = reinterpret_cast<unsigned short *>(h.segoff);
    }
    h.csp = reinterpret_cast<unsigned char *>(h.queue + (BATCH > 0 ? BATCH : 1) * h.qcap);
    h.filt = false;
    h.fstream = false;
    h.fx = h.fy = h.fz = nullptr;
    if (BIG && h.fcap > 0) {
        // (behind everything else of the big layout -- its wider segment tables included: pmc_lds_wave_bytes adds the float copy last)
        h.fx = reinterpret_cast<float *>(smem + pmc_lds_wave_bytes(h.cap, MULTI, BATCH, h.fcap, h.qcap) - (size_t)3 * h.fcap * 4);
        h.fy = h.fx + h.fcap;
        h.fz = h.fy + h.fcap;
    }
    DevModel *tab = reinterpret_cast<DevModel *>(smem + ((pmc_lds_wave_bytes(h.cap, MULTI, BATCH, h.fcap, h.qcap) + 15) & ~(size_t)15));
    if (MULTI) {
        for (int k = threadIdx.x; k < a.S * a.S; k += blockDim.x) tab[k] = a.models[k];
        __syncthreads();
    }
    return tab;
}

template <int D>
__device__ __forceinline__ bool decode_cell(const CbArgs &a, int colour, int aidx, int cell[3], int &lin)
{
    int r = aidx;
    if (!a.all_cells && a.order) {
        lin = a.order[colour * a.nactive + aidx];
        r = lin;
#pragma unroll
        for (int q = D - 1; q >= 0; q--) { cell[q] = r % a.n[q]; r /= a.n[q]; }
        return true;
    }
    if (a.all_cells) {
#pragma unroll
        for (int q = D - 1; q >= 0; q--) { cell[q] = r % a.n[q]; r /= a.n[q]; }
    } else {
#pragma unroll
        for (int q = D - 1; q >= 0; q--) {
            cell[q] = 2 * (r % a.half[q]) + ((colour >> q) & 1);
            r /= a.half[q];
        }
    }
    lin = 0;
#pragma unroll
    for (int q = 0; q < D; q++) lin = lin * a.n[q] + cell[q];
    return true;
}

// sum(local_energy) (src/atoms.jl:51-52) over the cells of this launch, exact:
// stats[16] += sum_i e_i * 2^32; optional per-particle energies.
template <int D, int KIND, bool MULTI>
__global__ __launch_bounds__(64) void k_total_energy(CbArgs a)
{
    extern __shared__ __align__(16) unsigned char smem[];
    WaveNbh h;
    const DevModel *tab = setup_lds<D, MULTI>(a, smem, h);
    int aidx = xcd_remap(blockIdx.x, gridDim.x);
    if (aidx >= a.nactive) return;
    int cell[3] = {0, 0, 0}, lin;
    decode_cell<D>(a, a.colour, aidx, cell, lin);
    h.lin = lin;
    stage_neighbourhood<D, MULTI>(a, cell, h);
    if (h.nc == 0) return;
    unsigned long long *const st = a.stats + (size_t)(blockIdx.x & (PMC_NSHARD - 1)) * PMC_NSTATS;
    if (h.ctr + h.nc > h.cap) { if (wave_lane() == 0) atomicOr(&st[17], 1ull); return; }
    long long tot = 0;
    for (int p = 0; p < h.nc; p++) {
        const int pc = h.ctr + p;
        double xi = h.cx[pc], yi = h.cy[pc], zi = (D == 3) ? h.cz[pc] : 0.0;
        int si = MULTI ? (int)h.csp[pc] : 0;
        double e = nbh_energy<D, KIND, MULTI>(a, h, tab, xi, yi, zi, si, pc, -1);
        tot += (long long)(e * PMC_FIX);
        if (a.e_particle && wave_lane() == 0) a.e_particle[a.ids[h.k0 + p]] = e;
    }
    if (wave_lane() == 0) atomicAdd(&st[16], (unsigned long long)tot);
}

// k-th (0-based) centre-cell particle of species sp, in cell order (index inside the cell); uniform
__device__ __forceinline__ int kth_of_species(const WaveNbh &h, int sp, int k)
{
    for (int c0 = 0; c0 < h.nc; c0 += PMC_WAVE) {
        int idx = c0 + wave_lane();
        bool is = idx < h.nc && (int)h.csp[h.ctr + idx] == sp;
        unsigned long long b = __ballot(is);
        int cnt = __popcll(b);
        if (k < cnt) {
            for (int q = 0; q < k; q++) b &= b - 1;
            return c0 + (__ffsll((long long)b) - 1);
        }
        k -= cnt;
    }
    return -1;
}

__device__ __forceinline__ int count_species(const WaveNbh &h, int sp)
{
    int n = 0;
    for (int c0 = 0; c0 < h.nc; c0 += PMC_WAVE) {
        int idx = c0 + wave_lane();
        n += __popcll(__ballot(idx < h.nc && (int)h.csp[h.ctr + idx] == sp));
    }
    return n;
}

// Fused sweep: a cell of colour c may start once every neighbour cell of a LOWER colour has finished
// its visit of this sweep (those are the only cells that wrote coordinates this cell stages; cells of a
// higher colour start after this one has published).  Tickets are handed out in colour order, so a
// workgroup only ever waits for workgroups that already hold a ticket, i.e. are resident or finished:
// no deadlock whatever the dispatch order.  Wavefront 0 polls (relaxed agent-scope loads, one flag
// per lane; inside stage_neighbourhood, overlapped with the CSR loads), a barrier releases the other
// wavefronts; the staged data is then read with sc1 loads.

// publish the end of a cell visit: every wavefront drains its stores, then one lane sets the flag
__device__ __forceinline__ void publish_done(const CbArgs &a, int lin, bool leader)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (leader) __hip_atomic_store(a.done + lin, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One colour pass of the checkerboard sweep: one workgroup (NW wavefronts) per active cell.
// The cell's n_c proposal slots run in order against the LDS-resident neighbourhood, the
// wavefronts splitting every candidate loop; Displacement slots that would leave the cell are
// rejected.
// ONE: the pool is a single Displacement move (no move selection, no per-move select chains)
// counters of one workgroup, summed over all the cells it visits and added to its counter copy once, at the end
struct CbTotals {
    unsigned long long att1, acc1;   // single-Displacement pools (multi-move pools count per move in LDS, WaveNbh::cnt)
    long long efix;                  // accepted energy change in units of 2^-32
    unsigned long long nnull;
    unsigned long long tests, energies;
    PairCount pcnt;                  // per-wavefront work counters of the current visit
    PMC_TIME(unsigned long long tp[16];)
};

template <int D, int KIND, bool MULTI, int NW, bool ONE, bool IMG, bool BIG = false>
__device__ __forceinline__ void cb_visit(const CbArgs &a, WaveNbh &h, const DevModel *tab, int colour, int aidx,
                                         unsigned long long *st, CbTotals &tot)
{
    const int lane = wave_lane();
    const int tid = team_tid<NW>(), wv = team_wave<NW>();
    constexpr int TS = NW * PMC_WAVE;
    // one wavefront evaluates one slot alone (FP32 pre-filter, survivor queue, dense FP64: wave_delta_e_filtered);
    // teams of NW >= 2 wavefronts evaluate NW slots of the cell speculatively at once and resolve them in order
    constexpr bool BATCH = (NW >= 2);
    unsigned short *const my_queue = h.queue + wv * h.qcap;
    PairCount &pcnt = tot.pcnt;
    unsigned nnull = 0;
    PMC_TIME(
    unsigned long long tp_pro = 0, tp_red = 0, tp_n = 0;
    const unsigned long long tf_base = pcnt.t_filter, td_base = pcnt.t_dense;
    )
    PMC_T(t_0);
    // The instruction arbiter prefers the oldest wavefront of a SIMD, so a workgroup that has just started -- the one whose
    // offsets, flags and coordinates are still to be requested -- issues last.  Raised priority until the slot loop gets
    // those requests out earlier (+1.2 % at N = 10^6, nothing for the teams).
    if (NW == 1) __builtin_amdgcn_s_setprio(3);
    int cell[3] = {0, 0, 0}, lin;
    decode_cell<D>(a, colour, aidx, cell, lin);
    h.lin = lin;
    PMC_T(t_1);
#if defined(PMC_DEBUG_STOP) && PMC_DEBUG_STOP == 1   // profiling builds only: price the visit's phases one by one (not a valid chain)
    { if (a.fused) publish_done(a, lin, tid == 0); return; }
#endif
    // (teams are for small and medium systems, where a visit is one latency chain and LDS is plentiful: the fused staging's
    // count exchange between the wavefronts costs them 2-9 %)
    // teams take the fused staging only in the multi-species instances, and there only where a.fuse_teams says so (dense
    // systems: see pmc_hip.hip, cb_fuse_teams); compiling it into the single-species team kernels as well cost them 20 KB of
    // code and 16-20 B more scratch per lane for a path their systems never take
    constexpr bool FUSE = PMC_FUSED_STAGE && (NW == 1 || MULTI);
    stage_neighbourhood<D, MULTI, NW, FUSE>(a, cell, h, a.fused ? colour : 0);
    PMC_T(t_2);
#if defined(PMC_DEBUG_STOP) && PMC_DEBUG_STOP == 2
    { if (a.fused) publish_done(a, lin, tid == 0); return; }
#endif
    const int nc = h.nc;
    // packed images of the centre-cell particles, lane p <-> particle p (the load travels while the candidates are compacted);
    // a visit of in-box particles only -- every visit of a run that started folded -- never looks at them again
    const int myimg = IMG ? h.myimg : PMC_IMG_ZERO;
    // (work counter of the roofline: candidates this visit loaded from global memory; fire-and-forget, no register lives on)
    if (tid == 0 && h.loaded > 0) atomicAdd(&st[31], (unsigned long long)h.loaded);
    if (nc == 0) { if (a.fused) publish_done(a, lin, tid == 0); return; }
    if (h.ctr + nc > h.cap) {
        if (tid == 0) atomicOr(&st[17], 1ull);
        if (a.fused) publish_done(a, lin, tid == 0);
        return;
    }
    // histogram of the box-filtered neighbourhood sizes, in steps of 64: the host sizes the LDS staging of the next call
    // from it (pmc_hip.hip: cb_cap_sweep)
    if (FUSE && NW == 1 && tid == 0) atomicAdd(&st[PMC_STAT_KEPT + min((h.kept + 63) >> 6, 15)], 1ull);
    compact_candidates<D, MULTI, NW, BIG>(a, h);
    FRegs fr;
    prepare_filter<D, NW, BIG>(a, h, fr);
    PMC_T(t_3);
#if defined(PMC_DEBUG_STOP) && PMC_DEBUG_STOP == 3
    { asm volatile("" :: "v"(fr.x[0]), "v"(fr.y[7]), "v"(fr.z[3])); if (a.fused) publish_done(a, lin, tid == 0); return; }   // (keeps the float copy alive)
#endif
    if (NW == 1) __builtin_amdgcn_s_setprio(0);
#ifdef PMC_DEBUG_COUNT   // profiling builds only
    if (tid == 0) { atomicAdd(&st[20], (unsigned long long)h.ncand); atomicAdd(&st[21], 1ull); atomicAdd(&st[22], (unsigned long long)h.nowrap); atomicAdd(&st[23], 0ull); atomicAdd(&st[28], (unsigned long long)h.compacted); atomicAdd(&st[29], (unsigned long long)h.filt); atomicAdd(&st[30], (unsigned long long)h.ncand * (unsigned long long)h.ncand); }
#endif

    // multi-move pools count per move in LDS (cnt[0..7] attempted, cnt[8..15] accepted); a
    // single-Displacement pool counts in two scalars
    int *cnt = h.cnt;
    unsigned long long att1 = 0, acc1 = 0;
    long long efix = 0;
    bool dirty = false;
    int nbatch = 0;
    const bool single = (a.n[0] == 1);       // the whole periodic box is ONE cell (boxes of 2 ... 4 cutoffs per side): no move leaves it
    const double my_shift = (lane == 0) ? a.shift[0] : ((lane == 1) ? a.shift[1] : a.shift[2]);
    const double my_w = (lane == 0) ? a.w[0] : ((lane == 1) ? a.w[1] : a.w[2]);
    const bool anyimg = IMG && a.simg && ((nc > PMC_WAVE) || (__builtin_amdgcn_ballot_w64(myimg != PMC_IMG_ZERO) != 0ull));
    // the folded position a proposal of centre particle p (image `kp`) is evaluated and stored at (img_settle)
    auto settle = [&](int p, double &xn, double &yn, double &zn) {
        const int kp = (p < PMC_WAVE) ? __builtin_amdgcn_readlane(myimg, p) : a.simg[h.k0 + p];
        if (kp != PMC_IMG_ZERO) {
            xn = img_settle(xn, img_get(kp, 0), a.L[0]);
            yn = img_settle(yn, img_get(kp, 1), a.L[1]);
            if (D == 3) zn = img_settle(zn, img_get(kp, 2), a.L[2]);
        }
    };

    const int nsl = nc * a.rounds;     // proposal slots of this visit
    for (int chunk = 0; chunk < nsl; chunk += PMC_WAVE) {
        // lanes generate the slots of this chunk in parallel
        int t = chunk + lane;
        int g_m = 0;
        uint32_t g_w1 = 0, g_w2 = 0;
        double g_dx = 0, g_dy = 0, g_dz = 0, g_tlnu = 0;
        if (NW >= 4 && chunk == 0) {
            g_m = ONE ? 0 : h.p_m;
            g_tlnu = h.p_tlnu;
            g_w1 = h.p_w1;
            g_w2 = h.p_w2;
            g_dx = h.p_dx;
            g_dy = h.p_dy;
            g_dz = h.p_dz;
        } else if (t < nsl) {
            const int ks = a.rounds * h.k0 + t;
            g_m = ONE ? 0 : (int)a.slot_m[ks];
            g_tlnu = a.slot_tlnu[ks];
            g_w1 = a.slot_w1[ks];
            if (!ONE) g_w2 = a.slot_w2[ks];
            g_dx = a.slot_dx[ks];
            g_dy = a.slot_dy[ks];
            if (D == 3) g_dz = a.slot_dz[ks];
        }
        int nslots = min(PMC_WAVE, nsl - chunk);
#ifdef PMC_DEBUG_NOSLOTS   // profiling builds only: price the staging alone
        nslots = 0;
#endif
        if constexpr (BATCH) {
            // Speculative batches (single-Displacement pools, NW > 1): wavefront w evaluates slot s0 + w
            // ALONE over all candidates against the state before the batch; wavefront 0 then resolves the
            // batch in slot order.  An earlier accepted move i changes one candidate of slot j, so
            //   dE_j' = dE_j - [q(new_j,old_i) - q(old_j,old_i)] + [q(new_j,new_i) - q(old_j,new_i)]
            // (four pair terms, every term and sum exact, hence bit-identical to the sequential
            // evaluation); a slot that re-selects an already moved particle ends the batch and is
            // evaluated again in the next one.  Two barriers per NW slots, no shared reduction.
            int s0 = 0;
            while (s0 < nslots) {
                const int K = min(NW, nslots - s0);
                double *const recs = h.rec + (nbatch & 1) * (NW * PMC_REC);   // records of this batch (two buffers alternate)
                nbatch++;
                PMC_T(tb_0);
                if (wv < K) {
                    const int my = __builtin_amdgcn_readfirstlane(s0 + wv);
                    const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)g_w1, my);
                    const int mm_ = ONE ? 0 : __builtin_amdgcn_readlane(g_m, my);
                    // (read here, with every lane active.  It used to be read inside `if (lane == 0)` below: a cross-lane read of
                    // a lane that is inactive at that point is undefined -- it works while the value sits in a register, and returned
                    // 0 once the compiler reloaded a spilled g_tlnu inside that branch, i.e. for lane 0 only: DESIGN.md 4.6)
                    const double tlnu_my = readlane_double(g_tlnu, my);
                    if (MULTI && !ONE && h.ptab[mm_] != 0) {
                        // DiscreteSwap slot: both particles and the fused four-energy difference by this wavefront alone
                        const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)g_w2, my);
                        const int sa = h.ptab[8 + mm_], sb = h.ptab[16 + mm_];
                        const int na = count_species(h, sa), nb = count_species(h, sb);
                        const bool valid = (na > 0 && nb > 0 && sa != sb);
                        int pi = 0, pj = 0;
                        double de = 0.0;
                        if (valid) {
                            pi = h.ctr + kth_of_species(h, sa, (int)mulhi32(w1, (uint32_t)na));
                            pj = h.ctr + kth_of_species(h, sb, (int)mulhi32(w2, (uint32_t)nb));
                            de = wave_swap_delta<D, KIND, MULTI, BIG>(a, h, tab, my_queue, fr, pi, pj, sa, sb, pcnt);
                        }
                        if (lane == 0) {
                            double *r = recs + wv * PMC_REC;
                            // (the positions of the two particles: what an earlier accepted Displacement of the batch corrects against)
                            r[0] = h.cx[pi]; r[1] = h.cy[pi]; r[2] = (D == 3) ? h.cz[pi] : 0.0;
                            r[3] = h.cx[pj]; r[4] = h.cy[pj]; r[5] = (D == 3) ? h.cz[pj] : 0.0;
                            r[6] = de;
                            r[7] = tlnu_my;
                            r[8] = (double)pi;
                            r[9] = valid ? 1.0 : 0.0;
                            r[10] = (double)pj;
                            r[11] = (double)mm_;
                        }
                    } else {
                    const int pc = h.ctr + (int)mulhi32(w1, (uint32_t)nc);
                    const double xo = h.cx[pc], yo = h.cy[pc], zo = (D == 3) ? h.cz[pc] : 0.0;
                    double xn = xo + readlane_double(g_dx, my);
                    double yn = yo + readlane_double(g_dy, my);
                    double zn = (D == 3) ? zo + readlane_double(g_dz, my) : 0.0;
                    if (anyimg) settle(__builtin_amdgcn_readfirstlane(pc - h.ctr), xn, yn, zn);
                    const bool inside = single || stays_in_cell<D>(a, h, pc - h.ctr, xo, yo, zo, xn, yn, zn, my_shift, my_w);
                    double de = 0.0;
                    if (inside) {
                        const int si = MULTI ? (int)h.csp[pc] : 0;
                        float fxn, fyn, fzn;
                        de = wave_delta_e<D, KIND, MULTI, BIG>(a, h, tab, my_queue, fr, pc, si, xo, yo, zo, xn, yn, zn, fxn, fyn, fzn, pcnt);
                    }
                    if (lane == 0) {
                        double *r = recs + wv * PMC_REC;
                        r[0] = xo; r[1] = yo; r[2] = zo;
                        r[3] = xn; r[4] = yn; r[5] = zn;
                        r[6] = de;
                        r[7] = tlnu_my;
                        r[8] = (double)pc;
                        r[9] = inside ? 1.0 : 0.0;
                        r[10] = -1.0;
                        r[11] = (double)mm_;
                    }
                    }
                }
                PMC_T(tb_1);
                __syncthreads();           // every evaluation has read the pre-batch state; records are posted
                PMC_T(tb_2);
                if (wv == 0) {
                    // this lane's record (lanes 0..K-1)
                    const int jl = min(lane, NW - 1);
                    const double *rl = recs + jl * PMC_REC;
                    const double l_de = rl[6], l_tlnu = rl[7];
                    const int l_pc = (int)rl[8];
                    const bool l_in = rl[9] != 0.0;
                    const int l_pj = (int)rl[10];
                    const int l_m = ONE ? 0 : (int)rl[11];
                    const bool l_swap = !ONE && MULTI && (l_pj >= 0);
                    // pair terms, lane-parallel: lane-term = 4 * pair(i,j) + t,  pair(i,j) = j (j - 1) / 2 + i,  i < j < K;
                    // 16 pairs per pass; the quad sums c(i,j) go to an NW x NW matrix in LDS
                    constexpr int NP = NW * (NW - 1) / 2;
#pragma unroll
                    for (int p0 = 0; p0 < NP; p0 += 16) {
                        const int pr = p0 + (lane >> 2), tt = lane & 3;
                        const int pj = 1 + (pr >= 1) + (pr >= 3) + (pr >= 6) + (pr >= 10) + (pr >= 15) + (pr >= 21);
                        const int pi = pr - ((pj * (pj - 1)) >> 1);
                        const int pjc = min(pj, NW - 1);
                        double v = 0.0;
                        // record[10] >= 0 marks a DiscreteSwap slot (its second particle)
                        const bool i_disp = ONE || recs[pi * PMC_REC + 10] < 0.0, j_disp = ONE || recs[pjc * PMC_REC + 10] < 0.0;
                        const bool both_disp = i_disp && j_disp;
                        if (!ONE && MULTI && pr < NP && pj < K && i_disp && !j_disp) {
                            // Displacement i before DiscreteSwap j (particles A = rj[8] <-> B = rj[10], species sa <-> sb): the swap was
                            // evaluated with particle i at its old position.  One lane per (position of i, particle of the swap):
                            // the pair energy with the species that particle takes minus the one with the species it has;
                            // + at i's new position, - at its old one.
                            const double *ri = recs + pi * PMC_REC, *rj = recs + pj * PMC_REC;
                            const int oi = (tt & 2) ? 0 : 3, oj = (tt & 1) ? 3 : 0;     // i: old or new; swap: particle B or A
                            const double xi_ = ri[oi], yi_ = ri[oi + 1], zi_ = ri[oi + 2];
                            const double xj_ = rj[oj], yj_ = rj[oj + 1], zj_ = rj[oj + 2];
                            const int sci = (int)h.csp[(int)ri[8]];
                            const int mj = (int)rj[11];
                            const int sa_ = h.ptab[8 + mj], sb_ = h.ptab[16 + mj];
                            const DevModel &mplus = tab[((tt & 1) ? sa_ : sb_) * a.S + sci], &mminus = tab[((tt & 1) ? sb_ : sa_) * a.S + sci];
                            const double r2 = h.nowrap ? dist2_nowrap<D>(xj_, yj_, zj_, xi_, yi_, zi_) : dist2_fast<D>(a, xj_, yj_, zj_, xi_, yi_, zi_);
                            if (!(r2 > mplus.rcut2)) v += dev_quant(pot_kind<KIND>(r2, mplus));
                            if (!(r2 > mminus.rcut2)) v -= dev_quant(pot_kind<KIND>(r2, mminus));
                            if (tt & 2) v = -v;
                        }
                        if (pr < NP && pj < K && both_disp) {
                            const double *ri = recs + pi * PMC_REC, *rj = recs + pj * PMC_REC;
                            const int oi = (tt & 2) ? 0 : 3, oj = (tt & 1) ? 0 : 3;   // old or new coordinates
                            const double xi_ = ri[oi], yi_ = ri[oi + 1], zi_ = ri[oi + 2];
                            const double xj_ = rj[oj], yj_ = rj[oj + 1], zj_ = rj[oj + 2];
                            const int sci = MULTI ? (int)h.csp[(int)ri[8]] : 0, scj = MULTI ? (int)h.csp[(int)rj[8]] : 0;
                            const DevModel &mm = MULTI ? tab[scj * a.S + sci] : a.m0;
                            const double r2 = h.nowrap ? dist2_nowrap<D>(xj_, yj_, zj_, xi_, yi_, zi_) : dist2_fast<D>(a, xj_, yj_, zj_, xi_, yi_, zi_);
                            if (!(r2 > mm.rcut2)) v = dev_quant(pot_kind<KIND>(r2, mm));
                            if (tt == 1 || tt == 2) v = -v;
                        }
                        v = dpp_add<0xB1>(v);
                        v = dpp_add<0x4E>(v);      // every lane of a quad holds the quad's (exact) sum
                        if (tt == 0 && pr < NP) h.cmat[pi * NW + pjc] = v;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
// Two forms of the in-order resolution below.  The first is round 2's (cross-lane reads of every operand, ~40 instructions per
// slot); the second lets every lane decide for itself (~8 per slot, +3 % for teams of 8).
                    // lane j holds column j of the matrix: the corrections it receives from earlier accepted slots i (none from i >= j)
                    double creg[NW - 1];
#pragma unroll
                    for (int i = 0; i < NW - 1; i++) {
                        const double c = h.cmat[i * NW + jl];
                        creg[i] = (lane > i) ? c : 0.0;
                    }
                    // samem, bit i: slot i (before this lane's slot) is a Displacement of the particle this slot uses -- if slot i is
                    // accepted, this slot was evaluated against a stale position: it ends the batch and is evaluated again (a swap after
                    // accepted Displacements of OTHER particles is corrected like a Displacement: c(i,j) above)
                    unsigned samem = 0;
#pragma unroll
                    for (int i = 0; i < NW - 1; i++) {
                        const int pci = __builtin_amdgcn_readlane(l_pc, i);
                        const bool dispi = ONE || !MULTI || __builtin_amdgcn_readlane(l_pj, i) < 0;
                        if (dispi && lane > i && ((l_pc == pci) || (l_swap && l_pj == pci))) samem |= 1u << i;
                    }
                    const unsigned long long inmask = __builtin_amdgcn_ballot_w64(l_in), swapmask = __builtin_amdgcn_ballot_w64(l_swap);
                    // decisions in slot order, every lane deciding for itself (lane i's answer is the one read at step i); lanes > i
                    // accumulate what an accepted slot i does to them.  The loop is a chain of ~8 instructions per slot on a CU where
                    // this wavefront runs alone: it used to be ~40 (cross-lane reads of every operand, a float-to-fixed conversion per
                    // accepted slot), a quarter of a team's time per batch.
                    double corr = 0.0;
                    int nres = 0;
                    unsigned accmask = 0;
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        if (i >= K) break;
                        const unsigned long long confm = __builtin_amdgcn_ballot_w64((samem & accmask) != 0u);
                        if ((confm >> i) & 1ull) break;
                        const unsigned long long okm = __builtin_amdgcn_ballot_w64(l_tlnu < -(l_de + corr)) & inmask;
                        nres = i + 1;
                        if ((okm >> i) & 1ull) {
                            accmask |= 1u << i;
                            if ((swapmask >> i) & 1ull) break;            // later slots saw the old species: next batch
                            if (i < NW - 1) corr += creg[i < NW - 1 ? i : 0];
                        }
                    }
                    {
                        // (lane i still holds the dE it decided with: it only ever received corrections from slots before it)
                        const bool accl = lane < NW && ((accmask >> lane) & 1u);
                        double dsum = accl ? (l_de + corr) : 0.0;
                        dsum = dpp_add<0xB1>(dsum);
                        dsum = dpp_add<0x4E>(dsum);
                        dsum = dpp_add<0x141>(dsum);          // lanes 0..7 hold the (exact) sum over the accepted slots
                        efix += (long long)(readlane_double(dsum, 0) * PMC_FIX);
                        acc1 += (unsigned long long)__popc(accmask);
                    }
                    att1 += (unsigned long long)nres;
#ifdef PMC_DEBUG_SLOTLOG
                    if (g_slotlog && lane < nres) {
                        double *q = g_slotlog + 8 * (size_t)(a.rounds * h.k0 + chunk + s0 + lane);
                        q[0] = l_de; q[1] = corr; q[2] = l_tlnu; q[3] = (double)l_pc; q[4] = l_in ? 1.0 : 0.0;
                        q[5] = (double)((accmask >> lane) & 1u); q[6] = rl[0]; q[7] = rl[3];
                    }
#endif
                    if (!ONE && lane < nres) {              // per-move counters of multi-move pools (this wavefront only)
                        atomicAdd(&cnt[l_m], 1);
                        if ((accmask >> lane) & 1u) atomicAdd(&cnt[8 + l_m], 1);
                    }
                    if (lane < NW && ((accmask >> lane) & 1u)) {
                        if (l_swap) {
                            const unsigned char si_ = h.csp[l_pc], sj_ = h.csp[l_pj];
                            h.csp[l_pc] = sj_;
                            h.csp[l_pj] = si_;
                        } else {
                            h.cx[l_pc] = rl[3];
                            h.cy[l_pc] = rl[4];
                            if (D == 3) h.cz[l_pc] = rl[5];
                            if (BIG && h.fstream) {         // the team's float copy in LDS (prepare_filter) follows the move
                                if (h.nowrap) {
                                    h.fx[l_pc] = rel_float<true>(rl[3], h.bc[0], a.L[0], a.invL[0]);
                                    h.fy[l_pc] = rel_float<true>(rl[4], h.bc[1], a.L[1], a.invL[1]);
                                    if (D == 3) h.fz[l_pc] = rel_float<true>(rl[5], h.bc[2], a.L[2], a.invL[2]);
                                } else {
                                    h.fx[l_pc] = rel_float<false>(rl[3], h.bc[0], a.L[0], a.invL[0]);
                                    h.fy[l_pc] = rel_float<false>(rl[4], h.bc[1], a.L[1], a.invL[1]);
                                    if (D == 3) h.fz[l_pc] = rel_float<false>(rl[5], h.bc[2], a.L[2], a.invL[2]);
                                }
                            }
                        }
                    }
                    nnull += (unsigned)__popcll(__ballot(lane < nres && !l_in));
                    if (lane == 0) { h.flag[3] = nres; h.flag[1] = (int)accmask; }
                }
                PMC_T(tb_3);
                __syncthreads();           // accepted moves and the number of resolved slots are visible
    PMC_TIME({ PMC_T(tb_4); tp_pro += tb_1 - tb_0; tp_red += tb_3 - tb_2; tp_n += 1; pcnt.t_filter += tb_2 - tb_1; pcnt.t_dense += tb_4 - tb_3; })
                s0 += h.flag[3];
                if (h.filt) {
                    // every wavefront brings its register copy of the float coordinates up to date with the accepted
                    // Displacements of this batch (the records stay intact: the next batch writes the other buffer)
                    const unsigned am = (unsigned)h.flag[1];
#pragma unroll
                    for (int i = 0; i < NW; i++) {
                        if ((am >> i) & 1u) {
                            const double *ri = recs + i * PMC_REC;
                            if (ONE || !MULTI || ri[10] < 0.0) {
                                const int pci = (int)ri[8];
                                float gx, gy, gz = 0.0f;
                                if (h.nowrap) {
                                    gx = rel_float<true>(ri[3], h.bc[0], a.L[0], a.invL[0]);
                                    gy = rel_float<true>(ri[4], h.bc[1], a.L[1], a.invL[1]);
                                    if (D == 3) gz = rel_float<true>(ri[5], h.bc[2], a.L[2], a.invL[2]);
                                } else {
                                    gx = rel_float<false>(ri[3], h.bc[0], a.L[0], a.invL[0]);
                                    gy = rel_float<false>(ri[4], h.bc[1], a.L[1], a.invL[1]);
                                    if (D == 3) gz = rel_float<false>(ri[5], h.bc[2], a.L[2], a.invL[2]);
                                }
                                update_fregs<D>(fr, pci, gx, gy, gz);
                            }
                        }
                    }
                }
            }
            dirty = true;
        } else
        for (int s = 0; s < nslots; s++) {
            PMC_T(ts_0);
            const int m = ONE ? 0 : __builtin_amdgcn_readlane(g_m, s);
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)g_w1, s);
            const double tlnu = readlane_double(g_tlnu, s);     // T * ln u: accept iff T ln u < -dE
            bool ok = false;
            if (ONE || h.ptab[m] == 0) {
                const int p = (int)mulhi32(w1, (uint32_t)nc);
                const int pc = h.ctr + p;
                const double xo = h.cx[pc], yo = h.cy[pc], zo = (D == 3) ? h.cz[pc] : 0.0;
                double xn = xo + readlane_double(g_dx, s);
                double yn = yo + readlane_double(g_dy, s);
                double zn = (D == 3) ? zo + readlane_double(g_dz, s) : 0.0;
                if (anyimg) settle(__builtin_amdgcn_readfirstlane(p), xn, yn, zn);
                const bool inside = single || stays_in_cell<D>(a, h, p, xo, yo, zo, xn, yn, zn, my_shift, my_w);
                if (inside) {
                    // one wavefront per cell here (teams of NW >= 2 take the batch branch above)
                    const int si = MULTI ? (int)h.csp[pc] : 0;
                    float fxn = 0.0f, fyn = 0.0f, fzn = 0.0f;
    PMC_TIME(
                    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 0" :: "v"(xn + yn + zn) : "memory");
                    PMC_T(ts_1);
                    tp_pro += ts_1 - ts_0;
                    tp_n += 1;
    )
                    const double de = wave_delta_e<D, KIND, MULTI>(a, h, tab, my_queue, fr, pc, si, xo, yo, zo, xn, yn, zn, fxn, fyn, fzn, pcnt);
                    PMC_T(ts_2);
                    ok = tlnu < -de;
                    if (ok) {
                        __syncthreads();
                        if (tid == 0) {
                            h.cx[pc] = xn;
                            h.cy[pc] = yn;
                            if (D == 3) h.cz[pc] = zn;
                        }
                        __syncthreads();
                        if (h.filt) update_fregs<D>(fr, pc, fxn, fyn, fzn, h.compacted && nc <= PMC_WAVE);
                        efix += (long long)(de * PMC_FIX);
                        dirty = true;
                    }
    PMC_TIME({ PMC_T(ts_3); tp_red += ts_3 - ts_2; })
                } else {
                    nnull += 1;
                }
            } else if (MULTI && !ONE) {
                const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)g_w2, s);
                const int sa = h.ptab[8 + m], sb = h.ptab[16 + m];
                const int na = count_species(h, sa), nb = count_species(h, sb);
                if (na > 0 && nb > 0 && sa != sb) {
                    const int pi = h.ctr + kth_of_species(h, sa, (int)mulhi32(w1, (uint32_t)na));
                    const int pj = h.ctr + kth_of_species(h, sb, (int)mulhi32(w2, (uint32_t)nb));
                    const double de = wave_swap_delta<D, KIND, MULTI>(a, h, tab, my_queue, fr, pi, pj, sa, sb, pcnt);
                    ok = tlnu < -de;
                    if (ok) {
                        __syncthreads();
                        if (tid == 0) {
                            h.csp[pi] = (unsigned char)sb;
                            h.csp[pj] = (unsigned char)sa;
                        }
                        __syncthreads();
                        efix += (long long)(de * PMC_FIX);
                        dirty = true;
                    }
                } else {
                    nnull += 1;
                }
            }
            if (ONE) {
                att1 += 1;
                acc1 += ok ? 1 : 0;
            } else {
                if (tid == 0) {
                    cnt[m] += 1;
                    if (ok) cnt[8 + m] += 1;
                }
            }
        }
    }
    PMC_T(t_4);
    if (dirty) {
        if (tid == 0) atomicAdd(&st[19], (unsigned long long)nc);
        for (int k = tid; k < nc; k += TS) {
            const int kc = h.ctr + k;
            const int id = a.ids[h.k0 + k];
            // the position array through the particle's image: x = xf + k L (k = 0: x = xf, the reference's x + delta).  Folding is
            // not exact for every x (adding L to a small negative number rounds), so a particle with an image is only written
            // if it has moved -- its folded coordinates differ from the cell-sorted copy this visit staged; visits of in-box
            // particles (every visit of a run that started folded) never take that test
            const double nx = h.cx[kc], ny = h.cy[kc], nz = (D == 3) ? h.cz[kc] : 0.0;
            const int im = !anyimg ? PMC_IMG_ZERO : ((k < PMC_WAVE && NW == 1) ? myimg : a.simg[h.k0 + k]);
            bool moved = true;
            if (im != PMC_IMG_ZERO) {
                const double ox = ld_agent(a.sx + h.k0 + k), oy = ld_agent(a.sy + h.k0 + k), oz = (D == 3) ? ld_agent(a.sz + h.k0 + k) : 0.0;
                moved = !(nx == ox && ny == oy && nz == oz);
            }
            if (moved) {
                st_agent(a.sx + h.k0 + k, nx);
                st_agent(a.sy + h.k0 + k, ny);
                a.pos[(size_t)id * D + 0] = img_unfold(nx, img_get(im, 0), a.L[0]);
                a.pos[(size_t)id * D + 1] = img_unfold(ny, img_get(im, 1), a.L[1]);
                if (D == 3) {
                    st_agent(a.sz + h.k0 + k, nz);
                    a.pos[(size_t)id * D + 2] = img_unfold(nz, img_get(im, 2), a.L[2]);
                }
            }
            if (MULTI) {
                st_agent(a.ssp + h.k0 + k, h.csp[kc]);
                a.species[id] = (int)h.csp[kc];
            }
        }
    }
    if (a.fused) publish_done(a, lin, tid == 0);
    tot.att1 += att1;
    tot.acc1 += acc1;
    tot.efix += efix;
    tot.nnull += nnull;
#ifdef PMC_DEBUG_TIME
    {
        PMC_T(t_5);
        tot.tp[0] += t_1 - t_0; tot.tp[1] += t_2 - t_1; tot.tp[2] += t_3 - t_2; tot.tp[3] += t_4 - t_3; tot.tp[4] += t_5 - t_4;
        tot.tp[5] += tp_pro; tot.tp[6] += pcnt.t_filter - tf_base; tot.tp[7] += pcnt.t_dense - td_base; tot.tp[8] += tp_red;
        tot.tp[9] += tp_n; tot.tp[15] += 1;
#if PMC_DEBUG_TIME == 2   // the slot-phase lines show the inside of stage_compacted instead: box, image test, box test, LDS writes (then sync + cfl)
        tot.tp[5] = tot.tp[5] - tp_pro + (h.ts[5] - h.ts[3]); tot.tp[6] = tot.tp[6] - (pcnt.t_filter - tf_base) + (h.ts[6] - h.ts[5]);
        tot.tp[7] = tot.tp[7] - (pcnt.t_dense - td_base) + (h.ts[7] - h.ts[6]); tot.tp[8] = tot.tp[8] - tp_red + (h.ts[8] - h.ts[7]);
#endif
        tot.tp[10] += h.ts[1] - h.ts[0]; tot.tp[11] += h.ts[2] - h.ts[1]; tot.tp[12] += h.ts[3] - h.ts[2]; tot.tp[13] += h.ts[4] - h.ts[3];
    }
#endif
}

// The Metropolis sweep kernel: one workgroup of NW wavefronts per cell visit (cb_visit).
//  * separate colour launches (a.fused == 0): block b visits one cell of colour a.colour;
//  * fused sweep (a.fused == 1): all 2^d colours in one launch; a workgroup takes a ticket (atomicAdd, so start order =
//    colour order whatever the dispatch order), ticket t <-> (colour t / nactive, cell t mod nactive).  A cell waits
//    (inside stage_neighbourhood) for the completion flags of its lower-colour neighbours: those have lower tickets,
//    i.e. are held by workgroups that are resident or finished: no deadlock.
//    (Persistent workgroups that loop over tickets were measured twice and rejected: the loop around the visit takes the
//    kernel from 84 to 188 VGPRs -- or 128 with scratch spills -- and chunks of tickets let workgroups run ahead by more
//    than one colour, so cells stall on their neighbours: 1.65 ms against 1.41 ms at N = 10^6.)
// ONE: the pool is a single Displacement move (no move selection, no per-move select chains)
// BIG (teams of eight, systems without periodic images): neighbourhoods beyond the 512 candidates of the register copy -- boxes of
// 4-5 cutoffs per side at liquid density -- are staged whole (up to PMC_CAP_BIG), with one float copy in LDS that every
// wavefront's pre-filter streams (prepare_filter, wave_delta_e_filtered<STREAM>; host: cb_big).  Separate instances: compiled
// into the default team kernels the extra state cost configs 2 and 3 (N = 32768, Kob-Andersen 10^5) 3-6 %.
template <int D, int KIND, bool MULTI, int NW, bool ONE, int MW = PMC_CB_MINWAVES, bool IMG = false, bool BIG = false>   // MW: wavefronts per SIMD it is compiled for
__global__ __launch_bounds__(NW * 64, MW) void k_cb_pass(const CbArgs *__restrict__ args, int nrep)
{
    // The arguments of a launch live in DEVICE memory, one CbArgs per replica of the launch (one for a single system,
    // n for pmc_sweep_many), written once per call for all its sweeps.  They are only ever read through references:
    // a by-value CbArgs that is modified -- or handed to a function the compiler declines to inline -- is copied to
    // scratch, and both times that happened (r02) the kernel produced wrong energies (the copy put more values into scratch,
    // and a spilled value was then read across lanes inside a lane-0 branch: the bug found in round 3).
    extern __shared__ __align__(16) unsigned char smem[];
    WaveNbh h;
    const CbArgs &a0 = args[0];                      // the shape (grid, models, pool, capacities) is common to all replicas
    const DevModel *tab = setup_lds<D, MULTI, (NW >= 2 ? NW : 0), BIG>(a0, smem, h);
    const int tid = team_tid<NW>();
    CbTotals tot{};
    tot.pcnt.count_dense = (NW == 1 && !MULTI) || BIG;
    if (!ONE) {
        if (tid < 16) h.cnt[tid] = 0;
        if (tid < 8) {
            h.ptab[tid] = a0.pool.kind[tid];
            h.ptab[8 + tid] = a0.pool.sp0[tid];
            h.ptab[16 + tid] = a0.pool.sp1[tid];
        }
        __syncthreads();
    }
    int aidx, colour = a0.colour, r = 0;
    if (a0.fused) {
        // ticket -> (colour, replica, cell), colour slowest: a cell's lower-colour neighbours -- cells of the SAME replica --
        // hold lower tickets, i.e. belong to workgroups that are resident or finished: no deadlock
        if (tid == 0) {
            if (a0.ticket_lanes == PMC_TICKET_LANES) {
                const int x = (int)(blockIdx.x & (PMC_TICKET_LANES - 1));
                h.flag[2] = (int)atomicAdd(a0.ticket + x * PMC_TICKET_STRIDE, 1ull) * PMC_TICKET_LANES + x;
            } else {
                h.flag[2] = (int)(atomicAdd(a0.ticket, 1ull) - a0.ticket_base);
            }
        }
        __syncthreads();
        const int t = h.flag[2];
        const int per_colour = nrep * a0.nactive;
        colour = t / per_colour;
        const int rem = t - colour * per_colour;
        r = __builtin_amdgcn_readfirstlane(rem / a0.nactive);
        aidx = rem - r * a0.nactive;
    } else {
        aidx = a0.order ? (int)blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);
        if (aidx >= a0.nactive) return;
    }
    const CbArgs &a = args[r];
    unsigned long long *const st = a.stats + (size_t)(aidx & (PMC_NSHARD - 1)) * PMC_NSTATS;   // this workgroup's counter copy
    cb_visit<D, KIND, MULTI, NW, ONE, IMG, BIG>(a, h, tab, colour, aidx, st, tot);
    if (tid == 0) {
        if (ONE) {
            if (tot.att1) atomicAdd(&st[0], tot.att1);
            if (tot.acc1) atomicAdd(&st[8], tot.acc1);
        } else {
            for (int q = 0; q < a.pool.npool; q++) {
                if (h.cnt[q]) atomicAdd(&st[q], (unsigned long long)h.cnt[q]);
                if (h.cnt[8 + q]) atomicAdd(&st[8 + q], (unsigned long long)h.cnt[8 + q]);
            }
        }
        if (tot.efix) atomicAdd(&st[16], (unsigned long long)tot.efix);
        if (tot.nnull) atomicAdd(&st[27], tot.nnull);
    PMC_TIME(
        for (int q = 0; q < 15; q++) atomicAdd(&st[32 + q], tot.tp[q]);
        atomicAdd(&st[47], tot.tp[15]);
    )
    }
    // work counters of the roofline (bench.py): every wavefront adds its own
    if (wave_lane() == 0 && tot.pcnt.tests) {
        atomicAdd(&st[25], (unsigned long long)tot.pcnt.tests);
        atomicAdd(&st[26], (unsigned long long)tot.pcnt.energies);
        if ((NW == 1 && !MULTI) || BIG) atomicAdd(&st[24], (unsigned long long)tot.pcnt.dense);
    }
}

// independent batch of Displacement dE (exact sums): one wavefront per proposal,
// candidates streamed straight from the CSR arrays of the cell of the old and of
// the new position.
template <int D>
__device__ double stream_energy(const CbArgs &a, double xi, double yi, double zi, int si, int self_id)
{
    constexpr int NN = (D == 3) ? 27 : 9;
    const double x[3] = {xi, yi, zi};
    int cell[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) {
        long long c = (long long)mc_coord<D>(x[q], a.shift[q], a.w[q]);
        int r = (int)(c % a.n[q]);
        if (r < 0) r += a.n[q];
        cell[q] = r;
    }
    double e = 0.0;
    for (int o = 0; o < NN; o++) {
        int lin2 = 0, r = o;
        bool dup = false;              // (single-cell boxes: stage_neighbourhood)
#pragma unroll
        for (int q = 0; q < D; q++) {
            const int off = (r % 3) - 1;
            int cq = cell[q] + off;
            r /= 3;
            dup = dup || (a.n[q] == 1 && off != 0);
            if (cq < 0) cq += a.n[q];
            if (cq >= a.n[q]) cq -= a.n[q];
            lin2 = lin2 * a.n[q] + cq;
        }
        if (dup) continue;
        int lo = a.start[lin2], hi = a.start[lin2 + 1];
        for (int k = lo + wave_lane(); k < hi; k += PMC_WAVE) {
            if (a.ids[k] == self_id) continue;
            const DevModel &m = a.models[si * a.S + (int)a.ssp[k]];
            double r2 = dist2_fast<D>(a, xi, yi, zi, a.sx[k], a.sy[k], (D == 3) ? a.sz[k] : 0.0);
            if (!(r2 > m.rcut2)) e += dev_quant(dev_potential_dyn(r2, m));
        }
    }
    return wave_sum_exact(e);
}

template <int D>
__global__ __launch_bounds__(64) void k_delta_e(CbArgs a, int n, const int *__restrict__ which,
                                                const double *__restrict__ delta, double *__restrict__ e1,
                                                double *__restrict__ e2)
{
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        int i = which[b];
        double x[3] = {0, 0, 0}, xn[3] = {0, 0, 0};
        const int im = a.img ? a.img[i] : PMC_IMG_ZERO;
#pragma unroll
        for (int q = 0; q < D; q++) {
            // (folded coordinates, as the sweep kernel evaluates a Displacement of particle i: img_fold, img_settle)
            x[q] = img_fold(a.pos[(size_t)i * D + q], img_get(im, q), a.L[q]);
            xn[q] = img_settle(x[q] + delta[(size_t)b * D + q], img_get(im, q), a.L[q]);
        }
        int si = a.species[i];
        double ea = stream_energy<D>(a, x[0], x[1], x[2], si, i);
        double eb = stream_energy<D>(a, xn[0], xn[1], xn[2], si, i);
        if (wave_lane() == 0) { e1[b] = ea; e2[b] = eb; }
    }
}

// Sum the counter copies 1 .. PMC_NSHARD-1 into copy 0 and clear them: one block per counter, one thread per copy
// (counter 17 holds error flags: OR).  Runs once per pmc_sweep call, after its last sweep.
PMC_GLOBAL __launch_bounds__(PMC_NSHARD) void k_stats_fold(unsigned long long *__restrict__ stats)
{
    __shared__ unsigned long long red[PMC_NSHARD / 64];
    const int idx = blockIdx.x, t = threadIdx.x;
    unsigned long long v = 0;
    if (t > 0) {
        v = stats[(size_t)t * PMC_NSTATS + idx];
        if (v) stats[(size_t)t * PMC_NSTATS + idx] = 0;
    }
    const bool flags = (idx == 17);
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) {
        const unsigned long long o = __shfl_xor(v, sft, 64);
        v = flags ? (v | o) : (v + o);
    }
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    if (t == 0) {
        unsigned long long tot = stats[idx];
        for (int w = 0; w < PMC_NSHARD / 64; w++) tot = flags ? (tot | red[w]) : (tot + red[w]);
        stats[idx] = tot;
    }
}

// test hook: the guard-free division against the IEEE one
PMC_GLOBAL void k_debug_div(int n, const double *__restrict__ num, const double *__restrict__ den,
                            double *__restrict__ q_ieee, double *__restrict__ q_fast)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    q_ieee[i] = num[i] / den[i];
    q_fast[i] = dev_div_inrange(num[i], den[i]);
}

PMC_GLOBAL void k_debug_rng(int n, const uint32_t *__restrict__ w1, const uint32_t *__restrict__ w2,
                            double *__restrict__ z0, double *__restrict__ z1, double *__restrict__ lnu)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    box_muller(w1[i], w2[i], &a, &b);
    z0[i] = a;
    z1[i] = b;
    lnu[i] = det_log(uniform53(w1[i], w2[i]));
}
