// pmc_ref.h -- the reference-order engine of the ParticlesMC hot path on gfx950: device-resident head / list linked cells
// exactly as src/neighbours.jl:224-302, per-particle energies summed in the iterator order of the list type
// (src/neighbours.jl:332-377, 209-217, 422-424, 42-44), the per-move verbs and the literal one-wavefront chain (k_ref_ops),
// and the EnergyBias policy (src/moves.jl:246-280).  Included by pmc_kernels.h (which defines PMC_GLOBAL, PMC_REF_IDS and the
// binning kernels these build on); the fast forms of the same chain are in pmc_seq.h.
#pragma once
// head/list exactly as build_neighbour_list! leaves them (src/neighbours.jl:259-267):
// every chain runs through the cell's particles in DESCENDING id.
PMC_GLOBAL void k_ref_links(const int *__restrict__ cell_of, int N, const int *__restrict__ start,
                            const int *__restrict__ ids, int ncell, int *__restrict__ head,
                            int *__restrict__ list, int *__restrict__ ccount)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < N) {
        int c = cell_of[t];
        // position of t in its (ascending) cell segment
        int lo = start[c], hi = start[c + 1];
        int k = lo;
        while (k < hi && ids[k] != t) k++;
        list[t] = (k > lo) ? ids[k - 1] : -1;
    }
    if (t < ncell) {
        int lo = start[t], hi = start[t + 1];
        head[t] = (hi > lo) ? ids[hi - 1] : -1;
        ccount[t] = hi - lo;
    }
}

// ------------------------------------------------------------------------------
// reference-order engine
// ------------------------------------------------------------------------------

struct RefArgs {
    int N, S;
    double box[3], cell[3];
    int nc[3];
    int nnmax;                 // 3^d
    double *pos;               // AoS
    int *species;              // 0-based
    const DevModel *models;
    int *cs, *head, *list, *ccount;
    const int *nbr, *nbrcnt;   // neighbour-cell table, reference order
    int *sp_off, *sp_ids, *sp_heads; // SpeciesList (utils.jl:31-49), 0-based
    double *energy;            // system.energy[1]
    double temperature;
    // alternative list types of the reference (src/neighbours.jl): 0 EmptyList, 1 LinkedList, 2 CellList, 3 VerletList
    int list_type;
    int *cl_items;             // cells[c] vectors (CellList / VerletList): [ncell][cl_cap], lengths in ccount
    int cl_cap;
    int *vl_nb, *vl_cnt;       // neighbours[i] vectors (VerletList): [N][vl_cap]
    int vl_cap;
    double *vl_last;           // positions_at_last_update
    double dr2o4, rcutoff2;
    int *err;                  // bit 0: a cell vector overflowed, bit 1: a Verlet vector overflowed
};

template <int D>
__device__ __forceinline__ int ref_cell_of(const RefArgs &a, const double *x)
{
    long long c = 0, stride = 1;
#pragma unroll
    for (int q = D - 1; q >= 0; q--) {
        long long mc = (long long)dev_jl_fld(x[q], a.cell[q]);
        long long f = mc - dev_fld_i(mc, a.nc[q]) * a.nc[q];
        c += f * stride;
        stride *= a.nc[q];
    }
    return (int)c;
}

// The literal chain is ONE wavefront that owns its system for the length of a launch: what a lane stores only ever has to reach
// the wavefront's own later loads, which a workgroup barrier (with the waits it implies) orders.  Until round 5 every such point
// was a device-scope __threadfence() -- a write-back of the XCD's L2 and an invalidate of the CU's L1, ~3.5 us each
// (MI355X_MICROARCH.md), six to ten per step -- for visibility nobody else needed.
__device__ __forceinline__ void ref_sync() { __syncthreads(); }

// compute_energy_ij, src/atoms.jl:66-75 (literal, division-form minimum image)
template <int D>
__device__ __forceinline__ double ref_energy_ij(const RefArgs &a, int i, int j, const double *xi, int spi)
{
    if (i == j) return 0.0;
    const DevModel &m = a.models[spi * a.S + a.species[j]];
    double v = dev_vector_1d(xi[0], a.pos[(size_t)j * D + 0], a.box[0]);
    double r2 = v * v;
#pragma unroll
    for (int q = 1; q < D; q++) {
        v = dev_vector_1d(xi[q], a.pos[(size_t)j * D + q], a.box[q]);
        r2 = r2 + v * v;
    }
    if (r2 > m.rcut2) return 0.0;
    return dev_potential_dyn(r2, m);
}

// nearest_image_distance_squared, src/utils.jl:21-28 (division form)
template <int D>
__device__ __forceinline__ double ref_dist2(const RefArgs &a, const double *xi, int j)
{
    double v = dev_vector_1d(xi[0], a.pos[(size_t)j * D + 0], a.box[0]);
    double r2 = v * v;
#pragma unroll
    for (int q = 1; q < D; q++) {
        v = dev_vector_1d(xi[q], a.pos[(size_t)j * D + q], a.box[q]);
        r2 = r2 + v * v;
    }
    return r2;
}

__device__ __forceinline__ double readlane_double(double v, int l)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// Ordered FP64 sums through LDS.  The chain acc += u_k must run in iterator order, one rounding per term; terms that are 0 can
// be left out.  Lanes deposit their non-zero terms, in lane order, behind the terms already there (sq_terms_push); the sum is
// then one chain of dependent additions fed by LDS reads that do not depend on it (sq_terms_sum): ~10 cycles per term, where
// a ballot / readlane loop over the lanes of a register takes ~100.  Buffers hold SQ_TERMS terms and are summed down whenever
// the next 64 might not fit, which keeps the order.
#define SQ_TERMS 192
#define SQ_TERMS_PAD 8
#define SQ_TERMS_BUF (SQ_TERMS + 64 + SQ_TERMS_PAD)     // doubles of one buffer

__device__ __forceinline__ double sq_terms_sum(double *buf, int n, double acc)
{
    const int lane = wave_lane();
    if (lane < SQ_TERMS_PAD) buf[n + lane] = 0.0;               // (+0.0 leaves the sum as it is: the loop runs in blocks of 8)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the terms are in LDS
    for (int k = 0; k < n; k += SQ_TERMS_PAD) {
        const double t0 = buf[k], t1 = buf[k + 1], t2 = buf[k + 2], t3 = buf[k + 3];
        const double t4 = buf[k + 4], t5 = buf[k + 5], t6 = buf[k + 6], t7 = buf[k + 7];
        acc += t0; acc += t1; acc += t2; acc += t3; acc += t4; acc += t5; acc += t6; acc += t7;
    }
    return acc;
}

// two sums at once: their chains are independent and interleave
__device__ __forceinline__ void sq_terms_sum2(double *bufa, int na, double &acca, double *bufb, int nb, double &accb)
{
    const int lane = wave_lane();
    if (lane < SQ_TERMS_PAD) { bufa[na + lane] = 0.0; bufb[nb + lane] = 0.0; }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    double a = acca, b = accb;
    const int n = max(na, nb);
    // (both buffers are SQ_TERMS + 64 + 8 long: reading one past its own terms is safe, and masked)
    for (int k = 0; k < n; k += SQ_TERMS_PAD) {
        double ta[SQ_TERMS_PAD], tb[SQ_TERMS_PAD];
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) { ta[q] = bufa[k + q]; tb[q] = bufb[k + q]; }     // sixteen reads in flight
        const bool ina = k < na, inb = k < nb;              // (a block that starts inside the terms ends inside their zero padding)
#pragma unroll
        for (int q = 0; q < SQ_TERMS_PAD; q++) {
            a += ina ? ta[q] : 0.0;
            b += inb ? tb[q] : 0.0;
        }
    }
    acca = a;
    accb = b;
}

__device__ __forceinline__ void sq_terms_push(double *buf, int &n, double u)
{
    const unsigned long long mk = __ballot(u != 0.0);
    if (mk) {
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        if (u != 0.0) buf[n + rank] = u;
        n += __popcll(mk);
    }
}

// the LDS of the literal loop's kernels: PMC_REF_IDS neighbour ids, then one buffer of ordered-sum terms
#define PMC_REF_LDS_INTS (PMC_REF_IDS + 2 * SQ_TERMS_BUF)
__device__ __forceinline__ double *ref_terms_of(int *lds_ids) { return reinterpret_cast<double *>(lds_ids + PMC_REF_IDS); }

// compute_energy_particle with the LinkedIterator order (src/atoms.jl:81-88,
// src/neighbours.jl:332-377).  One wavefront: the <= 27 chains are walked by <= 27 lanes
// concurrently into one LDS id buffer (offsets from the maintained per-cell counts), pair terms
// are evaluated 64 at a time, and the sum is taken strictly in iterator order.
// lds_ids: PMC_REF_IDS ints; neighbourhoods larger than that take the serial path.
template <int D>
__device__ double wave_ordered_energy(const RefArgs &a, int i, const double *xi, int spi, int *lds_ids)
{
    const int lane = wave_lane();
    if (a.list_type == 0 || a.list_type == 3) {
        // EmptyList: j in 1:N (src/neighbours.jl:42-44); VerletList: j in neighbours[i] (:422-424)
        // (round 5: the non-zero terms go through LDS in order -- sq_terms_push / sq_terms_sum, ~10 cycles per term -- instead of
        // a readlane loop over all 64 lanes of every trip, ~40 cycles per lane: an EmptyList system of 10^4 particles spent
        // 4 x 10^5 cycles per energy adding zeros.  A term that is 0 leaves the sum as it is.)
        const int total = (a.list_type == 0) ? a.N : a.vl_cnt[i];
        double *tb = ref_terms_of(lds_ids);
        double acc = 0.0;
        int nt = 0;
        constexpr int UN = 4;                                        // trips whose loads are in flight together
        __syncthreads();
        for (int base = 0; base < total; base += UN * PMC_WAVE) {
            double u[UN];
#pragma unroll
            for (int q = 0; q < UN; q++) {
                const int t = base + q * PMC_WAVE + lane;
                u[q] = 0.0;
                if (t < total) {
                    const int j = (a.list_type == 0) ? t : a.vl_nb[(size_t)i * a.vl_cap + t];
                    u[q] = ref_energy_ij<D>(a, i, j, xi, spi);
                }
            }
#pragma unroll
            for (int q = 0; q < UN; q++) {
                if (base + q * PMC_WAVE >= total) break;
                if (nt > SQ_TERMS) { acc = sq_terms_sum(tb, nt, acc); nt = 0; }
                sq_terms_push(tb, nt, u[q]);
            }
        }
        acc = sq_terms_sum(tb, nt, acc);
        __syncthreads();
        return acc;
    }
    int c = ref_cell_of<D>(a, xi);
    int cnt = a.nbrcnt[c];
    int c2 = (lane < cnt) ? a.nbr[(size_t)c * a.nnmax + lane] : 0;
    int len = (lane < cnt) ? a.ccount[c2] : 0;
    int incl = len;
#pragma unroll
    for (int s = 1; s < 32; s <<= 1) {
        int v = __shfl_up(incl, s, PMC_WAVE);
        if (lane >= s) incl += v;
    }
    const int off = incl - len;
    const int total = __shfl(incl, 31, PMC_WAVE);
    const bool fits = total <= PMC_REF_IDS;
    double acc = 0.0;
    __syncthreads();
    if (fits) {
        if (lane < cnt) {
            if (a.list_type == 2) {                 // CellList: cells[c2] in vector order (src/neighbours.jl:209-217)
                for (int k = 0; k < len; k++) lds_ids[off + k] = a.cl_items[(size_t)c2 * a.cl_cap + k];
            } else {
                int j = a.head[c2], k = 0;
                while (j != -1 && k < len) {
                    lds_ids[off + k] = j;
                    k++;
                    j = a.list[j];
                }
            }
        }
        __syncthreads();
        double *tb = ref_terms_of(lds_ids);
        int nt = 0;
        for (int base = 0; base < total; base += PMC_WAVE) {
            const int n = min(PMC_WAVE, total - base);
            double u = 0.0;
            if (lane < n) u = ref_energy_ij<D>(a, i, lds_ids[base + lane], xi, spi);
            if (nt > SQ_TERMS) { acc = sq_terms_sum(tb, nt, acc); nt = 0; }
            sq_terms_push(tb, nt, u);
        }
        acc = sq_terms_sum(tb, nt, acc);
    } else if (a.list_type == 2) {
        for (int l = 0; l < cnt; l++) {
            const int cc = a.nbr[(size_t)c * a.nnmax + l];
            const int n = a.ccount[cc];
            for (int k = 0; k < n; k++) acc += ref_energy_ij<D>(a, i, a.cl_items[(size_t)cc * a.cl_cap + k], xi, spi);
        }
    } else {
        for (int l = 0; l < cnt; l++) {
            int j = a.head[a.nbr[(size_t)c * a.nnmax + l]];
            while (j != -1) {
                acc += ref_energy_ij<D>(a, i, j, xi, spi);
                j = a.list[j];
            }
        }
    }
    __syncthreads();
    return acc;
}

// [compute_energy_particle(system, i) for i in ids] -- one wavefront per particle
template <int D>
__global__ __launch_bounds__(64) void k_ref_energy(RefArgs a, const int *__restrict__ which, int n,
                                                    double *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) int lds_ids[PMC_REF_LDS_INTS];
    for (int b = blockIdx.x; b < n; b += gridDim.x) {
        int i = which ? which[b] : b;
        double xi[3] = {0, 0, 0};
#pragma unroll
        for (int q = 0; q < D; q++) xi[q] = a.pos[(size_t)i * D + q];
        double e = wave_ordered_energy<D>(a, i, xi, a.species[i], lds_ids);
        if (wave_lane() == 0) out[b] = e;
    }
}

// update_neighbour_list!, src/neighbours.jl:276-302 (executed by lane 0)
template <int D>
__device__ void ref_relink(const RefArgs &a, int i, const double *x)
{
    int c = a.cs[i];
    int c2 = ref_cell_of<D>(a, x);
    if (c == c2) return;
    if (a.head[c] == i) {
        a.head[c] = a.list[i];
    } else {
        int j = a.head[c];
        while (a.list[j] != i) j = a.list[j];
        a.list[j] = a.list[i];
    }
    a.list[i] = a.head[c2];
    a.head[c2] = i;
    a.cs[i] = c2;
    if (a.cl_items) {
        // rows of the fast sequential engine (pmc_seq.h): a cell's chain read backwards, so push-front = append
        // (a row that overflowed earlier in this launch -- error bit 4, the host drops the rows after the launch -- counts more
        // particles than it holds: the scan stays inside the row, it never reaches the next cell's or the end of the array)
        int *v = a.cl_items + (size_t)c * a.cl_cap;
        const int n = min(a.ccount[c], a.cl_cap);
        int w = 0;
        for (int r = 0; r < n; r++) {
            const int t = v[r];
            if (t != i) v[w++] = t;
        }
        const int n2 = a.ccount[c2];
        if (n2 < a.cl_cap) a.cl_items[(size_t)c2 * a.cl_cap + n2] = i;
        else if (a.err) atomicOr(a.err, 4);
    }
    a.ccount[c]--;
    a.ccount[c2]++;
}

// update_neighbour_list!(i, c, c2, ::CellList) / update_cell_list!(::VerletList): filter! keeps the order of
// the old cell's vector, push! appends to the new one (src/neighbours.jl:162-170, 468-476).  Lane 0.
template <int D>
__device__ void ref_cellvec_update(const RefArgs &a, int i, const double *x)
{
    const int c = a.cs[i];
    const int c2 = ref_cell_of<D>(a, x);
    if (c == c2) return;
    int *v = a.cl_items + (size_t)c * a.cl_cap;
    const int n = a.ccount[c];
    int w = 0;
    for (int r = 0; r < n; r++) {
        const int t = v[r];
        if (t != i) v[w++] = t;
    }
    a.ccount[c] = w;
    const int n2 = a.ccount[c2];
    if (n2 >= a.cl_cap) { atomicOr(a.err, 1); return; }
    a.cl_items[(size_t)c2 * a.cl_cap + n2] = i;
    a.ccount[c2] = n2 + 1;
    a.cs[i] = c2;
}

// update_neighbour_list!(system, i, ::VerletList), src/neighbours.jl:494-543: nothing unless the particle moved
// dr/2 since its last update; then i leaves every list it is in (order kept), its cell vector is updated, and
// its own list is rebuilt by scanning the neighbourhood in cell / vector order.  Whole wavefront.
template <int D>
__device__ void ref_verlet_update(const RefArgs &a, int i, const double *x, int *lds_ids)
{
    const int lane = wave_lane();
    double d2 = 0.0;
#pragma unroll
    for (int q = 0; q < D; q++) {
        const double dx = x[q] - a.vl_last[(size_t)i * D + q];
        d2 = (q == 0) ? dx * dx : d2 + dx * dx;
    }
    if (d2 < a.dr2o4) return;
    const int ni = a.vl_cnt[i];
    for (int q = lane; q < ni; q += PMC_WAVE) {           // lists are duplicate-free: distinct lanes, distinct j
        const int j = a.vl_nb[(size_t)i * a.vl_cap + q];
        int *v = a.vl_nb + (size_t)j * a.vl_cap;
        const int n = a.vl_cnt[j];
        int w = 0;
        for (int r = 0; r < n; r++) {
            const int t = v[r];
            if (t != i) v[w++] = t;
        }
        a.vl_cnt[j] = w;
    }
    if (lane == 0) ref_cellvec_update<D>(a, i, x);
    ref_sync();
    const int c = ref_cell_of<D>(a, x);
    const int cnt = a.nbrcnt[c];
    int fill = 0;                                           // uniform: entries of neighbours[i] so far
    for (int l = 0; l < cnt; l++) {
        const int cc = a.nbr[(size_t)c * a.nnmax + l];
        const int n = a.ccount[cc];
        for (int base = 0; base < n; base += PMC_WAVE) {
            const int k = base + lane;
            int j = -1;
            bool in = false;
            if (k < n) {
                j = a.cl_items[(size_t)cc * a.cl_cap + k];
                in = (j != i) && (ref_dist2<D>(a, x, j) < a.rcutoff2);
            }
            const unsigned long long mk = __ballot(in);
            if (in) {
                const int pos = fill + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                const int nj = a.vl_cnt[j];
                if (pos < a.vl_cap && nj < a.vl_cap) {
                    a.vl_nb[(size_t)i * a.vl_cap + pos] = j;
                    a.vl_nb[(size_t)j * a.vl_cap + nj] = i;
                    a.vl_cnt[j] = nj + 1;
                } else {
                    atomicOr(a.err, 2);
                }
            }
            fill += __popcll(mk);
        }
    }
    if (lane == 0) {
        a.vl_cnt[i] = min(fill, a.vl_cap);
#pragma unroll
        for (int q = 0; q < D; q++) a.vl_last[(size_t)i * D + q] = x[q];
    }
    (void)lds_ids;
}

// update_neighbour_list!(system, i) for the active list type; called by the whole wavefront
template <int D>
__device__ void ref_update_list(const RefArgs &a, int i, const double *x, int *lds_ids)
{
    if (a.list_type == 1) { if (wave_lane() == 0) ref_relink<D>(a, i, x); }
    else if (a.list_type == 2) { if (wave_lane() == 0) ref_cellvec_update<D>(a, i, x); }
    else if (a.list_type == 3) ref_verlet_update<D>(a, i, x, lds_ids);
}

// build_neighbour_list!(::CellList): cells[c] = ids in ascending order (push! in the loop over i)
PMC_GLOBAL void k_ref_cellvec(int N, const int *__restrict__ cell_of, const int *__restrict__ start,
                              const int *__restrict__ ids, int ncell, int *__restrict__ items, int cap,
                              int *__restrict__ ccount, int *__restrict__ err)
{
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < ncell) {
        const int n = start[k + 1] - start[k];
        ccount[k] = min(n, cap);
        if (n > cap) atomicOr(err, 1);
    }
    if (k < N) {
        const int i = ids[k];
        const int c = cell_of[i];
        const int q = k - start[c];
        if (q < cap) items[(size_t)c * cap + q] = i;
    }
}

// build_neighbour_list!(::VerletList), src/neighbours.jl:429-462.  The serial loop over i pushes, for every
// pair within rcut + dr, j into neighbours[i] and i into neighbours[j] while scanning the cells around i for
// j > i; neighbours[k] therefore is [the i < k in ascending i] followed by [the j > k in k's own scan order].
template <int D>
__global__ void k_ref_verlet_build(RefArgs a)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= a.N) return;
    double x[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) x[q] = a.pos[(size_t)k * D + q];
    const int c = a.cs[k];
    const int cnt = a.nbrcnt[c];
    int *v = a.vl_nb + (size_t)k * a.vl_cap;
    int n = 0;
    bool over = false;
    for (int l = 0; l < cnt; l++) {
        const int cc = a.nbr[(size_t)c * a.nnmax + l];
        const int m = a.ccount[cc];
        for (int q = 0; q < m; q++) {
            const int j = a.cl_items[(size_t)cc * a.cl_cap + q];
            if (j < k && ref_dist2<D>(a, x, j) < a.rcutoff2) {
                if (n < a.vl_cap) {
                    int w = n++;                              // insertion into the ascending prefix
                    while (w > 0 && v[w - 1] > j) { v[w] = v[w - 1]; w--; }
                    v[w] = j;
                } else over = true;
            }
        }
    }
    for (int l = 0; l < cnt; l++) {
        const int cc = a.nbr[(size_t)c * a.nnmax + l];
        const int m = a.ccount[cc];
        for (int q = 0; q < m; q++) {
            const int j = a.cl_items[(size_t)cc * a.cl_cap + q];
            if (j > k && ref_dist2<D>(a, x, j) < a.rcutoff2) {
                if (n < a.vl_cap) v[n++] = j; else over = true;
            }
        }
    }
    a.vl_cnt[k] = n;
    if (over) atomicOr(a.err, 2);
#pragma unroll
    for (int q = 0; q < D; q++) a.vl_last[(size_t)k * D + q] = x[q];
}

enum { OP_DISP_PERFORM = 0, OP_DISP_REVERT = 1, OP_SWAP_PERFORM = 2, OP_SWAP_REVERT = 3, OP_STEPS = 4 };

__device__ __host__ __forceinline__ int select_move(const MovePool &p, uint32_t w)
{
    if (p.npool == 1) return 0;
    double tot = 0.0;
    for (int m = 0; m < p.npool; m++) tot += p.prob[m];
    double u = (double)w * (1.0 / 4294967296.0) * tot;
    double acc = 0.0;
    for (int m = 0; m < p.npool; m++) {
        acc += p.prob[m];
        if (u < acc) return m;
    }
    return p.npool - 1;
}

struct RefOp {
    int op;
    int i, j;                  // 0-based
    double delta[3];
    double de_in;
    unsigned long long step0;
    long long nsteps;
    uint32_t key0, key1;
    MovePool pool;
    double *result;            // e1, e2, de
    unsigned long long *stats; // attempted[8], accepted[8]
};

template <int D>
__device__ void ref_disp_perform(const RefArgs &a, int i, const double *delta, int *lds_ids,
                                 double *e1, double *e2, double *de)
{
    const int lane = wave_lane();
    double x[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) x[q] = a.pos[(size_t)i * D + q];
    int sp = a.species[i];
    double ea = wave_ordered_energy<D>(a, i, x, sp, lds_ids);
#pragma unroll
    for (int q = 0; q < D; q++) x[q] = x[q] + delta[q];   // update_position! src/moves.jl:46-48
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < D; q++) a.pos[(size_t)i * D + q] = x[q];
    }
    ref_sync();
    ref_update_list<D>(a, i, x, lds_ids);                 // update_neighbour_list!, src/moves.jl:47
    ref_sync();
    double eb = wave_ordered_energy<D>(a, i, x, sp, lds_ids);
    double d = 0.0;
    if (!(isinf(ea) || isinf(eb))) {                      // src/moves.jl:13-18
        d = eb - ea;
        if (lane == 0) *a.energy += d;
    }
    *e1 = ea; *e2 = eb; *de = d;
}

template <int D>
__device__ void ref_disp_revert(const RefArgs &a, int i, const double *mdelta, double de, int *lds_ids)
{
    double x[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) x[q] = a.pos[(size_t)i * D + q] + mdelta[q];
    __syncthreads();
    if (wave_lane() == 0) {
#pragma unroll
        for (int q = 0; q < D; q++) a.pos[(size_t)i * D + q] = x[q];
        *a.energy -= de;                                  // src/moves.jl:79
    }
    ref_sync();
    ref_update_list<D>(a, i, x, lds_ids);
    ref_sync();
}

// update_species_list!, src/moves.jl:175-179 (0-based)
__device__ __forceinline__ void ref_update_species_list(const RefArgs &a, int s1, int s2, int i, int j)
{
    a.sp_ids[a.sp_off[s1] + a.sp_heads[i]] = j;
    a.sp_ids[a.sp_off[s2] + a.sp_heads[j]] = i;
    int t = a.sp_heads[i];
    a.sp_heads[i] = a.sp_heads[j];
    a.sp_heads[j] = t;
}

template <int D>
__device__ void ref_swap_perform(const RefArgs &a, int i, int j, int s1, int s2, int *lds_ids,
                                 double *e1, double *e2, double *de)
{
    const int lane = wave_lane();
    double xi[3] = {0, 0, 0}, xj[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < D; q++) { xi[q] = a.pos[(size_t)i * D + q]; xj[q] = a.pos[(size_t)j * D + q]; }
    int spi = a.species[i], spj = a.species[j];
    double e1i = wave_ordered_energy<D>(a, i, xi, spi, lds_ids);
    double e1j = wave_ordered_energy<D>(a, j, xj, spj, lds_ids);
    if (lane == 0) { a.species[i] = spj; a.species[j] = spi; }
    ref_sync();
    double e2i = wave_ordered_energy<D>(a, i, xi, spj, lds_ids);
    double e2j = wave_ordered_energy<D>(a, j, xj, spi, lds_ids);
    double ea = e1i + e1j, eb = e2i + e2j;
    double d = 0.0;
    if (lane == 0) ref_update_species_list(a, s1, s2, i, j);
    if (!(isinf(ea) || isinf(eb))) {
        d = eb - ea;
        if (lane == 0) *a.energy += d;
    }
    ref_sync();
    *e1 = ea; *e2 = eb; *de = d;
}

__device__ void ref_swap_revert(const RefArgs &a, int i, int j, int s1, int s2, double de)
{
    if (wave_lane() == 0) {
        int spi = a.species[i], spj = a.species[j];
        a.species[j] = spi;                               // src/moves.jl:204 (as written)
        a.species[i] = spj;
        ref_update_species_list(a, s1, s2, i, j);
        *a.energy -= de;
    }
    ref_sync();
}

// `nsteps` Metropolis steps of the reference's chain following the Arianna mc_step! contract with the Philox stream
// of DESIGN.md; one wavefront
template <int D>
__device__ void ref_steps(const RefArgs &a, const RefOp &o, int *lds_ids)
{
    const int lane = wave_lane();
    double e1 = 0, e2 = 0, de = 0;
    unsigned long long att[8] = {0}, acc[8] = {0};
    for (long long t = 0; t < o.nsteps; t++) {
        unsigned long long g = o.step0 + (unsigned long long)t;
        uint32_t R[4], Q[4];
        philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ0, o.key0, o.key1, R);
        philox4x32((uint32_t)g, (uint32_t)(g >> 32), 0, TAG_SEQ1, o.key0, o.key1, Q);
        int m = select_move(o.pool, R[0]);
        double lnu = det_log(uniform53(R[2], R[3]));
        bool ok = false;
        if (o.pool.kind[m] == 0) {
            int i = (int)mulhi32(R[1], (uint32_t)a.N);
            double z[4], delta[3] = {0, 0, 0}, md[3] = {0, 0, 0};
            box_muller(Q[0], Q[1], &z[0], &z[1]);
            box_muller(Q[2], Q[3], &z[2], &z[3]);
#pragma unroll
            for (int q = 0; q < D; q++) { delta[q] = o.pool.sigma[m] * z[q]; md[q] = -delta[q]; }
            ref_disp_perform<D>(a, i, delta, lds_ids, &e1, &e2, &de);
            ok = lnu < -(e2 - e1) / a.temperature;
            if (!ok) ref_disp_revert<D>(a, i, md, de, lds_ids);
        } else {
            int sa = o.pool.sp0[m], sb = o.pool.sp1[m];
            int na = a.sp_off[sa + 1] - a.sp_off[sa], nb = a.sp_off[sb + 1] - a.sp_off[sb];
            if (na > 0 && nb > 0) {
                int i = a.sp_ids[a.sp_off[sa] + mulhi32(R[1], (uint32_t)na)];
                int j = a.sp_ids[a.sp_off[sb] + mulhi32(Q[0], (uint32_t)nb)];
                ref_swap_perform<D>(a, i, j, sa, sb, lds_ids, &e1, &e2, &de);
                ok = lnu < -(e2 - e1) / a.temperature;
                if (!ok) ref_swap_revert(a, j, i, sa, sb, de);
            }
        }
        // runtime-indexed counters stay in registers via a select chain
#pragma unroll
        for (int q = 0; q < 8; q++) {
            att[q] += (q == m);
            acc[q] += (q == m && ok);
        }
    }
    if (lane == 0)
        for (int q = 0; q < 8; q++) { o.stats[q] += att[q]; o.stats[8 + q] += acc[q]; }
}

// One wavefront runs the reference's strictly sequential chain: per-move verbs
// (batch of one) or `nsteps` Metropolis steps following the Arianna mc_step!
// contract with the Philox stream of DESIGN.md.
template <int D>
__global__ __launch_bounds__(64) void k_ref_ops(RefArgs a, RefOp o)
{
    __shared__ __attribute__((aligned(16))) int lds_ids[PMC_REF_LDS_INTS];
    const int lane = wave_lane();
    double e1 = 0, e2 = 0, de = 0;
    if (o.op == OP_DISP_PERFORM) {
        ref_disp_perform<D>(a, o.i, o.delta, lds_ids, &e1, &e2, &de);
    } else if (o.op == OP_DISP_REVERT) {
        ref_disp_revert<D>(a, o.i, o.delta, o.de_in, lds_ids);
    } else if (o.op == OP_SWAP_PERFORM) {
        ref_swap_perform<D>(a, o.i, o.j, a.species[o.i], a.species[o.j], lds_ids, &e1, &e2, &de);
    } else if (o.op == OP_SWAP_REVERT) {
        // called with the inverted action (src/moves.jl:212-214): o.i currently holds action.species[1]
        ref_swap_revert(a, o.i, o.j, a.species[o.i], a.species[o.j], o.de_in);
    } else {
        ref_steps<D>(a, o, lds_ids);
    }
    if (lane == 0 && o.result) { o.result[0] = e1; o.result[1] = e2; o.result[2] = de; }
}

// The chains of a `parallel = true` run (src/ParticlesMC.jl:164,246) on the literal loop, all in ONE launch: workgroup b runs
// chain b (its own arguments, any shape, any list type).  What the literal loop serves -- VerletList, EmptyList systems beyond
// the LDS, swap pools between a species and itself -- used to run its chains one after the other.
template <int D>
__global__ __launch_bounds__(64) void k_ref_steps_many(const RefArgs *__restrict__ args, const RefOp *__restrict__ ops)
{
    __shared__ __attribute__((aligned(16))) int lds_ids[PMC_REF_LDS_INTS];
    ref_steps<D>(args[blockIdx.x], ops[blockIdx.x], lds_ids);
}

// ------------------------------------------------------------------------------
// EnergyBias (src/moves.jl:246-280): energy-weighted choice of a DiscreteSwap pair, on the device
// ------------------------------------------------------------------------------

// Base sum(::Vector{Float64}): pairwise with 1024-element leaves (base/reduce.jl), as pmc_hip.hip:pairwise_sum
__device__ double dev_pairwise_sum(const double *v, long long lo, long long hi)
{
    if (hi - lo < 1024) {
        double a = v[lo];
        for (long long k = lo + 1; k <= hi; k++) a += v[k];
        return a;
    }
    const long long mid = lo + ((hi - lo) >> 1);
    return dev_pairwise_sum(v, lo, mid) + dev_pairwise_sum(v, mid + 1, hi);
}

struct BiasArgs {
    const double *e;           // per-particle energies, particle-id order (reference-order sums)
    const int *sp_off, *sp_ids;
    int sp[2];                 // 0-based species of the pair
    double theta[2], u[2];     // policy parameters; the caller's two uniforms in [0, 1)
    int given[2];              // >= 0: particle ids of a given pair (log density only), -1: sample
    double *w;                 // scratch, N doubles
    double *out;               // [0] i, [1] j (0-based ids), [2] log q, [3] log-sum-exp 1, [4] log-sum-exp 2
};

// One workgroup.  For each of the two species: w_k = exp(theta e_k) over sp_ids[species] in list order (parallel),
// S = sum(w) in Julia's pairwise order (one thread: the order is the result), p_k = w_k / S (parallel), and either the
// inverse-CDF pick `cp <= u` of rand(rng, Categorical(p)) (src/moves.jl:274-277) or nothing (log density of a given pair).
PMC_GLOBAL __launch_bounds__(256) void k_energy_bias(BiasArgs a)
{
    __shared__ double s_sum;
    __shared__ int s_pick;
    double numer = 0.0, lse[2] = {0.0, 0.0};
    int pick[2] = {0, 0};
    for (int side = 0; side < 2; side++) {
        const int lo = a.sp_off[a.sp[side]], n = a.sp_off[a.sp[side] + 1] - lo;
        double *w = a.w + lo;
        for (int k = threadIdx.x; k < n; k += blockDim.x) w[k] = det_exp(a.theta[side] * a.e[a.sp_ids[lo + k]]);
        __syncthreads();
        if (threadIdx.x == 0) s_sum = (n > 0) ? dev_pairwise_sum(w, 0, n - 1) : 0.0;
        __syncthreads();
        const double S = s_sum;
        if (a.given[side] < 0) {
            for (int k = threadIdx.x; k < n; k += blockDim.x) w[k] = w[k] / S;          // w1s .= w1s ./ sum(w1s)
            __syncthreads();
            if (threadIdx.x == 0) {
                int i = 0;
                double cp = (n > 0) ? w[0] : 1.0;
                while (cp <= a.u[side] && i < n - 1) { i++; cp += w[i]; }
                s_pick = (n > 0) ? a.sp_ids[lo + i] : -1;
            }
            __syncthreads();
            pick[side] = s_pick;
        } else {
            pick[side] = a.given[side];
        }
        lse[side] = det_log(S);
        if (pick[side] >= 0) numer += a.theta[side] * a.e[pick[side]];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.out[0] = (double)pick[0];
        a.out[1] = (double)pick[1];
        a.out[2] = numer - lse[0] - lse[1];       // src/moves.jl:257-262
        a.out[3] = lse[0];
        a.out[4] = lse[1];
    }
}

