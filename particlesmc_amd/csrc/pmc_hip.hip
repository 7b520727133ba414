// pmc_hip.hip -- C ABI (include/pmc.h) over the gfx950 kernels of pmc_kernels.h.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
#include "../../include/pmc.h"
#include "pmc_kernels.h"
#include "pmc_seq.h"

// The instances of the sweep kernel are compiled in their own translation units (pmc_cb_inst.hip, one per dimension and
// model kind, in parallel); here they are only declared.  -DPMC_SINGLE_TU (and the development builds) keep everything
// in this file.
#if !defined(PMC_SINGLE_TU) && !defined(PMC_DEV_BUILD)
#define PMC_DECL_3_0(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<3, 0, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_3_1(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<3, 1, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_3_2(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<3, 2, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_3_3(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<3, 3, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_3_4(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<3, 4, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_2_0(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<2, 0, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_2_1(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<2, 1, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_2_2(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<2, 2, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_2_3(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<2, 3, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
#define PMC_DECL_2_4(M_, NW_, ONE_, MW_) extern template __global__ void k_cb_pass<2, 4, M_, NW_, ONE_, MW_>(const CbArgs *__restrict__, int);
PMC_CB_VARIANTS(PMC_DECL_3_0) PMC_CB_VARIANTS(PMC_DECL_3_1) PMC_CB_VARIANTS(PMC_DECL_3_2) PMC_CB_VARIANTS(PMC_DECL_3_3) PMC_CB_VARIANTS(PMC_DECL_3_4)
PMC_CB_VARIANTS(PMC_DECL_2_0) PMC_CB_VARIANTS(PMC_DECL_2_1) PMC_CB_VARIANTS(PMC_DECL_2_2) PMC_CB_VARIANTS(PMC_DECL_2_3) PMC_CB_VARIANTS(PMC_DECL_2_4)
#define PMC_DECLI(DD_, KK_, M_, NW_, ONE_) extern template __global__ void k_cb_pass<DD_, KK_, M_, NW_, ONE_, PMC_IMG_MW(NW_), true>(const CbArgs *__restrict__, int);
#define PMC_DECLI_3_0(M_, NW_, ONE_) PMC_DECLI(3, 0, M_, NW_, ONE_)
#define PMC_DECLI_3_1(M_, NW_, ONE_) PMC_DECLI(3, 1, M_, NW_, ONE_)
#define PMC_DECLI_3_2(M_, NW_, ONE_) PMC_DECLI(3, 2, M_, NW_, ONE_)
#define PMC_DECLI_3_3(M_, NW_, ONE_) PMC_DECLI(3, 3, M_, NW_, ONE_)
#define PMC_DECLI_3_4(M_, NW_, ONE_) PMC_DECLI(3, 4, M_, NW_, ONE_)
#define PMC_DECLI_2_0(M_, NW_, ONE_) PMC_DECLI(2, 0, M_, NW_, ONE_)
#define PMC_DECLI_2_1(M_, NW_, ONE_) PMC_DECLI(2, 1, M_, NW_, ONE_)
#define PMC_DECLI_2_2(M_, NW_, ONE_) PMC_DECLI(2, 2, M_, NW_, ONE_)
#define PMC_DECLI_2_3(M_, NW_, ONE_) PMC_DECLI(2, 3, M_, NW_, ONE_)
#define PMC_DECLI_2_4(M_, NW_, ONE_) PMC_DECLI(2, 4, M_, NW_, ONE_)
PMC_CB_VARIANTS_IMG(PMC_DECLI_3_0) PMC_CB_VARIANTS_IMG(PMC_DECLI_3_1) PMC_CB_VARIANTS_IMG(PMC_DECLI_3_2) PMC_CB_VARIANTS_IMG(PMC_DECLI_3_3) PMC_CB_VARIANTS_IMG(PMC_DECLI_3_4)
PMC_CB_VARIANTS_IMG(PMC_DECLI_2_0) PMC_CB_VARIANTS_IMG(PMC_DECLI_2_1) PMC_CB_VARIANTS_IMG(PMC_DECLI_2_2) PMC_CB_VARIANTS_IMG(PMC_DECLI_2_3) PMC_CB_VARIANTS_IMG(PMC_DECLI_2_4)
// ... and the big-neighbourhood instances (teams of eight, no images)
#define PMC_DECLB(DD_, KK_, M_, ONE_) extern template __global__ void k_cb_pass<DD_, KK_, M_, 8, ONE_, PMC_TEAM_MINWAVES, false, true>(const CbArgs *__restrict__, int);
#define PMC_DECLB_3_0(M_, ONE_) PMC_DECLB(3, 0, M_, ONE_)
#define PMC_DECLB_3_1(M_, ONE_) PMC_DECLB(3, 1, M_, ONE_)
#define PMC_DECLB_3_2(M_, ONE_) PMC_DECLB(3, 2, M_, ONE_)
#define PMC_DECLB_3_3(M_, ONE_) PMC_DECLB(3, 3, M_, ONE_)
#define PMC_DECLB_3_4(M_, ONE_) PMC_DECLB(3, 4, M_, ONE_)
#define PMC_DECLB_2_0(M_, ONE_) PMC_DECLB(2, 0, M_, ONE_)
#define PMC_DECLB_2_1(M_, ONE_) PMC_DECLB(2, 1, M_, ONE_)
#define PMC_DECLB_2_2(M_, ONE_) PMC_DECLB(2, 2, M_, ONE_)
#define PMC_DECLB_2_3(M_, ONE_) PMC_DECLB(2, 3, M_, ONE_)
#define PMC_DECLB_2_4(M_, ONE_) PMC_DECLB(2, 4, M_, ONE_)
PMC_CB_VARIANTS_BIG(PMC_DECLB_3_0) PMC_CB_VARIANTS_BIG(PMC_DECLB_3_1) PMC_CB_VARIANTS_BIG(PMC_DECLB_3_2) PMC_CB_VARIANTS_BIG(PMC_DECLB_3_3) PMC_CB_VARIANTS_BIG(PMC_DECLB_3_4)
PMC_CB_VARIANTS_BIG(PMC_DECLB_2_0) PMC_CB_VARIANTS_BIG(PMC_DECLB_2_1) PMC_CB_VARIANTS_BIG(PMC_DECLB_2_2) PMC_CB_VARIANTS_BIG(PMC_DECLB_2_3) PMC_CB_VARIANTS_BIG(PMC_DECLB_2_4)
#endif

#include <rccl/rccl.h>     // types only: the library itself is opened with dlopen on first use
#include <dlfcn.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;

struct pmc_system {
    pmc_params p;
    std::vector<pmc_pair_model> models;
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> pass_ev;   // (begin, end) per sweep around the colour passes
    int N = 0, D = 0, S = 0;
    int kind = 0;                // common model kind, 4 = mixed
    double energy = 0.0;         // system.energy[1]
    long long e_fix = 0;
    bool e_fix_valid = false;
    bool energy_stale = false;   // a partial throughput sweep changed positions without folding its dE: recompute before use
    bool state_invalid = false;  // ... and until that recomputation (or an upload) no caller-side energy is taken (pmc_set_energy)
    bool cb_started = false;     // the current pmc_sweep call has enqueued sweep kernels (errors before that leave the state alone)
    uint64_t step_ctr = 0, sweep_ctr = 0;
    uint32_t key[2] = {0, 0};
    // particle-id-ordered state
    double *d_pos = nullptr;
    int *d_species = nullptr;
    DevModel *d_models = nullptr;
    // reference LinkedList
    int ncell_ref = 0, nn = 0;
    int *d_cs = nullptr, *d_head = nullptr, *d_list = nullptr, *d_ccount = nullptr;
    int *d_nbr = nullptr, *d_nbrcnt = nullptr;
    std::vector<int> h_nbr, h_nbrcnt;
    bool cells_built = false;
    int *d_sp_off = nullptr, *d_sp_ids = nullptr, *d_sp_heads = nullptr;
    bool species_list_built = false;
    bool species_list_stale = false;   // throughput sweeps ran since the SpeciesList was built
    double *d_energy = nullptr;
    // binning scratch
    int *d_cell_of = nullptr, *d_count = nullptr, *d_fill = nullptr, *d_tmp_ids = nullptr;
    int *d_arrival = nullptr, *d_tile_sum = nullptr, *d_order = nullptr, *d_hist = nullptr;
    bool use_order = true;
    // alternative reference list types (pmc_set_list_type): 0 EmptyList, 1 LinkedList, 2 CellList, 3 VerletList
    int list_type = 1;
    double list_dr = 0.0;
    int *d_cl_items = nullptr, *d_vl_nb = nullptr, *d_vl_cnt = nullptr, *d_ref_err = nullptr;
    double *d_vl_last = nullptr;
    int cl_cap = 0, vl_cap = 0;
    long long sqv_last[4] = {0, 0, 0, 0};     // engine 4, last call: batches, steps run by wavefront 0 alone, list rebuilds, steps it ran
    int cap_hint = 0;       // LDS staging capacity learnt from the last sweeps' histogram (cb_cap_learn); 0: none yet
    unsigned long long kept_seen[16] = {0};   // the part of the current call's histogram cb_cap_learn has already used
    // proposal slots of the current sweep (k_mc_slots)
    double *d_slot_f = nullptr;          // [4][rounds N]: T ln u, dx, dy, dz
    uint32_t *d_slot_w = nullptr;        // [2][rounds N]: particle word, second swap word
    unsigned char *d_slot_m = nullptr;   // [rounds N]: move index
    int rounds = 1;                      // schedule parameter m (pmc_set_rounds): sweeps per pass = proposal rounds per cell visit
    int slot_rounds = 1;                 // rounds the slot arrays are allocated for
    // fused sweep (all colours in one launch, per-cell completion flags): flags, ticket, epoch
    bool use_fused = true;
    int fused_lpt = 0;           // longest-first order inside a fused launch (PMC_FUSED_LPT=1): measured equal pass
                                 // time at N=1e6 and slower below, and it costs the three ordering kernels
    unsigned int *d_done = nullptr;
    unsigned long long *d_ticket = nullptr;
    unsigned int epoch = 0;
    unsigned long long ticket_next = 0;
    int ticket_lanes = PMC_TICKET_LANES;   // 8: one ticket counter per XCD, zeroed every sweep; 1: one monotonic counter (PMC_TICKETS=1)
    int team = 1;                // wavefronts that share one cell in k_cb_pass (1, 2 or 4)
    double maxcut = 0.0;
    bool use_filter = true;
    // a sweep in flight (pmc_sweep_begin .. pmc_sweep_end)
    bool pending = false;
    int pending_mode = 0;
    int64_t pending_nsweeps = 0, pending_timed = 0;
    int pending_launches_per_sweep = 0;
    unsigned long long *h_stats = nullptr;   // pinned host copy of d_stats
    pmc_stats done_stats{};
    int *d_start_ref = nullptr, *d_ids_ref = nullptr;
    // throughput grid
    bool mc_ok = false;
    GridGeom grid{};
    int ncell_mc = 0;
    int *d_start = nullptr, *d_ids = nullptr;
    double *d_sx = nullptr, *d_sy = nullptr, *d_sz = nullptr;
    unsigned char *d_ssp = nullptr;
    int *d_img = nullptr, *d_simg = nullptr;   // persistent packed images (id order; set by pmc_upload) and their cell-sorted copy
    bool any_img = false;        // some particle has a non-zero image (else the kernels get null image pointers and skip all of it)
    bool img_valid = false;      // the images belong to the current positions: cleared wherever positions change outside the
                                 // throughput engine (pmc_upload, the per-move verbs, reference-order sweeps, an aborted sweep) -- the
                                 // same events that invalidate the exact running energy, so a re-fold never meets a valid one
    bool mc_binned = false;      // CSR + SoA consistent with d_pos
    bool count_is_zero = false;  // d_count[0..ncell_mc] is all zero (left so by the last throughput scan)
    unsigned long long *d_stats = nullptr;
    CbArgs *d_cbtab = nullptr;   // launch arguments of the passes of one chunk (k_cb_pass reads them from device memory)
    size_t cbtab_cap = 0;        // entries d_cbtab holds
    double *d_result = nullptr;
    int max_count = 0;
    // fast reference-order engine (pmc_seq.h): speculative windows over the sequential chain
    int seq_engine = 0;          // 0 auto, 1 the one-wavefront loop (k_ref_ops), 2 speculative windows
    int seq_window = 0;          // window slots (0: from the cell count)
    int rows_min_cap = 0;        // row capacity a LinkedList's rows must at least have (doubled when a row overflowed mid-chain)
    SqSlot *d_sq_slots[2] = {nullptr, nullptr};
    SqGen *d_sq_gen = nullptr;
    unsigned long long *d_sq_M = nullptr, *d_sq_cp = nullptr;
    int *d_sq_moveto = nullptr, *d_sq_cellhead = nullptr, *d_sq_cellnext = nullptr;
    SqLog *d_sq_log = nullptr, *h_sq_log = nullptr;
    int sq_ring = 0;             // records the log ring holds (a power of two)
    int sq_cap_K = 0, sq_gen = 0;   // window the per-slot buffers are allocated for; steps the ring of random parts holds
    SqCtl *d_sq_ctl = nullptr, *h_sq_ctl = nullptr;
    SqCtl sq_last{};             // control block of the last fast sequential call (diagnostics: rounds, evaluations)
    long long sq_last_steps = 0; // ... and the steps of that call the host folded into system.energy (all of them when it ended well)
    bool single_cell = false;    // the throughput grid is one cell spanning the periodic box (boxes of 2 ... 4 cutoffs per side, N <= PMC_CAP_MAX)
    bool no_big = false;         // a shared launch with a replica that carries images: no big-neighbourhood instances for any of its chains (cb_big)
    bool inject_starve = false;  // pmc_debug_inject(PMC_INJECT_STARVED_LAUNCH): the next fused launch is declared starved (one shot)
    hipGraphExec_t sq_graph = nullptr;   // SQ_GRAPH_ROUNDS rounds of the windows, captured for sq_graph_args / sq_graph_K
    SqArgs sq_graph_args;
    int sq_graph_K = 0;
    SqArgs *d_sq_args = nullptr;         // the chain's entry of the rounds' argument table (a group of one)
};

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            char buf_[512];                                                                     \
            snprintf(buf_, sizeof buf_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                     __FILE__, __LINE__);                                                       \
            if (h) (h)->err = buf_;                                                             \
            g_err = buf_;                                                                       \
            return PMC_ERR_HIP;                                                                 \
        }                                                                                       \
    } while (0)

// temporary device allocation that is released on every return path
struct DevTmp {
    void *p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <class T> T *as() const { return static_cast<T *>(p); }
};

// Declared AFTER the host vectors and device temporaries that asynchronous copies on `s` read or write: whichever way the function
// is left -- an error return between the first copy and the function's own synchronize included -- the stream is drained before
// those buffers are destroyed (destruction runs in reverse order of declaration).  On the normal path the stream is already idle.
struct StreamDrain {
    hipStream_t s;
    explicit StreamDrain(hipStream_t st) : s(st) {}
    ~StreamDrain() { (void)hipStreamSynchronize(s); }
};

static int fail(pmc_system *h, int code, const char *msg)
{
    if (h) h->err = msg;
    g_err = msg;
    return code;
}

static int set_device(pmc_system *h) { HIPCHK(h, hipSetDevice(h->p.device)); return PMC_OK; }

extern "C" const char *pmc_version(void) { return "particlesmc-hip 0.1 (gfx950)"; }

extern "C" const char *pmc_last_error(const pmc_system *h) { return h ? h->err.c_str() : g_err.c_str(); }

extern "C" int pmc_device_count(int *n)
{
    if (!n) return PMC_ERR_BADARG;
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; g_err = hipGetErrorString(e); return PMC_ERR_HIP; }
    *n = c;
    return PMC_OK;
}

// ---- host-side restatements of the reference constructors -----------------------------

static double h_lennard_jones(double r2, double eps4, double sigma2)   // src/models.jl:30-34
{
    double x = sigma2 * (1.0 / r2);
    double x3 = x * x * x;
    return eps4 * (x3 * x3 - x3);
}

static double h_pow_int(double x, long long n)   // Julia Base pow_body, as dev_pow_int
{
    double y = 1.0, xnlo = 0.0, ynlo = 0.0, err, hi;
    if (n == 3) return x * x * x;
    if (n < 0) {
        double rx = 1.0 / x;
        if (n == -2) return rx * rx;
        if (std::isfinite(x)) xnlo = -std::fma(x, rx, -1.0) * rx;
        x = rx;
        n = -n;
    }
    while (n > 1) {
        if (n & 1) {
            err = std::fma(y, xnlo, x * ynlo);
            hi = x * y;
            ynlo = std::fma(x, y, -hi);
            y = hi;
            ynlo += err;
        }
        err = x * 2.0 * xnlo;
        hi = x * x;
        xnlo = std::fma(x, x, -hi);
        x = hi;
        xnlo += err;
        n >>= 1;
    }
    err = std::fma(y, xnlo, x * ynlo);
    if (std::isfinite(x) && std::isfinite(err)) return std::fma(x, y, err);
    return x * y;
}

static double h_potential(double r2, const pmc_pair_model &m)
{
    switch (m.kind) {
    case PMC_MODEL_LJ: return h_lennard_jones(r2, m.eps4, m.sigma2) - m.shift;
    case PMC_MODEL_SOFT_INT: return m.eps * h_pow_int(m.sigma2 / r2, m.ndiv2_i) - m.shift;
    case PMC_MODEL_SOFT_FLOAT: return m.eps * std::sqrt(h_pow_int(m.sigma2 / r2, (int64_t)(2.0 * m.ndiv2_f))) - m.shift;
    default: {
        double lj = h_lennard_jones(r2, m.eps4, m.sigma2);
        double sh = m.eps4 * (m.C0 + r2 * std::fma(r2, m.C4_s4, m.C2_s2));
        return lj + sh;
    }
    }
}

extern "C" int pmc_model_lennard_jones(double eps, double sigma, double rcut, int32_t shift_potential,
                                       pmc_pair_model *out)
{
    if (!out) return PMC_ERR_BADARG;
    memset(out, 0, sizeof *out);
    out->kind = PMC_MODEL_LJ;
    out->eps = eps; out->eps4 = 4.0 * eps; out->sigma = sigma; out->sigma2 = sigma * sigma;
    out->rcut = rcut; out->rcut2 = rcut * rcut;
    out->shift = shift_potential ? h_lennard_jones(out->rcut2, out->eps4, out->sigma2) : 0.0;
    return PMC_OK;
}

extern "C" int pmc_model_soft_spheres(double eps, double sigma, int32_t n, double rcut, pmc_pair_model *out)
{
    if (!out) return PMC_ERR_BADARG;
    memset(out, 0, sizeof *out);
    out->n = n; out->eps = eps; out->sigma = sigma; out->sigma2 = sigma * sigma;
    out->rcut = rcut; out->rcut2 = rcut * rcut;
    if (n & 1) { out->kind = PMC_MODEL_SOFT_FLOAT; out->ndiv2_f = (double)n / 2.0; }
    else { out->kind = PMC_MODEL_SOFT_INT; out->ndiv2_i = n / 2; }
    out->shift = 0.0;
    out->shift = h_potential(out->rcut2, *out);   // inverse_power(rcut2, ...) src/models.jl:68
    return PMC_OK;
}

extern "C" int pmc_model_smooth_lennard_jones(double eps, double sigma, double rcut, pmc_pair_model *out)
{
    if (!out) return PMC_ERR_BADARG;
    memset(out, 0, sizeof *out);
    out->kind = PMC_MODEL_SMOOTH_LJ;
    out->eps = eps; out->eps4 = 4.0 * eps; out->sigma = sigma; out->sigma2 = sigma * sigma;
    out->C0 = 0.04049023795;
    out->C2_s2 = -0.00970155098 / out->sigma2;
    out->C4_s4 = 0.00062012616 / (out->sigma2 * out->sigma2);
    out->rcut = rcut; out->rcut2 = rcut * rcut;
    return PMC_OK;
}

extern "C" int pmc_potential(double r2, const pmc_pair_model *m, double *u)
{
    if (!m || !u) return PMC_ERR_BADARG;
    *u = h_potential(r2, *m);
    return PMC_OK;
}

extern "C" int pmc_box_from_density(int64_t N, double density, int32_t d, double *box)
{
    if (!box || d < 1 || d > 3 || !(density > 0.0) || N < 1) return PMC_ERR_BADARG;
    double b = pow((double)N / density, 1.0 / (double)d);   // src/atoms.jl:45 (libm pow)
    for (int a = 0; a < d; a++) box[a] = b;
    return PMC_OK;
}

static double h_jl_fld(double x, double y)   // Julia Base fld for floats (base/div.jl)
{
    double r = fmod(x, y), m;
    if (r == 0.0) m = copysign(r, y);
    else if ((r > 0.0) != (y > 0.0)) m = r + y;
    else m = r;
    return rint((x - m) / y);
}

extern "C" int pmc_linked_list_geometry(int32_t d, const double *box, double rcut, double *cell, int64_t *ncells)
{
    if (!box || !cell || !ncells || d < 1 || d > 3 || !(rcut > 0.0)) return PMC_ERR_BADARG;
    for (int a = 0; a < d; a++) {
        cell[a] = box[a] / h_jl_fld(box[a], rcut);         // src/neighbours.jl:237
        double q = box[a] / cell[a];                       // src/neighbours.jl:238: Int(box / cell)
        if (!(q >= 1.0) || q != rint(q)) {
            g_err = "LinkedList: box / cell is not a positive integer (rcut larger than the box?)";
            return PMC_ERR_BADARG;
        }
        ncells[a] = (int64_t)q;
    }
    return PMC_OK;
}

static DevModel to_dev(const pmc_pair_model &m)
{
    DevModel d;
    d.kind = m.kind; d.n = m.n; d.eps = m.eps; d.eps4 = m.eps4; d.sigma2 = m.sigma2;
    d.rcut2 = m.rcut2; d.shift = m.shift; d.ndiv2_i = m.ndiv2_i; d.ndiv2_f = m.ndiv2_f;
    d.C0 = m.C0; d.C2_s2 = m.C2_s2; d.C4_s4 = m.C4_s4;
    return d;
}

static long long fld_i(long long a, long long b)
{
    long long q = a / b;
    if ((a % b != 0) && ((a < 0) != (b < 0))) q -= 1;
    return q;
}

// cell_index + build_neighbour_cells, src/neighbours.jl:79-111 (0-based output)
static void host_neighbour_cells(pmc_system *h)
{
    const int D = h->D;
    const int nn = (D == 3) ? 27 : 9;
    h->nn = nn;
    long long nc[3] = {h->p.ncells[0], h->p.ncells[1], D == 3 ? h->p.ncells[2] : 1};
    auto cidx = [&](const long long *mc) {
        long long c = 0, stride = 1;
        for (int a = D - 1; a >= 0; a--) {
            long long f = mc[a] - fld_i(mc[a], nc[a]) * nc[a];
            c += f * stride;
            stride *= nc[a];
        }
        return (int)c;
    };
    h->h_nbr.assign((size_t)h->ncell_ref * nn, -1);
    h->h_nbrcnt.assign((size_t)h->ncell_ref, 0);
    long long mc[3] = {1, 1, 1};
    for (int it = 0; it < h->ncell_ref; it++) {
        int c = cidx(mc);
        int cnt = 0;
        int *out = &h->h_nbr[(size_t)c * nn];
        for (int o = 0; o < nn; o++) {
            long long mc2[3] = {0, 0, 0};
            int r = o;
            for (int a = 0; a < D; a++) { mc2[a] = mc[a] + (r % 3) - 1; r /= 3; }
            int c2 = cidx(mc2);
            bool dup = false;
            for (int k = 0; k < cnt; k++) if (out[k] == c2) { dup = true; break; }   // unique!
            if (!dup) out[cnt++] = c2;
        }
        h->h_nbrcnt[c] = cnt;
        for (int a = 0; a < D; a++) {
            if (++mc[a] <= nc[a]) break;
            mc[a] = 1;
        }
    }
}

// wavefronts per cell visit from the number of cells that are active at the same time (one colour of one system, or of all the
// chains of a shared launch: pmc_sweep_many); the measurements behind the thresholds are in pmc_create
static int team_rule(int D, long long nact)
{
    // 2-d neighbourhoods hold ~70-100 candidates (two filter trips) and one wavefront per cell takes over earlier.  Binary r^-14
    // soft spheres, rho = 0.8, 10^8 moves/s for teams of 1 / 2 / 4 / 8 (end of round 3, m = 1): 100 cells per colour 0.33 / 0.36 /
    // 0.43 / 0.43; 400: 1.26 / 1.34 / 1.56 / 1.12; 1600: 4.47 / 4.38 / 3.16 / 1.40; 6561: 12.1 / 5.9 / 3.7 / 1.5; 26569 (N = 2^20):
    // 14.8 / 6.3 / 3.9 / 1.5.  (Rounds 2-3 kept two wavefronts per cell from 512 cells per colour on WITHOUT an upper bound -- a
    // rule fitted at N = 65536 before the fused staging existed: large 2-d systems ran at half their rate.)
    if (D == 2) return (nact <= 256) ? 8 : ((nact <= 900) ? 4 : ((nact <= 1500) ? 2 : 1));
    return (nact <= 256) ? 8 : ((nact <= 900) ? 4 : ((nact <= 2400) ? 2 : 1));
}

extern "C" int pmc_create(const pmc_params *p, pmc_system **out)
{
    if (!p || !out) return fail(nullptr, PMC_ERR_BADARG, "pmc_create: null argument");
    *out = nullptr;
    if (p->N < 1 || p->N > 2000000000LL) return fail(nullptr, PMC_ERR_BADARG, "pmc_create: N out of range");
    if (p->d != 2 && p->d != 3) return fail(nullptr, PMC_ERR_BADARG, "pmc_create: d must be 2 or 3");
    if (p->nspecies < 1 || p->nspecies > 8 || !p->models)
        return fail(nullptr, PMC_ERR_BADARG, "pmc_create: nspecies must be 1..8 with a model matrix");
    if (!(p->temperature > 0.0)) return fail(nullptr, PMC_ERR_BADARG, "pmc_create: temperature must be > 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, PMC_ERR_HIP, "pmc_create: no HIP device available (the HIP path has no CPU fallback)");
    if (p->device < 0 || p->device >= ndev) return fail(nullptr, PMC_ERR_BADARG, "pmc_create: bad device index");

    pmc_system *h = new pmc_system();
    h->p = *p;
    h->N = (int)p->N;
    h->D = p->d;
    h->S = p->nspecies;
    h->models.assign(p->models, p->models + (size_t)h->S * h->S);
    h->p.models = nullptr;
    h->key[0] = (uint32_t)p->seed;
    h->key[1] = (uint32_t)(p->seed >> 32) + p->chain;
    h->kind = h->models[0].kind;
    double maxcut = 0.0;
    for (auto &m : h->models) {
        if (m.kind != h->kind) h->kind = 4;
        if (m.kind < 0 || m.kind > 3) { delete h; return fail(nullptr, PMC_ERR_BADARG, "pmc_create: unknown model kind"); }
        if (m.kind == PMC_MODEL_SOFT_FLOAT) {   // n odd: ndiv2 = k + 1/2 (src/models.jl:66); evaluated as sqrt(x^n)
            const double n2 = 2.0 * m.ndiv2_f;
            if (!(n2 >= 1.0 && n2 < 1e6) || n2 != std::floor(n2) || ((int64_t)n2 & 1) == 0) {
                delete h;
                return fail(nullptr, PMC_ERR_BADARG, "pmc_create: PMC_MODEL_SOFT_FLOAT needs ndiv2_f = n/2 with n odd");
            }
        }
        maxcut = std::max(maxcut, m.rcut);
    }
    h->maxcut = maxcut;
    if (const char *e = getenv("PMC_NO_FILTER")) h->use_filter = !(e[0] == '1');
    // reference LinkedList geometry comes from the host (src/neighbours.jl:237-238)
    long long nref = 1;
    for (int a = 0; a < h->D; a++) {
        if (p->ncells[a] < 1 || !(p->cell[a] > 0.0) || !(p->box[a] > 0.0)) {
            delete h;
            return fail(nullptr, PMC_ERR_BADARG, "pmc_create: degenerate LinkedList grid (rcut > box?)");
        }
        nref *= p->ncells[a];
    }
    if (nref > 400000000LL) { delete h; return fail(nullptr, PMC_ERR_BADARG, "pmc_create: too many cells"); }
    h->ncell_ref = (int)nref;
    // throughput grid: widest even grid with cells wider than rcut by a margin
    // (DESIGN.md "schedule"; restated in oracle/pmc_oracle.c:orc_cb_create)
    h->mc_ok = true;
    long long nmc = 1;
    const double need = maxcut * (1.0 + 1.0 / 1048576.0);
    int nsmall = 0;
    for (int a = 0; a < h->D; a++) {
        long long n = (long long)floor(p->box[a] / need);
        n -= (n & 1);
        while (n >= 4 && p->box[a] / (double)n < need) n -= 2;
        if (n < 4) { nsmall++; n = 1; }
        h->grid.n[a] = (int)n;
        h->grid.w[a] = p->box[a] / (double)n;
        h->grid.shift[a] = 0.0;
        h->grid.L[a] = p->box[a];
        nmc *= n;
    }
    if (nsmall > 0) {
        // SINGLE-CELL BOXES (round 5; restated in oracle/pmc_oracle.c:orc_cb_create).  Fewer than 4 cutoffs per side cannot be
        // two-coloured periodically.  A box that is small in EVERY dimension, still >= 2 cutoffs wide (the two roundings of the
        // minimum image then only differ beyond the cutoff) and holds <= PMC_CAP_MAX particles is ONE cell: one colour, one visit
        // per pass -- its N proposal slots in order against all N particles, evaluated eight at a time by a team with the exact
        // pair corrections of any batch; no move "leaves the cell", DoubleUniform picks are the reference's global ones.  The
        // reference's own validation example is such a box (N = 1000, rcut = 4 sigma: examples/lj-mixture/run-validation.py:26-34).
        bool ok = (nsmall == h->D) && h->N <= PMC_CAP_MAX;
        for (int a = 0; a < h->D; a++) if (p->box[a] < 2.0 * maxcut * (1.0 + 1.0 / 1024.0)) ok = false;
        h->single_cell = ok;
        if (!ok) {
            h->mc_ok = false;
            nmc = 0;
            for (int a = 0; a < h->D; a++) { h->grid.n[a] = 0; h->grid.w[a] = 0.0; }
        }
    }
    if (h->D == 2) { h->grid.n[2] = 1; h->grid.w[2] = 1.0; h->grid.shift[2] = 0.0; h->grid.L[2] = 1.0; }
    h->ncell_mc = h->mc_ok ? (int)nmc : 0;

    int rc = set_device(h);
    if (rc) { std::string e = h->err; delete h; g_err = e; return rc; }
#define CR(call)                                                                      \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) {                                                       \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            pmc_destroy(h);                                                           \
            return PMC_ERR_HIP;                                                       \
        }                                                                             \
    } while (0)
    const size_t N = (size_t)h->N;
    CR(hipStreamCreate(&h->stream));
    for (int k = 0; k < 4; k++) CR(hipEventCreate(&h->ev[k]));
    CR(hipMalloc(&h->d_pos, N * h->D * sizeof(double)));
    CR(hipMalloc(&h->d_species, N * sizeof(int)));
    CR(hipMalloc(&h->d_models, (size_t)h->S * h->S * sizeof(DevModel)));
    std::vector<DevModel> dm;
    for (auto &m : h->models) dm.push_back(to_dev(m));
    CR(hipMemcpy(h->d_models, dm.data(), dm.size() * sizeof(DevModel), hipMemcpyHostToDevice));
    const size_t ncmax = (size_t)std::max(h->ncell_ref, h->ncell_mc) + 1;
    CR(hipMalloc(&h->d_cs, N * sizeof(int)));
    CR(hipMalloc(&h->d_head, (size_t)h->ncell_ref * sizeof(int)));
    CR(hipMalloc(&h->d_list, N * sizeof(int)));
    CR(hipMalloc(&h->d_ccount, (size_t)h->ncell_ref * sizeof(int)));
    host_neighbour_cells(h);
    CR(hipMalloc(&h->d_nbr, h->h_nbr.size() * sizeof(int)));
    CR(hipMalloc(&h->d_nbrcnt, h->h_nbrcnt.size() * sizeof(int)));
    CR(hipMemcpy(h->d_nbr, h->h_nbr.data(), h->h_nbr.size() * sizeof(int), hipMemcpyHostToDevice));
    CR(hipMemcpy(h->d_nbrcnt, h->h_nbrcnt.data(), h->h_nbrcnt.size() * sizeof(int), hipMemcpyHostToDevice));
    CR(hipMalloc(&h->d_sp_off, (size_t)(h->S + 1) * sizeof(int)));
    CR(hipMalloc(&h->d_sp_ids, N * sizeof(int)));
    CR(hipMalloc(&h->d_sp_heads, N * sizeof(int)));
    CR(hipMalloc(&h->d_energy, sizeof(double)));
    CR(hipMemset(h->d_energy, 0, sizeof(double)));
    CR(hipMalloc(&h->d_cell_of, N * sizeof(int)));
    CR(hipMalloc(&h->d_count, ncmax * sizeof(int)));
    CR(hipMalloc(&h->d_fill, ncmax * sizeof(int)));
    CR(hipMalloc(&h->d_tmp_ids, N * sizeof(int)));
    CR(hipMalloc(&h->d_arrival, N * sizeof(int)));
    CR(hipMalloc(&h->d_tile_sum, (ncmax / PMC_SCAN_TILE + 2) * sizeof(int)));
    CR(hipMalloc(&h->d_order, ncmax * sizeof(int)));
    CR(hipMalloc(&h->d_hist, (8 * 256 + 1) * sizeof(int)));
    if (const char *e = getenv("PMC_NO_ORDER")) h->use_order = !(e[0] == '1');
    {
        // Wavefronts per cell.  Every wavefront evaluates one proposal slot alone (wave_delta_e); a team of NW
        // wavefronts evaluates NW slots of its cell speculatively at once and resolves them in order.  Small systems
        // have too few cells per colour to fill 256 CUs with one wavefront each, so they take wide teams; from a few
        // thousand cells per colour on, one wavefront per cell wastes nothing on the resolution step.  Measured on
        // MI355X (r02, 3-D LJ, moves/s for teams of 1 / 2 / 4 / 8): N=32768 (216 cells per colour) 0.54 / 0.85 / 1.21 /
        // 1.30e8; N=110592 (729) 1.7 / 2.7 / 3.5 / 2.9e8; N=262144 (1728) 3.9 / 5.5 / 5.0 / 3.1e8; N=10^6 (6859)
        // 7.6 / 7.1 / 5.3 / 3.3e8.  Multi-species systems follow the same rule since the box filter is applied while staging
        // (Kob-Andersen N=10^6, 5832 cells per colour, Displacement only: 7.1 / 5.9 / 4.6e8 for teams of 1 / 2 / 4; before
        // that change two wavefronts per cell were ahead there, 4.0 vs 5.9e8).
        long long nact = 1;
        for (int a = 0; a < h->D; a++) nact *= std::max(1, h->grid.n[a] / 2);
        // End of round 2 (10^8 moves/s in the liquid, teams of 1 / 2 / 4 / 8): 343 cells per colour 1.3 / 1.5 / 2.0 / 2.1; 729:
        // 2.5 / 2.9 / 3.7 / 2.9; 1331: 4.2 / 4.8 / 4.7 / 2.9; 1728: 5.3 / 6.1 / 5.2 / 3.1; 2197: 6.7 / 6.9; 3375: 9.7 / 7.5 / 5.5 / 3.3.
        // Round 3: the teams are compiled for three wavefronts per SIMD (PMC_TEAM_MINWAVES: no scratch), i.e. twelve per CU -- one
        // team of eight, three of four or six of two -- so eight wavefronts per cell only pay while one colour's cells fit the
        // chip one per CU.  10^8 moves/s in the liquid, teams of 8 / 4 / 2 at that budget: 27 cells per colour 0.181 / 0.165 / 0.119;
        // 216: 1.32 / 1.26 / 0.91; 512: 1.73 / 2.04 / 1.48; 729: 1.81 / 3.70 / 2.93; 1331: 1.77 / 3.96 / 4.84; 1728: 1.93 / 4.30 / 5.72.
        h->team = team_rule(h->D, nact);
        if (const char *e = getenv("PMC_TEAM")) { int t = atoi(e); if (t == 1 || t == 2 || t == 4 || t == 8) h->team = t; }
    }
    CR(hipMalloc(&h->d_start_ref, ((size_t)h->ncell_ref + 1) * sizeof(int)));
    CR(hipMalloc(&h->d_ids_ref, N * sizeof(int)));
    if (h->mc_ok) {
        CR(hipMalloc(&h->d_start, ((size_t)h->ncell_mc + 1) * sizeof(int)));
        CR(hipMalloc(&h->d_ids, N * sizeof(int)));
        CR(hipMalloc(&h->d_sx, N * sizeof(double)));
        CR(hipMalloc(&h->d_sy, N * sizeof(double)));
        CR(hipMalloc(&h->d_sz, N * sizeof(double)));
        CR(hipMalloc(&h->d_ssp, N));
        CR(hipMalloc(&h->d_img, N * sizeof(int)));
        CR(hipMalloc(&h->d_simg, N * sizeof(int)));
        {
            std::vector<int> z((size_t)N, PMC_IMG_ZERO);
            CR(hipMemcpy(h->d_img, z.data(), N * sizeof(int), hipMemcpyHostToDevice));
        }
        CR(hipMalloc(&h->d_slot_f, 4 * N * sizeof(double)));
        CR(hipMalloc(&h->d_slot_w, 2 * N * sizeof(uint32_t)));
        CR(hipMalloc(&h->d_slot_m, N));
        CR(hipMalloc(&h->d_done, (size_t)h->ncell_mc * sizeof(unsigned int)));
        CR(hipMemset(h->d_done, 0, (size_t)h->ncell_mc * sizeof(unsigned int)));
        CR(hipMalloc(&h->d_ticket, PMC_TICKET_LANES * PMC_TICKET_STRIDE * sizeof(unsigned long long)));
        CR(hipMemset(h->d_ticket, 0, PMC_TICKET_LANES * PMC_TICKET_STRIDE * sizeof(unsigned long long)));
        {
            // one ticket counter per XCD only on a device that reports eight of them (the scheme's progress argument assumes
            // workgroups are dealt to the XCDs round-robin in launch order: DESIGN.md 4.3); anything else takes the single counter
            int nxcc = 0;
            if (hipDeviceGetAttribute(&nxcc, hipDeviceAttributeNumberOfXccs, h->p.device) != hipSuccess) nxcc = 0;
            h->ticket_lanes = (nxcc == PMC_TICKET_LANES) ? PMC_TICKET_LANES : 1;
        }
        if (const char *e = getenv("PMC_TICKETS")) h->ticket_lanes = atoi(e) == 1 ? 1 : PMC_TICKET_LANES;
        if (const char *e = getenv("PMC_NO_FUSED")) h->use_fused = !(e[0] == '1');
        if (const char *e = getenv("PMC_FUSED_LPT")) h->fused_lpt = (e[0] == '1') ? 1 : 0;
        if (const char *e = getenv("PMC_ROUNDS")) {      // experiments; the API is pmc_set_rounds
            const int rc = pmc_set_rounds(h, atoi(e));
            if (rc) { g_err = h->err; pmc_destroy(h); return rc; }
        }
    }
    CR(hipMalloc(&h->d_stats, (size_t)PMC_NSHARD * PMC_NSTATS * sizeof(unsigned long long)));
    CR(hipMemset(h->d_stats, 0, (size_t)PMC_NSHARD * PMC_NSTATS * sizeof(unsigned long long)));
    CR(hipMalloc(&h->d_result, 8 * sizeof(double)));
    CR(hipHostMalloc(&h->h_stats, PMC_NSTATS * sizeof(unsigned long long)));
#undef CR
    *out = h;
    return PMC_OK;
}

extern "C" int pmc_destroy(pmc_system *h)
{
    if (!h) return PMC_OK;
    (void)hipSetDevice(h->p.device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);   // a sweep may still be in flight
    void *ptrs[] = {h->d_pos, h->d_species, h->d_models, h->d_cs, h->d_head, h->d_list, h->d_ccount, h->d_nbr,
                    h->d_nbrcnt, h->d_sp_off, h->d_sp_ids, h->d_sp_heads, h->d_energy, h->d_cell_of, h->d_count,
                    h->d_fill, h->d_tmp_ids, h->d_arrival, h->d_tile_sum, h->d_order, h->d_hist, h->d_start_ref, h->d_ids_ref, h->d_start, h->d_ids, h->d_sx, h->d_sy,
                    h->d_sz, h->d_ssp, h->d_stats, h->d_cbtab, h->d_result, h->d_done, h->d_ticket, h->d_slot_f, h->d_slot_w, h->d_slot_m, h->d_img, h->d_simg,
                    h->d_cl_items, h->d_vl_nb, h->d_vl_cnt, h->d_ref_err, h->d_vl_last,
                    h->d_sq_slots[0], h->d_sq_slots[1], h->d_sq_gen, h->d_sq_M, h->d_sq_cp, h->d_sq_moveto, h->d_sq_cellhead, h->d_sq_cellnext, h->d_sq_log, h->d_sq_ctl, h->d_sq_args};
    for (void *q : ptrs) if (q) (void)hipFree(q);
    if (h->h_stats) (void)hipHostFree(h->h_stats);
    if (h->h_sq_ctl) (void)hipHostFree(h->h_sq_ctl);
    if (h->h_sq_log) (void)hipHostFree(h->h_sq_log);
    if (h->sq_graph) (void)hipGraphExecDestroy(h->sq_graph);
    for (int k = 0; k < 4; k++) if (h->ev[k]) (void)hipEventDestroy(h->ev[k]);
    for (hipEvent_t e : h->pass_ev) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return PMC_OK;
}

// ---- state -----------------------------------------------------------------------

static int build_species_list(pmc_system *h, const std::vector<int> &sp0)
{
    // SpeciesList(species), src/utils.jl:36-49 (0-based on the device)
    std::vector<int> off(h->S + 1, 0), ids(h->N), heads(h->N), fill(h->S, 0);
    for (int i = 0; i < h->N; i++) off[sp0[i] + 1]++;
    for (int s = 0; s < h->S; s++) off[s + 1] += off[s];
    for (int i = 0; i < h->N; i++) {
        int s = sp0[i];
        heads[i] = fill[s];
        ids[off[s] + fill[s]++] = i;
    }
    HIPCHK(h, hipMemcpy(h->d_sp_off, off.data(), off.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_sp_ids, ids.data(), ids.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_sp_heads, heads.data(), heads.size() * sizeof(int), hipMemcpyHostToDevice));
    h->species_list_built = true;
    h->species_list_stale = false;
    return PMC_OK;
}

// The throughput engine swaps species cell-locally and does not maintain the global SpeciesList; the next
// consumer (sequential chain, swap verbs, pmc_get_species_ids) gets SpeciesList(species) of the current
// state, i.e. what the reference constructor (src/utils.jl:36-49) would build from it.
static int ensure_species_list(pmc_system *h)
{
    if (!h->species_list_stale) return PMC_OK;
    std::vector<int> sp0(h->N);
    HIPCHK(h, hipMemcpy(sp0.data(), h->d_species, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
    return build_species_list(h, sp0);
}

extern "C" int pmc_upload(pmc_system *h, const double *pos, const int32_t *species)
{
    if (!h || !pos) return fail(h, PMC_ERR_BADARG, "pmc_upload: null argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = set_device(h);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(h->d_pos, pos, (size_t)h->N * h->D * sizeof(double), hipMemcpyHostToDevice));
    if (species) {
        std::vector<int> sp0(h->N);
        for (int i = 0; i < h->N; i++) {
            if (species[i] < 1 || species[i] > h->S) return fail(h, PMC_ERR_BADARG, "pmc_upload: species must be 1..nspecies");
            sp0[i] = species[i] - 1;
        }
        HIPCHK(h, hipMemcpy(h->d_species, sp0.data(), (size_t)h->N * sizeof(int), hipMemcpyHostToDevice));
        rc = build_species_list(h, sp0);
        if (rc) return rc;
    } else if (!h->species_list_built) {
        std::vector<int> sp0(h->N, 0);
        HIPCHK(h, hipMemcpy(h->d_species, sp0.data(), (size_t)h->N * sizeof(int), hipMemcpyHostToDevice));
        rc = build_species_list(h, sp0);
        if (rc) return rc;
    }
    h->img_valid = false;      // (re-derived by the next throughput-engine entry: ensure_images)
    h->state_invalid = false;
    h->cells_built = false;
    h->mc_binned = false;
    h->e_fix_valid = false;
    return PMC_OK;
}

extern "C" int pmc_download(pmc_system *h, double *pos, int32_t *species)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_download: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = set_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (pos) HIPCHK(h, hipMemcpy(pos, h->d_pos, (size_t)h->N * h->D * sizeof(double), hipMemcpyDeviceToHost));
    if (species) {
        std::vector<int> sp0(h->N);
        HIPCHK(h, hipMemcpy(sp0.data(), h->d_species, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->N; i++) species[i] = sp0[i] + 1;
    }
    return PMC_OK;
}

// Schedule parameter m of the throughput engine: the sweeps of a call are taken in passes of min(m, sweeps
// left) sweeps; a pass draws one grid shift, bins once, and every cell visit runs that many rounds of n_c proposal slots
// against the neighbourhood it staged once.  m = 1 (default) re-shifts the grid every N attempted moves.
extern "C" int pmc_set_rounds(pmc_system *h, int32_t rounds)
{
    if (!h || rounds < 1 || rounds > PMC_MAX_ROUNDS) return fail(h, PMC_ERR_BADARG, "pmc_set_rounds: rounds must be 1..16");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->mc_ok) return fail(h, PMC_ERR_UNSUPPORTED, "checkerboard grid needs >= 4 cells of width >= rcut per side");
    if ((long long)h->N * rounds > 0x7fffffffLL) return fail(h, PMC_ERR_BADARG, "pmc_set_rounds: N * rounds exceeds the 32-bit slot index");
    if (rounds > h->slot_rounds) {
        int rc = set_device(h);
        if (rc) return rc;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        const size_t n = (size_t)h->N * (size_t)rounds;
        double *f = nullptr; uint32_t *w = nullptr; unsigned char *m = nullptr;
        if (hipMalloc(&f, 4 * n * sizeof(double)) != hipSuccess || hipMalloc(&w, 2 * n * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc(&m, n) != hipSuccess) {
            if (f) (void)hipFree(f);
            if (w) (void)hipFree(w);
            if (m) (void)hipFree(m);
            return fail(h, PMC_ERR_HIP, "pmc_set_rounds: out of device memory for the proposal slots");
        }
        (void)hipFree(h->d_slot_f); (void)hipFree(h->d_slot_w); (void)hipFree(h->d_slot_m);
        h->d_slot_f = f; h->d_slot_w = w; h->d_slot_m = m;
        h->slot_rounds = rounds;
    }
    h->rounds = rounds;
    return PMC_OK;
}
extern "C" int pmc_get_rounds(pmc_system *h, int32_t *rounds)
{
    if (!h || !rounds) return fail(h, PMC_ERR_BADARG, "pmc_get_rounds: null argument");
    *rounds = h->rounds;
    return PMC_OK;
}

extern "C" int pmc_set_temperature(pmc_system *h, double T)
{
    if (!h || !(T > 0.0)) return fail(h, PMC_ERR_BADARG, "pmc_set_temperature: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    h->p.temperature = T;
    return PMC_OK;
}
extern "C" int pmc_get_temperature(pmc_system *h, double *T)
{
    if (!h || !T) return fail(h, PMC_ERR_BADARG, "pmc_get_temperature: null argument");
    *T = h->p.temperature;
    return PMC_OK;
}
static int refresh_stale_energy(pmc_system *h);

extern "C" int pmc_set_energy(pmc_system *h, double e)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_set_energy: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    // After a throughput sweep that failed part-way the positions have moved and the caller's value belongs to the state
    // before the call: it is not taken, and the running energy stays marked for recomputation (cb_invalidate).
    if (h->state_invalid) return PMC_OK;
    h->energy = e;
    h->energy_stale = false;
    return PMC_OK;
}
extern "C" int pmc_get_energy(pmc_system *h, double *e)
{
    if (!h || !e) return fail(h, PMC_ERR_BADARG, "pmc_get_energy: null argument");
    if (int rc = refresh_stale_energy(h)) return rc;
    *e = h->energy;
    return PMC_OK;
}
extern "C" int pmc_set_counters(pmc_system *h, uint64_t step, uint64_t sweep)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_set_counters: null handle");
    h->step_ctr = step;
    h->sweep_ctr = sweep;
    return PMC_OK;
}
extern "C" int pmc_get_counters(pmc_system *h, uint64_t *step, uint64_t *sweep)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_get_counters: null handle");
    if (step) *step = h->step_ctr;
    if (sweep) *sweep = h->sweep_ctr;
    return PMC_OK;
}

// ---- alternative list types of the reference -----------------------------------------------

extern "C" int pmc_cell_list_geometry(int32_t d, const double *box, double rcut, double *cell, int64_t *ncells)
{
    // CellList(box, rcut, N), src/neighbours.jl:126-133: box / floor(Int, box / rcut); max(1, floor(Int, box / cell))
    if (!box || !cell || !ncells || d < 1 || d > 3 || !(rcut > 0.0)) return fail(nullptr, PMC_ERR_BADARG, "pmc_cell_list_geometry: bad argument");
    for (int a = 0; a < d; a++) {
        const double nf = std::floor(box[a] / rcut);
        if (!(nf >= 1.0)) return fail(nullptr, PMC_ERR_BADARG, "CellList: rcut larger than the box (division by zero cells)");
        cell[a] = box[a] / (double)(int64_t)nf;
        const double q = std::floor(box[a] / cell[a]);
        ncells[a] = q >= 1.0 ? (int64_t)q : 1;
    }
    return PMC_OK;
}

extern "C" int pmc_verlet_list_geometry(int32_t d, const double *box, double rcut, double dr, double *cell, int64_t *ncells)
{
    // VerletList(box, rcut, N; dr), src/neighbours.jl:405-406: the LinkedList formula with rcut + dr
    if (!(dr > 0.0)) return fail(nullptr, PMC_ERR_BADARG, "Verlet list must have dr as a parameter");
    return pmc_linked_list_geometry(d, box, rcut + dr, cell, ncells);
}

extern "C" int pmc_set_list_type(pmc_system *h, int32_t list_type, double dr)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_set_list_type: null handle");
    if (list_type < 0 || list_type > 3) return fail(h, PMC_ERR_BADARG, "pmc_set_list_type: 0 EmptyList, 1 LinkedList, 2 CellList, 3 VerletList");
    if (list_type == 3 && !(dr > 0.0)) return fail(h, PMC_ERR_BADARG, "Verlet list must have dr as a parameter");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = set_device(h);
    if (rc) return rc;
    h->list_type = list_type;
    h->list_dr = dr;
    h->cells_built = false;
    const size_t N = (size_t)h->N;
    if (!h->d_ref_err) {
        HIPCHK(h, hipMalloc(&h->d_ref_err, 16));
        HIPCHK(h, hipMemset(h->d_ref_err, 0, 16));
    }
    if (list_type >= 2 && !h->d_cl_items) {
        const double mean = (double)h->N / (double)h->ncell_ref;
        h->cl_cap = std::max(32, (int)(3.0 * mean) + 32);
        HIPCHK(h, hipMalloc(&h->d_cl_items, (size_t)h->ncell_ref * h->cl_cap * sizeof(int)));
    }
    if (list_type == 3) {
        double vol = 1.0;
        for (int a = 0; a < h->D; a++) vol *= h->p.box[a];
        const double r = h->maxcut + dr;
        const double sphere = (h->D == 3) ? 4.18879020478639 * r * r * r : 3.14159265358979 * r * r;
        const int cap = std::max(64, (int)(2.5 * sphere * (double)h->N / vol) + 32);
        if (cap != h->vl_cap) {
            if (h->d_vl_nb) (void)hipFree(h->d_vl_nb);
            h->d_vl_nb = nullptr;
            h->vl_cap = cap;
        }
        if (!h->d_vl_nb) HIPCHK(h, hipMalloc(&h->d_vl_nb, N * (size_t)h->vl_cap * sizeof(int)));
        if (!h->d_vl_cnt) HIPCHK(h, hipMalloc(&h->d_vl_cnt, N * sizeof(int)));
        if (!h->d_vl_last) HIPCHK(h, hipMalloc(&h->d_vl_last, N * h->D * sizeof(double)));
    }
    return PMC_OK;
}

static int ref_list_errors(pmc_system *h)
{
    if (!h->d_ref_err) return PMC_OK;
    int e = 0;
    HIPCHK(h, hipMemcpy(&e, h->d_ref_err, sizeof(int), hipMemcpyDeviceToHost));
    if (e) {
        (void)hipMemset(h->d_ref_err, 0, sizeof(int));
        if ((e & 4) && h->list_type == 1) {
            // a LinkedList's row (the copy of its chains the fast sequential engines read: ref_relink) was full when the literal
            // loop moved a particle into its cell.  head / list / cs -- the list itself -- are right and the move stands; the
            // rows are not, so they are dropped: the next fast sequential call re-creates them FROM THE CHAINS (never from the
            // positions: moves have shaped the chains' order, seq_alloc_rows) with twice the capacity.  Not an error of the call.
            if (h->d_cl_items) (void)hipFree(h->d_cl_items);
            h->d_cl_items = nullptr;
            h->rows_min_cap = std::max(h->rows_min_cap, 2 * h->cl_cap);
            h->cl_cap = 0;
            if (e == 4) return PMC_OK;
        }
        h->cells_built = false;
        return fail(h, PMC_ERR_UNSUPPORTED, (e & 2) ? "VerletList: a neighbour vector exceeded its device capacity"
                                                   : "CellList: a cell vector exceeded its device capacity");
    }
    return PMC_OK;
}

extern "C" int pmc_get_cell_members(pmc_system *h, int64_t c, int64_t *out, int64_t cap, int64_t *n)
{
    if (!h || !out || !n || c < 1 || c > h->ncell_ref) return fail(h, PMC_ERR_BADARG, "pmc_get_cell_members: bad argument");
    if (h->list_type < 2) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_get_cell_members: only CellList and VerletList keep cells[c]");
    if (!h->cells_built) { int rb = pmc_build_cells(h); if (rb) return rb; }
    int rc = set_device(h);
    if (rc) return rc;
    int cnt = 0;
    HIPCHK(h, hipMemcpy(&cnt, h->d_ccount + (c - 1), sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> t((size_t)std::max(cnt, 1));
    if (cnt) HIPCHK(h, hipMemcpy(t.data(), h->d_cl_items + (size_t)(c - 1) * h->cl_cap, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost));
    for (int k = 0; k < cnt && k < cap; k++) out[k] = (int64_t)t[k] + 1;
    *n = cnt;
    return PMC_OK;
}

extern "C" int pmc_get_verlet_neighbours(pmc_system *h, int64_t i, int64_t *out, int64_t cap, int64_t *n)
{
    if (!h || !out || !n || i < 1 || i > h->N) return fail(h, PMC_ERR_BADARG, "pmc_get_verlet_neighbours: bad argument");
    if (h->list_type != 3) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_get_verlet_neighbours: the list type is not VerletList");
    if (!h->cells_built) { int rb = pmc_build_cells(h); if (rb) return rb; }
    int rc = set_device(h);
    if (rc) return rc;
    int cnt = 0;
    HIPCHK(h, hipMemcpy(&cnt, h->d_vl_cnt + (i - 1), sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> t((size_t)std::max(cnt, 1));
    if (cnt) HIPCHK(h, hipMemcpy(t.data(), h->d_vl_nb + (size_t)(i - 1) * h->vl_cap, (size_t)cnt * sizeof(int), hipMemcpyDeviceToHost));
    for (int k = 0; k < cnt && k < cap; k++) out[k] = (int64_t)t[k] + 1;
    *n = cnt;
    return PMC_OK;
}

// ---- binning ------------------------------------------------------------------------

static inline int nblk(int n, int b) { return (n + b - 1) / b; }

// exclusive scan of d_count into d_start; zero != 0 also clears d_count for the next binning pass
static int scan_counts(pmc_system *h, int ncell, int *d_start, int zero = 0)
{
    if (ncell <= PMC_SCAN_SMALL_MAX) {
        hipLaunchKernelGGL(k_scan_small, dim3(1), dim3(1024), 0, h->stream, h->d_count, ncell, d_start, zero);
        HIPCHK(h, hipGetLastError());
        return PMC_OK;
    }
    const int ntiles = nblk(ncell, PMC_SCAN_TILE);
    int *d_total = h->d_hist + 8 * 256;
    hipLaunchKernelGGL(k_scan_tiles, dim3(ntiles), dim3(1024), 0, h->stream, h->d_count, ncell, h->d_tile_sum);
    hipLaunchKernelGGL(k_scan_tops, dim3(1), dim3(1024), 0, h->stream, h->d_tile_sum, ntiles, d_total);
    hipLaunchKernelGGL(k_scan_apply, dim3(ntiles), dim3(1024), 0, h->stream, h->d_count, ncell, h->d_tile_sum, d_total,
                       d_start, zero);
    HIPCHK(h, hipGetLastError());
    return PMC_OK;
}

// cell_of (already filled) -> start / ids (ascending ids inside each cell)
static int bin_by_cell(pmc_system *h, int ncell, int *d_start, int *d_ids)
{
    const int N = h->N;
    HIPCHK(h, hipMemsetAsync(h->d_count, 0, ((size_t)ncell + 1) * sizeof(int), h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_fill, 0, ((size_t)ncell + 1) * sizeof(int), h->stream));
    h->count_is_zero = false;
    hipLaunchKernelGGL(k_count, dim3(nblk(N, 256)), dim3(256), 0, h->stream, h->d_cell_of, N, h->d_count);
    if (int rc = scan_counts(h, ncell, d_start)) return rc;
    hipLaunchKernelGGL(k_scatter, dim3(nblk(N, 256)), dim3(256), 0, h->stream, h->d_cell_of, N, d_start, h->d_fill,
                       h->d_tmp_ids);
    hipLaunchKernelGGL(k_rank, dim3(nblk(N, 256)), dim3(256), 0, h->stream, h->d_cell_of, N, d_start, h->d_tmp_ids,
                       d_ids);
    HIPCHK(h, hipGetLastError());
    return PMC_OK;
}

static RefArgs ref_args(pmc_system *h);
static int seq_build_rows(pmc_system *h);

extern "C" int pmc_build_cells(pmc_system *h)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_build_cells: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = set_device(h);
    if (rc) return rc;
    rc = ensure_species_list(h);
    if (rc) return rc;
    const int N = h->N;
    const pmc_params &p = h->p;
    if (h->D == 3)
        hipLaunchKernelGGL(k_ref_cell_index<3>, dim3(nblk(N, 256)), dim3(256), 0, h->stream, h->d_pos, N, p.cell[0],
                           p.cell[1], p.cell[2], (int)p.ncells[0], (int)p.ncells[1], (int)p.ncells[2], h->d_cell_of);
    else
        hipLaunchKernelGGL(k_ref_cell_index<2>, dim3(nblk(N, 256)), dim3(256), 0, h->stream, h->d_pos, N, p.cell[0],
                           p.cell[1], 1.0, (int)p.ncells[0], (int)p.ncells[1], 1, h->d_cell_of);
    rc = bin_by_cell(h, h->ncell_ref, h->d_start_ref, h->d_ids_ref);
    if (rc) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_cs, h->d_cell_of, (size_t)N * sizeof(int), hipMemcpyDeviceToDevice, h->stream));
    int tot = std::max(N, h->ncell_ref);
    hipLaunchKernelGGL(k_ref_links, dim3(nblk(tot, 256)), dim3(256), 0, h->stream, h->d_cell_of, N, h->d_start_ref,
                       h->d_ids_ref, h->ncell_ref, h->d_head, h->d_list, h->d_ccount);
    if (h->list_type >= 2) {
        // CellList / VerletList: cells[c] vectors in ascending id (build_neighbour_list!, src/neighbours.jl:145-156, 431-435).
        // The reference's vectors grow as they must; the device rows are re-allocated with twice the capacity until the fullest
        // cell fits (found by tests/test_gpu_seq_fuzz.py: CellList's floor(Int, box / cell) can come out one short of box / rcut's -- box /
        // (box / 7) = 6.999... -- and the last slab of every dimension then folds onto the first: cells of 2, 4 and 8 x the mean)
        bool headroom = false;
        for (int attempt = 0; ; attempt++) {
            hipLaunchKernelGGL(k_ref_cellvec, dim3(nblk(tot, 256)), dim3(256), 0, h->stream, N, h->d_cell_of, h->d_start_ref,
                               h->d_ids_ref, h->ncell_ref, h->d_cl_items, h->cl_cap, h->d_ccount, h->d_ref_err);
            HIPCHK(h, hipGetLastError());
            int e = 0;
            HIPCHK(h, hipMemcpyAsync(&e, h->d_ref_err, sizeof(int), hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            // (rows that had to grow get one doubling more than the fullest cell needs: the chain that follows moves particles
            // into that cell too, and a row that overflows mid-chain ends the call with PMC_ERR_STATE)
            if (!(e & 1) && (attempt == 0 || headroom)) break;
            if (!(e & 1)) headroom = true;
            if (attempt >= 12 || (double)h->ncell_ref * 2.0 * h->cl_cap * sizeof(int) > 32e9) break;   // reported below
            HIPCHK(h, hipMemsetAsync(h->d_ref_err, 0, sizeof(int), h->stream));
            (void)hipFree(h->d_cl_items);
            h->d_cl_items = nullptr;
            h->cl_cap *= 2;
            HIPCHK(h, hipMalloc(&h->d_cl_items, (size_t)h->ncell_ref * h->cl_cap * sizeof(int)));
        }
    } else if (h->list_type == 1 && h->d_cl_items) {
        // LinkedList rows of the fast sequential engine (pmc_seq.h): the chain of a cell read backwards = ids ascending.  Whenever
        // rows exist they are rebuilt here -- also rows a CellList / VerletList allocated before the handle was switched back to
        // LinkedList (pmc_set_list_type): ref_relink edits them and the fast engines read them, so they are never left stale
        rc = seq_build_rows(h);
        if (rc) return rc;
    }
    if (h->list_type == 3) {
        RefArgs a = ref_args(h);
        if (h->D == 3) hipLaunchKernelGGL(k_ref_verlet_build<3>, dim3(nblk(N, 64)), dim3(64), 0, h->stream, a);
        else hipLaunchKernelGGL(k_ref_verlet_build<2>, dim3(nblk(N, 64)), dim3(64), 0, h->stream, a);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->cells_built = true;
    return ref_list_errors(h);
}

extern "C" int pmc_get_cells(pmc_system *h, int64_t *cs)
{
    if (!h || !cs) return fail(h, PMC_ERR_BADARG, "pmc_get_cells: null argument");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<int> t(h->N);
    HIPCHK(h, hipMemcpy(t.data(), h->d_cs, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->N; i++) cs[i] = (int64_t)t[i] + 1;
    return PMC_OK;
}

extern "C" int pmc_get_linked_list(pmc_system *h, int64_t *head, int64_t *list)
{
    if (!h || !head || !list) return fail(h, PMC_ERR_BADARG, "pmc_get_linked_list: null argument");
    if (h->list_type != 1) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_get_linked_list: the list type is not LinkedList");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<int> t(std::max(h->N, h->ncell_ref));
    HIPCHK(h, hipMemcpy(t.data(), h->d_head, (size_t)h->ncell_ref * sizeof(int), hipMemcpyDeviceToHost));
    for (int c = 0; c < h->ncell_ref; c++) head[c] = t[c] < 0 ? -1 : (int64_t)t[c] + 1;
    HIPCHK(h, hipMemcpy(t.data(), h->d_list, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->N; i++) list[i] = t[i] < 0 ? -1 : (int64_t)t[i] + 1;
    return PMC_OK;
}

extern "C" int pmc_get_neighbour_cells(pmc_system *h, int64_t c, int64_t *out, int32_t *n)
{
    if (!h || !out || !n || c < 1 || c > h->ncell_ref) return fail(h, PMC_ERR_BADARG, "pmc_get_neighbour_cells: bad argument");
    int cnt = h->h_nbrcnt[c - 1];
    for (int k = 0; k < cnt; k++) out[k] = (int64_t)h->h_nbr[(size_t)(c - 1) * h->nn + k] + 1;
    *n = cnt;
    return PMC_OK;
}

// (L::LinkedList)(system, i) (src/neighbours.jl:371-377) materialised: the ids the LinkedIterator (:316-365) yields for
// particle i -- the chains of the neighbour cells of the cell that holds i, in table order, every chain from its head;
// i itself is among them, as in the reference.  Compatibility path of hosts that iterate the list themselves (two
// device-to-host copies of size N per call): the energy loops of this library never materialise it.
extern "C" int pmc_neighbours_of(pmc_system *h, int64_t i, int64_t *out, int64_t cap, int64_t *n)
{
    if (!h || !n || i < 1 || i > h->N || (cap > 0 && !out)) return fail(h, PMC_ERR_BADARG, "pmc_neighbours_of: bad argument");
    if (h->list_type != 1) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_neighbours_of: the list type is not LinkedList");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<int> head(h->ncell_ref), list(h->N);
    int c = 0;
    HIPCHK(h, hipMemcpy(head.data(), h->d_head, (size_t)h->ncell_ref * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(list.data(), h->d_list, (size_t)h->N * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(h, hipMemcpy(&c, h->d_cs + (i - 1), sizeof(int), hipMemcpyDeviceToHost));   // the list is current: cs[i] is the cell of x_i
    int64_t cnt = 0;
    for (int k = 0; k < h->h_nbrcnt[c]; k++) {
        for (int j = head[h->h_nbr[(size_t)c * h->nn + k]]; j >= 0; j = list[j]) {
            if (cnt < cap) out[cnt] = (int64_t)j + 1;
            cnt++;
        }
    }
    *n = cnt;
    return cnt > cap ? fail(h, PMC_ERR_BADARG, "pmc_neighbours_of: the output buffer is too small (n holds the needed size)") : PMC_OK;
}

static RefArgs ref_args(pmc_system *h)
{
    RefArgs a;
    memset(&a, 0, sizeof a);
    a.N = h->N; a.S = h->S;
    for (int q = 0; q < 3; q++) {
        a.box[q] = q < h->D ? h->p.box[q] : 1.0;
        a.cell[q] = q < h->D ? h->p.cell[q] : 1.0;
        a.nc[q] = q < h->D ? (int)h->p.ncells[q] : 1;
    }
    a.nnmax = h->nn;
    a.pos = h->d_pos; a.species = h->d_species; a.models = h->d_models;
    a.cs = h->d_cs; a.head = h->d_head; a.list = h->d_list; a.ccount = h->d_ccount;
    a.nbr = h->d_nbr; a.nbrcnt = h->d_nbrcnt;
    a.sp_off = h->d_sp_off; a.sp_ids = h->d_sp_ids; a.sp_heads = h->d_sp_heads;
    a.energy = h->d_energy;
    a.temperature = h->p.temperature;
    a.list_type = h->list_type;
    a.cl_items = h->d_cl_items; a.cl_cap = h->cl_cap;
    a.vl_nb = h->d_vl_nb; a.vl_cnt = h->d_vl_cnt; a.vl_cap = h->vl_cap; a.vl_last = h->d_vl_last;
    a.dr2o4 = (h->list_dr * h->list_dr) / 4;                                   // src/neighbours.jl:415
    a.rcutoff2 = (h->maxcut + h->list_dr) * (h->maxcut + h->list_dr);          // (rcut + dr)^2, :438
    a.err = h->d_ref_err;
    return a;
}

static int ref_energy_launch(pmc_system *h, const int *d_which, int n, double *d_out)
{
    RefArgs a = ref_args(h);
    int grid = std::min(n, 65536 * 4);
    if (h->D == 3) hipLaunchKernelGGL(k_ref_energy<3>, dim3(grid), dim3(64), 0, h->stream, a, d_which, n, d_out);
    else hipLaunchKernelGGL(k_ref_energy<2>, dim3(grid), dim3(64), 0, h->stream, a, d_which, n, d_out);
    HIPCHK(h, hipGetLastError());
    return PMC_OK;
}

extern "C" int pmc_energy_particles(pmc_system *h, double *e)
{
    if (!h || !e) return fail(h, PMC_ERR_BADARG, "pmc_energy_particles: null argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    DevTmp out;
    HIPCHK(h, out.alloc((size_t)h->N * sizeof(double)));
    rc = ref_energy_launch(h, nullptr, h->N, out.as<double>());
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(e, out.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost));
    return PMC_OK;
}

// Base sum(::Vector{Float64}) is pairwise with 1024-element leaves (base/reduce.jl)
static double pairwise_sum(const double *v, long long lo, long long hi)
{
    if (hi - lo < 1024) {
        double a = v[lo];
        for (long long k = lo + 1; k <= hi; k++) a += v[k];
        return a;
    }
    long long mid = lo + ((hi - lo) >> 1);
    return pairwise_sum(v, lo, mid) + pairwise_sum(v, mid + 1, hi);
}

extern "C" int pmc_init_energy(pmc_system *h, double *e)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_init_energy: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = pmc_build_cells(h);                       // src/atoms.jl:50
    if (rc) return rc;
    std::vector<double> loc((size_t)h->N);
    rc = pmc_energy_particles(h, loc.data());          // src/atoms.jl:51
    if (rc) return rc;
    double en = pairwise_sum(loc.data(), 0, h->N - 1) / 2.0;   // src/atoms.jl:52
    if (std::isinf(en) || std::isnan(en))              // src/atoms.jl:53-55
        return fail(h, PMC_ERR_NONFINITE, "Initial configuration has infinite or NaN energy.");
    h->energy = en;
    h->energy_stale = false;
    h->e_fix_valid = false;
    if (e) *e = en;
    return PMC_OK;
}

extern "C" int pmc_energy_particle(pmc_system *h, int64_t i, double *e)
{
    if (!h || !e || i < 1 || i > h->N) return fail(h, PMC_ERR_BADARG, "pmc_energy_particle: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    int i0 = (int)i - 1;
    int *d_which = reinterpret_cast<int *>(h->d_result + 4);
    HIPCHK(h, hipMemcpy(d_which, &i0, sizeof(int), hipMemcpyHostToDevice));
    rc = ref_energy_launch(h, d_which, 1, h->d_result);
    if (rc) return rc;
    HIPCHK(h, hipMemcpy(e, h->d_result, sizeof(double), hipMemcpyDeviceToHost));
    return PMC_OK;
}

extern "C" int pmc_species_count(pmc_system *h, int32_t sp, int64_t *n)
{
    if (!h || !n || sp < 1 || sp > h->S) return fail(h, PMC_ERR_BADARG, "pmc_species_count: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    int off[2];
    HIPCHK(h, hipMemcpy(off, h->d_sp_off + (sp - 1), 2 * sizeof(int), hipMemcpyDeviceToHost));
    *n = off[1] - off[0];
    return PMC_OK;
}

extern "C" int pmc_get_species_ids(pmc_system *h, int32_t sp, int64_t *ids)
{
    if (!h || !ids || sp < 1 || sp > h->S) return fail(h, PMC_ERR_BADARG, "pmc_get_species_ids: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    rc = ensure_species_list(h);
    if (rc) return rc;
    int off[2];
    HIPCHK(h, hipMemcpy(off, h->d_sp_off + (sp - 1), 2 * sizeof(int), hipMemcpyDeviceToHost));
    std::vector<int> t((size_t)(off[1] - off[0]));
    if (!t.empty()) HIPCHK(h, hipMemcpy(t.data(), h->d_sp_ids + off[0], t.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < t.size(); k++) ids[k] = (int64_t)t[k] + 1;
    return PMC_OK;
}

// ---- per-move verbs and the sequential chain ------------------------------------------

static MovePool make_pool(const pmc_move *pool, int npool)
{
    MovePool mp;
    memset(&mp, 0, sizeof mp);
    mp.npool = npool;
    for (int m = 0; m < npool; m++) {
        mp.kind[m] = pool[m].kind;
        mp.sp0[m] = pool[m].species[0] - 1;
        mp.sp1[m] = pool[m].species[1] - 1;
        mp.prob[m] = pool[m].probability;
        mp.sigma[m] = pool[m].sigma;
    }
    return mp;
}

// system.energy[1] after a throughput sweep that failed part-way (cb_invalidate): recomputed from the
// current positions (exact sum) before anything reads it
static int refresh_stale_energy(pmc_system *h)
{
    if (!h->energy_stale) return PMC_OK;
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    h->energy_stale = false;
    const int rc = pmc_total_energy(h, nullptr, nullptr);
    if (rc) h->energy_stale = true;
    else h->state_invalid = false;
    return rc;
}

static int ref_op_launch(pmc_system *h, RefOp &o)
{
    if (int rs = refresh_stale_energy(h)) return rs;
    RefArgs a = ref_args(h);
    HIPCHK(h, hipMemcpyAsync(h->d_energy, &h->energy, sizeof(double), hipMemcpyHostToDevice, h->stream));
    if (h->D == 3) hipLaunchKernelGGL(k_ref_ops<3>, dim3(1), dim3(64), 0, h->stream, a, o);
    else hipLaunchKernelGGL(k_ref_ops<2>, dim3(1), dim3(64), 0, h->stream, a, o);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(&h->energy, h->d_energy, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->mc_binned = false;
    h->e_fix_valid = false;
    h->img_valid = false;
    return (h->list_type >= 2 || h->d_cl_items) ? ref_list_errors(h) : PMC_OK;
}

extern "C" int pmc_displacement_perform(pmc_system *h, int64_t i, const double *delta, double *e1, double *e2,
                                        double *de)
{
    if (!h || !delta || i < 1 || i > h->N) return fail(h, PMC_ERR_BADARG, "pmc_displacement_perform: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    RefOp o;
    memset(&o, 0, sizeof o);
    o.op = OP_DISP_PERFORM;
    o.i = (int)i - 1;
    for (int q = 0; q < h->D; q++) o.delta[q] = delta[q];
    o.result = h->d_result;
    rc = ref_op_launch(h, o);
    if (rc) return rc;
    double r[3];
    HIPCHK(h, hipMemcpy(r, h->d_result, sizeof r, hipMemcpyDeviceToHost));
    if (e1) *e1 = r[0];
    if (e2) *e2 = r[1];
    if (de) *de = r[2];
    return PMC_OK;
}

extern "C" int pmc_displacement_revert(pmc_system *h, int64_t i, const double *minus_delta, double de)
{
    if (!h || !minus_delta || i < 1 || i > h->N) return fail(h, PMC_ERR_BADARG, "pmc_displacement_revert: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    RefOp o;
    memset(&o, 0, sizeof o);
    o.op = OP_DISP_REVERT;
    o.i = (int)i - 1;
    for (int q = 0; q < h->D; q++) o.delta[q] = minus_delta[q];
    o.de_in = de;
    return ref_op_launch(h, o);
}

extern "C" int pmc_swap_perform(pmc_system *h, int64_t i, int64_t j, double *e1, double *e2, double *de)
{
    if (!h || i < 1 || i > h->N || j < 1 || j > h->N) return fail(h, PMC_ERR_BADARG, "pmc_swap_perform: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    RefOp o;
    memset(&o, 0, sizeof o);
    o.op = OP_SWAP_PERFORM;
    o.i = (int)i - 1;
    o.j = (int)j - 1;
    o.result = h->d_result;
    rc = ref_op_launch(h, o);
    if (rc) return rc;
    double r[3];
    HIPCHK(h, hipMemcpy(r, h->d_result, sizeof r, hipMemcpyDeviceToHost));
    if (e1) *e1 = r[0];
    if (e2) *e2 = r[1];
    if (de) *de = r[2];
    return PMC_OK;
}

extern "C" int pmc_swap_revert(pmc_system *h, int64_t i, int64_t j, double de)
{
    if (!h || i < 1 || i > h->N || j < 1 || j > h->N) return fail(h, PMC_ERR_BADARG, "pmc_swap_revert: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->cells_built) {   // e.g. after checkerboard sweeps: the reference's list is always current
        int rb = pmc_build_cells(h);
        if (rb) return rb;
    }
    int rc = set_device(h);
    if (rc) return rc;
    RefOp o;
    memset(&o, 0, sizeof o);
    o.op = OP_SWAP_REVERT;
    o.i = (int)i - 1;
    o.j = (int)j - 1;
    o.de_in = de;
    return ref_op_launch(h, o);
}

// ---- throughput engine -----------------------------------------------------------------

// The persistent periodic image of every particle (pmc_kernels.h "images"): k = floor(x / L), derived from the positions
// the throughput engine takes over -- after an upload, after per-move verbs or reference-order sweeps, after an aborted sweep
// -- and then kept for as long as the engine's own state (sorted copy, exact running energy) stays valid.
static int ensure_images(pmc_system *h)
{
    if (h->img_valid || !h->mc_ok) return PMC_OK;
    const int nb = (h->N + 255) / 256;
    int *d_any = reinterpret_cast<int *>(h->d_result);      // (scratch of the per-move verbs: 8 doubles)
    HIPCHK(h, hipMemsetAsync(d_any, 0, sizeof(int), h->stream));
    if (h->D == 3) hipLaunchKernelGGL(k_fold_images<3>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, h->N, h->p.box[0], h->p.box[1], h->p.box[2], h->d_img, d_any);
    else hipLaunchKernelGGL(k_fold_images<2>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, h->N, h->p.box[0], h->p.box[1], 1.0, h->d_img, d_any);
    HIPCHK(h, hipGetLastError());
    int any = 0;
    HIPCHK(h, hipMemcpyAsync(&any, d_any, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->any_img = any != 0;
    h->img_valid = true;
    return PMC_OK;
}

extern "C" int pmc_get_images(pmc_system *h, int32_t *img)
{
    if (!h || !img) return fail(h, PMC_ERR_BADARG, "pmc_get_images: null argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    int rc = set_device(h);
    if (rc) return rc;
    std::fill(img, img + (size_t)h->N * h->D, 0);
    if (!h->mc_ok) return PMC_OK;                  // no throughput grid: no images are ever used
    rc = ensure_images(h);
    if (rc) return rc;
    std::vector<int> t((size_t)h->N);
    HIPCHK(h, hipMemcpy(t.data(), h->d_img, t.size() * sizeof(int), hipMemcpyDeviceToHost));
    for (int i = 0; i < h->N; i++)
        for (int q = 0; q < h->D; q++) img[(size_t)i * h->D + q] = img_get(t[i], q);
    return PMC_OK;
}

extern "C" int pmc_set_images(pmc_system *h, const int32_t *img)
{
    if (!h || !img) return fail(h, PMC_ERR_BADARG, "pmc_set_images: null argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->mc_ok) return PMC_OK;
    int rc = set_device(h);
    if (rc) return rc;
    std::vector<int> t((size_t)h->N);
    bool any = false;
    for (int i = 0; i < h->N; i++) {
        int k[3] = {0, 0, 0};
        for (int q = 0; q < h->D; q++) {
            k[q] = img[(size_t)i * h->D + q];
            if (k[q] < -PMC_IMG_MAX || k[q] > PMC_IMG_MAX) return fail(h, PMC_ERR_BADARG, "pmc_set_images: |image| must be <= 511");
        }
        t[i] = img_pack(k[0], k[1], k[2]);
        any = any || t[i] != PMC_IMG_ZERO;
    }
    HIPCHK(h, hipMemcpy(h->d_img, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice));
    h->any_img = any;
    h->img_valid = true;
    h->mc_binned = false;
    h->e_fix_valid = false;
    return PMC_OK;
}

static int mc_bin(pmc_system *h, const double shift[3], const SlotGen *slots = nullptr)
{
    const int N = h->N;
    if (int ri = ensure_images(h)) return ri;
    const int nb = nblk(N, 256);
    for (int q = 0; q < 3; q++) h->grid.shift[q] = q < h->D ? shift[q] : 0.0;
    SlotGen gen;
    memset(&gen, 0, sizeof gen);
    if (slots) gen = *slots;
    // the scan of the previous throughput binning left the counters zeroed (k_scan_*: zero = 1)
    if (!h->count_is_zero) HIPCHK(h, hipMemsetAsync(h->d_count, 0, ((size_t)h->ncell_mc + 1) * sizeof(int), h->stream));
    unsigned long long *const t8 = h->ticket_lanes == PMC_TICKET_LANES ? h->d_ticket : nullptr;
    const int *const img = h->any_img ? h->d_img : nullptr;
    int *const simg = h->any_img ? h->d_simg : nullptr;
    if (h->D == 3) hipLaunchKernelGGL(k_mc_bin_a<3>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, img, N, h->grid, h->d_cell_of, h->d_count, h->d_arrival, t8);
    else hipLaunchKernelGGL(k_mc_bin_a<2>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, img, N, h->grid, h->d_cell_of, h->d_count, h->d_arrival, t8);
    if (int rc = scan_counts(h, h->ncell_mc, h->d_start, 1)) return rc;
    h->count_is_zero = true;
    hipLaunchKernelGGL(k_mc_bin_b, dim3(nb), dim3(256), 0, h->stream, h->d_cell_of, h->d_arrival, N, h->d_start, h->d_tmp_ids);
    if (h->D == 3)
        hipLaunchKernelGGL(k_mc_bin_c<3>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, h->d_species, h->d_cell_of, N, h->d_start,
                           h->d_tmp_ids, h->d_ids, h->d_sx, h->d_sy, h->d_sz, h->d_ssp, gen, img, simg, h->p.box[0], h->p.box[1], h->p.box[2]);
    else
        hipLaunchKernelGGL(k_mc_bin_c<2>, dim3(nb), dim3(256), 0, h->stream, h->d_pos, h->d_species, h->d_cell_of, N, h->d_start,
                           h->d_tmp_ids, h->d_ids, h->d_sx, h->d_sy, h->d_sz, h->d_ssp, gen, img, simg, h->p.box[0], h->p.box[1], 1.0);
    HIPCHK(h, hipGetLastError());
    h->mc_binned = true;
    return PMC_OK;
}

// per colour: cells in descending population (longest processing time first)
static int mc_order(pmc_system *h, int nactive)
{
    const int ncol = 1 << h->D;
    const int nb = nblk(h->ncell_mc, 256);
    HIPCHK(h, hipMemsetAsync(h->d_hist, 0, 8 * 256 * sizeof(int), h->stream));
    if (h->D == 3) hipLaunchKernelGGL(k_order_hist<3>, dim3(nb), dim3(256), 0, h->stream, h->d_start, h->ncell_mc, h->grid, h->d_hist);
    else hipLaunchKernelGGL(k_order_hist<2>, dim3(nb), dim3(256), 0, h->stream, h->d_start, h->ncell_mc, h->grid, h->d_hist);
    hipLaunchKernelGGL(k_order_scan, dim3(ncol), dim3(64), 0, h->stream, h->d_hist, ncol);
    if (h->D == 3) hipLaunchKernelGGL(k_order_fill<3>, dim3(nb), dim3(256), 0, h->stream, h->d_start, h->ncell_mc, h->grid, h->d_hist, nactive, h->d_order);
    else hipLaunchKernelGGL(k_order_fill<2>, dim3(nb), dim3(256), 0, h->stream, h->d_start, h->ncell_mc, h->grid, h->d_hist, nactive, h->d_order);
    HIPCHK(h, hipGetLastError());
    return PMC_OK;
}

static int cb_cap(const pmc_system *h);

// Teams (NW >= 2 wavefronts per cell) stage the old way -- everything into LDS, then a separate compaction pass -- unless the
// runs of three cells that staging loads with one coalesced access per array are mostly longer than a wavefront: then the old
// path falls into its element-by-element loops, and the fused staging (27 single cells, box filter in registers, one count
// exchange per 9 segments) is faster.  Measured (r03, teams of 8): Kob-Andersen rho = 1.2 N = 10^5 (73 per run) 1.91 -> 2.21e8
// with swaps, 2.35 -> 2.61e8 without; LJ rho = 1 N = 32768 (57 per run) 1.33 -> 1.21e8: the switch sits between them.
static bool cb_fuse_teams(const pmc_system *h)
{
    if (const char *e = getenv("PMC_FUSE_TEAMS")) return e[0] == '1';      // experiments
    if (h->S < 2) return false;      // (compiled into the multi-species kernels only)
    if (!h->mc_ok || !h->use_filter || h->D != 3) return false;
    return 3.0 * (double)h->N / (double)h->ncell_mc > 62.0;
}

static CbArgs cb_args(pmc_system *h)
{
    CbArgs a;
    memset(&a, 0, sizeof a);
    for (int q = 0; q < 3; q++) {
        a.n[q] = q < h->D ? h->grid.n[q] : 1;
        a.half[q] = std::max(1, a.n[q] / 2);
        a.w[q] = q < h->D ? h->grid.w[q] : 1.0;
        a.shift[q] = h->grid.shift[q];
        a.L[q] = q < h->D ? h->p.box[q] : 1.0;
        a.invL[q] = 1.0 / a.L[q];
        a.invw[q] = 1.0 / a.w[q];
    }
    a.temperature = h->p.temperature;
    a.start = h->d_start; a.ids = h->d_ids;
    a.sx = h->d_sx; a.sy = h->d_sy; a.sz = h->d_sz; a.ssp = h->d_ssp;
    a.pos = h->d_pos; a.species = h->d_species;
    a.img = h->any_img ? h->d_img : nullptr;
    a.simg = h->any_img ? h->d_simg : nullptr;
    a.N = h->N; a.S = h->S;
    a.key0 = h->key[0]; a.key1 = h->key[1];
    a.m0 = to_dev(h->models[0]);
    a.models = h->d_models;
    a.stats = h->d_stats;
    a.cap = h->mc_ok ? cb_cap(h) : 128;
    a.fcap = 0;
    a.qcap = PMC_QCAP;
    a.rc2pad = h->maxcut * h->maxcut * (1.0 + 1e-9);
    // threshold of the FP32 pre-filter: rcut_max^2 (1 + 2^-12), rounded UP to float (pmc_kernels.h: error bound 2^-19)
    a.rc2f = std::nextafter((float)(h->maxcut * h->maxcut * (1.0 + 1.0 / 4096.0)), INFINITY);
    a.filter = h->use_filter ? 1 : 0;
    a.fuse_teams = cb_fuse_teams(h) ? 1 : 0;
    a.rounds = 1;
    return a;
}

// LDS staging capacity: the expected 3^d-cell population plus 4 %, rounded up to a multiple
// of 64 (candidates beyond it are read from HBM/L2 by the kernels' overflow loops)
static int cb_cap(const pmc_system *h)
{
    if (h->single_cell) return std::min(std::max((h->N + 63) / 64 * 64, 128), PMC_CAP_MAX);   // the one cell IS the neighbourhood: N candidates, exactly
    double mean = (double)h->N / (double)h->ncell_mc * (h->D == 3 ? 27.0 : 9.0);
    int cap = (int)((mean * 1.04 + 63.0) / 64.0) * 64;
    return std::min(std::max(cap, 128), PMC_CAP_MAX);
}

// Big neighbourhoods (round 5).  A box of 4-5 cutoffs per side at liquid density -- the published mixture's rcut = 4 sigma at
// N = 8000, examples/lj-mixture/run-validation.py:26-34 -- has 3^d-cell neighbourhoods of thousands of candidates: beyond the
// 1024 the staging holds by default and the 512 whose float copy a wavefront keeps in registers, so every slot took the
// direct FP64 loop over candidates read from L2.  Such systems have few cells per colour, i.e. they run teams with a CU to
// themselves: the team then stages the WHOLE neighbourhood (up to PMC_CAP_BIG candidates), keeps one float copy of it in LDS
// and every wavefront's pre-filter streams that copy (pmc_kernels.h: prepare_filter, wave_delta_e_filtered<STREAM>).
// Returns false -- the default capacities stay -- when the neighbourhood is small, the launch has no teams, or the LDS
// (160 KB per workgroup) cannot hold it.
struct CbBig { int cap, fcap, qcap; };
static bool cb_big(const pmc_system *h, int team, CbBig *out)
{
    if (!h->mc_ok || !h->use_filter || team != 8 || h->any_img || h->no_big || getenv("PMC_NO_BIG")) return false;   // (the instances that exist: k_cb_pass<..., BIG>)
    const double per_cell = (double)h->N / (double)h->ncell_mc;
    const double mean = h->single_cell ? (double)h->N : per_cell * (h->D == 3 ? 27.0 : 9.0);
    // (a single-cell box: exactly N candidates, and always these instances: theirs is the pre-filter that takes the minimum image of a
    // difference of two float coordinates, which the whole periodic box as ONE neighbourhood needs)
    if (!h->single_cell && mean * 1.04 <= (double)PMC_CAP_MAX) return false;
    // the whole 3^d-cell population with room for fluctuations (8 % + 64); the queue: the candidates within the largest cutoff of
    // either position -- a sphere's worth at the mean density, + 50 % + 128
    int cap = h->single_cell ? cb_cap(h) : (int)((mean * 1.08 + 64.0 + 63.0) / 64.0) * 64;
    if (cap > PMC_CAP_BIG) return false;
    double vol = 1.0;
    for (int q = 0; q < h->D; q++) vol *= h->p.box[q];
    const double r = h->maxcut, pi = 3.14159265358979323846;
    const double sphere = (h->D == 3) ? 4.0 / 3.0 * pi * r * r * r : pi * r * r;
    int qcap = (int)(((double)h->N / vol * sphere * 1.5 + 128.0 + 63.0) / 64.0) * 64;
    qcap = std::min(std::max(qcap, PMC_QCAP), 2048);
    size_t b = (pmc_lds_wave_bytes(cap, h->S > 1, team, cap, qcap) + 15) & ~(size_t)15;
    if (h->S > 1) b += (size_t)h->S * h->S * sizeof(DevModel);
    if (b > (size_t)160 * 1024 - 512) return false;
    if (out) { out->cap = cap; out->fcap = cap; out->qcap = qcap; }
    return true;
}

// ... of k_cb_pass, which applies the box filter while staging (stage_compacted) and only keeps the survivors in LDS --
// about 73 % of the 3^d-cell population at w ~ rcut.  How many that is depends on the state (a liquid: 343 +- 9 at
// N = 10^6; the jittered lattice it started from: 343 .. 512), so the kernel counts the visits by survivor count
// (stats[PMC_STAT_KEPT..]) and the next launches are sized for all but 0.2 % of them.  Those stage the old way, their
// overflow read from L2: results never depend on the capacity.
static int cb_cap_sweep(const pmc_system *h)
{
    if (const char *e = getenv("PMC_CAP")) return std::min(std::max(atoi(e) / 64 * 64, 128), PMC_CAP_MAX);   // experiments
    if (h->team > 1) { CbBig big; if (cb_big(h, h->team, &big)) return big.cap; }
    if (!h->use_filter || !PMC_FUSED_STAGE || h->team > 1) return cb_cap(h);   // (teams stage the old way)
    int cap = h->cap_hint;
    if (cap <= 0) {
        // nothing learnt yet: the expected population of the centre cell's box grown by the largest cutoff (Steiner's
        // formula) plus 6 % + 16
        const double r = h->maxcut * (1.0 + 1e-9), pi = 3.14159265358979323846;
        const double *w = h->grid.w;
        double vol = 1.0, v;
        for (int q = 0; q < h->D; q++) vol *= h->p.box[q];
        if (h->D == 3)
            v = w[0] * w[1] * w[2] + 2.0 * r * (w[0] * w[1] + w[1] * w[2] + w[2] * w[0]) + pi * r * r * (w[0] + w[1] + w[2]) + 4.0 / 3.0 * pi * r * r * r;
        else
            v = w[0] * w[1] + 2.0 * r * (w[0] + w[1]) + pi * r * r;
        cap = (int)(((double)h->N / vol * v * 1.06 + 16.0 + 63.0) / 64.0) * 64;
    }
    return std::min(std::max(cap, 128), cb_cap(h));
}

static bool cb_cap_sweep_adapts(const pmc_system *h)
{
    return h->use_filter && PMC_FUSED_STAGE && h->team == 1 && !getenv("PMC_CAP");
}

// `cum`: the call's histogram so far; what was already looked at (kept_seen) is subtracted
static void cb_cap_learn(pmc_system *h, const unsigned long long *cum)
{
    unsigned long long bk[16], total = 0, above = 0;
    for (int b = 0; b < 16; b++) {
        bk[b] = cum[b] - h->kept_seen[b];
        h->kept_seen[b] = cum[b];
        total += bk[b];
    }
    if (total == 0) return;
    int b = 15;
    while (b > 1 && above + bk[b] <= total / 500) above += bk[b--];
    h->cap_hint = b >= 15 ? PMC_CAP_MAX : 64 * b;
    if (getenv("PMC_DEBUG_CAP")) {
        fprintf(stderr, "[pmc cap] visits %llu:", total);
        for (int q = 0; q < 16; q++) if (bk[q]) fprintf(stderr, " <=%d:%llu", 64 * q, bk[q]);
        fprintf(stderr, " -> %d\n", h->cap_hint);
    }
}

// the float-copy and queue capacities that go with a big-neighbourhood staging capacity (cb_big)
static void cb_apply_big(const pmc_system *h, int cap_sweep, CbArgs &a)
{
    CbBig big;
    if (h->team > 1 && !getenv("PMC_CAP") && cb_big(h, h->team, &big) && cap_sweep == big.cap) { a.fcap = big.fcap; a.qcap = big.qcap; }
}

static size_t cb_lds_bytes(const pmc_system *h, int batch_nw = 0, int cap = 0)
{
    CbBig big;
    const bool isbig = batch_nw >= 2 && !getenv("PMC_CAP") && cb_big(h, batch_nw, &big) && cap == big.cap;
    size_t b = (pmc_lds_wave_bytes(cap > 0 ? cap : cb_cap(h), h->S > 1, batch_nw, isbig ? big.fcap : 0, isbig ? big.qcap : PMC_QCAP) + 15) & ~(size_t)15;
    if (h->S > 1) b += (size_t)h->S * h->S * sizeof(DevModel);
    if (const char *e = getenv("PMC_LDS_PAD")) b += (size_t)atoi(e);   // occupancy experiments
    return b;
}

#define PMC_ARGS_CHUNK 256     // passes whose launch arguments are uploaded together (blocking calls)
#define PMC_ARGS_ASYNC_MAX 8192   // ... by pmc_sweep_begin, which must not wait for the device: longer calls wait once per 8192 passes
// k_cb_pass additionally carries the team size NW (wavefronts per cell)
// The launch arguments (CbArgs) are read from device memory: `args` below is a device pointer to nrep_ of them.
#define DISPATCH_CB_ONE(DD, KK, NWV, MWV, h, grid, lds, args)   /* uses pool_ (MovePool), nrep_, mw_ (int) and img_ (bool) of the caller's scope */                                      \
    if (!done_ && img_ && (h)->D == DD && (h)->kind == KK && (h)->team == NWV && MWV == PMC_IMG_MW(NWV)) {   /* systems with periodic images: one register budget per team width */ \
        if ((h)->S > 1 && one_) hipLaunchKernelGGL((k_cb_pass<DD, KK, true, NWV, true, PMC_IMG_MW(NWV), true>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else if ((h)->S > 1) hipLaunchKernelGGL((k_cb_pass<DD, KK, true, NWV, false, PMC_IMG_MW(NWV), true>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else if (one_) hipLaunchKernelGGL((k_cb_pass<DD, KK, false, NWV, true, PMC_IMG_MW(NWV), true>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else hipLaunchKernelGGL((k_cb_pass<DD, KK, false, NWV, false, PMC_IMG_MW(NWV), true>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        done_ = true;                                                                                           \
    }                                                                                                           \
    if (!done_ && !img_ && (h)->D == DD && (h)->kind == KK && (h)->team == NWV && mw_ == MWV) {                   \
        if ((h)->S > 1 && one_) hipLaunchKernelGGL((k_cb_pass<DD, KK, true, NWV, true, MWV>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else if ((h)->S > 1) hipLaunchKernelGGL((k_cb_pass<DD, KK, true, NWV, false, MWV>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else if (one_) hipLaunchKernelGGL((k_cb_pass<DD, KK, false, NWV, true, MWV>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        else hipLaunchKernelGGL((k_cb_pass<DD, KK, false, NWV, false, MWV>), dim3(grid), dim3(64 * NWV), lds, (h)->stream, args, nrep_); \
        done_ = true;                                                                                           \
    }
// one wavefront per cell comes in two register budgets: 4 wavefronts per SIMD (128 VGPRs) pays when the LDS footprint lets
// more than 12 workgroups share a CU, 3 (168 VGPRs) is ~5 % faster when it does not (cb_minwaves)
// big neighbourhoods (cb_big; big_ of the caller's scope): the instance with the float copy in LDS; its dynamic LDS is beyond 64 KB,
// which needs the attribute on the instance that is launched
#define DISPATCH_CB_BIG(DD, KK, h, grid, lds, args)                                                             \
    if (!done_ && big_ && !img_ && (h)->D == DD && (h)->kind == KK && (h)->team == 8) {                          \
        const void *f_ = ((h)->S > 1 && one_) ? (const void *)k_cb_pass<DD, KK, true, 8, true, PMC_TEAM_MINWAVES, false, true>   \
                       : ((h)->S > 1)       ? (const void *)k_cb_pass<DD, KK, true, 8, false, PMC_TEAM_MINWAVES, false, true>  \
                       : one_               ? (const void *)k_cb_pass<DD, KK, false, 8, true, PMC_TEAM_MINWAVES, false, true>  \
                                            : (const void *)k_cb_pass<DD, KK, false, 8, false, PMC_TEAM_MINWAVES, false, true>; \
        if ((lds) > 64 * 1024) (void)hipFuncSetAttribute(f_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds)); \
        const CbArgs *a_ = (args);                                                                              \
        int n_ = nrep_;                                                                                         \
        void *ka_[2] = {(void *)&a_, (void *)&n_};                                                              \
        (void)hipLaunchKernel(f_, dim3(grid), dim3(64 * 8), ka_, lds, (h)->stream);                             \
        done_ = true;                                                                                           \
    }
#define DISPATCH_CB_KIND(DD, KK, h, grid, lds, args)                                                            \
    DISPATCH_CB_BIG(DD, KK, h, grid, lds, args)                                                                 \
    DISPATCH_CB_ONE(DD, KK, 1, 3, h, grid, lds, args)                                                           \
    DISPATCH_CB_ONE(DD, KK, 1, 4, h, grid, lds, args)                                                           \
    DISPATCH_CB_ONE(DD, KK, 2, PMC_TEAM_MINWAVES, h, grid, lds, args)                                           \
    DISPATCH_CB_ONE(DD, KK, 4, PMC_TEAM_MINWAVES, h, grid, lds, args)                                           \
    DISPATCH_CB_ONE(DD, KK, 8, PMC_TEAM_MINWAVES, h, grid, lds, args)
static int cb_minwaves(const pmc_system *h, size_t lds)
{
    if (h->team != 1) return PMC_TEAM_MINWAVES;
    if (const char *e = getenv("PMC_MINWAVES")) return atoi(e) == 3 ? 3 : 4;    // experiments
    // The 128-VGPR instances of multi-species / non-LJ systems keep 12 ... 72 B of scratch per lane, the 168-VGPR ones none: for
    // them a 13th or 14th workgroup per CU does not pay for the spills (Kob-Andersen N = 10^6, LDS good for 13 workgroups per CU:
    // 7.30 -> 7.74e8 moves/s with swaps, 8.28 -> 8.58e8 without, by taking the 12 of the larger budget); the LJ single-species
    // instance fits 127 VGPRs without scratch and takes the smaller budget from 13 workgroups on
    const size_t need = (h->S > 1 || h->kind != 0) ? 15 : 13;
    return lds * need <= 160 * 1024 ? 4 : 3;
}
#ifndef PMC_DEV_KIND
#define PMC_DEV_KIND 0
#endif
#ifdef PMC_DEV_BUILD   // development builds: 3-D instances of one model kind only (seconds instead of minutes to compile)
#define DISPATCH_CB(h, grid, lds, args)                                                                         \
    do {                                                                                                        \
        bool done_ = false;                                                                                     \
        const bool one_ = (pool_).npool == 1 && (pool_).kind[0] == 0;                                           \
        DISPATCH_CB_KIND(3, PMC_DEV_KIND, h, grid, lds, args)                                                   \
        (void)done_;                                                                                            \
    } while (0)
#else
#define DISPATCH_CB(h, grid, lds, args)                                                                         \
    do {                                                                                                        \
        bool done_ = false;                                                                                     \
        const bool one_ = (pool_).npool == 1 && (pool_).kind[0] == 0;                                           \
        DISPATCH_CB_KIND(3, 0, h, grid, lds, args) DISPATCH_CB_KIND(3, 1, h, grid, lds, args)                    \
        DISPATCH_CB_KIND(3, 2, h, grid, lds, args) DISPATCH_CB_KIND(3, 3, h, grid, lds, args)                    \
        DISPATCH_CB_KIND(3, 4, h, grid, lds, args) DISPATCH_CB_KIND(2, 0, h, grid, lds, args)                    \
        DISPATCH_CB_KIND(2, 1, h, grid, lds, args) DISPATCH_CB_KIND(2, 2, h, grid, lds, args)                    \
        DISPATCH_CB_KIND(2, 3, h, grid, lds, args) DISPATCH_CB_KIND(2, 4, h, grid, lds, args)                    \
        (void)done_;                                                                                            \
    } while (0)
#endif

#define DISPATCH_ONE(KERNEL, DD, KK, h, grid, lds, args)                                                     \
    if (!done_ && (h)->D == DD && (h)->kind == KK) {                                                            \
        if ((h)->S > 1) hipLaunchKernelGGL((KERNEL<DD, KK, true>), dim3(grid), dim3(64), lds, (h)->stream, args);   \
        else hipLaunchKernelGGL((KERNEL<DD, KK, false>), dim3(grid), dim3(64), lds, (h)->stream, args);         \
        done_ = true;                                                                                           \
    }

// template instance for (dimension, common model kind, multi-species)
#define DISPATCH_KERNEL(KERNEL, h, grid, lds, args)                                                             \
    do {                                                                                                        \
        bool done_ = false;                                                                                     \
        DISPATCH_ONE(KERNEL, 3, 0, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 3, 1, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 3, 2, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 3, 3, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 3, 4, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 2, 0, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 2, 1, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 2, 2, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 2, 3, h, grid, lds, args)                                                          \
        DISPATCH_ONE(KERNEL, 2, 4, h, grid, lds, args)                                                          \
        (void)done_;                                                                                            \
    } while (0)

static void ticket_fallback(pmc_system *h)
{
    h->ticket_lanes = 1;
    h->ticket_next = 0;
    if (h->d_ticket) (void)hipMemset(h->d_ticket, 0, PMC_TICKET_LANES * PMC_TICKET_STRIDE * sizeof(unsigned long long));
}

static int check_kernel_flags(pmc_system *h, const unsigned long long *st)
{
    if (st[17] & 1ull)
        return fail(h, PMC_ERR_UNSUPPORTED, "a cell holds more particles than the LDS staging capacity (that cell's slots were "
                                            "skipped; moves accepted in other cells are kept and the energy is recomputed on the next call)");
    bool starved = (st[17] & 2ull) != 0;
    if (h->inject_starve) {          // fault injection (pmc_debug_inject): take the fallback after a good launch
        h->inject_starve = false;
        if (h->ticket_lanes != 1) starved = true;
    }
    if (starved)
    {
        // the eight-counter ticket scheme assumes in-order round-robin dispatch to the XCDs for its progress argument (DESIGN.md
        // 4.3); whatever starved this launch, the handle takes the order-independent single counter from now on -- from a
        // counter that is in step with ticket_next again (the eight per-XCD counters are zeroed by every sweep's first kernel,
        // the single one is monotonic: after the switch it must not keep the aborted sweep's count).  The stream is idle
        // here: the caller has just read `st` back.
        ticket_fallback(h);
        return fail(h, PMC_ERR_STATE, "fused sweep: a cell's completion flag was never seen (positions were modified by a partial "
                                      "sweep: the chain is invalid; upload the configuration again -- the handle now draws its "
                                      "workgroup tickets from the single, dispatch-order-independent counter)");
    }
    return PMC_OK;
}

extern "C" int pmc_debug_inject(pmc_system *h, int32_t what)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_debug_inject: null handle");
    if (what != PMC_INJECT_STARVED_LAUNCH) return fail(h, PMC_ERR_BADARG, "pmc_debug_inject: unknown fault");
    h->inject_starve = true;
    return PMC_OK;
}

extern "C" int pmc_grid(pmc_system *h, int64_t *n_mc, double *w)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_grid: null handle");
    if (!h->mc_ok) return fail(h, PMC_ERR_UNSUPPORTED, "checkerboard grid needs >= 4 cells of width >= rcut per side");
    for (int q = 0; q < h->D; q++) {
        if (n_mc) n_mc[q] = h->grid.n[q];
        if (w) w[q] = h->grid.w[q];
    }
    return PMC_OK;
}

extern "C" int pmc_total_energy(pmc_system *h, double *E, int64_t *E_fix)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_total_energy: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->mc_ok) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_total_energy: box too small for the cell grid (need >= 4 cells per side)");
    int rc = set_device(h);
    if (rc) return rc;
    if (!h->mc_binned) {
        const double z[3] = {0, 0, 0};
        rc = mc_bin(h, z);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), h->stream));
    CbArgs a = cb_args(h);
    a.all_cells = 1;
    a.nactive = h->ncell_mc;
    size_t lds = cb_lds_bytes(h);
    DISPATCH_KERNEL(k_total_energy, h, a.nactive, lds, a);
    hipLaunchKernelGGL(k_stats_fold, dim3(PMC_NSTATS), dim3(PMC_NSHARD), 0, h->stream, h->d_stats);   // counter copies -> copy 0
    HIPCHK(h, hipGetLastError());
    unsigned long long st[PMC_NSTATS];
    HIPCHK(h, hipMemcpyAsync(st, h->d_stats, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    rc = check_kernel_flags(h, st);
    if (rc) return rc;
    long long tot = (long long)st[16];
    h->e_fix = tot / 2;
    h->e_fix_valid = true;
    h->energy = (double)h->e_fix / PMC_FIX;
    h->energy_stale = false;
    if (E) *E = h->energy;
    if (E_fix) *E_fix = h->e_fix;
    return PMC_OK;
}

extern "C" int pmc_energy_particles_exact(pmc_system *h, double *e)
{
    if (!h || !e) return fail(h, PMC_ERR_BADARG, "pmc_energy_particles_exact: null argument");
    if (!h->mc_ok) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_energy_particles_exact: box too small for the cell grid");
    if (h->pending) return fail(h, PMC_ERR_STATE, "pmc_energy_particles_exact: a sweep is in flight");
    int rc = set_device(h);
    if (rc) return rc;
    if (!h->mc_binned) {
        const double z[3] = {0, 0, 0};
        rc = mc_bin(h, z);
        if (rc) return rc;
    }
    DevTmp out;
    HIPCHK(h, out.alloc((size_t)h->N * sizeof(double)));
    HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), h->stream));
    CbArgs a = cb_args(h);
    a.all_cells = 1;
    a.nactive = h->ncell_mc;
    a.e_particle = out.as<double>();
    size_t lds = cb_lds_bytes(h);
    DISPATCH_KERNEL(k_total_energy, h, a.nactive, lds, a);
    hipLaunchKernelGGL(k_stats_fold, dim3(PMC_NSTATS), dim3(PMC_NSHARD), 0, h->stream, h->d_stats);   // counter copies -> copy 0
    HIPCHK(h, hipGetLastError());
    unsigned long long st[PMC_NSTATS];
    HIPCHK(h, hipMemcpyAsync(st, h->d_stats, sizeof st, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(e, out.p, (size_t)h->N * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return check_kernel_flags(h, st);
}

extern "C" int pmc_delta_e(pmc_system *h, int64_t n, const int64_t *i, const double *delta, double *e1, double *e2)
{
    if (!h || n < 0 || (n > 0 && (!i || !delta || !e1 || !e2))) return fail(h, PMC_ERR_BADARG, "pmc_delta_e: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (!h->mc_ok) return fail(h, PMC_ERR_UNSUPPORTED, "pmc_delta_e: box too small for the cell grid");
    if (n == 0) return PMC_OK;
    int rc = set_device(h);
    if (rc) return rc;
    if (!h->mc_binned) {
        const double z[3] = {0, 0, 0};
        rc = mc_bin(h, z);
        if (rc) return rc;
    }
    std::vector<int> w((size_t)n);
    for (int64_t k = 0; k < n; k++) {
        if (i[k] < 1 || i[k] > h->N) return fail(h, PMC_ERR_BADARG, "pmc_delta_e: particle id out of range");
        w[(size_t)k] = (int)i[k] - 1;
    }
    DevTmp tw, td, te;
    HIPCHK(h, tw.alloc((size_t)n * sizeof(int)));
    HIPCHK(h, td.alloc((size_t)n * h->D * sizeof(double)));
    HIPCHK(h, te.alloc((size_t)n * 2 * sizeof(double)));
    int *d_w = tw.as<int>();
    double *d_d = td.as<double>(), *d_e = te.as<double>();
    (void)hipMemcpyAsync(d_w, w.data(), (size_t)n * sizeof(int), hipMemcpyHostToDevice, h->stream);
    (void)hipMemcpyAsync(d_d, delta, (size_t)n * h->D * sizeof(double), hipMemcpyHostToDevice, h->stream);
    CbArgs a = cb_args(h);
    int grid = (int)std::min<int64_t>(n, 262144);
    if (h->D == 3) hipLaunchKernelGGL(k_delta_e<3>, dim3(grid), dim3(64), 0, h->stream, a, (int)n, d_w, d_d, d_e, d_e + n);
    else hipLaunchKernelGGL(k_delta_e<2>, dim3(grid), dim3(64), 0, h->stream, a, (int)n, d_w, d_d, d_e, d_e + n);
    hipError_t er = hipGetLastError();
    if (er == hipSuccess) er = hipMemcpyAsync(e1, d_e, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (er == hipSuccess) er = hipMemcpyAsync(e2, d_e + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    if (er != hipSuccess) return fail(h, PMC_ERR_HIP, hipGetErrorString(er));
    return PMC_OK;
}

#include "pmc_verlet.h"
#include "pmc_empty.h"
#include "pmc_seq_host.h"

extern "C" int pmc_set_sequential_engine(pmc_system *h, int32_t engine, int32_t window)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_set_sequential_engine: null handle");
    if (engine < 0 || engine > 5 || window < 0 || window > SQ_KMAX)
        return fail(h, PMC_ERR_BADARG, "pmc_set_sequential_engine: engine 0 (auto), 1 (one wavefront), 2 (speculative windows), 3 (state in LDS), 4 (VerletList state in LDS) or 5 (small EmptyList systems, evaluations repaired); window 0..4096");
    if (engine == 5 && window != 0 && window != 8 && window != 16)
        return fail(h, PMC_ERR_BADARG, "pmc_set_sequential_engine: engine 5 takes window 0 (the library's choice), 8 or 16 (wavefronts per chain)");
    if (engine == 5 && !seq_empty_fits(h))
        return fail(h, PMC_ERR_UNSUPPORTED, "pmc_set_sequential_engine: engine 5 needs an EmptyList system whose packed pair terms fit 160 KB of LDS (a few hundred particles)");
    if (engine == 4 && !seq_verlet_fits(h))
        return fail(h, PMC_ERR_UNSUPPORTED, "pmc_set_sequential_engine: engine 4 needs a VerletList system whose positions, cell vectors and neighbour lists fit 160 KB of LDS");
    if (engine == 3 && !seq_lds_fits(h, SQ_LDS_MAX))
        return fail(h, PMC_ERR_UNSUPPORTED, "pmc_set_sequential_engine: engine 3 needs a LinkedList, CellList or EmptyList system whose state fits 160 KB of LDS");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    h->seq_engine = engine;
    h->seq_window = window;
    return PMC_OK;
}

extern "C" int pmc_sequential_info(pmc_system *h, int64_t *out)
{
    if (!h || !out) return fail(h, PMC_ERR_BADARG, "pmc_sequential_info: null argument");
    out[0] = (int64_t)h->sq_last.rounds;
    out[1] = (int64_t)h->sq_last.evals;
    out[2] = (int64_t)seq_window(h);
    out[3] = (int64_t)h->sq_last_steps;
    return PMC_OK;
}

extern "C" int pmc_verlet_info(pmc_system *h, int64_t *out)
{
    if (!h || !out) return fail(h, PMC_ERR_BADARG, "pmc_verlet_info: null argument");
    for (int q = 0; q < 4; q++) out[q] = (int64_t)h->sqv_last[q];
    return PMC_OK;
}

// `nsteps` steps of the chain on the literal loop (engine 1, k_ref_ops); `credit_sweeps` whole sweeps are added to the sweep counter.
// add: the counts are added to *out (the steps of the call that another engine already ran), otherwise they replace it.
static int sweep_literal_steps(pmc_system *h, const pmc_move *pool, int npool, long long nsteps, int64_t credit_sweeps, pmc_stats *out, bool add)
{
    if (!h->cells_built) {
        int rc = pmc_build_cells(h);
        if (rc) return rc;
    }
    HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), h->stream));
    RefOp o;
    memset(&o, 0, sizeof o);
    o.op = OP_STEPS;
    o.step0 = h->step_ctr;
    o.nsteps = nsteps;
    o.key0 = h->key[0];
    o.key1 = h->key[1];
    o.pool = make_pool(pool, npool);
    o.stats = h->d_stats;
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    int rc = ref_op_launch(h, o);
    if (rc) return rc;
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    HIPCHK(h, hipEventSynchronize(h->ev[1]));
    h->step_ctr += (uint64_t)o.nsteps;
    h->sweep_ctr += (uint64_t)credit_sweeps;
    if (out) {
        unsigned long long st[PMC_NSTATS];
        HIPCHK(h, hipMemcpy(st, h->d_stats, sizeof st, hipMemcpyDeviceToHost));
        for (int m = 0; m < PMC_MAX_MOVES; m++) {
            out->attempted[m] = (add ? out->attempted[m] : 0) + (int64_t)st[m];
            out->accepted[m] = (add ? out->accepted[m] : 0) + (int64_t)st[8 + m];
        }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
        out->kernel_ms = (add ? out->kernel_ms : 0) + ms;
        out->sweep_kernel_ms = (add ? out->sweep_kernel_ms : 0) + ms;
        out->sweep_kernel_launches = (add ? out->sweep_kernel_launches : 0) + 1;
        out->energy = h->energy;
        out->energy_fix = 0;
    }
    return PMC_OK;
}

static int sweep_sequential(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    if (nsweeps > 0 && h->list_type == 0 && (h->seq_engine == 0 || h->seq_engine == 5) && seq_empty_pool_ok(pool, npool) && seq_empty_fits(h)) {
        // a small EmptyList system and Displacements only (the reference's published benchmark): state in LDS, the steps evaluated
        // ahead and repaired term by term (pmc_empty.h, engine 5)
        pmc_system *one[1] = {h};
        return sweep_sequential_empty(one, 1, pool, npool, nsweeps, out);
    }
    if (h->seq_engine == 5) return fail(h, PMC_ERR_UNSUPPORTED, "sequential engine 5: an EmptyList system of a few hundred particles and a pool of Displacements only");
    if (nsweeps > 0 && seq_pool_ok(pool, npool) && h->seq_engine != 1 && h->seq_engine != 2 && h->seq_engine != 4 &&
        (h->seq_engine == 3 || (h->ncell_ref <= 343 && seq_lds_fits(h, SQ_LDS_AUTO)))) {
        // few cells per side: every step interacts with every other, and the whole state fits the LDS (pmc_seq.h, engine 3)
        pmc_system *one[1] = {h};
        const int rl = sweep_sequential_lds(one, 1, pool, npool, nsweeps, out);
        // (a crowded cell can make the rows outgrow the LDS after all: nothing has run then, and the windows take over)
        if (rl != PMC_ERR_UNSUPPORTED || h->seq_engine == 3) return rl;
    }
    if (nsweeps > 0 && h->seq_engine != 3 && h->seq_engine != 4 && seq_fast_applicable(h, pool, npool)) {
        const int rf = sweep_sequential_fast(h, pool, npool, nsweeps, out);
        // (neighbourhood rows beyond the LDS -- a cell of thousands of particles -- are refused before anything runs: the
        // literal loop below has no such limit)
        if (rf != PMC_ERR_UNSUPPORTED || h->seq_engine == 2) return rf;
    }
    if (nsweeps > 0 && h->list_type == 3 && (h->seq_engine == 0 || h->seq_engine == 4) && seq_pool_ok(pool, npool)) {
        // VerletList (the reference's CI system): state in LDS, steps evaluated a batch at a time (pmc_verlet.h, engine 4)
        if (!h->cells_built) {
            int rc = pmc_build_cells(h);
            if (rc) return rc;
        }
        if (seq_verlet_fits(h)) {
            pmc_system *one[1] = {h};
            long long did = 0;
            const int rv = sweep_sequential_verlet(one, 1, pool, npool, nsweeps, out, &did);
            if (rv) return rv;
            const long long total = nsweeps * (int64_t)h->N;
            if (did == total) return PMC_OK;
            // a neighbour list or cell vector outgrew its LDS row: the rest of the call on the literal loop (rows in global memory)
            return sweep_literal_steps(h, pool, npool, total - did, nsweeps, out, true);
        }
        if (h->seq_engine == 4) return fail(h, PMC_ERR_UNSUPPORTED, "sequential engine 4: the system's state does not fit the LDS");
    }
    return sweep_literal_steps(h, pool, npool, nsweeps * (int64_t)h->N, nsweeps, out, false);   // sweepstep = N, src/ParticlesMC.jl:246
}

// The chains of a `parallel = true` run on the literal loop (engine 1: VerletList, EmptyList systems beyond the LDS, swap pools
// between a species and itself, or chains forced onto it), ALL IN ONE LAUNCH: workgroup b of k_ref_steps_many runs chain b from
// its own entry of two device tables (any shape, any list type; one dimension per launch).  Every chain ends exactly as
// pmc_sweep(PMC_SEQUENTIAL) on it alone (the same ref_steps).  Until round 5 these chains ran one after the other.
static int sweep_sequential_ref_many(pmc_system **hs, int n, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    pmc_system *m = hs[0];
    int rc = set_device(m);
    if (rc) return rc;
    std::vector<RefArgs> args((size_t)n);
    std::vector<RefOp> ops((size_t)n);
    std::vector<unsigned long long> st((size_t)n * PMC_NSTATS);
    DevTmp d_args, d_ops;
    StreamDrain drain(m->stream);       // (the copies below read args / ops / h->energy and write st / h->energy: none outlives the stream's work)
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        if (!h->cells_built) { rc = pmc_build_cells(h); if (rc) return rc; }
        if (int rs = refresh_stale_energy(h)) return rs;
        HIPCHK(h, hipStreamSynchronize(h->stream));                    // earlier work on the chain's own stream
        HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), m->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_energy, &h->energy, sizeof(double), hipMemcpyHostToDevice, m->stream));
        args[(size_t)r] = ref_args(h);
        RefOp &o = ops[(size_t)r];
        memset(&o, 0, sizeof o);
        o.op = OP_STEPS;
        o.step0 = h->step_ctr;
        o.nsteps = nsweeps * (int64_t)h->N;                             // sweepstep = N, src/ParticlesMC.jl:246
        o.key0 = h->key[0];
        o.key1 = h->key[1];
        o.pool = make_pool(pool, npool);
        o.stats = h->d_stats;
    }
    HIPCHK(m, d_args.alloc((size_t)n * sizeof(RefArgs)));
    HIPCHK(m, d_ops.alloc((size_t)n * sizeof(RefOp)));
    HIPCHK(m, hipMemcpyAsync(d_args.p, args.data(), (size_t)n * sizeof(RefArgs), hipMemcpyHostToDevice, m->stream));
    HIPCHK(m, hipMemcpyAsync(d_ops.p, ops.data(), (size_t)n * sizeof(RefOp), hipMemcpyHostToDevice, m->stream));
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    if (m->D == 3) hipLaunchKernelGGL(k_ref_steps_many<3>, dim3(n), dim3(64), 0, m->stream, d_args.as<RefArgs>(), d_ops.as<RefOp>());
    else hipLaunchKernelGGL(k_ref_steps_many<2>, dim3(n), dim3(64), 0, m->stream, d_args.as<RefArgs>(), d_ops.as<RefOp>());
    HIPCHK(m, hipGetLastError());
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    for (int r = 0; r < n; r++) {
        HIPCHK(hs[r], hipMemcpyAsync(&hs[r]->energy, hs[r]->d_energy, sizeof(double), hipMemcpyDeviceToHost, m->stream));
        HIPCHK(hs[r], hipMemcpyAsync(st.data() + (size_t)r * PMC_NSTATS, hs[r]->d_stats, PMC_NSTATS * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    int first = PMC_OK;
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        h->mc_binned = false;
        h->e_fix_valid = false;
        h->img_valid = false;
        h->step_ctr += (uint64_t)ops[(size_t)r].nsteps;
        h->sweep_ctr += (uint64_t)nsweeps;
        if (h->list_type >= 2 || h->d_cl_items) { const int re = ref_list_errors(h); if (re && !first) first = re; }
        if (out) {
            pmc_stats &o = out[r];
            const unsigned long long *s_ = st.data() + (size_t)r * PMC_NSTATS;
            for (int q = 0; q < PMC_MAX_MOVES; q++) { o.attempted[q] = (int64_t)s_[q]; o.accepted[q] = (int64_t)s_[8 + q]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = ms;
            o.sweep_kernel_launches = 1;
            o.energy = h->energy;
            o.energy_fix = 0;
        }
    }
    return first;
}

// after a failed or partial throughput sweep: positions may have changed, so the running energy, the
// cell-sorted copy, the reference lists and the SpeciesList are all recomputed by their next consumer
static void cb_invalidate(pmc_system *h)
{
    (void)hipMemsetAsync(h->d_stats, 0, (size_t)PMC_NSHARD * PMC_NSTATS * sizeof(unsigned long long), h->stream);   // every counter copy
    h->e_fix_valid = false;
    h->img_valid = false;
    h->energy_stale = true;
    h->state_invalid = true;
    h->mc_binned = false;
    h->cells_built = false;
    h->count_is_zero = false;
    if (h->S > 1) h->species_list_stale = true;
}

static int cb_enqueue_sweeps(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, bool want_timing, bool async);

// async (pmc_sweep_begin): the call must return without waiting for the device, so that the chains of several handles
// overlap: the launch arguments of ALL its passes (up to PMC_ARGS_ASYNC_MAX) go up at once and the LDS capacity is not
// re-learnt inside the call (pmc_sweep_end learns from the whole call).  Blocking calls (pmc_sweep) work in chunks of
// PMC_ARGS_CHUNK passes and re-fit the capacity between chunks.
static int cb_enqueue(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, bool want_timing, bool async = false)
{
    h->cb_started = false;
    const int rc = cb_enqueue_sweeps(h, pool, npool, nsweeps, want_timing, async);
    if (rc != PMC_OK && !h->pending && h->cb_started) {
        // an error part-way through the loop leaves launches queued: drain them, then distrust the state
        (void)hipStreamSynchronize(h->stream);
        cb_invalidate(h);
    }
    return rc;
}

static int cb_enqueue_sweeps(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, bool want_timing, bool async)
{
    pmc_stats *out = want_timing ? &h->done_stats : nullptr;
    if (!h->mc_ok)
        return fail(h, PMC_ERR_UNSUPPORTED, "checkerboard mode needs >= 4 cells of width >= rcut per side; use PMC_SEQUENTIAL");
    int rc;
    if (!h->e_fix_valid) {
        rc = pmc_total_energy(h, nullptr, nullptr);
        if (rc) return rc;
    }
#ifdef PMC_DEBUG_SLOTLOG
    if (getenv("PMC_SLOTLOG")) {
        double *d = nullptr;
        HIPCHK(h, hipMemcpyFromSymbol(&d, HIP_SYMBOL(g_slotlog), sizeof(d)));
        if (!d) {
            HIPCHK(h, hipMalloc(&d, (size_t)8 * h->slot_rounds * h->N * sizeof(double)));
            HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_slotlog), &d, sizeof(d)));
        }
        HIPCHK(h, hipMemset(d, 0, (size_t)8 * h->slot_rounds * h->N * sizeof(double)));
    }
#endif
    for (int m = 0; m < npool; m++) {
        if (pool[m].kind == PMC_MOVE_SWAP && h->S < 2)
            return fail(h, PMC_ERR_BADARG, "DiscreteSwap needs at least two species");
    }
    // uniform hand-over rule: after throughput sweeps the next consumer of the SpeciesList sees SpeciesList(species)
    // of the current state (swaps here are cell-local and do not maintain the global list)
    if (h->S > 1 && nsweeps > 0) h->species_list_stale = true;
    HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), h->stream));
    const MovePool mp = make_pool(pool, npool);
    const int ncol = h->single_cell ? 1 : (1 << h->D);     // (a single-cell box has one colour)
    int nactive = 1;
    for (int q = 0; q < h->D; q++) nactive *= std::max(1, h->grid.n[q] / 2);
    int cap_sweep = cb_cap_sweep(h);
    size_t lds = cb_lds_bytes(h, h->team >= 2 ? h->team : 0, cap_sweep);   // teams keep slot records and one survivor queue per wavefront in LDS
    int64_t launches = 0, timed_sweeps = 0;
    // fused launches stream the cells of every colour in the same spatial order, so that the cells a
    // workgroup waits for were ticketed long before it (a pipeline with one tail per sweep); separate
    // colour launches sort each colour longest-first to shorten their eight tails
    const bool fused = h->use_fused;
    const bool lpt = h->use_order && (!fused || h->fused_lpt == 1) && !h->single_cell;
    const bool fuse_slots = h->N < 500000;   // small systems: slots generated inside pass C of the binning (one launch fewer)
    const int per_sweep = fused ? 1 : ncol;  // launches, and CbArgs entries, per sweep
    const MovePool &pool_ = mp;
    const int nrep_ = 1;
    const bool img_ = h->any_img;          // some particle carries a periodic image: the kernel instances with the image path
    int mw_ = cb_minwaves(h, lds);
    const bool big_ = [&] { CbBig b; return !img_ && !getenv("PMC_CAP") && cb_big(h, h->team, &b) && cap_sweep == b.cap; }();   // (teams never re-learn their capacity)
    // long calls: the staging capacity is fitted again (cb_cap_learn) to what the last chunk of sweeps saw -- a first call on
    // a handle after 16 sweeps already
    const bool learn = cb_cap_sweep_adapts(h) && !async;
    const bool learn_early = learn && h->cap_hint <= 0 && (nsweeps + h->rounds - 1) / h->rounds > 48;
    memset(h->kept_seen, 0, sizeof h->kept_seen);
    // The launch arguments of the sweep kernel live in device memory (k_cb_pass): all the sweeps of a chunk are prepared
    // here and uploaded at once.
    const int64_t npass_all = (nsweeps + h->rounds - 1) / h->rounds;
    const int64_t chunk = async ? std::max<int64_t>(PMC_ARGS_CHUNK, std::min<int64_t>(npass_all, PMC_ARGS_ASYNC_MAX)) : PMC_ARGS_CHUNK;
    if (h->cbtab_cap < (size_t)chunk * per_sweep) {      // (use_fused is fixed per handle)
        if (h->d_cbtab) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->d_cbtab)); h->d_cbtab = nullptr; h->cbtab_cap = 0; }
        HIPCHK(h, hipMalloc(&h->d_cbtab, (size_t)chunk * per_sweep * sizeof(CbArgs)));
        h->cbtab_cap = (size_t)chunk * per_sweep;
    }
    std::vector<CbArgs> tabh;
    std::vector<std::array<double, 3>> shifts;
    const size_t Ns = (size_t)h->N * (size_t)h->slot_rounds;
    const SlotOut slots{h->d_slot_f, h->d_slot_f + Ns, h->d_slot_f + 2 * Ns, h->d_slot_f + 3 * Ns, h->d_slot_w, h->d_slot_w + Ns, h->d_slot_m};
    // the passes of this call (pmc_set_rounds): pass j covers the sweeps [pass_off[j], pass_off[j] + pass_k[j])
    const int64_t npass = (nsweeps + h->rounds - 1) / h->rounds;
    auto pass_off = [&](int64_t j) { return j * (int64_t)h->rounds; };
    auto pass_k = [&](int64_t j) { return (int)std::min<int64_t>(h->rounds, nsweeps - j * (int64_t)h->rounds); };
    HIPCHK(h, hipEventRecord(h->ev[0], h->stream));
    h->cb_started = nsweeps > 0;
    for (int64_t s0 = 0, ns = 0; s0 < npass; s0 += ns) {
        ns = std::min<int64_t>(learn_early && s0 == 0 ? 16 : chunk, npass - s0);
        if (learn && s0 > 0) {
            unsigned long long buckets[16];
            hipLaunchKernelGGL(k_stats_fold, dim3(PMC_NSTATS), dim3(PMC_NSHARD), 0, h->stream, h->d_stats);
            HIPCHK(h, hipGetLastError());
            HIPCHK(h, hipMemcpyAsync(buckets, h->d_stats + PMC_STAT_KEPT, sizeof buckets, hipMemcpyDeviceToHost, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            cb_cap_learn(h, buckets);
            cap_sweep = cb_cap_sweep(h);
            lds = cb_lds_bytes(h, h->team >= 2 ? h->team : 0, cap_sweep);
            mw_ = cb_minwaves(h, lds);
        }
        tabh.assign((size_t)(ns * per_sweep), CbArgs{});
        shifts.assign((size_t)ns, std::array<double, 3>{0.0, 0.0, 0.0});
        bool reset_flags = false;
        for (int64_t k = 0; k < ns; k++) {
            const uint64_t g = h->sweep_ctr + (uint64_t)pass_off(s0 + k);
            const uint32_t s_lo = (uint32_t)g, s_hi = (uint32_t)(g >> 32) & 0xffffffu;
            uint32_t W[4];
            philox4x32(0, 0, s_lo, s_hi | ((uint32_t)TAG_SHIFT << 24), h->key[0], h->key[1], W);
            for (int q = 0; q < h->D; q++) shifts[(size_t)k][(size_t)q] = ((double)W[q] * (1.0 / 4294967296.0)) * h->grid.w[q];
            for (int q = 0; q < 3; q++) h->grid.shift[q] = shifts[(size_t)k][(size_t)q];
            CbArgs a = cb_args(h);
            a.cap = cap_sweep;
            cb_apply_big(h, cap_sweep, a);
            a.order = lpt ? h->d_order : nullptr;
            a.slot_tlnu = slots.tlnu; a.slot_dx = slots.dx; a.slot_dy = slots.dy; a.slot_dz = slots.dz;
            a.slot_w1 = slots.w1; a.slot_w2 = slots.w2; a.slot_m = slots.m;
            a.s_lo = s_lo; a.s_hi = s_hi;
            a.rounds = pass_k(s0 + k);
            a.pool = mp;
            a.all_cells = 0;
            a.nactive = nactive;
            if (fused) {
                // one launch for all colours; cells wait on the completion flags of their lower-colour neighbours
                if (h->epoch == 0xffffffffu) {       // (once in 4e9 sweeps) the flags start over
                    reset_flags = true;
                    h->epoch = 0;
                }
                a.fused = 1;
                a.epoch = ++h->epoch;
                a.done = h->d_done;
                a.ticket = h->d_ticket;
                a.ticket_lanes = h->ticket_lanes;
                a.ticket_base = h->ticket_next;
                a.colour = 0;
                if (h->ticket_lanes == 1) h->ticket_next += (unsigned long long)(ncol * nactive);
                tabh[(size_t)k] = a;
            } else {
                for (int col = 0; col < ncol; col++) {
                    a.colour = col;
                    tabh[(size_t)(k * ncol + col)] = a;
                }
            }
        }
        if (s0 > 0 || reset_flags) HIPCHK(h, hipStreamSynchronize(h->stream));   // the previous chunk's entries are still being read
        if (reset_flags) HIPCHK(h, hipMemsetAsync(h->d_done, 0, (size_t)h->ncell_mc * sizeof(unsigned int), h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_cbtab, tabh.data(), tabh.size() * sizeof(CbArgs), hipMemcpyHostToDevice, h->stream));
        for (int64_t k = 0; k < ns; k++) {
            const int64_t sw = s0 + k;
            const CbArgs &a0 = tabh[(size_t)(k * per_sweep)];
            SlotGen sg;
            memset(&sg, 0, sizeof sg);
            sg.enabled = 1;
            sg.rounds = a0.rounds;
            sg.s_lo = a0.s_lo; sg.s_hi = a0.s_hi; sg.key0 = h->key[0]; sg.key1 = h->key[1];
            sg.temperature = h->p.temperature;
            sg.pool = mp;
            sg.out = slots;
            rc = mc_bin(h, shifts[(size_t)k].data(), fuse_slots ? &sg : nullptr);
            if (rc) return rc;
            if (!fuse_slots) {      // large systems: own launch (coalesced stores)
                const int nbs = (int)(((long long)h->N * sg.rounds + 255) / 256);
                if (h->D == 3) hipLaunchKernelGGL(k_mc_slots<3>, dim3(nbs), dim3(256), 0, h->stream, h->N, h->d_ids, h->d_cell_of, h->d_start, sg);
                else hipLaunchKernelGGL(k_mc_slots<2>, dim3(nbs), dim3(256), 0, h->stream, h->N, h->d_ids, h->d_cell_of, h->d_start, sg);
            }
            if (lpt) {
                rc = mc_order(h, nactive);
                if (rc) return rc;
            }
            // HIP events bracket the colour passes of every sweep on this stream; they are
            // read back after the final synchronize so the host never stalls the queue
            const bool timed = out && sw < 4096;      // (sw: index of the pass)
            if (timed) {
                while ((int64_t)h->pass_ev.size() < 2 * (sw + 1)) {
                    hipEvent_t e;
                    HIPCHK(h, hipEventCreate(&e));
                    h->pass_ev.push_back(e);
                }
                HIPCHK(h, hipEventRecord(h->pass_ev[2 * sw], h->stream));
            }
            for (int l = 0; l < per_sweep; l++) {
                const CbArgs *args = h->d_cbtab + (k * per_sweep + l);
                DISPATCH_CB(h, fused ? ncol * nactive : nactive, lds, args);
                launches++;
            }
            if (timed) {
                HIPCHK(h, hipEventRecord(h->pass_ev[2 * sw + 1], h->stream));
                timed_sweeps = sw + 1;
            }
            HIPCHK(h, hipGetLastError());
        }
    }
    HIPCHK(h, hipEventRecord(h->ev[1], h->stream));
    hipLaunchKernelGGL(k_stats_fold, dim3(PMC_NSTATS), dim3(PMC_NSHARD), 0, h->stream, h->d_stats);   // counter copies -> copy 0
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(h->h_stats, h->d_stats, PMC_NSTATS * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    h->pending_launches_per_sweep = npass > 0 ? (int)(launches / npass) : ncol;    // (per pass)
    h->pending = true;
    h->pending_mode = PMC_CHECKERBOARD;
    h->pending_nsweeps = nsweeps;
    h->pending_timed = timed_sweeps;
    return PMC_OK;
}

static int cb_finish(pmc_system *h, pmc_stats *out)
{
    HIPCHK(h, hipStreamSynchronize(h->stream));
#ifdef PMC_DEBUG_SLOTLOG
    if (const char *path = getenv("PMC_SLOTLOG")) {
        double *d = nullptr;
        HIPCHK(h, hipMemcpyFromSymbol(&d, HIP_SYMBOL(g_slotlog), sizeof(d)));
        if (d) {
            std::vector<double> buf((size_t)8 * h->slot_rounds * h->N);
            HIPCHK(h, hipMemcpy(buf.data(), d, buf.size() * sizeof(double), hipMemcpyDeviceToHost));
            if (FILE *f = fopen(path, "wb")) { fwrite(buf.data(), sizeof(double), buf.size(), f); fclose(f); }
        }
    }
#endif
    h->pending = false;
    const unsigned long long *st = h->h_stats;
    const int64_t nsweeps = h->pending_nsweeps, timed_sweeps = h->pending_timed;
    float pass_ms = 0.f;
    h->sweep_ctr += (uint64_t)nsweeps;
    h->cells_built = false;
    int rc = check_kernel_flags(h, st);
    if (rc) {
        // the other cells of these sweeps have committed their accepted moves: nothing derived from the
        // state before the call may be trusted any more (a caller that falls back to PMC_SEQUENTIAL gets a
        // freshly recomputed energy, a fresh binning and a fresh reference list)
        cb_invalidate(h);
        return rc;
    }
#ifdef PMC_DEBUG_TIME
    {
        // shader-clock cycles per cell visit, by phase (wavefront 0 of every workgroup; s_memtime)
        const double v = (double)st[47];
        static const char *names[] = {"ticket", "stage+poll", "compact+bounds+fcopy", "slot loop", "writeback+publish",
                                      "  slot prologue", "  filter", "  dense", "  reduce+accept", "  (slots evaluated)",
                                      "  stage: offsets", "  stage: flags of lower colours", "  stage: coordinates arrive", "  stage: box test + LDS"};
        fprintf(stderr, "[pmc time] visits=%.0f\n", v);
        for (int k = 0; k < 14; k++) fprintf(stderr, "[pmc time] %-24s %12.1f cycles/visit\n", names[k], (double)st[32 + k] / v);
    }
#endif
#ifdef PMC_DEBUG_COUNT
    fprintf(stderr, "[pmc debug] visits=%llu mean_ncand=%.1f nowrap=%.3f bounds=%.3f mean_nseg=%.2f compacted=%.4f filt=%.4f sd_ncand=%.1f\n", st[21], (double)st[20] / st[21], (double)st[22] / st[21], (double)st[23] / st[21], 0.0 /* (st[24] now counts the FP64 distance evaluations) */, (double)st[28] / st[21], (double)st[29] / st[21], sqrt((double)st[30] / st[21] - ((double)st[20] / st[21]) * ((double)st[20] / st[21])));
#endif
    h->e_fix += (long long)st[16];
    h->energy = (double)h->e_fix / PMC_FIX;
    cb_cap_learn(h, st + PMC_STAT_KEPT);
    if (out) {
        for (int m = 0; m < PMC_MAX_MOVES; m++) { out->attempted[m] = (int64_t)st[m]; out->accepted[m] = (int64_t)st[8 + m]; }
        float ms = 0;
        (void)hipEventElapsedTime(&ms, h->ev[0], h->ev[1]);
        out->kernel_ms = ms;
        for (int64_t k = 0; k < timed_sweeps; k++) {
            float t = 0;
            (void)hipEventElapsedTime(&t, h->pass_ev[2 * k], h->pass_ev[2 * k + 1]);
            pass_ms += t;
        }
        out->sweep_kernel_ms = pass_ms;
        out->sweep_kernel_launches = timed_sweeps * h->pending_launches_per_sweep;
        out->energy = h->energy;
        out->energy_fix = h->e_fix;
        out->pair_tests = (int64_t)st[25];
        out->pairs_in_range = (int64_t)st[26];
        out->null_moves = (int64_t)st[27];
        out->dense_evals = (int64_t)st[24];
        out->staged_candidates = (int64_t)st[31];
        out->written_back = (int64_t)st[19];
    }
    return PMC_OK;
}

static int sweep_checkerboard(pmc_system *h, const pmc_move *pool, int npool, int64_t nsweeps, pmc_stats *out)
{
    int rc = cb_enqueue(h, pool, npool, nsweeps, out != nullptr);
    if (rc) return rc;
    return cb_finish(h, out);
}

static int validate_pool(pmc_system *h, const pmc_move *pool, int32_t npool, int64_t nsweeps);

extern "C" int pmc_sweep_begin(pmc_system *h, int32_t mode, const pmc_move *pool, int32_t npool, int64_t nsweeps)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_sweep_begin: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "pmc_sweep_begin: a sweep is already in flight on this handle");
    int rc = validate_pool(h, pool, npool, nsweeps);
    if (rc) return rc;
    rc = set_device(h);
    if (rc) return rc;
    memset(&h->done_stats, 0, sizeof h->done_stats);
    if (mode == PMC_CHECKERBOARD) return cb_enqueue(h, pool, npool, nsweeps, true, true);
    if (mode == PMC_SEQUENTIAL) {
        rc = sweep_sequential(h, pool, npool, nsweeps, &h->done_stats);   // one wavefront: run it now
        if (rc) return rc;
        h->pending = true;
        h->pending_mode = PMC_SEQUENTIAL;
        return PMC_OK;
    }
    return fail(h, PMC_ERR_BADARG, "pmc_sweep_begin: unknown mode");
}

extern "C" int pmc_sweep_end(pmc_system *h, pmc_stats *out)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_sweep_end: null handle");
    if (!h->pending) return fail(h, PMC_ERR_STATE, "pmc_sweep_end: no sweep in flight");
    int rc = set_device(h);
    if (rc) return rc;
    if (h->pending_mode == PMC_CHECKERBOARD) {
        rc = cb_finish(h, &h->done_stats);
        if (rc) return rc;
    }
    h->pending = false;
    if (out) *out = h->done_stats;
    return PMC_OK;
}

static int validate_pool(pmc_system *h, const pmc_move *pool, int32_t npool, int64_t nsweeps)
{
    if (!h || !pool || npool < 1 || npool > PMC_MAX_MOVES || nsweeps < 0)
        return fail(h, PMC_ERR_BADARG, "pmc_sweep: bad argument");
    for (int m = 0; m < npool; m++) {
        if (pool[m].kind == PMC_MOVE_DISPLACEMENT) {
            if (!(pool[m].sigma > 0.0)) return fail(h, PMC_ERR_BADARG, "pmc_sweep: Displacement needs sigma > 0");
        } else if (pool[m].kind == PMC_MOVE_SWAP) {
            if (pool[m].species[0] < 1 || pool[m].species[0] > h->S || pool[m].species[1] < 1 || pool[m].species[1] > h->S)
                return fail(h, PMC_ERR_BADARG, "pmc_sweep: DiscreteSwap species out of range");
        } else {
            return fail(h, PMC_ERR_BADARG, "pmc_sweep: unsupported action");
        }
        if (!(pool[m].probability >= 0.0)) return fail(h, PMC_ERR_BADARG, "pmc_sweep: negative probability");
    }
    return PMC_OK;
}

// ---- many replicas of one shape per launch (chains with parallel = true, tempering ladders) ---------------------------
// src/ParticlesMC.jl:164,246: the chains of a run are independent Systems of the same shape.  One stream per handle
// (pmc_sweep_begin / _end) overlaps them, but a small system is a chain of ~6 short dependent launches per sweep and the
// launches of eight streams do not interleave well (8 x N = 4096: 5.6e7 moves/s against 1.8e7 for one chain).  Here the
// n handles share every launch: the binning kernels take the replica from blockIdx.y, the fused sweep kernel from the ticket.
static bool many_compatible(pmc_system **hs, int n)
{
    const pmc_system *m = hs[0];
    // (k_many_scan scans a replica's cell counts with ONE block, 1024 cells per step: up to 32768 cells that is <= 32 steps per
    // sweep, a few per cent of a sweep of such a system)
    if (!m->mc_ok || !m->use_fused || m->N >= 500000 || m->ncell_mc > 4 * PMC_SCAN_SMALL_MAX) return false;
    for (int r = 1; r < n; r++) {
        const pmc_system *h = hs[r];
        if (h == m) return false;
        for (int q = 0; q < r; q++) if (hs[q] == h) return false;
        if (h->p.device != m->p.device || h->D != m->D || h->N != m->N || h->S != m->S || h->kind != m->kind || h->team != m->team ||
            h->use_filter != m->use_filter || !h->mc_ok || !h->use_fused || h->ncell_mc != m->ncell_mc || h->rounds != m->rounds)
            return false;
        for (int a = 0; a < 3; a++)
            if (h->grid.n[a] != m->grid.n[a] || h->grid.w[a] != m->grid.w[a] || h->p.box[a] != m->p.box[a]) return false;
        if (h->models.size() != m->models.size() || memcmp(h->models.data(), m->models.data(), m->models.size() * sizeof(pmc_pair_model)) != 0)
            return false;
    }
    return true;
}

extern "C" int pmc_sweep_many(pmc_system **hs, int32_t n, int32_t mode, const pmc_move *pool, int32_t npool, int64_t nsweeps,
                              pmc_stats *out)
{
    if (!hs || n < 1) return fail(nullptr, PMC_ERR_BADARG, "pmc_sweep_many: bad argument");
    for (int r = 0; r < n; r++) {
        if (!hs[r]) return fail(nullptr, PMC_ERR_BADARG, "pmc_sweep_many: null handle");
        if (hs[r]->pending) return fail(hs[r], PMC_ERR_STATE, "pmc_sweep_many: a sweep is in flight on one of the handles (pmc_sweep_end first)");
        if (int rc = validate_pool(hs[r], pool, npool, nsweeps)) return rc;
    }
    if (out) memset(out, 0, (size_t)n * sizeof *out);
    if (mode == PMC_SEQUENTIAL && n > 1 && nsweeps > 0 && hs[0]->list_type == 0 && seq_empty_pool_ok(pool, npool) && seq_lds_compatible(hs, n)) {
        // small EmptyList chains of one shape and Displacements only: engine 5 (pmc_empty.h), one workgroup (a CU) per chain in one
        // launch -- 2.0e8 moves/s from 256 chains of the reference's benchmark system on, where engine 3 (a wavefront or four per
        // chain, below) reaches 1.3e8 with a thousand
        bool eok = seq_empty_fits(hs[0]);
        for (int r = 0; r < n && eok; r++) if (hs[r]->seq_engine != 0 && hs[r]->seq_engine != 5) eok = false;
        if (eok) return sweep_sequential_empty(hs, n, pool, npool, nsweeps, out);
    }
    if (mode == PMC_SEQUENTIAL && n > 1 && nsweeps > 0 && seq_pool_ok(pool, npool) && seq_lds_compatible(hs, n)) {
        // the chains of a `parallel = true` run on the reference's own chain (src/moves.jl:57-81): small systems of one shape
        // run one wavefront per chain with their state in LDS, all chains in one launch (pmc_seq.h, engine 3)
        bool lds_ok = true;
        for (int r = 0; r < n; r++) {
            const int e = hs[r]->seq_engine;
            if (e == 1 || e == 2 || !seq_lds_fits(hs[r], e == 3 ? SQ_LDS_MAX : SQ_LDS_AUTO)) lds_ok = false;
        }
        if (lds_ok) {
            const int rl = sweep_sequential_lds(hs, n, pool, npool, nsweeps, out);
            if (rl != PMC_ERR_UNSUPPORTED) return rl;
        }
    }
    if (mode == PMC_SEQUENTIAL && n > 1 && nsweeps > 0) {
        // larger chains on the speculative windows: every chain's rounds are enqueued before any is waited for
        bool fast = true;
        for (int r = 0; r < n; r++) {
            if (hs[r]->seq_engine == 1 || hs[r]->seq_engine == 3 || !seq_fast_applicable(hs[r], pool, npool)) fast = false;
            if (hs[r]->p.device != hs[0]->p.device || hs[r]->D != hs[0]->D) fast = false;
            for (int q = 0; q < r; q++) if (hs[q] == hs[r]) fast = false;
        }
        if (fast) return sweep_sequential_fast_many(hs, n, pool, npool, nsweeps, out);
    }
    if (mode == PMC_SEQUENTIAL && n > 1 && nsweeps > 0 && seq_pool_ok(pool, npool) && hs[0]->list_type == 3) {
        // VerletList chains of one shape: one workgroup per chain, state in LDS (pmc_verlet.h, engine 4)
        bool vok = true;
        for (int r = 0; r < n && vok; r++) {
            if (hs[r]->seq_engine != 0 && hs[r]->seq_engine != 4) vok = false;
            if (vok && !hs[r]->cells_built) { const int rb = pmc_build_cells(hs[r]); if (rb) return rb; }
        }
        if (vok && seq_verlet_compatible(hs, n) && seq_verlet_fits(hs[0])) {
            std::vector<long long> did((size_t)n, 0);
            const int rv = sweep_sequential_verlet(hs, n, pool, npool, nsweeps, out, did.data());
            if (rv) return rv;
            int first = PMC_OK;
            for (int r = 0; r < n; r++) {
                const long long total = nsweeps * (int64_t)hs[r]->N;
                if (did[(size_t)r] == total) continue;
                // (a list outgrew its LDS row on this chain: the rest of its call on the literal loop)
                const int rl = sweep_literal_steps(hs[r], pool, npool, total - did[(size_t)r], nsweeps, out ? &out[r] : nullptr, true);
                if (rl && !first) first = rl;
            }
            return first;
        }
    }
    if (mode == PMC_SEQUENTIAL && n > 1 && nsweeps > 0) {
        // chains only the literal loop serves (VerletList, EmptyList beyond the LDS, swaps within a species, engine 1 chosen): one
        // launch, one workgroup per chain
        bool lit = true;
        for (int r = 0; r < n; r++) {
            pmc_system *h = hs[r];
            const bool fast_ok = h->seq_engine != 1 && (seq_fast_applicable(h, pool, npool) ||
                                                        (seq_pool_ok(pool, npool) && h->seq_engine != 2 && seq_lds_fits(h, h->seq_engine == 3 ? SQ_LDS_MAX : SQ_LDS_AUTO)));
            if (fast_ok) lit = false;
            if (h->p.device != hs[0]->p.device || h->D != hs[0]->D) lit = false;
            for (int q = 0; q < r; q++) if (hs[q] == h) lit = false;
        }
        if (lit) return sweep_sequential_ref_many(hs, n, pool, npool, nsweeps, out);
    }
    if (mode != PMC_CHECKERBOARD || n == 1 || nsweeps == 0 || !many_compatible(hs, n)) {
        // different shapes, large systems (they fill the chip on their own) or the reference-order engine: one stream per handle
        int first = PMC_OK, begun = 0;
        for (; begun < n; begun++) {
            first = pmc_sweep_begin(hs[begun], mode, pool, npool, nsweeps);
            if (first) break;
        }
        for (int r = 0; r < begun; r++) {
            const int rc = pmc_sweep_end(hs[r], out ? &out[r] : nullptr);
            if (rc && !first) first = rc;
        }
        return first;
    }
    pmc_system *m = hs[0];
    int rc = set_device(m);
    if (rc) return rc;
    // The chains of a shared launch fill the chip together: the team width follows the cells that are active at the same time
    // over ALL of them (64 chains of N = 4096 -- 27 cells per colour each -- ran teams of eight at 1.8e8 moves/s, the rate of
    // eight chains).  Results do not depend on the width; the handles get their own width back when the call returns.
    struct TeamOverride {
        pmc_system **hs; int n; int saved;
        TeamOverride(pmc_system **hs_, int n_, int team) : hs(hs_), n(n_), saved(hs_[0]->team) { for (int r = 0; r < n; r++) hs[r]->team = team; }
        ~TeamOverride() { for (int r = 0; r < n; r++) hs[r]->team = saved; }
    };
    long long nact_all = n;
    for (int q = 0; q < m->D; q++) nact_all *= std::max(1, m->grid.n[q] / 2);
    TeamOverride team_guard(hs, n, getenv("PMC_TEAM") ? m->team : team_rule(m->D, nact_all));
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        rc = ensure_images(h);
        if (rc) return rc;
        if (!h->e_fix_valid) {
            rc = pmc_total_energy(h, nullptr, nullptr);
            if (rc) return rc;
        }
        for (int k = 0; k < npool; k++)
            if (pool[k].kind == PMC_MOVE_SWAP && h->S < 2) return fail(h, PMC_ERR_BADARG, "DiscreteSwap needs at least two species");
        if (h->S > 1) h->species_list_stale = true;
        if (h->epoch > 0xffffffffu - (unsigned int)std::min<int64_t>(nsweeps, 0x7fffffff) - 2u) {
            HIPCHK(h, hipMemsetAsync(h->d_done, 0, (size_t)h->ncell_mc * sizeof(unsigned int), m->stream));
            h->epoch = 0;
        }
        HIPCHK(h, hipMemsetAsync(h->d_stats, 0, PMC_NSTATS * sizeof(unsigned long long), m->stream));
        memset(h->kept_seen, 0, sizeof h->kept_seen);
        if (!h->count_is_zero) HIPCHK(h, hipMemsetAsync(h->d_count, 0, ((size_t)h->ncell_mc + 1) * sizeof(int), m->stream));
        h->count_is_zero = true;
    }
    const MovePool mp = make_pool(pool, npool);
    const int ncol = m->single_cell ? 1 : (1 << m->D), N = m->N;
    int nactive = 1;
    for (int q = 0; q < m->D; q++) nactive *= std::max(1, m->grid.n[q] / 2);
    // (one kernel instance serves the whole launch: a replica with periodic images takes the image instances, which have no
    // big-neighbourhood form -- then no chain of the launch stages big)
    struct NoBig {
        pmc_system **hs; int n;
        NoBig(pmc_system **hs_, int n_) : hs(hs_), n(n_) { bool any = false; for (int r = 0; r < n; r++) any = any || hs[r]->any_img; for (int r = 0; r < n; r++) hs[r]->no_big = any; }
        ~NoBig() { for (int r = 0; r < n; r++) hs[r]->no_big = false; }
    } nobig_guard(hs, n);
    int cap_sweep = 0;                                       // one LDS layout for all the replicas
    for (int r = 0; r < n; r++) cap_sweep = std::max(cap_sweep, cb_cap_sweep(hs[r]));
    const size_t lds = cb_lds_bytes(m, m->team >= 2 ? m->team : 0, cap_sweep);
    const int nb = nblk(N, 256);
    // replica table (once) and per-sweep table (all sweeps of the call, in chunks)
    std::vector<ManyRep> reps((size_t)n);
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        const size_t Ns = (size_t)h->N * (size_t)h->slot_rounds;
        ManyRep &q = reps[(size_t)r];
        memset(&q, 0, sizeof q);
        q.pos_ro = h->d_pos; q.pos = h->d_pos; q.species = h->d_species;
        q.cell_of = h->d_cell_of; q.count = h->d_count; q.arrival = h->d_arrival; q.start = h->d_start; q.tmp_ids = h->d_tmp_ids; q.ids = h->d_ids;
        q.sx = h->d_sx; q.sy = h->d_sy; q.sz = h->d_sz; q.ssp = h->d_ssp;
        q.img = h->any_img ? h->d_img : nullptr;
        q.simg = h->any_img ? h->d_simg : nullptr;
        q.slots = SlotOut{h->d_slot_f, h->d_slot_f + Ns, h->d_slot_f + 2 * Ns, h->d_slot_f + 3 * Ns, h->d_slot_w, h->d_slot_w + Ns, h->d_slot_m};
        q.key0 = h->key[0]; q.key1 = h->key[1];
        q.temperature = h->p.temperature;
        q.stats = h->d_stats; q.done = h->d_done;
    }
    // the passes of this call (pmc_set_rounds; many_compatible: one value for all the handles)
    const int rounds = m->rounds;
    const int64_t npass = (nsweeps + rounds - 1) / rounds;
    const int64_t chunk = PMC_ARGS_CHUNK;
    const int64_t cmax = std::min<int64_t>(chunk, npass);
    DevTmp d_reps, d_steps, d_args;
    HIPCHK(m, d_reps.alloc(reps.size() * sizeof(ManyRep)));
    HIPCHK(m, d_steps.alloc((size_t)cmax * n * sizeof(ManyStep)));
    HIPCHK(m, d_args.alloc((size_t)cmax * n * sizeof(CbArgs)));
    HIPCHK(m, hipMemcpyAsync(d_reps.p, reps.data(), reps.size() * sizeof(ManyRep), hipMemcpyHostToDevice, m->stream));
    std::vector<ManyStep> steps;
    std::vector<CbArgs> argsh;
    const MovePool &pool_ = mp;
    const int nrep_ = n;
    bool img_ = false;                     // (the image instances serve replicas without images too: their pointers are null)
    for (int r = 0; r < n; r++) img_ = img_ || hs[r]->any_img;
    const int mw_ = cb_minwaves(m, lds);
    const bool big_ = [&] { CbBig b; return !img_ && !getenv("PMC_CAP") && cb_big(m, m->team, &b) && cap_sweep == b.cap; }();
    const int grid = ncol * n * nactive;
    if (getenv("PMC_DEBUG_CAP")) fprintf(stderr, "[pmc many] chains %d team %d cap %d lds %zu minwaves %d\n", n, m->team, cap_sweep, lds, mw_);
    HIPCHK(m, hipEventRecord(m->ev[0], m->stream));
    for (int64_t s0 = 0; s0 < npass; s0 += chunk) {
        const int64_t ns = std::min<int64_t>(chunk, npass - s0);
        steps.assign((size_t)(ns * n), ManyStep{});
        argsh.assign((size_t)(ns * n), CbArgs{});
        for (int64_t sw = 0; sw < ns; sw++) {
            for (int r = 0; r < n; r++) {
                pmc_system *h = hs[r];
                const uint64_t g = h->sweep_ctr + (uint64_t)((s0 + sw) * rounds);
                ManyStep &st = steps[(size_t)(sw * n + r)];
                st.rounds = (int)std::min<int64_t>(rounds, nsweeps - (s0 + sw) * rounds);
                st.s_lo = (uint32_t)g;
                st.s_hi = (uint32_t)(g >> 32) & 0xffffffu;
                uint32_t W[4];
                philox4x32(0, 0, st.s_lo, st.s_hi | ((uint32_t)TAG_SHIFT << 24), h->key[0], h->key[1], W);
                for (int q = 0; q < h->D; q++) st.shift[q] = ((double)W[q] * (1.0 / 4294967296.0)) * h->grid.w[q];
                st.epoch = h->epoch + (unsigned int)(s0 + sw) + 1u;
                // the sweep kernel's arguments for this replica and sweep (its own arrays; the launch's shared ticket)
                for (int q = 0; q < 3; q++) h->grid.shift[q] = st.shift[q];
                CbArgs a = cb_args(h);
                a.cap = cap_sweep;
                cb_apply_big(h, cap_sweep, a);
                const size_t Ns = (size_t)h->N * (size_t)h->slot_rounds;
                a.order = nullptr;
                a.slot_tlnu = h->d_slot_f; a.slot_dx = h->d_slot_f + Ns; a.slot_dy = h->d_slot_f + 2 * Ns; a.slot_dz = h->d_slot_f + 3 * Ns;
                a.slot_w1 = h->d_slot_w; a.slot_w2 = h->d_slot_w + Ns; a.slot_m = h->d_slot_m;
                a.s_lo = st.s_lo; a.s_hi = st.s_hi;
                a.rounds = st.rounds;
                a.pool = mp;
                a.all_cells = 0;
                a.nactive = nactive;
                a.fused = 1;
                a.epoch = st.epoch;
                a.done = h->d_done;
                a.ticket = m->d_ticket;
                a.ticket_lanes = m->ticket_lanes;
                a.ticket_base = m->ticket_next + (unsigned long long)sw * (unsigned long long)grid;
                a.colour = 0;
                argsh[(size_t)(sw * n + r)] = a;
            }
        }
        if (s0 > 0) HIPCHK(m, hipStreamSynchronize(m->stream));      // the previous chunk's tables are still being read
        HIPCHK(m, hipMemcpyAsync(d_steps.p, steps.data(), steps.size() * sizeof(ManyStep), hipMemcpyHostToDevice, m->stream));
        HIPCHK(m, hipMemcpyAsync(d_args.p, argsh.data(), argsh.size() * sizeof(CbArgs), hipMemcpyHostToDevice, m->stream));
        for (int64_t sw = 0; sw < ns; sw++) {
            const ManyRep *dr = d_reps.as<ManyRep>();
            const ManyStep *ds = d_steps.as<ManyStep>() + sw * n;
            unsigned long long *const t8 = m->ticket_lanes == PMC_TICKET_LANES ? m->d_ticket : nullptr;
            if (m->D == 3) hipLaunchKernelGGL(k_many_bin_a<3>, dim3(nb, n), dim3(256), 0, m->stream, N, m->grid, dr, ds, t8);
            else hipLaunchKernelGGL(k_many_bin_a<2>, dim3(nb, n), dim3(256), 0, m->stream, N, m->grid, dr, ds, t8);
            hipLaunchKernelGGL(k_many_scan, dim3(n), dim3(1024), 0, m->stream, m->ncell_mc, dr);
            hipLaunchKernelGGL(k_many_bin_b, dim3(nb, n), dim3(256), 0, m->stream, N, dr);
            if (m->D == 3) hipLaunchKernelGGL(k_many_bin_c<3>, dim3(nb, n), dim3(256), 0, m->stream, N, mp, dr, ds, m->p.box[0], m->p.box[1], m->p.box[2]);
            else hipLaunchKernelGGL(k_many_bin_c<2>, dim3(nb, n), dim3(256), 0, m->stream, N, mp, dr, ds, m->p.box[0], m->p.box[1], 1.0);
            const CbArgs *args = d_args.as<CbArgs>() + sw * n;
            DISPATCH_CB(m, grid, lds, args);
            HIPCHK(m, hipGetLastError());
        }
        if (m->ticket_lanes == 1) m->ticket_next += (unsigned long long)ns * (unsigned long long)grid;
    }
    HIPCHK(m, hipEventRecord(m->ev[1], m->stream));
    for (int r = 0; r < n; r++) {
        hipLaunchKernelGGL(k_stats_fold, dim3(PMC_NSTATS), dim3(PMC_NSHARD), 0, m->stream, hs[r]->d_stats);
        HIPCHK(m, hipMemcpyAsync(hs[r]->h_stats, hs[r]->d_stats, PMC_NSTATS * sizeof(unsigned long long), hipMemcpyDeviceToHost, m->stream));
    }
    HIPCHK(m, hipGetLastError());
    HIPCHK(m, hipStreamSynchronize(m->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, m->ev[0], m->ev[1]);
    int first = PMC_OK;
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        const unsigned long long *st = h->h_stats;
        h->sweep_ctr += (uint64_t)nsweeps;
        h->epoch += (unsigned int)npass;
        h->cells_built = false;
        h->mc_binned = true;
        const int rcf = check_kernel_flags(h, st);
        if (rcf) {
            cb_invalidate(h);
            if (!first) first = rcf;
            if (rcf == PMC_ERR_STATE)      // a shared launch tickets through hs[0]: every chain of it leaves the eight-counter scheme
                for (int q = 0; q < n; q++) ticket_fallback(hs[q]);
            continue;
        }
        h->e_fix += (long long)st[16];
        h->energy = (double)h->e_fix / PMC_FIX;
        // one wavefront per cell (the launch's width, still in force here): the LDS staging of the next call is sized from this
        // call's survivor histogram, as after a single-chain call (256 chains of N = 4096: 7.4 -> 8.4e8 moves/s)
        if (cb_cap_sweep_adapts(h)) cb_cap_learn(h, st + PMC_STAT_KEPT);
        if (out) {
            pmc_stats &o = out[r];
            for (int k = 0; k < PMC_MAX_MOVES; k++) { o.attempted[k] = (int64_t)st[k]; o.accepted[k] = (int64_t)st[8 + k]; }
            o.kernel_ms = ms;
            o.sweep_kernel_ms = 0.0;
            o.sweep_kernel_launches = npass;
            o.energy = h->energy;
            o.energy_fix = h->e_fix;
            o.pair_tests = (int64_t)st[25];
            o.pairs_in_range = (int64_t)st[26];
            o.null_moves = (int64_t)st[27];
            o.dense_evals = (int64_t)st[24];
            o.staged_candidates = (int64_t)st[31];
            o.written_back = (int64_t)st[19];
        }
    }
    return first;
}

extern "C" int pmc_sweep(pmc_system *h, int32_t mode, const pmc_move *pool, int32_t npool, int64_t nsweeps,
                         pmc_stats *out)
{
    if (h && h->pending) return fail(h, PMC_ERR_STATE, "pmc_sweep: a sweep is in flight on this handle (pmc_sweep_end first)");
    int rc = validate_pool(h, pool, npool, nsweeps);
    if (rc) return rc;
    if (out) memset(out, 0, sizeof *out);
    rc = set_device(h);
    if (rc) return rc;
    if (mode == PMC_SEQUENTIAL) return sweep_sequential(h, pool, npool, nsweeps, out);
    if (mode == PMC_CHECKERBOARD) return sweep_checkerboard(h, pool, npool, nsweeps, out);
    return fail(h, PMC_ERR_BADARG, "pmc_sweep: unknown mode");
}

// ---- parallel tempering ---------------------------------------------------------------

extern "C" int pmc_pt_decide(uint64_t seed, uint64_t round, int32_t n, const double *energy,
                             const double *ladder, int32_t *label, int32_t *n_accepted)
{
    if (n < 1 || !energy || !ladder || !label) { g_err = "pmc_pt_decide: bad argument"; return PMC_ERR_BADARG; }
    std::vector<int> holder(n, -1);   // holder[t] = replica currently at temperature index t
    for (int r = 0; r < n; r++) {
        if (label[r] < 0 || label[r] >= n || holder[label[r]] != -1) { g_err = "pmc_pt_decide: label is not a permutation"; return PMC_ERR_BADARG; }
        holder[label[r]] = r;
    }
    int acc = 0;
    const int parity = (int)(round & 1);
    for (int t = parity; t + 1 < n; t += 2) {
        int ra = holder[t], rb = holder[t + 1];
        double ba = 1.0 / ladder[t], bb = 1.0 / ladder[t + 1];
        double arg = (ba - bb) * (energy[ra] - energy[rb]);
        uint32_t W[4];
        philox4x32((uint32_t)t, 0, (uint32_t)round, ((uint32_t)(round >> 32) & 0xffffffu) | ((uint32_t)TAG_PT << 24),
                   (uint32_t)seed, (uint32_t)(seed >> 32), W);
        double lnu = det_log(uniform53(W[0], W[1]));
        if (lnu < arg) {
            label[ra] = t + 1;
            label[rb] = t;
            acc++;
        }
    }
    if (n_accepted) *n_accepted = acc;
    return PMC_OK;
}

// One exchange round over replicas that are handles of THIS process (one GPU, or one handle per device of the node):
// the energies are the handles' running system.energy[1] (exact in checkerboard mode), the decisions those of
// pmc_pt_decide, and every handle's temperature becomes ladder[label[r]].  Configurations never move.
extern "C" int pmc_replica_exchange(pmc_system **hs, int32_t n, const double *ladder, uint64_t seed, uint64_t round,
                                    int32_t *label, int32_t *n_accepted)
{
    if (!hs || n < 1 || !ladder || !label) return fail(nullptr, PMC_ERR_BADARG, "pmc_replica_exchange: bad argument");
    std::vector<double> e((size_t)n);
    for (int r = 0; r < n; r++) {
        pmc_system *h = hs[r];
        if (!h) return fail(nullptr, PMC_ERR_BADARG, "pmc_replica_exchange: null handle");
        if (h->pending) return fail(h, PMC_ERR_STATE, "pmc_replica_exchange: a sweep is in flight on a replica (pmc_sweep_end first)");
        if (!(ladder[r] > 0.0)) return fail(h, PMC_ERR_BADARG, "pmc_replica_exchange: temperatures must be positive");
        if (int rc = set_device(h)) return rc;
        if (int rc = refresh_stale_energy(h)) return rc;
        e[(size_t)r] = h->energy;
    }
    const int rc = pmc_pt_decide(seed, round, n, e.data(), ladder, label, n_accepted);
    if (rc) return rc;
    for (int r = 0; r < n; r++) hs[r]->p.temperature = ladder[label[r]];
    return PMC_OK;
}

// ---- EnergyBias on the device (src/moves.jl:246-280) ---------------------------------------------------------------
static int energy_bias(pmc_system *h, int32_t sp1, int32_t sp2, double theta1, double theta2, double u1, double u2,
                       int64_t gi, int64_t gj, int64_t *i, int64_t *j, double *logq)
{
    if (!h) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias: null handle");
    if (h->pending) return fail(h, PMC_ERR_STATE, "a sweep is in flight on this handle: call pmc_sweep_end first");
    if (sp1 < 1 || sp1 > h->S || sp2 < 1 || sp2 > h->S) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias: species out of range");
    if (!(u1 >= 0.0 && u1 < 1.0 && u2 >= 0.0 && u2 < 1.0)) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias: uniforms must lie in [0, 1)");
    int rc = set_device(h);
    if (rc) return rc;
    if (!h->cells_built) {
        rc = pmc_build_cells(h);
        if (rc) return rc;
    }
    rc = ensure_species_list(h);
    if (rc) return rc;
    DevTmp buf;
    HIPCHK(h, buf.alloc(((size_t)2 * h->N + 8) * sizeof(double)));
    double *d_e = buf.as<double>(), *d_w = d_e + h->N, *d_out = d_w + h->N;
    rc = ref_energy_launch(h, nullptr, h->N, d_e);        // [compute_energy_particle(system, i) for i], reference order
    if (rc) return rc;
    BiasArgs a;
    memset(&a, 0, sizeof a);
    a.e = d_e; a.sp_off = h->d_sp_off; a.sp_ids = h->d_sp_ids;
    a.sp[0] = sp1 - 1; a.sp[1] = sp2 - 1;
    a.theta[0] = theta1; a.theta[1] = theta2;
    a.u[0] = u1; a.u[1] = u2;
    a.given[0] = gi > 0 ? (int)gi - 1 : -1;
    a.given[1] = gj > 0 ? (int)gj - 1 : -1;
    a.w = d_w; a.out = d_out;
    hipLaunchKernelGGL(k_energy_bias, dim3(1), dim3(256), 0, h->stream, a);
    HIPCHK(h, hipGetLastError());
    double out[5];
    HIPCHK(h, hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (out[0] < 0.0 || out[1] < 0.0) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias: a species of the pair has no particles");
    if (i) *i = (int64_t)out[0] + 1;
    if (j) *j = (int64_t)out[1] + 1;
    if (logq) *logq = out[2];
    return PMC_OK;
}

extern "C" int pmc_energy_bias_sample(pmc_system *h, int32_t sp1, int32_t sp2, double theta1, double theta2, double u1, double u2,
                                      int64_t *i, int64_t *j, double *logq)
{
    if (!i || !j) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias_sample: null argument");
    return energy_bias(h, sp1, sp2, theta1, theta2, u1, u2, 0, 0, i, j, logq);
}

extern "C" int pmc_energy_bias_logq(pmc_system *h, int32_t sp1, int32_t sp2, double theta1, double theta2, int64_t i, int64_t j,
                                    double *logq)
{
    if (!h || !logq || i < 1 || j < 1 || i > h->N || j > h->N) return fail(h, PMC_ERR_BADARG, "pmc_energy_bias_logq: bad argument");
    return energy_bias(h, sp1, sp2, theta1, theta2, 0.0, 0.0, i, j, nullptr, nullptr, logq);
}

extern "C" int pmc_exp(double x, double *y)          // the deterministic exp the weights use (bit-parity tests)
{
    if (!y) return PMC_ERR_BADARG;
    *y = det_exp(x);
    return PMC_OK;
}

// ---- RCCL (one process per GPU): librccl.so is 570 MB, so it is opened on first use, not linked -----------------
namespace {
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi *rccl_api()
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        // PMC_RCCL_LIB: another library with RCCL's entry points (tests/c/rccl_stub.c lets several processes that share one GPU
        // drive this path; the real RCCL needs one device per rank)
        if (const char *over = getenv("PMC_RCCL_LIB")) api.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        else
            for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
                api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
                if (api.lib) break;
            }
        if (api.lib) {
            api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(dlsym(api.lib, "ncclGetUniqueId"));
            api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(dlsym(api.lib, "ncclCommInitRank"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(dlsym(api.lib, "ncclCommDestroy"));
            api.AllGather = reinterpret_cast<decltype(api.AllGather)>(dlsym(api.lib, "ncclAllGather"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(dlsym(api.lib, "ncclGetErrorString"));
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather) { dlclose(api.lib); api.lib = nullptr; }
        }
    }
    return api.lib ? &api : nullptr;
}
int rccl_fail(pmc_system *h, RcclApi *api, const char *what, ncclResult_t r)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed: %s", what, (api && api->GetErrorString) ? api->GetErrorString(r) : "RCCL error");
    return fail(h, PMC_ERR_HIP, buf);
}
}   // namespace

extern "C" int pmc_rccl_unique_id(void *id128)
{
    if (!id128) return fail(nullptr, PMC_ERR_BADARG, "pmc_rccl_unique_id: null argument");
    RcclApi *api = rccl_api();
    if (!api) return fail(nullptr, PMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    const ncclResult_t r = api->GetUniqueId(static_cast<ncclUniqueId *>(id128));
    return r == ncclSuccess ? PMC_OK : rccl_fail(nullptr, api, "ncclGetUniqueId", r);
}

extern "C" int pmc_rccl_comm_create(const void *id128, int32_t rank, int32_t world, int32_t device, void **comm)
{
    if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) return fail(nullptr, PMC_ERR_BADARG, "pmc_rccl_comm_create: bad argument");
    RcclApi *api = rccl_api();
    if (!api) return fail(nullptr, PMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    HIPCHK((pmc_system *)nullptr, hipSetDevice(device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclComm_t c = nullptr;
    const ncclResult_t r = api->CommInitRank(&c, world, id, rank);
    if (r != ncclSuccess) return rccl_fail(nullptr, api, "ncclCommInitRank", r);
    *comm = c;
    return PMC_OK;
}

extern "C" int pmc_rccl_comm_destroy(void *comm)
{
    RcclApi *api = rccl_api();
    if (!api || !comm) return fail(nullptr, PMC_ERR_BADARG, "pmc_rccl_comm_destroy: bad argument");
    const ncclResult_t r = api->CommDestroy(static_cast<ncclComm_t>(comm));
    return r == ncclSuccess ? PMC_OK : rccl_fail(nullptr, api, "ncclCommDestroy", r);
}

// The same round with one replica per process: ONE double per rank goes through ncclAllGather (RCCL over xGMI) on
// the handle's stream, every rank then takes the same decisions and adopts its temperature.  label has `world` entries.
extern "C" int pmc_replica_exchange_rccl(pmc_system *h, void *comm, int32_t rank, int32_t world, const double *ladder,
                                         uint64_t seed, uint64_t round, int32_t *label, int32_t *n_accepted)
{
    if (!h || !comm || !ladder || !label || world < 1 || rank < 0 || rank >= world)
        return fail(h, PMC_ERR_BADARG, "pmc_replica_exchange_rccl: bad argument");
    if (h->pending) return fail(h, PMC_ERR_STATE, "pmc_replica_exchange_rccl: a sweep is in flight (pmc_sweep_end first)");
    RcclApi *api = rccl_api();
    if (!api) return fail(h, PMC_ERR_UNSUPPORTED, "librccl.so could not be loaded");
    if (int rc = set_device(h)) return rc;
    if (int rc = refresh_stale_energy(h)) return rc;
    DevTmp buf;
    HIPCHK(h, buf.alloc((size_t)(world + 1) * sizeof(double)));
    double *d_send = buf.as<double>(), *d_recv = d_send + 1;
    HIPCHK(h, hipMemcpyAsync(d_send, &h->energy, sizeof(double), hipMemcpyHostToDevice, h->stream));
    const ncclResult_t r = api->AllGather(d_send, d_recv, 1, ncclDouble, static_cast<ncclComm_t>(comm), h->stream);
    if (r != ncclSuccess) return rccl_fail(h, api, "ncclAllGather", r);
    std::vector<double> e((size_t)world);
    HIPCHK(h, hipMemcpyAsync(e.data(), d_recv, (size_t)world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    const int rc = pmc_pt_decide(seed, round, world, e.data(), ladder, label, n_accepted);
    if (rc) return rc;
    h->p.temperature = ladder[label[rank]];
    return PMC_OK;
}

// ---- host access to the random streams ---------------------------------------------------

extern "C" int pmc_philox4x32(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    if (!ctr || !key || !out) return PMC_ERR_BADARG;
    philox4x32(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
    return PMC_OK;
}

extern "C" int pmc_box_muller(uint32_t w1, uint32_t w2, double *z0, double *z1)
{
    if (!z0 || !z1) return PMC_ERR_BADARG;
    box_muller(w1, w2, z0, z1);
    return PMC_OK;
}

extern "C" int pmc_log_uniform(uint32_t hi, uint32_t lo, double *lnu)
{
    if (!lnu) return PMC_ERR_BADARG;
    *lnu = det_log(uniform53(hi, lo));
    return PMC_OK;
}

extern "C" int pmc_sequential_words(uint64_t seed, uint32_t chain, uint64_t step, uint32_t R[4], uint32_t Q[4])
{
    if (!R || !Q) return PMC_ERR_BADARG;
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32) + chain;
    philox4x32((uint32_t)step, (uint32_t)(step >> 32), 0, TAG_SEQ0, k0, k1, R);
    philox4x32((uint32_t)step, (uint32_t)(step >> 32), 0, TAG_SEQ1, k0, k1, Q);
    return PMC_OK;
}

extern "C" int pmc_debug_div(pmc_system *h, int64_t n, const double *num, const double *den, double *q_ieee,
                             double *q_fast)
{
    if (!h || n < 1 || !num || !den || !q_ieee || !q_fast) return fail(h, PMC_ERR_BADARG, "pmc_debug_div: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    DevTmp tmp;
    HIPCHK(h, tmp.alloc((size_t)n * 4 * sizeof(double)));
    double *d = tmp.as<double>();
    (void)hipMemcpy(d, num, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    (void)hipMemcpy(d + n, den, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_debug_div, dim3(nblk((int)n, 256)), dim3(256), 0, h->stream, (int)n, d, d + n, d + 2 * n, d + 3 * n);
    hipError_t er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess) er = hipMemcpy(q_ieee, d + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(q_fast, d + 3 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (er != hipSuccess) return fail(h, PMC_ERR_HIP, hipGetErrorString(er));
    return PMC_OK;
}

// ---- debug ----------------------------------------------------------------------------

extern "C" int pmc_debug_rng(pmc_system *h, int64_t n, const uint32_t *w1, const uint32_t *w2, double *z0, double *z1,
                             double *lnu)
{
    if (!h || n < 1 || !w1 || !w2 || !z0 || !z1 || !lnu) return fail(h, PMC_ERR_BADARG, "pmc_debug_rng: bad argument");
    int rc = set_device(h);
    if (rc) return rc;
    DevTmp tw, to;
    HIPCHK(h, tw.alloc((size_t)n * 2 * sizeof(uint32_t)));
    HIPCHK(h, to.alloc((size_t)n * 3 * sizeof(double)));
    uint32_t *d_w = tw.as<uint32_t>();
    double *d_o = to.as<double>();
    (void)hipMemcpy(d_w, w1, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_w + n, w2, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_debug_rng, dim3(nblk((int)n, 256)), dim3(256), 0, h->stream, (int)n, d_w, d_w + n, d_o, d_o + n,
                       d_o + 2 * n);
    hipError_t er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess) er = hipMemcpy(z0, d_o, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(z1, d_o + n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(lnu, d_o + 2 * n, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (er != hipSuccess) return fail(h, PMC_ERR_HIP, hipGetErrorString(er));
    return PMC_OK;
}
