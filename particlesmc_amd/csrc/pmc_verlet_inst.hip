// The instances of k_sq_verlet (pmc_verlet.h: engine 4, VerletList chains with their state in LDS) in a translation unit of their
// own: pmc_hip.hip sees the argument structures and asks for a kernel by (dimension, model kind); a change of the kernel
// rebuilds this file in seconds instead of pmc_hip.hip in minutes.
#define PMC_INST_ONLY
#include <cstring>
#include "pmc_verlet.h"
#include "pmc_empty.h"

static size_t sqv_align(size_t x, size_t a) { return (x + a - 1) / a * a; }

// LDS bytes of k_sq_verlet for a system of this shape with rows of `vcap` ids and cell vectors of `ccap` (and the offsets of its arrays)
size_t pmc_sqv_layout(int n_, int d, int nspecies, int ncell, int nn_, int vcap, int ccap, SqvArgs *s)
{
    const size_t N = (size_t)n_, nc = (size_t)ncell, nn = (size_t)nn_, nb = SQV_NB;
    const size_t t0 = (size_t)((vcap + 7) / 8 * 8 + SQ_TERMS_PAD), t1 = 3 * SQ_TERMS_PAD;      // (eight zeros behind the terms; a chain requests the block behind the one it adds)
    const size_t tstride = t0 > t1 ? t0 : t1;
    size_t o = sqv_align(N * d * sizeof(double), 16);
    const size_t o_vrows = o;     o = sqv_align(o + N * vcap * 2 + 8, 16);       // (+ the word a half row reads behind the last row)
    const size_t o_crows = o;     o = sqv_align(o + nc * ccap * 2, 16);
    const size_t o_cs = o;        o = sqv_align(o + N * 2, 16);
    const size_t o_vcnt = o;      o = sqv_align(o + N * 2, 16);
    const size_t o_ccnt = o;      o = sqv_align(o + nc * 2, 16);
    const size_t o_nbr = o;       o = sqv_align(o + nc * nn * 2, 16);
    const size_t o_nbrcnt = o;    o = sqv_align(o + nc * 2, 16);
    const size_t o_species = o;   o = sqv_align(o + N, 16);
    const size_t o_touch = o;     o = sqv_align(o + N * 2 + 4, 16);
    const size_t o_ctouch = o;    o = sqv_align(o + nc * 2 + 4, 16);
    const size_t o_gen = o;       o = sqv_align(o + SQV_GEN * sizeof(SqGen), 16);
    const size_t o_spec = o;      o = sqv_align(o + nb * sizeof(SqvSpec), 16);
    const size_t o_lst = o;       o = sqv_align(o + nb * (size_t)vcap * 2, 16);
    const size_t o_models = o;    o = sqv_align(o + (size_t)nspecies * nspecies * sizeof(DevModel), 16);
    const size_t o_terms = o;     o = sqv_align(o + nb * 2 * tstride * sizeof(double), 16);      // (the evaluators' only)
    const size_t o_ctl = o;       o = sqv_align(o + 64 * sizeof(int), 16);
    if (s) {
        s->vcap = vcap; s->ccap = ccap; s->tstride = (int)tstride;
        s->o_vrows = (int)o_vrows; s->o_crows = (int)o_crows; s->o_cs = (int)o_cs; s->o_vcnt = (int)o_vcnt;
        s->o_ccnt = (int)o_ccnt; s->o_nbr = (int)o_nbr; s->o_nbrcnt = (int)o_nbrcnt; s->o_species = (int)o_species; s->o_touch = (int)o_touch;
        s->o_ctouch = (int)o_ctouch; s->o_gen = (int)o_gen; s->o_spec = (int)o_spec; s->o_lst = (int)o_lst; s->o_models = (int)o_models;
        s->o_terms = (int)o_terms; s->o_ctl = (int)o_ctl;
    }
    return o;
}

int pmc_sqv_threads() { return SQV_NW * 64; }

const void *pmc_sqv_kernel(int d, int kind)
{
#define SQV_PICK(DD) (kind == 0 ? (const void *)k_sq_verlet<DD, 0> : kind == 1 ? (const void *)k_sq_verlet<DD, 1> : \
                      kind == 2 ? (const void *)k_sq_verlet<DD, 2> : kind == 3 ? (const void *)k_sq_verlet<DD, 3> : (const void *)k_sq_verlet<DD, 4>)
    return d == 3 ? SQV_PICK(3) : SQV_PICK(2);
#undef SQV_PICK
}

// LDS bytes of k_sq_empty for a system of this shape (and the offsets of its arrays)
size_t pmc_sqe_layout(int n_, int d, int nspecies, int nw, SqeArgs *s)
{
    const size_t N = (size_t)n_, nb = (size_t)(3 * nw / 4);
    const size_t k8 = (N + 7) / 8 * 8 + 2 * SQ_TERMS_PAD, kc = k8 > 32 ? k8 : 32;      // (eight zeros behind the terms, and the block a chain requests behind the one it adds)
    size_t o = sqv_align(N * d * sizeof(double), 16);
    const size_t o_species = o;   o = sqv_align(o + N, 16);
    const size_t o_gen = o;       o = sqv_align(o + SQV_GEN * sizeof(SqGen), 16);
    const size_t o_spec = o;      o = sqv_align(o + nb * sizeof(SqeSpec), 16);
    const size_t o_models = o;    o = sqv_align(o + (size_t)nspecies * nspecies * sizeof(DevModel), 16);
    const size_t o_log = o;       o = sqv_align(o + SQE_LOG * 2 * sizeof(int), 16);
    const size_t o_ctl = o;       o = sqv_align(o + 24 * sizeof(int), 16);
    const size_t ev_stride = sqv_align(2 * kc * sizeof(double) + 2 * (kc / 8) * sizeof(double) + 2 * kc * 2 + 2 * N * 2, 16);
    const size_t o_ev = o;        o = o + (nb + 1) * ev_stride;                    // the evaluators' blocks, then the resolver's own
    if (s) {
        s->kc = (int)kc;
        s->o_species = (int)o_species; s->o_gen = (int)o_gen; s->o_spec = (int)o_spec; s->o_models = (int)o_models; s->o_log = (int)o_log;
        s->o_ctl = (int)o_ctl; s->o_ev = (int)o_ev; s->ev_stride = (int)ev_stride;
    }
    return o;
}

const void *pmc_sqe_kernel(int d, int kind, int nw)
{
#define SQE_PICK(DD, WW) (kind == 0 ? (const void *)k_sq_empty<DD, 0, WW> : kind == 1 ? (const void *)k_sq_empty<DD, 1, WW> : \
                          kind == 2 ? (const void *)k_sq_empty<DD, 2, WW> : kind == 3 ? (const void *)k_sq_empty<DD, 3, WW> : (const void *)k_sq_empty<DD, 4, WW>)
    if (nw == 8) return d == 3 ? SQE_PICK(3, 8) : SQE_PICK(2, 8);
    return d == 3 ? SQE_PICK(3, 16) : SQE_PICK(2, 16);
#undef SQE_PICK
}

#ifdef PMC_DEBUG_TIME
__device__ unsigned long long g_sqvt[16];
void pmc_sqv_times(unsigned long long *out16)
{
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_sqvt), 16 * sizeof(unsigned long long)) != hipSuccess) std::memset(out16, 0, 16 * sizeof(unsigned long long));
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sqvt), z, sizeof z);
}
#else
void pmc_sqv_times(unsigned long long *out16) { std::memset(out16, 0, 16 * sizeof(unsigned long long)); }
#endif
