// Explicit instantiations of the sweep kernel k_cb_pass for ONE (dimension, model kind), given as -DPMC_INST_D / -DPMC_INST_KIND:
// the Makefile compiles them in parallel and links them with pmc_hip.hip, which only declares them (extern template).  One
// translation unit with all the instances takes minutes to compile; this way it is about one.  -DPMC_INST_IMG selects the
// instances for systems whose particles carry periodic images (pmc_kernels.h "images").
#define PMC_INST_ONLY
#include "pmc_kernels.h"

#ifdef PMC_INST_IMG
#define PMC_INSTANTIATE_IMG(M, NW, ONE) \
    template __global__ void k_cb_pass<PMC_INST_D, PMC_INST_KIND, M, NW, ONE, PMC_IMG_MW(NW), true>(const CbArgs *__restrict__ args, int nrep);
PMC_CB_VARIANTS_IMG(PMC_INSTANTIATE_IMG)
#else
#define PMC_INSTANTIATE(M, NW, ONE, MW) \
    template __global__ void k_cb_pass<PMC_INST_D, PMC_INST_KIND, M, NW, ONE, MW>(const CbArgs *__restrict__ args, int nrep);
PMC_CB_VARIANTS(PMC_INSTANTIATE)
#define PMC_INSTANTIATE_BIG(M, ONE) \
    template __global__ void k_cb_pass<PMC_INST_D, PMC_INST_KIND, M, 8, ONE, PMC_TEAM_MINWAVES, false, true>(const CbArgs *__restrict__ args, int nrep);
PMC_CB_VARIANTS_BIG(PMC_INSTANTIATE_BIG)
#endif
