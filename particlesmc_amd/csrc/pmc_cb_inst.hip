// Explicit instantiations of the sweep kernel k_cb_pass for ONE (dimension, model kind), given as -DPMC_INST_D / -DPMC_INST_KIND:
// the Makefile compiles the ten of them in parallel and links them with pmc_hip.hip, which only declares them
// (extern template).  One translation unit with all 200 instances takes four minutes to compile; this way it is one.
#define PMC_INST_ONLY
#include "pmc_kernels.h"

#define PMC_INSTANTIATE(M, NW, ONE, MW) \
    template __global__ void k_cb_pass<PMC_INST_D, PMC_INST_KIND, M, NW, ONE, MW>(const CbArgs *__restrict__ args, int nrep);
PMC_CB_VARIANTS(PMC_INSTANTIATE)
