// clrs_mw_inst.hip -- the device code of the multi-word kernels for ONE limb count (-DMW_INST_K=4, 5, 6, 8): explicit
// instantiations of the templates that clrs_mw.hip declares `extern template` (clrs_mw_inst.h).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "../../include/clrs_hip.h"
#include "clrs_mw_kernels.hip.h"
#include "clrs_mw_ipm.hip.h"
#include "clrs_mw_inst.h"

#ifndef MW_INST_K
#define MW_INST_K 5          // the build (clusteredlowranksolver.jl_amd/_lib.py) compiles this unit once per limb count 4, 5, 6, 8
#endif
MW_KERNELS_ALL(template, MW_INST_K)
