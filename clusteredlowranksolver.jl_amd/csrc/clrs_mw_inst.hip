// clrs_mw_inst.hip -- the device code of the multi-word kernels for ONE limb count (-DMW_INST_K=4, 5, 6, 8): explicit
// instantiations of the templates that clrs_mw.hip declares `extern template` (clrs_mw_inst.h).  Compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include "../../include/clrs_hip.h"
#include "clrs_mw_kernels.hip.h"
#include "clrs_mw_pipe.hip.h"
#include "clrs_mw_exact.hip.h"
#include "clrs_mw_ipm.hip.h"
#include "clrs_mw_inst.h"

#ifndef MW_INST_K
#define MW_INST_K 5          // the build (clusteredlowranksolver.jl_amd/_lib.py) compiles this unit once per limb count 4, 5, 6, 8
#endif
#ifndef MW_INST_PART
#define MW_INST_PART 0       // 0: every kernel of this limb count; 1 / 2 / 3 / 4: those without data limbs / with 1 / with 2 / with K (the largest counts are split further)
#endif
#if MW_INST_PART == 0
MW_KERNELS_ALL(template, MW_INST_K)
#elif MW_INST_PART == 1
MW_KERNELS_K(template, MW_INST_K)
#elif MW_INST_PART == 2
MW_KERNELS_KD(template, MW_INST_K, 1)
MW_KERNELS_KDX(template, MW_INST_K, 1)
#elif MW_INST_PART == 3
MW_KERNELS_KD(template, MW_INST_K, 2)
MW_KERNELS_KDX(template, MW_INST_K, 2)
#else
MW_KERNELS_KD(template, MW_INST_K, MW_INST_K)
#endif
