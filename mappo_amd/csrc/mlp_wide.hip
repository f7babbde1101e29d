// wide-input kernels (in_dim 65..512, mlp_wide16.h): layer-1 forward with LDS-resident weights, layer-1 weight gradient
// (the rollout forwards are compiled in mlp_wide_fwd.hip / mlp_wide_sk.hip: one translation unit took ten minutes)
#define MLP_TU_WIDE
#include "mlp_impl.h"
