// mlp_update16_kernel / mlp_update16_dual_kernel<RELU=true, LN=0> — one wave per 16-sample tile (mlp_upd16.h)
#define MLP_TU_UPD16
#define MLP_UPD_RELU true
#define MLP_UPD_LN 0
#include "mlp_impl.h"
