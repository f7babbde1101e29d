// mlp_update2_dual_kernel<RELU=false, LN=2, WIDE_A 0..1, WIDE_C 0..1> — actor + critic update in one launch (mlp_upd2.h)
#define MLP_TU_UPD2D
#define MLP_UPD_RELU false
#define MLP_UPD_LN 2
#include "mlp_impl.h"
