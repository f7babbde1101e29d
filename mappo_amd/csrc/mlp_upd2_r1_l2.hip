// mlp_update2_kernel<RELU=true, LN=2, HEAD 0..3, WIDE 0..1> — the pair update kernel (see mlp_upd2.h)
#define MLP_TU_UPD2
#define MLP_UPD_RELU true
#define MLP_UPD_LN 2
#include "mlp_impl.h"
