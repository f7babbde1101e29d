// mlp_update_kernel<RELU=false, LN=2, HEAD 0..3, XW 0..2> (see mlp_impl.h)
#define MLP_TU_UPD
#define MLP_UPD_RELU false
#define MLP_UPD_LN 2
#include "mlp_impl.h"
