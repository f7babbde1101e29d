// mlp_update_kernel<RELU=false, LN=0, HEAD 0..3, XW 0..2> (see mlp_impl.h)
#define MLP_TU_UPD
#define MLP_UPD_RELU false
#define MLP_UPD_LN 0
#include "mlp_impl.h"
