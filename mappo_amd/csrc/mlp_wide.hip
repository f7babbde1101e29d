// wide-input kernels (in_dim 65..512, mlp_wide16.h): layer-1 forward, rollout forward, layer-1 weight gradient
#define MLP_TU_WIDE
#include "mlp_impl.h"
