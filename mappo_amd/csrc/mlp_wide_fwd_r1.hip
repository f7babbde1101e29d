// wide-input kernels (mlp_wide16.h): the streamed one-launch forward and its dual form, activation = ReLU
#define MLP_TU_WIDE_FWD
#define MLP_WIDE_RELU true
#include "mlp_impl.h"
