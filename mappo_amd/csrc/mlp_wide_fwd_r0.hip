// wide-input kernels (mlp_wide16.h): the streamed one-launch forward and its dual form, activation = tanh
#define MLP_TU_WIDE_FWD
#define MLP_WIDE_RELU false
#include "mlp_impl.h"
