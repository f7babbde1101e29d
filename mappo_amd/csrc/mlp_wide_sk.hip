// wide-input kernels (mlp_wide16.h): the split-K forward for step-sized batches and its dual form
#define MLP_TU_WIDE_SK
#include "mlp_impl.h"
