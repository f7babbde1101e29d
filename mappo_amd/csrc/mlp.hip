// mlp.hip — host entry points of the MLP kernels + the forward / wide-input / statistics kernels (see mlp_impl.h).
// The update kernel's instantiations live in mlp_upd_r*_l*.hip so that they compile in parallel.
#define MLP_TU_MAIN
#include "mlp_impl.h"
