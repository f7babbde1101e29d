// mlp_step.hip — the fused rollout step: actor get_actions + critic get_values + MPE insert in one launch (see mlp_impl.h)
#define MLP_TU_STEP
#include "mlp_impl.h"
