// gu_rollout_uniform.hip -- instantiates the fused rollout kernel (gu_rollout.hpp) for GU_POLICY_UNIFORM.
#include "gu_rollout.hpp"

void gu_rollout_uniform(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs)
{
    gu_rollout_dispatch<GU_POLICY_UNIFORM>(h, a, auto_mode, traj, stats, bs);
}
