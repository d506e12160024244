// gu_rollout_sample.hip -- instantiates the fused rollout kernel (gu_rollout.hpp) for GU_POLICY_SAMPLE.
#include "gu_rollout.hpp"

void gu_rollout_sample(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs)
{
    gu_rollout_dispatch<GU_POLICY_SAMPLE>(h, a, auto_mode, traj, stats, bs);
}
