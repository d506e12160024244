// gu_rollout_stream.hip -- instantiates the fused rollout kernel (gu_rollout.hpp) for GU_POLICY_STREAM.
#include "gu_rollout.hpp"

void gu_rollout_stream(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs)
{
    gu_rollout_dispatch<GU_POLICY_STREAM>(h, a, auto_mode, traj, stats, bs);
}
