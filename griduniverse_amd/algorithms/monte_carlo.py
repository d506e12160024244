"""run_episode with the reference's signature (core/algorithms/monte_carlo.py:7-26): the
canonical reset / sample-action / step loop on the N = 1 facade.  For throughput use
`VecGridUniverse.rollout`, which fuses this loop for thousands of envs into one launch."""
import numpy as np


def run_episode(policy, env, max_steps_per_episode=1000):
    states_hist, rewards_hist = [], []
    observation = env.reset()
    states_hist.append(observation)
    done = False
    for _ in range(max_steps_per_episode):
        action = np.random.choice(policy[observation].size, p=policy[observation])  # numpy GLOBAL rng, as the reference
        observation, reward, done, _ = env.step(action)
        states_hist.append(observation)
        rewards_hist.append(reward)
        if done:
            break
    return states_hist, rewards_hist, done
