/* abi_parity.c -- the C ABI driven from plain C (no Python, no C++): build the config-4 grid as row bit-planes,
 * run reset + one fused rollout + a few single steps through libgu.so, and compare every value with the C oracle
 * (oracle/gu_oracle.c, linked in as the checker).  Built and run by tests/test_gpu_c_abi.py on the GPU box.
 *
 *   gcc -std=c11 -O2 -Iinclude tests/c_abi/abi_parity.c -o abi_parity \
 *       -Lgriduniverse_amd/lib -lgu -Loracle/_build -lgu_oracle -Wl,-rpath,... -lm
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gu.h"

typedef struct {
    int32_t W, H;
    const uint8_t *wall, *lava, *goal;
    const int32_t *reward, *starts;
    int32_t n_starts;
} gu_oracle_grid;

void gu_oracle_reset(const gu_oracle_grid *g, uint64_t seed, int64_t env_id0, int64_t n, const uint8_t *mask,
                     int32_t *pos, int32_t *done, uint32_t *episode);
void gu_oracle_rollout(const gu_oracle_grid *g, uint64_t seed, int64_t env_id0, int64_t n, int64_t T, int32_t auto_reset,
                       const int32_t *actions, const double *pi, int32_t *pos, int32_t *done, uint32_t *episode,
                       uint64_t *tcount, int32_t *obs_out, int32_t *reward_out, int32_t *done_out, int64_t *ret_out,
                       int32_t *episodes_out);

#define CHECK(call)                                                        \
    do {                                                                   \
        int rc_ = (call);                                                  \
        if (rc_ != GU_OK) {                                                \
            char msg[512];                                                 \
            gu_last_error(msg, sizeof msg);                                \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, msg);            \
            return 1;                                                      \
        }                                                                  \
    } while (0)

enum { W = 32, H = 32, S = W * H, N = 3000, T = 257 };

int main(void)
{
    /* config 4: open 32x32 grid, start 0, goal 1023, lava column 16 + 32 r (r < 24), plus two walls */
    static uint8_t wall[S], lava[S], goal[S];
    static int32_t reward[S];
    uint32_t wall_rows[H] = {0}, goal_rows[H] = {0}, lava_rows[H] = {0};
    for (int s = 0; s < S; ++s) reward[s] = -1;
    goal[S - 1] = 1; reward[S - 1] = 10; goal_rows[H - 1] |= 1u << (W - 1);
    for (int r = 0; r < 24; ++r) { int s = 16 + 32 * r; lava[s] = 1; reward[s] = -10; lava_rows[s / W] |= 1u << (s % W); }
    wall[40] = wall[75] = 1; wall_rows[40 / W] |= 1u << (40 % W); wall_rows[75 / W] |= 1u << (75 % W);
    const int32_t starts[2] = {0, 33};
    const gu_oracle_grid og = {W, H, wall, lava, goal, reward, starts, 2};
    const uint64_t seed = 0xC0FFEEull;
    const int64_t env_id0 = 96;

    gu_handle h = NULL;
    CHECK(gu_create(0, N, env_id0, &h));
    CHECK(gu_set_grid(h, W, H, 1, wall_rows, goal_rows, lava_rows, NULL, NULL, starts, 2));
    CHECK(gu_seed(h, seed));

    int32_t *obs = malloc(sizeof(int32_t) * N * T), *rew = malloc(sizeof(int32_t) * N * T), *don = malloc(sizeof(int32_t) * N * T);
    int32_t *o_obs = malloc(sizeof(int32_t) * N * T), *o_rew = malloc(sizeof(int32_t) * N * T), *o_don = malloc(sizeof(int32_t) * N * T);
    static int32_t pos[N], done[N], first[N], acts[N], s_obs[N], s_rew[N], s_don[N];
    static uint32_t episode[N];
    static uint64_t tcount[N];

    CHECK(gu_reset(h, NULL, NULL, first));
    gu_oracle_reset(&og, seed, env_id0, N, NULL, pos, done, episode);
    if (memcmp(first, pos, sizeof pos)) { fprintf(stderr, "reset differs\n"); return 1; }

    CHECK(gu_reserve_trajectory(h, T));
    CHECK(gu_rollout(h, T, GU_POLICY_UNIFORM, GU_F_AUTO_RESET | GU_F_TRAJECTORY));
    CHECK(gu_read_trajectory(h, 0, T, obs, rew, don));
    gu_oracle_rollout(&og, seed, env_id0, N, T, 1, NULL, NULL, pos, done, episode, tcount, o_obs, o_rew, o_don, NULL, NULL);
    if (memcmp(obs, o_obs, sizeof(int32_t) * N * T) || memcmp(rew, o_rew, sizeof(int32_t) * N * T) ||
        memcmp(don, o_don, sizeof(int32_t) * N * T)) { fprintf(stderr, "rollout differs\n"); return 1; }

    for (int k = 0; k < 5; ++k) {
        for (int i = 0; i < N; ++i) acts[i] = (i + k) & 3;
        CHECK(gu_step(h, acts, GU_F_AUTO_RESET, s_obs, s_rew, s_don));
        gu_oracle_rollout(&og, seed, env_id0, N, 1, 1, acts, NULL, pos, done, episode, tcount, o_obs, o_rew, o_don, NULL, NULL);
        if (memcmp(s_obs, o_obs, sizeof s_obs) || memcmp(s_rew, o_rew, sizeof s_rew) || memcmp(s_don, o_don, sizeof s_don)) {
            fprintf(stderr, "step %d differs\n", k);
            return 1;
        }
    }
    /* an action outside 0..3 is caught by the kernel: GU_ERR_INVALID, env 7 does not step, every other env does */
    acts[7] = 9;
    if (gu_step(h, acts, 0, NULL, NULL, NULL) != GU_ERR_INVALID) { fprintf(stderr, "bad action accepted\n"); return 1; }
    {
        const int32_t keep_pos = pos[7], keep_done = done[7];
        const uint32_t keep_ep = episode[7];
        const uint64_t keep_t = tcount[7];
        acts[7] = 0;
        gu_oracle_rollout(&og, seed, env_id0, N, 1, 0, acts, NULL, pos, done, episode, tcount, o_obs, o_rew, o_don, NULL, NULL);
        pos[7] = keep_pos, done[7] = keep_done, episode[7] = keep_ep, tcount[7] = keep_t;
        static int32_t g_pos[N], g_done[N];
        static uint32_t g_ep[N];
        static uint64_t g_t[N];
        CHECK(gu_get_state(h, g_pos, g_done, g_ep, g_t));
        if (memcmp(g_pos, pos, sizeof pos) || memcmp(g_done, done, sizeof done) || memcmp(g_ep, episode, sizeof episode) ||
            memcmp(g_t, tcount, sizeof tcount)) { fprintf(stderr, "state after a rejected action differs\n"); return 1; }
    }
    int32_t idx[N], count = -1;
    CHECK(gu_done_indices(h, idx, &count));
    int want = 0;
    for (int i = 0; i < N; ++i) want += done[i] != 0;
    if (count != want) { fprintf(stderr, "done count %d != %d\n", count, want); return 1; }
    CHECK(gu_destroy(h));
    printf("PASS abi_parity: %d envs x %d rollout steps + 5 steps bit-exact, %d envs done\n", N, T, count);
    return 0;
}
