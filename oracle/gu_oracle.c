/* gu_oracle.c -- batched CPU restatement of the reference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Plain C, scalar, one env
 * after the other; used by tests/ as the bit-exact checker at sizes where the
 * per-instance Python restatement (oracle/ref_env.py) would take minutes, and by
 * bench.py's cpu_baseline leg.  The product never links or loads this file.
 *
 * Restates TheMTank/GridUniverse (file:line into the reference):
 *   core/envs/griduniverse_env.py:51-54    boundary-checked moves (UP,RIGHT,DOWN,LEFT)
 *   core/envs/griduniverse_env.py:136-155  look_step_ahead
 *   core/envs/griduniverse_env.py:157-174  _is_wall / is_terminal / is_lava / is_terminal_goal
 *   core/envs/griduniverse_env.py:176-193  _step / _reset
 *   core/algorithms/utils.py:15-27         single_step_policy_evaluation (V1)
 *   core/algorithms/utils.py:55-72         greedy_policy_from_value_function (V2)
 * and the build-defined counter RNG specified in oracle/gu_rng.py.
 *
 * The grid is described exactly the way the reference holds it: per-cell wall
 * flags (wall_grid), membership flags for lava_states / goal_states, the int
 * reward_matrix, and the starting_states list.  Nothing is pre-compiled into
 * transition tables here, so that a mistake in the product's own cell-record
 * compiler cannot be mirrored by the checker.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

typedef struct {
    int32_t W, H;             /* x_max, y_max */
    const uint8_t *wall;      /* [S] wall_grid[s] == 1 */
    const uint8_t *lava;      /* [S] s in lava_states  */
    const uint8_t *goal;      /* [S] s in goal_states  */
    const int32_t *reward;    /* [S] reward_matrix     */
    const int32_t *starts;    /* starting_states       */
    int32_t n_starts;
} gu_oracle_grid;

/* ---------------------------------------------------------------- RNG (oracle/gu_rng.py) */
static uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static uint32_t mm3_block(uint32_t h, uint32_t k)
{
    k *= 0xCC9E2D51u; k = rotl32(k, 15); k *= 0x1B873593u;
    h ^= k; h = rotl32(h, 13);
    return h * 5u + 0xE6546B64u;
}

/* epoch: step count >> 32 of streams 0 and 2 -- hashed right behind the seed when it is not zero; the length word stays 16 */
static uint32_t rng_word_epoch(uint64_t seed, uint32_t epoch, uint32_t env, uint32_t stream, uint32_t ctr)
{
    uint32_t h = 0x9747B28Cu;
    h = mm3_block(h, (uint32_t)seed);
    h = mm3_block(h, (uint32_t)(seed >> 32));
    if (epoch) h = mm3_block(h, epoch);
    h = mm3_block(h, env);
    h = mm3_block(h, ((stream & 0xFu) << 28) | (ctr & 0x0FFFFFFFu));
    if (ctr >> 28) {   /* counters of 2^28 and more: fifth key word, hashed length 20 (oracle/gu_rng.py) */
        h = mm3_block(h, ctr >> 28);
        h ^= 20u;
    } else {
        h ^= 16u;
    }
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

uint32_t gu_oracle_rng_word(uint64_t seed, uint32_t env, uint32_t stream, uint32_t ctr)
{
    return rng_word_epoch(seed, 0u, env, stream, ctr);
}

/* the word of stream 0 or 2 that covers step t (a 64-bit step count): counter = bits 4 .. 31 of t, epoch = t >> 32 */
static uint32_t rng_word_at_step(uint64_t seed, uint32_t env, uint32_t stream, uint64_t t)
{
    return rng_word_epoch(seed, (uint32_t)(t >> 32), env, stream, (uint32_t)(t >> 4) & 0x0FFFFFFFu);
}

int32_t gu_oracle_rng_action(uint64_t seed, uint32_t env, uint64_t t)
{
    return (int32_t)((rng_word_at_step(seed, env, 0, t) >> (2 * (t & 15))) & 3u);
}

int32_t gu_oracle_rng_start(uint64_t seed, uint32_t env, uint32_t episode, int32_t n_starts)
{
    return (int32_t)(((uint64_t)gu_oracle_rng_word(seed, env, 1, episode) * (uint64_t)(uint32_t)n_starts) >> 32);
}

/* ---------------------------------------------------------------- transition */
static int is_terminal(const gu_oracle_grid *g, int32_t s) { return g->lava[s] || g->goal[s]; }

static int32_t move(const gu_oracle_grid *g, int32_t s, int32_t a)
{
    int32_t x = s % g->W, y = s / g->W;      /* world[s] = (x, y), env:116-117 */
    switch (a) {
    case 0: return y > 0 ? s - g->W : s;            /* UP    env:51 */
    case 1: return x < g->W - 1 ? s + 1 : s;        /* RIGHT env:52 */
    case 2: return y < g->H - 1 ? s + g->W : s;     /* DOWN  env:53 */
    default: return x > 0 ? s - 1 : s;              /* LEFT  env:54 */
    }
}

/* env:136-155 */
void gu_oracle_look_step_ahead(const gu_oracle_grid *g, int32_t s, int32_t a, int32_t care_about_terminal,
                               int32_t *next, int32_t *reward, int32_t *done)
{
    int32_t n;
    if (a < 0) a += 4;   /* env:148 indexes a Python list of four moves: -1 is LEFT ... -4 is UP */
    if (care_about_terminal && is_terminal(g, s)) {
        n = s;
    } else {
        int32_t cand = move(g, s, a);
        n = g->wall[cand] ? s : cand;
    }
    *next = n;
    *reward = g->reward[n];
    *done = is_terminal(g, n);
}

void gu_oracle_look_step_ahead_batch(const gu_oracle_grid *g, int64_t n, const int32_t *states,
                                     const int32_t *actions, int32_t care_about_terminal,
                                     int32_t *next, int32_t *reward, int32_t *done)
{
    for (int64_t i = 0; i < n; ++i)
        gu_oracle_look_step_ahead(g, states[i], actions[i], care_about_terminal, &next[i], &reward[i], &done[i]);
}

/* _reset (env:187-193) with the start chosen by RNG stream 1; bumps the episode counter */
static int32_t do_reset(const gu_oracle_grid *g, uint64_t seed, uint32_t env, uint32_t *episode)
{
    int32_t s = g->starts[gu_oracle_rng_start(seed, env, *episode, g->n_starts)];
    *episode += 1;
    return s;
}

void gu_oracle_reset(const gu_oracle_grid *g, uint64_t seed, int64_t env_id0, int64_t n, const uint8_t *mask,
                     int32_t *pos, int32_t *done, uint32_t *episode)
{
    for (int64_t i = 0; i < n; ++i) {
        if (mask && !mask[i]) continue;
        pos[i] = do_reset(g, seed, (uint32_t)(env_id0 + i), &episode[i]);
        done[i] = 0;
    }
}

/* the word behind the sampled action of step t (oracle/gu_rng.py: sample_word): one hashed word of stream 2 per sixteen steps,
 * the fifteen behind it by xorshift32 + a Weyl increment */
uint32_t gu_oracle_rng_sample_word(uint64_t seed, uint32_t env, uint64_t t)
{
    uint32_t w = rng_word_at_step(seed, env, 2, t);
    for (uint32_t i = 0; i < (t & 15u); ++i) {
        w ^= w << 13;
        w ^= w >> 17;
        w ^= w << 5;
        w += 0x9E3779B9u;
    }
    return w;
}

/* inverse-CDF sample of pi[s][0..3] on RNG stream 2 (oracle/gu_rng.py: sampled_action) */
int32_t gu_oracle_rng_sample(uint64_t seed, uint32_t env, uint64_t t, const double *p)
{
    double u = (double)gu_oracle_rng_sample_word(seed, env, t) / 4294967296.0;
    volatile double c0 = p[0];
    volatile double c1 = c0 + p[1];
    volatile double c2 = c1 + p[2];
    return (u >= c0) + (u >= c1) + (u >= c2);
}

/* T lock-stepped env-steps for n envs (global ids env_id0 ..), state in/out.
 *   pi           : [S][4] or NULL; used when actions == NULL: a ~ pi[s] (gu_oracle_rng_sample)
 *   actions      : [T][n] int32 or NULL -> uniform actions from RNG stream 0 at step counter tcount[i]
 *   auto_reset   : the harness's `if done: reset()` applied lazily, i.e. right before the next step
 *   obs/reward/done_out : [T][n] or NULL
 *   ret_out/len... : per-env sum of rewards over the T steps and number of finished episodes, or NULL */
void gu_oracle_rollout(const gu_oracle_grid *g, uint64_t seed, int64_t env_id0, int64_t n, int64_t T,
                       int32_t auto_reset, const int32_t *actions, const double *pi,
                       int32_t *pos, int32_t *done, uint32_t *episode, uint64_t *tcount,
                       int32_t *obs_out, int32_t *reward_out, int32_t *done_out,
                       int64_t *ret_out, int32_t *episodes_out)
{
    for (int64_t i = 0; i < n; ++i) {
        uint32_t env = (uint32_t)(env_id0 + i);
        int32_t s = pos[i], d = done[i];
        int64_t ret = 0;
        int32_t fin = 0;
        for (int64_t t = 0; t < T; ++t) {
            if (auto_reset && d) { s = do_reset(g, seed, env, &episode[i]); d = 0; }
            int32_t a = actions ? actions[t * n + i]
                      : pi ? gu_oracle_rng_sample(seed, env, tcount[i], pi + 4 * (int64_t)s)
                           : gu_oracle_rng_action(seed, env, tcount[i]);
            int32_t r;
            gu_oracle_look_step_ahead(g, s, a, 1, &s, &r, &d);   /* _step, env:180-181 */
            tcount[i] += 1;
            ret += r;
            fin += d;
            if (obs_out) obs_out[t * n + i] = s;
            if (reward_out) reward_out[t * n + i] = r;
            if (done_out) done_out[t * n + i] = d;
        }
        pos[i] = s; done[i] = d;
        if (ret_out) ret_out[i] = ret;
        if (episodes_out) episodes_out[i] = fin;
    }
}

/* ---------------------------------------------------------------- tabular DP (float64, no contraction) */
/* utils.py:15-27: v'[s] = ((0.0 + R[s]) + pi[s,0]*(g*v[n0])) + ... in action order */
void gu_oracle_policy_evaluation_sweep(const gu_oracle_grid *g, double gamma, const double *pi,
                                       const double *v, double *v_new)
{
    int32_t S = g->W * g->H;
    for (int32_t s = 0; s < S; ++s) {
        volatile double acc = 0.0;
        acc = acc + (double)g->reward[s];
        for (int32_t a = 0; a < 4; ++a) {
            int32_t n, r, d;
            gu_oracle_look_step_ahead(g, s, a, 1, &n, &r, &d);
            volatile double gv = gamma * v[n];
            volatile double term = pi[4 * s + a] * gv;
            acc = acc + term;
        }
        v_new[s] = acc;
    }
}

static double around8(double x)
{   /* numpy.around(x, 8) == rint(x * 1e8) / 1e8 for float64 */
    volatile double m = x * 100000000.0;
    volatile double r = rint(m);
    return r / 100000000.0;
}

/* utils.py:55-72: q = 0.0 + (R[n] + g*v[n]); ties after around(.,8); terminal rows all zero */
void gu_oracle_greedy_policy(const gu_oracle_grid *g, double gamma, const double *v, double *pi)
{
    int32_t S = g->W * g->H;
    for (int32_t s = 0; s < S; ++s) {
        double q[4], qmax;
        for (int32_t a = 0; a < 4; ++a) {
            int32_t n, r, d;
            gu_oracle_look_step_ahead(g, s, a, 1, &n, &r, &d);
            volatile double gv = gamma * v[n];
            volatile double rq = (double)r + gv;
            q[a] = 0.0 + rq;
        }
        qmax = q[0];
        for (int32_t a = 1; a < 4; ++a) if (q[a] > qmax) qmax = q[a];   /* np.amax (no NaNs arise here) */
        double rmax = around8(qmax);
        int32_t ties = 0, tie[4];
        for (int32_t a = 0; a < 4; ++a) { tie[a] = around8(q[a]) == rmax; ties += tie[a]; }
        int term = is_terminal(g, s);
        for (int32_t a = 0; a < 4; ++a)
            pi[4 * s + a] = (tie[a] && !term) ? 1.0 / (double)ties : 0.0;
    }
}

/* dynamic_programming.py:15-20, one iteration: V1, signed delta = max(v - v'), V2 (pi updated in place) */
double gu_oracle_value_iteration_step(const gu_oracle_grid *g, double gamma, double *pi,
                                      const double *v, double *v_new)
{
    int32_t S = g->W * g->H;
    gu_oracle_policy_evaluation_sweep(g, gamma, pi, v, v_new);
    double delta = v[0] - v_new[0];
    for (int32_t s = 1; s < S; ++s) { double d = v[s] - v_new[s]; if (d > delta) delta = d; }
    gu_oracle_greedy_policy(g, gamma, v_new, pi);
    return delta;
}

/* ---------------------------------------------------------------- on-device maze generator, restated
 * (csrc/gu_maze.hip; algorithm of core/envs/maze_generation.py:41-149 with the build's RNG stream 3:
 * draw k = mulhi(word(maze_seed, grid_id, 3, k), n)).  wall_out[S]: 1 = wall.  Returns the number of open cells. */
static uint32_t mulhi32(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

int32_t gu_oracle_generate_maze(uint64_t maze_seed, uint32_t grid_id, int32_t W, int32_t H, uint8_t *wall_out,
                                int32_t *start_out, int32_t *goal_out, int32_t *stack /* >= rooms */)
{
    int32_t S = W * H, sp = 0, moved = 0;
    uint32_t k = 0;
    for (int32_t s = 0; s < S; ++s) wall_out[s] = 1;
    int32_t x = (int32_t)mulhi32(gu_oracle_rng_word(maze_seed, grid_id, 3, k++), (uint32_t)W);
    int32_t y = (int32_t)mulhi32(gu_oracle_rng_word(maze_seed, grid_id, 3, k++), (uint32_t)H);
    int32_t origin = y * W + x;
    uint8_t *visited = wall_out + 0;  /* a room is visited iff carved, except the origin before the first move */
    (void)visited;
    for (;;) {
        int32_t opt[4], n = 0, cur = y * W + x;
        /* neighbour order +x, -x, +y, -y (maze_generation.py:61-68) */
        if (x + 2 < W && wall_out[cur + 2] && cur + 2 != origin) opt[n++] = cur + 2;
        if (x - 2 >= 0 && wall_out[cur - 2] && cur - 2 != origin) opt[n++] = cur - 2;
        if (y + 2 < H && wall_out[cur + 2 * W] && cur + 2 * W != origin) opt[n++] = cur + 2 * W;
        if (y - 2 >= 0 && wall_out[cur - 2 * W] && cur - 2 * W != origin) opt[n++] = cur - 2 * W;
        if (n > 0) {
            int32_t nb = opt[mulhi32(gu_oracle_rng_word(maze_seed, grid_id, 3, k++), (uint32_t)n)];
            stack[sp++] = cur;
            wall_out[(cur + nb) / 2] = 0;   /* :77-82 carve the wall between, the neighbour and the current cell */
            wall_out[nb] = 0;
            wall_out[cur] = 0;
            x = nb % W; y = nb / W;
            moved = 1;
        } else if (sp > 0) {
            cur = stack[--sp];
            x = cur % W; y = cur / W;
        } else {
            break;
        }
    }
    int32_t n_open = 0;
    for (int32_t s = 0; s < S; ++s) n_open += wall_out[s] == 0;
    if (!moved || n_open < 2) return n_open;
    /* two distinct open cells, by rank among the open cells in ascending order (:124-142) */
    uint32_t i = mulhi32(gu_oracle_rng_word(maze_seed, grid_id, 3, k++), (uint32_t)n_open);
    uint32_t j = mulhi32(gu_oracle_rng_word(maze_seed, grid_id, 3, k++), (uint32_t)(n_open - 1));
    if (j >= i) ++j;
    uint32_t seen = 0;
    for (int32_t s = 0; s < S; ++s) {
        if (wall_out[s]) continue;
        if (seen == i) *start_out = s;
        if (seen == j) *goal_out = s;
        ++seen;
    }
    return n_open;
}
