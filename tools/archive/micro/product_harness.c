/* product_harness.c -- the product's rollout launch driven from plain C in a tight loop (tuning aid): is the launch time the
 * Python tools report a property of the kernel or of the way they drive it?  Engines of 65 536 envs on a 32 x 32 grid (one start
 * cell), each with the first trajectory allocation it gets; per engine and rate-limiter setting (option rollout_pace) 2 + 30
 * launches between gu_timer_begin / gu_timer_end.
 *   gcc -std=c11 -O2 -Iinclude tools/archive/micro/product_harness.c -o tools/archive/micro/product_harness -Lgriduniverse_amd/lib -lgu<variant> \
 *       -Wl,-rpath,$PWD/griduniverse_amd/lib && tools/archive/micro/product_harness [engines] [T] [pace values ...] */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "gu.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != GU_OK) { char msg[512]; gu_last_error(msg, sizeof msg); fprintf(stderr, "%s -> %d: %s\n", #call, rc_, msg); return 1; } } while (0)

enum { W = 32, H = 32, N = 65536 };

int main(int argc, char **argv)
{
    const int engines = argc > 1 ? atoi(argv[1]) : 4;
    const int T = argc > 2 ? atoi(argv[2]) : 1000;
    uint32_t wall_rows[H] = {0}, goal_rows[H] = {0}, lava_rows[H] = {0};
    goal_rows[H - 1] |= 1u << (W - 1);
    for (int r = 4; r < 28; ++r) lava_rows[r] |= 1u << 16;
    wall_rows[1] |= 1u << 8; wall_rows[2] |= 1u << 11;
    const int32_t starts[1] = {0};
    CHECK(gu_set_option(NULL, GU_OPT_TRAJ_CANDIDATES, 1));
    gu_handle h[16];
    for (int b = 0; b < engines && b < 16; ++b) {
        CHECK(gu_create(0, N, 0, &h[b]));
        CHECK(gu_set_grid(h[b], W, H, 1, wall_rows, goal_rows, lava_rows, NULL, NULL, starts, 1));
        CHECK(gu_seed(h[b], 123));
        CHECK(gu_reset(h[b], NULL, NULL, NULL));
        CHECK(gu_reserve_trajectory(h[b], T));
    }
    printf("%-4s %9s", "eng", "unpaced");
    for (int k = 3; k < argc; ++k) printf(" %8s", argv[k]);
    printf("\n");
    for (int b = 0; b < engines && b < 16; ++b) {
        printf("%-4d", b);
        for (int k = 2; k < argc || k == 2; ++k) {
            const long pace = k == 2 ? 0 : atol(argv[k]);
            CHECK(gu_set_option(h[b], GU_OPT_ROLLOUT_PACE, pace));
            for (int i = 0; i < 2; ++i) CHECK(gu_rollout(h[b], T, GU_POLICY_UNIFORM, GU_F_AUTO_RESET | GU_F_TRAJECTORY));
            CHECK(gu_sync(h[b]));
            CHECK(gu_timer_begin(h[b]));
            for (int i = 0; i < 30; ++i) CHECK(gu_rollout(h[b], T, GU_POLICY_UNIFORM, GU_F_AUTO_RESET | GU_F_TRAJECTORY));
            float ms = 0;
            CHECK(gu_timer_end(h[b], &ms));
            printf(" %8.1f", ms / 30 * 1e3);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
