// gu_options.hip -- gu_set_option / gu_get_option, gu_device_info (include/gu.h "options").
//
// Every launch-shape switch of the library is a per-engine option with a process-wide default; the kernels' launchers ask
// gu_opt(h, GU_OPT_...) -- a few loads -- instead of the environment.  Only a -DGU_EXPERIMENTS build (libgu_exp.so, the A/B
// tools under tools/) still listens to the environment variables of round 1 / 2, on every call, and only such a build accepts
// the GU_OPT_X_* experiments; the product library refuses them.
#include "gu_internal.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

struct OptSpec {
    const char *name;       // (also the environment variable a GU_EXPERIMENTS build reads)
    int64_t builtin, lo, hi;
};

// built-in default, smallest and largest accepted value
const OptSpec g_spec[GU_OPT_COUNT] = {
    {nullptr, 0, 0, 0},
    {"GU_ROLLOUT_BLOCK", 256, 64, 1024},
    {"GU_ROLLOUT_ROWS", -1, -1, 3},
    {"GU_ROWS_COPIES", 0, 0, 32},
    {"GU_ROLLOUT_MULTI", -1, -1, 1},
    {"GU_ROLLOUT_MULTI_K", 0, 0, 4},
    {"GU_ROLLOUT_MULTI_COPIES", 1, 1, 2},
    {"GU_ROLLOUT_XCD", 0, 0, 1},
    {"GU_VI_PATH", 0, 0, 6},
    {"GU_MC_SCRATCH_MB", 2048, 1, 1 << 20},
    {"GU_MC_LANE_RETURNS", 0, 0, 1},
    {"GU_MC_GLOBAL_WALK", 0, 0, 1},
    {"GU_STEP_SYNC", 0, 0, 1},
    {"GU_TRAJ_CANDIDATES", 4, 1, 64},
    {"GU_TRAJ_FAR_CANDIDATES", 0, 0, 256},
    {"GU_TRAJ_STRIDE_MIB", 3072, 0, 1 << 20},
    {"GU_TRAJ_FAR_MIB", 49152, 0, 1 << 22},
    {"GU_TRAJ_PROBE_ALL", 0, 0, 1},
    {"GU_ROLLOUT_PACE", -1, -2, 0xFFFFF},
    {"GU_VI_XCD_BLOCK", 0, 0, 1024},
    {"GU_PACE_TARGET", 7200, 100, 100000},
    {"GU_PACE_BAR_NUM", 20, 1, 256},
    {"GU_PACE_GAIN_Q", 128, 1, 64 * 64},
    {"GU_PACE_DEC_Q", 8, 0, 64},
    {"GU_TRAJ_LAYOUT", -1, -1, 1},
    {"GU_PACE_RECORD", 1, 0, 1},
    {"GU_PACE_PROBE_EVERY", 1024, 0, 1 << 24},
    {"GU_PACE_ADAPT", 1, 0, 1},
    {"GU_ROLLOUT_HALF_WAVES", -1, -1, 1},
    {"GU_ROLLOUT_ENTRY", 1, 0, 1},
    {"GU_SYNC_SPIN_US", 5000, 0, 1000000},
};
const char *g_spec_x[GU_OPT_X_COUNT] = {"GU_TRAJ_UNCACHED", "GU_TRAJ_POISON", "GU_MC_POISON"};

std::atomic<int64_t> g_default[GU_OPT_COUNT];
std::atomic<int64_t> g_default_x[GU_OPT_X_COUNT];
std::atomic<bool> g_defaults_ready{false};

void defaults_init()
{
    if (g_defaults_ready.load(std::memory_order_acquire)) return;
    for (auto &d : g_default) d.store(GU_OPT_UNSET, std::memory_order_relaxed);
    for (auto &d : g_default_x) d.store(0, std::memory_order_relaxed);
    g_defaults_ready.store(true, std::memory_order_release);
}

bool block_size_ok(int64_t v) { return v == 64 || v == 128 || v == 256 || v == 512 || v == 1024; }
bool xcd_block_ok(int64_t v) { return v == 0 || v == 256 || v == 512 || v == 1024; }

}  // namespace

int64_t gu_opt(const gu_engine *h, int option)
{
    defaults_init();
    if (option >= GU_OPT_X_TRAJ_UNCACHED && option < GU_OPT_X_TRAJ_UNCACHED + GU_OPT_X_COUNT) {
#ifdef GU_EXPERIMENTS
        const int x = option - GU_OPT_X_TRAJ_UNCACHED;
        if (const char *s = std::getenv(g_spec_x[x])) return std::atoll(s);
        if (h && h->opt_x[x]) return h->opt_x[x];
        return g_default_x[x].load(std::memory_order_relaxed);
#else
        return 0;  // compiled out of the product library
#endif
    }
    if (option <= 0 || option >= GU_OPT_COUNT) return 0;
#ifdef GU_EXPERIMENTS
    // the A/B tools switch paths inside one process through the environment, as in rounds 1 and 2
    if (option == GU_OPT_VI_PATH) {
        if (std::getenv("GU_VI_MULTI_LAUNCH")) return 2;
        const char *c = std::getenv("GU_VI_CLUSTER");
        if (c && std::atoi(c) == 0) return 1;
    } else if (const char *s = std::getenv(g_spec[option].name)) {
        const int64_t v = std::atoll(s);
        if (v >= g_spec[option].lo && v <= g_spec[option].hi && (option != GU_OPT_ROLLOUT_BLOCK || block_size_ok(v)) && (option != GU_OPT_VI_XCD_BLOCK || xcd_block_ok(v))) return v;
    }
#endif
    if (h && h->opt[option] != GU_OPT_UNSET) return h->opt[option];
    const int64_t d = g_default[option].load(std::memory_order_relaxed);
    return d != GU_OPT_UNSET ? d : g_spec[option].builtin;
}

int gu_debug()
{
    static const int level = [] {
        const char *s = std::getenv("GU_DEBUG");
        return s ? std::atoi(s) : 0;
    }();
    return level;
}

extern "C" {

int gu_set_option(gu_handle h, int32_t option, int64_t value)
{
    defaults_init();
    if (option >= GU_OPT_X_TRAJ_UNCACHED && option < GU_OPT_X_TRAJ_UNCACHED + GU_OPT_X_COUNT) {
#ifdef GU_EXPERIMENTS
        const int x = option - GU_OPT_X_TRAJ_UNCACHED;
        const int64_t v = value == GU_OPT_UNSET ? 0 : (value != 0);
        if (h) h->opt_x[x] = v;
        else g_default_x[x].store(v, std::memory_order_relaxed);
        return GU_OK;
#else
        return gu_fail(GU_ERR_UNSUPPORTED, "option %d (%s) is an experiment that is compiled out of this library (build `make exp`)", option,
                       g_spec_x[option - GU_OPT_X_TRAJ_UNCACHED]);
#endif
    }
    GU_REQUIRE(option > 0 && option < GU_OPT_COUNT, GU_ERR_INVALID, "unknown option %d", option);
    const OptSpec &sp = g_spec[option];
    if (value != GU_OPT_UNSET) {
        GU_REQUIRE(value >= sp.lo && value <= sp.hi, GU_ERR_INVALID, "option %s: %lld outside %lld .. %lld", sp.name, (long long)value,
                   (long long)sp.lo, (long long)sp.hi);
        if (option == GU_OPT_VI_XCD_BLOCK) GU_REQUIRE(xcd_block_ok(value), GU_ERR_INVALID, "option %s: %lld is not 0, 256, 512 or 1024", sp.name, (long long)value);
        if (option == GU_OPT_ROLLOUT_BLOCK) GU_REQUIRE(block_size_ok(value), GU_ERR_INVALID, "option %s: %lld is not 64, 128, 256, 512 or 1024", sp.name, (long long)value);
        if (option == GU_OPT_ROWS_COPIES) GU_REQUIRE((value & (value - 1)) == 0, GU_ERR_INVALID, "option %s: %lld is not a power of two", sp.name, (long long)value);
        if (option == GU_OPT_ROLLOUT_MULTI_K) GU_REQUIRE(value == 0 || value == 2 || value == 4, GU_ERR_INVALID, "option %s: K is 2 or 4", sp.name);
    }
    if (h) {
        h->opt[option] = value;
        // a launch-shape option may change which kernel a launch kind runs on: the store pacing of every kind starts over
        if (option != GU_OPT_ROLLOUT_PACE && option != GU_OPT_SYNC_SPIN_US)
            for (gu_engine::PaceKind &k : h->pace) k.active = false;
    } else {
        g_default[option].store(value, std::memory_order_relaxed);
    }
    return GU_OK;
}

int gu_get_option(gu_handle h, int32_t option, int64_t *value)
{
    GU_REQUIRE(value != nullptr, GU_ERR_INVALID, "value is NULL");
    const bool x = option >= GU_OPT_X_TRAJ_UNCACHED && option < GU_OPT_X_TRAJ_UNCACHED + GU_OPT_X_COUNT;
    GU_REQUIRE(x || (option > 0 && option < GU_OPT_COUNT), GU_ERR_INVALID, "unknown option %d", option);
    *value = gu_opt(h, option);
    return GU_OK;
}

int gu_device_info(int device_id, char *buf, size_t len)
{
    GU_REQUIRE(buf != nullptr && len > 0, GU_ERR_INVALID, "buf is NULL or len == 0");
    hipDeviceProp_t p;
    memset(&p, 0, sizeof p);
    GU_HIP(hipGetDeviceProperties(&p, device_id));
    char pci[64] = "";
    if (hipDeviceGetPCIBusId(pci, (int)sizeof pci, device_id) != hipSuccess) {
        (void)hipGetLastError();
        pci[0] = 0;
    }
    size_t free_b = 0, total_b = 0;
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess && hipSetDevice(device_id) == hipSuccess) {
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
        (void)hipSetDevice(cur);
    }
    const int n = snprintf(buf, len, "name=%s;arch=%s;pci=%s;cus=%d;lds_per_cu=%zu;sclk_khz=%d;mclk_khz=%d;bus_bits=%d;l2_bytes=%d;hbm_bytes=%zu;hbm_free=%zu",
                           p.name, p.gcnArchName, pci, p.multiProcessorCount, (size_t)p.maxSharedMemoryPerMultiProcessor, p.clockRate,
                           p.memoryClockRate, p.memoryBusWidth, p.l2CacheSize, (size_t)p.totalGlobalMem, free_b);
    return n < 0 ? 0 : n;
}

}  // extern "C"
