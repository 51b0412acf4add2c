// Library-wide pieces of libmkb_hip.so: ABI version and the thread-local error message.
#include "common.h"

namespace mkb {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// ---- per-kernel event timing --------------------------------------------------------------------------
constexpr int kProfSlots = 8192;
struct ProfState {
    bool on = false;
    int every = 1;   // bracket every `every`-th launch (two event records cost ~6 us of stream time each)
    int seen = 0;
    int n = 0;
    hipEvent_t *start = nullptr, *stop = nullptr;
};
static ProfState g_prof[MKB_PROF_KINDS];

ProfScope::ProfScope(int kind_, hipStream_t st_) : kind(kind_), st(st_), slot(-1) {
    ProfState &p = g_prof[kind];
    if (!p.on || p.n >= kProfSlots) return;
    if (p.seen++ % p.every != 0) return;
    if (!p.start) {
        p.start = new hipEvent_t[kProfSlots]();
        p.stop = new hipEvent_t[kProfSlots]();
    }
    slot = p.n++;
    if (!p.start[slot]) {
        (void)hipEventCreate(&p.start[slot]);
        (void)hipEventCreate(&p.stop[slot]);
    }
    (void)hipEventRecord(p.start[slot], st);
}

ProfScope::~ProfScope() {
    if (slot >= 0) (void)hipEventRecord(g_prof[kind].stop[slot], st);
}

}  // namespace mkb

extern "C" int mkb_profile_enable(int kernel, int on) {
    MKB_REQUIRE(kernel >= 0 && kernel < MKB_PROF_KINDS, "bad kernel kind");
    mkb::g_prof[kernel].on = on != 0;
    mkb::g_prof[kernel].every = on > 1 ? on : 1;
    mkb::g_prof[kernel].seen = 0;
    return MKB_OK;
}

extern "C" int mkb_profile_read(int kernel, int64_t *launches, double *total_ms) {
    MKB_REQUIRE(kernel >= 0 && kernel < MKB_PROF_KINDS && launches && total_ms, "bad arguments");
    mkb::ProfState &p = mkb::g_prof[kernel];
    double tot = 0.0;
    for (int i = 0; i < p.n; ++i) {
        MKB_CHECK_HIP(hipEventSynchronize(p.stop[i]));
        float ms = 0.f;
        MKB_CHECK_HIP(hipEventElapsedTime(&ms, p.start[i], p.stop[i]));
        tot += ms;
    }
    *launches = p.n;
    *total_ms = tot;
    p.n = 0;
    return MKB_OK;
}

extern "C" int mkb_abi_version(void) { return MKB_ABI_VERSION; }
extern "C" const char *mkb_last_error(void) { return mkb::g_err; }
