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

// ---- id range check ---------------------------------------------------------------------------------------------
namespace mkb {
__global__ __launch_bounds__(256) void check_ids_kernel(const int64_t *__restrict__ sample, int64_t B, const int64_t *__restrict__ cand,
                                                        int64_t n_cand, int64_t n_entity, int64_t n_relation, int32_t *__restrict__ flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < 3 * B + n_cand; i += (int64_t)gridDim.x * 256) {
        if (i < 3 * B) {
            const int64_t v = sample[i];
            const bool rel = i % 3 == 1;
            if (v < 0 || v >= (rel ? n_relation : n_entity)) bad |= rel ? 2 : 1;
        } else {
            const int64_t v = cand[i - 3 * B];
            if (v < 0 || v >= n_entity) bad |= 4;
        }
    }
    if (bad) atomicOr(flag, bad);
}
}  // namespace mkb

extern "C" int mkb_check_ids(const int64_t *sample, int64_t B, const int64_t *cand, int64_t n_cand, int64_t n_entity,
                             int64_t n_relation, int32_t *flag, void *stream) {
    MKB_REQUIRE(flag && (sample || B == 0) && (cand || n_cand == 0) && B >= 0 && n_cand >= 0, "bad arguments");
    const int64_t n = 3 * B + n_cand;
    if (n == 0) return MKB_OK;
    const unsigned blocks = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
    hipLaunchKernelGGL(mkb::check_ids_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sample, B, cand, n_cand, n_entity,
                       n_relation, flag);
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

// ---- measurement aids -------------------------------------------------------------------------------------------
// shader clock right now: shader-clock cycles (s_memtime) per tick of the constant 100 MHz counter (s_memrealtime) over a short spin
namespace mkb {
__global__ void sclk_kernel(float *out_mhz, int spin_ticks) {
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < (unsigned long long)spin_ticks) { __builtin_amdgcn_s_sleep(8); r1 = __builtin_amdgcn_s_memrealtime(); }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) *out_mhz = (float)((double)(c1 - c0) / (double)(r1 - r0) * 100.0);
}
}  // namespace mkb

extern "C" int mkb_debug_sclk_mhz(float *out_mhz_device, void *stream) {
    MKB_REQUIRE(out_mhz_device != nullptr, "null pointer");
    hipLaunchKernelGGL(mkb::sclk_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_mhz_device, 2000);  // ~20 us
    MKB_LAUNCH_CHECK();
    return MKB_OK;
}

extern "C" int mkb_abi_version(void) { return MKB_ABI_VERSION; }
extern "C" const char *mkb_last_error(void) { return mkb::g_err; }
