// Host-side helpers shared by the C-ABI entry points of libmkb_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/mkb_hip.h"

namespace mkb {

int set_error(int code, const char *fmt, ...);

#define MKB_CHECK_HIP(expr)                                                                     \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return ::mkb::set_error(MKB_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                                    __FILE__, __LINE__);                                        \
    } while (0)

#define MKB_REQUIRE(cond, ...)                                          \
    do {                                                                \
        if (!(cond)) return ::mkb::set_error(MKB_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define MKB_LAUNCH_CHECK() MKB_CHECK_HIP(hipGetLastError())

inline int validate_tables(const mkb_tables_t *tb) {
    MKB_REQUIRE(tb != nullptr, "tables is null");
    MKB_REQUIRE(tb->model >= MKB_TRANSE && tb->model <= MKB_PROTATE, "unknown model id %d", tb->model);
    MKB_REQUIRE(tb->ent && tb->rel, "null embedding table");
    MKB_REQUIRE(tb->n_entity > 0 && tb->n_relation > 0 && tb->hidden_dim > 0, "empty table");
    const int64_t d = tb->hidden_dim;
    int64_t de = d, dr = d;
    if (tb->model == MKB_ROTATE) de = 2 * d;
    if (tb->model == MKB_COMPLEX) { de = 2 * d; dr = 2 * d; }
    MKB_REQUIRE(tb->entity_dim == de && tb->relation_dim == dr,
                "entity_dim/relation_dim (%lld, %lld) do not match model %d with hidden_dim %lld",
                (long long)tb->entity_dim, (long long)tb->relation_dim, tb->model, (long long)d);
    MKB_REQUIRE(tb->model != MKB_PROTATE || tb->modulus != nullptr, "pRotatE needs the modulus scalar");
    return MKB_OK;
}

inline bool mode_is_head(int mode) { return mode == MKB_MODE_HEAD; }

// Brackets a kernel launch with hipEvents on the launch stream when profiling of `kind` is enabled
// (mkb_profile_enable).  Usage:  { ProfScope ps(MKB_PROF_ADAM, st); hipLaunchKernelGGL(...); }
struct ProfScope {
    int kind;
    hipStream_t st;
    int slot;
    ProfScope(int kind, hipStream_t st);
    ~ProfScope();
};

}  // namespace mkb
