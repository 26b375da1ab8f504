// ABI version + process-wide tuning knobs of libunsloth_amd.so
#include <stdlib.h>

#include "common.h"

extern "C" int uamd_version(void) { return (0 << 16) | 2; }

namespace {
int g_knob[UAMD_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
// environment names only for the knobs whose choice still depends on the workload (attention forward kernel, fused-activation
// schedule, GEMM kernel family); the others' A/Bs are settled -- they remain `uamd_set_tuning` hooks for the parity tests
const char* const kEnv[UAMD_TUNE_COUNT] = {nullptr, nullptr, nullptr, nullptr, "UAMD_ATTN_VAR", nullptr, nullptr, nullptr, nullptr, nullptr,
                                            "UAMD_GLU_XA", "UAMD_GEMM_S"};
const int kDefault[UAMD_TUNE_COUNT] = {2, 8, 0, 1, 0, 1, 1, 1, 1, 1, 3, 1};
}  // namespace

// value of a knob: uamd_set_tuning() > environment variable > built-in default (the measured-fastest setting)
int uamd_tuning_get(int knob) {
    if (knob < 0 || knob >= UAMD_TUNE_COUNT) return 0;
    if (g_knob[knob] < 0) {
        const char* e = kEnv[knob] ? getenv(kEnv[knob]) : nullptr;
        g_knob[knob] = (e && *e) ? atoi(e) : kDefault[knob];
        if (g_knob[knob] < 0) g_knob[knob] = kDefault[knob];
    }
    return g_knob[knob];
}

extern "C" int uamd_set_tuning(int knob, int value) {
    if (knob < 0 || knob >= UAMD_TUNE_COUNT || value < 0) return UAMD_ERR_ARG;
    g_knob[knob] = value;
    return UAMD_OK;
}
