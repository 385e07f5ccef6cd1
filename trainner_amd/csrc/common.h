// Shared device/host helpers for libtrainner_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/trainner_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// K-chunk (input channels staged per LDS fill) and the padded LDS pixel stride in dwords.
// 20 dwords: ds_read_b128 of consecutive pixels lands on distinct 16-B slots (5*p mod 16 is a
// permutation), see DESIGN.md "LDS layout".
constexpr int TNR_CK = 16;
constexpr int TNR_PST = TNR_CK + 4;

void tnr_set_error(const char *fmt, ...);
int tnr_check_launch(const char *what);
unsigned *tnr_fault_word_or(unsigned *fallback);      // the registered fault latch (tnr_set_fault_word), else `fallback`

#define TNR_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            tnr_set_error(__VA_ARGS__);   \
            return TNR_EINVAL;            \
        }                                 \
    } while (0)

static inline int tnr_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t tnr_cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int tnr_round_up(int a, int b) { return tnr_cdiv(a, b) * b; }

__device__ __forceinline__ float tnr_act(float v, int act, float slope) {
    if (act == TNR_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == TNR_ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}
