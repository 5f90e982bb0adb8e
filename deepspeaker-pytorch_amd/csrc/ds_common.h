// ds_common.h -- host-side helpers shared by the C-ABI translation units.
#pragma once
#include <stdint.h>
#include "../../include/deepspeaker_hip.h"

#define DS_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
#define DS_ALIGNED16(p) ((((uintptr_t)(p)) & 15u) == 0)

static inline int ds_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline long long ds_ceil_div_ll(long long a, long long b) { return (a + b - 1) / b; }
