// Host-side helpers shared by every source of the library: the error slot behind grip_last_error() and the argument-check macro.
// No HIP here, so the host-only sources (leaderboard.cpp, bpe.cpp) also build with plain g++ -- `make sanitize` does that under
// ThreadSanitizer and AddressSanitizer + UBSan (GPU sanitizers are not available on this pool; the device code has its own parity tests).
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/grip_amd.h"

void grip_set_error(const char* fmt, ...);

#define GRIP_REQUIRE(cond, ...)             \
    do {                                    \
        if (!(cond)) {                      \
            grip_set_error(__VA_ARGS__);    \
            return GRIP_ERR_ARG;            \
        }                                   \
    } while (0)
