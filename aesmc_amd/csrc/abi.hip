// Library identification entry points of the C ABI (include/aesmc_hip.h).
#include "common.hpp"

extern "C" int aesmc_version(void) { return 100; /* 0.1.0 */ }
extern "C" const char *aesmc_target_arch(void) { return "gfx950"; }
