// ABI version + build info of libunsloth_amd.so
#include "common.h"

extern "C" int uamd_version(void) { return (0 << 16) | 1; }
