#include "common.h"
#include "../../include/maggie_hip.h"
extern "C" int mg_abi_version(void) { return 1; }
