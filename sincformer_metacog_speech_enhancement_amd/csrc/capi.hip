#include "sfm_common.h"
extern "C" int sfm_abi_version(void) { return 1; }
