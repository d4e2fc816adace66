// Library-level entry points of libpylc_hip.so: error text, ABI version, one-time kernel attribute setup.
#include "common.h"

namespace pylc {
thread_local char g_err[512] = "";
int conv_init();
}  // namespace pylc

extern "C" const char* pylc_last_error(void) { return pylc::g_err; }

extern "C" int pylc_abi_version(void) { return 5; }

extern "C" int pylc_init(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return pylc::fail(PYLC_ERR_HIP, "pylc_init: no HIP device visible");
    return pylc::conv_init();
}
