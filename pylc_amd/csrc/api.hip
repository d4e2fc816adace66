// Library-level entry points of libpylc_hip.so: error text, ABI version, one-time kernel attribute setup.
#include "common.h"

namespace pylc {
thread_local char g_err[512] = "";
int conv_init();
}  // namespace pylc

extern "C" const char* pylc_last_error(void) { return pylc::g_err; }

extern "C" int pylc_abi_version(void) { return 13; }      // r5: pylc_comm_*, PylcFwdEp, PylcConvDesc.w_planes_fmt, pylc_weight_prepare(interleave); 12: pylc_conv2d_wgrad_slabs, pylc_splitk_reduce_batch, PylcSlabSum; 13 (r6): pylc_comm_available

// 1 when the library was built with EXPERIMENTAL=1 (include/pylc_hip.h: the #ifdef PYLC_EXPERIMENTAL entry points exist)
extern "C" int pylc_experimental_build(void) {
#ifdef PYLC_EXPERIMENTAL
    return 1;
#else
    return 0;
#endif
}

extern "C" int pylc_init(void) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        return pylc::fail(PYLC_ERR_HIP, "pylc_init: no HIP device visible");
    return pylc::conv_init();
}

#ifdef PYLC_EXPERIMENTAL
extern "C" int pylc_stream_create_cu_mask(int n_cus, int from_top, void** stream_out) {
    PYLC_REQUIRE(stream_out != nullptr && n_cus >= 8 && n_cus <= pylc::kNumCU && n_cus % 8 == 0,
                 "stream_create_cu_mask: n_cus must be a multiple of 8 in [8, %d]", pylc::kNumCU);
    uint32_t mask[pylc::kNumCU / 32] = {};
    for (int i = 0; i < n_cus; ++i) {
        const int bit = from_top ? pylc::kNumCU - 1 - i : i;
        mask[bit / 32] |= 1u << (bit % 32);
    }
    hipStream_t st = nullptr;
    PYLC_HIP(hipExtStreamCreateWithCUMask(&st, pylc::kNumCU / 32, mask));
    *stream_out = st;
    return PYLC_OK;
}

extern "C" int pylc_stream_destroy(void* stream) {
    PYLC_REQUIRE(stream != nullptr, "stream_destroy: null stream");
    PYLC_HIP(hipStreamDestroy(pylc::as_stream(stream)));
    return PYLC_OK;
}
#endif
