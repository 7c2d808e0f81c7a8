// Library-level entry points: version + last-error text.
#include "bmc_common.h"

static thread_local char g_err[512] = "";

void bmc_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int bmc_version(void) { return 100; }
extern "C" const char* bmc_last_error(void) { return g_err; }

// A stream of the device's LOWEST priority (include/bmc_hip.h): the weight-gradient stream of the training step.
extern "C" int bmc_stream_create_low_priority(bmc_stream_t* out) {
    BMC_CHECK_ARG(out, "bmc_stream_create_low_priority: null argument");
    int least = 0, greatest = 0;
    hipStream_t s = nullptr;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess ||
        hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least) != hipSuccess) {
        bmc_set_error("bmc_stream_create_low_priority: %s", hipGetErrorString(hipGetLastError()));
        return 1;
    }
    *out = (bmc_stream_t)s;
    return 0;
}

int bmc_num_cus(void) {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}
