// libgte_hip.so: error reporting, version and device facts.
#include "gte_common.h"

#include <string.h>

namespace gte {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

const DeviceProps& device_props() {
    // read-only cache, filled once (C++11 magic static: thread-safe)
    static const DeviceProps props = [] {
        DeviceProps p;
        p.cus = 256;
        p.lds_bytes = 160 * 1024;
        strncpy(p.arch, "unknown", sizeof(p.arch));
        int dev = 0;
        hipDeviceProp_t hp;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&hp, dev) == hipSuccess) {
            p.cus = hp.multiProcessorCount;
            p.lds_bytes = (int)hp.maxSharedMemoryPerMultiProcessor;
            strncpy(p.arch, hp.gcnArchName, sizeof(p.arch) - 1);
            p.arch[sizeof(p.arch) - 1] = 0;
        }
        return p;
    }();
    return props;
}

}  // namespace gte

extern "C" int gte_version(void) { return GTE_VERSION; }

extern "C" const char* gte_last_error(void) { return gte::error_buffer(); }

extern "C" int gte_device_info(int* compute_units, int* wave_size, int* lds_bytes, char* arch_name,
                               int arch_name_len) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return gte::fail(GTE_ERR_UNSUPPORTED, "gte_device_info: no HIP device visible");
    const gte::DeviceProps& p = gte::device_props();
    if (compute_units) *compute_units = p.cus;
    if (wave_size) *wave_size = gte::kWave;
    if (lds_bytes) *lds_bytes = p.lds_bytes;
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, p.arch, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return GTE_OK;
}
