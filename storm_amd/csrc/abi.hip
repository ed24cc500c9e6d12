// Error channel, version and device info of libstorm_hip.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "common.h"

namespace storm {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

void bump_switch_epoch();
namespace {
struct SwitchName { const char* name; int Switches::*field; };
const SwitchName kSwitches[] = {
    {"STORM_CONV_VARIANT", &Switches::conv_variant}, {"STORM_CONV_PIPE128", &Switches::conv_pipe128}, 
    {"STORM_CONV_CUS", &Switches::conv_cus}, {"STORM_CONV_PERSIST", &Switches::conv_persist},
    {"STORM_CONV_DMA", &Switches::conv_dma}, {"STORM_CONV_ABLATE", &Switches::conv_ablate}, {"STORM_SPLITK", &Switches::splitk}, {"STORM_GN_WIDE", &Switches::gn_wide}, {"STORM_GN_DOWN_SHARE", &Switches::gn_down_share}, {"STORM_GN_ROWS", &Switches::gn_rows}, {"STORM_GN_NT", &Switches::gn_nt}, {"STORM_GRAPH", &Switches::graph}, {"STORM_SPLITK_SMALL", &Switches::splitk_small}, {"STORM_CONV_TABLE", &Switches::conv_table}, {"STORM_ATTN_SPLIT", &Switches::attn_split}, {"STORM_BATCH_INVARIANT", &Switches::batch_invariant},
};
}  // namespace

Switches& switches() {
    static Switches sw = [] {
        Switches v;
        for (const SwitchName& n : kSwitches)
            if (const char* e = getenv(n.name)) v.*(n.field) = atoi(e);
        if (const char* e = getenv("STORM_CONV_TRACE_PTR")) v.conv_trace_ptr = strtoull(e, nullptr, 0);
        return v;
    }();
    return sw;
}

static std::atomic<unsigned long long> g_switch_epoch{0};
unsigned long long switch_epoch() { return g_switch_epoch.load(std::memory_order_relaxed); }
void bump_switch_epoch() { g_switch_epoch.fetch_add(1, std::memory_order_relaxed); }

int device_cus() {
    const int forced = switches().conv_cus;
    if (forced > 0) return forced;
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0; hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
        if (n_cu <= 0) n_cu = 256;
    }
    return n_cu;
}
}  // namespace storm

// Test / tool hook (not part of the drop-in surface): set one of the switches above by its environment-variable name.
extern "C" int storm_set_switch(const char* name, long long value) {
    STORM_CHECK(name != nullptr, "storm_set_switch: null name");
    if (strcmp(name, "STORM_CONV_TRACE_PTR") == 0) { storm::switches().conv_trace_ptr = (unsigned long long)value; return STORM_OK; }
    for (const storm::SwitchName& n : storm::kSwitches)
        if (strcmp(name, n.name) == 0) { storm::switches().*(n.field) = (int)value; storm::bump_switch_epoch(); return STORM_OK; }
    STORM_CHECK(false, "storm_set_switch: unknown switch %s", name);
}
extern "C" long long storm_get_switch(const char* name) {
    if (name == nullptr) return 0;
    if (strcmp(name, "STORM_CONV_TRACE_PTR") == 0) return (long long)storm::switches().conv_trace_ptr;
    for (const storm::SwitchName& n : storm::kSwitches)
        if (strcmp(name, n.name) == 0) return storm::switches().*(n.field);
    return 0;
}

extern "C" const char* storm_last_error(void) { return storm::g_err; }
extern "C" int storm_abi_version(void) { return STORM_ABI_VERSION; }
extern "C" long long storm_abi_struct_bytes(int which) {
    switch (which) {
        case 0: return (long long)sizeof(storm_conv_args);
        case 1: return (long long)sizeof(storm_op);
        case 2: return (long long)sizeof(storm_conv_seg);
        case 3: return (long long)sizeof(storm_ncsnpp_config);
        default: return -1;
    }
}
extern "C" int storm_device_info(char* name, int name_len, int* n_cu, size_t* hbm_bytes) {
    int dev = 0;
    STORM_HIP(hipGetDevice(&dev));
    hipDeviceProp_t p;
    STORM_HIP(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.gcnArchName, name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = p.totalGlobalMem;
    return STORM_OK;
}
