// options.hip -- process-wide kernel-variant overrides (options.h); host code only.
#include "options.h"

#include <cstdlib>
#include <cstring>
#include <mutex>

#include "../../include/tfmpc_hip.h"

namespace tfmpc {
namespace {

constexpr int kMaxLen = 32;
const char *const kNames[kOptCount] = {"TFMPC_LQR_KERNEL", "TFMPC_LQR_MFMA", "TFMPC_ILQR_KERNEL", "TFMPC_COSTATE_WAVES", "TFMPC_ILQR_RETRY", "TFMPC_COSTATE_COUPLING", "TFMPC_LQR_WAVES", "TFMPC_BOX_HELPERS", "TFMPC_BOX_HELP_AFTER", "TFMPC_ILQR_LQ_REUSE", "TFMPC_GROUP_STORED", "TFMPC_BOX_SPECULATE"};

struct Table {
    char value[kOptCount][kMaxLen];
    std::mutex lock;
    Table()
    {
        for (int i = 0; i < kOptCount; ++i) {
            const char *v = std::getenv(kNames[i]);           // the only getenv of the library
            value[i][0] = 0;
            if (v) std::strncat(value[i], v, kMaxLen - 1);
        }
    }
};

Table &table()
{
    static Table t;       // constructed on first use, thread-safe in C++11
    return t;
}

}  // namespace

bool option_is(Option which, const char *value)
{
    Table &t = table();
    std::lock_guard<std::mutex> g(t.lock);
    return t.value[which][0] != 0 && std::strcmp(t.value[which], value) == 0;
}

int option_int(Option which, int fallback)
{
    Table &t = table();
    std::lock_guard<std::mutex> g(t.lock);
    if (t.value[which][0] == 0) return fallback;
    char *end = nullptr;
    const long v = std::strtol(t.value[which], &end, 10);
    return (end && *end == 0) ? (int)v : fallback;
}

}  // namespace tfmpc

extern "C" int tfmpc_set_option(const char *name, const char *value)
{
    using namespace tfmpc;
    if (!name) return TFMPC_ERR_ARG;
    if (value && std::strlen(value) >= (size_t)kMaxLen) return TFMPC_ERR_ARG;
    Table &t = table();
    for (int i = 0; i < kOptCount; ++i) {
        if (std::strcmp(name, kNames[i]) != 0) continue;
        std::lock_guard<std::mutex> g(t.lock);
        t.value[i][0] = 0;
        if (value) std::strncat(t.value[i], value, kMaxLen - 1);
        return TFMPC_OK;
    }
    return TFMPC_ERR_ARG;
}


// the current override of `name` copied into buf (empty string = no override): lets a caller restore what it replaces
extern "C" int tfmpc_get_option(const char *name, char *buf, int len)
{
    using namespace tfmpc;
    if (!name || !buf || len <= 0) return TFMPC_ERR_ARG;
    Table &t = table();
    for (int i = 0; i < kOptCount; ++i) {
        if (std::strcmp(name, kNames[i]) != 0) continue;
        std::lock_guard<std::mutex> g(t.lock);
        buf[0] = 0;
        std::strncat(buf, t.value[i], (size_t)len - 1);
        return TFMPC_OK;
    }
    return TFMPC_ERR_ARG;
}
