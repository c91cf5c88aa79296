// Error reporting, ABI version and host-side helpers of libw2l_hip.so.
#include "common.h"
#include <mutex>
#include <unordered_set>
#include <vector>
#include <algorithm>

static thread_local char g_err[512] = "";

void w2l_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

hipError_t w2l_allow_big_lds(const void* kernel) {
    static std::mutex mu;
    static std::unordered_set<const void*> done;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count(kernel)) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done.insert(kernel);
    return e;
}

extern "C" const char* w2l_last_error(void) { return g_err; }
extern "C" int w2l_abi_version(void) { return 1; }

// Edit distance over int32 symbols (stands in for python-Levenshtein's distance(),
// decoder.py:49,60).  Host code: the strings live on the host.
extern "C" int w2l_levenshtein_host(const int32_t* a, int na, const int32_t* b, int nb) {
    if (na < 0 || nb < 0) return -1;
    if (na < nb) { std::swap(a, b); std::swap(na, nb); }
    std::vector<int> prev(nb + 1), cur(nb + 1);
    for (int j = 0; j <= nb; ++j) prev[j] = j;
    for (int i = 1; i <= na; ++i) {
        cur[0] = i;
        for (int j = 1; j <= nb; ++j)
            cur[j] = std::min(std::min(prev[j] + 1, cur[j - 1] + 1), prev[j - 1] + (a[i - 1] != b[j - 1]));
        std::swap(prev, cur);
    }
    return prev[nb];
}
