#include "prof.h"

#include <cstdio>
#include <cstdlib>
#include <map>

#include <mutex>
#include <utility>

namespace e2v {

void kattr_max_lds(const void* kernel, int bytes) {
    if (dry_run()) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    static std::mutex m;
    static std::map<std::pair<const void*, int>, int> set;      // (kernel, device) -> bytes the attribute holds
    std::lock_guard<std::mutex> lk(m);
    int& have = set[{kernel, dev}];
    if (bytes <= have) return;
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    have = bytes;
}

Profiler& profiler() {
    static Profiler p;
    return p;
}

void Profiler::begin() {
    entries.clear();
    const char* d = std::getenv("E2V_PROFILE_DETAIL");
    detail = d && d[0] == '1';
    on = true;
}

std::string Profiler::end_json() {
    on = false;
    (void)hipDeviceSynchronize();
    struct Agg { long launches = 0; double ms = 0, flops = 0, bytes = 0; };
    std::map<std::string, Agg> agg;
    for (auto& e : entries) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e.a, e.b);
        Agg& a = agg[e.name];
        a.launches += 1; a.ms += ms; a.flops += e.flops; a.bytes += e.bytes;
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    entries.clear();
    std::string out = "{";
    bool first = true;
    char buf[512];
    for (auto& kv : agg) {
        std::snprintf(buf, sizeof(buf), "%s\"%s\": {\"launches\": %ld, \"ms\": %.6f, \"flops\": %.6e, \"bytes\": %.6e}",
                      first ? "" : ", ", kv.first.c_str(), kv.second.launches, kv.second.ms, kv.second.flops, kv.second.bytes);
        out += buf;
        first = false;
    }
    return out + "}";
}

}  // namespace e2v
