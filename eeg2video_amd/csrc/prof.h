// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline figures).
// Off by default: a disabled ProfScope is two branches.  When on, every kernel launcher brackets its
// launch with an event pair and records the ALGORITHMIC flops / bytes of that launch.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

namespace e2v {

// Dry run (e2v_op_describe_dispatch): the graph walker, the launch rules and every launcher run as usual on a host-only ctx, but no
// HIP call is made -- launches and E2V_HIP calls are skipped, workspace and weight "allocations" hand out distinct fake addresses --
// and every ProfScope records its (shape- and kernel-tagged) name instead of timing: the list IS the dispatch of that configuration.
inline bool& dry_run() { static thread_local bool on = false; return on; }
inline std::vector<std::string>& dry_log() { static thread_local std::vector<std::string> log; return log; }
inline void dry_tag(const std::string& tag) {               // a launcher's decision, appended to the record its ProfScope made
    if (dry_run() && !dry_log().empty()) dry_log().back() += tag;
}
inline float* dry_fake_ptr(size_t bytes) {                  // distinct, 256-byte aligned, never dereferenced
    static thread_local uintptr_t next = (uintptr_t)1 << 40;
    const uintptr_t p = next;
    next += (bytes + 255) / 256 * 256 + 256;
    return reinterpret_cast<float*>(p);
}

}  // namespace e2v

// every kernel launch of the library goes through this (skipped in a dry run)
#define E2V_KLAUNCH(...)                                          \
    do {                                                          \
        if (!::e2v::dry_run()) { hipLaunchKernelGGL(__VA_ARGS__); } \
    } while (0)

namespace e2v {

// Opt a kernel into its dynamic LDS size (hipFuncAttributeMaxDynamicSharedMemorySize; the default ceiling is 64 KB) -- once per
// (kernel, DEVICE): the attribute belongs to the device's code object, so a context on another device has to set it again; the
// largest size asked for so far is kept.  Never in a dry run: e2v_op_describe_dispatch makes no HIP call and leaves no state behind.
void kattr_max_lds(const void* kernel, int bytes);
}  // namespace e2v
#define E2V_KATTR(kernel, bytes) ::e2v::kattr_max_lds(reinterpret_cast<const void*>(kernel), (int)(bytes))
namespace e2v {

struct ProfEntry {
    std::string name;
    double flops, bytes;
    hipEvent_t a, b;
};

struct Profiler {
    bool on = false;
    bool detail = false;         // E2V_PROFILE_DETAIL=1: key igemm entries by shape as well
    std::vector<ProfEntry> entries;
    void begin();
    std::string end_json();      // synchronises, aggregates per kernel class, frees the events
};
Profiler& profiler();
// shape / kernel tags on a launcher's profile name: the detailed profile (tools/shape_profile.py) and every dry run
inline bool prof_detail() { return dry_run() || (profiler().on && profiler().detail); }

struct ProfScope {
    hipStream_t s;
    bool live;
    size_t idx = 0;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : s(stream), live(profiler().on) {
        if (dry_run()) { dry_log().push_back(name); live = false; return; }      // e2v_op_describe_dispatch: the name is the record
        if (!live) return;
        ProfEntry e{name, flops, bytes, nullptr, nullptr};
        (void)hipEventCreate(&e.a);
        (void)hipEventCreate(&e.b);
        (void)hipEventRecord(e.a, s);
        idx = profiler().entries.size();
        profiler().entries.push_back(e);
    }
    ~ProfScope() {
        if (live) (void)hipEventRecord(profiler().entries[idx].b, s);
    }
};

}  // namespace e2v
