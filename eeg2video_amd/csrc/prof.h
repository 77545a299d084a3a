// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline figures).
// Off by default: a disabled ProfScope is two branches.  When on, every kernel launcher brackets its
// launch with an event pair and records the ALGORITHMIC flops / bytes of that launch.
#pragma once
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

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

struct ProfScope {
    hipStream_t s;
    bool live;
    size_t idx = 0;
    ProfScope(const char* name, double flops, double bytes, hipStream_t stream) : s(stream), live(profiler().on) {
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
