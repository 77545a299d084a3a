// Host-side runtime of libeeg2video_hip: error plumbing, the stream-ordered workspace cache and the
// weight store keyed by the reference's state-dict names.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/eeg2video_hip.h"

namespace e2v {

struct Error : std::runtime_error {
    e2v_status code;
    Error(e2v_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define E2V_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            throw ::e2v::Error(E2V_EHIP, std::string(#expr) + ": " + hipGetErrorString(_e));            \
    } while (0)

#define E2V_REQUIRE(cond, code, msg)                         \
    do {                                                     \
        if (!(cond)) throw ::e2v::Error((code), (msg));      \
    } while (0)

// Stream-ordered workspace cache.  All work of a ctx runs on one stream at a time, so a buffer handed
// back is immediately reusable by later launches on that stream; blocks are kept by size and reused,
// which makes steady-state calls allocation-free (hipMalloc only while the shape mix is new).
class Pool {
public:
    ~Pool() { trim(); }
    float* get(size_t floats) {
        size_t bytes = ((floats * sizeof(float) + 255) / 256) * 256;
        if (bytes == 0) bytes = 256;
        auto it = free_.lower_bound(bytes);
        if (it != free_.end() && it->first <= bytes + bytes / 4) {
            void* p = it->second;
            size_t sz = it->first;
            free_.erase(it);
            live_[p] = sz;
            return static_cast<float*>(p);
        }
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            trim();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) throw Error(E2V_EHIP, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
        total_ += bytes;
        live_[p] = bytes;
        return static_cast<float*>(p);
    }
    void put(float* p) {
        if (!p) return;
        auto it = live_.find(p);
        if (it == live_.end()) return;
        free_.emplace(it->second, p);
        live_.erase(it);
    }
    void trim() {
        for (auto& kv : free_) {
            (void)hipFree(kv.second);
            total_ -= kv.first;
        }
        free_.clear();
    }
    size_t bytes() const { return total_; }

private:
    std::multimap<size_t, void*> free_;
    std::unordered_map<void*, size_t> live_;
    size_t total_ = 0;
};

// channel-last activation [rows][C] living in the pool
struct Act {
    float* p = nullptr;
    int64_t rows = 0;
    int C = 0;
    Pool* pool = nullptr;
    Act() = default;
    Act(Pool& pl, int64_t r, int c) : p(pl.get((size_t)r * c)), rows(r), C(c), pool(&pl) {}
    Act(const Act&) = delete;
    Act& operator=(const Act&) = delete;
    Act(Act&& o) noexcept { *this = std::move(o); }
    Act& operator=(Act&& o) noexcept {
        if (this != &o) {
            reset();
            p = o.p; rows = o.rows; C = o.C; pool = o.pool;
            o.p = nullptr; o.pool = nullptr;
        }
        return *this;
    }
    ~Act() { reset(); }
    void reset() {
        if (p && pool) pool->put(p);
        p = nullptr;
    }
};

struct WTensor {
    float* d = nullptr;             // device, torch layout, fp32
    std::vector<int64_t> shape;
    size_t numel = 0;
    bool loaded = false;
};

}  // namespace e2v
