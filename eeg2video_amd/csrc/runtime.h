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
#include "prof.h"

namespace e2v {

struct Error : std::runtime_error {
    e2v_status code;
    Error(e2v_status c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define E2V_HIP(expr)                                                                                   \
    do {                                                                                                \
        if (::e2v::dry_run()) break;                                                                    \
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
        if (dry_run()) return dry_fake_ptr(bytes);           // (put() does not know the address and ignores it)
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

// channel-last activation [rows][C] living in the pool; fp32, or bf16 (bf16-activation mode: `p` then points at 2-byte elements)
struct Act {
    float* p = nullptr;
    int64_t rows = 0;
    int C = 0;
    Pool* pool = nullptr;
    bool bf16 = false;
    float* rb = nullptr;           // row-block sums that came with the tensor (IgemmArgs::rbsum: [rows / 64][C][2]), pool-owned; null: none
    Act() = default;
    Act(Pool& pl, int64_t r, int c, bool half = false)
        : p(pl.get(half ? ((size_t)r * c + 1) / 2 : (size_t)r * c)), rows(r), C(c), pool(&pl), bf16(half) {}
    size_t bytes() const { return (size_t)rows * C * (bf16 ? 2 : 4); }
    const float* at(int64_t elem) const {          // address of element `elem` (counted in elements of the storage type)
        return reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + (size_t)elem * (bf16 ? 2 : 4));
    }
    Act(const Act&) = delete;
    Act& operator=(const Act&) = delete;
    Act(Act&& o) noexcept { *this = std::move(o); }
    Act& operator=(Act&& o) noexcept {
        if (this != &o) {
            reset();
            p = o.p; rows = o.rows; C = o.C; pool = o.pool; bf16 = o.bf16; rb = o.rb;
            o.p = nullptr; o.pool = nullptr; o.rb = nullptr;
        }
        return *this;
    }
    ~Act() { reset(); }
    void reset() {
        if (p && pool) pool->put(p);
        if (rb && pool) pool->put(rb);
        p = nullptr; rb = nullptr;
    }
};

struct WTensor {
    float* d = nullptr;             // device, torch layout, fp32
    std::vector<int64_t> shape;
    size_t numel = 0;
    bool loaded = false;
};

}  // namespace e2v
