// The one exchange of the path: the all-gather of the decoded frames of all ranks, over RCCL / xGMI, below the C ABI.
//
// One process per GPU, one ctx per process.  The library opens its own RCCL communicator: rank 0 draws a 128-byte id
// (e2v_comm_unique_id), the HOST side ships it to the other ranks by whatever channel it has (torch.distributed broadcast in the
// Python mirror: plumbing), every rank calls e2v_comm_init, and from then on e2v_allgather_frames is a single ncclAllGather on the
// caller's stream -- optionally of the uint8 form of the frames (tuneavideo/util.py:29: 4x fewer xGMI bytes when the consumer
// writes GIFs).  RCCL is resolved at run time (dlopen of the librccl already in the process -- torch's copy when the host is
// PyTorch -- else the ROCm one), so the library itself has no link-time dependency on it and loads on a box without RCCL.
#include <dlfcn.h>

#include <cstring>
#include <mutex>

#include "model.h"

using namespace e2v;

namespace {

// the four RCCL entry points used, with the ABI of <rccl/rccl.h> (ncclResult_t = int, ncclComm_t = opaque pointer,
// ncclUniqueId = 128 bytes passed BY VALUE, ncclDataType_t: ncclUint8 = 1, ncclFloat32 = 7)
struct UniqueId { char internal[128]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void**, int, UniqueId, int);
typedef int (*AllGatherFn)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*CommDestroyFn)(void*);
typedef const char* (*GetErrorStringFn)(int);

struct Rccl {
    void* handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllGatherFn all_gather = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn error_string = nullptr;
    std::string why;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // the soname first: that resolves to the copy ALREADY in the process (torch's, when the host is PyTorch) instead of loading a
        // second RCCL beside it through the unversioned development link
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.handle) break;
        }
        if (!r.handle) { r.why = std::string("dlopen(librccl.so): ") + dlerror(); return; }
        r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
        r.all_gather = reinterpret_cast<AllGatherFn>(dlsym(r.handle, "ncclAllGather"));
        r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
        r.error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.handle, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.all_gather || !r.comm_destroy) r.why = "librccl.so lacks an ncclAllGather entry point";
    });
    return r;
}

void need_rccl() {
    Rccl& r = rccl();
    E2V_REQUIRE(r.why.empty(), E2V_ESTATE, "RCCL is not available: " + r.why);
}

void check(int rc, const char* what) {
    if (rc == 0) return;
    Rccl& r = rccl();
    throw Error(E2V_EHIP, std::string(what) + ": " + (r.error_string ? r.error_string(rc) : "RCCL error " + std::to_string(rc)));
}

template <typename Fn>
e2v_status guarded(e2v_ctx* ctx, Fn&& fn) {
    static std::string g_err;
    try {
        if (ctx) {
            E2V_REQUIRE(ctx->device >= 0, E2V_ESTATE, "host-only context (device = -1): no GPU work possible");
            E2V_HIP(hipSetDevice(ctx->device));
        }
        fn();
        return E2V_OK;
    } catch (const Error& e) {
        if (ctx) ctx->err = e.what();
        return e.code;
    } catch (const std::exception& e) {
        if (ctx) ctx->err = e.what();
        return E2V_EINVAL;
    }
}

}  // namespace

extern "C" {

e2v_status e2v_comm_unique_id(void* id128_host) {
    if (!id128_host) return E2V_EINVAL;
    return guarded(nullptr, [&] {
        need_rccl();
        UniqueId id;
        check(rccl().get_unique_id(&id), "ncclGetUniqueId");
        std::memcpy(id128_host, id.internal, sizeof(id.internal));
    });
}

e2v_status e2v_comm_init(e2v_ctx* c, const void* id128_host, int rank, int world) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(id128_host && world >= 1 && rank >= 0 && rank < world, E2V_EINVAL, "e2v_comm_init: need the 128-byte id and 0 <= rank < world");
        E2V_REQUIRE(c->comm == nullptr, E2V_ESTATE, "e2v_comm_init: the ctx already holds a communicator (e2v_comm_destroy first)");
        need_rccl();
        UniqueId id;
        std::memcpy(id.internal, id128_host, sizeof(id.internal));
        void* comm = nullptr;
        check(rccl().comm_init_rank(&comm, world, id, rank), "ncclCommInitRank");
        c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    });
}

int e2v_comm_world(const e2v_ctx* c) { return c && c->comm ? c->comm_world : 0; }

e2v_status e2v_allgather_frames(e2v_ctx* c, const float* frames, int64_t count, int as_uint8, void* out, e2v_stream stream) {
    if (!c) return E2V_EINVAL;
    return guarded(c, [&] {
        E2V_REQUIRE(c->comm != nullptr, E2V_ESTATE, "e2v_allgather_frames: no communicator (e2v_comm_init)");
        E2V_REQUIRE(frames && out && count > 0, E2V_EINVAL, "e2v_allgather_frames: null buffer or empty shard");
        hipStream_t s = static_cast<hipStream_t>(stream);
        c->enter_stream(s);
        if (as_uint8) {
            // quantise into this rank's slot of the output, then gather in place (RCCL: sendbuff == recvbuff + rank * count is the in-place form)
            unsigned char* o = static_cast<unsigned char*>(out);
            unsigned char* mine = o + (size_t)c->comm_rank * (size_t)count;
            frames_to_u8(frames, mine, count, s);
            check(rccl().all_gather(mine, o, (size_t)count, /*ncclUint8*/ 1, c->comm, s), "ncclAllGather(uint8)");
        } else {
            check(rccl().all_gather(frames, out, (size_t)count, /*ncclFloat32*/ 7, c->comm, s), "ncclAllGather(float32)");
        }
    });
}

e2v_status e2v_comm_destroy(e2v_ctx* c) {
    if (!c) return E2V_EINVAL;
    if (!c->comm) return E2V_OK;
    return guarded(c, [&] {
        void* comm = c->comm;
        c->comm = nullptr; c->comm_world = 0; c->comm_rank = 0;
        check(rccl().comm_destroy(comm), "ncclCommDestroy");
    });
}

}  // extern "C"
