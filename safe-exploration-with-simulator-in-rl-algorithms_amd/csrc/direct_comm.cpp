// direct_comm.cpp -- the per-iteration all-gather issued straight into RCCL from native code
// (sw_comm_*, include/swimmer_hip.h), on the caller's stream, instead of through
// torch.distributed's ProcessGroupNCCL.  Replaces nothing in the reference by itself: it is the
// exchange step of the sharded form of the serial loop over directions (ars/ars_agent.py:160).
//
// RCCL is resolved at RUN time with dlopen, by soname: a process that has imported torch has
// torch's own librccl.so.1 loaded already and gets exactly that copy (two RCCL copies in one
// process would each own half of the state); a plain C caller gets the system library.  The
// library therefore has no link-time dependency on RCCL and loads on a machine without it.
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>

#include <initializer_list>
#include <new>

#include "../../include/swimmer_hip.h"

namespace {

// the five entry points used, with the C types of rccl.h (ncclResult_t / ncclDataType_t are ints,
// ncclComm_t is an opaque pointer, ncclUniqueId is 128 opaque bytes passed BY VALUE)
struct UniqueId { char internal[SW_COMM_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(void **, int, UniqueId, int);
typedef int (*CommDestroyFn)(void *);
typedef int (*AllGatherFn)(const void *, void *, size_t, int, void *, void *);
typedef const char *(*ErrorStringFn)(int);
constexpr int kNcclFloat64 = 8;   // rccl.h: ncclFloat64 = ncclDouble = 8

struct Rccl {
    void *handle = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    AllGatherFn all_gather = nullptr;
    ErrorStringFn error_string = nullptr;
    bool ok = false;
};

Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            x.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);      // the copy already in the process
            if (x.handle) break;
        }
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (x.handle) break;
            x.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!x.handle) return x;
        x.get_unique_id = (GetUniqueIdFn)dlsym(x.handle, "ncclGetUniqueId");
        x.comm_init_rank = (CommInitRankFn)dlsym(x.handle, "ncclCommInitRank");
        x.comm_destroy = (CommDestroyFn)dlsym(x.handle, "ncclCommDestroy");
        x.all_gather = (AllGatherFn)dlsym(x.handle, "ncclAllGather");
        x.error_string = (ErrorStringFn)dlsym(x.handle, "ncclGetErrorString");
        x.ok = x.get_unique_id && x.comm_init_rank && x.comm_destroy && x.all_gather;
        return x;
    }();
    return r;
}

}  // namespace

struct sw_comm {
    void *comm = nullptr;
    int world = 0, rank = 0;
    int last_error = 0;
};

extern "C" {

int sw_comm_available(void) { return rccl().ok ? 1 : 0; }

int sw_comm_unique_id(uint8_t *id)
{
    if (!id) return SW_ERR_NULL;
    if (!rccl().ok) return SW_ERR_LAUNCH;
    UniqueId u;
    if (rccl().get_unique_id(&u) != 0) return SW_ERR_LAUNCH;
    for (int i = 0; i < SW_COMM_ID_BYTES; ++i) id[i] = (uint8_t)u.internal[i];
    return SW_OK;
}

int sw_comm_create(sw_comm **out, const uint8_t *id, int32_t world, int32_t rank)
{
    if (!out || !id) return SW_ERR_NULL;
    if (world < 1 || rank < 0 || rank >= world) return SW_ERR_SIZE;
    if (!rccl().ok) return SW_ERR_LAUNCH;
    sw_comm *c = new (std::nothrow) sw_comm();
    if (!c) return SW_ERR_LAUNCH;
    UniqueId u;
    for (int i = 0; i < SW_COMM_ID_BYTES; ++i) u.internal[i] = (char)id[i];
    c->world = world;
    c->rank = rank;
    c->last_error = rccl().comm_init_rank(&c->comm, world, u, rank);
    if (c->last_error != 0) {
        delete c;
        return SW_ERR_LAUNCH;
    }
    *out = c;
    return SW_OK;
}

void sw_comm_destroy(sw_comm *c)
{
    if (!c) return;
    if (c->comm && rccl().ok) (void)rccl().comm_destroy(c->comm);
    delete c;
}

int sw_comm_all_gather_f64(sw_comm *c, const double *send, double *recv, int64_t count, void *stream)
{
    if (!c || !send || !recv) return SW_ERR_NULL;
    if (count < 0) return SW_ERR_SIZE;
    if (count == 0) return SW_OK;
    c->last_error = rccl().all_gather(send, recv, (size_t)count, kNcclFloat64, c->comm, stream);
    return c->last_error == 0 ? SW_OK : SW_ERR_LAUNCH;
}

const char *sw_comm_last_error(sw_comm *c)
{
    if (!c || !rccl().error_string) return "RCCL not available";
    return rccl().error_string(c->last_error);
}

}  // extern "C"
