/*
 * fake_rccl.cc — TEST INFRASTRUCTURE ONLY.  A stand-in for librccl.so.1 that lets TWO (or more) processes sharing ONE GPU run the
 * library's N > 1 collective path — ptx_comm_init(…, n) → ptx_allgather_digests → ptx_count_converged_digests, equal and unequal
 * blocks — on the one-GPU boxes the test-suite gets (tests/test_gpu_shard_ranks.py).  libperitext_hip.so binds RCCL at run time with
 * dlopen("librccl.so.1"): the test puts this directory first on LD_LIBRARY_PATH of its worker processes (which never import torch, so the
 * real RCCL is not in the process).  Nothing in the product links or loads this file; real RCCL over xGMI at N > 1 is what bench.py
 * --gpus N runs on a multi-GPU node.
 *
 * Implements the five entry points the library binds (peritext_hip.hip: PtxRccl): ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy,
 * ncclAllGather, ncclGetErrorString.  Ranks meet in a POSIX shared-memory segment named by the unique id; an all-gather is
 *   stream-sync → copy the send buffer to the segment → barrier → copy every rank's slot to the receive buffer → barrier.
 * It is synchronous with respect to the host (real RCCL is stream-ordered): work enqueued on the stream after the call still sees the
 * gathered data, which is all the callers rely on.
 */
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>

namespace {
constexpr size_t kSlotBytes = 8u << 20; /* per rank and call: 512 Ki digest pairs */
constexpr int kMaxRanks = 8;
struct Shared {
    std::atomic<uint32_t> joined;
    std::atomic<uint32_t> arrive[2]; /* sense-reversing barrier counters */
    std::atomic<uint32_t> calls;     /* all-gathers completed (the test asserts that the collective really ran) */
    uint32_t pad[12];
};
struct FakeComm {
    Shared* sh = nullptr;
    uint8_t* slots = nullptr;
    size_t map_bytes = 0;
    int rank = 0, n = 1;
    uint32_t phase = 0;
    char name[64];
};
bool wait_for(std::atomic<uint32_t>& a, uint32_t target, double seconds) {
    timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    while (a.load(std::memory_order_acquire) < target) {
        timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        if ((t.tv_sec - t0.tv_sec) + 1e-9 * (t.tv_nsec - t0.tv_nsec) > seconds) return false;
        usleep(50);
    }
    return true;
}
bool barrier(FakeComm* c) {
    const uint32_t k = c->phase++;
    /* counter k & 1 counts arrivals of barrier k: every barrier adds n to it, barrier k completes at n * (k / 2 + 1) */
    c->sh->arrive[k & 1].fetch_add(1, std::memory_order_acq_rel);
    return wait_for(c->sh->arrive[k & 1], (uint32_t)c->n * (k / 2 + 1), 60.0);
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    timespec t;
    clock_gettime(CLOCK_REALTIME, &t);
    snprintf(id->internal, sizeof(id->internal), "/ptxfakerccl_%d_%lx", (int)getpid(), (unsigned long)t.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    FakeComm* c = new FakeComm();
    c->rank = rank;
    c->n = nranks;
    strncpy(c->name, id.internal, sizeof(c->name) - 1);
    c->map_bytes = sizeof(Shared) + (size_t)nranks * kSlotBytes;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {
        if (fd >= 0) close(fd);
        delete c;
        return ncclSystemError;
    }
    void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) {
        delete c;
        return ncclSystemError;
    }
    c->sh = (Shared*)p; /* a fresh segment is zero-filled: the atomics start at 0 */
    c->slots = (uint8_t*)p + sizeof(Shared);
    c->sh->joined.fetch_add(1, std::memory_order_acq_rel);
    if (!wait_for(c->sh->joined, (uint32_t)nranks, 60.0)) {
        munmap(p, c->map_bytes);
        delete c;
        return ncclSystemError; /* the other ranks never came */
    }
    *comm = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    FakeComm* c = (FakeComm*)comm;
    if (!c) return ncclSuccess;
    if (c->sh) munmap((void*)c->sh, c->map_bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
    FakeComm* c = (FakeComm*)comm;
    if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
    const size_t esz = datatype == ncclUint64 || datatype == ncclInt64 || datatype == ncclFloat64 ? 8 : datatype == ncclUint8 || datatype == ncclInt8 ? 1 : 4;
    const size_t bytes = sendcount * esz;
    if (bytes > kSlotBytes) return ncclInvalidArgument;
    if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError; /* the send buffer is what the stream has produced so far */
    if (bytes && hipMemcpy(c->slots + (size_t)c->rank * kSlotBytes, sendbuff, bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError;
    for (int r = 0; r < c->n && bytes; ++r)
        if (hipMemcpy((uint8_t*)recvbuff + (size_t)r * bytes, c->slots + (size_t)r * kSlotBytes, bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclUnhandledCudaError;
    if (!barrier(c)) return ncclSystemError; /* nobody overwrites its slot before everyone has read it */
    if (c->rank == 0) c->sh->calls.fetch_add(1, std::memory_order_relaxed);
    return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : r == ncclInvalidArgument ? "invalid argument (fake rccl)" : "error (fake rccl)"; }

/* test hook: all-gathers completed on this communicator's segment */
uint32_t ptx_fake_rccl_calls(ncclComm_t comm) { return comm ? ((FakeComm*)comm)->sh->calls.load() : 0u; }
}
