// Test double of librccl.so for a box with ONE GPU (tests/test_native_sharded_gpu.py, two-process test; LQRRT_RCCL points here).
//
// lqrrt_engine_extend_sharded's only RCCL call site is ncclAllGather on the engine's stream, in place (the rank's block is its
// chunk of the receive buffer).  Real RCCL refuses two ranks on one device, and the loopback communicator of the library skips
// the collective altogether, so on a one-GPU box nothing would ever execute the in-place offsets, the per-rank tail cursors and
// the sequencing of a wave across PROCESSES.  This file implements the six entry points the engine resolves (+ ncclCommAbort)
// over POSIX shared memory: every rank copies its chunk to its slot (device -> host), a process-shared barrier, every rank
// copies all slots into its receive buffer (host -> device), a second barrier.  Synchronous with respect to the stream -- a
// correctness double, not a performance model.  Test infrastructure: nothing in lqrrt_amd/ links or names it.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {
constexpr size_t SLOT = 16u << 20;          // bytes per rank (a 256-sample boat wave is < 1 MB per rank)
constexpr int MAXW = 16;
struct Header {
    std::atomic<int> count;
    std::atomic<int> sense;
    std::atomic<int> joined;
};
struct Comm {
    int rank, world, local_sense;
    char name[64];
    Header* h;
    unsigned char* slots;
    size_t bytes;
};
bool barrier(Comm* c) {
    c->local_sense ^= 1;
    const auto t0 = std::chrono::steady_clock::now();
    if (c->h->count.fetch_add(1, std::memory_order_acq_rel) == c->world - 1) {
        c->h->count.store(0, std::memory_order_relaxed);
        c->h->sense.store(c->local_sense, std::memory_order_release);
        return true;
    }
    while (c->h->sense.load(std::memory_order_acquire) != c->local_sense) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) return false;
        usleep(50);
    }
    return true;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
    memset(id->internal, 0, 128);
    unsigned r = 0;
    FILE* f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(&r, sizeof r, 1, f) != 1) r = 12345u; fclose(f); }
    snprintf(id->internal, 64, "/lqrrt_stub_rccl_%d_%08x", (int)getpid(), r);
    return 0;
}

int ncclCommInitRank(void** comm, int world, ncclUniqueId id, int rank) {
    if (!comm || world < 1 || world > MAXW || rank < 0 || rank >= world) return 4;
    Comm* c = new Comm();
    c->rank = rank; c->world = world; c->local_sense = 0;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->bytes = sizeof(Header) + 64 + SLOT * (size_t)world;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return 2; }
    if (ftruncate(fd, (off_t)c->bytes) != 0) { close(fd); delete c; return 2; }       // (new pages are zero: count = sense = 0)
    void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return 2; }
    c->h = (Header*)p;
    c->slots = (unsigned char*)p + sizeof(Header) + 64;
    c->h->joined.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();       // everybody has to have mapped the segment before the first collective
    while (c->h->joined.load() < world) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) return 6;
        usleep(100);
    }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    munmap((void*)c->h, c->bytes);
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
    return 0;
}
int ncclCommAbort(void* comm) { return ncclCommDestroy(comm); }

// count elements of `type` per rank (the engine passes bytes as ncclUint8 = 1)
int ncclAllGather(const void* send, void* recv, size_t count, int type, void* comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    if (!c || type != 1 || count > SLOT) return 4;
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    if (hipMemcpy(c->slots + SLOT * (size_t)c->rank, send, count, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    if (!barrier(c)) return 6;
    for (int g = 0; g < c->world; ++g)
        if (hipMemcpy((unsigned char*)recv + count * (size_t)g, c->slots + SLOT * (size_t)g, count, hipMemcpyHostToDevice) != hipSuccess) return 1;
    if (!barrier(c)) return 6;                               // nobody refills a slot before everybody has read it
    return 0;
}

const char* ncclGetErrorString(int r) {
    switch (r) {
        case 0: return "success";
        case 1: return "stub: HIP call failed";
        case 2: return "stub: shared memory segment";
        case 4: return "stub: invalid argument";
        case 6: return "stub: a rank did not arrive within 120 s";
        default: return "stub: error";
    }
}
}
