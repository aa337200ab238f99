// Layer-boundary exchange of the head-sharded attention output (include/rsa.h "multi-GPU"):
// every rank holds O for its heads as [rows = B*S][Hl*D]; the consumer (to_out GEMM) of an unsharded model wants
// [rows][H*D] on every rank.  Two transports behind the C-ABI:
//   * rsa_allgather_heads      RCCL ncclAllGather (rank-major staging) + one unpack kernel into the head-major rows;
//   * rsa_allgather_heads_p2p  every rank writes its [rows][Hl*D] slab straight into its column range of every peer's
//                              full buffer with ONE copy kernel whose workgroups are dealt over the peers, so all xGMI
//                              links (point to point, one per peer) carry their slab at the same time: 16-byte peer
//                              stores, no staging, no unpack.  Arrival is signalled through IPC-shared device flags
//                              (an epoch per source rank, written behind a system-scope release by the last workgroup of
//                              a peer's share) and awaited by a one-workgroup kernel on the consumer's stream: the exchange
//                              is stream-ordered end to end -- no host barrier, no host synchronisation, capturable.
// RCCL is bound at run time (dlopen): a process that already carries an RCCL (PyTorch bundles one) keeps using that
// copy, and single-GPU hosts do not need the library at all.  Nothing here is on the attention path itself: the path
// shards by head with no collective.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "rsa.h"

extern int g_rsa_last_hip_error;

namespace {

typedef struct { char internal[128]; } rsa_nccl_id;   // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128)
typedef int (*fn_get_id)(rsa_nccl_id*);
typedef int (*fn_init_rank)(void**, int, rsa_nccl_id, int);
typedef int (*fn_destroy)(void*);
typedef int (*fn_allgather)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*fn_count)(void*, int*);

struct Rccl {
    void* h = nullptr;
    fn_get_id get_id = nullptr;
    fn_init_rank init_rank = nullptr;
    fn_destroy destroy = nullptr;
    fn_allgather allgather = nullptr;
    fn_count count = nullptr;
    bool tried = false;
};
Rccl g_rccl;

bool load_rccl() {
    if (g_rccl.tried) return g_rccl.allgather != nullptr;
    g_rccl.tried = true;
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names) {   // a copy the process already loaded (PyTorch's) first
        g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        if (g_rccl.h) break;
    }
    for (int i = 0; !g_rccl.h && i < 2; ++i) g_rccl.h = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
    if (!g_rccl.h) return false;
    g_rccl.get_id = (fn_get_id)dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_init_rank)dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.destroy = (fn_destroy)dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.allgather = (fn_allgather)dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.count = (fn_count)dlsym(g_rccl.h, "ncclCommCount");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.destroy || !g_rccl.allgather) g_rccl.allgather = nullptr;
    return g_rccl.allgather != nullptr;
}

int hip_status(hipError_t e) {
    if (e == hipSuccess) return RSA_OK;
    g_rsa_last_hip_error = (int)e;
    return RSA_ERR_LAUNCH;
}

// staging [world][rows][w16] (16-byte units) -> full [rows][world * w16]
__global__ __launch_bounds__(256) void unpack_heads_kernel(const uint4* __restrict__ staging, uint4* __restrict__ full,
                                                           long rows, int w16, int world) {
    const long n = rows * (long)w16 * world;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long row = i / ((long)w16 * world);
        const int col = (int)(i % ((long)w16 * world));
        const int rk = col / w16, c = col % w16;
        full[i] = staging[((long)rk * rows + row) * w16 + c];
    }
}

}  // namespace

extern "C" int rsa_comm_unique_id(void* id128) {
    if (!id128) return RSA_ERR_BAD_ARG;
    if (!load_rccl()) return RSA_ERR_UNSUPPORTED;
    rsa_nccl_id id;
    if (g_rccl.get_id(&id) != 0) return RSA_ERR_LAUNCH;
    memcpy(id128, &id, sizeof(id));
    return RSA_OK;
}

extern "C" int rsa_comm_create(int world, int rank, const void* id128, void** comm) {
    if (!id128 || !comm || world <= 0 || rank < 0 || rank >= world) return RSA_ERR_BAD_ARG;
    if (!load_rccl()) return RSA_ERR_UNSUPPORTED;
    rsa_nccl_id id;
    memcpy(&id, id128, sizeof(id));
    void* c = nullptr;
    if (g_rccl.init_rank(&c, world, id, rank) != 0) return RSA_ERR_LAUNCH;
    *comm = c;
    return RSA_OK;
}

extern "C" int rsa_comm_count(void* comm, int* ranks) {
    if (!comm || !ranks) return RSA_ERR_BAD_ARG;
    if (!load_rccl() || !g_rccl.count) return RSA_ERR_UNSUPPORTED;
    return g_rccl.count(comm, ranks) == 0 ? RSA_OK : RSA_ERR_LAUNCH;
}

extern "C" int rsa_comm_destroy(void* comm) {
    if (!comm) return RSA_ERR_BAD_ARG;
    if (!load_rccl()) return RSA_ERR_UNSUPPORTED;
    return g_rccl.destroy(comm) == 0 ? RSA_OK : RSA_ERR_LAUNCH;
}

extern "C" int rsa_allgather_heads(void* comm, int world, const void* local, void* staging, void* full, int64_t rows,
                                   int64_t local_row_bytes, void* stream) {
    if (!comm || !local || !staging || !full || world <= 0 || rows <= 0 || local_row_bytes <= 0) return RSA_ERR_BAD_ARG;
    if ((local_row_bytes & 15) || ((uintptr_t)local & 15) || ((uintptr_t)staging & 15) || ((uintptr_t)full & 15))
        return RSA_ERR_BAD_ARG;
    if (!load_rccl()) return RSA_ERR_UNSUPPORTED;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t count = (size_t)rows * (size_t)local_row_bytes;
    if (g_rccl.allgather(local, staging, count, /*ncclInt8*/ 0, comm, s) != 0) return RSA_ERR_LAUNCH;
    const int w16 = (int)(local_row_bytes / 16);
    const long n = rows * (long)w16 * world;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    unpack_heads_kernel<<<dim3((unsigned)blocks), 256, 0, s>>>(static_cast<const uint4*>(staging),
                                                               static_cast<uint4*>(full), rows, w16, world);
    return hip_status(hipGetLastError());
}

extern "C" int rsa_ipc_export(const void* dev_ptr, void* handle64) {
    if (!dev_ptr || !handle64) return RSA_ERR_BAD_ARG;
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "IPC handle does not fit the 64-byte slot of the C-ABI");
    hipIpcMemHandle_t h;
    const int st = hip_status(hipIpcGetMemHandle(&h, const_cast<void*>(dev_ptr)));
    if (st != RSA_OK) return st;
    memset(handle64, 0, 64);
    memcpy(handle64, &h, sizeof(h));
    return RSA_OK;
}

extern "C" int rsa_ipc_open(const void* handle64, int peer_device, void** dev_ptr) {
    if (!handle64 || !dev_ptr) return RSA_ERR_BAD_ARG;
    if (peer_device >= 0) {   // let this device address the peer's memory (idempotent)
        int cur = 0;
        int st = hip_status(hipGetDevice(&cur));
        if (st != RSA_OK) return st;
        if (peer_device != cur) {
            const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();   // clear exactly this sticky state
            else if (e != hipSuccess) return hip_status(e);
        }
    }
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    return hip_status(hipIpcOpenMemHandle(dev_ptr, h, hipIpcMemLazyEnablePeerAccess));
}

extern "C" int rsa_ipc_close(void* dev_ptr) {
    if (!dev_ptr) return RSA_ERR_BAD_ARG;
    return hip_status(hipIpcCloseMemHandle(dev_ptr));
}

extern "C" int rsa_ipc_offset(const void* dev_ptr, int64_t* offset) {
    // byte offset of dev_ptr inside its allocation: an IPC handle names the allocation (a framework's caching allocator
    // hands out interior pointers), so the opener adds this offset to the pointer rsa_ipc_open returns
    if (!dev_ptr || !offset) return RSA_ERR_BAD_ARG;
    hipDeviceptr_t base = nullptr;
    size_t size = 0;
    const int st = hip_status(hipMemGetAddressRange(&base, &size, const_cast<void*>(dev_ptr)));
    if (st != RSA_OK) return st;
    *offset = (int64_t)((const unsigned char*)dev_ptr - (const unsigned char*)base);
    return RSA_OK;
}

namespace {

// exchange state of one rank (device memory, zeroed once by its owner, IPC-shared): 32-bit words
//   [0] epoch = gathers completed by this rank   [1] time-out word (nonzero: a wait gave up)
//   [ARRIVE + p] workgroups of this rank's copy kernel done with peer p's share (local counter)
//   [FLAGS + r]  written BY rank r: the epoch of its slab that has landed in this rank's full buffer
enum { P2P_EPOCH = 0, P2P_TIMEOUT = 1, P2P_ARRIVE = 16, P2P_FLAGS = 128, P2P_WORDS = 256, P2P_MAX_WORLD = 64 };
constexpr int P2P_BLOCKS_PER_PEER = 48;

struct P2pPeers {
    unsigned char* full[P2P_MAX_WORLD];
    unsigned* state[P2P_MAX_WORLD];
};

__global__ __launch_bounds__(256) void p2p_copy_kernel(const uint4* __restrict__ local, P2pPeers peers, int world, int rank,
                                                        long rows, int w16) {
    const int p = blockIdx.x / P2P_BLOCKS_PER_PEER, part = blockIdx.x % P2P_BLOCKS_PER_PEER;
    unsigned* my_state = peers.state[rank];
    const unsigned epoch = my_state[P2P_EPOCH] + 1;   // (bumped by this gather's wait kernel, behind this kernel)
    uint4* dst = reinterpret_cast<uint4*>(peers.full[p]) + (long)rank * w16;
    const long n = rows * (long)w16, pitch16 = (long)w16 * world;
    for (long i = (long)part * 256 + threadIdx.x; i < n; i += (long)P2P_BLOCKS_PER_PEER * 256) {
        const long row = i / w16;
        const int c = (int)(i % w16);
        dst[row * pitch16 + c] = local[i];
    }
    __threadfence_system();   // this thread's peer stores before the arrival below
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = atomicAdd(&my_state[P2P_ARRIVE + p], 1u);
        if (old == P2P_BLOCKS_PER_PEER - 1) {   // the whole share of peer p has been written (every block fenced first)
            my_state[P2P_ARRIVE + p] = 0;
            __hip_atomic_store(&peers.state[p][P2P_FLAGS + rank], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// one workgroup on the consumer's stream: until every source's slab of this epoch has landed here (bounded spin)
__global__ __launch_bounds__(64) void p2p_wait_kernel(unsigned* state, int world, int rank, long long timeout_ticks) {
    const unsigned epoch = state[P2P_EPOCH] + 1;
    const int p = threadIdx.x;
    if (p < world && p != rank) {
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(&state[P2P_FLAGS + p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > timeout_ticks) {
                __hip_atomic_store(&state[P2P_TIMEOUT], (unsigned)(p + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) state[P2P_EPOCH] = epoch;
}

}  // namespace

extern "C" int rsa_p2p_state_bytes(void) { return P2P_WORDS * 4; }

// The exchange state is polled by this GPU while PEERS write it over xGMI (flags) -- and written by this GPU into peers.  Cross-
// agent visibility of such polling is only defined for FINE-GRAINED memory (the AMDGPU memory model; a plain hipMalloc block is
// coarse-grained: a system-scope acquire load may keep hitting a stale L2 line).  So the library allocates the state block itself,
// fine-grained, zeroed; the caller IPC-exports it like any allocation (the data buffers stay ordinary device memory).
extern "C" int rsa_p2p_state_alloc(void** state) {
    if (!state) return RSA_ERR_BAD_ARG;
    *state = nullptr;
    void* p = nullptr;
    int st = hip_status(hipExtMallocWithFlags(&p, 4096, hipDeviceMallocFinegrained));
    if (st != RSA_OK) return st;
    st = hip_status(hipMemset(p, 0, 4096));
    if (st == RSA_OK) st = hip_status(hipDeviceSynchronize());
    if (st != RSA_OK) { (void)hipFree(p); return st; }
    *state = p;
    return RSA_OK;
}

// word 1 of a state block (0, or the rank a wait gave up on + 1), read back synchronously
extern "C" int rsa_p2p_state_timeout(const void* state, int* missing_rank_plus_1) {
    if (!state || !missing_rank_plus_1) return RSA_ERR_BAD_ARG;
    unsigned w[2] = {0, 0};
    const int st = hip_status(hipMemcpy(w, state, 8, hipMemcpyDeviceToHost));
    *missing_rank_plus_1 = (int)w[P2P_TIMEOUT];
    return st;
}

extern "C" int rsa_p2p_state_free(void* state) {
    if (!state) return RSA_OK;
    return hip_status(hipFree(state));
}

extern "C" int rsa_allgather_heads_p2p(int world, int rank, const void* local, void* const* full_of_rank,
                                       void* const* state_of_rank, int64_t rows, int64_t local_row_bytes, void* stream) {
    if (!local || !full_of_rank || !state_of_rank || world <= 0 || world > P2P_MAX_WORLD || rank < 0 || rank >= world ||
        rows <= 0 || local_row_bytes <= 0)
        return RSA_ERR_BAD_ARG;
    if ((local_row_bytes & 15) || ((uintptr_t)local & 15)) return RSA_ERR_BAD_ARG;
    P2pPeers peers;
    for (int r = 0; r < world; ++r) {
        if (!full_of_rank[r] || !state_of_rank[r] || ((uintptr_t)full_of_rank[r] & 15)) return RSA_ERR_BAD_ARG;
        peers.full[r] = static_cast<unsigned char*>(full_of_rank[r]);
        peers.state[r] = static_cast<unsigned*>(state_of_rank[r]);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    p2p_copy_kernel<<<dim3((unsigned)(world * P2P_BLOCKS_PER_PEER)), 256, 0, s>>>(
        static_cast<const uint4*>(local), peers, world, rank, (long)rows, (int)(local_row_bytes / 16));
    int st = hip_status(hipGetLastError());
    if (st != RSA_OK) return st;
    // 100 MHz wall clock: give a peer 4 s before the wait gives up (the time-out word then names the missing rank + 1)
    p2p_wait_kernel<<<1, 64, 0, s>>>(peers.state[rank], world, rank, 400000000ll);
    return hip_status(hipGetLastError());
}
