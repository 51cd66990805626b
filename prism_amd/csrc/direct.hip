// Direct small-message all-reduce over peer-mapped gradient buffers (the reference has no
// collective at all -- /root/reference/prism/learner.py:95-125 is a single process).
//
// Why: the data-parallel step all-reduces 0.8 MB (configs[2]) to 6.2 MB (configs[3]) of fp32 gradient once per 80-250 us
// step.  A ring over xGMI is per-link bound and pays 2 (W - 1) hop latencies; on a fully connected node (7 links x ~153 GB/s
// per GPU) every GPU can instead pull its 1/W slice from all W - 1 peers at once, and then every slice from its owner:
//   reduce-scatter  rank r sums slice r of ALL ranks' buffers, in rank order 0..W-1, into slice r of its own buffer
//   all-gather      rank r copies slice s (s != r) from rank s's buffer
// Each slice is summed by exactly one rank in a fixed order, so all replicas end up with bit-identical gradients whatever the
// timing -- the property the redundant clip + Adam relies on.
//
// The peers' buffers are device pointers the HOST mapped into this process (hipIpcOpenMemHandle; the Python host lets
// torch do that).  Three points of the protocol need every rank to have arrived: gradients written -> reduce-scatter ->
// all-gather -> gradients may be overwritten.  They are either the host's business (use_flags = 0: the caller puts its own
// barrier between the calls -- the only legal form for ranks that SHARE a device, where a spinning kernel would starve the
// peer it waits for) or device flags (use_flags = 1, ranks on distinct devices): a one-workgroup kernel stores the phase
// number into slot [rank] of every peer's flag array (system-scope release) and polls its own array until every slot has
// reached it (bounded: gives up after the descriptor's wait_seconds, sets the sticky error slot and poisons the step).
//
// Memory types.  The flag arrays are UNCACHED device memory (prism_direct_flags_alloc: hipExtMallocWithFlags,
// hipDeviceMallocUncached): a peer's store into an ordinary hipMalloc (coarse-grained) allocation is only guaranteed to be
// seen at kernel boundaries -- the polling lane could spin on a stale line of its own L2.  The gradient buffers stay ordinary
// allocations: each is written by kernels that have ENDED (system-scope write-back) before the flag that announces them is
// stored by the NEXT kernel on the stream, and read by kernels that START (invalidate) after the wait kernel has ended.
#include <string.h>

#include "common.h"

namespace prism {

constexpr double DIRECT_WAIT_DEFAULT_S = 30.0;      // RCCL would wait for ever; a peer saving a checkpoint stalls for seconds

struct DirectArgs {
    float *bufs[PRISM_MAX_PEERS];
    unsigned int *flags[PRISM_MAX_PEERS];
    int world, rank;
    long long n;
    unsigned int phase;
    unsigned int *poison, *host_status;
    unsigned long long wait_ticks;      // of the 100 MHz real-time counter
};

// a wait of THIS rank has given up (sticky): every later kernel of the collective leaves the buffers alone
__device__ __forceinline__ bool direct_poisoned(const DirectArgs &a) {
    return a.flags[a.rank] != nullptr &&
           __hip_atomic_load(a.flags[a.rank] + PRISM_MAX_PEERS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// float4 range [lo, hi) of slice s; the n & 3 trailing floats belong to the last slice (handled by its owner as scalars)
__device__ __forceinline__ void slice_range(long long n, int world, int s, long long &lo, long long &hi) {
    const long long n4 = n >> 2, per = (n4 + world - 1) / world;
    lo = per * s < n4 ? per * s : n4;
    hi = per * (s + 1) < n4 ? per * (s + 1) : n4;
}

__global__ __launch_bounds__(256) void direct_reduce_scatter_kernel(DirectArgs a) {
    if (direct_poisoned(a)) return;
    long long lo, hi;
    slice_range(a.n, a.world, a.rank, lo, hi);
    float4 *own = reinterpret_cast<float4 *>(a.bufs[a.rank]);
    for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) {
        float4 v[PRISM_MAX_PEERS];
#pragma unroll
        for (int s = 0; s < PRISM_MAX_PEERS; ++s)            // every peer's piece requested before the first add
            if (s < a.world) v[s] = reinterpret_cast<const float4 *>(a.bufs[s])[i];
        float4 acc = v[0];
#pragma unroll
        for (int s = 1; s < PRISM_MAX_PEERS; ++s)            // rank order: the one summation order of this element
            if (s < a.world) {
                acc.x += v[s].x; acc.y += v[s].y; acc.z += v[s].z; acc.w += v[s].w;
            }
        own[i] = acc;
    }
    if (a.rank == a.world - 1 && blockIdx.x == 0 && (int)threadIdx.x < (int)(a.n & 3)) {
        const long long i = ((a.n >> 2) << 2) + threadIdx.x;
        float acc = a.bufs[0][i];
        for (int s = 1; s < a.world; ++s) acc += a.bufs[s][i];
        a.bufs[a.rank][i] = acc;
    }
}

__global__ __launch_bounds__(256) void direct_all_gather_kernel(DirectArgs a) {
    if (direct_poisoned(a)) return;
    float4 *own = reinterpret_cast<float4 *>(a.bufs[a.rank]);
    for (int d = 1; d < a.world; ++d) {
        const int s = (a.rank + d) % a.world;                 // (every rank starts at another peer: the links share the load)
        long long lo, hi;
        slice_range(a.n, a.world, s, lo, hi);
        const float4 *src = reinterpret_cast<const float4 *>(a.bufs[s]);
        for (long long i = lo + (long long)blockIdx.x * 256 + threadIdx.x; i < hi; i += (long long)gridDim.x * 256) own[i] = src[i];
    }
    if (a.rank != a.world - 1 && blockIdx.x == 0 && (int)threadIdx.x < (int)(a.n & 3)) {
        const long long i = ((a.n >> 2) << 2) + threadIdx.x;
        a.bufs[a.rank][i] = a.bufs[a.world - 1][i];
    }
}

// flags[r][s]: the last phase rank s has announced to rank r; slot PRISM_MAX_PEERS of the own array: sticky time-out;
// slot PRISM_MAX_PEERS + 1: the running all-reduce count E of this rank (all ranks advance it in lockstep).  The phase
// announced is 3 E + a.phase: a captured hipGraph replays with fresh phase numbers.  `what`: bit 0 announce, bit 1 wait
// (both in one launch on the step's path; apart -- with the host's barrier between them -- where ranks share a device).
__global__ __launch_bounds__(64) void direct_signal_wait_kernel(DirectArgs a_in, int what) {
    DirectArgs a = a_in;
    const int t = threadIdx.x;
    const unsigned int E = a.flags[a.rank][PRISM_MAX_PEERS + 1];
    a.phase = 3u * E + a_in.phase;
    __builtin_amdgcn_s_barrier();                             // (every lane has read E before lane 0 may advance it)
    if (a_in.phase == 3u && (what & 2) && t == 0) a.flags[a.rank][PRISM_MAX_PEERS + 1] = E + 1u;
    if (t < a.world) {
        if (what & 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: what this rank wrote is visible to the peers first
            __hip_atomic_store(a.flags[t] + a.rank, a.phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (what & 2) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            // a rank that has given up once announces its phases (the peers need not time out on it as well) but waits no more
            const bool dead = direct_poisoned(a);
            // (phase numbers only grow; compared as a signed difference so that the counter may wrap)
            while (!dead && (int)(__hip_atomic_load(a.flags[a.rank] + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.phase) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (__builtin_amdgcn_s_memrealtime() - t0 > a.wait_ticks) {
                    atomicOr(a.flags[a.rank] + PRISM_MAX_PEERS, 1u);
                    if (a.poison) atomicOr(a.poison, PRISM_WS_STATUS_COLLECTIVE_TIMEOUT);
                    if (a.host_status)      // (a plain system-scope store of word 1 = bit 1: needs no PCIe atomics)
                        __hip_atomic_store(a.host_status + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
    }
}

static int fill(const prism_direct_desc *d, DirectArgs &a, bool need_flags) {
    PRISM_CHECK_ARG(d != nullptr, "null descriptor");
    PRISM_CHECK_ARG(d->world >= 1 && d->world <= PRISM_MAX_PEERS && d->rank >= 0 && d->rank < d->world, "world / rank out of range");
    PRISM_CHECK_ARG(d->n > 0, "empty buffer");
    memset(&a, 0, sizeof(a));
    for (int s = 0; s < d->world; ++s) {
        PRISM_CHECK_ARG(d->bufs[s] != nullptr && ((uintptr_t)d->bufs[s] & 15) == 0, "peer buffers must be mapped and 16-byte aligned");
        PRISM_CHECK_ARG(!need_flags || d->flags[s] != nullptr, "flag arrays missing");
        a.bufs[s] = d->bufs[s];
        a.flags[s] = d->flags[s];
    }
    a.world = d->world;
    a.rank = d->rank;
    a.n = d->n;
    a.poison = d->poison;
    a.host_status = d->host_status;
    PRISM_CHECK_ARG(d->wait_seconds >= 0.0 && d->wait_seconds < 1e6, "wait_seconds out of range");
    a.wait_ticks = (unsigned long long)((d->wait_seconds > 0.0 ? d->wait_seconds : DIRECT_WAIT_DEFAULT_S) * 1e8);
    return PRISM_OK;
}

static int grid_for(long long n, int world) {
    const long long per = ((n >> 2) + world - 1) / world;
    long long b = (per + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace prism

using namespace prism;

static int signal_wait(DirectArgs &a, unsigned int phase, hipStream_t stream, int what = 3) {
    a.phase = phase;
    hipLaunchKernelGGL(direct_signal_wait_kernel, dim3(1), dim3(64), 0, stream, a, what);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_direct_phase(const prism_direct_desc *d, int32_t phase, int32_t what, prism_stream_t stream_) {
    DirectArgs a;
    int rc = fill(d, a, true);
    if (rc) return rc;
    PRISM_CHECK_ARG(phase >= 1 && phase <= 3 && what >= 1 && what <= 3, "phase is 1..3, what is 1 (announce), 2 (wait) or 3");
    return signal_wait(a, (unsigned int)phase, (hipStream_t)stream_, what);
}

extern "C" int prism_direct_reduce_scatter(const prism_direct_desc *d, int32_t use_flags, prism_stream_t stream_) {
    DirectArgs a;
    int rc = fill(d, a, use_flags != 0);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (use_flags && (rc = signal_wait(a, 1u, stream))) return rc;          // every rank's gradient is complete
    hipLaunchKernelGGL(direct_reduce_scatter_kernel, dim3(grid_for(a.n, a.world)), dim3(256), 0, stream, a);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_direct_all_gather(const prism_direct_desc *d, int32_t use_flags, prism_stream_t stream_) {
    DirectArgs a;
    int rc = fill(d, a, use_flags != 0);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    if (use_flags && (rc = signal_wait(a, 2u, stream))) return rc;          // every slice is reduced
    hipLaunchKernelGGL(direct_all_gather_kernel, dim3(grid_for(a.n, a.world)), dim3(256), 0, stream, a);
    PRISM_CHECK_LAUNCH();
    if (use_flags && (rc = signal_wait(a, 3u, stream))) return rc;          // nobody still reads this rank's buffer
    return PRISM_OK;
}

// ---- set-up helpers (host, synchronous) ------------------------------------------------------------------------------
#define PRISM_CHECK_HIP(call)                                                                  \
    do {                                                                                       \
        hipError_t e__ = (call);                                                               \
        if (e__ != hipSuccess) {                                                               \
            ::prism::set_error("%s: %s failed: %s", __func__, #call, hipGetErrorString(e__)); \
            return PRISM_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

static_assert(sizeof(hipIpcMemHandle_t) == PRISM_IPC_HANDLE_BYTES, "IPC handle size of the ABI");

extern "C" int prism_direct_flags_alloc(uint32_t **flags_out, void *ipc_handle_out) {
    PRISM_CHECK_ARG(flags_out != nullptr, "null out pointer");
    void *p = nullptr;
    // uncached: every access of every agent goes to memory -- what a word that peers store into while a kernel polls it needs
    hipError_t e = hipExtMallocWithFlags(&p, PRISM_DIRECT_FLAG_WORDS * sizeof(uint32_t), hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        PRISM_CHECK_HIP(hipExtMallocWithFlags(&p, PRISM_DIRECT_FLAG_WORDS * sizeof(uint32_t), hipDeviceMallocFinegrained));
    }
    PRISM_CHECK_HIP(hipMemset(p, 0, PRISM_DIRECT_FLAG_WORDS * sizeof(uint32_t)));
    PRISM_CHECK_HIP(hipDeviceSynchronize());
    if (ipc_handle_out) {
        hipIpcMemHandle_t h;
        e = hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) {
            (void)hipFree(p);
            set_error("%s: hipIpcGetMemHandle failed: %s", __func__, hipGetErrorString(e));
            return PRISM_ERR_HIP;
        }
        memcpy(ipc_handle_out, &h, sizeof(h));
    }
    *flags_out = (uint32_t *)p;
    return PRISM_OK;
}

extern "C" int prism_direct_flags_free(uint32_t *flags) {
    if (flags) PRISM_CHECK_HIP(hipFree(flags));
    return PRISM_OK;
}

extern "C" int prism_direct_flags_open(const void *ipc_handle, uint32_t **flags_out) {
    PRISM_CHECK_ARG(ipc_handle != nullptr && flags_out != nullptr, "null argument");
    hipIpcMemHandle_t h;
    memcpy(&h, ipc_handle, sizeof(h));
    void *p = nullptr;
    PRISM_CHECK_HIP(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
    *flags_out = (uint32_t *)p;
    return PRISM_OK;
}

extern "C" int prism_direct_flags_close(uint32_t *flags) {
    if (flags) PRISM_CHECK_HIP(hipIpcCloseMemHandle(flags));
    return PRISM_OK;
}

extern "C" int prism_direct_enable_peer(int32_t peer_device) {
    int cur = -1;
    PRISM_CHECK_HIP(hipGetDevice(&cur));
    if (cur == peer_device) return PRISM_OK;
    int can = 0;
    PRISM_CHECK_HIP(hipDeviceCanAccessPeer(&can, cur, peer_device));
    if (!can) {
        set_error("%s: device %d has no peer path to device %d", __func__, cur, (int)peer_device);
        return PRISM_ERR_HIP;
    }
    const hipError_t e = hipDeviceEnablePeerAccess(peer_device, 0);
    if (e == hipErrorPeerAccessAlreadyEnabled) {
        (void)hipGetLastError();
        return PRISM_OK;
    }
    PRISM_CHECK_HIP(e);
    return PRISM_OK;
}

extern "C" int prism_direct_flags_read(const uint32_t *flags, uint32_t *host_out, prism_stream_t stream_) {
    PRISM_CHECK_ARG(flags != nullptr && host_out != nullptr, "null argument");
    hipStream_t stream = (hipStream_t)stream_;
    PRISM_CHECK_HIP(hipMemcpyAsync(host_out, flags, PRISM_DIRECT_FLAG_WORDS * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    PRISM_CHECK_HIP(hipStreamSynchronize(stream));
    return PRISM_OK;
}
