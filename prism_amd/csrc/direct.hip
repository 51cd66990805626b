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
// reached it (bounded: gives up after DIRECT_WAIT_TICKS and sets the sticky error slot).
#include <string.h>

#include "common.h"

namespace prism {

constexpr unsigned long long DIRECT_WAIT_TICKS = 200000000ull;      // 2 s of the 100 MHz counter

struct DirectArgs {
    float *bufs[PRISM_MAX_PEERS];
    unsigned int *flags[PRISM_MAX_PEERS];
    int world, rank;
    long long n;
    unsigned int phase;
};

// float4 range [lo, hi) of slice s; the n & 3 trailing floats belong to the last slice (handled by its owner as scalars)
__device__ __forceinline__ void slice_range(long long n, int world, int s, long long &lo, long long &hi) {
    const long long n4 = n >> 2, per = (n4 + world - 1) / world;
    lo = per * s < n4 ? per * s : n4;
    hi = per * (s + 1) < n4 ? per * (s + 1) : n4;
}

__global__ __launch_bounds__(256) void direct_reduce_scatter_kernel(DirectArgs a) {
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
// announced is 3 E + a.phase: a captured hipGraph replays with fresh phase numbers.
__global__ __launch_bounds__(64) void direct_signal_wait_kernel(DirectArgs a_in) {
    DirectArgs a = a_in;
    const int t = threadIdx.x;
    const unsigned int E = a.flags[a.rank][PRISM_MAX_PEERS + 1];
    a.phase = 3u * E + a_in.phase;
    __builtin_amdgcn_s_barrier();                             // (every lane has read E before lane 0 may advance it)
    if (a_in.phase == 3u && t == 0) a.flags[a.rank][PRISM_MAX_PEERS + 1] = E + 1u;
    if (t < a.world) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");       // system scope: what this rank wrote is visible to the peers first
        __hip_atomic_store(a.flags[t] + a.rank, a.phase, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        // (phase numbers only grow; compared as a signed difference so that the counter may wrap)
        while ((int)(__hip_atomic_load(a.flags[a.rank] + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - a.phase) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (__builtin_amdgcn_s_memrealtime() - t0 > DIRECT_WAIT_TICKS) {
                atomicOr(a.flags[a.rank] + PRISM_MAX_PEERS, 1u);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
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
    return PRISM_OK;
}

static int grid_for(long long n, int world) {
    const long long per = ((n >> 2) + world - 1) / world;
    long long b = (per + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 512 ? 512 : b));
}

}  // namespace prism

using namespace prism;

static int signal_wait(DirectArgs &a, unsigned int phase, hipStream_t stream) {
    a.phase = phase;
    hipLaunchKernelGGL(direct_signal_wait_kernel, dim3(1), dim3(64), 0, stream, a);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
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
