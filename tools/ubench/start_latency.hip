// Microbenchmark: what the first memory accesses of a freshly launched kernel cost on MI355X, in shader-clock ticks.
// A flush kernel (touches 512 MB) runs before every measured launch; the measured kernel (one wave per workgroup)
// stamps s_memtime at entry (before any memory operation), after its kernarg-dependent first load (cold), after a
// dependent load from another 2 MB region (cold), after a load of the neighbouring cache line of the first (L2 / MALL),
// after an LDS round trip, after a release fence + atomic + acquire (the cost of one grid-barrier arrival), and after
// a second arrival-style atomic.  Build: hipcc --offload-arch=gfx950 -O3 start_latency.hip -o start_latency.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ void flush_kernel(float *buf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = buf[i] * 1.0001f + 1.0f;
}

struct Big {
    float pad[200];     // a fat by-value argument, like the learner's launch descriptors
};

__global__ __launch_bounds__(64) void probe(const int *chain, float *sink, unsigned long long *out, unsigned int *ctr, Big big) {
    __shared__ int s_x[64];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const int lane = threadIdx.x;
    const int *p = chain + (size_t)blockIdx.x * 1024;            // each workgroup its own 4 KB
    int a = p[0];                                                // cold line
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int b = chain[(size_t)a + (size_t)blockIdx.x * 1024];        // dependent, another region (a = offset in ints)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    int c = p[32 + (b & 1)];                                     // neighbouring line of the first one
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    s_x[lane] = c + lane;
    __syncthreads();
    int d = s_x[(lane + 1) & 63];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t4 = __builtin_amdgcn_s_memtime();
    sink[(size_t)blockIdx.x * 64 + lane] = (float)d + big.pad[lane];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    unsigned int old = 0;
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t6 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        old = atomicAdd(ctr, 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t7 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t8 = __builtin_amdgcn_s_memtime();
    unsigned int v = 0;
    if (lane == 0) {
        v = __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t9 = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
        unsigned long long *o = out + (size_t)blockIdx.x * 16;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4;
        o[5] = t6 - t5; o[6] = t7 - t6; o[7] = t8 - t7; o[8] = t9 - t8; o[9] = old + v;
    }
}

int main() {
    const int blocks = 256, reps = 20;
    const size_t fl = 128u << 20;        // floats (512 MB)
    float *flush, *sink;
    int *chain;
    unsigned long long *out;
    unsigned int *ctr;
    hipMalloc(&flush, fl * 4);
    hipMalloc(&sink, blocks * 64 * 4);
    hipMalloc(&chain, (size_t)64 << 20);
    hipMalloc(&out, blocks * 16 * 8);
    hipMalloc(&ctr, 4);
    hipMemset(flush, 0, fl * 4);
    hipMemset(ctr, 0, 4);
    std::vector<int> h((size_t)16 << 20, 0);
    for (int b = 0; b < blocks; ++b) h[(size_t)b * 1024] = (8 << 20) + 4096 * (b % 64);      // offset (ints): +32 MB, its own page
    hipMemcpy(chain, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    Big big;
    for (int i = 0; i < 200; ++i) big.pad[i] = 0.f;
    const char *names[9] = {"entry -> first load (kernarg + cold line)", "dependent cold load (another region)",
                            "neighbouring line of the first", "LDS store + barrier + load", "store + drain",
                            "release fence (L2 write-back)", "atomicAdd round trip", "acquire fence (invalidate)",
                            "agent-scope load round trip"};
    std::vector<std::vector<unsigned long long>> all(9);
    std::vector<unsigned long long> ho(blocks * 16);
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(flush_kernel, dim3(2048), dim3(256), 0, 0, flush, fl);
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, chain, sink, out, ctr, big);
        hipDeviceSynchronize();
        hipMemcpy(ho.data(), out, ho.size() * 8, hipMemcpyDeviceToHost);
        if (r < 2) continue;
        for (int k = 0; k < 9; ++k)
            for (int b = 0; b < blocks; ++b) all[k].push_back(ho[(size_t)b * 16 + k]);
    }
    for (int k = 0; k < 9; ++k) {
        std::sort(all[k].begin(), all[k].end());
        const size_t n = all[k].size();
        printf("%-44s median %6llu   p10 %6llu   p90 %6llu ticks\n", names[k], all[k][n / 2], all[k][n / 10], all[k][n * 9 / 10]);
    }
    return 0;
}
