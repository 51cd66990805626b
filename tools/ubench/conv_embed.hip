// Microbenchmark: the observation convolution of the front launch (conv_embed_rows, iqn_kernels.h) in isolation --
// 256 workgroups of 256 threads, operands already in LDS, s_memtime around a FIRST call and around a SECOND call of the
// same inlined code placed in a loop (so the second pass runs from a warm instruction cache), plus an empty stamp pair.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../prism_amd/csrc conv_embed.hip -o conv_embed.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#include "common.h"
#include "iqn_kernels.h"
namespace prism { __device__ void embed_extra_block(const IqnArgs &, int, float *) {} }
using namespace prism;

__global__ __launch_bounds__(256) void probe(const float *obs, const float *w, float *dst, unsigned long long *out, int C, int passes) {
    __shared__ __attribute__((aligned(16))) float s_obs[1000];
    __shared__ __attribute__((aligned(16))) float s_w[CONV_W_FLOATS];
    __shared__ float s_b[16];
    const int tid = threadIdx.x;
    if (tid < 25 * C) obs_to_lds(s_obs, reinterpret_cast<const float4 *>(obs + (size_t)blockIdx.x * 100 * C)[tid], tid, C);
    for (int i = tid; i < 16 * conv_kp(C); i += 256) s_w[i] = w[i];
    if (tid < 16) s_b[tid] = w[2000 + tid];
    __syncthreads();
    unsigned long long t[8];
    t[0] = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t[1] = __builtin_amdgcn_s_memtime();
    for (int p = 0; p < passes; ++p) {          // (run-time trip count: one copy of the code)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t[2 + 2 * (p & 1)] = __builtin_amdgcn_s_memtime();
        conv_embed_rows(s_obs, s_w, s_b, C, dst + ((size_t)blockIdx.x * 2 + (p & 1)) * 1024, tid);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t[3 + 2 * (p & 1)] = __builtin_amdgcn_s_memtime();
        __syncthreads();
    }
    if (tid == 0)
        for (int i = 0; i < 6; ++i) out[(size_t)blockIdx.x * 8 + i] = t[i];
}

int main() {
    const int NB = 256, C = 4;
    float *obs, *w, *dst;
    unsigned long long *out;
    hipMalloc(&obs, NB * 400 * sizeof(float));
    hipMalloc(&w, 4096 * sizeof(float));
    hipMalloc(&dst, NB * 2048 * sizeof(float));
    hipMalloc(&out, NB * 8 * sizeof(unsigned long long));
    hipMemset(obs, 0, NB * 400 * sizeof(float));
    hipMemset(w, 0, 4096 * sizeof(float));
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(NB), dim3(256), 0, 0, obs, w, dst, out, C, 2);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(NB * 8);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<long> e, c1, c2;
        for (int b = 0; b < NB; ++b) {
            e.push_back((long)(h[b * 8 + 1] - h[b * 8 + 0]));
            c1.push_back((long)(h[b * 8 + 3] - h[b * 8 + 2]));
            c2.push_back((long)(h[b * 8 + 5] - h[b * 8 + 4]));
        }
        auto med = [](std::vector<long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("launch %d: empty stamp pair %ld, first conv %ld, second conv (same code, warm) %ld ticks (median of %d workgroups)\n", rep, med(e),
               med(c1), med(c2), NB);
    }
    return 0;
}
