// Diagnostic: phase timing of the single-workgroup priority writer (GPU box only).
//   hipcc --offload-arch=gfx950 -O3 -I include -I prism_amd/csrc tools/ubench/tree_write.hip -o /tmp/tw && /tmp/tw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef STAMP_MASK
#define STAMP_MASK 0xffffffffffull
#endif
__device__ unsigned long long *g_stamp;
#define TREE_STAMP(k) do { if (threadIdx.x == 0 && g_stamp && (STAMP_MASK >> (k) & 1)) g_stamp[k] = __builtin_amdgcn_s_memtime(); } while (0)
#include "common.h"
#include "replay_kernels.h"
using namespace prism;
__global__ void set_stamp(unsigned long long *p) { g_stamp = p; }
int main() {
    const int64_t capacity = 100000, cap = 131072;
    const int B = 256;
    prism_replay_desc rp{};
    rp.capacity = capacity; rp.tree_capacity = cap;
    float *tree, *state;
    hipMalloc(&tree, 2 * cap * 8); hipMalloc(&state, 64);
    hipMemset(tree, 0, 2 * cap * 8); hipMemset(state, 0, 64);
    rp.tree = tree; rp.per_state = state;
    std::vector<int64_t> hidx(B); std::vector<float> hp(B);
    for (int i = 0; i < B; ++i) { hidx[i] = rand() % capacity; hp[i] = (rand() % 1000) / 1000.f; }
    int64_t *idx; float *pr; unsigned long long *st;
    hipMalloc(&idx, B * 8); hipMalloc(&pr, B * 4); hipMalloc(&st, 64 * 8); hipMemset(st, 0, 64 * 8);
    hipMemcpy(idx, hidx.data(), B * 8, hipMemcpyHostToDevice); hipMemcpy(pr, hp.data(), B * 4, hipMemcpyHostToDevice);
    set_stamp<<<1, 1>>>(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 1024}) for (int dense = 0; dense < 2; ++dense) {
        for (int it = 0; it < 5; ++it) (dense ? per_update_kernel<true> : per_update_kernel<false>)<<<1, threads>>>(rp, idx, pr, B, 0.5f, 1e-8f, 1);
        hipEventRecord(e0);
        for (int it = 0; it < 100; ++it) (dense ? per_update_kernel<true> : per_update_kernel<false>)<<<1, threads>>>(rp, idx, pr, B, 0.5f, 1e-8f, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[64]; hipMemcpy(h, st, sizeof h, hipMemcpyDeviceToHost);
        printf("threads %4d dense %d: %.2f us/launch; stamps:", threads, dense, ms * 10.f);
        for (int k = 1; k < 40; ++k) if (h[k]) printf(" %d:%llu", k, h[k] - h[0]);
        printf("\n");
    }
    return 0;
}
