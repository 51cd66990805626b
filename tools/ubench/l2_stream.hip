// Microbenchmark: every workgroup streams the SAME weight buffer (768 KB) from L2 into registers.
// Variants: (0) fully coalesced 1 KB per wave instruction; (1) MFMA-B-fragment shape: 16 rows x 64 B;
// (2) 8 rows x 128 B.  Reports bytes/clk/CU.  Build: hipcc --offload-arch=gfx950 -O3 l2_stream.hip -o l2_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void stream_kernel(const float *__restrict__ w, float *out, int n_floats,
                                                           unsigned long long *cyc) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    float4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // buffer viewed as rows of 1024 floats (4 KB); total rows = n_floats / 1024
    const int rows = n_floats / 1024;
    if (MODE == 0) {
        for (int i = (wv * 64 + lane) * 4; i < n_floats; i += WAVES * 64 * 4) {
            float4 v = *reinterpret_cast<const float4 *>(w + i);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    } else if (MODE == 3) {
        // coalesced, but every workgroup of an XCD starts at a different offset (no lockstep on identical lines)
        const int chunks = n_floats / (WAVES * 64 * 4);
        const int rot = (int)((blockIdx.x / 8) * 7919u % chunks);
        for (int c = 0; c < chunks; ++c) {
            int cc = c + rot;
            if (cc >= chunks) cc -= chunks;
            float4 v = *reinterpret_cast<const float4 *>(w + (size_t)cc * WAVES * 64 * 4 + (wv * 64 + lane) * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    } else if (MODE == 4) {
        // coalesced, unrolled x32: as many loads in flight per wave as the tile kernels keep
        for (int i = (wv * 64 + lane) * 4; i < n_floats; i += WAVES * 64 * 4 * 32) {
            float4 v[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) v[u] = (i + u * WAVES * 64 * 4 < n_floats) ? *reinterpret_cast<const float4 *>(w + i + u * WAVES * 64 * 4) : make_float4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < 32; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    } else if (MODE == 1) {
        // wave wv owns row tiles wv, wv+WAVES, ...; per tile 16 rows; per q a 64-B piece per row
        for (int rt = wv; rt < rows / 16; rt += WAVES) {
            const float *base = w + (size_t)(rt * 16 + li) * 1024 + 4 * g;
#pragma unroll 8
            for (int q = 0; q < 64; ++q) {
                float4 v = *reinterpret_cast<const float4 *>(base + 16 * q);
                acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
        }
    } else {
        for (int rt = wv; rt < rows / 16; rt += WAVES) {
            const float *base = w + (size_t)(rt * 16 + (li & 7)) * 1024 + 16 * (li >> 3) + 4 * g;
#pragma unroll 8
            for (int c = 0; c < 32; ++c) {
                float4 v = *reinterpret_cast<const float4 *>(base + 32 * c);
                float4 u = *reinterpret_cast<const float4 *>(base + 8 * 1024 + 32 * c);
                acc.x += v.x + u.x; acc.y += v.y + u.y; acc.z += v.z + u.z; acc.w += v.w + u.w;
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int WAVES>
void run(const char *name, const float *w, float *out, int n, unsigned long long *cyc, int blocks) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream_kernel<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, w, out, n, cyc);
    hipEventRecord(a);
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream_kernel<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, w, out, n, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto x : h) avg += x;
    avg /= blocks;
    const double bytes = (double)n * 4;
    printf("%-44s blocks=%4d waves=%d  %7.2f us/launch  in-kernel %8.0f clk  -> %6.1f B/clk/CU   aggregate %6.2f TB/s\n", name, blocks,
           WAVES, ms / reps * 1e3, avg, bytes / avg, bytes * blocks / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const int n = 192 * 1024;   // 768 KB
    float *w, *out;
    unsigned long long *cyc;
    hipMalloc(&w, n * 4);
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 4096 * 8);
    hipMemset(w, 0, n * 4);
    run<0, 8>("coalesced 1 KB/instr", w, out, n, cyc, 256);
    run<3, 8>("coalesced, rotated start per workgroup", w, out, n, cyc, 256);
    run<4, 8>("coalesced, 32 loads in flight per lane", w, out, n, cyc, 256);
    run<1, 8>("fragment 16 rows x 64 B", w, out, n, cyc, 256);
    run<2, 8>("fragment 8 rows x 128 B", w, out, n, cyc, 256);
    run<0, 4>("coalesced 1 KB/instr", w, out, n, cyc, 256);
    run<1, 4>("fragment 16 rows x 64 B", w, out, n, cyc, 256);
    run<0, 8>("coalesced, 512 blocks", w, out, n, cyc, 512);
    run<1, 8>("fragment 16x64, 512 blocks", w, out, n, cyc, 512);
    run<0, 8>("coalesced, 32 blocks", w, out, n, cyc, 32);
    run<0, 16>("coalesced 16 waves", w, out, n, cyc, 256);
    return 0;
}
