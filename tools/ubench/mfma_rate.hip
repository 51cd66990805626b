// Microbenchmark: how fast does one SIMD retire v_mfma_f32_16x16x4_f32 when 1 or 2 waves feed it, with
// (a) independent accumulators only, (b) the forward tile's pattern (8 trunk accumulators + one phi chain threaded between),
// (c) the same plus the VALU / dummy work between them.  Reports shader cycles per MFMA per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mf(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[8], p = {0, 0, 0, 0};
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f, x[4] = {a, b, a + b, a - b};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int hp = 0; hp < 4; ++hp)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[2 * hp] = mf(a + r, x[r], acc[2 * hp]);
                if (MODE >= 1 && hp < 2) p = mf(b, a, p);
                if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
                acc[2 * hp + 1] = mf(b + r, x[r], acc[2 * hp + 1]);
                if (MODE >= 1 && hp < 2) p = mf(a, b, p);
                if (MODE >= 2 && hp == 2) { x[r] = fmaxf(p[r], 0.f) * a; b += x[r]; a = fmaf(x[r], x[r], a); }
                if (MODE >= 1) __builtin_amdgcn_sched_barrier(0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = p[0];
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1234.5f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
void run(const char *name, int threads, float *out, unsigned long long *cyc) {
    const int iters = 8, blocks = 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0, av = 0; int n = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < threads / 64; ++w) { double v = (double)h[b * 8 + w]; av += v; if (v > mx) mx = v; ++n; }
    const int per_wave = iters * (MODE >= 1 ? 48 : 32), waves_per_simd = threads / 256;
    printf("%-58s waves/SIMD=%d  wave cycles avg %8.0f max %8.0f -> %5.1f cycles per MFMA per SIMD\n", name, waves_per_simd, av / n, mx,
           (av / n) / (per_wave * waves_per_simd));
}
int main() {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 64); hipMalloc(&cyc, 256 * 8 * 8);
    run<0>("8 independent accumulators", 256, out, cyc);
    run<0>("8 independent accumulators", 512, out, cyc);
    run<1>("forward pattern: 8 trunk accumulators + phi chain", 256, out, cyc);
    run<1>("forward pattern: 8 trunk accumulators + phi chain", 512, out, cyc);
    run<2>("forward pattern + phi epilogue VALU", 256, out, cyc);
    run<2>("forward pattern + phi epilogue VALU", 512, out, cyc);
    return 0;
}
