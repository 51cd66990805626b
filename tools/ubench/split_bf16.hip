// Microbenchmark: a fp32 product done as SIX bf16 MFMAs -- each fp32 operand split into three bf16
// pieces (hi + mid + lo = the fp32 value, 8 + 8 + 8 significant bits), the six piece products of weight >= 2^-16 kept,
// fp32 accumulation inside v_mfma_f32_16x16x32_bf16 -- against the exact fp32 chain v_mfma_f32_16x16x4_f32, at EQUAL MATH:
//   rate      one 32-column step of the forward tile's trunk product (H = 128: eight 16 x 16 accumulators, K = 32):
//             64 fp32 MFMAs (32 cycles each) vs 48 bf16 MFMAs (16 cycles each) + the VALU that splits the B operand on the
//             fly (the A operand = weights would come pre-split from the packing role); 1 and 2 waves per SIMD
//   accuracy  pre[h][m] = sum_n W[h][n] x[n][m], K = 1024 (the trunk product of one tile), both ways, against float64
// Build: hipcc --offload-arch=gfx950 -O3 split_bf16.hip -o split_bf16.bin
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mf32(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mbf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned int pk(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
// split eight fp32 values into three packed-bf16 fragments (4 registers each): x = hi + mid + lo (exactly, up to the last bit)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 &hi, u32x4 &mid, u32x4 &lo) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned int h = pk(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned int m = pk(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        hi[p] = h;
        mid[p] = m;
        lo[p] = pk(sa, sb);
    }
}

// ---------------------------------------------------------------------------------------------- rate
template <int MODE>      // 0: fp32 chain, 1: split-bf16 (operands pre-split), 2: split-bf16 + on-the-fly split of the B operand
__global__ __launch_bounds__(512) void rate_kernel(float *out, unsigned long long *cyc, int iters) {
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    const float a0 = threadIdx.x * 1e-3f + 0.1f, b0 = 1.0f + threadIdx.x * 1e-4f;
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = a0 + i * b0;
    u32x4 wh[8], wm[8], wl[8];       // pre-split A operands of the 8 accumulators (in the real kernel: streamed)
    for (int i = 0; i < 8; ++i) {
        float t[8];
        for (int j = 0; j < 8; ++j) t[j] = b0 * (i + 1) + j * a0;
        split8(t, wh[i], wm[i], wl[i]);
    }
    u32x4 xh, xm, xl;
    split8(x, xh, xm, xl);
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = mf32(__uint_as_float(wh[i][k & 3]) + k, x[k], acc[i]);
        } else {
            if (MODE == 2) {
                split8(x, xh, xm, xl);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] += 1.0f;       // (new values every step, as the phi epilogue would hand over)
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                // six products: hh | hm mh | hl mm lh   (small terms first would be the more accurate order; the MFMA adds them
                // to a running fp32 accumulator either way)
                acc[i] = mbf(wl[i], xh, acc[i]);
                acc[i] = mbf(wm[i], xm, acc[i]);
                acc[i] = mbf(wh[i], xl, acc[i]);
                acc[i] = mbf(wm[i], xh, acc[i]);
                acc[i] = mbf(wh[i], xm, acc[i]);
                acc[i] = mbf(wh[i], xh, acc[i]);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1234.5f) out[0] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE>
static double run_rate(const char *name, int threads, float *out, unsigned long long *cyc) {
    const int iters = 64, blocks = 256;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256 * 8);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double av = 0;
    int n = 0;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < threads / 64; ++w) { av += (double)h[b * 8 + w]; ++n; }
    const double per_step = av / n / iters * (256.0 / threads) * (threads / 256);       // cycles per 32-column step per wave
    const int waves = threads / 256;
    printf("%-52s waves/SIMD=%d  %7.0f cycles per K=32 step per SIMD (fp32 floor 2048)\n", name, waves, per_step * waves);
    return per_step * waves;
}

// ------------------------------------------------------------------------------------------ accuracy
// one wave: pre[h][m], h < 16 (one accumulator tile), m < 16, K = 1024.  W [16][1024], x [1024][16] in global memory.
__global__ void acc_kernel(const float *W, const float *X, float *out32, float *outsp) {
    const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
    f32x4 a32 = {0, 0, 0, 0}, asp = {0, 0, 0, 0};
    for (int k0 = 0; k0 < 1024; k0 += 4) a32 = mf32(W[li * 1024 + k0 + g], X[(k0 + g) * 16 + li], a32);
    for (int k0 = 0; k0 < 1024; k0 += 32) {
        float wv[8], xv[8];
        for (int j = 0; j < 8; ++j) {
            wv[j] = W[li * 1024 + k0 + 8 * g + j];
            xv[j] = X[(k0 + 8 * g + j) * 16 + li];
        }
        u32x4 wh, wm, wl, xh, xm, xl;
        split8(wv, wh, wm, wl);
        split8(xv, xh, xm, xl);
        asp = mbf(wl, xh, asp);
        asp = mbf(wm, xm, asp);
        asp = mbf(wh, xl, asp);
        asp = mbf(wm, xh, asp);
        asp = mbf(wh, xm, asp);
        asp = mbf(wh, xh, asp);
    }
    for (int r = 0; r < 4; ++r) {
        out32[(4 * g + r) * 16 + li] = a32[r];
        outsp[(4 * g + r) * 16 + li] = asp[r];
    }
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 64);
    hipMalloc(&cyc, 256 * 8 * 8);
    double r[6];
    r[0] = run_rate<0>("fp32 16x16x4 chain (64 MFMAs per step)", 256, out, cyc);
    r[1] = run_rate<0>("fp32 16x16x4 chain (64 MFMAs per step)", 512, out, cyc);
    r[2] = run_rate<1>("split bf16 16x16x32 x 6 (48 MFMAs per step)", 256, out, cyc);
    r[3] = run_rate<1>("split bf16 16x16x32 x 6 (48 MFMAs per step)", 512, out, cyc);
    r[4] = run_rate<2>("split bf16 x 6 + splitting the B operand (VALU)", 256, out, cyc);
    r[5] = run_rate<2>("split bf16 x 6 + splitting the B operand (VALU)", 512, out, cyc);
    printf("speed-up of the split form at equal math: %.2fx (pre-split), %.2fx (with the on-the-fly split), 2 waves per SIMD\n",
           r[1] / r[3], r[1] / r[5]);

    // accuracy on trunk-shaped data: W ~ U(-1/32, 1/32) (Linear(1024) init), x = ReLU-like non-negative activations
    srand(7);
    for (int trial = 0; trial < 3; ++trial) {
        std::vector<float> W(16 * 1024), X(1024 * 16);
        const float xs = trial == 0 ? 1.f : (trial == 1 ? 30.f : 0.01f);
        for (auto &v : W) v = ((rand() / (float)RAND_MAX) * 2.f - 1.f) / 32.f;
        for (auto &v : X) { float u = (rand() / (float)RAND_MAX) * 2.f - 0.6f; v = u > 0 ? u * xs : 0.f; }
        float *dW, *dX, *d32, *dsp;
        hipMalloc(&dW, W.size() * 4); hipMalloc(&dX, X.size() * 4); hipMalloc(&d32, 1024); hipMalloc(&dsp, 1024);
        hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, dW, dX, d32, dsp);
        float o32[256], osp[256];
        hipMemcpy(o32, d32, 1024, hipMemcpyDeviceToHost);
        hipMemcpy(osp, dsp, 1024, hipMemcpyDeviceToHost);
        double e32 = 0, esp = 0, r32 = 0, rsp = 0, mag = 0;
        for (int h = 0; h < 16; ++h)
            for (int m = 0; m < 16; ++m) {
                double ref = 0;
                for (int k = 0; k < 1024; ++k) ref += (double)W[h * 1024 + k] * (double)X[k * 16 + m];
                const double d1 = fabs(o32[h * 16 + m] - ref), d2 = fabs(osp[h * 16 + m] - ref);
                e32 = fmax(e32, d1); esp = fmax(esp, d2);
                r32 += d1 * d1; rsp += d2 * d2; mag = fmax(mag, fabs(ref));
            }
        printf("accuracy (K = 1024, x scale %g, max |pre| %.3g): fp32 MFMA max err %.3e rms %.3e | split-bf16 x 6 max err %.3e rms %.3e\n",
               xs, mag, e32, sqrt(r32 / 256), esp, sqrt(rsp / 256));
    }
    return 0;
}
