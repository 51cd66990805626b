// Microbenchmark behind the pair-cooperative forward tile: is the forward bound by the BYTES each CU
// streams out of the L2s, and what does a two-workgroup exchange of trunk partials cost?
//
// The real forward (fwd_tile_kernel<128, true, true>): 256 workgroups x 8 waves, each wave streams 144 KB of pre-split
// bf16 weight pieces (48 operand groups of 3 planes x 1 KB) straight into MFMA A operands and issues 6 MFMAs (16x16x32
// bf16) per group for ITS 16 rows: 1.18 MB per CU per tile, 288 MFMAs per wave.
//   MODE 0  that shape: every workgroup streams all 384 groups (48 per wave), 6 MFMAs per group
//   MODE 1  pair shape: a workgroup streams HALF the columns (24 groups per wave) for 32 rows = 12 MFMAs per group: the
//           same 288 MFMAs per wave, half the bytes per CU
//   MODE 2  MODE 1 + the exchange: each workgroup folds its 32 x 128 partials through LDS, publishes the partner's 16 rows
//           (8 KB + 256 B of row statistics, 16-byte write-through stores, one flag), polls the partner's flag and reads its
//           16 rows back with agent-scope loads -- the protocol of cdna_hip_programming.md Guideline 16 (R1, sc1 both sides)
//   MODE 3  MODE 0 with no MFMAs (the stream alone), MODE 4: MODE 1 with no MFMAs
// Reported: event time per launch, in-kernel cycles of the stream phase and of the exchange (median / max over workgroups).
// Build: hipcc --offload-arch=gfx950 -O3 pair_stream.hip -o pair_stream.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mbf(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

constexpr int GROUP_U4 = 3 * 64;          // u32x4 per operand group (3 planes x 64 lanes)
constexpr int RING = 4;

template <int MODE>
__global__ __launch_bounds__(512) void pair_kernel(const u32x4 *__restrict__ w, float *xch, unsigned int *flags, unsigned int epoch,
                                                   float *out, unsigned long long *cyc) {
    constexpr bool PAIR = MODE == 1 || MODE == 2 || MODE == 4, MFMA = MODE < 3, XCH = MODE == 2;
    constexpr int NG = PAIR ? 24 : 48, RT = PAIR ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float part[8 * 32 * 132];       // [wave][row][H + 4]
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int half = PAIR ? (blockIdx.x & 1) : 0;
    typedef const u32x4 __attribute__((address_space(1))) *gcu4;
    // wave wv of half `half` streams groups [(half * 8 + wv) * NG, +NG) of the 384-group (1.18 MB) buffer
    const gcu4 wp = (gcu4)w + (size_t)((PAIR ? half * 8 + wv : wv) * NG) * GROUP_U4 + lane;
    f32x4 acc[RT][8];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[r][i] = f32x4{0, 0, 0, 0};
    u32x4 xb[RT][3];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int p = 0; p < 3; ++p) xb[r][p] = u32x4{0x3f803f80u + tid + r, 0x3f003f00u, 0x3e803e80u + p, 0x3f803f80u};
    u32x4 ring[RING][3];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int q = 0; q < RING; ++q)
#pragma unroll
        for (int p = 0; p < 3; ++p) ring[q][p] = wp[(q * 3 + p) * 64];
    u32x4 sink = {0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < NG; ++q) {
        const int sl = q % RING;
        if (MFMA) {
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                f32x4 a = acc[r][q & 7];
                a = mbf(ring[sl][2], xb[r][0], a);
                a = mbf(ring[sl][1], xb[r][1], a);
                a = mbf(ring[sl][0], xb[r][2], a);
                a = mbf(ring[sl][1], xb[r][0], a);
                a = mbf(ring[sl][0], xb[r][1], a);
                a = mbf(ring[sl][0], xb[r][0], a);
                acc[r][q & 7] = a;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) sink ^= ring[sl][p];
        }
        if (q + RING < NG) {
#pragma unroll
            for (int p = 0; p < 3; ++p) ring[sl][p] = wp[((q + RING) * 3 + p) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    // partials -> LDS (as the real kernel's fold)
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int ht = 0; ht < 8; ++ht) {
            f32x4 v = acc[r][ht];
            if (!MFMA) v[0] += __uint_as_float(sink[0] ^ sink[1] ^ sink[2] ^ sink[3]);
            *reinterpret_cast<f32x4 *>(&part[(wv * 32 + 16 * r + li) * 132 + 16 * ht + 4 * g]) = v;
        }
    __syncthreads();
    // fold the eight waves: thread = (row fm of 16 x RT, 4 hidden units)
    const int fm = tid >> 5, fc = tid & 31;
    f32x4 s[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        s[r] = *reinterpret_cast<const f32x4 *>(&part[(16 * r + fm) * 132 + 4 * fc]);
#pragma unroll
        for (int ww = 1; ww < 8; ++ww) s[r] += *reinterpret_cast<const f32x4 *>(&part[(ww * 32 + 16 * r + fm) * 132 + 4 * fc]);
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    f32x4 total = s[0];
    if (XCH) {
        // the partner finalises row tile (1 - half): publish that tile's folded partials, keep tile `half`
        const int pair = blockIdx.x >> 1;
        float *mine = xch + ((size_t)pair * 2 + half) * (16 * 128 + 64);          // what THIS workgroup publishes
        const float *theirs = xch + ((size_t)pair * 2 + (1 - half)) * (16 * 128 + 64);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(mine, 0, (16 * 128 + 64) * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(theirs), 0, (16 * 128 + 64) * 4, 0x00020000);
        const f32x4 send = half == 0 ? s[RT - 1] : s[0];
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, send), rs_out, (fm * 128 + 4 * fc) * 4, 0, 16);      // aux 16 = sc1
        if (tid < 16) __builtin_amdgcn_raw_buffer_store_b128(u32x4{1u, 2u, 3u, (unsigned)tid}, rs_out, (16 * 128 + 4 * tid) * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned int *fl_mine = flags + (size_t)pair * 2 + half, *fl_theirs = flags + (size_t)pair * 2 + (1 - half);
        if (tid == 0) {
            __hip_atomic_store(fl_mine, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned int spins = 0;
            while (__hip_atomic_load(fl_theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 4000000u) break;      // bounded
            }
        }
        __syncthreads();
        const u32x4 got = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (fm * 128 + 4 * fc) * 4, 0, 16);
        u32x4 st = {0, 0, 0, 0};
        if (tid < 16) st = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (16 * 128 + 4 * tid) * 4, 0, 16);
        total = (half == 0 ? s[0] : s[RT - 1]) + __builtin_bit_cast(f32x4, got);
        total[0] += (float)st[0];
    }
    const unsigned long long t3 = __builtin_amdgcn_s_memtime();
    if (total[0] + total[1] + total[2] + total[3] == 12345.678f) out[tid] = total[0];
    if (tid == 0) {
        cyc[blockIdx.x * 4 + 0] = t1 - t0;
        cyc[blockIdx.x * 4 + 1] = t2 - t1;
        cyc[blockIdx.x * 4 + 2] = t3 - t2;
        cyc[blockIdx.x * 4 + 3] = t3 - t0;
    }
}

template <int MODE>
void run(const char *name, const u32x4 *w, float *xch, unsigned int *flags, float *out, unsigned long long *cyc) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    unsigned int epoch = 1;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((pair_kernel<MODE>), dim3(256), dim3(512), 0, 0, w, xch, flags, epoch++, out, cyc);
    hipEventRecord(a);
    const int reps = 50;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((pair_kernel<MODE>), dim3(256), dim3(512), 0, 0, w, xch, flags, epoch++, out, cyc);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    std::vector<unsigned long long> h(256 * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double med[4], mx[4];
    for (int k = 0; k < 4; ++k) {
        std::vector<unsigned long long> v;
        for (int bI = 0; bI < 256; ++bI) v.push_back(h[bI * 4 + k]);
        std::sort(v.begin(), v.end());
        med[k] = (double)v[128];
        mx[k] = (double)v[255];
    }
    printf("%-34s %7.2f us/launch | stream med %6.0f max %6.0f | fold %5.0f | exchange med %6.0f max %6.0f | total med %6.0f max %6.0f clk\n",
           name, ms / reps * 1e3, med[0], mx[0], med[1], med[2], mx[2], med[3], mx[3]);
}

int main() {
    const size_t n_u4 = (size_t)384 * GROUP_U4;      // 1.18 MB
    u32x4 *w;
    float *xch, *out;
    unsigned int *flags;
    unsigned long long *cyc;
    hipMalloc(&w, n_u4 * 16);
    hipMalloc(&xch, 256 * (16 * 128 + 64) * 4);
    hipMalloc(&flags, 256 * 4);
    hipMalloc(&out, 4096);
    hipMalloc(&cyc, 256 * 4 * 8);
    std::vector<unsigned int> hw(n_u4 * 4);
    for (auto &x : hw) x = 0x3f803f80u ^ (unsigned)(rand() & 0x007f007f);       // random bf16 pairs near 1
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMemset(flags, 0, 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>("16-row tile, all columns", w, xch, flags, out, cyc);
        run<1>("32-row pair, half the columns", w, xch, flags, out, cyc);
        run<2>("32-row pair + exchange", w, xch, flags, out, cyc);
        run<3>("stream only, all columns", w, xch, flags, out, cyc);
        run<4>("stream only, half the columns", w, xch, flags, out, cyc);
    }
    return 0;
}
