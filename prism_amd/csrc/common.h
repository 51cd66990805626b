// Shared host/device helpers for libprism_hip (gfx950 only: wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "prism_hip.h"

namespace prism {

void set_error(const char *fmt, ...);

#define PRISM_CHECK_ARG(cond, msg)                            \
    do {                                                      \
        if (!(cond)) {                                        \
            ::prism::set_error("%s: %s", __func__, msg);      \
            return PRISM_ERR_INVALID;                         \
        }                                                     \
    } while (0)

#define PRISM_CHECK_LAUNCH()                                                        \
    do {                                                                            \
        hipError_t e__ = hipGetLastError();                                         \
        if (e__ != hipSuccess) {                                                    \
            ::prism::set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return PRISM_ERR_HIP;                                                   \
        }                                                                           \
    } while (0)

// ---- optional HIP-event instrumentation (profile.hip) ----
enum KernelId { K_EMBED = 0, K_TILE_FWD, K_LOSS, K_BWD, K_POST, K_FRONT, K_CLIP_ADAM, K_PER_SAMPLE, K_GATHER,
                K_PER_UPDATE, K_BACK, K_Q_FWD, K_Q_BWD, K_TAIL };
extern thread_local int g_profile_on;
void profile_begin(int kernel_id, hipStream_t stream);
void profile_end(int kernel_id, hipStream_t stream);
struct ProfileScope {
    int id;
    hipStream_t s;
    bool on;
    ProfileScope(int id_, hipStream_t s_) : id(id_), s(s_), on(g_profile_on != 0) {
        if (on) profile_begin(id, s);
    }
    ~ProfileScope() {
        if (on) profile_end(id, s);
    }
};

constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 16-byte streaming store: data the NEXT launch consumes (it comes from HBM there anyway: kernel boundaries
// write back and invalidate the L2s) goes out as the kernel runs instead of in the flush at its end
__device__ __forceinline__ void stream_store4(float4 *dst, const float4 &v) {
    __builtin_nontemporal_store(f32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4 *>(dst));
}

// Pull the kernel's whole argument segment into the scalar cache in ONE round trip.  The launch descriptors here are
// 0.5 - 1.5 KB passed by value; the compiler fetches their fields where it first needs them, so a kernel that branches
// on a field, then reads a pointer, then a size ... pays one scalar-cache miss (~0.4 us after a launch boundary)
// PER step of that chain before its first vector load is even issued (measured: 2 us from entry to the first role
// instruction of the post kernel).  One dword of every 64-byte line is requested back to back, into a register
// nobody reads; the later field loads hit.
template <int BYTES>
__device__ __forceinline__ void kernarg_prefetch() {
    const char __attribute__((address_space(4))) *p =
        (const char __attribute__((address_space(4))) *)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int LINES = (BYTES - 4) / 64 + 1, GROUPS = (LINES + 7) / 8;
    static_assert(GROUPS <= 4, "argument segment larger than 2 KB");
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define PRISM_KA_GROUP(T, G)                                                                                          \
    if constexpr (GROUPS > G) {                                                                                       \
        constexpr int n = LINES - 8 * G >= 8 ? 8 : LINES - 8 * G;                                                     \
        const char __attribute__((address_space(4))) *q = p + 512 * G;                                                \
        if constexpr (n == 8)                                                                                         \
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\t"    \
                         "s_load_dword %0, %1, 0xc0\n\ts_load_dword %0, %1, 0x100\n\ts_load_dword %0, %1, 0x140\n\t" \
                         "s_load_dword %0, %1, 0x180\n\ts_load_dword %0, %1, 0x1c0" : "=&s"(T) : "s"(q) : "memory");             \
        else if constexpr (n >= 4)                                                                                    \
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\t"    \
                         "s_load_dword %0, %1, 0xc0\n\ts_load_dword %0, %1, %2\n\ts_load_dword %0, %1, %3\n\t"             \
                         "s_load_dword %0, %1, %4"                                                                    \
                         : "=&s"(T) : "s"(q), "n"(64 * (n - 3)), "n"(64 * (n - 2)), "n"(64 * (n - 1)) : "memory");      \
        else                                                                                                          \
            asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, %2" : "=&s"(T) : "s"(q), "n"(64 * (n - 1)) : "memory");  \
    }
    PRISM_KA_GROUP(t0, 0)
    PRISM_KA_GROUP(t1, 1)
    PRISM_KA_GROUP(t2, 2)
    PRISM_KA_GROUP(t3, 3)
#undef PRISM_KA_GROUP
    // the registers stay allocated until every request has landed
    // ("memory": nothing the compiler counts on lgkmcnt -- LDS traffic -- may move in between: it does not know these loads)
    asm volatile("s_waitcnt lgkmcnt(0)" ::"s"(t0), "s"(t1), "s"(t2), "s"(t3) : "memory");
}

// ---- LDS carve-ups, checked at compile time ------------------------------------------------------------------------
// Every kernel that splits one LDS allocation into regions states them in ONE constexpr table -- offset, size (in the unit
// the kernel indexes by) and the phases of the kernel in which somebody may still read or write the region -- takes its
// pointers from that table, and static_asserts lds_layout_ok() on it: two regions that are live in a common phase must not
// overlap, and every region must lie inside the allocation.  (Round 1 shipped a writer whose ranking partials overlapped
// the keys they were computed from; it showed as a one-in-fifty wrong rank.  That table no longer compiles: see
// replay_kernels.h.)  The function is plain constexpr, so launch code checks run-time-sized layouts with it as well.
struct LdsRegion {
    int off, size;
    unsigned live;      // bit mask of phases
};
constexpr unsigned LDS_ALWAYS = ~0u;
constexpr bool lds_overlap(const LdsRegion &a, const LdsRegion &b) {
    return (a.live & b.live) != 0u && a.size > 0 && b.size > 0 && a.off < b.off + b.size && b.off < a.off + a.size;
}
template <int N>
constexpr bool lds_layout_ok(const LdsRegion (&r)[N], int total) {
    for (int i = 0; i < N; ++i) {
        if (r[i].off < 0 || r[i].size < 0 || r[i].off + r[i].size > total) return false;
        for (int j = i + 1; j < N; ++j)
            if (lds_overlap(r[i], r[j])) return false;
    }
    return true;
}

// D = A(16x4) * B(4x16) + C, exact fp32 FMA chain.  Lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; D[row = 4*(l>>4) + r][col = l&15] lands in element r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- fp32 products on the bf16 matrix pipe ---------------------------------------------------------------------------
// v_mfma_f32_16x16x32_bf16 retires 16x the flops of v_mfma_f32_16x16x4_f32 per cycle and, unlike it, leaves the SIMD's
// vector issue free for half of its 16 cycles.  A fp32 value is the exact sum of three bf16 pieces (hi + mid + lo, 8 + 8 + 8
// significant bits, each rounded to nearest from what the pieces before it left); of the nine piece products of a * b the six
// of weight >= 2^-16 are kept (hh | hm mh | hl mm lh), accumulated in fp32 inside the MFMA -- what is dropped is of the size
// of the rounding of one fp32 multiply.  Measured (tools/ubench/split_bf16.hip, profiles/r03_split_bf16_ubench.txt): K = 1024
// dot products of trunk-shaped data come out with the SAME rms error against float64 as the exact fp32 chain (1.57e-7 at
// |pre| ~ 1), at 2.5x its rate with pre-split operands, 2.05x when one operand is split on the fly.
// Operand maps (16x16x32): lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15], j = 0..7,
// as four registers of packed bf16 pairs; D as for 16x16x4: D[row 4 (l >> 4) + r][col l & 15].
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}
// two fp32 -> one register of two bf16, round to nearest even (v_cvt_pk_bf16_f32); `lo` lands in the low half
__device__ __forceinline__ unsigned int pack_bf16(float lo, float hi) {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned int, v);
}
struct Split3 {
    u32x4 hi, mid, lo;
};
// eight fp32 values -> their three bf16 pieces, element j in half (j & 1) of register j >> 1
__device__ __forceinline__ Split3 split_bf16x3(const float (&x)[8]) {
    Split3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float a = x[2 * p], b = x[2 * p + 1];
        const unsigned int h = pack_bf16(a, b);
        const float ra = a - __uint_as_float(h << 16), rb = b - __uint_as_float(h & 0xffff0000u);
        const unsigned int m = pack_bf16(ra, rb);
        const float sa = ra - __uint_as_float(m << 16), sb = rb - __uint_as_float(m & 0xffff0000u);
        s.hi[p] = h;
        s.mid[p] = m;
        s.lo[p] = pack_bf16(sa, sb);
    }
    return s;
}
// acc += a * b to fp32 accuracy: the six piece products, small ones first
__device__ __forceinline__ f32x4 mfma_split(const u32x4 &ah, const u32x4 &am, const u32x4 &al, const Split3 &b, f32x4 acc) {
    acc = mfma_bf16(al, b.hi, acc);
    acc = mfma_bf16(am, b.mid, acc);
    acc = mfma_bf16(ah, b.lo, acc);
    acc = mfma_bf16(am, b.hi, acc);
    acc = mfma_bf16(ah, b.mid, acc);
    acc = mfma_bf16(ah, b.hi, acc);
    return acc;
}

// ---- wave64 reductions on the DPP cross-lane path (no LDS round trips, unlike __shfl_xor) --------
// quad_perm xor-1 / xor-2, row_ror 4 / 8, then row_bcast 15 / 31: the total lands in lane 63 and is
// returned wave-uniform through readlane.  Fixed association order -> bitwise reproducible.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1, 0xF>(0.f, v);     // quad_perm [1,0,3,2]
    v += dpp_move<0x4E, 0xF>(0.f, v);     // quad_perm [2,3,0,1]
    v += dpp_move<0x124, 0xF>(0.f, v);    // row_ror:4
    v += dpp_move<0x128, 0xF>(0.f, v);    // row_ror:8
    v += dpp_move<0x142, 0xA>(0.f, v);    // row_bcast:15 into rows 1,3
    v += dpp_move<0x143, 0xC>(0.f, v);    // row_bcast:31 into rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x4E, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x124, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x128, 0xF>(v, v));
    v = fmaxf(v, dpp_move<0x142, 0xA>(v, v));
    v = fmaxf(v, dpp_move<0x143, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// ---- Philox4x32-10 (Salmon et al., SC'11), counter-based: no state, reproducible per element ----
struct Philox {
    uint32_t k0, k1;
    __host__ __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __host__ __device__ static inline void mulhilo(uint32_t a, uint32_t b, uint32_t &hi, uint32_t &lo) {
        uint64_t p = (uint64_t)a * b;
        hi = (uint32_t)(p >> 32);
        lo = (uint32_t)p;
    }
    // counter = (ctr, stream); returns 4 words
    __host__ __device__ inline void operator()(uint64_t ctr, uint64_t stream, uint32_t out[4]) const {
        uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = (uint32_t)stream,
                 c3 = (uint32_t)(stream >> 32);
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            uint32_t h0, l0, h1, l1;
            mulhilo(0xD2511F53u, c0, h0, l0);
            mulhilo(0xCD9E8D57u, c2, h1, l1);
            uint32_t n0 = h1 ^ c1 ^ a, n1 = l1, n2 = h0 ^ c3 ^ b, n3 = l0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};
// [0,1) with 24 random bits, like torch's CPU/GPU uniform for fp32
__host__ __device__ inline float u32_to_unit_float(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// [0,1) with 53 random bits, like numpy's random_sample
__host__ __device__ inline double u64_to_unit_double(uint32_t hi, uint32_t lo) {
    uint64_t a = hi >> 5, b = lo >> 6;  // 27 + 26 bits
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

// value squish of the TD target and its inverse (/root/reference/prism/agents/squish_functions.py:4-18; id = PRISM_SQUISH_*),
// operation by operation as the reference's torch expressions evaluate them in fp32; sign(0) = 0
__device__ __forceinline__ float sign_of(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
__device__ __forceinline__ float squish_value(int id, float x) {
    if (id == PRISM_SQUISH_SYMLOG) return sign_of(x) * logf(fabsf(x) + 1.0f);
    if (id == PRISM_SQUISH_OBS_LOOK_FURTHER) return sign_of(x) * (sqrtf(fabsf(x) + 1.0f) - 1.0f) + 0.01f * x;
    return x;
}
__device__ __forceinline__ float unsquish_value(int id, float y) {
    if (id == PRISM_SQUISH_SYMLOG) return sign_of(y) * (expf(fabsf(y)) - 1.0f);
    if (id == PRISM_SQUISH_OBS_LOOK_FURTHER) {
        // torch: sqrt(1 + 4 * 0.01 * (|y| + 1 + 0.01)): the Python scalars fold first (4 * 0.01 in double, then to fp32)
        const float root = sqrtf(1.0f + (float)(4 * 0.01) * (fabsf(y) + 1.0f + 0.01f));
        const float q = (root - 1.0f) / (float)(2 * 0.01);
        return sign_of(y) * (q * q - 1.0f);
    }
    return y;
}
// squish(r + dg * unsquish(z)): separate multiply and add, as the reference writes it
__device__ __forceinline__ float td_target(int id, float r, float z, float dg) {
    if (id == PRISM_SQUISH_NONE) return r + z * dg;
    return squish_value(id, r + unsquish_value(id, z) * dg);
}

// priority exponent with the same special cases torch.pow uses for scalar exponents
// (0.5 -> sqrt, -0.5 -> 1/sqrt), so the default alpha = beta = 0.5 is correctly rounded on both
// host and device.
__host__ __device__ inline float pow_alpha(float x, float alpha) {
    if (alpha == 0.5f) return sqrtf(x);
    if (alpha == 1.0f) return x;
    return powf(x, alpha);
}
__host__ __device__ inline float pow_neg_beta(float x, float beta) {
    if (beta == 0.5f) return 1.0f / sqrtf(x);
    if (beta == 1.0f) return 1.0f / x;
    if (beta == 0.0f) return 1.0f;
    return powf(x, -beta);
}

}  // namespace prism
