// Backward of the Q ensemble's first layers as TWO plain GEMMs on the bf16 matrix pipe (three-piece operands, common.h).
//
// Restates the input side of /root/reference/prism/agents/models/q_ensemble.py:44-92 (autograd of the ten
// [LayerNorm(1024)] -> Linear(1024 -> 128) stacks of ffnn_model.py:61-76) for width 128.  qh_bwd_kernel (qhead_kernels.h)
// walks (head, 16-column slice) workgroups, each re-reading the head's 512 x 128 dpre1 in two layouts and writing a
// per-head embedding gradient (10 x 2 MB out, 10 x 2 MB back in by the conv role).  But every head normalises the SAME
// embedding: xhat[b][n] = (e[b][n] - mu[b]) rstd[b] does not depend on the head, only the affine (g1_h, beta1_h) does.  With
//   P  = [dpre1_0 | ... | dpre1_9]           B x 1280   (row b, column (h, hh))
//   Wg = [g1_0 * W1_0 ; ... ; g1_9 * W1_9]   1280 x 1024
// the whole input side is
//   G = P^T . xhat        1280 x 1024, K = B      dW1_h = g1_h * G_h + beta1_h (x) cs_h,   cs_h[hh] = sum_b dpre1_h[b][hh]
//                                                 dg1_h[n] = sum_hh W1_h[hh][n] G_h[hh][n],  dbeta1_h[n] = sum_hh W1_h[hh][n] cs_h[hh]
//   S = P . Wg            B x 1024,    K = 1280   de[b][n] = rstd[b] (S[b][n] - C1[b] / E - xhat[b][n] C2[b] / E),  C1 = sum_h c1_h, ...
// (without LayerNorm: xhat = e, g1 = 1, beta1 = 0, de = S).  Two regular GEMMs that share both operands; the sum over heads
// happens inside S's K loop, so ONE embedding gradient leaves the kernel (two slots: the K range is halved for occupancy).
//
// One launch, 256-thread workgroups, two roles:
//   G role  head h x 64 columns          128 x 64 tile, K = B in steps of 32:      heads x 16 workgroups
//   S role  K half x 64 rows x 64 cols   64 x 64 tile,  K = 640 in steps of 32:    2 x B/64 x 16 workgroups
// Per K step a workgroup stages its A and B tiles from fp32 global memory, SPLITS them once (three bf16 planes) into LDS --
// double-buffered, the next step's rows are requested before the current step's MFMAs -- and its four waves (2 x 2) take
// their operands from the shared planes: k-contiguous images by ds_read_b128, [k][x] images by ds_read_b64_tr_b16 (the
// hardware's transposed read delivers exactly the 16x16x32 operand).
#pragma once
#include "iqn_kernels.h"
#include "qhead_kernels.h"

namespace prism {
// (own debug bit: this kernel's stamps share slots 0..6 with the forward tiles')
#define QB2_STAMP(k)                                                                       \
    do {                                                                                   \
        if ((a.dbg & 32) && threadIdx.x == 0) {                                            \
            a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime();        \
            a.stamps[(size_t)blockIdx.x * 64 + 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                  \
    } while (0)


constexpr int QB2_H = 128;
constexpr int QB2_KS = 32;                        // K per step = K of one bf16 MFMA
// LDS images (bytes per row; +16 keeps rows 16-byte aligned and staggers the banks)
constexpr int QB2_G_AROW = 2 * 128 + 16, QB2_G_BROW = 2 * 64 + 16;       // G role: A [32 k][128 m], B [32 k][64 n]
constexpr int QB2_S_AROW = 2 * 32 + 16, QB2_S_BROW = 2 * 64 + 16;        // S role: A [64 m][32 k], B [32 k][64 n]
constexpr int QB2_G_A = 32 * QB2_G_AROW, QB2_G_B = 32 * QB2_G_BROW;      // one plane
constexpr int QB2_S_A = 64 * QB2_S_AROW, QB2_S_B = 32 * QB2_S_BROW;
constexpr int QB2_G_BUF = 3 * (QB2_G_A + QB2_G_B), QB2_S_BUF = 3 * (QB2_S_A + QB2_S_B);
constexpr int QB2_LDS_BYTES = 2 * (QB2_G_BUF > QB2_S_BUF ? QB2_G_BUF : QB2_S_BUF);
constexpr LdsRegion QB2_REGIONS[] = {{0, QB2_G_BUF, 1u}, {QB2_G_BUF, QB2_G_BUF, 1u}, {0, QB2_S_BUF, 2u}, {QB2_S_BUF, QB2_S_BUF, 2u}};
static_assert(lds_layout_ok(QB2_REGIONS, QB2_LDS_BYTES), "q backward: LDS buffers overlap");
static_assert(QB2_G_AROW % 16 == 0 && QB2_G_BROW % 16 == 0 && QB2_S_AROW % 16 == 0, "16-byte aligned rows");

// Workgroup -> tile, XCD-aware.  Workgroups go to the eight XCDs round-robin by index, each XCD has its own 4 MB L2, and
// every operand here is re-read by many workgroups (a head's dpre1 by all 16 column tiles, a column tile of xhat / Wg by all
// heads / row tiles).  Mapped naively the launch pulls ~200 MB through the L2s and runs at the Infinity Cache's rate
// (measured: 71 us).  So XCD x = (half, quarter) owns, in the G role, the heads of one half x the column tiles of one
// quarter (2 MB of dpre1 pieces + 0.8 MB of xhat), and in the S role the row tiles of one half x the column tiles of one
// quarter (2 MB + 1.3 MB): every operand byte enters an L2 about once.  Slots of an XCD: the G tiles first, then the S tiles.
__host__ __device__ inline int qb2_g_slots(int n_heads) { return 4 * ((n_heads + 1) / 2); }        // per XCD (heads of a half x 4 column tiles)
__host__ __device__ inline int qb2_s_slots(int B) { return 2 * (B / 128) * 4; }                   // per XCD (K halves x row tiles of a half x 4)
__host__ __device__ inline int qb2_blocks(int n_heads, int B) { return 8 * (qb2_g_slots(n_heads) + qb2_s_slots(B)); }
inline bool qb2_ok(int H, int B, int n_heads, int head_layers) { return H == QB2_H && head_layers == 2 && B % 128 == 0 && n_heads >= 2; }

typedef short s16x4_t __attribute__((ext_vector_type(4)));
// three planes of four consecutive elements of one image row: 8-byte stores
__device__ __forceinline__ void qb2_store4(char *img, int plane_bytes, int off, const float (&x)[4]) {
    const unsigned int h0 = pack_bf16(x[0], x[1]), h1 = pack_bf16(x[2], x[3]);
    const float r0 = x[0] - __uint_as_float(h0 << 16), r1 = x[1] - __uint_as_float(h0 & 0xffff0000u);
    const float r2 = x[2] - __uint_as_float(h1 << 16), r3 = x[3] - __uint_as_float(h1 & 0xffff0000u);
    const unsigned int m0 = pack_bf16(r0, r1), m1 = pack_bf16(r2, r3);
    const float s0 = r0 - __uint_as_float(m0 << 16), s1 = r1 - __uint_as_float(m0 & 0xffff0000u);
    const float s2 = r2 - __uint_as_float(m1 << 16), s3 = r3 - __uint_as_float(m1 & 0xffff0000u);
    *reinterpret_cast<uint2 *>(img + off) = make_uint2(h0, h1);
    *reinterpret_cast<uint2 *>(img + plane_bytes + off) = make_uint2(m0, m1);
    *reinterpret_cast<uint2 *>(img + 2 * plane_bytes + off) = make_uint2(pack_bf16(s0, s1), pack_bf16(s2, s3));
}
// 16x16x32 operand (A: rows = image columns x0 .. x0+15; B alike) from a [k][x] image: k = 8 g + j, two transposed reads
__device__ __forceinline__ u32x4 qb2_tr_operand(const char *plane, int row_bytes, int x0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char *a0 = plane + (8 * g + q) * row_bytes + 2 * (x0 + 4 * p);
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)a0);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a0 + 4 * row_bytes));
    const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
    return u32x4{u0.x, u0.y, u1.x, u1.y};
}
__device__ __forceinline__ Split3 qb2_tr_split(const char *img, int plane_bytes, int row_bytes, int x0, int lane) {
    Split3 s;
    s.hi = qb2_tr_operand(img, row_bytes, x0, lane);
    s.mid = qb2_tr_operand(img + plane_bytes, row_bytes, x0, lane);
    s.lo = qb2_tr_operand(img + 2 * plane_bytes, row_bytes, x0, lane);
    return s;
}

template <bool LN>
__global__ __launch_bounds__(256, 2) void qh_bwd2_kernel(IqnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char *smem = reinterpret_cast<char *>(smem_f);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int wm = w >> 1, wn = w & 1;
    const int B = a.B, Hd = a.n_heads;
    constexpr int H = QB2_H;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, xh = xcd >> 2, xq = xcd & 3;
    const int g_slots = qb2_g_slots(Hd);
    typedef const float4 *cf4;
    QB2_STAMP(0);
    if (slot < g_slots) {
        // =========================== G role: G_h = dpre1_h^T . xhat for 64 columns ===========================
        const int hd = xh * ((Hd + 1) / 2) + (slot >> 2), n0 = 64 * (4 * xq + (slot & 3));
        if (hd >= Hd) return;                     // (odd head counts: the second half has one head less)
        const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
        f32x4 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        // Staging: fp32 rows from global memory, split into the three bf16 planes on their way into LDS.  (Measured
        // alternative: the Q loss kernel writes dpre1 / xhat as bf16 pieces and this kernel only copies them -- no VALU here,
        // but 1.5x the bytes and, with one K step of prefetch, the copies' latency is exposed every step: 70 us against 38.
        // The way to use pieces is an LDS-DMA ring several K steps deep)
        const float *P = a.ws.q_dpre1 + (size_t)hd * B * H;
        float cs[4] = {0.f, 0.f, 0.f, 0.f};                 // partial column sums of dpre1_h: columns 4 (tid & 31) .. + 3
        float4 pa[4], pb[2];
        float pmu[2], prs[2];
        auto request = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                pa[i] = reinterpret_cast<cf4>(P + (size_t)(k0 + (idx >> 5)) * H)[idx & 31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i, b = k0 + (idx >> 4);
                pb[i] = reinterpret_cast<cf4>(a.ws.e_cur + (size_t)b * E_DIM + n0)[idx & 15];
                pmu[i] = LN ? a.ws.q_mu1[b] : 0.f;
                prs[i] = LN ? a.ws.q_rstd1[b] : 1.f;
            }
        };
        auto stage = [&](char *buf) __attribute__((always_inline)) {
            char *A = buf, *Bm = buf + 3 * QB2_G_A;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = tid + 256 * i;
                const float x[4] = {pa[i].x, pa[i].y, pa[i].z, pa[i].w};
                cs[0] += x[0]; cs[1] += x[1]; cs[2] += x[2]; cs[3] += x[3];
                qb2_store4(A, QB2_G_A, (idx >> 5) * QB2_G_AROW + 8 * (idx & 31), x);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                const float x[4] = {(pb[i].x - pmu[i]) * prs[i], (pb[i].y - pmu[i]) * prs[i], (pb[i].z - pmu[i]) * prs[i],
                                    (pb[i].w - pmu[i]) * prs[i]};
                qb2_store4(Bm, QB2_G_B, (idx >> 4) * QB2_G_BROW + 8 * (idx & 15), x);
            }
        };
        request(0);
        stage(smem);
        __syncthreads();
        QB2_STAMP(1);
        const int nk = B / QB2_KS;
        for (int ks = 0; ks < nk; ++ks) {
            char *cur = smem + (ks & 1) * QB2_G_BUF, *nxt = smem + ((ks + 1) & 1) * QB2_G_BUF;
            if (ks + 1 < nk) request((ks + 1) * QB2_KS);
            const char *A = cur, *Bm = cur + 3 * QB2_G_A;
            {
            Split3 bo[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bo[nt] = qb2_tr_split(Bm, QB2_G_B, QB2_G_BROW, 32 * wn + 16 * nt, lane);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const Split3 ao = qb2_tr_split(A, QB2_G_A, QB2_G_AROW, 64 * wm + 16 * mt, lane);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_split(ao.hi, ao.mid, ao.lo, bo[nt], acc[mt][nt]);

            }
            }
            if (ks + 1 < nk) stage(nxt);
            __syncthreads();
        }
        QB2_STAMP(2);
        // ---- epilogue.  cs: fold the 8 row groups of the staging threads (tid >> 5) in fixed order through LDS
        float *s_cs = reinterpret_cast<float *>(smem);          // [8][128]
        float *s_red = s_cs + 8 * 128;                          // [2 quantities][2 (wm)][64 n]
#pragma unroll
        for (int c = 0; c < 4; ++c) s_cs[(tid >> 5) * 128 + 4 * (tid & 31) + c] = cs[c];
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += s_cs[r * 128 + tid];
            s_cs[tid] = t;                                      // (row 0 now holds the totals; only thread tid touches column tid)
        }
        __syncthreads();
        float *slab = a.ws.q_slabs + (int64_t)hd * a.q_slab;
        constexpr int W1_OFF = LN ? 2 * E_DIM : 0;
        float dg[2] = {0.f, 0.f}, db[2] = {0.f, 0.f};
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + 32 * wn + 16 * nt + li;
            const float g1 = LN ? Ph[a.off.h_ln1_g + n] : 1.f, be1 = LN ? Ph[a.off.h_ln1_b + n] : 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int hh = 64 * wm + 16 * mt + 4 * g + r;
                    const float G = acc[mt][nt][r], c = s_cs[hh];
                    slab[W1_OFF + (int64_t)hh * E_DIM + n] = LN ? g1 * G + be1 * c : G;
                    if (LN) {
                        const float wv = Ph[a.off.h_w1 + (int64_t)hh * E_DIM + n];
                        dg[nt] = fmaf(wv, G, dg[nt]);
                        db[nt] = fmaf(wv, c, db[nt]);
                    }
                }
        }
        if (LN) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                dg[nt] += __shfl_xor(dg[nt], 16, 64);
                dg[nt] += __shfl_xor(dg[nt], 32, 64);
                db[nt] += __shfl_xor(db[nt], 16, 64);
                db[nt] += __shfl_xor(db[nt], 32, 64);
                if (g == 0) {
                    s_red[(0 * 2 + wm) * 64 + 32 * wn + 16 * nt + li] = dg[nt];
                    s_red[(1 * 2 + wm) * 64 + 32 * wn + 16 * nt + li] = db[nt];
                }
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, n = tid & 63;
                slab[which * E_DIM + n0 + n] = s_red[(which * 2 + 0) * 64 + n] + s_red[(which * 2 + 1) * 64 + n];
            }
        }
        QB2_STAMP(3);
        return;
    }
    // =========================== S role: S = P . Wg over half of the heads, 64 rows x 64 columns ===========================
    {
        const int x = slot - g_slots, nbh = B / 128;           // x = (kh * nbh + row tile of the half) * 4 + column tile of the quarter
        const int kh = x / (4 * nbh), bi = xh * nbh + (x / 4) % nbh, ni = 4 * xq + (x & 3);
        const int b0 = 64 * bi, n0 = 64 * ni;
        const int h_lo = kh ? (Hd + 1) / 2 : 0, h_hi = kh ? Hd : (Hd + 1) / 2;
        f32x4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i) acc[i][0] = acc[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        float4 pa[2], pb[2], pg;
        auto request = [&](int step) __attribute__((always_inline)) {
            const int hd = h_lo + (step >> 2), k0 = QB2_KS * (step & 3);
            const float *P = a.ws.q_dpre1 + (size_t)hd * B * H;
            const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                pa[i] = reinterpret_cast<cf4>(P + (size_t)(b0 + (idx >> 3)) * H + k0)[idx & 7];
                const float *wr = Ph + a.off.h_w1 + (int64_t)(k0 + (idx >> 4)) * E_DIM + n0 + 4 * (idx & 15);     // (head tensors: 4-byte aligned)
                pb[i] = make_float4(wr[0], wr[1], wr[2], wr[3]);
            }
            if (LN) {
                const float *gr = Ph + a.off.h_ln1_g + n0 + 4 * (tid & 15);
                pg = make_float4(gr[0], gr[1], gr[2], gr[3]);
            }
        };
        auto stage = [&](char *buf) __attribute__((always_inline)) {
            char *A = buf, *Bm = buf + 3 * QB2_S_A;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = tid + 256 * i;
                const float xa[4] = {pa[i].x, pa[i].y, pa[i].z, pa[i].w};
                qb2_store4(A, QB2_S_A, (idx >> 3) * QB2_S_AROW + 8 * (idx & 7), xa);
                const float xb[4] = {LN ? pb[i].x * pg.x : pb[i].x, LN ? pb[i].y * pg.y : pb[i].y, LN ? pb[i].z * pg.z : pb[i].z,
                                     LN ? pb[i].w * pg.w : pb[i].w};
                qb2_store4(Bm, QB2_S_B, (idx >> 4) * QB2_S_BROW + 8 * (idx & 15), xb);
            }
        };
        const int nk = (h_hi - h_lo) * (H / QB2_KS);
        request(0);
        stage(smem);
        __syncthreads();
        QB2_STAMP(1);
        for (int ks = 0; ks < nk; ++ks) {
            char *cur = smem + (ks & 1) * QB2_S_BUF, *nxt = smem + ((ks + 1) & 1) * QB2_S_BUF;
            if (ks + 1 < nk) request(ks + 1);
            const char *A = cur, *Bm = cur + 3 * QB2_S_A;
            {
            Split3 bo[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) bo[nt] = qb2_tr_split(Bm, QB2_S_B, QB2_S_BROW, 32 * wn + 16 * nt, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const char *ar = A + (32 * wm + 16 * mt + li) * QB2_S_AROW + 16 * g;          // [m][k]: eight consecutive k
                const u32x4 ah = *reinterpret_cast<const u32x4 *>(ar), am = *reinterpret_cast<const u32x4 *>(ar + QB2_S_A),
                            al = *reinterpret_cast<const u32x4 *>(ar + 2 * QB2_S_A);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = mfma_split(ah, am, al, bo[nt], acc[mt][nt]);
            }
            }
            if (ks + 1 < nk) stage(nxt);
            __syncthreads();
        }
        QB2_STAMP(2);
        // ---- epilogue: this K half's share of the embedding gradient (the LayerNorm terms ride with half 0).  The row
        // scalars -- rstd, and for half 0 mean and the sums over the heads of c1 / c2 (heads in order) -- are gathered once
        // per workgroup: thread = (row, head) requests, then one thread per row folds
        float *s_row = reinterpret_cast<float *>(smem);            // [4][64]: rstd | mean | C1 | C2
        float *s_c = s_row + 4 * 64;                               // [2][16 heads][64 rows]
        if (LN) {
            if (tid < 64) {
                s_row[tid] = a.ws.q_rstd1[b0 + tid];
                s_row[64 + tid] = a.ws.q_mu1[b0 + tid];
            }
            if (kh == 0) {
                for (int i = tid; i < Hd * 64; i += 256) {
                    const int hd = i >> 6, r = i & 63;
                    s_c[hd * 64 + r] = a.ws.q_c1[(size_t)hd * B + b0 + r];
                    s_c[(16 + hd) * 64 + r] = a.ws.q_c2[(size_t)hd * B + b0 + r];
                }
            }
            __syncthreads();
            if (kh == 0 && tid < 128) {
                const int which = tid >> 6, r = tid & 63;
                float t = 0.f;
                for (int hd = 0; hd < Hd; ++hd) t += s_c[(16 * which + hd) * 64 + r];
                s_row[(2 + which) * 64 + r] = t;
            }
            __syncthreads();
        }
        float *de = a.ws.de_q + (size_t)kh * B * E_DIM;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rl = 32 * wm + 16 * mt + 4 * g + r, b = b0 + rl;
                const float rs = LN ? s_row[rl] : 1.f, mu = LN ? s_row[64 + rl] : 0.f;
                const float C1 = (LN && kh == 0) ? s_row[128 + rl] : 0.f, C2 = (LN && kh == 0) ? s_row[192 + rl] : 0.f;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int n = n0 + 32 * wn + 16 * nt + li;
                    float v = acc[mt][nt][r];
                    if (LN) {
                        v = rs * v;
                        if (kh == 0) {
                            const float xhat = (a.ws.e_cur[(size_t)b * E_DIM + n] - mu) * rs;
                            v -= rs * (C1 * (1.0f / E_DIM) + xhat * (C2 * (1.0f / E_DIM)));
                        }
                    }
                    de[(size_t)b * E_DIM + n] = v;
                }
            }
        QB2_STAMP(3);
    }
}

}  // namespace prism
