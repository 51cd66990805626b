// IQN backward on the bf16 matrix pipe (width 128): dX, dWphi, dW1 with three-piece operands (common.h), the row operands
// split ONCE per workgroup into LDS and shared by its four waves.
//
// Restates the backward of /root/reference/prism/agents/models/iqn_model.py:48-93 (+ ffnn_model.py:61-76) as
// iqn_bwd_kernel (iqn_kernels.h) does -- the same products, the same element-wise LayerNorm / ReLU / Hadamard backward,
// the same column-slice ownership of every weight gradient -- with another division of labour:
//   iqn_bwd_kernel   workgroup = 16 columns x a row chunk; its four waves walk DIFFERENT rows, each loads the row operands
//                    (dpre1 in two layouts, cos) itself; fp32 MFMA, whose issue excludes every vector instruction
//   iqn_bwd3_kernel  workgroup = 64 columns x a row chunk; its four waves own 16 columns each and walk the SAME 32-row blocks:
//                    dpre1 [32][128] and cos [32][64] are fetched once, split into bf16 planes once (256 threads share the
//                    work) and parked in LDS; a wave takes dpre1 as the A operand of dX by row reads and as the A operand of
//                    dW1 by transposed reads of the same image, cos as the B operand of dWphi by transposed reads; the
//                    element-wise results (dphi, x: 8 values a lane) are split in registers and ARE the remaining operands
//                    (K index = the block's rows in accumulator order: two transposed reads at rows 4g and 16 + 4g).
// Eight waves, two teams.  Waves 0-3 compute (the matrix work and the element-wise step of their 16 columns).  Waves 4-7 help:
// they stage the NEXT block (fetch, split, park) while the computing waves work on the current one, and -- IQN-only models
// -- accumulate the conv-backward taps of the PREVIOUS block from the ReLU-masked d e values their partner wave left in LDS
// (observation rows by LDS-DMA, as in iqn_bwd_kernel).  Vector work of one team issues beside the MFMAs of the other: with
// everything on four waves (one per SIMD) a block took 6.9 k cycles and the taps another 1.1 k (measured); the taps alone
// made that form break even with iqn_bwd_kernel.
// No cross-wave reduction: a wave owns its columns over all rows of the chunk.  Per 32 x 16 block and wave: 120 bf16 MFMAs
// (1920 cycles, half of them open to vector issue) and ~450 vector instructions, against 160 fp32 MFMAs (5120) + ~900.
// L2 traffic per row operand byte drops four-fold (64 columns per fetch instead of 16).
#pragma once
#include "iqn_kernels.h"
#include "qbwd2_kernels.h"

namespace prism {
// (diagnostic stamp of the helper team: its first thread)
#define BW3_HSTAMP(k)                                                                      \
    do {                                                                                   \
        if ((a.dbg & 8) && threadIdx.x == 256)                                             \
            a.stamps[(size_t)(2048 + blockIdx.x) * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)


constexpr int BW3_H = 128, BW3_RB = 32;            // width; rows per block
constexpr int BW3_RC = 16;                        // row chunks (gradient slabs)
constexpr int BW3_PROW = 2 * 128 + 32, BW3_CROW = 2 * 64 + 32;      // LDS row strides (bytes): transposed reads conflict-free
constexpr int BW3_P = BW3_RB * BW3_PROW, BW3_C = BW3_RB * BW3_CROW;  // one plane
constexpr int BW3_BUF = 3 * (BW3_P + BW3_C);
constexpr int BW3_DCV = 2 * 4 * 2 * 64 * 4;        // [block parity][slice][tile][lane] ReLU-masked d e, compute -> helper waves (bytes)
constexpr int BW3_LDS_BYTES = 2 * BW3_BUF + BW3_DCV;
constexpr LdsRegion BW3_REGIONS[] = {{0, 3 * BW3_P, 1u}, {3 * BW3_P, 3 * BW3_C, 1u}, {BW3_BUF, 3 * BW3_P, 1u}, {BW3_BUF + 3 * BW3_P, 3 * BW3_C, 1u},
                                     {2 * BW3_BUF, BW3_DCV, 1u}};
static_assert(lds_layout_ok(BW3_REGIONS, BW3_LDS_BYTES), "backward (bf16): LDS images overlap");
static_assert(BW3_PROW % 16 == 0 && BW3_CROW % 16 == 0, "16-byte aligned image rows");

// conv-backward taps (IQN-only models): behind the fixed regions, per helper wave the observation rows [samples][4 rows][10][C]
// of the chunk's samples, then the lane-group partial sums [4 waves][4 groups][TAPS + 1], then the four helper waves'
// finished rows [4][BWD_CONV_ROW] and their arrival count: the wave that finishes last adds the four (slice order, the
// order the post launch used to add them in) and writes ONE row per (row chunk, channel) -- the post launch's single
// conv workgroup then reads 16 instead of 64 partials per output
__host__ __device__ inline int bw3_samples(int B, int T) { return (B * T / BW3_RC) / T; }
__host__ __device__ inline int bw3_lds_bytes(int B, int T, int C, bool conv) {
    return BW3_LDS_BYTES + (conv ? 4 * (4 * bw3_samples(B, T) * 40 * C + 4 * 4 * (BWD_CONV_TAPS + 1) + 4 * BWD_CONV_ROW + 4) : 0);
}
inline bool bw3_conv_ok(int use_iqn, int n_heads, int propagate_grad, int T, int C, int B) {
    const int share = bwd_conv_share(T);
    return use_iqn && n_heads == 0 && propagate_grad && T >= 8 && C % share == 0 && 9 * (C / share) <= BWD_CONV_TAPS && C / share <= 2 &&
           (C / share == 1 || C % 2 == 0) && 9 * C < BWD_CONV_ROW && bw3_samples(B, T) * 10 * C <= 16 * 64 &&
           bw3_lds_bytes(B, T, C, true) <= 160 * 1024;
}
inline bool bw3_ok(int H, int B, int T, bool phi_saved) { return H == BW3_H && phi_saved && (B * T) % (BW3_RC * BW3_RB) == 0 && ((B * T) / BW3_RC) % T == 0; }

// 16x16x32 operand whose K index runs over the block's rows in ACCUMULATOR order (k = (g, j): row 4 g + j for j < 4, row
// 16 + 4 g + (j - 4) above): two transposed reads of a [row][x] image, columns x0 .. x0 + 15
__device__ __forceinline__ u32x4 bw3_tr_rows(const char *plane, int row_bytes, int x0, int lane) {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
    const char *a0 = plane + (4 * g + q) * row_bytes + 2 * (x0 + 4 * p);
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)a0);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t *)(a0 + 16 * row_bytes));
    const uint2 u0 = __builtin_bit_cast(uint2, v0), u1 = __builtin_bit_cast(uint2, v1);
    return u32x4{u0.x, u0.y, u1.x, u1.y};
}

template <bool LN>
__global__ __launch_bounds__(512) void iqn_bwd3_kernel(IqnArgs a) {
    kernarg_prefetch<sizeof(IqnArgs)>();
    constexpr int H = BW3_H, NHT = H / 16;
    constexpr int SLAB_W1 = E_DIM * K_BASIS + E_DIM + (LN ? 2 * E_DIM : 0);
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char *smem = reinterpret_cast<char *>(smem_f);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int team = w >> 2, wq = w & 3, ht_tid = tid & 255;       // team 0 computes, team 1 helps; wq: the 16-column slice of the group
    // (the row chunk in the low bits of the index: the 16 column groups of a chunk share an XCD, i.e. one L2 holds its rows)
    const int rc = blockIdx.x % BW3_RC, cg = blockIdx.x / BW3_RC;
    const int n = 64 * cg + 16 * wq + li, cs = 4 * cg + wq;         // this lane's embed column; the wave's 16-column slice
    const int T = a.T, R = a.B * T, rpc = R / BW3_RC, row0 = rc * rpc, nblk = rpc / BW3_RB;
    typedef const float4 *cf4;
    const float *P = a.params;
    float *s_dcv = reinterpret_cast<float *>(smem + 2 * BW3_BUF);    // [parity][slice][tile][64]
    PRISM_STAMP(8);
    if (team == 1) {
        // =============================== helper waves: staging + conv taps ===============================
        const int C = a.C, y0 = (cs & 3) * 2;
        const int share = bwd_conv_share(T), cpl = a.conv_in_bwd ? C / share : 0, n_mine = 9 * cpl;
        const int sub = T == 8 ? (g & 1) : g;
        const int ws_lo = row0 / T, ws_n = n_mine ? rpc / T : 0;
        float *s_obs = reinterpret_cast<float *>(smem + BW3_LDS_BYTES) + wq * (ws_n * 40 * C);
        float *s_tap = reinterpret_cast<float *>(smem + BW3_LDS_BYTES) + 4 * (ws_n * 40 * C);
        float *s_row = s_tap + 4 * 4 * (BWD_CONV_TAPS + 1);
        unsigned int *s_cnt = reinterpret_cast<unsigned int *>(s_row + 4 * BWD_CONV_ROW);
        if (n_mine && ht_tid == 0) *s_cnt = 0u;          // (the block loop's barriers lie between this and the first arrival)
        float4 pd[4], pc[2];
        auto request = [&](int blk) __attribute__((always_inline)) {
            const int r0 = row0 + blk * BW3_RB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = ht_tid + 256 * i;
                pd[i] = reinterpret_cast<cf4>(a.ws.dpre1 + (size_t)(r0 + (idx >> 5)) * H)[idx & 31];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = ht_tid + 256 * i;
                pc[i] = reinterpret_cast<cf4>(a.ws.cosb + (size_t)(r0 + (idx >> 4)) * K_BASIS)[idx & 15];
            }
        };
        auto stage = [&](char *buf) __attribute__((always_inline)) {
            char *PD = buf, *CS = buf + 3 * BW3_P;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = ht_tid + 256 * i;
                const float x[4] = {pd[i].x, pd[i].y, pd[i].z, pd[i].w};
                qb2_store4(PD, BW3_P, (idx >> 5) * BW3_PROW + 8 * (idx & 31), x);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = ht_tid + 256 * i;
                const float x[4] = {pc[i].x, pc[i].y, pc[i].z, pc[i].w};
                qb2_store4(CS, BW3_C, (idx >> 4) * BW3_CROW + 8 * (idx & 15), x);
            }
        };
        float cacc[BWD_CONV_TAPS], cbias = 0.f;
#pragma unroll
        for (int i = 0; i < BWD_CONV_TAPS; ++i) cacc[i] = 0.f;
        // taps of block `blk` (its two tiles) from the masked d e values the partner wave parked
#define BW3_TAPS(blk_)                                                                                              \
    do {                                                                                                            \
        _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                                             \
            const int bsm = (row0 + (blk_) * BW3_RB + 16 * t + 4 * g) / T;                                          \
            const float dcv = s_dcv[((((blk_) & 1) * 4 + wq) * 2 + t) * 64 + lane];                                 \
            if (sub == 0) cbias += dcv;                                                                             \
            const float *src = s_obs + (bsm - ws_lo) * 40 * C + ((li >> 3) * 10 + (li & 7)) * C + sub * cpl;        \
            if (cpl == 2) {                                                                                         \
                float2 ob[9];                                                                                       \
                _Pragma("unroll") for (int i = 0; i < 9; ++i)                                                       \
                    ob[i] = *reinterpret_cast<const float2 *>(src + ((i / 3) * 10 + (i % 3)) * C);                  \
                _Pragma("unroll") for (int i = 0; i < 9; ++i) {                                                     \
                    cacc[i] = fmaf(dcv, ob[i].x, cacc[i]);                                                          \
                    cacc[9 + i] = fmaf(dcv, ob[i].y, cacc[9 + i]);                                                  \
                }                                                                                                   \
            } else {                                                                                                \
                float ob[9];                                                                                        \
                _Pragma("unroll") for (int i = 0; i < 9; ++i) ob[i] = src[((i / 3) * 10 + (i % 3)) * C];            \
                _Pragma("unroll") for (int i = 0; i < 9; ++i) cacc[i] = fmaf(dcv, ob[i], cacc[i]);                  \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)
        BW3_HSTAMP(9);
        request(0);
        BW3_HSTAMP(10);
        stage(smem);
        BW3_HSTAMP(11);
        lds_barrier();                                              // (1) block 0 is parked
        BW3_HSTAMP(12);
        // the observation rows are asked for only now: sixty-four more wave-loads in front of the first barrier kept the
        // address unit busy while every wave of the workgroup waited for block 0 (they are first needed at block 1's taps)
        // (idx / (10 C) by multiplication: idx < 1024, 10 C <= 100 -- sixteen run-time divisions here were 3 k cycles in front of
        // the first block's staging, i.e. of every wave's first barrier)
        const unsigned int rdiv = (1u << 20) / (unsigned int)(10 * C) + 1u;      // exact for idx < 1024, 10 C <= 100 (checked offline)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = lane + 64 * i;
            if (idx < ws_n * 10 * C) {
                const int sm = (int)(((unsigned int)idx * rdiv) >> 20), o4 = idx - sm * 10 * C;
                const float *src = a.obs + ((int64_t)(ws_lo + sm) * 100 + y0 * 10) * C + 4 * o4;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                                 (__attribute__((address_space(3))) void *)(s_obs + 256 * i), 16, 0, 0);
            }
        }
        for (int blk = 0; blk < nblk; ++blk) {
            if (n_mine && blk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the wave's observation rows have landed
            if (blk + 1 < nblk) request(blk + 1);
            if (n_mine && blk >= 1) BW3_TAPS(blk - 1);
            if (blk + 1 < nblk) stage(smem + ((blk + 1) & 1) * BW3_BUF);
            if (blk == 3) BW3_HSTAMP(20);
            lds_barrier();                                          // (2 + blk)
            if (blk == 3) BW3_HSTAMP(21);
            if (blk == 2) BW3_HSTAMP(19);
        }
        if (n_mine) {
            if (nblk == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BW3_TAPS(nblk - 1);
#undef BW3_TAPS
            // fold the 16 positions of a lane group, park one row per (wave, lane group), then the wave's own outputs
            constexpr int TS = BWD_CONV_TAPS + 1;
#pragma unroll
            for (int i = 0; i < TS; ++i) {
                float v = i < BWD_CONV_TAPS ? cacc[i < BWD_CONV_TAPS ? i : 0] : cbias;
                v += dpp_move<0xB1, 0xF>(0.f, v);
                v += dpp_move<0x4E, 0xF>(0.f, v);
                v += dpp_move<0x124, 0xF>(0.f, v);
                v += dpp_move<0x128, 0xF>(0.f, v);
                if (li == 0) s_tap[(wq * 4 + g) * TS + i] = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            if (lane <= 9 * C) {
                // output tap o belongs to lane subset so = o / n_mine; for T = 8 lane groups so and so + 2 hold it
                const int so = lane < 9 * C ? lane / n_mine : 0, i = lane < 9 * C ? lane - so * n_mine : BWD_CONV_TAPS;
                float tsum = s_tap[(wq * 4 + so) * TS + i];
                if (T == 8) tsum += s_tap[(wq * 4 + so + 2) * TS + i];
                s_row[wq * BWD_CONV_ROW + lane] = tsum;
            }
            // last of the four helper waves to get here folds (LDS operations of a wave retire in order: a wave's row is
            // in place before its count is)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            int arrived = 0;
            if (lane == 0) arrived = (int)__hip_atomic_fetch_add(s_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            arrived = __builtin_amdgcn_readfirstlane(arrived);
            if (arrived == 3 && lane <= 9 * C) {
                const float *r = s_row + lane;
                a.ws.convpart[(int64_t)(rc * (E_DIM / 16) + (cs & ~3)) * BWD_CONV_ROW + lane] =
                    ((r[0] + r[BWD_CONV_ROW]) + r[2 * BWD_CONV_ROW]) + r[3 * BWD_CONV_ROW];
            }
        }
        PRISM_STAMP(26);
        return;
    }
    // =============================== computing waves ===============================
    const bool want_dcv = a.conv_in_bwd != 0;
    // the W1 slice of the wave's columns as the B operand of dX (K = hidden unit), split once
    Split3 w1p[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = P[a.off.iqn_w1 + (int64_t)(32 * kb + 8 * g + j) * E_DIM + n];
        w1p[kb] = split_bf16x3(wv);
    }
    PRISM_STAMP2(13);
    const float g1 = LN ? P[a.off.iqn_ln1_g + n] : 1.f, be1 = LN ? P[a.off.iqn_ln1_b + n] : 0.f;
    const __amdgpu_buffer_rsrc_t rs_ph = __builtin_amdgcn_make_buffer_rsrc(a.ws.phis, 0, ((R + 15) / 16) * 16 * E_DIM * 4, 0x00020000);
    const int vo_ph = (4 * g * 16 + li) * 4;

    f32x4 accWphi[4], accW1[NHT];      // accWphi[c][r]: dWphi[n = 16 cs + 4g + r][k = 16c + li];  accW1[ht][r]: dW1[h = 16ht + 4g + r][n]
#pragma unroll
    for (int i = 0; i < 4; ++i) accWphi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NHT; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_dg = 0.f, s_db = 0.f, s_dbphi = 0.f, de_acc = 0.f;
    // per-wave row data of a block: saved ReLU(phi) of the wave's columns, LayerNorm row scalars, the samples' embeddings
    struct RowData {
        f32x4 ph[2], mu[2], rs[2], c1[2], c2[2];
        float ev[2];
    };
    auto load_rows = [&](RowData &D, int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r0 = row0 + blk * BW3_RB + 16 * t;
            const int so = ((r0 >> 4) * (E_DIM / 16) + cs) * 1024;
#pragma unroll
            for (int r = 0; r < 4; ++r) D.ph[t][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ph, vo_ph + 64 * r, so, 0));
            if (LN) {
                D.mu[t] = *reinterpret_cast<const f32x4 *>(a.ws.mu1 + r0 + 4 * g);
                D.rs[t] = *reinterpret_cast<const f32x4 *>(a.ws.rstd1 + r0 + 4 * g);
                D.c1[t] = *reinterpret_cast<const f32x4 *>(a.ws.c1 + r0 + 4 * g);
                D.c2[t] = *reinterpret_cast<const f32x4 *>(a.ws.c2 + r0 + 4 * g);
            }
            D.ev[t] = a.ws.e_cur[(int64_t)((r0 + 4 * g) / T) * E_DIM + n];
        }
    };
    RowData D;
    load_rows(D, 0);
    PRISM_STAMP2(14);
    lds_barrier();                                                  // (1)
    PRISM_STAMP2(15);
    PRISM_STAMP(9);
    for (int blk = 0; blk < nblk; ++blk) {
        const char *cur = smem + (blk & 1) * BW3_BUF;
        const bool more = blk + 1 < nblk;
        const char *PD = cur, *CS = cur + 3 * BW3_P;
        float xs8[8], dp8[8];
        u32x4 xa[2][3];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r0 = row0 + blk * BW3_RB + 16 * t;
            // dX[m = 4g + r][n] = sum_h dpre1[m][h] W1[h][n]: A = the image's rows (eight consecutive hidden units a lane)
            f32x4 adx = {0.f, 0.f, 0.f, 0.f};
            // (K blocks walked one ahead, like the operand groups below; the first block of tile 1 is asked for before
            // tile 0's element-wise work)
            auto xrd = [&](int tt, int kb, u32x4 (&o)[3]) __attribute__((always_inline)) {
                const char *ar = PD + (16 * tt + li) * BW3_PROW + 64 * kb + 16 * g;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) o[pl] = *reinterpret_cast<const u32x4 *>(ar + pl * BW3_P);
            };
            if (t == 0) xrd(0, 0, xa[0]);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (kb + 1 < 4) xrd(t, kb + 1, xa[(kb + 1) & 1]);
                else if (t == 0) xrd(1, 0, xa[0]);
                __builtin_amdgcn_sched_barrier(0);
                adx = mfma_split(xa[kb & 1][0], xa[kb & 1][1], xa[kb & 1][2], w1p[kb], adx);
                __builtin_amdgcn_sched_barrier(0);
            }
            // element-wise backward of the tile's rows 4g + r, column n (as iqn_bwd_kernel)
            const float ev = D.ev[t];
            float dep = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float phi = D.ph[t][r];
                const float h0 = phi * ev;
                const float xhat = LN ? (h0 - D.mu[t][r]) * D.rs[t][r] : h0;
                xs8[4 * t + r] = LN ? xhat * g1 + be1 : h0;
                const float dX = adx[r];
                s_dg += dX * xhat;
                s_db += dX;
                const float dh0 = LN ? D.rs[t][r] * (dX * g1 - D.c1[t][r] * (1.0f / E_DIM) - xhat * (D.c2[t][r] * (1.0f / E_DIM))) : dX;
                dep += dh0 * phi;
                const float dpp = (phi > 0.f) ? dh0 * ev : 0.f;
                dp8[4 * t + r] = dpp;
                s_dbphi += dpp;
            }
            // d e[b][n]: sum over the T rows of a sample
            const int bsm = (r0 + 4 * g) / T;
            const bool ev_pos = ev > 0.f;
            float dcv = 0.f;          // ReLU-masked d e of (sample, column n) when it is final in this lane
            if (T == 4) {
                a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
            } else if (T == 8) {
                dep += __shfl_xor(dep, 16, 64);
                if (!want_dcv && (g & 1) == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
                dcv = ev_pos ? dep : 0.f;
            } else {
                dep += __shfl_xor(dep, 16, 64);
                dep += __shfl_xor(dep, 32, 64);
                de_acc += dep;
                if (((r0 + 16) % T) == 0) {
                    if (!want_dcv && g == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = de_acc;
                    dcv = ev_pos ? de_acc : 0.f;
                    de_acc = 0.f;
                }
            }
            if (want_dcv) s_dcv[(((blk & 1) * 4 + wq) * 2 + t) * 64 + lane] = dcv;      // (the helper wave reads it behind the barrier)
        }
        if (more) load_rows(D, blk + 1);          // (this block's row data is consumed)
        const Split3 X = split_bf16x3(xs8), DP = split_bf16x3(dp8);
        // dWphi[n][k] += sum_m dphi[m][n] cos[m][k]: A = dphi (own registers), B = cos by transposed reads
        // dW1[h][n]   += sum_m dpre1[m][h] x[m][n]: A = dpre1 by transposed reads of the same image, B = x (own registers)
        // Twelve operand groups (4 cos column blocks, 8 hidden tiles), each three transposed piece reads + six MFMAs, walked
        // ONE GROUP AHEAD: the reads of group q + 1 are in flight while the MFMAs of group q issue.  (All 72 reads first and
        // then all 72 MFMAs -- what the straight-line form compiled to -- left this one wave per SIMD waiting for LDS: 5.2 k
        // cycles a block for 1.9 k of matrix work, with the helper team idle 70 % of the time.)
        auto grp = [&](int q, u32x4 (&o)[3]) __attribute__((always_inline)) {
            if (q < 4) {
                o[0] = bw3_tr_rows(CS, BW3_CROW, 16 * q, lane);
                o[1] = bw3_tr_rows(CS + BW3_C, BW3_CROW, 16 * q, lane);
                o[2] = bw3_tr_rows(CS + 2 * BW3_C, BW3_CROW, 16 * q, lane);
            } else {
                o[0] = bw3_tr_rows(PD, BW3_PROW, 16 * (q - 4), lane);
                o[1] = bw3_tr_rows(PD + BW3_P, BW3_PROW, 16 * (q - 4), lane);
                o[2] = bw3_tr_rows(PD + 2 * BW3_P, BW3_PROW, 16 * (q - 4), lane);
            }
        };
        u32x4 og[2][3];
        grp(0, og[0]);
#pragma unroll
        for (int q = 0; q < 4 + NHT; ++q) {
            if (q + 1 < 4 + NHT) grp(q + 1, og[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 4) {
                Split3 cbq;
                cbq.hi = og[q & 1][0];
                cbq.mid = og[q & 1][1];
                cbq.lo = og[q & 1][2];
                accWphi[q] = mfma_split(DP.hi, DP.mid, DP.lo, cbq, accWphi[q]);
            } else {
                accW1[q - 4] = mfma_split(og[q & 1][0], og[q & 1][1], og[q & 1][2], X, accW1[q - 4]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (blk == 3) PRISM_STAMP2(17);
        lds_barrier();                                              // (2 + blk)
        if (blk == 3) PRISM_STAMP2(18);
        if (blk == 2) PRISM_STAMP2(16);
    }
    PRISM_STAMP(10);
    // ---- this wave's part of the chunk's slab: nobody else holds these columns
    float *slab = a.ws.slabs + (int64_t)rc * a.slab;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(accWphi[c][r], slab + (int64_t)(16 * cs + 4 * g + r) * K_BASIS + 16 * c + li);
#pragma unroll
    for (int ht = 0; ht < NHT; ++ht)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(accW1[ht][r], slab + SLAB_W1 + (int64_t)(16 * ht + 4 * g + r) * E_DIM + n);
    s_dg += __shfl_xor(s_dg, 16, 64);
    s_dg += __shfl_xor(s_dg, 32, 64);
    s_db += __shfl_xor(s_db, 16, 64);
    s_db += __shfl_xor(s_db, 32, 64);
    s_dbphi += __shfl_xor(s_dbphi, 16, 64);
    s_dbphi += __shfl_xor(s_dbphi, 32, 64);
    if (g == 0) {
        slab[E_DIM * K_BASIS + n] = s_dbphi;
        if (LN) {
            slab[E_DIM * K_BASIS + E_DIM + n] = s_dg;
            slab[E_DIM * K_BASIS + 2 * E_DIM + n] = s_db;
        }
    }
    PRISM_STAMP(12);
}

}  // namespace prism
