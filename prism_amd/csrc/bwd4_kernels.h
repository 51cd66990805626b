// IQN backward on the bf16 matrix pipe at hidden width 256 -- the shape the reference's ablation presets run
// (/root/reference/prism/config/additive_ablation_base_config.py:21-25,38, subtractive_ablation_base_config.py:6-10,18,45).
//
// Same products, same element-wise LayerNorm / ReLU / Hadamard backward, same column-slice ownership of every weight
// gradient as iqn_bwd3_kernel (bwd3_kernels.h; math: the backward of /root/reference/prism/agents/models/iqn_model.py:48-93
// + ffnn_model.py:61-76).  What changes with the width is where the registers go: a wave that owned 16 columns over all 256
// hidden units would hold 96 registers of W1 pieces (B operand of dX) and 64 of dW1 accumulators -- more than the 256 a wave
// has at two waves per SIMD.  So the four computing waves of a workgroup are two PAIRS: a pair owns 16 embed columns, each
// wave of it one HALF of the hidden units:
//   phase 1   both waves: partial dX over their 128 hidden units (both 16-row tiles of the block), parked in LDS
//   barrier A
//   phase 2   both waves add the two partials in the same order (hidden half 0 first: bit-identical on both), do the element-wise
//             step of the 16 columns redundantly, and split the work that follows from it: the wave of half hh takes cos column
//             blocks 2 hh, 2 hh + 1 of dWphi and hidden tiles 8 hh .. 8 hh + 7 of dW1
//   barrier B (the block's LDS images are free)
// A workgroup thus covers 32 columns x one row chunk; per wave and 32-row block 48 + 12 + 48 bf16 MFMAs (iqn_bwd3_kernel: 120).
// Waves 4-7 stage the next block's row operands (dpre1 [32][256], cos [32][64]: fetched once, split into three bf16 planes once)
// while the computing waves work, as there.  The conv-backward taps stay with the post launch at this width.
#pragma once
#include "bwd3_kernels.h"

namespace prism {

constexpr int BW4_H = 256, BW4_RB = 32;
constexpr int BW4_PROW = 2 * BW4_H + 32, BW4_CROW = 2 * 64 + 32;      // LDS row strides (bytes): 8 dwords past a multiple of 64 banks, as at width 128
constexpr int BW4_P = BW4_RB * BW4_PROW, BW4_C = BW4_RB * BW4_CROW;   // one plane
constexpr int BW4_BUF = 3 * (BW4_P + BW4_C);
constexpr int BW4_XCH = 4 * 2 * 64 * 16;                              // [wave][tile][lane] partial dX (f32x4)
constexpr int BW4_LDS_BYTES = 2 * BW4_BUF + BW4_XCH;
constexpr LdsRegion BW4_REGIONS[] = {{0, 3 * BW4_P, 1u}, {3 * BW4_P, 3 * BW4_C, 1u}, {BW4_BUF, 3 * BW4_P, 1u}, {BW4_BUF + 3 * BW4_P, 3 * BW4_C, 1u},
                                     {2 * BW4_BUF, BW4_XCH, 1u}};
static_assert(lds_layout_ok(BW4_REGIONS, BW4_LDS_BYTES), "backward (bf16, width 256): LDS images overlap");
static_assert(BW4_PROW % 16 == 0 && BW4_CROW % 16 == 0 && BW4_LDS_BYTES <= 160 * 1024, "16-byte aligned image rows; one workgroup per CU");

// row chunks (gradient slabs): eight where the rows divide (256 workgroups: one round of the CUs), else the next that does
inline int bw4_chunks(int B, int T) {
    const int R = B * T;
    for (int rc : {8, 16, 4, 2, 1})
        if (R % (rc * BW4_RB) == 0 && (R / rc) % T == 0) return rc;
    return 0;
}
inline bool bw4_ok(int H, int B, int T) { return H == BW4_H && bw4_chunks(B, T) > 0; }

template <bool LN>
__global__ __launch_bounds__(512) void iqn_bwd4_kernel(IqnArgs a) {
    kernarg_prefetch<sizeof(IqnArgs)>();
    constexpr int H = BW4_H;
    constexpr int SLAB_W1 = E_DIM * K_BASIS + E_DIM + (LN ? 2 * E_DIM : 0);
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    char *smem = reinterpret_cast<char *>(smem_f);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int team = w >> 2, wq = w & 3, ht_tid = tid & 255;
    const int cq = wq >> 1, hh = wq & 1;              // the pair's 16-column slice of the group; this wave's half of the hidden units
    // (the row chunk in the low bits of the index: the column groups of a chunk share an XCD where the chunk count is a multiple of 8)
    const int RC = a.n_chunks, rc = blockIdx.x % RC, cg = blockIdx.x / RC;
    const int n = 32 * cg + 16 * cq + li, cs = 2 * cg + cq;           // this lane's embed column; the pair's 16-column slice
    const int T = a.T, R = a.B * T, rpc = R / RC, row0 = rc * rpc, nblk = rpc / BW4_RB;
    typedef const float4 *cf4;
    const float *P = a.params;
    f32x4 *xch = reinterpret_cast<f32x4 *>(smem + 2 * BW4_BUF);
    PRISM_STAMP(8);
    if (team == 1) {
        // =============================== helper waves: staging ===============================
        float4 pd[8], pc[2];
        auto request = [&](int blk) __attribute__((always_inline)) {
            const int r0 = row0 + blk * BW4_RB;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = ht_tid + 256 * i;
                pd[i] = reinterpret_cast<cf4>(a.ws.dpre1 + (size_t)(r0 + (idx >> 6)) * H)[idx & 63];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = ht_tid + 256 * i;
                pc[i] = reinterpret_cast<cf4>(a.ws.cosb + (size_t)(r0 + (idx >> 4)) * K_BASIS)[idx & 15];
            }
        };
        auto stage = [&](char *buf) __attribute__((always_inline)) {
            char *PD = buf, *CS = buf + 3 * BW4_P;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = ht_tid + 256 * i;
                const float x[4] = {pd[i].x, pd[i].y, pd[i].z, pd[i].w};
                qb2_store4(PD, BW4_P, (idx >> 6) * BW4_PROW + 8 * (idx & 63), x);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int idx = ht_tid + 256 * i;
                const float x[4] = {pc[i].x, pc[i].y, pc[i].z, pc[i].w};
                qb2_store4(CS, BW4_C, (idx >> 4) * BW4_CROW + 8 * (idx & 15), x);
            }
        };
        request(0);
        stage(smem);
        lds_barrier();                                              // (1) block 0 is parked
        for (int blk = 0; blk < nblk; ++blk) {
            if (blk + 1 < nblk) request(blk + 1);
            lds_barrier();                                          // (A of blk)
            if (blk + 1 < nblk) stage(smem + ((blk + 1) & 1) * BW4_BUF);      // (that buffer was last read before B of blk - 1)
            lds_barrier();                                          // (B of blk)
        }
        return;
    }
    // =============================== computing waves ===============================
    // the W1 slice of the wave's columns and hidden half as the B operand of dX (K = hidden unit), split once
    Split3 w1p[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wv[j] = P[a.off.iqn_w1 + (int64_t)(128 * hh + 32 * kb + 8 * g + j) * E_DIM + n];
        w1p[kb] = split_bf16x3(wv);
    }
    const float g1 = LN ? P[a.off.iqn_ln1_g + n] : 1.f, be1 = LN ? P[a.off.iqn_ln1_b + n] : 0.f;
    const __amdgpu_buffer_rsrc_t rs_ph = __builtin_amdgcn_make_buffer_rsrc(a.ws.phis, 0, ((R + 15) / 16) * 16 * E_DIM * 4, 0x00020000);
    const int vo_ph = (4 * g * 16 + li) * 4;

    f32x4 accWphi[2], accW1[8];      // accWphi[c][r]: dWphi[n = 16 cs + 4g + r][k = 16 (2 hh + c) + li];  accW1[i][r]: dW1[h = 16 (8 hh + i) + 4g + r][n]
#pragma unroll
    for (int i = 0; i < 2; ++i) accWphi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_dg = 0.f, s_db = 0.f, s_dbphi = 0.f, de_acc = 0.f;
    struct RowData {
        f32x4 ph[2], mu[2], rs[2], c1[2], c2[2];
        float ev[2];
    };
    auto load_rows = [&](RowData &D, int blk) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r0 = row0 + blk * BW4_RB + 16 * t;
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
    lds_barrier();                                                  // (1)
    PRISM_STAMP(9);
    for (int blk = 0; blk < nblk; ++blk) {
        const char *cur = smem + (blk & 1) * BW4_BUF;
        const bool more = blk + 1 < nblk;
        const char *PD = cur, *CS = cur + 3 * BW4_P;
        // ---- phase 1: dX[m = 4g + r][n] partial over this wave's hidden half: A = the image's rows, B = w1p
        {
            u32x4 xa[2][3];
            auto xrd = [&](int tt, int kb, u32x4 (&o)[3]) __attribute__((always_inline)) {
                const char *ar = PD + (16 * tt + li) * BW4_PROW + 256 * hh + 64 * kb + 16 * g;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) o[pl] = *reinterpret_cast<const u32x4 *>(ar + pl * BW4_P);
            };
            xrd(0, 0, xa[0]);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f32x4 adx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    if (kb + 1 < 4) xrd(t, kb + 1, xa[(kb + 1) & 1]);
                    else if (t == 0) xrd(1, 0, xa[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    adx = mfma_split(xa[kb & 1][0], xa[kb & 1][1], xa[kb & 1][2], w1p[kb], adx);
                    __builtin_amdgcn_sched_barrier(0);
                }
                xch[(wq * 2 + t) * 64 + lane] = adx;
            }
        }
        lds_barrier();                                              // (A) both halves of every pair are parked
        float xs8[8], dp8[8];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r0 = row0 + blk * BW4_RB + 16 * t;
            const f32x4 p0 = xch[((wq & 2) * 2 + t) * 64 + lane], p1 = xch[((wq | 1) * 2 + t) * 64 + lane];
            const f32x4 adx = p0 + p1;                              // (hidden half 0 + hidden half 1 on both waves of the pair)
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
            // d e[b][n]: sum over the T rows of a sample (written by the pair's first wave)
            const int bsm = (r0 + 4 * g) / T;
            if (T == 4) {
                if (hh == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
            } else if (T == 8) {
                dep += __shfl_xor(dep, 16, 64);
                if (hh == 0 && (g & 1) == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
            } else {
                dep += __shfl_xor(dep, 16, 64);
                dep += __shfl_xor(dep, 32, 64);
                de_acc += dep;
                if (((r0 + 16) % T) == 0) {
                    if (hh == 0 && g == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = de_acc;
                    de_acc = 0.f;
                }
            }
        }
        if (more) load_rows(D, blk + 1);          // (this block's row data is consumed)
        const Split3 X = split_bf16x3(xs8), DP = split_bf16x3(dp8);
        // ---- phase 2: this wave's share of dWphi (two cos column blocks) and dW1 (eight hidden tiles), operand groups walked
        // one ahead of their MFMAs (bwd3_kernels.h)
        auto grp = [&](int q, u32x4 (&o)[3]) __attribute__((always_inline)) {
            if (q < 2) {
                const int x0 = 16 * (2 * hh + q);
                o[0] = bw3_tr_rows(CS, BW4_CROW, x0, lane);
                o[1] = bw3_tr_rows(CS + BW4_C, BW4_CROW, x0, lane);
                o[2] = bw3_tr_rows(CS + 2 * BW4_C, BW4_CROW, x0, lane);
            } else {
                const int x0 = 16 * (8 * hh + q - 2);
                o[0] = bw3_tr_rows(PD, BW4_PROW, x0, lane);
                o[1] = bw3_tr_rows(PD + BW4_P, BW4_PROW, x0, lane);
                o[2] = bw3_tr_rows(PD + 2 * BW4_P, BW4_PROW, x0, lane);
            }
        };
        u32x4 og[2][3];
        grp(0, og[0]);
#pragma unroll
        for (int q = 0; q < 10; ++q) {
            if (q + 1 < 10) grp(q + 1, og[(q + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 2) {
                Split3 cbq;
                cbq.hi = og[q & 1][0];
                cbq.mid = og[q & 1][1];
                cbq.lo = og[q & 1][2];
                accWphi[q] = mfma_split(DP.hi, DP.mid, DP.lo, cbq, accWphi[q]);
            } else {
                accW1[q - 2] = mfma_split(og[q & 1][0], og[q & 1][1], og[q & 1][2], X, accW1[q - 2]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        lds_barrier();                                              // (B)
    }
    PRISM_STAMP(10);
    // ---- this wave's part of the chunk's slab: nobody else holds these (column, cos block / hidden tile) products
    float *slab = a.ws.slabs + (int64_t)rc * a.slab;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(accWphi[c][r], slab + (int64_t)(16 * cs + 4 * g + r) * K_BASIS + 16 * (2 * hh + c) + li);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __builtin_nontemporal_store(accW1[i][r], slab + SLAB_W1 + (int64_t)(16 * (8 * hh + i) + 4 * g + r) * E_DIM + n);
    s_dg += __shfl_xor(s_dg, 16, 64);
    s_dg += __shfl_xor(s_dg, 32, 64);
    s_db += __shfl_xor(s_db, 16, 64);
    s_db += __shfl_xor(s_db, 32, 64);
    s_dbphi += __shfl_xor(s_dbphi, 16, 64);
    s_dbphi += __shfl_xor(s_dbphi, 32, 64);
    if (g == 0 && hh == 0) {
        slab[E_DIM * K_BASIS + n] = s_dbphi;
        if (LN) {
            slab[E_DIM * K_BASIS + E_DIM + n] = s_dg;
            slab[E_DIM * K_BASIS + 2 * E_DIM + n] = s_db;
        }
    }
    PRISM_STAMP(12);
}

}  // namespace prism
