// IQN TD-update kernels for gfx950 (fp32 MFMA 16x16x4, wave64).
//
// Restates /root/reference/prism/agents/models/iqn_model.py:48-201 (+ ffnn_model.py:61-76,
// minatar_cnn_model.py:43-46) as these kernels and block routines:
//
//   embed      conv3x3+ReLU of obs / next_obs -> e_cur, e_next [B,1024]; extra workgroups compute
//              u = W1 g1, v = W1 beta1 (used to get LayerNorm-backward row sums without dX) and repack
//              the GEMM weights in MFMA-fragment order (in the fused step: part of step_front_kernel)
//   tile_fwd   one 16-row tile of (sample, tau) rows per workgroup, whole rows on chip:
//              cos basis -> phi GEMM (K=64) -> ReLU -> Hadamard with e -> LayerNorm(1024) ->
//              trunk GEMM (K=1024) -> ReLU -> LayerNorm(128) -> head -> Z[16, A]; the current-state
//              tiles then run the loss of their samples (iqn_loss_tile) once the next-state tiles of
//              the same samples have published their rows
//   loss       the same loss as a kernel of its own (one workgroup per sample) for shapes where a tile
//              does not hold whole samples: argmax / n-step target / pairwise quantile-Huber tile,
//              dL/dq, head + LayerNorm(128) backward -> dpre1 and the per-row scalars
//   bwd        column-sliced backward: workgroup (16 embed columns x a row chunk) recomputes its
//              columns of phi / LN from saved row statistics and accumulates dWphi, dW1, dLN, de
//              with NO cross-workgroup reduction (per-chunk slabs); IQN-only models: conv-backward taps
//   post roles conv-backward partials / fold, the small tensors (b1, LN2, W2, b2), slab sums -> flat
//              gradient + sum-of-squares partials (iqn_post_kernel in step_kernels.h)
//
// Rows are SAMPLE-major inside the workspace (row = b*T + t); the reference's tau-major order
// (row = t*B + b, iqn_model.py:70) only matters for how tau inputs are indexed.
#pragma once
#include "common.h"

namespace prism {

constexpr int E_DIM = 1024;   // 16 * 8 * 8 (minatar_cnn_model.py:14)
constexpr int K_BASIS = 64;   // iqn_n_basis_elements
constexpr int MAX_H = 256;    // hidden widths covered: 128 (MINATAR_CONFIG) and 256 (ablation presets)
constexpr int CS = K_BASIS + 4;   // LDS row stride (floats) of a cos tile, +4 breaks the 16-row bank alias
constexpr float LN_EPS = 1e-5f;
// one forward tile's prepared basis: [3 planes hi | mid | lo][16 rows][32 bf16 pairs] + [16] tau (float bits)
constexpr int CP_TILE = 3 * 16 * 32 + 16;
constexpr float PI_F = 3.14159274101257324f;  // fp32(np.pi), the scalar torch multiplies by

struct IqnPass {
    const float *params;  // weight set for this pass (online or target flat buffer)
    const float *wpk;     // stream-packed weights of that set (pack_weights_block / pack_head_w1_block)
    const float *uv;      // u | v of that set: IQN [2][H]; Q heads [heads][2][H]
    const float *e;       // [B][E] embedded observations feeding this pass
    const float *e2;      // kind 2: embedded NEXT observations (the tile's next-state rows)
    const float *tau_in;  // [T*B] tau-major, or NULL -> Philox
    const float *tau_in2; // kind 2: next-state quantile samples
    const unsigned int *cospk; // tiles of this pass prepared by cos_basis_block (CP_TILE dwords each), or NULL: the tile draws tau
                               // and evaluates the basis itself
    float *z_out;         // [B*T][A] sample-major
    float *z_out2;        // kind 2: next-state estimates
    int T;
    int n_tiles;
    int save;             // current-state pass: keep what backward needs
    int stream_id;        // 0 cur, 1 next-online, 2 next-target
    int kind;             // 0: IQN quantile rows of one pass, 1: Q-head rows, 2: IQN current + next rows of whole samples
};

struct IqnWs {           // workspace pointers (device)
    float *e_cur, *e_next;
    float *uv;           // [2 sets][UV_ROWS][Hi]: u = W1 g1 | v = W1 beta1 | u of each 128-column K slice, online / target IQN trunk
    float *wpk[2];       // [online, target] stream-packed {phi_w, w1 * ln1_g}
    float *cosb, *mu1, *rstd1, *pre1, *xhat2, *rstd2;
    unsigned int *cospk; // [IQN tiles of the learner's passes][CP_TILE]: quantile samples + cos basis as bf16 pieces (cos_basis_block)
    float *phis;         // ReLU(phi) of the current-state rows for the backward: [row / 16][column / 16][16 rows][16 columns]
    float *zcur, *zon, *ztg;
    float *dq, *c1, *c2, *dpre1, *Sb, *Pb, *Db, *lossw;
    float *de_iqn;
    // Q heads (rows indexed head*B + sample)
    float *q_mu1, *q_rstd1, *q_pre1, *q_xhat2, *q_rstd2;
    float *zq_cur, *zq_on, *zq_tg;
    float *q_dq, *q_c1, *q_c2, *q_dpre1, *q_lossw;
    unsigned short *q_pp;   // [3 planes][heads * B][Hq] bf16 pieces of q_dpre1 (hi | mid | lo), written by the Q loss for qh_bwd2_kernel
    unsigned short *q_xp;   // [3 planes][B][E] bf16 pieces of xhat = (e - mean) rstd of the current observations (LN off: e)
    float *q_uv;         // [2 sets][heads][UV_ROWS][Hq]
    float *q_kappa;      // [heads][Q_NORM_PARTS] partial ||theta_h||^2
    float *q_wpk[2];     // [online, target] packed W1 * ln1_g of every head
    float *de_q;         // [heads][B][E] embedding gradient of each Q head
    float *q_slabs;      // [heads][q_slab]
    float *slabs;        // [n_chunks][slab]
    float *convpart;     // conv-backward partial rows
    float *normpart;     // [NORM_SLOTS]
    float *sib;          // [TREE_MAX_LEVELS][B] float2: siblings of the sampled paths (front -> writeback)
    float *wb_plan;      // [B] int4: prepared priority writeback (post -> back)
    unsigned int *ticket;   // [8] {adam, conv, target set packed, sibling-record state | grid barrier of the fused tail:
                            // 64-bit arrival count, -, sticky status (GRID_STATUS_*)}, zero-initialised by the caller;
                            // adam / conv / sibling reset themselves
};

constexpr int NORM_SLOTS = 2560;
// rows of a u/v table: u (all columns) | v | u restricted to K slice w = embed columns [128 w, 128 w + 128), w = 0..7 -- the
// slice a forward wave streams.  The forward shifts its LayerNorm(1024) input by a per-(wave, row) constant before the
// trunk product (fwd_kernels.h); the slice sums put that constant's contribution back.
constexpr int UV_SLICES = 8, UV_ROWS = 2 + UV_SLICES;
// gradient slab of one row chunk, in flat-parameter order: phi_w | phi_b | [ln1_g | ln1_b] | w1
__host__ __device__ inline int iqn_slab_floats(int H, int ln) { return E_DIM * K_BASIS + E_DIM + (ln ? 2 * E_DIM : 0) + H * E_DIM; }

struct BwdGeom {
    int tsh, unit, units_total, nw_sh, share, cpl, spw, tpw, main_lds;
};

struct IqnArgs {
    IqnPass pass[6];
    int n_pass;
    int B, A, C, T, Tn;
    int Bt;                // row stride of the tau-major quantile-sample arrays (== B except in the acting forward)
    int Hi, Hq;            // hidden width of the IQN trunk / of the Q heads
    int ln;                // use_layer_norm
    int slab, q_slab;      // floats per gradient slab
    int n_chunks;          // row chunks of the backward
    int has_target, double_q, propagate_grad;
    int use_iqn, n_heads;  // Q ensemble: 0 = none
    int conv_in_bwd;       // conv-backward partials are produced by the tail of iqn_bwd_kernel (bwd_conv_ok)
    int conv_rows;         // ... as this many partial rows per (row chunk, channel): 4 column slices, or 1 (already added)
    BwdGeom bg;
    int local_loss;        // the IQN loss ran inside the forward tiles (kind 2): no iqn_loss_kernel launch
    int cos_tiles;         // IQN forward tiles whose quantile samples + cos basis the embed / front launch prepares (cos_basis_block)
    int head_layers;       // 2: [LN]-Linear-ReLU-[LN]-Linear heads (MFMA path); 1: single Linear DQN head
    float q_w, theil_coef;
    int q_de_slots;        // slots of ws.de_q that hold a share of the Q heads' embedding gradient (one per head, or the
                           // two K halves of qh_bwd2_kernel)
    int q_pieces;          // the Q loss also leaves dpre1 / xhat as bf16 pieces in ws.q_pp / ws.q_xp (experiments; 0)
    int squish;            // PRISM_SQUISH_*: value squish of the TD target (common.h td_target)
    int split;             // forward GEMMs on the bf16 matrix pipe (three-piece operands, common.h); packed copies laid out for it
    int dbg;               // experiment switches (PRISM_DBG env), 0 in production
    unsigned long long *stamps;   // diagnostic builds only: [block][64] shader-clock stamps (dbg & 8)
    float huber_k, dist_w;
    prism_param_offsets off;
    const float *params;
    const float *target_params;
    const float *obs, *next_obs, *reward, *gamma, *per_weights;
    const uint8_t *nonterminal;
    const int64_t *action;
    uint64_t seed, offset;
    const uint64_t *rng;   // device counters {PER draws, tau draws} or NULL
    uint64_t *act_rng;     // acting forward replayed from a hipGraph: the device counter of its quantile draws, advanced by
    uint64_t act_inc;      // act_inc by the embed launch (the tiles then read counter - act_inc through `rng` / `offset`)
    float *tau_out;        // [3][maxT*B] or NULL
    int maxT;
    float *out_dl, *out_ql, *out_td, *out_scalars;
    float *grads;
    IqnWs ws;
};

// ------------------------------------------------------------------------------------------
// Stream-packed weight copies.  A forward tile consumes its weights as MFMA A operands, one 16-column
// step of the embedding at a time; the copy lays them out in exactly that order, so each wave of a tile
// reads ONE linear stream, 1 KB per wave instruction (reading the canonical [n][k] matrices in operand
// shape touches 16 rows 4 KB apart per instruction and streams at half the rate: tools/ubench/l2_stream.hip).
// LayerNorm(1024)'s scale is folded into the trunk weight here (fwd_kernels.h).  The copies are rebuilt
// from the canonical parameters by extra blocks of the embed/front launch every step (off the critical
// path), so they can never go stale.  float4 index of the IQN copy, SL = 4 + H/16 slots per step:
//   [(step = n >> 4) * SL + slot][lane = g*16 + li]
//     slot q < 4      : Wphi[n = 16 step + li][16 q + 4 g .. +3]
//     slot 4 + ht     : W1[h = 16 ht + li][n = 16 step + 4 g .. +3] * g1[n]
// (wave w of a tile owns steps 8w .. 8w+7).  A Q head's copy is the same without the phi slots.
// ------------------------------------------------------------------------------------------
__host__ __device__ inline int iqn_pack_floats(int H) { return E_DIM * K_BASIS + H * E_DIM; }
__host__ __device__ inline int iqn_pack_blocks(int H) { return iqn_pack_floats(H) / 4 / 256; }   // one float4 per thread

__device__ __forceinline__ void pack_weights_block(const float *__restrict__ P, const prism_param_offsets &off, int H, int ln,
                                                   float *__restrict__ pk, int blk, int tid) {
    const int p4 = blk * 256 + tid;              // packed float4 index
    const int lane = p4 & 63, li = lane & 15, g = lane >> 4;
    const int SL = 4 + H / 16;
    const int grp = p4 >> 6, step = grp / SL, slot = grp - step * SL;
    float4 v;
    if (slot < 4) {
        v = *reinterpret_cast<const float4 *>(P + off.phi_w + (int64_t)(16 * step + li) * K_BASIS + 16 * slot + 4 * g);
    } else {
        const int n0 = 16 * step + 4 * g;
        v = *reinterpret_cast<const float4 *>(P + off.iqn_w1 + (int64_t)(16 * (slot - 4) + li) * E_DIM + n0);
        if (ln) {
            const float4 gg = *reinterpret_cast<const float4 *>(P + off.iqn_ln1_g + n0);
            v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
        }
    }
    reinterpret_cast<float4 *>(pk)[p4] = v;   // (plain store: every CU of the next launch streams these)
}

// ---- the same weights pre-split for the bf16 matrix pipe (fwd_kernels.h, stream_split; common.h mfma_split) -------------
// Wave w of a tile owns embed columns [128 w, 128 w + 128) as four 32-column double steps; per (wave, double step) the copy
// holds G = 4 + H/16 operand groups (Q heads: H/16), each as three u32x4 planes {hi, mid, lo} of packed bf16 per lane:
//   group nt2 * 2 + kb  (< 4): A operand of the phi product   Wphi[n = n0 + 16 nt2 + li][k = 32 kb + 8 g + j],  j = 0..7
//   group 4 + ht            : A operand of the trunk product  (g1 * W1)[h = 16 ht + li][n = n0 + perm(g, j)]
//     perm(g, j) = 4 g + j (j < 4), 16 + 4 g + (j - 4) (j >= 4): K index j of the lane group IS the accumulator register of
//     the phi product that holds that column (n-tile j >> 2, row 4 g + (j & 3)) -- no lane movement between the products.
// u32x4 index of (wave w, double step ds, group, plane): ((4 w + ds) * G + group) * 3 + plane) * 64 + lane.  6 bytes per weight.
__host__ __device__ inline int split_groups(int H, bool phi) { return (phi ? 4 : 0) + H / 16; }
__host__ __device__ inline int iqn_pack_split_floats(int H) { return 32 * split_groups(H, true) * 3 * 64 * 4; }      // in floats (16 B per u32x4)
__host__ __device__ inline int q_pack_split_floats(int H) { return 32 * split_groups(H, false) * 3 * 64 * 4; }       // per head
__host__ __device__ inline int iqn_pack_split_blocks(int H) { return 32 * split_groups(H, true) * 64 / 256; }         // one (group, lane) per thread
__host__ __device__ inline int q_pack_split_blocks_per_head(int H) { return 32 * split_groups(H, false) * 64 / 256; }

// `wbase`: W1 of the tensor (row stride E_DIM), `g1`: LayerNorm scale or NULL, `wphi`: phi weight or NULL (Q heads);
// `al`: the tensors are 16-byte aligned (the IQN's are, a Q head's are not)
__device__ __forceinline__ void pack_split_block(const float *__restrict__ wphi, const float *__restrict__ wbase,
                                                 const float *__restrict__ g1, int H, bool al, float *__restrict__ pk, int blk, int tid) {
    const int t = blk * 256 + tid, lane = t & 63, li = lane & 15, g = lane >> 4;
    const int G = split_groups(H, wphi != nullptr), gi = t >> 6, group = gi % G, wd = gi / G, n0 = 32 * wd;
    float x[8];
    auto ld4f = [&](const float *p, float *dst) {
        if (al) {
            const float4 v = *reinterpret_cast<const float4 *>(p);
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        } else {
            dst[0] = p[0]; dst[1] = p[1]; dst[2] = p[2]; dst[3] = p[3];
        }
    };
    const int P0 = wphi ? 4 : 0;
    if (group < P0) {
        const float *src = wphi + (int64_t)(n0 + 16 * (group >> 1) + li) * K_BASIS + 32 * (group & 1) + 8 * g;
        ld4f(src, x);
        ld4f(src + 4, x + 4);
    } else {
        const float *src = wbase + (int64_t)(16 * (group - P0) + li) * E_DIM + n0 + 4 * g;
        ld4f(src, x);
        ld4f(src + 16, x + 4);
        if (g1) {
            float gg[8];
            ld4f(g1 + n0 + 4 * g, gg);
            ld4f(g1 + n0 + 16 + 4 * g, gg + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] *= gg[j];
        }
    }
    const Split3 s = split_bf16x3(x);
    u32x4 *dst = reinterpret_cast<u32x4 *>(pk) + (size_t)gi * 3 * 64 + lane;
    dst[0] = s.hi;
    dst[64] = s.mid;
    dst[128] = s.lo;
}

// ------------------------------------------------------------------------------------------
// embed: blocks [0,B) conv(obs) online; [B,2B) conv(next_obs) with target-or-online weights;
//        blocks [2B, 2B + H/4) compute u,v.
// ------------------------------------------------------------------------------------------
// u[h] = sum_n W1[h][n] g1[n] (whole and per K slice),  v[h] = sum_n W1[h][n] beta1[n]   (one wave per h)
__device__ __forceinline__ void iqn_uv_block(const IqnArgs &a, int set, int h, int lane) {
    const float *P = set ? a.target_params : a.params;
    const float *W1 = P + a.off.iqn_w1 + (int64_t)h * E_DIM;
    const float *g1 = P + a.off.iqn_ln1_g, *b1 = P + a.off.iqn_ln1_b;
    float su[4], sv = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                 // columns 4 lane + 256 i: K slice (lane >> 5) + 2 i
        const int n = lane * 4 + 256 * i;
        const float4 w = *reinterpret_cast<const float4 *>(W1 + n);
        const float4 g = *reinterpret_cast<const float4 *>(g1 + n);
        const float4 bb = *reinterpret_cast<const float4 *>(b1 + n);
        su[i] = w.x * g.x + w.y * g.y + w.z * g.z + w.w * g.w;
        sv += w.x * bb.x + w.y * bb.y + w.z * bb.z + w.w * bb.w;
    }
    float *uv = a.ws.uv + (size_t)set * UV_ROWS * a.Hi;
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < UV_SLICES; ++s) {
        const float t = wave_sum(((lane >> 5) == (s & 1)) ? su[s >> 1] : 0.f);
        if (lane == 0) uv[(2 + s) * a.Hi + h] = t;
        tot += t;                                 // (slices in order: what the forward's fold adds up as well)
    }
    sv = wave_sum(sv);
    if (lane == 0) {
        uv[h] = tot;
        uv[a.Hi + h] = sv;
    }
}

// Conv2d(C->16, 3x3) + ReLU + channel-major flatten of one NHWC observation held in LDS, on the matrix core:
// out[16 channels][64 positions] = W[16][K] x patches[K][64], K = 9 C, as 16x16x4 fp32 MFMAs -- four waves, one 16-position
// tile each, K / 4 MFMAs per wave.  (As fmaf chains, one output position x four channels a thread, the convolution of one
// observation took 2.0-3.4 k cycles and was the largest single piece of the front launch; the MFMA form reads 2 x K / 4
// words per lane and issues K / 4 matrix instructions: ~0.5 k.)  K runs (ci, dy, dx)-major like the stored weights, so
// W[m][k] = conv_w[m * 9 C + k]; the bias is added to the finished sum.
//   s_obs   CHANNEL-major image of the observation, s_obs[ci * 100 + y * 10 + x] (obs_to_lds: the stored order is
//           position-major with C interleaved channels, which puts a wave's reads on 8 of the 32 banks)
//   s_w     [16][conv_kp(C)]: the weights of an output channel, zero-padded to a multiple of four
__host__ __device__ constexpr int conv_kp(int C) { return (9 * C + 3) & ~3; }
constexpr int CONV_W_FLOATS = 16 * conv_kp(10);
__device__ __forceinline__ int conv_w_slot(int i, int C) { return i + (i / (9 * C)) * (conv_kp(C) - 9 * C); }
// every thread of the workgroup: weights + padding + bias of one parameter set into LDS (P = the set's flat parameters)
__device__ __forceinline__ void conv_w_to_lds(const float *__restrict__ P, const prism_param_offsets &off, int C, float *s_w, float *s_b,
                                              int tid, int nthreads) {
    const int KP = conv_kp(C), K = 9 * C;
    for (int i = tid; i < 16 * KP; i += nthreads) {
        const int m = i / KP, k = i - m * KP;
        s_w[i] = k < K ? P[off.conv_w + m * K + k] : 0.f;
    }
    if (tid < 16) s_b[tid] = P[off.conv_b + tid];
}
// obs_to_lds(): the four consecutive stored elements 4 t .. 4 t + 3 to their places.
__device__ __forceinline__ void obs_to_lds(float *s_obs, const float4 v, int t, int C) {
    const unsigned int m = 65536u / (unsigned int)C + 1u;             // e / C == (e * m) >> 16 for e < 1000, C <= 10
    const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned int e = 4u * (unsigned int)t + j, pos = (e * m) >> 16, ci = e - pos * (unsigned int)C;
        s_obs[ci * 100u + pos] = x[j];
    }
}
// waves 0..3 of the caller (tid < 256).  Branch-free: all operand words of a run of K steps are requested before the
// first MFMA (written with a bounds test per step the loop was one LDS round trip per matrix instruction: 2.2 k cycles
// for nine of them, tools/ubench/conv_embed.hip); a padded step reads a valid word against a zero weight.
template <int KSTEPS>      // the number of K steps when it is known at compile time (C = 4: nine), else 0
__device__ __forceinline__ void conv_embed_rows_k(const float *s_obs, const float *s_w, const float *s_b, int C,
                                                  float *__restrict__ dst, int tid) {
    const int lane = tid & 63, j = tid >> 6, n = lane & 15, kq = lane >> 4;
    const int KP = KSTEPS ? 4 * KSTEPS : conv_kp(C), K = 9 * C;
    const int y = 2 * j + (n >> 3), x = n & 7;
    const float *wrow = s_w + n * KP + kq;                 // A operand: lane supplies W[m = lane & 15][k = 4 s + kq]
    const float *orow = s_obs + y * 10 + x;                // B operand: patch element k of position 16 j + n
    auto patch = [&](int s4) {
        int k = s4 + kq;
        k = k < K ? k : K - 1;
        const int ci = (k * 7282) >> 16;                   // k / 9 for k < 1000
        const int t = k - 9 * ci, dy = (t * 11) >> 5;      // t / 3 for t < 9
        return orow[ci * 100 + dy * 10 + (t - 3 * dy)];
    };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KSTEPS > 0) {
        float av[KSTEPS], bv[KSTEPS];
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            av[s] = wrow[4 * s];
            bv[s] = patch(4 * s);
        }
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) acc = mfma16(av[s], bv[s], acc);
    } else {
        for (int s4 = 0; s4 < KP; s4 += 12) {              // runs of three steps (KP = 4 (9 C + 3) / 4; at most one run reads past it)
            float av[3], bv[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int q = s4 + 4 * u < KP ? s4 + 4 * u : KP - 4;
                av[u] = s4 + 4 * u < KP ? wrow[q] : 0.f;
                bv[u] = patch(q);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) acc = mfma16(av[u], bv[u], acc);
        }
    }
    // D: lane holds channels 4 kq + r at position 16 j + n
#pragma unroll
    for (int r = 0; r < 4; ++r)
        __builtin_nontemporal_store(fmaxf(acc[r] + s_b[4 * kq + r], 0.f), &dst[(4 * kq + r) * 64 + 16 * j + n]);
}
__device__ __forceinline__ void conv_embed_rows(const float *s_obs, const float *s_w, const float *s_b, int C,
                                                float *__restrict__ dst, int tid) {
    if (C == 4) conv_embed_rows_k<9>(s_obs, s_w, s_b, C, dst, tid);
    else conv_embed_rows_k<0>(s_obs, s_w, s_b, C, dst, tid);
}

__device__ void embed_extra_block(const IqnArgs &a, int x, float *s_red);

__global__ __launch_bounds__(256) void iqn_embed_kernel(IqnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_obs[1024];
    __shared__ __attribute__((aligned(16))) float s_w[CONV_W_FLOATS];
    __shared__ float s_b[16];
    const int B = a.B, C = a.C;
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (a.act_rng && blk == 0 && tid == 0) *a.act_rng += a.act_inc;      // (the forward tiles run in the NEXT launch)
    if (blk >= 2 * B) {
        embed_extra_block(a, blk - 2 * B, s_b);      // defined in step_kernels.h (front_extra_block)
        return;
    }
    const bool is_next = blk >= B;
    const int b = is_next ? blk - B : blk;
    const float *P = (is_next && a.has_target) ? a.target_params : a.params;
    const float *src = (is_next ? a.next_obs : a.obs) + (int64_t)b * 100 * C;
    float *dst = (is_next ? a.ws.e_next : a.ws.e_cur) + (int64_t)b * E_DIM;
    if (tid < 25 * C) obs_to_lds(s_obs, reinterpret_cast<const float4 *>(src)[tid], tid, C);
    conv_w_to_lds(P, a.off, C, s_w, s_b, tid, 256);
    __syncthreads();
    conv_embed_rows(s_obs, s_w, s_b, C, dst, tid);
}

// Workgroup barrier that orders LDS traffic only: global loads issued before it stay in flight
// (a plain __syncthreads() also waits for vmcnt(0), which would serialise every weight prefetch).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// shader-clock stamps (s_memtime: per-XCD counters, only differences inside one workgroup mean anything) plus,
// in slots 60.. the chip-wide 100 MHz real-time counter of the same moments (for spans across workgroups)
#define PRISM_LOOP_STAMP(k)                                                                \
    do {                                                                                   \
        if ((a.dbg & 16) && threadIdx.x == 0 && blockIdx.x >= 256)                         \
            a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime();        \
    } while (0)
#define PRISM_STAMP2(k)                                                                    \
    do {                                                                                   \
        if ((a.dbg & 8) && threadIdx.x == 0)                                               \
            a.stamps[(size_t)(2048 + blockIdx.x) * 64 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#define PRISM_STAMP(k)                                                                     \
    do {                                                                                   \
        if ((a.dbg & 8) && threadIdx.x == 0) {                                             \
            a.stamps[(size_t)blockIdx.x * 64 + (k)] = __builtin_amdgcn_s_memtime();        \
            a.stamps[(size_t)blockIdx.x * 64 + 32 + (k)] = __builtin_amdgcn_s_memrealtime(); \
        }                                                                                  \
    } while (0)

constexpr int LOSS_WAVES = 8;    // waves of the workgroup that evaluates one sample (16 for T >= 32: its rows are walked wave by
                                 // wave, and with few samples per launch more threads per sample cost nothing)
__host__ __device__ constexpr int loss_waves(int T) { return T >= 32 ? 16 : LOSS_WAVES; }

// ------------------------------------------------------------------------------------------
// loss (stand-alone form, iqn_model.py:95-201): one workgroup (8 waves) per sample, for the shapes whose
// loss cannot finish inside a forward tile (target network, T > 8, T' != T).  Wave 0 evaluates the
// pairwise quantile-Huber tile while every wave already has the saved activations of its rows in
// flight; then the T current-state rows are back-propagated through head + LayerNorm(H), rows strided
// over waves.  T, T' must divide 64.  `xhat2` holds the head Linear's input: LayerNorm output before the
// affine (LN) or ReLU(pre1) (no LN).
// ------------------------------------------------------------------------------------------
// LDS of the stand-alone loss (floats): Z of the sample's rows (current / online-next / target-next), four 64-entry rows
// (target, prediction, tau, dq), one accumulator row per wave (at most 16 waves).  Everything is live throughout.
template <int H>
struct LossLds {
    static constexpr int AS = 2 * H + 4;
    static constexpr LdsRegion ZC{0, 64 * 16, LDS_ALWAYS}, ZO{64 * 16, 64 * 16, LDS_ALWAYS}, ZT{2 * 64 * 16, 64 * 16, LDS_ALWAYS};
    static constexpr LdsRegion Y{3 * 64 * 16, 64, LDS_ALWAYS}, Q{3 * 64 * 16 + 64, 64, LDS_ALWAYS};
    static constexpr LdsRegion TAU{3 * 64 * 16 + 128, 64, LDS_ALWAYS}, DQ{3 * 64 * 16 + 192, 64, LDS_ALWAYS};
    static constexpr LdsRegion ACC{3 * 64 * 16 + 256, 16 * AS, LDS_ALWAYS};
    static constexpr int TOTAL = ACC.off + ACC.size + 4;
    static constexpr LdsRegion ALL[] = {ZC, ZO, ZT, Y, Q, TAU, DQ, ACC};
    static_assert(lds_layout_ok(ALL, TOTAL), "loss kernel: LDS regions overlap");
};
template <int H>
__host__ __device__ constexpr int loss_lds_floats() { return LossLds<H>::TOTAL; }

template <int H, bool LN, int LW>
__device__ __forceinline__ void iqn_loss_body(const IqnArgs &a, const int b) {
    constexpr int LOSS_WAVES = LW, LOSS_RPW = 64 / LW;     // rows per wave (T <= 64)
    constexpr int KH = H / 64, AS = 2 * H + 4;
    __shared__ __attribute__((aligned(16))) float lds[loss_lds_floats<H>()];
    typedef LossLds<H> LD;
    static_assert(LW <= 16 && AS == LD::AS, "one accumulator row per wave");
    float *s_zc = lds + LD::ZC.off, *s_zo = lds + LD::ZO.off, *s_zt = lds + LD::ZT.off;
    float *s_y = lds + LD::Y.off, *s_q = lds + LD::Q.off, *s_tau = lds + LD::TAU.off, *s_dq = lds + LD::DQ.off;
    float *s_acc = lds + LD::ACC.off;             // [LOSS_WAVES][AS]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int B = a.B, A = a.A, T = a.T, Tn = a.Tn;
    const float kap = a.huber_k;

    // rows of this wave: t = w, w + 8, ...; start their loads now, they do not depend on the loss
    float xa[LOSS_RPW][KH], pa[LOSS_RPW][KH], rs[LOSS_RPW];
#pragma unroll
    for (int i = 0; i < LOSS_RPW; ++i) {
        const int t = w + LOSS_WAVES * i;
        rs[i] = 1.f;
        if (t < T) {
            const int64_t r = (int64_t)b * T + t;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                xa[i][k] = a.ws.xhat2[r * H + 64 * k + lane];
                pa[i][k] = a.ws.pre1[r * H + 64 * k + lane];
            }
            if (LN) rs[i] = a.ws.rstd2[r];
        }
    }
    const int act = (int)a.action[b];
    const float *P = a.params;
    const float *W2 = P + a.off.iqn_w2 + (int64_t)act * H;
    float w2a[KH], ua[KH], va[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        const int h = 64 * k + lane;
        w2a[k] = LN ? W2[h] * P[a.off.iqn_ln2_g + h] : W2[h];          // d (head input) / dq
        ua[k] = LN ? a.ws.uv[h] : 0.f;
        va[k] = (LN ? a.ws.uv[H + h] : 0.f) + P[a.off.iqn_b1 + h];
    }
    for (int i = tid; i < T * A; i += 64 * LOSS_WAVES) s_zc[i] = a.ws.zcur[(int64_t)b * T * A + i];
    for (int i = tid; i < Tn * A; i += 64 * LOSS_WAVES) {
        s_zo[i] = a.ws.zon[(int64_t)b * Tn * A + i];
        s_zt[i] = a.ws.ztg[(int64_t)b * Tn * A + i];
    }
    // quantile samples of the current-state pass (tau_out slot 0 always holds them)
    if (tid < T) s_tau[tid] = a.tau_out[(int64_t)tid * B + b];
    __syncthreads();
    PRISM_STAMP(20);
    if (w == 0) {
        // a* = argmax_a mean_j Zon[j][a]  (first maximum wins, iqn_model.py:129-133): lane = action, each lane adds its
        // column in the sequential order (one thread doing all A * T' reads one after the other was a third of this
        // kernel at T' = 32), then the first maximum over the lanes
        float mean = 0.f;
        {
            const int la = lane < A ? lane : 0;
            float sacc = 0.f;
            for (int j = 0; j < Tn; ++j) sacc += s_zo[j * A + la];
            mean = sacc / (float)Tn;
        }
        int astar = 0;
        float bestv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), 0));
#pragma unroll
        for (int aa = 1; aa < 16; ++aa) {
            const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mean), aa));
            if (aa < A && m > bestv) {
                bestv = m;
                astar = aa;
            }
        }
        PRISM_STAMP(21);
        const float R = a.reward[b];
        const float dg = a.gamma[b] * (a.nonterminal[b] ? 1.0f : 0.0f);
        if (lane < Tn) s_y[lane] = td_target(a.squish, R, s_zt[lane * A + astar], dg);     // separate mul and add (iqn_model.py:141-148)
        if (lane < T) s_q[lane] = s_zc[lane * A + act];
        __builtin_amdgcn_wave_barrier();
        // pairwise quantile-Huber tile: pair p = j*T + t; a lane keeps a fixed t because T | 64
        float lsum = 0.f, gq = 0.f;
        const bool t_pow2 = (T & (T - 1)) == 0;
        const int tsh = 31 - __clz(T);
        const int t_l = t_pow2 ? lane & (T - 1) : lane % T;
        const float q_l = s_q[t_l], tau_l = s_tau[t_l];
        for (int p = lane; p < T * Tn; p += 64) {
            const int j = t_pow2 ? p >> tsh : p / T;
            const float d = s_y[j] - q_l;
            const float ad = fabsf(d);
            const float hub = (ad <= kap) ? 0.5f * (d * d) : kap * (ad - 0.5f * kap);
            const float wgt = fabsf(tau_l - (d < 0.f ? 1.0f : 0.0f));
            lsum += (wgt * hub) / kap;
            const float cl = fminf(fmaxf(d, -kap), kap);
            gq += (wgt * cl) / kap;
        }
        lsum = wave_sum(lsum);
        for (int o = 32; o >= T; o >>= 1) gq += __shfl_xor(gq, o, 64);
        const float dl = (lsum / (float)Tn) * a.dist_w;
        const float wb = a.per_weights ? a.per_weights[b] : 1.0f;
        const float scale = -(wb / (float)B) * a.dist_w / (float)Tn;
        if (lane < T) s_dq[lane] = gq * scale;
        if (lane == 0) {
            a.out_dl[b] = dl;
            if (a.out_td && a.n_heads == 0) a.out_td[b] = dl;  // IQN only: td_errors = distribution_loss (composite_model.py:138-139)
            a.ws.lossw[b] = dl * wb;
        }
    }
    __syncthreads();
    PRISM_STAMP(22);

    // head + LayerNorm(H) backward for this wave's rows
    float Sa[KH], Pa[KH], Dsum = 0.f;
#pragma unroll
    for (int k = 0; k < KH; ++k) Sa[k] = Pa[k] = 0.f;
#pragma unroll
    for (int i = 0; i < LOSS_RPW; ++i) {
        const int t = w + LOSS_WAVES * i;
        if (t < T) {
            const int64_t r = (int64_t)b * T + t;
            const float dq = s_dq[t];
            float da[KH], ga[KH], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                da[k] = dq * w2a[k];
                m1 += da[k];
                m2 += da[k] * xa[i][k];
            }
            if (LN) {
                m1 = wave_sum(m1) * (1.0f / H);
                m2 = wave_sum(m2) * (1.0f / H);
            }
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const float gv = LN ? rs[i] * (da[k] - m1 - xa[i][k] * m2) : da[k];
                ga[k] = pa[i][k] > 0.f ? gv : 0.f;
                a.ws.dpre1[r * H + 64 * k + lane] = ga[k];
                c1 += ga[k] * ua[k];
                c2 += ga[k] * (pa[i][k] - va[k]);
                Sa[k] += dq * xa[i][k];
                Pa[k] += ga[k];
            }
            if (LN) {
                c1 = wave_sum(c1);
                c2 = wave_sum(c2);
            }
            if (lane == 0) {
                if (LN) {
                    a.ws.c1[r] = c1;
                    a.ws.c2[r] = c2;
                }
                a.ws.dq[r] = dq;
            }
            Dsum += dq;
        }
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) {
        s_acc[w * AS + 64 * k + lane] = Sa[k];
        s_acc[w * AS + H + 64 * k + lane] = Pa[k];
    }
    if (lane == 0) s_acc[w * AS + 2 * H] = Dsum;
    __syncthreads();
    PRISM_STAMP(23);
    for (int o = tid; o < 2 * H + 1; o += 64 * LOSS_WAVES) {
        float t = 0.f;
#pragma unroll
        for (int ww = 0; ww < LOSS_WAVES; ++ww) t += s_acc[ww * AS + o];
        if (o < H) a.ws.Sb[(int64_t)b * H + o] = t;
        else if (o < 2 * H) a.ws.Pb[(int64_t)b * H + (o - H)] = t;
        else a.ws.Db[b] = t;
    }
}
template <int H, bool LN, int LW>
__global__ __launch_bounds__(64 * LW) void iqn_loss_kernel(IqnArgs a) { iqn_loss_body<H, LN, LW>(a, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// bwd: grid = (E/16 column slices) x n_chunks row chunks, 256 threads = 4 waves; each wave walks a contiguous
// run of 16-row tiles.  Per tile and wave, four products on the fp32 MFMA (16x16x4):
//   phi[m][n]   = cos[m][:] . Wphi[n][:]           (recomputed; A = cos rows,   B = Wphi slice)
//   dX[m][n]    = dpre1[m][:] . W1[:][n]           (A = dpre1 rows, B = W1 slice)
//   dWphi[n][k] += dphi[m][n] cos[m][k]            (A = dphi: the D registers of the elementwise step, B = cos)
//   dW1[h][n]   += dpre1[m][h] x[m][n]             (A = dpre1, B = x: D registers again)
// The two row operands (cos, dpre1) are needed once with the row on the lane (first pair) and once with the
// row on the k index (second pair).  Both forms are loaded straight from global memory as whole-row 16-byte
// pieces -- the second form assigns the basis / hidden index to (lane, register) as {4 j + c}, a permutation
// that only changes where each accumulator element is written at the end -- so no LDS transpose, no LDS
// traffic in the loop except the workgroup's fixed weight slices (parked there once).  Every register set is
// refilled for the NEXT tile right after its last use (rolling prefetch, no extra registers), and the
// instruction order is pinned: the elementwise step of row group r+1 runs in the shadow of the MFMAs of
// row group r.
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int bwd_acc(int H) { return 16 + H / 4 + 3; }                 // accumulators reduced across waves
// conv-backward partials ride along in the tile loop (IQN-only models: d e[b][n] is final per
// (sample, column) right there).  The lanes that hold one (sample, position) value -- `share` of
// them: 2 for T = 8, 4 for T >= 16 -- split the input channels; each lane owns all 9 kernel taps
// of C/share channels in registers.
constexpr int BWD_CONV_TAPS = 18;                   // taps per lane: 9 * C / share
constexpr int BWD_CONV_ROW = 96;                    // floats per partial row: 9C taps (C <= 10) + bias
__host__ __device__ constexpr int bwd_w_lds(int H) { return 16 * K_BASIS + 16 * H; }        // Wphi slice | W1 slice, operand order
__host__ __device__ inline int bwd_conv_share(int T) { return T == 4 ? 1 : (T == 8 ? 2 : 4); }
// samples one wave may own (its observation rows are staged in LDS: 4 image rows x 10 x C floats each)
__host__ __device__ inline int bwd_conv_spw(int B, int n_chunks) { return (B + 4 * n_chunks - 1) / (4 * n_chunks) + 1; }
// tiles one wave may walk (their d e values are parked in LDS for the tap pass behind the tile loop)
__host__ __device__ inline int bwd_tiles_per_wave(int B, int T, int n_chunks) {
    return (B * T / 16 + 4 * n_chunks - 1) / (4 * n_chunks) + (T > 16 ? T / 16 : 1);
}
constexpr int BWD_CONV_PRE = 12;                    // 16-byte LDS-DMA pieces per lane staging a wave's observation rows
// LDS of a workgroup: [ weight slices | observation rows | d e values ] during the tile loop, overlaid afterwards by
// the cross-wave reduction buffer; the conv tap fold has a small region of its own behind it
__host__ __device__ inline int bwd_loop_lds(int H, int B, int C, int T, int n_chunks, bool conv) {
    return bwd_w_lds(H) + (conv ? 4 * bwd_conv_spw(B, n_chunks) * 40 * C + 4 * bwd_tiles_per_wave(B, T, n_chunks) * 64 : 0);
}
__host__ __device__ inline int bwd_main_lds(int H, int B, int C, int T, int n_chunks, bool conv) {
    const int red = 4 * bwd_acc(H) * 64, loop = bwd_loop_lds(H, B, C, T, n_chunks, conv);
    return red > loop ? red : loop;
}
__host__ __device__ inline int bwd_lds_floats(int H, int B, int C, int T, int n_chunks, bool conv) {
    return bwd_main_lds(H, B, C, T, n_chunks, conv) + (conv ? 4 * BWD_CONV_ROW : 0);
}
__host__ __device__ inline bool bwd_conv_ok(int use_iqn, int n_heads, int propagate_grad, int T, int C, int B,
                                            int n_chunks, int H) {
    const int share = bwd_conv_share(T);
    return use_iqn && n_heads == 0 && propagate_grad && C % share == 0 && 9 * (C / share) <= BWD_CONV_TAPS &&
           C / share <= 2 && (C / share == 1 || C % 2 == 0) &&      // 1 channel, or an aligned pair, per lane
           9 * C < BWD_CONV_ROW && bwd_conv_spw(B, n_chunks) * 10 * C <= BWD_CONV_PRE * 64 &&
           bwd_lds_floats(H, B, C, T, n_chunks, true) * 4 <= (H == 128 ? 76 : 152) * 1024;     // two (one) workgroups per CU
}
// LDS of a backward workgroup (floats).  Compile-time part: the two weight slices; the rest is sized by the launch
// (batch, channels, T), so the host checks the whole table with lds_layout_ok() before it launches (bwd_lds_layout_ok).
// Phases: LOOP = tile loop + conv tap pass (weight slices, each wave's observation rows and d e values), REDUCE = cross-wave
// fold (overlays the loop's regions behind a workgroup barrier), TAPS = the conv partial fold (a region of its own).
constexpr unsigned BW_LOOP = 1u, BW_REDUCE = 2u, BW_TAPS = 4u;
__host__ __device__ constexpr LdsRegion bwd_wphi_region() { return LdsRegion{0, 16 * K_BASIS, BW_LOOP}; }
__host__ __device__ constexpr LdsRegion bwd_w1_region(int H) { return LdsRegion{16 * K_BASIS, 16 * H, BW_LOOP}; }
__host__ __device__ constexpr LdsRegion bwd_obs_region(int H, int w, int spw, int C) {
    return LdsRegion{bwd_w_lds(H) + w * (spw * 40 * C), spw * 40 * C, BW_LOOP};
}
__host__ __device__ constexpr LdsRegion bwd_dcv_region(int H, int w, int spw, int C, int tpw) {
    return LdsRegion{bwd_w_lds(H) + 4 * (spw * 40 * C) + w * tpw * 64, tpw * 64, BW_LOOP};
}
__host__ __device__ constexpr LdsRegion bwd_red_region(int H) { return LdsRegion{0, 4 * bwd_acc(H) * 64, BW_REDUCE}; }
__host__ __device__ constexpr LdsRegion bwd_tap_region(int main_lds) {
    return LdsRegion{main_lds, 4 * 4 * (BWD_CONV_TAPS + 1), BW_TAPS | BW_REDUCE | BW_LOOP};
}
static_assert(!lds_overlap(bwd_wphi_region(), bwd_w1_region(128)) && bwd_w1_region(256).off + bwd_w1_region(256).size == bwd_w_lds(256),
              "backward: weight slices");
static_assert(4 * 4 * (BWD_CONV_TAPS + 1) <= 4 * BWD_CONV_ROW, "backward: tap fold region");
inline bool bwd_lds_layout_ok(int H, int B, int C, int T, int n_chunks, bool conv) {
    const int spw = bwd_conv_spw(B, n_chunks), tpw = bwd_tiles_per_wave(B, T, n_chunks);
    const int main_lds = bwd_main_lds(H, B, C, T, n_chunks, conv), total = bwd_lds_floats(H, B, C, T, n_chunks, conv);
    LdsRegion r[12];
    r[0] = bwd_wphi_region();
    r[1] = bwd_w1_region(H);
    r[2] = bwd_red_region(H);
    for (int w = 0; w < 4; ++w) {
        r[3 + w] = conv ? bwd_obs_region(H, w, spw, C) : LdsRegion{0, 0, 0u};
        r[7 + w] = conv ? bwd_dcv_region(H, w, spw, C, tpw) : LdsRegion{0, 0, 0u};
    }
    r[11] = conv ? bwd_tap_region(main_lds) : LdsRegion{0, 0, 0u};
    return lds_layout_ok(r, total);
}
inline BwdGeom bwd_geometry(int H, int B, int C, int T, int n_chunks, bool conv) {
    BwdGeom g;
    g.tsh = -1;
    for (int k = 0; k < 16; ++k)
        if ((1 << k) == T) g.tsh = k;
    g.unit = T > 16 ? T / 16 : 1;
    g.units_total = (B * T / 16) / g.unit;
    g.nw_sh = 0;
    while ((1 << g.nw_sh) < 4 * n_chunks) ++g.nw_sh;
    g.share = bwd_conv_share(T);
    g.cpl = C / g.share;
    g.spw = bwd_conv_spw(B, n_chunks);
    g.tpw = bwd_tiles_per_wave(B, T, n_chunks);
    g.main_lds = bwd_main_lds(H, B, C, T, n_chunks, conv);
    return g;
}
#ifndef BWD_CHUNKS
#define BWD_CHUNKS 8
#endif

// DB = true: ONE wave per SIMD (one workgroup per CU, twice the register budget): the workgroup's weight slices
// live in registers, tiles are double-buffered -- every operand of tile i+1 is requested at the start of tile i --
// and the wave feeds the matrix pipe on its own (three independent chains, VALU in the MFMA shadows).
// DB = false: operands roll forward inside one register set (every set is refilled right after its last use), weight
// slices are read from LDS; width 128: two workgroups per CU share each SIMD, width 256: one (the accumulators
// alone take 80 registers; two tile sets do not fit).
template <int H, bool LN, bool DB>
__global__ __launch_bounds__(256, (H == 128 && !DB) ? 2 : 1) void iqn_bwd_kernel(IqnArgs a) {
    kernarg_prefetch<sizeof(IqnArgs)>();
    constexpr int NHT = H / 16, NU = H / 64, BWD_ACC = bwd_acc(H);
    constexpr int SLAB_W1 = E_DIM * K_BASIS + E_DIM + (LN ? 2 * E_DIM : 0);     // slab: phi_w | phi_b | [ln1_g | ln1_b] | w1
    typedef const f32x4 __attribute__((address_space(1))) *g4;
    typedef const float __attribute__((address_space(1))) *g1p;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    // workgroups go to the XCDs round-robin by index: with the row chunk in the low bits all column slices of a chunk
    // share one XCD (8 chunks) or two (4), so an L2 holds one chunk's rows (the operands streamed all through the loop)
    // instead of every chunk's; the weight slices (768 KB, read once) are what each L2 then holds in full
    const int rc = blockIdx.x % a.n_chunks, cs = blockIdx.x / a.n_chunks;
    const int n = cs * 16 + j;
    const int T = a.T;
    const int R = a.B * T;
    const int gw = rc * 4 + w;
    const int tile_begin = __builtin_amdgcn_readfirstlane(((a.bg.units_total * gw) >> a.bg.nw_sh) * a.bg.unit);
    const int tiles_per_wave = __builtin_amdgcn_readfirstlane(((a.bg.units_total * (gw + 1)) >> a.bg.nw_sh) * a.bg.unit) - tile_begin;
    const g1p P = (g1p)a.params;
    const g1p e_cur = (g1p)a.ws.e_cur;
    // the row operands are read through buffer descriptors: one loop-invariant 32-bit lane offset per stream,
    // the tile offset in a scalar register, everything else in the instruction's immediate
    const __amdgpu_buffer_rsrc_t rs_cos = __builtin_amdgcn_make_buffer_rsrc(a.ws.cosb, 0, R * K_BASIS * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dp = __builtin_amdgcn_make_buffer_rsrc(a.ws.dpre1, 0, R * H * 4, 0x00020000);
    // ReLU(phi) as the forward tiles left it: one 16 x 16 block (1 KB, row-major) per (tile, column slice)
    const __amdgpu_buffer_rsrc_t rs_ph = __builtin_amdgcn_make_buffer_rsrc(a.ws.phis, 0, ((R + 15) / 16) * 16 * E_DIM * 4, 0x00020000);
    const int vo_ph = (4 * g * 16 + j) * 4;       // row 4 g (+ r), column j of the block
    const __amdgpu_buffer_rsrc_t rs_mu = __builtin_amdgcn_make_buffer_rsrc(a.ws.mu1, 0, R * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rs = __builtin_amdgcn_make_buffer_rsrc(a.ws.rstd1, 0, R * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c1 = __builtin_amdgcn_make_buffer_rsrc(a.ws.c1, 0, R * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_c2 = __builtin_amdgcn_make_buffer_rsrc(a.ws.c2, 0, R * 4, 0x00020000);
    auto bload4 = [](__amdgpu_buffer_rsrc_t rs, int voff, int soff) __attribute__((always_inline)) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
        return __builtin_bit_cast(f32x4, v);
    };
    // Width 128: ReLU(phi) comes saved from the forward tiles (a sixth less matrix work here).  Width 256: the forward
    // has no registers left to keep it, the phi columns are recomputed from the saved cos basis as before.
    constexpr bool PHI_SAVED = H == 128;
    const int vo_ca = (j * K_BASIS + 4 * g) * 4, vo_da = (j * H + 4 * g) * 4;          // row on the lane
    (void)vo_ca;
    const int vo_cb = (4 * g * K_BASIS + 4 * j) * 4, vo_db = (4 * g * H + 4 * j) * 4;  // row on the k index
    const int vo_sc = 4 * g * 4;
    f32x4 *wphil = reinterpret_cast<f32x4 *>(smem + bwd_wphi_region().off);     // [q][lane]: Wphi[n][16q + 4g + jj]
    f32x4 *w1l = reinterpret_cast<f32x4 *>(smem + bwd_w1_region(H).off);        // [q][lane]: W1[h = 16q + 4g + jj][n]
    // conv-backward taps of this lane: channels [sub * cpl, (sub + 1) * cpl) x 3 x 3
    const int C = a.C, y0 = (cs & 3) * 2;
    const int cpl = a.bg.cpl, n_mine = a.conv_in_bwd ? 9 * cpl : 0;
    const int sub = T == 8 ? (g & 1) : g;
    // observation rows y0..y0+3 of this wave's samples: requested first thing, straight into LDS (LDS-DMA: 16 B
    // per lane, lane-linear destination, no registers); the barrier in front of the tile loop covers them
    const int ws_lo = (tile_begin * 16) / T, ws_n = n_mine ? (tiles_per_wave * 16) / T : 0;
    const int spw = a.bg.spw, tpw = a.bg.tpw;
    float *s_obs = smem + bwd_obs_region(H, w, spw, C).off;
    // ReLU-masked d e of every tile of this wave, consumed by the tap pass behind the tile loop (keeps the tap
    // accumulators out of the loop's register budget)
    float *s_dcv = smem + bwd_dcv_region(H, w, spw, C, tpw).off;
#pragma unroll
    for (int i = 0; i < BWD_CONV_PRE; ++i) {
        const int idx = lane + 64 * i;
        if (idx < ws_n * 10 * C) {
            const int s = idx / (10 * C), o4 = idx - s * 10 * C;
            const float *src = a.obs + ((int64_t)(ws_lo + s) * 100 + y0 * 10) * C + 4 * o4;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(s_obs + 256 * i), 16, 0, 0);
        }
    }
    PRISM_STAMP(8);
    // ---- the workgroup's weight slices -> LDS in B-operand order -------------------------------
    {
        if (!PHI_SAVED) {
            // Wphi rows cs*16 .. +15 are one contiguous 4 KB run: thread t takes float4 t
            const int jj = tid >> 4, f4 = tid & 15;
            const f32x4 v = *reinterpret_cast<g4>(P + a.off.phi_w + (int64_t)(cs * 16 + jj) * K_BASIS + 4 * f4);
            wphil[(f4 >> 2) * 64 + (f4 & 3) * 16 + jj] = v;
        }
        // W1[h][cs*16 .. +15]: 64 B per row; thread takes (row h, float4 c4), scatters its 4 columns
        float *w1f = reinterpret_cast<float *>(w1l);
#pragma unroll
        for (int i = 0; i < H / 64; ++i) {
            const int h = (tid >> 2) + 64 * i, c4 = tid & 3;
            const f32x4 x = *reinterpret_cast<g4>(P + a.off.iqn_w1 + (int64_t)h * E_DIM + cs * 16 + 4 * c4);
            const int q = h >> 4, gg = (h >> 2) & 3, jh = h & 3;
#pragma unroll
            for (int c = 0; c < 4; ++c) w1f[((q * 64 + gg * 16 + 4 * c4 + c) << 2) + jh] = x[c];
        }
    }
    const float bphi = PHI_SAVED ? 0.f : P[a.off.phi_b + n];
    const float g1 = LN ? P[a.off.iqn_ln1_g + n] : 1.f, be1 = LN ? P[a.off.iqn_ln1_b + n] : 0.f;

    f32x4 accWphi[4], accW1[NHT];     // accWphi[c][r']: dWphi[n = 4g + r'][k = 4j + c];  accW1[4u + c][r']: dW1[h = 16g + 4r' + 64u + c][n = j]
#pragma unroll
    for (int i = 0; i < 4; ++i) accWphi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < NHT; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_dg = 0.f, s_db = 0.f, s_dbphi = 0.f, de_acc = 0.f;

    // ---- operands of a tile ---------------------------------------------------------------------
    struct TileSet {
        f32x4 ac[PHI_SAVED ? 1 : 4];      // row on the lane:   cos[r0 + j][16q + 4g ..] (only to recompute phi)
        f32x4 ad[NHT];                    // row on the lane:   dpre1[r0 + j][16q + 4g ..]
        f32x4 cB[4], dB[4][NU];           // row on the k index: cos[r0 + 4g + r][4j ..],  dpre1[r0 + 4g + r][64u + 4j ..]
        f32x4 mu, rs, c1, c2;
        f32x4 ph;                         // ReLU(phi)[r0 + 4g + r][n], as saved by the forward tiles
        float ev;
    };
    auto load_rows_a = [&](TileSet &S, int ti) __attribute__((always_inline)) {
        const int r0 = (tile_begin + ti) * 16;
        if (!PHI_SAVED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) S.ac[PHI_SAVED ? 0 : q] = bload4(rs_cos, vo_ca + 64 * q, r0 * K_BASIS * 4);
        }
#pragma unroll
        for (int q = 0; q < NHT; ++q) S.ad[q] = bload4(rs_dp, vo_da + 64 * q, r0 * H * 4);
    };
    auto load_rows_b = [&](TileSet &S, int ti) __attribute__((always_inline)) {
        const int r0 = (tile_begin + ti) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            S.cB[r] = bload4(rs_cos, vo_cb + r * K_BASIS * 4, r0 * K_BASIS * 4);
#pragma unroll
            for (int u = 0; u < NU; ++u) S.dB[r][u] = bload4(rs_dp, vo_db + r * H * 4 + 256 * u, r0 * H * 4);
        }
    };
    auto load_scalars = [&](TileSet &S, int ti) __attribute__((always_inline)) {
        const int r0 = (tile_begin + ti) * 16;
        if (LN) {
            S.mu = bload4(rs_mu, vo_sc, r0 * 4);
            S.rs = bload4(rs_rs, vo_sc, r0 * 4);
            S.c1 = bload4(rs_c1, vo_sc, r0 * 4);
            S.c2 = bload4(rs_c2, vo_sc, r0 * 4);
        }
        S.ev = e_cur[(int64_t)((r0 + 4 * g) / T) * E_DIM + n];
        if (PHI_SAVED) {
            const int so = ((tile_begin + ti) * (E_DIM / 16) + cs) * 1024;   // this tile's block of the workgroup's columns
#pragma unroll
            for (int r = 0; r < 4; ++r)
                S.ph[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ph, vo_ph + 64 * r, so, 0));
        }
    };
    TileSet S0, S1;
    S0.mu = S0.c1 = S0.c2 = S1.mu = S1.c1 = S1.c2 = f32x4{0.f, 0.f, 0.f, 0.f};
    S0.rs = S1.rs = f32x4{1.f, 1.f, 1.f, 1.f};
    if (tiles_per_wave > 0) {
        load_rows_a(S0, 0);
        load_scalars(S0, 0);
        load_rows_b(S0, 0);
    }
    __syncthreads();          // weight slices (and every wave's observation rows) are in LDS
    f32x4 wphr[4], w1r[NHT];  // DB: this lane's B operands of the phi / dX products, for the whole kernel
    if (DB) {
        if (!PHI_SAVED) {
#pragma unroll
            for (int q = 0; q < 4; ++q) wphr[q] = wphil[q * 64 + lane];
        }
#pragma unroll
        for (int q = 0; q < NHT; ++q) w1r[q] = w1l[q * 64 + lane];
    }
    PRISM_STAMP(9);

    auto tile = [&](TileSet &S, TileSet &Sn, int ti) __attribute__((always_inline)) {
        const int r0 = (tile_begin + ti) * 16;
        const bool more = ti + 1 < tiles_per_wave;
        const int bsm = (r0 + 4 * g) / T;
        if (DB && more) {          // the whole next tile, a full tile ahead of its first use
            load_rows_a(Sn, ti + 1);
            load_scalars(Sn, ti + 1);
            load_rows_b(Sn, ti + 1);
        }
        // ---- dX columns (and, width 256, the phi columns): independent MFMA chains interleaved -------------------
        f32x4 aphi = {0.f, 0.f, 0.f, 0.f}, adx = {0.f, 0.f, 0.f, 0.f}, adx2 = {0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_sched_barrier(0);
        if (ti < 4) PRISM_LOOP_STAMP(16 + 4 * ti);
#pragma unroll
        for (int q = 0; q < NHT / 2; ++q) {
            f32x4 wa, wb, wp = {0.f, 0.f, 0.f, 0.f};
            if (DB) {
                wa = w1r[2 * q];
                wb = w1r[2 * q + 1];
                if (!PHI_SAVED && q < 4) wp = wphr[q & 3];
            } else {
                wa = w1l[(2 * q) * 64 + lane];
                wb = w1l[(2 * q + 1) * 64 + lane];
                if (!PHI_SAVED && q < 4) wp = wphil[(q & 3) * 64 + lane];
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (!PHI_SAVED && q < 4) aphi = mfma16(S.ac[PHI_SAVED ? 0 : q & 3][c], wp[c], aphi);
                adx = mfma16(S.ad[2 * q][c], wa[c], adx);
                adx2 = mfma16(S.ad[2 * q + 1][c], wb[c], adx2);
            }
            if (q & 1) __builtin_amdgcn_sched_barrier(0);      // (bounds how many weight registers are in flight)
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ti < 4) PRISM_LOOP_STAMP(17 + 4 * ti);
        if (!DB && more) load_rows_a(S, ti + 1);  // the row-on-lane registers are free: refill them for the next tile
        __builtin_amdgcn_sched_barrier(0);
        // ---- elementwise backward of row group r (column n), then the weight-gradient MFMAs of that group:
        // the VALU work of group r + 1 issues in the shadow of the MFMAs of group r
        float dep = 0.f;
        float xv = 0.f, dpp = 0.f;
        const float ev = S.ev;
        auto elementwise = [&](int r) __attribute__((always_inline)) {
            const float phi = PHI_SAVED ? S.ph[r] : fmaxf(aphi[r] + bphi, 0.f);   // saved by the forward / recomputed (bias added last, as there)
            const float h0 = phi * ev;
            const float xhat = LN ? (h0 - S.mu[r]) * S.rs[r] : h0;
            xv = LN ? xhat * g1 + be1 : h0;                 // trunk input (B operand of dW1)
            const float dX = adx[r] + adx2[r];
            s_dg += dX * xhat;
            s_db += dX;
            const float dh0 = LN ? S.rs[r] * (dX * g1 - S.c1[r] * (1.0f / E_DIM) - xhat * (S.c2[r] * (1.0f / E_DIM))) : dX;
            dep += dh0 * phi;
            dpp = (phi > 0.f) ? dh0 * ev : 0.f;
            s_dbphi += dpp;
        };
        elementwise(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xr = xv, dr = dpp;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) accWphi[c] = mfma16(dr, S.cB[r][c], accWphi[c]);
            if (r < 3) elementwise(r + 1);
#pragma unroll
            for (int u = 0; u < NU; ++u)
#pragma unroll
                for (int c = 0; c < 4; ++c) accW1[4 * u + c] = mfma16(S.dB[r][u][c], xr, accW1[4 * u + c]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ti < 4) PRISM_LOOP_STAMP(18 + 4 * ti);
        const bool ev_pos = ev > 0.f;
        if (!DB && more) {
            load_rows_b(S, ti + 1);
            load_scalars(S, ti + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        // d e[b][n]: sum over the T rows of a sample
        float dcv = 0.f;          // ReLU-masked d e of (sample, column n) when it is final in this lane
        if (T == 4) {
            if (!n_mine) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;       // (only the post-kernel conv role reads it)
        } else if (T == 8) {
            dep += __shfl_xor(dep, 16, 64);
            if (!n_mine && (g & 1) == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
            dcv = ev_pos ? dep : 0.f;
        } else {
            dep += __shfl_xor(dep, 16, 64);
            dep += __shfl_xor(dep, 32, 64);
            de_acc += dep;
            if (((r0 + 16) % T) == 0) {
                if (!n_mine && g == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = de_acc;
                dcv = ev_pos ? de_acc : 0.f;
                de_acc = 0.f;
            }
        }
        if (n_mine) s_dcv[ti * 64 + lane] = dcv;
        if (ti < 4) PRISM_LOOP_STAMP(19 + 4 * ti);
    };
    if (DB) {
        for (int ti = 0; ti < tiles_per_wave; ti += 2) {
            tile(S0, S1, ti);
            if (ti + 1 < tiles_per_wave) tile(S1, S0, ti + 1);
        }
    } else {
        for (int ti = 0; ti < tiles_per_wave; ++ti) tile(S0, S0, ti);
    }
    // ---- conv taps of (sample, channel cs>>2, output position (y0 + (j>>3), j&7)) for every tile of this wave;
    // LDS reads of this wave's own earlier writes need no barrier
    float cacc[BWD_CONV_TAPS], cbias = 0.f;
#pragma unroll
    for (int i = 0; i < BWD_CONV_TAPS; ++i) cacc[i] = 0.f;
    if (n_mine) {
        for (int ti = 0; ti < tiles_per_wave; ++ti) {
            const int bsm = ((tile_begin + ti) * 16 + 4 * g) / T;
            const float dcv = s_dcv[ti * 64 + lane];
            if (sub == 0) cbias += dcv;
            const float *src = s_obs + (bsm - ws_lo) * 40 * C + ((j >> 3) * 10 + (j & 7)) * C + sub * cpl;
            // all reads of a lane issue back to back (no per-tap branches), then the FMAs
            if (cpl == 2) {
                float2 ob[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) ob[i] = *reinterpret_cast<const float2 *>(src + ((i / 3) * 10 + (i % 3)) * C);
#pragma unroll
                for (int i = 0; i < 9; ++i) {
                    cacc[i] = fmaf(dcv, ob[i].x, cacc[i]);
                    cacc[9 + i] = fmaf(dcv, ob[i].y, cacc[9 + i]);
                }
            } else {
                float ob[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) ob[i] = src[((i / 3) * 10 + (i % 3)) * C];
#pragma unroll
                for (int i = 0; i < 9; ++i) cacc[i] = fmaf(dcv, ob[i], cacc[i]);
            }
        }
    }

    PRISM_STAMP(10);
    // ---- reduce the four waves in fixed order and write this workgroup's slab part -------------
    __syncthreads();
    PRISM_STAMP(11);
    float *red = smem + bwd_red_region(H).off;    // [4 waves][BWD_ACC][64]
    {
        float *mine = red + (w * BWD_ACC) * 64 + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(c * 4 + r) * 64] = accWphi[c][r];
#pragma unroll
        for (int mt = 0; mt < NHT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 + mt * 4 + r) * 64] = accW1[mt][r];
        // column sums: fold the 4 row groups (lanes j, j+16, j+32, j+48)
        s_dg += __shfl_xor(s_dg, 16, 64);
        s_dg += __shfl_xor(s_dg, 32, 64);
        s_db += __shfl_xor(s_db, 16, 64);
        s_db += __shfl_xor(s_db, 32, 64);
        s_dbphi += __shfl_xor(s_dbphi, 16, 64);
        s_dbphi += __shfl_xor(s_dbphi, 32, 64);
        mine[(16 + 4 * NHT) * 64] = s_dg;
        mine[(17 + 4 * NHT) * 64] = s_db;
        mine[(18 + 4 * NHT) * 64] = s_dbphi;
    }
    __syncthreads();
    float *slab = a.ws.slabs + (int64_t)rc * a.slab;
    // this thread (wave w, lane) writes the register index r' = w of every accumulator: affine addresses
    auto fold4 = [&](int slot) {
        return ((red[(0 * BWD_ACC + slot) * 64 + lane] + red[(1 * BWD_ACC + slot) * 64 + lane]) +
                red[(2 * BWD_ACC + slot) * 64 + lane]) + red[(3 * BWD_ACC + slot) * 64 + lane];
    };
    {
        // accWphi[c][r' = w]: row n = cs*16 + 4g + w, basis k = 4j + c  -> one 16-byte store
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = fold4(c * 4 + w);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(slab + (int64_t)(cs * 16 + 4 * g + w) * K_BASIS + 4 * j));
    }
    {
        // accW1[4u + c][r' = w]: row h = 16g + 4w + 64u + c, col n = cs*16 + j
        float *dst = slab + (int64_t)SLAB_W1 + (int64_t)(16 * g + 4 * w) * E_DIM + cs * 16 + j;
        float v[NHT];
#pragma unroll
        for (int mt = 0; mt < NHT; ++mt) v[mt] = fold4(16 + w + 4 * mt);
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int c = 0; c < 4; ++c) __builtin_nontemporal_store(v[4 * u + c], dst + (int64_t)(64 * u + c) * E_DIM);
    }
    if (w < 3 && g == 0) {
        const float v = fold4(16 + 4 * NHT + w);
        const int nn = cs * 16 + j;
        if (w == 2) slab[E_DIM * K_BASIS + nn] = v;                   // d phi_b
        else if (LN) slab[E_DIM * K_BASIS + (1 + w) * E_DIM + nn] = v;   // d ln1_g (w = 0), d ln1_b (w = 1)
    }
    PRISM_STAMP(12);
    if (!n_mine) return;
    // ---- conv-backward partial row of this block: fold lanes (16 positions, then the lane groups that
    // hold the same taps for other samples), then the four waves in fixed order
    float *s_tap = smem + bwd_tap_region(a.bg.main_lds).off;   // [4 waves][4 lane groups][TAPS + 1]  (<= 4 * BWD_CONV_ROW)
    constexpr int TS = BWD_CONV_TAPS + 1;
#pragma unroll
    for (int i = 0; i < TS; ++i) {
        float v = i < BWD_CONV_TAPS ? cacc[i < BWD_CONV_TAPS ? i : 0] : cbias;
        v += dpp_move<0xB1, 0xF>(0.f, v);                // 16 positions of the row: quad swaps, then row rotates
        v += dpp_move<0x4E, 0xF>(0.f, v);
        v += dpp_move<0x124, 0xF>(0.f, v);
        v += dpp_move<0x128, 0xF>(0.f, v);
        if (j == 0) s_tap[(w * 4 + g) * TS + i] = v;
    }
    __syncthreads();
    if (tid <= 9 * C) {
        // output tap o belongs to lane subset so = o / n_mine; for T = 8 lane groups so and so + 2 hold it
        const int so = tid < 9 * C ? tid / n_mine : 0, i = tid < 9 * C ? tid - so * n_mine : BWD_CONV_TAPS;
        float t = 0.f;
        for (int ww = 0; ww < 4; ++ww) {
            t += s_tap[(ww * 4 + so) * TS + i];
            if (T == 8) t += s_tap[(ww * 4 + so + 2) * TS + i];
        }
        a.ws.convpart[(int64_t)(rc * (E_DIM / 16) + cs) * BWD_CONV_ROW + tid] = t;
    }
    PRISM_STAMP(26);
}

// ------------------------------------------------------------------------------------------
// Block routines of the post kernel (step_kernels.h): conv-backward partials and the small tensors.
// ------------------------------------------------------------------------------------------
constexpr int CONV_ROW = 16 * 90 + 16;   // floats per partial row: 16 * 9C weights (C <= 10) + 16 biases
constexpr int SMALL_MAX_B = 4096;

// A gradient element another workgroup of the SAME launch reads behind the fused tail's grid barrier: stored at agent
// scope (written through, past this XCD's L2), so that the producer's barrier arrival needs no L2 write-back -- it only
// waits for these stores.  The reader uses agent-scope loads (tail_clip_adam).
__device__ __forceinline__ void far_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float far_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// d conv_w / d conv_b partial sums over the samples of block cb, all 16 output channels, on the matrix core.
// (minatar_cnn_model.py:43-46 backward: dW[c][k] = sum_b sum_pos dconv[b][c][pos] * patch[b][pos][k], k = (ci, dy, dx);
// db[c] = sum_b sum_pos dconv[b][c][pos].)  That is a [16 x 64 B] x [64 B x 9C] product: per sample sixteen K steps (four
// positions each) of v_mfma_f32_16x16x4_f32 per 16-column tile of k, accumulated across the samples in the matrix core's
// registers; the bias gradient is one more column of the patch matrix (k = 9C, all ones: 9C is never a multiple of 16 for
// C <= 10).  A wave owns one k tile and every G-th sample of the block (G = 16 / tiles wave groups); the G partial tiles are
// added through LDS in group order.  Staging is ONE round of loads: every operand of the block's samples is requested before
// the first is used (observations as float4 into the channel-major image the forward convolution uses, the ReLU-masked
// embedding gradient as float4 of d e (IQN) + the Q heads' slots in fixed order).
// (Round 3's form -- one (sample, c, ci, dy) item a thread, fmaf chains out of LDS, four samples a block -- took 34 k cycles
// a block on c4, 11.6 k of them four dependent staging trips: the longest role of its post launch.)
__host__ __device__ constexpr int conv_spb(int C) { return C <= 5 ? 8 : 4; }          // samples per conv-backward block
constexpr int CV_DCS = 68;                                                              // row stride of a sample's [16][64] gradient image
__host__ __device__ constexpr int conv_sample_floats(int C) { return 100 * C + 16 * CV_DCS; }
constexpr int CONV_LDS_FLOATS = 8 * conv_sample_floats(5) > 4 * conv_sample_floats(10) ? 8 * conv_sample_floats(5) : 4 * conv_sample_floats(10);
static_assert(conv_spb(5) * conv_sample_floats(5) <= CONV_LDS_FLOATS && conv_spb(10) * conv_sample_floats(10) <= CONV_LDS_FLOATS &&
              16 * 256 <= CONV_LDS_FLOATS, "conv backward block: samples + reduction scratch fit the pool");
__device__ __forceinline__ void conv_bwd_partial_block(const IqnArgs &a, int cb, float *lds) {
    const int tid = threadIdx.x, B = a.B, C = a.C;
    const int SPB = conv_spb(C), SF = conv_sample_floats(C);
    const int b0 = cb * SPB, ns = min(SPB, B - b0);
    // ---- staging: one round of loads
    {
        // observations: float4 t of sample s (25 C per sample; SPB * 25 C <= 1024)
        const int per = 25 * C, so = tid / per, to = tid - so * per;
        const bool ob_ok = so < ns;
        float4 ov = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ob_ok) ov = reinterpret_cast<const float4 *>(a.obs + (int64_t)(b0 + so) * 100 * C)[to];
        // embedding gradient: float4 n4 of sample s, two per thread at eight samples a block
        const bool iq = a.use_iqn && a.propagate_grad;
        float4 di[2], ev[2], q0[2], q1[2];
        bool ok[2];
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 1024 * it, sm = idx >> 8, n4 = idx & 255;
            ok[it] = sm < ns;
            const int64_t o = ((int64_t)(b0 + (ok[it] ? sm : 0)) * E_DIM) / 4 + n4;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            di[it] = iq ? reinterpret_cast<const float4 *>(a.ws.de_iqn)[o] : z;
            ev[it] = reinterpret_cast<const float4 *>(a.ws.e_cur)[o];
            q0[it] = a.q_de_slots > 0 ? reinterpret_cast<const float4 *>(a.ws.de_q)[o] : z;
            q1[it] = a.q_de_slots > 1 ? reinterpret_cast<const float4 *>(a.ws.de_q + (size_t)B * E_DIM)[o] : z;
        }
        if (ob_ok) obs_to_lds(lds + so * SF, ov, to, C);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + 1024 * it, sm = idx >> 8, n4 = idx & 255;
            if (!ok[it]) continue;
            float d[4] = {di[it].x, di[it].y, di[it].z, di[it].w};
            if (a.q_de_slots > 0) { d[0] += q0[it].x; d[1] += q0[it].y; d[2] += q0[it].z; d[3] += q0[it].w; }
            if (a.q_de_slots > 1) { d[0] += q1[it].x; d[1] += q1[it].y; d[2] += q1[it].z; d[3] += q1[it].w; }
            // (more slots -- one per head where the two-GEMM backward does not apply: two per trip, slots in fixed order)
            for (int hd = 2; hd < a.q_de_slots; hd += 2) {
                const int64_t o = ((int64_t)(b0 + sm) * E_DIM) / 4 + n4;
                const float4 u = reinterpret_cast<const float4 *>(a.ws.de_q + (size_t)hd * B * E_DIM)[o];
                const float4 v = hd + 1 < a.q_de_slots ? reinterpret_cast<const float4 *>(a.ws.de_q + (size_t)(hd + 1) * B * E_DIM)[o]
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                d[0] += u.x; d[1] += u.y; d[2] += u.z; d[3] += u.w;
                if (hd + 1 < a.q_de_slots) { d[0] += v.x; d[1] += v.y; d[2] += v.z; d[3] += v.w; }
            }
            const float e4[4] = {ev[it].x, ev[it].y, ev[it].z, ev[it].w};
            const int n = 4 * n4;
            float4 m;
            m.x = e4[0] > 0.f ? d[0] : 0.f;
            m.y = e4[1] > 0.f ? d[1] : 0.f;
            m.z = e4[2] > 0.f ? d[2] : 0.f;
            m.w = e4[3] > 0.f ? d[3] : 0.f;
            *reinterpret_cast<float4 *>(lds + sm * SF + 100 * C + (n >> 6) * CV_DCS + (n & 63)) = m;
        }
    }
    __syncthreads();
    PRISM_STAMP(22);
    // ---- the product: wave = (sample group grp, k tile nt)
    const int K = 9 * C, NT = (K + 1 + 15) >> 4, G = 16 / NT;
    const int w = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
    const int nt = w % NT, grp = w / NT;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (grp < G) {
        const int k = 16 * nt + li, kc = k < K ? k : K - 1;
        const int ci = (kc * 7282) >> 16, t9 = kc - 9 * ci, dy = (t9 * 11) >> 5;      // kc / 9, t9 / 3 (kc < 1000)
        const int koff = ci * 100 + dy * 10 + (t9 - 3 * dy) + g;
        const float kone = k == K ? 1.f : 0.f;
        for (int s = grp; s < ns; s += G) {
            const float *ob = lds + s * SF + koff, *dc = lds + s * SF + 100 * C + li * CV_DCS + g;
            float av[16], bv[16];
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) {
                av[ks] = dc[4 * ks];                                        // A[c = li][pos = 4 ks + g]
                const float p = ob[(ks >> 1) * 10 + 4 * (ks & 1)];          // B[pos][k]: obs[ci][y + dy][x + dx], y = pos >> 3, x = pos & 7
                bv[ks] = k < K ? p : kone;
            }
#pragma unroll
            for (int ks = 0; ks < 16; ++ks) acc = mfma16(av[ks], bv[ks], acc);
        }
    }
    __syncthreads();                                                        // (the images are dead: the scratch aliases them)
    if (grp < G) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lds[(grp * NT + nt) * 256 + r * 64 + lane] = acc[r];
    }
    __syncthreads();
    PRISM_STAMP(23);
    float *out = a.ws.convpart + (int64_t)cb * CONV_ROW;
    for (int i = tid; i < NT * 256; i += 1024) {                            // (six k tiles at ten channels: two trips)
        const int tn = i >> 8, e = i & 255, r = e >> 6, ln = e & 63;
        const int c = 4 * (ln >> 4) + r, k = 16 * tn + (ln & 15);
        float t = 0.f;
        for (int q = 0; q < G; ++q) t += lds[(q * NT + tn) * 256 + e];
        // (written through: the workgroup that folds the rows -- the last to arrive -- reads them at agent scope, step_kernels.h)
        if (k < K) far_store(&out[c * K + k], t);
        else if (k == K) far_store(&out[16 * K + c], t);
    }
}


// b1, LN2 affine, W2, b2 gradients of one [LN -> Linear(H -> A)] head for the 16 hidden units
// [slice*16, slice*16+16), shared by the IQN head and the Q-ensemble heads:
//   S[a][h] = sum_{b: act=a} v(b,h);  D[a] = sum_{b: act=a} d(b);  db1[h] = sum_b p(b,h)
//   dW2[a][h] = g2[h] S[a][h] + beta2[h] D[a];  dg2[h] = sum_a W2[a][h] S[a][h];  dbeta2[h] = sum_a W2[a][h] D[a]
// (+ kappa * theta on every tensor when the Theil regulariser is on).
// 1024 threads = 16 units x 64 batch parts.  Latency-shaped: every global operand is requested up
// front, the 64 batch parts of all A+1 per-unit sums (and of D, riding in a 17th column) are folded
// in ONE LDS pass (16 reads + a quad reduction per thread), and the tail runs as (action, unit)
// threads.  `pool`: SMALL_POOL_FLOATS floats of LDS.
constexpr int SMALL_W = 16;
constexpr int SMALL_GROUP = 8;                                   // per-unit sums folded per LDS pass
constexpr int SMALL_POOL_FLOATS = SMALL_GROUP * 64 * 17;
struct SmallIo {
    const float *w2, *g2, *be2, *b1, *b2;   // parameters of the head (b1, b2 only read with kappa); g2 == NULL:
                                            // no affine in front of the Linear (g = 1, beta = 0, no dg / dbeta)
    float *gw2, *gg2, *gbe2, *gb1, *gb2;    // their gradients (gb1 == NULL: no b1 row)
    float kappa;
    bool use_kappa;
    const float *extra;                     // optional [B]: slice 0 leaves sum_b extra[b] in *extra_sum (thread 0)
    int w_stride;                           // row stride of w2 / gw2 (units per action row)
};
// load(b, h, want_d) -> {v, p, d}.  AMAX: action slots carried per thread (7: up to seven actions, one LDS pass).
template <int AMAX, typename Load>
__device__ __forceinline__ void small_fold_core(const IqnArgs &a, int slice, float &sq, float *pool, const SmallIo &io,
                                                Load load, float *extra_sum) {
    __shared__ float s_S[17][SMALL_W];      // [A] = b1 row
    __shared__ float s_D[16];
    __shared__ float s_lw[16];
    __shared__ float s_dgb[2][16][SMALL_W];
    const int tid = threadIdx.x, B = a.B, A = a.A;
    const int hl = tid & (SMALL_W - 1), part = tid >> 4, h = slice * SMALL_W + hl;
    // tail operands of thread (action ta, unit hl): requested now, used last
    const int ta = tid >> 4;
    float w2 = 0.f, g2 = 0.f, be2 = 0.f, b1v = 0.f, b2v = 0.f;
    if (ta < A) {
        w2 = io.w2[(int64_t)ta * io.w_stride + h];
        g2 = io.g2 ? io.g2[h] : 1.f;
        be2 = io.g2 ? io.be2[h] : 0.f;
    }
    if (io.use_kappa) {
        if (io.gb1 && tid >= 2 * SMALL_W && tid < 3 * SMALL_W) b1v = io.b1[h];
        if (slice == 0 && tid < A) b2v = io.b2[tid];
    }
    float sA[AMAX], dA[AMAX];
#pragma unroll
    for (int aa = 0; aa < AMAX; ++aa) sA[aa] = dA[aa] = 0.f;
    float pb = 0.f;
    float lw = (io.extra && slice == 0 && tid < B) ? io.extra[tid] : 0.f;
    // (four trips at batch 256: unrolled by exactly that -- under "unroll 8" those four ran in the ROLLED remainder loop,
    // one memory round trip each, and this role reached the fused tail's grid barrier last.  Batch 512 makes two rounds of
    // four; an eight-slot predicated form beside this one cost the kernel 5 k instructions and every launch shape ~1 us)
#pragma unroll 4
    for (int b = part; b < B; b += 64) {
        const float3 x = load(b, h, hl == 0);                   // d rides in column 16 of the fold
        pb += x.y;
        const int ab = (int)a.action[b];
#pragma unroll
        for (int aa = 0; aa < AMAX; ++aa) {
            sA[aa] += (ab == aa) ? x.x : 0.f;
            dA[aa] += (ab == aa) ? x.z : 0.f;
        }
    }
    PRISM_STAMP(16);
    if (io.extra && slice == 0) {
        for (int b = 1024 + tid; b < B; b += 1024) lw += io.extra[b];
        lw = wave_sum(lw);
        if ((tid & 63) == 0) s_lw[tid >> 6] = lw;
    }
    PRISM_STAMP(17);
    // fold the 64 batch parts of sums x = 0..A (x = A is the b1 row), SMALL_GROUP sums per pass;
    // waves 8..15 fold column 16 (the D[a] partials of each part)
    for (int x0 = 0; x0 <= A; x0 += SMALL_GROUP) {
        if (x0) __syncthreads();
#pragma unroll
        for (int aa = 0; aa < AMAX + 1; ++aa) {
            const int x = aa - x0;                               // (aa is compile-time, x0 uniform)
            if (aa <= A && x >= 0 && x < SMALL_GROUP) {
                pool[(x * 64 + part) * 17 + hl] = aa == A ? pb : sA[aa < AMAX ? aa : 0];
                if (hl == 0 && aa < A) pool[(x * 64 + part) * 17 + 16] = dA[aa < AMAX ? aa : 0];
            }
        }
        __syncthreads();
        if (tid < 512) {
            const int x = tid >> 6, rhl = (tid >> 2) & 15, q = tid & 3;
            if (x0 + x <= A) {
                float t = 0.f;
#pragma unroll
                for (int i = 0; i < 16; ++i) t += pool[(x * 64 + q * 16 + i) * 17 + rhl];
                t += __shfl_xor(t, 1, 64);
                t += __shfl_xor(t, 2, 64);
                if (q == 0) s_S[x0 + x][rhl] = t;
            }
        } else {
            const int x = (tid - 512) >> 6, lane = tid & 63;
            if (x0 + x < A) {
                const float t = wave_sum(pool[(x * 64 + lane) * 17 + 16]);
                if (lane == 0) s_D[x0 + x] = t;
            }
        }
    }
    __syncthreads();
    PRISM_STAMP(18);
    const float k = io.kappa;
    float dg = 0.f, db = 0.f;
    if (ta < A) {
        const float S = s_S[ta][hl], D = s_D[ta];
        float dw = g2 * S + be2 * D;
        if (io.use_kappa) dw += k * w2;
        far_store(&io.gw2[(int64_t)ta * io.w_stride + h], dw);
        sq += dw * dw;
        dg = w2 * S;
        db = w2 * D;
    }
    if (ta < 16) {
        s_dgb[0][ta][hl] = dg;
        s_dgb[1][ta][hl] = db;
    }
    __syncthreads();
    if (tid < 2 * SMALL_W) {                 // threads [0,16): d ln2_g; [16,32): d ln2_b
        if (io.g2) {
            const int which = tid >> 4;
            float t = 0.f;
            for (int aa = 0; aa < A; ++aa) t += s_dgb[which][aa][hl];
            if (io.use_kappa) t += k * (which ? io.be2[h] : io.g2[h]);
            far_store(&(which ? io.gbe2 : io.gg2)[h], t);
            sq += t * t;
        }
    } else if (tid < 3 * SMALL_W && io.gb1) {
        float t = s_S[A][hl];
        if (io.use_kappa) t += k * b1v;
        far_store(&io.gb1[h], t);
        sq += t * t;
    }
    PRISM_STAMP(19);
    if (slice == 0 && tid < A) {
        float D = s_D[tid];
        if (io.use_kappa) D += k * b2v;
        far_store(&io.gb2[tid], D);
        sq += D * D;
    }
    if (io.extra && slice == 0 && tid == 0) {
        float l = 0.f;
#pragma unroll
        for (int p = 0; p < 16; ++p) l += s_lw[p];
        *extra_sum = l;
    }
}

template <typename Load>
__device__ __forceinline__ void small_fold_block(const IqnArgs &a, int slice, float &sq, float *pool, const SmallIo &io,
                                                 Load load, float *extra_sum) {
    if (a.A <= 7) small_fold_core<7>(a, slice, sq, pool, io, load, extra_sum);       // (uniform)
    else small_fold_core<16>(a, slice, sq, pool, io, load, extra_sum);
}

// IQN head: Sb / Pb / Db were left per sample by the loss; slice 0 also writes the IQN part of the total loss
__device__ __forceinline__ void small_tensor_block(const IqnArgs &a, int slice, float &sq, float *pool) {
    const float *P = a.params;
    float *gr = a.grads;
    const bool ln = a.ln != 0;
    SmallIo io{P + a.off.iqn_w2, ln ? P + a.off.iqn_ln2_g : nullptr, ln ? P + a.off.iqn_ln2_b : nullptr, P + a.off.iqn_b1, P + a.off.iqn_b2,
               gr + a.off.iqn_w2, ln ? gr + a.off.iqn_ln2_g : nullptr, ln ? gr + a.off.iqn_ln2_b : nullptr, gr + a.off.iqn_b1, gr + a.off.iqn_b2,
               0.f, false, a.ws.lossw, a.Hi};
    float lsum = 0.f;
    small_fold_block(a, slice, sq, pool, io,
                     [&](int b, int h, bool want_d) {
                         return make_float3(a.ws.Sb[(int64_t)b * a.Hi + h], a.ws.Pb[(int64_t)b * a.Hi + h],
                                            want_d ? a.ws.Db[b] : 0.f);
                     },
                     &lsum);
    if (slice == 0 && threadIdx.x == 0) {
        const float l = lsum / (float)a.B;       // mean_b(dl_b * w_b)  (agent.py:58-64)
        a.out_scalars[1] = l;
        if (a.n_heads == 0) {
            a.out_scalars[0] = l;
            a.out_scalars[2] = 0.f;
        }
    }
}

}  // namespace prism
