// IQN TD-update kernels for gfx950 (fp32 MFMA 16x16x4, wave64).
//
// Restates /root/reference/prism/agents/models/iqn_model.py:48-201 (+ ffnn_model.py:61-76,
// minatar_cnn_model.py:43-46) as six launches:
//
//   embed      conv3x3+ReLU of obs / next_obs -> e_cur, e_next [B,1024]; extra workgroups compute
//              u = W1 g1, v = W1 beta1 (used to get LayerNorm-backward row sums without dX)
//   tile_fwd   one 16-row tile of (sample, tau) rows per workgroup, whole rows on chip:
//              cos basis -> phi GEMM (K=64) -> ReLU -> Hadamard with e -> LayerNorm(1024) ->
//              trunk GEMM (K=1024) -> ReLU -> LayerNorm(128) -> head -> Z[16, A]
//   loss       one wave per sample: argmax / n-step target / pairwise quantile-Huber tile,
//              dL/dq, head + LayerNorm(128) backward -> dpre1 and the per-row scalars
//   bwd        column-sliced backward: workgroup (16 embed columns x a row chunk) recomputes its
//              columns of phi / LN from saved row statistics and accumulates dWphi, dW1, dLN, de
//              with NO cross-workgroup reduction (per-chunk slabs)
//   small      conv-backward partials + the small tensors (b1, LN2, W2, b2)
//   reduce     slabs/partials -> flat gradient + sum-of-squares partials
//
// Rows are SAMPLE-major inside the workspace (row = b*T + t); the reference's tau-major order
// (row = t*B + b, iqn_model.py:70) only matters for how tau inputs are indexed.
#pragma once
#include "common.h"

namespace prism {

constexpr int E_DIM = 1024;   // 16 * 8 * 8 (minatar_cnn_model.py:14)
constexpr int K_BASIS = 64;   // iqn_n_basis_elements
constexpr int H_DIM = 128;    // iqn_quantile_model_feature_dim
constexpr int YS = E_DIM + 4; // LDS row strides (floats), +4 breaks the 16-row bank alias
constexpr int CS = K_BASIS + 4;
constexpr int HS = H_DIM + 4;
constexpr float LN_EPS = 1e-5f;
constexpr float PI_F = 3.14159274101257324f;  // fp32(np.pi), the scalar torch multiplies by

struct IqnPass {
    const float *params;  // weight set for this pass (online or target flat buffer)
    const float *e;       // [B][E] embedded observations feeding this pass
    const float *tau_in;  // [T*B] tau-major, or NULL -> Philox
    float *z_out;         // [B*T][A] sample-major
    int T;
    int n_tiles;          // B*T/16
    int save;             // current-state pass: keep what backward needs
    int stream_id;        // 0 cur, 1 next-online, 2 next-target
};

struct IqnWs {           // workspace pointers (device)
    float *e_cur, *e_next, *uv;
    float *cosb, *mu1, *rstd1, *pre1, *xhat2, *rstd2;
    float *zcur, *zon, *ztg;
    float *dq, *c1, *c2, *dpre1, *Sb, *Pb, *Db, *lossw;
    float *de_iqn;
    float *slabs;        // [n_chunks][SLAB]
    float *convpart;     // [16][CONV_CHUNKS][9C+1]
    float *normpart;     // [NORM_SLOTS]
    unsigned int *ticket;
};

constexpr int SLAB = E_DIM * K_BASIS + E_DIM + E_DIM + E_DIM + H_DIM * E_DIM;  // phi_w|phi_b|ln1_g|ln1_b|w1
constexpr int CONV_CHUNKS = 16;
constexpr int NORM_SLOTS = 1024;

struct IqnArgs {
    IqnPass pass[3];
    int n_pass;
    int B, A, C, T, Tn;
    int n_chunks;          // row chunks of the backward
    int has_target, double_q, propagate_grad;
    float huber_k, dist_w;
    prism_param_offsets off;
    const float *params;
    const float *target_params;
    const float *obs, *next_obs, *reward, *gamma, *per_weights;
    const uint8_t *nonterminal;
    const int64_t *action;
    uint64_t seed, offset;
    float *tau_out;        // [3][maxT*B] or NULL
    int maxT;
    float *out_dl, *out_td, *out_scalars;
    float *grads;
    IqnWs ws;
};

// ------------------------------------------------------------------------------------------
// embed: blocks [0,B) conv(obs) online; [B,2B) conv(next_obs) with target-or-online weights;
//        blocks [2B, 2B + H/4) compute u,v.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iqn_embed_kernel(IqnArgs a) {
    __shared__ float s_obs[1024];
    __shared__ float s_w[16 * 16 * 9];
    __shared__ float s_b[16];
    const int B = a.B, C = a.C;
    const int blk = blockIdx.x, tid = threadIdx.x;
    if (blk >= 2 * B) {
        // u[h] = sum_n W1[h][n] g1[n],  v[h] = sum_n W1[h][n] beta1[n]   (one wave per h)
        const int h = (blk - 2 * B) * 4 + (tid >> 6), lane = tid & 63;
        const float *W1 = a.params + a.off.iqn_w1 + (int64_t)h * E_DIM;
        const float *g1 = a.params + a.off.iqn_ln1_g, *b1 = a.params + a.off.iqn_ln1_b;
        float su = 0.f, sv = 0.f;
        for (int n = lane * 4; n < E_DIM; n += 256) {
            const float4 w = *reinterpret_cast<const float4 *>(W1 + n);
            const float4 g = *reinterpret_cast<const float4 *>(g1 + n);
            const float4 bb = *reinterpret_cast<const float4 *>(b1 + n);
            su += w.x * g.x + w.y * g.y + w.z * g.z + w.w * g.w;
            sv += w.x * bb.x + w.y * bb.y + w.z * bb.z + w.w * bb.w;
        }
        su = wave_sum(su);
        sv = wave_sum(sv);
        if (lane == 0) {
            a.ws.uv[h] = su;
            a.ws.uv[H_DIM + h] = sv;
        }
        return;
    }
    const bool is_next = blk >= B;
    const int b = is_next ? blk - B : blk;
    const float *P = (is_next && a.has_target) ? a.target_params : a.params;
    const float *src = (is_next ? a.next_obs : a.obs) + (int64_t)b * 100 * C;
    float *dst = (is_next ? a.ws.e_next : a.ws.e_cur) + (int64_t)b * E_DIM;
    for (int i = tid; i < 100 * C; i += 256) s_obs[i] = src[i];
    for (int i = tid; i < 16 * C * 9; i += 256) s_w[i] = P[a.off.conv_w + i];
    if (tid < 16) s_b[tid] = P[a.off.conv_b + tid];
    __syncthreads();
    for (int n = tid; n < E_DIM; n += 256) {
        const int c = n >> 6, y = (n >> 3) & 7, x = n & 7;
        float acc = s_b[c];
        const float *w = s_w + c * C * 9;
        for (int ci = 0; ci < C; ++ci)
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
                    acc = fmaf(w[ci * 9 + dy * 3 + dx], s_obs[((y + dy) * 10 + (x + dx)) * C + ci], acc);
        dst[n] = fmaxf(acc, 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// tile_fwd: 512 threads = 8 waves, one 16-row tile per workgroup.
// ------------------------------------------------------------------------------------------
constexpr int TILE_FWD_LDS_FLOATS = 16 * YS + 16 * CS + 16 * HS + 64;

__global__ __launch_bounds__(512) void iqn_tile_fwd_kernel(IqnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ytile = smem;
    float *cost = ytile + 16 * YS;
    float *h1 = cost + 16 * CS;
    float *rowf = h1 + 16 * HS;                 // [0,16) tau, [16,32) mu, [32,48) rstd
    int *rowb = reinterpret_cast<int *>(rowf + 48);  // [16] sample of each row

    int tile = blockIdx.x, pi = 0;
    while (pi < a.n_pass - 1 && tile >= a.pass[pi].n_tiles) {
        tile -= a.pass[pi].n_tiles;
        ++pi;
    }
    const IqnPass ps = a.pass[pi];
    const int B = a.B, A = a.A, T = ps.T;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int li = lane & 15, g = lane >> 4;
    const int r0 = tile * 16;
    const float *P = ps.params;

    if (tid < 16) {
        const int r = r0 + tid, b = r / T, t = r - b * T;
        float tau;
        if (ps.tau_in) {
            tau = ps.tau_in[(int64_t)t * B + b];
        } else {
            uint32_t rr[4];
            Philox ph(a.seed);
            ph(a.offset + (uint64_t)((int64_t)t * B + b), 0x54415530ull + (uint64_t)ps.stream_id, rr);
            tau = u32_to_unit_float(rr[0]);
        }
        if (a.tau_out) a.tau_out[(int64_t)ps.stream_id * a.maxT * B + (int64_t)t * B + b] = tau;
        rowf[tid] = tau;
        rowb[tid] = b;
    }
    __syncthreads();
    // cos basis: c[m][k] = cos(tau * (k+1) * pi), two fp32 multiplies as torch does (iqn_model.py:90-92)
    for (int idx = tid; idx < 16 * K_BASIS; idx += 512) {
        const int m = idx >> 6, k = idx & 63;
        const float x = (rowf[m] * (float)(k + 1)) * PI_F;
        const float c = cosf(x);
        cost[m * CS + k] = c;
        if (ps.save) a.ws.cosb[(int64_t)(r0 + m) * K_BASIS + k] = c;
    }
    __syncthreads();

    // ---- phi GEMM (16 x 1024, K = 64) + bias + ReLU + Hadamard with e -> ytile ----------------
    {
        float4 afr[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) afr[q] = *reinterpret_cast<const float4 *>(&cost[li * CS + 16 * q + 4 * g]);
        const float *Wphi = P + a.off.phi_w, *bphi = P + a.off.phi_b;
        const float *erow = ps.e + (int64_t)rowb[4 * g] * E_DIM;   // rows 4g..4g+3 share a sample (T % 4 == 0)
        float4 bfr[4], bnx[4];
        {
            const float *src = Wphi + (int64_t)(w * 128 + li) * K_BASIS + 4 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q) bfr[q] = *reinterpret_cast<const float4 *>(src + 16 * q);
        }
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) {
            const int n = w * 128 + nt * 16 + li;
            if (nt < 7) {
                const float *src = Wphi + (int64_t)(n + 16) * K_BASIS + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) bnx[q] = *reinterpret_cast<const float4 *>(src + 16 * q);
            }
            const float bias = bphi[n], ev = erow[n];
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc = mfma16(afr[q].x, bfr[q].x, acc);
                acc = mfma16(afr[q].y, bfr[q].y, acc);
                acc = mfma16(afr[q].z, bfr[q].z, acc);
                acc = mfma16(afr[q].w, bfr[q].w, acc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) ytile[(4 * g + r) * YS + n] = fmaxf(acc[r] + bias, 0.f) * ev;
#pragma unroll
            for (int q = 0; q < 4; ++q) bfr[q] = bnx[q];
        }
    }
    __syncthreads();

    // ---- LayerNorm(1024): wave w owns rows 2w, 2w+1; normalise in place ------------------------
    {
        const float *g1 = P + a.off.iqn_ln1_g, *be1 = P + a.off.iqn_ln1_b;
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = 2 * w + rr;
            float4 x[4];
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i] = *reinterpret_cast<const float4 *>(&ytile[m * YS + (i * 64 + lane) * 4]);
                s += (x[i].x + x[i].y) + (x[i].z + x[i].w);
            }
            const float mean = wave_sum(s) * (1.0f / E_DIM);
            float v = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i].x -= mean; x[i].y -= mean; x[i].z -= mean; x[i].w -= mean;
                v += (x[i].x * x[i].x + x[i].y * x[i].y) + (x[i].z * x[i].z + x[i].w * x[i].w);
            }
            const float var = wave_sum(v) * (1.0f / E_DIM);
            const float rstd = 1.0f / sqrtf(var + LN_EPS);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = (i * 64 + lane) * 4;
                const float4 gg = *reinterpret_cast<const float4 *>(g1 + n);
                const float4 bb = *reinterpret_cast<const float4 *>(be1 + n);
                float4 y;
                y.x = x[i].x * rstd * gg.x + bb.x;
                y.y = x[i].y * rstd * gg.y + bb.y;
                y.z = x[i].z * rstd * gg.z + bb.z;
                y.w = x[i].w * rstd * gg.w + bb.w;
                *reinterpret_cast<float4 *>(&ytile[m * YS + n]) = y;
            }
            if (ps.save && lane == 0) {
                a.ws.mu1[r0 + m] = mean;
                a.ws.rstd1[r0 + m] = rstd;
            }
        }
    }
    __syncthreads();

    // ---- trunk GEMM (16 x 128, K = 1024): wave w owns output columns 16w..16w+15 -------------
    {
        const float *brow = P + a.off.iqn_w1 + (int64_t)(16 * w + li) * E_DIM + 4 * g;
        const float *arow = ytile + li * YS + 4 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        float4 breg[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) breg[i] = *reinterpret_cast<const float4 *>(brow + 16 * i);
        for (int qo = 0; qo < 8; ++qo) {
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) {
                const int q = qo * 8 + qi;
                const float4 av = *reinterpret_cast<const float4 *>(arow + 16 * q);
                const float4 bv = breg[qi];
                if (qo < 7) breg[qi] = *reinterpret_cast<const float4 *>(brow + 16 * (q + 8));
                acc = mfma16(av.x, bv.x, acc);
                acc = mfma16(av.y, bv.y, acc);
                acc = mfma16(av.z, bv.z, acc);
                acc = mfma16(av.w, bv.w, acc);
            }
        }
        const int h = 16 * w + li;
        const float bias = P[a.off.iqn_b1 + h];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = acc[r] + bias;
            h1[(4 * g + r) * HS + h] = fmaxf(pre, 0.f);
            if (ps.save) a.ws.pre1[(int64_t)(r0 + 4 * g + r) * H_DIM + h] = pre;
        }
    }
    __syncthreads();

    // ---- LayerNorm(128) + head (128 -> A): wave w owns rows 2w, 2w+1 ---------------------------
    {
        const float *g2 = P + a.off.iqn_ln2_g, *be2 = P + a.off.iqn_ln2_b;
        const float *W2 = P + a.off.iqn_w2, *b2 = P + a.off.iqn_b2;
        const float g2a = g2[lane], g2b = g2[lane + 64], b2a = be2[lane], b2b = be2[lane + 64];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int m = 2 * w + rr;
            float x0 = h1[m * HS + lane], x1 = h1[m * HS + 64 + lane];
            const float mean = wave_sum(x0 + x1) * (1.0f / H_DIM);
            x0 -= mean;
            x1 -= mean;
            const float var = wave_sum(x0 * x0 + x1 * x1) * (1.0f / H_DIM);
            const float rstd = 1.0f / sqrtf(var + LN_EPS);
            const float xh0 = x0 * rstd, xh1 = x1 * rstd;
            const float y0 = xh0 * g2a + b2a, y1 = xh1 * g2b + b2b;
            if (ps.save) {
                a.ws.xhat2[(int64_t)(r0 + m) * H_DIM + lane] = xh0;
                a.ws.xhat2[(int64_t)(r0 + m) * H_DIM + 64 + lane] = xh1;
                if (lane == 0) a.ws.rstd2[r0 + m] = rstd;
            }
            for (int aa = 0; aa < A; ++aa) {
                const float z = wave_sum(y0 * W2[aa * H_DIM + lane] + y1 * W2[aa * H_DIM + 64 + lane]) + b2[aa];
                if (lane == 0) ps.z_out[(int64_t)(r0 + m) * A + aa] = z;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// loss: one wave (64 lanes) per sample.  T, T' must divide 64.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void iqn_loss_kernel(IqnArgs a) {
    __shared__ float s_zc[64 * 16], s_zo[64 * 16], s_zt[64 * 16];
    __shared__ float s_y[64], s_q[64], s_tau[64], s_dq[64];
    __shared__ int s_astar;
    const int b = blockIdx.x, lane = threadIdx.x;
    const int B = a.B, A = a.A, T = a.T, Tn = a.Tn;
    const float kap = a.huber_k;
    const float *zon_g = a.ws.zon, *ztg_g = a.ws.ztg;
    for (int i = lane; i < T * A; i += 64) s_zc[i] = a.ws.zcur[(int64_t)b * T * A + i];
    for (int i = lane; i < Tn * A; i += 64) {
        s_zo[i] = zon_g[(int64_t)b * Tn * A + i];
        s_zt[i] = ztg_g[(int64_t)b * Tn * A + i];
    }
    // quantile samples of the current-state pass (tau_out slot 0 always holds them)
    if (lane < T) s_tau[lane] = a.tau_out[(int64_t)lane * B + b];
    __syncthreads();
    if (lane == 0) {
        // a* = argmax_a mean_j Zon[j][a]  (first maximum wins, iqn_model.py:129-133)
        int best = 0;
        float bestv = 0.f;
        for (int aa = 0; aa < A; ++aa) {
            float s = 0.f;
            for (int j = 0; j < Tn; ++j) s += s_zo[j * A + aa];
            s = s / (float)Tn;
            if (aa == 0 || s > bestv) {
                bestv = s;
                best = aa;
            }
        }
        s_astar = best;
    }
    __syncthreads();
    const int act = (int)a.action[b];
    const float R = a.reward[b];
    const float dg = a.gamma[b] * (a.nonterminal[b] ? 1.0f : 0.0f);
    if (lane < Tn) s_y[lane] = R + s_zt[lane * A + s_astar] * dg;   // separate mul and add (iqn_model.py:145)
    if (lane < T) s_q[lane] = s_zc[lane * A + act];
    __syncthreads();
    // pairwise quantile-Huber tile: pair p = j*T + t; lane keeps a fixed t because T | 64
    float lsum = 0.f, gq = 0.f;
    const int t_l = lane % T;
    for (int p = lane; p < T * Tn; p += 64) {
        const int j = p / T;
        const float d = s_y[j] - s_q[t_l];
        const float ad = fabsf(d);
        const float hub = (ad <= kap) ? 0.5f * (d * d) : kap * (ad - 0.5f * kap);
        const float wgt = fabsf(s_tau[t_l] - (d < 0.f ? 1.0f : 0.0f));
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
        if (a.out_td) a.out_td[b] = dl;      // IQN only: td_errors = distribution_loss (composite_model.py:138-139)
        a.ws.lossw[b] = dl * wb;
    }
    __syncthreads();

    // head + LayerNorm(128) backward for the T current-state rows of this sample
    const float *P = a.params;
    const float *W2 = P + a.off.iqn_w2 + (int64_t)act * H_DIM;
    const float *g2 = P + a.off.iqn_ln2_g, *b1 = P + a.off.iqn_b1;
    const float w2a = W2[lane] * g2[lane], w2b = W2[lane + 64] * g2[lane + 64];  // d xhat2 / dq
    const float ua = a.ws.uv[lane], ub = a.ws.uv[lane + 64];
    const float va = a.ws.uv[H_DIM + lane] + b1[lane], vb = a.ws.uv[H_DIM + lane + 64] + b1[lane + 64];
    float Sa = 0.f, Sbb = 0.f, Pa = 0.f, Pbb = 0.f, Dsum = 0.f;
    for (int t = 0; t < T; ++t) {
        const int64_t r = (int64_t)b * T + t;
        const float dq = s_dq[t];
        const float xa = a.ws.xhat2[r * H_DIM + lane], xb = a.ws.xhat2[r * H_DIM + 64 + lane];
        const float pa = a.ws.pre1[r * H_DIM + lane], pb = a.ws.pre1[r * H_DIM + 64 + lane];
        const float rstd = a.ws.rstd2[r];
        const float da = dq * w2a, db = dq * w2b;
        const float m1 = wave_sum(da + db) * (1.0f / H_DIM);
        const float m2 = wave_sum(da * xa + db * xb) * (1.0f / H_DIM);
        float ga = rstd * (da - m1 - xa * m2), gb = rstd * (db - m1 - xb * m2);
        ga = pa > 0.f ? ga : 0.f;
        gb = pb > 0.f ? gb : 0.f;
        a.ws.dpre1[r * H_DIM + lane] = ga;
        a.ws.dpre1[r * H_DIM + 64 + lane] = gb;
        const float c1 = wave_sum(ga * ua + gb * ub);
        const float c2 = wave_sum(ga * (pa - va) + gb * (pb - vb));
        if (lane == 0) {
            a.ws.c1[r] = c1;
            a.ws.c2[r] = c2;
            a.ws.dq[r] = dq;
        }
        Sa += dq * xa;
        Sbb += dq * xb;
        Pa += ga;
        Pbb += gb;
        Dsum += dq;
    }
    a.ws.Sb[(int64_t)b * H_DIM + lane] = Sa;
    a.ws.Sb[(int64_t)b * H_DIM + 64 + lane] = Sbb;
    a.ws.Pb[(int64_t)b * H_DIM + lane] = Pa;
    a.ws.Pb[(int64_t)b * H_DIM + 64 + lane] = Pbb;
    if (lane == 0) a.ws.Db[b] = Dsum;
}

// ------------------------------------------------------------------------------------------
// bwd: grid = (E/16 column slices) x n_chunks row chunks, 256 threads = 4 waves.
// Each wave walks a contiguous run of 16-row tiles; everything a tile needs comes straight from
// global/L2 into registers, LDS only transposes the two operands that are needed k-major.
// ------------------------------------------------------------------------------------------
constexpr int BWD_WAVE_LDS = 16 * CS + 16 * HS;     // floats per wave (cos tile + dpre1 tile)
constexpr int BWD_ACC = 16 + 32 + 3;                // accumulators reduced across waves

__global__ __launch_bounds__(256) void iqn_bwd_kernel(IqnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int cs = blockIdx.x % (E_DIM / 16), rc = blockIdx.x / (E_DIM / 16);
    const int n = cs * 16 + j;
    const int T = a.T;
    const int R = a.B * T;
    // contiguous, balanced run of tiles per wave, in units that keep a sample's rows together
    const int unit = T > 16 ? T / 16 : 1;
    const int units_total = (R / 16) / unit;
    const int gw = rc * 4 + w, nw = a.n_chunks * 4;
    const int tile_begin = (int)(((int64_t)units_total * gw) / nw) * unit;
    const int tiles_per_wave = (int)(((int64_t)units_total * (gw + 1)) / nw) * unit - tile_begin;
    float *cosl = smem + w * BWD_WAVE_LDS;
    float *dpl = cosl + 16 * CS;
    const float *P = a.params;

    // per-lane constants ------------------------------------------------------------------------
    float4 wphi[4];   // B operand of phi: Wphi[n][16q + 4g + jj]
    {
        const float *src = P + a.off.phi_w + (int64_t)n * K_BASIS + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) wphi[q] = *reinterpret_cast<const float4 *>(src + 16 * q);
    }
    float w1f[32];    // B operand of dX: W1[h = 16q + 4g + jj][n]
    {
        const float *src = P + a.off.iqn_w1 + n;
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w1f[q * 4 + jj] = src[(int64_t)(16 * q + 4 * g + jj) * E_DIM];
    }
    const float bphi = P[a.off.phi_b + n], g1 = P[a.off.iqn_ln1_g + n], be1 = P[a.off.iqn_ln1_b + n];

    f32x4 accWphi[4], accW1[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) accWphi[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_dg = 0.f, s_db = 0.f, s_dbphi = 0.f, de_acc = 0.f;

    for (int ti = 0; ti < tiles_per_wave; ++ti) {
        const int r0 = (tile_begin + ti) * 16;
        // A fragments (row = r0 + j, k = 16q + 4g + jj)
        float4 ac[4], ad[8];
        {
            const float *src = a.ws.cosb + (int64_t)(r0 + j) * K_BASIS + 4 * g;
#pragma unroll
            for (int q = 0; q < 4; ++q) ac[q] = *reinterpret_cast<const float4 *>(src + 16 * q);
            const float *sd = a.ws.dpre1 + (int64_t)(r0 + j) * H_DIM + 4 * g;
#pragma unroll
            for (int q = 0; q < 8; ++q) ad[q] = *reinterpret_cast<const float4 *>(sd + 16 * q);
        }
        // stage both tiles in LDS for the k-major reads of the weight-gradient products
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4 *>(&cosl[j * CS + 16 * q + 4 * g]) = ac[q];
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float4 *>(&dpl[j * HS + 16 * q + 4 * g]) = ad[q];

        // row scalars for the D-layout rows 4g..4g+3
        const int rb = r0 + 4 * g;
        const int bsm = rb / T;
        const float4 mu = *reinterpret_cast<const float4 *>(a.ws.mu1 + rb);
        const float4 rs = *reinterpret_cast<const float4 *>(a.ws.rstd1 + rb);
        const float4 c1 = *reinterpret_cast<const float4 *>(a.ws.c1 + rb);
        const float4 c2 = *reinterpret_cast<const float4 *>(a.ws.c2 + rb);
        const float ev = a.ws.e_cur[(int64_t)bsm * E_DIM + n];

        // phi columns and dX columns, two independent MFMA chains interleaved
        f32x4 aphi = {0.f, 0.f, 0.f, 0.f}, adx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            aphi = mfma16(ac[q].x, wphi[q].x, aphi);
            adx = mfma16(ad[2 * q].x, w1f[8 * q + 0], adx);
            aphi = mfma16(ac[q].y, wphi[q].y, aphi);
            adx = mfma16(ad[2 * q].y, w1f[8 * q + 1], adx);
            aphi = mfma16(ac[q].z, wphi[q].z, aphi);
            adx = mfma16(ad[2 * q].z, w1f[8 * q + 2], adx);
            aphi = mfma16(ac[q].w, wphi[q].w, aphi);
            adx = mfma16(ad[2 * q].w, w1f[8 * q + 3], adx);
            adx = mfma16(ad[2 * q + 1].x, w1f[8 * q + 4], adx);
            adx = mfma16(ad[2 * q + 1].y, w1f[8 * q + 5], adx);
            adx = mfma16(ad[2 * q + 1].z, w1f[8 * q + 6], adx);
            adx = mfma16(ad[2 * q + 1].w, w1f[8 * q + 7], adx);
        }
        // elementwise backward on the 4 rows this lane holds (column n)
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
        const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, c2v[4] = {c2.x, c2.y, c2.z, c2.w};
        float xv[4], dpp[4];
        float dep = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float phi = fmaxf(aphi[r] + bphi, 0.f);
            const float h0 = phi * ev;
            const float xhat = (h0 - muv[r]) * rsv[r];
            xv[r] = xhat * g1 + be1;                  // LN output (B operand of dW1)
            const float dX = adx[r];
            s_dg += dX * xhat;
            s_db += dX;
            const float dh0 = rsv[r] * (dX * g1 - c1v[r] * (1.0f / E_DIM) - xhat * (c2v[r] * (1.0f / E_DIM)));
            dep += dh0 * phi;
            const float dphi = (phi > 0.f) ? dh0 * ev : 0.f;
            dpp[r] = dphi;
            s_dbphi += dphi;
        }
        // d e[b][n]: sum over the T rows of a sample
        if (T == 4) {
            a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
        } else if (T == 8) {
            dep += __shfl_xor(dep, 16, 64);
            if ((g & 1) == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = dep;
        } else {
            dep += __shfl_xor(dep, 16, 64);
            dep += __shfl_xor(dep, 32, 64);
            de_acc += dep;
            if (((r0 + 16) % T) == 0) {
                if (g == 0) a.ws.de_iqn[(int64_t)bsm * E_DIM + n] = de_acc;
                de_acc = 0.f;
            }
        }
        // dWphi[n-slice][64] += dphi^T (16 cols x 16 rows) . cos (16 rows x 64): A = dpp (D layout == A^T layout)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accWphi[kt] = mfma16(dpp[r], cosl[(4 * g + r) * CS + 16 * kt + j], accWphi[kt]);
        }
        // dW1[128][n-slice] += dpre1^T (128 x 16 rows) . X (16 rows x 16 cols): B = xv
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
#pragma unroll
            for (int r = 0; r < 4; ++r) accW1[mt] = mfma16(dpl[(4 * g + r) * HS + 16 * mt + j], xv[r], accW1[mt]);
        }
    }

    // ---- reduce the four waves in fixed order and write this workgroup's slab part -------------
    __syncthreads();
    float *red = smem;    // [4 waves][BWD_ACC][64]
    {
        float *mine = red + (w * BWD_ACC) * 64 + lane;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(kt * 4 + r) * 64] = accWphi[kt][r];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 + mt * 4 + r) * 64] = accW1[mt][r];
        // column sums: fold the 4 row groups (lanes j, j+16, j+32, j+48)
        s_dg += __shfl_xor(s_dg, 16, 64);
        s_dg += __shfl_xor(s_dg, 32, 64);
        s_db += __shfl_xor(s_db, 16, 64);
        s_db += __shfl_xor(s_db, 32, 64);
        s_dbphi += __shfl_xor(s_dbphi, 16, 64);
        s_dbphi += __shfl_xor(s_dbphi, 32, 64);
        mine[48 * 64] = s_dg;
        mine[49 * 64] = s_db;
        mine[50 * 64] = s_dbphi;
    }
    __syncthreads();
    float *slab = a.ws.slabs + (int64_t)rc * SLAB;
    for (int idx = tid; idx < BWD_ACC * 64; idx += 256) {
        const int slot = idx >> 6, l = idx & 63;
        const float v = ((red[(0 * BWD_ACC + slot) * 64 + l] + red[(1 * BWD_ACC + slot) * 64 + l]) +
                         red[(2 * BWD_ACC + slot) * 64 + l]) + red[(3 * BWD_ACC + slot) * 64 + l];
        const int lj = l & 15, lg = l >> 4;
        if (slot < 16) {
            // accWphi[kt][r]: row (n index) = 4*lg + r, col (basis) = 16*kt + lj
            const int kt = slot >> 2, r = slot & 3;
            slab[(int64_t)(cs * 16 + 4 * lg + r) * K_BASIS + 16 * kt + lj] = v;
        } else if (slot < 48) {
            // accW1[mt][r]: row h = 16*mt + 4*lg + r, col n = cs*16 + lj
            const int mt = (slot - 16) >> 2, r = (slot - 16) & 3;
            slab[(int64_t)(E_DIM * K_BASIS + 3 * E_DIM) + (int64_t)(16 * mt + 4 * lg + r) * E_DIM + cs * 16 + lj] = v;
        } else if (lg == 0) {
            const int nn = cs * 16 + lj;
            if (slot == 48) slab[E_DIM * K_BASIS + E_DIM + nn] = v;           // d ln1_g
            else if (slot == 49) slab[E_DIM * K_BASIS + 2 * E_DIM + nn] = v;  // d ln1_b
            else slab[E_DIM * K_BASIS + nn] = v;                              // d phi_b
        }
    }
}

// ------------------------------------------------------------------------------------------
// small: blocks [0, 16*CONV_CHUNKS): conv backward partials; last block: b1, LN2, W2, b2.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void iqn_small_kernel(IqnArgs a) {
    __shared__ float s_obs[1024];
    __shared__ float s_dc[64];
    __shared__ float s_red[1024 + 64];
    const int tid = threadIdx.x, B = a.B, C = a.C, A = a.A;
    const int n_conv_blocks = 16 * CONV_CHUNKS;
    if ((int)blockIdx.x < n_conv_blocks) {
        const int c = blockIdx.x / CONV_CHUNKS, ch = blockIdx.x % CONV_CHUNKS;
        const int per = (B + CONV_CHUNKS - 1) / CONV_CHUNKS;
        const int b0 = ch * per, b1 = min(B, b0 + per);
        const int nk = 9 * C;                 // (ci, dy, dx)
        const bool active = tid < nk * 8;     // thread = (k, y)
        const int k = tid >> 3, y = tid & 7;
        const int ci = k / 9, dy = (k % 9) / 3, dx = k % 3;
        float acc = 0.f, bacc = 0.f;
        for (int b = b0; b < b1; ++b) {
            __syncthreads();
            for (int i = tid; i < 100 * C; i += 1024) s_obs[i] = a.obs[(int64_t)b * 100 * C + i];
            if (tid < 64) {
                const int64_t o = (int64_t)b * E_DIM + c * 64 + tid;
                const float d = a.propagate_grad ? a.ws.de_iqn[o] : 0.f;
                s_dc[tid] = a.ws.e_cur[o] > 0.f ? d : 0.f;
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int x = 0; x < 8; ++x) acc = fmaf(s_dc[y * 8 + x], s_obs[((y + dy) * 10 + (x + dx)) * C + ci], acc);
            }
            if (tid < 64) bacc += s_dc[tid];
        }
        __syncthreads();
        s_red[tid] = active ? acc : 0.f;
        if (tid < 64) s_red[1024 + tid] = bacc;
        __syncthreads();
        float *out = a.ws.convpart + (int64_t)(c * CONV_CHUNKS + ch) * 96;
        if (tid < nk) {
            float s = 0.f;
#pragma unroll
            for (int yy = 0; yy < 8; ++yy) s += s_red[tid * 8 + yy];
            out[tid] = s;
        }
        if (tid == 0) {
            float s = 0.f;
            for (int i = 0; i < 64; ++i) s += s_red[1024 + i];
            out[95] = s;
        }
        return;
    }
    // ---- small tensors: S[a][h] = sum_{b: act=a} Sb[b][h]; D[a]; db1[h] = sum_b Pb[b][h] -------
    __shared__ float s_S[16 * H_DIM];
    __shared__ float s_D[16];
    const int h = tid & 127, part = tid >> 7;       // 8 parts over the batch
    const int per = (B + 7) / 8, b0 = part * per, b1 = min(B, b0 + per);
    float *gr = a.grads;
    const float *P = a.params;
    float sq = 0.f;
    for (int aa = 0; aa < A; ++aa) {
        float s = 0.f;
        for (int b = b0; b < b1; ++b)
            if ((int)a.action[b] == aa) s += a.ws.Sb[(int64_t)b * H_DIM + h];
        __syncthreads();
        s_red[tid] = s;
        __syncthreads();
        if (part == 0) {
            float t = 0.f;
            for (int p = 0; p < 8; ++p) t += s_red[p * 128 + h];
            s_S[aa * H_DIM + h] = t;
        }
    }
    {
        float s = 0.f;
        for (int b = b0; b < b1; ++b) s += a.ws.Pb[(int64_t)b * H_DIM + h];
        __syncthreads();
        s_red[tid] = s;
        __syncthreads();
        if (part == 0) {
            float t = 0.f;
            for (int p = 0; p < 8; ++p) t += s_red[p * 128 + h];
            gr[a.off.iqn_b1 + h] = t;
            sq += t * t;
        }
    }
    if (tid < A) {
        float s = 0.f;
        for (int b = 0; b < B; ++b)
            if ((int)a.action[b] == tid) s += a.ws.Db[b];
        s_D[tid] = s;
        gr[a.off.iqn_b2 + tid] = s;
        sq += s * s;
    }
    __syncthreads();
    if (part == 0) {
        const float g2 = P[a.off.iqn_ln2_g + h], be2 = P[a.off.iqn_ln2_b + h];
        float dg = 0.f, db = 0.f;
        for (int aa = 0; aa < A; ++aa) {
            const float w2 = P[a.off.iqn_w2 + aa * H_DIM + h];
            const float S = s_S[aa * H_DIM + h], D = s_D[aa];
            const float dw = g2 * S + be2 * D;
            gr[a.off.iqn_w2 + aa * H_DIM + h] = dw;
            sq += dw * dw;
            dg += w2 * S;
            db += w2 * D;
        }
        gr[a.off.iqn_ln2_g + h] = dg;
        gr[a.off.iqn_ln2_b + h] = db;
        sq += dg * dg + db * db;
    }
    // total loss (agent.py:58-64) and this block's sum of squares
    __syncthreads();
    s_red[tid] = sq;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
        for (int i = 0; i < 1024; ++i) t += s_red[i];
        a.ws.normpart[NORM_SLOTS - 1] = t;
        float l = 0.f;
        for (int b = 0; b < B; ++b) l += a.ws.lossw[b];
        l = l / (float)B;
        a.out_scalars[0] = l;
        a.out_scalars[1] = l;
        a.out_scalars[2] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// reduce: slabs -> grads[phi_w .. w1], conv partials -> grads[conv]; per-block sum of squares.
// grid = REDUCE_BLOCKS, 256 threads, grid-stride over SLAB/4 float4 + conv.
// ------------------------------------------------------------------------------------------
constexpr int REDUCE_BLOCKS = 195;   // SLAB/4 = 49920 float4 = 195 * 256

__global__ __launch_bounds__(256) void iqn_reduce_kernel(IqnArgs a) {
    __shared__ float s_red[256];
    const int tid = threadIdx.x;
    float sq = 0.f;
    const int nvec = SLAB / 4;
    float *gbase = a.grads + a.off.phi_w;
    for (int i = blockIdx.x * 256 + tid; i < nvec; i += gridDim.x * 256) {
        float4 s = reinterpret_cast<const float4 *>(a.ws.slabs)[i];
        for (int c = 1; c < a.n_chunks; ++c) {
            const float4 v = reinterpret_cast<const float4 *>(a.ws.slabs + (int64_t)c * SLAB)[i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        gbase[4 * i + 0] = s.x;
        gbase[4 * i + 1] = s.y;
        gbase[4 * i + 2] = s.z;
        gbase[4 * i + 3] = s.w;
        sq += (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
    }
    if (blockIdx.x == 0) {
        const int nk = 9 * a.C;
        for (int i = tid; i < 16 * nk + 16; i += 256) {
            float s = 0.f;
            if (i < 16 * nk) {
                const int c = i / nk, k = i % nk;
                for (int ch = 0; ch < CONV_CHUNKS; ++ch) s += a.ws.convpart[(int64_t)(c * CONV_CHUNKS + ch) * 96 + k];
                a.grads[a.off.conv_w + i] = s;
            } else {
                const int c = i - 16 * nk;
                for (int ch = 0; ch < CONV_CHUNKS; ++ch) s += a.ws.convpart[(int64_t)(c * CONV_CHUNKS + ch) * 96 + 95];
                a.grads[a.off.conv_b + c] = s;
            }
            sq += s * s;
        }
    }
    s_red[tid] = sq;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) s_red[tid] += s_red[tid + o];
        __syncthreads();
    }
    if (tid == 0) a.ws.normpart[blockIdx.x] = s_red[0];
}

}  // namespace prism
