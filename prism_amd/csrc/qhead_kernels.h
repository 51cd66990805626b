// Q-ensemble (IDS) / 2-layer DQN heads: loss, backward and the block routines of their gradient
// reduction.  Restates /root/reference/prism/agents/models/q_ensemble.py:44-92 for heads of the form
//   [LayerNorm(1024)] -> Linear(1024 -> H) -> ReLU -> [LayerNorm(H)] -> Linear(H -> A),  H in {128, 256}
// (ffnn_model.py:61-76 with n_layers = 2).  The forward of a head tile is kind 1 of fwd_tile_kernel.
// Rows of all per-head workspace arrays are indexed  head * B + sample.
#pragma once
#include "iqn_kernels.h"

namespace prism {

__host__ __device__ inline int q_slab_floats(int H, int ln) { return (ln ? 2 * E_DIM : 0) + H * E_DIM; }   // per head: [ln1_g | ln1_b |] w1
constexpr int Q_MAX_HEADS = 16;

// Head tensors start at arbitrary float offsets of the flat parameter buffer (head stride 134 278
// floats), so they are only 4-byte aligned: no 16-byte accesses on them.
__device__ __forceinline__ float4 ld4u(const float *p) { return float4{p[0], p[1], p[2], p[3]}; }

// ---- front-kernel roles -------------------------------------------------------------------------
// stream-packed W1 (* ln1_g) of head `hd`: the IQN copy's layout without the phi slots (iqn_kernels.h)
__device__ __forceinline__ void pack_head_w1_block(const float *__restrict__ P, const prism_param_offsets &off, int H, int ln,
                                                   float *__restrict__ pk, int hd, int blk, int tid) {
    const int r = blk * 256 + tid;               // packed float4 index within the head, < H*E/4
    const int lane = r & 63, li = lane & 15, g = lane >> 4;
    const int NHT = H / 16, grp = r >> 6, step = grp / NHT, ht = grp - step * NHT;
    const float *Ph = P + off.head_base + (int64_t)hd * off.head_stride;
    const int n0 = 16 * step + 4 * g;
    float4 v = ld4u(Ph + off.h_w1 + (int64_t)(16 * ht + li) * E_DIM + n0);
    if (ln) {
        const float4 gg = ld4u(Ph + off.h_ln1_g + n0);
        v.x *= gg.x; v.y *= gg.y; v.z *= gg.z; v.w *= gg.w;
    }
    reinterpret_cast<float4 *>(pk + (size_t)hd * H * E_DIM)[r] = v;
}
__host__ __device__ inline int q_pack_blocks_per_head(int H) { return H * E_DIM / 4 / 256; }

// u_h[hh] = sum_n W1_h[hh][n] g1_h[n] (whole and per K slice, iqn_kernels.h UV_ROWS),  v_h[hh] = sum_n W1_h[hh][n] beta1_h[n]
// (one wave per (set, head, hh))
__device__ __forceinline__ void q_uv_block(const IqnArgs &a, int set, int hd, int hh, int lane) {
    const float *Ph = (set ? a.target_params : a.params) + a.off.head_base + (int64_t)hd * a.off.head_stride;
    const float *W1 = Ph + a.off.h_w1 + (int64_t)hh * E_DIM, *g1 = Ph + a.off.h_ln1_g, *b1 = Ph + a.off.h_ln1_b;
    float su[UV_SLICES], sv = 0.f;
#pragma unroll
    for (int s = 0; s < UV_SLICES; ++s) {         // columns lane + 64 k, k = 2 s, 2 s + 1: K slice s
        const int n0 = lane + 128 * s, n1 = n0 + 64;
        const float w0 = W1[n0], w1 = W1[n1];
        su[s] = w0 * g1[n0] + w1 * g1[n1];
        sv += w0 * b1[n0] + w1 * b1[n1];
    }
    float *uv = a.ws.q_uv + ((size_t)set * a.n_heads + hd) * UV_ROWS * a.Hq;
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < UV_SLICES; ++s) {
        const float t = wave_sum(su[s]);
        if (lane == 0) uv[(2 + s) * a.Hq + hh] = t;
        tot += t;
    }
    sv = wave_sum(sv);
    if (lane == 0) {
        uv[hh] = tot;
        uv[a.Hq + hh] = sv;
    }
}

// ||theta_h||^2 of one head, Q_NORM_PARTS blocks of 256 threads each -> q_kappa[(hd * Q_NORM_PARTS + part)]
constexpr int Q_NORM_PARTS = 16;
__device__ __forceinline__ void q_head_norm_block(const IqnArgs &a, int hd, int part, float *s_red) {
    const int tid = threadIdx.x;
    const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
    const int64_t per = (a.off.head_stride + Q_NORM_PARTS - 1) / Q_NORM_PARTS;
    const int64_t i0 = part * per, i1 = i0 + per < a.off.head_stride ? i0 + per : a.off.head_stride;
    float s = 0.f;
#pragma unroll 8
    for (int64_t i = i0 + tid; i < i1; i += 256) {
        const float x = Ph[i];
        s += x * x;
    }
    s = wave_sum(s);
    if ((tid & 63) == 0) s_red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) a.ws.q_kappa[hd * Q_NORM_PARTS + part] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// Theil index of the head norms and its gradient factors (q_ensemble.py:86-90):
//   r_h = n_h / mean(n), theil = mean(r log r);  d theil / d theta_{h,i} = c_h * theta_{h,i}
__device__ __forceinline__ void theil_factors(const float *norm2, int Hd, float *c_out, float &theil) {
    float n[Q_MAX_HEADS];
    float m = 0.f;
    for (int h = 0; h < Hd; ++h) {
        n[h] = sqrtf(norm2[h]);
        m += n[h];
    }
    m /= (float)Hd;
    float t = 0.f, mix = 0.f;
    for (int h = 0; h < Hd; ++h) {
        const float r = n[h] / m;
        t += r * logf(r);
        mix += (logf(r) + 1.0f) * r;
    }
    theil = t / (float)Hd;
    mix /= (float)Hd;
    for (int h = 0; h < Hd; ++h) {
        const float r = n[h] / m;
        c_out[h] = ((logf(r) + 1.0f) - mix) / ((float)Hd * m * n[h]);
    }
}
// all threads: gather the Q_NORM_PARTS partial sums of every head into s_norm2[Hd] (one parallel round
// of loads instead of a serial chain in one lane); caller synchronises before reading
__device__ __forceinline__ void stage_head_norms(const IqnArgs &a, float *s_parts, float *s_norm2) {
    const int tid = threadIdx.x, n = a.n_heads * Q_NORM_PARTS;
    if (tid < n) s_parts[tid] = a.ws.q_kappa[tid];
    __syncthreads();
    if (tid < a.n_heads) {
        float s = 0.f;
#pragma unroll
        for (int p = 0; p < Q_NORM_PARTS; ++p) s += s_parts[tid * Q_NORM_PARTS + p];
        s_norm2[tid] = s;
    }
}

// ------------------------------------------------------------------------------------------
// q loss: one workgroup (8 waves) per sample.  MSE against the n-step target per head, then the
// head + LayerNorm(H) backward of the sample's row in every head.
// ------------------------------------------------------------------------------------------
template <int H, bool LN>
__device__ __forceinline__ void qh_loss_body(const IqnArgs &a, const int b) {
    constexpr int KH = H / 64;
    __shared__ float s_zc[Q_MAX_HEADS * 16], s_zo[Q_MAX_HEADS * 16], s_zt[Q_MAX_HEADS * 16];
    __shared__ float s_dq[Q_MAX_HEADS], s_sq[Q_MAX_HEADS];
    __shared__ float s_parts[Q_MAX_HEADS * Q_NORM_PARTS], s_norm2[Q_MAX_HEADS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int B = a.B, A = a.A, Hd = a.n_heads;
    // saved activations of this wave's rows (heads w, w + 8): in flight before the loss is known
    float xa[2][KH], pa[2][KH], rs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hd = w + 8 * i;
        rs[i] = 1.f;
        if (hd < Hd) {
            const int64_t r = (int64_t)hd * B + b;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                xa[i][k] = a.ws.q_xhat2[r * H + 64 * k + lane];
                pa[i][k] = a.ws.q_pre1[r * H + 64 * k + lane];
            }
            if (LN) rs[i] = a.ws.q_rstd2[r];
        }
    }
    const int act = (int)a.action[b];
    if (a.theil_coef != 0.f) stage_head_norms(a, s_parts, s_norm2);
    for (int i = tid; i < Hd * A; i += 512) {
        const int hd = i / A, aa = i - hd * A;
        const int64_t o = ((int64_t)hd * B + b) * A + aa;
        s_zc[i] = a.ws.zq_cur[o];
        s_zo[i] = a.ws.zq_on[o];
        s_zt[i] = a.ws.zq_tg[o];
    }
    __syncthreads();
    const float wb = a.per_weights ? a.per_weights[b] : 1.0f;
    if (tid < Hd) {
        // a*_{b,h} = argmax_a Qon_next[b,a,h] (first maximum); y = R + Qtg_next[b,a*,h] * gamma*nonterminal
        int best = 0;
        float bv = s_zo[tid * A];
        for (int aa = 1; aa < A; ++aa) {
            const float v = s_zo[tid * A + aa];
            if (v > bv) {
                bv = v;
                best = aa;
            }
        }
        const float dg = a.gamma[b] * (a.nonterminal[b] ? 1.0f : 0.0f);
        const float y = td_target(a.squish, a.reward[b], s_zt[tid * A + best], dg);      // (q_ensemble.py:77-82)
        const float diff = s_zc[tid * A + act] - y;
        s_sq[tid] = diff * diff;
        s_dq[tid] = (wb / (float)B) * a.q_w * (2.0f / (float)Hd) * diff;
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int h = 0; h < Hd; ++h) s += s_sq[h];
        s = s / (float)Hd;
        float theil = 0.f;
        if (a.theil_coef != 0.f) {
            float c[Q_MAX_HEADS];
            theil_factors(s_norm2, Hd, c, theil);
        }
        const float ql = a.q_w * (s - theil * a.theil_coef);
        a.out_ql[b] = ql;
        a.ws.q_lossw[b] = ql * wb;
        // td errors (composite_model.py:135-142): |ql| for a Q-only model; with an IQN part the post launch combines
        // dl / 2 + ql / 2 (this kernel runs beside the IQN loss, not behind it)
        if (!a.use_iqn) a.out_td[b] = fabsf(ql);
        if (b == 0) a.out_scalars[4] = theil;
    }
    if (a.q_pieces) {
        // xhat of this sample's embedding as three bf16 pieces (the shared B operand of qh_bwd2_kernel's G product); the
        // row statistics are head 0's (every head normalises the same embedding)
        const float mu = LN ? a.ws.q_mu1[b] : 0.f, rs = LN ? a.ws.q_rstd1[b] : 1.f;
        const float2 ev = *reinterpret_cast<const float2 *>(a.ws.e_cur + (size_t)b * E_DIM + 2 * tid);
        const float x0 = (ev.x - mu) * rs, x1 = (ev.y - mu) * rs;
        const unsigned int hp = pack_bf16(x0, x1);
        const float r0 = x0 - __uint_as_float(hp << 16), r1 = x1 - __uint_as_float(hp & 0xffff0000u);
        const unsigned int mp = pack_bf16(r0, r1);
        const float s0 = r0 - __uint_as_float(mp << 16), s1 = r1 - __uint_as_float(mp & 0xffff0000u);
        unsigned int *xp = reinterpret_cast<unsigned int *>(a.ws.q_xp);
        const size_t plane = (size_t)B * E_DIM / 2, o = (size_t)b * (E_DIM / 2) + tid;
        xp[o] = hp;
        xp[plane + o] = mp;
        xp[2 * plane + o] = pack_bf16(s0, s1);
    }
    // head + LayerNorm(H) backward of row (head, b)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hd = w + 8 * i;
        if (hd < Hd) {
            const int64_t r = (int64_t)hd * B + b;
            const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
            const float *W2 = Ph + a.off.h_w2 + (int64_t)act * H;
            const float *uv = a.ws.q_uv + (size_t)hd * UV_ROWS * H;  // (online set)
            const float dq = s_dq[hd];
            float da[KH], ga[KH], ua[KH], va[KH], m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int h = 64 * k + lane;
                const float w2 = LN ? W2[h] * Ph[a.off.h_ln2_g + h] : W2[h];
                ua[k] = LN ? uv[h] : 0.f;
                va[k] = (LN ? uv[H + h] : 0.f) + Ph[a.off.h_b1 + h];
                da[k] = dq * w2;
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
                a.ws.q_dpre1[r * H + 64 * k + lane] = ga[k];
                if (a.q_pieces) {                 // (a consumer of bf16 pieces of dpre1: split here, once; none at present -- qbwd2_kernels.h)
                    const size_t plane = (size_t)Hd * B * H, o = (size_t)r * H + 64 * k + lane;
                    const unsigned int hp = pack_bf16(ga[k], 0.f);
                    const float r1 = ga[k] - __uint_as_float(hp << 16);
                    const unsigned int mp = pack_bf16(r1, 0.f);
                    const float r2 = r1 - __uint_as_float(mp << 16);
                    a.ws.q_pp[o] = (unsigned short)hp;
                    a.ws.q_pp[plane + o] = (unsigned short)mp;
                    a.ws.q_pp[2 * plane + o] = (unsigned short)pack_bf16(r2, 0.f);
                }
                c1 += ga[k] * ua[k];
                c2 += ga[k] * (pa[i][k] - va[k]);
            }
            if (LN) {
                c1 = wave_sum(c1);
                c2 = wave_sum(c2);
            }
            if (lane == 0) {
                if (LN) {
                    a.ws.q_c1[r] = c1;
                    a.ws.q_c2[r] = c2;
                }
                a.ws.q_dq[r] = dq;
            }
        }
    }
}

template <int H, bool LN>
__global__ __launch_bounds__(512) void qh_loss_kernel(IqnArgs a) { qh_loss_body<H, LN>(a, blockIdx.x); }

// ------------------------------------------------------------------------------------------
// q bwd: grid = (E/16 column slices) x heads, 256 threads = 4 waves, each wave a strided set of the
// B/16 sample tiles.  dX = dpre1 . W1 (columns of the slice), LayerNorm(1024) backward with the saved
// row stats, dW1 / dLN written straight to this head's gradient slab (one slab: every workgroup owns
// its (head, column slice) outright), embedding gradient per head -> de_q[head][b][n] (summed over
// heads where it is consumed, conv_bwd_partial_block).
// ------------------------------------------------------------------------------------------
__host__ __device__ constexpr int qb_acc(int H) { return H / 4 + 2; }             // accumulators folded across the four waves
__host__ __device__ constexpr int qb_stage(int H) { return qb_acc(H) * 64; }       // per-wave staging floats: dpre1 tile (16*(H+4)) aliased with the fold buffer
__host__ __device__ constexpr int qb_lds_floats(int H) { return 4 * qb_stage(H); }

// COLS (small batches: B <= 128): a workgroup takes (head, FOUR column slices), each wave ONE slice over ALL sample tiles, so a
// wave owns its columns outright -- no fold across waves, no barrier, a quarter of the workgroups.  At B = 64 the row-split form
// gives every wave ONE tile: 640 workgroups of 22.9 k cycles each (16.8 k of them the wave's 80 operand loads and their round
// trip for 4 k cycles of matrix work), one resident per CU (its registers), i.e. three rounds: 30.5 us for 0.7 GFLOP (stamps,
// subtractive ablation preset); this form: 25.2 us (22.6 without LayerNorm).  Requesting the next tile's rows before the
// current tile's products (software pipeline through the wave's LDS image) measured no better (27.0 / 22.2 us): not kept.
template <int H, bool LN, bool COLS = false>
__global__ __launch_bounds__(256) void qh_bwd_kernel(IqnArgs a) {
    constexpr int NHT = H / 16, HS = H + 4, QB_ACC = qb_acc(H), QB_STAGE = qb_stage(H);
    static_assert(16 * HS <= QB_STAGE, "dpre1 tile must fit the per-wave staging area");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int j = lane & 15, g = lane >> 4;
    const int cs = COLS ? 4 * (blockIdx.x % (E_DIM / 64)) + w : blockIdx.x % (E_DIM / 16);
    const int hd = COLS ? blockIdx.x / (E_DIM / 64) : blockIdx.x / (E_DIM / 16);
    const int n = cs * 16 + j;
    const int B = a.B;
    const int tiles_total = B / 16;
    float *dpl = smem + w * QB_STAGE;
    float *red = smem;                            // [4 waves][QB_ACC][64], reuses the staging area after the barrier
    const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
    float w1f[4 * NHT];    // B operand of dX: W1_h[hh = 16q + 4g + jj][n]
    {
        const float *src = Ph + a.off.h_w1 + n;
#pragma unroll
        for (int q = 0; q < NHT; ++q)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) w1f[q * 4 + jj] = src[(int64_t)(16 * q + 4 * g + jj) * E_DIM];
    }
    const float g1 = LN ? Ph[a.off.h_ln1_g + n] : 1.f, be1 = LN ? Ph[a.off.h_ln1_b + n] : 0.f;
    f32x4 accW1[NHT];
#pragma unroll
    for (int i = 0; i < NHT; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float s_dg = 0.f, s_db = 0.f;
    float *de_h = a.ws.de_q + (size_t)hd * B * E_DIM;
    for (int t = COLS ? 0 : w; t < tiles_total; t += COLS ? 1 : 4) {
        const int b0 = t * 16;
        const int64_t row0 = (int64_t)hd * B + b0;
        float4 ad[NHT];
        const float *sd = a.ws.q_dpre1 + (row0 + j) * H + 4 * g;
#pragma unroll
        for (int q = 0; q < NHT; ++q) ad[q] = *reinterpret_cast<const float4 *>(sd + 16 * q);
#pragma unroll
        for (int q = 0; q < NHT; ++q) *reinterpret_cast<float4 *>(&dpl[j * HS + 16 * q + 4 * g]) = ad[q];
        const int64_t rb = row0 + 4 * g;
        float4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {1.f, 1.f, 1.f, 1.f}, c1 = mu, c2 = mu;
        if (LN) {
            mu = *reinterpret_cast<const float4 *>(a.ws.q_mu1 + rb);
            rs = *reinterpret_cast<const float4 *>(a.ws.q_rstd1 + rb);
            c1 = *reinterpret_cast<const float4 *>(a.ws.q_c1 + rb);
            c2 = *reinterpret_cast<const float4 *>(a.ws.q_c2 + rb);
        }
        float ev[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ev[r] = a.ws.e_cur[(int64_t)(b0 + 4 * g + r) * E_DIM + n];
        f32x4 adx = {0.f, 0.f, 0.f, 0.f}, adx2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < NHT / 2; ++q) {
            adx = mfma16(ad[2 * q].x, w1f[8 * q + 0], adx);
            adx2 = mfma16(ad[2 * q + 1].x, w1f[8 * q + 4], adx2);
            adx = mfma16(ad[2 * q].y, w1f[8 * q + 1], adx);
            adx2 = mfma16(ad[2 * q + 1].y, w1f[8 * q + 5], adx2);
            adx = mfma16(ad[2 * q].z, w1f[8 * q + 2], adx);
            adx2 = mfma16(ad[2 * q + 1].z, w1f[8 * q + 6], adx2);
            adx = mfma16(ad[2 * q].w, w1f[8 * q + 3], adx);
            adx2 = mfma16(ad[2 * q + 1].w, w1f[8 * q + 7], adx2);
        }
        const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w};
        const float c1v[4] = {c1.x, c1.y, c1.z, c1.w}, c2v[4] = {c2.x, c2.y, c2.z, c2.w};
        float xv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xhat = LN ? (ev[r] - muv[r]) * rsv[r] : ev[r];
            xv[r] = LN ? xhat * g1 + be1 : ev[r];
            const float dX = adx[r] + adx2[r];
            s_dg += dX * xhat;
            s_db += dX;
            const float dh0 = LN ? rsv[r] * (dX * g1 - c1v[r] * (1.0f / E_DIM) - xhat * (c2v[r] * (1.0f / E_DIM))) : dX;
            de_h[(int64_t)(b0 + 4 * g + r) * E_DIM + n] = dh0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mt = 0; mt < NHT; ++mt)
                accW1[mt] = mfma16(dpl[(4 * g + r) * HS + 16 * mt + j], xv[r], accW1[mt]);
        }
    }
    float *slab = a.ws.q_slabs + (int64_t)hd * a.q_slab;
    constexpr int W1_OFF = LN ? 2 * E_DIM : 0;
    if constexpr (COLS) {
        // the wave's columns over all rows: straight to the slab
#pragma unroll
        for (int mt = 0; mt < NHT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[W1_OFF + (int64_t)(16 * mt + 4 * g + r) * E_DIM + n] = accW1[mt][r];
        s_dg += __shfl_xor(s_dg, 16, 64);
        s_dg += __shfl_xor(s_dg, 32, 64);
        s_db += __shfl_xor(s_db, 16, 64);
        s_db += __shfl_xor(s_db, 32, 64);
        if (LN && g == 0) {
            slab[n] = s_dg;
            slab[E_DIM + n] = s_db;
        }
        return;
    }
    // fold the four waves in fixed order and write this head's slab slice
    __syncthreads();
    {
        float *mine = red + (w * QB_ACC) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < NHT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(mt * 4 + r) * 64] = accW1[mt][r];
        s_dg += __shfl_xor(s_dg, 16, 64);
        s_dg += __shfl_xor(s_dg, 32, 64);
        s_db += __shfl_xor(s_db, 16, 64);
        s_db += __shfl_xor(s_db, 32, 64);
        mine[(4 * NHT) * 64] = s_dg;
        mine[(4 * NHT + 1) * 64] = s_db;
    }
    __syncthreads();
    for (int idx = tid; idx < QB_ACC * 64; idx += 256) {
        const int slot = idx >> 6, l = idx & 63;
        const float v = ((red[(0 * QB_ACC + slot) * 64 + l] + red[(1 * QB_ACC + slot) * 64 + l]) +
                         red[(2 * QB_ACC + slot) * 64 + l]) + red[(3 * QB_ACC + slot) * 64 + l];
        const int lj = l & 15, lg = l >> 4;
        if (slot < 4 * NHT) {
            const int mt = slot >> 2, r = slot & 3;        // row hh = 16*mt + 4*lg + r, col n = cs*16 + lj
            slab[W1_OFF + (int64_t)(16 * mt + 4 * lg + r) * E_DIM + cs * 16 + lj] = v;
        } else if (LN && lg == 0) {
            if (slot == 4 * NHT) slab[cs * 16 + lj] = v;       // d ln1_g
            else slab[E_DIM + cs * 16 + lj] = v;               // d ln1_b
        }
    }
}

// ---- post-kernel roles ---------------------------------------------------------------------------
// slab sum of the Q heads: float4 index i over [heads][Q_SLAB/4]; adds the Theil term kappa_h * theta
__device__ __forceinline__ void q_slab_sum(const IqnArgs &a, int64_t i, const float *kappa, float &sq) {
    const int per_head = a.q_slab / 4;
    const int hd = (int)(i / per_head);
    const int64_t o = (i - (int64_t)hd * per_head) * 4;
    float4 s = *reinterpret_cast<const float4 *>(a.ws.q_slabs + (int64_t)hd * a.q_slab + o);
    // parameter order inside a head: [ln1_g | ln1_b |] w1 (contiguous), the slab has the same order
    const int64_t po = a.off.head_base + (int64_t)hd * a.off.head_stride + (a.ln ? a.off.h_ln1_g : a.off.h_w1) + o;
    if (kappa) {
        const float4 th = ld4u(a.params + po);
        const float k = kappa[hd];
        s.x += k * th.x; s.y += k * th.y; s.z += k * th.z; s.w += k * th.w;
    }
    float *gp = a.grads + po;
    gp[0] = s.x; gp[1] = s.y; gp[2] = s.z; gp[3] = s.w;
    sq += (s.x * s.x + s.y * s.y) + (s.z * s.z + s.w * s.w);
}

// b1, LN2 affine, W2, b2 gradients of head `hd` for the 16 hidden units [slice*16, +16).  1024 threads.
__device__ __forceinline__ void q_small_tensor_block(const IqnArgs &a, int hd, int slice, const float *kappa, float &sq,
                                                     float *pool) {
    const float *Ph = a.params + a.off.head_base + (int64_t)hd * a.off.head_stride;
    float *Gh = a.grads + a.off.head_base + (int64_t)hd * a.off.head_stride;
    const bool ln = a.ln != 0;
    const int H = a.Hq;
    SmallIo io{Ph + a.off.h_w2, ln ? Ph + a.off.h_ln2_g : nullptr, ln ? Ph + a.off.h_ln2_b : nullptr, Ph + a.off.h_b1, Ph + a.off.h_b2,
               Gh + a.off.h_w2, ln ? Gh + a.off.h_ln2_g : nullptr, ln ? Gh + a.off.h_ln2_b : nullptr, Gh + a.off.h_b1, Gh + a.off.h_b2,
               kappa ? kappa[hd] : 0.f, kappa != nullptr, nullptr, H};
    const int64_t r0 = (int64_t)hd * a.B;
    small_fold_block(a, slice, sq, pool, io,
                     [&](int b, int h, bool) {
                         const float dq = a.ws.q_dq[r0 + b];
                         return make_float3(dq * a.ws.q_xhat2[(r0 + b) * H + h], a.ws.q_dpre1[(r0 + b) * H + h], dq);
                     },
                     nullptr);
}

}  // namespace prism

// ==========================================================================================
// One-layer DQN head: Q = [LayerNorm(1024)](e) . W^T + b with W [A][1024]
// (q_ensemble.py:25-38 with n_model_layers = 1 through FFNNModel, n_heads = 1; model_factory.py:130-145).
// Everything is tiny (A dot products of length 1024 per observation), so there is no MFMA here: one
// workgroup per sample for loss + input gradient, and a reduction role in the post kernel for dW.
// ==========================================================================================
namespace prism {

// 1024-float dot products of one observation embedding against the A rows of W: 256 threads.
// x in LDS (already normalised if LN), W in global; results -> out[A] (LDS), all threads return after a barrier.
__device__ __forceinline__ void dqn_q_values(const float *s_x, const float *W, const float *bias, int A, float *s_out,
                                             float *s_red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int aa = 0; aa < A; ++aa) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = tid + 256 * i;
            s += s_x[n] * W[aa * E_DIM + n];
        }
        s = wave_sum(s);
        if (lane == 0) s_red[aa * 4 + w] = s;
    }
    __syncthreads();
    if (tid < A) s_out[tid] = ((s_red[tid * 4] + s_red[tid * 4 + 1]) + (s_red[tid * 4 + 2] + s_red[tid * 4 + 3])) + bias[tid];
    __syncthreads();
}

// LayerNorm(1024) of a row held in LDS (in place -> y = xhat*g + beta); returns mean / rstd; keeps xhat in s_xhat
__device__ __forceinline__ void dqn_layernorm(float *s_x, float *s_xhat, const float *g, const float *be, float &mean,
                                              float &rstd, float *s_red) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float x[4], s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] = s_x[tid + 256 * i];
        s += x[i];
    }
    s = wave_sum(s);
    if (lane == 0) s_red[w] = s;
    __syncthreads();
    mean = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * (1.0f / E_DIM);
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        x[i] -= mean;
        v += x[i] * x[i];
    }
    v = wave_sum(v);
    if (lane == 0) s_red[w] = v;
    __syncthreads();
    rstd = 1.0f / sqrtf(((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * (1.0f / E_DIM) + LN_EPS);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = tid + 256 * i;
        const float xh = x[i] * rstd;
        s_xhat[n] = xh;
        s_x[n] = xh * g[n] + be[n];
    }
    __syncthreads();
}

// conv-backward partial row of ONE sample (the block that just produced d e of that sample has everything
// at hand): out[c*9C + ci*9 + dy*3 + dx] = sum_{y,x} dc[c][y][x] * obs[y+dy][x+dx][ci], out[16*9C + c] = sum dc[c].
// s_dc: ReLU-masked d e [1024] (channel-major), s_ob: the observation [100*C]; 256 threads.
__device__ __forceinline__ void conv_bwd_sample_row(const float *s_dc, const float *s_ob, int C, float *__restrict__ out) {
    const int tid = threadIdx.x, nk = 9 * C;
    // work item = (out channel c, in channel ci, kernel row dy): its three dx outputs share every operand
    // (8 gradient + 10 observation values of an image row feed 24 MACs in three independent chains)
    for (int item = tid; item < 16 * C * 3; item += 256) {
        const int c = item / (3 * C), r = item - c * 3 * C, ci = r / 3, dy = r - ci * 3;
        const float *dc = s_dc + c * 64, *ob = s_ob + dy * 10 * C + ci;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int y = 0; y < 8; ++y) {
            const float4 d0 = *reinterpret_cast<const float4 *>(dc + y * 8), d1 = *reinterpret_cast<const float4 *>(dc + y * 8 + 4);
            const float d[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            float row[10];
#pragma unroll
            for (int x = 0; x < 10; ++x) row[x] = ob[(y * 10 + x) * C];
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                a0 = fmaf(d[x], row[x], a0);
                a1 = fmaf(d[x], row[x + 1], a1);
                a2 = fmaf(d[x], row[x + 2], a2);
            }
        }
        float *o = out + c * nk + ci * 9 + dy * 3;
        o[0] = a0;
        o[1] = a1;
        o[2] = a2;
    }
    const int w = tid >> 6, lane = tid & 63;
    for (int c = w; c < 16; c += 4) {
        const float s = wave_sum(s_dc[c * 64 + lane]);
        if (lane == 0) out[16 * nk + c] = s;
    }
}

__global__ __launch_bounds__(256) void dqn_loss_kernel(IqnArgs a) {
    __shared__ __attribute__((aligned(16))) float s_x[E_DIM], s_xhat[E_DIM], s_nx[E_DIM], s_tmp[E_DIM];
    __shared__ float s_ob[1000];
    __shared__ float s_q[16], s_qo[16], s_qt[16], s_red[64];
    __shared__ float s_sc[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int B = a.B, A = a.A;
    const bool ln = a.off.h_ln1_g >= 0;
    const float *P = a.params + a.off.head_base, *Pt = (a.has_target ? a.target_params : a.params) + a.off.head_base;
    float ec[4];                                         // conv output of this sample (ReLU mask of the tail)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ec[i] = a.ws.e_cur[(int64_t)b * E_DIM + tid + 256 * i];
        s_x[tid + 256 * i] = ec[i];
        s_nx[tid + 256 * i] = a.ws.e_next[(int64_t)b * E_DIM + tid + 256 * i];
    }
    for (int i = tid; i < 100 * a.C; i += 256) s_ob[i] = a.obs[(int64_t)b * 100 * a.C + i];   // for the conv-backward tail
    __syncthreads();
    float mean = 0.f, rstd = 1.f;
    if (ln) dqn_layernorm(s_x, s_xhat, P + a.off.h_ln1_g, P + a.off.h_ln1_b, mean, rstd, s_red);
    dqn_q_values(s_x, P + a.off.h_w1, P + a.off.h_b1, A, s_q, s_red);
    // next-state values: online (no target, or double-Q) and/or target (q_ensemble.py:62-68)
    const bool need_on = !a.has_target || a.double_q;
    if (need_on) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_tmp[tid + 256 * i] = s_nx[tid + 256 * i];
        __syncthreads();
        float m2, r2;
        if (ln) dqn_layernorm(s_tmp, s_xhat + 0, P + a.off.h_ln1_g, P + a.off.h_ln1_b, m2, r2, s_red);   // xhat of cur is re-made below
        dqn_q_values(s_tmp, P + a.off.h_w1, P + a.off.h_b1, A, s_qo, s_red);
    }
    if (a.has_target) {
#pragma unroll
        for (int i = 0; i < 4; ++i) s_tmp[tid + 256 * i] = s_nx[tid + 256 * i];
        __syncthreads();
        float m2, r2;
        if (ln) dqn_layernorm(s_tmp, s_xhat + 0, Pt + a.off.h_ln1_g, Pt + a.off.h_ln1_b, m2, r2, s_red);
        dqn_q_values(s_tmp, Pt + a.off.h_w1, Pt + a.off.h_b1, A, s_qt, s_red);
    }
    if (ln && (need_on || a.has_target)) {   // restore xhat of the CURRENT observation (clobbered above)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = tid + 256 * i;
            s_xhat[n] = (a.ws.e_cur[(int64_t)b * E_DIM + n] - mean) * rstd;
        }
        __syncthreads();
    }
    const int act = (int)a.action[b];
    if (tid == 0) {
        const float *qon = need_on ? s_qo : s_qt, *qtg = a.has_target ? s_qt : s_qo;
        int best = 0;
        float bv = qon[0];
        for (int aa = 1; aa < A; ++aa)
            if (qon[aa] > bv) {
                bv = qon[aa];
                best = aa;
            }
        const float dg = a.gamma[b] * (a.nonterminal[b] ? 1.0f : 0.0f);
        const float y = td_target(a.squish, a.reward[b], qtg[best], dg);
        const float diff = s_q[act] - y;
        const float wb = a.per_weights ? a.per_weights[b] : 1.0f;
        const float ql = a.q_w * (diff * diff);
        a.out_ql[b] = ql;
        a.ws.q_lossw[b] = ql * wb;
        a.out_td[b] = fabsf(ql);                       // composite_model.py:141-142
        const float dq = (wb / (float)B) * a.q_w * 2.0f * diff;
        a.ws.q_dq[b] = dq;
        s_sc[0] = dq;
        if (ln) {
            a.ws.q_mu1[b] = mean;
            a.ws.q_rstd1[b] = rstd;
        }
        if (b == 0) a.out_scalars[4] = 0.f;
    }
    __syncthreads();
    // input gradient: dy = dq * W[act]; through LayerNorm if present; masked by the conv ReLU later
    const float dq = s_sc[0];
    const float *Wa = P + a.off.h_w1 + (int64_t)act * E_DIM;
    float dy[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) dy[i] = dq * Wa[tid + 256 * i];
    if (ln) {
        const float *g = P + a.off.h_ln1_g;
        float s1 = 0.f, s2 = 0.f, dxh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = tid + 256 * i;
            dxh[i] = dy[i] * g[n];
            s1 += dxh[i];
            s2 += dxh[i] * s_xhat[n];
        }
        s1 = wave_sum(s1);
        s2 = wave_sum(s2);
        if (lane == 0) {
            s_red[w] = s1;
            s_red[4 + w] = s2;
        }
        __syncthreads();
        const float m1 = ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3])) * (1.0f / E_DIM);
        const float m2 = ((s_red[4] + s_red[5]) + (s_red[6] + s_red[7])) * (1.0f / E_DIM);
#pragma unroll
        for (int i = 0; i < 4; ++i) dy[i] = rstd * (dxh[i] - m1 - s_xhat[tid + 256 * i] * m2);
    }
    // d e of this sample -> workspace; its ReLU-masked copy feeds the conv-backward partial row right here
    // (one row per sample, folded by the post kernel: no partial/ticket/fold chain there)
    __syncthreads();                                     // s_tmp (next-state scratch) is free
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = tid + 256 * i;
        a.ws.de_q[(int64_t)b * E_DIM + n] = dy[i];
        s_tmp[n] = ec[i] > 0.f ? dy[i] : 0.f;
    }
    __syncthreads();
    conv_bwd_sample_row(s_tmp, s_ob, a.C, a.ws.convpart + (int64_t)b * CONV_ROW);
}

// acting forward of the one-layer head (agent.py:31-41 -> q_ensemble.py:44-48): Q values of observation b from its
// embedding (left in ws.e_cur by the embed launch), with the routines the loss kernel uses.  out_q: [1][n_pad][A].
__global__ __launch_bounds__(256) void dqn1_act_kernel(IqnArgs a, float *__restrict__ out_q) {
    __shared__ __attribute__((aligned(16))) float s_x[E_DIM], s_xhat[E_DIM];
    __shared__ float s_q[16], s_red[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool ln = a.off.h_ln1_g >= 0;
    const float *P = a.params + a.off.head_base;
#pragma unroll
    for (int i = 0; i < 4; ++i) s_x[tid + 256 * i] = a.ws.e_cur[(int64_t)b * E_DIM + tid + 256 * i];
    __syncthreads();
    float mean = 0.f, rstd = 1.f;
    if (ln) dqn_layernorm(s_x, s_xhat, P + a.off.h_ln1_g, P + a.off.h_ln1_b, mean, rstd, s_red);
    dqn_q_values(s_x, P + a.off.h_w1, P + a.off.h_b1, a.A, s_q, s_red);
    if (tid < a.A) out_q[(int64_t)b * a.A + tid] = s_q[tid];
}

// post role: dW[a][n] = sum_{b: act=a} dq_b * y_b[n], db[a], and with LayerNorm dg[n] = sum_b dy_b[n] xhat_b[n],
// dbeta[n] = sum_b dy_b[n] (dy_b = dq_b W[act_b]).  With y = g * xhat + beta this is the algebra of
// small_fold_block with the 1024 embedding columns as "units":
//   S[a][n] = sum_{b: act=a} dq_b xhat_b[n],  D[a] = sum_{b: act=a} dq_b,
//   dW = g S + beta D,  dg = sum_a W S,  dbeta = sum_a W D,  db = D        (no LayerNorm: xhat := e, g = 1, beta = 0)
// 16 columns per workgroup -> E/16 workgroups; slice 0 also writes db and the total loss.
constexpr int DQN_GRAD_BLOCKS = E_DIM / SMALL_W;

__device__ __forceinline__ void dqn_grad_block(const IqnArgs &a, int slice, float &sq, float *pool) {
    const bool ln = a.off.h_ln1_g >= 0;
    const float *P = a.params + a.off.head_base;
    float *G = a.grads + a.off.head_base;
    SmallIo io{P + a.off.h_w1, ln ? P + a.off.h_ln1_g : nullptr, ln ? P + a.off.h_ln1_b : nullptr, nullptr, P + a.off.h_b1,
               G + a.off.h_w1, ln ? G + a.off.h_ln1_g : nullptr, ln ? G + a.off.h_ln1_b : nullptr, nullptr, G + a.off.h_b1,
               0.f, false, a.ws.q_lossw, E_DIM};
    float lsum = 0.f;
    small_fold_block(a, slice, sq, pool, io,
                     [&](int b, int n, bool) {
                         const float dq = a.ws.q_dq[b];
                         float x = a.ws.e_cur[(int64_t)b * E_DIM + n];
                         if (ln) x = (x - a.ws.q_mu1[b]) * a.ws.q_rstd1[b];
                         return make_float3(dq * x, 0.f, dq);
                     },
                     &lsum);
    if (slice == 0 && threadIdx.x == 0) {
        const float l = lsum / (float)a.B;          // total loss = mean(ql * w)
        a.out_scalars[0] = l;
        a.out_scalars[1] = 0.f;
        a.out_scalars[2] = l;
    }
}

}  // namespace prism
