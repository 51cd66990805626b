// TD update entry points: get_losses + backward (prism_learner_fwd_bwd), clip_grad_norm_ + Adam
// (prism_learner_clip_adam), and the fused step front/back (prism_step_front / prism_step_back).
// Reference: /root/reference/prism/agents/models/composite_model.py:94-144,
// prism/agents/agent.py:53-79, prism/learner.py:95-125.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "step_kernels.h"
#include "fwd_kernels.h"
#include "act_kernels.h"
#include "qbwd2_kernels.h"
#include "bwd3_kernels.h"
#include "bwd4_kernels.h"

namespace prism {

// sum of squares of the (scaled) gradient, one partial per block — data-parallel path, where the
// partials written by the backward kernels predate the all-reduce
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float *__restrict__ g, int64_t n, float scale,
                                                        float *__restrict__ normpart) {
    __shared__ float s_red[256];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = g[i] * scale;
        s += x * x;
    }
    s_red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_red[threadIdx.x] += s_red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) normpart[blockIdx.x] = s_red[0];
}

__global__ __launch_bounds__(256) void clip_adam_kernel(AdamArgs a) { clip_adam_block<256>(a, blockIdx.x, gridDim.x); }

// back: block 0 = priority writeback (+ RNG counters); blocks [1, gridDim) = clip + Adam.
__global__ __launch_bounds__(256) void step_back_kernel(AdamArgs a, prism_replay_desc rp, BackArgs k) {
    kernarg_prefetch<sizeof(AdamArgs) + sizeof(prism_replay_desc) + sizeof(BackArgs)>();
    if (blockIdx.x == 0) {
        __shared__ __attribute__((aligned(16))) char s_pool[PER_UPDATE_LDS_BYTES];
        if (k.plan) per_update_finish(rp, k.plan, k.n, s_pool, k.sib, k.n, k.sib_state);
        else if (k.use_per) per_update_block(rp, k.index, k.priority, k.n, k.alpha, k.eps, k.take_abs, s_pool);
        if (k.rng && threadIdx.x == 0) {
            k.rng[0] += k.inc_per;
            k.rng[1] += k.inc_tau;
        }
        return;
    }
    clip_adam_block<256>(a, blockIdx.x - 1, gridDim.x - 1);
}

// Models with both parts (IDS: IQN + Q ensemble): the two per-sample losses in ONE launch -- blocks [0, B) the quantile
// loss, [B, 2 B) the ensemble loss (both 512-thread routines; neither reads what the other writes: td = dl / 2 + ql / 2 is
// combined by the post launch).  Two latency-bound launches and a boundary become one.
template <int H, bool LN>
__global__ __launch_bounds__(512) void loss_both_kernel(IqnArgs a) {
    if ((int)blockIdx.x < a.B) iqn_loss_body<H, LN, LOSS_WAVES>(a, blockIdx.x);
    else qh_loss_body<H, LN>(a, (int)blockIdx.x - a.B);
}

// ---- workspace carving -----------------------------------------------------------------------
struct Carver {
    char *base;
    size_t off;
    explicit Carver(void *b) : base((char *)b), off(0) {}
    float *f(size_t n) {
        float *p = base ? (float *)(base + off) : nullptr;
        off += ((n + 3) / 4) * 16;   // n floats rounded up to 16 bytes
        return p;
    }
};

// backward decomposition: one workgroup per CU with double-buffered tiles (4 row chunks x 64 column slices = 256
// workgroups), or two workgroups per CU sharing each SIMD (8 row chunks; width 128 only).  Measured on MI355X
// (configs[2]): 33.4 us vs 30.3 us -- one wave per SIMD cannot keep the matrix pipe fed through its own VALU and
// hazard bubbles, the shared form is the default; PRISM_BWD_MODE=1 picks the first form for A/B runs.
static constexpr int MAX_CHUNKS = 16;
static int bwd_mode() {
    static const int mode = [] { const char *e = getenv("PRISM_BWD_MODE"); return e ? atoi(e) : 0; }();
    return mode;
}
static bool bwd_double_buffered(int H) { return H == 128 && bwd_mode() != 0; }
static int bwd_chunks(int H) { return (H == 128 && bwd_mode() == 0) ? 8 : 4; }      // workgroups per CU x 4


static bool width_ok(int h) { return h == 128 || h == 256; }

// Forward GEMMs: 1 = exact fp32 chain (v_mfma_f32_16x16x4_f32), 2 = three-piece bf16 operands on v_mfma_f32_16x16x32_bf16
// (fp32 accuracy, common.h).  prism_learner_desc.gemm_mode picks one, 0 = the library default (PRISM_GEMM=fp32|bf16x3
// overrides it).
static int default_gemm_mode() {
    static const int mode = [] {
        const char *e = getenv("PRISM_GEMM");
        if (e && !strcmp(e, "fp32")) return 1;
        if (e && !strcmp(e, "bf16x3")) return 2;
        return PRISM_GEMM_DEFAULT;
    }();
    return mode;
}
static int use_split(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    const int mode = ld->gemm_mode == PRISM_GEMM_FP32 || ld->gemm_mode == PRISM_GEMM_BF16X3 ? ld->gemm_mode : default_gemm_mode();
    if (mode != PRISM_GEMM_BF16X3) return 0;
    return (d.use_iqn || (d.n_heads && d.head_layers == 2)) ? 1 : 0;
}

// the 64-column bf16 backward (bwd3_kernels.h) where it applies: bf16 mode, width 128 (the forward then saves ReLU(phi))
static bool use_bw3(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    static const bool off = [] { const char *e = getenv("PRISM_NO_BWD3"); return e && atoi(e) != 0; }();
    return !off && d.use_iqn && use_split(ld) && bw3_ok(d.iqn_width, ld->batch, d.n_tau, true);
}

// ... and its width-256 form (bwd4_kernels.h: pairs of waves share 16 columns, one hidden half each)
static bool use_bw4(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    static const bool off = [] { const char *e = getenv("PRISM_NO_BWD4"); return e && atoi(e) != 0; }();
    return !off && d.use_iqn && use_split(ld) && bw4_ok(d.iqn_width, ld->batch, d.n_tau);
}
static int iqn_supported(const prism_model_dims *d, int32_t B) {
    auto pow2_ok = [](int t) { return t == 4 || t == 8 || t == 16 || t == 32 || t == 64; };
    if (!d->use_iqn && d->n_heads == 0) return PRISM_ERR_UNSUPPORTED;
    if (d->embed_dim != E_DIM) return PRISM_ERR_UNSUPPORTED;
    if (d->squish_fn < PRISM_SQUISH_NONE || d->squish_fn > PRISM_SQUISH_SYMLOG) return PRISM_ERR_UNSUPPORTED;
    if (d->use_iqn) {
        if (d->n_basis != K_BASIS || d->iqn_layers != 1 || !width_ok(d->iqn_width)) return PRISM_ERR_UNSUPPORTED;
        if (!pow2_ok(d->n_tau) || !pow2_ok(d->n_tau_next)) return PRISM_ERR_UNSUPPORTED;
        if ((B * d->n_tau) % 16 || (B * d->n_tau_next) % 16) return PRISM_ERR_UNSUPPORTED;
    }
    if (d->n_heads != 0) {
        // ensemble / DQN heads of the form [LN] -> Linear(1024,H) -> ReLU -> [LN] -> Linear(H,A)
        if (d->head_layers == 1) {
            // single Linear(1024 -> A) DQN head, with or without LayerNorm, no IQN beside it
            if (d->n_heads != 1 || d->use_iqn) return PRISM_ERR_UNSUPPORTED;
        } else {
            if (d->n_heads < 0 || d->n_heads > Q_MAX_HEADS || d->head_layers != 2 || !width_ok(d->head_width))
                return PRISM_ERR_UNSUPPORTED;
            if (B % 16) return PRISM_ERR_UNSUPPORTED;
        }
    }
    if (B < 1 || B > SMALL_MAX_B) return PRISM_ERR_UNSUPPORTED;
    if (d->n_actions < 1 || d->n_actions > 16 || d->in_channels < 1 || d->in_channels > 10) return PRISM_ERR_UNSUPPORTED;
    return PRISM_OK;
}

static int iqn_width(const prism_model_dims &d) { return d.use_iqn ? d.iqn_width : 128; }
static int head_width(const prism_model_dims &d) { return d.n_heads && d.head_layers == 2 ? d.head_width : 128; }

// the IQN loss finishes inside the forward tiles (kind 2) when current- and next-state rows of a sample run
// through the SAME weights (no target network) and a 16-row tile holds whole samples (2 T <= 16)
static int local_loss(const prism_model_dims &d) {
    return d.use_iqn && !d.has_target && d.n_tau_next == d.n_tau && d.n_tau <= 8 ? 1 : 0;
}

static size_t carve_iqn(const prism_model_dims *d, int B, void *base, IqnWs *ws, float **tau_buf, float **dl_buf) {
    Carver c(base);
    const size_t R = (size_t)B * d->n_tau, Rn = (size_t)B * d->n_tau_next, A = d->n_actions;
    const size_t maxT = d->n_tau > d->n_tau_next ? d->n_tau : d->n_tau_next;
    const size_t Hi = iqn_width(*d), Hq = head_width(*d);
    const int ln = d->use_layer_norm;
    IqnWs w;
    w.ticket = (unsigned int *)c.f(8);     // first 32 bytes: the self-resetting tickets (zeroed once by the caller)
    w.e_cur = c.f((size_t)B * E_DIM);
    w.e_next = c.f((size_t)B * E_DIM);
    w.uv = c.f(2 * UV_ROWS * Hi);
    w.wpk[0] = c.f(iqn_pack_split_floats((int)Hi));      // (the larger of the two layouts: fp32 stream order / bf16 pieces)
    w.wpk[1] = c.f(iqn_pack_split_floats((int)Hi));
    w.cosb = c.f(R * K_BASIS);
    w.cospk = (unsigned int *)c.f((size_t)((R + 2 * Rn + 15) / 16 + 3) * CP_TILE);
    w.phis = c.f(((R + 15) / 16) * 16 * (size_t)E_DIM);
    w.mu1 = c.f(R);
    w.rstd1 = c.f(R);
    w.pre1 = c.f(R * Hi);
    w.xhat2 = c.f(R * Hi);
    w.rstd2 = c.f(R);
    w.zcur = c.f(R * A);
    w.zon = c.f(Rn * A);
    w.ztg = c.f(Rn * A);
    w.dq = c.f(R);
    w.c1 = c.f(R);
    w.c2 = c.f(R);
    w.dpre1 = c.f(R * Hi);
    w.Sb = c.f((size_t)B * Hi);
    w.Pb = c.f((size_t)B * Hi);
    w.Db = c.f(B);
    w.lossw = c.f(B);
    w.de_iqn = c.f((size_t)B * E_DIM);
    w.slabs = c.f((size_t)MAX_CHUNKS * iqn_slab_floats((int)Hi, ln));
    {
        const size_t post_rows = (size_t)((B + 3) / 4) * CONV_ROW;      // (post_conv_blocks(B, C) <= this)
        const size_t bwd_rows = (size_t)(E_DIM / 16) * MAX_CHUNKS * BWD_CONV_ROW;
        const size_t dqn_rows = d->head_layers == 1 && d->n_heads ? (size_t)B * CONV_ROW : 0;   // one row per sample
        const size_t m = post_rows > bwd_rows ? post_rows : bwd_rows;
        w.convpart = c.f(m > dqn_rows ? m : dqn_rows);
    }
    w.normpart = c.f(NORM_SLOTS);
    w.sib = c.f((size_t)TREE_MAX_LEVELS * B * 2);
    w.wb_plan = c.f((size_t)B * 4);
    {
        const size_t Hd = d->n_heads, RQ = Hd * (size_t)B;
        w.q_mu1 = c.f(RQ);
        w.q_rstd1 = c.f(RQ);
        w.q_pre1 = c.f(RQ * Hq);
        w.q_xhat2 = c.f(RQ * Hq);
        w.q_rstd2 = c.f(RQ);
        w.zq_cur = c.f(RQ * A);
        w.zq_on = c.f(RQ * A);
        w.zq_tg = c.f(RQ * A);
        w.q_dq = c.f(RQ);
        w.q_c1 = c.f(RQ);
        w.q_c2 = c.f(RQ);
        w.q_dpre1 = c.f(RQ * Hq);
        w.q_pp = (unsigned short *)c.f(RQ * Hq * 3 / 2);
        w.q_xp = (unsigned short *)c.f((size_t)B * E_DIM * 3 / 2);
        w.q_lossw = c.f(B);
        w.q_uv = c.f(2 * Hd * UV_ROWS * Hq);
        w.q_kappa = c.f(Q_MAX_HEADS * Q_NORM_PARTS);
        w.q_wpk[0] = c.f(Hd * (size_t)q_pack_split_floats((int)Hq));
        w.q_wpk[1] = c.f(d->has_target ? Hd * (size_t)q_pack_split_floats((int)Hq) : 0);
        w.de_q = c.f(Hd * (size_t)B * E_DIM);      // (one slot for the single-Linear DQN head)
        w.q_slabs = c.f(Hd * (size_t)q_slab_floats((int)Hq, ln));
    }
    float *tb = c.f(3 * maxT * B);
    float *db = c.f(2 * (size_t)B);       // per-sample dl | ql when the caller binds no output arrays (the post launch reads both)
    if (ws) *ws = w;
    if (tau_buf) *tau_buf = tb;
    if (dl_buf) *dl_buf = db;
    return c.off;
}


}  // namespace prism

using namespace prism;

extern "C" int prism_learner_supported(const prism_model_dims *dims, int32_t batch) {
    if (!dims) return PRISM_ERR_INVALID;
    return iqn_supported(dims, batch);
}

extern "C" size_t prism_learner_workspace_bytes(const prism_model_dims *dims, int32_t batch) {
    if (!dims || iqn_supported(dims, batch) != PRISM_OK) return 0;
    return carve_iqn(dims, batch, nullptr, nullptr, nullptr, nullptr);
}

// `need_batch`: the minibatch arrays and outputs of an update must be bound (not for the acting forward, which may well
// run before the first update)
static int check_learner(const prism_learner_desc *ld, bool need_batch = true) {
    PRISM_CHECK_ARG(ld != nullptr, "null descriptor");
    if (iqn_supported(&ld->dims, ld->batch) != PRISM_OK) {
        set_error("prism_learner: model dims / batch not covered by the HIP kernels "
                  "(need E=1024; IQN: K=64, H in {128,256}, one trunk layer, T in {4,8,16,32,64}, B*T %% 16 == 0; "
                  "Q heads: one layer, or two layers of width 128/256 with B %% 16 == 0)");
        return PRISM_ERR_UNSUPPORTED;
    }
    PRISM_CHECK_ARG(ld->params && ld->grads && ld->adam_m && ld->adam_v && ld->adam_step, "null parameter buffers");
    PRISM_CHECK_ARG(!ld->dims.has_target || ld->target_params, "has_target without target_params");
    PRISM_CHECK_ARG(ld->workspace && ld->workspace_bytes >= prism_learner_workspace_bytes(&ld->dims, ld->batch),
                    "workspace too small");
    PRISM_CHECK_ARG(((uintptr_t)ld->workspace & 15) == 0, "workspace must be 16-byte aligned");
    PRISM_CHECK_ARG((((uintptr_t)ld->params | (uintptr_t)ld->grads | (uintptr_t)ld->adam_m | (uintptr_t)ld->adam_v) & 15) == 0,
                    "parameter / gradient / Adam buffers must be 16-byte aligned");
    PRISM_CHECK_ARG(ld->off.n_params > 0 && (!ld->dims.use_iqn || (ld->off.phi_w & 3) == 0),
                    "n_params / phi_w offset alignment");
    PRISM_CHECK_ARG(ld->dims.n_heads == 0 || (ld->off.head_base >= 0 && ld->off.h_w1 >= 0 && ld->off.h_b1 >= 0),
                    "Q-head parameter offsets missing");
    PRISM_CHECK_ARG(ld->dims.n_heads == 0 || ld->dims.head_layers == 1 ||
                        (ld->off.h_w2 >= 0 && (!ld->dims.use_layer_norm || (ld->off.h_ln1_g >= 0 && ld->off.h_ln2_g >= 0))),
                    "two-layer Q-head parameter offsets missing");
    PRISM_CHECK_ARG(!ld->dims.use_iqn || (ld->off.iqn_w1 >= 0 && ld->off.iqn_w2 >= 0 &&
                                          (!ld->dims.use_layer_norm || (ld->off.iqn_ln1_g >= 0 && ld->off.iqn_ln2_g >= 0))),
                    "IQN parameter offsets missing");
    if (!need_batch) return PRISM_OK;
    PRISM_CHECK_ARG(ld->obs && ld->next_obs && ld->reward && ld->nonterminal && ld->gamma && ld->action,
                    "null batch arrays");
    PRISM_CHECK_ARG(ld->out_td && ld->out_scalars, "null outputs");
    return PRISM_OK;
}

// fused writeback in two halves: preparation beside the gradient reduction (post launch), level walk
// beside clip + Adam (back launch, 256-thread workgroups: one leaf per thread)
static bool split_writeback(const prism_learner_desc *ld) { return ld->batch <= 256; }

static int bwd_row_chunks(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    if (use_bw3(ld)) return BW3_RC;
    if (use_bw4(ld)) return bw4_chunks(ld->batch, d.n_tau);
    return bwd_chunks(iqn_width(d));
}

static bool conv_in_bwd(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    if (use_bw3(ld)) return bw3_conv_ok(d.use_iqn, d.n_heads, d.propagate_grad, d.n_tau, d.in_channels, ld->batch);
    if (use_bw4(ld)) return false;
    return bwd_conv_ok(d.use_iqn, d.n_heads, d.propagate_grad, d.n_tau, d.in_channels, ld->batch, bwd_chunks(iqn_width(d)),
                       iqn_width(d));
}

static int post_block_count(const prism_learner_desc *ld) {
    const prism_model_dims &d = ld->dims;
    if (d.head_layers == 1 && d.n_heads) return post_blocks_dqn1(d.in_channels);
    return post_blocks(ld->batch, d.in_channels, d.use_iqn, d.n_heads, conv_in_bwd(ld), iqn_slab_floats(iqn_width(d), d.use_layer_norm),
                       q_slab_floats(head_width(d), d.use_layer_norm), iqn_width(d), head_width(d),
                       bwd_row_chunks(ld));
}

static void fill_iqn_args(const prism_learner_desc *ld, IqnArgs &a) {
    const prism_model_dims &d = ld->dims;
    const int B = ld->batch;
    memset(&a, 0, sizeof(a));
    float *tau_buf = nullptr, *dl_buf = nullptr;
    carve_iqn(&d, B, ld->workspace, &a.ws, &tau_buf, &dl_buf);
    a.B = B;
    a.Bt = B;
    a.A = d.n_actions;
    a.C = d.in_channels;
    a.T = d.n_tau;
    a.Tn = d.n_tau_next;
    a.Hi = iqn_width(d);
    a.Hq = head_width(d);
    a.ln = d.use_layer_norm;
    a.slab = iqn_slab_floats(a.Hi, a.ln);
    a.q_slab = q_slab_floats(a.Hq, a.ln);
    a.n_chunks = bwd_row_chunks(ld);
    a.has_target = d.has_target;
    a.double_q = d.double_q;
    a.propagate_grad = d.propagate_grad;
    a.squish = d.squish_fn;
    a.split = use_split(ld);
    // the Q heads' input-side backward as two shared-operand GEMMs on the bf16 pipe (qbwd2_kernels.h) where it applies
    a.q_de_slots = d.n_heads;
    if (a.split && qb2_ok(a.Hq, B, d.n_heads, d.head_layers)) a.q_de_slots = 2;
    a.conv_in_bwd = conv_in_bwd(ld);
    a.conv_rows = use_bw3(ld) ? 1 : 4;
    a.bg = bwd_geometry(a.Hi, a.B, a.C, a.T, a.n_chunks, a.conv_in_bwd != 0);
    a.huber_k = d.huber_k;
    a.dist_w = d.dist_loss_weight;
    a.use_iqn = d.use_iqn;
    a.n_heads = d.n_heads;
    a.head_layers = d.head_layers;
    a.q_w = d.q_loss_weight;
    a.theil_coef = d.n_heads > 1 ? d.theil_coef : 0.f;
    { const char *e = getenv("PRISM_DBG"); a.dbg = e ? atoi(e) : 0; }
    a.stamps = (unsigned long long *)ld->dbg_stamps;
    if (!a.stamps) a.dbg &= ~(8 | 16 | 32);
    a.off = ld->off;
    a.params = ld->params;
    a.target_params = ld->target_params;
    a.obs = ld->obs;
    a.next_obs = ld->next_obs;
    a.reward = ld->reward;
    a.gamma = ld->gamma;
    a.per_weights = ld->per_weights;
    a.nonterminal = ld->nonterminal;
    a.action = ld->action;
    a.seed = ld->seed;
    a.offset = ld->offset;
    a.rng = ld->rng_counters;
    a.tau_out = ld->tau_out ? ld->tau_out : tau_buf;
    a.maxT = d.n_tau > d.n_tau_next ? d.n_tau : d.n_tau_next;
    a.out_dl = ld->out_dist_loss ? ld->out_dist_loss : dl_buf;
    a.out_ql = ld->out_q_loss ? ld->out_q_loss : dl_buf + B;
    a.out_td = ld->out_td;
    a.out_scalars = ld->out_scalars;
    a.grads = ld->grads;
    // passes; the tau draws are tied to stream_id in the reference's draw order (iqn_model.py:104,112-126)
    int np = 0;
    const float *uv0 = a.ws.uv, *uv1 = a.ws.uv + UV_ROWS * a.Hi;
    if (d.use_iqn) {
        a.local_loss = local_loss(d);
        if (a.local_loss) {
            // one pass of mixed tiles: each holds whole samples (T current-state + T next-state rows)
            IqnPass p;
            memset(&p, 0, sizeof(p));
            p.params = ld->params;
            p.wpk = a.ws.wpk[0];
            p.uv = uv0;
            p.e = a.ws.e_cur;
            p.e2 = a.ws.e_next;
            p.tau_in = ld->tau_cur;
            p.tau_in2 = ld->tau_next_online;
            p.z_out = a.ws.zcur;
            p.z_out2 = a.ws.zon;
            p.T = d.n_tau;
            p.n_tiles = B * 2 * d.n_tau / 16;
            p.save = 1;
            p.stream_id = 0;
            p.kind = 2;
            a.pass[np++] = p;
            a.ws.ztg = a.ws.zon;
        } else {
            auto iqn_pass = [&](const float *params, int set, const float *e, const float *tau, float *z, int T, int save,
                                int sid) {
                IqnPass p;
                memset(&p, 0, sizeof(p));
                p.params = params;
                p.wpk = a.ws.wpk[set];
                p.uv = set ? uv1 : uv0;
                p.e = e;
                p.tau_in = tau;
                p.z_out = z;
                p.T = T;
                p.n_tiles = B * T / 16;
                p.save = save;
                p.stream_id = sid;
                p.kind = 0;
                return p;
            };
            a.pass[np++] = iqn_pass(ld->params, 0, a.ws.e_cur, ld->tau_cur, a.ws.zcur, d.n_tau, 1, 0);
            if (!d.has_target || d.double_q)
                a.pass[np++] = iqn_pass(ld->params, 0, a.ws.e_next, ld->tau_next_online, a.ws.zon, d.n_tau_next, 0, 1);
            if (d.has_target)
                a.pass[np++] = iqn_pass(ld->target_params, 1, a.ws.e_next, ld->tau_next_target, a.ws.ztg, d.n_tau_next, 0, 2);
            if (!d.has_target) a.ws.ztg = a.ws.zon;            // bootstrap from self
            else if (!d.double_q) a.ws.zon = a.ws.ztg;         // DQN-style: target picks the action too
        }
    }
    if (d.n_heads > 0 && d.head_layers == 2) {
        // Q-head tiles: (B/16) x heads per pass; same online/target selection (q_ensemble.py:62-68)
        const int nt = (B / 16) * d.n_heads;
        const float *quv0 = a.ws.q_uv, *quv1 = a.ws.q_uv + (size_t)d.n_heads * UV_ROWS * a.Hq;
        auto q_pass = [&](const float *params, int set, const float *e, float *z, int save, int sid) {
            IqnPass p;
            memset(&p, 0, sizeof(p));
            p.params = params;
            p.wpk = a.ws.q_wpk[set];
            p.uv = set ? quv1 : quv0;
            p.e = e;
            p.z_out = z;
            p.T = 1;
            p.n_tiles = nt;
            p.save = save;
            p.stream_id = sid;
            p.kind = 1;
            return p;
        };
        a.pass[np++] = q_pass(ld->params, 0, a.ws.e_cur, a.ws.zq_cur, 1, 0);
        if (!d.has_target || d.double_q) a.pass[np++] = q_pass(ld->params, 0, a.ws.e_next, a.ws.zq_on, 0, 1);
        if (d.has_target) a.pass[np++] = q_pass(ld->target_params, 1, a.ws.e_next, a.ws.zq_tg, 0, 2);
        if (!d.has_target) a.ws.zq_tg = a.ws.zq_on;
        else if (!d.double_q) a.ws.zq_on = a.ws.zq_tg;
    }
    a.n_pass = np;
    // the IQN tiles' quantile samples + cos basis come prepared from the embed / front launch (bf16 mode: the prologue they
    // replace is the split forward's)
    a.cos_tiles = 0;
    if (a.split)
        for (int i = 0; i < np; ++i)
            if (a.pass[i].kind != 1) {
                a.pass[i].cospk = a.ws.cospk + (size_t)a.cos_tiles * CP_TILE;
                a.cos_tiles += a.pass[i].n_tiles;
            }
}

static void fill_adam_args(const prism_learner_desc *ld, const IqnWs &ws, AdamArgs &a) {
    a.p = ld->params;
    a.g = ld->grads;
    a.m = ld->adam_m;
    a.v = ld->adam_v;
    a.n = ld->off.n_params;
    a.step = ld->adam_step;
    a.normpart = ws.normpart;
    a.lr = ld->hyper.lr;
    a.b1 = ld->hyper.beta1;
    a.b2 = ld->hyper.beta2;
    a.eps = ld->hyper.eps;
    a.max_norm = ld->hyper.max_grad_norm;
    a.grad_scale = ld->hyper.grad_scale;
    a.out_scalars = ld->out_scalars;
    a.ticket = ws.ticket;
    // data parallel (the gradient is scaled by 1 / world): a collective that gave up on a peer poisons the step
    a.poison = ld->hyper.grad_scale != 1.0f ? ws.ticket + PRISM_WS_STATUS_WORD : nullptr;
}

// (one workgroup per CU -- every workgroup folds the norm partials itself before its first update, a fixed cost per workgroup:
// measured on c4's 1.5 M parameters, clip + Adam + writeback: 2048 workgroups 25.5 us, 512: 15.5, 256: 13.4, 128: 15.4;
// subtractive preset, 3.0 M: 33.8 / 20.7 / 18.8 / 23.8)
#ifndef ADAM_MAX_BLOCKS
#define ADAM_MAX_BLOCKS 256
#endif
static int adam_blocks(int64_t n) {
    int blocks = (int)(((n >> 2) + 255) / 256);
    if (blocks > ADAM_MAX_BLOCKS) blocks = ADAM_MAX_BLOCKS;
    return blocks < 1 ? 1 : blocks;
}

// template dispatch over the hidden width and LayerNorm on/off
template <typename F>
static void dispatch_hl(int H, int ln, F &&f) {
    if (H == 128) {
        if (ln) f(std::integral_constant<int, 128>{}, std::true_type{});
        else f(std::integral_constant<int, 128>{}, std::false_type{});
    } else {
        if (ln) f(std::integral_constant<int, 256>{}, std::true_type{});
        else f(std::integral_constant<int, 256>{}, std::false_type{});
    }
}

// dynamic-LDS opt-in of the kernels that need more than 64 KB: once per device and instantiation
static hipError_t set_max_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

static int device_cus() {
    static std::mutex mu;
    static std::map<int, int> cache;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(dev);
    if (it != cache.end()) return it->second;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    return cache[dev] = cus;
}

// Forward tiles on four waves (fwd_kernels.h, WAVES = 4: two 256-thread workgroups per CU, the latency-bound phases of one tile
// beside the weight stream of another): where the launch has at least two tiles per CU and no mixed tile (their loss tail
// needs all sixteen rows in registers at once).  PRISM_FWD_WAVES=8 / 4 forces a form (A/B runs).
static bool fwd_four_waves(const IqnArgs &aa, int tiles) {
    if (!aa.split) return false;
    for (int i = 0; i < aa.n_pass; ++i)
        if (aa.pass[i].kind == 2) return false;
    static const int forced = [] { const char *e = getenv("PRISM_FWD_WAVES"); return e ? atoi(e) : 0; }();
    if (forced == 8) return false;
    if (forced == 4) return true;
    return tiles >= 2 * device_cus();
}

// ---- the post launch: gradient slabs / small tensors / conv fold (+ priority writeback block); with `tail` also the
// clip + Adam update behind a grid barrier (single GPU) ------------------------------------------------------------
// workgroups of the fused-tail instantiation `dense` that fit the CURRENT device at once (per device: processes that
// drive several GPUs, and per instantiation: the two forms differ in registers)
static int post_max_resident(bool dense) {
    static std::mutex mu;
    static std::map<std::pair<int, bool>, int> cache;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, dense});
    if (it != cache.end()) return it->second;
    int cus = 0, per = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const hipError_t e = dense ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, iqn_post_kernel<true, true, true>, 1024, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, iqn_post_kernel<true, true, false>, 1024, 0);
    if (e != hipSuccess) return 0;
    return cache[{dev, dense}] = cus * per;
}

static bool writeback_rides(const prism_learner_desc *ld) { return ld->fused_replay && ld->fused_replay->tree && ld->fused_index; }
// the writeback block of the post launch recomputes the top of the tree whole when it may (tree_dense_finish)
static bool post_dense(const prism_learner_desc *ld) {
    // the full writer's live threads: 2 x the batch rounded up to waves (iqn_post_kernel)
    const int wb_live = ld->batch <= UPD_MAX ? std::min(1024, 2 * ((ld->batch + 63) & ~63)) : 1024;
    return writeback_rides(ld) && tree_dense_ok(ld->fused_replay->tree_capacity, ld->batch, wb_live);
}

// The fused tail needs every workgroup of the launch resident at once (it has a grid barrier; the device must not be
// shared with other processes' kernels -- the barrier gives up after GRID_WAIT_TICKS and flags it, step_kernels.h) and
// nothing between the gradient and the optimizer step (no all-reduce: grad_scale 1).
static bool tail_fused(const prism_learner_desc *ld) {
    if (!ld->fuse_tail || ld->hyper.grad_scale != 1.0f) return false;
    // without a priority writeback riding along there is nothing for the fused launch to hide behind the barrier
    // (measured, uniform replay + one-layer DQN head: 32.7 us fused vs 30.1 us as two launches)
    static const bool always = [] { const char *e = getenv("PRISM_FUSED_TAIL_ALWAYS"); return e && atoi(e) != 0; }();
    // (an IQN's gradient slabs are enough to hide: additive ablation base, uniform replay, width 256: 95.5 vs 97.3 us per step)
    if (!always && !writeback_rides(ld) && !ld->dims.use_iqn) return false;
    static const bool off = [] { const char *e = getenv("PRISM_NO_FUSED_TAIL"); return e && atoi(e) != 0; }();
    if (off) return false;
    return post_block_count(ld) + 1 <= post_max_resident(post_dense(ld));
}

static int launch_post(const prism_learner_desc *ld, const IqnArgs &a, const TailArgs *tail, hipStream_t stream) {
    ProfileScope ps_(tail ? K_TAIL : K_POST, stream);
    int nb = post_block_count(ld);
    PostWriteback wb;
    memset(&wb, 0, sizeof(wb));
    if (writeback_rides(ld)) {
        // TD errors are final: the priority writeback rides along as one more block of this launch
        wb.enabled = 1;
        wb.rp = *ld->fused_replay;
        wb.index = ld->fused_index;
        wb.sib = reinterpret_cast<const float2 *>(a.ws.sib);
        wb.sib_state = a.ws.ticket + 3;
        wb.plan = (!tail && split_writeback(ld)) ? reinterpret_cast<int4 *>(a.ws.wb_plan) : nullptr;
        wb.alpha = ld->fused_alpha;
        wb.eps = ld->fused_eps;
        wb.block = nb;
        nb += 1;
    }
    TailArgs none;
    memset(&none, 0, sizeof(none));
    const bool dense = post_dense(ld);
    if (tail && dense) hipLaunchKernelGGL((iqn_post_kernel<true, true, true>), dim3(nb), dim3(1024), 0, stream, a, wb, *tail);
    else if (tail) hipLaunchKernelGGL((iqn_post_kernel<true, true, false>), dim3(nb), dim3(1024), 0, stream, a, wb, *tail);
    else if (wb.plan) hipLaunchKernelGGL((iqn_post_kernel<false, false, false>), dim3(nb), dim3(1024), 0, stream, a, wb, none);
    else if (dense) hipLaunchKernelGGL((iqn_post_kernel<true, false, true>), dim3(nb), dim3(1024), 0, stream, a, wb, none);
    else hipLaunchKernelGGL((iqn_post_kernel<true, false, false>), dim3(nb), dim3(1024), 0, stream, a, wb, none);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_learner_fwd_bwd(const prism_learner_desc *ld, prism_stream_t stream_) {
    int rc = check_learner(ld);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int B = ld->batch;
    IqnArgs a;
    fill_iqn_args(ld, a);
    hipError_t herr = hipSuccess;

    if (!ld->embed_done) {
        ProfileScope ps_(K_EMBED, stream);
        hipLaunchKernelGGL(iqn_embed_kernel, dim3(2 * B + front_extra_blocks(extra_dims(a))), dim3(256), 0, stream, a);
        PRISM_CHECK_LAUNCH();
    }
    // forward tiles: one launch when every tile kind has the same hidden width, else one per kind
    {
        int n_iqn = 0, n_q = 0;
        for (int i = 0; i < a.n_pass; ++i) (a.pass[i].kind == 1 ? n_q : n_iqn) += a.pass[i].n_tiles;
        auto launch = [&](const IqnArgs &aa, int H, int tiles) {
            dispatch_hl(H, aa.ln, [&](auto h, auto l) {
                constexpr int HH = decltype(h)::value;
                constexpr bool LL = decltype(l)::value;
                const size_t lds = fw_lds_floats<HH>() * sizeof(float);
                if constexpr (HH == 128) {
                    if (aa.split && fwd_four_waves(aa, tiles)) {
                        const size_t lds4 = fw_lds_floats<HH, 4>() * sizeof(float);
                        herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL, true, 4>, lds4);
                        if (herr == hipSuccess)
                            hipLaunchKernelGGL((fwd_tile_kernel<HH, LL, true, 4>), dim3(tiles), dim3(256), lds4, stream, aa);
                        return;
                    }
                }
                if (aa.split) {
                    herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL, true>, lds);
                    if (herr == hipSuccess)
                        hipLaunchKernelGGL((fwd_tile_kernel<HH, LL, true>), dim3(tiles), dim3(fw_threads(HH)), lds, stream, aa);
                    return;
                }
                herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL>, lds);
                if (herr == hipSuccess) hipLaunchKernelGGL((fwd_tile_kernel<HH, LL>), dim3(tiles), dim3(fw_threads(HH)), lds, stream, aa);
            });
        };
        ProfileScope ps_(K_TILE_FWD, stream);
        if (n_iqn && n_q && a.Hi != a.Hq) {
            IqnArgs a1 = a, a2 = a;
            a1.n_pass = a2.n_pass = 0;
            for (int i = 0; i < a.n_pass; ++i) {
                if (a.pass[i].kind == 1) a2.pass[a2.n_pass++] = a.pass[i];
                else a1.pass[a1.n_pass++] = a.pass[i];
            }
            launch(a1, a.Hi, n_iqn);
            launch(a2, a.Hq, n_q);
        } else if (n_iqn + n_q > 0) {
            launch(a, n_iqn ? a.Hi : a.Hq, n_iqn + n_q);
        }
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(fwd_tile): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    }
    // losses first (TD errors final), then the backward kernels
    const bool merged_loss = ld->dims.use_iqn && !a.local_loss && ld->dims.n_heads > 0 && ld->dims.head_layers == 2 && a.Hi == a.Hq &&
                             loss_waves(a.T) == LOSS_WAVES;
    if (merged_loss) {
        ProfileScope ps_(K_LOSS, stream);
        dispatch_hl(a.Hi, a.ln, [&](auto h, auto l) {
            hipLaunchKernelGGL((loss_both_kernel<decltype(h)::value, decltype(l)::value>), dim3(2 * B), dim3(512), 0, stream, a);
        });
        PRISM_CHECK_LAUNCH();
    }
    if (ld->dims.use_iqn && !a.local_loss && !merged_loss) {
        ProfileScope ps_(K_LOSS, stream);
        dispatch_hl(a.Hi, a.ln, [&](auto h, auto l) {
            if (loss_waves(a.T) == 16)
                hipLaunchKernelGGL((iqn_loss_kernel<decltype(h)::value, decltype(l)::value, 16>), dim3(B), dim3(64 * 16), 0, stream, a);
            else
                hipLaunchKernelGGL((iqn_loss_kernel<decltype(h)::value, decltype(l)::value, LOSS_WAVES>), dim3(B),
                                   dim3(64 * LOSS_WAVES), 0, stream, a);
        });
        PRISM_CHECK_LAUNCH();
    }
    if (ld->dims.n_heads > 0 && !merged_loss) {
        ProfileScope ps_(K_Q_FWD, stream);
        if (ld->dims.head_layers == 1) {
            hipLaunchKernelGGL(dqn_loss_kernel, dim3(B), dim3(256), 0, stream, a);
        } else {
            dispatch_hl(a.Hq, a.ln, [&](auto h, auto l) {
                hipLaunchKernelGGL((qh_loss_kernel<decltype(h)::value, decltype(l)::value>), dim3(B), dim3(512), 0, stream, a);
            });
        }
        PRISM_CHECK_LAUNCH();
    }
    if (ld->dims.use_iqn && use_bw3(ld)) {
        ProfileScope ps_(K_BWD, stream);
        const dim3 grid((E_DIM / 64) * BW3_RC);
        const size_t lds = (size_t)bw3_lds_bytes(B, a.T, a.C, a.conv_in_bwd != 0);
        if (a.ln) {
            herr = set_max_lds((const void *)iqn_bwd3_kernel<true>, 160 * 1024);
            if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd3_kernel<true>), grid, dim3(512), lds, stream, a);
        } else {
            herr = set_max_lds((const void *)iqn_bwd3_kernel<false>, 160 * 1024);
            if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd3_kernel<false>), grid, dim3(512), lds, stream, a);
        }
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(iqn_bwd3): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    } else if (ld->dims.use_iqn && use_bw4(ld)) {
        ProfileScope ps_(K_BWD, stream);
        const dim3 grid((E_DIM / 32) * a.n_chunks);
        if (a.ln) {
            herr = set_max_lds((const void *)iqn_bwd4_kernel<true>, BW4_LDS_BYTES);
            if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd4_kernel<true>), grid, dim3(512), BW4_LDS_BYTES, stream, a);
        } else {
            herr = set_max_lds((const void *)iqn_bwd4_kernel<false>, BW4_LDS_BYTES);
            if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd4_kernel<false>), grid, dim3(512), BW4_LDS_BYTES, stream, a);
        }
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(iqn_bwd4): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    } else if (ld->dims.use_iqn) {
        if (!bwd_lds_layout_ok(a.Hi, B, a.C, a.T, a.n_chunks, a.conv_in_bwd != 0)) {
            set_error("prism_learner_fwd_bwd: internal: LDS layout of the backward kernel overlaps for this shape");
            return PRISM_ERR_INVALID;
        }
        ProfileScope ps_(K_BWD, stream);
        const bool db = bwd_double_buffered(a.Hi);
        dispatch_hl(a.Hi, a.ln, [&](auto h, auto l) {
            constexpr int HH = decltype(h)::value;
            constexpr bool LL = decltype(l)::value;
            const size_t lds = (size_t)bwd_lds_floats(HH, B, a.C, a.T, a.n_chunks, a.conv_in_bwd != 0) * sizeof(float);
            const dim3 grid((E_DIM / 16) * a.n_chunks);
            if constexpr (HH == 128) {
                if (db) {
                    herr = set_max_lds((const void *)iqn_bwd_kernel<HH, LL, true>, lds);
                    if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd_kernel<HH, LL, true>), grid, dim3(256), lds, stream, a);
                    return;
                }
            }
            herr = set_max_lds((const void *)iqn_bwd_kernel<HH, LL, false>, lds);
            if (herr == hipSuccess) hipLaunchKernelGGL((iqn_bwd_kernel<HH, LL, false>), grid, dim3(256), lds, stream, a);
        });
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(iqn_bwd): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    }
    if (ld->dims.n_heads > 0 && ld->dims.head_layers == 2 && a.split && qb2_ok(a.Hq, B, ld->dims.n_heads, ld->dims.head_layers)) {
        ProfileScope ps_(K_Q_BWD, stream);
        const dim3 grid(qb2_blocks(ld->dims.n_heads, B));
        if (a.ln) {
            herr = set_max_lds((const void *)qh_bwd2_kernel<true>, QB2_LDS_BYTES);
            if (herr == hipSuccess) hipLaunchKernelGGL((qh_bwd2_kernel<true>), grid, dim3(256), QB2_LDS_BYTES, stream, a);
        } else {
            herr = set_max_lds((const void *)qh_bwd2_kernel<false>, QB2_LDS_BYTES);
            if (herr == hipSuccess) hipLaunchKernelGGL((qh_bwd2_kernel<false>), grid, dim3(256), QB2_LDS_BYTES, stream, a);
        }
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(qh_bwd2): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    } else if (ld->dims.n_heads > 0 && ld->dims.head_layers == 2) {
        ProfileScope ps_(K_Q_BWD, stream);
        dispatch_hl(a.Hq, a.ln, [&](auto h, auto l) {
            constexpr int HH = decltype(h)::value;
            constexpr bool LL = decltype(l)::value;
            const size_t lds = qb_lds_floats(HH) * sizeof(float);
            static const bool cols_off = [] { const char *e = getenv("PRISM_QB_COLS"); return e && atoi(e) == 0; }();      // (A/B runs)
            if (B <= 128 && !cols_off) {      // small batches: a wave per column slice over all rows (qhead_kernels.h, COLS)
                herr = set_max_lds((const void *)qh_bwd_kernel<HH, LL, true>, lds);
                if (herr == hipSuccess)
                    hipLaunchKernelGGL((qh_bwd_kernel<HH, LL, true>), dim3((E_DIM / 64) * ld->dims.n_heads), dim3(256), lds, stream, a);
                return;
            }
            herr = set_max_lds((const void *)qh_bwd_kernel<HH, LL>, lds);
            if (herr == hipSuccess)
                hipLaunchKernelGGL((qh_bwd_kernel<HH, LL>), dim3((E_DIM / 16) * ld->dims.n_heads), dim3(256), lds, stream, a);
        });
        if (herr != hipSuccess) {
            set_error("hipFuncSetAttribute(qh_bwd): %s", hipGetErrorString(herr));
            return PRISM_ERR_HIP;
        }
        PRISM_CHECK_LAUNCH();
    }
    if (!tail_fused(ld)) {
        rc = launch_post(ld, a, nullptr, stream);
        if (rc) return rc;
    }
    if (ld->dbg_z) {
        const size_t R = (size_t)B * ld->dims.n_tau, Rn = (size_t)B * ld->dims.n_tau_next, A = ld->dims.n_actions;
        (void)hipMemcpyAsync(ld->dbg_z, a.ws.zcur, R * A * 4, hipMemcpyDeviceToDevice, stream);
        (void)hipMemcpyAsync(ld->dbg_z + R * A, a.ws.ztg, Rn * A * 4, hipMemcpyDeviceToDevice, stream);
    }
    return PRISM_OK;
}

// Acting forward (agent.py:31-41 -> composite_model.py:51-70, iqn_model.py:61-87 with for_action=True): embeds `n`
// observations, runs n_tau quantile rows per observation through the IQN tiles and the observations through every
// Q head -- the same forward tiles as the update, on the learner's workspace (the stream-packed weights are rebuilt
// first: the last Adam step left them stale).
extern "C" int prism_act_forward(const prism_learner_desc *ld, const float *obs, int32_t n, int32_t n_tau,
                                 const float *tau_in, uint64_t seed, uint64_t offset, float *out_z, float *out_q,
                                 prism_stream_t stream_) {
    int rc = check_learner(ld, false);
    if (rc) return rc;
    PRISM_CHECK_ARG(obs != nullptr && n >= 1 && n <= ld->batch, "n must be in [1, batch]");
    const int n_pad = (n + 15) / 16 * 16;
    PRISM_CHECK_ARG(n_pad <= ld->batch || ld->dims.n_heads == 0 || ld->dims.head_layers != 2,
                    "Q-head tiles need the workspace of a batch >= 16-padded n");
    PRISM_CHECK_ARG(!ld->dims.use_iqn || (n_tau >= 1 && out_z), "quantile samples per action / output buffer");
    PRISM_CHECK_ARG(!(ld->dims.n_heads && ld->dims.head_layers == 2) || out_q, "Q output buffer");
    hipStream_t stream = (hipStream_t)stream_;
    IqnArgs a;
    fill_iqn_args(ld, a);
    hipError_t herr = hipSuccess;
    // embed (+ the parameter-only roles): the n observations stand in for both batch halves
    a.cos_tiles = 0;          // (acting tiles draw and evaluate their basis themselves: no learner passes here)
    a.B = n;
    a.obs = a.next_obs = obs;
    // quantile draws counted on the device (ld->rng_counters[2], added to `offset`): the call can be captured into a hipGraph
    // and replayed -- the embed launch advances the counter by this call's n * n_tau draws, the tiles of the next launch
    // start from counter - n * n_tau
    const uint64_t act_inc = ld->dims.use_iqn ? (uint64_t)n * (uint64_t)n_tau : 0ull;
    a.act_rng = (ld->rng_counters && !tau_in && act_inc) ? ld->rng_counters + 2 : nullptr;
    a.act_inc = act_inc;
    // PRISM_ACT_WEIGHTS_CURRENT: the stream-packed weight copies / LayerNorm helpers in the workspace were built from the
    // parameters as they are now (by an earlier acting call since the last update): the launch is the n embeddings alone
    const int extra = (ld->act_flags & PRISM_ACT_WEIGHTS_CURRENT) ? 0 : front_extra_blocks(extra_dims(a));
    hipLaunchKernelGGL(iqn_embed_kernel, dim3((extra ? 2 * n : n) + extra), dim3(256), 0, stream, a);
    PRISM_CHECK_LAUNCH();
    if (ld->dims.n_heads > 0 && ld->dims.head_layers == 1) {
        // single-Linear DQN head: one workgroup per observation on the embeddings just written
        PRISM_CHECK_ARG(out_q != nullptr, "Q output buffer");
        hipLaunchKernelGGL(dqn1_act_kernel, dim3(n), dim3(256), 0, stream, a, out_q);
        PRISM_CHECK_LAUNCH();
        return PRISM_OK;
    }
    a.B = n_pad;
    a.Bt = n;
    a.seed = seed;
    a.offset = a.act_rng ? offset - act_inc : offset;
    a.rng = a.act_rng ? ld->rng_counters + 1 : nullptr;      // (the tiles read word [1] of what they are given)
    a.act_rng = nullptr;
    a.tau_out = nullptr;
    a.local_loss = 0;
    int np = 0, n_iqn = 0, n_q = 0;
    IqnPass p;
    if (ld->dims.use_iqn) {
        memset(&p, 0, sizeof(p));
        p.params = ld->params;
        p.wpk = a.ws.wpk[0];
        p.uv = a.ws.uv;
        p.e = a.ws.e_cur;
        p.tau_in = tau_in;
        p.z_out = out_z;
        p.T = n_tau;
        p.n_tiles = n_iqn = (n * n_tau + 15) / 16;
        p.kind = 0;
        p.stream_id = 3;                 // a Philox stream of its own: acting draws never repeat an update's
        a.pass[np++] = p;
    }
    if (ld->dims.n_heads > 0 && ld->dims.head_layers == 2) {
        memset(&p, 0, sizeof(p));
        p.params = ld->params;
        p.wpk = a.ws.q_wpk[0];
        p.uv = a.ws.q_uv;
        p.e = a.ws.e_cur;
        p.z_out = out_q;
        p.T = 1;
        p.n_tiles = n_q = (n_pad / 16) * ld->dims.n_heads;
        p.kind = 1;
        a.pass[np++] = p;
    }
    PRISM_CHECK_ARG(np > 0, "nothing to run");
    auto launch = [&](const IqnArgs &aa, int H, int tiles) {
        dispatch_hl(H, aa.ln, [&](auto h, auto l) {
            constexpr int HH = decltype(h)::value;
            constexpr bool LL = decltype(l)::value;
            const size_t lds = fw_lds_floats<HH>() * sizeof(float);
            if constexpr (HH == 128) {
                if (aa.split && fwd_four_waves(aa, tiles)) {
                    const size_t lds4 = fw_lds_floats<HH, 4>() * sizeof(float);
                    herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL, true, 4>, lds4);
                    if (herr == hipSuccess)
                        hipLaunchKernelGGL((fwd_tile_kernel<HH, LL, true, 4>), dim3(tiles), dim3(256), lds4, stream, aa);
                    return;
                }
            }
            if (aa.split) {
                herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL, true>, lds);
                if (herr == hipSuccess)
                    hipLaunchKernelGGL((fwd_tile_kernel<HH, LL, true>), dim3(tiles), dim3(fw_threads(HH)), lds, stream, aa);
                return;
            }
            herr = set_max_lds((const void *)fwd_tile_kernel<HH, LL>, lds);
            if (herr == hipSuccess) hipLaunchKernelGGL((fwd_tile_kernel<HH, LL>), dim3(tiles), dim3(fw_threads(HH)), lds, stream, aa);
        });
    };
    a.n_pass = np;
    if (n_iqn && n_q && a.Hi != a.Hq) {
        IqnArgs a1 = a, a2 = a;
        a1.n_pass = a2.n_pass = 1;
        a2.pass[0] = a.pass[1];
        launch(a1, a.Hi, n_iqn);
        launch(a2, a.Hq, n_q);
    } else {
        launch(a, n_iqn ? a.Hi : a.Hq, n_iqn + n_q);
    }
    if (herr != hipSuccess) {
        set_error("hipFuncSetAttribute(fwd_tile): %s", hipGetErrorString(herr));
        return PRISM_ERR_HIP;
    }
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

// IDSActionSelector.generate_action_probs + select_action without random sampling (action_selectors.py:125-176).
extern "C" int prism_ids_select(const float *z, const float *q, int32_t n, int32_t n_pad, int32_t n_tau, int32_t n_actions,
                                int32_t n_heads, float lmbda, float epsilon, float rho_lower_bound, int32_t unsquish_fn,
                                float *out_scores, float *out_aux, int64_t *out_action, int64_t *out_action_host,
                                prism_stream_t stream_) {
    PRISM_CHECK_ARG(z && q && out_scores && out_action, "null buffers");
    PRISM_CHECK_ARG(unsquish_fn >= PRISM_SQUISH_NONE && unsquish_fn <= PRISM_SQUISH_SYMLOG, "unknown unsquish function");
    PRISM_CHECK_ARG(n >= 1 && n_pad >= n && n_tau >= 1 && n_actions >= 1 && n_actions <= 16 && n_heads >= 1, "bad sizes");
    const int stage = n_tau * n_actions <= ACT_STAGE_MAX_FLOATS;
    IdsArgs k{z, q, n, n_pad, n_tau, n_actions, n_heads, lmbda, epsilon, rho_lower_bound, out_scores, out_aux, out_action,
              out_action_host, stage, unsquish_fn};
    hipLaunchKernelGGL(ids_score_kernel, dim3(n), dim3(ACT_THREADS), stage ? (size_t)n_tau * n_actions * 4 : 0, (hipStream_t)stream_, k);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

// GreedyActionSelector (action_selectors.py:70-83) on the buffers prism_act_forward filled.
extern "C" int prism_greedy_select(const float *z, const float *q, int32_t n, int32_t n_pad, int32_t n_tau, int32_t n_actions,
                                   int32_t n_heads, int64_t *out_action, float *out_mean, int64_t *out_action_host,
                                   prism_stream_t stream_) {
    PRISM_CHECK_ARG((z || q) && out_action, "null buffers");
    PRISM_CHECK_ARG(n >= 1 && n_pad >= n && n_actions >= 1 && n_actions <= 16, "bad sizes");
    PRISM_CHECK_ARG(q ? n_heads >= 1 : n_tau >= 1, "bad sizes");
    const int stage = !q && n_tau * n_actions <= ACT_STAGE_MAX_FLOATS;
    GreedyArgs k{z, q, n, n_pad, n_tau, n_actions, n_heads, out_action, out_mean, out_action_host, stage};
    hipLaunchKernelGGL(greedy_select_kernel, dim3(n), dim3(ACT_THREADS), stage ? (size_t)n_tau * n_actions * 4 : 0, (hipStream_t)stream_, k);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

// grid-norm partial slots valid for the Adam kernels: either what post left, or a fresh pass
static int prepare_norm(const prism_learner_desc *ld, const IqnWs &ws, AdamArgs &a, hipStream_t stream) {
    if (ld->hyper.grad_scale == 1.0f) {
        a.n_slots = post_block_count(ld);
    } else {
        // data parallel: the gradient was all-reduced after the backward; recompute the partials
        const int nb = 256;
        hipLaunchKernelGGL(grad_sumsq_kernel, dim3(nb), dim3(256), 0, stream, ld->grads, a.n, a.grad_scale,
                           ws.normpart);
        PRISM_CHECK_LAUNCH();
        a.n_slots = nb;
    }
    return PRISM_OK;
}

extern "C" int prism_learner_clip_adam(const prism_learner_desc *ld, prism_stream_t stream_) {
    int rc = check_learner(ld);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    IqnWs ws;
    carve_iqn(&ld->dims, ld->batch, ld->workspace, &ws, nullptr, nullptr);
    AdamArgs a;
    fill_adam_args(ld, ws, a);
    rc = prepare_norm(ld, ws, a, stream);
    if (rc) return rc;
    {
        ProfileScope ps_(K_CLIP_ADAM, stream);
        hipLaunchKernelGGL(clip_adam_kernel, dim3(adam_blocks(a.n)), dim3(256), 0, stream, a);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

static int check_replay_for_step(const prism_learner_desc *ld, const prism_replay_desc *rp) {
    PRISM_CHECK_ARG(rp != nullptr, "null replay descriptor");
    PRISM_CHECK_ARG(rp->obs_elems == 100 * ld->dims.in_channels && (rp->obs_elems & 3) == 0,
                    "replay obs_elems must equal 10*10*C");
    PRISM_CHECK_ARG(rp->n_step >= 1 && rp->n_step <= PRISM_MAX_NSTEP, "n_step out of range");
    return PRISM_OK;
}

extern "C" int prism_step_front(const prism_learner_desc *ld, const prism_replay_desc *rp, int64_t size,
                                const float *mass, uint64_t seed, uint64_t offset, float beta, int64_t *out_index,
                                float *out_weight, prism_stream_t stream_) {
    int rc = check_learner(ld);
    if (rc) return rc;
    rc = check_replay_for_step(ld, rp);
    if (rc) return rc;
    PRISM_CHECK_ARG(size > 0 && size <= rp->capacity, "size must be in (0, capacity] (empty storage)");
    PRISM_CHECK_ARG(out_index && (rp->tree == nullptr || out_weight), "null outputs");
    hipStream_t stream = (hipStream_t)stream_;
    IqnArgs a;
    fill_iqn_args(ld, a);
    FrontArgs f;
    f.size = size;
    f.mass = mass;
    f.seed = seed;
    f.offset = offset;
    f.rng = ld->rng_counters;
    f.beta = beta;
    f.use_per = rp->tree != nullptr;
    f.out_index = out_index;
    f.out_weight = out_weight;
    f.obs = const_cast<float *>(ld->obs);
    f.next_obs = const_cast<float *>(ld->next_obs);
    f.reward = const_cast<float *>(ld->reward);
    f.gamma = const_cast<float *>(ld->gamma);
    f.nonterminal = const_cast<uint8_t *>(ld->nonterminal);
    f.action = const_cast<int64_t *>(ld->action);
    {
        ProfileScope ps_(K_FRONT, stream);
        const int extra = front_extra_blocks(extra_dims(a));
        hipLaunchKernelGGL(step_front_kernel, dim3(ld->batch + extra), dim3(256), 0, stream, a, *rp, f);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

extern "C" int prism_step_back(const prism_learner_desc *ld, const prism_replay_desc *rp, const int64_t *index,
                               float alpha, float eps, prism_stream_t stream_) {
    int rc = check_learner(ld);
    if (rc) return rc;
    rc = check_replay_for_step(ld, rp);
    if (rc) return rc;
    PRISM_CHECK_ARG(index != nullptr, "null index");
    hipStream_t stream = (hipStream_t)stream_;
    IqnWs ws;
    carve_iqn(&ld->dims, ld->batch, ld->workspace, &ws, nullptr, nullptr);
    AdamArgs a;
    fill_adam_args(ld, ws, a);
    rc = prepare_norm(ld, ws, a, stream);
    if (rc) return rc;
    BackArgs k;
    k.index = index;
    k.priority = ld->out_td;
    k.n = ld->batch;
    k.alpha = alpha;
    k.eps = eps;
    k.take_abs = 1;
    k.use_per = rp->tree != nullptr && !ld->fused_replay;   // already written back beside the backward pass
    k.plan = nullptr;
    k.sib = nullptr;
    k.sib_state = nullptr;
    if (ld->fused_replay && ld->fused_replay->tree && ld->fused_index && split_writeback(ld)) {
        k.plan = reinterpret_cast<const int4 *>(ws.wb_plan);
        k.sib = reinterpret_cast<const float2 *>(ws.sib);
        k.sib_state = ws.ticket + 3;
    }
    k.rng = ld->rng_counters;
    const int maxT = ld->dims.n_tau > ld->dims.n_tau_next ? ld->dims.n_tau : ld->dims.n_tau_next;
    k.inc_per = (uint64_t)ld->batch;
    k.inc_tau = (uint64_t)3 * maxT * ld->batch;
    if (tail_fused(ld)) {
        // single GPU: gradient reduction + clip + Adam + writeback in one launch
        IqnArgs ia;
        fill_iqn_args(ld, ia);
        TailArgs t;
        t.adam = a;
        t.barrier = reinterpret_cast<unsigned long long *>(ws.ticket + 4);
        t.status = ws.ticket + PRISM_WS_STATUS_WORD;
        t.host_status = ld->host_status;
        t.rng = k.rng;
        t.inc_per = k.inc_per;
        t.inc_tau = k.inc_tau;
        rc = launch_post(ld, ia, &t, stream);
        if (rc) return rc;
        if (k.use_per) {          // prioritised replay that is not riding in the learner's launches: its own update
            const int threads = ld->batch >= 1024 ? 1024 : ((ld->batch + 127) / 128) * 128;
            if (tree_dense_ok(rp->tree_capacity, ld->batch, threads))
                hipLaunchKernelGGL(per_update_kernel<true>, dim3(1), dim3(threads), 0, stream, *rp, index, ld->out_td, ld->batch,
                                   alpha, eps, 1);
            else
                hipLaunchKernelGGL(per_update_kernel<false>, dim3(1), dim3(threads), 0, stream, *rp, index, ld->out_td, ld->batch,
                                   alpha, eps, 1);
            PRISM_CHECK_LAUNCH();
        }
        return PRISM_OK;
    }
    {
        ProfileScope ps_(K_BACK, stream);
        hipLaunchKernelGGL(step_back_kernel, dim3(1 + adam_blocks(a.n)), dim3(256), 0, stream, a, *rp, k);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}
