/*
 * prism_hip.h — C ABI of libprism_hip.so (gfx950 / MI355X).
 *
 * The reference (AechPro/Prism) has no FFI layer: its hot path is Python calling torch + the
 * third-party torchrl segment trees.  These entry points are what a binding for that path would
 * call; each one names the reference interface it replaces (paths under /root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller unless
 *     the parameter says "host";
 *   - every call enqueues work on `stream` (a hipStream_t passed as void*) and returns without
 *     synchronising; no hidden allocation, no hidden sync — safe to capture into a hipGraph;
 *   - return value: PRISM_OK or a negative PRISM_ERR_* code; prism_last_error() returns a
 *     thread-local message; nothing throws across the boundary;
 *   - single-threaded caller per stream, as the reference's learner loop
 *     (prism/learner.py:75-143).
 */
#ifndef PRISM_HIP_H
#define PRISM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PRISM_ABI_VERSION 3

#define PRISM_OK 0
#define PRISM_ERR_INVALID (-1)     /* bad argument / shape the kernels do not cover          */
#define PRISM_ERR_HIP (-2)         /* a HIP runtime call failed                             */
#define PRISM_ERR_UNSUPPORTED (-3) /* valid reference configuration not implemented here    */

#define PRISM_MAX_NSTEP 15

#define PRISM_GEMM_FP32 1
#define PRISM_GEMM_BF16X3 2
#define PRISM_GEMM_DEFAULT PRISM_GEMM_BF16X3

#define PRISM_ACT_WEIGHTS_CURRENT 1

/* per-slot flag bits of the replay ring */
#define PRISM_FLAG_DONE 1u      /* Timestep.done                                            */
#define PRISM_FLAG_TRUNC 2u     /* Timestep.truncated                                       */
#define PRISM_FLAG_HAS_NEXT 4u  /* Timestep.next is not None                                */

/* device status word bits (prism_replay_desc.status) */
#define PRISM_WS_STATUS_WORD 7             /* index of the sticky status word in prism_learner_desc.workspace (uint32) */
#define PRISM_WS_STATUS_BARRIER_TIMEOUT 1u
#define PRISM_WS_STATUS_COLLECTIVE_TIMEOUT 2u /* a direct all-reduce wait gave up (prism_direct_desc.poison): clip + Adam are skipped */
#define PRISM_STATUS_NONPOSITIVE_PSUM 1
#define PRISM_STATUS_NONPOSITIVE_PMIN 2

typedef void *prism_stream_t;

const char *prism_last_error(void);
int prism_abi_version(void);
/* host out-params; multiprocessor count and gcnArch name ("gfx950") of `device`. */
int prism_device_info(int device, int *cu_count, char *arch_name, int arch_name_len);

/* ------------------------------------------------------------------------------------------
 * Replay ring + prioritized sum/min trees, resident in HBM.
 *
 * Replaces torchrl PrioritizedReplayBuffer(ListStorage) + prism TimestepBuffer
 * (prism/factory/exp_buffer_factory.py:22-33, prism/experience/timestep_buffer.py:10-238).
 * Layout: structure-of-arrays ring of `capacity` slots; trees are binary segment trees with
 * `tree_capacity` = smallest power of two STRICTLY greater than `capacity`, 2*tree_capacity fp32
 * nodes each, leaf i at node (i | tree_capacity), root at node 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct prism_replay_desc {
    int64_t capacity;
    int64_t tree_capacity;
    int32_t obs_elems;   /* floats per observation (10*10*C for MinAtar)                      */
    int32_t n_step;      /* n-step return length, 1..PRISM_MAX_NSTEP                          */
    float *obs;          /* [capacity][obs_elems]                                             */
    float *succ_obs;     /* [capacity][obs_elems] observation of the slot's immediate successor */
    float *reward;       /* [capacity]                                                        */
    int32_t *action;     /* [capacity]                                                        */
    uint8_t *flags;      /* [capacity] PRISM_FLAG_*                                           */
    int32_t *link;       /* [capacity] ring slot of the stored successor, -1 if none yet      */
    int32_t *back;       /* [capacity] ring slot of the predecessor that links here, or -1    */
    float *tree;         /* [2*tree_capacity][2] {sum, min} of every node interleaved (both children
                            of a node are one aligned 16-byte load); NULL for uniform replay   */
    float *per_state;    /* [4] {running max raw priority, last p_sum, last p_min, unused}    */
    int32_t *status;     /* [1] sticky PRISM_STATUS_* bits                                    */
    double gammas[PRISM_MAX_NSTEP + 1]; /* gamma**k as Python float64 (timestep_buffer.py:17)  */
} prism_replay_desc;

/* trees := identity (0 / FLT_MAX), per_state := {1,0,0,0}, status := 0, link/back := -1.
 * torchrl PrioritizedSampler._init. */
int prism_replay_init(const prism_replay_desc *rp, prism_stream_t stream);

/* TimestepBuffer.extend (timestep_buffer.py:32-33) for n completed timesteps, applied in order
 * i = 0..n-1: overwrite ring slot slots[i] (detaching any predecessor that linked to the old row),
 * store the row, link prev_slot[i] -> slots[i] when prev_slot[i] >= 0, and give the slot the
 * sampler's default priority (max_priority + eps) ** alpha in both trees
 * (torchrl PrioritizedSampler.extend).  All array arguments are device staging buffers. */
int prism_replay_insert(const prism_replay_desc *rp, int32_t n, const int32_t *slots,
                        const float *obs, const float *succ_obs, const float *reward,
                        const int32_t *action, const uint8_t *flags, const int32_t *prev_slot,
                        float alpha, float eps, prism_stream_t stream);

/* PrioritizedSampler.sample (called at timestep_buffer.py:37): p_sum/p_min = query(0,size);
 * index[i] = min(scan_lower_bound(mass[i]), size-1); weight[i] = (leaf/p_min) ** -beta.
 * mass == NULL: masses are drawn on the device, U(0,p_sum) in float64 narrowed to fp32, from
 * Philox4x32-10 keyed by (seed, offset); else `mass` holds `batch` fp32 values (parity mode: the
 * reference draws them with NumPy's global RNG, which cannot be reproduced on the device).
 * `size` = number of stored items (len(storage)).  Indices are int64 as in torchrl. */
int prism_per_sample(const prism_replay_desc *rp, int64_t size, int32_t batch, const float *mass,
                     uint64_t seed, uint64_t offset, float beta, int64_t *out_index,
                     float *out_weight, prism_stream_t stream);

/* torchrl RandomSampler (uniform replay, exp_buffer_factory.py:30-33): index ~ U{0..size-1}. */
int prism_uniform_sample(int64_t size, int32_t batch, uint64_t seed, uint64_t offset,
                         int64_t *out_index, prism_stream_t stream);

/* _compute_n_step + _timesteps_to_batch (timestep_buffer.py:79-238) for the sampled slots:
 * walks <= n_step links per sample (fp64 accumulate, fp32 store) and gathers rows.
 * out_nonterminal is torch.bool storage (1 byte), out_action int64. */
int prism_replay_gather(const prism_replay_desc *rp, const int64_t *index, int32_t batch,
                        float *out_obs, float *out_next_obs, float *out_reward,
                        uint8_t *out_nonterminal, float *out_gamma, int64_t *out_action,
                        prism_stream_t stream);

/* PrioritizedSampler.update_priority (timestep_buffer.py:53-54 <- learner.py:120):
 * max_priority = max(max_priority, max(priority)); leaf = (priority + eps) ** alpha in both
 * trees; duplicate indices resolve as the sequential loop would (last occurrence wins); every
 * ancestor is recomputed as op(left, right).  take_abs != 0 applies |.| first (learner.py:120
 * passes new_per_weights.abs()). */
int prism_per_update(const prism_replay_desc *rp, const int64_t *index, const float *priority,
                     int32_t batch, float alpha, float eps, int32_t take_abs,
                     prism_stream_t stream);

/* Recompute every internal node from the leaves (after the caller wrote leaves in bulk: restoring a
 * saved sampler, torchrl PrioritizedSampler.loads, or a synthetic pre-fill). */
int prism_per_rebuild(const prism_replay_desc *rp, prism_stream_t stream);

/* out[0] = sum.query(0,size), out[1] = min.query(0,size) (device floats). */
int prism_per_query(const prism_replay_desc *rp, int64_t size, float *out2, prism_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * TD update: CompositeModel.get_losses + Agent._update_without_cuda_graph
 * (prism/agents/models/composite_model.py:94-144, iqn_model.py:48-201, q_ensemble.py:44-92,
 *  prism/agents/agent.py:53-79), fp32 throughout.
 * ------------------------------------------------------------------------------------------ */
/* prism_model_dims.squish_fn (/root/reference/prism/agents/squish_functions.py:4-18) */
#define PRISM_SQUISH_NONE 0
#define PRISM_SQUISH_OBS_LOOK_FURTHER 1   /* sign(x) (sqrt(|x| + 1) - 1) + 0.01 x  and its inverse   */
#define PRISM_SQUISH_SYMLOG 2             /* sign(x) log(|x| + 1)  /  sign(x) (exp(|x|) - 1)          */

typedef struct prism_model_dims {
    int32_t in_channels;   /* C; observations are (10,10,C) NHWC                              */
    int32_t n_actions;     /* A <= 16                                                         */
    int32_t embed_dim;     /* 1024 = 16*8*8 (minatar_cnn_model.py:14)                         */
    int32_t use_iqn;
    int32_t n_basis;       /* 64                                                              */
    int32_t iqn_layers;    /* iqn_quantile_model_layers (1)                                   */
    int32_t iqn_width;     /* iqn_quantile_model_feature_dim (128)                            */
    int32_t n_tau;         /* current-state quantile samples T                                */
    int32_t n_tau_next;    /* next-state quantile samples T'                                  */
    int32_t use_layer_norm;
    int32_t n_heads;       /* 0 none, 1 DQN, >1 IDS ensemble                                  */
    int32_t head_layers;   /* linear layers per head (1 or 2)                                 */
    int32_t head_width;    /* hidden width of a 2-layer head (128)                            */
    int32_t has_target;    /* target network present                                          */
    int32_t double_q;
    int32_t propagate_grad; /* IQN gradients reach the embedding (model_factory.py:87)        */
    float huber_k;
    float dist_loss_weight;
    float q_loss_weight;
    float theil_coef;      /* ids_ensemble_variation_coef                                     */
    int32_t squish_fn;     /* PRISM_SQUISH_*: loss_squish_fn_id (model_factory.py:16-23): the TD target is
                            * squish(r + gamma' * unsquish(z_next)) (iqn_model.py:141-148, q_ensemble.py:77-82) */
} prism_model_dims;

/* Offsets (in floats) of each tensor inside the flat parameter buffer, which is laid out in
 * model.parameters() order.  -1 = tensor absent. */
typedef struct prism_param_offsets {
    int64_t n_params;
    int64_t conv_w, conv_b;
    int64_t phi_w, phi_b;
    int64_t iqn_ln1_g, iqn_ln1_b, iqn_w1, iqn_b1;
    int64_t iqn_ln2_g, iqn_ln2_b, iqn_w2, iqn_b2;
    int64_t head_base;    /* first float of head 0                                           */
    int64_t head_stride;  /* floats per head                                                 */
    int64_t h_ln1_g, h_ln1_b, h_w1, h_b1; /* relative to the head's base                     */
    int64_t h_ln2_g, h_ln2_b, h_w2, h_b2; /* (1-layer head: only h_ln1_* (if LN) and h_w1/h_b1) */
} prism_param_offsets;

typedef struct prism_adam_hyper {
    double lr, beta1, beta2, eps; /* doubles: torch evaluates the bias corrections in Python floats */
    float max_grad_norm;
    float grad_scale;      /* multiplied into the gradient before the norm (1/world_size)    */
} prism_adam_hyper;

typedef struct prism_learner_desc {
    prism_model_dims dims;
    prism_param_offsets off;
    int32_t batch;            /* B                                                           */
    int32_t embed_done;       /* != 0: prism_step_front already produced the embeddings for this batch */
    /* parameters and optimizer state, flat fp32 [n_params] */
    float *params;
    const float *target_params; /* NULL when !has_target                                     */
    float *grads;             /* out: dL/dparams (unclipped, unscaled)                        */
    float *adam_m;
    float *adam_v;
    int64_t *adam_step;       /* [1] device step counter, incremented by prism_learner_clip_adam */
    /* minibatch (the static batch of timestep_buffer.py:84-104) */
    const float *obs;         /* [B][10][10][C]                                               */
    const float *next_obs;
    const float *reward;      /* [B] n-step return                                            */
    const uint8_t *nonterminal; /* [B] torch.bool                                            */
    const float *gamma;       /* [B] gamma ** m                                               */
    const int64_t *action;    /* [B]                                                          */
    const float *per_weights; /* [B] or NULL (== 1, learner.py:109)                           */
    /* quantile samples, tau-major rows (row = t*B + b, iqn_model.py:70).  NULL => drawn in-kernel
     * from Philox(seed, offset) and written to tau_out. */
    const float *tau_cur;     /* [T*B]                                                        */
    const float *tau_next_online; /* [T'*B] used when !has_target or double_q                 */
    const float *tau_next_target; /* [T'*B] used when has_target                              */
    float *tau_out;           /* [3][max(T,T')*B] or NULL                                     */
    uint64_t seed, offset;
    uint64_t *rng_counters;   /* optional device [3] {PER draws, tau draws, acting draws} added to the immediate
                                 offsets; [0], [1] are advanced by prism_step_back, [2] by prism_act_forward (lets a
                                 captured hipGraph draw fresh numbers on every replay); NULL = immediate offsets only */
    /* Optional: when fused_replay is set, prism_per_update(fused_index, |out_td|) rides along in the
     * learner's launches instead of being a workgroup of its own in prism_step_back: one more workgroup
     * of prism_learner_fwd_bwd's last launch prepares it (|TD|^alpha, ranking, duplicate resolution --
     * the TD errors are final by then) and, for batches up to 256, prism_step_back walks the tree levels
     * using the sibling values prism_step_front recorded (larger batches: all of it in fwd_bwd).
     * With fused_replay set, prism_learner_fwd_bwd MUST be followed by prism_step_back on the same
     * stream before the tree is read again.  Results are identical to the stand-alone update. */
    const struct prism_replay_desc *fused_replay;   /* host pointer or NULL */
    const int64_t *fused_index;                     /* [B] sampled slots */
    float fused_alpha, fused_eps;
    /* != 0 (fused hot path, single GPU only): prism_learner_fwd_bwd leaves the gradient partials unreduced and
     * prism_step_back reduces them, clips and applies Adam in ONE launch (a grid barrier stands where the launch
     * boundary was).  ld->grads is complete only after prism_step_back then, so nothing may sit between the two
     * calls -- leave it 0 when an all-reduce does (hyper.grad_scale != 1 ignores it).  The library falls back to the
     * separate launches by itself when the launch would not be resident at once or no priority writeback rides along
     * (fused_replay unset: nothing to hide behind the barrier).  Results are bit-identical.
     * The residency proof (occupancy of the launched instantiation x CUs of the current device) assumes the process has the
     * GPU to itself: with other processes' kernels on the device leave it 0 (or set PRISM_NO_FUSED_TAIL=1).  A barrier that
     * is not through after 100 ms is abandoned: the workgroup sets PRISM_WS_STATUS_BARRIER_TIMEOUT in the workspace's
     * status word and skips its update instead of spinning for ever. */
    int32_t fuse_tail;
    /* prism_act_forward only.  PRISM_ACT_WEIGHTS_CURRENT: the stream-packed weight copies and LayerNorm helper vectors in
     * the workspace were built from `params` as they are now (an earlier prism_act_forward without this flag ran since the
     * parameters last changed): the call skips rebuilding them.  0 is always correct. */
    int32_t act_flags;
    /* How the forward GEMMs (quantile embedding, trunk, Q-head first layers) are multiplied: PRISM_GEMM_FP32 = the exact
     * fp32 MFMA chain; PRISM_GEMM_BF16X3 = every fp32 operand as the sum of three bf16 pieces, the six leading piece
     * products on the bf16 matrix pipe with fp32 accumulation (what is dropped is of the size of one fp32 rounding; same rms
     * error against float64 as the fp32 chain, profiles/r03_split_bf16_ubench.txt); 0 = library default (environment
     * PRISM_GEMM=fp32|bf16x3 overrides the default).  Both forms cover hidden widths 128 and 256. */
    int32_t gemm_mode;
    /* outputs */
    float *out_dist_loss;     /* [B] or NULL  (Agent._static_distribution_loss)               */
    float *out_q_loss;        /* [B] or NULL  (Agent._static_q_loss)                          */
    float *out_td;            /* [B]          (td_errors, composite_model.py:135-142)         */
    float *out_scalars;       /* [8] {total loss, mean dl*w, mean ql*w, grad norm, theil, clip coef, -, -} */
    float *dbg_z;             /* optional [ (T+T')*B*A ] quantile estimates (tests) or NULL    */
    void *dbg_stamps;         /* diagnostics only (PRISM_DBG & 8): [4096][64] uint64 shader-clock stamps, else NULL */
    void *workspace;          /* >= prism_learner_workspace_bytes(), zero-filled once.  32-bit word 2 of it is the
                                 "target set packed" flag: whoever writes target_params (prism_sync_target's caller,
                                 a checkpoint load) stores 0 there, on the stream; the library sets it.  32-bit word 7
                                 is a sticky status word: bit 0 = a workgroup of the fused tail abandoned its grid
                                 barrier (PRISM_WS_STATUS_BARRIER_TIMEOUT): the step that set it is incomplete     */
    size_t workspace_bytes;
    prism_adam_hyper hyper;
    /* Optional: uint32[4] in pinned, device-mapped HOST memory (hipHostMalloc; NULL = none), zeroed by the caller.  When
     * the library raises sticky bit k of the workspace's status word it also stores 1 into word k here (a plain
     * system-scope store: no PCIe atomics needed), so the host can poll for an abandoned barrier / collective on every
     * step WITHOUT synchronising with the device (the workspace word itself costs a device-to-host copy to read). */
    uint32_t *host_status;
} prism_learner_desc;

/* host: bytes of scratch prism_learner_fwd_bwd needs for (dims, batch). 0 on unsupported dims. */
size_t prism_learner_workspace_bytes(const prism_model_dims *dims, int32_t batch);

/* host: PRISM_OK if the kernels cover this (dims, batch) combination. */
int prism_learner_supported(const prism_model_dims *dims, int32_t batch);

/* get_losses + backward: fills out_*, grads (complete dL/dparams of
 * mean(dl*w) + mean(ql*w), agent.py:58-64,71-72) — the RCCL all-reduce of `grads` sits between
 * this call and prism_learner_clip_adam in data-parallel runs. */
int prism_learner_fwd_bwd(const prism_learner_desc *ld, prism_stream_t stream);

/* clip_grad_norm_(max_grad_norm) + Adam step (agent.py:73-74; torch.optim.Adam defaults,
 * agent_factory.py:44-47) over the flat buffers; adam_step += 1. */
int prism_learner_clip_adam(const prism_learner_desc *ld, prism_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused hot path — one Learner iteration (prism/learner.py:95-125) in four launches on one GPU (IQN; one or
 * two more with Q heads), five when an all-reduce sits in the middle:
 *   prism_step_front                        PER sample + n-step gather + conv embed (+ LayerNorm helpers)
 *   prism_learner_fwd_bwd (embed_done = 1)  quantile forward tiles with the loss in their tail, column-sliced
 *                                           backward [, post: gradient reduction + first half of the writeback]
 *   [RCCL all-reduce of ld->grads when data-parallel]
 *   prism_step_back                         [fuse_tail: gradient reduction, then behind a grid barrier] clip + Adam,
 *                                           priority writeback with |td|, RNG counters
 * Results are identical to per_sample -> replay_gather -> fwd_bwd -> clip_adam -> per_update.
 * ------------------------------------------------------------------------------------------ */
/* TimestepBuffer.sample (timestep_buffer.py:35-51) fused with the embedding of both observations.
 * Writes the minibatch into ld->obs/next_obs/reward/nonterminal/gamma/action (the static batch),
 * out_index [B] int64 and out_weight [B] (set ld->per_weights = out_weight to use them).
 * rp->tree == NULL selects uniform replay. */
int prism_step_front(const prism_learner_desc *ld, const prism_replay_desc *rp, int64_t size,
                     const float *mass, uint64_t seed, uint64_t offset, float beta,
                     int64_t *out_index, float *out_weight, prism_stream_t stream);

/* prism_learner_clip_adam + prism_per_update(index, |ld->out_td|) in one launch. */
int prism_step_back(const prism_learner_desc *ld, const prism_replay_desc *rp, const int64_t *index,
                    float alpha, float eps, prism_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Acting forward -- Agent.forward (prism/agents/agent.py:31-41): CompositeModel.forward(for_action=True)
 * (composite_model.py:51-70; iqn_model.py:61-87 with n_quantile_samples_per_action rows per observation;
 * q_ensemble.py:44-48) on the learner's parameters and workspace, with the same forward tiles as the update.
 *   obs      [n][10][10][C], 1 <= n <= ld->batch (Q heads: the 16-padded n as well); device memory, or pinned
 *            device-mapped host memory read in place (no host-to-device copy in front of a small batch)
 *   tau_in   [n_tau*n] tau-major (row = t*n + b, iqn_model.py:66-70) or NULL -> Philox(seed, offset)
 *   out_z    [ceil16(n*n_tau)][A]  quantile estimates, SAMPLE-major (row = b*n_tau + t); reference layout =
 *            view(n, n_tau, A).permute(1, 0, 2)
 *   out_q    [heads][ceil16(n)][A] ensemble estimates; reference layout = [:, :n].permute(1, 2, 0)
 *            (the single-Linear DQN head, dqn_n_model_layers = 1: heads = 1, one workgroup per observation)
 * Either output may be NULL when the model has no such part.
 * Quantile draws: Philox(seed, offset [+ ld->rng_counters[2] when ld->rng_counters is set and tau_in is NULL; the call then
 * advances that device word by n * n_tau: capture the call into a hipGraph and every replay draws fresh samples]). */
int prism_act_forward(const prism_learner_desc *ld, const float *obs, int32_t n, int32_t n_tau,
                      const float *tau_in, uint64_t seed, uint64_t offset, float *out_z, float *out_q,
                      prism_stream_t stream);

/* IDSActionSelector.generate_action_probs + select_action for ids_use_random_samples = False
 * (prism/agents/action_selectors.py:125-176): scores [n][A] = regret^2 / information gain, action [n] = argmin.
 * z / q: the buffers prism_act_forward filled.  out_aux (optional) [n][4][A]: ensemble mean, ensemble spread
 * (torch.std), return-distribution variance, information gain -- what the selector logs.  out_action_host (optional): a
 * second destination of the actions -- pinned, device-mapped host memory, so that the caller needs no device-to-host copy.
 * unsquish_fn (PRISM_SQUISH_*): the selector's unsquish function, applied to both estimate arrays first (:128-130). */
int prism_ids_select(const float *z, const float *q, int32_t n, int32_t n_pad, int32_t n_tau, int32_t n_actions,
                     int32_t n_heads, float lmbda, float epsilon, float rho_lower_bound, int32_t unsquish_fn,
                     float *out_scores, float *out_aux, int64_t *out_action, int64_t *out_action_host, prism_stream_t stream);

/* GreedyActionSelector.generate_action_probs + select_action (prism/agents/action_selectors.py:70-83; also the greedy
 * branch of EGreedyActionSelector, :24-45): action [n] = argmax_a mean(q_estimates[:, a, :]).  q != NULL: mean over the
 * n_heads ensemble estimates prism_act_forward left in out_q; q == NULL: mean over the n_tau quantile estimates in z
 * (composite_model.py:66-68, models without Q heads).  out_mean (optional) [n][A]; out_action_host as for prism_ids_select. */
int prism_greedy_select(const float *z, const float *q, int32_t n, int32_t n_pad, int32_t n_tau, int32_t n_actions,
                        int32_t n_heads, int64_t *out_action, float *out_mean, int64_t *out_action_host,
                        prism_stream_t stream);

/* Agent.sync_target_model (agent.py:149-152): target := online (device-to-device copy). */
int prism_sync_target(float *target_params, const float *params, int64_t n_params,
                      prism_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Direct all-reduce (sum) of the flat gradient over peer-mapped buffers -- the data-parallel step's one exchange
 * (the reference has no collective: prism/learner.py:95-125 is a single process).  An
 * alternative to the RCCL all-reduce between prism_learner_fwd_bwd and prism_step_back for the 0.8 - 6 MB messages of
 * this path on a fully connected xGMI node: two shots, every GPU pulling 1/world of the buffer from all peers at once.
 *   bufs[s]   rank s's gradient buffer as mapped into THIS process (own buffer at [rank]); the host maps the peers
 *             (hipIpcGetMemHandle / hipIpcOpenMemHandle, or the sharing of its tensor library)
 *   flags[s]  rank s's flag array, PRISM_MAX_PEERS + 2 uint32 zeroed once, mapped likewise (use_flags only)
 * prism_direct_reduce_scatter: slice `rank` of the own buffer := sum over s = 0..world-1 (in that order) of slice `rank`
 * of bufs[s];  prism_direct_all_gather: every other slice := its owner's.  Afterwards all ranks hold bit-identical sums.
 * Synchronisation: all ranks must have finished writing before the reduce-scatter, finished the reduce-scatter before
 * the all-gather, and finished the all-gather before anyone overwrites its buffer.  use_flags = 0: the CALLER provides
 * these three barriers (the only legal form when ranks share a device); use_flags = 1 (one device per rank): one-workgroup
 * kernels signal and poll the flag arrays on the stream (phase numbers come from a counter in the flag array itself, so the
 * calls may be captured into a hipGraph and replayed).
 *
 * Memory types (every word another DEVICE reads or writes while a kernel runs):
 *   flags[s]  MUST be fine-grained / uncached device memory -- prism_direct_flags_alloc() -- not an ordinary hipMalloc
 *             (coarse-grained) allocation: a store from a peer into coarse-grained memory is only guaranteed visible at
 *             kernel boundaries, a kernel polling it may spin on a stale L2 line.  (RCCL keeps its flags the same way.)
 *   bufs[s]   ordinary (coarse-grained) device memory: written by kernels that END before the flag announcing them is
 *             stored (the end of a kernel writes its L2 back at system scope), read by kernels that START after the wait
 *             kernel has ended (the start of a kernel invalidates).  No kernel ever reads a peer's gradient word that is
 *             written while it runs.
 * Failure: a wait that is not through after `wait_seconds` (0 = 30 s; RCCL would wait for ever, a rank that saves a
 * checkpoint or captures a graph can stall for seconds) gives up: it sets flags[rank][PRISM_MAX_PEERS] (sticky), ORs
 * PRISM_WS_STATUS_COLLECTIVE_TIMEOUT into *poison and into *host_status when given, and every later kernel of the
 * collective returns at once, as do prism_step_back / prism_learner_clip_adam of a learner whose workspace status word is
 * `poison`: a timed-out step applies NO update (never a partial sum) and the host sees the bit at its next poll.
 * ------------------------------------------------------------------------------------------ */
#define PRISM_MAX_PEERS 8
#define PRISM_DIRECT_FLAG_WORDS (PRISM_MAX_PEERS + 2)
#define PRISM_IPC_HANDLE_BYTES 64
typedef struct prism_direct_desc {
    int32_t world, rank;
    float *bufs[PRISM_MAX_PEERS];
    uint32_t *flags[PRISM_MAX_PEERS];
    int64_t n;             /* floats */
    uint32_t *poison;      /* optional device word (the learner workspace's status word): see Failure above */
    uint32_t *host_status; /* optional pinned, device-mapped host words [4]: see prism_learner_desc.host_status */
    double wait_seconds;   /* bound of one flag wait; 0 = 30 s */
} prism_direct_desc;
int prism_direct_reduce_scatter(const prism_direct_desc *d, int32_t use_flags, prism_stream_t stream);
int prism_direct_all_gather(const prism_direct_desc *d, int32_t use_flags, prism_stream_t stream);
/* One synchronisation point of the protocol as its own launch: phase 1 (gradients written), 2 (slices reduced) or 3
 * (slices gathered; its wait advances the all-reduce count), what = 1 announce to the peers, 2 wait for them, 3 both
 * (what use_flags = 1 enqueues).  For callers that pace the phases from the host -- announce, host barrier, wait, the
 * use_flags = 0 kernel -- e.g. ranks that SHARE a device, where a kernel that spins would keep the peer it waits for
 * off the device: the flag addressing and the phase arithmetic run exactly as on the step's path. */
int prism_direct_phase(const prism_direct_desc *d, int32_t phase, int32_t what, prism_stream_t stream);

/* host, synchronous (set-up time, never on the step's path).  The flag array of one rank: PRISM_DIRECT_FLAG_WORDS uint32
 * of UNCACHED (fine-grained) device memory on the current device, zeroed, plus its inter-process handle
 * (hipIpcGetMemHandle; PRISM_IPC_HANDLE_BYTES bytes the host ships to the peers however it likes). */
int prism_direct_flags_alloc(uint32_t **flags_out, void *ipc_handle_out);
int prism_direct_flags_free(uint32_t *flags);
/* map a peer's flag array into this process (hipIpcOpenMemHandle) / unmap it */
int prism_direct_flags_open(const void *ipc_handle, uint32_t **flags_out);
int prism_direct_flags_close(uint32_t *flags);
/* hipDeviceEnablePeerAccess(peer_device) from the current device, explicitly (idempotent; PRISM_ERR_HIP when the two
 * devices have no peer path) -- a mapped peer buffer is only dereferenceable after this. */
int prism_direct_enable_peer(int32_t peer_device);
/* copy the PRISM_DIRECT_FLAG_WORDS words of the own flag array to the host (synchronises with `stream`) */
int prism_direct_flags_read(const uint32_t *flags, uint32_t *host_out, prism_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optional per-kernel timing with HIP events on the launch stream (used by bench.py for the
 * roofline figure; off by default, adds two event records per instrumented launch).
 * Kernel ids: 0 embed, 1 fwd_tile, 2 loss, 3 bwd, 4 post, 5 front, 6 clip_adam, 7 per_sample,
 * 8 gather, 9 per_update, 10 back, 11 q_loss, 12 q_bwd, 13 tail (post + clip + Adam in one launch).
 * ------------------------------------------------------------------------------------------ */
#define PRISM_N_KERNEL_IDS 16
/* on != 0: instrument every launch issued by the calling thread until switched off */
int prism_profile_enable(int on);
/* host: synchronises the recorded events, ADDS elapsed milliseconds and launch counts per kernel id
 * into ms_sum[PRISM_N_KERNEL_IDS] / count[PRISM_N_KERNEL_IDS], and recycles the events. */
int prism_profile_collect(double *ms_sum, int64_t *count);
const char *prism_profile_kernel_name(int id);

#ifdef __cplusplus
}
#endif
#endif /* PRISM_HIP_H */
