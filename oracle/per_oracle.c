/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported, linked or called by the product path
 * (the prism_amd package).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it.
 *
 * CPU restatement (plain C, scalar, single thread) of the prioritized-replay arithmetic on the
 * reference's sample -> collate -> priority-writeback path.
 *
 * PARITY UNPINNED for the segment-tree part: the reference delegates it to the third-party package
 * `torchrl` (PrioritizedReplayBuffer / PrioritizedSampler with C++ Sum/MinSegmentTreeFp32; call
 * sites /root/reference/prism/factory/exp_buffer_factory.py:22-28,
 * prism/experience/timestep_buffer.py:37,54, prism/learner.py:100-107,120).  torchrl is not
 * vendored in /root/reference, has no pinned version there (no requirements / lock file) and is
 * not installed in the build container, and the reference holds no golden vectors for sampled
 * indices or tree values.  What follows restates torchrl's published algorithm:
 *   - binary segment tree, `capacity` = smallest power of two strictly greater than `size`,
 *     2*capacity nodes, leaf i at node (i | capacity), root at node 1;
 *   - update: write the leaf, then recompute each ancestor as op(node, sibling) up to the root;
 *   - query(l, r): whole-range shortcut returns the root, else the bottom-up half-open walk;
 *   - scan_lower_bound(v): top-down descent with fp32 subtract-as-you-go;
 *   - sampler: mass ~ U(0, p_sum) drawn in float64 and narrowed to fp32, index clamped to
 *     len-1, weight = (p_i / p_min) ** -beta, writeback p = (|td| + eps) ** alpha, running max of
 *     the raw priorities, new items get (max + eps) ** alpha (max starts at 1).
 *
 * The n-step / collate part restates code the reference owns and IS pinned by golden vectors
 * generated from the live reference (tests/golden/nstep_*.npz):
 *   prism/experience/timestep_buffer.py:198-238 (_compute_n_step) and :129-178 (collate rules).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int64_t size;      /* logical number of leaves                  */
    int64_t capacity;  /* power of two strictly greater than size   */
    int is_min;        /* 0: sum tree, 1: min tree                  */
    float *values;     /* 2 * capacity nodes                        */
} seg_tree;

/* torch.pow with a scalar exponent special-cases 0.5 -> sqrt and -0.5 -> 1/sqrt (ATen
 * pow_tensor_scalar_optimized_kernel); both are correctly rounded, which is what makes the
 * default alpha = beta = 0.5 reproducible bit-for-bit on any IEEE machine. */
static float pow_alpha(float x, float alpha) {
    if (alpha == 0.5f) return sqrtf(x);
    if (alpha == 1.0f) return x;
    return powf(x, alpha);
}
static float pow_neg_beta(float x, float beta) {
    if (beta == 0.5f) return 1.0f / sqrtf(x);
    if (beta == 1.0f) return 1.0f / x;
    if (beta == 0.0f) return 1.0f;
    return powf(x, -beta);
}

static float seg_op(const seg_tree *t, float a, float b) {
    if (t->is_min) return a < b ? a : b; /* std::min(a, b) */
    return a + b;
}

seg_tree *oracle_tree_create(int64_t size, int is_min) {
    seg_tree *t = (seg_tree *)malloc(sizeof(seg_tree));
    t->size = size;
    t->is_min = is_min;
    for (t->capacity = 1; t->capacity <= size; t->capacity <<= 1) {
    }
    t->values = (float *)malloc(sizeof(float) * 2 * (size_t)t->capacity);
    float ident = is_min ? FLT_MAX : 0.0f;
    for (int64_t i = 0; i < 2 * t->capacity; ++i) t->values[i] = ident;
    return t;
}

void oracle_tree_destroy(seg_tree *t) {
    if (t) {
        free(t->values);
        free(t);
    }
}

int64_t oracle_tree_capacity(const seg_tree *t) { return t->capacity; }
float *oracle_tree_values(seg_tree *t) { return t->values; }

void oracle_tree_update(seg_tree *t, int64_t index, float value) {
    index |= t->capacity;
    for (t->values[index] = value; index > 1; index >>= 1) {
        t->values[index >> 1] = seg_op(t, t->values[index], t->values[index ^ 1]);
    }
}

/* sequential batched update: duplicates resolve as "last occurrence wins" */
void oracle_tree_update_batch(seg_tree *t, const int64_t *index, const float *value, int64_t n) {
    for (int64_t i = 0; i < n; ++i) oracle_tree_update(t, index[i], value[i]);
}

float oracle_tree_get(const seg_tree *t, int64_t index) { return t->values[index | t->capacity]; }

float oracle_tree_query(const seg_tree *t, int64_t l, int64_t r) {
    if (l <= 0 && r >= t->size) return t->values[1];
    float ret = t->is_min ? FLT_MAX : 0.0f;
    l |= t->capacity;
    r |= t->capacity;
    while (l < r) {
        if (l & 1) ret = seg_op(t, ret, t->values[l++]);
        if (r & 1) ret = seg_op(t, ret, t->values[--r]);
        l >>= 1;
        r >>= 1;
    }
    return ret;
}

int64_t oracle_tree_scan_lower_bound(const seg_tree *t, float value) {
    if (value > t->values[1]) return t->size;
    int64_t index = 1;
    float current = value;
    while (index < t->capacity) {
        index <<= 1;
        float lvalue = t->values[index];
        if (current > lvalue) {
            current -= lvalue;
            index |= 1;
        }
    }
    return index ^ t->capacity;
}

/*
 * One PrioritizedSampler.sample() given the masses (already narrowed to fp32 by the caller):
 * index = min(scan_lower_bound(mass), len - 1); weight = (sum_tree[index] / p_min) ** -beta.
 * The weight is evaluated the way torch does it on fp32 tensors: fp32 divide, then fp32 pow.
 */
void oracle_per_sample(const seg_tree *sum_t, const seg_tree *min_t, int64_t len, const float *mass,
                       int64_t n, float beta, int64_t *out_index, float *out_weight,
                       float *out_psum_pmin) {
    float p_sum = oracle_tree_query(sum_t, 0, len);
    float p_min = oracle_tree_query(min_t, 0, len);
    if (out_psum_pmin) {
        out_psum_pmin[0] = p_sum;
        out_psum_pmin[1] = p_min;
    }
    for (int64_t i = 0; i < n; ++i) {
        int64_t idx = oracle_tree_scan_lower_bound(sum_t, mass[i]);
        if (idx > len - 1) idx = len - 1;
        out_index[i] = idx;
        float w = oracle_tree_get(sum_t, idx) / p_min;
        out_weight[i] = pow_neg_beta(w, beta);
    }
}

/*
 * Version variants of PrioritizedSampler.sample in torchrl's history (the reference pins no version, so both are
 * restated and tests/test_oracle_replay.py states what each one changes):
 *   weight_form 0  torch.pow(p / p_min, -beta) on fp32 tensors (newer torchrl): what oracle_per_sample evaluates --
 *                  ATen special-cases the exponent -0.5 to 1 / sqrt(x), two correctly rounded fp32 operations;
 *   weight_form 1  np.power(p / p_min, -beta) on fp32 NumPy arrays (older torchrl): the C library's powf(x, -beta), one
 *                  rounding (glibc's powf is correctly rounded in almost every case).
 *   query_full != 0  p_sum / p_min taken over query(0, max_capacity) instead of query(0, len): the same nodes once the
 *                  storage is full (len == size -- the benched state); while it fills, the whole-range shortcut returns the
 *                  root, whose fp32 sum may round differently from the bottom-up walk over [0, len).
 * The sampled INDICES depend on neither the weight form nor -- once the storage is full -- the query range.
 */
void oracle_per_sample_variant(const seg_tree *sum_t, const seg_tree *min_t, int64_t len, const float *mass,
                               int64_t n, float beta, int weight_form, int query_full, int64_t *out_index,
                               float *out_weight, float *out_psum_pmin) {
    float p_sum = oracle_tree_query(sum_t, 0, query_full ? sum_t->size : len);
    float p_min = oracle_tree_query(min_t, 0, query_full ? min_t->size : len);
    if (out_psum_pmin) {
        out_psum_pmin[0] = p_sum;
        out_psum_pmin[1] = p_min;
    }
    for (int64_t i = 0; i < n; ++i) {
        int64_t idx = oracle_tree_scan_lower_bound(sum_t, mass[i]);
        if (idx > len - 1) idx = len - 1;
        out_index[i] = idx;
        float w = oracle_tree_get(sum_t, idx) / p_min;
        out_weight[i] = weight_form == 1 ? powf(w, -beta) : pow_neg_beta(w, beta);
    }
}

/*
 * PrioritizedSampler.update_priority: running max over the raw priorities, then
 * p = (priority + eps) ** alpha written to both trees, sequentially (last duplicate wins).
 * Returns the new running max.
 */
float oracle_default_priority(float max_priority, float alpha, float eps) {
    return pow_alpha(max_priority + eps, alpha);
}

float oracle_per_update(seg_tree *sum_t, seg_tree *min_t, const int64_t *index, const float *priority,
                        int64_t n, float alpha, float eps, float max_priority) {
    for (int64_t i = 0; i < n; ++i)
        if (priority[i] > max_priority) max_priority = priority[i];
    for (int64_t i = 0; i < n; ++i) {
        float p = pow_alpha(priority[i] + eps, alpha);
        oracle_tree_update(sum_t, index[i], p);
        oracle_tree_update(min_t, index[i], p);
    }
    return max_priority;
}

/*
 * n-step return + collate over a slot-indexed structure-of-arrays replay ring.
 *
 * The reference walks Python `Timestep` objects linked by weakrefs
 * (timestep_buffer.py:198-238); this restates the same walk over arrays:
 *   reward[s], flags[s] (bit0 done, bit1 truncated, bit2 "next is not None"), link[s] = ring slot
 *   of the successor if that successor has itself been completed and stored (its reward is not
 *   None), else -1; succ_obs[s] = observation of the immediate successor node (the real next
 *   timestep or the strong-referenced truncation node).
 * gammas[k] = gamma ** k as Python float64 (timestep_buffer.py:17).  The return accumulates in
 * float64 and is narrowed to fp32 when stored into the batch (timestep_buffer.py:175).
 *
 * Collate rules (timestep_buffer.py:145-178): if the last node of the walk has no `next`
 * (terminal), next_obs = obs of the SAMPLED timestep; else next_obs = that node's successor obs.
 * nonterminal = 1 - done(last); gamma = gammas[m], m = number of rewards summed.
 */
#define ORACLE_FLAG_DONE 1
#define ORACLE_FLAG_TRUNC 2
#define ORACLE_FLAG_HAS_NEXT 4

void oracle_nstep_gather(const float *obs, const float *succ_obs, const float *reward,
                         const int32_t *action, const uint8_t *flags, const int32_t *link,
                         int64_t obs_elems, int32_t n_step, const double *gammas,
                         const int64_t *index, int64_t n, float *out_obs, float *out_next_obs,
                         float *out_reward, uint8_t *out_nonterminal, float *out_gamma,
                         int64_t *out_action, uint8_t *out_needs_n_step) {
    for (int64_t b = 0; b < n; ++b) {
        int64_t first = index[b];
        int64_t cur = first;
        double ret = 0.0;
        double gamma = 1.0;
        int incomplete = 0;
        for (int32_t k = 0; k < n_step; ++k) {
            ret += (double)reward[cur] * gammas[k];
            gamma = gammas[k + 1];
            incomplete = (k != n_step - 1);
            int has_next = (flags[cur] & ORACLE_FLAG_HAS_NEXT) != 0;
            int trunc = (flags[cur] & ORACLE_FLAG_TRUNC) != 0;
            if (has_next && !trunc && incomplete) {
                if (link[cur] >= 0)
                    cur = link[cur];
                else
                    break;
            } else {
                break;
            }
        }
        int done = (flags[cur] & ORACLE_FLAG_DONE) != 0;
        int trunc = (flags[cur] & ORACLE_FLAG_TRUNC) != 0;
        int has_next = (flags[cur] & ORACLE_FLAG_HAS_NEXT) != 0;
        memcpy(out_obs + b * obs_elems, obs + first * obs_elems, sizeof(float) * (size_t)obs_elems);
        if (has_next)
            memcpy(out_next_obs + b * obs_elems, succ_obs + cur * obs_elems,
                   sizeof(float) * (size_t)obs_elems);
        else
            memcpy(out_next_obs + b * obs_elems, obs + first * obs_elems,
                   sizeof(float) * (size_t)obs_elems);
        out_reward[b] = (float)ret;
        out_nonterminal[b] = (uint8_t)(1 - done);
        out_gamma[b] = (float)gamma;
        out_action[b] = (int64_t)action[first];
        if (out_needs_n_step) out_needs_n_step[b] = (uint8_t)(incomplete && !done && !trunc);
    }
}
