// Replay ring + prioritized sum/min segment trees resident in HBM (gfx950).
//
// Semantics restated in oracle/per_oracle.c; reference call sites:
//   /root/reference/prism/experience/timestep_buffer.py:32-54,79-238 and the torchrl
//   PrioritizedSampler it wraps (prism/factory/exp_buffer_factory.py:22-28).
//
// All of this is latency-bound integer/fp32 pointer chasing (17-21 dependent tree levels), so the
// kernels are shaped for few launches, wave-parallelism ACROSS samples, LDS-cached tree tops and
// coalesced 16-byte row copies — not for MFMA.
#include <float.h>

#include "common.h"

namespace prism {

constexpr int TOP_LEVELS = 11;                 // nodes [1, 2^11) of the sum tree cached in LDS (8 KB)
constexpr int TOP_NODES = 1 << TOP_LEVELS;

template <bool MIN>
__device__ __forceinline__ float tree_op(float a, float b) {
    if (MIN) return a < b ? a : b;
    return a + b;
}

// SegmentTree::Query(0, r) restated so that all node loads are issued in parallel (one lane per
// (level, side)) and then folded by one lane in exactly the sequential order.
// Must be called by all threads of the block; needs blockDim.x >= 128; `scratch` holds 128 floats.
template <bool MIN>
__device__ float block_tree_query(const float *__restrict__ v, int64_t cap, int64_t tree_size, int64_t r_in,
                                  float *scratch) {
    const float ident = MIN ? FLT_MAX : 0.0f;
    if (r_in >= tree_size) return v[1];
    const int t = threadIdx.x;
    if (t < 128) {
        const int level = t >> 1, side = t & 1;
        int64_t l = cap, r = r_in | cap;
        float val = ident;
        bool live = true;
        for (int i = 0; i < level && live; ++i) {
            if (!(l < r)) { live = false; break; }
            if (l & 1) ++l;
            if (r & 1) --r;
            l >>= 1;
            r >>= 1;
        }
        if (live && l < r) {
            if (side == 0) {
                if (l & 1) val = v[l];
            } else {
                if (r & 1) val = v[r - 1];
            }
        }
        scratch[t] = val;
    }
    __syncthreads();
    float ret = ident;
    // identity entries fold as no-ops for min; for the sum they add +0.0f which is exact
    // (ret is never -0.0f here), so folding all 128 slots in order equals the sequential walk.
    for (int i = 0; i < 128; ++i) ret = tree_op<MIN>(ret, scratch[i]);
    __syncthreads();
    return ret;
}

__global__ void replay_init_kernel(prism_replay_desc rp) {
    const int64_t n = 2 * rp.tree_capacity;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (rp.sum_tree) {
        for (int64_t i = tid; i < n; i += stride) {
            rp.sum_tree[i] = 0.0f;
            rp.min_tree[i] = FLT_MAX;
        }
    }
    for (int64_t i = tid; i < rp.capacity; i += stride) {
        rp.link[i] = -1;
        rp.back[i] = -1;
        rp.flags[i] = 0;
    }
    if (tid == 0) {
        rp.per_state[0] = 1.0f;
        rp.per_state[1] = 0.0f;
        rp.per_state[2] = 0.0f;
        rp.per_state[3] = 0.0f;
        rp.status[0] = 0;
    }
}

// ---- priority write + ancestor recompute for up to `n` leaves, one workgroup ----------------
// Duplicates: the sequential reference loop leaves the LAST occurrence's value in the leaf, and
// every ancestor equals op(left, right) of the final children.  We write only the winning
// occurrence per leaf, then recompute ancestors level-synchronously; threads sharing an ancestor
// compute the same value from the same finished children, so the races are benign.
__device__ void block_tree_write(const prism_replay_desc &rp, const int64_t *s_idx, const float *s_val,
                                 const uint8_t *s_win, int n) {
    const int64_t cap = rp.tree_capacity;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (s_win[i]) {
            const int64_t leaf = s_idx[i] | cap;
            rp.sum_tree[leaf] = s_val[i];
            rp.min_tree[leaf] = s_val[i];
        }
    }
    __syncthreads();
    for (int64_t width = cap >> 1, shift = 1; width >= 1; width >>= 1, ++shift) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t p = (s_idx[i] | cap) >> shift;
            const float a = rp.sum_tree[2 * p], b = rp.sum_tree[2 * p + 1];
            const float c = rp.min_tree[2 * p], d = rp.min_tree[2 * p + 1];
            rp.sum_tree[p] = a + b;
            rp.min_tree[p] = c < d ? c : d;
        }
        __syncthreads();
    }
}

constexpr int UPD_MAX = 4096;  // leaves per pass of the single-workgroup writer

__global__ __launch_bounds__(1024) void per_update_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                         const float *__restrict__ priority, int n,
                                                         float alpha, float eps, int take_abs) {
    __shared__ int64_t s_idx[UPD_MAX];
    __shared__ float s_val[UPD_MAX];
    __shared__ uint8_t s_win[UPD_MAX];
    __shared__ float s_red[16];
    // running max of the raw priorities (torchrl tracks it before the +eps, **alpha)
    float m = -FLT_MAX;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float p = priority[i];
        if (take_abs) p = fabsf(p);
        m = fmaxf(m, p);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = rp.per_state[0];
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) mm = fmaxf(mm, s_red[w]);
        rp.per_state[0] = mm;
    }
    for (int base = 0; base < n; base += UPD_MAX) {
        const int cnt = min(UPD_MAX, n - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            float p = priority[base + i];
            if (take_abs) p = fabsf(p);
            s_idx[i] = index[base + i];
            s_val[i] = pow_alpha(p + eps, alpha);
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int64_t me = s_idx[i];
            bool win = true;
            for (int j = i + 1; j < cnt; ++j) win &= (s_idx[j] != me);
            s_win[i] = win;
        }
        __syncthreads();
        block_tree_write(rp, s_idx, s_val, s_win, cnt);
    }
}

// rows of an insert batch -> ring slots (any number of workgroups)
__global__ void replay_store_rows_kernel(prism_replay_desc rp, int n, const int32_t *__restrict__ slots,
                                         const float *__restrict__ obs, const float *__restrict__ succ_obs,
                                         const float *__restrict__ reward, const int32_t *__restrict__ action,
                                         const uint8_t *__restrict__ flags) {
    const int O = rp.obs_elems;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int64_t s = slots[i];
        float *d0 = rp.obs + s * O, *d1 = rp.succ_obs + s * O;
        const float *s0 = obs + (int64_t)i * O, *s1 = succ_obs + (int64_t)i * O;
        if ((O & 3) == 0) {
            for (int k = threadIdx.x; k < O / 4; k += blockDim.x) {
                reinterpret_cast<float4 *>(d0)[k] = reinterpret_cast<const float4 *>(s0)[k];
                reinterpret_cast<float4 *>(d1)[k] = reinterpret_cast<const float4 *>(s1)[k];
            }
        } else {
            for (int k = threadIdx.x; k < O; k += blockDim.x) {
                d0[k] = s0[k];
                d1[k] = s1[k];
            }
        }
        if (threadIdx.x == 0) {
            rp.reward[s] = reward[i];
            rp.action[s] = action[i];
            rp.flags[s] = flags[i];
        }
    }
}

// links (sequential, as the collector would have produced them) + default priority
__global__ __launch_bounds__(1024) void replay_link_kernel(prism_replay_desc rp, int n,
                                                          const int32_t *__restrict__ slots,
                                                          const int32_t *__restrict__ prev_slot, float alpha,
                                                          float eps) {
    __shared__ int64_t s_idx[UPD_MAX];
    __shared__ float s_val[UPD_MAX];
    __shared__ uint8_t s_win[UPD_MAX];
    if (threadIdx.x == 0) {
        for (int i = 0; i < n; ++i) {
            const int32_t s = slots[i];
            const int32_t b = rp.back[s];
            if (b >= 0 && rp.link[b] == s) rp.link[b] = -1;   // predecessor of the overwritten row
            const int32_t q = rp.link[s];
            if (q >= 0 && rp.back[q] == s) rp.back[q] = -1;   // successor of the overwritten row
            rp.link[s] = -1;
            rp.back[s] = -1;
            const int32_t p = prev_slot[i];
            if (p >= 0) {
                rp.link[p] = s;
                rp.back[s] = p;
            }
        }
    }
    if (!rp.sum_tree) return;
    const float prio = pow_alpha(rp.per_state[0] + eps, alpha);
    for (int base = 0; base < n; base += UPD_MAX) {
        const int cnt = min(UPD_MAX, n - base);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            s_idx[i] = slots[base + i];
            s_val[i] = prio;
            s_win[i] = 1;
        }
        __syncthreads();
        block_tree_write(rp, s_idx, s_val, s_win, cnt);
    }
}

// internal nodes of one level from their children (used after leaves were written in bulk)
__global__ void per_rebuild_level_kernel(prism_replay_desc rp, int64_t first, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int64_t p = first + i;
        const float a = rp.sum_tree[2 * p], b = rp.sum_tree[2 * p + 1];
        const float c = rp.min_tree[2 * p], d = rp.min_tree[2 * p + 1];
        rp.sum_tree[p] = a + b;
        rp.min_tree[p] = c < d ? c : d;
    }
}

// ---- PER sample: one lane per sample, tree top in LDS --------------------------------------
__global__ __launch_bounds__(256) void per_sample_kernel(prism_replay_desc rp, int64_t size, int batch,
                                                        const float *__restrict__ mass_in, uint64_t seed,
                                                        uint64_t offset, float beta,
                                                        int64_t *__restrict__ out_index,
                                                        float *__restrict__ out_weight) {
    __shared__ float s_top[TOP_NODES];
    __shared__ float s_scratch[128];
    const int64_t cap = rp.tree_capacity;
    const int64_t top = cap < TOP_NODES ? cap : TOP_NODES;   // nodes [1, top) are internal or leaves of a tiny tree
    for (int i = threadIdx.x; i < top; i += blockDim.x) s_top[i] = i ? rp.sum_tree[i] : 0.0f;
    const float p_sum = block_tree_query<false>(rp.sum_tree, cap, rp.capacity, size, s_scratch);
    const float p_min = block_tree_query<true>(rp.min_tree, cap, rp.capacity, size, s_scratch);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        rp.per_state[1] = p_sum;
        rp.per_state[2] = p_min;
        int st = 0;
        if (!(p_sum > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PSUM;
        if (!(p_min > 0.0f)) st |= PRISM_STATUS_NONPOSITIVE_PMIN;
        if (st) atomicOr(rp.status, st);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    float mass;
    if (mass_in) {
        mass = mass_in[i];
    } else {
        uint32_t r[4];
        Philox ph(seed);
        ph(offset + (uint64_t)i, 0x5045524dull /* "PERM" */, r);
        mass = (float)(0.0 + ((double)p_sum - 0.0) * u64_to_unit_double(r[0], r[1]));
    }
    int64_t idx;
    if (mass > s_top[1]) {
        idx = rp.capacity;
    } else {
        int64_t node = 1;
        float v = mass;
        while (node < cap) {
            node <<= 1;
            const float lv = node < top ? s_top[node] : rp.sum_tree[node];
            if (v > lv) {
                v -= lv;
                node |= 1;
            }
        }
        idx = node ^ cap;
    }
    if (idx > size - 1) idx = size - 1;
    out_index[i] = idx;
    const float leaf = rp.sum_tree[idx | cap];
    out_weight[i] = pow_neg_beta(leaf / p_min, beta);
}

__global__ void per_query_kernel(prism_replay_desc rp, int64_t size, float *out2) {
    __shared__ float s_scratch[128];
    const float p_sum = block_tree_query<false>(rp.sum_tree, rp.tree_capacity, rp.capacity, size, s_scratch);
    const float p_min = block_tree_query<true>(rp.min_tree, rp.tree_capacity, rp.capacity, size, s_scratch);
    if (threadIdx.x == 0) {
        out2[0] = p_sum;
        out2[1] = p_min;
    }
}

__global__ void uniform_sample_kernel(int64_t size, int batch, uint64_t seed, uint64_t offset,
                                      int64_t *__restrict__ out_index) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch) return;
    uint32_t r[4];
    Philox ph(seed);
    ph(offset + (uint64_t)i, 0x554e4946ull /* "UNIF" */, r);
    const uint64_t x = ((uint64_t)r[0] << 32) | r[1];
    out_index[i] = (int64_t)__umul64hi(x, (uint64_t)size);
}

// ---- n-step walk + row gather: one workgroup (128 lanes) per sampled slot ----------------------
__global__ __launch_bounds__(128) void replay_gather_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                           int batch, float *__restrict__ out_obs,
                                                           float *__restrict__ out_next_obs,
                                                           float *__restrict__ out_reward,
                                                           uint8_t *__restrict__ out_nonterminal,
                                                           float *__restrict__ out_gamma,
                                                           int64_t *__restrict__ out_action) {
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int64_t first = index[b];
    // wave-uniform walk (every lane runs it; the loads are uniform so they cost one request each)
    int64_t cur = first;
    double ret = 0.0, gamma = 1.0;
    for (int k = 0; k < rp.n_step; ++k) {
        ret += (double)rp.reward[cur] * rp.gammas[k];
        gamma = rp.gammas[k + 1];
        const bool incomplete = (k != rp.n_step - 1);
        const uint32_t f = rp.flags[cur];
        if ((f & PRISM_FLAG_HAS_NEXT) && !(f & PRISM_FLAG_TRUNC) && incomplete) {
            const int32_t nx = rp.link[cur];
            if (nx >= 0)
                cur = nx;
            else
                break;
        } else {
            break;
        }
    }
    const uint32_t fl = rp.flags[cur];
    const int O = rp.obs_elems;
    const float *src_obs = rp.obs + first * O;
    const float *src_next = (fl & PRISM_FLAG_HAS_NEXT) ? rp.succ_obs + cur * O : src_obs;
    float *d0 = out_obs + (int64_t)b * O, *d1 = out_next_obs + (int64_t)b * O;
    if ((O & 3) == 0) {
        for (int k = threadIdx.x; k < O / 4; k += blockDim.x) {
            const float4 a = reinterpret_cast<const float4 *>(src_obs)[k];
            const float4 c = reinterpret_cast<const float4 *>(src_next)[k];
            reinterpret_cast<float4 *>(d0)[k] = a;
            reinterpret_cast<float4 *>(d1)[k] = c;
        }
    } else {
        for (int k = threadIdx.x; k < O; k += blockDim.x) {
            d0[k] = src_obs[k];
            d1[k] = src_next[k];
        }
    }
    if (threadIdx.x == 0) {
        out_reward[b] = (float)ret;
        out_nonterminal[b] = (fl & PRISM_FLAG_DONE) ? 0 : 1;
        out_gamma[b] = (float)gamma;
        out_action[b] = (int64_t)rp.action[first];
    }
}

}  // namespace prism

using namespace prism;

static int check_replay(const prism_replay_desc *rp, bool need_tree) {
    PRISM_CHECK_ARG(rp != nullptr, "null descriptor");
    PRISM_CHECK_ARG(rp->capacity > 0 && rp->capacity < (1ll << 31), "capacity out of range");
    int64_t cap = 1;
    while (cap <= rp->capacity) cap <<= 1;
    PRISM_CHECK_ARG(rp->tree_capacity == cap, "tree_capacity must be the smallest power of two > capacity");
    PRISM_CHECK_ARG(rp->obs_elems > 0, "obs_elems");
    PRISM_CHECK_ARG(rp->n_step >= 1 && rp->n_step <= PRISM_MAX_NSTEP, "n_step out of range");
    PRISM_CHECK_ARG(rp->obs && rp->succ_obs && rp->reward && rp->action && rp->flags && rp->link && rp->back,
                    "null ring array");
    PRISM_CHECK_ARG(rp->per_state && rp->status, "null per_state/status");
    if (need_tree) PRISM_CHECK_ARG(rp->sum_tree && rp->min_tree, "prioritized call on a ring without trees");
    PRISM_CHECK_ARG((rp->sum_tree == nullptr) == (rp->min_tree == nullptr), "sum/min tree must come together");
    return PRISM_OK;
}

extern "C" int prism_replay_init(const prism_replay_desc *rp, prism_stream_t stream) {
    int rc = check_replay(rp, false);
    if (rc) return rc;
    hipLaunchKernelGGL(replay_init_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, *rp);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_replay_insert(const prism_replay_desc *rp, int32_t n, const int32_t *slots, const float *obs,
                                   const float *succ_obs, const float *reward, const int32_t *action,
                                   const uint8_t *flags, const int32_t *prev_slot, float alpha, float eps,
                                   prism_stream_t stream) {
    int rc = check_replay(rp, false);
    if (rc) return rc;
    if (n == 0) return PRISM_OK;
    PRISM_CHECK_ARG(n > 0 && n <= rp->capacity, "n must be in (0, capacity]");
    PRISM_CHECK_ARG(slots && obs && succ_obs && reward && action && flags && prev_slot, "null staging array");
    const int grid = n < 2048 ? n : 2048;
    hipLaunchKernelGGL(replay_store_rows_kernel, dim3(grid), dim3(128), 0, (hipStream_t)stream, *rp, n, slots, obs,
                       succ_obs, reward, action, flags);
    PRISM_CHECK_LAUNCH();
    const int threads = n >= 1024 ? 1024 : ((n + 127) / 128) * 128;
    hipLaunchKernelGGL(replay_link_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, *rp, n, slots, prev_slot,
                       alpha, eps);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_per_rebuild(const prism_replay_desc *rp, prism_stream_t stream) {
    int rc = check_replay(rp, true);
    if (rc) return rc;
    for (int64_t first = rp->tree_capacity >> 1; first >= 1; first >>= 1) {
        const int64_t count = first;
        int grid = (int)((count + 255) / 256);
        if (grid > 2048) grid = 2048;
        hipLaunchKernelGGL(per_rebuild_level_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, *rp, first, count);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

extern "C" int prism_per_sample(const prism_replay_desc *rp, int64_t size, int32_t batch, const float *mass,
                                uint64_t seed, uint64_t offset, float beta, int64_t *out_index, float *out_weight,
                                prism_stream_t stream) {
    int rc = check_replay(rp, true);
    if (rc) return rc;
    PRISM_CHECK_ARG(size > 0 && size <= rp->capacity, "size must be in (0, capacity] (empty storage)");
    PRISM_CHECK_ARG(batch > 0 && out_index && out_weight, "batch/out");
    {
        ProfileScope ps_(K_PER_SAMPLE, (hipStream_t)stream);
        hipLaunchKernelGGL(per_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, *rp, size,
                           batch, mass, seed, offset, beta, out_index, out_weight);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

extern "C" int prism_uniform_sample(int64_t size, int32_t batch, uint64_t seed, uint64_t offset, int64_t *out_index,
                                    prism_stream_t stream) {
    PRISM_CHECK_ARG(size > 0 && batch > 0 && out_index, "size/batch/out");
    hipLaunchKernelGGL(uniform_sample_kernel, dim3((batch + 255) / 256), dim3(256), 0, (hipStream_t)stream, size,
                       batch, seed, offset, out_index);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}

extern "C" int prism_replay_gather(const prism_replay_desc *rp, const int64_t *index, int32_t batch, float *out_obs,
                                   float *out_next_obs, float *out_reward, uint8_t *out_nonterminal, float *out_gamma,
                                   int64_t *out_action, prism_stream_t stream) {
    int rc = check_replay(rp, false);
    if (rc) return rc;
    PRISM_CHECK_ARG(batch > 0 && index && out_obs && out_next_obs && out_reward && out_nonterminal && out_gamma &&
                        out_action,
                    "batch/out");
    {
        ProfileScope ps_(K_GATHER, (hipStream_t)stream);
        hipLaunchKernelGGL(replay_gather_kernel, dim3(batch), dim3(128), 0, (hipStream_t)stream, *rp, index, batch,
                           out_obs, out_next_obs, out_reward, out_nonterminal, out_gamma, out_action);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

extern "C" int prism_per_update(const prism_replay_desc *rp, const int64_t *index, const float *priority,
                                int32_t batch, float alpha, float eps, int32_t take_abs, prism_stream_t stream) {
    int rc = check_replay(rp, true);
    if (rc) return rc;
    PRISM_CHECK_ARG(batch > 0 && index && priority, "batch/index/priority");
    const int threads = batch >= 1024 ? 1024 : ((batch + 127) / 128) * 128;
    {
        ProfileScope ps_(K_PER_UPDATE, (hipStream_t)stream);
        hipLaunchKernelGGL(per_update_kernel, dim3(1), dim3(threads), 0, (hipStream_t)stream, *rp, index, priority, batch,
                           alpha, eps, take_abs);
        PRISM_CHECK_LAUNCH();
    }
    return PRISM_OK;
}

extern "C" int prism_per_query(const prism_replay_desc *rp, int64_t size, float *out2, prism_stream_t stream) {
    int rc = check_replay(rp, true);
    if (rc) return rc;
    PRISM_CHECK_ARG(size > 0 && size <= rp->capacity && out2, "size/out");
    hipLaunchKernelGGL(per_query_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, *rp, size, out2);
    PRISM_CHECK_LAUNCH();
    return PRISM_OK;
}
