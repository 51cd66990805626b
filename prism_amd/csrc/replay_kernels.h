// Device code of the replay ring + PER trees (kernels and the block-level routines the fused step
// kernels reuse).  Semantics restated in oracle/per_oracle.c; reference call sites:
//   /root/reference/prism/experience/timestep_buffer.py:32-54,79-238 and the torchrl
//   PrioritizedSampler it wraps (prism/factory/exp_buffer_factory.py:22-28).
//
// All of this is latency-bound integer/fp32 pointer chasing (17-21 dependent tree levels), so the
// kernels are shaped for few launches, wave-parallelism ACROSS samples, LDS-cached tree tops and
// coalesced 16-byte row copies — not for MFMA.
#pragma once
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

static __global__ void replay_init_kernel(prism_replay_desc rp) {
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
// The lower levels go through global memory (one barrier + L2 round trip each); the top
// WTOP_LEVELS levels of both trees are recomputed in LDS and written back once.
constexpr int WTOP_LEVELS = 11;
constexpr int WTOP = 1 << WTOP_LEVELS;          // nodes [1, WTOP) live in LDS, children up to 2*WTOP

__device__ void block_tree_write(const prism_replay_desc &rp, const int32_t *s_idx, const float *s_val,
                                 const uint8_t *s_win, int n, float *s_tsum, float *s_tmin) {
    const int64_t cap = rp.tree_capacity;
    const int64_t top = cap < WTOP ? cap : WTOP;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (s_win[i]) {
            const int64_t leaf = (int64_t)s_idx[i] | cap;
            rp.sum_tree[leaf] = s_val[i];
            rp.min_tree[leaf] = s_val[i];
        }
    }
    __syncthreads();
    int64_t shift = 1;
    for (; (cap >> shift) >= top; ++shift) {      // parents still at or below node `top`: via global
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t p = ((int64_t)s_idx[i] | cap) >> shift;
            const float a = rp.sum_tree[2 * p], b = rp.sum_tree[2 * p + 1];
            const float c = rp.min_tree[2 * p], d = rp.min_tree[2 * p + 1];
            rp.sum_tree[p] = a + b;
            rp.min_tree[p] = c < d ? c : d;
        }
        __syncthreads();
    }
    // nodes [1, 2*top): the children of the first LDS level were finished above (or are leaves)
#pragma unroll 8
    for (int64_t i = threadIdx.x; i < 2 * top; i += blockDim.x) {   // independent loads: keep many in flight
        s_tsum[i] = rp.sum_tree[i];
        s_tmin[i] = rp.min_tree[i];
    }
    __syncthreads();
    const int64_t first_lds_shift = shift;
    for (; (cap >> shift) >= 1; ++shift) {
        for (int i = threadIdx.x; i < n; i += blockDim.x) {
            const int64_t p = ((int64_t)s_idx[i] | cap) >> shift;
            const float a = s_tsum[2 * p], b = s_tsum[2 * p + 1];
            const float c = s_tmin[2 * p], d = s_tmin[2 * p + 1];
            s_tsum[p] = a + b;
            s_tmin[p] = c < d ? c : d;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        for (int64_t sh = first_lds_shift; (cap >> sh) >= 1; ++sh) {
            const int64_t p = ((int64_t)s_idx[i] | cap) >> sh;
            rp.sum_tree[p] = s_tsum[p];
            rp.min_tree[p] = s_tmin[p];
        }
    }
}

constexpr int UPD_MAX = 1024;  // leaves per pass of the single-workgroup writer
constexpr int HT_BITS = 11, HT_SIZE = 1 << HT_BITS;   // duplicate-detection hash table (load factor <= 0.5)

// whole PrioritizedSampler.update_priority for one batch, executed by ONE workgroup (any size)
// `lds`: PER_UPDATE_LDS_BYTES of 16-byte aligned LDS supplied by the calling kernel (so that kernels
// hosting this routine as one role among others can alias it with their own scratch).
constexpr int PER_UPDATE_LDS_BYTES = UPD_MAX * 4 + UPD_MAX * 4 + UPD_MAX + 64 + 2 * (2 * WTOP) * 4 + 2 * HT_SIZE * 4;
__device__ void per_update_block(const prism_replay_desc &rp, const int64_t *__restrict__ index,
                                 const float *__restrict__ priority, int n, float alpha, float eps, int take_abs,
                                 char *lds) {
    int32_t *s_idx = reinterpret_cast<int32_t *>(lds);
    float *s_val = reinterpret_cast<float *>(lds + UPD_MAX * 4);
    float *s_red = reinterpret_cast<float *>(lds + UPD_MAX * 8);
    float *s_tsum = s_red + 16, *s_tmin = s_tsum + 2 * WTOP;
    int32_t *s_hkey = reinterpret_cast<int32_t *>(s_tmin + 2 * WTOP), *s_hpos = s_hkey + HT_SIZE;
    uint8_t *s_win = reinterpret_cast<uint8_t *>(s_hpos + HT_SIZE);
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
        const int cnt4 = (cnt + 3) & ~3;
        for (int i = threadIdx.x; i < cnt4; i += blockDim.x) {
            if (i < cnt) {
                float p = priority[base + i];
                if (take_abs) p = fabsf(p);
                s_idx[i] = (int32_t)index[base + i];
                s_val[i] = pow_alpha(p + eps, alpha);
            } else {
                s_idx[i] = -1;
            }
        }
        __syncthreads();
        // last occurrence wins: LDS hash table keyed by leaf, value = highest batch position seen
        for (int i = threadIdx.x; i < HT_SIZE; i += blockDim.x) {
            s_hkey[i] = -1;
            s_hpos[i] = -1;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int32_t me = s_idx[i];
            uint32_t slot = ((uint32_t)me * 2654435761u) >> (32 - HT_BITS);
            for (;;) {
                const int32_t prev = atomicCAS(&s_hkey[slot], -1, me);
                if (prev == -1 || prev == me) {
                    atomicMax(&s_hpos[slot], i);
                    break;
                }
                slot = (slot + 1) & (HT_SIZE - 1);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            const int32_t me = s_idx[i];
            uint32_t slot = ((uint32_t)me * 2654435761u) >> (32 - HT_BITS);
            while (s_hkey[slot] != me) slot = (slot + 1) & (HT_SIZE - 1);
            s_win[i] = (s_hpos[slot] == i);
        }
        __syncthreads();
        block_tree_write(rp, s_idx, s_val, s_win, cnt, s_tsum, s_tmin);
    }
}

static __global__ __launch_bounds__(1024) void per_update_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                         const float *__restrict__ priority, int n,
                                                         float alpha, float eps, int take_abs) {
    __shared__ __attribute__((aligned(16))) char s_pool[PER_UPDATE_LDS_BYTES];
    per_update_block(rp, index, priority, n, alpha, eps, take_abs, s_pool);
}

// rows of an insert batch -> ring slots (any number of workgroups)
static __global__ void replay_store_rows_kernel(prism_replay_desc rp, int n, const int32_t *__restrict__ slots,
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
static __global__ __launch_bounds__(1024) void replay_link_kernel(prism_replay_desc rp, int n,
                                                          const int32_t *__restrict__ slots,
                                                          const int32_t *__restrict__ prev_slot, float alpha,
                                                          float eps) {
    __shared__ __attribute__((aligned(16))) int32_t s_idx[UPD_MAX];
    __shared__ float s_val[UPD_MAX];
    __shared__ uint8_t s_win[UPD_MAX];
    __shared__ float s_tsum[2 * WTOP], s_tmin[2 * WTOP];
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
            s_idx[i] = (int32_t)slots[base + i];
            s_val[i] = prio;
            s_win[i] = 1;
        }
        __syncthreads();
        block_tree_write(rp, s_idx, s_val, s_win, cnt, s_tsum, s_tmin);
    }
}

// internal nodes of one level from their children (used after leaves were written in bulk)
static __global__ void per_rebuild_level_kernel(prism_replay_desc rp, int64_t first, int64_t count) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const int64_t p = first + i;
        const float a = rp.sum_tree[2 * p], b = rp.sum_tree[2 * p + 1];
        const float c = rp.min_tree[2 * p], d = rp.min_tree[2 * p + 1];
        rp.sum_tree[p] = a + b;
        rp.min_tree[p] = c < d ? c : d;
    }
}

// SumSegmentTree::ScanLowerBound with fp32 subtract-as-you-go; nodes below `top` come from LDS
__device__ __forceinline__ int64_t tree_descend(const prism_replay_desc &rp, const float *s_top, int64_t top,
                                                float mass) {
    const int64_t cap = rp.tree_capacity;
    if (mass > s_top[1]) return rp.capacity;
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
    return node ^ cap;
}

// _compute_n_step (timestep_buffer.py:198-238) over the slot arrays; wave-uniform
struct NStepResult {
    int64_t last;      // slot where the walk stopped
    double ret;        // sum_k reward_k * gamma**k (float64 accumulate)
    double gamma;      // gamma ** (number of rewards summed)
    uint32_t flags;    // flags of the last slot
};
__device__ __forceinline__ NStepResult nstep_walk(const prism_replay_desc &rp, int64_t first) {
    NStepResult r;
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
    r.last = cur;
    r.ret = ret;
    r.gamma = gamma;
    r.flags = rp.flags[cur];
    return r;
}

// ---- PER sample: one lane per sample, tree top in LDS --------------------------------------
static __global__ __launch_bounds__(256) void per_sample_kernel(prism_replay_desc rp, int64_t size, int batch,
                                                        const float *__restrict__ mass_in, uint64_t seed,
                                                        uint64_t offset, float beta,
                                                        int64_t *__restrict__ out_index,
                                                        float *__restrict__ out_weight) {
    __shared__ float s_top[TOP_NODES];
    __shared__ float s_scratch[128];
    const int64_t cap = rp.tree_capacity;
    const int64_t top = cap < TOP_NODES ? cap : TOP_NODES;   // nodes [1, top) are internal or leaves of a tiny tree
#pragma unroll 8
    for (int i = threadIdx.x; i < top; i += blockDim.x) s_top[i] = rp.sum_tree[i];
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
    int64_t idx = tree_descend(rp, s_top, top, mass);
    if (idx > size - 1) idx = size - 1;
    out_index[i] = idx;
    const float leaf = rp.sum_tree[idx | cap];
    out_weight[i] = pow_neg_beta(leaf / p_min, beta);
}

static __global__ void per_query_kernel(prism_replay_desc rp, int64_t size, float *out2) {
    __shared__ float s_scratch[128];
    const float p_sum = block_tree_query<false>(rp.sum_tree, rp.tree_capacity, rp.capacity, size, s_scratch);
    const float p_min = block_tree_query<true>(rp.min_tree, rp.tree_capacity, rp.capacity, size, s_scratch);
    if (threadIdx.x == 0) {
        out2[0] = p_sum;
        out2[1] = p_min;
    }
}

static __global__ void uniform_sample_kernel(int64_t size, int batch, uint64_t seed, uint64_t offset,
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
static __global__ __launch_bounds__(128) void replay_gather_kernel(prism_replay_desc rp, const int64_t *__restrict__ index,
                                                           int batch, float *__restrict__ out_obs,
                                                           float *__restrict__ out_next_obs,
                                                           float *__restrict__ out_reward,
                                                           uint8_t *__restrict__ out_nonterminal,
                                                           float *__restrict__ out_gamma,
                                                           int64_t *__restrict__ out_action) {
    const int b = blockIdx.x;
    if (b >= batch) return;
    const int64_t first = index[b];
    const NStepResult ns = nstep_walk(rp, first);   // wave-uniform: every lane runs it
    const int64_t cur = ns.last;
    const uint32_t fl = ns.flags;
    const double ret = ns.ret, gamma = ns.gamma;
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
