// Acting-time pieces: the information-directed-sampling score of every action
// (/root/reference/prism/agents/action_selectors.py:125-176) from the quantile estimates Z [n][T][A] and the
// ensemble estimates Q [heads][n_pad][A] the forward tiles leave behind (fwd_kernels.h, kinds 0 and 1).
// One wave per observation: lanes split the T quantile samples; per action two passes (mean, then squared
// deviations) as torch.var / torch.std do, unbiased (n - 1).
#pragma once
#include "common.h"

namespace prism {

struct IdsArgs {
    const float *z;      // [n][T][A] sample-major quantile estimates
    const float *q;      // [heads][n_pad][A]
    int n, n_pad, T, A, heads;
    float lmbda, eps, rho_lb;
    float *scores;       // [n][A] IDS scores (regret^2 / information gain)
    float *aux;          // [n][4][A]: ensemble mean | ensemble "variance" (torch.std) | return-distribution variance | information gain
    int64_t *action;     // [n] argmin of the scores (first minimum)
    int64_t *action2;    // optional second copy of the actions (pinned host memory: the caller needs no device-to-host copy)
    int stage;           // the observation's T x A estimates fit the launch's dynamic LDS: staged there in one round of loads
    int unsquish;        // PRISM_SQUISH_*: the selector's unsquish function, applied to every estimate as it is read
};

constexpr int ACT_THREADS = 256;
constexpr int ACT_STAGE_MAX_FLOATS = 12288;      // 48 KB
// The T x A quantile estimates of observation b, contiguous in memory, into LDS with every thread of the workgroup: ONE round
// of coalesced loads where the selectors used to walk them from global memory in T dependent rounds of A-lane loads
// (greedy_select_kernel: 19 us for T = 200, ids_score_kernel: 9.7 us -- longer than the forward tiles they follow).
__device__ __forceinline__ const float *act_stage(const float *z, int64_t b, int T, int A, float *s_z, bool stage) {
    const float *src = z + b * (int64_t)T * A;
    if (!stage) return src;
    const int n = T * A;
    if (((n & 3) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
        for (int i = threadIdx.x; i < n / 4; i += ACT_THREADS) reinterpret_cast<float4 *>(s_z)[i] = reinterpret_cast<const float4 *>(src)[i];
    } else {
        for (int i = threadIdx.x; i < n; i += ACT_THREADS) s_z[i] = src[i];
    }
    __syncthreads();
    return s_z;
}

__global__ __launch_bounds__(ACT_THREADS) void ids_score_kernel(IdsArgs k) {
    extern __shared__ __attribute__((aligned(16))) float s_act[];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int A = k.A, T = k.T, Hd = k.heads;
    const float *zb = act_stage(k.z, b, T, A, s_act, k.stage != 0);
    if (lane >= 64) return;
    // ensemble statistics: lane = action
    float mean = 0.f, spread = 0.f;
    if (lane < A) {
        float s = 0.f;
        for (int h = 0; h < Hd; ++h) s += unsquish_value(k.unsquish, k.q[((int64_t)h * k.n_pad + b) * A + lane]);
        mean = s / (float)Hd;
        float v = 0.f;
        for (int h = 0; h < Hd; ++h) {
            const float d = unsquish_value(k.unsquish, k.q[((int64_t)h * k.n_pad + b) * A + lane]) - mean;
            v += d * d;
        }
        spread = sqrtf(v / (float)(Hd > 1 ? Hd - 1 : 1));          // q_estimates.std(dim=-1): called "variance" there
    }
    const float sd = sqrtf(spread);                                 // torch.sqrt(variance)
    const float hi = lane < A ? mean + k.lmbda * sd : -INFINITY;
    const float upper = wave_max(hi);
    const float regret = upper - (mean - k.lmbda * sd);
    const float regret_sq = regret * regret;
    // return-distribution variance per action over the T quantile samples (unbiased)
    float var_mine = 0.f;
    for (int a = 0; a < A; ++a) {
        float s = 0.f;
        for (int t = lane; t < T; t += 64) s += unsquish_value(k.unsquish, zb[t * A + a]);
        const float m = wave_sum(s) / (float)T;
        float v = 0.f;
        for (int t = lane; t < T; t += 64) {
            const float d = unsquish_value(k.unsquish, zb[t * A + a]) - m;
            v += d * d;
        }
        const float var = wave_sum(v) / (float)(T > 1 ? T - 1 : 1);
        if (lane == a) var_mine = var;
    }
    const float var_mean = wave_sum(lane < A ? var_mine : 0.f) / (float)A;
    const float rho = fmaxf(var_mine / (k.eps + var_mean), k.rho_lb);
    const float gain = logf(1.0f + spread / rho) + k.eps;
    const float score = regret_sq / gain;
    if (lane < A) {
        k.scores[(int64_t)b * A + lane] = score;
        if (k.aux) {
            float *x = k.aux + (int64_t)b * 4 * A;
            x[lane] = mean;
            x[A + lane] = spread;
            x[2 * A + lane] = var_mine;
            x[3 * A + lane] = gain;
        }
    }
    // argmin, first minimum wins (torch.argmin)
    int best = 0;
    float bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(score), 0));
    for (int a = 1; a < A; ++a) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(score), a));
        if (v < bv) {
            bv = v;
            best = a;
        }
    }
    if (lane == 0) {
        k.action[b] = best;
        if (k.action2) k.action2[b] = best;
    }
}

// Greedy arg-max of the mean action values (GreedyActionSelector, /root/reference/prism/agents/action_selectors.py:70-83:
// q_estimates.mean(dim=-1).argmax(dim=-1)): the ensemble mean over the heads when the model has Q heads, else the mean of
// the T quantile samples (composite_model.py:66-68: q_estimates = return_distribution.mean(dim=0)).  One wave per
// observation, lane = action, sums in index order, first maximum wins (torch.argmax).
struct GreedyArgs {
    const float *z;      // [n][T][A] or NULL
    const float *q;      // [heads][n_pad][A] or NULL
    int n, n_pad, T, A, heads;
    int64_t *action;     // [n]
    float *mean;         // optional [n][A]
    int64_t *action2;    // optional second copy of the actions (pinned host memory)
    int stage;           // as IdsArgs.stage
};

__global__ __launch_bounds__(ACT_THREADS) void greedy_select_kernel(GreedyArgs k) {
    extern __shared__ __attribute__((aligned(16))) float s_act[];
    const int b = blockIdx.x, lane = threadIdx.x, A = k.A;
    const float *zb = k.q ? nullptr : act_stage(k.z, b, k.T, A, s_act, k.stage != 0);
    if (lane >= 64) return;
    float m = -INFINITY;
    if (lane < A) {
        float s = 0.f;
        if (k.q) {
            for (int h = 0; h < k.heads; ++h) s += k.q[((int64_t)h * k.n_pad + b) * A + lane];
            m = s / (float)k.heads;
        } else {
#pragma unroll 8
            for (int t = 0; t < k.T; ++t) s += zb[t * A + lane];            // (index order, as before: same bits)
            m = s / (float)k.T;
        }
        if (k.mean) k.mean[(int64_t)b * A + lane] = m;
    }
    int best = 0;
    float bv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 0));
    for (int a = 1; a < A; ++a) {
        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), a));
        if (v > bv) {
            bv = v;
            best = a;
        }
    }
    if (lane == 0) {
        k.action[b] = best;
        if (k.action2) k.action2[b] = best;
    }
}

}  // namespace prism
