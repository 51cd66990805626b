// Host entry points of the replay ring + PER trees (C ABI in include/prism_hip.h).
#include "replay_kernels.h"

using namespace prism;

static int check_replay(const prism_replay_desc *rp, bool need_tree) {
    PRISM_CHECK_ARG(rp != nullptr, "null descriptor");
    PRISM_CHECK_ARG(rp->capacity > 0 && rp->capacity < (1ll << 31), "capacity out of range");
    int64_t cap = 1;
    while (cap <= rp->capacity) cap <<= 1;
    PRISM_CHECK_ARG(rp->tree_capacity == cap, "tree_capacity must be the smallest power of two > capacity");
    PRISM_CHECK_ARG(!need_tree || cap <= (1ll << TREE_MAX_LEVELS), "prioritized capacity above 2^24 - 1 rows");
    PRISM_CHECK_ARG(rp->obs_elems > 0, "obs_elems");
    PRISM_CHECK_ARG(rp->n_step >= 1 && rp->n_step <= PRISM_MAX_NSTEP, "n_step out of range");
    PRISM_CHECK_ARG(rp->obs && rp->succ_obs && rp->reward && rp->action && rp->flags && rp->link && rp->back,
                    "null ring array");
    PRISM_CHECK_ARG(rp->per_state && rp->status, "null per_state/status");
    if (need_tree) PRISM_CHECK_ARG(rp->tree, "prioritized call on a ring without trees");
    PRISM_CHECK_ARG((reinterpret_cast<uintptr_t>(rp->tree) & 15) == 0, "tree must be 16-byte aligned");
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
        if (tree_dense_ok(rp->tree_capacity, batch, threads))
            hipLaunchKernelGGL(per_update_kernel<true>, dim3(1), dim3(threads), 0, (hipStream_t)stream, *rp, index, priority, batch,
                               alpha, eps, take_abs);
        else
            hipLaunchKernelGGL(per_update_kernel<false>, dim3(1), dim3(threads), 0, (hipStream_t)stream, *rp, index, priority,
                               batch, alpha, eps, take_abs);
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
