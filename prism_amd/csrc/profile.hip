// HIP-event brackets around individual kernel launches (bench.py's roofline leg).
#include <vector>

#include "common.h"

namespace prism {
thread_local int g_profile_on = 0;

struct Span {
    int id;
    hipEvent_t a, b;
};
static thread_local std::vector<Span> g_spans;
static thread_local std::vector<hipEvent_t> g_free;
static thread_local hipEvent_t g_open[PRISM_N_KERNEL_IDS];

static hipEvent_t get_event() {
    if (!g_free.empty()) {
        hipEvent_t e = g_free.back();
        g_free.pop_back();
        return e;
    }
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

void profile_begin(int id, hipStream_t stream) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &st) == hipSuccess && st != hipStreamCaptureStatusNone) {
        g_open[id] = nullptr;
        return;
    }
    hipEvent_t e = get_event();
    g_open[id] = e;
    if (e) (void)hipEventRecord(e, stream);
}

void profile_end(int id, hipStream_t stream) {
    if (!g_open[id]) return;
    hipEvent_t e = get_event();
    if (!e) return;
    (void)hipEventRecord(e, stream);
    g_spans.push_back(Span{id, g_open[id], e});
    g_open[id] = nullptr;
}
}  // namespace prism

using namespace prism;

extern "C" int prism_profile_enable(int on) {
    g_profile_on = on;
    return PRISM_OK;
}

extern "C" int prism_profile_collect(double *ms_sum, int64_t *count) {
    PRISM_CHECK_ARG(ms_sum && count, "null outputs");
    for (const Span &s : g_spans) {
        float ms = 0.f;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            ms_sum[s.id] += (double)ms;
            count[s.id] += 1;
        }
        g_free.push_back(s.a);
        g_free.push_back(s.b);
    }
    g_spans.clear();
    return PRISM_OK;
}

extern "C" const char *prism_profile_kernel_name(int id) {
    static const char *names[PRISM_N_KERNEL_IDS] = {
        "iqn_embed_kernel", "fwd_tile_kernel",     "iqn_loss_kernel",      "iqn_bwd_kernel",
        "iqn_post_kernel",  "step_front_kernel",   "clip_adam_kernel",     "per_sample_kernel",
        "replay_gather_kernel", "per_update_kernel", "step_back_kernel",   "qh_loss_kernel",
        "qh_bwd_kernel",     "step_tail_kernel",    "",                     ""};
    if (id < 0 || id >= PRISM_N_KERNEL_IDS) return "";
    return names[id];
}
