// prof.hip -- live per-kernel timing with hipEvents recorded on the launch stream.
// bench.py enables it for an instrumented pass; the hot path pays one predictable branch when off.
#include "common.h"
#include <vector>
#include <mutex>

int g_act_prof_on = 0;

namespace {
const char* kNames[KID_COUNT] = {
    "fps", "knn_group", "gather_points", "gather_points_bwd", "scale_translate", "chamfer_fwd", "chamfer_bwd",
    "sgemm_nt", "sgemm_nn", "sgemm_tn", "layernorm_fwd", "layernorm_bwd", "attention_fwd", "attention_bwd",
    "colsum", "gelu_bwd", "cosine_loss_fwd", "cosine_loss_bwd", "bn_stats", "bn_apply", "bn_bwd",
    "group_maxpool", "group_maxpool_bwd", "gn_lrelu_max", "graph_feature", "gumbel_argmax", "row_gather",
    "row_scatter", "adamw", "eltwise", "sgemm_nt_bf16x3"};

struct Rec { int kid; hipEvent_t a, b; };
struct State {
    std::mutex mu;
    std::vector<Rec> recs;             // pending (not yet folded)
    std::vector<hipEvent_t> pool;      // free events
    double ms[KID_COUNT] = {0}, flops[KID_COUNT] = {0}, bytes[KID_COUNT] = {0};
    long long n[KID_COUNT] = {0};
    hipEvent_t open_a[KID_COUNT];
    bool open[KID_COUNT] = {false};
};
State& st() { static State s; return s; }

hipEvent_t get_event(State& s) {
    if (!s.pool.empty()) { hipEvent_t e = s.pool.back(); s.pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
void fold(State& s) {
    for (auto& r : s.recs) {
        hipEventSynchronize(r.b);
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) s.ms[r.kid] += t;
        s.pool.push_back(r.a); s.pool.push_back(r.b);
    }
    s.recs.clear();
}
}  // namespace

double act_prof_live_fraction(const int32_t* flags, int n, hipStream_t stream) {
    if (!g_act_prof_on || !flags || n <= 0) return 1.0;
    std::vector<int32_t> h((size_t)n);
    if (hipMemcpyAsync(h.data(), flags, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess) return 1.0;
    if (hipStreamSynchronize(stream) != hipSuccess) return 1.0;
    long long live = 0;
    for (int32_t v : h) live += v != 0;
    return (double)live / (double)n;
}

void act_prof_begin(int kid, hipStream_t stream, double flops, double bytes) {
    State& s = st();
    std::lock_guard<std::mutex> g(s.mu);
    hipEvent_t a = get_event(s);
    hipEventRecord(a, stream);
    s.open_a[kid] = a; s.open[kid] = true;
    s.flops[kid] += flops; s.bytes[kid] += bytes; s.n[kid] += 1;
}
void act_prof_end(int kid, hipStream_t stream) {
    State& s = st();
    std::lock_guard<std::mutex> g(s.mu);
    if (!s.open[kid]) return;
    hipEvent_t b = get_event(s);
    hipEventRecord(b, stream);
    s.recs.push_back({kid, s.open_a[kid], b});
    s.open[kid] = false;
    if (s.recs.size() > 8192) fold(s);
}

extern "C" {
int act_version(void) { return 100; }
const char* act_arch(void) { return "gfx950"; }
int act_prof_enable(int on) { int p = g_act_prof_on; g_act_prof_on = on ? 1 : 0; return p; }
int act_prof_reset(void) {
    State& s = st();
    std::lock_guard<std::mutex> g(s.mu);
    fold(s);
    for (int i = 0; i < KID_COUNT; ++i) { s.ms[i] = s.flops[i] = s.bytes[i] = 0; s.n[i] = 0; }
    return 0;
}
int act_prof_num_kernels(void) { return KID_COUNT; }
const char* act_prof_kernel_name(int id) { return (id >= 0 && id < KID_COUNT) ? kNames[id] : ""; }
int act_prof_read(int id, double* total_ms, long long* launches, double* flops, double* bytes) {
    if (id < 0 || id >= KID_COUNT) return ACT_E_BADARG;
    State& s = st();
    std::lock_guard<std::mutex> g(s.mu);
    fold(s);
    if (total_ms) *total_ms = s.ms[id];
    if (launches) *launches = s.n[id];
    if (flops) *flops = s.flops[id];
    if (bytes) *bytes = s.bytes[id];
    return 0;
}
}
