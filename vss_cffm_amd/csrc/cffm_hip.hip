// cffm_hip.hip -- C-ABI entry points of libcffm_hip.so (see include/cffm_hip.h) and the host-side
// orchestration of one CFFM block / layer on a HIP stream.  gfx950 only.
#include "cfm_attn_kernels.h"
#include "clip_kernels.h"
#include "rowops_kernels.h"
#include "gtc_kernels.h"
#include "gemm.h"
#include "segfuse_kernels.h"
#include "segloss_kernels.h"
#include "headfuse_kernels.h"

#include <stdarg.h>
#include <stdio.h>

#include "../../include/cffm_hip.h"

// A boolean / integer tuning switch: a constant in the product build, an environment variable in -DCFFM_EXPERIMENTS builds (see
// cffm_tune in cffm_common.h).  The alternative legs these switches select are dead code the compiler removes from libcffm_hip.so.
#ifdef CFFM_EXPERIMENTS
#define CFFM_SWITCH(fn, env, dflt) static int fn() { static int v = -(1 << 30); if (v == -(1 << 30)) { const char* e = getenv(env); v = e ? atoi(e) : (dflt); } return v; }
#define constexpr_or_not
#else
#define CFFM_SWITCH(fn, env, dflt) static constexpr int fn() { return (dflt); }
#define constexpr_or_not constexpr
#endif

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define CHECK_LAUNCH(name)                                                                        \
    do {                                                                                          \
        hipError_t e_ = hipGetLastError();                                                        \
        if (e_ != hipSuccess) return fail(-2, "%s: launch failed: %s", name, hipGetErrorString(e_)); \
    } while (0)
#define REQUIRE(cond, ...) \
    do { if (!(cond)) return fail(-1, __VA_ARGS__); } while (0)
#define TRY(call) \
    do { int rc_ = (call); if (rc_) return rc_; } while (0)

static Geo to_geo(const cffm_geom* g) {
    Geo G;
    G.B = g->B; G.H0 = g->H0; G.W0 = g->W0; G.Hp = g->Hp; G.Wp = g->Wp; G.gy = g->gy; G.gx = g->gx;
    G.nW = g->nW; G.HW = g->HW; G.RC = g->RC;
    return G;
}


// ------------------------------------------------------------------------------------------- stage timing
// Optional per-stage HIP-event timing on the caller's stream (bench.py's live `roofline` numbers):
// when enabled every stage-level entry point brackets its launches with two events.
enum { ST_TRANSPOSE, ST_POOLMAT, ST_LN_POOL_FWD, ST_LN_POOL_BWD, ST_BIAS_ASM, ST_BIAS_SCT, ST_ATTN_FWD, ST_ATTN_BWD,
       ST_GEMM, ST_COLSUM, ST_RES_LN, ST_LN_BWD, ST_GELU, ST_GELU_BWD, ST_RES_OUT, ST_GTC_FWD, ST_GTC_BWD, ST_LN, ST_ADAMW, ST_NULL_PAIR,
       // per-kernel-family stages of the block (bench.py's `roofline_kernels`); the ST_GEMM / ST_ATTN_BWD totals above stay
       ST_G_QKV_FWD, ST_G_PROJ_FWD, ST_G_FC1_FWD, ST_G_FC2_FWD, ST_G_FC2_DX, ST_G_FC1_DX, ST_G_PROJ_DX, ST_G_QKV_DX, ST_G_DW,
       ST_ATTN_BWD_Q, ST_ATTN_BWD_KV, ST_DKV_GATHER, ST_MLP_FWD, ST_MLP_BWD, ST_LN_POOL_BWD_REF, ST_COUNT };
static const char* const k_stage_names[ST_COUNT] = {"transpose", "pool_matrix", "ln_pool_fwd", "ln_pool_bwd", "bias_assemble",
    "bias_scatter", "cfm_attn_fwd", "cfm_attn_bwd", "linear_gemm", "colsum", "residual_ln", "ln_bwd", "bias_gelu", "gelu_bwd",
    "residual_out", "gtc_attn_fwd", "gtc_attn_bwd", "layernorm", "adamw", "event_pair_null",
    "gemm_qkv_fwd", "gemm_proj_fwd", "gemm_fc1_fwd", "gemm_fc2_fwd", "gemm_fc2_dx_gelu", "gemm_fc1_dx", "gemm_proj_dx", "gemm_qkv_dx",
    "gemm_dw_group", "attn_bwd_fused", "attn_bwd_bias_sum", "attn_dkv_gather", "mlp_fwd_fused", "mlp_bwd_fused", "ln_pool_bwd_ref"};
#ifndef CFFM_EMU
#include <vector>
struct ProfRec { int stage; hipEvent_t e0, e1; };
static unsigned long long g_prof_mask = 0;   // bit i = time stage i
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static hipEvent_t prof_event() {
    hipEvent_t e;
    if (!g_prof_pool.empty()) { e = g_prof_pool.back(); g_prof_pool.pop_back(); return e; }
    (void)hipEventCreate(&e);
    return e;
}
// Event pairs recorded while the stream was being captured into a HIP graph: they become event-record NODES of the graph,
// every replay re-records them, and they stay readable for the life of the graph.  The node is added explicitly
// (capture info -> hipGraphAddEventRecordNode -> new capture dependency): the HIP 7.0 runtime PyTorch bundles rejects
// hipEventRecordWithFlags(hipEventRecordExternal), the explicit form works on 7.0 and 7.2 (scripts/graph_bench_check.sh).
static std::vector<ProfRec> g_prof_graph_recs;
static bool g_prof_capture_failed = false;
static hipEvent_t prof_capture_event(hipStream_t st) {
    hipEvent_t ev = nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    hipGraph_t graph = nullptr;
    const hipGraphNode_t* deps = nullptr;
    size_t ndeps = 0;
    hipGraphNode_t node;
    bool ok = hipEventCreate(&ev) == hipSuccess
        && hipStreamGetCaptureInfo_v2(st, &cs, &id, &graph, &deps, &ndeps) == hipSuccess
        && hipGraphAddEventRecordNode(&node, graph, deps, ndeps, ev) == hipSuccess
        && hipStreamUpdateCaptureDependencies(st, &node, 1, hipStreamSetCaptureDependencies) == hipSuccess;
    if (!ok) { g_prof_capture_failed = true; (void)hipGetLastError(); }
    return ev;
}
static int g_prof_period = 1;
static unsigned g_prof_seen[ST_COUNT] = {};
struct ProfScope {
    int stage; hipStream_t st; hipEvent_t e0; bool on, cap;
    ProfScope(int s, void* stream) : stage(s), st((hipStream_t)stream), on((g_prof_mask >> s) & 1ull), cap(false) {
        if (on && g_prof_period > 1) on = (g_prof_seen[s]++ % g_prof_period) == 0;     // cffm_profile_sample_every
        if (!on) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        cap = hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive;
        if (cap) e0 = prof_capture_event(st);
        else { e0 = prof_event(); (void)hipEventRecord(e0, st); }
    }
    ~ProfScope() {
        if (!on) return;
        if (cap) {
            hipEvent_t e1 = prof_capture_event(st);
            g_prof_graph_recs.push_back({stage, e0, e1});
        } else {
            hipEvent_t e1 = prof_event(); (void)hipEventRecord(e1, st); g_prof_recs.push_back({stage, e0, e1});
        }
    }
};
#define PROF(stage) ProfScope prof_scope_(stage, stream)
#define PROF2(stage) ProfScope prof_scope2_(stage, stream)      // a second (inner / family) interval in the same scope
#else
#define PROF(stage) (void)0
#define PROF2(stage) (void)0
#endif

static bool g_side_off = false;     // cffm_side_streams(0): everything on the caller's stream (per-kernel timing passes)
extern "C" {
// mask: bit i enables stage i (cffm_profile_stage_name); 0 = off, -1 = every stage.  Each timed launch costs two event
// records on the stream (~2 us of GPU time), so timing every stage perturbs a ~1.5 ms step by ~25 %: bench.py times only
// the roofline kernel inside its timed region and takes the full breakdown in a separate pass.
int cffm_profile_enable(long long mask) {
#ifndef CFFM_EMU
    g_prof_mask = (unsigned long long)mask;
    for (int i = 0; i < ST_COUNT; ++i) g_prof_seen[i] = 0;
#endif
    return 0;
}
// time only every `period`-th launch of an enabled stage (counted from the next cffm_profile_enable): an event pair around a kernel
// of a captured step keeps its neighbours from being dispatched back to back (~10 us per pair on the chain of a replayed graph)
int cffm_profile_sample_every(int period) {
#ifndef CFFM_EMU
    g_prof_period = period > 1 ? period : 1;
#endif
    (void)period;
    return 0;
}
// an event pair with NOTHING between its two records (stage "event_pair_null"): what a timed interval costs by itself on this
// stream / in this graph -- bench.py reports it next to the roofline kernel's interval (22.6 us between events vs 19.7 us in
// rocprofv3's kernel trace; the empty pair measures 5-6 us, so it is not simply additive and is not subtracted)
int cffm_profile_null_pair(void* stream) {
#ifndef CFFM_EMU
    PROF(ST_NULL_PAIR);
#endif
    (void)stream;
    return 0;
}
// on = 0: the library's side streams are not used until switched on again -- every kernel of a block backward then runs on the caller's
// stream, one after the other (bench.py's per-kernel pass: an interval measured while another stream's GEMMs share the chip says
// little about the kernel).  Returns the previous setting.  Not to be changed while a stream is being captured.
int cffm_side_streams(int on) {
    const int was = g_side_off ? 0 : 1;
    g_side_off = on == 0;
    return was;
}
int cffm_profile_stage_count(void) { return ST_COUNT; }
const char* cffm_profile_stage_name(int i) { return (i >= 0 && i < ST_COUNT) ? k_stage_names[i] : ""; }
// sums the recorded intervals per stage (ms) and launch counts, then clears the records; synchronises.
int cffm_profile_collect(float* ms /*[count]*/, int* calls /*[count]*/) {
    for (int i = 0; i < ST_COUNT; ++i) { ms[i] = 0.f; calls[i] = 0; }
#ifndef CFFM_EMU
    for (auto& r : g_prof_recs) {
        float t = 0.f;
        (void)hipEventSynchronize(r.e1);
        (void)hipEventElapsedTime(&t, r.e0, r.e1);
        ms[r.stage] += t;
        calls[r.stage] += 1;
        g_prof_pool.push_back(r.e0);
        g_prof_pool.push_back(r.e1);
    }
    g_prof_recs.clear();
#endif
    return 0;
}
// the event pairs captured INTO a HIP graph (stages enabled while the caller's stream was capturing): per stage, the
// intervals of the graph's most recent replay (ms) and the number of pairs; synchronises; the pairs stay valid.
// reset != 0 forgets them (call before capturing another graph).
int cffm_profile_collect_graph(float* ms /*[count]*/, int* calls /*[count]*/, int reset) {
    for (int i = 0; i < ST_COUNT; ++i) { ms[i] = 0.f; calls[i] = 0; }
#ifndef CFFM_EMU
    if (g_prof_capture_failed) {
        g_prof_capture_failed = false;
        g_prof_graph_recs.clear();
        return fail(-2, "cffm_profile_collect_graph: the runtime refused an event-record node during the capture");
    }
    for (auto& r : g_prof_graph_recs) {
        float t = 0.f;
        if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) {
            (void)hipGetLastError();
            return fail(-2, "cffm_profile_collect_graph: captured events are not readable (graph not replayed yet?)");
        }
        ms[r.stage] += t;
        calls[r.stage] += 1;
    }
    if (reset) { g_prof_graph_recs.clear(); (void)hipGetLastError(); }   // (the events themselves belong to the graph's nodes: not destroyed here)
#endif
    return 0;
}
}

extern "C" {

int cffm_abi_version(void) { return CFFM_ABI_VERSION; }
const char* cffm_last_error(void) { return g_err; }

int cffm_geom_init(cffm_geom* g, int B, int H0, int W0) {
    REQUIRE(g, "geom_init: null");
    REQUIRE(B >= 1 && H0 >= 1 && W0 >= 1, "geom_init: bad sizes B=%d H0=%d W0=%d", B, H0, W0);
    g->B = B; g->H0 = H0; g->W0 = W0;
    g->Hp = (H0 + 6) / 7 * 7; g->Wp = (W0 + 6) / 7 * 7;
    g->gy = g->Hp / 7; g->gx = g->Wp / 7;
    g->nW = g->gy * g->gx; g->HW = H0 * W0; g->RC = 64 * g->nW;
    return 0;
}

static long up(long v) { return (v + 63) / 64 * 64; }  // keep every carve 256-B aligned

int cffm_block_ws_layout(const cffm_geom* g, cffm_block_ws* o) {
    REQUIRE(g && o, "ws_layout: null");
    const long B = g->B, HW = g->HW, RC = g->RC, nW = g->nW;
    long p = 0;
    o->mean1 = p; p += up(B * 4 * HW);
    o->rstd1 = p; p += up(B * 4 * HW);
    o->M = p; p += up(CFFM_NCELL * CFFM_WA);
    o->zall = p; p += up(B * RC * CFFM_C);
    o->qkv = p; p += up(B * RC * 768 / 2);   // f16 q|k|v
    o->bias = p; p += up((long)BIASH_HALFS / 2);                  // biasH: f16 B-operand fragments of the position bias (tile pairs)
    o->biasT = o->bias;                                           // (kept in the struct for ABI stability: no key-major table any more)
    o->lse = p; p += up(B * nW * CFFM_HEADS * CFFM_NQ_PAD);
    o->ao = p; p += up(B * HW * CFFM_C);
    o->x1 = p; p += up(B * HW * CFFM_C);
    o->mean2 = p; p += up(B * HW);
    o->rstd2 = p; p += up(B * HW);
    const long NP32 = (B * HW + 31) / 32 * 32, NR32 = (B * RC + 31) / 32 * 32;   // T-frag storage pads the token rows to whole k-steps of 32
    o->z2 = p; p += up(NP32 * CFFM_C);
    o->hraw = p; p += up(B * HW * CFFM_HID);
    o->act = p; p += up(NP32 * CFFM_HID);
    o->x2 = p; p += up(B * HW * CFFM_C);
    o->w_split = p; p += up(PREP_WFLOATS);   // qkv | proj | fc1 | fc2 weights in split-4 storage (k_param_prep)
    o->w_frag = p; p += up(2 * PREP_WFLOATS); // the same in MFMA-fragment order: forward forms, then input-gradient forms (row-panel kernels)
    o->ao_t = p; p += up(NP32 * CFFM_C);      // T-frag copies of ao and zall: operands of the streaming weight gradients (dws_kernels.h; ABI 9)
    o->zall_t = p; p += up(NR32 * CFFM_C);
    o->total = p;
    return 0;
}

long cffm_layer_saved_floats(const cffm_geom* g, int depth) {
    cffm_block_ws w;
    if (cffm_block_ws_layout(g, &w)) return -1;
    return up((long)g->B * 4 * g->HW * CFFM_C) + depth * w.total;
}

// scratch carve (floats): fwd uses [0, BHW*C); bwd uses all of it
struct Scratch { long a, a2, b, dz2, dao, dact, dqkv, dzall, dM, dbiasT, dxs, dkvp, alt, tf, tf_stride, total; };
static Scratch scratch_layout(const cffm_geom* g) {
    const long B = g->B, HW = g->HW, RC = g->RC;
    Scratch s;
    long p = 0;
    s.a = p; p += up(B * HW * CFFM_C);
    s.a2 = p; p += up(B * HW * CFFM_C);   // the blocks' target-frame gradient ping-pongs between a and a2 (see layer_backward_impl)
    s.b = p; p += up(B * HW * CFFM_C);
    s.dz2 = p; p += up(B * HW * CFFM_C);
    s.dao = p; p += up(B * HW * CFFM_C);
    const long NP32 = (B * HW + 31) / 32 * 32, NR32 = (B * RC + 31) / 32 * 32;   // T-frag storage pads the token rows to whole k-steps of 32
    s.dact = p; p += up(NP32 * CFFM_HID);
    s.dqkv = p; p += up(B * RC * 768);
    s.dzall = p; p += up(B * RC * CFFM_C);
    s.dM = p; p += up(CFFM_NCELL * CFFM_WA);
    s.dbiasT = p; p += up((long)CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD);
    s.dxs = p; p += up(B * 4 * HW * CFFM_C);
    s.dkvp = p; p += up(B * g->nW * ((long)CFFM_NKEY_PAD * 256 + CFFM_HEADS));   // f16 partial rows + their (window, head) scales
    // second copies of what a block's side work (weight-gradient GEMMs, column sums, bias-tile sum) still reads after the chain has
    // moved on to the next block: blocks alternate between the two sets (block_backward_impl `par`), so the next block's chain never
    // waits for the side streams.  Offsets relative to `alt`: b | dact | dqkv | dbiasT | dM
    s.alt = p;
    p += up(B * HW * CFFM_C) + up(NP32 * CFFM_HID) + up(B * RC * 768) + up((long)CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD) + up(CFFM_NCELL * CFFM_WA);
    // T-frag copies of dout | dx1 | dqkv for the streaming weight gradients, one set per parity (the side stream reads them after the chain has moved on)
    s.tf = p;
    s.tf_stride = 2 * up(NP32 * CFFM_C) + up(NR32 * 768);
    p += 2 * s.tf_stride;
    s.total = p;
    return s;
}
long cffm_layer_scratch_floats(const cffm_geom* g) { return scratch_layout(g).total; }


// ------------------------------------------------------------------------------------------- internal scratch
// Small library-owned device buffer for the block-partial records of the two-stage reductions
// (grows on demand; stream-ordered reuse, one stream at a time as the ABI's threading rule says).
// One pool per stream a stage call can arrive on: the caller's (0), the four branches of cffm_branch_begin / _take (1..4) and the deferred
// branch of cffm_defer_begin (5) -- stage calls
// that need scratch (split-K slabs of the weight gradients, column-sum records) may then run on different branches at the same time.
// ScratchFor selects the pool for the duration of a public stage call from its `stream` argument; everything else uses pool 0.
static float* g_scr[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
static size_t g_scr_floats[6] = {0, 0, 0, 0, 0, 0};
static int g_scr_sel = 0;
extern "C++" __attribute__((visibility("hidden"))) float* lib_scratch(size_t nfloats) {
    float*& buf = g_scr[g_scr_sel];
    size_t& have = g_scr_floats[g_scr_sel];
    if (nfloats <= have) return buf;
#ifdef CFFM_EMU
    free(buf);
    buf = (float*)malloc(nfloats * sizeof(float));
#else
    if (buf) { (void)hipDeviceSynchronize(); (void)hipFree(buf); }
    if (hipMalloc((void**)&buf, nfloats * sizeof(float)) != hipSuccess) buf = nullptr;
#endif
    have = buf ? nfloats : 0;
    return buf;
}
// second scratch buffer, for work issued on the library's side stream (runs concurrently with users of lib_scratch)
static float* g_scr2 = nullptr;
static size_t g_scr2_floats = 0;
static float* lib_scratch2(size_t nfloats) {
    if (nfloats <= g_scr2_floats) return g_scr2;
#ifdef CFFM_EMU
    free(g_scr2);
    g_scr2 = (float*)malloc(nfloats * sizeof(float));
#else
    if (g_scr2) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr2); }
    if (hipMalloc((void**)&g_scr2, nfloats * sizeof(float)) != hipSuccess) g_scr2 = nullptr;
#endif
    g_scr2_floats = g_scr2 ? nfloats : 0;
    return g_scr2;
}
// third scratch buffer: the per-group bias-gradient tiles of the attention backward (read by a side-stream kernel while the
// caller's stream goes on using lib_scratch)
static float* g_scr3 = nullptr;
static size_t g_scr3_floats = 0;
static float* lib_scratch3(size_t nfloats) {
    if (nfloats <= g_scr3_floats) return g_scr3;
#ifdef CFFM_EMU
    free(g_scr3);
    g_scr3 = (float*)malloc(nfloats * sizeof(float));
#else
    if (g_scr3) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr3); }
    if (hipMalloc((void**)&g_scr3, nfloats * sizeof(float)) != hipSuccess) g_scr3 = nullptr;
#endif
    g_scr3_floats = g_scr3 ? nfloats : 0;
    return g_scr3;
}
// ---- side stream of the block backward -----------------------------------------------------------------------------------
// Everything of a block backward that only produces PARAMETER gradients feeds nothing inside the chain (only the optimizer reads
// it): the four weight-gradient GEMMs, the bias-gradient tile sum + scatter, the q|k|v bias column sum, the record reductions and
// the pooling-matrix backward.  The small ones are launch- / latency-bound kernels of a few workgroups each (~45 us per block
// back to back) that truly overlap with the chain's kernels; the GEMMs need the whole chip and overlap only partly.
// All of it is issued on a library-owned second stream, forked from / joined to the caller's stream with events:
// launched eagerly the two streams overlap on the device; captured into a HIP graph (bench.py) the fork becomes a parallel
// branch of the graph.  CFFM_SIDE_STREAM=0 keeps everything on the caller's stream (A/B measurement, debugging).
struct SideStream {
    bool on = false;
    int ns = 1;                 // side streams in use (CFFM_SIDE_STREAMS = 1 | 2 | 4): with two, branches 0 / 2 (the GEMM-sized work) and
                                // 1 / 3 (the small kernels) do not queue behind each other
#ifndef CFFM_EMU
    hipStream_t st[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork[4] = {nullptr, nullptr, nullptr, nullptr}, join[2] = {nullptr, nullptr}, order = nullptr, tail_order = nullptr, cs_order = nullptr;
    // deferred join of a block backward (layer_backward_impl): what the NEXT block has to wait for
    hipEvent_t dw_done[2] = {nullptr, nullptr}, bias_done[2] = {nullptr, nullptr}, tail_done[2] = {nullptr, nullptr}, join_all[4] = {nullptr, nullptr, nullptr, nullptr};
    bool bias_pending[2] = {false, false}, dw_pending[2] = {false, false}, tail_pending[2] = {false, false};
    const float* dw_dout[2] = {nullptr, nullptr};   // the `dout` a pending weight-gradient group still reads
    int par = 0;                // scratch set of the block backward being issued
    unsigned used = 0;          // side streams forked since the last full join
#endif
};
static SideStream g_side;
#ifndef CFFM_EMU
static struct { hipStream_t st = nullptr; hipEvent_t fork = nullptr, done = nullptr; bool used = false; int state = -1; } g_defer;   // cffm_defer_begin
#endif
struct ScratchFor {       // RAII: pool of the branch `stream` is (cffm_branch_begin / _take, cffm_defer_begin), else the caller's pool
    explicit ScratchFor(void* stream) {
        g_scr_sel = 0;
#ifndef CFFM_EMU
        for (int i = 0; i < 4; ++i)
            if (stream && (hipStream_t)stream == g_side.st[i]) g_scr_sel = i + 1;
        if (stream && (hipStream_t)stream == g_defer.st) g_scr_sel = 5;
#endif
        (void)stream;
    }
    ~ScratchFor() { g_scr_sel = 0; }
};
static int stream_is_capturing(hipStream_t st) {
#ifdef CFFM_EMU
    (void)st;
    return 0;
#else
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return cs == hipStreamCaptureStatusActive ? 1 : 0;
#endif
}
// `main`: the caller's stream.  Launched eagerly ONE side stream measured best (0.877 vs 0.929 ms per step with two); under stream
// capture more (two: 0.882-0.895 vs 0.902-0.906 replayed: branches that do not queue behind each other give the graph executor more
// to run side by side; FOUR -- every branch of a block backward on its own stream -- another 0.8 %: 0.8894 vs 0.8963 as means of five
// alternating runs).  CFFM_SIDE_STREAMS=1|2|4 forces a count.
static bool side_init(hipStream_t main) {
    static int state = -1, forced_ns = 0;
    if (state < 0) {
        state = 0;
#ifndef CFFM_EMU
        const char* e = cffm_tune("CFFM_SIDE_STREAM");
        const char* n = cffm_tune("CFFM_SIDE_STREAMS");
        forced_ns = !n ? 0 : (n[0] == '1' ? 1 : (n[0] == '4' ? 4 : 2));
        // CFFM_SIDE_PRIO=low: the side streams get the lowest stream priority (parameter-gradient work yields to the chain's kernels
        // when both have workgroups ready)
        int lo = 0, hi = 0;
        const char* pr = cffm_tune("CFFM_SIDE_PRIO");
        const bool low = pr && pr[0] == 'l' && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess;
        auto mk = [&](hipStream_t* st_) { return (low ? hipStreamCreateWithPriority(st_, hipStreamNonBlocking, lo) : hipStreamCreateWithFlags(st_, hipStreamNonBlocking)) == hipSuccess; };
        if (!(e && e[0] == '0') && mk(&g_side.st[0]) && mk(&g_side.st[1]) && mk(&g_side.st[2]) && mk(&g_side.st[3])) {
            bool ok = hipEventCreateWithFlags(&g_side.order, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&g_side.tail_order, hipEventDisableTiming) == hipSuccess &&
                      hipEventCreateWithFlags(&g_side.cs_order, hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 4; ++i) ok = ok && hipEventCreateWithFlags(&g_side.fork[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 2; ++i) ok = ok && hipEventCreateWithFlags(&g_side.join[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 2; ++i) ok = ok && hipEventCreateWithFlags(&g_side.tail_done[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 4; ++i) ok = ok && hipEventCreateWithFlags(&g_side.join_all[i], hipEventDisableTiming) == hipSuccess;
            for (int i = 0; i < 2; ++i) ok = ok && hipEventCreateWithFlags(&g_side.dw_done[i], hipEventDisableTiming) == hipSuccess &&
                                             hipEventCreateWithFlags(&g_side.bias_done[i], hipEventDisableTiming) == hipSuccess;
            state = ok ? 1 : 0;
        }
        (void)hipGetLastError();
#endif
    }
    g_side.on = state == 1 && !g_side_off;
    g_side.ns = forced_ns ? forced_ns : (stream_is_capturing(main) ? 4 : 1);
    return g_side.on;
}
// the stream branch `i` (0 / 1) of the side work runs on, ordered after everything issued on `main` so far
// How the four weight-gradient GEMMs of a block backward are issued.  `one`: ONE grouped launch late in the block (59 us of GEMM +
// sum); `split`: two groups of two, each forked to the side stream as early as its operands allow (2 x 53 us, but off the chain
// sooner).  Launched eagerly the side stream really runs beside the chain and `split` wins (0.886 vs 0.911 ms per step); a replayed
// HIP graph executes its branches almost serially, so there the cheaper `one` wins (0.927 vs 0.950).  Default: by whether the
// caller's stream is being captured; CFFM_DW_GROUP=one|split forces a form.
// How the side work of a block backward is attached to the chain (bit mask; default 38 = 2 | 4 | 32, the rest for A/B measurements).
// Under stream capture the graph executor (ROCm 7.2) gives a node's FIRST-captured dependant the node's own stream and every further
// dependant the next of its (four) streams, depth first -- so whatever is launched first behind a fork stays on the chain's stream,
// and side branches that reach the same stream number run one after the other in topological order.  The chain must therefore be
// launched first at every fork and must never wait for a side branch inside the step:
//   bit 0  the dK/dV gather is launched before the bias-tile sum of the attention backward
//   bit 1  the q|k|v input-gradient GEMM is launched before the column sum / weight-gradient group
//   bit 2  the bias-tile sum + scatter get no branch of their own: they follow the weight-gradient group on its stream
//   bit 5  the record reductions + pooling-matrix backward are launched only after the NEXT chain kernel (tail_flush)
// Measured (B = 2, depth 2, replayed graph, same box): 0 -> 0.830-0.833 ms per step, 4 -> 0.808-0.810, 38 (with the two scratch sets
// of scratch_layout, so that no block waits for the previous block's weight gradients) -> 0.798-0.811 against 0.821-0.848 for 4;
// 35 (the bias-tile sum early again, now that every side branch lands on the same graph stream in launch order: it runs in the idle
// stretch beside the gather instead of behind the weight gradients on the step's tail) -> 0.773-0.785 against 0.783-0.786 for 38; with
// the last block's tail on the caller's stream and the L2 warm-up of the fused kernels 38 is ahead again (one fork less on the chain:
// a node with a dependant on another stream costs its same-stream successor ~5 us): 0.724-0.728 against 0.732-0.737 for 35.
CFFM_SWITCH(fork_order, "CFFM_FORK_ORDER", 38)
static int dw_one_group(hipStream_t st) {
    static int forced = -2;
    if (forced == -2) { const char* e = getenv("CFFM_DW_GROUP"); forced = !e ? -1 : (e[0] == 's' ? 0 : 1); }
    if (forced >= 0) return forced;
#ifdef CFFM_EMU
    (void)st;
    return 1;
#else
    return stream_is_capturing(st);
#endif
}
// Launch ORDER matters under stream capture: of the kernels that depend on one node, the graph executor keeps the FIRST-launched on
// the node's own hardware queue and moves the others to different queues -- and a dependency that crosses queues costs 5-10 us where
// same-queue succession costs ~1.5 (round-3 timeline: the chain hopped queues at every fork because the side work was launched first).
// So a fork is split: side_fork_mark() records the point on `main`, the chain's next kernel is launched, and only then
// side_fork_take() makes the side stream wait for the mark and hands it out.
static void side_fork_mark(hipStream_t main, int i) {
#ifndef CFFM_EMU
    if (g_side.on) (void)hipEventRecord(g_side.fork[i], main);
#endif
    (void)main; (void)i;
}
static hipStream_t side_fork_take(hipStream_t main, int i) {
#ifndef CFFM_EMU
    const int idx = g_side.ns >= 4 ? (i & 3) : (g_side.ns > 1 ? (i & 1) : 0);
    hipStream_t s = g_side.st[idx];
    if (g_side.on && hipStreamWaitEvent(s, g_side.fork[i], 0) == hipSuccess) {
        g_side.used |= 1u << idx;
        return s;
    }
    (void)hipGetLastError();
#endif
    (void)i;
    return main;
}
static hipStream_t side_fork(hipStream_t main, int i) {
#ifndef CFFM_EMU
    const int idx = g_side.ns >= 4 ? (i & 3) : (g_side.ns > 1 ? (i & 1) : 0);
    hipStream_t s = g_side.st[idx];
    if (g_side.on && hipEventRecord(g_side.fork[i], main) == hipSuccess && hipStreamWaitEvent(s, g_side.fork[i], 0) == hipSuccess) {
        g_side.used |= 1u << idx;
        return s;
    }
    (void)hipGetLastError();
#endif
    (void)i;
    return main;
}
// what is issued on `later` from here on runs after what has been issued on `earlier` (two different side streams)
static void side_order(hipStream_t earlier, hipStream_t later) {
#ifndef CFFM_EMU
    if (earlier != later && hipEventRecord(g_side.order, earlier) == hipSuccess) (void)hipStreamWaitEvent(later, g_side.order, 0);
#endif
    (void)earlier; (void)later;
}
static void side_mark(hipStream_t side, hipStream_t main, int i) {   // end of branch i
#ifndef CFFM_EMU
    if (side != main) (void)hipEventRecord(g_side.join[i], side);
#endif
    (void)side; (void)main; (void)i;
}
static void side_join(hipStream_t side, hipStream_t main, int i) {   // `main` continues after branch i
#ifndef CFFM_EMU
    if (side != main) (void)hipStreamWaitEvent(main, g_side.join[i], 0);
#endif
    (void)side; (void)main; (void)i;
}

// `main` continues after everything issued on any side stream since the last full join
static void side_join_all(hipStream_t main) {
#ifndef CFFM_EMU
    for (int i = 0; i < 4; ++i)
        if ((g_side.used >> i) & 1u) {
            if (hipEventRecord(g_side.join_all[i], g_side.st[i]) == hipSuccess) (void)hipStreamWaitEvent(main, g_side.join_all[i], 0);
        }
    g_side.used = 0;
    g_side.bias_pending[0] = g_side.bias_pending[1] = g_side.dw_pending[0] = g_side.dw_pending[1] = g_side.tail_pending[0] = g_side.tail_pending[1] = false;
    (void)hipGetLastError();
#endif
    (void)main;
}
// ---- branches for the CALLER (ABI 10): independent stage calls of the head's rows path side by side -------------------------------
// The head around the hot path (rows f.1 / f.2 of SURVEY 8: embedding per scale, composed weights, classifiers) is a sequence of stage
// calls from Python, most of them 5-15 us kernels that do not depend on each other (four scales, weight vs input gradients): on ONE
// stream a replayed step pays ~5 us of launch latency per kernel (profiles/r06_head_timeline_before.txt: ~60 such kernels).
// cffm_branch_begin(stream, i) hands out library-owned stream i (0..3), ordered behind everything issued on `stream` so far; the caller
// passes it as the `stream` argument of the stage calls of that branch; cffm_branch_join(stream) makes `stream` continue behind all
// branches.  Under stream capture the branches become parallel branches of the graph.  Rules: every tensor a branch touches stays alive
// until the join; the stage calls that use the library's own scratch (weight gradients, column sums) take it from the pool of the
// branch they are issued on (ScratchFor); with side streams off (cffm_side_streams(0)), on failure, or in the emulator build the
// caller's own stream comes back and everything simply runs in order.
// Launch ORDER matters under capture (see side_fork_mark): of the dependants of one node the graph executor keeps the FIRST-launched on the
// node's own hardware queue, the others start on other queues ~10 us later each.  So the caller marks the fork point (cffm_branch_mark), launches
// its own -- longest -- chain on `stream` first and only then takes the branches (cffm_branch_take) for the shorter ones.
// cffm_branch_begin = mark + take of one branch at the current point.
int cffm_branch_mark(void* stream) {
#ifndef CFFM_EMU
    hipStream_t main = (hipStream_t)stream;
    if (!side_init(main)) return 0;
    g_side.ns = 4;
    for (int i = 0; i < 4; ++i)
        if (hipEventRecord(g_side.fork[i], main) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return 1;
#else
    (void)stream;
    return 0;
#endif
}
void* cffm_branch_take(void* stream, int i) {
#ifndef CFFM_EMU
    if (i < 0 || i > 3 || !g_side.on) return stream;
    hipStream_t s = g_side.st[i];
    if (hipStreamWaitEvent(s, g_side.fork[i], 0) == hipSuccess) {
        g_side.used |= 1u << i;
        return (void*)s;
    }
    (void)hipGetLastError();
#endif
    (void)i;
    return stream;
}
void* cffm_branch_begin(void* stream, int i) {
    hipStream_t main = (hipStream_t)stream;
#ifndef CFFM_EMU
    if (i < 0 || i > 3 || !side_init(main)) return stream;
    g_side.ns = 4;
    hipStream_t s = g_side.st[i];
    if (hipEventRecord(g_side.fork[i], main) == hipSuccess && hipStreamWaitEvent(s, g_side.fork[i], 0) == hipSuccess) {
        g_side.used |= 1u << i;
        return (void*)s;
    }
    (void)hipGetLastError();
#endif
    (void)i;
    return (void*)main;
}
int cffm_branch_join(void* stream) {
    side_join_all((hipStream_t)stream);
    return 0;
}
// A DEFERRED branch for the caller (ABI 10): work whose results the caller's stream needs much later than the next call -- the frame
// classifier's backward of the CFFM heads (cffm_head.py:121: its input gradient is needed by linear_fuse's BatchNorm backward, its weight /
// bias gradients by the optimizer, and the whole CFFM layer's backward lies between).  cffm_defer_begin(stream): a library-owned stream
// of its own (not one of the four branches: the layer's own side work uses those and must not queue behind 270 us of classifier kernels),
// ordered behind everything issued on `stream` so far; cffm_defer_join(stream): `stream` continues behind it (a no-op when nothing is
// pending).  Everything the deferred work touches must stay alive, and nothing on `stream` may read its results, until the join.
void* cffm_defer_begin(void* stream) {
#ifndef CFFM_EMU
    hipStream_t main = (hipStream_t)stream;
    if (g_side_off) return stream;
    if (g_defer.state < 0) {
        g_defer.state = (hipStreamCreateWithFlags(&g_defer.st, hipStreamNonBlocking) == hipSuccess &&
                         hipEventCreateWithFlags(&g_defer.fork, hipEventDisableTiming) == hipSuccess &&
                         hipEventCreateWithFlags(&g_defer.done, hipEventDisableTiming) == hipSuccess) ? 1 : 0;
        (void)hipGetLastError();
    }
    if (g_defer.state == 1 && hipEventRecord(g_defer.fork, main) == hipSuccess && hipStreamWaitEvent(g_defer.st, g_defer.fork, 0) == hipSuccess) {
        g_defer.used = true;
        return (void*)g_defer.st;
    }
    (void)hipGetLastError();
#endif
    return stream;
}
int cffm_defer_join(void* stream) {
#ifndef CFFM_EMU
    if (g_defer.used) {
        if (hipEventRecord(g_defer.done, g_defer.st) == hipSuccess) (void)hipStreamWaitEvent((hipStream_t)stream, g_defer.done, 0);
        g_defer.used = false;
        (void)hipGetLastError();
    }
#endif
    (void)stream;
    return 0;
}
// event helpers of the deferred join (no-ops when the work ran on `main` itself)
static void side_record(hipStream_t side, hipStream_t main, void* ev) {
#ifndef CFFM_EMU
    if (side != main) (void)hipEventRecord((hipEvent_t)ev, side);
#endif
    (void)side; (void)main; (void)ev;
}
static void side_wait(hipStream_t main, void* ev) {
#ifndef CFFM_EMU
    (void)hipStreamWaitEvent(main, (hipEvent_t)ev, 0);
#endif
    (void)main; (void)ev;
}
#ifndef CFFM_EMU
static void dw_group_launched(hipStream_t side) { (void)hipEventRecord(g_side.dw_done[g_side.par], side); }
#else
static void dw_group_launched(hipStream_t) {}
#endif

static void seg_add(RedSegs& r, int off, int width, float* out, int accumulate) {
    if (!out) return;
    const int k = r.nseg++;
    r.off[k] = off; r.width[k] = width; r.out[k] = out; r.accumulate[k] = accumulate;
}
// Deferred second stages: inside a block backward (RedScope) the reductions are queued -- each with its own slice of a
// library-owned record buffer -- and run as ONE launch at the end of the scope; outside a scope they launch at once.
// Two record buffers, used alternately by consecutive scopes (block backwards): the reduction of block i (side stream) may still
// read its records while block i - 1 already writes its own (deferred join, see block_backward_impl)
static float* g_redbuf[2] = {nullptr, nullptr};
static size_t g_red_floats_[2] = {0, 0};
static int g_red_parity = 0;
#define g_red g_redbuf[g_red_parity]
#define g_red_floats g_red_floats_[g_red_parity]
static struct { bool active; size_t bump; RedJobs jobs; } g_rq = {false, 0, {}};
static void redq_launch(RedJobs& J, hipStream_t st) {
    if (J.njob == 1) {
        CFFM_LAUNCH(k_reduce_records, ((J.total[0] + 63) / 64), (1024), 0, st, J.part[0], J.nblk[0], J.stride[0], J.total[0], J.segs[0]);
    } else if (J.njob > 1) {
        CFFM_LAUNCH(k_reduce_records_multi, ((unsigned)J.blk_end[J.njob - 1]), (1024), 0, st, J);
    }
    J.njob = 0;
}
static void redq_flush(hipStream_t st) { redq_launch(g_rq.jobs, st); }
struct RedScope {
    hipStream_t st;
    explicit RedScope(hipStream_t s) : st(s) { g_red_parity ^= 1; g_rq.active = true; g_rq.bump = 0; g_rq.jobs.njob = 0; }
    void finish() { redq_flush(st); g_rq.active = false; }
    void finish_on(hipStream_t other) { redq_flush(other); g_rq.active = false; }
    ~RedScope() { g_rq.active = false; g_rq.jobs.njob = 0; }
};
// block-partial record buffer of one reduction
static float* red_scratch(size_t nfloats, hipStream_t st) {
    if (!g_rq.active) return lib_scratch(nfloats);
    nfloats = (nfloats + 63) / 64 * 64;
    if (g_rq.bump + nfloats > g_red_floats) {
#ifndef CFFM_EMU
        (void)hipDeviceSynchronize();   // (growth happens in the first step only) the records queued so far may have been written on another stream
#endif
        redq_flush(st);   // queued jobs read the old buffer
        const size_t want = 2 * (g_rq.bump + nfloats);
#ifdef CFFM_EMU
        free(g_red);
        g_red = (float*)malloc(want * sizeof(float));
#else
        if (g_red) { (void)hipDeviceSynchronize(); (void)hipFree(g_red); }
        if (hipMalloc((void**)&g_red, want * sizeof(float)) != hipSuccess) g_red = nullptr;
#endif
        g_red_floats = g_red ? want : 0;
        g_rq.bump = 0;
        if (!g_red) return nullptr;
    }
    float* p = g_red + g_rq.bump;
    g_rq.bump += nfloats;
    return p;
}
static void reduce_records(const float* part, int nblk, int stride, int total, const RedSegs& segs, hipStream_t st) {
    if (!g_rq.active) {
        CFFM_LAUNCH(k_reduce_records, ((total + 63) / 64), (1024), 0, st, part, nblk, stride, total, segs);
        return;
    }
    RedJobs& J = g_rq.jobs;
    if (J.njob == RED_MAXJOB) redq_flush(st);
    const int j = J.njob++;
    J.part[j] = part; J.nblk[j] = nblk; J.stride[j] = stride; J.total[j] = total; J.segs[j] = segs;
    J.blk_end[j] = (j ? J.blk_end[j - 1] : 0) + (total + 63) / 64;
}

static unsigned ew_grid(long n4) {
    long b = (n4 + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

// ------------------------------------------------------------------------------------------- stages
static int transpose_add(const float* src, float* dst, int batch, int rows, int cols, long src_bs, long dst_bs, const float* addend,
                         int add_mod, int add_skip, void* stream, float* copy_dst = nullptr) {
    PROF(ST_TRANSPOSE);
    REQUIRE(src && dst && batch > 0 && rows > 0 && cols > 0, "transpose: bad arguments");
    CFFM_LAUNCH(k_transpose, ((cols + 63) / 64, (rows + 63) / 64, batch), (256), 0, (hipStream_t)stream, src, dst, rows, cols,
                src_bs, dst_bs, addend, add_mod > 0 ? add_mod : 1, add_skip, copy_dst);
    CHECK_LAUNCH("transpose");
    return 0;
}
// a += b over n floats (n % 4 == 0): summing the per-clip pieces of a gradient on the stream they were computed on
int cffm_add_inplace(float* a, const float* b, long n, void* stream) {
    REQUIRE(a && b && n >= 0 && n % 4 == 0, "add_inplace: n must be a multiple of 4");
    if (!n) return 0;
    CFFM_LAUNCH(k_add_inplace, (ew_grid(n / 4)), (256), 0, (hipStream_t)stream, a, b, n / 4);
    CHECK_LAUNCH("add_inplace");
    return 0;
}
int cffm_transpose(const float* src, float* dst, int batch, int rows, int cols, long src_bs, long dst_bs, void* stream) {
    return transpose_add(src, dst, batch, rows, cols, src_bs, dst_bs, nullptr, 1, -1, stream);
}

int cffm_pool_matrix(const float* const pool_w[4], float* M, void* stream) {
    PROF(ST_POOLMAT);
    PoolW pw;
    for (int i = 0; i < 4; ++i) { REQUIRE(pool_w[i], "pool_matrix: null weight %d", i); pw.w[i] = pool_w[i]; }
    CFFM_LAUNCH(k_pool_matrix, (1), (256), 0, (hipStream_t)stream, pw, M);
    CHECK_LAUNCH("pool_matrix");
    return 0;
}

static thread_local int g_grad_pads = 0;   // per calling thread (autograd runs one backward thread per device: ADVICE r4)
void cffm_grad_slices_padded(int yes) { g_grad_pads = yes ? 1 : 0; }
static int pool_matrix_bwd_impl(const float* dM, float* const dpool_w[4], float* const dpool_b[4], void* stream) {
    PROF(ST_POOLMAT);
    PoolWG gw;
    for (int i = 0; i < 4; ++i) { gw.w[i] = dpool_w[i]; gw.b[i] = dpool_b ? dpool_b[i] : nullptr; }
    CFFM_LAUNCH(k_pool_matrix_bwd, (111), (64), 0, (hipStream_t)stream, dM, gw);
    CHECK_LAUNCH("pool_matrix_bwd");
    return 0;
}
int cffm_pool_matrix_bwd(const float* dM, float* const dpool_w[4], void* stream) { return pool_matrix_bwd_impl(dM, dpool_w, nullptr, stream); }

// `split`: the rows a GEMM is the only reader of are written in split-4 storage (cffm_common.h); the public stage
// functions always write plain fp32, the block orchestration asks for split-4 when the hand-written GEMMs are in use
static int ln_pool_fwd_impl(const cffm_geom* g, const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs,
                            const float* gamma, const float* beta, const float* M, const float* const pool_b[4],
                            float* zall, float* mean, float* rstd, int split, void* stream, const float* const* pool_w = nullptr) {
    PROF(ST_LN_POOL_FWD);
    REQUIRE(g && x_ref && x_tgt && zall, "ln_pool_fwd: null");
    PoolB pb;
    PoolW pw;
    for (int i = 0; i < 4; ++i) { pb.b[i] = pool_b[i]; pw.w[i] = pool_w ? pool_w[i] : nullptr; }
    CFFM_LAUNCH(k_ln_pool_fwd, (g->nW, 4, g->B), (LNP_THREADS), 0, (hipStream_t)stream, to_geo(g), x_ref, ref_bs, x_tgt, tgt_bs, gamma,
                beta, pool_w ? nullptr : M, pw, pb, zall, mean, rstd, split);
    CHECK_LAUNCH("ln_pool_fwd");
    return 0;
}
int cffm_ln_pool_fwd(const cffm_geom* g, const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs,
                     const float* gamma, const float* beta, const float* M, const float* const pool_b[4],
                     float* zall, float* mean, float* rstd, void* stream) {
    return ln_pool_fwd_impl(g, x_ref, ref_bs, x_tgt, tgt_bs, gamma, beta, M, pool_b, zall, mean, rstd, 0, stream);
}

// ---- CFFA backward (cffa_kernels.h): target frame per block, reference frames once per range of blocks ------------------------------
static long cffa_rows(const cffm_geom* g) { return (long)g->B * g->nW * LNB_SPLIT; }     // records per kernel launch and block
static int ln_pool_bwd_tgt(const cffm_geom* g, const float* x_tgt, long tgt_bs, const float* gamma, const float* beta, const float* M,
                           const float* mean, const float* rstd, const float* dzall, const float* dres, float* dx_tgt, long dtgt_bs,
                           float* rec, void* stream) {
    PROF(ST_LN_POOL_BWD);
    CFFM_LAUNCH(k_ln_pool_bwd_tgt, ((unsigned)cffa_rows(g)), (LNB_THREADS), 0, (hipStream_t)stream, to_geo(g), x_tgt, tgt_bs, gamma, beta, M, mean, rstd,
                dzall, dres, dx_tgt, dtgt_bs, rec);
    CHECK_LAUNCH("ln_pool_bwd_tgt");
    return 0;
}
extern "C++" {
template <int D>
static int ln_pool_bwd_ref_d(const cffm_geom* g, const float* x_ref, long ref_bs, const CffaRefBlocks& Bk, const float* mean, const float* rstd,
                             float* dx_ref, long dref_bs, int accum, hipStream_t st) {
    constexpr int lds = cffa_ref_lds(D);
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted && lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)k_ln_pool_bwd_ref<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_ln_pool_bwd_ref<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return -1;
        granted = true;
    }
#endif
    if (accum) CFFM_LAUNCH((k_ln_pool_bwd_ref<D, true>), ((unsigned)cffa_rows(g)), (LNB_THREADS), lds, st, to_geo(g), x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs);
    else CFFM_LAUNCH((k_ln_pool_bwd_ref<D, false>), ((unsigned)cffa_rows(g)), (LNB_THREADS), lds, st, to_geo(g), x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs);
    return 0;
}
}  // extern "C++"
// the reference frames of Bk.n <= RB_MAXD blocks; accum: dx_ref already holds the share of blocks handled earlier
static int ln_pool_bwd_ref(const cffm_geom* g, const float* x_ref, long ref_bs, const CffaRefBlocks& Bk, const float* mean, const float* rstd,
                           float* dx_ref, long dref_bs, int accum, void* stream) {
    PROF(ST_LN_POOL_BWD_REF);
    hipStream_t st = (hipStream_t)stream;
    int rc = -1;
    switch (Bk.n) {
        case 1: rc = ln_pool_bwd_ref_d<1>(g, x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs, accum, st); break;
        case 2: rc = ln_pool_bwd_ref_d<2>(g, x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs, accum, st); break;
        case 3: rc = ln_pool_bwd_ref_d<3>(g, x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs, accum, st); break;
        case 4: rc = ln_pool_bwd_ref_d<4>(g, x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs, accum, st); break;
        default: break;
    }
    REQUIRE(!rc, "ln_pool_bwd_ref: bad block count %d or LDS grant failed", Bk.n);
    CHECK_LAUNCH("ln_pool_bwd_ref");
    return 0;
}
// The three record sums of one block (its target records in rec[0, rows), its reference records in rec[rows, 2 rows)): norm1's
// dgamma | dbeta over both, the target tail (dM cell 0, pool bias 0), the reference tail (dM cells 1..14, pool biases 1..3) -- appended
// to J (launched when it is full; the caller launches the rest with redq_launch).
static void job_add(RedJobs& J, const float* part, int nblk, int stride, int total, const RedSegs& segs, hipStream_t st) {
    if (J.njob == RED_MAXJOB) redq_launch(J, st);
    const int j = J.njob++;
    J.part[j] = part; J.nblk[j] = nblk; J.stride[j] = stride; J.total[j] = total; J.segs[j] = segs;
    J.blk_end[j] = (j ? J.blk_end[j - 1] : 0) + (total + 63) / 64;
}
static void cffa_jobs(RedJobs& J, const float* rec, long rows, float* dgamma, float* dbeta, float* dM, float* const dpool_b[4], hipStream_t st) {
    RedSegs a, b, c;
    a.nseg = b.nseg = c.nseg = 0;
    seg_add(a, 0, CFFM_C, dgamma, 0);
    seg_add(a, CFFM_C, CFFM_C, dbeta, 0);
    job_add(J, rec, (int)(2 * rows), LNP_RSTRIDE, 2 * CFFM_C, a, st);
    seg_add(b, 0, CFFM_WA, dM, 0);
    seg_add(b, CFFM_WA, 1, dpool_b[0], 0);
    job_add(J, rec + 2 * CFFM_C, (int)rows, LNP_RSTRIDE, LNP_TAIL_TGT, b, st);
    seg_add(c, 0, 14 * CFFM_WA, dM + CFFM_WA, 0);
    for (int i = 1; i < 4; ++i) seg_add(c, 14 * CFFM_WA + i - 1, 1, dpool_b[i], 0);
    job_add(J, rec + rows * LNP_RSTRIDE + 2 * CFFM_C, (int)rows, LNP_RSTRIDE, LNP_TAIL_REF, c, st);
}

int cffm_ln_pool_bwd(const cffm_geom* g, const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs,
                     const float* gamma, const float* beta, const float* M, const float* mean, const float* rstd,
                     const float* dzall, const float* dres, float* dx_ref, long dref_bs, int accum_ref,
                     float* dx_tgt, long dtgt_bs, float* dgamma, float* dbeta, float* dM, float* const dpool_b[4],
                     void* stream) {
    REQUIRE(g && x_ref && x_tgt && gamma && beta && M && mean && rstd && dzall && dx_ref && dx_tgt && dgamma && dbeta && dM && dpool_b, "ln_pool_bwd: null");
    hipStream_t st = (hipStream_t)stream;
    const long rows = cffa_rows(g);
    float* rec = red_scratch((size_t)2 * rows * LNP_RSTRIDE, st);
    REQUIRE(rec, "ln_pool_bwd: scratch allocation failed");
    TRY(ln_pool_bwd_tgt(g, x_tgt, tgt_bs, gamma, beta, M, mean, rstd, dzall, dres, dx_tgt, dtgt_bs, rec, stream));
    CffaRefBlocks Bk;
    Bk.n = 1;
    for (int d = 0; d < RB_MAXD; ++d) { Bk.gamma[d] = gamma; Bk.beta[d] = beta; Bk.M[d] = M; Bk.dzall[d] = dzall; Bk.rec[d] = rec + rows * LNP_RSTRIDE; }
    TRY(ln_pool_bwd_ref(g, x_ref, ref_bs, Bk, mean, rstd, dx_ref, dref_bs, accum_ref, stream));
    RedJobs J;
    J.njob = 0;
    cffa_jobs(J, rec, rows, dgamma, dbeta, dM, dpool_b, st);
    redq_launch(J, st);
    CHECK_LAUNCH("ln_pool_bwd");
    return 0;
}

// Library-owned state of the CFFA backward, one SLOT per block of the layer: the block's token-row gradient dzall (written by the q|k|v
// input-gradient GEMM, its pooled-cell rows are read again by the reference pass at the end of the range), the records of its target
// and reference workgroups, its pooling-matrix gradient.
static float* g_scr4 = nullptr;
static size_t g_scr4_floats = 0;
static float* lib_scratch4(size_t nfloats) {
    if (nfloats <= g_scr4_floats) return g_scr4;
#ifdef CFFM_EMU
    free(g_scr4);
    g_scr4 = (float*)malloc(nfloats * sizeof(float));
#else
    if (g_scr4) { (void)hipDeviceSynchronize(); (void)hipFree(g_scr4); }
    if (hipMalloc((void**)&g_scr4, nfloats * sizeof(float)) != hipSuccess) g_scr4 = nullptr;
#endif
    g_scr4_floats = g_scr4 ? nfloats : 0;
    return g_scr4;
}
struct CffaSlot { float* dzall; float* rec; float* dM; };
static int cffa_slot(const cffm_geom* g, int slot, int nslots, CffaSlot* o) {
    const long a = up((long)g->B * g->RC * CFFM_C), b = up(2 * cffa_rows(g) * LNP_RSTRIDE), c = up(CFFM_NCELL * CFFM_WA);
    float* base = lib_scratch4((size_t)(a + b + c) * nslots);
    REQUIRE(base && slot >= 0 && slot < nslots, "cffa: scratch allocation failed");
    float* p = base + (a + b + c) * slot;
    o->dzall = p; o->rec = p + a; o->dM = p + a + b;
    return 0;
}
// End of a range of block backwards (blocks first, first - 1, ..., last of a layer with `nslots` blocks; block i's parameters / gradients
// at params[i] / grads[i], its workspace at ws0 + i * ws_stride): the reference frames of all of them in one pass (RB_MAXD blocks per
// launch) -> dx_ref, then every CFFA parameter gradient of the range: norm1, the pooling matrix -> the pooling Linears, the pool biases.
// `extra` (or NULL): record reductions of the caller that are due at this point of the stream anyway (the last block's parameter-gradient
// tail): they join the final reduction launch instead of taking one of their own.
static int cffa_finish(const cffm_geom* g, int nslots, const cffm_block_params* params, const cffm_block_grads* grads, int first, int last,
                       const float* ws0, long ws_stride, const float* x_ref, long ref_bs, float* dx_ref, long dref_bs, int accum, void* stream,
                       RedJobs* extra = nullptr) {
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    hipStream_t st = (hipStream_t)stream;
    const long rows = cffa_rows(g);
    static_assert(PMB_MAXD == RB_MAXD, "one pooling-matrix launch per reference launch");
    for (int hi = first; hi >= last; hi -= RB_MAXD) {
        const int n = hi - last + 1 < RB_MAXD ? hi - last + 1 : RB_MAXD;
        CffaRefBlocks Bk;
        PoolWGN pw;
        RedJobs J;
        Bk.n = n; J.njob = 0;
        for (int d = 0; d < RB_MAXD; ++d) {
            const int i = hi - (d < n ? d : 0);
            CffaSlot S;
            TRY(cffa_slot(g, i, nslots, &S));
            Bk.gamma[d] = params[i].norm1_w; Bk.beta[d] = params[i].norm1_b; Bk.M[d] = ws0 + i * ws_stride + L.M;
            Bk.dzall[d] = S.dzall; Bk.rec[d] = S.rec + rows * LNP_RSTRIDE;
            pw.dM[d] = S.dM;
            for (int q = 0; q < 4; ++q) { pw.gw[d].w[q] = grads[i].pool_w[q]; pw.gw[d].b[q] = g_grad_pads ? grads[i].pool_b[q] : nullptr; }
        }
        // (the LayerNorm statistics of the reference frames are the same in every block: any block's saved copy serves)
        const float* ws = ws0 + (long)hi * ws_stride;
        TRY(ln_pool_bwd_ref(g, x_ref, ref_bs, Bk, ws + L.mean1, ws + L.rstd1, dx_ref, dref_bs, accum || hi != first, stream));
        for (int d = 0; d < n; ++d) {
            const int i = hi - d;
            CffaSlot S;
            TRY(cffa_slot(g, i, nslots, &S));
            cffa_jobs(J, S.rec, rows, grads[i].norm1_w, grads[i].norm1_b, S.dM, grads[i].pool_b, st);
        }
        if (extra && hi - RB_MAXD < last) {        // (the range's last launch)
            for (int j = 0; j < extra->njob; ++j) job_add(J, extra->part[j], extra->nblk[j], extra->stride[j], extra->total[j], extra->segs[j], st);
            extra->njob = 0;
        }
        redq_launch(J, st);
        CHECK_LAUNCH("cffa reductions");
        {
            PROF(ST_POOLMAT);
            CFFM_LAUNCH(k_pool_matrix_bwd_n, (111, n), (64), 0, st, pw);
            CHECK_LAUNCH("pool_matrix_bwd");
        }
    }
    return 0;
}

int cffm_bias_assemble(const float* own, const float* ring, const float* const pool[4], float* bias, void* biasH, void* stream) {
    PROF(ST_BIAS_ASM);
    BiasTables t;
    t.own = own; t.ring = ring;
    for (int i = 0; i < 4; ++i) t.pool[i] = pool[i];
    const int n = CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD;
    CFFM_LAUNCH(k_bias_assemble, ((n + 255) / 256), (256), 0, (hipStream_t)stream, t, bias, (h16*)biasH);
    CHECK_LAUNCH("bias_assemble");
    return 0;
}

int cffm_bias_scatter(const float* dbiasT, float* down, float* dring, float* const dpool[4], void* stream) {
    PROF(ST_BIAS_SCT);
    BiasTablesG t;
    t.own = down; t.ring = dring;
    for (int i = 0; i < 4; ++i) t.pool[i] = dpool[i];
    CFFM_LAUNCH(k_bias_scatter, ((BIAS_SCATTER_THREADS + 255) / 256), (256), 0, (hipStream_t)stream, dbiasT, t);
    CHECK_LAUNCH("bias_scatter");
    return 0;
}

int cffm_attn_fwd(const cffm_geom* g, const void* qkv16, const int* key_src, const int* q_dst, const void* biasH, float* ao,
                  float* lse, void* stream) {
    PROF(ST_ATTN_FWD);
    REQUIRE(g && qkv16 && key_src && q_dst && biasH && ao && lse, "attn_fwd: null");
    const h16* bias = (const h16*)biasH;
    // one workgroup per (window, head), four per CU
    CFFM_LAUNCH(k_cfm_attn_fwd, (g->B * g->nW * CFFM_HEADS), (256), ATT_FWD_LDS, (hipStream_t)stream, to_geo(g), (const h16*)qkv16,
                key_src, q_dst, bias, ao, lse);
    CHECK_LAUNCH("attn_fwd");
    return 0;
}

// window groups of the fused backward: 8 heads x groups <= one 12-wave workgroup per CU on 256 CUs, every group the same length
static int attn_bwd_groups(const cffm_geom* g, int* per_group) {
    const int total = g->B * g->nW;
    static int want = -1;   // tuning aid: CFFM_BWD_GROUPS
    if (want < 0) { const char* e = cffm_tune("CFFM_BWD_GROUPS"); want = e ? atoi(e) : 32; if (want < 1) want = 32; }
    int ng = total < want ? total : want;
    *per_group = (total + ng - 1) / ng;
    return (total + *per_group - 1) / *per_group;
}

// the three pieces of the attention backward (the block backward puts the bias-gradient sum on its side stream)
static int attn_bwd_fused(const cffm_geom* g, const void* qkv16, const int* key_src, const int* q_dst, const h16* bias, const float* ao,
                          const float* dao, const float* lse, float* dqkv, float* dkv_part, float** dbp_out, int* ng_out, int par, void* stream) {
    PROF2(ST_ATTN_BWD_Q);
    int per;
    const int ng = attn_bwd_groups(g, &per);
    const long nb = (long)CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD;   // one bias-gradient tile set per window group
    float* dbp = lib_scratch3((size_t)2 * ng * nb);      // two sets: see scratch_layout (alt)
    REQUIRE(dbp, "attn_bwd: scratch allocation failed");
    dbp += (size_t)(par ? 1 : 0) * ng * nb;
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        REQUIRE(hipFuncSetAttribute((const void*)k_cfm_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, ATT_BK_LDS) == hipSuccess,
                "attn_bwd: LDS grant failed");
        granted = true;
    }
#endif
    CFFM_LAUNCH(k_cfm_attn_bwd, (CFFM_HEADS, ng), (ATT_BK_THREADS), ATT_BK_LDS, (hipStream_t)stream, to_geo(g), (const h16*)qkv16, key_src, q_dst, bias,
                ao, dao, lse, dqkv, dbp, dkv_part, per);
    CHECK_LAUNCH("attn_bwd");
    *dbp_out = dbp; *ng_out = ng;
    return 0;
}
static int attn_bwd_bias_sum(const float* dbp, int ng, float* dbiasT, void* stream) {
    PROF2(ST_ATTN_BWD_KV);
    const long nb = (long)CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD;
    CFFM_LAUNCH(k_sum_splits, ((unsigned)((nb / 4 + 63) / 64)), (256), 0, (hipStream_t)stream, dbp, ng, nb, dbiasT);
    CHECK_LAUNCH("attn_bwd bias sum");
    return 0;
}
static int attn_bwd_gather(const cffm_geom* g, const int* inv_ptr, const int* inv_idx, const float* dkv_part, float* dqkv, void* stream) {
    PROF2(ST_DKV_GATHER);
    CFFM_LAUNCH(k_dkv_gather, ((g->RC + 3) / 4, g->B), (256), 0, (hipStream_t)stream, to_geo(g), inv_ptr, inv_idx, dkv_part, dqkv);
    CHECK_LAUNCH("attn_bwd gather");
    return 0;
}
int cffm_attn_bwd(const cffm_geom* g, const void* qkv16, const int* key_src, const int* q_dst,
                  const int* inv_ptr, const int* inv_idx, const void* biasH, const float* ao,
                  const float* dao, const float* lse, float* dqkv, float* dbiasT, float* dkv_part, void* stream) {
    PROF(ST_ATTN_BWD);
    const h16* bias = (const h16*)biasH;
    REQUIRE(g && qkv16 && bias && dao && dqkv && dbiasT && dkv_part && inv_ptr && inv_idx, "attn_bwd: null");
    float* dbp;
    int ng;
    TRY(attn_bwd_fused(g, qkv16, key_src, q_dst, bias, ao, dao, lse, dqkv, dkv_part, &dbp, &ng, 0, stream));
    TRY(attn_bwd_bias_sum(dbp, ng, dbiasT, stream));
    TRY(attn_bwd_gather(g, inv_ptr, inv_idx, dkv_part, dqkv, stream));
    return 0;
}

int cffm_linear_fwd(const float* x, const float* w, float* y, long M, int N, int K, void* stream) {
    PROF(ST_GEMM);
    return gemm_nt(x, w, y, M, N, K, (hipStream_t)stream) ? fail(-3, "linear_fwd: gemm failed") : 0;
}
int cffm_linear_bias_fwd(const float* x, const float* w, const float* b, float* y, long M, int N, int K, void* stream) {
    REQUIRE(M >= 0 && N >= 4 && N % 4 == 0 && K >= 1, "linear_bias_fwd: bad sizes (N must be a multiple of 4)");
    if (!M) return 0;
    REQUIRE(x && w && b && y, "linear_bias_fwd: null");
    PROF(ST_GEMM);
    return gemm_split_launch<false, false>(x, w, y, (int)M, N, K, K, K, N, 1, (hipStream_t)stream, b) ? fail(-3, "linear_bias_fwd: gemm failed") : 0;
}
int cffm_linear_bwd_input(const float* dy, const float* w, float* dx, long M, int N, int K, void* stream) {
    PROF(ST_GEMM);
    return gemm_nn(dy, w, dx, M, N, K, (hipStream_t)stream) ? fail(-3, "linear_bwd_input: gemm failed") : 0;
}
int cffm_linear_bwd_weight(const float* dy, const float* x, float* dw, long M, int N, int K, void* stream) {
    ScratchFor pool(stream);
    PROF(ST_GEMM);
    return gemm_tn(dy, x, dw, M, N, K, (hipStream_t)stream) ? fail(-3, "linear_bwd_weight: gemm failed") : 0;
}

int cffm_linear_bwd_weight_group(const cffm_wgrad* problems, int n, void* stream) {
    REQUIRE(problems && n >= 1 && n <= 4, "linear_bwd_weight_group: 1..4 problems");
    static_assert(sizeof(cffm_wgrad) == sizeof(GemmTN), "wgrad layout");
    for (int i = 0; i < n; ++i) REQUIRE(problems[i].dy && problems[i].x && problems[i].dw, "linear_bwd_weight_group: null operand");
    ScratchFor pool(stream);
    PROF(ST_GEMM);
    if (gemm_tn_group((const GemmTN*)problems, n, (hipStream_t)stream)) return fail(-3, "linear_bwd_weight_group: gemm failed");
    CHECK_LAUNCH("linear_bwd_weight_group");
    return 0;
}

// plain fp32 -> split-4 storage (16 bytes = {bf16 hi x4 | bf16 lo x4} of four consecutive floats, cffm_common.h): what the operands of
// cffm_linear_bwd_weight_split are stored as.  n: floats, a multiple of 4
int cffm_split4(const float* src, float* dst, long n, void* stream) {
    REQUIRE(src && dst && n >= 0 && n % 4 == 0, "split4: n must be a multiple of 4");
    if (!n) return 0;
    CFFM_LAUNCH(k_split4, (ew_grid(n / 4)), (256), 0, (hipStream_t)stream, src, dst, n / 4);
    CHECK_LAUNCH("split4");
    return 0;
}
// dw[N,K] = dy[M,N]^T x[M,K] with dy and x in split-4 storage: the LDS-DMA weight-gradient kernel of the block backward (dw_kernels.h) as
// a stage.  N, K multiples of 128
int cffm_linear_bwd_weight_split_group(const cffm_wgrad* problems, int n, void* stream) {
    REQUIRE(problems && n >= 1 && n <= DWD_MAX, "linear_bwd_weight_split_group: 1..%d problems", DWD_MAX);
    for (int i = 0; i < n; ++i)
        REQUIRE(problems[i].dy && problems[i].x && problems[i].dw && problems[i].M >= 1 && problems[i].N % 128 == 0 && problems[i].K % 128 == 0 &&
                problems[i].N >= 128 && problems[i].K >= 128, "linear_bwd_weight_split_group: N, K must be multiples of 128");
    PROF(ST_GEMM);
    int klen;
    static int wgs = -1;      // tuning aid (experiment builds): CFFM_DWD_WGS
    if (wgs < 0) { const char* e = cffm_tune("CFFM_DWD_WGS"); wgs = e ? atoi(e) : 480; if (wgs < 1) wgs = 480; }
    const size_t need = dw_dma_partial_floats((const GemmTN*)problems, n, wgs, &klen);
    float* part = need ? lib_scratch(need) : nullptr;
    REQUIRE(!need || part, "linear_bwd_weight_split_group: scratch allocation failed");
    REQUIRE(!dw_group_dma((const GemmTN*)problems, n, (hipStream_t)stream, part, wgs), "linear_bwd_weight_split_group: launch failed");
    CHECK_LAUNCH("linear_bwd_weight_split_group");
    return 0;
}
int cffm_linear_bwd_weight_split(const float* dy_s, const float* x_s, float* dw, long M, int N, int K, void* stream) {
    const cffm_wgrad pr = {dy_s, x_s, dw, M, N, K};
    return cffm_linear_bwd_weight_split_group(&pr, 1, stream);
}

// row-major fp32 x[R][C] -> T-frag storage (dws_kernels.h: MFMA fragments along the contraction, bf16 hi | lo, rows padded to a multiple of
// 32 with zeros): what the operands of cffm_linear_bwd_weight_tfrag are stored as.  dst: cffm_tfrag_floats(R, C) floats.  A utility: inside
// the block the row-panel kernels leave their operands in this order themselves.
long cffm_tfrag_floats(long R, int C) { return R < 0 || C < 16 ? 0 : ((R + 31) / 32) * 32 * (long)C; }
int cffm_tfrag_pack(const float* x, float* dst, long R, int C, void* stream) {
    REQUIRE(x && dst && R >= 1 && C >= 16 && C % 16 == 0, "tfrag_pack: C must be a multiple of 16");
    const long items = ((R + 31) / 32) * (C / 16) * 64;
    CFFM_LAUNCH(k_tfrag_pack, ((unsigned)((items + 255) / 256)), (256), 0, (hipStream_t)stream, x, R, C, 0, (f32x4*)dst, items);
    CHECK_LAUNCH("tfrag_pack");
    return 0;
}
// dw[N,K] = dy[M,N]^T x[M,K] with dy and x in T-frag storage: the streaming weight-gradient kernel of the block backward (dws_kernels.h) as
// a stage.  N a multiple of 64, K a multiple of 128
int cffm_linear_bwd_weight_tfrag_group(const cffm_wgrad* problems, int n, void* stream) {
    REQUIRE(problems && n >= 1 && n <= DWS_MAX, "linear_bwd_weight_tfrag_group: 1..%d problems", DWS_MAX);
    for (int i = 0; i < n; ++i) REQUIRE(problems[i].dy && problems[i].x && problems[i].dw, "linear_bwd_weight_tfrag_group: null operand");
    PROF(ST_GEMM);
    DwsPlan P;
    REQUIRE(dw_stream_plan((const GemmTN*)problems, n, dw_stream_target(), &P), "linear_bwd_weight_tfrag_group: N must be a multiple of 64, K of 128, operands under 4 GB");
    float* part = P.part_floats ? lib_scratch(P.part_floats) : nullptr;
    REQUIRE(!P.part_floats || part, "linear_bwd_weight_tfrag_group: scratch allocation failed");
    REQUIRE(!dw_group_stream((const GemmTN*)problems, n, (hipStream_t)stream, part, dw_stream_target()), "linear_bwd_weight_tfrag_group: launch failed");
    CHECK_LAUNCH("linear_bwd_weight_tfrag_group");
    return 0;
}
int cffm_linear_bwd_weight_tfrag(const float* dy_t, const float* x_t, float* dw, long M, int N, int K, void* stream) {
    const cffm_wgrad pr = {dy_t, x_t, dw, M, N, K};
    return cffm_linear_bwd_weight_tfrag_group(&pr, 1, stream);
}

// q|k|v Linear feeding the CFM kernels: qkv16[M,768] (f16) = x w^T + b, q third times 32^-0.5 (cffm_transformer.py:374, :528)
int cffm_linear_qkv_fwd(const float* x, const float* w, const float* b, void* qkv16, long M, void* stream) {
    REQUIRE(x && w && b && qkv16, "linear_qkv_fwd: null");
    PROF(ST_GEMM);
    return gemm_nt_qkv16_split(x, w, b, (h16*)qkv16, M, 768, CFFM_C, (hipStream_t)stream) ? fail(-3, "linear_qkv_fwd: gemm failed") : 0;
}

// fused Mlp halves (one launch each)
int cffm_linear_gelu_fwd(const float* x, const float* w, const float* b, float* hraw, float* act, long M, int N, int K, void* stream) {
    PROF(ST_GEMM);
    REQUIRE(N % 4 == 0, "linear_gelu_fwd: N %% 4");
    return gemm_nt_gelu_split(x, w, b, hraw, act, M, N, K, (hipStream_t)stream) ? fail(-3, "linear_gelu_fwd: gemm failed") : 0;
}
int cffm_linear_residual_fwd(const float* x, const float* w, const float* b, const float* res, float* out, long M, int N, int K,
                             void* stream) {
    PROF(ST_GEMM);
    REQUIRE(N % 4 == 0, "linear_residual_fwd: N %% 4");
    return gemm_nt_residual_split(x, w, b, res, out, M, N, K, (hipStream_t)stream) ? fail(-3, "linear_residual_fwd: gemm failed") : 0;
}

int cffm_colsum(const float* a, long rows, int cols, float* out, void* stream) {
    ScratchFor pool(stream);
    PROF(ST_COLSUM);
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(a && out && cols >= 4 && cols % 4 == 0, "colsum: cols must be a multiple of 4");
    const int slices = (int)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS);
    float* part = red_scratch((size_t)slices * cols, st);
    REQUIRE(part, "colsum: scratch allocation failed");
    CFFM_LAUNCH(k_colsum_partial, (slices), (256), 0, st, a, rows, cols, part);
    RedSegs segs;
    segs.nseg = 0;
    seg_add(segs, 0, cols, out, 0);
    reduce_records(part, slices, cols, cols, segs, st);
    CHECK_LAUNCH("colsum");
    return 0;
}

static int residual_ln_impl(const float* xt, long xt_bs, int rows_per_batch, const float* yraw, const float* bproj,
                            const float* gamma, const float* beta, float* x1, float* z2, float* mean, float* rstd,
                            long nrows, int split, void* stream) {
    PROF(ST_RES_LN);
    CFFM_LAUNCH(k_residual_ln, ((unsigned)((nrows + 3) / 4)), (256), 0, (hipStream_t)stream, xt, xt_bs, rows_per_batch, yraw, bproj,
                gamma, beta, x1, z2, mean, rstd, nrows, split);
    CHECK_LAUNCH("residual_ln");
    return 0;
}
int cffm_residual_ln(const float* xt, long xt_bs, int rows_per_batch, const float* yraw, const float* bproj,
                     const float* gamma, const float* beta, float* x1, float* z2, float* mean, float* rstd,
                     long nrows, void* stream) {
    return residual_ln_impl(xt, xt_bs, rows_per_batch, yraw, bproj, gamma, beta, x1, z2, mean, rstd, nrows, 0, stream);
}

int cffm_ln_bwd_residual(const float* x1, const float* mean, const float* rstd, const float* gamma, const float* dz2,
                         const float* dres, float* dx1, float* dgamma, float* dbeta, long nrows, int zero_grads,
                         float* dres_colsum, float* dx1_colsum, void* stream) {
    PROF(ST_LN_BWD);
    hipStream_t st = (hipStream_t)stream;
    const int rpb = LNB_ROWS;
    const int nblk = (int)((nrows + rpb - 1) / rpb);
    float* part = red_scratch((size_t)nblk * 1024, st);
    REQUIRE(part, "ln_bwd_residual: scratch allocation failed");
    CFFM_LAUNCH(k_ln_bwd_residual, (nblk), (256), 0, st, x1, mean, rstd, gamma, dz2, dres, dx1, part, nrows, rpb);
    RedSegs segs;
    segs.nseg = 0;
    seg_add(segs, 0, CFFM_C, dgamma, !zero_grads);
    seg_add(segs, CFFM_C, CFFM_C, dbeta, !zero_grads);
    seg_add(segs, 2 * CFFM_C, CFFM_C, dres_colsum, 0);
    seg_add(segs, 3 * CFFM_C, CFFM_C, dx1_colsum, 0);
    reduce_records(part, nblk, 1024, 1024, segs, st);
    CHECK_LAUNCH("ln_bwd_residual");
    return 0;
}


int cffm_bias_gelu(const float* hraw, const float* b1, float* act, long rows, int cols, void* stream) {
    PROF(ST_GELU);
    REQUIRE(cols % 4 == 0, "bias_gelu: cols %% 4");
    const long n4 = rows * cols / 4;
    CFFM_LAUNCH(k_bias_gelu, (ew_grid(n4)), (256), 0, (hipStream_t)stream, hraw, b1, act, n4, cols / 4);
    CHECK_LAUNCH("bias_gelu");
    return 0;
}
static int gelu_bwd_impl(const float* hraw, const float* b1, float* dact, long rows, int cols, float* db1, int split, void* stream) {
    PROF(ST_GELU_BWD);
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(cols == CFFM_HID, "gelu_bwd: cols must be %d", CFFM_HID);
    const int rpb = GELU_BWD_ROWS;
    const int nblk = (int)((rows + rpb - 1) / rpb);
    float* part = nullptr;
    if (db1) {
        part = red_scratch((size_t)nblk * CFFM_HID, st);
        REQUIRE(part, "gelu_bwd: scratch allocation failed");
    }
    CFFM_LAUNCH(k_gelu_bwd, (nblk), (256), 0, st, hraw, b1, dact, part, rows, split);
    if (db1) {
        RedSegs segs;
        segs.nseg = 0;
        seg_add(segs, 0, CFFM_HID, db1, 0);
        reduce_records(part, nblk, CFFM_HID, CFFM_HID, segs, st);
    }
    CHECK_LAUNCH("gelu_bwd");
    return 0;
}
int cffm_gelu_bwd(const float* hraw, const float* b1, float* dact, long rows, int cols, float* db1, void* stream) {
    return gelu_bwd_impl(hraw, b1, dact, rows, cols, db1, 0, stream);
}
int cffm_residual_out(const float* x1, const float* oraw, const float* b2, float* out, long rows, void* stream) {
    PROF(ST_RES_OUT);
    const long n4 = rows * CFFM_C / 4;
    CFFM_LAUNCH(k_residual_out, (ew_grid(n4)), (256), 0, (hipStream_t)stream, x1, oraw, b2, out, n4);
    CHECK_LAUNCH("residual_out");
    return 0;
}


// ------------------------------------------------------------------------------------------- optimizer
int cffm_adamw_step(const cffm_adamw_chunk* chunks, int nchunks, double lr, double beta1, double beta2, double eps, double weight_decay,
                    int step, void* stream) {
    static_assert(sizeof(cffm_adamw_chunk) == sizeof(AdamwChunk), "chunk layout");
    if (nchunks <= 0) return 0;
    REQUIRE(step >= 1, "adamw: step must be >= 1, got %d", step);
    PROF(ST_ADAMW);
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    CFFM_LAUNCH(k_adamw, ((unsigned)nchunks), (256), 0, (hipStream_t)stream, (const AdamwChunk*)chunks, (float)(1.0 - lr * weight_decay), (float)beta1,
                (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)));
    CHECK_LAUNCH("adamw");
    return 0;
}

int cffm_adamw_step_dev(const cffm_adamw_chunk* chunks, int nchunks, const float* grad_base, double lr, double beta1, double beta2,
                        double eps, double weight_decay, float* state, void* stream) {
    if (nchunks <= 0) return 0;
    REQUIRE(chunks && state, "adamw_step_dev: null");
    PROF(ST_ADAMW);
    CFFM_LAUNCH(k_adamw_tick, (1), (1), 0, (hipStream_t)stream, state, lr, beta1, beta2);
    CFFM_LAUNCH(k_adamw_dev, ((unsigned)nchunks), (256), 0, (hipStream_t)stream, (const AdamwChunk*)chunks, grad_base,
                (float)(1.0 - lr * weight_decay), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps,
                (const float*)state);
    CHECK_LAUNCH("adamw_dev");
    return 0;
}

int cffm_adamw_step_rows(const cffm_adamw_chunk2* chunks, int nchunks, const float* grad_base, float* state, const float* sched,
                         double* consts, int nrows, const int* active_rows, int* ticket, void* stream) {
    static_assert(sizeof(cffm_adamw_chunk2) == sizeof(AdamwChunk2), "chunk layout");
    if (nchunks <= 0) return 0;
    REQUIRE(chunks && state && sched && consts && nrows >= 1, "adamw_step_rows: null table or no rows");
    PROF(ST_ADAMW);
    if (ticket) {
        CFFM_LAUNCH(k_adamw_rows_tick, ((unsigned)nchunks), (256), 0, (hipStream_t)stream, (const AdamwChunk2*)chunks, grad_base, state, sched, consts,
                    nrows, active_rows, ticket);
    } else {
        CFFM_LAUNCH(k_adamw_tick_rows, ((unsigned)((nrows + 63) / 64)), (64), 0, (hipStream_t)stream, state, sched, consts, nrows, active_rows);
        CFFM_LAUNCH(k_adamw_rows, ((unsigned)nchunks), (256), 0, (hipStream_t)stream, (const AdamwChunk2*)chunks, grad_base, (const float*)state, consts);
    }
    CHECK_LAUNCH("adamw_rows");
    return 0;
}

// ------------------------------------------------------------------------------------------- CFFM++ (GTC) stages
int cffm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* z, float* mean, float* rstd,
                       long nrows, void* stream) {
    PROF(ST_LN);
    CFFM_LAUNCH(k_layernorm, ((unsigned)((nrows + 3) / 4)), (256), 0, (hipStream_t)stream, x, gamma, beta, z, mean, rstd, nrows, 0);
    CHECK_LAUNCH("layernorm_fwd");
    return 0;
}

// the prototype attention on the matrix pipe for K <= 128 (gtc_kernels.h).  CFFM_GTC_MFMA=0 (experiment builds): the VALU kernels for every K
CFFM_SWITCH(gtc_mfma_sw, "CFFM_GTC_MFMA", 1)
// workgroups the three matrix-pipe kernels aim for (a wave walks ceil(tiles / (4 x workgroups)) tiles)
#ifndef GTM_WGS_FWD
#define GTM_WGS_FWD 512
#endif
#ifndef GTM_WGS_DQ
#define GTM_WGS_DQ 512
#endif
#ifndef GTM_WGS_DKV
#define GTM_WGS_DKV 256
#endif
int cffm_gtc_attn_fwd(const float* q_raw, const float* q_b, const float* kv_raw, const float* kv_b, float* o, float* lse,
                      int B, int T, int K, void* stream) {
    PROF(ST_GTC_FWD);
    REQUIRE(K >= 1 && K <= 256, "gtc_attn_fwd: K=%d outside 1..256", K);
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        REQUIRE(hipFuncSetAttribute((const void*)k_gtc_attn_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, gtf_lds(256)) == hipSuccess, "gtc_attn_fwd: LDS grant failed");
        granted = true;
    }
#endif
    if (K <= 128 && gtc_mfma_sw()) {
        // the matrix-pipe form (three-pass bf16 split): ~512 workgroups of four waves, each wave a run of 16-token tiles
        const int tiles = (T + 15) / 16;
        int tpw = (tiles * CFFM_HEADS * B + 4 * GTM_WGS_FWD - 1) / (4 * GTM_WGS_FWD);
        if (tpw < 1) tpw = 1;
        const unsigned gx = (unsigned)((tiles + 4 * tpw - 1) / (4 * tpw));
        const int U = (K + 31) / 32;
        hipStream_t st = (hipStream_t)stream;
        f32x4* fr = (f32x4*)lib_scratch((size_t)B * CFFM_HEADS * gtm_frag_units(U) * 256);
        REQUIRE(fr, "gtc_attn_fwd: scratch allocation failed");
#define GTM_FWD(U_)                                                                                                                          \
        do {                                                                                                                             \
            if (U_ > 1) CFFM_LAUNCH(k_gtc_pack_frags<U_>, (CFFM_HEADS, B, 2 * U_), (256), 0, st, kv_raw, kv_b, fr, K);                   \
            CFFM_LAUNCH(k_gtc_attn_fwd_mfma<U_>, (gx, CFFM_HEADS, B), (256), 0, st, q_raw, q_b, (const f32x4*)(U_ > 1 ? fr : nullptr), kv_raw, kv_b, o, lse, T, K, tpw);   \
        } while (0)
        if (U == 1) GTM_FWD(1); else if (U == 2) GTM_FWD(2); else if (U == 3) GTM_FWD(3); else GTM_FWD(4);
#undef GTM_FWD
        CHECK_LAUNCH("gtc_attn_fwd");
        return 0;
    }
    CFFM_LAUNCH(k_gtc_attn_fwd, ((T + GTC_TOK - 1) / GTC_TOK, CFFM_HEADS, B), (256), (size_t)gtf_lds(K), (hipStream_t)stream, q_raw, q_b, kv_raw, kv_b, o,
                lse, T, K);
    CHECK_LAUNCH("gtc_attn_fwd");
    return 0;
}

int cffm_gtc_attn_bwd(const float* q_raw, const float* q_b, const float* kv_raw, const float* kv_b, const float* o,
                      const float* dout, const float* lse, float* dq_raw, float* dkv, int B, int T, int K, void* stream) {
    PROF(ST_GTC_BWD);
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(K >= 1 && K <= 256 && B >= 1 && T >= 1, "gtc_attn_bwd: K=%d outside 1..256", K);
    if (K <= 128 && gtc_mfma_sw()) {
        // the matrix-pipe form (gtc_kernels.h): dq per 16-token tile, then dKc / dVc per pair of tiles into one record per workgroup
        const int tiles = (T + 15) / 16, pairs = (T + 31) / 32, U = (K + 31) / 32;
        int tpw = (tiles * CFFM_HEADS * B + 4 * GTM_WGS_DQ - 1) / (4 * GTM_WGS_DQ), ppw = (pairs * CFFM_HEADS * B + 4 * GTM_WGS_DKV - 1) / (4 * GTM_WGS_DKV);
        if (tpw < 1) tpw = 1;
        if (ppw < 1) ppw = 1;
        const unsigned g1 = (unsigned)((tiles + 4 * tpw - 1) / (4 * tpw)), g2 = (unsigned)((pairs + 4 * ppw - 1) / (4 * ppw));
        const size_t nD = ((size_t)B * T * CFFM_HEADS + 63) / 64 * 64;
        const size_t nrecf = ((size_t)B * CFFM_HEADS * g2 * K * 64 + 63) / 64 * 64;
        float* scr = lib_scratch(nD + nrecf + (size_t)B * CFFM_HEADS * gtm_frag_units(U) * 256);
        REQUIRE(scr, "gtc_attn_bwd: scratch allocation failed");
        float* Dbuf = scr;
        float* rec = scr + nD;
        f32x4* fr = (f32x4*)(scr + nD + nrecf);
#ifndef CFFM_EMU
        static bool granted_m = false;
        if (!granted_m) {
            REQUIRE(hipFuncSetAttribute((const void*)k_gtc_attn_bwd_dkv_mfma<3>, hipFuncAttributeMaxDynamicSharedMemorySize, gtm_bwd_lds(4)) == hipSuccess &&
                    hipFuncSetAttribute((const void*)k_gtc_attn_bwd_dkv_mfma<4>, hipFuncAttributeMaxDynamicSharedMemorySize, gtm_bwd_lds(4)) == hipSuccess, "gtc_attn_bwd: LDS grant failed");
            granted_m = true;
        }
#endif
#define GTM_BWD(U_)                                                                                                                                              \
        do {                                                                                                                                                 \
            const f32x4* fq = U_ > 1 ? (const f32x4*)fr : nullptr;        /* U = 1: the kernels make their own copy (gtm_pack_local) */                        \
            if (U_ > 1) CFFM_LAUNCH(k_gtc_pack_frags<U_>, (CFFM_HEADS, B, 2 * U_), (256), 0, st, kv_raw, kv_b, fr, K);                                       \
            CFFM_LAUNCH(k_gtc_attn_bwd_dq_mfma<U_>, (g1, CFFM_HEADS, B), (256), (size_t)(8 * U_ * 1024), st, q_raw, q_b, fq, kv_raw, kv_b, dout, lse, dq_raw, Dbuf, T, K, tpw); \
            CFFM_LAUNCH(k_gtc_attn_bwd_dkv_mfma<U_>, (g2, CFFM_HEADS, B), (256), (size_t)gtm_bwd_lds(U_), st, q_raw, q_b, fq, kv_raw, kv_b, dout, lse, (const float*)Dbuf, rec, T, K, ppw); \
        } while (0)
        if (U == 1) GTM_BWD(1); else if (U == 2) GTM_BWD(2); else if (U == 3) GTM_BWD(3); else GTM_BWD(4);
        (void)o;
#undef GTM_BWD
        CFFM_LAUNCH(k_gtc_dkv_sum, (K, CFFM_HEADS, B), (64), 0, st, (const float*)rec, (int)g2, K, dkv);
        CHECK_LAUNCH("gtc_attn_bwd");
        return 0;
    }
    const int per = gtb_tok(K) * GTB_CHUNKS, nwg = (T + per - 1) / per;
    float* rec = lib_scratch((size_t)B * CFFM_HEADS * nwg * K * 64);       // one [K][64] record per workgroup: k_gtc_dkv_sum adds them in order
    REQUIRE(rec, "gtc_attn_bwd: scratch allocation failed");
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        REQUIRE(hipFuncSetAttribute((const void*)k_gtc_attn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess, "gtc_attn_bwd: LDS grant failed");
        granted = true;
    }
#endif
    CFFM_LAUNCH(k_gtc_attn_bwd, (nwg, CFFM_HEADS, B), (256), (size_t)gtb_lds(K), st, q_raw, q_b, kv_raw, kv_b, o, dout, lse, dq_raw, rec, T, K);
    CFFM_LAUNCH(k_gtc_dkv_sum, (K, CFFM_HEADS, B), (64), 0, st, (const float*)rec, nwg, K, dkv);
    CHECK_LAUNCH("gtc_attn_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------- fused row-panel stages
// CFFM_PANEL=0 keeps the round-2 sequence of tiled GEMMs + row kernels (A/B measurements)
// CFFM_STORE_ACT=0: the fused forward does NOT store gelu(hraw + b1); the fc2 weight gradient re-applies bias + GELU to hraw while it
// stages its tiles (29.5 MB less written per block, but the GELU then sits in the weight-gradient GEMM's staging path: measured
// 0.8914 vs 0.8843 ms per step without / with the stored activation, means of three alternating runs -- storing is the default)
CFFM_SWITCH(store_act, "CFFM_STORE_ACT", 1)
CFFM_SWITCH(panel_on, "CFFM_PANEL", 1)
#ifndef MLP_MT
#define MLP_MT 2
#endif
#define MLP_D 4
static int mlp_lds_grant() {
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute((const void*)k_mlp_fwd<MLP_MT, MLP_D>, hipFuncAttributeMaxDynamicSharedMemorySize, PNL_FUSED_LDS(MLP_MT)) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_mlp_bwd<MLP_MT, MLP_D>, hipFuncAttributeMaxDynamicSharedMemorySize, PNL_FUSED_LDS(MLP_MT)) != hipSuccess)
            return -1;
        granted = true;
    }
#endif
    return 0;
}
// q|k|v Linear and its input gradient as row-panel GEMMs (weights in fragment order): rows per workgroup (32 or 48) chosen so that
// the grid wastes the least of its last round on 256 CUs
static int panel_mt(long M) {
    const long w2 = (M + 31) / 32, w3 = (M + 47) / 48;
    return ((w3 + 255) / 256) * 3 < ((w2 + 255) / 256) * 2 ? 3 : 2;
}
extern "C++" {
template <int MT, int NTW, bool A_PRE, int EPI>
static int panel_gemm_launch(const float* A, int lda, long M, int K, const float* wf, float* Cout, int ldc, const float* bias, void* aux,
                             hipStream_t st, float* colrec = nullptr, float* a_t = nullptr) {
    auto kern = k_panel_gemm<MT, NTW, 2, 4, A_PRE, EPI>;
#ifndef CFFM_EMU
    static bool granted = false;
    if (!granted) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, PNL_LDS(MT)) != hipSuccess) return -1;
        granted = true;
    }
#endif
    CFFM_LAUNCH(kern, ((unsigned)((M + 16 * MT - 1) / (16 * MT))), (PNL_THREADS), PNL_LDS(MT), st, A, lda, (int)M, K, (const f32x4*)wf, Cout, ldc, bias, aux, colrec, (f32x4*)a_t);
    return 0;
}
}  // extern "C++"
// qkv16[M][768] (f16) = f16((x W^T + b) [q third * 32^-0.5]), x in split-4 storage, W fragment-ordered (forward form)
// x_t (or NULL): T-frag copy of x for the streaming weight gradient
static int panel_qkv_fwd(const float* x_s, const float* wf, const float* b, h16* qkv16, long M, hipStream_t st, float* x_t = nullptr) {
    return panel_mt(M) == 3 ? panel_gemm_launch<3, 6, true, 3>(x_s, 256, M, 256, wf, nullptr, 768, b, qkv16, st, nullptr, x_t)
                            : panel_gemm_launch<2, 6, true, 3>(x_s, 256, M, 256, wf, nullptr, 768, b, qkv16, st, nullptr, x_t);
}
// dx[M][256] = dqkv[M][768] W, W fragment-ordered (input-gradient form); colrec (or NULL): panel_qkv_records(M) records of 768
// column sums of dqkv (the q|k|v bias gradient before its reduction)
static long panel_qkv_records(long M) { return (M + 16 * panel_mt(M) - 1) / (16 * panel_mt(M)); }
// dqkv_t (or NULL): T-frag copy of dqkv for the streaming weight gradient
static int panel_qkv_dx(const float* dqkv, const float* wfn, float* dx, long M, hipStream_t st, float* colrec = nullptr, float* dqkv_t = nullptr) {
    return panel_mt(M) == 3 ? panel_gemm_launch<3, 2, false, 0>(dqkv, 768, M, 768, wfn, dx, 256, nullptr, nullptr, st, colrec, dqkv_t)
                            : panel_gemm_launch<2, 2, false, 0>(dqkv, 768, M, 768, wfn, dx, 256, nullptr, nullptr, st, colrec, dqkv_t);
}
// CFFM_PANEL_QKV=0: the q|k|v Linear keeps the tiled GEMMs (A/B: 0.8843 tiled vs 0.8729 ms per step as row panels, means of three
// alternating runs with the activation stored)
CFFM_SWITCH(qkv_colrec_on, "CFFM_QKV_COLREC", 1)
// The block's four weight gradients by the streaming kernel (dws_kernels.h): their operands in T-frag storage, written by the row-panel
// kernels that have them in LDS anyway.  CFFM_DW_STREAM=0 (experiment builds): the LDS-staged group on split-4 / fp32 operands, as before.
#ifndef CFFM_DW_STREAM_DEFAULT
#define CFFM_DW_STREAM_DEFAULT 0
#endif
static int g_dw_stream = CFFM_DW_STREAM_DEFAULT;
static int dw_stream_sw() { return g_dw_stream; }
// Process-wide choice between the two forms of the block's weight gradients (both tested): 1 = streaming kernel on T-frag operands, 0 = the
// LDS-staged group.  The forward leaves the operands in the form the backward will read: do not change it between a forward and its backward.
int cffm_dw_stream(int on) { const int was = g_dw_stream; if (on >= 0) g_dw_stream = on != 0; return was; }
CFFM_SWITCH(panel_qkv_sw, "CFFM_PANEL_QKV", 1)
static constexpr_or_not int panel_qkv_on() { return panel_qkv_sw() && panel_on(); }

long cffm_mlp_records(long NP) { return (NP + 16 * MLP_MT - 1) / (16 * MLP_MT); }

int cffm_panel_pack_weight(const float* w, int N, int K, int form, float* w_frag, void* stream) {
    REQUIRE(w && w_frag && N >= 16 && K >= 16 && N % 32 == 0 && K % 32 == 0 && (form == 0 || form == 1), "panel_pack_weight: bad arguments");
    CFFM_LAUNCH(k_pnl_pack_weight, ((unsigned)(((long)N * K / 8 + 255) / 256)), (256), 0, (hipStream_t)stream, w, N, K, form, (f32x4*)w_frag);
    CHECK_LAUNCH("panel_pack_weight");
    return 0;
}

// z2s / acts: split-4 copies (what the LDS-staged weight-gradient kernels read), ao_t / z2_t / act_t: T-frag copies (what the streaming one
// reads, cffm_tfrag_floats(NP, 256 | 256 | 1024) floats each); every one of the five may be NULL
int cffm_mlp_fwd_tfrag(const float* ao, const float* xt, long xt_bs, int rows_per_batch, const float* wp_f, const float* w1_f, const float* w2_f,
                       const float* bp, const float* b1, const float* b2, const float* g2, const float* be2, float* x1, float* z2s, float* mean2,
                       float* rstd2, float* hraw, float* acts, float* x2, float* ao_t, float* z2_t, float* act_t, long NP, void* stream) {
    REQUIRE(NP >= 0 && NP < (1L << 21) && rows_per_batch >= 1, "mlp_fwd: bad sizes");
    if (!NP) return 0;
    REQUIRE(ao && xt && wp_f && w1_f && w2_f && bp && b1 && b2 && g2 && be2 && x1 && mean2 && rstd2 && hraw && x2, "mlp_fwd: null");
    REQUIRE(!mlp_lds_grant(), "mlp_fwd: LDS grant failed");
    PROF2(ST_MLP_FWD);
    MlpFwdArgs a;
    a.ao = ao; a.xt = xt; a.xt_bs = xt_bs; a.rows_per_batch = rows_per_batch;
    a.wp = (const f32x4*)wp_f; a.w1 = (const f32x4*)w1_f; a.w2 = (const f32x4*)w2_f;
    a.bp = bp; a.b1 = b1; a.b2 = b2; a.g2 = g2; a.be2 = be2;
    a.x1 = x1; a.z2s = z2s; a.mean2 = mean2; a.rstd2 = rstd2; a.hraw = hraw; a.acts = acts; a.x2 = x2; a.NP = (int)NP;
    a.ao_t = (f32x4*)ao_t; a.z2_t = (f32x4*)z2_t; a.act_t = (f32x4*)act_t;
    CFFM_LAUNCH((k_mlp_fwd<MLP_MT, MLP_D>), ((unsigned)cffm_mlp_records(NP)), (PNL_THREADS), PNL_FUSED_LDS(MLP_MT), (hipStream_t)stream, a);
    CHECK_LAUNCH("mlp_fwd");
    return 0;
}
int cffm_mlp_fwd(const float* ao, const float* xt, long xt_bs, int rows_per_batch, const float* wp_f, const float* w1_f, const float* w2_f,
                 const float* bp, const float* b1, const float* b2, const float* g2, const float* be2, float* x1, float* z2s, float* mean2,
                 float* rstd2, float* hraw, float* acts, float* x2, long NP, void* stream) {
    REQUIRE(z2s || !NP, "mlp_fwd: null");   // (acts may be NULL: not stored)
    return cffm_mlp_fwd_tfrag(ao, xt, xt_bs, rows_per_batch, wp_f, w1_f, w2_f, bp, b1, b2, g2, be2, x1, z2s, mean2, rstd2, hraw, acts, x2, nullptr, nullptr,
                              nullptr, NP, stream);
}

// dhs: split-4 copy of dh; dout_t / dh_t / dx1_t: T-frag copies of dout, dh, dx1 (cffm_tfrag_floats(NP, 256 | 1024 | 256) floats); each may be NULL
int cffm_mlp_bwd_tfrag(const float* dout, const float* hraw, const float* b1, const float* x1, const float* mean2, const float* rstd2,
                       const float* g2, const float* w2_n, const float* w1_n, const float* wp_n, float* dhs, float* dx1, float* dao, float* dg2,
                       float* dbe2, float* db1, float* db2, float* dbp, float* dout_t, float* dh_t, float* dx1_t, long NP, void* stream) {
    REQUIRE(NP >= 0 && NP < (1L << 21), "mlp_bwd: bad sizes");
    if (!NP) return 0;
    REQUIRE(dout && hraw && b1 && x1 && mean2 && rstd2 && g2 && w2_n && w1_n && wp_n && dx1 && dao, "mlp_bwd: null");
    REQUIRE(!mlp_lds_grant(), "mlp_bwd: LDS grant failed");
    PROF2(ST_MLP_BWD);
    hipStream_t st = (hipStream_t)stream;
    const int nrec = (int)cffm_mlp_records(NP);
    float* rec = red_scratch((size_t)nrec * 2048, st);
    REQUIRE(rec, "mlp_bwd: scratch allocation failed");
    MlpBwdArgs a;
    a.dout = dout; a.hraw = hraw; a.b1 = b1; a.x1 = x1; a.mean2 = mean2; a.rstd2 = rstd2; a.g2 = g2;
    a.w2n = (const f32x4*)w2_n; a.w1n = (const f32x4*)w1_n; a.wpn = (const f32x4*)wp_n;
    a.dhs = dhs; a.dx1 = dx1; a.dao = dao; a.rec_b1 = rec; a.rec_ln = rec + (size_t)nrec * 1024; a.NP = (int)NP;
    a.dout_t = (f32x4*)dout_t; a.dh_t = (f32x4*)dh_t; a.dx1_t = (f32x4*)dx1_t;
    CFFM_LAUNCH((k_mlp_bwd<MLP_MT, MLP_D>), ((unsigned)nrec), (PNL_THREADS), PNL_FUSED_LDS(MLP_MT), st, a);
    RedSegs s1, s2;
    s1.nseg = 0;
    seg_add(s1, 0, CFFM_HID, db1, 0);
    reduce_records(a.rec_b1, nrec, CFFM_HID, CFFM_HID, s1, st);
    s2.nseg = 0;
    seg_add(s2, 0, CFFM_C, dg2, 0);
    seg_add(s2, CFFM_C, CFFM_C, dbe2, 0);
    seg_add(s2, 2 * CFFM_C, CFFM_C, db2, 0);
    seg_add(s2, 3 * CFFM_C, CFFM_C, dbp, 0);
    reduce_records(a.rec_ln, nrec, 1024, 1024, s2, st);
    CHECK_LAUNCH("mlp_bwd");
    return 0;
}
int cffm_mlp_bwd(const float* dout, const float* hraw, const float* b1, const float* x1, const float* mean2, const float* rstd2,
                 const float* g2, const float* w2_n, const float* w1_n, const float* wp_n, float* dhs, float* dx1, float* dao, float* dg2,
                 float* dbe2, float* db1, float* db2, float* dbp, long NP, void* stream) {
    REQUIRE(dhs || !NP, "mlp_bwd: null");
    return cffm_mlp_bwd_tfrag(dout, hraw, b1, x1, mean2, rstd2, g2, w2_n, w1_n, wp_n, dhs, dx1, dao, dg2, dbe2, db1, db2, dbp, nullptr, nullptr, nullptr, NP,
                              stream);
}

// ------------------------------------------------------------------------------------------- CFFM++ (GTC) block, fused (round 5)
// SwinTransformerBlock_cluster.forward (pvt/swin_transformer_2d.py:605-665) with shift 0 and only_use_cluster_center_as_context (:216):
//   z = LN1(x), cn = LN1(centers) (:619-622);  q = (z Wq^T + bq) 32^-0.5, q third of `qkv` only (:219-226);  [Kc|Vc] = cn Wkv^T + bkv
//   (:223-224);  o = softmax_K(q Kc^T) Vc (:232-257);  x1 = x + o Wpc^T + bpc (:258, :662);  x2 = x1 + fc2(GELU(fc1(LN2(x1)))) (:663).
// Round 1-4 sequenced ~13 forward / ~20 backward stage launches from Python (vss_cffm_amd/ops.py).  Now ONE entry point per
// direction, built from the base head's kernels: the q projection and its input gradient are row-panel GEMMs (k_panel_gemm), everything
// from proj_cluster to the block output is the fused Mlp launch of the base block (k_mlp_fwd / k_mlp_bwd: same shapes), the four large
// weight gradients are one grouped launch (k_gemm_group_tt), every bias / norm gradient is a record reduction in one launch.
struct GtcWs {     // float offsets into the caller's workspace: what the forward keeps for the backward, then the backward's temporaries
    long z, mean1, rstd1, cn, cmean, crstd, qraw, kvraw, ao, lse, x1, z2, mean2, rstd2, hraw, act, wf, wn, z_t, ao_t, saved;
    long dh, dx1, dao, dq, dz, dkv, dcn, dout_t, dx1_t, dq_t, total;
};
// The block's four large weight gradients by the streaming kernel (dws_kernels.h) -- everything of this block runs on ONE stream, so what the
// kernel saves alone (51 -> 38 us) is saved in the step.  z2 / act / dh are then kept in T-frag storage only, z / ao / dout / dx1 / dq get
// T-frag copies from the row-panel kernels that stage them.  GTC_DW_STREAM=0: the LDS-staged group on split-4 / fp32 operands.
#ifndef GTC_DW_STREAM
#define GTC_DW_STREAM 1
#endif
#define GTC_WFLOATS (2 * 256 * 256 + 2 * 1024 * 256)     // q third | proj_cluster | fc1 | fc2 in fragment order
static GtcWs gtc_ws_layout(long nt, long nk) {
    GtcWs w;
    long p = 0;
    w.z = p; p += up(nt * CFFM_C);
    w.mean1 = p; p += up(nt); w.rstd1 = p; p += up(nt);
    w.cn = p; p += up(nk * CFFM_C); w.cmean = p; p += up(nk); w.crstd = p; p += up(nk);
    w.qraw = p; p += up(nt * CFFM_C); w.kvraw = p; p += up(nk * 512);
    w.ao = p; p += up(nt * CFFM_C); w.lse = p; p += up(nt * CFFM_HEADS);
    const long nt32 = (nt + 31) / 32 * 32;      // T-frag storage pads the token rows to whole k-steps of 32
    w.x1 = p; p += up(nt * CFFM_C); w.z2 = p; p += up(nt32 * CFFM_C); w.mean2 = p; p += up(nt); w.rstd2 = p; p += up(nt);
    w.hraw = p; p += up(nt * CFFM_HID); w.act = p; p += up(nt32 * CFFM_HID);
    w.wf = p; p += up(GTC_WFLOATS); w.wn = p; p += up(GTC_WFLOATS);
    w.z_t = p; p += up(nt32 * CFFM_C); w.ao_t = p; p += up(nt32 * CFFM_C);
    w.saved = p;
    w.dh = p; p += up(nt32 * CFFM_HID);
    w.dx1 = p; p += up(nt * CFFM_C); w.dao = p; p += up(nt * CFFM_C); w.dq = p; p += up(nt * CFFM_C); w.dz = p; p += up(nt * CFFM_C);
    w.dkv = p; p += up(nk * 512); w.dcn = p; p += up(nk * CFFM_C);
    w.dout_t = p; p += up(nt32 * CFFM_C); w.dx1_t = p; p += up(nt32 * CFFM_C); w.dq_t = p; p += up(nt32 * CFFM_C);
    w.total = p;
    return w;
}
long cffm_gtc_ws_floats(int B, int T, int K) { return (B < 1 || T < 1 || K < 1) ? -1 : gtc_ws_layout((long)B * T, (long)B * K).total; }

static int gtc_pack(const cffm_gtc_params* p, float* wfrag, int form, void* stream) {
    PnlPackJobs J;
    const float* w[4] = {p->qkv_w /* rows 0..255 of qkv.weight: the q third */, p->proj_w, p->fc1_w, p->fc2_w};
    const int N[4] = {256, 256, CFFM_HID, 256}, K[4] = {256, 256, 256, CFFM_HID};
    const long off[4] = {0, 256 * 256, 2 * 256 * 256, 2 * 256 * 256 + CFFM_HID * 256};
    long end = 0;
    for (int j = 0; j < 4; ++j) {
        J.w[j] = w[j]; J.dst[j] = (f32x4*)(wfrag + off[j]); J.N[j] = N[j]; J.K[j] = K[j];
        end += ((long)N[j] * K[j] / 8 + 255) / 256 * 256;       // whole workgroups per weight
        J.end[j] = end;
    }
    J.n = 4; J.nn = form;
    CFFM_LAUNCH(k_pnl_pack_weights, ((unsigned)(end / 256)), (256), 0, (hipStream_t)stream, J);
    CHECK_LAUNCH("gtc weight pack");
    return 0;
}
static int gtc_check(const cffm_gtc_params* p, const void* a, const void* b, const void* c, const void* ws, int B, int T, int K, const char* who) {
    REQUIRE(p && a && b && c && ws && B >= 1 && T >= 1 && K >= 1 && K <= 256, "%s: bad arguments (1 <= K <= 256)", who);
    REQUIRE(p->norm1_w && p->norm1_b && p->qkv_w && p->qkv_b && p->kv_w && p->kv_b && p->proj_w && p->proj_b && p->norm2_w && p->norm2_b &&
            p->fc1_w && p->fc1_b && p->fc2_w && p->fc2_b, "%s: null parameter", who);
    return 0;
}
int cffm_gtc_block_forward(const cffm_gtc_params* p, const float* x, const float* centers, float* out, float* ws, int B, int T, int K,
                           void* stream) {
    TRY(gtc_check(p, x, centers, out, ws, B, T, K, "gtc_block_forward"));
    hipStream_t st = (hipStream_t)stream;
    const long nt = (long)B * T, nk = (long)B * K;
    const GtcWs W = gtc_ws_layout(nt, nk);
    float* wf = ws + W.wf;
    TRY(gtc_pack(p, wf, 0, stream));
    {
        PROF(ST_LN);
        CFFM_LAUNCH(k_layernorm, ((unsigned)((nt + 3) / 4)), (256), 0, st, x, p->norm1_w, p->norm1_b, ws + W.z, ws + W.mean1, ws + W.rstd1, nt, 1);
        CFFM_LAUNCH(k_layernorm, ((unsigned)((nk + 3) / 4)), (256), 0, st, centers, p->norm1_w, p->norm1_b, ws + W.cn, ws + W.cmean, ws + W.crstd, nk, 0);
        CHECK_LAUNCH("gtc layernorm");
    }
    {
        PROF(ST_GEMM);
        float* z_t = GTC_DW_STREAM ? ws + W.z_t : nullptr;
        const int rc = panel_mt(nt) == 3 ? panel_gemm_launch<3, 2, true, 0>(ws + W.z, 256, nt, 256, wf, ws + W.qraw, 256, nullptr, nullptr, st, nullptr, z_t)
                                         : panel_gemm_launch<2, 2, true, 0>(ws + W.z, 256, nt, 256, wf, ws + W.qraw, 256, nullptr, nullptr, st, nullptr, z_t);
        REQUIRE(!rc, "gtc_block_forward: q gemm failed");
        REQUIRE(!gemm_nt(ws + W.cn, p->kv_w, ws + W.kvraw, nk, 512, 256, st), "gtc_block_forward: kv gemm failed");
    }
    TRY(cffm_gtc_attn_fwd(ws + W.qraw, p->qkv_b, ws + W.kvraw, p->kv_b, ws + W.ao, ws + W.lse, B, T, K, stream));
    if (GTC_DW_STREAM)
        TRY(cffm_mlp_fwd_tfrag(ws + W.ao, x, 0, (int)nt, wf + 256 * 256, wf + 2 * 256 * 256, wf + 2 * 256 * 256 + CFFM_HID * 256, p->proj_b, p->fc1_b, p->fc2_b,
                               p->norm2_w, p->norm2_b, ws + W.x1, nullptr, ws + W.mean2, ws + W.rstd2, ws + W.hraw, nullptr, out, ws + W.ao_t, ws + W.z2,
                               ws + W.act, nt, stream));
    else
    TRY(cffm_mlp_fwd(ws + W.ao, x, 0, (int)nt, wf + 256 * 256, wf + 2 * 256 * 256, wf + 2 * 256 * 256 + CFFM_HID * 256, p->proj_b, p->fc1_b, p->fc2_b,
                     p->norm2_w, p->norm2_b, ws + W.x1, ws + W.z2, ws + W.mean2, ws + W.rstd2, ws + W.hraw, ws + W.act, out, nt, stream));
    return 0;
}

// every gradient of cffm_gtc_grads is written (qkv_w / qkv_b: rows / entries >= 256 -- the unused k, v thirds, :216 -- are zeroed)
int cffm_gtc_block_backward(const cffm_gtc_params* p, const cffm_gtc_grads* g, const float* x, const float* centers, const float* dout,
                            float* dx, float* dcenters, float* ws, int B, int T, int K, void* stream) {
    TRY(gtc_check(p, x, centers, dout, ws, B, T, K, "gtc_block_backward"));
    REQUIRE(g && dx && dcenters && g->norm1_w && g->norm1_b && g->qkv_w && g->qkv_b && g->kv_w && g->kv_b && g->proj_w && g->proj_b && g->norm2_w &&
            g->norm2_b && g->fc1_w && g->fc1_b && g->fc2_w && g->fc2_b, "gtc_block_backward: null gradient");
    hipStream_t st = (hipStream_t)stream;
    const long nt = (long)B * T, nk = (long)B * K;
    const GtcWs W = gtc_ws_layout(nt, nk);
    float* wn = ws + W.wn;
    TRY(gtc_pack(p, wn, 1, stream));
#ifdef CFFM_EMU
    memset(g->qkv_w + 256 * 256, 0, sizeof(float) * 512 * 256);
    memset(g->qkv_b + 256, 0, sizeof(float) * 512);
#else
    REQUIRE(hipMemsetAsync(g->qkv_w + 256 * 256, 0, sizeof(float) * 512 * 256, st) == hipSuccess &&
            hipMemsetAsync(g->qkv_b + 256, 0, sizeof(float) * 512, st) == hipSuccess, "gtc_block_backward: memset failed");
#endif
    RedScope reductions(st);      // the bias / norm record reductions below run as ONE launch (finish())
    // x2 = x1 + Mlp(LN2(x1)), x1 = x + o Wpc^T + bpc: the fused input-gradient chain of the base block
    if (GTC_DW_STREAM)
        TRY(cffm_mlp_bwd_tfrag(dout, ws + W.hraw, p->fc1_b, ws + W.x1, ws + W.mean2, ws + W.rstd2, p->norm2_w, wn + 2 * 256 * 256 + CFFM_HID * 256,
                               wn + 2 * 256 * 256, wn + 256 * 256, nullptr, ws + W.dx1, ws + W.dao, g->norm2_w, g->norm2_b, g->fc1_b, g->fc2_b, g->proj_b,
                               ws + W.dout_t, ws + W.dh, ws + W.dx1_t, nt, stream));
    else
    TRY(cffm_mlp_bwd(dout, ws + W.hraw, p->fc1_b, ws + W.x1, ws + W.mean2, ws + W.rstd2, p->norm2_w, wn + 2 * 256 * 256 + CFFM_HID * 256, wn + 2 * 256 * 256,
                     wn + 256 * 256, ws + W.dh, ws + W.dx1, ws + W.dao, g->norm2_w, g->norm2_b, g->fc1_b, g->fc2_b, g->proj_b, nt, stream));
    TRY(cffm_gtc_attn_bwd(ws + W.qraw, p->qkv_b, ws + W.kvraw, p->kv_b, ws + W.ao, ws + W.dao, ws + W.lse, ws + W.dq, ws + W.dkv, B, T, K, stream));
    {
        // dz = dq Wq (row panels; the q bias gradient = column sums of dq out of the staging registers)
        PROF(ST_GEMM);
        const int mt = panel_mt(nt);
        const long nrec = (nt + 16 * mt - 1) / (16 * mt);
        float* qrec = red_scratch((size_t)nrec * 256, st);
        REQUIRE(qrec, "gtc_block_backward: scratch allocation failed");
        float* dq_t = GTC_DW_STREAM ? ws + W.dq_t : nullptr;
        const int rc = mt == 3 ? panel_gemm_launch<3, 2, false, 0>(ws + W.dq, 256, nt, 256, wn, ws + W.dz, 256, nullptr, nullptr, st, qrec, dq_t)
                               : panel_gemm_launch<2, 2, false, 0>(ws + W.dq, 256, nt, 256, wn, ws + W.dz, 256, nullptr, nullptr, st, qrec, dq_t);
        REQUIRE(!rc, "gtc_block_backward: q input-gradient gemm failed");
        RedSegs segs;
        segs.nseg = 0;
        seg_add(segs, 0, 256, g->qkv_b, 0);
        reduce_records(qrec, (int)nrec, 256, 256, segs, st);
    }
    {
        // the four large weight gradients in one grouped launch (z, z2, act and dh are in split-4 storage)
        PROF(ST_GEMM); PROF2(ST_G_DW);
        const cffm_wgrad wg4[4] = {{ws + W.dq, ws + W.z, g->qkv_w, nt, 256, 256}, {ws + W.dh, ws + W.z2, g->fc1_w, nt, CFFM_HID, 256},
                                   {dout, ws + W.act, g->fc2_w, nt, 256, CFFM_HID}, {ws + W.dx1, ws + W.ao, g->proj_w, nt, 256, 256}};
        const GemmTNPre pre4[4] = {{0, 1, nullptr}, {1, 1, nullptr}, {0, 1, nullptr}, {0, 0, nullptr}};
        if (GTC_DW_STREAM) {
            const cffm_wgrad wt4[4] = {{ws + W.dq_t, ws + W.z_t, g->qkv_w, nt, 256, 256}, {ws + W.dh, ws + W.z2, g->fc1_w, nt, CFFM_HID, 256},
                                       {ws + W.dout_t, ws + W.act, g->fc2_w, nt, 256, CFFM_HID}, {ws + W.dx1_t, ws + W.ao_t, g->proj_w, nt, 256, 256}};
            DwsPlan P;
            REQUIRE(dw_stream_plan((const GemmTN*)wt4, 4, dw_stream_target(), &P), "gtc_block_backward: weight-gradient plan failed");
            float* part = P.part_floats ? lib_scratch(P.part_floats) : nullptr;
            REQUIRE(!P.part_floats || part, "gtc_block_backward: scratch allocation failed");
            // (on the library's side stream, beside the prototype side's chain of small launches below: measured 0.268 vs 0.262 ms at K = 8, 0.296 vs
            //  0.293 at K = 100 -- the fork and the join cost a replayed graph more than the overlap returns; it stays on the caller's stream)
            REQUIRE(!dw_group_stream((const GemmTN*)wt4, 4, st, part, dw_stream_target()), "gtc_block_backward: weight-gradient gemm failed");
        } else
        REQUIRE(!gemm_tn_group((const GemmTN*)wg4, 4, st, pre4, lib_scratch, 480), "gtc_block_backward: weight-gradient gemm failed");
    }
    // the prototype side: [Kc|Vc] = cn Wkv^T + bkv on B*K rows
    TRY(cffm_colsum(ws + W.dkv, nk, 512, g->kv_b, stream));
    {
        PROF(ST_GEMM);
        REQUIRE(!gemm_tn(ws + W.dkv, ws + W.cn, g->kv_w, nk, 512, 256, st), "gtc_block_backward: kv weight-gradient gemm failed");
        REQUIRE(!gemm_nn(ws + W.dkv, p->kv_w, ws + W.dcn, nk, 512, 256, st), "gtc_block_backward: kv input-gradient gemm failed");
    }
    // LN1 backward of the tokens (+ the residual path dx1), then of the prototypes: the same norm1 (:622), so the second ACCUMULATES
    TRY(cffm_ln_bwd_residual(x, ws + W.mean1, ws + W.rstd1, p->norm1_w, ws + W.dz, ws + W.dx1, dx, g->norm1_w, g->norm1_b, nt, 1, nullptr, nullptr, stream));
    reductions.finish();
    TRY(cffm_ln_bwd_residual(centers, ws + W.cmean, ws + W.crstd, p->norm1_w, ws + W.dcn, nullptr, dcenters, g->norm1_w, g->norm1_b, nk, 0, nullptr, nullptr,
                             stream));
    CHECK_LAUNCH("gtc_block_backward");
    return 0;
}

// ------------------------------------------------------------------------------------------- block
// bias tiles + pooling matrices of `n` blocks (workspaces ws0 + i*ws_stride) in ceil(n / PREP_MAXD) launches
static int prep_args(PrepArgs& a, const cffm_block_params* params, int d0, int nd, float* ws0, long ws_stride, const cffm_block_ws& L) {
    const int nb = (CFFM_HEADS * CFFM_NQ_PAD * CFFM_NKEY_PAD + 255) / 256;
    for (int d = 0; d < PREP_MAXD; ++d) {
        const cffm_block_params& p = params[d0 + (d < nd ? d : 0)];
        float* ws = ws0 + (long)(d0 + (d < nd ? d : 0)) * ws_stride;
        a.t[d].own = p.rpb_own; a.t[d].ring = p.rpb_ring;
        for (int i = 0; i < 4; ++i) { a.t[d].pool[i] = p.rpb_pool[i]; a.pw[d].w[i] = p.pool_w[i]; }
        a.bias[d] = ws + L.bias; a.M[d] = ws + L.M;
        a.w[d][0] = p.qkv_w; a.w[d][1] = p.proj_w; a.w[d][2] = p.fc1_w; a.w[d][3] = p.fc2_w;
        a.w_s[d] = ws + L.w_split;
        a.w_f[d] = ws + L.w_frag;
    }
    a.nbias = nb;
    // pack: 0 nothing (library GEMMs) | 1 split-4 copy + fragment-ordered copies | 2 fragment-ordered copies only -- every Linear of the
    // block runs as a row-panel kernel then and nobody reads the split-4 copy (3 MB read + 3 MB written per block saved)
    a.pack = (panel_on() && panel_qkv_on()) ? 2 : 1;
    return nb + 1 + (a.pack ? PREP_WBLOCKS + PREP_FBLOCKS : 0);      // workgroups per block
}
static int param_prep(const cffm_block_params* params, int n, float* ws0, long ws_stride, const cffm_block_ws& L, void* stream) {
    PROF(ST_BIAS_ASM);
    for (int d0 = 0; d0 < n; d0 += PREP_MAXD) {
        const int nd = (n - d0 < PREP_MAXD) ? n - d0 : PREP_MAXD;
        PrepArgs a;
        const int pnx = prep_args(a, params, d0, nd, ws0, ws_stride, L);
        CFFM_LAUNCH(k_param_prep, (pnx, nd), (256), 0, (hipStream_t)stream, a);
    }
    CHECK_LAUNCH("param_prep");
    return 0;
}

static int block_forward_impl(const cffm_geom* g, const cffm_block_params* p, const float* x_ref, long ref_bs,
                              const float* x_tgt, long tgt_bs, const int* key_src, const int* q_dst, float* ws,
                              float* scratch, void* stream);
int cffm_block_forward(const cffm_geom* g, const cffm_block_params* p, const float* x_ref, long ref_bs,
                       const float* x_tgt, long tgt_bs, const int* key_src, const int* q_dst, float* ws,
                       float* scratch, void* stream) {
    REQUIRE(g && p && ws && scratch, "block_forward: null");
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    TRY(param_prep(p, 1, ws, 0, L, stream));
    return block_forward_impl(g, p, x_ref, ref_bs, x_tgt, tgt_bs, key_src, q_dst, ws, scratch, stream);
}
// the block after its parameter-only inputs (ws.bias, ws.biasT, ws.M) have been prepared
static struct { bool pending; hipStream_t side; } g_prep_join = {false, nullptr};
static int block_forward_impl(const cffm_geom* g, const cffm_block_params* p, const float* x_ref, long ref_bs,
                              const float* x_tgt, long tgt_bs, const int* key_src, const int* q_dst, float* ws,
                              float* scratch, void* stream) {
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const long NR = (long)g->B * g->RC, NP = (long)g->B * g->HW;
    hipStream_t st = (hipStream_t)stream;
    // Hand-written GEMMs: the tensors only they read (zall, z2, act) are produced in split-4 storage and the weights come
    // from the split-4 copy k_param_prep made (ws.w_split), so no operand tile is split while it is staged -- except ao,
    // which the attention backward also reads.  CFFM_GEMM=lib (rocBLAS cross-check) keeps everything plain fp32.
    constexpr int sp = 1;   // split-4 storage of what only the GEMMs read (zall, z2, act, dh) and of the weights: always (the library-GEMM form that kept plain fp32 left in round 4)
    // (a pending join with the side-stream parameter prep -- layer_forward_impl -- is taken AFTER this launch: the kernel builds its
    //  pooling-matrix rows from the raw weights then, and the chain's first kernel does not wait for another stream)
    const bool late_join = g_prep_join.pending;
    TRY(ln_pool_fwd_impl(g, x_ref, ref_bs, x_tgt, tgt_bs, p->norm1_w, p->norm1_b, ws + L.M, p->pool_b, ws + L.zall,
                         ws + L.mean1, ws + L.rstd1, sp, stream, late_join ? p->pool_w : nullptr));
    if (late_join) {
        g_prep_join.pending = false;
        side_join(g_prep_join.side, st, 0);
    }
    if (sp && panel_qkv_on()) {
        PROF(ST_GEMM); PROF2(ST_G_QKV_FWD);
        REQUIRE(!panel_qkv_fwd(ws + L.zall, ws + L.w_frag, p->qkv_b, (h16*)(ws + L.qkv), NR, st, dw_stream_sw() ? ws + L.zall_t : nullptr), "block_forward: qkv gemm failed");
    }
    TRY(cffm_attn_fwd(g, ws + L.qkv, key_src, q_dst, (const void*)(ws + L.bias), ws + L.ao, ws + L.lse, stream));
    if (sp && panel_on()) {
        // proj + residual + norm2 + Mlp in one row-panel launch (panel_kernels.h); weights in fragment order from k_param_prep
        const float* wf = ws + L.w_frag;
        PROF(ST_GEMM);
        if (dw_stream_sw())     // z2 / act (and a copy of ao) in T-frag storage: operands of the streaming weight gradients
            TRY(cffm_mlp_fwd_tfrag(ws + L.ao, x_tgt, tgt_bs, g->HW, wf + 768 * 256, wf + 768 * 256 + 256 * 256, wf + 768 * 256 + 256 * 256 + 1024 * 256,
                                   p->proj_b, p->fc1_b, p->fc2_b, p->norm2_w, p->norm2_b, ws + L.x1, nullptr, ws + L.mean2, ws + L.rstd2, ws + L.hraw,
                                   nullptr, ws + L.x2, ws + L.ao_t, ws + L.z2, ws + L.act, NP, stream));
        else
        TRY(cffm_mlp_fwd(ws + L.ao, x_tgt, tgt_bs, g->HW, wf + 768 * 256, wf + 768 * 256 + 256 * 256, wf + 768 * 256 + 256 * 256 + 1024 * 256,
                         p->proj_b, p->fc1_b, p->fc2_b, p->norm2_w, p->norm2_b, ws + L.x1, ws + L.z2, ws + L.mean2, ws + L.rstd2, ws + L.hraw,
                         store_act() ? ws + L.act : nullptr /* CFFM_STORE_ACT=0: the fc2 weight gradient re-applies bias + GELU to hraw */,
                         ws + L.x2, NP, stream));
        CHECK_LAUNCH("block_forward");
        return 0;
    }
    (void)scratch;
    return fail(-3, "block_forward: unreachable");
}

static int block_backward_impl(const cffm_geom* g, const cffm_block_params* p, const cffm_block_grads* gr,
                               const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs, const int* key_src,
                               const int* q_dst, const int* inv_ptr, const int* inv_idx, const float* ws, const float* dout,
                               float* dx_tgt, long dtgt_bs, float* scratch, int defer, int par, int slot, int nslots, void* stream);
// The parameter-gradient tail of a block backward (record reductions, pooling-matrix backward), LAUNCHED LATE: its fork point is the end
// of ln_pool_bwd (event fork[3]), but the launches happen only after the chain's next kernel has been launched (tail_flush), so that
// under stream capture the chain's kernel is ln_pool_bwd's FIRST dependant and keeps its place on the graph's first stream (the graph
// executor hands every further dependant of a node the next stream, see side_fork_mark).
#ifndef CFFM_EMU
static struct { bool has, cs; RedJobs jobs; int parity; } g_tail = {false, false, {}, 0};
// on_main (end of a layer backward): the optimizer is what waits for these gradients, so they are the chain now -- launched on the
// caller's stream itself, in front of everything else that follows the last ln_pool_bwd
// `on` (fork_order bit 6): a side stream that is already ordered behind the ln_pool_bwd in question (the NEXT block's weight-gradient
// stream, forked behind that block's gather): no fork of its own, so ln_pool_bwd keeps a single dependant
static int tail_flush(hipStream_t st, bool on_main = false, hipStream_t on = nullptr) {
    if (!g_tail.has) return 0;
    g_tail.has = false;
    hipStream_t s3 = on_main ? st : (on ? on : side_fork_take(st, 3));
    if (on && !on_main) {
        g_tail.cs = false;
        redq_launch(g_tail.jobs, s3);
        CHECK_LAUNCH("block_backward reductions");
        side_record(s3, st, g_side.tail_done[g_tail.parity]);
        g_side.tail_pending[g_tail.parity] = true;
        return 0;
    }
    // the q|k|v bias records come from the column sum on the weight-gradient stream.  On a side stream the tail waits for that stream's
    // whole block (tail_order: the graph executor then queues it right behind the block's side work; waiting for the column sum alone
    // it was queued behind the NEXT block's side work); on the caller's stream only for the column sum (cs_order)
    if (!on_main) (void)hipStreamWaitEvent(s3, g_side.tail_order, 0);
    else if (g_tail.cs) (void)hipStreamWaitEvent(s3, g_side.cs_order, 0);
    g_tail.cs = false;
    redq_launch(g_tail.jobs, s3);
    CHECK_LAUNCH("block_backward reductions");
    if (s3 != st) {
        side_record(s3, st, g_side.tail_done[g_tail.parity]);
        g_side.tail_pending[g_tail.parity] = true;
    }
    return 0;
}
#else
static int tail_flush(hipStream_t, bool = false, hipStream_t = nullptr) { return 0; }
#endif
static bool tail_on_main() {     // CFFM_TAIL_MAIN=0: the last block's tail on the side stream like every other block's (A/B)
    static int v = -1;
    if (v < 0) { const char* e = cffm_tune("CFFM_TAIL_MAIN"); v = (e && e[0] == '0') ? 0 : 1; }
    return v != 0;
}
static void tail_reset() {      // entry of a layer backward: nothing of a previous (failed) call is left to launch or to wait for
#ifndef CFFM_EMU
    g_tail.has = g_tail.cs = false;
    for (int i = 0; i < 2; ++i) g_side.dw_pending[i] = g_side.bias_pending[i] = g_side.tail_pending[i] = false;   // (a call that returned normally ended in side_join_all)
#endif
}
int cffm_block_backward(const cffm_geom* g, const cffm_block_params* p, const cffm_block_grads* gr,
                        const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs, const int* key_src,
                        const int* q_dst, const int* inv_ptr, const int* inv_idx, const float* ws, const float* dout,
                        float* dx_ref, long dref_bs, int accum_ref, float* dx_tgt, long dtgt_bs, float* scratch, void* stream) {
    REQUIRE(x_ref && dx_ref, "block_backward: null");
    TRY(block_backward_impl(g, p, gr, x_ref, ref_bs, x_tgt, tgt_bs, key_src, q_dst, inv_ptr, inv_idx, ws, dout, dx_tgt, dtgt_bs, scratch, 0, 0, 0, 1,
                            stream));
    return cffa_finish(g, 1, p, gr, 0, 0, ws, 0, x_ref, ref_bs, dx_ref, dref_bs, accum_ref, stream);
}
// `defer` (layer backward, round 3): the call returns with this block's parameter-gradient tail (partial-slab sums, record reductions,
// pooling-matrix backward, bias-table scatter) still running on the side streams; the caller's stream has only waited for what reads
// the scratch operands the NEXT block overwrites (the weight-gradient GEMMs).  The next block_backward_impl waits for the rest where
// it needs it; the caller ends the sequence with side_join_all().  (Measured on the round-3 timeline: the chain idled ~32 us per
// block between ln_pool_bwd and the next block's first kernel, behind four small side kernels.)
static int block_backward_impl(const cffm_geom* g, const cffm_block_params* p, const cffm_block_grads* gr,
                               const float* x_ref, long ref_bs, const float* x_tgt, long tgt_bs, const int* key_src,
                               const int* q_dst, const int* inv_ptr, const int* inv_idx, const float* ws, const float* dout,
                               float* dx_tgt, long dtgt_bs, float* scratch, int defer, int par, int slot, int nslots, void* stream) {
    REQUIRE(g && p && gr && ws && dout && scratch, "block_backward: null");
    {
        static int defer_env = -1;   // CFFM_DEFER_JOIN=0: every block ends fully joined (round-2 behaviour; A/B measurements)
        if (defer_env < 0) { const char* e = cffm_tune("CFFM_DEFER_JOIN"); defer_env = (e && e[0] == '0') ? 0 : 1; }
        defer = defer && defer_env;
    }
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const Scratch S = scratch_layout(g);
    const long NR = (long)g->B * g->RC, NP = (long)g->B * g->HW;
    par = (defer && par) ? 1 : 0;
    const long altb = S.alt, altact = altb + up(NP * CFFM_C), altqkv = altact + up((NP + 31) / 32 * 32 * CFFM_HID), altbias = altqkv + up(NR * 768);
    float* dx1 = scratch + (par ? altb : S.b);
    float* dz2 = scratch + S.dz2;
    float* dao = scratch + S.dao;
    float* dact = scratch + (par ? altact : S.dact);
    float* dqkv = scratch + (par ? altqkv : S.dqkv);
    // T-frag copies of dout | dx1 | dqkv (streaming weight gradients), one set per parity
    const long NP32 = (NP + 31) / 32 * 32;
    float* dout_t = scratch + S.tf + (par ? S.tf_stride : 0);
    float* dx1_t = dout_t + up(NP32 * CFFM_C);
    float* dqkv_t = dx1_t + up(NP32 * CFFM_C);
    const int stream_dw = dw_stream_sw();
    CffaSlot cffa;           // library-owned: this block's dzall lives until the reference pass at the end of the range (cffa_finish)
    TRY(cffa_slot(g, slot, nslots, &cffa));
    float* dzall = cffa.dzall;
    float* dbiasT = scratch + (par ? altbias : S.dbiasT);
    RedScope reductions((hipStream_t)stream);   // the four parameter-gradient reductions below run as one launch (finish())
    hipStream_t st = (hipStream_t)stream;
    hipStream_t sa = st;
    side_init(st);
#ifndef CFFM_EMU
    if (g_side.tail_pending[g_red_parity]) {    // the reduction that last read this scope's record buffer (two blocks ago)
        side_wait(st, g_side.tail_done[g_red_parity]);
        g_side.tail_pending[g_red_parity] = false;
    }
    g_side.par = par;
    if (g_side.dw_pending[par]) {               // the weight-gradient group that last read this scratch set (two blocks ago)
        side_wait(st, g_side.dw_done[par]);
        g_side.dw_pending[par] = false;
    }
#endif
    constexpr int sp = 1;   // see block_forward_impl: zall / z2 / act / weights (and dh below) in split-4 storage
    const int one_group = sp ? dw_one_group(st) : 0;
    if (sp && !one_group) {
        // the grouped form (chosen under stream capture) needs larger partial slabs than the split form: size the side scratch for it
        // now, while allocating is still allowed -- a capture that follows eager steps must not grow it
        const GemmTN all4[4] = {{nullptr, nullptr, nullptr, NR, 768, CFFM_C}, {nullptr, nullptr, nullptr, NP, CFFM_HID, CFFM_C},
                                {nullptr, nullptr, nullptr, NP, CFFM_C, CFFM_HID}, {nullptr, nullptr, nullptr, NP, CFFM_C, CFFM_C}};
        size_t need = gemm_tn_group_partial_floats(all4, 4, 480);
        DwsPlan P4;
        if (stream_dw && dw_stream_plan(all4, 4, dw_stream_target(), &P4)) need = P4.part_floats;
        REQUIRE(!need || lib_scratch2(need), "block_backward: scratch allocation failed");
    }
    // a group of the block's weight gradients: the streaming kernel on T-frag operands (default), or the LDS-staged group
    auto dw_group = [&](const cffm_wgrad* wg, const GemmTNPre* pre, int n, hipStream_t s, int target, void (*after)(hipStream_t)) -> int {
        if (!stream_dw) return gemm_tn_group((const GemmTN*)wg, n, s, pre, s == st ? lib_scratch : lib_scratch2, target, after);
        DwsPlan P;
        if (!dw_stream_plan((const GemmTN*)wg, n, dw_stream_target(), &P)) return -1;
        float* part = P.part_floats ? (s == st ? lib_scratch : lib_scratch2)(P.part_floats) : nullptr;
        if (P.part_floats && !part) return -1;
        return dw_group_stream((const GemmTN*)wg, n, s, part, dw_stream_target(), after);
    };
    // the four problems' operands: T-frag copies (streaming) or the split-4 / fp32 tensors themselves
    const float* fc2_x = stream_dw ? ws + L.act : ((sp && panel_on() && !store_act()) ? ws + L.hraw : ws + L.act);
    const cffm_wgrad wg_qkv = {stream_dw ? dqkv_t : dqkv, stream_dw ? ws + L.zall_t : ws + L.zall, gr->qkv_w, NR, 768, CFFM_C};
    const cffm_wgrad wg_fc1 = {dact, ws + L.z2, gr->fc1_w, NP, CFFM_HID, CFFM_C};
    const cffm_wgrad wg_fc2 = {stream_dw ? dout_t : dout, fc2_x, gr->fc2_w, NP, CFFM_C, CFFM_HID};
    const cffm_wgrad wg_proj = {stream_dw ? dx1_t : dx1, stream_dw ? ws + L.ao_t : ws + L.ao, gr->proj_w, NP, CFFM_C, CFFM_C};
    // x2 = x1 + act W2^T + b2
    // act = gelu(hraw + b1); hraw = z2 W1^T: the GELU backward runs in the epilogue of the fc2 input-gradient GEMM (dact is
    // never materialised; what is stored is dh, in split-4 storage since only GEMMs read it, plus column-sum records of it)
    const int panel = sp && panel_on();
    if (panel) {
        // the input-gradient chain fc2 -> GELU' -> fc1 -> norm2 -> proj in one row-panel launch; dh (split-4), dx1 and the bias / norm
        // gradient records come out of it for the weight-gradient groups and the reductions below
        const float* wfn = ws + L.w_frag + PREP_WFLOATS;
        PROF(ST_GEMM);
        if (stream_dw)
            TRY(cffm_mlp_bwd_tfrag(dout, ws + L.hraw, p->fc1_b, ws + L.x1, ws + L.mean2, ws + L.rstd2, p->norm2_w, wfn + 768 * 256 + 256 * 256 + 1024 * 256,
                                   wfn + 768 * 256 + 256 * 256, wfn + 768 * 256, nullptr, dx1, dao, gr->norm2_w, gr->norm2_b, gr->fc1_b, gr->fc2_b, gr->proj_b,
                                   dout_t, dact, dx1_t, NP, stream));
        else
        TRY(cffm_mlp_bwd(dout, ws + L.hraw, p->fc1_b, ws + L.x1, ws + L.mean2, ws + L.rstd2, p->norm2_w, wfn + 768 * 256 + 256 * 256 + 1024 * 256,
                         wfn + 768 * 256 + 256 * 256, wfn + 768 * 256, dact, dx1, dao, gr->norm2_w, gr->norm2_b, gr->fc1_b, gr->fc2_b, gr->proj_b, NP,
                         stream));
        if (!(fork_order() & 64)) TRY(tail_flush(st));    // the previous block's parameter-gradient tail, now that this block's first kernel is ln_pool_bwd's first dependant
        if (!one_group) {
            sa = side_fork(st, 0);
            void* stream_a = (void*)sa;
            const cffm_wgrad wga[2] = {wg_fc1, wg_fc2};
            const GemmTNPre prea[2] = {{1, 1, nullptr}, {0, (panel && !store_act()) ? 2 : 1, p->fc1_b}};
            {
                void* stream = stream_a;
                PROF2(ST_G_DW);
                REQUIRE(!dw_group(wga, prea, 2, sa, 320, nullptr), "block_backward: weight-gradient gemm failed");
            }
            side_mark(sa, st, 0);
        }
    }
    // attention: the fused kernel and the dK/dV gather stay on the chain; the bias-gradient tile sum and its scatter into the six
    // tables go to the side stream (branch 1), the q|k|v bias column sum and the weight gradients of q|k|v / proj after the gather
    // (branch 2)
    if (!(fork_order() & 64)) TRY(tail_flush(st));        // (forms of the block whose first kernel is not the fused Mlp backward)
    hipStream_t sb = st, s1 = st;
    int bias_late = 0, late_ng = 0;
    float* late_dbp = nullptr;
    {
        PROF(ST_ATTN_BWD);
        float* dbp;
        int ng;
#ifndef CFFM_EMU
        if (g_side.bias_pending[par]) {   // the bias-gradient tile sum of two blocks ago still owns this set's tile buffer / dbiasT
            side_wait(st, g_side.bias_done[par]);
            g_side.bias_pending[par] = false;
        }
#endif
        TRY(attn_bwd_fused(g, ws + L.qkv, key_src, q_dst, (const h16*)(ws + L.bias), ws + L.ao, dao, ws + L.lse, dqkv, scratch + S.dkvp, &dbp, &ng, par, stream));
        // (side work first here, although that makes the chain change hardware queues under graph replay -- see side_fork_mark: with the
        //  chain launched first the executor parked these side kernels behind the NEXT block's chain and the step's tail grew:
        //  0.864 vs 0.849 ms per step, means of three alternating runs)
        if (sp && (fork_order() & 4)) {
            // no branch of its own: the bias-gradient tile sum and scatter follow the weight-gradient group on ITS stream (below), so
            // the attention backward has ONE dependant and the chain stays on its hardware queue
            TRY(attn_bwd_gather(g, inv_ptr, inv_idx, scratch + S.dkvp, dqkv, stream));
            bias_late = 1; late_dbp = dbp; late_ng = ng;
        } else if (sp && (fork_order() & 1)) {
            side_fork_mark(st, 1);
            TRY(attn_bwd_gather(g, inv_ptr, inv_idx, scratch + S.dkvp, dqkv, stream));
            s1 = side_fork_take(st, 1);
            TRY(attn_bwd_bias_sum(dbp, ng, dbiasT, (void*)s1));
            TRY(cffm_bias_scatter(dbiasT, gr->rpb_own, gr->rpb_ring, gr->rpb_pool, (void*)s1));
        } else {
        s1 = sp ? side_fork(st, 1) : st;
        TRY(attn_bwd_bias_sum(dbp, ng, dbiasT, (void*)s1));
        TRY(cffm_bias_scatter(dbiasT, gr->rpb_own, gr->rpb_ring, gr->rpb_pool, (void*)s1));
        TRY(attn_bwd_gather(g, inv_ptr, inv_idx, scratch + S.dkvp, dqkv, stream));
        }
    }
    // q|k|v = zall Wqkv^T + b (bias folded into the f16 epilogue; its gradient is the column sum of dqkv)
    int dx_done = 0;
    // the q|k|v bias gradient = column sums of dqkv: as records out of the row-panel input-gradient GEMM, which stages every row of
    // dqkv anyway (CFFM_QKV_COLREC=0: the separate column-sum pass over the 32 MB on the side stream, as before)
    float* qkv_rec = nullptr;
    if (sp && panel_qkv_on() && qkv_colrec_on()) {
        const long nrec = panel_qkv_records(NR);
        qkv_rec = red_scratch((size_t)nrec * 768, st);
        REQUIRE(qkv_rec, "block_backward: scratch allocation failed");
        RedSegs segs;
        segs.nseg = 0;
        seg_add(segs, 0, 768, gr->qkv_b, 0);
        reduce_records(qkv_rec, (int)nrec, 768, 768, segs, st);
    }
    if (sp && ((fork_order() & 2) || stream_dw) && panel_qkv_on()) {     // chain first: the q|k|v input gradient is launched before the side work
        // (streaming weight gradients: the side work forks BEHIND this kernel -- it leaves the T-frag copy of dqkv the q|k|v weight gradient reads)
        if (!stream_dw) side_fork_mark(st, 2);
        PROF(ST_GEMM); PROF2(ST_G_QKV_DX);
        REQUIRE(!panel_qkv_dx(dqkv, ws + L.w_frag + PREP_WFLOATS, dzall, NR, st, qkv_rec, stream_dw ? dqkv_t : nullptr), "block_backward: q|k|v input-gradient gemm failed");
        if (stream_dw) side_fork_mark(st, 2);
        dx_done = 1;
    }
    // the block's side work behind the dK/dV gather (weight gradients, bias-gradient tiles, the previous block's tail)
    auto side_b = [&]() -> int {
    if (sp) {
        sb = dx_done ? side_fork_take(st, 2) : side_fork(st, 2);
        void* stream_b = (void*)sb;
        if (!qkv_rec) {
            TRY(cffm_colsum(dqkv, NR, 768, gr->qkv_b, stream_b));
#ifndef CFFM_EMU
            if (sb != st) { (void)hipEventRecord(g_side.cs_order, sb); g_tail.cs = true; }
#endif
        }
        if (one_group) {
            // ALL four weight gradients as one grouped launch (~480 workgroups with one slice length, one partial-sum launch: 59 us
            // where two groups of two take 2 x 53), on the side stream beside q|k|v's input gradient and the CFFA backward; every
            // operand lives until the end of the block now that ln_pool_bwd no longer writes over `dout`
            const cffm_wgrad wg4[4] = {wg_qkv, wg_fc1, wg_fc2, wg_proj};
            const GemmTNPre pre4[4] = {{0, 1, nullptr}, {1, 1, nullptr}, {0, (panel && !store_act()) ? 2 : 1, p->fc1_b}, {0, 0, nullptr}};
            void* stream = stream_b;
            PROF(ST_GEMM); PROF2(ST_G_DW);
            REQUIRE(!dw_group(wg4, pre4, 4, sb, 480, dw_group_launched), "block_backward: weight-gradient gemm failed");
            sa = sb;
            side_mark(sa, st, 0);
            if (bias_late) {
                PROF(ST_ATTN_BWD);
                s1 = sb;
                TRY(attn_bwd_bias_sum(late_dbp, late_ng, dbiasT, (void*)s1));
                TRY(cffm_bias_scatter(dbiasT, gr->rpb_own, gr->rpb_ring, gr->rpb_pool, (void*)s1));
                bias_late = 0;
            }
            if ((fork_order() & 64) && sb != st) TRY(tail_flush(st, false, sb));   // the PREVIOUS block's record reductions: no fork of their own
        } else {
        const cffm_wgrad wgb[2] = {wg_qkv, wg_proj};
        const GemmTNPre preb[2] = {{0, 1, nullptr}, {0, 0, nullptr}};
        {
            void* stream = stream_b;
            PROF(ST_GEMM); PROF2(ST_G_DW);
            REQUIRE(!dw_group(wgb, preb, 2, sb, 320, dw_group_launched), "block_backward: weight-gradient gemm failed");
        }
        if (bias_late) {
            PROF(ST_ATTN_BWD);
            s1 = sb;
            TRY(attn_bwd_bias_sum(late_dbp, late_ng, dbiasT, (void*)s1));
            TRY(cffm_bias_scatter(dbiasT, gr->rpb_own, gr->rpb_ring, gr->rpb_pool, (void*)s1));
            bias_late = 0;
        }
        }
    }
    return 0;
    };
    // Streaming weight gradients under capture: the group is launched AFTER the chain's next kernel (ln_pool_bwd_tgt below), so that kernel
    // is the q|k|v input gradient's first-launched dependant and keeps the chain's hardware queue (see side_fork_mark); nothing the
    // group reads is written by it (fc2's weight gradient reads the T-frag copy of dout, not dout).
    const bool side_late = stream_dw && one_group && dx_done;
    if (!side_late) TRY(side_b());
    if (dx_done) {
    } else if (sp && panel_qkv_on()) {
        PROF(ST_GEMM); PROF2(ST_G_QKV_DX);
        REQUIRE(!panel_qkv_dx(dqkv, ws + L.w_frag + PREP_WFLOATS, dzall, NR, st, qkv_rec), "block_backward: q|k|v input-gradient gemm failed");
    }
    if (!side_late && (!one_group || dx_tgt == dout)) side_join(sa, st, 0);    // fc2's weight gradient has read dout before an in-place ln_pool_bwd overwrites it
#ifndef CFFM_EMU
    if (g_side.dw_pending[par ^ 1] && g_side.dw_dout[par ^ 1] == dx_tgt) {   // (depth >= 3: the block before still reads its dout = our dx_tgt)
        side_wait(st, g_side.dw_done[par ^ 1]);
        g_side.dw_pending[par ^ 1] = false;
    }
#endif
    // CFFA, target frame: dx_tgt for the next block + this block's target records.  The reference frames of every block of the range
    // follow in ONE pass at its end (cffa_finish): they need this block's pooled-cell rows of dzall and nothing else.
    TRY(ln_pool_bwd_tgt(g, x_tgt, tgt_bs, p->norm1_w, p->norm1_b, ws + L.M, ws + L.mean1, ws + L.rstd1, dzall, dx1, dx_tgt, dtgt_bs, cffa.rec,
                        stream));
    if (side_late) TRY(side_b());
    // the record reductions (every block-partial record of this backward except the CFFA's is written by now):
    // parameter gradients only -> side stream (branch 3); the caller's stream then waits for the side stream once
#ifndef CFFM_EMU
    if (defer && sp && sb != st && s1 != st && g_side.on && (fork_order() & 32)) {
        if (g_tail.has) TRY(tail_flush(st));    // (an earlier block's tail nobody has launched yet: before its slot is reused)
        side_fork_mark(st, 3);
        if (s1 != sb) side_order(s1, sb);
        (void)hipEventRecord(g_side.tail_order, sb);
        g_tail.has = true; g_tail.jobs = g_rq.jobs; g_tail.parity = g_red_parity;
        g_rq.jobs.njob = 0; g_rq.active = false;
        g_side.dw_pending[par] = true;
        g_side.dw_dout[par] = dout;
        side_record(s1, st, g_side.bias_done[par]);
        g_side.bias_pending[par] = true;
        return 0;
    }
#endif
    hipStream_t s3 = sp ? side_fork(st, 3) : st;
    if (sp && sb != st && s3 != st) side_order(sb, s3);     // the q|k|v bias records come from branch 2's column sum
    if (sp && s1 != st && s3 != st && s1 != s3) side_order(s1, s3);   // (four side streams: branch 1 joins through branch 3)
    reductions.finish_on(s3);
    CHECK_LAUNCH("block_backward reductions");
#ifndef CFFM_EMU
    if (defer && sp && sb != st && s3 != st) {
        // (no wait for the weight-gradient GEMMs: the next block works in the other scratch set; whoever reuses THIS set, or writes
        //  the `dout` they read, waits for dw_done[par] then)
        g_side.dw_pending[par] = true;
        g_side.dw_dout[par] = dout;
        side_record(s1, st, g_side.bias_done[par]);
        g_side.bias_pending[par] = s1 != st;
        side_record(s3, st, g_side.tail_done[g_red_parity]);
        g_side.tail_pending[g_red_parity] = true;
        return 0;
    }
#endif
    (void)defer;
    side_mark(s3, st, 1);
    side_join(s3, st, 1);    // the caller's stream owns every gradient (and the scratch operands) again
    if (sp && sb != st && sb != s3) { side_mark(sb, st, 0); side_join(sb, st, 0); }   // ... on both side streams
    return 0;
}

// ------------------------------------------------------------------------------------------- SegFormer embedding
static int segf_maps(SegfMaps& mp, const int* h, const int* w, int nmaps, int H, int W, const char* who) {
    if (nmaps < 0 || nmaps > 3 || H < 1 || W < 1) return fail(-1, "%s: bad map count / size", who);
    mp.cnt = nmaps;
    mp.nseg = 0; mp.per_frame = 0;
    for (int m = 0; m < 3; ++m) {
        mp.z[m] = nullptr; mp.dz[m] = nullptr; mp.h[m] = mp.w[m] = 1;
        if (m >= nmaps) continue;
        // the 2x2-patch forward assumes every tap of a patch lies in a 3x3 neighbourhood and the adjoint windows assume an
        // UPsampling map: a map larger than the output (or more than SEGF_MAX_RATIO times smaller) is rejected, not mis-resized
        if (h[m] < 1 || w[m] < 1 || h[m] > H || w[m] > W || (long)H > (long)SEGF_MAX_RATIO * h[m] || (long)W > (long)SEGF_MAX_RATIO * w[m])
            return fail(-1, "%s: map %d is %dx%d for a %dx%d output (only upsampling by factors 1..%d is supported)", who, m, h[m], w[m],
                        H, W, SEGF_MAX_RATIO);
        mp.h[m] = h[m]; mp.w[m] = w[m];
    }
    return 0;
}
// ---- composed embedding weights (ABI 10) --------------------------------------------------------------------------------------------
// A_i = Wf_i W_i for the k <= 4 scales (Wf_i = input-channel block k - 1 - i of linear_fuse.conv.weight [e][k e], read in place through
// its leading dimension; W_i = linear_c{i+1}.proj.weight [e][C_i]) and d = sum_i Wf_i b_i -- what ops.segformer_fuse feeds the per-scale
// embedding GEMMs (cffm_head.py:102-119 without the 1024-channel concat).  The k products are independent 10-MFLOP GEMMs: in the backward
// (2 k of them) each scale on its own branch; round 5 ran them from Python behind two layout copies of the fuse weight, with torch's
// cat / gemv / ger for the constant (97 us of a replayed head step, 120 us for the backward).
static void fuse_bias_ptrs(FuseBias& fb, const float* const* lin_b, float* const* dlin_b, int k) {
    for (int j = 0; j < 4; ++j) { fb.b[j] = j < k ? lin_b[k - 1 - j] : nullptr; fb.db[j] = (j < k && dlin_b) ? dlin_b[k - 1 - j] : nullptr; }
}
int cffm_fuse_compose_fwd(const float* fuse_w, const float* const* lin_w, const float* const* lin_b, const int* C_in, int k, int e,
                          float* const* mats, float* d, void* stream) {
    REQUIRE(fuse_w && lin_w && lin_b && C_in && mats && d && k >= 1 && k <= 4 && e >= 64 && e % 64 == 0, "fuse_compose_fwd: bad arguments");
    for (int i = 0; i < k; ++i) REQUIRE(lin_w[i] && lin_b[i] && mats[i] && C_in[i] >= 4 && C_in[i] % 4 == 0, "fuse_compose_fwd: bad scale %d", i);
    hipStream_t st = (hipStream_t)stream;
    FuseBias fb;
    fuse_bias_ptrs(fb, lin_b, nullptr, k);
    // (one behind the other: these are the first nodes of a replayed step, where starting a second hardware queue costs more -- ~11 us per
    // branch, profiles/r06_head_timeline_branches_first_version.txt -- than a 7 us GEMM)
    for (int i = 0; i < k; ++i) {
        // A_i [e][C_i] = Wf_i [e][e] (lda = k e) x W_i [e][C_i]
        if (gemm_split_launch<false, true>(fuse_w + (long)(k - 1 - i) * e, lin_w[i], mats[i], e, C_in[i], e, k * e, C_in[i], C_in[i], 1, st))
            return fail(-3, "fuse_compose_fwd: gemm failed");
    }
    CFFM_LAUNCH(k_fuse_const, ((e + 3) / 4), (256), 0, st, fuse_w, fb, k, e, d);
    CHECK_LAUNCH("fuse_compose_fwd");
    return 0;
}
// gradients of the nine tensors from dmats[i] [e][C_i] and dd [e]: dfuse_w [e][k e] (every element written), dlin_w[i] [e][C_i], dlin_b[i] [e]
int cffm_fuse_compose_bwd(const float* fuse_w, const float* const* lin_w, const float* const* lin_b, const int* C_in, int k, int e,
                          const float* const* dmats, const float* dd, float* dfuse_w, float* const* dlin_w, float* const* dlin_b, void* stream) {
    REQUIRE(fuse_w && lin_w && lin_b && C_in && dmats && dd && dfuse_w && dlin_w && dlin_b && k >= 1 && k <= 4 && e >= 64 && e % 64 == 0,
            "fuse_compose_bwd: bad arguments");
    for (int i = 0; i < k; ++i)
        REQUIRE(lin_w[i] && lin_b[i] && dmats[i] && dlin_w[i] && dlin_b[i] && C_in[i] >= 4 && C_in[i] % 4 == 0, "fuse_compose_bwd: bad scale %d", i);
    hipStream_t st = (hipStream_t)stream;
    cffm_branch_mark(stream);
    for (int i = 0; i < k; ++i) {          // (the caller's own chain first: see cffm_branch_mark)
        hipStream_t s = i ? (hipStream_t)cffm_branch_take(stream, i) : st;
        const float* wf = fuse_w + (long)(k - 1 - i) * e;
        // dWf_i [e][e] (ldc = k e) = dA_i [e][C_i] x W_i^T;  dW_i [e][C_i] = Wf_i^T x dA_i (one slice: no library scratch on a branch)
        if (gemm_split_launch<false, false>(dmats[i], lin_w[i], dfuse_w + (long)(k - 1 - i) * e, e, e, C_in[i], C_in[i], C_in[i], k * e, 1, s) ||
            gemm_split_launch<true, true>(wf, dmats[i], dlin_w[i], e, C_in[i], e, k * e, C_in[i], C_in[i], 1, s))
            return fail(-3, "fuse_compose_bwd: gemm failed");
    }
    cffm_branch_join(stream);
    FuseBias fb;
    fuse_bias_ptrs(fb, lin_b, dlin_b, k);
    CFFM_LAUNCH(k_fuse_const_bwd, (k * e / 64), (256), 0, st, fuse_w, dd, fb, k, e, dfuse_w);
    CHECK_LAUNCH("fuse_compose_bwd");
    return 0;
}
int cffm_segfuse_fwd(float* y, const float* d, const float* const z[3], const int h[3], const int w[3], int nmaps, int N, int H,
                     int W, void* stream) {
    REQUIRE(y && d && N >= 0 && (nmaps == 0 || (z && h && w)), "segfuse_fwd: bad arguments");
    SegfMaps mp;
    TRY(segf_maps(mp, h, w, nmaps, H, W, "segfuse_fwd"));
    for (int m = 0; m < nmaps; ++m) { REQUIRE(z[m], "segfuse_fwd: null map"); mp.z[m] = z[m]; }
    const long rows = (long)N * H * W;
    if (!rows) return 0;
    REQUIRE(rows < (1L << 31), "segfuse_fwd: too many output pixels");
    hipStream_t st = (hipStream_t)stream;
    const long patches = (long)N * ((H + 1) / 2) * ((W + 1) / 2);
    CFFM_LAUNCH(k_segfuse_fwd, ((unsigned)((patches + 3) / 4)), (256), 0, st, y, d, mp, N, H, W);
    CHECK_LAUNCH("segfuse_fwd");
    return 0;
}
int cffm_segfuse_bwd(const float* g, float* const dz[3], const int h[3], const int w[3], int nmaps, int N, int H, int W,
                     void* stream) {
    REQUIRE(g && N >= 0 && nmaps >= 1 && dz && h && w, "segfuse_bwd: bad arguments");
    SegfMaps mp;
    TRY(segf_maps(mp, h, w, nmaps, H, W, "segfuse_bwd"));
    for (int m = 0; m < nmaps; ++m) { REQUIRE(dz[m], "segfuse_bwd: null map"); mp.dz[m] = dz[m]; }
    // bands of BR output rows (BR = the largest resize factor, at most SEGF_MAX_BANDS bands); a low-resolution row belongs to
    // the band its centre falls into
    int BR = 1;
    for (int m = 0; m < nmaps; ++m) BR = std::max(BR, (H + h[m] - 1) / h[m]);
    BR = std::max(BR, (H + SEGF_MAX_BANDS - 1) / SEGF_MAX_BANDS);
    const int nb = (H + BR - 1) / BR;
    long items = 0;
    int next_q[3] = {0, 0, 0};
    for (int b = 0; b < nb; ++b)
        for (int m = 0; m < nmaps; ++m) {
            const int q0 = next_q[m];
            int q = q0;
            while (q < h[m] && (b == nb - 1 || (int)(((double)q + 0.5) * H / h[m]) / BR <= b)) ++q;
            next_q[m] = q;
            items += (long)(q - q0) * ((w[m] + SEGF_XQ - 1) / SEGF_XQ);
            mp.seg_q0[b * nmaps + m] = q0;
            mp.seg_end[b * nmaps + m] = (int)items;
        }
    mp.nseg = nb * nmaps;
    mp.per_frame = (int)items;
    const long blocks = items * N;
    REQUIRE(blocks < (1L << 31), "segfuse_bwd: too many work items");
    if (!blocks) return 0;
    hipStream_t st = (hipStream_t)stream;
    CFFM_LAUNCH(k_segfuse_bwd, ((unsigned)blocks), (256), 0, st, g, mp, N, H, W);
    CHECK_LAUNCH("segfuse_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------- resize + cross entropy
static int upce_geom(UpceGeom& G, int M, int K, int h, int w, int H, int W, int ignore, const char* who) {
    if (M < 0 || K < 1 || K > 256 || h < 1 || w < 1 || H < h || W < w || (long)H > (long)UPCE_MAX_RATIO * h ||
        (long)W > (long)UPCE_MAX_RATIO * w)
        return fail(-1, "%s: unsupported sizes (K=%d, %dx%d -> %dx%d; 1 <= K <= 256, resize factor 1..%d)", who, K, h, w, H, W,
                    UPCE_MAX_RATIO);
    G.M = M; G.K = K; G.h = h; G.w = w; G.H = H; G.W = W; G.ignore = ignore; G.rn = G.cn = 0; G.foot = G.win = 0;
    G.inner = 1; G.ms_outer = (long)K * h * w; G.ms_inner = 0; G.ks = h * w; G.ps = 1;       // plain [M,K,h,w]
    G.label_idx = nullptr; G.map_scale = nullptr;
    return 0;
}
// the caller's layout of the logits (include/cffm_hip.h: cffm_upce_maps_*); every element offset must fit 31 bits past its map base
static int upce_layout(UpceGeom& G, int inner, long ms_outer, long ms_inner, int ks, int ps, const char* who) {
    REQUIRE(inner >= 1 && ks >= 1 && ps >= 1 && ms_outer >= 0 && ms_inner >= 0, "%s: bad logits layout", who);
    REQUIRE((ks == 1) != (ps == 1) || (G.K == 1 || G.h * G.w == 1), "%s: either classes (ks = 1, token rows) or cells (ps = 1, [K][h][w]) must be contiguous", who);
    REQUIRE((long)(G.K - 1) * ks + (long)(G.h * G.w - 1) * ps < (1L << 31), "%s: a logits map spans more than 2^31 elements", who);
    G.inner = inner; G.ms_outer = ms_outer; G.ms_inner = ms_inner; G.ks = ks; G.ps = ps;
    return 0;
}
// host restatement of segf_taps (the LDS tile of the forward pass has to cover first tap .. last tap of a 16-pixel span)
static void host_taps(int dst, int in, int out, int& i0, int& i1) {
    const float scale = (float)in / (float)out;
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    i0 = std::min((int)s, in - 1);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
}
static int span_taps(int in, int out) {
    int need = 1;
    for (int t = 0; t < out; t += UPCE_TILE) {
        int a0, a1, b0, b1;
        host_taps(t, in, out, a0, a1);
        host_taps(std::min(t + UPCE_TILE, out) - 1, in, out, b0, b1);
        need = std::max(need, b1 - a0 + 1);
    }
    return need;
}
static int upce_lds(const void* kernel, size_t bytes, const char* who) {
    if (bytes > 160 * 1024) return fail(-1, "%s: the low-resolution tile needs %zu bytes of LDS", who, bytes);
#ifndef CFFM_EMU
    if (bytes > 65536 && hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
        return fail(-2, "%s: cannot reserve %zu bytes of LDS", who, bytes);
#endif
    return 0;
}
long cffm_upce_blocks(int M, int H, int W) {
    return (long)M * ((H + UPCE_TILE - 1) / UPCE_TILE) * ((W + UPCE_TILE - 1) / UPCE_TILE);
}
int cffm_upce_fwd(const float* logits, const long long* labels, float* lse, float* part, int M, int K, int h, int w, int H, int W,
                  int ignore_index, void* stream) {
    return cffm_upce_maps_fwd(logits, labels, nullptr, lse, part, M, K, h, w, H, W, ignore_index, 1, (long)K * h * w, 0, h * w, 1, stream);
}
int cffm_upce_maps_fwd(const float* logits, const long long* labels, const int* label_idx, float* lse, float* part, int M, int K, int h,
                       int w, int H, int W, int ignore_index, int inner, long ms_outer, long ms_inner, int ks, int ps, void* stream) {
    UpceGeom G;
    TRY(upce_geom(G, M, K, h, w, H, W, ignore_index, "upce_fwd"));
    if (!M) return 0;
    REQUIRE(logits && labels && lse && part, "upce_fwd: null");
    TRY(upce_layout(G, inner, ms_outer, ms_inner, ks, ps, "upce_fwd"));
    G.label_idx = label_idx;
    G.rn = span_taps(h, H);
    G.cn = span_taps(w, W);
    const int cells = G.rn * G.cn, kper = ps != 1 ? UPCE_ROWS_KPER : (cells >= 256 ? 1 : 256 / cells);
    const size_t lds = ((size_t)cells * (UPCE_KP(K) + 1) + (size_t)cells * kper + 8) * sizeof(float);
    TRY(upce_lds((const void*)k_upce_fwd, lds, "upce_fwd"));
    hipStream_t st = (hipStream_t)stream;
    REQUIRE(cffm_upce_blocks(M, H, W) < (1L << 31), "upce_fwd: too many tiles");
    CFFM_LAUNCH(k_upce_fwd, ((unsigned)cffm_upce_blocks(M, H, W)), (256), lds, st, logits, labels, lse, part, G);
    CHECK_LAUNCH("upce_fwd");
    return 0;
}
// out[0] = sum_m wl[m] * (sum of map m's loss records), out[1] = sum_m wh[m] * (sum of its hit records); part as cffm_upce_maps_fwd left it
// (cffm_upce_blocks(M, H, W) / M records per map, map-major); wl / wh: M doubles each on the device
int cffm_upce_maps_finalize(const float* part, int M, long per, const double* wl, const double* wh, float* out, void* stream) {
    REQUIRE(part && wl && wh && out && M >= 1 && per >= 1 && per < (1L << 30), "upce_maps_finalize: bad arguments");
    CFFM_LAUNCH(k_upce_finalize, (1), (256), 0, (hipStream_t)stream, part, M, (int)per, wl, wh, out);
    CHECK_LAUNCH("upce_maps_finalize");
    return 0;
}
int cffm_upce_bwd(const float* logits, const long long* labels, const float* lse, const float* gscale, float scale,
                  float* dlogits, int M, int K, int h, int w, int H, int W, int ignore_index, void* stream) {
    return cffm_upce_maps_bwd(logits, labels, nullptr, lse, gscale, nullptr, scale, dlogits, M, K, h, w, H, W, ignore_index, 1,
                              (long)K * h * w, 0, h * w, 1, stream);
}
int cffm_upce_maps_bwd(const float* logits, const long long* labels, const int* label_idx, const float* lse, const float* gscale,
                       const float* map_scale, float scale, float* dlogits, int M, int K, int h, int w, int H, int W, int ignore_index,
                       int inner, long ms_outer, long ms_inner, int ks, int ps, void* stream) {
    UpceGeom G;
    TRY(upce_geom(G, M, K, h, w, H, W, ignore_index, "upce_bwd"));
    if (!M) return 0;
    REQUIRE(logits && labels && lse && dlogits, "upce_bwd: null");
    TRY(upce_layout(G, inner, ms_outer, ms_inner, ks, ps, "upce_bwd"));
    G.label_idx = label_idx;
    G.map_scale = map_scale;
    const bool plain = inner == 1 && ms_outer == (long)K * h * w && ks == h * w && ps == 1;
    G.rn = std::min(h, UPCE_QT + 2);
    G.cn = std::min(w, UPCE_QT + 2);
    // LDS capacities for this resize factor, from the kernel's own formulas (the same float expressions): the largest output
    // footprint of a 4 x 4 tile along each dimension and the largest candidate window of one pixel
    auto span = [](int q0, int q1, int in, int out) {     // candidate range of low-resolution indices q0..q1, clamped
        const float is = (float)out / (float)in;
        int lo = (int)floorf(((float)q0 - 0.5f) * is - 0.5f) - 1, hi = (int)ceilf(((float)q1 + 1.5f) * is - 0.5f) + 1;
        lo = std::max(lo, 0); hi = std::min(hi, out - 1);
        return hi - lo + 1;
    };
    int fy = 1, fx = 1, win = 1;
    for (int q0 = 0; q0 < h; q0 += UPCE_QT) fy = std::max(fy, span(q0, std::min(q0 + UPCE_QT - 1, h - 1), h, H));
    for (int q0 = 0; q0 < w; q0 += UPCE_QT) fx = std::max(fx, span(q0, std::min(q0 + UPCE_QT - 1, w - 1), w, W));
    for (int q = 0; q < h; ++q) win = std::max(win, span(q, q, h, H));
    for (int q = 0; q < w; ++q) win = std::max(win, span(q, q, w, W));
    G.foot = (fy * fx + 1) & ~1;      // even: the tap tables behind the two footprint arrays stay 16-byte aligned
    G.win = win;
    hipStream_t st = (hipStream_t)stream;
    REQUIRE((long)M * ((w + UPCE_QT - 1) / UPCE_QT) * ((h + UPCE_QT - 1) / UPCE_QT) < (1L << 31), "upce_bwd: too many tiles");
    const unsigned grid = (unsigned)((long)M * ((w + UPCE_QT - 1) / UPCE_QT) * ((h + UPCE_QT - 1) / UPCE_QT));
    // two forms (CFFM_UPCE_BWD = block | gather): the block form evaluates every output pixel's soft-max about 1.2 x per launch from
    // register-resident cells (default); the gather form re-evaluates it in each low-resolution pixel it taps (round 1; also the
    // fallback for down-sampling geometries, where blocks are not contiguous runs)
    static int form = -1, ty_env = 0, slots = 0;
    if (form < 0) {
        const char* e = cffm_tune("CFFM_UPCE_BWD");
        const char* t = cffm_tune("CFFM_UPCE_TY");
        ty_env = t ? atoi(t) : 0;
        slots = 1024;                                   // resident workgroups of the block kernel on the whole device
#ifndef CFFM_EMU
        int per_cu = 0, dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k_upce_bwd_blk, UPCE_BLK_THREADS, 16 * 1024) == hipSuccess &&
            per_cu > 0 && cus > 0)
            slots = per_cu * cus;
#endif
        form = (e && e[0] == 'g') ? 1 : 0;
    }
    if ((form == 0 || !plain) && H >= h && W >= w) {
        // owned cell rows per workgroup: the ring row costs (ty + 1) / ty, a partly filled last wave of workgroups costs its idle slots
        const int TXo = UPCE_BLK_COLS - 1, gxb = (w + TXo - 1) / TXo, cpw = UPCE_BLK_THREADS / UPCE_BLK_COLS, nch = (UPCE_KP(K) / 4 + cpw - 1) / cpw;
        int ty = ty_env;
        if (ty <= 0) {
            double best = -1.0;
            for (int t = 2; t <= 16 && t <= std::max(h, 2); ++t) {
                const long nwg = (long)M * gxb * ((h + t - 1) / t) * nch;
                const long waves = (nwg + slots - 1) / slots;
                const double eff = (double)t / (t + 1) * (double)nwg / (double)(waves * slots);
                if (eff > best) { best = eff; ty = t; }
            }
        }
        UpceBlkGeom Bk;
        Bk.ty = ty;
        Bk.xcap = (int)ceil((double)UPCE_BLK_COLS * W / w) + 4;
        Bk.ycap = (int)ceil((double)(ty + 1) * H / h) + 4;
        Bk.fcap = Bk.xcap * Bk.ycap;
        const size_t lds_blk = ((size_t)2 * (Bk.fcap + Bk.xcap + Bk.ycap) + (ty + 3) + (UPCE_BLK_COLS + 3)) * sizeof(float);
        const long nwg = (long)M * gxb * ((h + ty - 1) / ty) * nch;
        REQUIRE(nwg < (1L << 31), "upce_bwd: too many tiles");
        TRY(upce_lds((const void*)k_upce_bwd_blk, lds_blk, "upce_bwd"));
        CFFM_LAUNCH(k_upce_bwd_blk, ((unsigned)nwg), (UPCE_BLK_THREADS), lds_blk, st, logits, labels, lse, gscale, scale, dlogits, G, Bk);
    } else {
        REQUIRE(plain && (long)M * K * h * w < (1L << 31), "upce_bwd: the gather form reads plain [M,K,h,w] logits below 2^31 elements");
        const size_t lds = ((size_t)G.rn * G.cn * UPCE_KP(K) + 2 * (size_t)G.foot + 2 * 16 * G.win * 4) * sizeof(float);
        TRY(upce_lds((const void*)k_upce_bwd, lds, "upce_bwd"));
        CFFM_LAUNCH(k_upce_bwd, (grid), (256), lds, st, logits, labels, lse, gscale, scale, dlogits, G);
    }
    CHECK_LAUNCH("upce_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------- BatchNorm + ReLU + 1/8 stack (head)
static unsigned hf_stat_grid(long units) {
    long b = (units + 63) / 64;
    return (unsigned)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
long cffm_colstats_records(long rows) { return (long)hf_stat_grid((rows + 3) / 4); }
int cffm_colstats(const float* y, long rows, float* part, void* stream) {
    REQUIRE(rows >= 0 && (rows == 0 || (y && part)), "colstats: null");
    if (!rows) return 0;
    CFFM_LAUNCH(k_colstats_partial, (hf_stat_grid((rows + 3) / 4)), (256), 0, (hipStream_t)stream, y, rows, part);
    CHECK_LAUNCH("colstats");
    return 0;
}
long cffm_bn_relu_pool_records(int N, int H, int W) { return ((long)N * (H / 2) * (W / 2) + HF_BLOCKS_PER_WG - 1) / HF_BLOCKS_PER_WG; }
int cffm_bn_relu_pool_fwd(const float* y, const float* scale, const float* shift, const float* mask, float* fused, float* stack, int N,
                          int H, int W, void* stream) {
    REQUIRE(N >= 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "bn_relu_pool_fwd: even map sides expected, got %dx%d", H, W);
    if (!N) return 0;
    REQUIRE(y && scale && shift && fused, "bn_relu_pool_fwd: null");
    CFFM_LAUNCH(k_bn_relu_pool_fwd, ((unsigned)cffm_bn_relu_pool_records(N, H, W)), (256), 0, (hipStream_t)stream, y, scale, shift, mask, fused,
                stack, N, H, W);
    CHECK_LAUNCH("bn_relu_pool_fwd");
    return 0;
}
int cffm_bn_relu_pool_bwd1(const float* y, const float* scale, const float* shift, const float* xs, const float* xo, const float* mask,
                           const float* dfused, const float* dstack, float* g, float* part, int N, int H, int W, void* stream) {
    REQUIRE(N >= 0 && H >= 2 && W >= 2 && H % 2 == 0 && W % 2 == 0, "bn_relu_pool_bwd1: even map sides expected, got %dx%d", H, W);
    if (!N) return 0;
    REQUIRE(y && scale && shift && xs && xo && g && part, "bn_relu_pool_bwd1: null");
    CFFM_LAUNCH(k_bn_relu_pool_bwd1, ((unsigned)cffm_bn_relu_pool_records(N, H, W)), (256), 0, (hipStream_t)stream, y, scale, shift, xs, xo,
                mask, dfused, dstack, g, part, N, H, W);
    CHECK_LAUNCH("bn_relu_pool_bwd1");
    return 0;
}
int cffm_bn_bwd2(float* g, const float* y, const float* xs, const float* xo, const float* c1, const float* mg, const float* mgx, long rows,
                 void* stream) {
    if (!rows) return 0;
    REQUIRE(g && y && xs && xo && c1 && mg && mgx, "bn_bwd2: null");
    CFFM_LAUNCH(k_bn_bwd2, (hf_stat_grid((rows + 3) / 4) * 4), (256), 0, (hipStream_t)stream, g, y, xs, xo, c1, mg, mgx, rows);
    CHECK_LAUNCH("bn_bwd2");
    return 0;
}

int cffm_bn_finalize_fwd(const float* part, long nrec, double count, const float* weight, const float* bias, float* running_mean,
                         float* running_var, float momentum, float eps, float* coef, void* stream) {
    REQUIRE(weight && bias && coef, "bn_finalize_fwd: null");
    REQUIRE(part ? (nrec >= 1 && count >= 1.0) : (running_mean && running_var), "bn_finalize_fwd: records (training) or running statistics (eval) needed");
    REQUIRE(!running_mean == !running_var, "bn_finalize_fwd: running_mean and running_var come together");
    CFFM_LAUNCH(k_bn_finalize_fwd, (CFFM_C / HF_FIN_CH), (1024), 0, (hipStream_t)stream, part, nrec, count, weight, bias, running_mean, running_var, momentum,
                eps, coef);
    CHECK_LAUNCH("bn_finalize_fwd");
    return 0;
}
int cffm_bn_finalize_bwd(const float* part, long nrec, double count, const float* weight, const float* xs, int training, float* out,
                         void* stream) {
    REQUIRE(part && weight && xs && out && nrec >= 1 && count >= 1.0, "bn_finalize_bwd: null / empty");
    CFFM_LAUNCH(k_bn_finalize_bwd, (CFFM_C / HF_FIN_CH), (1024), 0, (hipStream_t)stream, part, nrec, count, weight, xs, training, out);
    CHECK_LAUNCH("bn_finalize_bwd");
    return 0;
}

int cffm_rows_resize_fwd(const float* src, long src_map_stride, float* dst, long dst_map_stride, int N, int h, int w, int H, int W, int C,
                         void* stream) {
    REQUIRE(N >= 0 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0, "rows_resize_fwd: bad sizes");
    if (!N) return 0;
    REQUIRE(src && dst && src_map_stride >= (long)h * w * C && dst_map_stride >= (long)H * W * C && src_map_stride % 4 == 0 && dst_map_stride % 4 == 0,
            "rows_resize_fwd: null / overlapping maps");
    const long total = (long)N * H * W * (C / 4);
    CFFM_LAUNCH(k_rows_resize_fwd, ((unsigned)std::min<long>((total + 255) / 256, 1 << 16)), (256), 0, (hipStream_t)stream, src, src_map_stride, dst,
                dst_map_stride, N, h, w, H, W, C);
    CHECK_LAUNCH("rows_resize_fwd");
    return 0;
}
int cffm_rows_resize_bwd(const float* ddst, long ddst_map_stride, float* dsrc, long dsrc_map_stride, int N, int h, int w, int H, int W, int C,
                         void* stream) {
    REQUIRE(N >= 0 && h >= 1 && w >= 1 && H >= 1 && W >= 1 && C >= 4 && C % 4 == 0, "rows_resize_bwd: bad sizes");
    if (!N) return 0;
    REQUIRE(ddst && dsrc && dsrc_map_stride >= (long)h * w * C && ddst_map_stride >= (long)H * W * C && dsrc_map_stride % 4 == 0 && ddst_map_stride % 4 == 0,
            "rows_resize_bwd: null / overlapping maps");
    const long total = (long)N * h * w * (C / 4);
    CFFM_LAUNCH(k_rows_resize_bwd, ((unsigned)std::min<long>((total + 255) / 256, 1 << 16)), (256), 0, (hipStream_t)stream, ddst, ddst_map_stride, dsrc,
                dsrc_map_stride, N, h, w, H, W, C);
    CHECK_LAUNCH("rows_resize_bwd");
    return 0;
}

// ------------------------------------------------------------------------------------------- clip data path
int cffm_clip_format(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H, int W,
                     int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3], int to_rgb,
                     float pad_val, int seg_pad_val, int reduce_zero_label, void* stream) {
    return cffm_clip_format_photo(frames, labels, out_img, out_lab, T, H, W, y1, x1, ch, cw, flip, Ho, Wo, mean, std, to_rgb, pad_val,
                                  seg_pad_val, reduce_zero_label, nullptr, nullptr, stream);
}
int cffm_clip_format_photo(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H, int W,
                           int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3], int to_rgb,
                           float pad_val, int seg_pad_val, int reduce_zero_label, const float* brightness_beta, const float* contrast_alpha,
                           void* stream) {
    return cffm_clip_format_hsv(frames, labels, out_img, out_lab, T, H, W, y1, x1, ch, cw, flip, Ho, Wo, mean, std, to_rgb, pad_val, seg_pad_val,
                                reduce_zero_label, brightness_beta, contrast_alpha, nullptr, nullptr, nullptr, stream);
}
int cffm_clip_format_hsv(const unsigned char* frames, const unsigned char* labels, float* out_img, long long* out_lab, int T, int H, int W,
                         int y1, int x1, int ch, int cw, int flip, int Ho, int Wo, const float mean[3], const float std[3], int to_rgb,
                         float pad_val, int seg_pad_val, int reduce_zero_label, const float* brightness_beta, const float* contrast_alpha,
                         const int* contrast_first, const float* saturation, const float* hue_shift, void* stream) {
    REQUIRE(T >= 0 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1, "clip_format: bad sizes");
    const bool any_photo = brightness_beta || contrast_alpha || saturation || hue_shift;
    REQUIRE(!any_photo || T <= CLIP_MAXT, "clip_format: at most %d frames with photometric parameters", CLIP_MAXT);
    REQUIRE(y1 >= 0 && x1 >= 0 && ch >= 0 && cw >= 0 && y1 + ch <= H && x1 + cw <= W && ch <= Ho && cw <= Wo,
            "clip_format: crop box %d+%d x %d+%d does not fit a %dx%d frame / %dx%d output", y1, ch, x1, cw, H, W, Ho, Wo);
    if (!T) return 0;
    REQUIRE(frames && out_img && mean && std && (!out_lab || labels), "clip_format: null");
    ClipFmt P;
    P.T = T; P.H = H; P.W = W; P.y1 = y1; P.x1 = x1; P.ch = ch; P.cw = cw; P.flip = flip ? 1 : 0; P.Ho = Ho; P.Wo = Wo;
    P.to_rgb = to_rgb ? 1 : 0; P.reduce_zero_label = reduce_zero_label ? 1 : 0; P.seg_pad = seg_pad_val; P.pad_val = pad_val;
    P.photo = any_photo ? 1 : 0;
    for (int t = 0; t < CLIP_MAXT; ++t) {   // NaN = "not taken" (the reference draws a parameter only when the branch is taken)
        const float bt = (brightness_beta && t < T) ? brightness_beta[t] : NAN, al = (contrast_alpha && t < T) ? contrast_alpha[t] : NAN;
        const float sa = (saturation && t < T) ? saturation[t] : NAN, hu = (hue_shift && t < T) ? hue_shift[t] : NAN;
        P.has_b[t] = bt == bt; P.has_c[t] = al == al; P.has_s[t] = sa == sa; P.has_h[t] = hu == hu;
        P.beta[t] = P.has_b[t] ? bt : 0.f; P.alpha[t] = P.has_c[t] ? al : 1.f;
        P.sat[t] = P.has_s[t] ? sa : 1.f;
        REQUIRE(!P.has_h[t] || (hu == (float)(int)hu && hu > -100000.f && hu < 100000.f), "clip_format: the hue shift of frame %d is not an integer", t);
        P.hue[t] = P.has_h[t] ? (int)hu : 0;
        P.c_first[t] = (contrast_first && t < T && contrast_first[t]) ? 1 : 0;
    }
    for (int c = 0; c < 3; ++c) {
        REQUIRE(std[c] != 0.f, "clip_format: zero std");
        P.mean[c] = mean[c];
        P.stdinv[c] = (float)(1.0 / (double)std[c]);     // mmcv.imnormalize: stdinv = 1 / np.float64(std), applied in float32
    }
    const long n = (long)T * Ho * Wo, want = (n + 255) / 256;
    CFFM_LAUNCH(k_clip_format, ((unsigned)(want < 8192 ? want : 8192)), (256), 0, (hipStream_t)stream, frames, labels, out_img, out_lab, P);
    CHECK_LAUNCH("clip_format");
    return 0;
}
int cffm_clip_resize(const unsigned char* frames, const unsigned char* labels, int T, int H, int W, unsigned char* out_frames,
                     unsigned char* out_labels, int Ho, int Wo, void* stream) {
    REQUIRE(T >= 0 && H >= 1 && W >= 1 && Ho >= 1 && Wo >= 1 && H < (1 << 15) && W < (1 << 15) && Ho < (1 << 15) && Wo < (1 << 15), "clip_resize: bad sizes");
    if (!T) return 0;
    REQUIRE((frames || labels) && (!frames || out_frames) && (!labels || out_labels), "clip_resize: null");
    ClipResize P;
    P.T = T; P.H = H; P.W = W; P.Ho = Ho; P.Wo = Wo;
    P.scale_x = 1.0 / ((double)Wo / (double)W);          // scale_x = 1. / inv_scale_x, as cv::resize computes it
    P.scale_y = 1.0 / ((double)Ho / (double)H);
    P.half = (W == 2 * Wo && H == 2 * Ho) ? 1 : 0;
    const long n = (long)T * Ho * Wo, want = (n + 255) / 256;
    CFFM_LAUNCH(k_clip_resize, ((unsigned)(want < 8192 ? want : 8192)), (256), 0, (hipStream_t)stream, frames, labels, out_frames, out_labels, P);
    CHECK_LAUNCH("clip_resize");
    return 0;
}

// ------------------------------------------------------------------------------------------- evaluation counts
int cffm_seg_counts(const long long* pred, const long long* label, long n, int num_classes, int ignore_index, int reduce_zero_label,
                    long long* counts, void* stream) {
    REQUIRE(n >= 0 && n < (1L << 32) && num_classes >= 1 && num_classes <= SEGCNT_MAXK, "seg_counts: bad sizes");
    if (!n) return 0;
    REQUIRE(pred && label && counts, "seg_counts: null");
    const long want = (n + 255) / 256;
    const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
    CFFM_LAUNCH(k_seg_counts, (blocks), (256), 0, (hipStream_t)stream, pred, label, n, num_classes, ignore_index, reduce_zero_label,
                (unsigned long long*)counts);
    CHECK_LAUNCH("seg_counts");
    return 0;
}

int cffm_vc_counts(const long long* gt, const long long* pred, int F, long npix, int n, long long* counts, void* stream) {
    REQUIRE(F >= 0 && npix >= 0 && n >= 1 && npix < (1L << 31), "vc_counts: bad sizes");
    if (F - n <= 0 || !npix) return 0;
    REQUIRE(gt && pred && counts, "vc_counts: null");
    REQUIRE(F - n <= 65535, "vc_counts: more than 65535 start frames in one call");
    CFFM_LAUNCH(k_vc_counts, ((unsigned)((npix + 255) / 256), (unsigned)(F - n)), (256), 0, (hipStream_t)stream, gt, pred, npix, n,
                (unsigned long long*)counts);
    CHECK_LAUNCH("vc_counts");
    return 0;
}

// ------------------------------------------------------------------------------------------- layer
// The layer on token rows (channels-last) on both sides: x_rows [B,4,HW,256] -> y_rows [B,HW,256] (new target frame).  No layout
// transposes and no copy of the input: `x_rows` itself is the NHWC stack the blocks read (the caller keeps it for the backward).
int cffm_layer_forward_rows(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_rows, float* y_rows,
                            const int* key_src, const int* q_dst, float* saved, float* scratch, void* stream) {
    REQUIRE(g && params && x_rows && y_rows && saved && scratch && depth >= 1, "layer_forward_rows: bad arguments");
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const long HW = g->HW, img = HW * CFFM_C;
    float* blk0 = saved + up((long)g->B * 4 * img);
    TRY(param_prep(params, depth, blk0, L.total, L, stream));
    for (int i = 0; i < depth; ++i) {
        float* ws = blk0 + (long)i * L.total;
        const float* tgt = (i == 0) ? x_rows + 3 * img : blk0 + (long)(i - 1) * L.total + L.x2;
        const long tgt_bs = (i == 0) ? 4 * img : img;
        TRY(block_forward_impl(g, &params[i], x_rows, 4 * img, tgt, tgt_bs, key_src, q_dst, ws, scratch, stream));
    }
#ifdef CFFM_EMU
    memcpy(y_rows, blk0 + (long)(depth - 1) * L.total + L.x2, (size_t)g->B * img * sizeof(float));
#else
    REQUIRE(hipMemcpyAsync(y_rows, blk0 + (long)(depth - 1) * L.total + L.x2, (size_t)g->B * img * sizeof(float), hipMemcpyDeviceToDevice,
                           (hipStream_t)stream) == hipSuccess, "layer_forward_rows: copy failed");
#endif
    return 0;
}
// dy_rows [B,HW,256] -> dx_rows [B,4,HW,256] and every parameter gradient; x_rows as given to cffm_layer_forward_rows
int cffm_layer_backward_rows(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                             const float* x_rows, const float* dy_rows, float* dx_rows, const int* key_src, const int* q_dst,
                             const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, void* stream) {
    REQUIRE(g && params && grads && x_rows && dy_rows && dx_rows && saved && scratch && depth >= 1, "layer_backward_rows: bad arguments");
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const Scratch S = scratch_layout(g);
    const long HW = g->HW, img = HW * CFFM_C;
    const float* blk0 = saved + up((long)g->B * 4 * img);
    tail_reset();
    for (int i = depth - 1; i >= 0; --i) {
        const float* ws = blk0 + (long)i * L.total;
        const float* tgt = (i == 0) ? x_rows + 3 * img : blk0 + (long)(i - 1) * L.total + L.x2;
        const long tgt_bs = (i == 0) ? 4 * img : img;
        // block i reads the gradient of its output from buffer (depth - 1 - i) & 1 and writes the next block's into the other one
        // (never in place: its weight-gradient GEMMs on the side stream still read `dout` while ln_pool_bwd writes dtgt)
        float* dtgt = (i == 0) ? dx_rows + 3 * img : scratch + (((depth - i) & 1) ? S.a2 : S.a);
        const long dtgt_bs = (i == 0) ? 4 * img : img;
        // the last block reads the caller's gradient directly
        const float* dout = (i == depth - 1) ? dy_rows : scratch + (((depth - 1 - i) & 1) ? S.a2 : S.a);
        TRY(block_backward_impl(g, &params[i], &grads[i], x_rows, 4 * img, tgt, tgt_bs, key_src, q_dst, inv_ptr, inv_idx, ws, dout, dtgt, dtgt_bs,
                                scratch, 1, (depth - 1 - i) & 1, i, depth, stream));
    }
    // the reference frames of every block in one pass + the CFFA parameter gradients, then the last block's parameter-gradient tail
    RedJobs* tail_jobs = nullptr;
#ifndef CFFM_EMU
    if (tail_on_main() && g_tail.has) {
        if (g_tail.cs) (void)hipStreamWaitEvent((hipStream_t)stream, g_side.cs_order, 0);
        g_tail.has = g_tail.cs = false;
        tail_jobs = &g_tail.jobs;
    }
#endif
    TRY(cffa_finish(g, depth, params, grads, depth - 1, 0, blk0, L.total, x_rows, 4 * img, dx_rows, 4 * img, 0, stream, tail_jobs));
    TRY(tail_flush((hipStream_t)stream, tail_on_main()));
    side_join_all((hipStream_t)stream);
    return 0;
}

// y_tgt_nchw: the new target frame, images y_bs floats apart; y_full (or NULL): the reference's whole output [B,4,C,H,W], whose
// frames 0..2 are copies of the input (cffm_transformer.py:826) -- copied on the side stream while the blocks run
static int layer_forward_impl(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_nchw,
                              float* y_tgt_nchw, long y_bs, float* y_full, const int* key_src, const int* q_dst, float* saved,
                              float* scratch, void* stream) {
    REQUIRE(g && params && x_nchw && y_tgt_nchw && saved && scratch && depth >= 1, "layer_forward: bad arguments");
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const long HW = g->HW, img = HW * CFFM_C;
    float* xs = saved;  // NHWC stack [B,4,HW,C]
    float* blk0 = saved + up((long)g->B * 4 * img);
    // the parameter-derived tables do not depend on x: they are built on the side stream while the input is transposed
    hipStream_t st = (hipStream_t)stream;
    side_init(st);
    static int fused_prep = -1;      // CFFM_PREP_FUSED=0: the prep on the side stream beside the transpose (A/B)
    if (fused_prep < 0) { const char* e = cffm_tune("CFFM_PREP_FUSED"); fused_prep = (e && e[0] == '0') ? 0 : 1; }
    if (fused_prep && depth <= PREP_MAXD) {
        // ONE launch: the input transpose (frames 0..2 also copied to y_full) + the parameter prep of every block (k_transpose_prep)
        PROF(ST_TRANSPOSE);
        PrepArgs a;
        const int pnx = prep_args(a, params, 0, depth, blk0, L.total, L);
        TrArgs T;
        T.src = x_nchw; T.dst = xs; T.rows = CFFM_C; T.cols = (int)HW; T.src_bs = img; T.dst_bs = img; T.add_mod = 4; T.add_skip = 3; T.copy_dst = y_full;
        T.gx = ((int)HW + 63) / 64; T.gy = (CFFM_C + 63) / 64; T.gz = g->B * 4;
        CFFM_LAUNCH(k_transpose_prep, ((unsigned)(T.gx * T.gy * T.gz + pnx * depth)), (256), 0, st, T, a, pnx);
        CHECK_LAUNCH("transpose + param_prep");
        g_prep_join.pending = false;
    } else {
    side_fork_mark(st, 0);
    // NCHW -> NHWC of the four frames; frames 0..2 also go to y_full as they are (the reference's pass-through frames) with the same read
    TRY(transpose_add(x_nchw, xs, g->B * 4, CFFM_C, (int)HW, img, img, nullptr, 4, 3, stream, y_full));
    hipStream_t sd = side_fork_take(st, 0);
    TRY(param_prep(params, depth, blk0, L.total, L, (void*)sd));
    side_mark(sd, st, 0);
    {
        static int early = -1;           // CFFM_PREP_JOIN=early: join in front of the first block (A/B)
        if (early < 0) { const char* e = cffm_tune("CFFM_PREP_JOIN"); early = (e && e[0] == 'e') ? 1 : 0; }
        g_prep_join.pending = sd != st && !early;      // joined behind the first block's ln_pool_fwd (block_forward_impl)
        g_prep_join.side = sd;
        if (!g_prep_join.pending) side_join(sd, st, 0);
    }
    }
#ifndef CFFM_EMU
    g_side.used = 0;   // (that was this call's only side branch, joined here)
#endif
    for (int i = 0; i < depth; ++i) {
        float* ws = blk0 + (long)i * L.total;
        const float* tgt = (i == 0) ? xs + 3 * img : blk0 + (long)(i - 1) * L.total + L.x2;
        const long tgt_bs = (i == 0) ? 4 * img : img;
        TRY(block_forward_impl(g, &params[i], xs, 4 * img, tgt, tgt_bs, key_src, q_dst, ws, scratch, stream));
    }
    TRY(cffm_transpose(blk0 + (long)(depth - 1) * L.total + L.x2, y_tgt_nchw, g->B, (int)HW, CFFM_C, img, y_bs, stream));
    return 0;
}
int cffm_layer_forward(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_nchw,
                       float* y_tgt_nchw, const int* key_src, const int* q_dst, float* saved, float* scratch,
                       void* stream) {
    REQUIRE(g, "layer_forward: bad arguments");
    return layer_forward_impl(g, depth, params, x_nchw, y_tgt_nchw, g->HW * CFFM_C, nullptr, key_src, q_dst, saved, scratch, stream);
}
int cffm_layer_forward_full(const cffm_geom* g, int depth, const cffm_block_params* params, const float* x_nchw,
                            float* y_full_nchw, const int* key_src, const int* q_dst, float* saved, float* scratch,
                            void* stream) {
    REQUIRE(g && y_full_nchw && y_full_nchw != x_nchw, "layer_forward_full: bad arguments");
    const long img = g->HW * CFFM_C;
    return layer_forward_impl(g, depth, params, x_nchw, y_full_nchw + 3 * img, 4 * img, y_full_nchw, key_src, q_dst, saved, scratch, stream);
}

// blocks first_block, first_block - 1, ..., last_block of the layer backward (depth - 1 >= first >= last >= 0): the piece with
// first == depth - 1 starts from dy (its layout transpose included), the piece with last == 0 ends with the transpose of the
// gradient stack into dx; the intermediate state lives in `scratch`, so consecutive pieces must be issued in order on one stream.
// Data-parallel training calls it block by block and starts the gradient all-reduce of block i while block i - 1 runs
// (vss_cffm_amd/distributed.py; the reference's DDP does the same with its buckets: mmseg/apis/train.py:57-65).
static int layer_backward_impl(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                               const float* dy_tgt_nchw, long dy_bs, const float* dy_full, float* dx_nchw, const int* key_src,
                               const int* q_dst, const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch,
                               int first_block, int last_block, void* stream) {
    REQUIRE(g && params && grads && dy_tgt_nchw && dx_nchw && saved && scratch && depth >= 1, "layer_backward: bad arguments");
    REQUIRE(first_block < depth && last_block >= 0 && first_block >= last_block, "layer_backward: bad block range %d..%d of %d", first_block,
            last_block, depth);
    cffm_block_ws L;
    cffm_block_ws_layout(g, &L);
    const Scratch S = scratch_layout(g);
    const long HW = g->HW, img = HW * CFFM_C;
    const float* xs = saved;
    const float* blk0 = saved + up((long)g->B * 4 * img);
    float* dxs = scratch + S.dxs;   // NHWC gradient stack [B,4,HW,C]
    REQUIRE(dy_bs >= img, "layer_backward: dy batch stride %ld < %ld", dy_bs, img);
    if (first_block == depth - 1) TRY(cffm_transpose(dy_tgt_nchw, scratch + S.a, g->B, CFFM_C, (int)HW, dy_bs, img, stream));
    tail_reset();
    for (int i = first_block; i >= last_block; --i) {
        const float* ws = blk0 + (long)i * L.total;
        const float* tgt = (i == 0) ? xs + 3 * img : blk0 + (long)(i - 1) * L.total + L.x2;
        const long tgt_bs = (i == 0) ? 4 * img : img;
        // gradient of the current block's output target [B,HW,C]: buffer (depth - 1 - i) & 1 of {a, a2}; the target-frame gradient
        // goes to the stack for block 0, otherwise into the OTHER buffer (the block's weight-gradient GEMMs on the side stream
        // still read `dout` while ln_pool_bwd writes it)
        float* dcur = scratch + (((depth - 1 - i) & 1) ? S.a2 : S.a);
        float* dtgt = (i == 0) ? dxs + 3 * img : scratch + (((depth - i) & 1) ? S.a2 : S.a);
        const long dtgt_bs = (i == 0) ? 4 * img : img;
        TRY(block_backward_impl(g, &params[i], &grads[i], xs, 4 * img, tgt, tgt_bs, key_src, q_dst, inv_ptr, inv_idx, ws, dcur, dtgt, dtgt_bs, scratch,
                                1, (depth - 1 - i) & 1, i, depth, stream));
    }
    // The reference frames of every block of the range in ONE pass (they pass through the layer unchanged: one read of x_ref, one write
    // of dx_ref per range instead of a read + read-modify-write per block) and the range's CFFA parameter gradients.  A caller that
    // walks the layer block by block (data-parallel training, BlockwiseReducer) gets a pass per block, accumulating into dx_ref, so
    // that every block's gradient slice is complete when its range returns.
    RedJobs* tail_jobs = nullptr;
#ifndef CFFM_EMU
    if (last_block == 0 && tail_on_main() && g_tail.has) {     // the last block's record reductions ride in the CFFA's final reduction launch
        if (g_tail.cs) (void)hipStreamWaitEvent((hipStream_t)stream, g_side.cs_order, 0);
        g_tail.has = g_tail.cs = false;
        tail_jobs = &g_tail.jobs;
    }
#endif
    TRY(cffa_finish(g, depth, params, grads, first_block, last_block, blk0, L.total, xs, 4 * img, dxs, 4 * img, first_block != depth - 1, stream, tail_jobs));
    // (dy_full: the upstream gradient of the whole [B,4,C,H,W] output -- its pass-through frames 0..2 join dx in the same pass)
    if (last_block == 0) {
        // the last block's parameter-gradient tail stays on the caller's stream, in front of the output transpose: no fork behind the
        // last ln_pool_bwd at all (the transpose on a side stream of its own was queued AHEAD of the blocks' side work by the graph
        // executor and held it up until the chain had finished: 0.95 ms per step)
        hipStream_t st = (hipStream_t)stream;
        if (tail_on_main()) TRY(tail_flush(st, true));
        TRY(transpose_add(dxs, dx_nchw, g->B * 4, (int)HW, CFFM_C, img, img, dy_full, 4, 3, stream));
    }
    TRY(tail_flush((hipStream_t)stream));
    side_join_all((hipStream_t)stream);   // every parameter gradient of the range is complete behind this point of the stream
    return 0;
}
int cffm_layer_backward_range(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                              const float* dy_tgt_nchw, long dy_bs, float* dx_nchw, const int* key_src, const int* q_dst,
                              const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, int first_block,
                              int last_block, void* stream) {
    return layer_backward_impl(g, depth, params, grads, dy_tgt_nchw, dy_bs, nullptr, dx_nchw, key_src, q_dst, inv_ptr, inv_idx, saved,
                               scratch, first_block, last_block, stream);
}
int cffm_layer_backward_full(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                             const float* dy_full_nchw, float* dx_nchw, const int* key_src, const int* q_dst, const int* inv_ptr,
                             const int* inv_idx, const float* saved, float* scratch, int first_block, int last_block, void* stream) {
    REQUIRE(g && dy_full_nchw && dy_full_nchw != dx_nchw, "layer_backward_full: bad arguments");
    const long img = g->HW * CFFM_C;
    return layer_backward_impl(g, depth, params, grads, dy_full_nchw + 3 * img, 4 * img, dy_full_nchw, dx_nchw, key_src, q_dst, inv_ptr,
                               inv_idx, saved, scratch, first_block, last_block, stream);
}
int cffm_layer_backward(const cffm_geom* g, int depth, const cffm_block_params* params, const cffm_block_grads* grads,
                        const float* dy_tgt_nchw, long dy_bs, float* dx_nchw, const int* key_src, const int* q_dst,
                        const int* inv_ptr, const int* inv_idx, const float* saved, float* scratch, void* stream) {
    return cffm_layer_backward_range(g, depth, params, grads, dy_tgt_nchw, dy_bs, dx_nchw, key_src, q_dst, inv_ptr, inv_idx, saved, scratch,
                                     depth - 1, 0, stream);
}

}  // extern "C"

#ifdef CFFM_EMU
#include "hipemu_impl.h"
#endif
