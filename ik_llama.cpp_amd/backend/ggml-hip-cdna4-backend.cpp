// ggml-hip-cdna4-backend.cpp -- the ggml-backend shim: makes libggml-hip-cdna4.so a drop-in for ik_llama.cpp's CUDA backend.
//
// The reference binds its GPU backend at COMPILE time through the 13 `ggml_backend_cuda_*` C symbols of
// ggml/include/ggml-cuda.h:24-47 (+ ggml_backend_cuda_reg_devices, ggml-backend.cpp:456-489) and the three vtables of
// ggml/src/ggml-backend-impl.h:18-130.  This file exports exactly those symbols and fills those vtables; every
// op is forwarded to the C ABI of include/ggml_hip_cdna4.h.  It is compiled against the reference's own headers (which are NOT
// copied into this repo), so it only builds where a reference checkout exists (`make REF=/path/to/ik_llama.cpp`).
//
// What the unmodified libllama / llama-bench get from it (Makefile.llama links them against this library; tests/test_gpu_llama.py):
//   * device buffers with the CUDA backend's row over-allocation; `_R4` tensors are un-interleaved ONCE at upload into the MI355X-native
//     base tiling (get_tensor re-interleaves, so a round trip is exact) -- no pointer-keyed shadow cache;
//   * the split buffer type of `-sm graph` (ggml-cuda.cu:805-1402): per-device slices of a tensor by split_dim, uploaded by set_tensor;
//   * GGML_OP_REDUCE across the backends of one process (peer access over xGMI), MUL_MAT / MUL_MAT_ID / FUSED_UP_GATE /
//     MOE_FUSED_UP_GATE (+ the 2-node MoE decode fusion of ggml-cuda.cu:3062-3185) on the supported quant types;
//   * HIP-graph capture of repeated compute graphs (ggml-cuda.cu:4408-4760), the `k=v` parameter string (:5299-5389).
// supports_op is true only for the hot path; every other op stays on whichever backend owns it (ggml-backend.cpp:1314-1360).
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-cuda.h"
#include "ggml_hip_cdna4.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <set>
#include <vector>

#define SHIM_MAX_DEVICES GGML_CUDA_MAX_DEVICES
#define MATRIX_ROW_PADDING 512          // ggml-cuda/common.cuh:63 -- quantized rows are over-allocated like the CUDA backend does

static ggml_log_callback g_log_cb = nullptr; static void *g_log_ud = nullptr;
static void shim_log(enum ggml_log_level lvl, const char *fmt, ...) {
    char buf[768]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
    if (g_log_cb) g_log_cb(lvl, buf, g_log_ud); else fputs(buf, stderr);
}
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "ggml-hip-cdna4: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); GGML_ABORT("HIP error"); } } while (0)
// A C-ABI call that fails while a HIP graph is being captured (an entry point stricter than supports_op, a workspace that would have to grow: CDNA4_E_NOMEM under
// capture) is not fatal: the capture is abandoned and the graph runs eagerly (graph_compute_impl catches capture_failed).  Outside a capture it aborts like the
// reference's CUDA_CHECK.
struct capture_failed { int rc; };
static thread_local bool t_capturing = false;
static void check(int rc, const char *what) {
    if (rc == CDNA4_OK) return;
    if (t_capturing) { fprintf(stderr, "ggml-hip-cdna4: %s failed during graph capture (%s): running this graph eagerly\n", what, cdna4_last_error()); throw capture_failed{rc}; }
    fprintf(stderr, "ggml-hip-cdna4: %s: %s\n", what, cdna4_last_error()); GGML_ABORT("cdna4 op failed");
}

// ---------------------------------------------------------------------------------------------- devices
// Logical device d runs on physical device d % (real count).  GGML_CDNA4_FAKE_DEVICES=N presents N logical devices on a box with fewer
// GPUs: the multi-backend paths of libllama (-sm layer / -sm graph: split buffers, GGML_OP_REDUCE) can then be exercised end to end on a
// single MI355X (tests/test_gpu_llama.py); with N unset logical == physical.
static int real_device_count() { static int n = cdna4_get_device_count(); return n; }
static int device_count() {
    static int n = [] { const char *e = getenv("GGML_CDNA4_FAKE_DEVICES"); int r = real_device_count(); if (e && r > 0) { int f = atoi(e); if (f > 0) r = std::min(f, SHIM_MAX_DEVICES); } return r; }();
    return n;
}
static int phys(int dev) { const int r = real_device_count(); return r > 0 ? dev % r : 0; }
static void set_device(int dev) { HIP_CHECK(hipSetDevice(phys(dev))); }

// one utility context per physical device for the buffer-level kernels (_R4 re-tiling at upload)
static cdna4_context *util_ctx(int dev) {
    static std::mutex mu; static cdna4_context *ctxs[SHIM_MAX_DEVICES] = {nullptr};
    std::lock_guard<std::mutex> lock(mu);
    const int p = phys(dev);
    if (!ctxs[p]) { ctxs[p] = cdna4_init(p); if (!ctxs[p]) { fprintf(stderr, "ggml-hip-cdna4: %s\n", cdna4_last_error()); GGML_ABORT("cdna4_init failed"); } }
    return ctxs[p];
}

// two kinds of row-interleaved weight types: the six of SURVEY 8 a8 are re-tiled on the DEVICE (cdna4_unrepack_r4) and served with their _R4 kernels' activation
// arithmetic (CDNA4_TYPE_PRETILED); every other interleaved form of a served base type (IQ2_K_R4 ... IQ5_KS_R4, the ones the CUDA backend lists, ggml-cuda.cu:4893-4898,
// and the CPU-only Q4_0_R8 ... IQ2_BN_R4) is re-tiled on the HOST (cdna4_retile_r4_host) and from then on IS a tensor of its base type as far as the C ABI is concerned
static bool is_r4h_type(int t) { return cdna4_retile_r4_host_base_type(t) >= 0; }
static bool is_r4_type(int t) { return (t >= 200 && t < 300 && cdna4_type_supported(t)) || is_r4h_type(t); }
static int  r4_base_type(int t) { return is_r4h_type(t) ? cdna4_retile_r4_host_base_type(t) : t - 200; }      // (device re-tiled: enum ggml_type has the _R4 ids of the six at base + 200, ggml.h:466-475)
static int  r4_rows(int t) { return is_r4h_type(t) ? cdna4_retile_r4_host_rows(t) : 4; }                   // rows per interleaved group

#include "shim_buffers.inc"
#include "shim_support.inc"
#include "shim_fusion.inc"
#include "shim_compute.inc"
#include "shim_graphs.inc"


static GGML_CALL const char *be_name(ggml_backend_t be) { return ((shim_context *)be->context)->name.c_str(); }
static GGML_CALL void be_free(ggml_backend_t be) {
    auto *c = (shim_context *)be->context; set_device(c->device);
    { std::lock_guard<std::mutex> lock(g_shims_mu); if (c->device < GGML_CUDA_MAX_DEVICES && g_shims[c->device] == c) g_shims[c->device] = nullptr; }
    (void)hipStreamSynchronize(c->stream); drop_graphs(c);
    if (getenv("GGML_CDNA4_STATS") && c->device < GGML_CUDA_MAX_DEVICES) fprintf(stderr, "cdna4[%d] small uploads queued on the compute stream instead of blocking copies: %ld\n", c->device, g_stage[c->device].n_staged);
    stage_detach(c->device, c->stream);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] graph_compute calls: %ld eager, %ld captured, %ld replayed, %ld capture failures, %ld too small / not capturable; fused attention + attn_output launches issued or captured: %ld\n", c->device, c->n_eager, c->n_captured, c->n_replayed, c->n_capture_failed, c->n_small, c->n_fused_attn);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] fused launches issued or captured: ADD+RMS_NORM %ld, ROPE+ROPE+KV stores %ld, shared-input MUL_MATs %ld, RMS_NORM in mat-mul %ld, MUL_MAT+ADD %ld, "
                                            "RMS_NORM+q,k,v+ROPE+KV store %ld, MoE blocks %ld, attention -> attn_output (q8 hand-off, or one launch) %ld, q/k norms+ROPE+KV stores %ld, RMS_NORM in MoE router %ld\n", c->device, c->n_fuse[0], c->n_fuse[1], c->n_fuse[2], c->n_fuse[3], c->n_fuse[4], c->n_fuse[5], c->n_fuse[6], c->n_fuse[7], c->n_fuse[8], c->n_fuse[9]);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] host time: graph_compute %.1f ms, synchronize %.1f ms (%ld calls), set_async %.1f ms (%ld calls, %.1f MB), get_async %.1f ms (%ld calls, %.1f MB)\n", c->device,
                                            c->t_compute * 1e3, c->t_sync * 1e3, c->n_sync, c->t_set * 1e3, c->n_set, c->b_set / 1e6, c->t_get * 1e3, c->n_get, c->b_get / 1e6);
    if (c->x32) (void)hipFree(c->x32);
    if (c->attn_q8) (void)hipFree(c->attn_q8);
    if (c->slots_ev) (void)hipEventDestroy(c->slots_ev); if (c->slots_host) (void)hipHostFree(c->slots_host); if (c->slots_dev) (void)hipFree(c->slots_dev);
    if (c->ev) (void)hipEventDestroy(c->ev); if (c->ev2) (void)hipEventDestroy(c->ev2); (void)hipStreamDestroy(c->stream); cdna4_free(c->ctx); delete c; delete be;
}
static GGML_CALL ggml_backend_buffer_type_t be_default_buft(ggml_backend_t be) { return ggml_backend_cuda_buffer_type(((shim_context *)be->context)->device); }
static GGML_CALL void be_set_async(ggml_backend_t be, ggml_tensor *t, const void *d, size_t off, size_t size) {
    auto *c = (shim_context *)be->context; set_device(c->device);
    if (buffer_is_ours(t->buffer) && r4_candidate(t)) r4_set_state(t->buffer, t, false);          // bytes arrive interleaved; re-tiled at first use (abi_type)
    const double t0 = now_s();
    HIP_CHECK(hipMemcpyAsync((char *)t->data + off, d, size, hipMemcpyHostToDevice, c->stream));
    c->t_set += now_s() - t0; ++c->n_set; c->b_set += size;
}
static GGML_CALL void be_get_async(ggml_backend_t be, const ggml_tensor *t, void *d, size_t off, size_t size) {
    auto *c = (shim_context *)be->context; set_device(c->device);
    if (buffer_is_ours(t->buffer) && r4_candidate(t) && r4_is_tiled(t->buffer, t)) { HIP_CHECK(hipStreamSynchronize(c->stream)); buf_get_tensor(t->buffer, t, d, off, size); return; }
    const double t0 = now_s();
    HIP_CHECK(hipMemcpyAsync(d, (const char *)t->data + off, size, hipMemcpyDeviceToHost, c->stream));
    c->t_get += now_s() - t0; ++c->n_get; c->b_get += size;
}
static GGML_CALL bool be_cpy_async(ggml_backend_t src_be, ggml_backend_t dst_be, const ggml_tensor *src, ggml_tensor *dst) {
    if (!ggml_backend_is_cuda(src_be) || !ggml_backend_is_cuda(dst_be) || !buffer_is_ours(src->buffer) || !buffer_is_ours(dst->buffer)) return false;
    if (r4_candidate(src) || r4_candidate(dst)) return false;                                     // (weights: the synchronous path keeps their tiling state)
    auto *s = (shim_context *)src_be->context; auto *d = (shim_context *)dst_be->context;
    if (s != d) {               // copy on the destination's stream, after the source's queued work (ggml-cuda.cu cpy_tensor_async)
        set_device(s->device); HIP_CHECK(hipEventRecord(s->ev, s->stream));
        set_device(d->device); HIP_CHECK(hipStreamWaitEvent(d->stream, s->ev, 0));
    }
    set_device(d->device);
    HIP_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, d->stream));
    return true;
}
static GGML_CALL void be_sync(ggml_backend_t be) { auto *c = (shim_context *)be->context; set_device(c->device); const double t0 = now_s(); const unsigned long long sq = stage_seq(c->device); HIP_CHECK(hipStreamSynchronize(c->stream)); stage_synced(c->device, c->stream, sq); c->t_sync += now_s() - t0; ++c->n_sync; }
static GGML_CALL bool be_supports_buft(ggml_backend_t be, ggml_backend_buffer_type_t t) {
    if (t->iface.get_name == split_buft_name) return true;
    return t->iface.get_name == buft_get_name && ((shim_buft_ctx *)t->context)->device == ((shim_context *)be->context)->device;
}
static GGML_CALL bool be_offload_op(ggml_backend_t be, const ggml_tensor *op) {      // ggml-cuda.cu:5180-5216: large batches only; MoE scaled by the expert fan-out
    auto *c = (shim_context *)be->context; int min_batch = c->params.offload_batch_size;
    if (op->op != GGML_OP_MUL_MAT && op->op != GGML_OP_MUL_MAT_ID && op->op != GGML_OP_MOE_FUSED_UP_GATE && op->op != GGML_OP_FUSED_UP_GATE) return false;
    if (is_r4_type(op->src[0]->type)) return false;          // _R4 weights in host memory stay with the CPU kernels built for them
    if (op->op == GGML_OP_MUL_MAT_ID || op->op == GGML_OP_MOE_FUSED_UP_GATE) {
        if (c->params.offload_batch_size_per_byte >= 0) { const ggml_tensor *w = op->src[0]; min_batch = (int)(1. * c->params.offload_batch_size_per_byte * ggml_row_size(w->type, w->ne[0]) / w->ne[0]); }
        const ggml_tensor *ids = op->op == GGML_OP_MUL_MAT_ID ? op->src[2] : op->src[3];
        const int64_t batch = op->ne[2]; if (batch < min_batch) return false;
        return batch * ids->ne[0] >= (int64_t)min_batch * op->src[0]->ne[2];
    }
    return op->ne[1] >= min_batch;
}
static GGML_CALL ggml_backend_event_t be_event_new(ggml_backend_t be) { auto *c = (shim_context *)be->context; set_device(c->device); hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return new ggml_backend_event{be, e}; }
static GGML_CALL void be_event_free(ggml_backend_event_t ev) { HIP_CHECK(hipEventDestroy((hipEvent_t)ev->context)); delete ev; }
static GGML_CALL void be_event_record(ggml_backend_event_t ev) { auto *c = (shim_context *)ev->backend->context; set_device(c->device); HIP_CHECK(hipEventRecord((hipEvent_t)ev->context, c->stream)); }
static GGML_CALL void be_event_wait(ggml_backend_t be, ggml_backend_event_t ev) { auto *c = (shim_context *)be->context; set_device(c->device); HIP_CHECK(hipStreamWaitEvent(c->stream, (hipEvent_t)ev->context, 0)); }
static GGML_CALL void be_event_sync(ggml_backend_event_t ev) { HIP_CHECK(hipEventSynchronize((hipEvent_t)ev->context)); }

static const ggml_backend_i k_backend_iface = { be_name, be_free, be_default_buft, be_set_async, be_get_async, be_cpy_async, be_sync,
                                                nullptr, nullptr, nullptr, nullptr, be_graph_compute, be_supports_op, be_supports_buft, be_offload_op,
                                                be_event_new, be_event_free, be_event_record, be_event_wait, be_event_sync };

extern "C" {

// `params` is the reference's "k=v,..." string (ggml-cuda.cu:5299-5389); `model` the opaque key of ggml_backend_cuda_invalidate_graphs
GGML_CALL ggml_backend_t ggml_backend_cuda_init(int device, const void *params, const void *model) {
    if (device < 0 || device >= device_count()) { shim_log(GGML_LOG_LEVEL_ERROR, "ggml-hip-cdna4: invalid device %d\n", device); return nullptr; }     // ggml-cuda.cu:5392-5395
    if (const char *pm = getenv("GGML_CDNA4_POISON_MB")) {      // debug (see GGML_CDNA4_CHECK_NAN): later allocations of this process start as NaN bits, not as fresh zero pages
        void *pz = nullptr; const size_t pb = (size_t)atol(pm) << 20; set_device(device);
        if (pb && hipMalloc(&pz, pb) == hipSuccess) { (void)hipMemset(pz, 0xff, pb); (void)hipDeviceSynchronize(); (void)hipFree(pz); } else (void)hipGetLastError();
    }
    cdna4_context *ctx = cdna4_init(phys(device));
    if (!ctx) { shim_log(GGML_LOG_LEVEL_ERROR, "ggml-hip-cdna4: %s\n", cdna4_last_error()); return nullptr; }
    set_device(device); hipStream_t st; HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto *c = new shim_context{device, ctx, st, std::string(GGML_CUDA_NAME) + std::to_string(device)};
    if (getenv("GGML_CDNA4_PREFILL_INT8")) check(cdna4_set_prefill_mode(ctx, CDNA4_PREFILL_INT8_DOT), "prefill mode");      // CPU-arithmetic parity mode for prompts
    c->params = parse_params(getenv("GGML_CDNA4_PARAMS") ? getenv("GGML_CDNA4_PARAMS") : (const char *)params); c->model = model;      // (env: developer override)
    HIP_CHECK(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming)); HIP_CHECK(hipEventCreateWithFlags(&c->ev2, hipEventDisableTiming));
    HIP_CHECK(hipEventCreateWithFlags(&c->slots_ev, hipEventDisableTiming));
    if (hipHostMalloc((void **)&c->slots_host, sizeof(void *) * shim_context::MAX_SLOTS, hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void **)&c->slots_dev, sizeof(void *) * shim_context::MAX_SLOTS) != hipSuccess) { (void)hipGetLastError(); c->slots_host = nullptr; }        // (no table: no graph capture)
    if (c->params.enable_p2p) for (int p = 0, n = real_device_count(); p < n; ++p) {          // REDUCE reads / writes the peers' buffers directly (xGMI)
        int can = 0;
        if (p != phys(device) && hipDeviceCanAccessPeer(&can, phys(device), p) == hipSuccess && can) { if (hipDeviceEnablePeerAccess(p, 0) != hipSuccess) (void)hipGetLastError(); }
    }
    { std::lock_guard<std::mutex> lock(g_shims_mu); if (device < GGML_CUDA_MAX_DEVICES) g_shims[device] = c; }
    stage_attach(device, st);
    if (hipMalloc(&c->attn_q8, shim_context::ATTN_Q8_BYTES) != hipSuccess) { (void)hipGetLastError(); c->attn_q8 = nullptr; }      // (without it the attention hands its row over as f32)
    // the library's scratch workspace (activation images, split-K slabs, V^T of the prompt attention) grows on demand -- free + allocate + device synchronize, 3-4 times inside
    // the FIRST prompt pass a context sees (llama-bench warms up with a one-token prompt: its first timed pp repetition paid for it).  Sized once here for 512-token ubatches
    // of rows up to 28672 values; larger needs still grow it.  GGML_CDNA4_WS_MB=<n> (0: grow on demand only)
    { const long mb = getenv("GGML_CDNA4_WS_MB") ? atol(getenv("GGML_CDNA4_WS_MB")) : 192; if (mb > 0 && cdna4_reserve_workspace(ctx, (size_t)mb << 20) != CDNA4_OK) { (void)hipGetLastError(); shim_log(GGML_LOG_LEVEL_WARN, "ggml-hip-cdna4: device %d: could not reserve the %ld MB prompt workspace (%s); it will be grown on demand\n", device, mb, cdna4_last_error()); } }
    return new ggml_backend{shim_guid(), k_backend_iface, c};
}
GGML_CALL bool ggml_backend_is_cuda(ggml_backend_t be) { return be != nullptr && ggml_guid_matches(be->guid, shim_guid()); }
GGML_CALL int  ggml_backend_cuda_get_device_count(void) { return device_count(); }
GGML_CALL void ggml_backend_cuda_get_device_description(int device, char *d, size_t n) { if (cdna4_get_device_description(phys(device), d, n) != CDNA4_OK && n) d[0] = 0; }
GGML_CALL void ggml_backend_cuda_get_device_memory(int device, size_t *fr, size_t *tot) { if (cdna4_get_device_memory(phys(device), fr, tot) != CDNA4_OK) { *fr = 0; *tot = 0; } }
GGML_CALL bool ggml_backend_cuda_register_host_buffer(void *p, size_t n) { if (getenv("GGML_CUDA_REGISTER_HOST") == nullptr) return false; if (hipHostRegister(p, n, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return false; } return true; }
GGML_CALL void ggml_backend_cuda_unregister_host_buffer(void *p) { if (getenv("GGML_CUDA_REGISTER_HOST") == nullptr) return; if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError(); }
GGML_CALL void ggml_backend_cuda_log_set_callback(ggml_log_callback cb, void *ud) { g_log_cb = cb; g_log_ud = ud; }
// captured graphs hold raw device addresses of the model's tensors: drop them when the model behind them is swapped (src/llama-reload.cpp:1121)
GGML_CALL void ggml_backend_cuda_invalidate_graphs(const void *model) {
    std::lock_guard<std::mutex> lock(g_shims_mu);
    for (auto *c : g_shims) if (c && (model == nullptr || c->model == model)) { set_device(c->device); (void)hipStreamSynchronize(c->stream); drop_graphs(c); }
}

static GGML_CALL ggml_backend_t reg_init(const char *params, void *user) { return ggml_backend_cuda_init((int)(intptr_t)user, params, nullptr); }
GGML_CALL int ggml_backend_cuda_reg_devices(void) {                       // ggml-backend.cpp:456-489, ggml-cuda.cu:5518-5529
    const int n = ggml_backend_cuda_get_device_count();
    for (int i = 0; i < n; ++i) { char name[64]; snprintf(name, sizeof(name), "%s%d", GGML_CUDA_NAME, i); ggml_backend_register(name, reg_init, ggml_backend_cuda_buffer_type(i), (void *)(intptr_t)i); }
    return n;
}

} // extern "C"
