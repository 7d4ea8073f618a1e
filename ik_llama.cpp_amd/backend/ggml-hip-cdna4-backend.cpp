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

// ---------------------------------------------------------------------------------------------- device buffer
struct shim_buffer_ctx {
    int device; void *base;
    // _R4 tensors of this buffer (keyed by tensor->data): `tiled` = the bytes currently are in the base tiling (un-interleaved)
    struct r4_state { bool tiled; };
    std::mutex mu; std::unordered_map<const void *, r4_state> r4;
};
struct shim_buft_ctx { int device; std::string name; };

static GGML_CALL const char *buf_get_name(ggml_backend_buffer_t b) { return ((shim_buft_ctx *)b->buft->context)->name.c_str(); }
static bool buffer_is_ours(ggml_backend_buffer_t b) { return b && b->iface.get_name == buf_get_name; }
static GGML_CALL void buf_free(ggml_backend_buffer_t b) { auto *c = (shim_buffer_ctx *)b->context; set_device(c->device); HIP_CHECK(hipFree(c->base)); delete c; }
static GGML_CALL void *buf_get_base(ggml_backend_buffer_t b) { return ((shim_buffer_ctx *)b->context)->base; }
static size_t padded_nbytes(const ggml_tensor *t) {
    size_t n = ggml_nbytes(t); const int64_t ne0 = t->ne[0];
    if (ggml_is_quantized(t->type) && ne0 % MATRIX_ROW_PADDING != 0) n += ggml_row_size(t->type, MATRIX_ROW_PADDING - ne0 % MATRIX_ROW_PADDING);
    return n;
}
static GGML_CALL void buf_init_tensor(ggml_backend_buffer_t b, ggml_tensor *t) {
    if (t->view_src != nullptr) return;
    if (ggml_is_quantized(t->type)) {   // zero the row padding (ggml-cuda.cu:621-639)
        const size_t orig = ggml_nbytes(t), padded = padded_nbytes(t);
        if (padded > orig) { auto *c = (shim_buffer_ctx *)b->context; set_device(c->device); HIP_CHECK(hipMemset((char *)t->data + orig, 0, padded - orig)); }
    }
}

// ---- _R4 tensors: the 4-row interleave exists so that one AVX load of activations feeds 4 rows; a wavefront amortises the activations over
// 64 lanes, so the MI355X-native tiling is the base one.  A complete upload is re-tiled at once; a tensor written piecewise
// (the loader's chunked async upload, llama-model-loader.cpp:1204-1240) is re-tiled at its first use in a graph (ensure_tiled).
static bool r4_candidate(const ggml_tensor *t) {
    if (!is_r4_type(t->type) || t->view_src != nullptr || !ggml_is_contiguous(t) || t->ne[1] % r4_rows(t->type)) return false;
    return !is_r4h_type(t->type) || t->ne[0] % cdna4_blck_size(r4_base_type(t->type)) == 0;
}
static void r4_retile(shim_buffer_ctx *c, const ggml_tensor *t, bool to_base) {        // in place through a temporary (upload-time cost only)
    if (t_capturing) throw capture_failed{-1};       // (synchronous copies / a device sync: not inside a stream capture -- the graph falls back to the eager walk, which re-tiles)
    set_device(c->device);
    const size_t nb = ggml_nbytes(t);
    const int64_t nrows = ggml_nrows(t);
    if (is_r4h_type(t->type)) {                          // host re-tiled types whose bytes are already on the device (piecewise upload, memset, download state): round trip through the host
        std::vector<uint8_t> a(nb), b(nb);
        HIP_CHECK(hipDeviceSynchronize());                // (an asynchronous upload of the pieces may still be in flight on a backend stream)
        HIP_CHECK(hipMemcpy(a.data(), t->data, nb, hipMemcpyDeviceToHost));
        check(cdna4_retile_r4_host(t->type, a.data(), b.data(), nrows, t->ne[0], to_base ? 1 : 0, 0), "_R4 host re-tiling");
        HIP_CHECK(hipMemcpy(t->data, b.data(), nb, hipMemcpyHostToDevice));
        return;
    }
    void *tmp = nullptr; HIP_CHECK(hipMalloc(&tmp, nb));
    cdna4_context *u = util_ctx(c->device);
    check(to_base ? cdna4_unrepack_r4(u, r4_base_type(t->type), t->data, nrows, t->ne[0], tmp, nullptr)
                  : cdna4_repack_r4(u, r4_base_type(t->type), t->data, nrows, t->ne[0], tmp, nullptr), "_R4 re-tiling");
    HIP_CHECK(hipMemcpy(t->data, tmp, nb, hipMemcpyDeviceToDevice)); HIP_CHECK(hipDeviceSynchronize()); HIP_CHECK(hipFree(tmp));
}
// make sure the tensor's bytes are interleaved (`want_tiled` false) or in the base tiling (true); returns the state it found
static void r4_set_state(ggml_backend_buffer_t b, const ggml_tensor *t, bool want_tiled) {
    auto *c = (shim_buffer_ctx *)b->context;
    std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->r4.find(t->data);
    const bool tiled = it != c->r4.end() && it->second.tiled;
    if (tiled != want_tiled) r4_retile(c, t, want_tiled);
    c->r4[t->data] = {want_tiled};
}
static bool r4_is_tiled(ggml_backend_buffer_t b, const ggml_tensor *t) {
    auto *c = (shim_buffer_ctx *)b->context; std::lock_guard<std::mutex> lock(c->mu);
    auto it = c->r4.find(t->data); return it != c->r4.end() && it->second.tiled;
}

// ---- small synchronous uploads ride the device's compute stream ------------------------------------------------------------------------------------------------------------
// The inputs of a decode step (the embedding row, the position, the mask, the output ids) reach the device through 3-4 ggml_backend_tensor_set calls per token: each a blocking
// hipMemcpy -- a blit launch and a host round trip with the GPU idle.  With a backend attached to the device, uploads of up to 64 KiB are copied into a pinned ring slot and
// queued on the backend's stream instead (the caller's bytes are consumed before the call returns, as the interface demands).  Everything that reads the tensor afterwards is
// either queued on that stream (graph launches, get_async, cpy_async with its event) or flushes first (get_tensor, cpy_tensor, memset, clear, a large upload).
// GGML_CDNA4_SYNC_SET=1: the blocking copies of rounds 1-3.
// (slot reuse: a slot's copy has run once the stream was synchronized after it was queued -- every token does that; sequence numbers instead of an event pair per upload)
// Only while ONE backend is attached to the device: with two (two llama contexts on one GPU, e.g. a draft and a target model) an upload queued on one backend's stream would not be
// ordered before a graph on the other's -- then every upload is the blocking copy again.
struct set_stage {
    hipStream_t stream = nullptr; char *host = nullptr; unsigned long long slot_seq[32] = {}, seq = 0, done_seq = 0; int next = 0, n_backends = 0; bool pending = false; long n_staged = 0;
    std::vector<hipStream_t> attached;        // every backend stream of the device: when the ring's owner leaves and ONE backend remains, the ring moves to it
};
static set_stage g_stage[GGML_CUDA_MAX_DEVICES]; static std::mutex g_stage_mus[GGML_CUDA_MAX_DEVICES];      // (one lock per device: an upload's stream synchronize must not serialise the other GPUs)
#define g_stage_mu g_stage_mus[device]
static void stage_ring_to(set_stage &g, hipStream_t st) {
    if (hipHostMalloc((void **)&g.host, (size_t)(64u << 10) * 32, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); g.host = nullptr; return; }
    g.stream = st; g.next = 0; g.pending = false; g.seq = g.done_seq = 0; for (auto &q : g.slot_seq) q = 0;
}
static constexpr size_t STAGE_SLOT = 64u << 10; static constexpr int STAGE_SLOTS = 32;
static void stage_attach(int device, hipStream_t st) {
    static const bool off = getenv("GGML_CDNA4_SYNC_SET") && atoi(getenv("GGML_CDNA4_SYNC_SET")) != 0;
    if (off || device < 0 || device >= GGML_CUDA_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_stage_mu); set_stage &g = g_stage[device];
    ++g.n_backends; g.attached.push_back(st);
    if (g.stream) {                                             // a second backend on the device: what is queued on the first one's stream completes now, nothing is queued from here on
        if (g.pending) { (void)hipStreamSynchronize(g.stream); g.pending = false; g.done_seq = g.seq; }
        return;
    }
    if (g.n_backends > 1) return;                               // (the ring's owner is gone, others remain: no ring until the device has a single backend again)
    stage_ring_to(g, st);
}
static void stage_detach(int device, hipStream_t st) {
    if (device < 0 || device >= GGML_CUDA_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_stage_mu); set_stage &g = g_stage[device];
    if (g.n_backends > 0) --g.n_backends;
    { auto it = std::find(g.attached.begin(), g.attached.end(), st); if (it != g.attached.end()) g.attached.erase(it); }
    if (g.stream == st && st) {
        (void)hipStreamSynchronize(st);
        if (g.host) (void)hipHostFree(g.host);
        { const int nb = g.n_backends; auto keep = g.attached; g = set_stage(); g.n_backends = nb; g.attached = keep; }
    }
    // one backend left on the device and no ring (its owner has just gone, or went earlier): the survivor's small uploads are queued again instead of blocking for the rest of the process
    if (!g.stream && g.n_backends == 1 && g.attached.size() == 1 && g.attached[0]) stage_ring_to(g, g.attached[0]);
}
static bool stage_upload(int device, void *dst, const void *src, size_t size) {
    if (size == 0 || size > STAGE_SLOT || device < 0 || device >= GGML_CUDA_MAX_DEVICES) return false;
    std::lock_guard<std::mutex> lock(g_stage_mu); set_stage &g = g_stage[device];
    if (!g.stream || !g.host || g.n_backends != 1) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(g.stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    const int sl = g.next; g.next = (g.next + 1) % STAGE_SLOTS;
    if (g.slot_seq[sl] > g.done_seq) { HIP_CHECK(hipStreamSynchronize(g.stream)); g.done_seq = g.seq; }      // (more than 32 uploads without a synchronize in between)
    memcpy(g.host + sl * STAGE_SLOT, src, size);
    HIP_CHECK(hipMemcpyAsync(dst, g.host + sl * STAGE_SLOT, size, hipMemcpyHostToDevice, g.stream));
    g.slot_seq[sl] = ++g.seq;
    g.pending = true; ++g.n_staged;
    return true;
}
static void stage_flush(int device) {       // before anything that touches device memory outside the backend's stream
    if (device < 0 || device >= GGML_CUDA_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_stage_mu); set_stage &g = g_stage[device];
    if (g.stream && g.pending) { HIP_CHECK(hipStreamSynchronize(g.stream)); g.pending = false; g.done_seq = g.seq; }
}
static unsigned long long stage_seq(int device) {          // uploads queued so far (read BEFORE a synchronize: what that synchronize is known to have completed)
    if (device < 0 || device >= GGML_CUDA_MAX_DEVICES) return 0;
    std::lock_guard<std::mutex> lock(g_stage_mu); return g_stage[device].seq;
}
static void stage_synced(int device, hipStream_t st, unsigned long long seq_before) {      // the backend has just synchronized its stream
    if (device < 0 || device >= GGML_CUDA_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_stage_mu); set_stage &g = g_stage[device];
    if (g.stream == st) { if (seq_before > g.done_seq) g.done_seq = seq_before; if (g.done_seq == g.seq) g.pending = false; }
}

static GGML_CALL void buf_memset_tensor(ggml_backend_buffer_t b, ggml_tensor *t, uint8_t v, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; set_device(c->device); stage_flush(c->device);
    if (r4_candidate(t)) r4_set_state(b, t, false);
    HIP_CHECK(hipMemset((char *)t->data + off, v, size)); HIP_CHECK(hipDeviceSynchronize());
}
// first upload of weights of a type to a device: load the prompt kernels of that type now, not inside the first prompt pass (cdna4_preload_type)
static void preload_kernels_for(int device, const ggml_tensor *t) {
    static std::mutex mu; static std::set<std::pair<int, int>> seen;
    if (!ggml_is_quantized(t->type) || ggml_n_dims(t) < 2 || !cdna4_type_supported((int)t->type)) return;
    { std::lock_guard<std::mutex> lock(mu); if (!seen.insert({device, (int)t->type}).second) return; }
    static const bool off = getenv("GGML_CDNA4_NO_PRELOAD") != nullptr;
    if (!off && cdna4_preload_type((int)t->type) != CDNA4_OK) (void)hipGetLastError();
}
// A byte range of a host re-tiled tensor touches whole interleaved row groups only: group g occupies the same bytes [g * G, (g + 1) * G) in the file layout and in the base
// tiling (G = rows per group x row size), and groups are re-tiled independently.  Partial reads and partial overwrites of a re-tiled tensor therefore move and re-tile the covered
// groups, not the tensor (a chunked reload of a multi-GB expert tensor was O(chunks x tensor) traffic and 2 x the tensor in host memory per call).
struct r4h_span { size_t g_bytes, b0, b1; int64_t rows; };
static r4h_span r4h_cover(const ggml_tensor *t, size_t off, size_t size) {
    const int gr = r4_rows(t->type); const size_t nb = ggml_nbytes(t), g_bytes = nb / (size_t)(ggml_nrows(t) / gr);
    const size_t g0 = off / g_bytes, g1 = (off + size + g_bytes - 1) / g_bytes;
    return {g_bytes, g0 * g_bytes, std::min(nb, g1 * g_bytes), (int64_t)(g1 - g0) * gr};
}
static GGML_CALL void buf_set_tensor(ggml_backend_buffer_t b, ggml_tensor *t, const void *data, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; set_device(c->device);
    preload_kernels_for(c->device, t);
    const bool r4 = r4_candidate(t);
    if (r4 && is_r4h_type(t->type) && size > 0 && !(off == 0 && size == ggml_nbytes(t)) && r4_is_tiled(b, t)) {      // partial overwrite of a re-tiled tensor: patch the covered groups
        stage_flush(c->device);
        const r4h_span sp = r4h_cover(t, off, size); std::vector<uint8_t> tiled(sp.b1 - sp.b0), file(sp.b1 - sp.b0);
        HIP_CHECK(hipDeviceSynchronize());
        HIP_CHECK(hipMemcpy(tiled.data(), (const char *)t->data + sp.b0, tiled.size(), hipMemcpyDeviceToHost));
        check(cdna4_retile_r4_host(t->type, tiled.data(), file.data(), sp.rows, t->ne[0], 0, 1), "_R4 host re-interleave");
        memcpy(file.data() + (off - sp.b0), data, size);
        check(cdna4_retile_r4_host(t->type, file.data(), tiled.data(), sp.rows, t->ne[0], 1, 1), "_R4 host re-tiling");
        HIP_CHECK(hipMemcpy((char *)t->data + sp.b0, tiled.data(), tiled.size(), hipMemcpyHostToDevice));
        return;
    }
    if (r4 && is_r4h_type(t->type) && off == 0 && size == ggml_nbytes(t)) {       // a complete upload of a host re-tiled type: file bytes -> base tiling -> device, one copy
        std::vector<uint8_t> tiled(size);
        check(cdna4_retile_r4_host(t->type, data, tiled.data(), ggml_nrows(t), t->ne[0], 1, 0), "_R4 host re-tiling");
        HIP_CHECK(hipMemcpy(t->data, tiled.data(), size, hipMemcpyHostToDevice));
        std::lock_guard<std::mutex> lock(c->mu); c->r4[t->data] = {true};
        return;
    }
    if (r4) r4_set_state(b, t, false);               // (a partial write into an already re-tiled tensor: back to the file layout first)
    if (!r4 && stage_upload(c->device, (char *)t->data + off, data, size)) return;
    stage_flush(c->device);
    HIP_CHECK(hipMemcpy((char *)t->data + off, data, size, hipMemcpyHostToDevice));
    if (r4 && off == 0 && size == ggml_nbytes(t)) r4_set_state(b, t, true);
}
static GGML_CALL void buf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor *t, void *data, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; set_device(c->device); stage_flush(c->device);
    if (r4_candidate(t) && r4_is_tiled(b, t)) {      // hand back the file (interleaved) layout: exact inverse of the upload re-tiling
        const size_t nb = ggml_nbytes(t);
        if (is_r4h_type(t->type)) {                   // (only the interleaved row groups the range touches)
            if (size == 0) return;
            const r4h_span sp = r4h_cover(t, off, size); std::vector<uint8_t> tiled(sp.b1 - sp.b0), file(sp.b1 - sp.b0);
            HIP_CHECK(hipMemcpy(tiled.data(), (const char *)t->data + sp.b0, tiled.size(), hipMemcpyDeviceToHost));
            check(cdna4_retile_r4_host(t->type, tiled.data(), file.data(), sp.rows, t->ne[0], 0, 0), "_R4 host re-interleave");
            memcpy(data, file.data() + (off - sp.b0), size);
            return;
        }
        void *tmp = nullptr; HIP_CHECK(hipMalloc(&tmp, nb));
        check(cdna4_repack_r4(util_ctx(c->device), r4_base_type(t->type), t->data, ggml_nrows(t), t->ne[0], tmp, nullptr), "_R4 re-interleave");
        HIP_CHECK(hipMemcpy(data, (const char *)tmp + off, size, hipMemcpyDeviceToHost)); HIP_CHECK(hipFree(tmp));
        return;
    }
    HIP_CHECK(hipMemcpy(data, (const char *)t->data + off, size, hipMemcpyDeviceToHost));
}
static GGML_CALL bool buf_cpy_tensor(ggml_backend_buffer_t b, const ggml_tensor *src, ggml_tensor *dst) {
    if (!buffer_is_ours(src->buffer)) return false;
    stage_flush(((shim_buffer_ctx *)src->buffer->context)->device); stage_flush(((shim_buffer_ctx *)b->context)->device);
    bool src_tiled = false;
    if (r4_candidate(src)) src_tiled = r4_is_tiled(src->buffer, src);
    HIP_CHECK(hipMemcpy(dst->data, src->data, ggml_nbytes(src), hipMemcpyDeviceToDevice));      // same or peer device
    if (r4_candidate(dst)) { auto *c = (shim_buffer_ctx *)b->context; std::lock_guard<std::mutex> lock(c->mu); c->r4[dst->data] = {src_tiled && dst->type == src->type}; }
    return true;
}
static GGML_CALL void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    auto *c = (shim_buffer_ctx *)b->context; set_device(c->device); stage_flush(c->device);
    { std::lock_guard<std::mutex> lock(c->mu); c->r4.clear(); }
    HIP_CHECK(hipMemset(c->base, v, b->size)); HIP_CHECK(hipDeviceSynchronize());
}
static GGML_CALL void buf_reset(ggml_backend_buffer_t b) { auto *c = (shim_buffer_ctx *)b->context; std::lock_guard<std::mutex> lock(c->mu); c->r4.clear(); }
static const ggml_backend_buffer_i k_buffer_iface = { buf_get_name, buf_free, buf_get_base, buf_init_tensor, buf_memset_tensor, buf_set_tensor, buf_get_tensor, buf_cpy_tensor, buf_clear, buf_reset };

static GGML_CALL const char *buft_get_name(ggml_backend_buffer_type_t t) { return ((shim_buft_ctx *)t->context)->name.c_str(); }
static ggml_backend_buffer_t device_buffer_alloc(ggml_backend_buffer_type_t t, int device, size_t size) {
    set_device(device);
    size = size ? size : 1; void *p = nullptr;
    if (hipMalloc(&p, size) != hipSuccess) { (void)hipGetLastError(); shim_log(GGML_LOG_LEVEL_ERROR, "ggml-hip-cdna4: allocating %.2f MiB on device %d failed\n", size / 1048576.0, device); return nullptr; }
    auto *c = new shim_buffer_ctx(); c->device = device; c->base = p;
    return ggml_backend_buffer_init(t, k_buffer_iface, c, size);
}
static GGML_CALL ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) { return device_buffer_alloc(t, ((shim_buft_ctx *)t->context)->device, size); }
static GGML_CALL size_t buft_alignment(ggml_backend_buffer_type_t) { return 128; }
static GGML_CALL size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor *t) { return padded_nbytes(t); }     // ggml-cuda.cu:754-767
static GGML_CALL bool buft_is_host(ggml_backend_buffer_type_t) { return false; }

extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_buffer_type(int device) {
    static std::mutex mu; static ggml_backend_buffer_type types[SHIM_MAX_DEVICES]; static bool init = false;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= device_count() || device >= SHIM_MAX_DEVICES) return nullptr;
    if (!init) {
        for (int i = 0; i < SHIM_MAX_DEVICES; ++i) {
            types[i].iface = { buft_get_name, buft_alloc, buft_alignment, nullptr, buft_alloc_size, buft_is_host };
            types[i].context = new shim_buft_ctx{i, std::string(GGML_CUDA_NAME) + std::to_string(i)};
        }
        init = true;
    }
    return &types[device];
}

// ---------------------------------------------------------------------------------------------- pinned host buffer (ggml-cuda.cu host buffer type)
static GGML_CALL const char *host_buft_name(ggml_backend_buffer_type_t) { return GGML_CUDA_NAME "_Host"; }
static GGML_CALL void host_buf_free(ggml_backend_buffer_t b) { HIP_CHECK(hipHostFree(b->context)); }
static GGML_CALL ggml_backend_buffer_t host_buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    void *p = nullptr;
    if (real_device_count() <= 0 || hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size); }
    ggml_backend_buffer_t b = ggml_backend_cpu_buffer_from_ptr(p, size);
    b->buft = t; b->iface.free_buffer = host_buf_free;
    return b;
}
extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_host_buffer_type(void) {
    static ggml_backend_buffer_type t = { { host_buft_name, host_buft_alloc, ggml_backend_cpu_buffer_type()->iface.get_alignment, nullptr,
                                            ggml_backend_cpu_buffer_type()->iface.get_alloc_size, ggml_backend_cpu_buffer_type()->iface.is_host }, nullptr };
    return &t;
}

// ---------------------------------------------------------------------------------------------- split buffer type (-sm graph)
// A tensor placed in the split buffer type carries tensor->extra -> ggml_split_tensor_t {n_device, split_dim, tensor, splits[]} (ggml.h:3333-3338).
// libllama's graph builder works on the per-device `splits[i]` directly (they are ordinary tensors of device i); this buffer type's job is to
// give every split its device memory (init_tensor) and to cut the file bytes of the whole tensor into the splits at upload (set_tensor):
//   split_dim -1 replicate | 0 along ne[0] (K: whole quant blocks, the row-parallel wo / ffn_down) | 1 along ne[1] (rows: q,k,v,up,gate)
//   | 2 along ne[2] (experts);  an optional list of explicit (first, count) ranges per device rides in tensor->op_params (a pointer).
// Re-stated from the behaviour of ggml_backend_cuda_split_buffer_{init,set,get}_tensor (ggml-cuda.cu:852-1402), the special case of merged
// ffn_gate_up_exps views (:890-968) included (split_buf_set_merged_view).
static GGML_CALL const char *split_buf_name(ggml_backend_buffer_t) { return GGML_CUDA_NAME "_Split"; }
struct split_buffer_ctx { std::vector<ggml_backend_buffer_t> owned; };
static GGML_CALL void split_buf_free(ggml_backend_buffer_t b) { auto *c = (split_buffer_ctx *)b->context; for (auto *o : c->owned) ggml_backend_buffer_free(o); delete c; }
static GGML_CALL void *split_buf_base(ggml_backend_buffer_t) { return (void *)0x1000; }       // never dereferenced: the data lives in the splits (ggml-cuda.cu:845-850)
static GGML_CALL void split_buf_init_tensor(ggml_backend_buffer_t b, ggml_tensor *t) {
    if (!t->extra) return;
    auto *ex = (ggml_split_tensor_t *)t->extra; auto *c = (split_buffer_ctx *)b->context;
    GGML_ASSERT(ex->n_device <= device_count());
    for (int i = 0; i < ex->n_device; ++i) {
        ggml_tensor *s = ex->splits[i]; if (!s) continue;
        ggml_backend_buffer_type_t dt = ggml_backend_cuda_buffer_type(i);
        const size_t padded = padded_nbytes(s), size = ggml_nbytes(s);
        ggml_backend_buffer_t sb = device_buffer_alloc(dt, i, padded);
        if (!sb) GGML_ABORT("ggml-hip-cdna4: split allocation failed");
        ggml_backend_buffer_set_usage(sb, GGML_BACKEND_BUFFER_USAGE_WEIGHTS);
        s->data = ggml_backend_buffer_get_base(sb); s->buffer = sb;
        if (padded > size) { set_device(i); HIP_CHECK(hipMemset((char *)s->data + size, 0, padded - size)); }
        c->owned.push_back(sb);
    }
}
typedef std::vector<std::vector<std::pair<int, int>>> split_ranges_t;
static const split_ranges_t *split_ranges_of(const ggml_tensor *t) { void *p = nullptr; memcpy(&p, t->op_params, sizeof(p)); return (const split_ranges_t *)p; }
// rows of an interleaved type travel in groups (ggml-cuda.cu:969-1000 k_map): the _R4 types on this path interleave 4 rows
static int rows_interleaved(enum ggml_type t) { return is_r4_type(t) ? r4_rows(t) : 1; }

// gather (upload) or scatter (download) between the whole tensor's host bytes and ONE split's staging image
template <bool UPLOAD>
static void split_xfer(const ggml_tensor *t, const ggml_split_tensor_t *ex, int idev, char *whole, std::vector<char> &stage, int64_t &acc) {
    const ggml_tensor *s = ex->splits[idev];
    const auto tt = ggml_internal_get_type_traits(t->type);
    const size_t nb = ggml_nbytes(s); if (stage.size() < nb) stage.resize(nb);
    const split_ranges_t *ranges = split_ranges_of(t);
    auto mv = [](char *split_side, char *whole_side, size_t n) { if (UPLOAD) memcpy(split_side, whole_side, n); else memcpy(whole_side, split_side, n); };
    if (ex->split_dim < 0) { GGML_ASSERT(ggml_is_contiguous(t) && ggml_nbytes(t) == nb); mv(stage.data(), whole, nb); return; }
    if (ex->split_dim == 0) {       // K split: a byte-column range of every (group of interleaved) row(s)
        // row-scaled types (IQ4_KS, IQ5_KS, IQ2_KS, IQ3_KS, IQ4_KSS, IQ2_KL) keep `row_meta_size` bytes (the row scale) in front of a row's blocks: every split
        // gets a copy of them in front of its block range (ggml-cuda.cu:1073-1086); only the explicit-ranges form has no such types (the builder never asks for it)
        GGML_ASSERT(ggml_is_contiguous(t) && (tt.row_meta_size == 0 || !split_ranges_of(t)) && "explicit K ranges of a type with per-row meta data");
        const int il = rows_interleaved(t->type); const int64_t nrows = ggml_nrows(t);
        const size_t srow = ggml_row_size(s->type, s->ne[0]), wrow = t->nb[1];
        GGML_ASSERT(ggml_nrows(s) == nrows && s->ne[0] % tt.blck_size == 0 && nrows % il == 0);
        if (ranges) {
            GGML_ASSERT(il == 1 && t->ne[2] * t->ne[3] == 1);
            for (int64_t r = 0; r < nrows; ++r) { char *d = stage.data() + r * srow;
                for (auto &p : (*ranges)[idev]) { GGML_ASSERT(p.first % tt.blck_size == 0 && p.second % tt.blck_size == 0);
                    const size_t n = (size_t)(p.second / tt.blck_size) * tt.type_size; mv(d, whole + r * wrow + (size_t)(p.first / tt.blck_size) * tt.type_size, n); d += n; } }
        } else {
            const size_t meta = (size_t)il * tt.row_meta_size;                             // a group of `il` interleaved rows: [il x meta][blocks ...]
            const size_t off = meta + (size_t)il * (acc / tt.blck_size) * tt.type_size;      // byte offset of this split's blocks inside the group
            for (int64_t g = 0; g < nrows / il; ++g) {
                if (meta) mv(stage.data() + g * il * srow, whole + g * il * wrow, meta);     // (download: every split writes the same bytes back)
                mv(stage.data() + g * il * srow + meta, whole + g * il * wrow + off, il * srow - meta);
            }
            acc += s->ne[0];
        }
        return;
    }
    if (ex->split_dim == 1) {       // row split: contiguous row ranges of every ne[2] slice
        const size_t row = ggml_row_size(t->type, t->ne[0]);
        for (int64_t i2 = 0; i2 < s->ne[2] * s->ne[3]; ++i2) {
            char *d = stage.data() + i2 * s->ne[1] * row;
            if (ranges) for (auto &p : (*ranges)[idev]) { mv(d, whole + i2 * t->nb[2] + (size_t)p.first * t->nb[1], (size_t)p.second * t->nb[1]); d += (size_t)p.second * t->nb[1]; }
            else mv(d, whole + i2 * t->nb[2] + (size_t)acc * t->nb[1], (size_t)s->ne[1] * row);
        }
        if (!ranges) acc += s->ne[1];
        return;
    }
    if (ex->split_dim == 2) { mv(stage.data(), whole + (size_t)acc * t->nb[2], nb); acc += s->ne[2]; return; }      // experts
    GGML_ABORT("ggml-hip-cdna4: split_dim not implemented");
}
// `-muge` (merge_up_gate_exps, src/llama-load-tensors.cpp:4403-4470): the loader creates ONE split tensor blk.N.ffn_gate_up_exps.weight [K, 2 n_ff, n_expert] (rows split over the
// devices with explicit ranges: per device {a range of the gate half, a range of the up half}) and loads the file's ffn_gate_exps / ffn_up_exps into VIEWS of it: set_tensor arrives
// for the view (no extra of its own) with the whole gate (or up) tensor.  Every device's split holds [its gate rows ; its up rows] per expert: the view's rows go to the first / second
// half of each expert slice, device after device in row order.  Same for the bias pair ([2 n_ff, n_expert] f32, split along dim 0).  Behaviour of ggml-cuda.cu:890-968.
static bool split_buf_set_merged_view(ggml_tensor *t, const void *data, size_t off, size_t size) {
    const ggml_tensor *vs = t->view_src; auto *ex = (ggml_split_tensor_t *)vs->extra; const split_ranges_t *ranges = split_ranges_of(vs);
    const bool is_w = strstr(vs->name, "ffn_gate_up_exps.weight") != nullptr, is_b = strstr(vs->name, "ffn_gate_up_exps.bias") != nullptr;
    const bool gate = strstr(t->name, is_w ? "ffn_gate_exps.weight" : "ffn_gate_exps.bias") != nullptr, up = strstr(t->name, is_w ? "ffn_up_exps.weight" : "ffn_up_exps.bias") != nullptr;
    if (!ranges || !(is_w || is_b) || gate == up) return false;
    GGML_ASSERT(off == 0 && size == ggml_nbytes(t) && ex->split_dim == (is_w ? 1 : 0) && (int)ranges->size() >= ex->n_device);
    const int part = gate ? 0 : 1; int64_t acc = 0;                          // rows (weights) / columns (bias) of the view handed out so far
    for (int i = 0; i < ex->n_device; ++i) {
        ggml_tensor *s = ex->splits[i]; const auto &r = (*ranges)[i];
        GGML_ASSERT((s != nullptr) == !r.empty());
        if (!s) continue;
        GGML_ASSERT(r.size() == 2);
        const int64_t n = r[part].second;
        if (is_w) {
            GGML_ASSERT(s->ne[1] % 2 == 0 && n == s->ne[1] / 2 && s->ne[0] == t->ne[0] && acc + n <= t->ne[1] && n % rows_interleaved(t->type) == 0);
            const size_t half = (size_t)(s->ne[1] / 2) * s->nb[1];
            for (int64_t e = 0; e < s->ne[2] * s->ne[3]; ++e)                 // (a partial write into the split's device buffer: interleaved slices are re-tiled at their first use)
                ggml_backend_tensor_set(s, (const char *)data + e * t->nb[2] + acc * t->nb[1], e * s->nb[2] + part * half, (size_t)n * t->nb[1]);
        } else {
            GGML_ASSERT(s->ne[0] % 2 == 0 && n == s->ne[0] / 2 && acc + n <= t->ne[0]);
            const size_t half = (size_t)(s->ne[0] / 2) * s->nb[0];
            for (int64_t e = 0; e < ggml_nrows(s); ++e)
                ggml_backend_tensor_set(s, (const char *)data + e * t->nb[1] + acc * t->nb[0], e * s->nb[1] + part * half, (size_t)n * t->nb[0]);
        }
        acc += n;
    }
    return true;
}
static GGML_CALL void split_buf_set_tensor(ggml_backend_buffer_t, ggml_tensor *t, const void *data, size_t off, size_t size) {
    if (!t->extra) {
        if (t->view_src && t->view_src->extra && !split_buf_set_merged_view(t, data, off, size)) GGML_ABORT("ggml-hip-cdna4: set_tensor on a view of split tensor %s (%s): only the merged ffn_gate_up_exps views are known", t->view_src->name, t->name);
        return;
    }
    GGML_ASSERT(off == 0 && size == ggml_nbytes(t));            // split tensors are always set in their entirety (ggml-cuda.cu:1003-1005)
    auto *ex = (ggml_split_tensor_t *)t->extra; std::vector<char> stage; int64_t acc = 0;
    for (int i = 0; i < ex->n_device; ++i) {
        ggml_tensor *s = ex->splits[i]; if (!s) continue;
        split_xfer<true>(t, ex, i, (char *)data, stage, acc);
        ggml_backend_tensor_set(s, stage.data(), 0, ggml_nbytes(s));          // the split's own device buffer (re-tiles _R4 slices like any upload)
    }
}
static GGML_CALL void split_buf_get_tensor(ggml_backend_buffer_t, const ggml_tensor *t, void *data, size_t off, size_t size) {
    if (!t->extra) return;
    GGML_ASSERT(off == 0 && size == ggml_nbytes(t));
    auto *ex = (ggml_split_tensor_t *)t->extra; std::vector<char> stage; int64_t acc = 0;
    for (int i = 0; i < ex->n_device; ++i) {
        ggml_tensor *s = ex->splits[i]; if (!s) continue;
        const size_t nb = ggml_nbytes(s); if (stage.size() < nb) stage.resize(nb);
        ggml_backend_tensor_get(s, stage.data(), 0, nb);
        split_xfer<false>(t, ex, i, (char *)data, stage, acc);
        if (ex->split_dim < 0) return;                                        // replicated: the first copy is the tensor
    }
}
static GGML_CALL bool split_buf_cpy_tensor(ggml_backend_buffer_t, const ggml_tensor *, ggml_tensor *) { return false; }
static GGML_CALL void split_buf_clear(ggml_backend_buffer_t, uint8_t) {}
static GGML_CALL void split_buf_memset(ggml_backend_buffer_t, ggml_tensor *, uint8_t, size_t, size_t) {}
static const ggml_backend_buffer_i k_split_iface = { split_buf_name, split_buf_free, split_buf_base, split_buf_init_tensor, split_buf_memset, split_buf_set_tensor, split_buf_get_tensor,
                                                     split_buf_cpy_tensor, split_buf_clear, nullptr };
static GGML_CALL const char *split_buft_name(ggml_backend_buffer_type_t) { return GGML_CUDA_NAME "_Split"; }
static GGML_CALL ggml_backend_buffer_t split_buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    // the tensors' bytes live in per-split device buffers created by init_tensor; this object only owns them (ggml-cuda.cu:1404-1420)
    return ggml_backend_buffer_init(t, k_split_iface, new split_buffer_ctx(), size);
}
static GGML_CALL size_t split_buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor *t) { return t->extra ? 0 : ggml_nbytes(t); }
extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_split_buffer_type(const float *) {
    static ggml_backend_buffer_type t = { { split_buft_name, split_buft_alloc, buft_alignment, nullptr, split_buft_alloc_size, buft_is_host }, nullptr };
    return &t;
}

// ---------------------------------------------------------------------------------------------- backend
struct shim_params {            // the reference's "k=v,..." backend parameter string (ggml-cuda.cu:5299-5389)
    int fusion = 1; int offload_batch_size = 32; int offload_batch_size_per_byte = -1; int mmq_id_thresh = 32; float fa_offset = 0.6931f;
    bool use_graphs = true; bool enable_p2p = true;
};
static shim_params parse_params(const char *s) {
    shim_params p; if (!s || !*s) return p;
    std::stringstream ss(s); std::string kv;
    while (std::getline(ss, kv, ',')) {
        const size_t eq = kv.find('='); bool good = false;
        if (eq != std::string::npos) {
            const std::string k = kv.substr(0, eq), v = kv.substr(eq + 1); char *end = nullptr;
            const double d = strtod(v.c_str(), &end); good = end && end != v.c_str();
            if (!good) {}
            else if (k == "fusion") p.fusion = (int)d;
            else if (k == "offload-batch-size") p.offload_batch_size = (int)d;
            else if (k == "offload-batch-size-per-byte") p.offload_batch_size_per_byte = (int)d;
            else if (k == "mmq-id-size") p.mmq_id_thresh = (int)d;
            else if (k == "enable-p2p") p.enable_p2p = d != 0;
            else if (k == "graphs") p.use_graphs = d != 0;
            else if (k == "fa-offset") { if (d >= 0 && d <= 3) p.fa_offset = (float)d; else shim_log(GGML_LOG_LEVEL_WARN, "ggml-hip-cdna4: bad value for fa-offset (%g): must be in [0...3]\n", d); }
            else good = false;
        }
        if (!good) shim_log(GGML_LOG_LEVEL_WARN, "ggml-hip-cdna4: invalid parameter %s -> ignored\n", kv.c_str());
    }
    return p;
}

// HIP graphs (ggml-cuda.cu:4408-4760): a compute graph seen twice in a row with identical nodes is captured and replayed afterwards.
// The KV-cache write position moves every token, so the destinations of the cache-write nodes (CPY) are not part of the key: the captured kernels read
// them from a device-side slot table that one captured H2D copy refreshes from a pinned host table before the graph's first kernel (ggml-cuda.cu:4480-4560
// patches the copy kernels' parameters in the instantiated graph for the same purpose).
struct graph_key {
    // everything a captured launch bakes in (the reference compares the same set, ggml-cuda.cu:4524-4558): addresses, types, the full ne / nb of the node and of
    // its sources, every op parameter (ROPE reads op_params up to [14])
    struct node { int op, type; const void *data, *src[6]; int64_t ne[4], nb[4]; int src_type[6]; int64_t src_ne[6][4], src_nb[6][4]; int32_t params[GGML_MAX_OP_PARAMS / sizeof(int32_t)]; };
    std::vector<node> nodes;
    bool operator==(const graph_key &o) const { return nodes.size() == o.nodes.size() && (nodes.empty() || memcmp(nodes.data(), o.nodes.data(), nodes.size() * sizeof(node)) == 0); }
};
struct cached_graph { graph_key key; hipGraphExec_t exec = nullptr; int seen = 0; bool failed = false; long ws_epoch = -1; };

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
struct shim_context {
    int device; cdna4_context *ctx; hipStream_t stream; std::string name; hipEvent_t ev = nullptr, ev2 = nullptr;
    shim_params params; const void *model = nullptr;
    std::vector<cached_graph> graphs; int last_graph = -1;      // (index of the entry the previous call used)
    // cache-write destinations of the graph being run: slot i = dst address of the i-th CPY node (node order)
    static constexpr int MAX_SLOTS = 1024;
    void **slots_host = nullptr, **slots_dev = nullptr; hipEvent_t slots_ev = nullptr; bool slots_busy = false;
    bool capturing = false; int slot_next = 0;
    const void *rope_pos = nullptr; int32_t rope_params[16] = {0}; int rope_fills = 0;      // the (cos, sin) cache of this graph's rope nodes
    long n_eager = 0, n_replayed = 0, n_captured = 0, n_capture_failed = 0, n_small = 0, n_fused_attn = 0;      // GGML_CDNA4_STATS
    void *x32 = nullptr; size_t x32_bytes = 0;         // f32 copy of an f16 src1 of a quantised MUL_MAT
    long n_fuse[8] = {0};       // fused launches issued or captured, by GGML_CDNA4_FUSION_OFF bit (0: ADD+RMS_NORM ... 7: attention + attn_output)
    double t_compute = 0, t_sync = 0, t_set = 0, t_get = 0; long n_sync = 0, n_set = 0, n_get = 0; size_t b_set = 0, b_get = 0;
};
static void drop_graphs(shim_context *c);      // (defined with the graph cache below)

// device -> most recent backend of this process: the REDUCE node runs on ONE backend and orders every peer's stream around its launch
// (the reference keeps the same kind of map, model -> ctx[device]: ggml-cuda/common.cuh:765, reduce.cu:140-145)
static shim_context *g_shims[GGML_CUDA_MAX_DEVICES] = {nullptr};
static std::mutex g_shims_mu;

static ggml_guid_t shim_guid() { static ggml_guid g = {0xc4, 0xd1, 0x4a, 0x04, 0x95, 0x0f, 0x11, 0xee, 0x9a, 0x33, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30}; return &g; }

static bool weight_ok(const ggml_tensor *w) {
    if (!cdna4_type_supported(w->type) && !is_r4h_type(w->type)) return false;
    // _R4 weights are served from their re-tiled bytes, which only exist for tensors that live in one of our device buffers
    if (is_r4_type(w->type)) return buffer_is_ours(w->buffer) && r4_candidate(w);
    return true;
}
static bool mm_types_ok(const ggml_tensor *w, const ggml_tensor *x, const ggml_tensor *dst) {
    return weight_ok(w) && x->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32 &&
           w->nb[0] == ggml_type_size(w->type) && x->nb[0] == sizeof(float) && dst->nb[0] == sizeof(float) &&
           w->ne[0] % 64 == 0 && !ggml_is_transposed(w) && !ggml_is_transposed(x);
}
// BitNet weights (IQ1_BN / IQ2_BN): plain MUL_MAT only on the device (csrc/gemv_bitnet.hip); the fused / MoE / GET_ROWS forms stay on the CPU backend
static bool is_bitnet(const ggml_tensor *w) { return w->type == GGML_TYPE_IQ1_BN || w->type == GGML_TYPE_IQ2_BN || w->type == GGML_TYPE_IQ2_BN_R4; }
static bool up_gate_unary_ok(int u) { return u == GGML_UNARY_OP_SILU || u == GGML_UNARY_OP_GELU || u == GGML_UNARY_OP_RELU || u == GGML_UNARY_OP_SWIGLU_OAI; }
// per-expert bias [M, n_expert] f32 (ggml_moe_up_gate_ext, ggml.c:8066-8080)
static bool bias_ok(const ggml_tensor *b, const ggml_tensor *w) { return !b || (b->type == GGML_TYPE_F32 && b->nb[0] == sizeof(float) && b->ne[0] == w->ne[1]); }
// developer knob: GGML_CDNA4_DISABLE_OPS="MUL_MAT,FUSED_UP_GATE,..." leaves those ops to the CPU backend (bisecting a parity failure)
static bool op_disabled(const ggml_tensor *op) {
    static const char *e = getenv("GGML_CDNA4_DISABLE_OPS");
    if (!e || !*e) return false;
    const std::string list = std::string(",") + e + ",", name = std::string(",") + ggml_op_name(op->op) + ",";
    return list.find(name) != std::string::npos;
}
static bool is_f32_f16(ggml_type t) { return t == GGML_TYPE_F32 || t == GGML_TYPE_F16; }
static bool supports_op_impl(const ggml_tensor *op);
static GGML_CALL bool be_supports_op(ggml_backend_t, const ggml_tensor *op) {
    const bool ok = supports_op_impl(op);
    static const bool log_unsupported = getenv("GGML_CDNA4_LOG_UNSUPPORTED") != nullptr;       // developer knob: which ops of a graph stay on the CPU backend
    if (!ok && log_unsupported) {
        static std::mutex mu; static std::vector<std::string> seen; std::lock_guard<std::mutex> lock(mu);
        char sig[512]; int n = snprintf(sig, sizeof(sig), "%s %s [%ld,%ld,%ld,%ld]", ggml_op_name(op->op), ggml_type_name(op->type), (long)op->ne[0], (long)op->ne[1], (long)op->ne[2], (long)op->ne[3]);
        for (int i = 0; i < 4 && op->src[i]; ++i) n += snprintf(sig + n, sizeof(sig) - n, " | src%d %s [%ld,%ld,%ld,%ld]%s%s", i, ggml_type_name(op->src[i]->type), (long)op->src[i]->ne[0], (long)op->src[i]->ne[1],
                                                         (long)op->src[i]->ne[2], (long)op->src[i]->ne[3], ggml_is_contiguous(op->src[i]) ? "" : " nc",
                                                         op->src[i]->buffer && op->src[i]->buffer->iface.get_name == split_buf_name ? " SPLIT-PARENT" : "");
        n += snprintf(sig + n, sizeof(sig) - n, " params %d %d %d %d", op->op_params[0], op->op_params[1], op->op_params[2], op->op_params[3]);
        if (std::find(seen.begin(), seen.end(), sig) == seen.end()) { seen.push_back(sig); fprintf(stderr, "cdna4-unsupported: %s\n", sig); }
    }
    return ok;
}
static bool supports_op_impl(const ggml_tensor *op) {
    if (op_disabled(op)) return false;
    // A tensor that lives in the split buffer type has no bytes of its own (its per-device `splits` do; libllama's -sm graph builder works on
    // those).  When a builder path uses such a parent directly (e.g. attention without a split KV cache), the op is declined: the scheduler
    // then runs it on the CPU backend with a copy gathered by the split buffer's get_tensor.
    for (int i = 0; i < GGML_MAX_SRC; ++i) if (op->src[i] && op->src[i]->buffer && op->src[i]->buffer->iface.get_name == split_buf_name) return false;
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return true;
        case GGML_OP_MUL_MAT: {
            const ggml_tensor *w = op->src[0], *x = op->src[1];
            if ((w->type == GGML_TYPE_F32 || w->type == GGML_TYPE_F16) && w->op == GGML_OP_NONE)        // small dense weights: the MoE router (ffn_gate_inp)
                return x->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && w->ne[1] <= 1024 && w->ne[2] == 1 && w->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1;
            // f16 activations with quantised weights (the CUDA backend's supports_op tolerates them, ggml-cuda.cu:4844-4847; the CPU path asserts f32, ggml.c:18162): converted to
            // f32 in a scratch buffer, then the f32 path -- the results are those of the f32 path on the same (f16-representable) values
            if (x->type == GGML_TYPE_F16 && ggml_is_quantized(w->type) && ggml_is_contiguous(x) && x->nb[0] == sizeof(ggml_fp16_t)) {
                return weight_ok(w) && op->type == GGML_TYPE_F32 && w->nb[0] == ggml_type_size(w->type) && op->nb[0] == sizeof(float) && w->ne[0] % 64 == 0 && !ggml_is_transposed(w) &&
                       x->ne[2] % w->ne[2] == 0 && x->ne[3] % w->ne[3] == 0;
            }
            return mm_types_ok(w, x, op) && x->ne[2] % w->ne[2] == 0 && x->ne[3] % w->ne[3] == 0;
        }
        case GGML_OP_MUL_MAT_ID: return mm_types_ok(op->src[0], op->src[1], op) && !is_bitnet(op->src[0]) && op->src[2]->type == GGML_TYPE_I32 && op->src[1]->ne[3] == 1;
        case GGML_OP_FUSED_UP_GATE: {
            return op->src[1] && op->src[0]->type == op->src[1]->type && ggml_are_same_shape(op->src[0], op->src[1]) && mm_types_ok(op->src[0], op->src[2], op) && !is_bitnet(op->src[0]) &&
                   weight_ok(op->src[1]) && op->src[2]->ne[2] == 1 && op->src[2]->ne[3] == 1 && up_gate_unary_ok(op->op_params[0]);
        }
        case GGML_OP_MOE_FUSED_UP_GATE: {
            // src[1] == NULL: up and gate MERGED in one tensor per expert -- rows [0, ne01 / 2) are gate, rows [ne01 / 2, ne01) up, the biases of both in src[4] the same way
            // (ggml.c:18470-18600: src0_1_cur = src0_2_cur + nb02 / 2); served from the same kernels with two pointers into the one tensor (not for _R4 tensors)
            const ggml_tensor *up = op->src[0];
            if (!op->src[1]) return mm_types_ok(up, op->src[2], op) && !is_bitnet(up) && !is_r4_type(up->type) && up->ne[1] % 2 == 0 && up->nb[2] == (size_t)up->ne[1] * up->nb[1] && !op->src[5] &&
                                    op->ne[0] == up->ne[1] / 2 && op->src[3] && op->src[3]->type == GGML_TYPE_I32 && bias_ok(op->src[4], up) && up_gate_unary_ok(op->op_params[0]);
            return op->src[0]->type == op->src[1]->type && mm_types_ok(op->src[0], op->src[2], op) && !is_bitnet(op->src[0]) && weight_ok(op->src[1]) && op->src[3] && op->src[3]->type == GGML_TYPE_I32 &&
                   bias_ok(op->src[4], op->src[0]) && bias_ok(op->src[5], op->src[0]) && up_gate_unary_ok(op->op_params[0]);
        }
        // ---- the non-mat-mul ops of a Llama / Mixtral graph, same conditions as the C ABI entries (ops.hip)
        case GGML_OP_ADD: case GGML_OP_MUL: case GGML_OP_DIV:
            return is_f32_f16(op->type) && is_f32_f16(op->src[0]->type) && is_f32_f16(op->src[1]->type) && ggml_can_repeat(op->src[1], op->src[0]);
        case GGML_OP_RMS_NORM: case GGML_OP_FUSED_RMS_NORM:
            return op->type == GGML_TYPE_F32 && is_f32_f16(op->src[0]->type) && op->src[0]->nb[0] == ggml_type_size(op->src[0]->type) && op->nb[0] == sizeof(float) &&
                   (!op->src[1] || (op->src[1]->type == GGML_TYPE_F32 && ggml_nrows(op->src[1]) == 1 && op->src[1]->ne[0] == op->src[0]->ne[0] && op->src[1]->nb[0] == sizeof(float)));
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2], n_dims = op->op_params[1];
            return op->type == GGML_TYPE_F32 && op->src[0]->type == GGML_TYPE_F32 && op->src[1]->type == GGML_TYPE_I32 && (mode == 0 || (mode == 2 && n_dims == op->ne[0])) &&
                   n_dims > 0 && n_dims % 2 == 0 && n_dims <= op->ne[0] && op->ne[0] % 2 == 0 && ggml_are_same_shape(op, op->src[0]) &&        // (ops.hip cdna4_op_rope)
                   op->op_params[15] != 1 && op->src[0]->nb[0] == sizeof(float) && op->nb[0] == sizeof(float) && (!op->src[2] || op->src[2]->type == GGML_TYPE_F32);
        }
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const ggml_tensor *d = op->op == GGML_OP_CPY ? op->src[1] : op;
            return (op->src[0]->type == GGML_TYPE_F32 || op->src[0]->type == GGML_TYPE_F16) && (d->type == GGML_TYPE_F32 || d->type == GGML_TYPE_F16) && ggml_nelements(op->src[0]) == ggml_nelements(d);
        }
        case GGML_OP_GET_ROWS: {
            const ggml_tensor *a = op->src[0];
            const bool t_ok = a->type == GGML_TYPE_F32 || a->type == GGML_TYPE_F16 || (cdna4_type_supported(a->type) && !is_r4_type(a->type) && !is_bitnet(a));
            return t_ok && op->type == GGML_TYPE_F32 && op->src[1]->type == GGML_TYPE_I32 && a->nb[0] == ggml_type_size(a->type) && (a->ne[2] == 1 || a->ne[2] == op->src[1]->ne[1]) && (a->ne[3] == 1 || a->ne[3] == op->src[1]->ne[2]);
        }
        case GGML_OP_SOFT_MAX:
            return op->type == GGML_TYPE_F32 && op->src[0]->type == GGML_TYPE_F32 && !op->src[2] && ggml_is_contiguous(op->src[0]) && op->nb[0] == sizeof(float) &&
                   (!op->src[1] || ((op->src[1]->type == GGML_TYPE_F16 || op->src[1]->type == GGML_TYPE_F32) && op->src[1]->ne[0] >= op->src[0]->ne[0] && op->src[1]->ne[1] >= op->src[0]->ne[1]));
        case GGML_OP_FLASH_ATTN_EXT: {
            const ggml_tensor *q = op->src[0], *k = op->src[1], *v = op->src[2], *m = op->src[3];
            const bool ok_types = q->type == GGML_TYPE_F32 && k->type == GGML_TYPE_F16 && v->type == GGML_TYPE_F16 && op->type == GGML_TYPE_F32 && !op->src[4] &&
                                  (q->ne[0] == 64 || q->ne[0] == 128 || q->ne[0] == 256);
            if (!ok_types) {      // attention on the CPU backend means a PCIe round trip of q / the KV window / the result in every layer: say so once, do not fail silently
                static std::atomic<bool> said{false};
                if (!said.exchange(true)) shim_log(GGML_LOG_LEVEL_WARN, "ggml-hip-cdna4: FLASH_ATTN_EXT with q %s, K %s, V %s, head size %lld%s is not served on the device "
                    "(f32 q, f16 K / V, head size 64 / 128 / 256, no sinks): attention falls to the CPU backend in every layer -- expect a large slow-down (use an f16 KV cache)\n",
                    ggml_type_name(q->type), ggml_type_name(k->type), ggml_type_name(v->type), (long long)q->ne[0], op->src[4] ? ", attention sinks" : "");
                return false;
            }
            return q->type == GGML_TYPE_F32 && k->type == GGML_TYPE_F16 && v->type == GGML_TYPE_F16 && op->type == GGML_TYPE_F32 && !op->src[4] && (q->ne[0] == 64 || q->ne[0] == 128 || q->ne[0] == 256) &&
                   k->ne[0] == q->ne[0] && v->ne[0] == q->ne[0] && q->nb[0] == 4 && k->nb[0] == 2 && v->nb[0] == 2 && k->nb[1] % 16 == 0 && k->nb[2] % 16 == 0 && k->nb[3] % 16 == 0 && v->nb[1] % 4 == 0 &&
                   (!m || (m->type == GGML_TYPE_F16 && m->nb[0] == 2 && m->ne[0] >= k->ne[1] && m->ne[1] >= q->ne[1])) && q->ne[2] <= 65535 && q->ne[3] <= 65535 &&
                   op->nb[0] == 4 && op->ne[0] == q->ne[0] && op->ne[1] == q->ne[2] && op->ne[2] == q->ne[1] && k->ne[1] == v->ne[1] && k->ne[2] > 0 && v->ne[2] > 0 &&
                   q->ne[2] % k->ne[2] == 0 && q->ne[2] % v->ne[2] == 0 && q->ne[3] % k->ne[3] == 0 && q->ne[3] % v->ne[3] == 0 &&
                   // (the entry point also wants K rows 16-byte and V rows 4-byte aligned: every tensor / view offset ggml hands out is a multiple of the row size on
                   // a base aligned to the buffer type's 128 bytes, which the stride conditions above turn into exactly that)
                   (!k->data || ((uintptr_t)k->data % 16 == 0 && (uintptr_t)v->data % 4 == 0));
        }
        case GGML_OP_ARGSORT: return op->src[0]->type == GGML_TYPE_F32 && op->src[0]->ne[0] <= 16384 && op->src[0]->nb[0] == sizeof(float) && op->type == GGML_TYPE_I32 &&
                                     ggml_are_same_shape(op, op->src[0]) && op->nb[0] == sizeof(int32_t);
        case GGML_OP_SUM_ROWS: return op->src[0]->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32;
        case GGML_OP_MUL_MULTI_ADD: return !op->src[2] && !op->src[3] && op->src[0]->type == GGML_TYPE_F32 && op->src[1]->type == GGML_TYPE_F32 && op->src[0]->ne[2] <= 65535;
        case GGML_OP_REDUCE:                 // reduce.cu:125-134 (F32 / F16 / BF16 and Q8_0 partial sums)
            return op->op_params[0] == GGML_OP_ADD && (op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_BF16 || op->type == GGML_TYPE_Q8_0) && ggml_is_contiguous(op) &&
                   op->op_params[1] >= 1 && op->op_params[1] <= GGML_CUDA_MAX_DEVICES;
        default: return false;
    }
}

// the type id handed to the C ABI for a weight tensor: an _R4 tensor is re-tiled now if the upload came in pieces (llama-model-loader.cpp
// chunked async upload) and goes down as CDNA4_TYPE_PRETILED (base-layout kernels on the bytes as they are)
static int abi_type(const ggml_tensor *w) {
    if (!is_r4_type(w->type)) return w->type;
    if (!r4_is_tiled(w->buffer, w)) r4_set_state(w->buffer, w, true);
    return is_r4h_type(w->type) ? r4_base_type(w->type) : CDNA4_TYPE_PRETILED(w->type);        // (host re-tiled: the bytes now ARE a tensor of the base type)
}

// is tensor t (or a view of it) read by any node from index `from` on, or a graph output?  (fusions that skip materializing t must know)
// The eager walk asks this for every fusion candidate; as a scan over the rest of the graph it was quadratic in the node count and most of the walk's host time (measured on the
// stand-in runtime, where nothing else costs anything: 1.55 ms per decoded token of an 8B graph).  run_nodes() therefore indexes the graph once -- the LAST node that reads each
// tensor (as a source, through a view of it, or as a view node of it) -- and the question becomes one hash lookup.  GGML_CDNA4_CHECK_USES=1 answers both ways and aborts on a
// difference (tests/test_shim_host_logic.py runs the libllama graphs with it).
static bool used_from_scan(const ggml_cgraph *g, int from, const ggml_tensor *t) {
    if (t->flags & GGML_TENSOR_FLAG_OUTPUT) return true;
    for (int k = from; k < g->n_nodes; ++k) { const ggml_tensor *m = g->nodes[k];
        if (m->view_src == t) return true;
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (m->src[s] && (m->src[s] == t || m->src[s]->view_src == t)) return true; }
    return false;
}
struct use_index { const ggml_cgraph *g = nullptr; std::unordered_map<const ggml_tensor *, int> last; };
static thread_local use_index t_uses;
static void index_uses(const ggml_cgraph *g) {
    t_uses.g = g; t_uses.last.clear(); t_uses.last.reserve((size_t)g->n_nodes * 4);
    for (int k = 0; k < g->n_nodes; ++k) { const ggml_tensor *m = g->nodes[k];
        if (m->view_src) t_uses.last[m->view_src] = k;
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (m->src[s]) { t_uses.last[m->src[s]] = k; if (m->src[s]->view_src) t_uses.last[m->src[s]->view_src] = k; } }
}
static bool used_from(const ggml_cgraph *g, int from, const ggml_tensor *t) {
    if (t_uses.g != g) return used_from_scan(g, from, t);                 // (not inside an indexed walk)
    bool r = (t->flags & GGML_TENSOR_FLAG_OUTPUT) != 0;
    if (!r) { auto it = t_uses.last.find(t); r = it != t_uses.last.end() && it->second >= from; }
    static const bool check_both = getenv("GGML_CDNA4_CHECK_USES") != nullptr;
    if (check_both && r != used_from_scan(g, from, t)) GGML_ABORT("ggml-hip-cdna4: use index disagrees with the scan");
    return r;
}
// do the bytes of a and b overlap?  A fused launch reads its inputs while other workgroups already write results, and the graph allocator may
// place a result in the memory of an input whose last consumer (the node fused away) has "run": such pairs must not be fused.
static bool overlaps(const ggml_tensor *a, const ggml_tensor *b) {
    const char *a0 = (const char *)a->data, *b0 = (const char *)b->data;
    return a0 < b0 + ggml_nbytes(b) && b0 < a0 + ggml_nbytes(a);
}
// A fused launch reads operands while sibling workgroups already write results.  The graph allocator knows nothing of that: it may place a LATER node's result in the memory of
// an EARLIER node's dead operand (round 4's soak: the rotated K of ROPE(k) lay exactly over the un-rotated Q that ROPE(q), two nodes earlier, had consumed -- the fused
// ROPE + ROPE + KV-store launch then overwrote Q rows that other workgroups had not read yet: a wrong prompt once in ~50 passes).  Rule for every fusion site: a result may
// coincide EXACTLY with the element-wise operand of its OWN role (x -> rope(x) in place, residual += ...), and must be disjoint from every other operand and result.
static bool same_or_disjoint(const ggml_tensor *o, const ggml_tensor *a) { return !o || !a || !o->data || !a->data || !overlaps(o, a) || (o->data == a->data && ggml_nbytes(o) == ggml_nbytes(a)); }
struct alias_pair { const ggml_tensor *out, *in; };
// true when the operand layout is safe for ONE launch: `outs` vs `ins` disjoint, outs pairwise disjoint, and for each {out, in} of `own`: out exactly over in or disjoint from it;
// a result that is NOT listed with an `in` of `own` must be disjoint from it
static bool fusable_layout(std::initializer_list<const ggml_tensor *> outs, std::initializer_list<const ggml_tensor *> ins, std::initializer_list<alias_pair> own = {}) {
    for (const ggml_tensor *o : outs) { if (!o || !o->data) continue;
        for (const ggml_tensor *i : ins) if (i && i->data && overlaps(o, i)) return false;
        for (const ggml_tensor *o2 : outs) if (o2 && o2 != o && o2->data && overlaps(o, o2)) return false;
        for (const alias_pair &p : own) { if (!p.in || !p.in->data) continue;
            bool mine = false; for (const alias_pair &q : own) mine = mine || (q.out == o && q.in == p.in);          // (an operand may be listed for several results)
            if (mine ? !same_or_disjoint(o, p.in) : overlaps(o, p.in)) return false; } }
    return true;
}
// GGML_CDNA4_CHECK_OVERLAP=1 (debug switch, scripts/soak_logits.py): every FUSED launch asserts the rule above on the host before it is issued -- the conditions the fusion
// sites test, checked once more in one place and over ALL operands
static void assert_disjoint(const char *what, std::initializer_list<const ggml_tensor *> outs, std::initializer_list<const ggml_tensor *> ins, std::initializer_list<alias_pair> own = {}) {
    static const bool on = getenv("GGML_CDNA4_CHECK_OVERLAP") != nullptr;
    if (on && !fusable_layout(outs, ins, own)) {
        for (const ggml_tensor *o : outs) if (o) fprintf(stderr, "  result  %-24s [%p, +%zu)\n", o->name, o->data, ggml_nbytes(o));
        for (const ggml_tensor *i : ins) if (i) fprintf(stderr, "  operand %-24s [%p, +%zu)\n", i->name, i->data, ggml_nbytes(i));
        for (const alias_pair &p : own) if (p.in) fprintf(stderr, "  element-wise operand %-24s [%p, +%zu) of result %s\n", p.in->name, p.in->data, ggml_nbytes(p.in), p.out ? p.out->name : "-");
        GGML_ABORT("ggml-hip-cdna4: %s: a result of the fused launch overlaps an operand it must not", what);
    }
}
static cdna4_tensor td(const ggml_tensor *t) { cdna4_tensor d; d.data = t->data; d.type = t->type; for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t)t->nb[i]; } return d; }
static float f32_param(const ggml_tensor *n, int i) { float f; memcpy(&f, n->op_params + i, sizeof(f)); return f; }

// GGML_CDNA4_FUSION_OFF=<mask> (debug switch, scripts/soak_logits.py --bisect): switches single fusions off.  1 ADD + RMS_NORM; 2 ROPE + ROPE + KV stores; 4 MUL_MATs sharing src1;
// 8 RMS_NORM inside the mat-mul launch; 16 MUL_MAT + residual ADD; 32 q,k,v + ROPE + KV store epilogue; 64 MoE router chain / MUL_MULTI_ADD + ADD / expert FFN block;
// 128 FLASH_ATTN_EXT + attn_output MUL_MAT + ADD
static bool fusion_off(int bit) { static const int mask = getenv("GGML_CDNA4_FUSION_OFF") ? atoi(getenv("GGML_CDNA4_FUSION_OFF")) : 0; return (mask & bit) != 0; }
static bool node_is_noop(const ggml_tensor *n);
static int next_real(const ggml_cgraph *g, int i) { for (; i < g->n_nodes; ++i) if (!node_is_noop(g->nodes[i])) return i; return -1; }
// slot of the next cache-write node while a HIP graph is being captured (nullptr otherwise: the kernel then uses the address it is given)
static void *const *take_slot(shim_context *c) { return c->capturing ? c->slots_dev + c->slot_next++ : nullptr; }

static bool node_is_noop(const ggml_tensor *n) { return n->op == GGML_OP_NONE || n->op == GGML_OP_RESHAPE || n->op == GGML_OP_VIEW || n->op == GGML_OP_PERMUTE || n->op == GGML_OP_TRANSPOSE; }

// run nodes [i, ...) ; returns the number of nodes consumed (>= 1): consecutive same-src1 MUL_MATs and the 2-node MoE block are fused
// Consecutive MUL_MATs of leaf weights sharing src1 (q,k,v) go out as one call, like ggml.c:17984-18000 / ggml-cuda.cu:2570-2600: same-type
// matrices (and a K-quant group + a Q6_K matrix) become ONE decode launch.  Returns the number of graph nodes of the group starting at node i.
static int mm_group_size(ggml_backend_t be, shim_context *c, const ggml_cgraph *g, int i) {
    const ggml_tensor *n = g->nodes[i], *w = n->src[0], *x = n->src[1];
    auto plain2d = [](const ggml_tensor *t) { return t->ne[2] == 1 && t->ne[3] == 1; };
    int cnt = 1;
    if (c->params.fusion && !fusion_off(4) && plain2d(w) && plain2d(x) && w->op == GGML_OP_NONE) {
        while (i + cnt < g->n_nodes && cnt < 5) {
            const ggml_tensor *m = g->nodes[i + cnt];
            if (m->op != GGML_OP_MUL_MAT || m->src[1] != x || m->src[0]->op != GGML_OP_NONE || !plain2d(m->src[0]) || !be_supports_op(be, m) ||
                m->src[0]->ne[0] != w->ne[0]) break;
            ++cnt;
        }
    }
    return cnt;
}
// runs the group; with `norm` the activation row is `norm->src[0]`, RMS-normed with norm->src[1] inside the launch (returns -1 if that form is unsupported)
static int mm_group_run(shim_context *c, const ggml_cgraph *g, int i, int cnt, const ggml_tensor *norm) {
    const ggml_tensor *n = g->nodes[i], *w = n->src[0], *x = norm ? norm->src[0] : n->src[1];
    if (cnt > 1 || norm) {
        long nx[5], sa[5], sc[5]; int ty[5]; const void *ap[5]; float *cp[5];
        for (int j = 0; j < cnt; ++j) {
            const ggml_tensor *m = g->nodes[i + j];
            nx[j] = m->src[0]->ne[1]; sa[j] = m->src[0]->nb[1]; sc[j] = m->nb[1] / sizeof(float); ty[j] = abi_type(m->src[0]); ap[j] = m->src[0]->data; cp[j] = (float *)m->data;
        }
        if (norm) {
            for (int j = 0; j < cnt; ++j) { assert_disjoint("RMS_NORM + MUL_MAT", {g->nodes[i + j]}, {x, norm->src[1], g->nodes[i + j]->src[0]}); for (int k = j + 1; k < cnt; ++k) assert_disjoint("RMS_NORM + MUL_MAT", {g->nodes[i + j], g->nodes[i + k]}, {}); }
            cdna4_fusion fx = {(const float *)norm->src[1]->data, f32_param(norm, 0), nullptr, nullptr};
            const int rc = cdna4_mul_mat_multi_fused(c->ctx, cnt, nx, 1, w->ne[0], ty, ap, sa, x->type, x->data, x->nb[1], cp, sc, &fx, c->stream);
            if (rc == CDNA4_E_UNSUPPORTED) return -1;
            check(rc, "RMS_NORM + MUL_MAT"); ++c->n_fuse[3]; return cnt;
        }
        for (int j = 0; j < cnt; ++j) { assert_disjoint("MUL_MAT (shared src1)", {g->nodes[i + j]}, {x, g->nodes[i + j]->src[0]}); for (int k = j + 1; k < cnt; ++k) assert_disjoint("MUL_MAT (shared src1)", {g->nodes[i + j], g->nodes[i + k]}, {}); }
        check(cdna4_mul_mat_multi(c->ctx, cnt, nx, x->ne[1], w->ne[0], ty, ap, sa, x->type, x->data, x->nb[1], cp, sc, c->stream), "MUL_MAT (fused, shared src1)");
        ++c->n_fuse[2]; return cnt;
    }
    check(cdna4_mul_mat_4d(c->ctx, w->ne[1], x->ne[1], w->ne[0], w->ne[2], w->ne[3], x->ne[2], x->ne[3], w->nb[2], w->nb[3], x->nb[2], x->nb[3],
                           n->nb[2] / sizeof(float), n->nb[3] / sizeof(float), abi_type(w), w->data, w->nb[1], x->type, x->data, x->nb[1],
                           (float *)n->data, n->nb[1] / sizeof(float), c->stream), "MUL_MAT");
    return 1;
}

// every layer of a graph rotates with the same angles: (cos, sin) are computed once per graph (ggml_rope_cache_init on the CPU); a model that changes the parameters from
// layer to layer stops caching after two refills.  Returns true when the context's cache describes rope node `n`.
static bool ensure_rope_cache(shim_context *c, const ggml_tensor *n) {
    static const bool rope_cache = getenv("GGML_CDNA4_NO_ROPE_CACHE") == nullptr;
    if (!rope_cache) return false;
    const bool same = c->rope_pos == n->src[1]->data && memcmp(c->rope_params, n->op_params, sizeof(c->rope_params)) == 0;
    if (same) return true;
    if (c->rope_fills > 2) return false;
    const int rc = cdna4_op_rope_cache(c->ctx, (const int32_t *)n->src[1]->data, n->ne[2], n->src[2] ? (const float *)n->src[2]->data : nullptr, n->op_params[1], n->op_params[4],
                                       f32_param(n, 5), f32_param(n, 6), f32_param(n, 7), f32_param(n, 8), f32_param(n, 9), f32_param(n, 10), c->stream);
    ++c->rope_fills;
    if (rc == CDNA4_OK) { c->rope_pos = n->src[1]->data; memcpy(c->rope_params, n->op_params, sizeof(c->rope_params)); return true; }
    c->rope_pos = nullptr; return false;
}

// One decoded token: FUSED_RMS_NORM + the q,k,v MUL_MATs + ROPE(q) + ROPE(k) + CPY(k -> K cache) + CPY(v -> V cache) as ONE launch (the rotation and the f16 cache writes ride in
// the mat-mul's epilogue, cdna4_fusion.qkv).  `j` = first of the `cnt` mat-mul nodes consuming norm node `norm`.  Returns the index after the last node consumed, or -1.
static int try_norm_qkv_rope(shim_context *c, const ggml_cgraph *g, const ggml_tensor *norm, int j, int cnt) {
    static const bool on = getenv("GGML_CDNA4_NO_QKV_ROPE_FUSION") == nullptr;
    if (!on || fusion_off(32) || cnt != 3) return -1;
    const int j1 = next_real(g, j + cnt), j2 = j1 >= 0 ? next_real(g, j1 + 1) : -1, j3 = j2 >= 0 ? next_real(g, j2 + 1) : -1, j4 = j3 >= 0 ? next_real(g, j3 + 1) : -1;
    if (j4 < 0) return -1;
    const ggml_tensor *rq = g->nodes[j1], *rk = g->nodes[j2], *ck = g->nodes[j3], *cv = g->nodes[j4];
    if (rq->op != GGML_OP_ROPE || rk->op != GGML_OP_ROPE || ck->op != GGML_OP_CPY || cv->op != GGML_OP_CPY || ck->src[0] != rk) return -1;
    auto root = [](const ggml_tensor *t) { return t->view_src ? t->view_src : t; };
    int iq = -1, ik = -1, iv = -1;
    for (int q = 0; q < cnt; ++q) { const ggml_tensor *m = g->nodes[j + q]; if (root(rq->src[0]) == m) iq = q; else if (root(rk->src[0]) == m) ik = q; else if (root(cv->src[0]) == m) iv = q; }
    if (iq < 0 || ik < 0 || iv < 0) return -1;
    const ggml_tensor *mq = g->nodes[j + iq], *mk = g->nodes[j + ik], *mv = g->nodes[j + iv], *kc = ck->src[1], *vc = cv->src[1];
    // ROPE: NORM mode, one token, same parameters on q and k; head size from the rope views
    if (rq->op_params[2] != 0 || memcmp(rk->op_params, rq->op_params, sizeof(rq->op_params)) != 0 || rk->src[1] != rq->src[1] || rk->src[2] != rq->src[2] || rq->ne[2] != 1 || rk->ne[2] != 1 ||
        rq->ne[3] != 1 || rk->ne[3] != 1 || rk->ne[0] != rq->ne[0] || rq->type != GGML_TYPE_F32 || !ggml_is_contiguous(rq) || rq->src[0]->type != GGML_TYPE_F32 || rk->src[0]->type != GGML_TYPE_F32) return -1;
    const long hd = rq->ne[0];
    if (ggml_nelements(rq) != mq->ne[0] || ggml_nelements(rk) != mk->ne[0] || mq->ne[0] % hd || mk->ne[0] % hd || mv->ne[0] % 2) return -1;
    if (kc->type != GGML_TYPE_F16 || vc->type != GGML_TYPE_F16 || !ggml_is_contiguous(kc) || !ggml_is_contiguous(vc) || ggml_nelements(kc) != mk->ne[0] || ggml_nelements(vc) != mv->ne[0] ||
        cv->src[0]->type != GGML_TYPE_F32 || ggml_nelements(cv->src[0]) != mv->ne[0]) return -1;
    // nothing else may read the intermediates that are no longer written (mat-mul results, rotated K), and no result may lie over the un-normed input row
    if (used_from(g, j4 + 1, mq) || used_from(g, j4 + 1, mk) || used_from(g, j4 + 1, mv) || used_from(g, j4 + 1, rk) || used_from(g, j + cnt, norm)) return -1;
    for (int q = j + cnt; q <= j4; ++q) { const ggml_tensor *m = g->nodes[q]; if (!node_is_noop(m) && m != rq && m != rk && m != ck && m != cv) return -1; }
    if (!fusable_layout({rq, kc, vc}, {norm->src[0], rq->src[1], rq->src[2]})) return -1;
    if (!ensure_rope_cache(c, rq)) return -1;
    cdna4_qkv_epilogue qe; memset(&qe, 0, sizeof(qe)); qe.head_dim = (int)hd; qe.n_dims = rq->op_params[1];
    const int slot0 = c->slot_next;
    void *const *ks = take_slot(c), *const *vs = take_slot(c);
    qe.kind[iq] = 0; qe.kind[ik] = 1; qe.kind[iv] = 2; qe.kv_dst[ik] = kc->data; qe.kv_dst[iv] = vc->data; qe.kv_slot[ik] = ks; qe.kv_slot[iv] = vs;
    long nx[3], sa[3], sc[3]; int ty[3]; const void *ap[3]; float *cp[3];
    const ggml_tensor *w0 = g->nodes[j]->src[0], *x = norm->src[0];
    for (int q = 0; q < cnt; ++q) {
        const ggml_tensor *m = g->nodes[j + q];
        nx[q] = m->src[0]->ne[1]; sa[q] = m->src[0]->nb[1]; sc[q] = m->nb[1] / sizeof(float); ty[q] = abi_type(m->src[0]); ap[q] = m->src[0]->data; cp[q] = (float *)(q == iq ? rq->data : m->data);
    }
    assert_disjoint("RMS_NORM + q,k,v + ROPE + KV store", {rq, kc, vc}, {x, norm->src[1], mq->src[0], mk->src[0], mv->src[0], rq->src[1]});
    cdna4_fusion fx = {(const float *)norm->src[1]->data, f32_param(norm, 0), nullptr, &qe};
    const int rc = cdna4_mul_mat_multi_fused(c->ctx, cnt, nx, 1, w0->ne[0], ty, ap, sa, x->type, x->data, x->nb[1], cp, sc, &fx, c->stream);
    if (rc == CDNA4_E_UNSUPPORTED) { c->slot_next = slot0; return -1; }
    check(rc, "RMS_NORM + q,k,v MUL_MAT + ROPE + KV store");
    ++c->n_fuse[5]; return j4 + 1;
}

// PROMPT batch: [ADD ->] FUSED_RMS_NORM -> {MUL_MATs sharing it | FUSED_UP_GATE}, the normed rows read by nothing else: ONE C-ABI call -- the norm (and the ADD) ride in the launch
// that builds the f16 activation image of the GEMM (cdna4_fusion: prompt batches), the normed f32 rows are never written.  ia = the ADD node or -1, in = the norm node.
// Layout: the image launch reads x (and the second addend) completely before the first GEMM starts, so a mat-mul result MAY lie over a dead operand of the ADD; results must be
// clear of the weights, the norm weights, the sum and each other; the sum may coincide exactly with one of its addends (element-wise, read before written by the same thread).
// Returns the index one past the last node consumed, or -1 (the caller issues the nodes one by one).
static int try_prompt_norm_mm(ggml_backend_t be, shim_context *c, const ggml_cgraph *g, int ia, int in) {
    static const bool mm_fusion = getenv("GGML_CDNA4_NO_MM_FUSION") == nullptr;
    if (!mm_fusion || !c->params.fusion || fusion_off(8) || (ia >= 0 && fusion_off(1))) return -1;
    const ggml_tensor *n = g->nodes[in], *add = ia >= 0 ? g->nodes[ia] : nullptr, *x = add ? add->src[0] : n->src[0], *wn = n->src[1];
    if (n->op != GGML_OP_FUSED_RMS_NORM || !wn || wn->type != GGML_TYPE_F32 || n->type != GGML_TYPE_F32 || x->type != GGML_TYPE_F32 || ggml_nrows(n) <= 8 || n->ne[2] != 1 || n->ne[3] != 1 ||
        n->ne[0] % 128 || n->ne[0] > 16384 || !ggml_is_contiguous(x)) return -1;
    if (add && (add->type != GGML_TYPE_F32 || add->src[1]->type != GGML_TYPE_F32 || !ggml_are_same_shape(add->src[0], add->src[1]) || !ggml_is_contiguous(add) || !ggml_is_contiguous(add->src[1]) ||
                !same_or_disjoint(add, add->src[0]) || !same_or_disjoint(add, add->src[1]) || overlaps(add, wn))) return -1;
    const int j = next_real(g, in + 1); const ggml_tensor *m = j >= 0 ? g->nodes[j] : nullptr;
    if (!m) return -1;
    auto out_ok = [&](const ggml_tensor *o, const ggml_tensor *w1, const ggml_tensor *w2) {
        return !overlaps(o, wn) && !overlaps(o, w1) && (!w2 || !overlaps(o, w2)) && (!add || (!overlaps(o, add) && !overlaps(add, w1) && (!w2 || !overlaps(add, w2))));
    };
    cdna4_fusion fx = {(const float *)wn->data, f32_param(n, 0), nullptr, nullptr, add ? (const float *)add->src[1]->data : nullptr, add ? (float *)add->data : nullptr};
    if (m->op == GGML_OP_MUL_MAT && m->src[1] == n && ggml_is_quantized(m->src[0]->type) && m->src[0]->ne[2] == 1 && m->src[0]->ne[3] == 1 && be_supports_op(be, m)) {
        const int cnt = mm_group_size(be, c, g, j);
        if (used_from(g, j + cnt, n)) return -1;
        long nx[5], sa[5], sc[5]; int ty[5]; const void *ap[5]; float *cp[5];
        for (int q = 0; q < cnt; ++q) {
            const ggml_tensor *mq = g->nodes[j + q];
            if (is_r4_type(mq->src[0]->type) || !out_ok(mq, mq->src[0], nullptr)) return -1;
            for (int k = 0; k < cnt; ++k) if (k != q && (overlaps(mq, g->nodes[j + k]) || overlaps(mq, g->nodes[j + k]->src[0]))) return -1;
            nx[q] = mq->src[0]->ne[1]; sa[q] = mq->src[0]->nb[1]; sc[q] = mq->nb[1] / sizeof(float); ty[q] = abi_type(mq->src[0]); ap[q] = mq->src[0]->data; cp[q] = (float *)mq->data;
        }
        const int rc = cdna4_mul_mat_multi_fused(c->ctx, cnt, nx, x->ne[1], m->src[0]->ne[0], ty, ap, sa, GGML_TYPE_F32, x->data, x->nb[1], cp, sc, &fx, c->stream);
        if (rc == CDNA4_E_UNSUPPORTED) return -1;
        check(rc, add ? "ADD + RMS_NORM + MUL_MAT (prompt)" : "RMS_NORM + MUL_MAT (prompt)"); ++c->n_fuse[3]; if (add) ++c->n_fuse[0];
        return j + cnt;
    }
    if (m->op == GGML_OP_FUSED_UP_GATE && m->src[2] == n && !is_r4_type(m->src[0]->type) && be_supports_op(be, m) && !used_from(g, j + 1, n) && out_ok(m, m->src[0], m->src[1])) {
        const ggml_tensor *up = m->src[0], *gate = m->src[1]; const float limit = *(const float *)(m->op_params + 1);
        const int ty = abi_type(up); (void)abi_type(gate);
        const int rc = cdna4_fused_up_gate_fused(c->ctx, up->ne[1], x->ne[1], up->ne[0], m->op_params[0], ty, up->data, gate->data, up->nb[1], GGML_TYPE_F32, x->data, x->nb[1],
                                                 nullptr, nullptr, limit, (float *)m->data, m->nb[1] / sizeof(float), &fx, c->stream);
        if (rc == CDNA4_E_UNSUPPORTED) return -1;
        check(rc, add ? "ADD + RMS_NORM + FUSED_UP_GATE (prompt)" : "RMS_NORM + FUSED_UP_GATE (prompt)"); ++c->n_fuse[3]; if (add) ++c->n_fuse[0];
        return j + 1;
    }
    return -1;
}

// one decoded token: would FUSED_RMS_NORM node `jn` ride in the prologue of the mat-mul(s) that consume it (compute_node, RMS_NORM case)?  Then the residual ADD in front of it
// is better left alone (ADD + norm as one kernel would keep the norm -- and with it the q,k,v epilogue fusion -- out of the mat-mul launch)
static bool norm_rides_in_matmul(ggml_backend_t be, shim_context *c, const ggml_cgraph *g, int jn) {
    static const bool mm_fusion = getenv("GGML_CDNA4_NO_MM_FUSION") == nullptr;
    const ggml_tensor *n = g->nodes[jn];
    if (!mm_fusion || !c->params.fusion || fusion_off(8) || n->op != GGML_OP_FUSED_RMS_NORM || !n->src[1] || ggml_nrows(n) != 1 || n->src[0]->type != GGML_TYPE_F32 || !ggml_is_contiguous(n->src[0]) ||
        n->src[1]->type != GGML_TYPE_F32 || n->ne[0] > 8192 || n->ne[0] % 256) return false;
    const int j = next_real(g, jn + 1); const ggml_tensor *m = j >= 0 ? g->nodes[j] : nullptr;
    if (!m) return false;
    if (m->op == GGML_OP_MUL_MAT && m->src[1] == n && ggml_is_quantized(m->src[0]->type) && m->src[0]->ne[2] == 1 && m->src[0]->ne[3] == 1 && be_supports_op(be, m)) {
        const int cnt = mm_group_size(be, c, g, j);
        for (int q = 0; q < cnt; ++q) if (is_r4_type(g->nodes[j + q]->src[0]->type)) return false;
        return !used_from(g, j + cnt, n);
    }
    return m->op == GGML_OP_FUSED_UP_GATE && m->src[2] == n && !is_r4_type(m->src[0]->type) && be_supports_op(be, m) && !used_from(g, j + 1, n);
}

static int compute_node(ggml_backend_t be, shim_context *c, ggml_cgraph *g, int i) {
    ggml_tensor *n = g->nodes[i];
    switch (n->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return 1;
        case GGML_OP_ADD: case GGML_OP_MUL: case GGML_OP_DIV: {
            const cdna4_tensor a = td(n->src[0]), b = td(n->src[1]), d = td(n);
            if (n->op == GGML_OP_ADD && c->params.fusion && !fusion_off(1)) {         // ADD + FUSED_RMS_NORM of its result (residual add followed by the next norm)
                const int j = next_real(g, i + 1); const ggml_tensor *m = j >= 0 ? g->nodes[j] : nullptr;
                if (m && m->op == GGML_OP_FUSED_RMS_NORM && m->src[0] == n && ggml_nrows(n) > 8) { const int e = try_prompt_norm_mm(be, c, g, i, j); if (e > 0) return e - i; }
                if (m && m->op == GGML_OP_FUSED_RMS_NORM && m->src[0] == n && m->src[1] && n->type == GGML_TYPE_F32 && n->src[0]->type == GGML_TYPE_F32 && n->src[1]->type == GGML_TYPE_F32 &&
                    ggml_are_same_shape(n->src[0], n->src[1]) && n->src[0]->nb[0] == 4 && n->src[1]->nb[0] == 4 && n->nb[0] == 4 && m->nb[0] == 4 && m->data != n->data && supports_op_impl(m) &&
                    fusable_layout({n, m}, {m->src[1]}, {{n, n->src[0]}, {n, n->src[1]}, {m, n->src[0]}, {m, n->src[1]}}) &&      // (row r of a result over row r of an operand: read before written by the row's own workgroup)
                    !norm_rides_in_matmul(be, c, g, j)) {
                    assert_disjoint("ADD + RMS_NORM", {n, m}, {m->src[1]}, {{n, n->src[0]}, {n, n->src[1]}, {m, n->src[0]}, {m, n->src[1]}});
                    const cdna4_tensor w = td(m->src[1]), y = td(m);
                    check(cdna4_op_add_rms_norm(c->ctx, &a, &b, &d, &w, f32_param(m, 0), &y, c->stream), "ADD + RMS_NORM"); ++c->n_fuse[0]; return j + 1 - i;
                }
            }
            check(cdna4_op_binary(c->ctx, n->op == GGML_OP_ADD ? 0 : n->op == GGML_OP_MUL ? 1 : 2, &a, &b, &d, c->stream), ggml_op_name(n->op)); return 1;
        }
        case GGML_OP_RMS_NORM: case GGML_OP_FUSED_RMS_NORM: {
            // one decoded token: the norm rides in the prologue of the mat-mul(s) that consume it (q,k,v after attn_norm, up*gate after ffn_norm)
            static const bool mm_fusion = getenv("GGML_CDNA4_NO_MM_FUSION") == nullptr;
            if (n->op == GGML_OP_FUSED_RMS_NORM && ggml_nrows(n) > 8) { const int e = try_prompt_norm_mm(be, c, g, -1, i); if (e > 0) return e - i; }
            if (mm_fusion && c->params.fusion && !fusion_off(8) && n->op == GGML_OP_FUSED_RMS_NORM && n->src[1] && ggml_nrows(n) == 1 && n->src[0]->type == GGML_TYPE_F32 && ggml_is_contiguous(n->src[0]) &&
                n->src[1]->type == GGML_TYPE_F32 && n->ne[0] <= 8192 && n->ne[0] % 256 == 0) {
                const int j = next_real(g, i + 1); const ggml_tensor *m = j >= 0 ? g->nodes[j] : nullptr;
                if (m && m->op == GGML_OP_MUL_MAT && m->src[1] == n && ggml_is_quantized(m->src[0]->type) && m->src[0]->ne[2] == 1 && m->src[0]->ne[3] == 1 && be_supports_op(be, m)) {
                    const int cnt = mm_group_size(be, c, g, j);
                    bool plain = true; for (int q = 0; q < cnt; ++q) plain = plain && !is_r4_type(g->nodes[j + q]->src[0]->type);
                    for (int q = 0; q < cnt; ++q) plain = plain && !overlaps(n->src[0], g->nodes[j + q]);       // (an output in the memory of the un-normed row: not fusable)
                    if (plain) { const int e = try_norm_qkv_rope(c, g, n, j, cnt); if (e > 0) return e - i; }
                    if (plain && !used_from(g, j + cnt, n)) { const int done = mm_group_run(c, g, j, cnt, n); if (done > 0) return j + done - i; }
                } else if (m && m->op == GGML_OP_FUSED_UP_GATE && m->src[2] == n && !is_r4_type(m->src[0]->type) && be_supports_op(be, m) && !used_from(g, j + 1, n) &&
                           !overlaps(n->src[0], m)) {
                    const ggml_tensor *up = m->src[0], *gate = m->src[1]; const float limit = *(const float *)(m->op_params + 1);
                    const int ty = abi_type(up); (void)abi_type(gate);
                    assert_disjoint("RMS_NORM + FUSED_UP_GATE", {m}, {n->src[0], n->src[1], up, gate});
                    cdna4_fusion fx = {(const float *)n->src[1]->data, f32_param(n, 0), nullptr, nullptr};
                    const int rc = cdna4_fused_up_gate_fused(c->ctx, up->ne[1], 1, up->ne[0], m->op_params[0], ty, up->data, gate->data, up->nb[1], GGML_TYPE_F32, n->src[0]->data, n->src[0]->nb[1],
                                                             nullptr, nullptr, limit, (float *)m->data, m->nb[1] / sizeof(float), &fx, c->stream);
                    if (rc == CDNA4_OK) { ++c->n_fuse[3]; return j + 1 - i; }
                    if (rc != CDNA4_E_UNSUPPORTED) check(rc, "RMS_NORM + FUSED_UP_GATE");
                }
            }
            const cdna4_tensor x = td(n->src[0]), d = td(n); cdna4_tensor w; if (n->src[1]) w = td(n->src[1]);
            check(cdna4_op_rms_norm(c->ctx, &x, n->src[1] ? &w : nullptr, f32_param(n, 0), &d, c->stream), "RMS_NORM"); return 1;
        }
        case GGML_OP_ROPE: {
            const cdna4_tensor x = td(n->src[0]), d = td(n);
            // every layer of a graph rotates with the same angles: (cos, sin) are computed once per graph (ggml_rope_cache_init on the CPU); a model that changes the
            // parameters from layer to layer stops caching after two refills
            (void)ensure_rope_cache(c, n);
            if (c->params.fusion && !fusion_off(2)) {       // ROPE(q), ROPE(k), CPY(k -> K cache), CPY(v -> V cache): the four nodes between the QKV mat-muls and the attention
                const int j1 = next_real(g, i + 1), j2 = j1 >= 0 ? next_real(g, j1 + 1) : -1, j3 = j2 >= 0 ? next_real(g, j2 + 1) : -1;
                const ggml_tensor *rk = j1 >= 0 ? g->nodes[j1] : nullptr, *ck = j2 >= 0 ? g->nodes[j2] : nullptr, *cv = j3 >= 0 ? g->nodes[j3] : nullptr;
                if (rk && ck && cv && rk->op == GGML_OP_ROPE && ck->op == GGML_OP_CPY && cv->op == GGML_OP_CPY && ck->src[0] == rk && cv->src[0] != rk && cv->src[0] != n &&
                    memcmp(rk->op_params, n->op_params, sizeof(n->op_params)) == 0 && rk->src[1] == n->src[1] && rk->src[2] == n->src[2] && rk->ne[0] == n->ne[0] && rk->ne[2] == n->ne[2] &&
                    ck->src[1]->type == GGML_TYPE_F16 && cv->src[1]->type == GGML_TYPE_F16 && cv->src[0]->type == GGML_TYPE_F32 && rk->src[0]->type == GGML_TYPE_F32 &&
                    supports_op_impl(rk) && supports_op_impl(ck) && supports_op_impl(cv)) {
                    // the rotated K in f32 is written only when a later node reads it (the K-cache copy, its usual only reader, is part of this launch): the allocator likes to put it
                    // exactly over the un-rotated Q, which other workgroups of this launch are still reading
                    const bool kd_needed = used_from(g, j3 + 1, rk);
                    if (!fusable_layout({n, kd_needed ? rk : nullptr, ck->src[1], cv->src[1]}, {n->src[1], n->src[2], cv->src[0]}, {{n, n->src[0]}, {rk, rk->src[0]}})) goto rope_unfused;
                    const cdna4_tensor kx = td(rk->src[0]), kd = td(rk), kc = td(ck->src[1]), vx = td(cv->src[0]), vc = td(cv->src[1]);
                    assert_disjoint("ROPE + KV store", {n, kd_needed ? rk : nullptr, ck->src[1], cv->src[1]}, {n->src[1], n->src[2], cv->src[0]}, {{n, n->src[0]}, {rk, rk->src[0]}});
                    static const bool trace_rope = getenv("GGML_CDNA4_TRACE") != nullptr;
                    if (trace_rope) fprintf(stderr, "cdna4 rope+kv: q %p +%zu -> qd %p | k %p +%zu -> kd %p | v %p +%zu | kc %p +%zu vc %p +%zu | ck src %p rk %p\n", n->src[0]->data, ggml_nbytes(n->src[0]), n->data,
                                                            rk->src[0]->data, ggml_nbytes(rk->src[0]), rk->data, cv->src[0]->data, ggml_nbytes(cv->src[0]), ck->src[1]->data, ggml_nbytes(ck->src[1]), cv->src[1]->data, ggml_nbytes(cv->src[1]), ck->src[0]->data, rk->data);
                    void *const *ks = take_slot(c), *const *vs = take_slot(c);
                    check(cdna4_op_rope_store_kv(c->ctx, &x, &d, &kx, kd_needed ? &kd : nullptr, &kc, ks, &vx, &vc, vs, (const int32_t *)n->src[1]->data, n->src[2] ? (const float *)n->src[2]->data : nullptr, n->op_params[1], n->op_params[2],
                                                 n->op_params[4], f32_param(n, 5), f32_param(n, 6), f32_param(n, 7), f32_param(n, 8), f32_param(n, 9), f32_param(n, 10), c->stream), "ROPE + KV store");
                    ++c->n_fuse[1]; return j3 + 1 - i;
                }
            }
            rope_unfused:
            check(cdna4_op_rope(c->ctx, &x, (const int32_t *)n->src[1]->data, n->src[2] ? (const float *)n->src[2]->data : nullptr, &d, n->op_params[1], n->op_params[2], n->op_params[4],
                                f32_param(n, 5), f32_param(n, 6), f32_param(n, 7), f32_param(n, 8), f32_param(n, 9), f32_param(n, 10), c->stream), "ROPE"); return 1;
        }
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const cdna4_tensor a = td(n->src[0]), d = td(n->op == GGML_OP_CPY ? n->src[1] : n);
            check(cdna4_op_cpy_indirect(c->ctx, &a, &d, n->op == GGML_OP_CPY ? take_slot(c) : nullptr, c->stream), "CPY"); return 1;
        }
        case GGML_OP_GET_ROWS: { const cdna4_tensor a = td(n->src[0]), ids = td(n->src[1]), d = td(n); check(cdna4_op_get_rows(c->ctx, &a, &ids, &d, c->stream), "GET_ROWS"); return 1; }
        case GGML_OP_SOFT_MAX: {
            const cdna4_tensor x = td(n->src[0]), d = td(n); cdna4_tensor m; if (n->src[1]) m = td(n->src[1]);
            check(cdna4_op_soft_max(c->ctx, &x, n->src[1] ? &m : nullptr, &d, f32_param(n, 0), f32_param(n, 1), c->stream), "SOFT_MAX"); return 1;
        }
        case GGML_OP_FLASH_ATTN_EXT: {
            const cdna4_tensor q = td(n->src[0]), k = td(n->src[1]), v = td(n->src[2]), d = td(n); cdna4_tensor m; if (n->src[3]) m = td(n->src[3]);
            // one decoded token: FLASH_ATTN_EXT + MUL_MAT(attn_output) + ADD(residual) as ONE launch (cdna4_attn_out_fused: the attention runs on the first n_head workgroups
            // while the mat-vec workgroups already stream their weights)
            // MEASURED SLOWER and therefore opt-in (GGML_CDNA4_ATTN_FUSION=1): llama-bench tg128 of the 8B model 523.2 -> 494.7 tok/s (+3.4 us per layer) -- the tickets, the
            // polling of 256 workgroups and the weight stream's interference with the attention's loads cost more than the kernel boundary and the ramp they replace
            // (profiles/r04_notes.md; the same verdict the launch-chaining probe of round 3 reached).  Kept: bit-identical to the two launches (tests/test_gpu_attn_fused.py).
            static const bool mm_fusion = getenv("GGML_CDNA4_NO_MM_FUSION") == nullptr && getenv("GGML_CDNA4_ATTN_FUSION") != nullptr;
            if (mm_fusion && c->params.fusion && !fusion_off(128) && n->ne[2] == 1 && n->ne[3] == 1 && n->type == GGML_TYPE_F32 && ggml_is_contiguous(n)) {
                const int j1 = next_real(g, i + 1), j2 = j1 >= 0 ? next_real(g, j1 + 1) : -1;
                const ggml_tensor *mm = j1 >= 0 ? g->nodes[j1] : nullptr, *ad = j2 >= 0 ? g->nodes[j2] : nullptr;
                if (mm && ad && mm->op == GGML_OP_MUL_MAT && mm->src[1] && (mm->src[1] == n || mm->src[1]->view_src == n) && mm->src[1]->data == n->data && mm->src[1]->type == GGML_TYPE_F32 &&
                    mm->src[1]->ne[1] == 1 && mm->src[1]->ne[2] == 1 && mm->src[1]->ne[3] == 1 && mm->src[1]->ne[0] == n->ne[0] * n->ne[1] && ggml_is_contiguous(mm->src[1]) &&
                    ggml_is_quantized(mm->src[0]->type) && mm->src[0]->op == GGML_OP_NONE && mm->src[0]->ne[2] == 1 && mm->src[0]->ne[3] == 1 && !is_r4_type(mm->src[0]->type) && be_supports_op(be, mm) &&
                    mm_group_size(be, c, g, j1) == 1 && ad->op == GGML_OP_ADD && (ad->src[0] == mm || ad->src[1] == mm) && ad->type == GGML_TYPE_F32 && supports_op_impl(ad)) {
                    const ggml_tensor *w = mm->src[0], *r = ad->src[0] == mm ? ad->src[1] : ad->src[0];
                    if (r != mm && r->type == GGML_TYPE_F32 && ggml_are_same_shape(r, mm) && ggml_is_contiguous(r) && ggml_is_contiguous(ad) && ggml_is_contiguous(mm) && !used_from(g, j2 + 1, mm) &&
                        // (the mat-vec workgroups write `ad` only after EVERY attention workgroup has published: `ad` may lie over q or the mask -- the allocator does put it there --
                        //  but not over the attention row, which its sibling workgroups are still reading)
                        fusable_layout({n}, {n->src[0], n->src[1], n->src[2], n->src[3]}) && fusable_layout({ad}, {n, n->src[1], n->src[2], w}, {{ad, r}})) {
                        const int rc = cdna4_attn_out_fused(c->ctx, &q, &k, &v, n->src[3] ? &m : nullptr, &d, f32_param(n, 0), f32_param(n, 1), f32_param(n, 2), w->ne[1], w->ne[0], abi_type(w), w->data, w->nb[1],
                                                            (const float *)r->data, (float *)ad->data, c->stream);
                        if (rc == CDNA4_OK) { ++c->n_fused_attn; ++c->n_fuse[7]; return j2 + 1 - i; }
                        if (rc != CDNA4_E_UNSUPPORTED) check(rc, "FLASH_ATTN_EXT + MUL_MAT + ADD");
                    }
                }
            }
            check(cdna4_op_flash_attn(c->ctx, &q, &k, &v, n->src[3] ? &m : nullptr, &d, f32_param(n, 0), f32_param(n, 1), f32_param(n, 2), c->stream), "FLASH_ATTN_EXT"); return 1;
        }
        case GGML_OP_ARGSORT: { const cdna4_tensor x = td(n->src[0]), d = td(n); check(cdna4_op_argsort(c->ctx, &x, &d, n->op_params[0] == GGML_SORT_ORDER_DESC, c->stream), "ARGSORT"); return 1; }
        case GGML_OP_SUM_ROWS: { const cdna4_tensor x = td(n->src[0]), d = td(n); check(cdna4_op_sum_rows(c->ctx, &x, &d, c->stream), "SUM_ROWS"); return 1; }
        case GGML_OP_MUL_MULTI_ADD: {
            const cdna4_tensor a = td(n->src[0]), b = td(n->src[1]), d = td(n);
            if (c->params.fusion && !fusion_off(64)) {       // + the residual ADD that follows the experts' weighted sum (llm_build_moe_ffn -> ffn_out + ffn_inp)
                const int j = next_real(g, i + 1); const ggml_tensor *m = j >= 0 ? g->nodes[j] : nullptr;
                if (m && m->op == GGML_OP_ADD && (m->src[0] == n || m->src[1] == n) && m->type == GGML_TYPE_F32 && !used_from(g, j + 1, n)) {
                    const ggml_tensor *r = m->src[0] == n ? m->src[1] : m->src[0];
                    if (r != n && r->type == GGML_TYPE_F32 && ggml_are_same_shape(r, n) && r->nb[0] == 4 && m->nb[0] == 4 && r->ne[2] == 1 && r->ne[3] == 1 && fusable_layout({m}, {n->src[0], n->src[1]}, {{m, r}})) {
                        assert_disjoint("MUL_MULTI_ADD + ADD", {m}, {n->src[0], n->src[1]}, {{m, r}});
                        const cdna4_tensor rt = td(r), md = td(m);
                        check(cdna4_op_mul_multi_add_res(c->ctx, &a, &b, &rt, &md, c->stream), "MUL_MULTI_ADD + ADD"); ++c->n_fuse[6]; return j + 1 - i;
                    }
                }
            }
            check(cdna4_op_mul_multi_add(c->ctx, &a, &b, &d, c->stream), "MUL_MULTI_ADD"); return 1;
        }
        case GGML_OP_MUL_MAT: {     // ggml_compute_forward_mul_mat (ggml.c:17863) -> iqk_mul_mat_4d
            const ggml_tensor *w = n->src[0], *x = n->src[1];
            if (w->type == GGML_TYPE_F32 || w->type == GGML_TYPE_F16) {         // small dense weights (MoE router)
                const cdna4_tensor wt = td(w), xt = td(x), d = td(n);
                // the whole router in one launch: MUL_MAT + SOFT_MAX + ARGSORT (top-k view) + GET_ROWS + SUM_ROWS + DIV (llm_build_moe_ffn, softmax gating, normalized weights)
                if (c->params.fusion && !fusion_off(64) && w->ne[1] <= 64 && ggml_is_contiguous(n)) {
                    const int j1 = next_real(g, i + 1), j2 = j1 >= 0 ? next_real(g, j1 + 1) : -1, j3 = j2 >= 0 ? next_real(g, j2 + 1) : -1, j4 = j3 >= 0 ? next_real(g, j3 + 1) : -1, j5 = j4 >= 0 ? next_real(g, j4 + 1) : -1;
                    const ggml_tensor *sm = j1 >= 0 ? g->nodes[j1] : nullptr, *as = j2 >= 0 ? g->nodes[j2] : nullptr, *gr = j3 >= 0 ? g->nodes[j3] : nullptr, *sr = j4 >= 0 ? g->nodes[j4] : nullptr, *dv = j5 >= 0 ? g->nodes[j5] : nullptr;
                    if (sm && as && gr && sr && dv && sm->op == GGML_OP_SOFT_MAX && sm->src[0] == n && !sm->src[1] && !sm->src[2] && f32_param(sm, 0) == 1.0f && f32_param(sm, 1) == 0.0f && ggml_is_contiguous(sm) &&
                        as->op == GGML_OP_ARGSORT && as->src[0] == sm && as->op_params[0] == GGML_SORT_ORDER_DESC && as->type == GGML_TYPE_I32 && ggml_is_contiguous(as) &&
                        gr->op == GGML_OP_GET_ROWS && gr->src[0]->data == sm->data && gr->src[0]->ne[0] == 1 && gr->src[0]->ne[1] == sm->ne[0] && gr->src[1]->data == as->data && gr->src[1]->nb[1] == as->nb[1] &&
                        gr->src[1]->type == GGML_TYPE_I32 && gr->type == GGML_TYPE_F32 && gr->ne[0] == 1 &&
                        sr->op == GGML_OP_SUM_ROWS && sr->src[0]->data == gr->data && sr->src[0]->ne[0] == gr->ne[1] && sr->type == GGML_TYPE_F32 &&
                        dv->op == GGML_OP_DIV && dv->src[0]->data == gr->data && dv->src[1] == sr && dv->type == GGML_TYPE_F32 && dv->ne[0] == gr->ne[1] && dv->data != gr->data &&
                        supports_op_impl(sm) && supports_op_impl(as) && supports_op_impl(gr) && supports_op_impl(sr) && supports_op_impl(dv)) {
                        // an intermediate whose memory the allocator already gave to a LATER result of the chain is dead by then: it must not be written
                        // (workgroups of different tokens are unordered, a late "early" store would clobber the later result)
                        const ggml_tensor *chain[6] = {n, sm, as, gr, sr, dv}; cdna4_tensor tt[6];
                        for (int a = 0; a < 6; ++a) { tt[a] = td(chain[a]); for (int b = a + 1; b < 6; ++b) if (overlaps(chain[a], chain[b])) tt[a].data = nullptr; }
                        for (int a = 0; a < 6; ++a) if (tt[a].data) assert_disjoint("MoE router", {chain[a]}, {w, x});
                        const int rc = tt[2].data ? cdna4_op_moe_router(c->ctx, &wt, &xt, &tt[0], &tt[1], &tt[2], &tt[3], &tt[4], &tt[5], (int)gr->ne[1], c->stream) : CDNA4_E_UNSUPPORTED;
                        if (rc == CDNA4_OK) { ++c->n_fuse[6]; return j5 + 1 - i; }
                        if (rc != CDNA4_E_UNSUPPORTED) check(rc, "MoE router");
                    }
                }
                check(cdna4_op_mul_mat_dense(c->ctx, &wt, &xt, &d, c->stream), "MUL_MAT (dense)"); return 1;
            }
            if (x->type == GGML_TYPE_F16) {            // f16 activations: f32 copy in the scratch buffer, then the f32 mat-mul (supports_op: contiguous)
                const size_t need = (size_t)ggml_nelements(x) * sizeof(float);
                if (need > c->x32_bytes) {
                    if (t_capturing) throw capture_failed();           // (allocation inside a stream capture: this graph runs eagerly once, sized by then)
                    HIP_CHECK(hipStreamSynchronize(c->stream));
                    if (c->x32) { drop_graphs(c); HIP_CHECK(hipFree(c->x32)); }      // (a decode graph captured earlier holds the old address: it must not be replayed)
                    c->x32_bytes = (need + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1); HIP_CHECK(hipMalloc(&c->x32, c->x32_bytes));
                }
                cdna4_tensor xs = td(x), xd = td(x); xd.data = c->x32; xd.type = GGML_TYPE_F32; xd.nb[0] = sizeof(float);
                for (int d = 1; d < 4; ++d) xd.nb[d] = xd.nb[d - 1] * xd.ne[d - 1];
                check(cdna4_op_cpy_indirect(c->ctx, &xs, &xd, nullptr, c->stream), "MUL_MAT (f16 src1 -> f32)");
                check(cdna4_mul_mat_4d(c->ctx, w->ne[1], x->ne[1], w->ne[0], w->ne[2], w->ne[3], x->ne[2], x->ne[3], w->nb[2], w->nb[3], xd.nb[2], xd.nb[3],
                                       n->nb[2] / sizeof(float), n->nb[3] / sizeof(float), abi_type(w), w->data, w->nb[1], GGML_TYPE_F32, c->x32, xd.nb[1],
                                       (float *)n->data, n->nb[1] / sizeof(float), c->stream), "MUL_MAT (f16 src1)");
                return 1;
            }
            const int cnt = mm_group_size(be, c, g, i);
            // one decoded token, one matrix, followed by the residual ADD of its result (attn_output / ffn_down): C = W x + R in one launch
            static const bool mm_fusion = getenv("GGML_CDNA4_NO_MM_FUSION") == nullptr;          // (developer A/B knob)
            if (mm_fusion && c->params.fusion && !fusion_off(16) && cnt == 1 && x->ne[1] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && x->type == GGML_TYPE_F32 && w->ne[2] == 1 && w->ne[3] == 1 && !is_r4_type(w->type)) {
                const int j = next_real(g, i + 1); const ggml_tensor *ad = j >= 0 ? g->nodes[j] : nullptr;
                if (ad && ad->op == GGML_OP_ADD && (ad->src[0] == n || ad->src[1] == n) && ad->type == GGML_TYPE_F32 && supports_op_impl(ad)) {
                    const ggml_tensor *r = ad->src[0] == n ? ad->src[1] : ad->src[0];
                    if (r != n && r->type == GGML_TYPE_F32 && ggml_are_same_shape(r, n) && ggml_is_contiguous(r) && ggml_is_contiguous(ad) && ggml_is_contiguous(n) && !used_from(g, j + 1, n) && fusable_layout({ad}, {x, w}, {{ad, r}})) {
                        const long nx = w->ne[1], sa = w->nb[1], sc = ad->nb[1] / sizeof(float); const int ty = abi_type(w); const void *ap = w->data; float *cp = (float *)ad->data;
                        assert_disjoint("MUL_MAT + ADD", {ad}, {x, w}, {{ad, r}});
                        cdna4_fusion fx = {nullptr, 0.f, (const float *)r->data};
                        const int rc = cdna4_mul_mat_multi_fused(c->ctx, 1, &nx, 1, w->ne[0], &ty, &ap, &sa, x->type, x->data, x->nb[1], &cp, &sc, &fx, c->stream);
                        if (rc == CDNA4_OK) { ++c->n_fuse[4]; return j + 1 - i; }
                        if (rc != CDNA4_E_UNSUPPORTED) check(rc, "MUL_MAT + ADD");
                    }
                }
            }
            return mm_group_run(c, g, i, cnt, nullptr);
        }
        case GGML_OP_FUSED_UP_GATE: {
            const ggml_tensor *up = n->src[0], *gate = n->src[1], *x = n->src[2];
            const float limit = *(const float *)(n->op_params + 1);                      // ggml.c:18708
            const int ty = abi_type(up); (void)abi_type(gate);
            check(cdna4_fused_up_gate_ext(c->ctx, up->ne[1], x->ne[1], up->ne[0], n->op_params[0], ty, up->data, gate->data, up->nb[1], x->type, x->data, x->nb[1],
                                          nullptr, nullptr, limit, (float *)n->data, n->nb[1] / sizeof(float), c->stream), "FUSED_UP_GATE");
            return 1;
        }
        case GGML_OP_MUL_MAT_ID: {  // ids: src[2] i32 [n_used, n_tokens]; b: [K, n_b, n_tokens]; dst [M, n_used, n_tokens]
            const ggml_tensor *as = n->src[0], *b = n->src[1], *ids = n->src[2];
            check(cdna4_mul_mat_id(c->ctx, as->ne[1], as->ne[0], (int)as->ne[2], (int)ids->ne[0], b->ne[2], abi_type(as), as->data, as->nb[1], as->nb[2],
                                   (const float *)b->data, (int)b->ne[1], b->nb[1], b->nb[2], (const int32_t *)ids->data, ids->nb[1],
                                   (float *)n->data, n->nb[1] / sizeof(float), n->nb[2] / sizeof(float), c->stream), "MUL_MAT_ID");
            return 1;
        }
        case GGML_OP_MOE_FUSED_UP_GATE: {
            const ggml_tensor *up = n->src[0], *gate = n->src[1], *b = n->src[2], *ids = n->src[3], *up_b = n->src[4], *gate_b = n->src[5];
            const float limit = *(const float *)(n->op_params + 1);
            const int ty = abi_type(up); if (gate) (void)abi_type(gate);
            // merged form (gate == NULL): gate = the first ne01 / 2 rows of every expert's matrix, up = the second half; same for the bias vector
            const bool merged = gate == nullptr;
            const long nx_ff = merged ? up->ne[1] / 2 : up->ne[1];
            const void *up_w = merged ? (const void *)((const char *)up->data + up->nb[2] / 2) : up->data, *gate_w = merged ? up->data : gate->data;
            const float *up_bp = up_b ? (const float *)((const char *)up_b->data + (merged ? up_b->nb[1] / 2 : 0)) : nullptr;
            const float *gate_bp = merged ? (up_b ? (const float *)up_b->data : nullptr) : (gate_b ? (const float *)gate_b->data : nullptr);
            const long up_bs = up_b ? (long)up_b->nb[1] : 0, gate_bs = merged ? up_bs : (gate_b ? (long)gate_b->nb[1] : 0);
            // The CUDA backend consumes the FOLLOWING MUL_MAT_ID (the down projection on the fused result, same ids) in the same call for
            // decode-size batches (ggml-cuda.cu:3062-3185: up,gate,act -> re-quantise -> down with ids, two graph nodes).  Same here: one
            // C-ABI call runs the whole expert FFN block, the intermediate is the first node's own output tensor.
            const ggml_tensor *nx = (c->params.fusion && !fusion_off(64) && i + 1 < g->n_nodes) ? g->nodes[i + 1] : nullptr;
            if (nx && nx->op == GGML_OP_MUL_MAT_ID && nx->src[1] == n && nx->src[2] == ids && be_supports_op(be, nx) && b->ne[2] <= 8 && fusable_layout({n, nx}, {b, ids})) {
                const ggml_tensor *dn = nx->src[0];
                assert_disjoint("MOE_FUSED_UP_GATE + MUL_MAT_ID", {n, nx}, {b, ids, up, gate, dn, up_b, gate_b});
                check(cdna4_moe_ffn(c->ctx, nx_ff, up->ne[0], dn->ne[1], (int)up->ne[2], (int)ids->ne[0], b->ne[2], n->op_params[0], ty, up_w, gate_w, up->nb[1], up->nb[2],
                                    abi_type(dn), dn->data, dn->nb[1], dn->nb[2], (const float *)b->data, (int)b->ne[1], b->nb[1], b->nb[2], (const int32_t *)ids->data, ids->nb[1],
                                    up_bp, up_bs, gate_bp, gate_bs, limit,
                                    (float *)n->data, n->nb[1] / sizeof(float), n->nb[2] / sizeof(float), (float *)nx->data, nx->nb[1] / sizeof(float), nx->nb[2] / sizeof(float), c->stream),
                      "MOE_FUSED_UP_GATE + MUL_MAT_ID");
                ++c->n_fuse[6]; return 2;
            }
            check(cdna4_moe_fused_up_gate_ext(c->ctx, nx_ff, up->ne[0], (int)up->ne[2], (int)ids->ne[0], b->ne[2], n->op_params[0], ty, up_w, gate_w,
                                              up->nb[1], up->nb[2], (const float *)b->data, (int)b->ne[1], b->nb[1], b->nb[2], (const int32_t *)ids->data, ids->nb[1],
                                              up_bp, up_bs, gate_bp, gate_bs, limit,
                                              (float *)n->data, n->nb[1] / sizeof(float), n->nb[2] / sizeof(float), c->stream), "MOE_FUSED_UP_GATE");
            return 1;
        }
        case GGML_OP_REDUCE: {      // ggml_cuda_op_reduce (reduce.cu:125-598): src[j] = device j's partial (or, bit j of op_params[4], a copy target)
            if (n->op_params[3] == 1) return 1;                                // container only (reduce.cu:135-138)
            const int nred = n->op_params[1]; void *bufs[GGML_CUDA_MAX_DEVICES] = {nullptr}; unsigned partial = 0;
            for (int j = 0; j < nred; ++j) if (n->src[j]) { bufs[j] = n->src[j]->data; if (!((unsigned)n->op_params[4] & (1u << j))) partial |= 1u << j; }
            // order: every peer backend's queued work (its partial) before the launch, the launch before the peers' later work
            shim_context *peers[GGML_CUDA_MAX_DEVICES] = {nullptr};
            { std::lock_guard<std::mutex> lock(g_shims_mu); for (int j = 0; j < nred && j < GGML_CUDA_MAX_DEVICES; ++j) if (n->src[j] && j != c->device) peers[j] = g_shims[j]; }
            // Prompt-size messages: every participating device reduces its own 1 / N slice on its own stream (the reference's form above its small-message threshold,
            // reduce.cu:448-533) -- all partials ready before any slice starts, all slices done before any device continues.  Token-size messages: one launch here.
            static const long slice_min = getenv("GGML_CDNA4_REDUCE_SLICE_MIN") ? atol(getenv("GGML_CDNA4_REDUCE_SLICE_MIN")) : 256 * 1024;
            int ndev = 0, devs[GGML_CUDA_MAX_DEVICES]; bool all_here = true;
            for (int j = 0; j < nred; ++j) if (n->src[j]) { devs[ndev++] = j; if (j != c->device && !peers[j]) all_here = false; }
            if (all_here && ndev >= 2 && (long)ggml_nbytes(n) >= slice_min) {
                shim_context *ctxs[GGML_CUDA_MAX_DEVICES];
                for (int k = 0; k < ndev; ++k) ctxs[k] = devs[k] == c->device ? c : peers[devs[k]];
                for (int k = 0; k < ndev; ++k) { set_device(devs[k]); HIP_CHECK(hipEventRecord(ctxs[k]->ev, ctxs[k]->stream)); }
                for (int k = 0; k < ndev; ++k) { set_device(devs[k]); for (int m = 0; m < ndev; ++m) if (m != k) HIP_CHECK(hipStreamWaitEvent(ctxs[k]->stream, ctxs[m]->ev, 0)); }
                for (int k = 0; k < ndev; ++k) {
                    set_device(devs[k]);
                    check(cdna4_reduce_peers_slice(ctxs[k]->ctx, bufs, nred, partial, ggml_nelements(n), n->type, k, ndev, ctxs[k]->stream), "REDUCE (slice)");
                    HIP_CHECK(hipEventRecord(ctxs[k]->ev2, ctxs[k]->stream));
                }
                for (int k = 0; k < ndev; ++k) { set_device(devs[k]); for (int m = 0; m < ndev; ++m) if (m != k) HIP_CHECK(hipStreamWaitEvent(ctxs[k]->stream, ctxs[m]->ev2, 0)); }
                set_device(c->device);
                return 1;
            }
            for (int j = 0; j < nred; ++j) if (peers[j]) {
                set_device(j); HIP_CHECK(hipEventRecord(peers[j]->ev, peers[j]->stream));
                set_device(c->device); HIP_CHECK(hipStreamWaitEvent(c->stream, peers[j]->ev, 0));
            }
            set_device(c->device);
            check(cdna4_reduce_peers(c->ctx, bufs, nred, partial, ggml_nelements(n), n->type, c->stream), "REDUCE");
            HIP_CHECK(hipEventRecord(c->ev, c->stream));
            for (int j = 0; j < nred; ++j) if (peers[j]) { set_device(j); HIP_CHECK(hipStreamWaitEvent(peers[j]->stream, c->ev, 0)); }
            set_device(c->device);
            return 1;
        }
        default: fprintf(stderr, "ggml-hip-cdna4: op %s reached graph_compute (supports_op is false for it)\n", ggml_op_name(n->op)); return -1;
    }
}
// GGML_CDNA4_CHECK_REPRO=<R> (debug switch, eager walks only; scripts/soak_logits.py --bisect): every call of the walk -- a single node or a fused group -- is issued R more
// times and must write the same bytes every time.  Localizes a kernel whose result depends on the order in which its workgroups run (a race inside ONE launch) to the node.
// Groups that write over one of their own inputs (in-place element-wise nodes) cannot be repeated and are skipped.
static void check_repro(ggml_backend_t be, shim_context *c, ggml_cgraph *g, int i, int k, int reps) {
    std::vector<const ggml_tensor *> outs;
    for (int q = i; q < i + k; ++q) { const ggml_tensor *n = g->nodes[q]; if (node_is_noop(n) || !n->data) continue; outs.push_back(n->op == GGML_OP_CPY ? n->src[1] : n); }
    for (int q = i; q < i + k; ++q) { const ggml_tensor *n = g->nodes[q]; if (node_is_noop(n)) continue;
        for (int s = 0; s < GGML_MAX_SRC; ++s) { const ggml_tensor *x = n->src[s]; if (!x || !x->data) continue; if (n->op == GGML_OP_CPY && s == 1) continue;
            bool internal = false; for (int r = i; r < q; ++r) if (g->nodes[r] == x || (x->view_src && g->nodes[r] == x->view_src)) internal = true;
            if (internal) continue;
            for (const ggml_tensor *o : outs) if (overlaps(o, x)) return; } }           // in place: not repeatable
    if (outs.empty()) return;
    auto snapshot = [&](std::vector<std::vector<uint8_t>> &dst) {
        HIP_CHECK(hipStreamSynchronize(c->stream)); dst.resize(outs.size());
        for (size_t o = 0; o < outs.size(); ++o) { dst[o].resize(ggml_nbytes(outs[o])); HIP_CHECK(hipMemcpy(dst[o].data(), outs[o]->data, dst[o].size(), hipMemcpyDeviceToHost)); } };
    std::vector<std::vector<uint8_t>> first, again; snapshot(first);
    static long n_checked = 0, n_bad = 0;
    for (int r = 0; r < reps; ++r) {
        const int k2 = compute_node(be, c, g, i); if (k2 != k) GGML_ABORT("ggml-hip-cdna4: repro check: the walk consumed %d nodes, then %d", k, k2);
        snapshot(again); ++n_checked;
        for (size_t o = 0; o < outs.size(); ++o) if (memcmp(first[o].data(), again[o].data(), first[o].size()) != 0) {
            size_t nd = 0, firstd = 0; for (size_t b = 0; b < first[o].size(); ++b) if (first[o][b] != again[o][b]) { if (!nd) firstd = b; ++nd; }
            const ggml_tensor *n = g->nodes[i];
            fprintf(stderr, "cdna4 REPRO MISMATCH #%ld (of %ld repeats): node %d %s '%s' (+%d fused) result '%s' %s [%ld,%ld,%ld,%ld]: %zu of %zu bytes differ from byte %zu, repeat %d; src0 %s %s [%ld,%ld,%ld] src1 [%ld,%ld,%ld]\n",
                    ++n_bad, n_checked, i, ggml_op_name(n->op), n->name, k - 1, outs[o]->name, ggml_type_name(outs[o]->type), (long)outs[o]->ne[0], (long)outs[o]->ne[1], (long)outs[o]->ne[2], (long)outs[o]->ne[3],
                    nd, first[o].size(), firstd, r, n->src[0] ? n->src[0]->name : "-", n->src[0] ? ggml_type_name(n->src[0]->type) : "-", n->src[0] ? (long)n->src[0]->ne[0] : 0, n->src[0] ? (long)n->src[0]->ne[1] : 0,
                    n->src[0] ? (long)n->src[0]->ne[2] : 0, n->src[1] ? (long)n->src[1]->ne[0] : 0, n->src[1] ? (long)n->src[1]->ne[1] : 0, n->src[1] ? (long)n->src[1]->ne[2] : 0);
        }
    }
}
static enum ggml_status run_nodes(ggml_backend_t be, shim_context *c, ggml_cgraph *g) {
    static const bool trace = getenv("GGML_CDNA4_TRACE") != nullptr;
    static const int repro = getenv("GGML_CDNA4_CHECK_REPRO") ? atoi(getenv("GGML_CDNA4_CHECK_REPRO")) : 0;
    (void)cdna4_op_rope_cache_reset(c->ctx); c->rope_pos = nullptr; c->rope_fills = 0;
    index_uses(g);
    struct unindex { ~unindex() { t_uses.g = nullptr; } } unindex_at_exit;          // (the graph object may be rebuilt in place before the next walk)
    for (int i = 0; i < g->n_nodes;) {
        if (trace && !node_is_noop(g->nodes[i])) { const ggml_tensor *n = g->nodes[i]; fprintf(stderr, "cdna4[%d] %s %s [%ld,%ld,%ld,%ld] src0 %s %s [%ld,%ld,%ld] nb1 %zu src1 [%ld,%ld,%ld] nb1 %zu\n", c->device, ggml_op_name(n->op), n->name,
            (long)n->ne[0], (long)n->ne[1], (long)n->ne[2], (long)n->ne[3], n->src[0] ? n->src[0]->name : "-", n->src[0] ? ggml_type_name(n->src[0]->type) : "-", n->src[0] ? (long)n->src[0]->ne[0] : 0, n->src[0] ? (long)n->src[0]->ne[1] : 0, n->src[0] ? (long)n->src[0]->ne[2] : 0,
            n->src[0] ? n->src[0]->nb[1] : 0, n->src[1] ? (long)n->src[1]->ne[0] : 0, n->src[1] ? (long)n->src[1]->ne[1] : 0, n->src[1] ? (long)n->src[1]->ne[2] : 0, n->src[1] ? n->src[1]->nb[1] : 0); }
        const int k = compute_node(be, c, g, i); if (k < 0) return GGML_STATUS_FAILED;
        if (repro > 0 && !c->capturing) check_repro(be, c, g, i, k, repro);
        i += k;
    }
    return GGML_STATUS_SUCCESS;
}

static bool node_is_cache_write(const ggml_tensor *n) { return n->op == GGML_OP_CPY; }
// fills the pinned slot table with the current destinations of the graph's cache-write nodes; returns their count (-1: too many)
static int fill_slots(shim_context *c, const ggml_cgraph *g) {
    if (c->slots_busy) { HIP_CHECK(hipEventSynchronize(c->slots_ev)); c->slots_busy = false; }      // the previous replay's H2D copy must have read the table
    int n = 0;
    for (int i = 0; i < g->n_nodes; ++i) if (node_is_cache_write(g->nodes[i])) { if (n >= shim_context::MAX_SLOTS) return -1; c->slots_host[n++] = g->nodes[i]->src[1]->data; }
    return n;
}
// GGML_CDNA4_CHECK_SLOTS=1 (debug switch, scripts/soak_logits.py): the captured H2D copy of the slot table is bracketed -- the device table holds a canary before the graph is
// launched, and after the launch (stream synchronized) it must hold exactly the host table of THIS call: a replay whose copy node read a stale or half-written host table,
// or did not run in front of the kernels, aborts here instead of writing K / V rows to the wrong place.
static void launch_graph(shim_context *c, hipGraphExec_t exec, int n_slots) {
    static const bool chk = getenv("GGML_CDNA4_CHECK_SLOTS") != nullptr;
    if (chk && n_slots > 0) {
        std::vector<void *> canary((size_t)n_slots, (void *)(uintptr_t)0xdeadbeefdeadbeefull);
        HIP_CHECK(hipStreamSynchronize(c->stream)); HIP_CHECK(hipMemcpy(c->slots_dev, canary.data(), sizeof(void *) * (size_t)n_slots, hipMemcpyHostToDevice));
    }
    HIP_CHECK(hipGraphLaunch(exec, c->stream));
    if (n_slots > 0) { HIP_CHECK(hipEventRecord(c->slots_ev, c->stream)); c->slots_busy = true; }
    if (chk && n_slots > 0) {
        std::vector<void *> seen((size_t)n_slots);
        HIP_CHECK(hipStreamSynchronize(c->stream)); HIP_CHECK(hipMemcpy(seen.data(), c->slots_dev, sizeof(void *) * (size_t)n_slots, hipMemcpyDeviceToHost));
        for (int i = 0; i < n_slots; ++i) if (seen[(size_t)i] != c->slots_host[i]) GGML_ABORT("ggml-hip-cdna4: slot table check: device slot %d holds %p, this call's host table %p", i, seen[(size_t)i], c->slots_host[i]);
    }
}
static enum ggml_status graph_compute_impl(ggml_backend_t be, ggml_cgraph *g);
static GGML_CALL enum ggml_status be_graph_compute(ggml_backend_t be, ggml_cgraph *g) {
    auto *c = (shim_context *)be->context; const double t0 = now_s();
    const enum ggml_status st = graph_compute_impl(be, g);
    c->t_compute += now_s() - t0; return st;
}
static enum ggml_status graph_compute_impl(ggml_backend_t be, ggml_cgraph *g) {
    auto *c = (shim_context *)be->context; set_device(c->device);
    // HIP graph: worth it from a handful of launches on; not with REDUCE nodes (cross-device event ordering is done on the host).
    int n_real = 0; bool capturable = c->params.use_graphs && c->slots_host;
    // Prompt-size batches run eagerly: a real prompt never repeats a graph (every ubatch sees a longer KV window), and where one does repeat -- llama-bench's pp repetitions --
    // the capture + instantiate of ~450 launches cost the second pass 6 ms and the first replay 1.3 ms of a 15 ms pass, for a steady state the eager walk reaches as well
    // (the host stays ahead of the GPU by itself at 30 us per launch).  GGML_CDNA4_GRAPH_MAX_BATCH=<rows> moves the limit (default 8: decode and small speculative batches).
    static const long max_batch = getenv("GGML_CDNA4_GRAPH_MAX_BATCH") ? atol(getenv("GGML_CDNA4_GRAPH_MAX_BATCH")) : 8;
    for (int i = 0; i < g->n_nodes && capturable; ++i) { const ggml_tensor *n = g->nodes[i]; if (node_is_noop(n)) continue; ++n_real; if (n->op == GGML_OP_REDUCE) capturable = false;
                                                         if ((n->op == GGML_OP_MUL_MAT || n->op == GGML_OP_FUSED_UP_GATE) && n->ne[1] > max_batch) capturable = false; }      // (ne[1] = tokens, also for the K.Q / V.P products of a graph without flash attention)
    if (!capturable || n_real < 8) { ++c->n_small; return run_nodes(be, c, g); }
    auto key_node = [](const ggml_tensor *n, graph_key::node &k) {
        memset(&k, 0, sizeof(k));
        const bool cw = node_is_cache_write(n);
        k.op = n->op; k.type = n->type; k.data = cw ? nullptr : n->data;
        for (int j = 0; j < 6; ++j) if (n->src[j]) {
            k.src[j] = cw && j == 1 ? nullptr : n->src[j]->data; k.src_type[j] = n->src[j]->type;
            for (int d = 0; d < 4; ++d) { k.src_ne[j][d] = n->src[j]->ne[d]; k.src_nb[j][d] = (int64_t)n->src[j]->nb[d]; }
        }
        for (int d = 0; d < 4; ++d) { k.ne[d] = n->ne[d]; k.nb[d] = (int64_t)n->nb[d]; }
        static_assert(sizeof(k.params) == sizeof(n->op_params), "graph key: op_params"); memcpy(k.params, n->op_params, sizeof(k.params));
    };
    cached_graph *cg = nullptr;
    // a decode loop presents the graph it presented last time: compare node by node against that entry while the key is derived (no 0.5 MB vector to allocate, fill and
    // compare -- an mmap / munmap pair per token); any difference falls through to the full key and the search over all entries
    if (c->last_graph >= 0 && c->last_graph < (int)c->graphs.size() && (int)c->graphs[c->last_graph].key.nodes.size() == n_real) {
        const graph_key::node *ref = c->graphs[c->last_graph].key.nodes.data(); bool same = true; int idx = 0; graph_key::node k;
        for (int i = 0; i < g->n_nodes && same; ++i) {
            const ggml_tensor *n = g->nodes[i]; if (node_is_noop(n)) continue;
            key_node(n, k); same = memcmp(&k, ref + idx++, sizeof(k)) == 0;
        }
        if (same) cg = &c->graphs[c->last_graph];
    }
    graph_key key;
    if (!cg) {
        key.nodes.reserve(n_real);
        for (int i = 0; i < g->n_nodes; ++i) {
            const ggml_tensor *n = g->nodes[i]; if (node_is_noop(n)) continue;
            graph_key::node k; key_node(n, k); key.nodes.push_back(k);
        }
        for (auto &e : c->graphs) if (e.key == key) { cg = &e; break; }
    }
    c->last_graph = cg ? (int)(cg - c->graphs.data()) : -1;
    if (!cg) {                                                  // first sighting: run eagerly (sizes the workspace, re-tiles late _R4 uploads)
        if (c->graphs.size() >= 8) { if (c->graphs.front().exec) (void)hipGraphExecDestroy(c->graphs.front().exec); c->graphs.erase(c->graphs.begin()); }
        c->graphs.push_back({key, nullptr, 1, false, -1}); c->last_graph = (int)c->graphs.size() - 1;
        ++c->n_eager; return run_nodes(be, c, g);
    }
    if (cg->failed) { ++c->n_eager; return run_nodes(be, c, g); }
    const int n_slots = fill_slots(c, g);
    if (n_slots < 0) { cg->failed = true; return run_nodes(be, c, g); }
    if (cg->exec && cg->ws_epoch != cdna4_workspace_epoch(c->ctx)) { (void)hipGraphExecDestroy(cg->exec); cg->exec = nullptr; }      // the workspace moved since the capture: capture again
    if (cg->exec) {
        ++c->n_replayed;
        launch_graph(c, cg->exec, n_slots);
        return GGML_STATUS_SUCCESS;
    }
    ++cg->seen;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); cg->failed = true; return run_nodes(be, c, g); }
    c->capturing = true; c->slot_next = 0;
    hipError_t e = n_slots > 0 ? hipMemcpyAsync(c->slots_dev, c->slots_host, sizeof(void *) * (size_t)n_slots, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    enum ggml_status st = GGML_STATUS_FAILED;
    t_capturing = true;
    try { if (e == hipSuccess) st = run_nodes(be, c, g); } catch (const capture_failed &) { st = GGML_STATUS_FAILED; }
    t_capturing = false;
    c->capturing = false;
    const bool slots_ok = c->slot_next == n_slots;              // every cache-write node took exactly one slot, in node order
    e = hipStreamEndCapture(c->stream, &graph);
    if (st != GGML_STATUS_SUCCESS || e != hipSuccess || !graph || !slots_ok) { (void)hipGetLastError(); cg->failed = true; ++c->n_capture_failed; if (graph) (void)hipGraphDestroy(graph); return run_nodes(be, c, g); }
    if (hipGraphInstantiate(&cg->exec, graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); cg->exec = nullptr; cg->failed = true; (void)hipGraphDestroy(graph); return run_nodes(be, c, g); }
    (void)hipGraphDestroy(graph); ++c->n_captured; cg->ws_epoch = cdna4_workspace_epoch(c->ctx);
    launch_graph(c, cg->exec, n_slots);
    return GGML_STATUS_SUCCESS;
}
static void drop_graphs(shim_context *c) { for (auto &e : c->graphs) if (e.exec) (void)hipGraphExecDestroy(e.exec); c->graphs.clear(); c->last_graph = -1; }

static GGML_CALL const char *be_name(ggml_backend_t be) { return ((shim_context *)be->context)->name.c_str(); }
static GGML_CALL void be_free(ggml_backend_t be) {
    auto *c = (shim_context *)be->context; set_device(c->device);
    { std::lock_guard<std::mutex> lock(g_shims_mu); if (c->device < GGML_CUDA_MAX_DEVICES && g_shims[c->device] == c) g_shims[c->device] = nullptr; }
    (void)hipStreamSynchronize(c->stream); drop_graphs(c);
    if (getenv("GGML_CDNA4_STATS") && c->device < GGML_CUDA_MAX_DEVICES) fprintf(stderr, "cdna4[%d] small uploads queued on the compute stream instead of blocking copies: %ld\n", c->device, g_stage[c->device].n_staged);
    stage_detach(c->device, c->stream);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] graph_compute calls: %ld eager, %ld captured, %ld replayed, %ld capture failures, %ld too small / not capturable; fused attention + attn_output launches issued or captured: %ld\n", c->device, c->n_eager, c->n_captured, c->n_replayed, c->n_capture_failed, c->n_small, c->n_fused_attn);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] fused launches issued or captured: ADD+RMS_NORM %ld, ROPE+ROPE+KV stores %ld, shared-input MUL_MATs %ld, RMS_NORM in mat-mul %ld, MUL_MAT+ADD %ld, "
                                            "RMS_NORM+q,k,v+ROPE+KV store %ld, MoE blocks %ld, attention+attn_output %ld\n", c->device, c->n_fuse[0], c->n_fuse[1], c->n_fuse[2], c->n_fuse[3], c->n_fuse[4], c->n_fuse[5], c->n_fuse[6], c->n_fuse[7]);
    if (getenv("GGML_CDNA4_STATS")) fprintf(stderr, "cdna4[%d] host time: graph_compute %.1f ms, synchronize %.1f ms (%ld calls), set_async %.1f ms (%ld calls, %.1f MB), get_async %.1f ms (%ld calls, %.1f MB)\n", c->device,
                                            c->t_compute * 1e3, c->t_sync * 1e3, c->n_sync, c->t_set * 1e3, c->n_set, c->b_set / 1e6, c->t_get * 1e3, c->n_get, c->b_get / 1e6);
    if (c->x32) (void)hipFree(c->x32);
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
