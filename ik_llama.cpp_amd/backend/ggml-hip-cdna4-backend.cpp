// ggml-hip-cdna4-backend.cpp -- the ggml-backend shim: makes libggml-hip-cdna4.so a drop-in for ik_llama.cpp's CUDA backend.
//
// The reference binds its GPU backend at COMPILE time through the 13 `ggml_backend_cuda_*` C symbols of
// ggml/include/ggml-cuda.h:24-47 (+ ggml_backend_cuda_reg_devices, ggml-backend.cpp:456-489) and the three vtables of
// ggml/src/ggml-backend-impl.h:18-130.  This file exports exactly those symbols and fills those vtables; every
// op is forwarded to the C ABI of include/ggml_hip_cdna4.h.  It is compiled against the reference's own headers (which are NOT
// copied into this repo), so it only builds where a reference checkout exists (`make REF=/path/to/ik_llama.cpp`).
//
// supports_op is true only for the hot path (MUL_MAT / MUL_MAT_ID / FUSED_UP_GATE / MOE_FUSED_UP_GATE on the supported quant
// types with f32 activations); every other op stays on whichever backend owns it (ggml-backend.cpp:1314-1360 scheduler rule).
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-cuda.h"
#include "ggml_hip_cdna4.h"

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#define SHIM_MAX_DEVICES GGML_CUDA_MAX_DEVICES
#define MATRIX_ROW_PADDING 512          // ggml-cuda/common.cuh:63 -- quantized rows are over-allocated like the CUDA backend does

static ggml_log_callback g_log_cb = nullptr; static void *g_log_ud = nullptr;
static void shim_log(enum ggml_log_level lvl, const char *fmt, const char *a = "") {
    char buf[512]; snprintf(buf, sizeof(buf), fmt, a);
    if (g_log_cb) g_log_cb(lvl, buf, g_log_ud); else fputs(buf, stderr);
}
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "ggml-hip-cdna4: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); GGML_ABORT("HIP error"); } } while (0)

struct shim_context { int device; cdna4_context *ctx; hipStream_t stream; std::string name; hipEvent_t ev = nullptr; };
// device -> most recent backend of this process: the REDUCE node runs on ONE backend and orders every peer's stream around its launch
// (the reference keeps the same kind of map, model -> ctx[device]: ggml-cuda/common.cuh:765, reduce.cu:140-145)
static shim_context *g_shims[GGML_CUDA_MAX_DEVICES] = {nullptr};

// ---------------------------------------------------------------------------------------------- device buffer
struct shim_buffer_ctx { int device; void *base; };
struct shim_buft_ctx { int device; std::string name; };

static GGML_CALL const char *buf_get_name(ggml_backend_buffer_t b) { return ((shim_buft_ctx *)b->buft->context)->name.c_str(); }
static GGML_CALL void buf_free(ggml_backend_buffer_t b) { auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipFree(c->base)); delete c; }
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
        if (padded > orig) { auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemset((char *)t->data + orig, 0, padded - orig)); }
    }
}
static GGML_CALL void buf_memset_tensor(ggml_backend_buffer_t b, ggml_tensor *t, uint8_t v, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemset((char *)t->data + off, v, size)); HIP_CHECK(hipDeviceSynchronize());
}
static GGML_CALL void buf_set_tensor(ggml_backend_buffer_t b, ggml_tensor *t, const void *data, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpy((char *)t->data + off, data, size, hipMemcpyHostToDevice));
}
static GGML_CALL void buf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor *t, void *data, size_t off, size_t size) {
    auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpy(data, (const char *)t->data + off, size, hipMemcpyDeviceToHost));
}
static GGML_CALL bool buf_cpy_tensor(ggml_backend_buffer_t b, const ggml_tensor *src, ggml_tensor *dst);
static GGML_CALL void buf_clear(ggml_backend_buffer_t b, uint8_t v) { auto *c = (shim_buffer_ctx *)b->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemset(c->base, v, b->size)); HIP_CHECK(hipDeviceSynchronize()); }
static const ggml_backend_buffer_i k_buffer_iface = { buf_get_name, buf_free, buf_get_base, buf_init_tensor, buf_memset_tensor, buf_set_tensor, buf_get_tensor, buf_cpy_tensor, buf_clear, nullptr };
static bool buffer_is_ours(ggml_backend_buffer_t b) { return b && b->iface.get_name == buf_get_name; }
static GGML_CALL bool buf_cpy_tensor(ggml_backend_buffer_t, const ggml_tensor *src, ggml_tensor *dst) {
    if (!buffer_is_ours(src->buffer)) return false;
    HIP_CHECK(hipMemcpy(dst->data, src->data, ggml_nbytes(src), hipMemcpyDeviceToDevice));      // same or peer device
    return true;
}

static GGML_CALL const char *buft_get_name(ggml_backend_buffer_type_t t) { return ((shim_buft_ctx *)t->context)->name.c_str(); }
static GGML_CALL ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    auto *bc = (shim_buft_ctx *)t->context; HIP_CHECK(hipSetDevice(bc->device));
    size = size ? size : 1; void *p = nullptr;
    if (hipMalloc(&p, size) != hipSuccess) { (void)hipGetLastError(); shim_log(GGML_LOG_LEVEL_ERROR, "ggml-hip-cdna4: device allocation failed%s\n"); return nullptr; }
    return ggml_backend_buffer_init(t, k_buffer_iface, new shim_buffer_ctx{bc->device, p}, size);
}
static GGML_CALL size_t buft_alignment(ggml_backend_buffer_type_t) { return 128; }
static GGML_CALL size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor *t) { return padded_nbytes(t); }     // ggml-cuda.cu:754-767
static GGML_CALL bool buft_is_host(ggml_backend_buffer_type_t) { return false; }

extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_buffer_type(int device) {
    static std::mutex mu; static ggml_backend_buffer_type types[SHIM_MAX_DEVICES]; static bool init = false;
    std::lock_guard<std::mutex> lock(mu);
    if (device < 0 || device >= cdna4_get_device_count() || device >= SHIM_MAX_DEVICES) return nullptr;
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
    if (hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return ggml_backend_buft_alloc_buffer(ggml_backend_cpu_buffer_type(), size); }
    ggml_backend_buffer_t b = ggml_backend_cpu_buffer_from_ptr(p, size);
    b->buft = t; b->iface.free_buffer = host_buf_free;
    return b;
}
extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_host_buffer_type(void) {
    static ggml_backend_buffer_type t = { { host_buft_name, host_buft_alloc, ggml_backend_cpu_buffer_type()->iface.get_alignment, nullptr,
                                            ggml_backend_cpu_buffer_type()->iface.get_alloc_size, ggml_backend_cpu_buffer_type()->iface.is_host }, nullptr };
    return &t;
}
// One process drives one GPU in this backend: the single-process multi-GPU split buffer does not exist.
extern "C" GGML_CALL ggml_backend_buffer_type_t ggml_backend_cuda_split_buffer_type(const float *) { return nullptr; }

// ---------------------------------------------------------------------------------------------- backend
static ggml_guid_t shim_guid() { static ggml_guid g = {0xc4, 0xd1, 0x4a, 0x04, 0x95, 0x0f, 0x11, 0xee, 0x9a, 0x33, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30}; return &g; }

static bool mm_types_ok(const ggml_tensor *w, const ggml_tensor *x, const ggml_tensor *dst) {
    return cdna4_type_supported(w->type) && x->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32 &&
           w->nb[0] == ggml_type_size(w->type) && x->nb[0] == sizeof(float) && dst->nb[0] == sizeof(float) &&
           w->ne[0] % 64 == 0 && !ggml_is_transposed(w) && !ggml_is_transposed(x);
}
static bool up_gate_unary_ok(int u) { return u == GGML_UNARY_OP_SILU || u == GGML_UNARY_OP_GELU || u == GGML_UNARY_OP_RELU || u == GGML_UNARY_OP_SWIGLU_OAI; }
// per-expert bias [M, n_expert] f32 (ggml_moe_up_gate_ext, ggml.c:8066-8080)
static bool bias_ok(const ggml_tensor *b, const ggml_tensor *w) { return !b || (b->type == GGML_TYPE_F32 && b->nb[0] == sizeof(float) && b->ne[0] == w->ne[1]); }
static GGML_CALL bool be_supports_op(ggml_backend_t, const ggml_tensor *op) {
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return true;
        case GGML_OP_MUL_MAT: return mm_types_ok(op->src[0], op->src[1], op) && op->src[1]->ne[2] % op->src[0]->ne[2] == 0 && op->src[1]->ne[3] % op->src[0]->ne[3] == 0;
        case GGML_OP_MUL_MAT_ID: return mm_types_ok(op->src[0], op->src[1], op) && op->src[2]->type == GGML_TYPE_I32 && op->src[1]->ne[3] == 1 &&
                                        (op->src[0]->type < GGML_TYPE_Q4_0_R8 || (size_t)op->src[0]->nb[2] == (size_t)op->src[0]->ne[1] * op->src[0]->nb[1]);   // (_R4 experts: contiguous only)
        case GGML_OP_FUSED_UP_GATE: {
            return op->src[1] && op->src[0]->type == op->src[1]->type && ggml_are_same_shape(op->src[0], op->src[1]) && mm_types_ok(op->src[0], op->src[2], op) &&
                   op->src[2]->ne[2] == 1 && op->src[2]->ne[3] == 1 && up_gate_unary_ok(op->op_params[0]);
        }
        case GGML_OP_MOE_FUSED_UP_GATE: {   // (the merged up+gate single-tensor form, src[1] == NULL, is left to the CPU backend)
            return op->src[1] && op->src[0]->type == op->src[1]->type && mm_types_ok(op->src[0], op->src[2], op) && op->src[3] && op->src[3]->type == GGML_TYPE_I32 &&
                   bias_ok(op->src[4], op->src[0]) && bias_ok(op->src[5], op->src[0]) && up_gate_unary_ok(op->op_params[0]) &&
                   (op->src[0]->type < GGML_TYPE_Q4_0_R8 || (size_t)op->src[0]->nb[2] == (size_t)op->src[0]->ne[1] * op->src[0]->nb[1]);
        }
        case GGML_OP_REDUCE:                 // reduce.cu:125-134 (Q8_0 partial sums: left to the reference path)
            return op->op_params[0] == GGML_OP_ADD && (op->type == GGML_TYPE_F32 || op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_BF16) && ggml_is_contiguous(op) &&
                   op->op_params[1] >= 1 && op->op_params[1] <= GGML_CUDA_MAX_DEVICES;
        default: return false;
    }
}
static void check(int rc, const char *what) { if (rc != CDNA4_OK) { fprintf(stderr, "ggml-hip-cdna4: %s: %s\n", what, cdna4_last_error()); GGML_ABORT("cdna4 op failed"); } }

static GGML_CALL enum ggml_status be_graph_compute(ggml_backend_t be, ggml_cgraph *g) {
    auto *c = (shim_context *)be->context; HIP_CHECK(hipSetDevice(c->device));
    for (int i = 0; i < g->n_nodes; ++i) {
        ggml_tensor *n = g->nodes[i];
        switch (n->op) {
            case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_MUL_MAT: {     // ggml_compute_forward_mul_mat (ggml.c:17863) -> iqk_mul_mat_4d
                const ggml_tensor *w = n->src[0], *x = n->src[1];
                // Consecutive MUL_MATs of leaf weights sharing src1 (q,k,v) go out as one call, like ggml.c:17984-18000 /
                // ggml-cuda.cu:2570-2600: same-type matrices (and a K-quant group + a Q6_K matrix) become ONE decode launch.
                auto plain2d = [](const ggml_tensor *t) { return t->ne[2] == 1 && t->ne[3] == 1; };
                int cnt = 1;
                if (plain2d(w) && plain2d(x) && w->op == GGML_OP_NONE) {
                    while (i + cnt < g->n_nodes && cnt < 5) {
                        const ggml_tensor *m = g->nodes[i + cnt];
                        if (m->op != GGML_OP_MUL_MAT || m->src[1] != x || m->src[0]->op != GGML_OP_NONE || !plain2d(m->src[0]) || !be_supports_op(be, m) ||
                            m->src[0]->ne[0] != w->ne[0]) break;
                        ++cnt;
                    }
                }
                if (cnt > 1) {
                    long nx[5], sa[5], sc[5]; int ty[5]; const void *ap[5]; float *cp[5];
                    for (int j = 0; j < cnt; ++j) {
                        const ggml_tensor *m = g->nodes[i + j];
                        nx[j] = m->src[0]->ne[1]; sa[j] = m->src[0]->nb[1]; sc[j] = m->nb[1] / sizeof(float); ty[j] = m->src[0]->type; ap[j] = m->src[0]->data; cp[j] = (float *)m->data;
                    }
                    check(cdna4_mul_mat_multi(c->ctx, cnt, nx, x->ne[1], w->ne[0], ty, ap, sa, x->type, x->data, x->nb[1], cp, sc, c->stream), "MUL_MAT (fused, shared src1)");
                    i += cnt - 1;
                    break;
                }
                check(cdna4_mul_mat_4d(c->ctx, w->ne[1], x->ne[1], w->ne[0], w->ne[2], w->ne[3], x->ne[2], x->ne[3], w->nb[2], w->nb[3], x->nb[2], x->nb[3],
                                       n->nb[2] / sizeof(float), n->nb[3] / sizeof(float), w->type, w->data, w->nb[1], x->type, x->data, x->nb[1],
                                       (float *)n->data, n->nb[1] / sizeof(float), c->stream), "MUL_MAT");
            } break;
            case GGML_OP_FUSED_UP_GATE: {
                const ggml_tensor *up = n->src[0], *gate = n->src[1], *x = n->src[2];
                const float limit = *(const float *)(n->op_params + 1);                      // ggml.c:18708
                check(cdna4_fused_up_gate_ext(c->ctx, up->ne[1], x->ne[1], up->ne[0], n->op_params[0], up->type, up->data, gate->data, up->nb[1], x->type, x->data, x->nb[1],
                                              nullptr, nullptr, limit, (float *)n->data, n->nb[1] / sizeof(float), c->stream), "FUSED_UP_GATE");
            } break;
            case GGML_OP_MUL_MAT_ID: {  // ids: src[2] i32 [n_used, n_tokens]; b: [K, n_b, n_tokens]; dst [M, n_used, n_tokens]
                const ggml_tensor *as = n->src[0], *b = n->src[1], *ids = n->src[2];
                check(cdna4_mul_mat_id(c->ctx, as->ne[1], as->ne[0], (int)as->ne[2], (int)ids->ne[0], b->ne[2], as->type, as->data, as->nb[1], as->nb[2],
                                       (const float *)b->data, (int)b->ne[1], b->nb[1], b->nb[2], (const int32_t *)ids->data, ids->nb[1],
                                       (float *)n->data, n->nb[1] / sizeof(float), n->nb[2] / sizeof(float), c->stream), "MUL_MAT_ID");
            } break;
            case GGML_OP_MOE_FUSED_UP_GATE: {
                const ggml_tensor *up = n->src[0], *gate = n->src[1], *b = n->src[2], *ids = n->src[3], *up_b = n->src[4], *gate_b = n->src[5];
                const float limit = *(const float *)(n->op_params + 1);
                check(cdna4_moe_fused_up_gate_ext(c->ctx, up->ne[1], up->ne[0], (int)up->ne[2], (int)ids->ne[0], b->ne[2], n->op_params[0], up->type, up->data, gate->data,
                                                  up->nb[1], up->nb[2], (const float *)b->data, (int)b->ne[1], b->nb[1], b->nb[2], (const int32_t *)ids->data, ids->nb[1],
                                                  up_b ? (const float *)up_b->data : nullptr, up_b ? (long)up_b->nb[1] : 0,
                                                  gate_b ? (const float *)gate_b->data : nullptr, gate_b ? (long)gate_b->nb[1] : 0, limit,
                                                  (float *)n->data, n->nb[1] / sizeof(float), n->nb[2] / sizeof(float), c->stream), "MOE_FUSED_UP_GATE");
            } break;
            case GGML_OP_REDUCE: {      // ggml_cuda_op_reduce (reduce.cu:125-598): src[j] = device j's partial (or, bit j of op_params[4], a copy target)
                if (n->op_params[3] == 1) break;                                   // container only (reduce.cu:135-138)
                const int nred = n->op_params[1]; void *bufs[GGML_CUDA_MAX_DEVICES] = {nullptr}; unsigned partial = 0;
                for (int j = 0; j < nred; ++j) if (n->src[j]) { bufs[j] = n->src[j]->data; if (!((unsigned)n->op_params[4] & (1u << j))) partial |= 1u << j; }
                // order: every peer backend's queued work (its partial) before the launch, the launch before the peers' later work
                for (int j = 0; j < nred; ++j) if (n->src[j] && j != c->device && j < GGML_CUDA_MAX_DEVICES && g_shims[j]) {
                    HIP_CHECK(hipSetDevice(j)); HIP_CHECK(hipEventRecord(g_shims[j]->ev, g_shims[j]->stream));
                    HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipStreamWaitEvent(c->stream, g_shims[j]->ev, 0));
                }
                HIP_CHECK(hipSetDevice(c->device));
                check(cdna4_reduce_peers(c->ctx, bufs, nred, partial, ggml_nelements(n), n->type, c->stream), "REDUCE");
                HIP_CHECK(hipEventRecord(c->ev, c->stream));
                for (int j = 0; j < nred; ++j) if (n->src[j] && j != c->device && j < GGML_CUDA_MAX_DEVICES && g_shims[j]) {
                    HIP_CHECK(hipSetDevice(j)); HIP_CHECK(hipStreamWaitEvent(g_shims[j]->stream, c->ev, 0));
                }
                HIP_CHECK(hipSetDevice(c->device));
            } break;
            default: fprintf(stderr, "ggml-hip-cdna4: op %s reached graph_compute (supports_op is false for it)\n", ggml_op_name(n->op)); return GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
}

static GGML_CALL const char *be_name(ggml_backend_t be) { return ((shim_context *)be->context)->name.c_str(); }
static GGML_CALL void be_free(ggml_backend_t be) { auto *c = (shim_context *)be->context; (void)hipSetDevice(c->device); if (g_shims[c->device] == c) g_shims[c->device] = nullptr; if (c->ev) (void)hipEventDestroy(c->ev); (void)hipStreamDestroy(c->stream); cdna4_free(c->ctx); delete c; delete be; }
static GGML_CALL ggml_backend_buffer_type_t be_default_buft(ggml_backend_t be) { return ggml_backend_cuda_buffer_type(((shim_context *)be->context)->device); }
static GGML_CALL void be_set_async(ggml_backend_t be, ggml_tensor *t, const void *d, size_t off, size_t size) { auto *c = (shim_context *)be->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpyAsync((char *)t->data + off, d, size, hipMemcpyHostToDevice, c->stream)); }
static GGML_CALL void be_get_async(ggml_backend_t be, const ggml_tensor *t, void *d, size_t off, size_t size) { auto *c = (shim_context *)be->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipMemcpyAsync(d, (const char *)t->data + off, size, hipMemcpyDeviceToHost, c->stream)); }
static GGML_CALL bool be_cpy_async(ggml_backend_t src_be, ggml_backend_t dst_be, const ggml_tensor *src, ggml_tensor *dst) {
    if (!ggml_backend_is_cuda(src_be) || !ggml_backend_is_cuda(dst_be) || !buffer_is_ours(src->buffer) || !buffer_is_ours(dst->buffer)) return false;
    auto *d = (shim_context *)dst_be->context; HIP_CHECK(hipSetDevice(d->device));
    HIP_CHECK(hipMemcpyAsync(dst->data, src->data, ggml_nbytes(dst), hipMemcpyDeviceToDevice, d->stream));
    return true;
}
static GGML_CALL void be_sync(ggml_backend_t be) { auto *c = (shim_context *)be->context; HIP_CHECK(hipSetDevice(c->device)); HIP_CHECK(hipStreamSynchronize(c->stream)); }
static GGML_CALL bool be_supports_buft(ggml_backend_t be, ggml_backend_buffer_type_t t) {
    return t->iface.get_name == buft_get_name && ((shim_buft_ctx *)t->context)->device == ((shim_context *)be->context)->device;
}
static GGML_CALL bool be_offload_op(ggml_backend_t, const ggml_tensor *op) {      // ggml-cuda.cu offload rule: large batches only
    return (op->op == GGML_OP_MUL_MAT && op->ne[1] >= 32) || (op->op == GGML_OP_MUL_MAT_ID && op->ne[2] >= 32);
}
static GGML_CALL ggml_backend_event_t be_event_new(ggml_backend_t be) { auto *c = (shim_context *)be->context; HIP_CHECK(hipSetDevice(c->device)); hipEvent_t e; HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); return new ggml_backend_event{be, e}; }
static GGML_CALL void be_event_free(ggml_backend_event_t ev) { HIP_CHECK(hipEventDestroy((hipEvent_t)ev->context)); delete ev; }
static GGML_CALL void be_event_record(ggml_backend_event_t ev) { auto *c = (shim_context *)ev->backend->context; HIP_CHECK(hipEventRecord((hipEvent_t)ev->context, c->stream)); }
static GGML_CALL void be_event_wait(ggml_backend_t be, ggml_backend_event_t ev) { auto *c = (shim_context *)be->context; HIP_CHECK(hipStreamWaitEvent(c->stream, (hipEvent_t)ev->context, 0)); }
static GGML_CALL void be_event_sync(ggml_backend_event_t ev) { HIP_CHECK(hipEventSynchronize((hipEvent_t)ev->context)); }

static const ggml_backend_i k_backend_iface = { be_name, be_free, be_default_buft, be_set_async, be_get_async, be_cpy_async, be_sync,
                                                nullptr, nullptr, nullptr, nullptr, be_graph_compute, be_supports_op, be_supports_buft, be_offload_op,
                                                be_event_new, be_event_free, be_event_record, be_event_wait, be_event_sync };

extern "C" {

// `params` is the reference's "k=v,..." string (ggml-cuda.cu:5299-5389); unknown keys are ignored, none is needed here.
GGML_CALL ggml_backend_t ggml_backend_cuda_init(int device, const void *, const void *) {
    cdna4_context *ctx = cdna4_init(device);
    if (!ctx) { shim_log(GGML_LOG_LEVEL_ERROR, "ggml-hip-cdna4: %s\n", cdna4_last_error()); return nullptr; }     // ggml-cuda.cu:5392-5395
    HIP_CHECK(hipSetDevice(device)); hipStream_t st; HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    auto *c = new shim_context{device, ctx, st, std::string(GGML_CUDA_NAME) + std::to_string(device)};
    HIP_CHECK(hipEventCreateWithFlags(&c->ev, hipEventDisableTiming));
    for (int p = 0, n = cdna4_get_device_count(); p < n; ++p) {          // REDUCE reads / writes the peers' buffers directly (xGMI)
        int can = 0;
        if (p != device && hipDeviceCanAccessPeer(&can, device, p) == hipSuccess && can) { if (hipDeviceEnablePeerAccess(p, 0) != hipSuccess) (void)hipGetLastError(); }
    }
    if (device < GGML_CUDA_MAX_DEVICES) g_shims[device] = c;
    return new ggml_backend{shim_guid(), k_backend_iface, c};
}
GGML_CALL bool ggml_backend_is_cuda(ggml_backend_t be) { return be != nullptr && ggml_guid_matches(be->guid, shim_guid()); }
GGML_CALL int  ggml_backend_cuda_get_device_count(void) { return cdna4_get_device_count(); }
GGML_CALL void ggml_backend_cuda_get_device_description(int device, char *d, size_t n) { if (cdna4_get_device_description(device, d, n) != CDNA4_OK && n) d[0] = 0; }
GGML_CALL void ggml_backend_cuda_get_device_memory(int device, size_t *fr, size_t *tot) { if (cdna4_get_device_memory(device, fr, tot) != CDNA4_OK) { *fr = 0; *tot = 0; } }
GGML_CALL bool ggml_backend_cuda_register_host_buffer(void *p, size_t n) { if (getenv("GGML_CUDA_REGISTER_HOST") == nullptr) return false; if (hipHostRegister(p, n, hipHostRegisterPortable) != hipSuccess) { (void)hipGetLastError(); return false; } return true; }
GGML_CALL void ggml_backend_cuda_unregister_host_buffer(void *p) { if (getenv("GGML_CUDA_REGISTER_HOST") == nullptr) return; if (hipHostUnregister(p) != hipSuccess) (void)hipGetLastError(); }
GGML_CALL void ggml_backend_cuda_log_set_callback(ggml_log_callback cb, void *ud) { g_log_cb = cb; g_log_ud = ud; }
GGML_CALL void ggml_backend_cuda_invalidate_graphs(const void *) {}      // no captured graphs are kept across calls

static GGML_CALL ggml_backend_t reg_init(const char *params, void *user) { return ggml_backend_cuda_init((int)(intptr_t)user, params, nullptr); }
GGML_CALL void ggml_backend_cuda_reg_devices(void) {                      // ggml-backend.cpp:456-489, ggml-cuda.cu:5518-5529
    const int n = ggml_backend_cuda_get_device_count();
    for (int i = 0; i < n; ++i) { char name[64]; snprintf(name, sizeof(name), "%s%d", GGML_CUDA_NAME, i); ggml_backend_register(name, reg_init, ggml_backend_cuda_buffer_type(i), (void *)(intptr_t)i); }
}

} // extern "C"
