// TEST INFRASTRUCTURE: a stand-in for the handful of HIP runtime entry points the backend shim and the C-ABI library call, so that the shim's HOST logic -- buffer
// set / get / re-tiling state, supports_op decisions, eager graph walking -- can run in the CPU-only container (tests/test_shim_host_logic.py runs tests/shim_host_case.py in a
// child process with LD_PRELOAD=<this library>).  "Device" memory is host memory, copies are memcpy, kernel launches do nothing (results of compute nodes are garbage and never
// looked at), stream capture is refused (the shim then walks its graphs eagerly).  Nothing here is linked into, loaded by or shipped with the product.
#include <hip/hip_runtime_api.h>

#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) {
    memset(p, 0, sizeof(*p)); strcpy(p->name, "fake gfx950 (host memory)"); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256; p->maxSharedMemoryPerMultiProcessor = 160 * 1024; p->totalGlobalMem = (size_t)8 << 30; p->warpSize = 64; return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 0; return hipSuccess; }
hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)8 << 30; *t = (size_t)8 << 30; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake hip error"; }
hipError_t hipDeviceCanAccessPeer(int *c, int, int) { *c = 0; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

static long g_handles = 0x1000;
static hipError_t alloc(void **p, size_t n) { void *q = nullptr; if (posix_memalign(&q, 256, n ? n : 1)) return hipErrorOutOfMemory; memset(q, 0, n ? n : 1); *p = q; return hipSuccess; }      // (zero-filled: whatever reads results no kernel wrote sees zeros, run after run)
hipError_t hipMalloc(void **p, size_t n) { return alloc(p, n); }
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned) { return alloc(p, n); }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return alloc(p, n); }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(g_handles += 16); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
// stream capture: refused by default (the shim then walks its graphs eagerly); FAKE_HIP_CAPTURE=1 accepts captures and "replays" them as no-ops, which exercises the shim's
// graph cache (keys, slot table, replay bookkeeping) -- nothing a captured graph would have computed exists here anyway
static bool capture_ok() { static const bool on = getenv("FAKE_HIP_CAPTURE") != nullptr; return on; }
static bool g_capturing = false;
hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus *st) { *st = g_capturing ? hipStreamCaptureStatusActive : hipStreamCaptureStatusNone; return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { if (!capture_ok()) return hipErrorNotSupported; g_capturing = true; return hipSuccess; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { if (!capture_ok()) { *g = nullptr; return hipErrorNotSupported; } g_capturing = false; *g = (hipGraph_t)(g_handles += 16); return hipSuccess; }
hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t, hipGraphNode_t *, char *, size_t) { if (!capture_ok()) return hipErrorNotSupported; *e = (hipGraphExec_t)(g_handles += 16); return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return capture_ok() ? hipSuccess : hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)(g_handles += 16); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)(g_handles += 16); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *, void *) { return hipErrorNotSupported; }
hipError_t hipIpcOpenMemHandle(void **, hipIpcMemHandle_t, unsigned) { return hipErrorNotSupported; }
hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *) { if (a) memset(a, 0, sizeof(*a)); return hipSuccess; }

// kernel registration and launches: accepted and ignored
void **__hipRegisterFatBinary(const void *) { static void *h = nullptr; return &h; }
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
void __hipRegisterVar(void **, void *, char *, char *, int, size_t, int, int) {}
void __hipUnregisterFatBinary(void **) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *s, hipStream_t *st) { *g = dim3(1, 1, 1); *b = dim3(1, 1, 1); *s = 0; *st = nullptr; return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { return hipSuccess; }
}
