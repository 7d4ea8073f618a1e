"""Probe: how to get a process back to a usable state after a HIP graph capture was invalidated half way (ROCm 7 / torch 2.10).
Invalidate on purpose with a host sync inside the capture, then try recovery steps and test with a fresh allocation."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda", 0)
x = torch.ones(1024, device=dev)

def usable():
    try:
        y = torch.empty(1 << 20, device=dev); y.fill_(1); torch.cuda.synchronize(); return True
    except Exception as e:
        return "no: " + str(e).split("\n")[0][:90]

def status(st):
    s = ctypes.c_int(-1); rc = hip.hipStreamIsCapturing(ctypes.c_void_p(st.cuda_stream), ctypes.byref(s)); return rc, s.value

for mode in ("global", "thread_local", "relaxed"):
    st = torch.cuda.Stream(device=dev); g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, stream=st, capture_error_mode=mode):
            x.add_(1)
            import torch.distributed  # noqa
            x.cpu()                    # illegal during capture -> invalidates
    except Exception as e:
        print(mode, "capture failed:", str(e).split("\n")[0][:100])
    print(" after failure: capturing(rc,status) =", status(st), "usable:", usable())
    gph = ctypes.c_void_p(0); rc = hip.hipStreamEndCapture(ctypes.c_void_p(st.cuda_stream), ctypes.byref(gph))
    print(" hipStreamEndCapture rc =", rc, "graph =", gph.value, "-> capturing", status(st), "usable:", usable())
    print(" hipGetLastError:", hip.hipGetLastError(), hip.hipGetLastError(), "usable:", usable())
    m = ctypes.c_int(2); rc = hip.hipThreadExchangeStreamCaptureMode(ctypes.byref(m)); print(" exchange capture mode rc", rc, "old mode", m.value, "usable:", usable())
    del g
    torch.cuda.empty_cache()
    print(" after del graph + empty_cache: usable:", usable())
