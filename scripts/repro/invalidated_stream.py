"""After torch's capture_end() fails on an invalidated capture, what state is the capture stream in, and can it be ended?"""
import ctypes
import torch

x = torch.zeros(1024, device="cuda")
path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
print("hip runtime:", path)
hip = ctypes.CDLL(path)
hip.hipStreamIsCapturing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
hip.hipStreamEndCapture.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]


def status(s):
    st = ctypes.c_int(-1)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(s.cuda_stream), ctypes.byref(st))
    return rc, st.value   # 0 none, 1 active, 2 invalidated


side = torch.cuda.Stream()
graph = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
print("before", status(side))
with torch.cuda.stream(side):
    graph.capture_begin(capture_error_mode="thread_local")
    print("capturing", status(side))
    try:
        float(x.sum())
    except Exception as e:
        print("in-capture failure:", str(e)[:50])
    print("after the illegal call", status(side))
    try:
        graph.capture_end()
    except Exception as e:
        print("capture_end:", str(e)[:60])
    print("after capture_end", status(side))
for i in range(2):
    g = ctypes.c_void_p()
    rc = hip.hipStreamEndCapture(ctypes.c_void_p(side.cuda_stream), ctypes.byref(g))
    print("raw hipStreamEndCapture rc", rc, "graph", g.value, "status", status(side))
print("hipGetLastError", hip.hipGetLastError())
try:
    with torch.cuda.stream(side):
        y = x + 1
    torch.cuda.synchronize()
    print("stream usable again:", float(y.sum()))
except Exception as e:
    print("stream still unusable:", str(e)[:80])
