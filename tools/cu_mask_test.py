"""Developer check (GPU box): does a HIP CU mask take effect for kernels launched through torch.cuda.ExternalStream?"""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(enabled_cus, n_cu=256):
    words = (n_cu + 31) // 32
    mask = [0] * words
    for cu in enabled_cus:
        mask[cu // 32] |= 1 << (cu % 32)
    arr = (ctypes.c_uint32 * words)(*mask)
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value)
x = torch.randn(64 * 1024 * 1024, device="cuda")
def t(stream):
    with torch.cuda.stream(stream):
        for _ in range(3):
            y = torch.sin(x) * 2 + 1
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(10):
            y = torch.sin(torch.sin(torch.sin(x)))
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10
print("all CUs        %.3f ms" % t(torch.cuda.Stream()))
print("first 128 CUs  %.3f ms" % t(masked_stream(range(128))))
print("first 32 CUs   %.3f ms" % t(masked_stream(range(32))))
print("every 8th CU   %.3f ms" % t(masked_stream(range(0, 256, 8))))
