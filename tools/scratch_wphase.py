import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepspeaker_pytorch_amd import _native
from conv_probe import LAYERS
lib = _native.NativeLib(sys.argv[1])
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
B = 256
for name, H, W, Cin, Cout, KS, s in LAYERS:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev); gy = torch.randn(B, Ho, Wo, Cout, device=dev)
    gw = torch.empty(Cout, Cin, KS, KS, device=dev)
    shp = _native.ConvShape(B, H, W, Cin, Cout, KS, s)
    n = lib.raw("ds_conv_wgrad_bf16_workspace_floats")(ctypes.byref(shp))
    ws = torch.zeros(n, device=dev)
    for _ in range(2):
        lib.call("ds_conv_wgrad_bf16", ctypes.byref(shp), p(x), p(gy), p(ws), p(gw), st)
    torch.cuda.synchronize()
    d = ws[n - 4096 * 16:].view(torch.int64).view(4096, 8).double().cpu()
    d = d[d[:, 5] > 0]
    m = d.mean(0)
    print(f"{name:16s} WGs {len(d)} tiles/WG {m[5]:.1f}  clk per WG: setup {m[0]:7.0f} barrier {m[1]:7.0f} write {m[2]:7.0f} issue {m[6]:7.0f} contract {m[3]:7.0f} total {m[4]:8.0f}  (ideal MFMA {m[5]*4*(9 if KS==3 else 5)*3*32:.0f})")
