import ctypes, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepspeaker_pytorch_amd import _native
from conv_probe import LAYERS
lib = _native.NativeLib(sys.argv[1])
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
for B in [int(a) for a in sys.argv[2:]] or [768]:
  for name, H, W, Cin, Cout, KS, s in LAYERS:
      g = torch.Generator(device="cpu").manual_seed(1)
      x = torch.rand(B, H, W, Cin, generator=g).to(dev)
      w = (torch.randn(Cout, Cin, KS, KS, generator=g) * 0.05).to(dev)
      n = Cout * Cin * KS * KS
      whi = torch.empty(n, dtype=torch.bfloat16, device=dev); wlo = torch.empty(n, dtype=torch.bfloat16, device=dev)
      lib.call("ds_pack_conv_weight_bf16", p(w), p(whi), p(wlo), Cout, Cin, KS, st)
      Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
      y = torch.empty(B, Ho, Wo, Cout, device=dev); res = torch.rand(B, Ho, Wo, Cout, device=dev)
      sc = torch.ones(Cout, device=dev); sh = torch.zeros(Cout, device=dev)
      shp = _native.ConvShape(B, H, W, Cin, Cout, KS, s)
      o = (ctypes.c_int * 8)()
      lib.call("ds_conv_bf16_plan_describe", ctypes.byref(shp), 1, o)
      grid = o[4]
      dbg = torch.zeros(grid * 8, dtype=torch.int64, device=dev)
      flags = 1 | 4 | (2 if s == 1 else 0) | 0x2000
      for _ in range(3):
          lib.call("ds_conv_fwd_bf16", ctypes.byref(shp), p(x), p(whi), p(wlo), p(sc), p(sh), p(res) if s == 1 else None, p(y), p(dbg), flags, st)
      torch.cuda.synchronize()
      d = dbg.view(grid, 8).double().cpu()
      m = d.mean(0)
      tot = d[:, 6] - d[:, 5]
      print(f'B={B} {name:16s} cfg={list(o)} per-WG clk: prologue {m[0]:7.0f} barrier {m[1]:6.0f} stage {m[2]:7.0f} taps {m[3]:7.0f} epilogue {m[4]:7.0f} total {float(tot.mean()):8.0f}')
