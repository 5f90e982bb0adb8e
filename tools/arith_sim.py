"""CPU simulation of reduced-precision operand rounding at the bench configuration (BASELINE configs[1]: 768 utterances):
the stage convolutions are evaluated by ATen with their operands rounded to fp16 / bf16 (f32 accumulation), everything
else in f32 -- embedding error, error of d_n - d_p and flipped filter decisions against the f32 run.  This is what
the choice of the fp16 arithmetic (DESIGN_LOG.md 3.1) was made on before the kernel existed.  python tools/arith_sim.py"""
import os, sys, time, numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
torch.set_num_threads(8)
sd_np = synthetic_state_dict(0, 1211)
sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
g = torch.Generator(device="cpu").manual_seed(1234)
x_all = torch.randn(768, 1, 160, 64, generator=g)

def rnd(t, mode):
    if mode == 'f32': return t
    if mode == 'f16': return t.half().float()
    if mode == 'bf16': return t.bfloat16().float()
    raise ValueError(mode)

def forward(x, amode, wmode, store=None, dbl=False):
    """amode/wmode: rounding of conv operands (activations / weights) for stage convs (not conv1).
    store: rounding of stored activations (affects residual too)."""
    dt = torch.float64 if dbl else torch.float32
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"].to(dt), sd[name + ".running_var"].to(dt), sd[name + ".weight"].to(dt),
                            sd[name + ".bias"].to(dt), False, 0.1, 1e-5)
    x = x.to(dt)
    for i in range(1, 5):
        w = sd[f"model.conv{i}.weight"]
        if i == 1:
            x = F.conv2d(x, w.to(dt), None, 2, 2)
        else:
            x = F.conv2d(rnd(x.float(), amode).to(dt), rnd(w, wmode).to(dt), None, 2, 2)
        x = F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0)
        if store: x = rnd(x.float(), store).to(dt)
        r = x
        y = F.conv2d(rnd(x.float(), amode).to(dt), rnd(sd[f"model.layer{i}.0.conv1.weight"], wmode).to(dt), None, 1, 1)
        y = F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0)
        if store: y = rnd(y.float(), store).to(dt)
        y = F.conv2d(rnd(y.float(), amode).to(dt), rnd(sd[f"model.layer{i}.0.conv2.weight"], wmode).to(dt), None, 1, 1)
        y = bn(y, f"model.layer{i}.0.bn2")
        x = F.hardtanh(y + r, 0.0, 20.0)
        if store: x = rnd(x.float(), store).to(dt)
    x = F.adaptive_avg_pool2d(x, (1, None)).reshape(x.size(0), -1)
    x = F.linear(x, sd["model.fc.weight"].to(dt), sd["model.fc.bias"].to(dt))
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)
    return (x / norm.view(-1, 1) * 10).float()

def dist(a, b):
    return torch.sqrt(((a - b) ** 2).sum(1) + 1e-4 / 512)

def run(name, **kw):
    t0 = time.time()
    with torch.no_grad():
        e = torch.cat([forward(x_all[i:i+64], **kw) for i in range(0, 768, 64)])
    print(f"{name}: {time.time()-t0:.1f}s", flush=True)
    return e

ref = run('f32', amode='f32', wmode='f32')
a, p, n = ref[:256], ref[256:512], ref[512:]
dp, dn = dist(a, p), dist(a, n)
gap = dn - dp - 0.1
print("ref: d_p mean %.4f d_n mean %.4f; gap min|.| %.3e; sorted |gap| first 8:" % (dp.mean(), dn.mean(), gap.abs().min()), np.sort(gap.abs().numpy())[:8])
print("selected", int((gap < 0).sum()), "of 256; gap std %.4f" % gap.std())

for name, kw in [('f16/f16', dict(amode='f16', wmode='f16')),
                 ('f16/f16 store f16', dict(amode='f16', wmode='f16', store='f16')),
                 ('f32act/f16w', dict(amode='f32', wmode='f16')),
                 ('f16act/f32w', dict(amode='f16', wmode='f32')),
                 ('bf16/bf16', dict(amode='bf16', wmode='bf16'))]:
    e = run(name, **kw)
    rel = ((e - ref).norm(dim=1) / ref.norm(dim=1))
    relmax = ((e - ref).abs().max() / ref.abs().max())
    a2, p2, n2 = e[:256], e[256:512], e[512:]
    g2 = dist(a2, n2) - dist(a2, p2) - 0.1
    flips = int(((g2 < 0) != (gap < 0)).sum())
    print(f"  {name}: rel-L2 mean {rel.mean():.3e} max {rel.max():.3e}; max-abs/max {relmax:.3e}; gap err max {(g2-gap).abs().max():.3e} rms {(g2-gap).pow(2).mean().sqrt():.3e}; flips {flips}", flush=True)
