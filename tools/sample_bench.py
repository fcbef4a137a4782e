"""DiT sampling throughput at generation batch sizes (v-DDIM, CFG scale 6 -> batch doubled inside the model):
python tools/sample_bench.py [B] [steps]   - eager launches vs one HIP-graph replay per sampler step"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kalle_audio_amd
kalle_audio_amd.install()
from kalle_audio_amd.graph import GraphedForward
from stable_audio_tools.models.dit import DiffusionTransformer
from stable_audio_tools.inference.sampling import sample
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    m = DiffusionTransformer(io_channels=1024, embed_dim=1536, depth=24, num_heads=24, cond_token_dim=768,
                             project_cond_tokens=False, global_cond_dim=1536, transformer_type="continuous_transformer",
                             global_cond_type="prepend")
m.eval().requires_grad_(False)
for p in m.parameters():
    if p.dim() > 1 and float(p.abs().max()) == 0.0:
        torch.nn.init.normal_(p, std=0.02)
x = torch.randn(B, 1024, 125, device=dev)
cond = torch.randn(B, 130, 768, device=dev)
glob = torch.randn(B, 1536, device=dev)
kw = dict(cross_attn_cond=cond, global_embed=glob, cfg_scale=6.0)
pre = dict(kw, **m.precompute_conditioning(cross_attn_cond=cond, global_embed=glob, cfg_scale=6.0))   # as generate_diffusion_cond
outs = {}
for name, model, kw in (("eager", m, kw), ("graph", GraphedForward(m), kw), ("graph+precond", GraphedForward(m), pre)):
    sample(model, x, 2, 0.0, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs[name] = sample(model, x, steps, 0.0, **kw).clone()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: B={B} x {steps} DDIM steps (CFG 6): {dt*1e3:.0f} ms = {dt/steps*1e3:.2f} ms/step = {B*10/dt:.1f} audio-s/s generated")
d = (outs["eager"].float() - outs["graph"].float()).norm() / outs["eager"].float().norm()
print(f"graph vs eager rel-L2 {d.item():.2e}; conditioning hoisted out of the loop: equal = {torch.equal(outs['graph'], outs['graph+precond'])}")
