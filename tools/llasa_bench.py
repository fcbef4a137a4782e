"""Llasa train step (model_sigmaVAE.Llasa.forward + backward + fused AdamW) at the Llama-3.2-1B shape the reference trains
(hidden 2048, 16 layers, 32 heads / 8 kv heads, intermediate 8192, vocab 128256 + 8 special tokens, latent_dim 64),
random weights, synthetic batch: text prefix + audio frames per sample.   python tools/llasa_bench.py [B] [L] [steps]"""
import json, os, sys, tempfile, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kalle_audio_amd.model_sigmaVAE import Llasa
from kalle_audio_amd.engine import DataParallelTrainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
cfg = dict(model_type="llama", vocab_size=128256, hidden_size=2048, intermediate_size=8192, num_hidden_layers=16,
           num_attention_heads=32, num_key_value_heads=8, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0,
           rope_scaling=dict(rope_type="llama3", factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                             original_max_position_embeddings=8192), tie_word_embeddings=True)
d = tempfile.mkdtemp(prefix="kalle_llama_")
json.dump(cfg, open(os.path.join(d, "config.json"), "w"))


class Tok:
    def __len__(self):
        return 128264


dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    m = Llasa({"llm_model_name_or_path": d, "latent_dim": 64, "audio_proj_dim": 2048}, Tok(), use_flash_attention=False)
INFER_ONLY = "--infer-only" in sys.argv      # generation with the KV cache only (for profiling the decode step)
tr = None if INFER_ONLY else DataParallelTrainer(m, lr=1e-5, optimizer="AdamW", weight_decay=0.01)
nparam = sum(p.numel() for p in m.parameters())
ids = torch.randint(0, 128264, (B, L), device=dev)
lat = torch.randn(B, L, 64, device=dev)
lbl = torch.randn(B, L, 64, device=dev)
ids_mask = torch.zeros(B, L, device=dev); ids_mask[:, :64] = 1
audio_mask = 1 - ids_mask
target_mask = torch.zeros(B, L, device=dev); target_mask[:, 63:L - 1] = 1
end_mask = torch.zeros(B, L, device=dev); end_mask[:, L - 1] = 1


def step():
    out = m(ids, lat, lbl, ids_mask, audio_mask, target_mask, end_mask)
    tr.backward(out["audio_loss"] * 1.0 + out["end_loss"] * 1.0)
    return out


for _ in range(0 if INFER_ONLY else 2):
    out = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(0 if INFER_ONLY else steps):
    out = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# algorithmic FLOPs: 6 * (non-embedding params) per token + attention 12 * L * D per token per layer (causal: half)
nonemb = nparam - 128264 * 2048
fl = (6.0 * nonemb + 16 * 12.0 * L * 2048 * 0.5) * B * L
if not INFER_ONLY:
  print(f"Llasa Llama-3.2-1B-shape train step B={B} L={L}: {dt*1e3:.1f} ms/step, {B*L/dt:.0f} tokens/s, "
        f"{B*L/12.5/dt:.0f} audio-s/s (12.5 Hz frames), {fl/dt/1e12:.0f} TFLOP/s algorithmic, params {nparam/1e9:.2f} B, "
        f"loss {out['audio_loss'].item():.3f}")

if "--infer" in sys.argv or INFER_ONLY:
    # frame-by-frame generation (Llasa.infer): 64 prompt tokens, 200 frames, KV cache vs the reference's full re-forward
    m.eval()
    pid = torch.randint(0, 128264, (64,), device=dev)
    for use_cache, nfr in ((True, 200),) if INFER_ONLY else ((True, 200), (False, 200)):
        m.infer(pid, None, end_disp_kl_thres=-1.0, max_length=4, use_cache=use_cache)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = m.infer(pid, None, end_disp_kl_thres=-1.0, max_length=nfr, use_cache=use_cache)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"infer use_cache={use_cache}: {nfr} frames in {dt*1e3:.0f} ms = {nfr/dt:.1f} frames/s "
              f"({nfr/dt/12.5:.2f} x real time at 12.5 Hz), out {tuple(out.shape)}")
