"""one warm-up + N timed steps of the bench workload without the extras (for rocprofv3 --pmc passes):
python tools/bench_one_step.py [B] [steps]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kalle_audio_amd import engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
model = bench.build_model(dev)
tr = engine.DataParallelTrainer(model, lr=1e-5, optimizer="Adam")
lat, noise, t, cond = bench.make_batch(B, dev, 1234)
for _ in range(1 + steps):
    tr.train_step(model, lat, t, noise, cond, objective="v")
torch.cuda.synchronize()
