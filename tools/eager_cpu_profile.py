"""Host-side cost of one eager training step (the N > 1 path): cProfile over 5 steps, GPU kept async."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = bench.parse()
device = torch.device("cuda", 0)
from flamingo_mini_amd import FusedAdamW
from flamingo_mini_amd.backbones import load_stock_gemm_tuning
load_stock_gemm_tuning()
model, cfg = bench.build_model(args, device, torch.bfloat16)
batch = bench.synthetic_batch(args, cfg, device, torch.bfloat16, 0)
opt = FusedAdamW(list(model.parameters_trainable()), lr=1e-4)

def step():
    model.zero_grad(set_to_none=True)
    out = model(**batch)
    out.loss.backward()
    opt.step()

for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
t_issue = (time.perf_counter() - t0) / 5 * 1e3
torch.cuda.synchronize()
t_total = (time.perf_counter() - t0) / 5 * 1e3
print(f"host issue time {t_issue:.2f} ms/step, wall {t_total:.2f} ms/step")
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
