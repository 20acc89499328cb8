"""Tune the hipBLASLt / rocBLAS solution of every GEMM shape the *stock* backbones (CLIP, GPT-2, lm_head) run at the benchmark
configuration with PyTorch's TunableOp, and write the selection file bench.py loads (flamingo-mini_amd/tuning/).
The fusion library's own GEMMs do not go through torch and are unaffected.
    python tools/tune_stock_gemms.py [--out gpurun_out/tunableop_gfx950.csv]      # on an MI355X, ~1-2 min
"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="gpurun_out/tunableop_gfx950.csv")
ap.add_argument("--max-ms", type=int, default=15)
ap.add_argument("--max-iter", type=int, default=20)
a, rest = ap.parse_known_args()
sys.argv = [sys.argv[0]] + rest
args = bench.parse()
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
tun = torch.cuda.tunable
tun.enable(True)
tun.tuning_enable(True)
tun.set_max_tuning_duration(a.max_ms)
tun.set_max_tuning_iterations(a.max_iter)
tun.set_filename(a.out)
device = torch.device("cuda", 0)
model, cfg = bench.build_model(args, device, torch.bfloat16)
batch = bench.synthetic_batch(args, cfg, device, torch.bfloat16, 0)
for _ in range(2):
    model.zero_grad(set_to_none=True)
    model(**batch).loss.backward()
torch.cuda.synchronize()
tun.write_file(a.out) if hasattr(tun, "write_file") else None
print("wrote", a.out, "entries", len(tun.get_results()))
