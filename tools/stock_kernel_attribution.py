"""Which Python line launches the stock (ATen) kernels of one eager training step: torch.profiler with stacks, device time of every
ATen op grouped by (op, innermost frame inside flamingo_mini_amd / transformers).  Answers "are the fills / casts / cats of a step the
backbones' or this package's glue?" - the rocprofv3 summary only has kernel names.

    python tools/stock_kernel_attribution.py [bench.py flags]            (GPU box; --toy runs a CPU toy model to check the script)
"""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity

VERBOSE = torch._C._profiler._ExperimentalConfig(verbose=True)      # (without it the events carry no Python stacks)
TOY = "--toy" in sys.argv
if TOY:
    sys.argv.remove("--toy")


def frame_of(stack):
    """innermost frame that lies in this package, else in transformers, else the innermost frame at all"""
    for key in ("flamingo_mini_amd", "flamingo-mini_amd", "transformers/models", "bench.py"):
        for fr in stack:
            if key in fr:
                return fr.strip()
    return stack[0].strip() if stack else "?"


def report(prof, steps):
    by = collections.defaultdict(lambda: [0.0, 0])
    for ev in prof.events():
        t = getattr(ev, "self_device_time_total", None)
        if t is None:
            t = getattr(ev, "self_cuda_time_total", 0.0)
        if not t and not TOY:
            continue
        if TOY:
            t = ev.self_cpu_time_total
        where = frame_of(ev.stack or [])
        node, up = None, ev
        while up is not None:                      # an op of the backward pass: name the autograd node it runs under
            if up.name.startswith("autograd::engine::evaluate_function"):
                node = up.name.split(": ", 1)[-1]
                break
            up = up.cpu_parent
        if node is not None:
            where = "backward of " + node
        key = (ev.name, where)
        by[key][0] += t
        by[key][1] += 1
    rows = sorted(by.items(), key=lambda kv: -kv[1][0])
    tot = sum(v[0] for v in by.values())
    print(f"device time attributed: {tot / 1e3 / steps:.2f} ms/step over {steps} step(s)")
    print("ms/step  calls/step  op  |  frame")
    for (name, fr), (t, n) in rows[:70]:
        print(f"{t / 1e3 / steps:7.3f}  {n / steps:8.1f}  {name[:48]:48s} | {fr[-150:]}")
    # the same, op names folded: which frames fill / cast / concatenate
    for pat in ("fill_", "zero_", "copy_", "_to_copy", "cat", "zeros", "mul", "add"):
        sel = [(k, v) for k, v in rows if k[0].split("::")[-1] == pat or k[0].split("::")[-1].startswith(pat)]
        if not sel:
            continue
        print(f"--- aten ops matching '{pat}': {sum(v[0] for _, v in sel) / 1e3 / steps:.3f} ms/step in {sum(v[1] for _, v in sel) / steps:.0f} calls/step")
        for (name, fr), (t, n) in sel[:14]:
            print(f"    {t / 1e3 / steps:7.3f}  {n / steps:7.1f}  {name[:32]:32s} | {fr[-150:]}")


if TOY:
    m = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.GELU(), torch.nn.Linear(64, 8))
    x = torch.randn(16, 64)
    with profile(activities=[ProfilerActivity.CPU], with_stack=True, experimental_config=VERBOSE) as prof:
        m(x).sum().backward()
    for ev in prof.events()[:5]:
        print(ev.name, (ev.stack or [])[:2], getattr(ev, "self_device_time_total", None))
    report(prof, 1)
    sys.exit(0)

import bench
args = bench.parse()
device = torch.device("cuda", 0)
from flamingo_mini_amd import FusedAdamW
model, cfg = bench.build_model(args, device, torch.bfloat16)
batch = bench.synthetic_batch(args, cfg, device, torch.bfloat16, 0)
model.set_launch_structure(hoist_kv=True)
opt = FusedAdamW(list(model.parameters_trainable()), lr=1e-4)


def step():
    model.zero_grad(set_to_none=True)
    out = model(**batch)
    out.loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
STEPS = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, experimental_config=VERBOSE) as prof:
    for _ in range(STEPS):
        step()
    torch.cuda.synchronize()
report(prof, STEPS)
