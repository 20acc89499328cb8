"""GraphedTrainStep: a captured HIP graph (forward + backward + FusedAdamW) replayed n times must equal n eager steps."""
import copy

import pytest
import torch

from util import dev, rel, rnd

pytestmark = pytest.mark.gpu


class _Toy(torch.nn.Module):
    """Resampler + gated cross-attention block -> scalar loss: every kernel family of the library in one tiny step."""

    def __init__(self):
        super().__init__()
        from flamingo_mini_amd import GatedCrossAttentionBlock, PerceiverResampler
        self.resampler = PerceiverResampler(dim=64, depth=1, heads=2, dim_head=32, num_latents=8, num_time_embeds=2)
        self.block = GatedCrossAttentionBlock(dim=64, dim_visual=64, dim_head=32, heads=2, n_visual=8)
        with torch.no_grad():
            self.block.alpha_attn.fill_(0.5)
            self.block.alpha_ffw.fill_(0.5)

    def forward(self, x_f, y, media_locations):
        vf = self.resampler(x_f).unsqueeze(1)
        out, _ = self.block(y, vf, media_locations)
        return out.float().pow(2).mean()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_graph_replay_equals_eager_steps(dtype):
    from flamingo_mini_amd import FusedAdamW, GraphedTrainStep
    torch.manual_seed(0)
    eager = _Toy().cuda().to(dtype)
    graphed = copy.deepcopy(eager)
    ml = torch.zeros(2, 16, dtype=torch.int64, device="cuda")
    ml[:, 0] = 1
    batches = [dict(x_f=dev(rnd((2, 1, 24, 64), 10 + i), dtype), y=dev(rnd((2, 16, 64), 20 + i), dtype), media_locations=ml) for i in range(6)]
    opt_e = FusedAdamW(eager.parameters(), lr=1e-2)
    losses_e = []
    for b in batches:
        eager.zero_grad(set_to_none=True)
        loss = eager(**b)
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    # the graphed twin: 2 eager warm-up steps on batches 0, 1 (inside the constructor), then replays on batches 2..5
    opt_g = FusedAdamW(graphed.parameters(), lr=1e-2, capturable=True)

    step = GraphedTrainStep(graphed, opt_g, batches[0], warmup=1, loss_fn=lambda out: out)
    # constructor ran exactly one eager step on batch 0; continue with replays on batches 1..5
    losses_g = [None]
    for b in batches[1:]:
        losses_g.append(float(step(b)))
    tol = 2e-5 if dtype == torch.float32 else 3e-2
    for le, lg in zip(losses_e[1:], losses_g[1:]):
        assert abs(le - lg) <= tol * max(1.0, abs(le)), (losses_e, losses_g)
    for (n, pe), (_, pg) in zip(eager.named_parameters(), graphed.named_parameters()):
        assert rel(pg, pe) < (1e-4 if dtype == torch.float32 else 3e-2), n
    assert {float(s["step"]) for s in opt_g.state_dict()["state"].values()} == {6.0}
