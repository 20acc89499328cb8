"""Fused shifted cross-entropy (ff_shifted_ce_fwd/bwd) against torch's F.cross_entropy on shifted logits (float64 on CPU)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from util import dev, rel, rnd

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("reduction", ["mean", "sum", "none"])
def test_shifted_cross_entropy_matches_torch(dtype, reduction):
    from flamingo_mini_amd import functional as F
    b, L, V = 3, 9, 50258                                        # gpt2 vocabulary + <EOC>: rows are not 16-byte aligned
    logits = dev(rnd((b, L, V), 1, 3.0), dtype).requires_grad_(True)
    labels = torch.from_numpy(np.random.default_rng(2).integers(0, V, (b, L))).cuda()
    labels[1, 4] = -100                                          # ignored position
    loss = F.shifted_cross_entropy(logits, labels, reduction=reduction)
    w = dev(rnd(tuple(loss.shape), 3)) if reduction == "none" else None
    (loss * w).sum().backward() if w is not None else loss.backward()
    ref_logits = logits.detach().double().cpu().requires_grad_(True)
    ref = TF.cross_entropy(ref_logits[:, :-1].reshape(-1, V), labels[:, 1:].reshape(-1).cpu(), reduction=reduction)
    (ref * w.double().cpu()).sum().backward() if w is not None else ref.backward()
    assert rel(loss, ref.detach()) < (1e-6 if dtype == torch.float32 else 1e-5)          # fp32 math on the same (rounded) logits
    assert rel(logits.grad, ref_logits.grad) < (1e-5 if dtype == torch.float32 else 8e-3)   # gradient is stored in the logits dtype
    assert float(logits.grad[:, -1].abs().max()) == 0.0 and float(logits.grad[1, 3].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
@pytest.mark.parametrize("n", [1, 7, 4096 * 257 + 3])
def test_quick_gelu_matches_torch(dtype, n):
    from flamingo_mini_amd import functional as F
    x = dev(rnd((n,), 5, 2.0), dtype).requires_grad_(True)
    w = dev(rnd((n,), 6), dtype)
    y = F.quick_gelu(x)
    (y * w).sum().backward()
    xr = x.detach().double().cpu().requires_grad_(True)
    yr = xr * torch.sigmoid(1.702 * xr)
    (yr * w.double().cpu()).sum().backward()
    tol = 2e-6 if dtype == torch.float32 else 6e-3
    assert rel(y, yr.detach()) < tol and rel(x.grad, xr.grad) < tol


def test_shifted_ce_flags_out_of_range_labels_and_all_ignored_batches():
    """A label outside [0, vocab) that is not ignore_index must not read out of bounds: its row (and the mean) become NaN, as loud as
    torch's device assert; an all-ignored batch gives NaN like F.cross_entropy(reduction='mean')."""
    from flamingo_mini_amd import functional as F
    logits = torch.randn(2, 5, 33, device="cuda", requires_grad=True)
    labels = torch.randint(0, 33, (2, 5), device="cuda")
    bad = labels.clone(); bad[1, 3] = 33
    rows = F.shifted_cross_entropy(logits, bad, reduction="none")
    assert torch.isnan(rows[4 + 2]) and int(torch.isnan(rows).sum()) == 1
    loss = F.shifted_cross_entropy(logits, bad)
    assert torch.isnan(loss)
    loss.backward()
    assert torch.isnan(logits.grad[1, 2]).all() and not torch.isnan(logits.grad[0]).any()
    ignored = torch.full_like(labels, -100)
    assert torch.isnan(F.shifted_cross_entropy(logits.detach(), ignored))
    ok = labels.clone(); ok[0, 1:] = -100
    ref = torch.nn.functional.cross_entropy(logits.detach()[:, :-1].reshape(-1, 33), ok[:, 1:].reshape(-1))
    assert abs(float(F.shifted_cross_entropy(logits.detach(), ok)) - float(ref)) < 1e-5
