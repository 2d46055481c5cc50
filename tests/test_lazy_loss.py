"""CPU: the algebra of loss_utils.LazyLoss (opt-in lazy loss scalars, `run.py --lazy-loss`) on stand-in node outputs -- the
trainers' loss line (prune_finetune.py:161-164) must give the value and the gradients of the eager formula, and anything the class
does not know must fall back to real tensors."""
import math

import pytest
import torch

from lightgaussian_amd import loss_utils as LU


def _pair():
    x = torch.tensor(0.25, requires_grad=True); y = torch.tensor(0.9, requires_grad=True)
    l1, ss = x * 1.0, y * 1.0
    return x, y, l1, ss


@pytest.mark.parametrize("lam", [0.2, 0.0, 1.0, 0.35])
def test_the_trainers_loss_line_on_lazy_scalars(lam):
    x, y, l1, ss = _pair()
    Ll1, S = LU.LazyLoss(l1, ss, (0, 1, 0)), LU.LazyLoss(l1, ss, (0, 0, 1))
    loss = (1.0 - lam) * Ll1 + lam * (1.0 - S)                              # the reference's expression, literally
    assert isinstance(loss, LU.LazyLoss)
    c0, c1, c2 = loss._lz[2]
    assert (c0, c1, c2) == (lam, 1.0 - lam, -lam)
    ref = (1.0 - lam) * l1 + lam * (1.0 - ss)
    assert loss.item() == pytest.approx(float(ref.detach()), rel=1e-6)
    if lam in (0.0, 1.0):
        pass                                                               # one coefficient is zero: that output gets no gradient
    loss.backward(retain_graph=True)
    gx, gy = x.grad, y.grad
    x.grad = y.grad = None
    ref.backward()
    for lazy_g, ref_g in ((gx, x.grad), (gy, y.grad)):
        if lazy_g is None:                                                  # coefficient 0: never entered the graph
            assert float(ref_g) == 0.0
        else:
            assert torch.equal(lazy_g, ref_g)                               # float32(1 - lam) and float32(-lam): the same bits


def test_metadata_and_scalar_algebra_do_not_materialise():
    x, y, l1, ss = _pair()
    a = LU.LazyLoss(l1, ss, (0, 1, 0))
    assert a.shape == torch.Size([]) and a.dtype == torch.float32 and a.dim() == 0 and a.requires_grad and not a.is_cuda
    for expr, want in (((a * 2.0), (0, 2, 0)), ((2.0 * a), (0, 2, 0)), ((a / 4), (0, 0.25, 0)), ((-a), (0, -1, 0)), ((a + 1), (1, 1, 0)),
                       ((1 + a), (1, 1, 0)), ((a - 0.5), (-0.5, 1, 0)), ((1.0 - a), (1, -1, 0)), ((a + a), (0, 2, 0)), ((a - a), (0, 0, 0)),
                       (a.mean(), (0, 1, 0)), (a.detach(), (0, 1, 0))):
        assert isinstance(expr, LU.LazyLoss) and expr._lz[2] == tuple(float(v) for v in want)
    assert not a.detach().requires_grad
    assert float(a) == pytest.approx(0.25) and math.isfinite(a)


def test_everything_else_falls_back_to_real_tensors():
    x, y, l1, ss = _pair()
    a, s = LU.LazyLoss(l1, ss, (0.5, 2, 0)), LU.LazyLoss(l1, ss, (0, 0, 1))
    t = a * torch.tensor(2.0)
    assert type(t) is torch.Tensor and float(t) == pytest.approx(2.0)
    assert type(a + torch.tensor(1.0)) is torch.Tensor
    assert bool(a > 0) and not bool(torch.isnan(a)) and float(2.0 / a) == pytest.approx(2.0)
    assert "tensor(1." in repr(a)
    assert torch.equal(torch.stack([a.detach(), s.detach()]), torch.tensor([1.0, 0.9]))
    other = LU.LazyLoss(l1 * 1.0, ss, (0, 1, 0))                            # another node: no coefficient merge
    assert type(a + other) is torch.Tensor
    (a * torch.tensor(3.0)).backward()                                      # the materialised tensor carries the graph
    assert float(x.grad) == pytest.approx(6.0)
    with pytest.raises(RuntimeError):
        LU.LazyLoss(l1.detach(), ss.detach(), (0, 1, 0)).backward()


def test_lazy_mode_is_per_thread_and_off_by_default():
    import threading
    assert not LU._lazy_on()
    prev = LU.set_lazy(True)
    try:
        seen = []
        th = threading.Thread(target=lambda: seen.append(LU._lazy_on()))
        th.start(); th.join()
        assert LU._lazy_on() and seen == [False]
    finally:
        LU.set_lazy(prev)
    assert not LU._lazy_on()
