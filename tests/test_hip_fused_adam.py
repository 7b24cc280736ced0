"""ls2fm.optim.FusedAdam (csrc/adam.hip through the C ABI) against torch.optim.Adam on the GPU: same trajectory over
several steps incl. an ExponentialLR schedule, weight decay, ragged sizes (non-multiples of 4, unaligned views) and a
state_dict round trip."""
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(1 << 20) + 3, (64, 35), (64,), (17, 64), (1,), (3, 64), (5,)]
    ps = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    big = torch.randn(1000, generator=g).to(DEV)
    ps.append(big[1:998])                       # a 4-byte aligned, not 16-byte aligned contiguous view
    return ps


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_fused_adam_follows_torch_adam(wd):
    from ls2fm.optim import FusedAdam
    a = [torch.nn.Parameter(p.clone()) for p in _params(1)]
    b = [torch.nn.Parameter(p.clone()) for p in _params(1)]
    oa = torch.optim.Adam(a, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=wd)
    ob = FusedAdam(b, lr=1e-2, betas=(0.9, 0.99), eps=1e-15, weight_decay=wd)
    sa = torch.optim.lr_scheduler.ExponentialLR(oa, 0.9)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, 0.9)
    g = torch.Generator().manual_seed(7)
    for it in range(6):
        for pa, pb in zip(a, b):
            gr = torch.randn(pa.shape, generator=g).to(DEV) * (10.0 ** (it - 3))
            pa.grad = gr.clone(); pb.grad = gr.clone()
        if it == 2:
            b[4].grad = None; a[4].grad = None           # a parameter without gradient keeps its own step count
        oa.step(); ob.step(); sa.step(); sb.step()
        if it == 3:                                      # state_dict round trip
            ob2 = FusedAdam(b, lr=1.0)
            ob2.load_state_dict(ob.state_dict())
            ob = ob2
            sb.optimizer = ob                            # the scheduler keeps driving the re-created optimizer's groups
    for pa, pb in zip(a, b):
        assert rel_err(pb.detach().cpu(), pa.detach().cpu()) < 2e-6
    for pa, pb in zip(a, b):
        for k in ("exp_avg", "exp_avg_sq"):
            assert rel_err(ob.state[pb][k].cpu(), oa.state[pa][k].cpu()) < 2e-6, k
        assert int(ob.state[pb]["step"]) == int(oa.state[pa]["step"])


def test_fused_adam_refuses_cpu_tensors():
    from ls2fm.optim import FusedAdam
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        FusedAdam([p]).step()
