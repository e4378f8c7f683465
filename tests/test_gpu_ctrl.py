"""GPU parity of the fused LPC control transform (golf_rc2lpc_{fwd,bwd}_f32; reference rc2lpc, models/utils.py:581-593,
as used by models/filters.py:91-97): the reference's own output (golden g1), the PyTorch restatement in float64 at the
BASELINE shape, gradients against autograd of that restatement, and the stability property of the step-up."""
import numpy as np
import pytest
import torch

from conftest import rel_err

pytestmark = pytest.mark.gpu


def test_golden_g1(golden):
    from golf_amd import functional as GF

    g = golden("g1_rc2lpc")
    for rc_key, out_key in (("rc", "lpc"), ("rc1", "lpc1")):
        rc = torch.as_tensor(np.asarray(g[rc_key]), dtype=torch.float32).cuda()
        a = GF.rc2lpc(rc).cpu().numpy()
        emax, el2 = rel_err(a, np.asarray(g[out_key]))
        print("g1", rc_key, emax, el2)
        assert emax < 2e-6 and el2 < 2e-6


@pytest.mark.parametrize("B,F,M,max_abs", [(32, 200, 22, 1.0), (3, 7, 1, 1.0), (2, 5, 2, 0.99), (2, 9, 7, 0.9),
                                             (1, 300, 64, 0.95)])
def test_forward_and_backward_vs_torch(B, F, M, max_abs):
    from golf_amd import functional as GF
    from golf_amd.utils import rc2lpc

    gen = torch.Generator().manual_seed(B * 100 + M)
    logits = torch.randn(B, F, M, generator=gen) * 0.7
    gy = torch.randn(B, F, M, generator=gen)
    ref_in = logits.double().requires_grad_(True)
    ref = rc2lpc(torch.tanh(ref_in) * max_abs)
    (ref * gy.double()).sum().backward()
    x = logits.cuda().requires_grad_(True)
    a = GF.rc2lpc_logits(x, max_abs)
    assert a.shape == (B, F, M)
    (a * gy.cuda()).sum().backward()
    emax, el2 = rel_err(a.detach().cpu().numpy(), ref.detach().numpy())
    gmax, gl2 = rel_err(x.grad.cpu().numpy(), ref_in.grad.numpy())
    print(f"rc2lpc B{B} F{F} M{M}: fwd {emax:.2e} {el2:.2e}  grad {gmax:.2e} {gl2:.2e}")
    tol = 2e-5 if M > 32 else 5e-6
    assert emax < tol and el2 < tol and gmax < 10 * tol and gl2 < 10 * tol


def test_module_ctrl_uses_the_fused_kernel_and_matches_cpu_path():
    """The filter's .ctrl transform on GPU tensors (one kernel) equals the same transform on CPU tensors (PyTorch ops),
    values and gradients, and the result is a minimum-phase polynomial (|k| < 1 <=> all roots inside the unit circle)."""
    from golf_amd.audiotensor import AudioTensor
    from golf_amd.filters import LTVMinimumPhaseFilterPrecise

    m = LTVMinimumPhaseFilterPrecise(lpc_order=22, lpc_parameterisation="rc2lpc", max_abs_value=0.99)
    (split, trsfms) = m.ctrl(lambda s_, t_: (s_, t_))((), ())
    assert split[0] == (1, 22)
    gen = torch.Generator().manual_seed(4)
    lg, lo = torch.randn(4, 50, generator=gen) * 0.1, torch.randn(4, 50, 22, generator=gen)
    outs = {}
    for dev in ("cpu", "cuda"):
        l1 = lo.detach().clone().to(dev).requires_grad_(True)
        gain, a = trsfms[0](AudioTensor(lg.to(dev), 240), AudioTensor(l1, 240))
        a.as_tensor().square().sum().backward()
        outs[dev] = (gain.as_tensor().detach().cpu(), a.as_tensor().detach().cpu(), l1.grad.cpu())
        assert a.hop_length == 240
    for what, u, v in zip(("gain", "a", "grad"), outs["cpu"], outs["cuda"]):
        emax, el2 = rel_err(v.numpy(), u.numpy())        # both sides fp32: max-norm relative difference
        print("ctrl cpu vs gpu", what, emax, el2)
        assert emax < 1e-4 and el2 < 1e-4
    poly = np.concatenate([np.ones((4, 50, 1)), outs["cuda"][1].double().numpy()], -1)
    radii = np.abs(np.roots(poly[0, 0]))
    assert radii.max() < 1.0


def test_errors():
    from golf_amd import _lib
    from golf_amd import functional as GF

    with pytest.raises(_lib.GolfError):
        GF.rc2lpc_logits(torch.zeros(2, 3, 65, device="cuda"))   # order above the kernel's limit
    with pytest.raises(_lib.GolfError):
        GF.rc2lpc_logits(torch.zeros(2, 3, 4))                     # CPU tensor: there is no CPU path in the library


@pytest.mark.parametrize("rep", ["coef", "conj", "real"])
@pytest.mark.parametrize("B,F,K", [(32, 200, 11), (2, 9, 1), (3, 17, 13), (1, 40, 32)])
def test_biquad_parameterisations_vs_torch(rep, B, F, K):
    """golf_sos2lpc_{fwd,bwd}_f32 (logits -> K biquads -> direct form, the ISMIR'23 parameterisations) against the
    PyTorch restatement pinned by golden g2, in float64, values and gradients."""
    from golf_amd import functional as GF
    from golf_amd.utils import biquads2lpc, get_logits2biquads

    gen = torch.Generator().manual_seed(K * 10 + len(rep))
    logits = torch.randn(B, F, 2 * K, generator=gen) * 0.8
    gy = torch.randn(B, F, 2 * K, generator=gen)
    ref_in = logits.double().requires_grad_(True)
    ref = biquads2lpc(get_logits2biquads(rep, 0.97)(ref_in.view(B, F, K, 2)))
    (ref * gy.double()).sum().backward()
    x = logits.cuda().requires_grad_(True)
    a = GF.biquad_logits2lpc(x, rep, 0.97)
    assert a.shape == (B, F, 2 * K)
    (a * gy.cuda()).sum().backward()
    emax, el2 = rel_err(a.detach().cpu().numpy(), ref.detach().numpy())
    gmax, gl2 = rel_err(x.grad.cpu().numpy(), ref_in.grad.numpy())
    print(f"sos2lpc {rep} B{B} F{F} K{K}: fwd {emax:.2e} {el2:.2e}  grad {gmax:.2e} {gl2:.2e}")
    tol = 5e-5 if K > 16 else 5e-6          # a degree-64 product in fp32
    assert emax < tol and el2 < tol and gmax < 10 * tol and gl2 < 10 * tol


def test_golden_g2_biquads(golden):
    """The reference's own logits -> biquads -> direct form (g2, default pole radius 0.99) through the fused kernel."""
    from golf_amd import functional as GF

    g = golden("g2_biquads")
    lg = torch.as_tensor(np.asarray(g["logits"]), dtype=torch.float32)
    flat = lg.reshape(*lg.shape[:-2], -1).cuda()
    for rep in ("coef", "conj", "real"):
        a = GF.biquad_logits2lpc(flat, rep, 0.99).cpu().numpy()
        emax, el2 = rel_err(a, np.asarray(g[f"lpc_{rep}"]))
        print("g2", rep, emax, el2)
        assert emax < 5e-6 and el2 < 5e-6
