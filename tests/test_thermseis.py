"""thermal -> seismic conversions and the OceanMantleHybrid layer against
fixtures captured from the imported reference (tests/golden/make_golden_therm.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from pysurfinv_amd import thermseis as ts
from pysurfinv_amd.layers_batch import Model1DBatch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from settings_therm import HYBRID_RITZ, HYBRID_YAMA, HYBRID_STATIC, HYBRID_STATIC_YAMA, PERIODS   # noqa: E402

G = np.load(os.path.join(HERE, "golden", "ref_therm.npz"))
AGES = torch.as_tensor(G["ages"])


def _ther(tag):
    if tag == "default":
        return ts.hscm(AGES)
    return ts.hscm(AGES, zdeps=torch.as_tensor(G["hscm/custom/zdeps"]), Tp=1350)


def rel(a, b):
    a = a.numpy() if torch.is_tensor(a) else a
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize("tag", ["default", "custom"])
def test_hscm_fields(tag):
    th = _ther(tag)
    assert rel(th.P, G[f"hscm/{tag}/P"]) < 1e-14
    assert rel(th.T, G[f"hscm/{tag}/T"]) < 1e-12          # erf of two libraries + a 16-step bisection
    assert rel(th.rho, G[f"hscm/{tag}/rho"]) < 1e-12


@pytest.mark.parametrize("tag", ["default", "custom"])
def test_seismic_conversions(tag):
    th = _ther(tag)
    for k in ("raw", "corrected", "from_thermal"):
        assert rel(ts.ritz_vs(th, rho_type=k)[0], G[f"ritz/{tag}/{k}"]) < 1e-12, k
    vs, qs, vsu = ts.ruan(th, period=1)
    assert rel(vs, G[f"ruan1/{tag}/vs"]) < 1e-11
    assert rel(qs, G[f"ruan1/{tag}/qs"]) < 1e-10
    assert rel(vsu, G[f"ruan1/{tag}/vsu"]) < 1e-12
    vs, qs, _ = ts.ruan(th, period=10)
    assert rel(vs, G[f"ruan10/{tag}/vs"]) < 1e-11 and rel(qs, G[f"ruan10/{tag}/qs"]) < 1e-10
    for k in ("Takei2017", "Hirschmann2009", "Ruan2018"):
        vs, qs, _ = ts.yata(th, Tm=k, period=50)
        assert rel(vs, G[f"yata50/{tag}/{k}/vs"]) < 1e-11, k
        assert rel(qs, G[f"yata50/{tag}/{k}/qs"]) < 1e-10, k
    assert rel(ts.bass(th), G[f"other/{tag}/bass"]) < 1e-12
    assert rel(ts.stix(th), G[f"other/{tag}/stix"]) < 1e-12
    assert rel(ts.pm13(th, period=1), G[f"other/{tag}/pm13"]) < 1e-10
    assert rel(ts.yata_unrelaxed(th), G[f"other/{tag}/yata_unrelaxed"]) < 1e-12


def test_behn2009():
    q, sf = ts.behn2009_shear(1.0, 1e-3, torch.as_tensor(G["behn/T"]), torch.as_tensor(G["behn/P"]), 100)
    assert rel(q, G["behn/Qinv"]) < 1e-12 and rel(sf, G["behn/shear"]) < 1e-12


def test_cubic_spline_matches_scipy():
    from scipy.interpolate import CubicSpline
    rng = np.random.default_rng(3)
    B, N = 40, 31
    x = np.sort(rng.uniform(0, 100, (B, N)), axis=1)
    y = rng.normal(size=(B, N))
    keep = rng.uniform(size=(B, N)) < 0.6
    for b, n in enumerate([2, 3, 4, 5]):                  # short rows: line, parabola, smallest general cases
        keep[b] = False
        keep[b, rng.choice(N, n, replace=False)] = True
    keep[4, :5] = False                                    # extrapolation at both ends
    keep[4, -5:] = False
    keep[5] = True
    out = ts.cubic_spline_through(torch.as_tensor(x), torch.as_tensor(y), torch.as_tensor(keep)).numpy()
    for b in range(B):
        want = CubicSpline(x[b, keep[b]], y[b, keep[b]])(x[b])
        assert np.max(np.abs(out[b] - want)) <= 1e-9 * max(1.0, np.max(np.abs(want))), b


@pytest.mark.parametrize("name,setting", [("hyb_ritz", HYBRID_RITZ), ("hyb_yama", HYBRID_YAMA)])
def test_hybrid_layer_stack(name, setting):
    mb = Model1DBatch(setting, device="cpu")
    params = torch.as_tensor(G[f"{name}/params"])
    assert mb.spec.n == params.shape[1]
    (h, vs, vp, rho, qs, qp), nlay = mb.seis_prop_layers(params)
    want = G[f"{name}/layers"]
    assert np.array_equal(nlay.numpy(), G[f"{name}/nlay"])
    for a, w, nm in zip((h, vs, vp, rho, qs, qp), want.transpose(1, 0, 2), "h vs vp rho qs qp".split()):
        err = np.max(np.abs(a.numpy() - w) / np.maximum(np.abs(w), 1e-9))
        assert err < 1e-10, (nm, err)                       # measured 2e-14
    assert np.max(np.abs(mb.layers[-1]["zmelt_last"].numpy() - G[f"{name}/zmelt"])) < 1e-12
    assert mb.native_descriptor() is None                 # thermal layers take the torch path


def test_hybrid_forward_oracle():
    """Dispersion predicted from the batched thermal stack = the reference's own forward() values."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import cport
    mb = Model1DBatch(HYBRID_RITZ, device="cpu")
    params = torch.as_tensor(G["hyb_ritz/params"][:6])
    model, nlay = mb.to_model_torch(params)
    per = np.asarray(PERIODS, np.float32)
    c, _, st = cport.forward_batch(model.numpy(), per, 2, nlay=nlay.numpy())
    assert (st == 0).all()
    assert np.max(np.abs(c - G["hyb_ritz/c"])) < 2e-5


def test_metropolis_on_thermal_model():
    """The batched sampler drives the thermal parameterisation (ThermAge is a random-walk variable
    like any other): chains stay inside the prior box and the misfit of the truth is ~0."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import cport
    from pysurfinv_amd.mcmc import MetropolisBatch
    per = np.asarray(PERIODS, np.float32)

    def fwd(model, nlay):
        c, u, st = cport.forward_batch(model.cpu().numpy(), per, 2, nlay=nlay.cpu().numpy(), nthreads=4)
        return torch.from_numpy(c.astype(np.float64)), torch.from_numpy(st)
    mb = Model1DBatch(HYBRID_RITZ, device="cpu")
    c_obs = G["hyb_ritz/c"][1]
    mc = MetropolisBatch(mb.spec, mb.to_model, PERIODS, c_obs, np.full(len(PERIODS), 0.01), device="cpu",
                         forward=fwd, seed=3)
    mis, _, _ = mc.misfit(torch.as_tensor(G["hyb_ritz/params"][1:2]))
    assert mis.item() < 2e-3
    track = mc.run(n_chains=4, chainL=12).numpy()
    assert track.shape == (4, 12, 3 + mb.spec.n)
    p = track[:, :, 3:]
    assert (p > mb.spec.vmin).all() and (p < mb.spec.vmax).all()
    assert np.isfinite(track[:, :, 0]).all()


@pytest.mark.gpu
def test_hybrid_stack_and_forward_on_gpu():
    """Thermal stack built on the device (torch path), joint Rayleigh + Love forward through the HIP
    library: Rayleigh phase velocities = the reference's forward() values; Love against the oracle."""
    sys.path.insert(0, os.path.dirname(HERE))
    from oracle import cport
    from pysurfinv_amd.forward import forward_batch_torch
    mb = Model1DBatch(HYBRID_RITZ, device="cuda:0")
    params = torch.as_tensor(G["hyb_ritz/params"], device="cuda:0")
    model, nlay = mb.to_model(params)
    assert np.max(np.abs(model.cpu().numpy()[:, 1] - G["hyb_ritz/layers"][:, 1])) < 1e-6
    per = torch.as_tensor(np.asarray(PERIODS, np.float32), device="cuda:0")
    c, u, st = forward_batch_torch(model, per, kind=2, nlay=nlay)
    assert int((st != 0).sum()) == 0
    c2, st2 = mb.forward(params, PERIODS)                 # Model1D.forward() counterpart
    assert float((c2 - c).abs().max()) < 5e-6 and int(st2.abs().sum()) == 0   # team sizes may differ (phase-only)
    assert np.max(np.abs(c.cpu().numpy()[:6] - G["hyb_ritz/c"])) < 2e-5
    for kind in (1, 2):
        co, uo, so = cport.forward_batch(model.cpu().numpy(), np.asarray(PERIODS, np.float32), kind,
                                         nlay=nlay.cpu().numpy())
        cg, ug, sg = forward_batch_torch(model, per, kind=kind, nlay=nlay)
        assert np.array_equal(sg.cpu().numpy(), so)
        assert np.max(np.abs(cg.cpu().numpy() - co)) < 2e-5, kind
        fin = np.isfinite(uo)                              # the reference's Love U is NaN on some water stacks
        assert np.max(np.abs(ug.cpu().numpy() - uo)[fin]) < 5e-5, kind


@pytest.mark.gpu
def test_graphed_metropolis_on_thermal_model():
    """Static thermal parameterisation: the whole Metropolis step (proposal, two half-space models,
    mineral physics, spline, HIP forward, accept/reject) is captured in one HIP graph; every recorded
    row's misfit equals the eagerly recomputed misfit of its parameters."""
    from pysurfinv_amd.mcmc import MetropolisBatch
    mb = Model1DBatch(HYBRID_STATIC, device="cuda:0")
    assert mb._static_sig is not None
    mc = MetropolisBatch(mb.spec, mb.to_model, PERIODS, G["hyb_ritz/c"][0] * 1.002, np.full(len(PERIODS), 0.01),
                         device="cuda:0", seed=5)
    track = mc.run_graphed(16, 10)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(track).all())
    for k in (0, 1, 5, 9):
        mis, _, _ = mc.misfit(track[:, k, 3:].contiguous())
        assert float((mis - track[:, k, 0]).abs().max()) < 1e-9, k
    acc = track[:, 1:, 2]
    assert 0.0 < float(acc.mean()) < 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("setting", [HYBRID_STATIC, HYBRID_STATIC_YAMA], ids=["ritzwoller", "yamauchi_tp"])
def test_native_thermal_kernel_matches_torch_mirror(setting):
    """surfdisp_thermal_kernel + surfdisp_layers_kernel (HIP) against the torch mirror of the same
    rules, itself pinned to the reference at 1e-14: Vs and Qs on the thermal layer's grid in fp64, and
    the assembled fp32 stacks."""
    from pysurfinv_amd.brownian import TorchProposer
    mb = Model1DBatch(setting, device="cuda:0")
    assert mb.native_descriptor() is not None and mb._native_thermal
    params = TorchProposer(mb.spec, "cuda:0", seed=4).reset(300)
    params[0] = torch.as_tensor(mb.spec.v0, device="cuda:0")
    m_nat, nl = mb.to_model(params)
    assert nl is None
    m_ref, nlay = mb.to_model_torch(params)
    assert m_ref.shape == m_nat.shape and int((nlay != m_nat.shape[2]).sum()) == 0
    vs_t, qs_t = mb.layers[-1]["grid_last"]
    npts = vs_t.shape[1]
    sc = mb._thermal_scratch[:, :npts]
    assert float(((sc[:, :, 0] - vs_t).abs() / vs_t.abs()).max()) < 1e-10
    assert float(((sc[:, :, 1] - qs_t).abs() / qs_t.abs()).max()) < 1e-9
    d = (m_nat - m_ref).abs() / m_ref.abs().clamp(min=1e-3)
    assert float(d.max()) < 2e-6
