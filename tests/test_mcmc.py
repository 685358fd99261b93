"""the batched Metropolis driver against a trace captured from the reference's
``Point.MCinv`` (tests/golden/ref_driver.npz: runN=240, chainL=80, seed=7).

CPU tests replay the reference's exact CPython ``random`` stream through ``PythonRandomProposer``
with the forward solve supplied by the CPU oracle (test infrastructure); the GPU test does the same
through the HIP path, and runs the vectorised sampler statistically.
"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from settings import CONT, PERIODS                   # noqa: E402
from pysurfinv_amd.layers_batch import Model1DBatch
from pysurfinv_amd.mcmc import MetropolisBatch
from pysurfinv_amd import brownian

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_driver.npz"))


def oracle_forward(periods):
    from oracle import cport

    def fwd(model, nlay):
        c, u, st = cport.forward_batch(model.cpu().numpy(), periods, 2,
                                       nlay=None if nlay is None else nlay.cpu().numpy(), nthreads=4)
        return torch.from_numpy(c.astype(np.float64)), torch.from_numpy(st)
    return fwd


def replay(device, forward, independent=False):
    runN, chainL, seed = (int(x) for x in G["trace/meta"])
    mb = Model1DBatch(CONT, device=device)
    prop = brownian.PythonRandomProposer(mb.spec, device=device, seed=seed)
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"],
                         device=device, proposer=prop, forward=forward, independent=independent)
    chunks = [mc.run(1, chainL, init_first=(i == 0))[0] for i in range(runN // chainL)]
    return torch.cat(chunks, dim=0).cpu().numpy(), mc


def check_trace(track):
    ref = G["trace/mcTrack"]
    assert track.shape == ref.shape == (240, 16)
    # proposals (columns 3:) come from the same random stream: identical as long as every accept
    # decision agrees, which the accepted column checks
    assert np.array_equal(track[:, 2], ref[:, 2]), np.nonzero(track[:, 2] != ref[:, 2])
    assert np.abs(track[:, 3:] - ref[:, 3:]).max() < 1e-12
    assert np.abs(track[:, 0] / ref[:, 0] - 1).max() < 2e-4         # misfit = sqrt(chi2/N)
    ok = ref[:, 1] > 1e-300
    assert np.abs(np.log(track[ok, 1]) - np.log(ref[ok, 1])).max() < 5e-2   # L = exp(-chi2/2), chi2 up to ~1e3


def test_replay_reference_trace_cpu():
    track, mc = replay("cpu", oracle_forward(G["trace/periods"].astype(np.float32)))
    check_trace(track)
    assert mc.n_forward == 240                                      # one forward solve per step


def test_misfit_formula_and_failure_convention():
    mb = Model1DBatch(CONT)
    P = len(PERIODS)

    def fake(model, nlay):                                          # c = 3.5 everywhere, stack 1 fails
        c = torch.full((model.shape[0], P), 3.5, dtype=torch.float64)
        st = torch.zeros(model.shape[0], dtype=torch.int32); st[1] = 1
        return c, st
    c_obs = np.full(P, 3.6); c_obs[3] = np.nan                      # masked observation
    mc = MetropolisBatch(mb.spec, mb.to_model, PERIODS, c_obs, np.full(P, 0.01), device="cpu", forward=fake)
    p = torch.as_tensor(mb.spec.v0)[None, :].repeat(3, 1)
    mis, chi, L = mc.misfit(p)
    chi_raw = (P - 1) * (0.1 / 0.01) ** 2                           # 1800 >= 50 -> sqrt(50*chi)
    assert abs(mis[0].item() - np.sqrt(chi_raw / (P - 1))) < 1e-9
    assert abs(chi[0].item() - np.sqrt(chi_raw * 50)) < 1e-9
    assert mis[1].item() == 88888 and chi[1].item() == 88888 and L[1].item() == 0    # point.py:20-21


def test_batched_chains_are_independent_and_bounded():
    """Vectorised sampler, oracle forward: chains differ, stay inside the prior, accept ~ sometimes."""
    mb = Model1DBatch(CONT)
    per = G["trace/periods"].astype(np.float32)
    mc = MetropolisBatch(mb.spec, mb.to_model, per, G["trace/c_obs"], G["trace/uncer"], device="cpu",
                         seed=3, forward=oracle_forward(per))
    tr = mc.run(6, 12).numpy()
    assert tr.shape == (6, 12, 16)
    assert np.allclose(tr[0, 0, 3:], mb.spec.v0)                    # chain 0 starts at the initial model
    assert not np.allclose(tr[1, 0, 3:], tr[2, 0, 3:])              # the others from prior draws
    assert (tr[:, :, 3:] > mb.spec.vmin).all() and (tr[:, :, 3:] < mb.spec.vmax).all()
    assert (tr[:, 0, 2] == 1).all() and set(np.unique(tr[:, :, 2])) <= {0.0, 1.0}


def test_npz_schema(tmp_path):
    tr = np.zeros((2, 5, 16))
    f = MetropolisBatch.save_npz(str(tmp_path), "12.5_45.0", tr, CONT, {"T": PERIODS, "c": [], "uncer": []}, 5)
    d = np.load(f, allow_pickle=True)
    assert set(d.files) == {"mcTrack", "setting", "obs", "invMeta"}  # point.py:82-85
    assert d["mcTrack"].shape == (10, 16) and d["invMeta"][()]["chainL"] == 5


@pytest.mark.gpu
def test_replay_reference_trace_gpu():
    track, mc = replay("cuda:0", None)                               # HIP root-search kernel
    check_trace(track)


@pytest.mark.gpu
def test_replay_reference_trace_gpu_independent_mode():
    """The smooth continental parameterisation with its dense 8-80 s period list is the regime where
    the period-parallel root search equals the reference: the captured trace is reproduced step for
    step in that mode too."""
    track, mc = replay("cuda:0", None, independent=True)
    check_trace(track)


@pytest.mark.gpu
def test_replay_reference_trace_gpu_auto_decomposition():
    """independent="auto" (opt-in): one chain is far below AUTO_INDEP_CHAINS, so the (stack, period) decomposition runs -
    and the reference's trace is still reproduced step for step; a batch above the threshold takes the faithful walk."""
    track, mc = replay("cuda:0", None, independent="auto")
    check_trace(track)
    assert MetropolisBatch.AUTO_INDEP_CHAINS == 3072
    with pytest.raises(ValueError):
        replay("cuda:0", None, independent="sometimes")


@pytest.mark.gpu
def test_gpu_sampler_statistics_match_reference_trace():
    """512 chains x 80 steps on the GPU vs the reference's 3 chains: same acceptance behaviour
    (rate within a generous band) and the best misfit found is at least as good."""
    mb = Model1DBatch(CONT, device="cuda:0")
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"],
                         device="cuda:0", seed=1)
    tr = mc.run(512, 80).cpu().numpy()                               # (default: speculative lock steps of depth 2 for 512 chains)
    ref = G["trace/mcTrack"]
    acc, acc_ref = tr[:, 1:, 2].mean(), ref[np.arange(240) % 80 != 0, 2].mean()
    assert abs(acc - acc_ref) < 0.15, (acc, acc_ref)
    assert tr[:, :, 0].min() <= ref[:, 0].min() * 1.05
    assert mc.auto_spec_depth(512) == 2 and mc.n_forward == 512 * (1 + 3 * 40)      # 79 steps = 40 lock steps of 3 proposals


def test_speculative_sampler_walks_a_consistent_chain():
    """spec_depth > 1: rows must form a valid Metropolis chain (each accepted row becomes the state
    the following proposals are centred on; rejected rows leave it unchanged) and the accept rate
    must match the plain sampler's."""
    mb = Model1DBatch(CONT)
    per = G["trace/periods"].astype(np.float32)
    kw = dict(device="cpu", forward=oracle_forward(per))
    mc = MetropolisBatch(mb.spec, mb.to_model, per, G["trace/c_obs"], G["trace/uncer"], seed=4, **kw)
    tr = mc.run(8, 31, spec_depth=3).numpy()
    assert tr.shape == (8, 31, 16) and mc.n_forward == 8 + 8 * 7 * 10      # 10 lock steps of 2^3-1 proposals
    step = np.asarray(mb.spec.step)
    for c in range(8):
        state = tr[c, 0, 3:]
        for i in range(1, 31):
            prop = tr[c, i, 3:]
            assert (np.abs(prop - state) < 8 * step).all()          # drawn around the CURRENT state
            if tr[c, i, 2] == 1:
                state = prop
    mc1 = MetropolisBatch(mb.spec, mb.to_model, per, G["trace/c_obs"], G["trace/uncer"], seed=5, **kw)
    tr1 = mc1.run(8, 31).numpy()
    assert abs(tr[:, 1:, 2].mean() - tr1[:, 1:, 2].mean()) < 0.2


@pytest.mark.gpu
def test_graphed_sampler_matches_plain_sampler_statistics():
    """run_graphed(): the whole Metropolis step replayed from one HIP graph.  Same chain semantics as
    run(): rows form valid chains, bounds hold, acceptance and best misfit agree with run()."""
    mb = Model1DBatch(CONT, device="cuda:0")
    kw = dict(device="cuda:0")
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], seed=1, **kw)
    torch.cuda.manual_seed(11)
    tg = mc.run_graphed(256, 60).cpu().numpy()
    tp = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], seed=2, **kw).run(256, 60).cpu().numpy()
    assert tg.shape == tp.shape == (256, 60, 16)
    assert np.isfinite(tg).all() and (tg[:, :, 3:] > mb.spec.vmin).all() and (tg[:, :, 3:] < mb.spec.vmax).all()
    step = np.asarray(mb.spec.step)
    for c in range(0, 256, 37):                                     # chain consistency
        state = tg[c, 0, 3:]
        for i in range(1, 60):
            assert (np.abs(tg[c, i, 3:] - state) < 8 * step).all()
            if tg[c, i, 2] == 1:
                state = tg[c, i, 3:]
    assert abs(tg[:, 1:, 2].mean() - tp[:, 1:, 2].mean()) < 0.05
    assert abs(np.median(tg[:, -1, 0]) / np.median(tp[:, -1, 0]) - 1) < 0.25


def test_point_dropin_writes_reference_npz(tmp_path):
    """pysurfinv_amd.point.Point: the reference's constructor / MCinvMP signature; the .npz carries the keys
    and row layout point.py:82-85 writes (CPU run with the oracle forward injected)."""
    from pysurfinv_amd import point as pt
    per = G["trace/periods"]
    p = pt.Point(CONT, localInfo={"note": 1}, periods=per, vels=G["trace/c_obs"], uncers=G["trace/uncer"], device="cpu")
    assert p.setting["Info"]["note"] == 1 and p.initMod.spec.n == 13
    fwd = oracle_forward(per.astype(np.float32))
    p._sampler = lambda seed=None, **kw: MetropolisBatch(p.initMod.spec, p.initMod.to_model, per, G["trace/c_obs"],
                                                         G["trace/uncer"], device="cpu", seed=seed, forward=fwd,
                                                         **{k: v for k, v in kw.items() if k == "isgood"})
    mis, chi, L = p.misfit()
    assert abs(mis - G["trace/mcTrack"][0, 0]) < 2e-4 * mis          # the reference trace starts at the same model
    arr = p.MCinvMP(outdir=str(tmp_path / "MCtest"), pid="7.0_3.0", runN=24, chainL=6, seed=1)
    f = np.load(tmp_path / "MCtest" / "7.0_3.0.npz", allow_pickle=True)
    assert f["mcTrack"].shape == (24, 16) and np.array_equal(f["mcTrack"], arr)
    assert f["invMeta"][()] == {"pid": "7.0_3.0", "chainL": 6}
    assert set(f["obs"][()].keys()) == {"T", "c", "uncer"} and "Crust" in f["setting"][()]
    assert (arr[::6, 2] == 1).all()                                   # every chain's first row is "accepted"


@pytest.mark.gpu
def test_point_dropin_on_gpu(tmp_path):
    """Point.MCinvMP end to end on the device (native parameters -> stack kernel, HIP forward, speculative
    and exact-scan options): file written, chains inside the prior box, misfit of the start model as in the
    reference trace."""
    from pysurfinv_amd import point as pt
    p = pt.Point(CONT, periods=G["trace/periods"], vels=G["trace/c_obs"], uncers=G["trace/uncer"], device="cuda:0")
    mis, chi, L = p.misfit()
    assert abs(mis - G["trace/mcTrack"][0, 0]) < 2e-4 * mis
    arr = p.MCinvMP(outdir=str(tmp_path / "mc"), pid="1_2", runN=64 * 12, chainL=12, seed=3)
    assert arr.shape == (64 * 12, 16) and np.isfinite(arr).all()
    spec = p.initMod.spec
    assert (arr[:, 3:] > spec.vmin).all() and (arr[:, 3:] < spec.vmax).all()
    arr2 = p.MCinvMP(outdir=str(tmp_path / "mc"), pid="1_3", runN=64 * 12, chainL=12, seed=3, spec_depth=3, fast_scan=True)
    assert arr2.shape == arr.shape and np.isfinite(arr2).all()
    assert os.path.exists(tmp_path / "mc" / "1_3.npz")


def test_speculative_sampler_with_per_chain_observations():
    """spec_depth > 1 with c_obs [C, P] (the grid driver's case): proposal m of chain i is held against chain i's
    observations.  Chains whose observations are shifted by +-5 % must end up with different misfits for the SAME
    proposals, and the plain and the speculative sampler must both use every chain's own row."""
    import torch
    from pysurfinv_amd.layers_batch import Model1DBatch
    from pysurfinv_amd.mcmc import MetropolisBatch
    from oracle import cport
    per = G["trace/periods"].astype(np.float32)

    def fwd(model, nlay):
        c, u, st = cport.forward_batch(model.cpu().numpy(), per, 2, nlay=None if nlay is None else nlay.cpu().numpy(), nthreads=2)
        return torch.from_numpy(c.astype(np.float64)), torch.from_numpy(st)
    mb = Model1DBatch(CONT)
    C = 4
    c_obs = np.tile(G["trace/c_obs"], (C, 1)) * np.array([1.0, 1.05, 0.95, 1.0])[:, None]
    unc = np.tile(G["trace/uncer"], (C, 1))
    mc = MetropolisBatch(mb.spec, mb.to_model, per, c_obs, unc, device="cpu", seed=4, forward=fwd)
    tr = mc.run(C, 5, spec_depth=2)
    assert tr.shape == (C, 5, 16) and np.isfinite(tr.numpy()).all()
    # row 0 of every chain after the first is a prior draw; evaluate all four chains' observations on ONE model
    p = torch.as_tensor(np.tile(mb.spec.v0, (C, 1)))
    mis, chi, L = mc.misfit(p)
    assert mis[0] == mis[3] and mis[1] != mis[0] and mis[2] != mis[0]
    mis2, _, _ = mc.misfit(p[[0, 0]], rows=torch.tensor([1, 2]))
    assert mis2[0] == mis[1] and mis2[1] == mis[2]
    with pytest.raises(ValueError):
        mc.misfit(p[:3])


@pytest.mark.gpu
def test_fused_device_kernels_proposal_and_accept():
    """csrc/surfdisp_mcmc.hip against the torch formulas: (1) proposals stay inside the bounds with the requested step
    (plain Gaussian where the bounds are far), uniform resets cover the prior box; (2) misfit, likelihood and the
    mcTrack row of the accept kernel equal MetropolisBatch.misfit of the same proposals, the state moves exactly where
    the row says 'accepted', better models are always accepted, failed solves never."""
    import ctypes
    from pysurfinv_amd import _lib
    dev = torch.device("cuda:0")
    mb = Model1DBatch(CONT, device=dev)
    C, N, P = 20000, mb.spec.n, len(G["trace/periods"])
    rng = np.random.default_rng(0)
    c_obs = np.tile(G["trace/c_obs"], (C, 1)) * (1 + 0.01 * rng.standard_normal((C, 1)))
    c_obs[::7, 3] = np.nan                                           # masked periods
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, np.tile(G["trace/uncer"], (C, 1)), device=dev, seed=11)
    assert mc.fused_available()
    L = _lib.lib()
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    pr = mc.proposer
    v = torch.as_tensor(mb.spec.v0, device=dev)[None, :].repeat(C, 1).contiguous()
    new = torch.empty_like(v)
    _lib.check(L.surfdisp_mcmc_propose_device(None, C, N, ptr(v), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step), 11, 1, 0, ptr(new), 0))
    torch.cuda.synchronize()
    assert bool(((new > pr.vmin) & (new < pr.vmax)).all())
    d = (new - v).cpu().numpy()
    assert abs(d[:, 4].std() / mb.spec.step[4] - 1) < 0.03 and abs(d[:, 4].mean()) < 3 * mb.spec.step[4] / 100
    new2 = torch.empty_like(v)
    _lib.check(L.surfdisp_mcmc_propose_device(None, C, N, ptr(v), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step), 11, 2, 0, ptr(new2), 0))
    assert float((new2 - new).abs().max()) > 0                       # another counter, another draw
    rs = torch.empty_like(v)
    _lib.check(L.surfdisp_mcmc_propose_device(None, C, N, ptr(v), ptr(pr.vmin), ptr(pr.vmax), ptr(pr.step), 11, 3, 1, ptr(rs), 0))
    torch.cuda.synchronize()
    assert bool(((rs >= pr.vmin) & (rs <= pr.vmax)).all()) and abs(rs[:, 3].mean().item() - 35.0) < 0.3
    # accept kernel: two lock steps through the sampler's own entry, rows compared with the torch misfit
    p = rs.clone().contiguous()
    row = torch.zeros((C, 3 + N), dtype=torch.float64, device=dev)
    mc.fused_step(p, row=row, row_stride=3 + N, first=True)
    torch.cuda.synchronize()
    mis, chi, Lk = mc.misfit(rs)
    assert float((row[:, 0] - mis).abs().max()) < 1e-9 and float((row[:, 1] - Lk).abs().max()) < 1e-12
    assert bool((row[:, 2] == 1).all()) and torch.equal(row[:, 3:], rs) and torch.equal(p, rs)
    chi0 = mc._fz["chi"].clone()
    assert float((chi0 - chi).abs().max()) < 1e-9
    before = p.clone()
    mc.fused_step(p, row=row, row_stride=3 + N)
    torch.cuda.synchronize()
    prop = row[:, 3:].contiguous()
    mis1, chi1, L1 = mc.misfit(prop)
    assert float((row[:, 0] - mis1).abs().max()) < 1e-9 and float((row[:, 1] - L1).abs().max()) < 1e-12
    acc = row[:, 2] > 0.5
    assert torch.equal(p[acc], prop[acc]) and torch.equal(p[~acc], before[~acc])
    assert bool(acc[chi1 < chi0].all())                              # better models are always accepted
    assert not bool(acc[mis1 >= 88888].any()) or bool((chi0[acc & (mis1 >= 88888)] >= 88888).all())
    assert 0.05 < float(acc.double().mean()) < 0.95
    assert float((mc._fz["chi"] - torch.where(acc, chi1, chi0)).abs().max()) < 1e-9


@pytest.mark.gpu
def test_fused_run_records_consistent_rows():
    """MetropolisBatch.run on the device path: every recorded row's misfit equals the eagerly recomputed misfit of its
    parameters, rows form a valid chain, acceptance comparable with the torch-glue path."""
    dev = torch.device("cuda:0")
    mb = Model1DBatch(CONT, device=dev)
    kw = dict(device=dev, seed=4)
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], **kw)
    tr = mc.run(256, 40, spec_depth=1)
    assert mc.n_forward == 256 * 40
    for k in (0, 1, 17, 39):
        mis, _, Lk = mc.misfit(tr[:, k, 3:].contiguous())
        assert float((mis - tr[:, k, 0]).abs().max()) < 1e-9, k
    assert bool((tr[:, 0, 2] == 1).all()) and np.allclose(tr[0, 0, 3:].cpu().numpy(), mb.spec.v0)
    mc2 = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], **kw)
    tr2 = mc2.run(256, 40, fused=False)
    a1, a2 = float(tr[:, 1:, 2].mean()), float(tr2[:, 1:, 2].mean())
    assert abs(a1 - a2) < 0.05, (a1, a2)


def _check_chains(tr, mb, mc, rows=(0, 1, 2, 3, 17, 38, 39)):
    """mcTrack [C, chainL, 3+N]: recorded misfits are the misfits of the recorded proposals; every proposal lies within a
    few step widths of the state it was drawn from (the last accepted row before it) and inside the bounds."""
    for k in rows:
        mis, _, Lk = mc.misfit(tr[:, k, 3:].contiguous())
        assert float((mis - tr[:, k, 0]).abs().max()) < 1e-9 and float((Lk - tr[:, k, 1]).abs().max()) < 1e-12, k
    t = tr.cpu().numpy()
    step = np.asarray(mb.spec.step)
    assert (t[:, :, 3:] > mb.spec.vmin).all() and (t[:, :, 3:] < mb.spec.vmax).all() and (t[:, 0, 2] == 1).all()
    for c in range(0, t.shape[0], 23):
        state = t[c, 0, 3:]
        for i in range(1, t.shape[1]):
            assert (np.abs(t[c, i, 3:] - state) < 8 * step).all(), (c, i)
            if t[c, i, 2] == 1:
                state = t[c, i, 3:]


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [None, 2, 4])
def test_fused_speculative_lock_steps(depth):
    """The device path's speculative lock step (surfdisp_mcmc_propose_tree_device / _accept_tree_device): depth steps per
    batched solve of C * (2^depth - 1) proposals.  Rows form valid chains (every proposal drawn around the state the chain
    was in, recorded misfit = misfit of the recorded proposal), acceptance as the plain sampler's, also with a chain length
    that is not a multiple of the depth and with per-chain observations."""
    dev = torch.device("cuda:0")
    mb = Model1DBatch(CONT, device=dev)
    C, chainL = 256, 40
    rng = np.random.default_rng(3)
    c_obs = np.tile(G["trace/c_obs"], (C, 1)) * (1 + 0.01 * rng.standard_normal((C, 1)))
    unc = np.tile(G["trace/uncer"], (C, 1))
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, unc, device=dev, seed=4)
    tr = mc.run(C, chainL, spec_depth=depth)
    d = depth or 3
    assert mc.auto_spec_depth(100) == 4 and mc.auto_spec_depth(137) == 3 and mc.auto_spec_depth(C) == 3 and mc.auto_spec_depth(293) == 2 and mc.auto_spec_depth(683) == 1
    assert mc.n_forward == C * (1 + ((1 << d) - 1) * -(-(chainL - 1) // d))
    _check_chains(tr, mb, mc)
    mc1 = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, unc, device=dev, seed=4)
    tr1 = mc1.run(C, chainL, spec_depth=1)
    assert torch.equal(tr[:, 0], tr1[:, 0])                            # same start models, same first rows
    a, a1 = float(tr[:, 1:, 2].mean()), float(tr1[:, 1:, 2].mean())
    assert abs(a - a1) < 0.04, (a, a1)
    assert abs(float(tr[:, -1, 0].median()) / float(tr1[:, -1, 0].median()) - 1) < 0.3
    # depth 1 through the tree entries is the plain lock step, bit for bit (same random streams)
    if depth == 2:
        mcA = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, unc, device=dev, seed=9)
        mcB = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, unc, device=dev, seed=9)
        pA = mcA.reset(C).contiguous(); pB = pA.clone()
        rA = torch.zeros((C, 3 + mb.spec.n), dtype=torch.float64, device=dev); rB = torch.zeros_like(rA)
        mcA.fused_step(pA, first=True); mcB.fused_step(pB, first=True)
        for _ in range(3):
            mcA.fused_step(pA, row=rA, row_stride=3 + mb.spec.n)
            mcB.fused_tree_step(pB, 1, 1, row=rB, row_stride=3 + mb.spec.n, step_stride=3 + mb.spec.n)
            assert torch.equal(pA, pB) and torch.equal(rA, rB)


@pytest.mark.gpu
def test_chain_groups_do_not_change_the_chains():
    """MetropolisBatch.chain_groups: the chains advance as groups on their own streams; the random streams are indexed by
    the chain's index in the whole sampler (chain0 of the C ABI), so every mcTrack row is the same for 1, 2 and 3 groups -
    with per-chain observations and per-point local information threaded through the groups."""
    dev = torch.device("cuda:0")
    mb = Model1DBatch(CONT, device=dev)
    C, chainL, P = 600, 12, len(G["trace/periods"])
    rng = np.random.default_rng(5)
    c_obs = np.tile(G["trace/c_obs"], (C, 1)) * (1 + 0.01 * rng.standard_normal((C, 1)))
    c_obs[::5, 2] = np.nan
    unc = np.tile(G["trace/uncer"], (C, 1))
    tracks = []
    for groups in (1, 2, 3):
        mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], c_obs, unc, device=dev, seed=21)
        first = (torch.arange(C, device=dev) % 50) == 0
        tr = mc.run(C, chainL, init_first=False, _init_mask=first, groups=groups, spec_depth=1)
        torch.cuda.synchronize()
        assert mc.n_forward == C * chainL and mc._counter == chainL
        assert (mc.chain_groups(C, groups) is None) == (groups == 1)
        tracks.append(tr.cpu().numpy())
        # a second run continues the counter: other random numbers
        tr2 = mc.run(C, chainL, init_first=False, _init_mask=first, groups=groups, spec_depth=1)
        assert mc._counter == 2 * chainL and not np.array_equal(tr2.cpu().numpy()[:, 1:], tracks[-1][:, 1:])
    assert np.array_equal(tracks[0], tracks[1]) and np.array_equal(tracks[0], tracks[2])
    acc = tracks[0][:, 1:, 2].mean()
    assert 0.05 < acc < 0.95
    # default rule: one group below GROUP_MIN_CHAINS, two from there on
    mc = MetropolisBatch(mb.spec, mb.to_model, G["trace/periods"], G["trace/c_obs"], G["trace/uncer"], device=dev, seed=1)
    assert mc.chain_groups(MetropolisBatch.GROUP_MIN_CHAINS - 1) is None
    assert mc.chain_groups(MetropolisBatch.GROUP_MIN_CHAINS).G == 2


# ------------------------------------------------------------------ PostPoint (point.py:134-175, 307-335)
GP = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_post.npz"), allow_pickle=True)
POST_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "post_trace.npz")


def _check_post(p, key, tol_misfit):
    assert np.array_equal(p.MCparas, GP[f"{key}/MCparas"])
    assert np.array_equal(p.minMod.params, GP[f"{key}/min_params"]) and p.minMod.misfit == float(GP[f"{key}/min_misfit"])
    assert p.minMod.L == float(GP[f"{key}/min_L"]) and p.thres == float(GP[f"{key}/thres"])
    assert np.array_equal(p.accFinal, GP[f"{key}/accFinal"])
    assert np.abs(p.avgMod.params - GP[f"{key}/avg_params"]).max() < 1e-12
    assert np.array_equal(p._loadValues(), GP[f"{key}/values"]) and np.array_equal(p._loadValues(indVars=[0, 3, 7]), GP[f"{key}/values_sub"])
    assert len(list(p._model_generator())) == int(GP[f"{key}/accFinal"].sum())
    vz = p._loadValues(zdeps=GP["zdeps"])                               # Vs at depth of every final model (Model1D.value)
    assert vz.shape == GP[f"{key}/values_z"].shape and np.nanmax(np.abs(vz - GP[f"{key}/values_z"])) < 1e-9
    if tol_misfit is not None:
        assert abs(p.avgMod.misfit / float(GP[f"{key}/avg_misfit"]) - 1) < tol_misfit
        assert abs(p.avgMod.L / float(GP[f"{key}/avg_L"]) - 1) < 50 * tol_misfit


@pytest.mark.parametrize("tmc,key", [(True, "tmc"), (False, "raw")])
def test_postpoint_matches_reference_postpoint(tmc, key):
    """The reference's PostPoint on the reference's own 240-row trace file (tests/golden/make_golden_post.py): parameters
    after the true-Markov-chain substitution, minimum-misfit model, threshold, final mask, average model, _loadValues -
    exactly; the average model's misfit through the CPU oracle as forward (the checker standing in for the device here)."""
    from pysurfinv_amd.point import PostPoint
    p = PostPoint(POST_NPZ, trueMarkovChain=tmc, device=None, _forward=oracle_forward(np.asarray(G["trace/periods"], np.float32)))
    assert p.N == 240 and p.invMeta["chainL"] == 80 and list(p.obs["T"]) == list(G["trace/periods"])
    _check_post(p, key, 1e-6)
    q = PostPoint(POST_NPZ, trueMarkovChain=tmc, device=None)          # no device, no forward: everything but the one solve
    assert q.avgMod.misfit is None and np.array_equal(q.accFinal, p.accFinal)


def test_postpoint_reads_files_with_the_references_toYML_setting(tmp_path):
    """The reference's Point.MCinv stores `setting = initMod.toYML()` (point.py:83): LayerName keys ('LandSediment',
    'LandCrust', ...) and every random-walk entry as [v, vmin, vmax, step].  Such a file loads like one with the original
    setting (the reference's own PostPoint cannot read its land files back: its layerClassDict lacks the two land names)."""
    import json
    from pysurfinv_amd.point import PostPoint
    GGr = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_grids.npz"))
    toyml = json.loads(str(GGr["cont/toyml"]))[0]
    assert "LandSediment" in toyml and "LandCrust" in toyml
    src = np.load(POST_NPZ, allow_pickle=True)
    f = os.path.join(str(tmp_path), "ref_style.npz")
    np.savez_compressed(f, mcTrack=src["mcTrack"], setting=toyml, obs=src["obs"][()], invMeta=src["invMeta"][()])
    fwd = oracle_forward(np.asarray(G["trace/periods"], np.float32))
    a = PostPoint(f, device=None, _forward=fwd)
    b = PostPoint(POST_NPZ, device=None, _forward=fwd)
    assert np.array_equal(a.MCparas, b.MCparas) and np.array_equal(a.avgMod.params, b.avgMod.params)
    assert a.avgMod.misfit == b.avgMod.misfit and a.thres == b.thres
    assert np.allclose(a.initMod.spec.vmin, b.initMod.spec.vmin) and np.allclose(a.initMod.spec.step, b.initMod.spec.step)


@pytest.mark.gpu
def test_postpoint_on_the_device_and_on_our_own_files(tmp_path):
    from pysurfinv_amd.point import Point, PostPoint
    p = PostPoint(POST_NPZ, device="cuda:0")
    _check_post(p, "tmc", 1e-5)
    # a file written by this package's Point.MCinvMP reads back the same way
    pt = Point(CONT, periods=list(G["trace/periods"]), vels=list(G["trace/c_obs"]), uncers=list(G["trace/uncer"]), device="cuda:0")
    arr = pt.MCinvMP(outdir=str(tmp_path), pid="here", runN=600, chainL=60, seed=3)
    q = PostPoint(os.path.join(str(tmp_path), "here.npz"), device="cuda:0")
    assert q.N == 600 and np.array_equal(q.misfits, arr[:, 0]) and q.pid == "here"
    assert q.minMod.misfit == np.nanmin(arr[:, 0]) and q.accFinal.sum() >= 1 and np.isfinite(q.avgMod.misfit)
    assert q.avgMod.misfit < q.thres * 1.5
