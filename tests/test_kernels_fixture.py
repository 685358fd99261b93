"""finite-difference sensitivity kernels against the reference's own analytic kernel
fixtures (senskernel-1.0/TEST1/test.phv.{R,L}_0_{T}: (dc/c)/(db/b) per km on a 2 km depth grid, parsed
by tests/golden/make_golden_kernels.py).  A layer's finite-difference kernel
(c(1.001 Vs_i) - c(0.999 Vs_i)) / 0.002 / H_i / c is compared with the mean of the fixture's depth
samples inside that layer, for layers at least 4 km thick; measured deviation <= 3.4 % of the period's
peak (2 km sampling of the fixture + the 1e-6 km/s resolution of fp32 phase velocities)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
EUS = np.load(os.path.join(HERE, "golden", "test1_eus.npz"))
KER = np.load(os.path.join(HERE, "golden", "test1_kernels.npz"))
PERIODS = list(range(10, 101, 10))


def layer_means(w, H):
    z = KER[f"phv_{w}_depth"]
    kb = KER[f"phv_{w}_kernels"][:, :, 0]
    top = np.concatenate([[0], np.cumsum(H)[:-1]])
    bot = np.cumsum(H)
    out = np.full((kb.shape[0], H.size), np.nan)
    for i in range(H.size):
        sel = (z >= top[i]) & (z < bot[i])
        if sel.sum() >= 2 and H[i] >= 4:
            out[:, i] = kb[:, sel].mean(axis=1)
    return out


def eus_columns():
    m = EUS["model"][0].astype(np.float64)
    return m[3].copy(), m[1], m[0], m[2], 1.0 / m[4]


def check(w, c):
    """c float [1+2L, P]: unperturbed, L x 0.999, L x 1.001 (senskernel.perturbed_batch order)."""
    H = eus_columns()[0]
    L = H.size
    fd = ((c[1 + L:].astype(np.float64) - c[1:1 + L]) / 0.002 / H[:, None] / c[0][None, :]).T
    ref = layer_means(w, H)
    assert np.isfinite(ref).sum() > 200
    for ip in range(len(PERIODS)):
        o = np.isfinite(ref[ip])
        assert np.abs(fd[ip][o] - ref[ip][o]).max() < 0.05 * np.abs(ref[ip][o]).max(), (w, PERIODS[ip])
    assert np.abs(c[0] - KER[f"phv_{w}_header"][:, 1]).max() < 2e-4       # c printed with 4 decimals


@pytest.mark.parametrize("w,kind", [("R", 2), ("L", 1)])
def test_oracle_finite_difference_kernels_vs_reference_fixture(w, kind):
    from oracle import cport
    from pysurfinv_amd import senskernel
    batch, kept = senskernel.perturbed_batch(*eus_columns())
    assert kept.size == batch.shape[2]
    c, u, st = cport.forward_batch(batch, np.asarray(PERIODS, np.float32), kind, nthreads=8)
    assert (st == 0).all()
    check(w, c)


@pytest.mark.gpu
@pytest.mark.parametrize("w,kind", [("R", 2), ("L", 1)])
def test_hip_finite_difference_kernels_vs_reference_fixture(w, kind):
    import torch
    from pysurfinv_amd import forward, senskernel
    H, Vs, Vp, Rho, Qs = eus_columns()
    batch, kept = senskernel.perturbed_batch(H, Vs, Vp, Rho, Qs)
    c, u, st = forward.forward_batch(batch, np.asarray(PERIODS, np.float32), kind=kind)
    assert (st == 0).all()
    check(w, c)
    # and the whole-batch device version: [1, P, L] kernels in the reference's (vH - vL)/0.2/H units
    per = torch.as_tensor(np.asarray(PERIODS, np.float32)).cuda()
    out = senskernel.sens_kernel_pert_batch(torch.from_numpy(batch[:1]).cuda(), per, wtype=w)
    k = out["phv"][0].cpu().numpy().astype(np.float64) * 100.0 / c[0][:, None]
    ref = layer_means(w, H)
    o = np.isfinite(ref)
    assert np.abs(k[o] - ref[o]).max() < 0.05 * np.abs(ref[o]).max()


def _fd_oracle(m, per, kind, row, eps=0.01):
    """Central differences of the CPU oracle, 1 % perturbations of one column (1 Vs, 0 Vp, 2 rho)."""
    from oracle import cport
    L = m.shape[2]
    big = np.repeat(m, 2 * L, axis=0)
    for i in range(L):
        big[i, row, i] *= (1 - eps); big[L + i, row, i] *= (1 + eps)
    co, _, so = cport.forward_batch(big, per, kind, nthreads=8)
    assert (so == 0).all()
    fd = ((co[L:].astype(np.float64) - co[:L]) / (2 * eps * np.where(m[0, row] != 0, m[0, row], 1.0)[:, None])).T
    fd[:, m[0, row] == 0] = 0
    return fd


def _kernel_cases():
    from pysurfinv_amd import synth
    cases = {"synth_L12": synth.synth_models(2, 12, seed=3)[:1], "eus_L68": EUS["model"].astype(np.float32)}
    wm = synth.synth_models(1, 9, seed=5)
    wm[0, 1, 0] = 0.0; wm[0, 0, 0] = 1.5; wm[0, 2, 0] = 1.03; wm[0, 3, 0] = 3.0
    cases["water_L9"] = wm
    return cases


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["synth_L12", "eus_L68", "water_L9"])
@pytest.mark.parametrize("w,kind", [("R", 2), ("L", 1)])
def test_analytic_kernels_match_oracle_finite_differences(name, w, kind):
    """surfdisp_forward_kernels_device: dc/dVs, dc/dVp, dc/drho of every layer from the energy integrals
    of ONE solve against central finite differences of the CPU oracle (2L solves per column).
    Measured 3e-4 .. 2.5e-3 of each period's largest entry = the noise of fp32 differences; bar 6e-3."""
    import torch
    from pysurfinv_amd import senskernel
    m = _kernel_cases()[name]
    per = np.asarray(PERIODS, np.float32)
    out = senskernel.analytic_kernels(torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda(), wtype=w)
    assert int(out["status"][0]) == 0
    for row, key in ((1, "dcdb"), (0, "dcda"), (2, "dcdr")):
        if out[key] is None:
            assert w == "L" and key == "dcda"
            continue
        an = out[key][0].cpu().numpy().astype(np.float64)
        fd = _fd_oracle(m, per, kind, row)
        scale = np.abs(fd).max(axis=1, keepdims=True)
        assert (np.abs(an - fd) / scale).max() < 6e-3, (name, w, key)
        assert np.abs(an.sum(1) - fd.sum(1)).max() < 8e-3 * np.abs(fd.sum(1)).max()


@pytest.mark.gpu
@pytest.mark.parametrize("w", ["R", "L"])
def test_analytic_kernels_vs_reference_fixture(w):
    """The same analytic (dc/c)/(db/b) per km as the reference's senskernel-1.0 known-answer files."""
    import torch
    from pysurfinv_amd import senskernel
    H = eus_columns()[0]
    m = EUS["model"].astype(np.float32)
    per = torch.as_tensor(np.asarray(PERIODS, np.float32)).cuda()
    out = senskernel.analytic_kernels(torch.from_numpy(m).cuda(), per, wtype=w)
    c0 = out["c0"][0].cpu().numpy().astype(np.float64)
    k = out["dcdb"][0].cpu().numpy().astype(np.float64) * m[0, 1][None, :] / H[None, :] / c0[:, None]
    ref = layer_means(w, H)
    for ip in range(len(PERIODS)):
        o = np.isfinite(ref[ip])
        assert np.abs(k[ip][o] - ref[ip][o]).max() < 0.05 * np.abs(ref[ip][o]).max(), (w, PERIODS[ip])


@pytest.mark.gpu
def test_analytic_kernels_batch_rows_and_failures():
    """Rows of a ragged batch equal the one-stack results; unsolved periods and bad stacks give zeros."""
    import torch
    from pysurfinv_amd import forward, synth
    m = synth.synth_models(6, 10, seed=2)
    m[3, 1, 4] = 0.4                                        # a strong low-velocity layer
    m[5, 0, 2] = -1.0                                       # bad stack
    nlay = np.array([10, 7, 10, 10, 5, 10], np.int32)
    per = synth.default_periods(20)
    plan = forward.BatchPlan(6, 10, 20)
    c, u, st, kb, ka, kr = plan.run_kernels(torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda(), kind=2,
                                            nlay=torch.from_numpy(nlay).cuda())
    c, st, kb = c.cpu().numpy(), st.cpu().numpy(), kb.cpu().numpy()
    assert st[5] == 4 and not kb[5].any()
    assert not kb[c == 0].any()                             # unsolved (stack, period): zero rows
    for i in (0, 1, 4):
        n = int(nlay[i])
        p1 = forward.BatchPlan(1, n, 20)
        c1, _, s1, kb1, _, _ = p1.run_kernels(torch.from_numpy(np.ascontiguousarray(m[i:i + 1, :, :n])).cuda(),
                                              torch.from_numpy(per).cuda(), kind=2)
        assert np.array_equal(kb[i, :, :n], kb1[0].cpu().numpy()) and not kb[i, :, n:].any()


@pytest.mark.gpu
def test_senskernelpert_class_fd_and_analytic_agree():
    """Drop-in SensKernelPert: DataFrame in, kernel['Vs'] / kernel['Vp'] out (reference units); the
    finite-difference route and the one-solve analytic route give the same kernels, also when Vp, Rho
    follow from Vs through the group rules."""
    import pandas as pd
    from pysurfinv_amd import senskernel
    H, Vs, Vp, Rho, Qs = eus_columns()
    H = H.copy(); H[-1] = 50.0
    df = pd.DataFrame(dict(H=H, Vs=Vs, Vp=Vp, Rho=Rho, Qs=Qs))
    a = senskernel.SensKernelPert(df, wtype="R", method="fd")
    b = senskernel.SensKernelPert(df, wtype="R", method="analytic")
    assert list(a.periods) == list(range(20, 101, 10)) and a.kernel["Vs"].shape == (9, H.size)
    thick = H >= 4.0
    for key in ("Vs", "Vp"):
        sc = np.abs(b.kernel[key][:, thick]).max()
        assert np.abs(a.kernel[key][:, thick] - b.kernel[key][:, thick]).max() < 0.03 * sc, key   # fd noise
    grp = ["sediment"] * 5 + ["crust"] * 8 + ["mantle"] * (H.size - 13)
    dfg = pd.DataFrame(dict(H=H, Vs=Vs, Grp=grp))
    a = senskernel.SensKernelPert(dfg, wtype="L", method="fd")
    b = senskernel.SensKernelPert(dfg, wtype="L", method="analytic")
    sc = np.abs(b.kernel["Vs"][:, thick]).max()
    assert np.abs(a.kernel["Vs"][:, thick] - b.kernel["Vs"][:, thick]).max() < 0.03 * sc


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 1])
def test_partials_through_the_scratch_equal_the_direct_route(kind):
    """surfdisp_forward_kernels_device accumulates the analytic partials in a layer-major scratch of its workspace
    (surfdisp_kernels_workspace_bytes: coalesced) and transposes them into the caller's [B][P][Lmax] rows; with a
    workspace of only surfdisp_workspace_bytes it writes the rows directly.  Same sums in the same order: bit-identical,
    ragged layer counts, a water layer and unsolved periods included."""
    import torch
    from pysurfinv_amd import forward, synth, _lib
    B, L, P = 300, 37, 11
    assert _lib.lib().surfdisp_kernels_workspace_bytes(B, L, P) >= _lib.lib().surfdisp_workspace_bytes(B, L, P) + 3 * L * P * B * 4
    m = synth.synth_models(B, L, seed=4, noise=0.08, monotone=False)
    m[:20, 1, 0] = 0.0; m[:20, 0, 0] = 1.475; m[:20, 2, 0] = 1.027; m[:20, 4, 0] = 1e-4; m[:20, 3, 0] = 2.0
    nlay = np.random.default_rng(2).integers(3, L + 1, B).astype(np.int32)
    per = torch.from_numpy(synth.default_periods(P)).cuda()
    mt, nt = torch.from_numpy(m).cuda(), torch.from_numpy(nlay).cuda()
    plan = forward.BatchPlan(B, L, P)
    big = [t.clone() if t is not None else None for t in plan.run_kernels(mt, per, kind=kind, nlay=nt)]
    small = plan.run_kernels(mt, per, kind=kind, nlay=nt, small_workspace=True)
    for a, b in zip(big, small):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert float(big[3].abs().max()) > 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("kind", [2, 1])
def test_partials_do_not_depend_on_the_workgroup_order(kind):
    """A launch of the group-velocity kernel that also writes the partials, over a whole number of 2 048-stack blocks, deals its
    workgroups to the XCDs so that the periods of a stack block share one L2 (launch_group); any other batch size runs in plain
    period-major order.  A unit's arithmetic does not depend on where it runs: 4 096 stacks in one call (XCD-aware order) equal
    the same stacks in two calls of 2 000 and 2 096 (plain order), bit for bit."""
    import torch
    from pysurfinv_amd import forward, synth
    B, L, P = 4096, 9, 7
    m = torch.from_numpy(synth.synth_models(B, L, seed=21, noise=0.05)).cuda()
    per = torch.from_numpy(synth.default_periods(P)).cuda()
    whole = [t.clone() if t is not None else None for t in forward.BatchPlan(B, L, P).run_kernels(m, per, kind=kind)]
    parts = []
    for lo, hi in ((0, 2000), (2000, B)):
        parts.append([t.clone() if t is not None else None
                      for t in forward.BatchPlan(hi - lo, L, P).run_kernels(m[lo:hi].contiguous(), per, kind=kind)])
    for q, a in enumerate(whole):
        if a is None:
            assert parts[0][q] is None
            continue
        assert torch.equal(a, torch.cat([parts[0][q], parts[1][q]]))
    assert float(whole[3].abs().max()) > 0.0


# ---- the analytic partials against the reference's OWN numbers (COMMON /rar1/, tests/golden/make_golden_partials.py)
PART = np.load(os.path.join(HERE, "golden", "ref_partials.npz"))


@pytest.mark.gpu
@pytest.mark.parametrize("name", [str(n) for n in PART["names"]])
@pytest.mark.parametrize("w,kind", [("R", 2), ("L", 1)])
def test_analytic_kernels_vs_reference_common_block(name, w, kind):
    """dc/d(b, a, rho) of every layer in the reference's coordinates (SURFDISP_KERN_REFCOORD: the flattened, attenuated
    layer values, no chain factors) against what REIGEN / LEIGEN themselves leave in COMMON /rar1/ (surfa.f:1133-1135,
    1182-1184, 1204-1207; Love 511-512, 564-565, 582-583), summed over each layer's sublayers.  The fixture holds
    one-period calls (the block is overwritten at every period) = SURFDISP_INDEPENDENT's start rule.  Bar: 1e-4 of the
    period's largest entry (measured <= 3e-5), both routes (scratch + transposition kernel, direct rows)."""
    import torch
    from oracle import cport
    from pysurfinv_amd import _lib, forward
    m = np.ascontiguousarray(PART[f"{name}_model"][None], np.float32)
    L = m.shape[2]
    per = PART["periods"].astype(np.float32)
    blk, meta = PART[f"{name}_{w}_rar1"], PART[f"{name}_{w}_meta"]
    plan = forward.BatchPlan(1, L, per.size)
    worst = 0.0
    for small in (False, True):
        c, u, st, kb, ka, kr = plan.run_kernels(torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda(),
                                                kind=kind | _lib.INDEPENDENT | _lib.KERN_REFCOORD, small_workspace=small)
        c = c.cpu().numpy()[0]
        got = {"dcdb": kb[0].cpu().numpy(), "dcdr": kr[0].cpu().numpy()}
        if kind == 2:
            got["dcda"] = ka[0].cpu().numpy()
        water = m[0, 1, 0] <= 0.0
        nchk = 0
        for ip in range(per.size):
            cref, uref, mm, ndiv = meta[ip]
            if cref <= 0:
                assert c[ip] == 0 and not got["dcdb"][ip].any()
                continue
            assert abs(c[ip] / cref - 1) < 2e-5, (name, w, per[ip], c[ip], cref)
            for i, key in enumerate(("dcda", "dcdb", "dcdr")):
                if key not in got:
                    continue
                ref = cport.sum_sublayers(blk[ip, i], L, int(ndiv), int(mm), water)
                # the period's peak: the largest entry the reference itself holds, i.e. per SUBLAYER (a layer's sum may be the
                # remainder of a cancellation between its sublayers: dc/drho of a 40 km layer, -0.111 + 0.056 + 0.038 + ... = -5e-4)
                peak = np.abs(blk[ip, i]).max()
                err = np.abs(got[key][ip].astype(np.float64) - ref).max() / peak
                worst = max(worst, err)
                assert err < 1e-4, (name, w, key, per[ip], err, small)
                nchk += 1
        assert nchk >= 8
    print(f"partials vs COMMON /rar1/ {name} {w}: worst {worst:.2e} of a period's peak")
