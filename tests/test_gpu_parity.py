"""Parity tests proper: the HIP path (through the C ABI) against
  (a) the committed golden vectors captured from the reference Fortran, and
  (b) the CPU oracle on seeded inputs.
Floating-point bar (BASELINE.json north_star): 1e-4 relative on c and U.  Held here: c <= 2e-5 everywhere; U <= 5e-5 on
every golden case except the rough (unsorted, low-velocity-zone) families, which are held to the 1e-4 bar itself -
EVERY entry, apart from the entries listed in tests/golden/u_exceptions.json: (stack, period) pairs next to
osculating modes at which the reference's own two builds (FMA contraction on / off) disagree with each other by
more than 2e-5 or return NaN (tests/golden/make_golden_spread.py).  Measured on MI355X (scripts/parity_table.py,
profiles/r02a/parity_table.txt): worst unlisted entry 7.5e-5, the listed ones <= 1.2e-4.
Zero patterns (= the reference's failure convention) coincide exactly on every case and every team size.
"""
import ctypes
import json
import os

import numpy as np
import pytest

from conftest import relerr, load_cases, GOLDEN

pytestmark = pytest.mark.gpu

TOL_C = 2e-5
TOL_U = 5e-5            # smooth / water / sparse cases
TOL_U_BAR = 1e-4        # north_star's bar: rough cases, every entry not in u_exceptions.json
TOL_U_LISTED = 5e-4     # listed entries (the reference's own builds differ there / one returns NaN)
CASES = sorted(load_cases().keys())
TEAMS = (1, 2, 4, 8, 16, 32, 64)
with open(os.path.join(GOLDEN, "u_exceptions.json")) as _f:
    U_EXCEPTIONS = {}
    for _e in json.load(_f)["entries"]:
        U_EXCEPTIONS.setdefault(_e["case"], []).append((_e["stack"], _e["period_index"]))


@pytest.fixture(scope="module")
def hip():
    from pysurfinv_amd import _lib, forward
    L = _lib.lib()
    assert L.surfdisp_device_count() >= 1, "no HIP device"
    yield forward
    L.surfdisp_set_team(0)


def _check_u(u, ref_u, case):
    """U parity, entry by entry (see the module docstring)."""
    u = np.asarray(u, np.float64); ref_u = np.asarray(ref_u, np.float64)
    ok = ref_u != 0
    if not ok.any():
        return
    e = np.zeros_like(ref_u)
    e[ok] = np.abs(u[ok] / ref_u[ok] - 1.0)
    e = np.where(np.isfinite(e), e, np.inf)
    listed = np.zeros_like(ok)
    for b, k in U_EXCEPTIONS.get(case, ()):
        listed[b, k] = True
    tol = TOL_U_BAR if case.startswith("rough") else TOL_U
    assert e[ok & ~listed].max() < tol, (case, e[ok & ~listed].max())
    if (ok & listed).any():
        assert e[ok & listed].max() < TOL_U_LISTED, (case, e[ok & listed].max())


def _same_zero_pattern(c, ref_c):
    """Unsolved periods are zeros in both: the reference's failure convention, stack by stack."""
    return np.array_equal(np.asarray(c) > 0, np.asarray(ref_c) > 0)


@pytest.mark.parametrize("case", CASES)
def test_golden_cases_default_team(hip, ref_cases, case):
    d = ref_cases[case]
    c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"])
    assert _same_zero_pattern(c, d["c"])
    assert relerr(c, d["c"]) < TOL_C
    _check_u(u, d["u"], case)
    assert np.array_equal(st == 0, np.all(d["c"] > 0, axis=1))


@pytest.mark.parametrize("team", TEAMS)
@pytest.mark.parametrize("case", ["synth_L10_R", "synth_L10_L", "synth_L64_R", "water_L9_R", "rough_L10_R", "rough_L64_R",
                                  "rough_thick_L22_R", "rough_thick_L12_L"])
def test_every_team_size_matches_golden(hip, ref_cases, case, team):
    from pysurfinv_amd import _lib
    d = ref_cases[case]
    assert _lib.lib().surfdisp_set_team(team) == 0
    try:
        c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"])
    finally:
        _lib.lib().surfdisp_set_team(0)
    assert _same_zero_pattern(c, d["c"])
    assert relerr(c, d["c"]) < TOL_C
    _check_u(u, d["u"], case)


@pytest.mark.parametrize("case", CASES)
def test_group_velocity_equals_the_oracles_at_the_same_phase_velocity(hip, ref_cases, case):
    """Separates the group-velocity arithmetic from the conditioning of U(c): the oracle's REIGEN / LEIGEN (and
    ellipticity) evaluated at the phase velocities the HIP path found (oracle test hook surfdisp_oracle_forward_at)
    against the HIP group velocities - 1e-5 everywhere (measured <= 6.5e-6), the listed ill-conditioned entries and
    their like 1e-4 (the ellipticity, taken at that c by two different arithmetics, feeds U with a large factor
    there: measured 5.5e-5 at one entry for teams of 4 and 8 lanes)."""
    from oracle import cport
    from pysurfinv_amd import _lib
    d = ref_cases[case]
    for team in (0, 4):
        _lib.lib().surfdisp_set_team(team)
        try:
            c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"])
        finally:
            _lib.lib().surfdisp_set_team(0)
        co, uo = cport.group_at(d["model"], d["periods"], d["kind"], c)
        ok = (uo != 0) & np.isfinite(uo) & (u != 0)
        if not ok.any():
            continue
        e = np.abs(u[ok].astype(np.float64) / uo[ok] - 1.0)
        assert np.quantile(e, 0.995) < 1e-5 and e.max() < 1e-4, (case, team, e.max())


@pytest.mark.parametrize("wave", ["R", "L"])
def test_reference_known_answers_TEST1(hip, eus, wave):
    """senskernel-1.0/TEST1 (fp64 twin); the fp32 reference itself is 2.5e-5..8.7e-5 away."""
    kind = 2 if wave == "R" else 1
    c, u, st = hip.forward_batch(eus["model"], eus["periods"], kind)
    assert st[0] == 0
    assert relerr(c[0], eus[f"c_{wave}_fp64twin"]) < 1e-4
    assert relerr(u[0], eus[f"u_{wave}_fp64twin"]) < 1.2e-4
    assert relerr(c[0], eus[f"c_{wave}_ref"]) < TOL_C
    assert relerr(u[0], eus[f"u_{wave}_ref"]) < TOL_U


@pytest.mark.parametrize("kind", [2, 1])
@pytest.mark.parametrize("L", [5, 10, 33])
def test_against_oracle_seeded(hip, kind, L):
    from oracle import cport
    from pysurfinv_amd import synth
    model = synth.synth_models(512, L, seed=100 + L)
    per = synth.default_periods(20)
    c, u, st = hip.forward_batch(model, per, kind)
    co, uo, so = cport.forward_batch(model, per, kind, nthreads=8)
    assert np.array_equal(c > 0, co > 0)
    assert relerr(c, co) < TOL_C and relerr(u, uo) < TOL_U


def test_ragged_layer_counts(hip):
    """nlay[b] < Lmax: the tail of each row is ignored (f2py passes exactly nlay elements)."""
    from oracle import cport
    from pysurfinv_amd import synth
    per = synth.default_periods(20)
    rng = np.random.default_rng(0)
    B, Lmax = 96, 16
    nlay = rng.integers(2, Lmax + 1, B).astype(np.int32)
    model = np.full((B, 5, Lmax), np.nan, np.float32)       # poison the unused tail
    for i, n in enumerate(nlay):
        model[i, :, :n] = synth.synth_models(1, int(n), seed=1000 + i)[0]
    c, u, st = hip.forward_batch(model, per, 2, nlay=nlay)
    co, uo, so = cport.forward_batch(np.nan_to_num(model), per, 2, nlay=nlay, nthreads=8)
    assert np.array_equal(c > 0, co > 0)
    assert relerr(c, co) < TOL_C and relerr(u, uo) < TOL_U


def test_bad_models_are_flagged_not_crashing(hip):
    from pysurfinv_amd import synth, _lib
    per = synth.default_periods(20)
    model = synth.synth_models(8, 6, seed=3)
    model[1, 0, 2] = np.nan
    model[2, 3, 0] = -1.0
    nlay = np.full(8, 6, np.int32); nlay[3] = 1; nlay[4] = 7
    c, u, st = hip.forward_batch(model, per, 2, nlay=nlay)
    for i in (1, 2, 3, 4):
        assert st[i] == _lib.BADMODEL and not c[i].any() and not u[i].any()
    for i in (0, 5, 6, 7):
        assert st[i] == 0 and (c[i] > 0).all()


def test_f2py_shaped_fast_surf_and_calForward(hip, ref_cases):
    from pysurfinv_amd import fast_surf, forward
    d = ref_cases["c1_single_L5_R"]                           # BASELINE config 1
    m = d["model"][0].astype(np.float64)
    per = np.zeros(200); per[:20] = d["periods"]
    ur, ul, cr, cl = fast_surf.fast_surf(5, 2, m[0], m[1], m[2], m[3], m[4], per, 20)
    assert cr.shape == (200,) and cr.dtype == np.float32 and not cr[20:].any()
    assert not ul.any() and not cl.any()
    assert relerr(cr[:20], d["c"][0]) < TOL_C and relerr(ur[:20], d["u"][0]) < TOL_U
    with pytest.raises(ValueError):
        fast_surf.fast_surf(4, 2, m[0], m[1], m[2], m[3], m[4], per, 20)
    # _calForward: profile rows (h, Vs, Vp, rho, qs, qp), thin layers filtered (models.py:20)
    prof = np.stack([m[3], m[1], m[0], m[2], 1.0 / m[4], 2.0 / m[4]])
    prof[0, -1] = 50.0                                      # half-space thickness: ignored
    thin = np.concatenate([prof[:, :2], np.array([[1e-4, 3.1, 5.5, 2.5, 600, 1200]]).T, prof[:, 2:]], axis=1)
    out = forward._calForward(thin, "Ray", list(d["periods"]))
    assert out is not None and relerr(out, d["c"][0]) < TOL_C
    with pytest.raises(ValueError):
        forward._calForward(prof, "Stoneley", [10.0])
    # failing stack -> None (models.py:29-33)
    lv = ref_cases["lvz_halfspace_L4_R"]["model"][0].astype(np.float64)
    profl = np.stack([lv[3], lv[1], lv[0], lv[2], 1.0 / lv[4], 2.0 / lv[4]]); profl[0, -1] = 10.0
    assert forward._calForward(profl, "Ray", list(np.linspace(6, 80, 18))) is None


def test_device_pointer_entry_torch(hip):
    import torch
    from oracle import cport
    from pysurfinv_amd import synth, forward
    model = synth.synth_models(1024, 10, seed=21)
    per = synth.default_periods(20)
    dm = torch.from_numpy(model).cuda(); dp = torch.from_numpy(per).cuda()
    plan = forward.BatchPlan(1024, 10, 20)
    for kind in (2, 1):
        c, u, st = plan.run(dm, dp, kind=kind)
        torch.cuda.synchronize()
        co, uo, so = cport.forward_batch(model, per, kind, nthreads=8)
        assert relerr(c.cpu().numpy(), co) < TOL_C and relerr(u.cpu().numpy(), uo) < TOL_U
    with pytest.raises(ValueError):
        plan.run(dm[:10], dp)


def test_full_size_properties_config2(hip):
    """BASELINE config 2 at full size (B=65 536, L=10, P=20, Rayleigh c+U) through
    size-independent properties: every stack solved, 0 < U < c < max Vs, batch-order
    independence (a permuted batch gives the permuted answer bit for bit), determinism,
    and a 1 024-stack sample against the oracle."""
    import torch
    from oracle import cport
    from pysurfinv_amd import synth, forward
    B = 65536
    model = synth.synth_models(B, 10, seed=0)
    per = synth.default_periods(20)
    dm = torch.from_numpy(model).cuda(); dp = torch.from_numpy(per).cuda()
    plan = forward.BatchPlan(B, 10, 20)
    c, u, st = plan.run(dm, dp, kind=2); torch.cuda.synchronize()
    c = c.cpu().numpy(); u = u.cpu().numpy(); st = st.cpu().numpy()
    assert (st == 0).all() and (c > 0).all()
    assert (u > 0).all() and (u < c * 1.0001).all() and (c < model[:, 1].max(axis=1, keepdims=True) * 1.05).all()
    perm = np.random.default_rng(1).permutation(B)
    c2, u2, _ = plan.run(torch.from_numpy(model[perm]).cuda(), dp, kind=2); torch.cuda.synchronize()
    assert np.array_equal(c2.cpu().numpy(), c[perm]) and np.array_equal(u2.cpu().numpy(), u[perm])
    idx = np.random.default_rng(2).choice(B, 1024, replace=False)
    co, uo, so = cport.forward_batch(model[idx], per, 2, nthreads=8)
    assert relerr(c[idx], co) < TOL_C and relerr(u[idx], uo) < TOL_U


@pytest.mark.parametrize("B,L,kind", [(70000, 10, 2), (70000, 10, 1), (40000, 20, 2), (20000, 64, 2), (150001, 6, 2)])
def test_large_host_buffer_calls_are_chunked_and_pipelined(hip, B, L, kind):
    """surfdisp_forward_batch on large host buffers: the batch goes through the device in chunks alternating between two
    streams (copies beside kernels, chunks in flight beside each other; uneven last chunk, ragged layer counts) - same
    answers as the one-launch device entry; a phase-only call may pass u = NULL."""
    import torch
    from pysurfinv_amd import _lib, synth, forward
    per = synth.default_periods(17)
    model = synth.synth_models(B, L, seed=9, **({} if L == 10 else {"total_thickness": 220.0}))
    nlay = np.full(B, L, np.int32)
    nlay[::11] = max(2, L - 3)
    # (stacks of more than 20 layers go through in one piece, on the calling thread's cached buffer and stream)
    c, u, st = forward.forward_batch(model, per, kind=kind, nlay=nlay)
    plan = forward.BatchPlan(B, L, len(per))
    cd, ud, sd = plan.run(torch.from_numpy(model).cuda(), torch.from_numpy(per).cuda(), kind=kind, nlay=torch.from_numpy(nlay).cuda())
    torch.cuda.synchronize()
    # same zero pattern and status; values to the spread between team sizes (a chunk's launch may use other teams than
    # the whole batch's: the refinement subdivides its bracket G ways; 99.9 % of the roots agree to 2e-6, all to 3e-5)
    cd, ud = cd.cpu().numpy(), ud.cpu().numpy()
    assert np.array_equal(st, sd.cpu().numpy()) and np.array_equal(c > 0, cd > 0)
    q = lambda a, b: float(np.quantile(np.abs(a[b > 0].astype(np.float64) / b[b > 0] - 1), 0.999))
    assert relerr(c, cd) < 3e-5 and q(c, cd) < 2e-6, (relerr(c, cd), q(c, cd))
    assert relerr(u, ud) < 2e-3 and q(u, ud) < 2e-5, (relerr(u, ud), q(u, ud))
    assert (c > 0).mean() > 0.99
    # the calling thread's cached buffers and streams can be given back; the next call builds them again
    lib0 = _lib.lib()
    lib0.surfdisp_thread_release()
    c3, u3, st3 = forward.forward_batch(model[:4096], per, kind=kind, nlay=nlay[:4096])
    assert np.array_equal(st3, st[:4096]) and relerr(c3, cd[:4096]) < 3e-5
    lib0.surfdisp_thread_release()
    # phase only, no group-velocity and no status array
    lib = _lib.lib()
    c2 = np.full((B, len(per)), -1.0, np.float32)
    fp = lambda x: x.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    _lib.check(lib.surfdisp_forward_batch(0, B, L, nlay.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), fp(model), len(per), fp(per),
                                          kind | _lib.PHASE_ONLY, fp(c2), None, None))
    assert np.array_equal(c2, c)


@pytest.mark.parametrize("B,L,team,rough", [(20000, 10, 0, False), (20000, 10, 4, True), (4096, 64, 8, False), (4096, 30, 8, True),
                                             (3000, 21, 2, True), (30000, 30, 0, True)])
def test_love_certified_scan_is_the_point_by_point_scan(hip, B, L, team, rough):
    """Love root searches skip grid points between two coarse points whose Sturm counts agree (a theorem,
    not a heuristic).  Every result - c, U, status, zeros - equals the point-by-point scan's (SURFDISP_EXACTSCAN) BIT FOR
    BIT: same brackets, hence the same refinement; smooth and rough (unsorted, sigma 0.15) stacks, water on top of some,
    ragged layer counts, every team size the certified scan is instantiated for and the library's own choice."""
    import torch
    from pysurfinv_amd import _lib, synth, forward
    per = synth.default_periods(23)
    kw = {} if L == 10 else {"total_thickness": 220.0}
    model = synth.synth_models(B, L, seed=31, **(dict(kw, noise=0.15, monotone=False) if rough else kw))
    if L >= 6:
        model[::9, 1, 0] = 0.0; model[::9, 0, 0] = 1.475; model[::9, 2, 0] = 1.027; model[::9, 4, 0] = 1e-4      # a water layer on top
    nlay = np.full(B, L, np.int32); nlay[::7] = max(2, L - 2)
    dm, dp, dn = torch.from_numpy(model).cuda(), torch.from_numpy(per).cuda(), torch.from_numpy(nlay).cuda()
    _lib.lib().surfdisp_set_team(team)
    try:
        plan = forward.BatchPlan(B, L, len(per))
        c1, u1, s1 = (t.clone() for t in plan.run(dm, dp, kind=1, nlay=dn))
        c0, u0, s0 = (t.clone() for t in plan.run(dm, dp, kind=1 | _lib.EXACTSCAN, nlay=dn))
        torch.cuda.synchronize()
    finally:
        _lib.lib().surfdisp_set_team(0)
    assert torch.equal(c1, c0) and torch.equal(s1, s0) and torch.equal(u1.nan_to_num(-7.0), u0.nan_to_num(-7.0))
    assert float((c1 > 0).float().mean()) > 0.9


def test_extreme_velocities_follow_the_reference(hip):
    """Stacks far outside seismology.  Vs x 3 (roots up to ~14 km/s) must agree with the oracle.  Vs x 6 puts the
    roots above 16 km/s, where one fp32 ulp (1.9e-6) exceeds NEVILL's 1e-6 bracket tolerance: the REFERENCE never
    converges there (50 cycles, LSTOP, calcul.f:172-189 -> 9999: nothing is returned; oracle status 3).  The HIP
    path reports exactly those stacks as SURFDISP_NUMERIC with all-zero outputs."""
    from oracle import cport
    from pysurfinv_amd import synth, _lib
    per = synth.default_periods(12)
    model = synth.synth_models(64, 8, seed=9)
    model[:, 0:2] *= 3.0
    c, u, st = hip.forward_batch(model, per, 2)
    co, uo, so = cport.forward_batch(model, per, 2, nthreads=8)
    assert np.array_equal(c > 0, co > 0)
    assert relerr(c, co) < 2e-5 and relerr(u, uo) < 1e-4
    model[:, 0:2] *= 2.0
    for team in (0, 1, 4, 64):
        _lib.lib().surfdisp_set_team(team)
        try:
            c, u, st = hip.forward_batch(model, per, 2)
        finally:
            _lib.lib().surfdisp_set_team(0)
        co, uo, so = cport.forward_batch(model, per, 2, nthreads=8)
        assert (so == cport.NEVILL).all() and not co.any()
        assert (st == _lib.NUMERIC).all() and not c.any() and not u.any()


def test_sensitivity_kernels_match_oracle_finite_differences(hip, eus):
    """SensKernelPert as one batched solve (2L+1 stacks) vs the same finite
    differences taken with the CPU oracle."""
    from oracle import cport
    from pysurfinv_amd import senskernel
    m = eus["model"][0].astype(np.float64)
    H, Vs, Vp, Rho, Qs = m[3].copy(), m[1], m[0], m[2], 1.0 / m[4]
    H[-1] = 50.0
    periods = list(range(20, 101, 10))
    k = senskernel.sens_kernel_pert(H, Vs, Vp, Rho, Qs, periods=periods, wtype="R")
    batch, kept = senskernel.perturbed_batch(H, Vs, Vp, Rho, Qs)
    co, uo, so = cport.forward_batch(batch, np.asarray(periods, np.float32), 2, nthreads=8)
    L = kept.size
    kref = ((co[1 + L:].astype(np.float64) - co[1:1 + L]) / 0.2 / H[kept][:, None]).T
    assert k["phv"].shape == (len(periods), H.size)
    # a kernel is a difference of two fp32 c's 0.2 % apart divided by 0.2*H (H down to 0.25 km here):
    # it inherits the c parity (~1e-6 km/s) amplified by 1/(0.2 H), in the reference just as here.
    # Compare the underlying differences, and the kernel itself where layers are thick enough.
    dd = (k["phv"] - kref) * 0.2 * H[None, :]
    assert np.abs(dd).max() < 1.2e-5          # = 6 units of c parity (TOL_C x 4 km/s = 8e-5 is the bar for ONE c)
    thick = H >= 10.0
    scale = np.abs(kref[:, thick]).max()
    assert np.abs(k["phv"][:, thick] - kref[:, thick]).max() < 2e-2 * scale


def test_batched_device_sensitivity_kernels(hip):
    """sens_kernel_pert_batch (many stacks, built and differenced on the device) = the one-stack host
    version, stack by stack, for both wave types."""
    import torch
    from pysurfinv_amd import senskernel, synth
    M, L = 5, 12
    model = synth.synth_models(M, L, seed=9)
    periods = list(range(20, 101, 10))
    per = torch.as_tensor(np.asarray(periods, np.float32)).cuda()
    for wtype in ("R", "L"):
        out = senskernel.sens_kernel_pert_batch(torch.from_numpy(model).cuda(), per, wtype=wtype, chunk=2)
        for i in range(M):
            m = model[i].astype(np.float64)
            k = senskernel.sens_kernel_pert(m[3], m[1], m[0], m[2], 1.0 / m[4], periods=periods, wtype=wtype)
            for name in ("phv", "grv"):
                a, b = out[name][i].cpu().numpy(), k[name]
                assert a.shape == b.shape == (len(periods), L)
                # same HIP solves; the only difference is fp32 vs fp64 differencing of the outputs
                assert np.nanmax(np.abs(a - b)) <= 1e-5 * max(1.0, np.nanmax(np.abs(b))), (wtype, name, i)
            assert np.array_equal(out["c0"][i].cpu().numpy(), k["c0"])
    assert int(out["status"].abs().sum()) == 0


def test_device_entry_is_graph_capturable(hip):
    """surfdisp_forward_batch_device allocates nothing and never synchronises: the three kernels +
    finish can be captured in a HIP graph and replayed (launch-bound inner loops of the Metropolis
    driver)."""
    import torch
    from pysurfinv_amd import synth, forward
    model = torch.from_numpy(synth.synth_models(2048, 10, seed=5)).cuda()
    per = torch.from_numpy(synth.default_periods(20)).cuda()
    plan = forward.BatchPlan(2048, 10, 20)
    c0, u0, _ = plan.run(model, per, kind=2); torch.cuda.synchronize()
    c0, u0 = c0.clone(), u0.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.run(model, per, kind=2)                       # warm-up on the side stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        plan.run(model, per, kind=2)
    model2 = torch.from_numpy(synth.synth_models(2048, 10, seed=6)).cuda()
    model.copy_(model2)                                    # new inputs in the captured buffers
    plan.c.zero_(); plan.u.zero_()
    g.replay(); torch.cuda.synchronize()
    from oracle import cport
    co, uo, _ = cport.forward_batch(model2.cpu().numpy(), per.cpu().numpy(), 2, nthreads=8)
    assert relerr(plan.c.cpu().numpy(), co) < TOL_C and relerr(plan.u.cpu().numpy(), uo) < TOL_U
    assert not np.array_equal(plan.c.cpu().numpy(), c0.cpu().numpy())


def test_joint_rayleigh_love_two_streams(hip, ref_cases):
    """BASELINE configs[4] shape: R+L, c+U, 64-layer stacks, both wave types in flight together."""
    import torch
    from pysurfinv_amd import forward
    dR, dL = ref_cases["synth_L64_R"], ref_cases["synth_L64_L"]
    assert np.array_equal(dR["model"], dL["model"])
    model = torch.from_numpy(dR["model"]).cuda(); per = torch.from_numpy(dR["periods"]).cuda()
    plan = forward.JointPlan(model.shape[0], 64, per.numel())
    out = plan.run(model, per); torch.cuda.synchronize()
    for w, d in (("R", dR), ("L", dL)):
        assert relerr(out[f"c{w}"].cpu().numpy(), d["c"]) < TOL_C
        assert relerr(out[f"u{w}"].cpu().numpy(), d["u"]) < TOL_U


@pytest.mark.parametrize("B,L,P", [(1, 2, 1), (3, 5, 200), (2, 200, 7), (70000, 4, 3)])
def test_shape_extremes_against_oracle(hip, B, L, P):
    """Smallest and largest shapes the ABI accepts (P <= 200 as fast_surf.pyf:14-19, L <= 200)."""
    from oracle import cport
    from pysurfinv_amd import synth
    model = synth.synth_models(B, L, seed=L + P)
    per = np.linspace(6.0, 150.0, P).astype(np.float32) if P > 1 else np.array([20.0], np.float32)
    c, u, st = hip.forward_batch(model, per, 2)
    n = min(B, 256)
    co, uo, so = cport.forward_batch(model[:n], per, 2, nthreads=8)
    assert np.array_equal(c[:n] > 0, co > 0)
    assert relerr(c[:n], co) < TOL_C and relerr(u[:n], uo) < TOL_U


def test_unsorted_and_degenerate_periods_terminate(hip):
    """Descending / repeated / non-positive periods are outside the contract (ascending list), but
    must come back (zeros or values), never hang."""
    from pysurfinv_amd import synth
    model = synth.synth_models(64, 10, seed=1)
    for per in ([80.0, 40.0, 20.0, 10.0], [20.0, 20.0, 20.0], [0.0, 10.0], [-5.0, 10.0], [np.nan, 10.0]):
        c, u, st = hip.forward_batch(model, np.asarray(per, np.float32), 2)
        assert c.shape == (64, len(per))
        c, u, st = hip.forward_batch(model, np.asarray(per, np.float32), 1)
        assert c.shape == (64, len(per))


@pytest.mark.parametrize("case", ["synth_L5_R", "synth_L10_R", "synth_L10_L", "synth_L21_R", "synth_L64_R",
                                  "synth_L64_L", "c1_single_L5_R", "two_layer_R"])
def test_independent_mode_matches_reference_on_monotone_stacks(hip, ref_cases, case):
    """SURFDISP_INDEPENDENT (one team per (stack, period), BASELINE north_star's work unit): on
    monotone stacks with the dense 8-100 s period list it agrees with the reference to ~1e-6."""
    d = ref_cases[case]
    c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"], independent=True)
    assert np.array_equal(c > 0, d["c"] > 0) and (st == 0).all()
    assert relerr(c, d["c"]) < 1e-5 and relerr(u, d["u"]) < 2e-5


@pytest.mark.parametrize("case,bound", [("dense18_L10_R", 5e-4), ("sparse_L10_R", 2e-2), ("water_L9_R", 2e-3)])
def test_independent_mode_known_deviation_is_the_references_period_list_dependence(hip, ref_cases, case, bound):
    """Where the reference's answer depends on the period LIST (mmax carry-over: only the layers
    inside the previous period's effective half space are refreshed, calcul.f:112,133 - SURVEY.md
    section 4 defect 2 measured 6e-5..2.4e-3) the independent mode solves each period on a fully
    rebuilt stack and therefore differs by that amount; the faithful mode (default) does not.
    This pins the size of the documented difference."""
    d = ref_cases[case]
    ci, ui, _ = hip.forward_batch(d["model"], d["periods"], d["kind"], independent=True)
    cf, uf, _ = hip.forward_batch(d["model"], d["periods"], d["kind"])
    assert relerr(cf, d["c"]) < TOL_C                       # faithful mode: parity
    assert 1e-5 < relerr(ci, d["c"]) < bound                # independent mode: the carry-over effect


def test_independent_mode_failure_cascade_and_water(hip, ref_cases):
    """A failing period zeroes itself and all later ones (calcul.f:203-219), reduced over the
    stack's teams; water stacks start the scan at 0.5 km/s in every period."""
    d = ref_cases["lvz_halfspace_L4_R"]
    c, u, st = hip.forward_batch(d["model"], d["periods"], 2, independent=True)
    bad = np.argmax(c[0] == 0) if (c[0] == 0).any() else len(c[0])
    assert not c[0, bad:].any() and not u[0, bad:].any()
    assert st[0] == (0 if bad == len(c[0]) else (2 if bad == 0 else 1))
    d = ref_cases["water_L9_L"]
    c, u, st = hip.forward_batch(d["model"], d["periods"], 1, independent=True)
    assert np.isfinite(c).all() and c.shape == d["c"].shape


def test_default_scan_on_the_zero_group_velocity_stack(hip):
    """tests/golden/ref_zgv_stack.npz: a soft-sediment stack (Vs 0.28 ... 1.3 km/s, Vp/Vs 2.9, over 4.6 km/s) whose secular function
    dips below zero between c = 0.946 and 0.981 at T = 23.78 s - two roots 0.035 km/s apart on a branch with a zero-group-velocity
    point.  The DEFAULT scan walks every grid point and returns the reference's root for every team size; the opt-in count-guided
    scan (SURFDISP_FASTSCAN) steps over the pair - the count is the same on both sides - and is allowed to: the fixture records what
    it returned, and the test only checks that it returns either of the two."""
    from pysurfinv_amd import _lib
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_zgv_stack.npz"))
    m = d["model"][None]; per = d["periods"]
    for team in (0, 1, 2, 4, 8, 16):
        assert _lib.lib().surfdisp_set_team(team) == 0
        try:
            c, u, st = hip.forward_batch(m, per, 2)
            cf, uf, sf = hip.forward_batch(m, per, 2, fast_scan=True)
        finally:
            _lib.lib().surfdisp_set_team(0)
        assert np.array_equal(c[0] > 0, d["c"] > 0)
        ok = d["c"] > 0
        assert np.abs(c[0][ok] / d["c"][ok] - 1).max() < 2e-5
        near = lambda a, b: np.abs(a[ok] / np.where(b[ok] > 0, b[ok], 1) - 1) < 2e-5
        assert (near(cf[0], d["c"]) | near(cf[0], d["c_count_guided"])).all()


def test_love_overflow_stacks_go_through_the_exact_kernel(hip):
    """DLTAR1 carries (ut, tt) through the layers without normalisation (surfa.f:143-179): three 300 km layers at 2-4 s put
    e^(several hundred) into the pair, the reference scans NaNs and returns the edge of the overflowed region.  The production
    Love recursion stays finite there, so the root search bounds the growth per period (sum of k d sqrt(1 - c^2/b^2) over the
    evanescent layers at the lowest trial velocity) and hands such stacks to the exact fallback kernel: every one of them is
    re-solved there, and the default call returns what SURFDISP_STRICT returns, bit for bit, for every team size."""
    import torch
    from pysurfinv_amd import forward, synth, _lib
    B, L, P = 192, 4, 6
    m = synth.synth_models(B, L, seed=11, noise=0.05, monotone=True, total_thickness=900.0)
    per = np.linspace(2.0, 4.0, P).astype(np.float32)
    mt, pt = torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda()
    plan = forward.BatchPlan(B, L, P)
    cs, us, ss = [t.clone() for t in plan.run(mt, pt, kind=1 | _lib.STRICT)]
    for team in (0, 1, 2, 8, 64):
        assert _lib.lib().surfdisp_set_team(team) == 0
        try:
            c, u, st = plan.run(mt, pt, kind=1)
            torch.cuda.synchronize()
            assert plan.counters()[0] == B                          # every stack was handed over
        finally:
            _lib.lib().surfdisp_set_team(0)
        assert torch.equal(st, ss) and torch.equal(c, cs)
        assert torch.equal(torch.nan_to_num(u, nan=-1.0), torch.nan_to_num(us, nan=-1.0))


def test_fp32_overflow_regime_follows_the_reference(hip):
    """Two 200 km layers at T = 5 s: the un-normalised secular function overflows fp32 (NaN / inf) below ~3.0 km/s.
    The reference's scan treats a NaN as positive, brackets the edge of the overflowed region, and NEVILL - whose
    arithmetic IFs send a NaN abscissa to a bisection step (surfa.f:32-34) - converges onto that edge: the first
    period gets a spurious root (and a NaN group velocity), the later periods are solved normally.  The HIP path
    does the same, for every team size; neighbours in the same batch are unaffected."""
    from oracle import cport
    from pysurfinv_amd import synth, _lib
    m = synth.synth_models(64, 2, seed=2, noise=0.05, monotone=False, total_thickness=400.0)
    ok = synth.synth_models(64, 2, seed=3, noise=0.1, monotone=True, total_thickness=60.0)
    model = np.concatenate([m, ok]).astype(np.float32)
    per = np.linspace(5, 100, 24).astype(np.float32)
    co, uo, so = cport.forward_batch(model, per, 2, nthreads=8)
    assert (so == 0).all() and (co > 0).all() and np.isnan(uo[:64, 0]).all()     # what the reference returns
    for team in (0, 1, 4, 64):
        assert _lib.lib().surfdisp_set_team(team) == 0
        try:
            c, u, st = hip.forward_batch(model, per, 2)
        finally:
            _lib.lib().surfdisp_set_team(0)
        assert (st == 0).all() and (c > 0).all()
        assert relerr(c, co) < 2e-5                                   # also the spurious first-period root
        assert relerr(c[:, 1:], co[:, 1:]) < 2e-5 and relerr(u[:, 1:], uo[:, 1:]) < 1e-4
        assert relerr(u[64:], uo[64:]) < 1e-4


FAMILY_BARS = {
    # family: (stacks that may differ in zero pattern or root choice, c tolerance)
    "sediment_R": (0, 1e-4), "sediment_L": (0, 2e-5),
    "ragged_R": (0, 2e-5), "ragged_L": (0, 2e-5),
    "overflow_R": (0, 2e-5),
    # (one 2-layer stack, 200 km over a half space at 3 and 6.4 s, has three Love modes inside one 0.01 km/s bracket:
    # which of them NEVILL lands on depends on its evaluation sequence - reproduced by the exact fallback kernel)
    "overflow_L": (0, 2e-5),
}
with open(os.path.join(GOLDEN, "u_exceptions_families.json")) as _f:
    FAMILY_EXCEPTIONS = {}
    for _e in json.load(_f)["entries"]:
        FAMILY_EXCEPTIONS.setdefault(_e["family"], []).append((_e["stack"], _e["period_index"], _e["why"]))


@pytest.mark.parametrize("family", sorted(FAMILY_BARS))
def test_soak_family_fixtures(hip, ref_families, family):
    """One small fixture per family of the builder-run differential soaks (tests/golden/make_golden_families.py:
    what the reference Fortran returns for soft sediments at short periods, rough ragged stacks, the fp32-overflow
    regime).  Zero patterns and root choice as the reference's, c within tolerance, and the group velocity within 1e-4
    of the reference's own at EVERY entry (no quantiles) - except the entries of tests/golden/u_exceptions_families.json
    (tests/golden/make_golden_spread_families.py records for each: the reference's own FMA / non-FMA spread, |dlnU/dlnc|,
    the HIP path's error in default and SURFDISP_STRICT mode).  Since r04 the list holds FIVE entries, every one of them a
    (stack, period) at which the reference's own two builds disagree with each other by 4e-5 .. 2.8e-3 (|dlnU/dlnc| 160 ..
    1100; SURFDISP_STRICT misses there as the default does); they are held to 5e-3.  The r03 list's other four entries are
    inside the 1e-4 bar now: three soft-sediment guided-wave entries (c ~ 0.25 km/s) whose ellipticity is evaluated with the
    reference's own arithmetic (surfdisp_ellip_kernel: c far below the stack's fastest S velocity), and one entry whose root
    lies 5e-6 km/s from a layer's S velocity (the refinement no longer interpolates across such a kink)."""
    from pysurfinv_amd import _lib
    d = ref_families[family]
    nbad_max, tol_c = FAMILY_BARS[family]
    B, P = d["c"].shape
    listed = np.zeros((B, P)); own = np.zeros((B, P), bool)
    for b, k, why in FAMILY_EXCEPTIONS.get(family, ()):
        assert why.startswith("reference builds disagree"), why     # (the only kind of exception left)
        listed[b, k] = 5e-3
    for team in (0, 1, 4, 16):
        _lib.lib().surfdisp_set_team(team)
        try:
            c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"], nlay=d["nlay"])
        finally:
            _lib.lib().surfdisp_set_team(0)
        same = (c > 0) == (d["c"] > 0)
        with np.errstate(all="ignore"):
            ec = np.where(d["c"] > 0, np.abs(c.astype(np.float64) / d["c"] - 1), 0.0)
        good = same.all(axis=1) & (ec.max(axis=1) < tol_c)
        assert (~good).sum() <= nbad_max, (family, team, int((~good).sum()))
        ok = np.isfinite(d["u"]) & (np.abs(d["u"]) > 1e-3) & (c > 0) & (d["c"] > 0) & good[:, None]
        with np.errstate(all="ignore"):
            e = np.abs(u.astype(np.float64) / d["u"] - 1)
        e = np.where(np.isfinite(e), e, np.inf)                  # a NaN of ours where the reference is finite counts
        e = np.where(ok, e, 0.0)
        bar = np.where(listed > 0, listed, 1e-4)
        worst = np.unravel_index(np.argmax(e / bar), e.shape)
        assert (e < bar).all(), (family, team, worst, float(e[worst]))


@pytest.mark.parametrize("wave", ["R", "L"])
def test_wild_family_terminates_and_mostly_agrees(hip, ref_families, wave):
    """Anything monotone: layers of 0.1-5 km/s, 10 m - 50 km thick, periods 0.1-300 s.  Kilometres of 0.1 km/s material
    at 0.15 s are hundreds of wavelengths: the fp32 secular function overflows or is rounding noise over most of the
    scan, and the reference brackets rounding-decided sign changes (its own U is NaN at a fifth of these entries).
    Nearly all of these stacks go through the exact fallback kernel (the reference's arithmetic restated statement by
    statement), which reproduces most of that: measured (profiles/r02a/parity_table.txt) Rayleigh - every zero
    pattern, 96 % of the phase velocities within 1e-4; Love - 94-97 % of the zero patterns, 75-82 % of the values
    (its modes crowd far below the 0.01 km/s walk in the slow layers, and an early period that lands on another
    overtone hands a different start value to every later one).  Also: termination, finite-or-zero outputs, the
    failure cascade (zeros only at the tail)."""
    d = ref_families[f"wild_{wave}"]
    c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"], nlay=d["nlay"])
    assert np.isfinite(c).all() and (c >= 0).all()
    nz = c > 0
    assert (nz[:, :-1] | ~nz[:, 1:]).all()                       # once a period failed, all later ones are zero
    both = nz & (d["c"] > 0)
    e = np.abs(c[both].astype(np.float64) / d["c"][both] - 1)
    same = ((c > 0) == (d["c"] > 0)).all(axis=1).mean()
    assert np.median(e) < 1e-5
    assert (e < 1e-4).mean() > (0.9 if wave == "R" else 0.7) and same >= (0.97 if wave == "R" else 0.9)


@pytest.mark.parametrize("team", [2, 4, 8, 64])
def test_fast_scan_same_brackets_as_default_scan_on_golden_cases(hip, ref_cases, team):
    """Opt-in SURFDISP_FASTSCAN (coarse steps over intervals its tests find free of roots, point-by-point rescans
    elsewhere) against the default scan (every grid point, as the reference): bit-identical outputs on every
    golden case, the rough ones included (teams above 8 lanes always scan point by point)."""
    from pysurfinv_amd import _lib
    assert _lib.lib().surfdisp_set_team(team) == 0
    try:
        for name, d in ref_cases.items():
            c0, u0, s0 = hip.forward_batch(d["model"], d["periods"], d["kind"])
            c1, u1, s1 = hip.forward_batch(d["model"], d["periods"], d["kind"], fast_scan=True)
            assert np.array_equal(c0, c1) and np.array_equal(s0, s1), name
            assert np.array_equal(u0, u1, equal_nan=True), name
    finally:
        _lib.lib().surfdisp_set_team(0)


def test_fast_scan_random_stacks_and_independent_mode(hip):
    """Random stacks (smooth and rough, with and without low-velocity layers, thin and thick): the opt-in
    fast scan is bit-identical to the default point-by-point scan; also under SURFDISP_INDEPENDENT."""
    from pysurfinv_amd import synth, _lib
    rng = np.random.default_rng(8)
    for it in range(12):
        L = int(rng.integers(3, 30))
        m = synth.synth_models(1024, L, seed=int(rng.integers(1 << 30)), noise=float(rng.choice([0.02, 0.05, 0.1, 0.2])),
                               monotone=bool(it % 2), total_thickness=float(rng.choice([30., 60., 120., 200., 400.])))
        per = np.sort(rng.uniform(4.0, 120.0, int(rng.integers(5, 30)))).astype(np.float32)
        kind = 1 + it % 2
        c0, u0, s0 = hip.forward_batch(m, per, kind)
        c1, u1, s1 = hip.forward_batch(m, per, kind, fast_scan=True)
        assert np.array_equal(c0, c1) and np.array_equal(s0, s1) and np.array_equal(u0, u1, equal_nan=True)
    m = synth.synth_models(512, 10, seed=1)
    per = synth.default_periods(20)
    c0, u0, s0 = hip.forward_batch(m, per, 2, independent=True)
    c1, u1, s1 = hip.forward_batch(m, per, 2, independent=True, fast_scan=True)
    assert np.array_equal(c0, c1) and np.array_equal(s0, s1)


def test_fast_scan_soft_sediments_over_rock(hip):
    """Soft sediments over rock at short periods: fundamental and first higher mode 0.02 km/s apart, e^{kd} factors
    of many orders of magnitude in the secular function (a curvature test on the function itself let 5e-5 of these
    values slip; the vertical-phase rule and the test on ln|Delta| close it).  ~1.5 M phase velocities, bit-identical."""
    from pysurfinv_amd import synth, _lib
    rng = np.random.default_rng(78)
    try:
        for it in range(6):
            L = int(rng.integers(4, 40)); team = (2, 4, 8)[it % 3]
            _lib.lib().surfdisp_set_team(team)
            m = synth.sediment_models(8192, L, seed=100 + it, noise=float(rng.choice([0.02, 0.1, 0.2])),
                                      total_thickness=float(rng.choice([30., 120., 400.])))
            per = np.sort(rng.uniform(0.3, 30.0, 32)).astype(np.float32)
            for kind in (1, 2):
                c0, u0, s0 = hip.forward_batch(m, per, kind | 0x10)
                c1, u1, s1 = hip.forward_batch(m, per, kind | 0x10, fast_scan=True)
                assert np.array_equal(c0, c1) and np.array_equal(s0, s1), (it, kind, int((c0 != c1).sum()))
    finally:
        _lib.lib().surfdisp_set_team(0)


def test_pipelined_hint_changes_only_the_launch(hip, ref_cases):
    """SURFDISP_PIPELINED picks the team size for two batches in flight; the results stay within the parity
    band of every team size and the zero pattern is the same."""
    import torch
    from pysurfinv_amd import forward
    d = ref_cases["synth_L10_R"]
    model = torch.from_numpy(np.ascontiguousarray(np.repeat(d["model"], 512, axis=0))).cuda()
    per = torch.from_numpy(d["periods"]).cuda()
    plan = forward.BatchPlan(model.shape[0], model.shape[2], per.numel())
    c0, u0, s0 = (t.clone() for t in plan.run(model, per, kind=2))
    c1, u1, s1 = plan.run(model, per, kind=2, pipelined=True)
    assert torch.equal(s0, s1) and torch.equal(c0 > 0, c1 > 0)
    assert float((c1 / c0 - 1).abs().max()) < 4e-6 and float((u1 / u0 - 1).abs().max()) < 1e-5


def test_exact_fallback_takes_only_what_it_must(hip, ref_families):
    """Two-tier root search: the bench-like workload hands nothing to the exact fallback kernel; stacks whose
    secular function leaves the fp32 range all go there (and come back with the reference's answers,
    test_fp32_overflow_regime_follows_the_reference / test_soak_family_fixtures)."""
    import torch
    from pysurfinv_amd import synth, forward
    per = torch.from_numpy(synth.default_periods(20)).cuda()
    plan = forward.BatchPlan(4096, 10, 20)
    plan.run(torch.from_numpy(synth.synth_models(4096, 10, seed=0)).cuda(), per, kind=2)
    assert plan.fallback_count() == 0
    plan.run(torch.from_numpy(synth.synth_models(4096, 10, seed=0)).cuda(), per, kind=1)
    assert plan.fallback_count() == 0
    d = ref_families["overflow_R"]
    m = torch.from_numpy(np.ascontiguousarray(d["model"])).cuda()
    plan = forward.BatchPlan(m.shape[0], m.shape[2], len(d["periods"]))
    plan.run(m, torch.from_numpy(d["periods"]).cuda(), kind=2, nlay=torch.from_numpy(d["nlay"]).cuda())
    assert plan.fallback_count() == m.shape[0]


def test_prep_lanes_per_stack_do_not_change_the_answer(hip):
    """The prep kernel spreads a stack over 1 .. 64 lanes depending on the batch size (flattening factors: fp64 pow /
    log per layer).  The same stacks alone (a wavefront per stack) and inside a batch of 70 000 (two lanes per stack)
    give bit-identical c and U; ragged layer counts and a water layer included."""
    from pysurfinv_amd import synth
    per = synth.default_periods(12)
    L = 21
    small = synth.synth_models(48, L, seed=77, noise=0.08, monotone=False)
    small[:8, 1, 0] = 0.0; small[:8, 0, 0] = 1.475; small[:8, 2, 0] = 1.027; small[:8, 4, 0] = 1e-4; small[:8, 3, 0] = 2.0
    nl_small = np.random.default_rng(1).integers(3, L + 1, 48).astype(np.int32)
    big = synth.synth_models(70000, L, seed=78)
    big[:48] = small
    nl_big = np.full(70000, L, np.int32); nl_big[:48] = nl_small
    from pysurfinv_amd import _lib
    _lib.lib().surfdisp_set_team(4)                       # same root-search teams in both launches
    try:
        for kind in (2, 1):
            c0, u0, s0 = hip.forward_batch(small, per, kind, nlay=nl_small)
            c1, u1, s1 = hip.forward_batch(big, per, kind, nlay=nl_big)
            assert np.array_equal(c0, c1[:48]) and np.array_equal(u0, u1[:48], equal_nan=True) and np.array_equal(s0, s1[:48])
    finally:
        _lib.lib().surfdisp_set_team(0)


def test_liquid_layer_below_the_top_goes_through_the_exact_kernel(hip):
    """The production Rayleigh recursion tests for a liquid layer only at the top (water); a stack with Vs = 0 further
    down - the reference applies its liquid-layer matrix there too (surfa.f:216-251) - is re-solved by the exact
    fallback kernel and still equals the oracle."""
    import torch
    from oracle import cport
    from pysurfinv_amd import synth, forward
    per = synth.default_periods(10)
    m = synth.synth_models(32, 8, seed=12)
    m[:16, 1, 2] = 0.0                                    # a liquid layer inside half of the stacks
    c, u, st = hip.forward_batch(m, per, 2)
    co, uo, so = cport.forward_batch(m, per, 2, nthreads=4)
    assert np.array_equal(c > 0, co > 0)
    assert relerr(c, co) < TOL_C
    plan = forward.BatchPlan(32, 8, 10)
    plan.run(torch.from_numpy(m).cuda(), torch.from_numpy(per).cuda(), kind=2)
    assert plan.fallback_count() == 16


@pytest.mark.parametrize("case", CASES)
def test_strict_mode_golden_cases(hip, ref_cases, case):
    """SURFDISP_STRICT sends EVERY stack through the kernel that restates DLTAR4 / DLTAR1 / NEVILL statement by statement
    (the one the default mode keeps for stacks that leave the fp32 range).  Against the reference Fortran's vectors:
    c <= 1e-6-ish (libm's last bit is all that differs), same zero patterns - except at the osculation entries of
    u_exceptions.json, where the reference's own two builds disagree and a last-bit difference decides whether the
    grid sees the sign change (measured: one entry, rough_L64_R stack 0 period 13; scripts/strict_check.py)."""
    d = ref_cases[case]
    c, u, st = hip.forward_batch(d["model"], d["periods"], d["kind"], strict=True)
    keep = np.ones(c.shape[0], bool)
    for b, _k in U_EXCEPTIONS.get(case, ()):
        keep[b] = False
    assert _same_zero_pattern(c[keep], d["c"][keep])
    both = (c > 0) & (d["c"] > 0)
    if both.any():
        assert np.abs(c[both].astype(np.float64) / d["c"][both] - 1.0).max() < 2e-6
    u_keep = np.where(both, u, 0.0)
    _check_u(u_keep, np.where(both, d["u"], 0.0), case)


@pytest.mark.parametrize("kind,B,L", [(2, 65536, 10), (1, 65536, 10), (2, 16384, 64)])
def test_default_mode_agrees_with_strict_mode_at_full_size(hip, kind, B, L):
    """Full-size differential without a CPU in the loop: the production root search (factorised recursion, team
    subdivision) against the statement-by-statement kernel on the bench batch (BASELINE configs[1]: 65 536 x L10 x P20)
    and on a 64-layer batch.  Measured (scripts/strict_check.py): same zero patterns, c within 5.5e-6 (two root finders
    stopping inside the same 1e-6-wide NEVILL tolerance), U within 2.3e-5, 99.9 % of U within 7e-7."""
    import torch
    from pysurfinv_amd import synth, forward
    per = synth.default_periods(20)
    m = torch.from_numpy(synth.synth_models(B, L, seed=1)).cuda()
    pt = torch.from_numpy(per).cuda()
    plan = forward.BatchPlan(B, L, 20)
    plan.run(m, pt, kind=kind, strict=True)
    assert plan.fallback_count() == B                      # every stack went through the exact kernel
    cs, us = plan.c.cpu().numpy().copy(), plan.u.cpu().numpy().copy()
    plan.run(m, pt, kind=kind)
    assert plan.fallback_count() == 0
    cd, ud = plan.c.cpu().numpy(), plan.u.cpu().numpy()
    assert (cs > 0).all() and np.array_equal(cd > 0, cs > 0)
    # (L64: 9.3e-6 - brackets that hold a layer's S velocity are refined on signs down to the rounding of a 64-layer
    # recursion, where the two arithmetics' sign changes sit up to 1e-5 apart; r03, which interpolated across the kink: 5.4e-6)
    assert np.abs(cd.astype(np.float64) / cs - 1.0).max() < 1.3e-5
    eu = np.abs(ud.astype(np.float64) / us - 1.0)
    assert eu.max() < 1e-4 and np.quantile(eu, 0.999) < 5e-6


def test_entries_may_be_called_from_several_threads(hip):
    """include/surfdisp.h: the solve entries keep no state between calls and may be called from several threads.  Four
    host threads (ctypes releases the GIL) hammer the host-buffer entry and the f2py-shaped single-stack entry at the same
    time, Rayleigh and Love; every result equals the one obtained serially, bit for bit."""
    import threading
    from pysurfinv_amd import synth, fast_surf
    per = synth.default_periods(12)
    jobs = [(synth.synth_models(700 + 37 * i, 9, seed=300 + i), 2 - (i % 2)) for i in range(8)]
    serial = [hip.forward_batch(m, per, kind) for m, kind in jobs]
    one = jobs[0][0][0]
    per200 = np.zeros(200, np.float32); per200[:len(per)] = per          # f2py shape: cvper is real*4[200]
    f_serial = fast_surf.fast_surf(9, 2, one[0], one[1], one[2], one[3], one[4], per200, len(per))
    errors = []

    def worker(t):
        try:
            for it in range(6):
                i = (t * 3 + it) % len(jobs)
                c, u, st = hip.forward_batch(jobs[i][0], per, jobs[i][1])
                if not (np.array_equal(c, serial[i][0]) and np.array_equal(u, serial[i][1], equal_nan=True)
                        and np.array_equal(st, serial[i][2])):
                    errors.append(("batch", t, it, i))
                f = fast_surf.fast_surf(9, 2, one[0], one[1], one[2], one[3], one[4], per200, len(per))
                if not all(np.array_equal(a, b) for a, b in zip(f, f_serial)):
                    errors.append(("fast_surf", t, it))
        except Exception as e:                                   # noqa: BLE001 - reported through the list
            errors.append(("exception", t, repr(e)))

    # ... and two more threads push LARGE host-buffer calls through (chunked over each thread's own three streams and
    # grow-only device buffer), then give their buffers back
    big = synth.synth_models(70000, 10, seed=77)
    big_serial = hip.forward_batch(big, per, 2)

    def big_worker(t):
        try:
            from pysurfinv_amd import _lib
            for it in range(3):
                c, u, st = hip.forward_batch(big, per, 2)
                if not (np.array_equal(c, big_serial[0]) and np.array_equal(u, big_serial[1], equal_nan=True)
                        and np.array_equal(st, big_serial[2])):
                    errors.append(("big batch", t, it))
            _lib.lib().surfdisp_thread_release()
        except Exception as e:                                   # noqa: BLE001
            errors.append(("exception", t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)] + \
              [threading.Thread(target=big_worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads)
    assert not errors, errors[:4]


def _oracle_ratio(model, per, nlay=None):
    """Ellipticities of the CPU oracle (surfdisp_oracle_forward_dbg), stack by stack."""
    from oracle import cport
    O = cport.lib()
    fp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    B, _, L = model.shape
    P = len(per)
    c = np.zeros((B, P), np.float32); u = np.zeros((B, P), np.float32); r = np.zeros((B, P), np.float32)
    p32 = np.ascontiguousarray(per, np.float32)
    for i in range(B):
        n = L if nlay is None else int(nlay[i])
        m = np.ascontiguousarray(model[i][:, :n])
        O.surfdisp_oracle_forward_dbg(n, 2, fp(m[0]), fp(m[1]), fp(m[2]), fp(m[3]), fp(m[4]), fp(p32), P, fp(c[i]), fp(u[i]), fp(r[i]))
    return c, u, r


@pytest.mark.parametrize("case", ["synth_L10_R", "synth_L64_R", "water_L9_R", "two_layer_R", "synth_L21_R",
                                  "rough_L10_R", "rough_L64_R", "rough_thick_L22_R"])
def test_ellipticity_output_abi3(hip, ref_cases, case):
    """ABI 3 (surfdisp_forward_batch_device2): the Rayleigh ellipticity the reference computes and keeps in COMMON /o/
    ratio(k, 1) (calcul.f:195, surfa.f:360-363).  Checked against the reference itself where its shared object travelled
    (oracle/_ref, read through oracle/refso.last_ratio) and against the CPU oracle, whose value is bit-identical to the
    reference's.  Tolerance 1e-4 relative (the north_star bar), c + U call and phase-only call, every team size."""
    import torch
    from oracle import refso
    from pysurfinv_amd import _lib
    d = ref_cases[case]
    model = np.ascontiguousarray(d["model"], np.float32)
    per = np.ascontiguousarray(d["periods"], np.float32)
    co, uo, ro = _oracle_ratio(model, per)
    if refso.available():
        cr, ur, rr = refso.forward_batch(model[:, 0], model[:, 1], model[:, 2], model[:, 3], model[:, 4], per, 2, want_ratio=True)
        solved = cr > 0
        assert np.array_equal(rr[solved], ro[solved])              # oracle == reference, bit for bit
    B, _, L = model.shape
    dm, dp = torch.from_numpy(model).cuda(), torch.from_numpy(per).cuda()
    plan = hip.BatchPlan(B, L, len(per))
    strict_r = None
    for team in (0, 4, 16, 64):
        _lib.lib().surfdisp_set_team(team)
        for kind in (2, 2 | 0x10):
            c, u, st, r = plan.run(dm, dp, kind=kind, want_ratio=True)
            torch.cuda.synchronize()
            c, r = c.cpu().numpy(), r.cpu().numpy()
            assert _same_zero_pattern(c, co)
            ok = co > 0
            assert np.array_equal(r[~ok], np.zeros_like(r[~ok]))   # unsolved periods: 0
            # (the ratio changes sign along the period list of a water-covered stack: absolute floor for the entries near zero)
            # rough stacks (low-velocity zones, stale deep layers: what the ellipticity kernel's replay of the working stack's
            # history is for): held to 1e-3, except where the ratio is ill-conditioned in c - there the exact arithmetic on the
            # same device (SURFDISP_STRICT, root 1e-7 away) moves it by more than the bar as well (rough_thick_L22_R stack 30,
            # 3.3 s: oracle 0.648, strict 0.671, default 0.641 - and a group velocity within 2.4e-5 all the same)
            rel = 1e-3 if case.startswith("rough") else 1e-4
            okk = ok.copy()
            for bb, kk in U_EXCEPTIONS.get(case, ()):              # the listed osculation entries: as for U
                okk[bb, kk] = False
            viol = np.where(okk, np.abs(r - ro) - (rel * np.abs(ro) + 3e-5), -1.0)
            if case.startswith("rough") and (viol > 0).any():
                if strict_r is None:
                    strict_r = plan.run(dm, dp, kind=2, want_ratio=True, strict=True)[3].cpu().numpy().copy()
                illc = np.abs(strict_r - ro) > 0.3 * (rel * np.abs(ro) + 3e-5)
                assert (viol > 0).sum() <= 0.01 * okk.sum() and ((viol <= 0) | illc).all(), (case, team, kind, int((viol > 0).sum()))
                continue
            w = np.unravel_index(np.argmax(viol), viol.shape)
            assert viol[w] <= 0, (case, team, kind, w, float(r[w]), float(ro[w]), float(c[w]), float(co[w]))
    _lib.lib().surfdisp_set_team(0)
    # Love: zeros
    c, u, st, r = plan.run(dm, dp, kind=1, want_ratio=True)
    torch.cuda.synchronize()
    assert float(r.abs().max()) == 0.0


def test_soak_offender_fixture_every_team_size(hip):
    """The stacks on which the r03 library returned ANOTHER ROOT than the reference (tests/golden/ref_offenders.npz, from
    the r04 differential soak; reference outputs by make_golden_offenders.py): water layer over soft sediments with
    |Delta| ~ 1e20 (the refinement's reciprocals flushed to zero and the bracket's low end came back), and Love overtones
    1e-3 km/s apart (several roots per bracket, invisible to a small team's subdivision).  Every stack whose roots are
    DEFINED by the reference's formulas (its FMA build and two other roundings of exp / flattening agree to 2e-5) must come
    back on the reference's roots - zero pattern equal, c within 1e-4 - for every team size, and at most one of them may
    miss for any team size (the soak's residual rate on such stacks is 1e-6)."""
    from pysurfinv_amd import _lib
    f = np.load(os.path.join(GOLDEN, "ref_offenders.npz"))
    nst = len(f["nlay"])
    r03_bad = 0
    bad = {}
    for q in range(nst):
        n, P, kind = int(f["nlay"][q]), int(f["P"][q]), int(f["kind"][q])
        m = np.ascontiguousarray(f["model"][q][None, :, :n]); per = f["per"][q][:P]; cr = f["c"][q][:P]
        with np.errstate(all="ignore"):
            r03_bad += int((np.abs(f["c_r03"][q][:P] / cr - 1)[cr > 0] > 1e-4).any())
        if not f["defined"][q]:
            continue
        for team in (0, 1, 2, 4, 8, 16, 64):
            _lib.lib().surfdisp_set_team(team)
            try:
                c, u, st = hip.forward_batch(m, per, kind)
            finally:
                _lib.lib().surfdisp_set_team(0)
            with np.errstate(all="ignore"):
                ok = np.array_equal(c[0] > 0, cr > 0) and (np.abs(c[0][cr > 0] / cr[cr > 0] - 1) < 1e-4).all()
            if not ok:
                bad.setdefault(q, []).append(team)
    assert r03_bad == nst                                      # (the fixture is what it says: every stack was off in r03)
    assert len(bad) <= 1, bad
