"""Pins the CPU oracle (oracle/surfdisp_oracle.c) -- CPU only.

(i)  against the reference's own known-answer data senskernel-1.0/TEST1
     (fp64 twin of the same algorithm; the fp32 reference itself differs from it
     by 2.5e-5 (c_R), 5.9e-5 (U_R), 1.9e-5 (c_L), 8.7e-5 (U_L), so the bound is 1e-4);
(ii) against outputs captured from the reference Fortran (tests/golden/ref_cases.npz):
     the restatement is expected to be BIT-EXACT (same operation order, no FMA).
"""
import numpy as np
import pytest

from conftest import relerr, load_cases
from oracle import cport

CASES = sorted(load_cases().keys())


def test_oracle_builds_and_loads():
    cport.lib()


@pytest.mark.parametrize("wave", ["R", "L"])
def test_oracle_vs_reference_known_answers_TEST1(eus, wave):
    kind = 2 if wave == "R" else 1
    c, u, st = cport.forward_batch(eus["model"], eus["periods"], kind)
    assert st[0] == cport.OK
    assert relerr(c[0], eus[f"c_{wave}_fp64twin"]) < 1e-4
    assert relerr(u[0], eus[f"u_{wave}_fp64twin"]) < 1e-4
    # and bit-exact against the fp32 reference run on the same stack
    assert np.array_equal(c[0], eus[f"c_{wave}_ref"])
    assert np.array_equal(u[0], eus[f"u_{wave}_ref"])


@pytest.mark.parametrize("case", CASES)
def test_oracle_bit_exact_vs_captured_reference(ref_cases, case):
    d = ref_cases[case]
    c, u, st = cport.forward_batch(d["model"], d["periods"], d["kind"])
    assert np.array_equal(c, d["c"]), f"c differs: {relerr(c, d['c'])}"
    if case.startswith("water") and d["kind"] == 2:
        # REIGEN's water-layer terms use COMPLEX csin/ccos/csqrt (surfa.f:879-910); the oracle
        # restates them on the two real branches (sin/cos, sinh/cosh), which rounds
        # differently from flang's complex runtime in the last bit of a few values.
        assert np.array_equal(u != 0, d["u"] != 0)
        assert relerr(u, d["u"]) <= 5e-7
    else:
        assert np.array_equal(u, d["u"]), f"u differs: {relerr(u, d['u'])}"
    # status words agree with the zero pattern (fast_surf.f:197, calcul.f:203-219)
    solved_all = np.all(d["c"] > 0, axis=1)
    assert np.array_equal(st == cport.OK, solved_all)


def test_oracle_f2py_shaped_call(ref_cases):
    d = ref_cases["c1_single_L5_R"]
    m = d["model"][0]
    per = np.zeros(200); per[:20] = d["periods"]
    ur, ul, cr, cl = cport.fast_surf(5, 2, m[0], m[1], m[2], m[3], m[4], per, 20)
    assert cr.shape == (200,) and cr.dtype == np.float32
    assert np.array_equal(cr[:20], d["c"][0]) and np.all(cr[20:] == 0)
    assert np.array_equal(ur[:20], d["u"][0])
    assert not ul.any() and not cl.any()


def test_oracle_rejects_bad_arguments():
    r = cport.forward([6.], [3.], [2.7], [0.], [0.01], [10.], 2)     # one layer
    assert r["status"] == cport.EINVAL


def test_oracle_openmp_batch_is_deterministic(ref_cases):
    d = ref_cases["synth_L10_R"]
    c1, u1, _ = cport.forward_batch(d["model"], d["periods"], 2, nthreads=1)
    c4, u4, _ = cport.forward_batch(d["model"], d["periods"], 2, nthreads=4)
    assert np.array_equal(c1, c4) and np.array_equal(u1, u4)


def test_work_model_delta_evaluations(ref_cases):
    """~1 035 secular-function evaluations per 20-period Rayleigh solve."""
    d = ref_cases["synth_L10_R"]
    m = d["model"][0]
    r = cport.forward(m[0], m[1], m[2], m[3], m[4], d["periods"], 2)
    assert 700 < r["n_delta"] < 1500


def test_scan_trace_brackets_the_oracle_root(ref_cases):
    """Developer aid ``surfdisp_oracle_scan_trace`` (scripts/dev_scanfail.py): the walk it records for period k
    starts at 0.9 c(k-1), advances on the fp32 0.01 km/s grid and its first sign change brackets the oracle's c(k)."""
    import ctypes
    L = cport.lib()
    fp = ctypes.POINTER(ctypes.c_float); ip = ctypes.POINTER(ctypes.c_int)
    d = ref_cases["synth_L10_R"]
    m = np.ascontiguousarray(d["model"][0], np.float32); per = np.ascontiguousarray(d["periods"], np.float32)
    c_ref = d["c"][0]
    rows = [np.ascontiguousarray(m[i]) for i in range(5)]
    for k in (0, 3, len(per) - 1):
        cap = 2000
        ct = np.zeros(cap, np.float32); dt = np.zeros(cap, np.float32); mt = np.zeros(cap, np.int32)
        n = L.surfdisp_oracle_scan_trace(m.shape[1], int(d["kind"]), *[r.ctypes.data_as(fp) for r in rows],
                                         per.ctypes.data_as(fp), len(per), k, 8,
                                         ct.ctypes.data_as(fp), dt.ctypes.data_as(fp), mt.ctypes.data_as(ip), cap)
        assert 10 < n <= cap
        if k:
            assert ct[0] == np.float32(0.9) * c_ref[k - 1]
        step = np.diff(ct[:n])
        assert np.all(np.abs(step - 0.01) < 1e-5)
        sg = np.signbit(dt[:n])
        first = int(np.nonzero(sg[1:] != sg[:-1])[0][0])
        assert n - (first + 2) == 8                                # eight points recorded past the bracket
        assert ct[first] <= c_ref[k] <= ct[first + 1]
        assert np.all((mt[:n] >= 2) & (mt[:n] <= m.shape[1]))


def test_group_at_hook_reproduces_the_oracles_own_group_velocities(ref_cases):
    """surfdisp_oracle_forward_at with the oracle's own roots = the plain oracle, bit for bit; with a phase velocity
    one ulp off, U moves (that is what the hook is for)."""
    from oracle import cport
    for name in ("synth_L10_R", "synth_L10_L", "rough_L64_R", "water_L9_R"):
        d = ref_cases[name]
        c0, u0, _ = cport.forward_batch(d["model"], d["periods"], d["kind"])
        c1, u1 = cport.group_at(d["model"], d["periods"], d["kind"], c0)
        assert np.array_equal(c0, c1) and np.array_equal(u0, u1, equal_nan=True), name
        c2, u2 = cport.group_at(d["model"], d["periods"], d["kind"], np.nextafter(c0, np.float32(10)))
        assert np.array_equal(c0, c2) and not np.array_equal(u0, u2, equal_nan=True), name


def test_exception_list_is_what_the_spread_fixture_says(ref_cases):
    """tests/golden/u_exceptions.json lists exactly the entries at which the reference's own two builds
    (ref_cases.npz vs ref_spread.npz, tests/golden/make_golden_spread.py) differ by > 2e-5 in U or return NaN;
    their zero patterns and phase velocities agree (<= 2e-6) everywhere."""
    import json
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "ref_spread.npz"))
    listed = {(e["case"], e["stack"], e["period_index"]) for e in json.load(open(os.path.join(GOLDEN, "u_exceptions.json")))["entries"]}
    found = set()
    for name, d in ref_cases.items():
        c, u = z[f"{name}/c_fma"], z[f"{name}/u_fma"]
        assert np.array_equal(c > 0, d["c"] > 0)
        ok = d["c"] > 0
        if not ok.any():
            continue
        assert np.abs(c[ok].astype(np.float64) / d["c"][ok] - 1).max() < 2e-6
        with np.errstate(invalid="ignore"):
            su = np.abs(u.astype(np.float64) / np.where(ok, d["u"], 1) - 1)
        for b, k in zip(*np.where(ok & ~(su <= 2e-5))):
            found.add((name, int(b), int(k)))
    assert found == listed and len(listed) <= 4


@pytest.mark.parametrize("family", ["sediment_R", "sediment_L", "wild_R", "wild_L", "overflow_R", "overflow_L",
                                    "ragged_R", "ragged_L"])
def test_oracle_bit_exact_on_soak_family_fixtures(ref_families, family):
    """The C restatement against what the reference Fortran returned for the soak families
    (tests/golden/make_golden_families.py) - bit for bit, NaN for NaN.  overflow_* and wild_* pin the NaN semantics:
    the scan's SIGN(1., NaN) and NEVILL's arithmetic IFs (a NaN Neville abscissa takes the third label = a
    bisection step, surfa.f:32-34), which is why the reference returns roots - not failures - next to the
    overflowed region."""
    from oracle import cport
    d = ref_families[family]
    c, u, st = cport.forward_batch(d["model"], d["periods"], d["kind"], nlay=d["nlay"], nthreads=4)
    assert np.array_equal(c, d["c"]) and np.array_equal(u, d["u"], equal_nan=True)


# ---- analytic partials: the oracle against the reference's own COMMON /rar1/ (tests/golden/make_golden_partials.py)
PARTIALS = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "ref_partials.npz"))


@pytest.mark.parametrize("name", [str(n) for n in PARTIALS["names"]])
@pytest.mark.parametrize("wave", ["R", "L"])
def test_oracle_partials_bit_exact_vs_reference_common_block(name, wave):
    """dcda, dcdb, dcdr, dwx of every sublayer as REIGEN / LEIGEN leave them (surfa.f:1133-1135, 1182-1184, 1204-1207;
    Love 511-512, 564-565, 582-583), one-period calls, first mmax entries of the block: bit for bit."""
    kind = 2 if wave == "R" else 1
    m = PARTIALS[f"{name}_model"]
    blk, meta = PARTIALS[f"{name}_{wave}_rar1"], PARTIALS[f"{name}_{wave}_meta"]
    nz = 0
    for ip, T in enumerate(PARTIALS["periods"]):
        o = cport.partials(m[0], m[1], m[2], m[3], m[4], T, kind)
        c, u, mm, ndiv = meta[ip]
        assert np.float32(o["c"]) == np.float32(c) and o["ndiv"] == int(ndiv)
        if c <= 0:
            continue
        assert o["mmax"] == int(mm)
        if not (name.startswith("water") and kind == 2):           # (complex water-layer terms: last-bit differences)
            assert np.float32(o["u"]) == np.float32(u)
        for i, k in enumerate(("dcda", "dcdb", "dcdr", "dwx")):
            ref = blk[ip, i, :int(mm)]
            if name.startswith("water") and kind == 2:
                assert np.abs(o[k][:int(mm)] - ref).max() <= 2e-6 * max(np.abs(ref).max(), 1e-30)
            else:
                assert np.array_equal(o[k][:int(mm)], ref), (name, wave, T, k)
            nz += int(np.count_nonzero(ref))
    assert nz > 50


def test_oracle_bit_exact_on_the_soak_offender_fixture():
    """tests/golden/ref_offenders.npz (stacks on which the r03 library returned another root than the reference;
    make_golden_offenders.py): the restatement returns the reference's phase velocities bit for bit."""
    import os
    f = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_offenders.npz"))
    for q in range(len(f["nlay"])):
        n, P = int(f["nlay"][q]), int(f["P"][q])
        c, u, st = cport.forward_batch(np.ascontiguousarray(f["model"][q][None, :, :n]), f["per"][q][:P], int(f["kind"][q]))
        assert np.array_equal(c[0], f["c"][q][:P]), q


def test_oracle_on_the_zero_group_velocity_stack():
    """tests/golden/ref_zgv_stack.npz (reference outputs of the one stack on which the opt-in count-guided scan left the point-by-point
    scan): the oracle returns the reference's phase and group velocities bit for bit."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_zgv_stack.npz"))
    c, u, s = cport.forward_batch(d["model"][None], d["periods"], 2)
    assert np.array_equal(c[0], d["c"])
    assert np.array_equal(np.nan_to_num(u[0], nan=-1.0), np.nan_to_num(d["u"], nan=-1.0))
