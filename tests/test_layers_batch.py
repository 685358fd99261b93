"""batched parameters -> layer stack, pinned by outputs of the imported reference
(tests/golden/ref_driver.npz, captured by tests/golden/make_golden_driver.py).  CPU only."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
from settings import CONT, OCEAN, PERIODS           # noqa: E402
from pysurfinv_amd.layers_batch import Model1DBatch, bspline_basis
from pysurfinv_amd import brownian

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_driver.npz"))


@pytest.mark.parametrize("key", [k for k in G.files if k.startswith("bspl/")])
def test_bspline_basis_matches_reference_BsplBasis(key):
    N, nb, deg = key.split("/")[1].split("_")
    deg = None if deg == "None" else int(deg)
    mine = bspline_basis(int(N) + 1, int(nb), deg)
    assert mine.shape == G[key].shape
    assert np.abs(mine - G[key]).max() < 1e-12


@pytest.mark.parametrize("name,setting", [("cont", CONT), ("ocean", OCEAN)])
def test_layer_stacks_match_reference_seisPropLayers(name, setting):
    m = Model1DBatch(setting)
    params = torch.from_numpy(G[f"{name}/params"])
    assert m.spec.n == params.shape[1]
    assert np.allclose(m.spec.v0, G[f"{name}/params"][0])          # first captured model = initial model
    (h, vs, vp, rho, qs, qp), nlay = m.seis_prop_layers(params)
    ref = G[f"{name}/layers"]
    assert np.array_equal(nlay.numpy(), G[f"{name}/nlay"])         # ragged: ocean has 64..67 layers
    for a, r in zip((h, vs, vp, rho, qs, qp), np.moveaxis(ref, 1, 0)):
        assert a.shape == r.shape
        assert np.abs(a.numpy() - r).max() < 1e-9
    model, nl = m.to_model(params)
    assert model.dtype == torch.float32 and model.shape[1] == 5
    # rows are (vp, vs, rho, h, 1/Qs): the fast_surf argument order
    assert np.allclose(model[:, 1].numpy(), ref[:, 1], atol=1e-6)
    assert np.allclose(model[:, 3].numpy(), ref[:, 0], atol=1e-5)


GG = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_grids.npz"))


@pytest.mark.parametrize("name,setting", [("cont", CONT), ("ocean", OCEAN)])
def test_grid_points_match_reference_seisPropGrids(name, setting):
    """Model1D.seisPropGrids (models.py:72-91) - depth, Vs, Vp, rho, Qs, Qp at the grid points of every layer, interface
    points doubled, and the group of each point - for 12 parameter vectors drawn by the reference's own reset()
    (tests/golden/make_golden_grids.py): <= 1e-9, ragged point counts (the ocean models have 68..71) included."""
    m = Model1DBatch(setting)
    params = torch.from_numpy(GG[f"{name}/params"])
    (z, vs, vp, rho, qs, qp), grp, ngrid = m.seis_prop_grids(params)
    ref = GG[f"{name}/grids"]
    assert np.array_equal(ngrid.numpy(), GG[f"{name}/ngrid"]) and len(np.unique(GG["ocean/ngrid"])) > 1
    assert list(Model1DBatch.GROUP_NAMES) == list(GG["groups"])
    for a, r in zip((z, vs, vp, rho, qs, qp), np.moveaxis(ref[:, :6], 1, 0)):
        assert a.shape == r.shape and np.abs(a.numpy() - r).max() < 1e-9
    for i, n in enumerate(GG[f"{name}/ngrid"]):
        assert np.array_equal(grp[i, :n].numpy(), ref[i, 6, :n].astype(np.int64)) and bool((grp[i, n:] == -1).all())
    # Model1D.value(zdeps) and moho() (models.py:104-112: on the grids WITHOUT the reference mantle, NaN outside)
    val = m.value(params, GG["zdeps"])
    assert val.shape == GG[f"{name}/value"].shape and np.array_equal(np.isnan(val), np.isnan(GG[f"{name}/value"]))
    assert np.nanmax(np.abs(val - GG[f"{name}/value"])) < 1e-9 and np.isnan(val).any()
    assert np.abs(m.moho(params) - GG[f"{name}/moho"]).max() < 1e-9
    with pytest.raises(ValueError):
        m.value(params, [1.0], type="vp")
    # seisPropLayers is the midpoint form of these grids (models.py:93-102)
    (h, vsl, *_), nlay = m.seis_prop_layers(params)
    i = 0
    zz, vv = z[i, :ngrid[i]].numpy(), vs[i, :ngrid[i]].numpy()
    keep = np.diff(zz) > 0.01
    assert np.allclose(h[i, :nlay[i]].numpy(), np.diff(zz)[keep]) and np.allclose(vsl[i, :nlay[i]].numpy(), ((vv[1:] + vv[:-1]) / 2)[keep])


def _same(a, b, tol=1e-12):
    if isinstance(a, dict):
        return isinstance(b, dict) and list(a) == list(b) and all(_same(a[k], b[k], tol) for k in a)
    if isinstance(a, (list, tuple)):
        return isinstance(b, (list, tuple)) and len(a) == len(b) and all(_same(x, y, tol) for x, y in zip(a, b))
    if isinstance(a, (int, float)) and not isinstance(a, bool) and isinstance(b, (int, float)):
        return abs(a - b) <= tol * max(1.0, abs(b))
    return a == b


@pytest.mark.parametrize("name,setting", [("cont", CONT), ("ocean", OCEAN)])
def test_to_yml_is_the_references_toYML_and_reads_back(name, setting, tmp_path):
    """Model1D.toYML() (models.py:60-70) - the `setting` the reference stores in its {pid}.npz files: LayerName keys, every
    random-walk entry as [v, vmin, vmax, step] - for the initial model and for a prior draw; Model1DBatch reads that form
    back (also from a YAML file, as buildModel1D does) and builds the same model."""
    import json
    import yaml
    m = Model1DBatch(setting)
    ref0, ref3 = json.loads(str(GG[f"{name}/toyml"]))
    assert _same(m.to_yml(), ref0), (m.to_yml(), ref0)
    assert _same(m.to_yml(GG[f"{name}/params"][3]), ref3)
    f = tmp_path / "setting.yml"
    f.write_text(yaml.safe_dump(ref3, sort_keys=False))
    for src in (ref3, str(f)):
        m3 = Model1DBatch(src)
        assert np.allclose(m3.spec.v0, GG[f"{name}/params"][3], atol=1e-12)
        assert np.allclose(m3.spec.vmin, m.spec.vmin) and np.allclose(m3.spec.vmax, m.spec.vmax) and np.allclose(m3.spec.step, m.spec.step)
        p = torch.from_numpy(GG[f"{name}/params"][:5])
        (h1, v1, *_), n1 = m.seis_prop_layers(p)
        (h3, v3, *_), n3 = m3.seis_prop_layers(p)
        assert torch.equal(n1, n3) and torch.equal(h1, h3) and torch.equal(v1, v3)


def test_param_spec_bounds_follow_BrownianVarMC():
    s = brownian.ParamSpec.from_entries([[2., 'abs_pos', 3., 0.1], [10., 'rel', 30, 9.], [1., 0.5, 1.6, 0.05],
                                         [4., 'rel_pos', 200, 0.4], [0., 'abs', 0.4, 0.01]])
    assert np.allclose(s.vmin, [0, 7, 0.5, 0, -0.4]) and np.allclose(s.vmax, [5, 13, 1.6, 12, 0.4])
    assert np.allclose(s.step, [0.1, 3.0, 0.05, 0.4, 0.01])        # step clipped to |vmax-vmin|/2


def test_torch_proposer_respects_bounds_and_step():
    spec = Model1DBatch(CONT).spec
    pr = brownian.TorchProposer(spec, "cpu", seed=0)
    v = torch.as_tensor(spec.v0)[None, :].repeat(20000, 1)
    new = pr.move(v)
    lo, hi = torch.as_tensor(spec.vmin), torch.as_tensor(spec.vmax)
    assert bool(((new > lo) & (new < hi)).all())
    d = (new - v).numpy()
    # interior parameters (bounds many steps away): plain Gaussian with the requested step
    assert abs(d[:, 4].std() / spec.step[4] - 1) < 0.03 and abs(d[:, 4].mean()) < 3 * spec.step[4] / 100
    r = pr.reset(20000)
    assert bool(((r >= lo) & (r <= hi)).all())
    assert abs(r[:, 3].mean().item() - 35.0) < 0.3                 # uniform on (25, 45)


@pytest.mark.gpu
def test_native_params_to_model_kernel_matches_torch_path_and_reference():
    """csrc/surfdisp_layers.hip (one launch) against the torch implementation and, through it, the
    reference's seisPropLayers outputs."""
    m = Model1DBatch(CONT, device="cuda:0")
    assert m.native_descriptor() is not None
    params = torch.from_numpy(G["cont/params"]).cuda()
    mt, nl = m.to_model_torch(params)
    mn, nn = m.to_model_native(params)
    torch.cuda.synchronize()
    assert nn is None and mn.shape == mt.shape == (40, 5, 96)
    assert float((mn - mt).abs().max()) < 2e-6                       # fp32 rounding of fp64 values
    ref = G["cont/layers"]
    assert np.allclose(mn[:, 1].cpu().numpy(), ref[:, 1], atol=1e-6)  # vs
    assert np.allclose(mn[:, 3].cpu().numpy(), ref[:, 0], atol=1e-5)  # h
    assert np.allclose(mn[:, 0].cpu().numpy(), ref[:, 2], atol=1e-6)  # vp
    assert np.allclose(mn[:, 2].cpu().numpy(), ref[:, 3], atol=1e-6)  # rho
    assert np.allclose(1.0 / mn[:, 4].cpu().numpy(), ref[:, 4], rtol=1e-6)
    # the oceanic setting's structure is not static inside its prior box: torch path is used
    mo = Model1DBatch(OCEAN, device="cuda:0")
    assert mo.native_descriptor() is None
    model, nlay = mo.to_model(torch.from_numpy(G["ocean/params"]).cuda())
    assert np.array_equal(nlay.cpu().numpy(), G["ocean/nlay"])


def test_unknown_random_walk_keys_raise_and_crust_ignores_deg():
    """A Brownian entry under a key the batch parser does not implement must not be dropped silently (the
    reference's _brownians() would count it, models.py:240-253); 'deg' only acts on the mantle profile
    (layers.py:169-172 vs 258-260)."""
    import copy
    from pysurfinv_amd.layers_batch import Model1DBatch
    bad = copy.deepcopy(CONT)
    bad["Crust"]["Moho"] = [0.0, "abs", 1.0, 0.1]
    with pytest.raises(ValueError):
        Model1DBatch(bad)
    a = copy.deepcopy(CONT); a["Crust"]["deg"] = 2
    ma, mb = Model1DBatch(a), Model1DBatch(CONT)
    import torch
    p = torch.as_tensor(np.asarray(mb.spec.v0)[None, :])
    assert torch.equal(ma.to_model(p)[0], mb.to_model(p)[0])


@pytest.mark.parametrize("tag", ["pg", "pg_nodup"])
def test_puregird_matches_reference(tag):
    """PureGird (models.py:163-186): a model frozen as grid profiles, cut into one piece per group; seisPropGrids /
    seisPropLayers / value / moho exactly as the reference's (fixtures: the continental start model with its reference
    mantle, with and without doubled interface points - without them the pieces close up at the group boundaries)."""
    from pysurfinv_amd.puregird import PureGird
    names = list(GG["groups"])
    prof = tuple(GG[f"{tag}/in"]) + ([names[i] for i in GG[f"{tag}/in_grp"]],)
    pg = PureGird(prof, info={})
    g = pg.seisPropGrids()
    assert np.array_equal(np.array(g[:6]), GG[f"{tag}/grids"]) and [names.index(x) for x in g[6]] == list(GG[f"{tag}/grids_grp"])
    L = pg.seisPropLayers()
    assert np.array_equal(np.array(L[:6]), GG[f"{tag}/layers"]) and len(L[6]) == GG[f"{tag}/layers"].shape[1]
    v = pg.value(GG["zdeps"])
    assert np.array_equal(np.isnan(v), np.isnan(GG[f"{tag}/value"])) and np.nanmax(np.abs(v - GG[f"{tag}/value"])) == 0
    assert pg.moho() == float(GG[f"{tag}/moho"])
    assert len(pg.layers) == 3 and pg.copy().moho() == pg.moho()


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["pg", "pg_nodup"])
def test_puregird_forward_on_the_device(tag):
    from pysurfinv_amd.puregird import PureGird
    names = list(GG["groups"])
    pg = PureGird(tuple(GG[f"{tag}/in"]) + ([names[i] for i in GG[f"{tag}/in_grp"]],), info={})
    c = pg.forward([8, 12, 20, 30, 45, 60, 80])
    assert c is not None and np.abs(np.asarray(c) / GG[f"{tag}/c"] - 1).max() < 1e-4      # the reference's own forward (flang build)
