"""Per-point local information (``Point(setting, localInfo)``, point.py:8-14; models.py:54-59,74; layers.py:350-363)
and the Crust 'Gauss' option (layers.py:176-183): pinned by tests/golden/ref_local.npz - every point built as its own
model by the imported reference (tests/golden/make_golden_local.py) - on CPU (torch path), on the GPU (HIP kernels), and
through the sharded grid driver with heterogeneous points (2 gloo ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from settings import OCEAN, PERIODS                                   # noqa: E402
from settings_therm import HYBRID_STATIC                              # noqa: E402
from settings_local import GAUSS, LOCAL_TABLES                        # noqa: E402
from pysurfinv_amd.layers_batch import Model1DBatch                   # noqa: E402

G = np.load(os.path.join(HERE, "golden", "ref_local.npz"))
SETTINGS = {"hyb": HYBRID_STATIC, "ocean": OCEAN, "gauss": GAUSS}


def _model(name, device="cpu"):
    keys, table = LOCAL_TABLES[name]
    m = Model1DBatch(SETTINGS[name], device=device, local_keys=keys)
    assert m.aux_names == list(keys)
    m.set_local_info(np.asarray(table, float))
    return m


@pytest.mark.parametrize("name", ["hyb", "ocean", "gauss"])
def test_per_point_stacks_match_reference_models(name):
    """>= 8 points with different topo / lithoAge / period / fixed thickness (or Gaussian centre): every parameter
    vector's stack equals what the reference builds for THAT point's own (setting, localInfo), to 1e-9."""
    m = _model(name)
    params = torch.from_numpy(G[f"{name}/params"])
    rows = torch.from_numpy(G[f"{name}/rows"])
    assert len(np.unique(G[f"{name}/rows"])) >= 8
    (h, vs, vp, rho, qs, qp), nlay = m.seis_prop_layers(params, rows)
    ref = G[f"{name}/layers"]
    assert np.array_equal(nlay.numpy(), G[f"{name}/nlay"])
    for a, r in zip((h, vs, vp, rho, qs, qp), np.moveaxis(ref, 1, 0)):
        assert a.shape == r.shape
        assert np.abs(a.numpy() - r).max() < 1e-9
    # and the points really differ from one another
    assert np.abs(ref[0] - ref[3 * 3]).max() > 1e-3


def test_local_keys_are_checked():
    with pytest.raises(ValueError):
        Model1DBatch(OCEAN, local_keys=["OceanCrust.H"])                # a random-walk entry cannot be per point
    with pytest.raises(ValueError):
        Model1DBatch(OCEAN, local_keys=["Moho"])                        # names nothing
    m = Model1DBatch(OCEAN, local_keys=["topo"])
    p = torch.as_tensor(np.asarray(m.spec.v0)[None, :])
    with pytest.raises(ValueError):
        m.to_model(p)                                                   # no table yet
    m.set_local_info([[0.0], [1.0]])
    with pytest.raises(ValueError):
        m.to_model(p.repeat(3, 1))                                      # 3 vectors, 2 rows, no rows=
    a, _ = m.to_model(p.repeat(2, 1))
    assert float((a[0] - a[1]).abs().max()) > 1e-3                      # topo moved the BottomDepth layer
    st = m.setting_for_row([1.0])
    assert st["Info"]["topo"] == 1.0 and "topo" not in OCEAN["Info"]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["hyb", "gauss"])
def test_per_point_stacks_on_gpu(name):
    """The HIP kernels (surfdisp_layers_kernel / surfdisp_thermal_kernel read the local constants as extra columns of
    the parameter rows) against the reference fixtures, fp32 stacks."""
    m = _model(name, device="cuda:0")
    params = torch.from_numpy(G[f"{name}/params"]).cuda()
    rows = torch.from_numpy(G[f"{name}/rows"]).cuda()
    assert (m.native_descriptor() is not None) == (name == "hyb")       # the Gaussian term stays on the torch path
    model, nlay = m.to_model(params, rows)
    torch.cuda.synchronize()
    ref = G[f"{name}/layers"]
    mm = model.cpu().numpy()
    assert np.allclose(mm[:, 3], ref[:, 0], atol=2e-5)                  # h
    assert np.allclose(mm[:, 1], ref[:, 1], atol=2e-6)                  # vs
    assert np.allclose(mm[:, 0], ref[:, 2], atol=2e-6)                  # vp
    assert np.allclose(mm[:, 2], ref[:, 3], atol=2e-6)                  # rho
    assert np.allclose(1.0 / mm[:, 4], ref[:, 4], rtol=3e-6)            # qs (thermal: depends on lithoAge and period)
    if name == "hyb":
        mt, _ = m.to_model_torch(params, rows)
        assert float(((model - mt).abs() / mt.abs().clamp(min=1e-3)).max()) < 2e-6


@pytest.mark.gpu
def test_grid_with_local_info_is_the_same_for_any_chain_grouping():
    """run_grid on the device path with per-point constants: the chains advance as one launch per step or as chain groups on
    their own streams (each group reads its own rows of the local-information table and of the observations) - same mcTrack."""
    from pysurfinv_amd import grid
    keys, table = LOCAL_TABLES["hyb"]
    table = np.asarray(table, float)
    npts, chains, chainL = table.shape[0], 7, 5
    periods = np.asarray([8.0, 12.0, 18.0, 25.0, 33.0, 45.0, 60.0], np.float32)
    rng = np.random.default_rng(2)
    c_obs = 3.6 + 0.4 * np.linspace(0, 1, len(periods))[None, :] + 0.02 * rng.standard_normal((npts, len(periods)))
    unc = np.full_like(c_obs, 0.03)
    out = []
    for g in (1, 2, 3):
        m = Model1DBatch(SETTINGS["hyb"], device="cuda:0", local_keys=keys)
        r = grid.run_grid(m, np.arange(npts), np.zeros(npts), periods, c_obs, unc, chains, chainL, device="cuda:0", seed=5,
                          local_info=table, chain_groups=g, spec_depth=1)
        out.append((r["mcTrack"], r["summaries"]))
    assert out[0][0].shape == (npts, chains * chainL, 3 + m.spec.n)
    for tr, sm in out[1:]:
        assert np.array_equal(tr, out[0][0]) and np.allclose(sm, out[0][1], rtol=0, atol=1e-12, equal_nan=True)
    # the points' constants really reach the chains: two points' first rows (the initial model) have different misfits
    assert len(np.unique(np.round(out[0][0][:, 0, 0], 9))) > npts // 2


# ------------------------------------------------------------------ sharded grid with heterogeneous points (gloo, CPU)
NPTS, CHAINS, CHAINL = 6, 2, 3


def _oracle_forward(periods):
    from oracle import cport

    def fwd(model, nlay):
        c, u, st = cport.forward_batch(model.cpu().numpy(), periods, 2,
                                       nlay=None if nlay is None else nlay.cpu().numpy(), nthreads=2)
        return torch.from_numpy(c.astype(np.float64)), torch.from_numpy(st)
    return fwd


def _grid_inputs():
    keys, table = LOCAL_TABLES["ocean"]
    table = np.asarray(table, float)[:NPTS]
    per = np.asarray(PERIODS, np.float32)
    fwd = _oracle_forward(per)
    m = Model1DBatch(OCEAN, local_keys=keys).set_local_info(table)
    v0 = torch.as_tensor(np.tile(np.asarray(m.spec.v0)[None, :], (NPTS, 1)))
    model, nlay = m.to_model(v0)
    c0 = fwd(model, nlay)[0].numpy()                                    # every point's own start-model curve
    return keys, table, per, c0 * 1.004, np.full_like(c0, 0.02), c0


def _worker(rank, world, port, outdir, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pysurfinv_amd import grid
    keys, table, per, c_obs, unc, _ = _grid_inputs()
    mb = Model1DBatch(OCEAN, local_keys=keys)
    r = grid.run_grid(mb, np.arange(NPTS) * 0.5 + 230, np.arange(NPTS) * 0.25 + 44, per, c_obs, unc, CHAINS, CHAINL,
                      outdir=outdir, rank=rank, world=world, device="cpu", seed=3, forward=_oracle_forward(per),
                      local_info=table)
    q.put((rank, r["points"], r["mcTrack"], r["summaries"]))
    dist.barrier(); dist.destroy_process_group()


def test_two_rank_grid_with_heterogeneous_points(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn"); q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted((q.get(timeout=300) for _ in range(2)), key=lambda t: t[0])
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    (_, span0, tr0, sum0), (_, span1, tr1, sum1) = res
    assert span0 == (0, 3) and span1 == (3, 6) and np.array_equal(sum0, sum1)
    keys, table, per, c_obs, unc, c0 = _grid_inputs()
    assert np.abs(c0[0] - c0[3]).max() > 5e-4                           # the points' own curves differ (topo 0 vs 2.5 km)
    tr = np.concatenate([tr0, tr1], axis=0)                             # [NPTS, CHAINS * CHAINL, 3 + N]
    # first chain of every point starts at the initial model: its misfit is that POINT's start curve against that
    # point's observations - a chain reading another point's topo would miss by far more than the tolerance
    mis0 = np.sqrt((((c_obs - c0) / unc) ** 2).mean(axis=1))
    assert np.allclose(tr[:, 0, 0], mis0, rtol=1e-6)
    # the files carry each point's own setting (what Point(setting, localInfo) would have held)
    for i in range(NPTS):
        d = np.load(os.path.join(tmp_path, f"{230 + 0.5 * i}_{44 + 0.25 * i}.npz"), allow_pickle=True)
        assert d["setting"].item()["Info"]["topo"] == table[i, 0]
