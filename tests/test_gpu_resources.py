"""Device resources of the C-ABI objects: what cup3d_sim_create / the solvers / the mesh adaptation allocate, cup3d_sim_destroy gives
back (a long run of the reference adapts its mesh every 20 steps and rebuilds its device mirror each time, SURVEY 8b "Ownership")."""
import gc

import numpy as np
import pytest

import cup3d_amd as cu
from cup3d_amd.capi import check, lib

pytestmark = pytest.mark.gpu


def _free_bytes():
    import torch
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info()[0]


def _one_life(block_solver, adapt):
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=5 if adapt else 4, levelStart=3, extent=2 * np.pi, nu=0.01, BC_x="wall", BC_y="periodic",
                            BC_z="freespace", blockSolver=block_solver)
    g = sim.grid
    rng = np.random.default_rng(7)
    sim.upload("vel", rng.uniform(-1, 1, (g.nblocks, 8, 8, 8, 3)))
    S = cu.Simulation(sim)
    if adapt:
        cu.ComputeVorticity(S.sim)(0)
        w = S.sim.download("tmpV")
        linf = np.sqrt((w ** 2).sum(axis=-1)).reshape(S.sim.nblocks, -1).max(axis=1)
        S.adaptMesh(float(np.quantile(linf, 0.8)), -1.0)   # a second device mirror replaces the first
        assert S.sim.nblocks > g.nblocks
    S.sim.step = 5
    S.advance(0.01)
    assert S.sim.last_poisson.iterations > 0


@pytest.mark.parametrize("block_solver,adapt", [(0, False), (1, False), (5, False), (0, True)])
def test_create_step_destroy_returns_the_device_memory(block_solver, adapt):
    """20 lives of a simulation (create, upload, [adapt], one full step, destroy): the free device memory afterwards is what it was
    after the first life (pinned staging buffers and the profiler's event pool are allocated once per process)."""
    _one_life(block_solver, adapt)
    gc.collect()
    before = _free_bytes()
    for _ in range(20):
        _one_life(block_solver, adapt)
        gc.collect()
    after = _free_bytes()
    assert before - after < (8 << 20), f"{(before - after) / 2 ** 20:.1f} MiB of device memory lost over 20 create / destroy cycles"


def test_error_codes_instead_of_crashes():
    """No exception and no abort crosses the C boundary (SURVEY 8b "Errors"): misuse comes back as an error code with a message."""
    L = lib()
    assert L.cup3d_sim_upload(None, 0, np.zeros(1)) != 0
    sim = cu.SimulationData(bpdx=1, bpdy=1, bpdz=1, levelMax=2, levelStart=1, extent=1.0, BC_x="periodic", BC_y="periodic", BC_z="periodic")
    buf = np.zeros((sim.nblocks, 8, 8, 8))
    assert L.cup3d_sim_upload(sim.handle, 99, buf) != 0          # unknown field
    assert b"field" in L.cup3d_last_error()
    with pytest.raises(cu.capi.Cup3dError):
        check(L.cup3d_sim_upload(sim.handle, 99, buf))
    sim.blockSolver = 7                                                                              # unknown block solver code
    sim.upload("lhs", np.random.default_rng(0).uniform(-1, 1, buf.shape))
    with pytest.raises(cu.capi.Cup3dError):
        cu.makePoissonSolver(sim).solve()
