"""cup3d_amd — MI355X-native (HIP, gfx950, RCCL) implementation of CUP3D's per-block
stencil + pressure-Poisson hot path behind the reference's Operator /
PoissonSolverBase surface.  See DESIGN.md, INTEGRATION.md and include/cup3d_hip.h."""
from .capi import Cup3dError, PoissonParams, PoissonResult, device_count, device_init, lib  # noqa: F401
from .operators import (AdvectionDiffusion, AdvectionDiffusionImplicit, DiffusionSolver, RankView, ComputeLHS, ComputeVorticity, ExternalForcing, GradChiOnTmp, Grid, MeshAdaptation, ObstacleData, Operator, Penalization, PoissonSolverBase,  # noqa: F401
                        PoissonSolverHIP, PressureProjection, Simulation, SimulationData, findMaxU,
                        makePoissonSolver)
