"""The C++ shim (cup3d_amd/host/cup3d_hip_operators.h) compiles and links into the unmodified reference TU -- against the single-rank
MPI stub (the binary the GPU drop-in tests run) and against a REAL <mpi.h> (its multi-rank branch: RCCL bootstrap over MPI_Bcast, the
collective choice of transport, the gathered leaf list).  Needs the reference sources and the build container's MPICH: skipped elsewhere
(the GPU box only uses prebuilt files)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference/main.cpp"
MPI = "/opt/conda/lib/libmpi.so"


@pytest.mark.skipif(not (os.path.exists(REFERENCE) and os.path.exists(MPI) and os.path.exists(os.path.join(ROOT, "cup3d_amd", "libcup3d_hip.so"))),
                    reason="needs /root/reference, the container's MPICH and the built product library")
def test_shim_links_against_a_real_mpi(tmp_path):
    out = tmp_path / "ref_tool_hip_mpi"
    cmd = ["g++", "-O0", "-std=c++17", "-DCUBISM_ALIGNMENT=64", "-D_BS_=8", "-DDIMENSION=3", "-DNDEBUG", "-fopenmp", "-w", "-DCUP3D_WITH_HIP",
           "-I/opt/conda/include", "-I" + os.path.join(ROOT, "oracle", "refbuild"), f'-DCUP3D_REFERENCE_MAIN="{REFERENCE}"', "-o", str(out),
           os.path.join(ROOT, "oracle", "ref_harness.cpp"), MPI, "-L" + os.path.join(ROOT, "cup3d_amd"), "-lcup3d_hip", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    syms = subprocess.run(["nm", "-D", "--undefined-only", str(out)], stdout=subprocess.PIPE, check=True).stdout.decode()
    for s in ("MPI_Allgatherv", "MPI_Bcast", "cup3d_grid_rank_view", "cup3d_comm_init", "cup3d_comm_unique_id"):   # the multi-rank branch is in the binary
        assert s in syms, s
