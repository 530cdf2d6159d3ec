"""tests/fake_rccl/librccl_fake.so (TEST INFRASTRUCTURE: a stand-in for librccl at RCCL's own API, see fake_rccl.cpp) builds here and
exports every entry point cup3d_amd/csrc/comm.hip resolves with dlsym -- so that `CUP3D_RCCL_LIBRARY=<it>` can take librccl's place in
the GPU tests that run comm.hip's RCCL branch with several ranks on one device."""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_stand_in_exports_what_comm_hip_resolves():
    d = os.path.join(ROOT, "tests", "fake_rccl")
    subprocess.check_call(["make", "-s", "-C", d])
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(d, "librccl_fake.so")], stdout=subprocess.PIPE, check=True, text=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines()}
    wanted = set(re.findall(r'SYM\(\w+, "(nccl\w+)"\)', open(os.path.join(ROOT, "cup3d_amd", "csrc", "comm.hip")).read()))
    assert len(wanted) >= 10 and wanted <= exported, wanted - exported
