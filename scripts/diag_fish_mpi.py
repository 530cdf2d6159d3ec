"""Diagnostic (GPU box): where does the velocity difference between the CPU reference and the shim run come from -- ranks or levels?"""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as O
import test_gpu_00_dropin_mpi as T
REF_HIP = os.path.join(O.ORACLE_DIR, "_ref", "ref_tool_hip")

def run(tool, n, pre, args, script, env=None):
    wd = tempfile.mkdtemp()
    open(os.path.join(wd, "script.txt"), "w").write("\n".join(pre + script) + "\n")
    cmd = (T.launcher() + ["-n", str(n)] if n > 1 else []) + [tool, "script.txt", "--"] + args
    out = subprocess.run(cmd, cwd=wd, env=dict(T.ENV, **(env or {})), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return wd

def fields(wd, n, tag):
    res = []
    for r in range(n):
        suf = f".r{r}" if n > 1 else ""
        t = O.read_tables(os.path.join(wd, f"t{tag}.bin{suf}"))[0]
        res.append((t, O.read_blocks(os.path.join(wd, f"v{tag}.bin{suf}"), len(t), 3), O.read_blocks(os.path.join(wd, f"p{tag}.bin{suf}"), len(t), 1)))
    return res

steps = [1, 2, 3, 5, 8, 12, 20, 30]
script, prev = [], 0
for s in steps:
    script += [f"op steps {s - prev}", f"tables t{s}.bin", f"dump vel v{s}.bin", f"dump pres p{s}.bin"]
    prev = s
for n, lm in ((1, 3), (1, 4), (2, 3), (2, 4)):
    args = T.COMMON + ["-levelMax", str(lm), "-factory-content", T.ONE_FISH]
    cpu = run(T.REF_MPI if n > 1 else O.REF_TOOL, n, [], args, script)
    hip = run(T.REF_HIP_MPI if n > 1 else REF_HIP, n, ["hip on"], args, script, {"CUP3D_HIP_HOST_TRANSPORT": "1"})
    line = []
    for s in steps:
        c, h = fields(cpu, n, s), fields(hip, n, s)
        same = all(np.array_equal(a[0], b[0]) for a, b in zip(c, h))
        dv = max(np.abs(a[1] - b[1]).max() for a, b in zip(c, h)) if same else float("nan")
        vm = max(np.abs(a[1]).max() for a in c)
        dp = max(np.abs(a[2] - b[2]).max() for a, b in zip(c, h)) if same else float("nan")
        pm = max(np.abs(a[2]).max() for a in c)
        line.append(f"{s}:{'=' if same else 'X'} dv/v {dv / vm:.1e} dp/p {dp / pm:.1e}")
    print(f"ranks {n} levelMax {lm}: " + " | ".join(line), flush=True)
