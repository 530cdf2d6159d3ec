"""The scalar recurrences of the pipelined BiCGSTAB (PoissonSolverAMR::solve, main.cpp:14493, 14558-14601) as the product evaluates them:
ONE pair of functions (ctl_step1 / ctl_step2, poisson.hip) compiled for host and device -- on the device they run in the kernel that
totals the dot products, so the host never sits on the solver's critical path.  Here their HOST compilation (cup3d_debug_ctl_step, no
GPU needed) is compared with a line-by-line restatement of the reference's statements in IEEE double arithmetic (numpy float64 scalars:
every operation rounded once, like the reference's x86-64 baseline build), bit for bit, over random and adversarial inputs: both branches
of the alpha / alpha-tilde choice, serious breakdowns with and without restarts left, the x_opt bookkeeping, both stopping tests.  The device
compilation of the same functions is what every GPU solver test exercises (iteration and restart counts against the oracle)."""
import numpy as np

import cup3d_amd as cu

f64 = np.float64
EPS = f64(1e-100)


def reference_step1(st, t):
    st = dict(st)
    st["omega"] = f64(t[0]) / (f64(t[1]) + EPS)                                                   # 14493
    return st


def reference_step2(st, t):
    """14558-14601; the restart's own launches (14569-14590) are the host's business: state 2 asks for them"""
    st = dict(st)
    r0r, r0w, r0s, r0z, norm_1, norm_2 = (f64(v) for v in t[:6])
    norm = np.sqrt(f64(t[6]))
    alpha, omega = st["alpha"], st["omega"]
    beta = alpha / (omega + EPS) * r0r / (st["r0r_prev"] + EPS)                                   # 14558
    alpha = r0r / (r0w + beta * r0s - beta * omega * r0z)                                         # 14559
    alphat = f64(1.0) / (omega + EPS) + r0w / (r0r + EPS) - beta * omega * r0z / (r0r + EPS)      # 14560-14561
    alphat = f64(1.0) / (alphat + EPS)
    if np.abs(alphat) < f64(10) * np.abs(alpha):                                                  # 14563-14564
        alpha = alphat
    st.update(alpha=alpha, beta=beta, r0r_prev=r0r, norm=norm)
    xw = 1 - st["xcur"] if st["xopt"] == st["xcur"] else st["xcur"]                               # x_opt = x without a copy (solve())
    st["xcur"] = xw
    st["iter"] += 1
    state = 0
    if r0r * r0r < f64(1e-16) * norm_1 * norm_2 and st["restarts"] < st["max_restarts"]:          # 14566-14568
        st["restarts"] += 1
        state = 2
    if norm < st["min_norm"]:                                                                     # 14594-14600
        st["min_norm"] = norm
        st["xopt"] = st["xcur"]
    if norm < st["tol"] or norm / (st["init_norm"] + EPS) < st["tol_rel"]:                        # 14601
        state = 1
    st["state"] = state
    return st


KEYS = ["alpha", "beta", "omega", "r0r_prev", "norm", "init_norm", "min_norm", "tol", "tol_rel", "state", "restarts", "max_restarts", "xcur", "xopt", "iter"]


def product_step(step, st, t):
    io = np.zeros(16)
    for i, k in enumerate(KEYS):
        io[i] = st[k]
    tt = np.ascontiguousarray(np.asarray(t, dtype=np.float64))
    cu.capi.check(cu.lib().cup3d_debug_ctl_step(step, io, tt))
    out = dict(st)
    for i, k in enumerate(KEYS):
        out[k] = f64(io[i]) if i < 9 else int(io[i])
    return out


def same(a, b):
    for k in KEYS:
        va, vb = a[k], b[k]
        if isinstance(va, (int, np.integer)) or k in ("state", "restarts", "max_restarts", "xcur", "xopt", "iter"):
            if int(va) != int(vb):
                return False, k
        elif f64(va).tobytes() != f64(vb).tobytes() and not (np.isnan(va) and np.isnan(vb)):
            return False, k
    return True, None


def test_recurrences_are_the_references_bit_for_bit():
    rng = np.random.default_rng(2024)
    seen = {"alphat": 0, "alpha": 0, "restart": 0, "capped": 0, "done_abs": 0, "done_rel": 0, "xopt": 0, "inplace": 0, "other_buffer": 0}
    for trial in range(4000):
        mag = 10.0 ** rng.uniform(-8, 4)
        st = dict(alpha=f64(rng.normal() * mag), beta=f64(rng.normal()), omega=f64(rng.normal() * 10.0 ** rng.uniform(-3, 1)),
                  r0r_prev=f64(rng.normal() * mag), norm=f64(0), init_norm=f64(10.0 ** rng.uniform(-6, 2)), min_norm=f64(10.0 ** rng.uniform(-7, 2)),
                  tol=f64(1e-6), tol_rel=f64(1e-4), state=0, restarts=int(rng.integers(0, 4)), max_restarts=int(rng.choice([2, 100])),
                  xcur=int(rng.integers(0, 2)), xopt=int(rng.integers(-1, 2)), iter=int(rng.integers(0, 300)))
        if trial % 7 == 0:
            st["omega"] = f64(0.0)          # right after a restart (14591-14592)
            st["beta"] = f64(0.0)
        t1 = rng.normal(size=2) * mag
        t1[1] = abs(t1[1])
        a, b = product_step(1, st, t1), reference_step1(st, t1)
        ok, k = same(a, b)
        assert ok, (trial, "step1", k, a[k], b[k])
        t2 = rng.normal(size=7) * mag
        t2[4:] = np.abs(t2[4:])
        if trial % 5 == 0:
            t2[0] *= 1e-10                  # r0.r tiny against |r||r0|: a serious breakdown (14566)
        if trial % 11 == 0:
            t2[6] = 10.0 ** rng.uniform(-16, -9)   # converged in the absolute or the relative sense
        got, ref = product_step(2, a, t2), reference_step2(b, t2)
        ok, k = same(got, ref)
        assert ok, (trial, "step2", k, got[k], ref[k])
        # bookkeeping of which branches the inputs reached
        seen["restart"] += ref["state"] == 2
        seen["capped"] += (f64(t2[0]) ** 2 < f64(1e-16) * f64(t2[4]) * f64(t2[5])) and b["restarts"] >= b["max_restarts"]
        seen["done_abs"] += ref["state"] == 1 and ref["norm"] < 1e-6
        seen["done_rel"] += ref["state"] == 1 and ref["norm"] >= 1e-6
        seen["xopt"] += ref["xopt"] == ref["xcur"] and ref["min_norm"] == ref["norm"]
        seen["inplace"] += b["xopt"] != b["xcur"]
        seen["other_buffer"] += b["xopt"] == b["xcur"]
        with np.errstate(all="ignore"):
            beta = b["alpha"] / (b["omega"] + EPS) * f64(t2[0]) / (b["r0r_prev"] + EPS)
            al = f64(t2[0]) / (f64(t2[1]) + beta * f64(t2[2]) - beta * b["omega"] * f64(t2[3]))
        seen["alphat" if ref["alpha"].tobytes() != f64(al).tobytes() else "alpha"] += 1
    assert all(v > 20 for v in seen.values()), seen     # every branch of 14558-14601 was taken many times


def test_release_build_does_not_carry_the_test_entry_point():
    import ctypes as C
    import os
    rel = C.CDLL(os.path.join(os.path.dirname(cu.capi.LIB_PATH), "libcup3d_hip.so"))
    assert not hasattr(rel, "cup3d_debug_ctl_step")
