"""The host-callback path of aar::SparseLevMarq ("a host-side class for API compatibility"; automatic-ar_amd/host/host_levmarq.cpp)
against the REAL ucoslam::SparseLevMarq<double> of libs/sparselevmarq.h compiled into oracle/_ref/libref_lm.so, on the toy problems of
tests/tools/toy_problems.h: a caller's own residual / Jacobian functions, solve(z, f, J), solve(z, f) with the solver's central differences, a stop function,
and init + step by step.  Per-step trace (error, damping, accepted) and final vector must agree to 1e-9 -- rejected tries (mu grows inside a step), the
exits of libs/sparselevmarq.h:458-461 and the stop-function quirk (prevErr stays the initial error) included.  CPU only: no kernel runs."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "automatic-ar_amd")


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("hostlm") / "host_lm_main")
    cc = subprocess.run(["g++", "-O1", "-std=c++17", "-Wall", os.path.join(ROOT, "tests", "tools", "host_lm_main.cpp"), "-o", exe, "-L" + PKG, "-laar",
                         "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    return exe


def ref_toy(problem, mode, max_iters, min_error, min_step, min_avg, tau, der_eps, stop_after, steps):
    if not os.path.exists(ol.REF_SO):
        pytest.skip("oracle/_ref/libref_lm.so not built (needs /root/reference)")
    L = ol.ref()
    L.ref_lm_toy.restype = C.c_double
    cap = 256
    err, mu, acc = np.zeros(cap), np.zeros(cap), np.zeros(cap, dtype=np.int32)
    n = C.c_int32()
    z = np.zeros(6)
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    fe = L.ref_lm_toy(C.c_int(problem), C.c_int(mode), C.c_int(max_iters), C.c_double(min_error), C.c_double(min_step), C.c_double(min_avg), C.c_double(tau),
                      C.c_double(der_eps), C.c_int(stop_after), C.c_int(steps), err.ctypes.data_as(dp), mu.ctypes.data_as(dp), acc.ctypes.data_as(ip), C.c_int32(cap),
                      C.byref(n), z.ctypes.data_as(dp))
    return dict(err=err[:n.value], mu=mu[:n.value], acc=acc[:n.value], final_err=fe, z=z)


def host_toy(tool, *args):
    out = subprocess.run([tool] + [repr(a) if isinstance(a, float) else str(a) for a in args], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    err, mu, acc, z, fin = [], [], [], [], None
    for line in out.stdout.splitlines():
        t = line.split()
        if t[0] == "step":
            err.append(float(t[3])); mu.append(float(t[5])); acc.append(int(t[7]))
        elif t[0] == "z":
            z.append(float(t[2]))
        elif t[0] == "final_err":
            fin = dict(final_err=float(t[1]), exit_code=int(t[3]), iterations=int(t[5]))
    return dict(err=np.array(err), mu=np.array(mu), acc=np.array(acc), z=np.array(z), **fin)


def close(a, b, tol=1e-9):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return a.shape == b.shape and np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))


CASES = [
    # problem, mode, max_iters, min_error, min_step, min_avg, tau, der_eps, stop_after, steps
    ("rosenbrock solve(z, f, J), tau 1e-3: rejected tries, exit on minError", (0, 0, 100, 1e-10, 0.0, 1e-12, 1e-3, 1e-3, 0, 0)),
    ("rosenbrock solve(z, f, J), tau 1: exit on the average step", (0, 0, 100, 1e-10, 0.0, 1e-3, 1.0, 1e-3, 0, 0)),
    ("rosenbrock solve(z, f): the solver's central differences", (0, 1, 100, 1e-10, 0.0, 1e-12, 1e-3, 1e-4, 0, 0)),
    ("rosenbrock, maxIters binds", (0, 0, 5, 1e-10, 0.0, 1e-12, 1e-3, 1e-3, 0, 0)),
    ("pose fit solve(z, f): numeric Jacobian, sparsified at 1e-4", (1, 1, 100, 1e-12, 0.0, 1e-10, 1.0, 1e-4, 0, 0)),
    ("pose fit, tau 1e-6 (far too little damping at the start)", (1, 1, 100, 1e-12, 0.0, 1e-10, 1e-6, 1e-4, 0, 0)),
    ("rosenbrock under a stop function (prevErr stays the initial error)", (0, 2, 100, 1e-10, 0.0, 1e-12, 1e-3, 1e-3, 14, 0)),
    ("pose fit under a stop function", (1, 2, 100, 1e-12, 0.0, 1e-10, 1.0, 1e-4, 9, 0)),
    ("rosenbrock step by step: init + step(f, J)", (0, 3, 100, 1e-10, 0.0, 1e-12, 1e-3, 1e-3, 0, 12)),
    ("pose fit step by step: init + step(f)", (1, 3, 100, 1e-12, 0.0, 1e-10, 1e-2, 1e-4, 0, 8)),
]


@pytest.mark.parametrize("name,args", CASES, ids=[c[0] for c in CASES])
def test_host_loop_equals_the_reference_solver(tool, name, args):
    r = ref_toy(*args)
    h = host_toy(tool, *args)
    assert len(h["err"]) == len(r["err"]) and len(r["err"]) > 0, (len(h["err"]), len(r["err"]))
    assert close(h["err"], r["err"]), np.abs(h["err"] - r["err"]).max()
    assert close(h["mu"], r["mu"]), (h["mu"], r["mu"])
    if args[1] == 3:
        assert np.array_equal(h["acc"], r["acc"])
    assert close(h["final_err"], r["final_err"]) and close(h["z"], r["z"], 1e-8)


def test_the_cases_cover_a_rejected_try_and_every_exit(tool):
    # a step whose first try is rejected multiplies mu by v before the accepted try divides it again: mu_after > 0.33^... is not a test; look for a step
    # after which the damping is LARGER than before although the step was accepted
    h = host_toy(tool, *CASES[0][1])
    assert np.any(np.diff(h["mu"]) > 0) and np.all(np.diff(h["err"]) <= 0)
    exits = {host_toy(tool, *c[1])["exit_code"] for c in CASES[:6]}
    assert {1, 2, 0} <= exits, exits


def test_mixing_a_mapper_function_with_a_host_function_is_refused():
    # (compile-time shape only: the refusal is a std::logic_error thrown by SparseLevMarq::identify_pair; exercised on the GPU box by solver_seam_main)
    src = open(os.path.join(PKG, "host", "multicam_mapper.h")).read()
    assert "or both host functions" in src and "no CPU loop for host callbacks" not in src
