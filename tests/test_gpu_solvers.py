"""The solvers of the damped normal equations behind aar_solver_options (include/aar.h) -- DIRECT (the reference's step to rounding,
libs/sparselevmarq.h:394-400), SPCG (CG on the explicit reduced system, csrc/spcg_kernels.hip), PCG (CG through the frame blocks,
csrc/pcg_kernels.hip), AUTO -- against each other, the CPU oracle and the compiled reference.  Needs a real MI355X.

Bars: at a tight forcing term an inexact solver's damped step IS the direct step (<= 1e-8 relative); at its default forcing term the LM run
ends within 1e-4 px of the reference-faithful CPU run (north star) and within 1e-5 px of the direct path, in as many LM steps (+- 1).
"""
import os
import threading

import numpy as np
import pytest

import aar
import oracle_lib as ol
from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if aar.device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the GPU box (the product has no CPU path)")


def _rel(a, b):
    return np.abs(a - b).max() / np.abs(b).max()


@pytest.mark.parametrize("name,kw", [("g1_cfg2", {}), ("g1_cfg3_cut", {}), ("g1_cfg3_cut", {"optimize": (False, True, True)}), ("g1_cfg3_cut", {"optimize": (True, False, True)}),
                                     ("g1_cfg2_intr", {"intrinsics": True}), ("g2_small", {})])
def test_spcg_damped_step_is_the_direct_step_at_a_tight_forcing_term(name, kw):
    # one to four tiles, gauge rows, switched-off groups, the intrinsics entities: the CG on the explicit reduced system converged to 1e-12 gives the
    # step of the dense LDL^T chain and of the oracle's sparse LDL^T
    ds, g = load_golden(name)
    o = ol.Oracle(ds, **({"optimize": kw["optimize"]} if "optimize" in kw else {}), **({"intrinsics": True} if kw.get("intrinsics") else {}))
    with aar.Problem(ds, **kw) as pd, aar.Problem(ds, solver="spcg", pcg_eta=1e-12, **kw) as ps:
        x0 = pd.x_with_intrinsics(ds.x_full) if kw.get("intrinsics") else ds.x_full
        assert ps.solver_stats()["solver"] == "spcg" and pd.solver_stats()["solver"] == "direct"
        for mu in (1e2, 1e5, 1e8):
            dd, dsp = pd.eval_damped_step(x0, mu), ps.eval_damped_step(x0, mu)
            st = ps.solver_stats()
            # (1e-9 .. 1e-12 usually; 2.3e-7 seen once in ~10 runs at mu = 1e2 with 62 iterations: a stopping rule on r^T M^-1 r does not bound the error below
            #  cond x 1e-12, and the order of the atomics behind S moves the iteration count)
            assert _rel(dsp, dd) < 1e-6, (mu, st)
            if not kw.get("intrinsics"):
                do = o.damped_solve(ds.x_full, mu, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
                assert _rel(dsp, do) < 1e-7, mu
        st = ps.solver_stats()
        assert st["solves"] >= 3 and st["total_iterations"] > 0


def test_auto_at_full_size_config3_against_compiled_reference_and_direct_path():
    # BASELINE.json's metric configuration through solver = AUTO (which is SPCG there): the bar of test_full_size_config3_against_compiled_reference
    # (final RMSE within 1e-4 px of the reference-faithful CPU run by the REAL SparseLevMarq), within 1e-5 px of the direct path, the same number of LM
    # steps, no try redone by the direct chain
    ds = aar.synth(3)
    o = ol.Oracle(ds)
    solve = o.ref_lm_solve if ol.have_ref() else o.lm_solve
    x_ref, rep_ref = solve(ds.x_full, jac_mode=ol.JAC_NUMERIC_F32, res_mode=ol.RES_F32, threads=min(32, os.cpu_count() or 1))
    with aar.Problem(ds, solver="direct") as p:
        x_d, rep_d = p.lm_solve(ds.x_full)
        rmse_d, _ = p.reproj_stats(x_d)
    with aar.Problem(ds, solver="auto") as p:
        assert p.solver_stats()["solver"] == "spcg"
        x, rep = p.lm_solve(ds.x_full)
        rmse, _ = p.reproj_stats(x)
        st = p.solver_stats()
    ref = o.reproj_stats(x_ref)["rmse"]
    assert abs(rmse - ref) < 1e-4, (rmse, ref)
    assert abs(rmse - rmse_d) < 1e-5, (rmse, rmse_d)
    assert rep["iterations"] == rep_d["iterations"] and abs(rep["iterations"] - rep_ref["iterations"]) <= 1
    assert all(t["tries"] == 1 and t["accepted"] == 1 for t in rep["trace"])
    assert st["fallbacks"] == 0 and 0 < st["total_iterations"] <= 64 * st["solves"]
    np.testing.assert_allclose(rep["final_err"], rep_ref["final_err"], rtol=1e-4)


@pytest.mark.parametrize("name,huber", [("g1_cfg2_retry", False), ("g1_cfg2_far", False), ("g1_cfg2_huber_retry", True), ("g1_cfg3_cut", False)])
def test_spcg_through_the_retry_far_start_and_huber_fixtures(name, huber):
    # the fixtures whose rejected tries are where an inexact step could change the branch of libs/sparselevmarq.h:406-419: far starts, tau = 1e-6,
    # -with-huber with a rejected try.  The inexact run still takes its rejected tries and ends within 1e-4 px of the reference-faithful run
    ds, g = load_golden(name)
    prm = aar.lm_default_params(tau=float(g["tau"][0])) if "tau" in g else None
    with aar.Problem(ds, with_huber=huber) as p:
        x_d, rep_d = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
        rmse_d, _ = p.reproj_stats(x_d)
    with aar.Problem(ds, with_huber=huber, solver="spcg") as p:
        x, rep = p.lm_solve(ds.x_full, params=prm, trace_cap=600)
        rmse, _ = p.reproj_stats(x)
        st = p.solver_stats()
    assert abs(rmse - g["faithful_rmse"][0]) < 1e-4, (rmse, g["faithful_rmse"][0])
    assert abs(rmse - rmse_d) < 1e-4
    if "retry" in name:
        assert max(t["tries"] for t in rep["trace"]) > 1 and max(t["tries"] for t in rep_d["trace"]) > 1
    assert rep["trial_points"] == sum(t["tries"] for t in rep["trace"])
    # (from a far start with tau = 1e-6 the number of steps is not a stable quantity: the direct path itself is given +- 2 against the real solver there)
    assert abs(rep["iterations"] - rep_d["iterations"]) <= max(5 if "retry" in name else 2, rep_d["iterations"] // 50)
    assert 0 < st["solves"] <= rep["trial_points"] + st["fallbacks"]       # (after a fall-back the direct chain keeps the next tries)


def test_spcg_iteration_cap_falls_back_to_the_direct_chain():
    # a cap of one iteration at a forcing term nobody reaches in one: the first try of every solve raises device flag 8 and is redone by the direct chain,
    # which then keeps the next 8 tries (the damping only falls along accepted steps: the systems get harder), 16 after the next fall-back ... -- the run
    # is the direct run (to the rounding of rebuilt blocks), and says how often it fell back
    ds, g = load_golden("g1_cfg3_cut")
    with aar.Problem(ds) as p:
        x_d, rep_d = p.lm_solve(ds.x_full)
        d_d = p.eval_damped_step(ds.x_full, 1e4)
    with aar.Problem(ds, solver="spcg", pcg_eta=1e-12, pcg_max_it=1) as p:
        x, rep = p.lm_solve(ds.x_full)
        st = p.solver_stats()
        assert st["fallbacks"] == 2 and st["solves"] == 2 and rep["trial_points"] == sum(t["tries"] for t in rep["trace"]) == 15      # tries 1 and 10
        x2, rep2 = p.lm_solve(ds.x_full)
        assert p.solver_stats()["fallbacks"] == 4 and p.solver_stats()["solves"] == 4          # every solve starts with its own solver again
        d = p.eval_damped_step(ds.x_full, 1e4)
        assert p.solver_stats()["fallbacks"] == 5
    assert rep["iterations"] == rep_d["iterations"]
    np.testing.assert_allclose([t["err"] for t in rep["trace"]], [t["err"] for t in rep_d["trace"]], rtol=1e-7)      # (the bar of the direct path's own traces: fp64 atomics)
    np.testing.assert_allclose(x, x_d, atol=1e-7)
    assert _rel(d, d_d) < 1e-11


def test_spcg_handover_timeout_falls_back_to_the_direct_chain(monkeypatch):
    # a wavefront of the CG grid that never shows up (test hook AAR_SPCG_TEST_DROP: what a device shared with another process can do): the others
    # give up after ~1 s, raise device flag 4, and the try is redone by the direct chain -- no error, no hang, the same step
    ds, g = load_golden("g1_cfg3_cut")
    with aar.Problem(ds) as p:
        d_d = p.eval_damped_step(ds.x_full, 1e4)
    monkeypatch.setenv("AAR_SPCG_TEST_DROP", "5")
    with aar.Problem(ds, solver="spcg") as p:
        d = p.eval_damped_step(ds.x_full, 1e4)
        assert p.solver_stats()["fallbacks"] == 1
    monkeypatch.delenv("AAR_SPCG_TEST_DROP")
    assert _rel(d, d_d) < 1e-11
    with aar.Problem(ds, solver="spcg", pcg_eta=1e-9) as p:       # (and the next problem's hand-over buffers are clean)
        d8 = p.eval_damped_step(ds.x_full, 1e8)
        assert p.solver_stats()["fallbacks"] == 0 and 0 < p.solver_stats()["last_iterations"] < 64
    with aar.Problem(ds) as p:
        assert _rel(d8, p.eval_damped_step(ds.x_full, 1e8)) < 1e-6


def test_problems_with_different_solvers_live_side_by_side_in_one_process():
    # the solver is a property of the problem (aar_solver_options), not of the process: three problems of one process, stepped in turn
    ds, g = load_golden("g1_cfg3_cut")
    alone = {}
    for s in ("direct", "spcg", "pcg"):
        with aar.Problem(ds, solver=s) as p:
            alone[s] = p.lm_solve(ds.x_full)
    ps = {s: aar.Problem(ds, solver=s) for s in ("direct", "spcg", "pcg")}
    try:
        assert [ps[s].solver_stats()["solver"] for s in ("direct", "spcg", "pcg")] == ["direct", "spcg", "pcg"]
        for s in ps:
            ps[s].lm_init(ds.x_full)
        errs = {s: [] for s in ps}
        for k in range(max(r["iterations"] for _, r in alone.values())):
            for s in ps:
                if k < alone[s][1]["iterations"]:
                    errs[s].append(ps[s].lm_step()["err"])
        for s in ps:
            np.testing.assert_allclose(errs[s], [t["err"] for t in alone[s][1]["trace"]], rtol=1e-6 if s == "direct" else 1e-4, err_msg=s)
        assert ps["direct"].solver_stats()["total_iterations"] == 0 and ps["spcg"].solver_stats()["total_iterations"] > 0 and ps["pcg"].solver_stats()["total_iterations"] > 0
    finally:
        for p in ps.values():
            p.close()


@pytest.mark.parametrize("cfg", [3, 4, 5])
def test_inexact_solvers_at_full_size_against_the_direct_path(cfg):
    # configs 3, 4, 5 at FULL size: PCG through the frame blocks (what AUTO picks at config 5) and, where the reduced system fits the wavefronts'
    # registers, CG on the explicit system (what AUTO picks at configs 3 and 4): final RMSE within 1e-5 px of the direct path, LM steps <= direct + 1
    ds = aar.synth(cfg)
    with aar.Problem(ds) as p:
        x_d, rep_d = p.lm_solve(ds.x_full)
        rmse_d, _ = p.reproj_stats(x_d)
    with aar.Problem(ds, solver="auto") as p:
        auto = p.solver_stats()["solver"]
    assert auto == ("pcg" if cfg == 5 else "spcg")
    for s in ("pcg", "spcg"):
        with aar.Problem(ds, solver=s) as p:
            x, rep = p.lm_solve(ds.x_full)
            rmse, _ = p.reproj_stats(x)
            st = p.solver_stats()
        assert abs(rmse - rmse_d) < 1e-5, (cfg, s, rmse, rmse_d)
        assert rep["iterations"] <= rep_d["iterations"] + 1, (cfg, s)
        assert st["total_iterations"] > 0 and (s == "pcg" or st["fallbacks"] <= 1)


def _run_ranks(world, fn):
    grp = aar.LocalGroup(world)
    out, errs = [None] * world, []
    def run(rank):
        try:
            comm = aar.Comm.local(grp, rank)
            try:
                out[rank] = fn(comm, rank)
            finally:
                comm.close()
        except Exception as e:      # noqa: BLE001
            errs.append((rank, e))
    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    grp.close()
    assert not errs, errs
    return out


def test_config5_shaped_problem_against_the_oracle():
    # config 5's entity count (16 cameras / 200 markers: 14 tiles of LDL^T, MFMA Schur kernel, look-ahead, pass A in 128-thread workgroups) on a cut of
    # 300 frames the oracle can factor (75 612 marker observations): residual rows bit for bit, the damped step at two dampings against the oracle's sparse
    # LDL^T, the LM run against the oracle's (analytic Jacobian) -- direct, PCG and CG on the explicit system, on one rank and on four
    ds = aar.synth(5, num_frames=300)
    o = ol.Oracle(ds)
    do = {mu: o.damped_solve(ds.x_full, mu, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32) for mu in (1e3, 1e7)}
    xo, repo = o.lm_solve(ds.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F32)
    rmse_o = o.reproj_stats(xo)["rmse"]
    with aar.Problem(ds) as p:
        r, ss = p.eval_residuals(ds.x_full)
        assert np.array_equal(r, o.residuals(ds.x_full, res_mode=ol.RES_F32))
        for mu, d_ref in do.items():
            assert _rel(p.eval_damped_step(ds.x_full, mu), d_ref) < 1e-8, mu
        x, rep = p.lm_solve(ds.x_full)
        rmse, _ = p.reproj_stats(x)
    assert rep["iterations"] == repo["iterations"]
    np.testing.assert_allclose([t["err"] for t in rep["trace"]], [t["err"] for t in repo["trace"]], rtol=1e-6)
    assert abs(rmse - rmse_o) < 1e-7
    for s, eta in (("pcg", 1e-11), ("spcg", 1e-12)):
        with aar.Problem(ds, solver=s, pcg_eta=eta, pcg_max_it=2000) as p:
            for mu, d_ref in do.items():
                assert _rel(p.eval_damped_step(ds.x_full, mu), d_ref) < 1e-7, (s, mu)
    for s in ("pcg", "spcg"):
        with aar.Problem(ds, solver=s) as p:
            x, rep = p.lm_solve(ds.x_full)
            assert abs(p.reproj_stats(x)[0] - rmse_o) < 3e-5, s        # (north star: 1e-4; seen: PCG 1e-6, SPCG 1.1e-5)
            assert abs(rep["iterations"] - repo["iterations"]) <= 1, s
    for s in ("direct", "pcg", "spcg"):
        def solve(comm, rank, s=s):
            with aar.Problem(ds, comm=comm, solver=s) as q:
                xs, reps = q.lm_solve(ds.x_full)
                return q.reproj_stats(xs)[0], reps
        for rmse_r, reps in _run_ranks(4, solve):
            assert abs(rmse_r - rmse_o) < (1e-7 if s == "direct" else 3e-5), (s, rmse_r, rmse_o)
            assert abs(reps["iterations"] - repo["iterations"]) <= 1, s


def test_init_head_start_leaves_the_trajectory_alone(monkeypatch):
    # aar_lm_init queues the first step's frame inverses and Schur complement before the host has read mu_0 (AAR_INIT_HEADSTART, a per-problem
    # tuning switch): with fixed-order sums the whole LM run is the same bit for bit with and without it -- also when the step then wants another
    # damping than the one the head start assumed (tau changed between init and step: the complement is taken back and redone)
    ds, g = load_golden("g1_cfg3_cut")
    runs = {}
    for hs in ("1", "0"):
        monkeypatch.setenv("AAR_INIT_HEADSTART", hs)
        with aar.Problem(ds, deterministic=True) as p:
            x, rep = p.lm_solve(ds.x_full)
            p.lm_init(ds.x_full, params=aar.lm_default_params(tau=1.0))
            p.lm_init(ds.x_full, params=aar.lm_default_params(tau=1e-3))     # a second init: the first one's head start is discarded
            steps = [p.lm_step() for _ in range(4)]
            runs[hs] = (x, [t["err"] for t in rep["trace"]], [t["mu"] for t in rep["trace"]], [s["err"] for s in steps], [s["mu"] for s in steps])
    monkeypatch.delenv("AAR_INIT_HEADSTART")
    for a, b in zip(runs["1"], runs["0"]):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    np.testing.assert_allclose(runs["1"][1], g["analytic_err"], rtol=1e-7)


def test_cpp_mirror_and_driver_take_the_solver_as_an_option(tmp_path):
    # aar::MultiCamMapper::set_solver_options (beside SparseLevMarq::Params, libs/sparselevmarq.h:30-50) through the driver: aar_find_solution -solver spcg
    # on a 4-camera / 12-marker recording written in the reference's file formats ends where the default (direct) run ends, and says what its solver did
    import subprocess
    from conftest import PKG
    exe = os.path.join(PKG, "aar_find_solution")
    runs = {}
    for s in ("direct", "spcg", "auto"):
        folder = str(tmp_path / s)
        assert subprocess.run([exe, "--synth", "2", folder], capture_output=True, text=True).returncode == 0
        run = subprocess.run([exe, folder, "0.05", "x", "-from-initial"] + ([] if s == "direct" else ["-solver", s]), capture_output=True, text=True)
        assert run.returncode == 0, run.stderr + run.stdout
        fin = aar.solution_read(os.path.join(folder, "final.solution"))
        runs[s] = (ol.Oracle(fin).reproj_stats(fin.x_full)["rmse"], run.stdout)
    assert "solver: spcg" in runs["spcg"][1] and "CG iterations" in runs["spcg"][1] and "solver:" not in runs["direct"][1]
    assert "solver: direct" in runs["auto"][1]                      # (one tile of unknowns: AUTO keeps the direct chain)
    assert abs(runs["spcg"][0] - runs["direct"][0]) < 1e-5 and abs(runs["auto"][0] - runs["direct"][0]) < 1e-9
    bad = subprocess.run([exe, str(tmp_path / "direct"), "0.05", "x", "-solver", "nonsense"], capture_output=True, text=True)
    assert bad.returncode != 0


def test_spcg_in_deterministic_mode_gives_the_same_bits_twice():
    # k_spcg has no atomics (every wavefront adds the same shares in the same order); with fixed-order sums in the passes and the Schur kernel the whole
    # INEXACT run -- CG iteration counts included -- is bit-reproducible, on one rank and on two
    ds, g = load_golden("g1_cfg3_cut")
    def run():
        with aar.Problem(ds, solver="spcg", deterministic=True) as p:
            x, rep = p.lm_solve(ds.x_full)
            return x, [t["err"] for t in rep["trace"]], [t["mu"] for t in rep["trace"]], p.solver_stats()["total_iterations"]
    a, b = run(), run()
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and a[2] == b[2] and a[3] == b[3] > 0
    def solve(comm, rank):
        with aar.Problem(ds, comm=comm, solver="spcg", deterministic=True) as q:
            xs, reps = q.lm_solve(ds.x_full)
            return xs, [t["err"] for t in reps["trace"]]
    r1, r2 = _run_ranks(2, solve), _run_ranks(2, solve)
    for (xa, ea), (xb, eb) in zip(r1, r2):
        assert np.array_equal(xa, xb) and ea == eb
    assert np.array_equal(r1[0][0], r1[1][0])
    np.testing.assert_allclose(r1[0][1], a[1], rtol=1e-6)


def test_solver_options_struct_is_forward_compatible_and_validated():
    # aar_solver_options carries its own size: a caller built against a shorter struct (only struct_size + solver) keeps working, the fields it does not know
    # take their defaults; an unset struct_size and values outside the enums are refused with AAR_ERR_INVALID (no silent fall-back to another solver)
    import ctypes as C
    ds, g = load_golden("g1_cfg3_cut")
    cds = ds.as_c()
    d = aar.CProblemDesc()
    aar.lib().aar_problem_desc_from_dataset(C.byref(cds), C.byref(d))
    so = aar.CSolverOptions()
    aar.lib().aar_solver_default_options(C.byref(so))
    assert so.struct_size == C.sizeof(aar.CSolverOptions) and so.solver == aar.SOLVER_DIRECT and so.deterministic == 0
    h = C.c_void_p()
    so.solver, so.struct_size, so.deterministic, so.pcg_eta = aar.SOLVER_SPCG, 8, 1, 0.5      # a "short" caller: deterministic / eta lie beyond its struct
    assert aar.lib().aar_problem_create_ex(C.byref(d), C.byref(so), C.byref(h)) == 0
    st = aar.CSolverStats()
    assert aar.lib().aar_problem_get_solver_stats(h, C.byref(st)) == 0
    assert st.solver == aar.SOLVER_SPCG and st.deterministic == 0 and abs(st.pcg_eta - 0.02) < 1e-15
    aar.lib().aar_problem_destroy(h)
    for bad in (dict(struct_size=0), dict(solver=7), dict(pcg_eta=-1.0), dict(pcg_max_it=-3)):
        aar.lib().aar_solver_default_options(C.byref(so))
        for k, v in bad.items():
            setattr(so, k, v)
        h = C.c_void_p()
        assert aar.lib().aar_problem_create_ex(C.byref(d), C.byref(so), C.byref(h)) == aar.AAR_ERR_INVALID, bad
    # a solver the problem is too large for is refused, not replaced: CG on the explicit system beyond 14 tiles of unknowns
    big = aar.synth(3, num_cams=4, num_markers=300, num_frames=12)
    with pytest.raises(aar.AarError) as e:
        aar.Problem(big, solver="spcg")
    assert e.value.code == aar.AAR_ERR_UNSUPPORTED
    with aar.Problem(big, solver="auto") as p:
        assert p.solver_stats()["solver"] == "pcg"


def test_auto_pcg_forcing_sequence():
    # AUTO resolving to the frame-block PCG (>= 96 shared entities) with the forcing term left at its default solves the early LM steps to 0.3 and the
    # steps near the stopping rule to 0.1 (include/aar.h, profiles/r04_pcg_eta_sweep.txt): same LM step count as the direct solver (+- 1), final RMSE
    # within 1e-5 px of it, fewer CG iterations than the fixed 0.1; an explicit solver or an explicit forcing term switches the sequence off
    ds = aar.synth(5, num_frames=120)
    with aar.Problem(ds, solver="direct") as p:
        xd, rd = p.lm_solve(ds.x_full)
        rmse_d = p.reproj_stats(xd)[0]
    runs = {}
    for name, kw in (("auto", dict(solver="auto")), ("pcg", dict(solver="pcg")), ("auto_eta", dict(solver="auto", pcg_eta=0.1))):
        with aar.Problem(ds, **kw) as p:
            st = p.solver_stats()
            assert st["solver"] == "pcg" and st["pcg_eta"] == 0.1
            assert st["pcg_eta_loose"] == (0.3 if name == "auto" else 0.0)
            x, rep = p.lm_solve(ds.x_full)
            runs[name] = (p.reproj_stats(x)[0], rep["iterations"], p.solver_stats()["total_iterations"])
    for name, (rmse, its, cg) in runs.items():
        assert abs(rmse - rmse_d) < 1e-5, (name, rmse, rmse_d)
        assert abs(its - rd["iterations"]) <= 1, (name, its, rd["iterations"])
    assert runs["auto"][2] < runs["pcg"][2]
    assert runs["auto_eta"][2] == runs["pcg"][2]


def test_auto_spcg_forcing_sequence():
    # the same for AUTO resolving to the CG on the explicit reduced system (two tiles and more, below 96 entities): 0.1 early, the solver's 0.02 near the
    # stopping rule (measured in the preconditioner's norm); full-size config 3: the direct solver's 15 LM steps, final RMSE within 1e-5 px of it
    ds = aar.synth(3)
    with aar.Problem(ds, solver="direct") as p:
        xd, rd = p.lm_solve(ds.x_full)
        rmse_d = p.reproj_stats(xd)[0]
    out = {}
    for name, kw in (("auto", dict(solver="auto")), ("spcg", dict(solver="spcg"))):
        with aar.Problem(ds, **kw) as p:
            st = p.solver_stats()
            assert st["solver"] == "spcg" and abs(st["pcg_eta"] - 0.02) < 1e-15 and st["pcg_eta_loose"] == (0.1 if name == "auto" else 0.0)
            x, rep = p.lm_solve(ds.x_full)
            st = p.solver_stats()
            out[name] = (p.reproj_stats(x)[0], rep["iterations"], st["total_iterations"], st["fallbacks"])
    for name, (rmse, its, cg, fb) in out.items():
        assert abs(rmse - rmse_d) < 1e-5 and its == rd["iterations"] and fb == 0, (name, rmse, rmse_d, its)
    assert out["auto"][2] < out["spcg"][2]
