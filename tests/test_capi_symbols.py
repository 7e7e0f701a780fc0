"""The C-ABI library loads without a GPU and exports every symbol include/aar.h declares; compute entry points
refuse to run without a device (there is no CPU fallback).  CPU only."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import aar
from conftest import ROOT, load_golden


def header_functions():
    txt = open(os.path.join(ROOT, "include", "aar.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aar_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    names = header_functions()
    assert len(names) >= 30
    lib = C.CDLL(aar.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libaar.so does not export %s" % n
    assert sorted(aar.SYMBOLS) == names   # the Python binding covers the whole header


def test_boundary_has_no_torch_or_cxx_types():
    txt = open(os.path.join(ROOT, "include", "aar.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)   # signatures only, comments aside
    for bad in ("torch", "at::", "std::", "hipStream", "cv::", "Eigen"):
        assert bad not in txt
    assert 'extern "C"' in txt


def test_oracle_is_not_linked_into_the_product():
    # the product path must not route through the CPU checker
    import subprocess
    needed = subprocess.run(["objdump", "-p", aar.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in needed and "libref_lm" not in needed
    syms = subprocess.run(["nm", "-D", "--defined-only", aar.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in syms and "ref_lm_solve" not in syms
    for dirpath, _, files in os.walk(os.path.join(ROOT, "automatic-ar_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle_lib" not in src and "ba_oracle" not in src and "liboracle" not in src, f


@pytest.mark.skipif(aar.device_count() > 0, reason="this check is for machines without a GPU")
def test_compute_entry_points_fail_loudly_without_a_device():
    ds, _ = load_golden("g2_small")
    with pytest.raises(aar.AarError) as e:
        aar.Problem(ds)
    assert e.value.code == aar.AAR_ERR_NO_DEVICE
    assert "no CPU path" in str(e.value)


def test_invalid_problems_are_rejected_before_touching_the_device():
    ds, _ = load_golden("g2_small")
    bad = aar.Dataset.__new__(aar.Dataset)
    bad.__dict__.update(ds.__dict__)
    bad.obs_cam = ds.obs_cam.copy()
    bad.obs_cam[3] = ds.num_cams + 2
    with pytest.raises(aar.AarError) as e:
        aar.Problem(bad)
    assert e.value.code == aar.AAR_ERR_INVALID
    bad.obs_cam = ds.obs_cam
    bad.obs_frame = ds.obs_frame[::-1].copy()      # not in reference (frame-major) order
    with pytest.raises(aar.AarError) as e:
        aar.Problem(bad)
    assert e.value.code == aar.AAR_ERR_INVALID


def test_default_lm_params_are_the_mappers():
    p = aar.lm_default_params()   # libs/multicam_mapper.cpp:326-330 over libs/sparselevmarq.h:41-49
    assert (p.max_iters, p.min_error, p.min_step_error_diff, p.min_average_step_error_diff, p.tau) == (10000, 1e-5, 0.0, 1e-4, 1.0)


def test_default_solver_options_are_auto_with_one_forcing_term():
    # what a NULL options pointer means (include/aar.h): solver AUTO, not deterministic, every forcing-term field at "default"
    so = aar.CSolverOptions()
    aar.lib().aar_solver_default_options(C.byref(so))
    assert so.struct_size == C.sizeof(aar.CSolverOptions) and so.solver == aar.SOLVER_AUTO and so.deterministic == 0
    assert so.pcg_eta == 0.0 and so.pcg_eta_loose == 0.0 and so.pcg_eta_switch == 0.0 and so.pcg_abs_tol == 0.0 and so.pcg_max_it == 0
