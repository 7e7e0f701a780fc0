"""N > 1 by construction: the frame-range sharding and the one exchange step of the path, checked with two CPU
processes over gloo.  Each rank evaluates ITS frames' block normal equations with the CPU oracle, eliminates its
frame poses (per-frame Schur complement), and the ranks all-reduce [S, rhs, sum r^2] -- the same decomposition
libaar runs over RCCL (automatic-ar_amd/csrc/ba_capi.hip, damped_try).  The sum must equal the unsharded system.
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "automatic-ar_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def reduced_system(ds, H, B, mu):
    """Schur complement of the frame blocks of (H + mu I) onto cameras+markers."""
    ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
    U, W, V = H[:ns, :ns], H[:ns, ns:], H[ns:, ns:]
    S = U.copy()
    rhs = B[:ns].copy()
    for f in range(ds.num_frames):
        sl = slice(6 * f, 6 * f + 6)
        Vi = np.linalg.inv(V[sl, sl] + mu * np.eye(6))
        Y = W[:, sl] @ Vi
        S -= Y @ W[:, sl].T
        rhs -= Y @ B[ns + 6 * f: ns + 6 * f + 6]
    return S, rhs


def shard_dataset(ds, f0, f1):
    import aar

    sub = aar.Dataset.__new__(aar.Dataset)
    sub.__dict__.update(ds.__dict__)
    keep = (ds.obs_frame >= f0) & (ds.obs_frame < f1)
    for k in ("obs_frame", "obs_cam", "obs_marker", "obs_uv"):
        setattr(sub, k, getattr(ds, k)[keep])
    sub.num_obs = int(keep.sum())
    return sub


def _worker(rank, world, port, out):
    import torch
    import torch.distributed as dist

    import aar
    import oracle_lib as ol
    from conftest import load_golden

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ds, _ = load_golden("g2_small")
    begin = aar.plan_shards(np.bincount(ds.obs_frame, minlength=ds.num_frames), world)
    sub = shard_dataset(ds, begin[rank], begin[rank + 1])
    o = ol.Oracle(sub)   # all frames keep their columns; the other ranks' frames simply have no observations here
    H, B = o.normal_equations(sub.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    r = o.residuals(sub.x_full, res_mode=ol.RES_F64)
    mu = 123.0
    # eliminate only the frames this rank owns
    ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
    S = H[:ns, :ns].copy()
    rhs = B[:ns].copy()
    for f in range(begin[rank], begin[rank + 1]):
        sl = slice(ns + 6 * f, ns + 6 * f + 6)
        Vi = np.linalg.inv(H[sl, sl] + mu * np.eye(6))
        Y = H[:ns, sl] @ Vi
        S -= Y @ H[:ns, sl].T
        rhs -= Y @ B[sl]
    payload = torch.from_numpy(np.concatenate([S.reshape(-1), rhs, [float((r ** 2).sum())]]))
    dist.all_reduce(payload)                       # the one exchange step: [S, rhs, sum r^2]
    if rank == 0:
        np.save(out, payload.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_schur_allreduce_equals_unsharded(tmp_path):
    import torch.multiprocessing as mp

    import oracle_lib as ol
    from conftest import load_golden

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "reduced.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    ds, _ = load_golden("g2_small")
    o = ol.Oracle(ds)
    H, B = o.normal_equations(ds.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    S, rhs = reduced_system(ds, H, B, 123.0)
    ns = S.shape[0]
    np.testing.assert_allclose(got[: ns * ns].reshape(ns, ns), S, rtol=1e-10, atol=1e-9 * np.abs(S).max())
    np.testing.assert_allclose(got[ns * ns: ns * ns + ns], rhs, rtol=1e-10, atol=1e-9 * np.abs(rhs).max())
    r = o.residuals(ds.x_full, res_mode=ol.RES_F64)
    np.testing.assert_allclose(got[-1], float((r ** 2).sum()), rtol=1e-12)
    # and the reduced solve reproduces the full damped step for the shared parameters
    d = o.damped_solve(ds.x_full, 123.0, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    ds_ = np.linalg.solve(S + 123.0 * np.eye(ns), rhs)
    np.testing.assert_allclose(ds_, d[:ns], rtol=1e-7, atol=1e-9 * np.abs(d).max())


def test_shard_plan_is_identical_on_every_rank_and_covers_all_frames():
    import aar

    ds = aar.synth(3)
    counts = np.bincount(ds.obs_frame, minlength=ds.num_frames)
    for world in (2, 4, 8):
        b = aar.plan_shards(counts, world)
        per = [counts[b[r]:b[r + 1]].sum() for r in range(world)]
        assert sum(per) == ds.num_obs and max(per) / (ds.num_obs / world) < 1.05


def _pcg_worker(rank, world, port, out):
    """The sharded PCG of csrc/pcg_kernels.hip (k_pcgd_*) in numpy, one process per rank: a rank's own frames' W blocks and its PARTIAL U, g0;
    set-up shares all-reduced once, then one all-reduce of the n-vector y per CG iteration; the vector updates replicated."""
    import torch
    import torch.distributed as dist

    import aar
    import oracle_lib as ol
    from conftest import load_golden

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    ds, _ = load_golden("g2_small")
    begin = aar.plan_shards(np.bincount(ds.obs_frame, minlength=ds.num_frames), world)
    sub = shard_dataset(ds, begin[rank], begin[rank + 1])
    H, B = ol.Oracle(sub).normal_equations(sub.x_full, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    mu, eta = 123.0, 1e-10
    ns = 6 * (ds.num_cams - 1) + 6 * (ds.num_markers - 1)
    U, g0 = H[:ns, :ns], B[:ns]                                     # this rank's partial sums (pass B over its observations)
    mine = range(begin[rank], begin[rank + 1])
    Wf = {f: H[:ns, ns + 6 * f: ns + 6 * f + 6] for f in mine}
    Vi = {f: np.linalg.inv(H[ns + 6 * f: ns + 6 * f + 6, ns + 6 * f: ns + 6 * f + 6] + mu * np.eye(6)) for f in mine}
    hf = {f: Vi[f] @ B[ns + 6 * f: ns + 6 * f + 6] for f in mine}

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64).copy())
        dist.all_reduce(t)
        return t.numpy()

    # set-up: the rank's share of the diagonal blocks of S (without mu) and of the right-hand side -> one all-reduce
    ne = ns // 6
    share = np.zeros((ne, 6, 7))
    for e in range(ne):
        sl = slice(6 * e, 6 * e + 6)
        share[e, :, :6] = U[sl, sl] - sum(Wf[f][sl] @ Vi[f] @ Wf[f][sl].T for f in mine)
        share[e, :, 6] = g0[sl] - sum(Wf[f][sl] @ hf[f] for f in mine)
    share = allreduce(share.reshape(-1)).reshape(ne, 6, 7)
    Minv = np.stack([np.linalg.inv(share[e, :, :6] + mu * np.eye(6)) for e in range(ne)])
    b = share[:, :, 6].reshape(-1)
    prec = lambda r: np.einsum("eij,ej->ei", Minv, r.reshape(ne, 6)).reshape(-1)
    x, r = np.zeros(ns), b.copy()
    z = prec(r); p = z.copy(); rz = r @ z; bb = r @ r
    its, collectives = 0, 1
    while r @ r > eta * eta * bb and its < 500:
        y_local = U @ p - sum(Wf[f] @ (Vi[f] @ (Wf[f].T @ p)) for f in mine)        # frame pass + entity pass over THIS rank's frames
        y = allreduce(y_local) + mu * p                                               # the iteration's one collective: 8 n bytes
        collectives += 1
        alpha = rz / (p @ y)
        x += alpha * p; r -= alpha * y
        z = prec(r); rz_new = r @ z
        p = z + (rz_new / rz) * p; rz = rz_new
        its += 1
    if rank == 0:
        np.save(out, np.concatenate([x, [its, collectives]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_pcg_through_the_frame_blocks_equals_the_direct_step(tmp_path):
    # AAR_SOLVER=pcg with a communicator: nothing but n-vectors is ever exchanged, and the converged CG solution
    # of the rank-summed operator is the shared part of the full damped step
    import torch.multiprocessing as mp

    import oracle_lib as ol
    from conftest import load_golden

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "pcg.npy")
    mp.spawn(_pcg_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    ds, _ = load_golden("g2_small")
    d = ol.Oracle(ds).damped_solve(ds.x_full, 123.0, jac_mode=ol.JAC_ANALYTIC, res_mode=ol.RES_F64)
    ns = len(got) - 2
    np.testing.assert_allclose(got[:ns], d[:ns], rtol=1e-6, atol=1e-8 * np.abs(d).max())
    assert 3 < got[ns] < 500 and got[ns + 1] == got[ns] + 1            # one all-reduce per iteration + the set-up's
