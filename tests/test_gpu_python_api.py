"""The reference's Python call sequence (/root/reference/fcc_qp_test.py:77-89) and the new batched
API, through the pybind11 module, on the GPU."""
import numpy as np
import pytest

from fcc_qp_amd import problems as P
from tests.helpers import Z_TOL

pytestmark = pytest.mark.gpu


def harness_options(F):
    o = F.FCCQPOptions()
    o.rho = 5e-5; o.eps_fcone = 1e-6; o.eps_bound = 1e-6; o.max_iter = 100     # fcc_qp_test.py:78-83
    return o


def test_reference_harness_loop_matches_oracle(oracle):
    """solver.set_warm_start(i > 0); solver.Solve(...); solver.GetSolution()  over a sequence."""
    from fcc_qp import FCCQP, FCCQPOptions, FCCQPSolution      # the reference's import line (fcc_qp_test.py:6)
    import fcc_qp_amd as F
    qps = P.gait_sequence(8, P.WALKING, seed0=9100)
    solver = FCCQP(60, 38, 12, 38)
    solver.set_options(harness_options(F))
    ref = oracle.OracleFCCQP(60, 38, 12, 38)
    op = oracle.FCCQPOptions(); op.__dict__.update(P.HARNESS_OPTIONS); ref.set_options(op)
    for i, qp in enumerate(qps):
        solver.set_warm_start(i > 0); ref.set_warm_start(i > 0)
        solver.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
        result = solver.GetSolution()
        assert isinstance(result, FCCQPSolution)
        ref.Solve(**qp); r = ref.GetSolution()
        assert np.max(np.abs(result.z - r.z)) <= Z_TOL * (1 + np.max(np.abs(r.z)))
        d = result.details
        assert abs(d.n_iter - r.details.n_iter) <= 1
        assert abs(d.bounds_viol - r.details.bounds_viol) <= 1e-6 * (1 + r.details.bounds_viol)
        assert abs(d.friction_cone_viol - r.details.friction_cone_viol) <= 1e-6 * (1 + r.details.friction_cone_viol)
        assert d.solve_time > 0 and d.factorization_time > 0 and d.factorization_time < d.solve_time


def test_keyword_arguments_lists_and_foreign_layouts(oracle):
    """Eigen::Ref casters of the reference accept F-ordered / non-f64 arrays and lists."""
    import fcc_qp_amd as F
    qp = P.osc_qp(1000, P.WALKING)
    s = F.FCCQP(num_vars=60, num_equality_constraints=38, nc=12, lambda_c_start=38)
    s.set_options(harness_options(F))
    s.Solve(Q=qp["Q"], b=qp["b"], A_eq=qp["A_eq"], b_eq=qp["b_eq"], friction_coeffs=qp["friction_coeffs"],
            lb=qp["lb"], ub=qp["ub"])
    z0 = s.GetSolution().z
    s.Solve(np.asfortranarray(qp["Q"]), list(qp["b"]), np.asfortranarray(qp["A_eq"]), qp["b_eq"].reshape(-1, 1),
            list(qp["friction_coeffs"]), qp["lb"], qp["ub"][::1])
    assert np.array_equal(s.GetSolution().z, z0)
    s.set_rho(5e-5); s.set_max_iter(100)
    s.Solve(**{k: qp[k] for k in ("Q", "b", "A_eq", "b_eq", "friction_coeffs", "lb", "ub")})
    assert np.array_equal(s.GetSolution().z, z0)


def test_error_behaviour():
    import fcc_qp_amd as F
    qp = P.osc_qp(1000, P.WALKING)
    s = F.FCCQP(60, 38, 12, 38)
    with pytest.raises(ValueError):
        s.Solve(qp["Q"][:59, :59], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
    with pytest.raises(ValueError):
        s.Solve(qp["Q"], qp["b"], qp["A_eq"][:37], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
    with pytest.raises(IndexError):          # std::out_of_range from .at() in the reference
        s.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"][:3], qp["lb"], qp["ub"])
    with pytest.raises(ValueError):
        F.FCCQP(60, 38, 13, 38)
    z = s.GetSolution()                      # before any successful Solve: zeros, like the reference
    assert np.array_equal(z.z, np.zeros(60)) and z.details.n_iter == 0


def test_batch_api_matches_oracle_and_single_instance_api(oracle):
    import fcc_qp_amd as F
    nb = 40
    B = P.workload("W-mix", nb, P.WALKING)
    bs = F.FCCQPBatch(60, 38, 12, 38, nb)
    bs.set_options(harness_options(F))
    bs.Solve(*B.arrays())
    sol = bs.GetSolution()
    z_ref, it_ref, _, _ = oracle.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *B.arrays(), nthreads=4)
    err = np.max(np.abs(sol.z - z_ref), axis=1) / (1 + np.max(np.abs(z_ref), axis=1))
    assert err.max() <= Z_TOL and np.abs(sol.n_iter - it_ref).max() <= 1
    assert len(sol) == nb and sol.z.shape == (nb, 60)
    one = F.FCCQP(60, 38, 12, 38); one.set_options(harness_options(F))
    qp = B.instance(7)
    one.Solve(qp["Q"], qp["b"], qp["A_eq"], qp["b_eq"], qp["friction_coeffs"], qp["lb"], qp["ub"])
    assert np.array_equal(one.GetSolution().z, sol[7].z)          # same kernel, same bits
    assert sol[7].details.n_iter == sol.n_iter[7]
    # a single friction row is shared by every instance
    bs.Solve(B.Q, B.b, B.A_eq, B.b_eq, B.friction_coeffs[0], B.lb, B.ub)
    assert np.array_equal(bs.GetSolution().z, sol.z)
    with pytest.raises(ValueError):
        bs.Solve(B.Q[:-1], B.b, B.A_eq, B.b_eq, B.friction_coeffs, B.lb, B.ub)
    with pytest.raises(IndexError):
        bs.Solve(B.Q, B.b, B.A_eq, B.b_eq, B.friction_coeffs[:, :3], B.lb, B.ub)


def test_torch_tensors_resident_in_hbm_give_the_same_bits():
    import torch
    import fcc_qp_amd as F
    nb = 64
    B = P.workload("W-sat", nb, P.WALKING)
    bs = F.FCCQPBatch(60, 38, 12, 38, nb); bs.set_options(harness_options(F))
    bs.Solve(*B.arrays()); host = bs.GetSolution()
    tens = [torch.from_numpy(a).cuda() for a in B.arrays()]
    F.solve_tensors(bs, *tens)
    torch.cuda.synchronize()
    dev = bs.GetSolution()
    assert np.array_equal(dev.z, host.z) and np.array_equal(dev.n_iter, host.n_iter)
    assert bs.LastKernelMilliseconds() > 0
    assert bs.AlgorithmicBytes() == 8 * nb * (60 * 60 + 38 * 60 + 4 * 60 + 38 + 4)
    with pytest.raises(ValueError):                                                    # mixed dtypes
        F.solve_tensors(bs, tens[0].float(), *tens[1:])
    # float32 tensors / arrays take the fp32-storage path: same bits as the widened float64 problem
    t32 = [t.float() for t in tens]
    F.solve_tensors(bs, *t32)
    torch.cuda.synchronize()
    dev32 = bs.GetSolution()
    assert bs.AlgorithmicBytes() == nb * (4 * 6102 + 8 * 60)
    bs.Solve(*[t.cpu().numpy() for t in t32]); host32 = bs.GetSolution()               # float32 numpy
    bs.Solve(*[t.double().cpu().numpy() for t in t32]); wide = bs.GetSolution()        # widened float64
    assert np.array_equal(dev32.z, wide.z) and np.array_equal(host32.z, wide.z)
    assert np.array_equal(dev32.n_iter, wide.n_iter)
    ptrs = bs.device_pointers()
    assert ptrs["z"] != 0 and ptrs["n_iter"] != 0


def test_device_arrays_need_only_natural_alignment():
    """Q and A_eq views that start 8 bytes (fp64) / 4 bytes (fp32) into an allocation: the fused kernel reads
    them one element per lane, so they take the same kernel and give the same bits as aligned copies."""
    import torch
    import fcc_qp_amd as F
    nb = 32
    B = P.workload("W-mix", nb, P.WALKING)
    bs = F.FCCQPBatch(60, 38, 12, 38, nb); bs.set_options(harness_options(F))
    for dtype in (torch.float64, torch.float32):
        tens = [torch.from_numpy(a).cuda().to(dtype) for a in B.arrays()]
        F.solve_tensors(bs, *tens); torch.cuda.synchronize()
        ref = bs.GetSolution()
        shifted = []
        for t in tens:
            buf = torch.empty(t.numel() + 1, dtype=dtype, device="cuda")
            v = buf[1:].view(t.shape)
            v.copy_(t)
            assert v.data_ptr() % 16 != 0 and v.is_contiguous()
            shifted.append(v)
        F.solve_tensors(bs, *shifted); torch.cuda.synchronize()
        got = bs.GetSolution()
        assert np.array_equal(got.z, ref.z) and np.array_equal(got.n_iter, ref.n_iter)


def test_harness_reproduces_reference_loop_and_series(oracle, tmp_path):
    """fcc_qp_amd.harness: the reference's load -> warm-started loop -> eight series
    (/root/reference/fcc_qp_test.py:22-32,44-69,72-89) from a log file in the reference's format."""
    from fcc_qp_amd import harness as H
    qps = P.gait_sequence(10, P.WALKING, seed0=9300)
    path = str(tmp_path / "id_qp_log_walking.npz")
    P.save_qp_log(path, qps)
    rep = H.main([path, "--out", str(tmp_path / "r.json")])
    seq, bat = rep["sequence"], rep["batch"]
    assert seq["count"] == bat["count"] == 10
    for key, width in (("u", 10), ("vdot", 22), ("lambda_h", 6), ("lambda_c", 12)):     # slices at :52-56
        assert np.asarray(seq[key]).shape == (10, width)
    for key in ("solve_time", "n_iter", "friction_cone_viol", "bounds_viol"):
        assert len(seq[key]) == 10
    # the sequential series equals the oracle driven through the same loop
    o = oracle.OracleFCCQP(60, 38, 12, 38)
    op = oracle.FCCQPOptions(); op.__dict__.update(P.HARNESS_OPTIONS); o.set_options(op)
    for i, qp in enumerate(qps):
        o.set_warm_start(i > 0); o.Solve(**qp); s = o.GetSolution()
        assert np.max(np.abs(np.asarray(seq["u"][i]) - s.z[22:32])) <= 1e-6 * (1 + np.max(np.abs(s.z)))
        assert abs(seq["n_iter"][i] - s.details.n_iter) <= 1
    # the batch series is the cold-start solution of every QP: first entries coincide
    assert np.allclose(bat["lambda_c"][0], seq["lambda_c"][0], rtol=1e-9, atol=1e-9)
    import json
    assert json.load(open(str(tmp_path / "r.json")))["sequence"]["count"] == 10


def test_run_chains_equals_one_oracle_chain_per_log(oracle):
    """harness.run_chains: K logs of different lengths advanced in lock-step as one warm-started batch; every
    chain must reproduce the reference loop (/root/reference/fcc_qp_test.py:86-89) run on that log alone."""
    from fcc_qp_amd import harness as H
    lengths = [7, 4, 9, 1, 6]
    logs = [P.gait_sequence(L, P.WALKING, seed0=8100 + 40 * k) for k, L in enumerate(lengths)]
    reps = H.run_chains(logs, options=P.PAPER_OPTIONS)
    assert [r["count"] for r in reps] == lengths
    for k, (log, rep) in enumerate(zip(logs, reps)):
        o = oracle.OracleFCCQP(60, 38, 12, 38)
        op = oracle.FCCQPOptions(); op.__dict__.update(P.PAPER_OPTIONS); o.set_options(op)
        for i, qp in enumerate(log):
            o.set_warm_start(i > 0); o.Solve(**qp); s = o.GetSolution()
            z = np.r_[rep["vdot"][i], rep["u"][i], rep["lambda_h"][i], rep["lambda_c"][i]]
            assert np.max(np.abs(z - s.z[:50])) <= 1e-6 * (1 + np.max(np.abs(s.z))), (k, i)
            assert abs(rep["n_iter"][i] - s.details.n_iter) <= 1
            assert abs(rep["friction_cone_viol"][i] - s.details.friction_cone_viol) <= 1e-6 * (1 + abs(s.details.friction_cone_viol))
            assert abs(rep["bounds_viol"][i] - s.details.bounds_viol) <= 1e-6 * (1 + abs(s.details.bounds_viol))
        # ... and the one-chain driver on the same log
        one = H.run_sequence(log, options=P.PAPER_OPTIONS)
        assert np.array_equal(np.asarray(one["u"]), np.asarray(rep["u"])) and one["n_iter"] == rep["n_iter"]
    with pytest.raises(ValueError):
        H.run_chains([logs[0], []])


# ---- zero-copy API: __dlpack__ in, tensors out, against the oracle ----------------
class _DLPackOnly:
    """A foreign device array: exports nothing but the DLPack protocol."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__(stream=stream) if stream is not None else self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def test_dlpack_in_tensors_out_match_oracle(oracle):
    import torch
    import fcc_qp_amd as F
    nb = 96
    B = P.workload("W-mix", nb, P.WALKING)
    z_ref, it_ref, det_ref, _ = oracle.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *B.arrays(), nthreads=8)
    bs = F.FCCQPBatch(60, 38, 12, 38, nb); bs.set_options(harness_options(F))
    tens = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in B.arrays()]
    F.solve_tensors(bs, *[_DLPackOnly(t) for t in tens])           # any __dlpack__ producer
    out = F.result_tensors(bs)                                      # torch tensors aliasing the solver's buffers
    assert out["z"].is_cuda and out["z"].shape == (nb, 60) and out["z"].dtype == torch.float64
    assert out["z"].data_ptr() == bs.device_pointers()["z"]         # no copy
    torch.cuda.synchronize()
    z = out["z"].cpu().numpy(); it = out["n_iter"].cpu().numpy()
    err = np.max(np.abs(z - z_ref), axis=1) / (1 + np.max(np.abs(z_ref), axis=1))
    assert err.max() <= 1e-9 and np.max(np.abs(it - it_ref)) <= 1
    for key, ref in (("bounds_viol", [d.bounds_viol for d in det_ref]), ("friction_cone_viol", [d.friction_cone_viol for d in det_ref]),
                     ("eps_bounds", [d.admm_residual_bounds for d in det_ref]),
                     ("eps_friction_cone", [d.admm_residual_friction_cone for d in det_ref])):
        got = out[key].cpu().numpy(); ref = np.asarray(ref)
        same = it == it_ref
        assert np.all(np.abs(got - ref)[same] <= 1e-6 * (1 + np.abs(ref)[same])), key
    # results stay on the device for the next consumer: hand z on through DLPack, reduce there
    z2 = torch.from_dlpack(out["z"])
    assert z2.data_ptr() == out["z"].data_ptr()
    assert abs(float(z2.abs().max()) - np.abs(z).max()) <= 1e-9 * (1 + np.abs(z).max())
    # the tensors keep the solver alive
    del bs
    assert np.array_equal(out["z"].cpu().numpy(), z)


def test_solve_tensors_validates_shapes_and_device():
    import torch
    import fcc_qp_amd as F
    nb = 8
    B = P.workload("W-sat", nb, P.WALKING)
    bs = F.FCCQPBatch(60, 38, 12, 38, nb); bs.set_options(harness_options(F))
    assert (bs.batch, bs.num_vars, bs.num_equality_constraints, bs.nc, bs.device) == (nb, 60, 38, 12, 0)
    tens = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in B.arrays()]
    bad = list(tens); bad[0] = tens[0][:-1]                               # one instance short
    with pytest.raises(ValueError, match="Q has shape"):
        F.solve_tensors(bs, *bad)
    bad = list(tens); bad[4] = tens[4][0]                                 # one shared friction row: host path only
    with pytest.raises(ValueError, match="friction_coeffs has shape"):
        F.solve_tensors(bs, *bad)
    bad = list(tens); bad[2] = tens[2].transpose(1, 2)                    # right numel, wrong layout
    with pytest.raises(ValueError):
        F.solve_tensors(bs, *bad)
    bad = list(tens); bad[1] = tens[1].cpu()
    with pytest.raises(ValueError, match="not on a GPU"):
        F.solve_tensors(bs, *bad)
    with pytest.raises(TypeError):
        F.solve_tensors(bs, B.Q.tolist(), *tens[1:])
    F.solve_tensors(bs, *tens); torch.cuda.synchronize()                  # the good call still works


# ---- one host batch over several handles -----------------------------------------------
def test_sharded_solve_equals_unsharded_bit_for_bit(oracle):
    """Two shards on device 0 (this box has one GPU): [0, B/2) and [B/2, B) solved by two handles
    from two host threads, gathered into one result -- bit-identical to one handle solving all."""
    import fcc_qp_amd as F
    nb = 100                                        # not a multiple of 3: ragged shards below
    B = P.workload("W-mix", nb, P.WALKING)
    one = F.FCCQPBatch(60, 38, 12, 38, nb); one.set_options(harness_options(F))
    one.Solve(*B.arrays()); ref = one.GetSolution()
    for devices in ([0, 0], [0, 0, 0]):
        sh = F.FCCQPSharded(60, 38, 12, 38, nb, devices); sh.set_options(harness_options(F))
        assert sh.shards == len(devices) and sh.batch == nb
        assert [sh.shard_begin(g) for g in range(len(devices) + 1)] == [nb * g // len(devices) for g in range(len(devices) + 1)]
        sh.Solve(*B.arrays()); got = sh.GetSolution()
        assert np.array_equal(got.z, ref.z) and np.array_equal(got.n_iter, ref.n_iter)
        assert np.array_equal(got.flags, ref.flags) and np.array_equal(got.bounds_viol, ref.bounds_viol)
        assert sh.LastKernelMilliseconds() > 0
        # warm-start state stays on the owning handle: a warm chain equals the unsharded warm chain
        B2 = P.workload("W-mix", nb, P.WALKING, seed0=4000)
        sh.set_warm_start(True); sh.Solve(*B2.arrays()); w = sh.GetSolution()
        one.set_warm_start(True); one.Solve(*B2.arrays()); wref = one.GetSolution()
        assert np.array_equal(w.z, wref.z) and np.array_equal(w.n_iter, wref.n_iter)
        one.set_warm_start(False); one.Solve(*B.arrays())
    z_ref, it_ref, _, _ = oracle.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *B.arrays(), nthreads=8)
    err = np.max(np.abs(ref.z - z_ref), axis=1) / (1 + np.max(np.abs(z_ref), axis=1))
    assert err.max() <= 1e-9
    with pytest.raises(Exception):
        F.FCCQPSharded(60, 38, 12, 38, 2, [0, 0, 0])          # fewer instances than shards
    if F.device_count() == 1:
        with pytest.raises(F.FCCQPError):
            F.FCCQPSharded(60, 38, 12, 38, 8, [0, 1])          # no such device


def test_sharded_device_and_pinned_paths_equal_the_unsharded_solve_bit_for_bit():
    """FCCQPSharded from ONE process at the device-resident rate: SolveDevice (one set of device pointers per shard,
    asynchronous) and SolveAsync + GetZAsync (page-locked host arrays, every shard's copies and kernel on its own
    stream) -- three shards on device 0, ragged, cold and warm -- bit-identical to one handle solving the whole batch."""
    import torch
    import fcc_qp_amd as F
    nb = 100
    B = P.workload("W-mix", nb, P.WALKING)
    B2 = P.workload("W-mix", nb, P.WALKING, seed0=4000)
    one = F.FCCQPBatch(60, 38, 12, 38, nb); one.set_options(harness_options(F))
    one.Solve(*B.arrays()); ref = one.GetSolution()
    one.set_warm_start(True); one.Solve(*B2.arrays()); wref = one.GetSolution()
    dev = torch.device("cuda", 0)
    devices = [0, 0, 0]

    def shard_tensors(sh, batch):
        cuts = [sh.shard_begin(g) for g in range(sh.shards)] + [nb]
        return [[torch.from_numpy(np.ascontiguousarray(a[cuts[g]:cuts[g + 1]])).to(dev) for a in batch.arrays()] for g in range(sh.shards)]

    # ---- device pointers per shard
    sh = F.FCCQPSharded(60, 38, 12, 38, nb, devices); sh.set_options(harness_options(F))
    t1 = shard_tensors(sh, B)
    F.solve_tensors_sharded(sh, t1)                       # every shard on its own stream
    sh.Synchronize(); got = sh.GetSolution()
    assert np.array_equal(got.z, ref.z) and np.array_equal(got.n_iter, ref.n_iter) and np.array_equal(got.flags, ref.flags)
    t2 = shard_tensors(sh, B2)
    streams = [torch.cuda.Stream(device=dev) for _ in devices]
    sh.set_warm_start(True)
    F.solve_tensors_sharded(sh, t2, streams=streams)      # the caller's streams
    sh.Synchronize(); w = sh.GetSolution()
    assert np.array_equal(w.z, wref.z) and np.array_equal(w.n_iter, wref.n_iter)
    with pytest.raises(ValueError):
        F.solve_tensors_sharded(sh, t2[:2])
    with pytest.raises(ValueError):
        F.solve_tensors_sharded(sh, [t2[1], t2[0], t2[2]] if t2[0][0].shape != t2[1][0].shape else [[t.float() for t in t2[0]], t2[1], t2[2]])
    # ---- page-locked host arrays, asynchronous
    sh2 = F.FCCQPSharded(60, 38, 12, 38, nb, devices); sh2.set_options(harness_options(F))
    pin = [torch.from_numpy(np.ascontiguousarray(a)).pin_memory() for a in B.arrays()]
    z = torch.empty((nb, 60), dtype=torch.float64).pin_memory()
    sh2.SolveAsync(*(t.numpy() for t in pin))
    sh2.GetZAsync(z.numpy())
    sh2.Synchronize()
    assert np.array_equal(z.numpy(), ref.z)
    got2 = sh2.GetSolution()
    assert np.array_equal(got2.z, ref.z) and np.array_equal(got2.n_iter, ref.n_iter)
    with pytest.raises(ValueError):
        sh2.SolveAsync(*(a.astype(np.float32) if i == 0 else a for i, a in enumerate(B.arrays())))     # no silent conversion
    with pytest.raises(ValueError):
        sh2.SolveAsync(B.Q[:, ::-1, :], *B.arrays()[1:])                                                # no silent copy


def test_async_f32_host_solve_equals_the_synchronous_f32_path():
    """fccqp_solve_async_f32: float32 arrays all the way to the device, small vectors widened there --
    bit-identical to fccqp_solve_f32 (host-side widening) on the same arrays."""
    import ctypes
    from tests import capi as C
    L = C.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    L.fccqp_solve_async_f32.argtypes = [ctypes.c_void_p, fp, fp, fp, fp, fp, ctypes.c_int, fp, fp]
    nb = 48
    B = P.workload("W-mix", nb, P.WALKING)
    f32 = [np.ascontiguousarray(a.astype(np.float32)) for a in B.arrays()]
    s = C.Solver(*P.WALKING.ctor_args, batch=nb); s.set_options(**P.HARNESS_OPTIONS)
    s.solve_f32_ok(*f32); z_sync, d_sync = s.solution()
    q = [x.ctypes.data_as(fp) for x in f32]
    assert L.fccqp_solve_async_f32(s.h, q[0], q[1], q[2], q[3], q[4], P.WALKING.nc // 3, q[5], q[6]) == C.OK
    z_async, d_async = s.solution()
    assert np.array_equal(z_async, z_sync)
    assert [x.n_iter for x in d_async] == [x.n_iter for x in d_sync]
    assert L.fccqp_solve_async_f32(s.h, q[0], q[1], q[2], q[3], q[4], 3, q[5], q[6]) != C.OK     # n_friction must be nc/3


def test_async_host_solve_from_pinned_memory(oracle):
    """fccqp_solve_async + fccqp_get_z_async on page-locked buffers: two handles in flight, results
    equal to the synchronous path."""
    from tests import capi as C
    import ctypes
    L = C.lib()
    dp = ctypes.POINTER(ctypes.c_double)
    L.fccqp_solve_async.argtypes = [ctypes.c_void_p, dp, dp, dp, dp, dp, ctypes.c_int, dp, dp]
    L.fccqp_get_z_async.argtypes = [ctypes.c_void_p, dp]
    L.fccqp_host_register.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    L.fccqp_host_unregister.argtypes = [ctypes.c_void_p]
    nb = 256
    batches = [P.workload("W-mix", nb, P.WALKING, seed0=1000 + 500 * k) for k in range(2)]
    solvers = [C.Solver(*P.WALKING.ctor_args, batch=nb) for _ in range(2)]
    zs = [np.zeros((nb, 60)) for _ in range(2)]
    pinned = []
    for k in range(2):
        solvers[k].set_options(**P.HARNESS_OPTIONS)
        for a in list(batches[k].arrays()) + [zs[k]]:
            assert L.fccqp_host_register(a.ctypes.data, a.nbytes) == C.OK
            pinned.append(a)
    for k in range(2):                                        # both enqueued before either is waited for
        arrs = batches[k].arrays()
        rc = L.fccqp_solve_async(solvers[k].h, C._p(arrs[0]), C._p(arrs[1]), C._p(arrs[2]), C._p(arrs[3]), C._p(arrs[4]), 4,
                                 C._p(arrs[5]), C._p(arrs[6]))
        assert rc == C.OK
        assert L.fccqp_get_z_async(solvers[k].h, C._p(zs[k])) == C.OK
    for k in range(2):
        assert L.fccqp_synchronize(solvers[k].h) == C.OK
        z_sync, d = solvers[k].solution()
        assert np.array_equal(zs[k], z_sync)
        z_ref, it_ref, _, _ = oracle.solve_batch(P.WALKING.ctor_args, P.HARNESS_OPTIONS, *batches[k].arrays(), nthreads=8)
        err = np.max(np.abs(zs[k] - z_ref), axis=1) / (1 + np.max(np.abs(z_ref), axis=1))
        assert err.max() <= 1e-9
    arrs = batches[0].arrays()                                # a stride of friction rows is a host-path feature only
    wide = np.ascontiguousarray(np.concatenate([arrs[4], arrs[4]], axis=1))
    assert L.fccqp_solve_async(solvers[0].h, C._p(arrs[0]), C._p(arrs[1]), C._p(arrs[2]), C._p(arrs[3]), C._p(wide), 8,
                               C._p(arrs[5]), C._p(arrs[6])) == C.ERR_FRICTION
    for a in pinned:
        assert L.fccqp_host_unregister(a.ctypes.data) == C.OK


@pytest.mark.gpu
def test_kernel_timing_switch():
    """fccqp_set_kernel_timing(0): no start event in front of the kernel; results unchanged, LastKernelMilliseconds
    refuses, solve_time of a device-pointer solve is 0; switching it back on restores both."""
    import torch
    import fcc_qp_amd as F
    nb = 64
    B = P.workload("W-mix", nb, P.WALKING)
    tens = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in B.arrays()]
    s = F.FCCQPBatch(*P.WALKING.ctor_args, nb, 0)
    o = F.FCCQPOptions(); h = P.HARNESS_OPTIONS
    o.max_iter, o.rho, o.eps_fcone, o.eps_bound = h["max_iter"], h["rho"], h["eps_fcone"], h["eps_bound"]
    s.set_options(o)
    F.solve_tensors(s, *tens); torch.cuda.synchronize()
    z_on = np.array(s.GetSolution().z, copy=True)
    assert s.LastKernelMilliseconds() > 0.0
    s.set_kernel_timing(False)
    F.solve_tensors(s, *tens); torch.cuda.synchronize()
    sol = s.GetSolution()
    assert np.array_equal(np.asarray(sol.z), z_on)
    assert sol[0].details.solve_time == 0.0
    with pytest.raises(Exception):
        s.LastKernelMilliseconds()
    s.set_kernel_timing(True)
    F.solve_tensors(s, *tens); torch.cuda.synchronize()
    assert s.LastKernelMilliseconds() > 0.0 and s.GetSolution()[0].details.solve_time > 0.0


def test_batch_class_dispatch_order_switch():
    """FCCQPBatch.set_dispatch_order_by_previous_iterations (fccqp_set_dispatch_order): a chain's next launch longest
    instance first.  Results never depend on the order: two warm-started steps with the switch on and off, bit for bit."""
    import torch
    import fcc_qp_amd as F
    nb = 8 * torch.cuda.get_device_properties(0).multi_processor_count        # (where the order applies)
    base = P.workload("W-mix", nb, P.WALKING, distinct=64)
    moved = P.Batch(*[a.copy() for a in base.arrays()])
    moved.b[:] = moved.b + 0.03 * np.random.default_rng(1).standard_normal(moved.b.shape)
    out = []
    for on in (False, True):
        s = F.FCCQPBatch(60, 38, 12, 38, nb, 0)
        s.set_options(harness_options(F))
        s.set_dispatch_order_by_previous_iterations(on)
        s.Solve(*base.arrays())
        s.set_warm_start(True)
        s.Solve(*moved.arrays())
        sol = s.GetSolution()
        out.append((np.asarray(sol.z).copy(), np.asarray(sol.n_iter).copy()))
        assert s.kernel_in_use() == F.KERNEL_FUSED
    assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
    assert len(set(out[0][1].tolist())) > 1
