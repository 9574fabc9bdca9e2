"""Soak test: the same 1024-instance W-mix batch solved 200 times on three handles that run concurrently
on three streams; every repetition must reproduce the first BIT FOR BIT (z, n_iter, flags, details).

The fused kernel carries hand-placed wait states that no compiler pass checks (tests/test_code_objects.py
checks them statically), and this code base has seen one timing-dependent wrong-answer bug it could not
explain (DESIGN.md section 7, N > 256): a kernel whose result depends on what else the chip is doing shows
up here as a repetition that differs.  Three kernels are covered: the whole-solve fused kernel, its
two-launch form and the generic kernel.  Reference work: one FCCQP::Solve per instance,
/root/reference/src/fcc_qp.cpp:114-191 (deterministic by construction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kernels", [("fused", "split", "fused"), ("generic", "split", "generic")])
def test_two_hundred_repetitions_on_three_concurrent_handles(kernels):
    import torch
    import fcc_qp_amd as F
    from fcc_qp_amd import problems as P

    nb, reps = 1024, 200
    kmap = {"fused": F.KERNEL_FUSED, "split": F.KERNEL_FUSED_SPLIT, "generic": F.KERNEL_GENERIC}
    dev = torch.device("cuda", 0)
    base = P.workload("W-mix", nb, P.WALKING, distinct=128)
    handles = []
    for k, name in enumerate(kernels):
        solver = F.FCCQPBatch(*P.WALKING.ctor_args, nb, 0)
        o = F.FCCQPOptions()
        o.max_iter, o.rho, o.eps_fcone, o.eps_bound = (P.HARNESS_OPTIONS[x] for x in ("max_iter", "rho", "eps_fcone", "eps_bound"))
        solver.set_options(o)
        solver.set_kernel(kmap[name])
        roll = 7 * k                                       # every handle its own order of the same instances
        tens = [torch.from_numpy(np.ascontiguousarray(np.roll(a, roll, axis=0))).to(dev) for a in base.arrays()]
        handles.append((solver, tens, torch.cuda.Stream(device=dev)))
    torch.cuda.synchronize()
    first, bad = [], []
    for solver, tens, st in handles:
        with torch.cuda.stream(st):
            F.solve_tensors(solver, *tens, stream=st)
            r = F.result_tensors(solver)
            first.append({k: v.clone() for k, v in r.items()})
            bad.append(torch.zeros((), dtype=torch.int64, device=dev))
    for _ in range(reps):
        for h, (solver, tens, st) in enumerate(handles):
            with torch.cuda.stream(st):
                F.solve_tensors(solver, *tens, stream=st)
                r = F.result_tensors(solver)
                for key, ref in first[h].items():
                    # (bitwise: NaN-safe comparison of the raw words)
                    a = r[key].view(torch.int64) if r[key].dtype == torch.float64 else r[key]
                    b = ref.view(torch.int64) if ref.dtype == torch.float64 else ref
                    bad[h] += (a != b).sum()
    torch.cuda.synchronize()
    assert [int(b.item()) for b in bad] == [0, 0, 0], f"repetitions differed: {[int(b.item()) for b in bad]} words"
    # and the three handles agree with each other on the instances they share (order rolled back)
    z0 = first[0]["z"].cpu().numpy()
    for h in (1, 2):
        zh = np.roll(first[h]["z"].cpu().numpy(), -7 * h, axis=0)
        if kernels[h] != "generic" and kernels[0] != "generic":
            assert np.array_equal(zh, z0)
        else:
            assert np.max(np.abs(zh - z0)) <= 1e-9 * (1 + np.max(np.abs(z0)))
    it = first[0]["n_iter"].cpu().numpy()
    assert (it == 100).sum() > 100 and (it < 5).sum() > 100          # the mix is a mix
