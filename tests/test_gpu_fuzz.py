"""A bounded run of tools/fuzz_parity.py: random shapes (listed and unlisted, one to eight register slots), kernels,
options, batch sizes and inputs, cold and warm, f64 and f32 storage, against the oracle; disagreements beyond the stated
tolerance are refereed in 80-bit arithmetic (tools/referee_longdouble.py).  The long runs (75 000 cases per seed) are a
tool; this keeps a fixed 1 500-case prefix of one seed in the suite (bounded by the case count: a run that the time
budget cuts short fails).  Round 4: the tool found the generic kernel's
four-wave race between the presolve's sweeps and the copy of rho G (tests/test_gpu_parity.py, shapes (89, 40), ...)."""
import pytest

pytestmark = pytest.mark.gpu


def test_fixed_prefix_of_the_random_parity_run():
    from tools import fuzz_parity
    assert fuzz_parity.main(budget=1800.0, seed=4, max_cases=1500) == 0


def test_fixed_prefix_of_the_random_call_sequences():
    """tools/fuzz_api.py: chains of option / warm-start / kernel changes, state transplants and solves through every
    entry point (host, host-async, device pointers on three streams, f32), fetched by GetSolution, by the asynchronous
    copy or not at all, mirrored by one oracle object per instance (50 000 chains / 236 000 solves clean in round 4)."""
    from tools import fuzz_api
    assert fuzz_api.main(budget=1800.0, seed=5, max_chains=400) == 0


def test_fixed_prefix_of_the_random_python_surface_run():
    """tools/fuzz_python.py: the reference's FCCQP class (and FCCQPBatch) fed numpy arrays in every layout the
    reference's Eigen::Ref casters accept -- C / Fortran order, strided and reversed views, float32, (n,1) / (1,n)
    vectors, lists, keyword arguments (50 000 cases clean in round 4)."""
    from tools import fuzz_python
    assert fuzz_python.main(budget=1800.0, seed=2, max_cases=300) == 0
