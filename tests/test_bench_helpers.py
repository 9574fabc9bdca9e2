"""bench.py's host logic that needs no GPU: the clock-settling loop (runs untimed steps until the step time is steady,
bounded) with stand-ins for torch's events, and the MFMA count restated from the kernel's loop bounds."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


class _FakeTorch:
    """torch.cuda.Event / synchronize over a virtual clock that `step` advances"""
    def __init__(self):
        self.now = 0.0
        outer = self

        class Event:
            def __init__(self, enable_timing=True): self.t = None
            def record(self): self.t = outer.now
            def elapsed_time(self, other): return other.t - self.t

        class Cuda:
            pass
        self.cuda = Cuda()
        self.cuda.Event = Event
        self.cuda.synchronize = lambda: None


def test_settle_stops_when_the_step_time_is_steady():
    t = _FakeTorch()
    times = iter([0.30] * 10 + [0.29] * 10 + [0.28] * 10 + [0.27] * 10 + [0.265] * 1000)     # a ramp, then steady
    def step(): t.now += next(times)
    n = bench._settle(t, step, True)
    assert n == 70 and len(bench._settle.trace) == 7            # four ramp blocks, then three that agree
    assert abs(bench._settle.trace[0] - 0.30) < 1e-9 and abs(bench._settle.trace[-1] - 0.265) < 1e-9


def test_settle_is_bounded_and_can_be_switched_off():
    t = _FakeTorch()
    k = [0]
    def step():
        k[0] += 1; t.now += 0.2 + 0.05 * ((k[0] // 10) % 2)       # never steady
    assert bench._settle(t, step, True) == 400
    assert bench._settle(t, step, False) == 0 and bench._settle.trace == []


def test_settle_on_the_host_clock_for_steps_on_side_streams(monkeypatch):
    """two_streams launches on streams of its own: an event pair on the current stream brackets nothing there, so the
    settling loop times its blocks on the host clock instead."""
    t = _FakeTorch()
    times = iter([0.30] * 5 + [0.28] * 5 + [0.265] * 1000)
    def step(): t.now += next(times)
    monkeypatch.setattr(bench.time, "perf_counter", lambda: t.now * 1e-3)
    n = bench._settle(t, step, True, block=5, host_clock=True)
    assert n == 25 and abs(bench._settle.trace[-1] - 0.265) < 1e-9


def test_mfma_count_of_the_walking_shape():
    f, g, w = bench.fused_mfma_count(60, 38)
    assert (f, g, w) == (456, 196, 64)                           # 716 in the shipped code object (tests/test_code_objects.py)
