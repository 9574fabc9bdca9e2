"""AddressSanitizer + UndefinedBehaviorSanitizer build of the host layer (fcc_qp_amd/csrc/fcc_qp.cpp:
FCCQP, FCCQPBatch, FCCQPSharded) over a stub of the C ABI -- the CPU-side check of the shard
arithmetic, the host threads and the view packing (GPU sanitizers are not available on the pool).
The reference has no sanitizer configuration."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_layer_asan")
    srcs = [os.path.join(ROOT, "tests", "host_layer", f) for f in ("driver.cpp", "stub_capi.cpp")] + \
           [os.path.join(ROOT, "fcc_qp_amd", "csrc", "fcc_qp.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
           "-pthread", f"-I{os.path.join(ROOT, 'include')}", f"-I{os.path.join(ROOT, 'fcc_qp_amd', 'csrc')}", *srcs, "-o", exe]
    subprocess.check_call(cmd)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all checks passed" in out.stdout
