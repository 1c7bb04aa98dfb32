"""The host side of libnerf_hip.so under AddressSanitizer + UndefinedBehaviorSanitizer: `make asan`
builds the library's sources with the host code instrumented and a driver (tests/host/abi_argument_checks.cpp) that walks
every argument-validation path of the C ABI -- null pointers, sizes out of range, malformed level sets and point sources,
unsupported network shapes, misaligned buffers, the no-op sizes.  None of these calls reaches a kernel launch, so the run
needs no GPU.  Passing = every call returned what include/nerf_hip.h promises and neither sanitizer reported anything."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_abi_argument_checks_under_asan_and_ubsan():
    csrc = os.path.join(ROOT, 'nerf_amd', 'csrc')
    subprocess.check_call(['make', '-C', csrc, '-j8', 'asan'], stdout=subprocess.DEVNULL)      # up to date after build()
    exe = os.path.join(ROOT, 'nerf_amd', 'lib', 'abi_argument_checks_asan')
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=300)
    out = r.stdout.decode(errors='replace')
    assert r.returncode == 0, out[-4000:]
    assert 'all argument checks passed' in out and 'runtime error' not in out and 'AddressSanitizer' not in out, out[-4000:]
