"""AddressSanitizer + UndefinedBehaviourSanitizer build of the host-side C++ behind the C ABI (csrc/flac.cpp: a parser of
untrusted files; csrc/negidx.cpp: the MT19937 sampler and its worker thread), driven by tests/native/host_sanitize_driver.cpp
on the reference's FLAC fixtures (intact, truncated, bit-flipped) -- and a ThreadSanitizer build of the same driver for the
sampler's hand-over.  CPU only (sanitizers on the host code; no GPU sanitizer runs on this pool)."""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cpc2_amd", "csrc")
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "test_db", "**", "*.flac"), recursive=True))


def _build_and_run(tmp_path, name, flags, env):
    exe = str(tmp_path / name)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-pthread", *flags,
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "host_sanitize_driver.cpp"), os.path.join(CSRC, "flac.cpp"),
           os.path.join(CSRC, "negidx.cpp"), "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=600)
    res = subprocess.run([exe, str(tmp_path)] + FIXTURES, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, **env))
    assert res.returncode == 0, (res.stdout + res.stderr)[-4000:]
    assert "host sanitizer driver ok" in res.stdout
    return res.stdout


@pytest.mark.skipif(shutil.which("g++") is None or not FIXTURES, reason="needs g++ and the FLAC fixtures")
def test_flac_parser_and_sampler_under_asan_ubsan(tmp_path):
    out = _build_and_run(tmp_path, "drv_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"],
                         {"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert f"{len(FIXTURES)} fixtures decoded" in out


@pytest.mark.skipif(shutil.which("g++") is None or not FIXTURES, reason="needs g++ and the FLAC fixtures")
def test_sampler_worker_thread_under_tsan(tmp_path):
    _build_and_run(tmp_path, "drv_tsan", ["-fsanitize=thread"], {"TSAN_OPTIONS": "halt_on_error=1"})
