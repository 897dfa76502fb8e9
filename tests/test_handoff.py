"""The cross-thread hand-off of the libzl-named layer (libzl_amd/csrc/zl_handoff.h: the request queue play / stop / queue calls
post into and the parameter snapshots the setters publish) under ThreadSanitizer, on the CPU: four posting threads, two setter
threads, one draining "cycle" thread (tests/cpu_harness/handoff_tsan.cpp).  VERDICT r2 item 4: wait-free parameter edits must
stay clean under TSan."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_request_queue_and_parameter_snapshots_are_race_free_under_tsan(tmp_path):
    exe = str(tmp_path / "handoff_tsan")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", os.path.join(ROOT, "libzl_amd", "csrc"), "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpu_harness", "handoff_tsan.cpp"), "-lpthread", "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0 and ("tsan" in res.stderr.lower() or "sanitize" in res.stderr.lower()):
        pytest.skip("ThreadSanitizer runtime not available: " + res.stderr.strip().splitlines()[-1])
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    if run.returncode != 0 and "unexpected memory mapping" in run.stderr:
        # this gcc's ThreadSanitizer runtime cannot place its shadow under the kernel's address-space randomisation (seen on the GPU
        # boxes): once more without ASLR, or skip -- an environment limit, not a finding
        import shutil
        if shutil.which("setarch"):
            run = subprocess.run(["setarch", os.uname().machine, "-R", exe], capture_output=True, text=True, timeout=300, env=env)
        if run.returncode != 0 and ("unexpected memory mapping" in run.stderr or "setarch" in run.stderr):
            pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel: " + run.stderr.strip().splitlines()[-1])
    assert run.returncode == 0, run.stdout + run.stderr
    assert "ThreadSanitizer" not in run.stderr, run.stderr
    assert "out of order 0, torn 0" in run.stdout and "requests 80000 of 80000" in run.stdout, run.stdout
