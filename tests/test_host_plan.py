"""The host planner of librdgan_hip.so (csrc/rdgan_hostplan.h: geometry, gather plans, row tables, border-class boxes,
weight-gradient tilings, partial-slab workspace bound) compiled WITHOUT HIP by g++ under AddressSanitizer + UBSan and run on
the CPU: tests/host/plan_check.cpp builds every plan for ndomain {8,16,24,32,64,120} x {1,2,3} condition channels x max_batch
{1,9,96,256,2048} and checks that (1) every product the validity mask lets through reads and writes inside its tensors and
every destination row is written exactly once, (2) a border-class box plan computes exactly the parent plan's products,
(3) the tap lookup tables agree with the phases, (4) the box kernels' workgroup decode covers every partial slab, and the slab
need of EVERY batch size up to max_batch stays inside the bound the workspace is sized with."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_planner_under_sanitizers(tmp_path):
    exe = str(tmp_path / "plan_check")
    cmd = ["g++", "-std=c++17", "-O2", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-Wall", "-Wextra", "-Wno-unused-function", os.path.join(ROOT, "tests", "host", "plan_check.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    res = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "0 failures" in res.stdout and "runtime error" not in res.stderr and "AddressSanitizer" not in res.stderr, res.stderr[-4000:]
