"""The row-task scheduler (x265-amod_amd/host/xa_fiber.cpp, csrc/xa_fiber.h): tasks with start conditions, tasks that park and are resumed (possibly by
another worker thread), two submitters at once.  Host code only: runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_tasks_park_and_resume(tmp_path):
    exe = str(tmp_path / "fiber_check")
    pkg = os.path.join(ROOT, "x265-amod_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-I", os.path.join(pkg, "csrc"), "-o", exe,
                           os.path.join(ROOT, "tests", "native", "fiber_check.cpp"), os.path.join(pkg, "host", "xa_fiber.cpp")])
    for workers in ("1", "3", "6"):
        r = subprocess.run([exe], env=dict(os.environ, X265AMD_WORKERS=workers), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and "ok sum" in r.stdout, r.stdout + r.stderr
