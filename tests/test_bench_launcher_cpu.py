"""`python bench.py --gpus N` without a launcher spawns its own ranks (bench.spawn_ranks).  On a
box without a GPU every rank fails at kkt_create ("no HIP device available": the product has no
CPU path); the parent must notice, stop the other ranks and exit non-zero -- not hang, not print
a line."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_self_launched_ranks_fail_loudly_without_a_gpu():
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        n = ctypes.c_int(0)
        if hip.hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value > 0:
            import pytest
            pytest.skip("a GPU is present: the failure path is not reachable")
    except OSError:
        pass
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "16",
                          "--n_t", "4", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
                          "--no-config4", "--launch-timeout", "240"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert time.time() - t0 < 240
    assert "exited with code" in out.stderr and "no HIP device" in out.stderr, out.stderr[-2000:]
    assert out.stdout.strip() == ""


def test_free_port_and_rank_environment():
    sys.path.insert(0, ROOT)
    import bench
    p = bench.free_port()
    assert 1024 < p < 65536
    # a fraction outside (0, 1] is flagged in the line instead of aborting the run
    roof = {"frac": 1.2}
    assert not bench.roofline_check(roof) and "error" in roof
    assert bench.roofline_check({"frac": 0.7})
