"""bench.py's protection of the first multi-GPU run (no GPU needed): the watchdog that prints what was already measured when a
phase does not come back, and the self-launcher's time limit with one fresh retry."""
import os
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, env=None, timeout=120):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)


def test_watchdog_emits_the_prepared_record_and_ends_the_process():
    r = _run("""
        import sys, time
        sys.path.insert(0, ".")
        import bench
        bench.WATCHDOG["emit"] = lambda what: print("RECORD after [%s]" % what, flush=True)
        bench.watchdog_arm(1, "captured replay")
        time.sleep(30)          # the "hung" main thread
        print("not reached")
    """)
    assert r.returncode == 0, r.stderr
    assert "RECORD after [captured replay]" in r.stdout and "not reached" not in r.stdout
    assert "did not finish in time" in r.stderr


def test_watchdog_disarmed_or_without_a_record():
    r = _run("""
        import sys, time
        sys.path.insert(0, ".")
        import bench
        bench.WATCHDOG["emit"] = lambda what: print("RECORD")
        bench.watchdog_arm(1, "phase")
        bench.watchdog_disarm()
        time.sleep(2.5)
        print("finished normally")
    """)
    assert r.returncode == 0 and "finished normally" in r.stdout and "RECORD" not in r.stdout
    r = _run("""
        import sys, time
        sys.path.insert(0, ".")
        import bench
        bench.watchdog_arm(1, "nothing measured yet")
        time.sleep(30)
    """)
    assert r.returncode == 3   # nothing to report: a failure, not a fake record


def test_self_launch_kills_a_stuck_rank_tree_and_retries_once():
    t0 = time.time()
    r = _run("""
        import sys, types
        sys.path.insert(0, ".")
        import bench
        args = types.SimpleNamespace(gpus=2, launch_timeout=4)
        print("rc", bench.self_launch(args, ["--gpus", "2"]))
    """, env={"FHVAE_BENCH_TEST_SLEEP": "600"}, timeout=300)
    assert "rc 124" in r.stdout, (r.stdout, r.stderr[-2000:])
    assert "starting a fresh one with --no-dist-graph --no-alt" in r.stderr and "second attempt did not finish either" in r.stderr
    assert time.time() - t0 < 200
