"""GPU tests of the spin-wait host-call forms on a hostile host: kernels that were enqueued before their rows were
packed (launch_first) and the relay through member 0's workgroups (launch_relay) poll memory the host -- or another workgroup -- fills.
Round 5 tested their liveness on an idle 256-core box with one client.  Here the host is made to misbehave; every call must return the
packed-first bits or raise, and the process must end (each scenario is a child under a timeout: a hang fails the test, not the suite).

The scenarios live in tools/runs/r6_hostile_host.py (also run stand-alone for profiles/r6_hostile_host.log)."""
import json
import os
import random
import signal
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "runs", "r6_hostile_host.py")


def _run(mode, seconds, stopper=None, timeout=240):
    p = subprocess.Popen([sys.executable, SCRIPT, mode, str(seconds)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT)
    try:
        if stopper is not None:
            # wait for the child's READY line (models built, reference bits taken), then misbehave
            line = p.stdout.readline()
            while line and not line.startswith(("READY", "{")):
                line = p.stdout.readline()
            if line.startswith("READY"):
                stopper(p)
            head = line if line.startswith("{") else ""
        else:
            head = ""
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        pytest.fail(f"mode {mode}: the process did not end within {timeout} s (a hang)\n{err[-1500:]}")
    assert p.returncode == 0, err[-2000:]
    lines = [ln for ln in (head + out).splitlines() if ln.startswith("{")]
    assert lines, (out, err[-1000:])
    d = json.loads(lines[-1])
    if "skipped" in d:
        pytest.skip(d["skipped"])
    return d


def _check(d, min_calls):
    assert d["calls"] >= min_calls, d
    assert d["mismatches"] == 0, d                                # the packed-first bits, call after call
    assert d["n_errors"] == 0, d                                  # (a raise would be legal; none is expected: every string can be packed)
    assert d["launch_first_calls"] >= 1, d                        # the forms under test did run
    return d


def test_idle_host_reference_row():
    d = _check(_run("idle", 2.0), 50)
    assert d["launch_first_redone"] <= 1, d                       # an idle host packs in time


def test_packing_threads_pinned_to_one_busy_core():
    """(a) the caller, the packing pool and a busy-loop neighbour all share ONE core: packing is slow and pre-empted; the waves wait
    (0.25 s bound) or starve and the call is redone -- never a hang, never other bits."""
    d = _check(_run("oversubscribed", 3.0), 5)
    assert d["cpus_allowed"] == 1, d
    assert d["launch_first_redone"] <= d["launch_first_calls"], d


def test_process_stopped_for_300_ms_mid_call():
    """(b) SIGSTOP for 300 ms at random moments while the calls run (stop signals freeze every thread of the process: the caller and all
    packing threads stall mid-call, longer than the kernels' 0.25 s starvation bound); the call a stop lands in is run once more
    (`launch_first_redone`) and still gives the packed-first bits."""
    def stopper(p):
        rng = random.Random(3)
        t_end = time.time() + 4.0
        while time.time() < t_end and p.poll() is None:
            time.sleep(rng.uniform(0.05, 0.25))
            os.kill(p.pid, signal.SIGSTOP)                        # (the exact child started above)
            time.sleep(0.3)
            os.kill(p.pid, signal.SIGCONT)

    d = _check(_run("stopped", 5.0, stopper=stopper), 20)
    assert d["call_ms_p50_p90_p99_max"][3] >= 250.0, d            # at least one call did contain a stop
    assert d["launch_first_redone"] >= 1, d                       # ... and was accounted for as a redone launch


def test_foreign_stream_of_kernels_during_the_calls():
    """(c) another thread keeps a torch stream full of matrix kernels: the persistent launches share the CUs with foreign work (and, in
    a relay, member 0's workgroups may get onto the device after their readers)."""
    d = _check(_run("foreign", 3.0), 20)
    assert d["launch_relay_calls"] >= 1, d
