#!/usr/bin/env python3
"""Round 6: the spin-wait forms of the host calls -- kernels enqueued before their rows are packed
(launch_first), the relay through member 0's workgroups (launch_relay) -- on a HOSTILE host.  One process, one mode:

    oversubscribed   the whole process (Python thread + the packing pool) pinned to ONE core that a busy-loop process also sits on
    stopped          the parent SIGSTOPs this process for 300 ms at random moments while it issues calls (stop signals are
                     process-wide: every packing thread and the caller freeze mid-call; the kernels' 0.25 s starvation bail-out fires)
    foreign          a second thread keeps a foreign torch stream full of kernels during the calls
    idle             none of the above (the reference row)

Every call's result is compared with the packed-first bits (launch_first = 0) computed up front.  Prints ONE JSON object: calls,
mismatches, exceptions, launch_first_calls / launch_first_redone / launch_relay_calls deltas, call-time percentiles.  Must never hang:
the caller (tests/test_gpu_hostile_host.py) runs it under a timeout.

    python tools/runs/r6_hostile_host.py MODE [seconds]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

AAS = "ILVAGMFYWEDQNHCRKSTP"


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "idle"
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
    import torch

    import flexs_amd
    from flexs_amd import _native, synth
    from flexs_amd.baselines import models as bm

    eng = _native.Engine.get(0)
    if not eng.get_option("large_bar"):
        print(json.dumps({"mode": mode, "skipped": "no large BAR: launched-first calls are not offered"}))
        return 0
    work = []
    for kind, L, alpha, M, n in (("cnn", 8, "TGCA", 3, 70_001), ("mlp", 14, "UGCA", 1, 60_000), ("ge", 90, AAS, 8, 40_003)):
        make = {"cnn": lambda s: bm.CNN(L, 32, 100, alpha, seed=s), "mlp": lambda s: bm.MLP(L, 100, alpha, seed=s),
                "ge": lambda s: bm.GlobalEpistasisModel(L, 100, alpha, seed=s)}[kind]
        members = [make(s) for s in range(M)]
        model = members[0] if M == 1 else flexs_amd.Ensemble(members)
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, alpha, 5))
        eng.set_option("launch_first", 0)
        eng.set_option("launch_relay", 0)
        want = np.asarray(model.get_fitness(seqs)).copy()
        eng.set_option("launch_first", 1)
        eng.set_option("launch_relay", 1)
        model.get_fitness(seqs)                                  # (warm: staging areas, pools)
        work.append((f"{M}x{kind} L={L} n={n}", model, seqs, want))

    c0 = {k: eng.get_option(k) for k in ("launch_first_calls", "launch_first_redone", "launch_relay_calls")}
    stop = threading.Event()
    helpers = []
    if mode == "oversubscribed":
        cpu = sorted(os.sched_getaffinity(0))[-1]
        os.sched_setaffinity(0, {cpu})                           # this thread ...
        for t in os.listdir("/proc/self/task"):                  # ... and every thread already started (the packing pool, HIP's)
            try:
                os.sched_setaffinity(int(t), {cpu})
            except OSError:
                pass
        import subprocess

        busy = ("import os, sys, time\nos.sched_setaffinity(0, {int(sys.argv[1])})\nend = time.time() + float(sys.argv[2])\nx = 0\n"
                "while time.time() < end:\n    x += 1\n")
        pid = subprocess.Popen([sys.executable, "-c", busy, str(cpu), str(seconds + 30)]).pid   # the busy neighbour on the same core (no HIP)
        helpers.append(pid)
    elif mode == "foreign":
        def flood():
            torch.cuda.set_device(0)
            st = torch.cuda.Stream()
            a = torch.randn(2048, 2048, device="cuda")
            with torch.cuda.stream(st):
                while not stop.is_set():
                    for _ in range(8):
                        a = torch.tanh(a @ a * 1e-3)             # ~1 ms of foreign matrix work per iteration, every CU
                    st.synchronize()
        th = threading.Thread(target=flood, daemon=True)
        th.start()
        time.sleep(0.2)

    if mode == "stopped" and _native._strpack is not None:
        # ONE packing thread: packing is then about half of a call, so that a stop at a random moment lands where it matters -- kernels
        # launched, rows not all packed -- often enough for the test to insist on a redone launch (with the pool it is ~15 % of a call)
        _native._strpack.set_threads(1)
    print("READY", flush=True)                                   # (the parent of mode `stopped` starts its SIGSTOPs now)
    calls = mismatches = 0
    errors, times = [], []
    t_end = time.time() + seconds
    while time.time() < t_end:
        for name, model, seqs, want in work:
            t0 = time.perf_counter()
            try:
                got = np.asarray(model.get_fitness(seqs))
                if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                    mismatches += 1
            except Exception as ex:  # noqa: BLE001 -- "returns the packed-first bits or raises": a raise is recorded, not fatal
                errors.append(f"{name}: {type(ex).__name__}: {ex}"[:200])
            times.append(time.perf_counter() - t0)
            calls += 1
    stop.set()
    for pid in helpers:
        try:
            os.kill(pid, 9)
            os.waitpid(pid, 0)
        except OSError:
            pass
    c1 = {k: eng.get_option(k) for k in c0}
    ms = np.array(times) * 1e3
    print(json.dumps({"mode": mode, "seconds": seconds, "calls": calls, "mismatches": mismatches, "errors": errors[:5], "n_errors": len(errors),
                      **{k: int(c1[k] - c0[k]) for k in c0},
                      "call_ms_p50_p90_p99_max": [float(np.percentile(ms, q)) for q in (50, 90, 99)] + [float(ms.max())],
                      "cpus_allowed": len(os.sched_getaffinity(0))}), flush=True)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
