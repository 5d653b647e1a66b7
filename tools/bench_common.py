"""Shared pieces of bench.py and tools/bench_blocks.py: the workload constants, the synthetic members, the roofline arithmetic and the
two ways a step is timed (pipelined launch / finish through DistributedEnsemble; back-to-back launches issued from C)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

L, ALPHABET, F, H, K, M, BATCH = 8, "TGCA", 32, 100, 5, 3, 100_000
AAS = "ILVAGMFYWEDQNHCRKSTP"
PEAK_TF = 157.3                  # f32-input MFMA, dense (MI355X_MICROARCH.md)
MFMA_FLOP = 2048                 # one v_mfma_f32_16x16x4_f32: 16*16*4 MACs
MIN_TIMED_S = 0.5                # the settled figure covers at least this much GPU time
KINDS = {"cnn": 0, "mlp": 1, "ge": 2}


def _sig(x, digits=4):
    return None if x is None else float(f"{float(x):.{digits}g}")


def roofline_block(kind, Lx, A, Hx, Fx, Kx, members, n, kern_ms, kernel_name):
    """Both MFMA fractions of one scoring launch, from its measured duration.

    frac (= frac_issued)  MFMA instructions the launch really issues (fx_debug_mfma_per_tile: the kernels' loop bounds
                 restated on the host, x ceil(n/16) tiles x members) x 2048 FLOP / kernel_ms / peak: the physical
                 fraction of the matrix pipe, <= 1.  `achieved` is that rate in TFLOP/s.
    frac_algorithmic  ALGORITHMIC FLOP (2 x dense MACs x members x sequences, not discounted for
                 one-hot sparsity or 'same'-padding zeros) / kernel_ms / peak (`achieved_algorithmic`).  The kernels do
                 not issue those structural zeros, so this figure can exceed 1 on long launches (round-4 verdict:
                 a headline `frac` that can exceed 1 is not a roofline fraction -- it moved here)."""
    from flexs_amd import _native, synth

    macs = synth.algorithmic_macs(kind, Lx, A, Hx, Fx, Kx)
    flop = 2.0 * macs * members * n
    per_tile = _native.mfma_per_tile(KINDS[kind], Lx, A, Fx, Hx, Kx)
    issued = float(per_tile) * ((n + 15) // 16) * members * MFMA_FLOP
    ach_alg = flop / (kern_ms * 1e-3) / 1e12
    ach = issued / (kern_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": kernel_name, "achieved": ach, "peak": PEAK_TF, "unit": "TFLOP/s",
            "frac": ach / PEAK_TF, "frac_issued": ach / PEAK_TF,
            "achieved_algorithmic": ach_alg, "frac_algorithmic": ach_alg / PEAK_TF,
            "kernel_ms": kern_ms, "flop_per_launch": flop, "issued_flop_per_launch": issued,
            "mfma_per_tile": per_tile, "algorithmic_bytes_per_launch": (Lx + 4 * members) * n}


def build_members(kind, Lx, alphabet, members, device, Hx=H, Fx=F, Kx=K):
    """`members` surrogates of the product API with synthetic (Glorot + non-zero bias) weights, seeds 1000 + m."""
    from flexs_amd import synth
    from flexs_amd.baselines.models import CNN, MLP, GlobalEpistasisModel

    out = []
    for m in range(members):
        if kind == "cnn":
            mod = CNN(Lx, Fx, Hx, alphabet, kernel_size=Kx, device=device)
        elif kind == "mlp":
            mod = MLP(Lx, Hx, alphabet, device=device)
        else:
            mod = GlobalEpistasisModel(Lx, Hx, alphabet, device=device)
        mod.model.set_weights(synth.synthetic_weights(mod.model.shapes(), 1000 + m))
        out.append(mod)
    return out


def events_every(steps):
    """Which launches of a timed loop get a HIP event pair: every 4th for short loops, 16 samples for long ones.  A pair around EVERY
    launch cost the loop 7.8 us per step (192.7 vs 184.9 us, profiles/r6_bench_loop_events.log) -- the instrument was 4 % of the
    measurement -- and a pair around every 4th 1.6 us; the bracketed launches still read ~5 us long (event latency), which is why the
    report carries a back-to-back figure (one pair around many launches) beside them."""
    return max(4, steps // 16)


def run_pipelined(ens, d_seq, n, steps, warmup, torch, dist, use_dist, want_events=True):
    """W untimed + K timed steps of ens.launch / ens.finish, double-buffered: the gather of step k (communication
    stream) overlaps the scoring of step k + 1; barrier + synchronize on both sides of the timed region.
    Returns (elapsed_s, host_issue_s, kernel_ms)."""
    st = ens.stream

    def go(count, events):
        for i in range(count):
            ens.launch(d_seq, n, slot=i & 1, want="mean", timing=events[i] if events else None)
            if i:
                ens.finish((i - 1) & 1)
        if count:
            ens.finish((count - 1) & 1)

    go(warmup, None)
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    # want_events: True / 1 = an event pair around every launch, k > 1 = around every k-th (events_every), 0 = none
    every = int(want_events) if want_events else 0
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if i % every == 0 else None
              for i in range(steps)] if every else None
    t0 = time.perf_counter()
    go(steps, events)
    host_issue = time.perf_counter() - t0                # host time to ENQUEUE the K steps (GPU still running)
    st.synchronize()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ens._engine().sync()                                 # raises if any character was outside the alphabet
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)         # MAX over ranks
        elapsed = float(t.item())
    kern_ms = float(np.mean([ev[0].elapsed_time(ev[1]) for ev in events if ev])) if events else None
    return elapsed, host_issue, kern_ms


def time_launches(eng, models, d_ptr, n, Lx, lut, d_planes, stride, min_ms=60.0, reps0=50):
    """Mean duration of one scoring launch: one HIP event pair on the engine's stream around `reps` back-to-back
    launches issued from C (fx_debug_time_score -- Python cannot enqueue a ~15 us kernel fast enough to keep the GPU
    busy, and the idle gaps would be booked as kernel time), repeated until the bracket covers >= min_ms."""
    natives = [m.native() for m in models]
    eng.time_score_planes(natives, d_ptr, n, Lx, lut, d_planes.data_ptr(), stride, 20)      # warm-up
    reps = reps0
    while True:
        ms = eng.time_score_planes(natives, d_ptr, n, Lx, lut, d_planes.data_ptr(), stride, reps)
        if ms >= min_ms or reps >= 20000:
            return ms / reps, reps
        reps = int(min(20000, max(reps * 2, reps * min_ms / max(ms, 1e-3) * 1.1)))

