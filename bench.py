#!/usr/bin/env python3
"""Benchmark of the hot path: BASELINE.json configs[1] -- TF-binding L=8 (alphabet
TGCA), 3-member CNN(32,100,k=5) Ensemble, batch = 1e5 sequences per virtual-screen
call -- on N GPUs of one node (one process per GPU, weak scaling over sequences).

A step = one `Ensemble.get_fitness`-equivalent pass over one 1e5-sequence batch
that is already resident in HBM: the fused encode+CNN scoring kernel for all
three members, the ensemble mean kernel and, for N > 1, ONE RCCL all-gather of the
per-rank means (the north-star's exchange step).  Prints one JSON line on rank 0.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import faulthandler
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
faulthandler.enable()

L, ALPHABET, F, H, K, M, BATCH = 8, "TGCA", 32, 100, 5, 3, 100_000


def cpu_baseline(budget_s=12.0, hard_timeout_s=150.0, nam=False):
    """Reference-style CPU path (oracle/cpu_baseline_cli.py -> oracle/torch_twin.py),
    timed on a bounded sample of the same workload in a child process with a hard
    timeout, so that the bench line is always printed.  Checker code: used ONLY here."""
    import subprocess

    cmd = [sys.executable, "-m", "oracle.cpu_baseline_cli", "--L", str(L), "--alphabet", ALPHABET,
           "--filters", str(F), "--hidden", str(H), "--kernel", str(K), "--members", str(M),
           "--sample", "50000", "--budget", str(budget_s)] + (["--nam"] if nam else [])
    try:
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=hard_timeout_s)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001 - the GPU numbers must still be reported
        return {"value": None, "unit": "sequences/s", "cores": None, "kind": "port",
                "sample": f"cpu baseline failed: {type(e).__name__}: {str(e)[:200]}"}


def make_report(world, N, steps, warmup, elapsed, host_issue_s, kern_ms, use_dist):
    """The one JSON line of the bench contract (pure function: unit-tested on the CPU)."""
    from flexs_amd import synth

    macs = synth.algorithmic_macs("cnn", L, len(ALPHABET), H, F, K)
    flop_per_launch = 2.0 * macs * M * N                 # 2 x dense MACs x members x sequences
    peak = 157.3                                         # f32-input MFMA, MI355X_MICROARCH.md
    achieved = flop_per_launch / (kern_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
    if os.path.exists(tpath):
        traffic = json.load(open(tpath)).get("hbm_bytes_per_launch")
    return {
        "metric": "sequences scored/sec (virtual-screen batch)",
        "value": world * N * steps / elapsed,
        "unit": "sequences/s",
        "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "host_issue_ms_per_step": host_issue_s / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"TF-binding L={L} alphabet={ALPHABET}, {M}-member CNN(num_filters={F}, "
                               f"hidden_size={H}, kernel_size={K}) Ensemble, batch={N} virtual-screen per GPU "
                               "(BASELINE.json configs[1]); inputs resident in HBM; step = fused encode+CNN "
                               "scoring kernel + ensemble-mean kernel"
                               + (" + one RCCL all-gather of the per-rank means (own stream, overlapped with the "
                                  "next step's compute)" if use_dist else ""),
                   "global_batch": world * N, "seq_len": L, "members": M,
                   "parallelism": f"sequence-parallel x{world}" if world > 1 else "single GPU"},
        "roofline": {"bound": "mfma", "kernel": "k_score_cnn_mfma", "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "traffic": traffic,
                     "kernel_ms": kern_ms, "flop_per_launch": flop_per_launch,
                     "algorithmic_bytes_per_launch": (L + 4 * M) * N,
                     "note": "f32-input MFMA peak (157.3 TFLOP/s); algorithmic FLOP = 2*MACs, not discounted "
                             "for one-hot sparsity / zero padding; traffic = FETCH_SIZE x2 + WRITE_SIZE from the "
                             "PMC passes under profiles/"},
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 100 + 1000 launches of ~0.19 ms -- long enough for the clocks to settle (the first ~50 launches of a
    # cold process run ~10 % slower: profiles/r1_run22 trace), still a fraction of a second
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-nam", action="store_true",
                    help="cpu_baseline additionally times the NoisyAbstractModel CPU path (adds ~20 s)")
    ap.add_argument("--variant", type=int, default=0, help="cnn kernel variant (0 = auto)")
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="CUs left free for RCCL when running distributed; -1 = 4 when WORLD_SIZE > 1, else 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise RCCL and run the all-gather even with one rank (exercises the N>1 code path)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from flexs_amd import _native, synth
    from flexs_amd.baselines.models.keras_model import Architecture

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no HIP device visible); there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    saved_stdout = None
    if use_dist:
        # RCCL prints a five-line version banner on STDOUT when its first communicator comes up; the contract is one
        # JSON line there, so file descriptor 1 points at stderr until the line is printed
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # the exchanged messages are 4*N bytes per rank (0.4 MB): latency-bound.  Keep RCCL to a couple of
        # channels so its (overlapped) kernel does not take CUs away from the MFMA-bound scoring kernel.
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "2")
        os.environ.setdefault("NCCL_MIN_NCHANNELS", "1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    N = args.batch
    eng = _native.Engine.get(local_rank)
    if args.variant:
        eng.set_option("cnn_variant", args.variant)
    reserve = args.reserve_cus if args.reserve_cus >= 0 else (4 if world > 1 else 0)
    if use_dist and reserve > 0:
        # K1 is a persistent one-workgroup-per-CU kernel whose 16 waves x 128 VGPRs fill a CU's register file, so
        # RCCL's channel workgroups cannot co-reside with it: with every CU taken, the all-gather of step k would
        # start only when two K1 workgroups of step k+1 have been held back for it, stretching that launch by the
        # collective's latency.  Leaving a few CUs free (1.6 % of K1's throughput) lets it run next to step k+1.
        # (With one rank the collective is a copy, so this could not be measured on the 1-GPU boxes:
        # profiles/r1_run6_rccl_overlap_probe.md.)
        eng.set_option("grid_blocks", max(1, eng.get_option("num_cus") - reserve))
    stream = torch.cuda.Stream()
    lut = _native.make_lut(ALPHABET)
    arch = Architecture("cnn", L, len(ALPHABET), H, num_filters=F, kernel_size=K)
    weight_sets = [synth.synthetic_weights(arch.shapes(), 1000 + m) for m in range(M)]
    models = []
    for ws in weight_sets:
        nm = _native.NativeModel(eng, _native.FX_CNN, L, len(ALPHABET), F, H, K)
        nm.set_weights(ws)
        models.append(nm)
    seq_bytes = synth.random_sequence_bytes(N, L, ALPHABET, seed=rank)

    with torch.cuda.stream(stream):
        eng.set_stream(stream.cuda_stream)
        d_ascii = torch.from_numpy(seq_bytes).cuda()
        # two buffer sets: the all-gather of step k runs on RCCL's stream while step k+1 computes
        # the M members' scores as member-major planes (the layout fx_score_dev uses when only the mean is wanted)
        stride = (N + 63) // 64 * 64
        d_nm = [torch.empty((M, stride), dtype=torch.float32, device="cuda") for _ in range(2)]
        d_mean = [torch.empty((N,), dtype=torch.float32, device="cuda") for _ in range(2)]
        d_all = [torch.empty((world * N,), dtype=torch.float32, device="cuda") for _ in range(2)] if use_dist else None
        pending = [None, None]
        ev_a = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

        comm = torch.cuda.Stream() if use_dist else None          # the collective gets its own stream
        done = [torch.cuda.Event(), torch.cuda.Event()] if use_dist else None

        def step(i, k=None):
            b = i & 1
            if pending[b] is not None:
                stream.wait_event(pending[b])            # buffer set b is free again (stream-side wait, no host sync)
                pending[b] = None
            if k is not None:
                ev_a[k].record(stream)
            eng.score_planes_dev(models, d_ascii.data_ptr(), N, L, lut, d_nm[b].data_ptr(), stride)   # K1 fused encode+CNN x3
            if k is not None:
                ev_b[k].record(stream)
            eng.ensemble_mean_planes_dev(d_nm[b].data_ptr(), N, M, stride, d_mean[b].data_ptr())     # K3 np.mean order
            if use_dist:
                comm.wait_stream(stream)
                with torch.cuda.stream(comm):
                    dist.all_gather_into_tensor(d_all[b], d_mean[b])                         # RCCL over xGMI
                    done[b].record(comm)
                pending[b] = done[b]

        def drain():
            for b in (0, 1):
                if pending[b] is not None:
                    stream.wait_event(pending[b])
                    pending[b] = None

        for i in range(args.warmup):
            step(i)
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step(k, k)
        host_issue_s = time.perf_counter() - t0          # host time to ENQUEUE the K steps (GPU still running)
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        eng.sync()                                       # raises if any bad character was met
        # untimed completeness check: every score must have been written by the kernels (scores are
        # nan_to_num'ed, so a NaN that survives a step is an element nobody wrote)
        d_nm[0].fill_(float("nan"))
        d_mean[0].fill_(float("nan"))
        step(0)
        drain()
        torch.cuda.synchronize()
        assert not bool(torch.isnan(d_nm[0][:, :N]).any()) and not bool(torch.isnan(d_mean[0]).any()), "unwritten scores"
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev_a, ev_b)]))
        last = (args.steps - 1) & 1
        got_mean = d_mean[last].cpu().numpy()
        got_nm = d_nm[last][:, :N].t().contiguous().cpu().numpy()        # (N, M), i.e. np.stack(axis=1)
        if use_dist:
            gathered = d_all[last].cpu().numpy()
            assert np.array_equal(gathered[rank * N:(rank + 1) * N], got_mean), "all-gather lost this rank's shard"
        eng.set_stream(None)

    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        assert np.array_equal(np.mean(got_nm, axis=1), got_mean), "device mean is not np.mean bit-for-bit"
        out = make_report(world, N, args.steps, args.warmup, elapsed, host_issue_s, kern_ms, use_dist)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nam=args.cpu_nam)
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
