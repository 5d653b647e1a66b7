#!/usr/bin/env python3
"""Kernel-level and end-to-end measurements over the BASELINE.json configs (GPU box only).

Writes gpurun_out/perf_survey.json + a markdown table.  Kernel times are HIP-event
brackets on the engine's stream around `reps` back-to-back launches with inputs
resident in HBM; end-to-end times are wall-clock through the Python API (host
strings in, host array out, i.e. PCIe + marshalling inclusive)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import flexs_amd  # noqa: E402
from flexs_amd import _native, synth  # noqa: E402
from flexs_amd.baselines import models as bm  # noqa: E402
from flexs_amd.baselines.models.keras_model import Architecture  # noqa: E402

AAS = "ILVAGMFYWEDQNHCRKSTP"
PEAK_TF, PEAK_HBM = 157.3, 8000.0
eng = _native.Engine.get(0)
rows = []


def natives(kind, L, A, H, M, F=0, K=0):
    arch = Architecture(kind, L, A, H, num_filters=F, kernel_size=K)
    out = []
    for m in range(M):
        nm = _native.NativeModel(eng, {"cnn": 0, "mlp": 1, "ge": 2}[kind], L, A, F, H, K)
        nm.set_weights(synth.synthetic_weights(arch.shapes(), 1000 + m))
        out.append(nm)
    return out


def time_score(kind, L, alpha, H, M, N, F=0, K=0, reps=20, generic=False, variant=0, label=None, opts=None):
    A = len(alpha)
    ms_ = natives(kind, L, A, H, M, F, K)
    lut = _native.make_lut(alpha)
    d_in = torch.from_numpy(synth.random_sequence_bytes(N, L, alpha, 0)).cuda()
    stride = (N + 63) // 64 * 64
    d_pl = torch.empty((M, stride), dtype=torch.float32, device="cuda")      # member-major planes (the engine's own layout)
    torch.cuda.synchronize()
    eng.set_option("force_generic", int(generic))
    eng.set_option("cnn_variant", variant)
    for k_, v_ in (opts or {}).items():
        eng.set_option(k_, v_)
    # launches issued from C (fx_debug_time_score): a Python loop cannot keep the GPU busy with < 25 us kernels
    eng.time_score_planes(ms_, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride, 3)
    ms = eng.time_score_planes(ms_, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride, reps) / reps
    eng.set_option("force_generic", 0)
    eng.set_option("cnn_variant", 0)
    for k_ in (opts or {}):
        eng.set_option(k_, {"cnn_pair": 1, "cnn_big_units": 12, "cnn_seg": -1, "cnn_pair_seg": -1, "dense_slab": 1, "ge_bytetab": 1, "mlp_pair": 1,
                               "wave_prio": 1, "stage_bytes": 1, "stage_fill": 1, "cnn_quad": 1, "dma_fill": 1, "cnn_seg_multi": 1, "cnn_pair_seg4": 1, "dense_small": 1, "dense_coop": 1, "quad_rotate": 1, "serve_small": 1}.get(k_, 0))
    macs = synth.algorithmic_macs(kind, L, A, H, F, K)
    tf = 2.0 * macs * M * N / (ms * 1e-3) / 1e12
    gbs = (L + 4 * M) * N / (ms * 1e-3) / 1e9
    rows.append({"what": label or f"{kind} L={L} A={A} H={H} M={M} N={N}" + (" [generic]" if generic else ""),
                 "kernel_ms": ms, "seq_per_s": N / (ms * 1e-3), "alg_TFLOPs": tf, "frac_mfma_peak": tf / PEAK_TF,
                 "alg_GBs": gbs, "macs_per_seq_member": macs})
    print(rows[-1], flush=True)


def time_hbm_kernels():
    for (N, L, alpha) in ((2_000_000, 8, "TGCA"), (100_000, 237, AAS)):
        A = len(alpha)
        d_in = torch.from_numpy(synth.random_sequence_bytes(N, L, alpha, 0)).cuda()
        d_out = torch.empty((N, L, A), dtype=torch.float32, device="cuda")
        lut = _native.make_lut(alpha)
        eng.encode_onehot_dev(d_in.data_ptr(), N, L, lut, A, d_out.data_ptr()); eng.sync()
        eng.timer_start()
        for _ in range(20):
            eng.encode_onehot_dev(d_in.data_ptr(), N, L, lut, A, d_out.data_ptr())
        ms = eng.timer_stop() / 20
        b = N * (L + 4 * L * A)
        rows.append({"what": f"encode_onehot N={N} L={L} A={A}", "kernel_ms": ms, "alg_GBs": b / (ms * 1e-3) / 1e9,
                     "frac_hbm_peak": b / (ms * 1e-3) / 1e9 / PEAK_HBM, "seq_per_s": N / (ms * 1e-3)})
        print(rows[-1], flush=True)
    for (N, M) in ((10_000_000, 3), (10_000_000, 8), (100_000, 3)):
        d_in = torch.rand((N, M), dtype=torch.float32, device="cuda")
        d_out = torch.empty((N,), dtype=torch.float32, device="cuda")
        eng.ensemble_reduce_dev(d_in.data_ptr(), N, M, d_out.data_ptr()); eng.sync()
        eng.timer_start()
        for _ in range(20):
            eng.ensemble_reduce_dev(d_in.data_ptr(), N, M, d_out.data_ptr())
        ms = eng.timer_stop() / 20
        b = N * (4 * M + 4)
        rows.append({"what": f"ensemble_mean N={N} M={M}", "kernel_ms": ms, "alg_GBs": b / (ms * 1e-3) / 1e9,
                     "frac_hbm_peak": b / (ms * 1e-3) / 1e9 / PEAK_HBM, "seq_per_s": N / (ms * 1e-3)})
        print(rows[-1], flush=True)


def end_to_end():
    L, alpha, M = 8, "TGCA", 3
    members = [bm.CNN(L, 32, 100, alpha, seed=m) for m in range(M)]
    ens = flexs_amd.Ensemble(members)
    for N in (100_000,):
        b = synth.random_sequence_bytes(N, L, alpha, 1)
        seqs = synth.bytes_to_strings(b)
        arr_s = np.array(seqs, dtype="S")
        arr_u = np.array(seqs)
        for name, inp in (("list[str]", seqs), ("ndarray dtype=S", arr_s), ("ndarray dtype=U", arr_u)):
            ens.get_fitness(inp)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter(); ens.get_fitness(inp); ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            rows.append({"what": f"end-to-end Ensemble(3xCNN).get_fitness({name}) N={N} L=8 (PCIe + marshalling inclusive)",
                         "wall_ms": t * 1e3, "seq_per_s": N / t})
            print(rows[-1], flush=True)
        t0 = time.perf_counter()
        for _ in range(5):
            _native.sequences_to_bytes(seqs, L=L)
        rows.append({"what": f"host marshalling only: list[str] -> (N,L) uint8, N={N}", "wall_ms": (time.perf_counter() - t0) / 5 * 1e3})
        print(rows[-1], flush=True)
    # small-call latency: what Adalead (<=20), CbAS (100), DynaPPO (4), CMA-ES (1), Random (2001) issue
    for N in (1, 4, 20, 100, 2001):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, alpha, 2))
        for _ in range(20):
            ens.get_fitness(seqs)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        rows.append({"what": f"small call Ensemble(3xCNN).get_fitness N={N}", "median_us": float(np.median(ts)) * 1e6,
                     "p90_us": float(np.percentile(ts, 90)) * 1e6, "seq_per_s": N / float(np.median(ts))})
        print(rows[-1], flush=True)


def population():
    """CMA-ES iteration (cmaes.py:83-108): P solutions decoded and scored, one by one vs one fused round trip."""
    from flexs_amd.utils.population import PopulationEvaluator
    from flexs_amd.utils import sequence_utils as s_utils

    rng = np.random.default_rng(0)
    for L, alpha, P in ((8, "TGCA", 16), (8, "TGCA", 40), (237, s_utils.AAS, 40)):
        members = [bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)]
        ens = flexs_amd.Ensemble(members)
        ev = PopulationEvaluator(ens, alpha, L)
        x = rng.standard_normal((P, L * len(alpha)))
        for _ in range(5):
            ev.evaluate(x)
        ts = []
        for _ in range(100):
            t0 = time.perf_counter(); ev.evaluate(x); ts.append(time.perf_counter() - t0)
        fused = float(np.median(ts))
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            for r in x:                                    # the reference loop: decode on the host, one call per solution
                oh = np.zeros((L, len(alpha)))
                oh[np.arange(L), np.argmax(r.reshape(L, len(alpha)), axis=1)] = 1
                ens.get_fitness([s_utils.one_hot_to_string(oh, alpha)]).item()
            ts.append(time.perf_counter() - t0)
        loop = float(np.median(ts))
        rows.append({"what": f"population step P={P} L={L} A={len(alpha)} Ensemble(3xCNN): fused decode+score vs {P} single calls",
                     "fused_us": fused * 1e6, "one_by_one_us": loop * 1e6, "speedup": loop / fused})
        print(rows[-1], flush=True)


def protein_small_calls():
    """Latency of small calls on the GFP-length CNN ensemble, whole-sequence vs position-segmented pair kernel."""
    from flexs_amd.utils import sequence_utils as s_utils

    L, alpha = 237, s_utils.AAS
    ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, alpha, seed=m) for m in range(3)])
    for N in (1, 4, 16, 40, 100, 400):
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, alpha, 3))
        for seg, name in ((0, "whole-sequence form"), (-1, "segmented form (auto)")):
            eng.set_option("cnn_pair_seg", seg)
            for _ in range(3):
                ens.get_fitness(seqs)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
            rows.append({"what": f"small call Ensemble(3xCNN L=237 A=20).get_fitness N={N}, {name}",
                         "median_us": float(np.median(ts)) * 1e6, "seq_per_s": N / float(np.median(ts))})
            print(rows[-1], flush=True)
    eng.set_option("cnn_pair_seg", -1)
    for L in (50, 100):
        ens = flexs_amd.Ensemble([bm.CNN(L, 32, 100, "UGCA", seed=m) for m in range(3)])
        for N in (1, 20, 100):
            seqs = synth.bytes_to_strings(synth.random_sequence_bytes(N, L, "UGCA", 3))
            for seg, name in ((0, "one wave per tile"), (-1, "waves split the positions (auto)")):
                eng.set_option("cnn_seg", seg)
                for _ in range(5):
                    ens.get_fitness(seqs)
                ts = []
                for _ in range(100):
                    t0 = time.perf_counter(); ens.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                rows.append({"what": f"small call Ensemble(3xCNN L={L} A=4).get_fitness N={N}, {name}",
                             "median_us": float(np.median(ts)) * 1e6, "seq_per_s": N / float(np.median(ts))})
                print(rows[-1], flush=True)
    eng.set_option("cnn_seg", -1)


def nam():
    rng = np.random.default_rng(0)
    for (L, nsym, Q, C) in ((14, 4, 100, 100), (14, 4, 100, 1000), (14, 4, 100, 20000), (14, 4, 2000, 20000),
                            (8, 4, 2000, 20000), (90, 20, 100, 20000), (238, 20, 100, 2000)):
        cache = rng.integers(65, 65 + nsym, (C, L)).astype(np.uint8)
        q = rng.integers(65, 65 + nsym, (Q, L)).astype(np.uint8)
        dc = _native.NativeCache(eng, L)
        dc.append(cache)
        for mode, mname in ((0, "levenshtein"), (1, "hamming")):
            dc.min_dist(q, mode)
            ts = []
            for _ in range(10):
                t0 = time.perf_counter(); dc.min_dist(q, mode); ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            rows.append({"what": f"min_dist {mname} L={L} Q={Q} C={C} (host queries in, results out)", "wall_ms": t * 1e3,
                         "pair_evals_per_s": Q * C / t, "queries_per_s": Q / t})
            print(rows[-1], flush=True)

    class Table(flexs_amd.Landscape):
        def __init__(self):
            super().__init__("table")

        def _fitness_function(self, seqs):
            return np.array([(hash(str(s)) % 1000) / 1000.0 for s in seqs])

    np.random.seed(0)
    model = bm.NoisyAbstractModel(Table(), 0.9)
    alpha, L = "UGCA", 14
    seqs0 = synth.bytes_to_strings(synth.random_sequence_bytes(1000, L, alpha, 5))
    model.train(seqs0, np.random.random(1000))
    t0 = time.perf_counter()
    total = 0
    for call in range(20):                                     # CbAS pattern: 100 / call, 20 calls
        batch = synth.bytes_to_strings(synth.random_sequence_bytes(100, L, alpha, 100 + call))
        model.get_fitness(batch); total += 100
    t = time.perf_counter() - t0
    rows.append({"what": "NoisyAbstractModel(ss=.9).get_fitness, RNA L=14, cache 1000 -> 3000, 20 calls x 100 (CbAS pattern)",
                 "wall_ms": t * 1e3, "seq_per_s": total / t})
    print(rows[-1], flush=True)


def main():
    which = sys.argv[1:] or ["score", "sweep", "hbm", "e2e", "nam", "population", "protein"]
    if "score" in which:
        for v in (1, 2, 3, 4):
            time_score("cnn", 8, "TGCA", 100, 3, 100_000, 32, 5, variant=v, label=f"C2 cnn L=8 M=3 N=1e5 variant {v} conv1=gather")
            time_score("cnn", 8, "TGCA", 100, 3, 100_000, 32, 5, variant=v, label=f"C2 cnn L=8 M=3 N=1e5 variant {v} conv1=mfma",
                       opts={"cnn_conv1_mfma": 1})
        time_score("cnn", 8, "TGCA", 100, 1, 10_000, 32, 5, label="C1 cnn L=8 M=1 N=1e4")
        time_score("cnn", 8, "TGCA", 100, 1, 100_000, 32, 5)
        time_score("cnn", 8, "TGCA", 100, 3, 1_000_000, 32, 5, reps=5)
        time_score("cnn", 8, "TGCA", 100, 8, 100_000, 32, 5, reps=5)
        time_score("cnn", 8, "TGCA", 100, 3, 100_000, 32, 5, reps=2, generic=True)
        time_score("cnn", 14, "UGCA", 100, 1, 100_000, 32, 5)
        time_score("cnn", 50, "UGCA", 100, 1, 100_000, 32, 5, reps=5)
        time_score("cnn", 100, "UGCA", 100, 1, 100_000, 32, 5, reps=3)
        time_score("mlp", 14, "UGCA", 100, 1, 100_000, label="C3 mlp L=14 H=100 M=1 N=1e5 l1=gather")
        time_score("mlp", 14, "UGCA", 100, 1, 100_000, label="C3 mlp L=14 H=100 M=1 N=1e5 l1=mfma", opts={"mlp_l1_mfma": 1})
        time_score("mlp", 90, AAS, 100, 1, 100_000, reps=5, label="mlp L=90 A=20 H=100 M=1 N=1e5 (layer-1 rows gathered from L2)")
        time_score("cnn", 237, AAS, 100, 1, 16_384, 32, 5, reps=2, label="C5 cnn L=237 A=20 M=1 N=16384 single-wave form", opts={"cnn_pair": 0})
        time_score("cnn", 237, AAS, 100, 3, 65_536, 32, 5, reps=1, label="C5 cnn L=237 A=20 M=3 N=65536 (pair form)")
        time_score("mlp", 14, "UGCA", 100, 1, 1_000_000, reps=5)
        time_score("mlp", 14, "UGCA", 100, 1, 100_000, reps=2, generic=True)
        time_score("ge", 90, AAS, 100, 8, 100_000, label="C4 ge L=90 A=20 H=100 M=8 N=1e5")
        time_score("ge", 90, AAS, 100, 8, 1_000_000, reps=5)
        time_score("ge", 90, AAS, 100, 1, 100_000)
        time_score("ge", 90, AAS, 100, 8, 100_000, reps=2, generic=True)
        time_score("cnn", 90, AAS, 100, 1, 16_384, 32, 5, reps=3)
        time_score("mlp", 14, "UGCA", 200, 1, 100_000, label="mlp L=14 H=200 M=1 N=1e5 (HxH blocks streamed from L2)")
        time_score("mlp", 14, "UGCA", 64, 1, 100_000, label="mlp L=14 H=64 M=1 N=1e5")
        time_score("cnn", 8, "TGCA", 64, 3, 100_000, 32, 5, label="cnn L=8 H=64 M=3 N=1e5")
        time_score("cnn", 8, "TGCA", 200, 3, 100_000, 32, 5, label="cnn L=8 H=200 M=3 N=1e5 (dense head from L2)")
        time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 3, label="cnn L=14 kernel_size=3 M=3 N=1e5")
        time_score("cnn", 14, "UGCA", 100, 3, 100_000, 32, 7, label="cnn L=14 kernel_size=7 M=3 N=1e5")
        time_score("cnn", 14, "UGCA", 100, 3, 100_000, 24, 5, label="cnn L=14 num_filters=24 M=3 N=1e5")
        time_score("cnn", 14, "UGCA", 50, 3, 100_000, 32, 3, label="cnn L=14 kernel_size=3 hidden=50 M=3 N=1e5 (conv + head kernels)")
        time_score("cnn", 14, "UGCA", 200, 3, 100_000, 32, 4, label="cnn L=14 kernel_size=4 hidden=200 M=3 N=1e5 (conv + head kernels)")
        time_score("cnn", 237, AAS, 100, 1, 16_384, 32, 5, reps=2, label="C5 cnn L=237 A=20 M=1 N=16384")
        time_score("cnn", 237, AAS, 100, 3, 16_384, 32, 5, reps=1, label="C5 cnn L=237 A=20 M=3 N=16384")
    if "sweep" in which:
        for N in (1_000, 10_000, 30_000, 100_000, 300_000, 1_000_000, 10_000_000):
            time_score("cnn", 8, "TGCA", 100, 3, N, 32, 5, reps=(20 if N <= 1_000_000 else 3), label=f"sweep C2 cnn L=8 M=3 N={N}")
    if "hbm" in which:
        time_hbm_kernels()
    if "e2e" in which:
        end_to_end()
    if "nam" in which:
        nam()
    if "population" in which:
        population()
    if "protein" in which:
        protein_small_calls()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "perf_survey.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
