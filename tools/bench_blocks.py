"""The verbose measurement blocks of bench.py (one GPU, rank 0): every BASELINE.json config's kernel time and MFMA fractions, the
NoisyAbstractModel half of configs[2], `get_fitness(list[str])` end to end, the explorer call patterns and the explorer round, the
member-parallel split.  They go to the FULL record (stderr + gpurun_out/bench_full.json); bench.contract_line() copies a bounded set of
scalars out of them into the one contract line on stdout."""
import json
import os
import sys
import time

import numpy as np

from tools.bench_common import (AAS, ALPHABET, BATCH, F, H, K, KINDS, L, M, PEAK_TF, ROOT, _sig, build_members, roofline_block,
                                run_pipelined, time_launches)


def configs_block(eng, device, torch):
    """Kernel time and both MFMA fractions for the BASELINE.json configs that are not the headline, at the sizes the
    judge named: C1 (1 CNN, L=8, N=1e4), C2 at N=1e4 (3 CNN), C3 (MLP L=14, N=1e5), C4 (8 x GE L=90 A=20, N=1e5),
    C5 (3 x CNN L=237 A=20, one GPU's 62 500-row share of the 5e5 batch)."""
    from flexs_amd import _native, synth

    specs = [
        ("C1 cnn L=8 A=4 M=1 N=1e4", "cnn", 8, "TGCA", 1, 10_000, "k_score_cnn_mfma"),
        ("C2 cnn L=8 A=4 M=3 N=1e4", "cnn", 8, "TGCA", 3, 10_000, "k_score_cnn_mfma"),
        ("C3 mlp L=14 A=4 H=100 M=1 N=1e5", "mlp", 14, "UGCA", 1, 100_000, "k_score_dense_mfma<MLP>"),
        ("C4 ge L=90 A=20 H=100 M=8 N=1e5", "ge", 90, AAS, 8, 100_000, "k_score_dense_mfma<GE>"),
        ("C5 cnn L=237 A=20 M=3 N=62500 (one GPU's share of 5e5)", "cnn", 237, AAS, 3, 62_500, "k_score_cnn_pair"),
        # wide hidden layers (round-4 verdict item 4): H = 200 is DynaPPOEnsemble's default MLP member (dyna_ppo.py:52-55) and the Tutorial's
        ("survey mlp H200 L14 N1e5", "mlp", 14, "UGCA", 1, 100_000, "k_score_dense_mfma<MLP>", 200),
        ("survey cnn H200 L8 N1e5", "cnn", 8, "TGCA", 1, 100_000, "k_score_cnn_mfma", 200),
        ("survey ge M1 L90 N1e5", "ge", 90, AAS, 1, 100_000, "k_score_dense_mfma<GE>", 100),
        # the same member on a protein landscape (AAV: 90 residues x 20 letters): first-layer rows do not fit LDS (position-major first layer, round 6)
        ("survey mlp H200 L90 A20 N1e5", "mlp", 90, AAS, 1, 100_000, "k_mlp_l1_pos + k_score_dense_mfma<MLP>", 200),
    ]
    out = {}
    for spec in specs:
        name, kind, Lx, alpha, members, n, kname = spec[:7]
        Hx = spec[7] if len(spec) > 7 else H
        mods = build_members(kind, Lx, alpha, members, device, Hx=Hx)
        d_in = torch.from_numpy(synth.random_sequence_bytes(n, Lx, alpha, 0)).cuda()
        stride = (n + 63) // 64 * 64
        d_planes = torch.empty((members, stride), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        ms, reps = time_launches(eng, mods, d_in.data_ptr(), n, Lx, mods[0]._lut, d_planes, stride)
        Fx, Kx = (F, K) if kind == "cnn" else (0, 0)
        blk = roofline_block(kind, Lx, len(alpha), Hx, Fx, Kx, members, n, ms, kname)
        blk["seq_per_s"] = n / (ms * 1e-3)
        blk["reps"] = reps
        out[name] = blk
        del mods, d_in, d_planes
    return out

VALU_LANE_OPS = 256 * 4 * 16 * 2.4e9        # 256 CUs x 4 SIMDs x 16 lanes per cycle x 2.4 GHz = 3.93e13 lane-ops/s
K4_LANE_OPS_PER_CHAR = 24.5                 # issued VALU lane-ops per (pair, text character), L <= 32: PMC, profiles/archive/r2_run1_pmc_targets.md


def nam_block(eng, device):
    """configs[2]'s other half: NoisyAbstractModel (noisy_abstract_model.py:42-101) on RNA L=14.
    K4 (bit-parallel Levenshtein + first-arg-min) kernel time for Q=2000 uncached queries against C in {1e2, 1e3, 2e4}
    cached sequences, launches issued from C (fx_debug_time_min_dist): pair evaluations/s and the fraction of the
    integer-VALU issue rate (the roofline that binds K4: cache rows are L2-resident, HBM traffic ~ 0);
    and `NoisyAbstractModel.get_fitness` end to end on the CbAS call pattern (20 calls x 100 sequences, cache 1000 -> 3000)
    for a plain landscape (2 oracle calls + 1 RNG draw per query from a Python loop, as the reference) and a
    `batch_safe` one (two batched oracle calls)."""
    import flexs_amd
    from flexs_amd import _native, synth
    from flexs_amd.baselines.models import NoisyAbstractModel

    Lx, alpha, Q = 14, "UGCA", 2000
    out = {"k4": {}}
    q = synth.random_sequence_bytes(Q, Lx, alpha, 77)
    for C_ in (100, 1000, 20000):
        cache = _native.NativeCache(eng, Lx)
        cache.append(synth.random_sequence_bytes(C_, Lx, alpha, 78))
        reps = 20
        while True:
            ms = cache.time_min_dist(q, _native.FX_LEVENSHTEIN, reps)
            if ms >= 40.0 or reps >= 20000:
                break
            reps = int(min(20000, max(reps * 2, reps * 40.0 / max(ms, 1e-3) * 1.1)))
        t = ms / reps * 1e-3
        pairs = Q * C_
        lane_ops = K4_LANE_OPS_PER_CHAR * Lx * pairs
        out["k4"][f"L=14 Q=2000 C={C_}"] = {
            "kernel_ms": t * 1e3, "pair_evals_per_s": pairs / t, "queries_per_s": Q / t,
            "roofline": {"bound": "valu-int", "achieved": lane_ops / t / 1e12, "peak": VALU_LANE_OPS / 1e12,
                         "unit": "T lane-ops/s", "frac": lane_ops / t / VALU_LANE_OPS,
                         "lane_ops_per_pair_char": K4_LANE_OPS_PER_CHAR, "traffic": None},
            "workgroups": -(-C_ // 1024) * Q, "reps": reps}
        del cache
    out["k4"]["note"] = ("frac = 24.5 issued VALU lane-ops per (pair, text character) [PMC, profiles/archive/r2_run1_pmc_targets.md] x L x "
                         "pairs / kernel time / (256 CU x 4 SIMD x 16 lanes x 2.4 GHz); one workgroup = one query x <= 1024 "
                         "cache rows, so C = 100 runs 100 of 256 lanes per workgroup")

    class _Synth(flexs_amd.Landscape):
        """Deterministic table-like oracle: fitness = hash of the bytes in [0, 1) (ViennaRNA is absent)."""

        def __init__(self, batch_safe):
            super().__init__("synth")
            self.batch_safe = batch_safe
            self._w = (np.arange(1, Lx + 1, dtype=np.int64) * 2654435761) % 1000003

        def _fitness_function(self, seqs):
            b = _native.sequences_to_bytes([str(s_) for s_ in seqs], L=Lx).astype(np.int64)
            return ((b * self._w).sum(axis=1) % 1000) / 1000.0

    for name, safe in (("plain_landscape", False), ("batch_safe_landscape", True)):
        ts = []
        for rep in range(2):                                   # second pass: engine and caches warm
            np.random.seed(0)
            model = NoisyAbstractModel(_Synth(safe), 0.9, device=device)
            model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, Lx, alpha, 5)), np.random.random(1000))
            batches = [synth.bytes_to_strings(synth.random_sequence_bytes(100, Lx, alpha, 100 + c)) for c in range(20)]
            t0 = time.perf_counter()
            for bch in batches:
                model.get_fitness(bch)
            ts.append(time.perf_counter() - t0)
        out[name] = {"value": 2000 / ts[-1], "unit": "sequences/s", "wall_ms": ts[-1] * 1e3, "cache_after": len(model.cache),
                     "oracle_calls": int(model.landscape.cost)}
    # the same pattern over a DEVICE table landscape (TF-binding style: every 8-mer has a value; flexs_amd.landscapes.TFBinding
    # keeps such a table on the GPU): the whole uncached batch is one device round trip (fx_cache_nam_query)
    class _Table(flexs_amd.Landscape):
        batch_safe = True

        def __init__(self):
            super().__init__("table")
            self._L, self._t = 8, None
            self._vals = np.random.default_rng(3).random(4 ** 8)

        def _native_table(self):
            if self._t is None:
                self._t = _native.NativeTable(_native.Engine.get(device), self._vals, "ACGT", bits=2)
            return self._t

        def _fitness_function(self, seqs):
            return self._native_table().lookup(_native.sequences_to_bytes([str(s_) for s_ in seqs], L=8))

    ts = []
    for rep in range(2):
        np.random.seed(0)
        model = NoisyAbstractModel(_Table(), 0.9, device=device)
        model.train(synth.bytes_to_strings(synth.random_sequence_bytes(1000, 8, "ACGT", 5)), np.random.random(1000))
        batches = [synth.bytes_to_strings(synth.random_sequence_bytes(100, 8, "ACGT", 100 + c)) for c in range(20)]
        t0 = time.perf_counter()
        for bch in batches:
            model.get_fitness(bch)
        ts.append(time.perf_counter() - t0)
    out["device_table_landscape_L8"] = {"value": 2000 / ts[-1], "unit": "sequences/s", "wall_ms": ts[-1] * 1e3, "cache_after": len(model.cache),
                                        "oracle_calls": int(model.landscape.cost),
                                        "what": "TF-binding style table of all 8-mers on the device: neighbour search + both look-ups + blend "
                                                "of a batch in one device round trip (fx_cache_nam_query), RNG draws on the host"}
    out["what"] = ("NoisyAbstractModel(ss=0.9).get_fitness, RNA L=14, CbAS pattern: 20 calls x 100 sequences, cache 1000 -> ~3000. "
                   "Host-bound by construction: per call one K4 launch (~20 us) + K5, but the 2 oracle calls and the RNG draw "
                   "per uncached query stay in a Python loop in the reference's order (plain landscape); a batch_safe landscape "
                   "gets two batched oracle calls instead")
    return out


def end_to_end_block(device, configs=None):
    """SURVEY.md 8(d)'s primary metric: `get_fitness(list[str])` -> np.ndarray, host strings in, host array out (string
    marshalling + PCIe both ways inclusive), for configs[1] and -- marshalling cost grows with L -- for C3 (MLP L=14), C4
    (8 x GE L=90) and C5 (3 x CNN L=237, one GPU's 62 500-row share); plus the small-call latency.  Each row carries its
    split: `pack_ms` = the string marshalling alone (csrc/strpack.c into the pinned staging area, worker threads as the
    call uses them; `pack_1thread_ms` beside it), `kernel_ms` = the scoring launch from `configs`, and what is left of
    the wall time is PCIe + synchronisation + Python (pieces overlap in the chunked call, so the parts can exceed the whole)."""
    import flexs_amd
    from flexs_amd import _native, synth

    strpack = _native._strpack
    out = {}

    def pack_ms(seqs, Lx, threads):
        buf = np.empty((len(seqs), Lx), np.uint8)
        prev = strpack.set_threads(threads)
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); strpack.pack(seqs, Lx, buf); ts.append(time.perf_counter() - t0)
        strpack.set_threads(prev)
        return float(np.median(ts)) * 1e3

    rows = (("C2 3xCNN L=8", "cnn", L, ALPHABET, M, BATCH, "C2 full (headline kernel)"),
            ("C3 MLP L=14", "mlp", 14, "UGCA", 1, 100_000, "C3 mlp L=14 A=4 H=100 M=1 N=1e5"),
            ("C4 8xGE L=90", "ge", 90, AAS, 8, 100_000, "C4 ge L=90 A=20 H=100 M=8 N=1e5"),
            ("C5 3xCNN L=237", "cnn", 237, AAS, 3, 62_500, "C5 cnn L=237 A=20 M=3 N=62500 (one GPU's share of 5e5)"))
    for name, kind, Lx, alpha, members, n, cfg_key in rows:
        mods = build_members(kind, Lx, alpha, members, device)
        model = flexs_amd.Ensemble(mods) if members > 1 else mods[0]
        seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, Lx, alpha, 1))
        model.get_fitness(seqs)
        ts = []
        for _ in range(5 if Lx > 100 else 9):
            t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        row = {"value": n / t, "unit": "sequences/s", "wall_ms": t * 1e3, "n": n}
        # launched first, packed behind (fx_score_begin_staged) where the plan and the kernel allow it: the A/B beside it
        eng_ab = mods[0]._engine()
        try:
            c0 = eng_ab.get_option("launch_first_calls")
            model.get_fitness(seqs)
            row["launched_first"] = bool(eng_ab.get_option("launch_first_calls") - c0)
            if row["launched_first"]:
                eng_ab.set_option("launch_first", 0)
                model.get_fitness(seqs)
                ts = []
                for _ in range(5 if Lx > 100 else 9):
                    t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                row["wall_ms_packed_first"] = float(np.median(ts)) * 1e3
                # ... and with the results copied into a fresh array (FLEXS_AMD_RESULTS_IN_PLACE = 0; in place is the default since round 6)
                eng_ab.set_option("launch_first", 1)
                prev_in_place, _native.RESULTS_IN_PLACE = _native.RESULTS_IN_PLACE, 0
                try:
                    model.get_fitness(seqs)
                    ts = []
                    for _ in range(5 if Lx > 100 else 9):
                        t0 = time.perf_counter(); model.get_fitness(seqs); ts.append(time.perf_counter() - t0)
                    row["wall_ms_results_copied"] = float(np.median(ts)) * 1e3
                finally:
                    _native.RESULTS_IN_PLACE = prev_in_place
        except Exception:                                   # (a library without the option: the row stays as it is)
            pass
        finally:
            try:
                eng_ab.set_option("launch_first", 1)
            except Exception:
                pass
        if strpack is not None:
            row["pack_ms"] = pack_ms(seqs, Lx, 0)
            row["pack_1thread_ms"] = pack_ms(seqs, Lx, 1)
        kern = (configs or {}).get(cfg_key, {}).get("kernel_ms")
        if kern:
            row["kernel_ms"] = kern
            row["frac_of_kernel_rate"] = (n / t) / (n / (kern * 1e-3))
        row["h2d_bytes"], row["d2h_bytes"] = n * Lx, 4 * n
        out[name + " list_str"] = row
        if name.startswith("C2"):
            arr_s = np.array(seqs, dtype="S")
            model.get_fitness(arr_s)
            ts = []
            for _ in range(9):
                t0 = time.perf_counter(); model.get_fitness(arr_s); ts.append(time.perf_counter() - t0)
            t = float(np.median(ts))
            out["C2 3xCNN L=8 ndarray_S"] = {"value": n / t, "unit": "sequences/s", "wall_ms": t * 1e3}
            # small-call latency at N in {1, 4, 20, 100, 2001}, host strings -> host scores, median of 200 calls;
            # resident form (narrow generation up to 256 sequences, wide up to 4096, streamed from 384) beside a launch per call (serve_small = 0: the form of rounds 1-2)
            eng = mods[0]._engine()

            def call_us(batch):
                for _ in range(20):
                    model.get_fitness(batch)
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter(); model.get_fitness(batch); ts.append(time.perf_counter() - t0)
                return float(np.median(ts)) * 1e6

            sizes = (1, 4, 20, 100, 2001)
            out["small_call_us"] = {str(k): call_us(seqs[:k]) for k in sizes}
            try:
                eng.set_option("serve_small", 0)
                out["small_call_us_launch_per_call"] = {str(k): call_us(seqs[:k]) for k in sizes}
            finally:
                eng.set_option("serve_small", 1)
            out["small_call_N20_us"] = out["small_call_us"]["20"]          # (round-2 key, kept)
            out["small_call_N20_us_launch_per_call"] = out["small_call_us_launch_per_call"]["20"]
            out["small_call_resident_requests"] = int(eng.get_option("server_calls"))
        del model, mods, seqs
    out["list_str"] = out["C2 3xCNN L=8 list_str"]          # (round-2 key, kept)
    out["what"] = ("get_fitness on host strings -> host float32 array, median wall time, marshalling + PCIe inclusive; "
                   "pack_ms = strpack.pack alone (auto threads) / pack_1thread_ms single-threaded; kernel_ms from `configs`")
    return out


def explorer_patterns_block(device):
    """SURVEY.md 8(d)'s explorer call patterns for configs[3] and configs[4] (the C2 pattern is `small_call_us`):
      DyNA-PPO  `environments/dyna_ppo.py:144-163`: the environment step scores 4-10 sequences per call with the ensemble
                -- here 8 x GlobalEpistasis(100), L = 90, protein alphabet, `Ensemble.get_fitness(list[str])`;
      CMA-ES    `cmaes.py:61-67, 83-108`: P = 15 / 40 solutions are argmax-decoded and scored one population at a time
                -- here `PopulationEvaluator.evaluate` (fx_decode_score) on 3 x CNN(32,100), L = 237, plus the plain
                one-sequence call of the reference loop.
    Host arrays / strings in, host values out; median of 200 calls after 20 warm-up calls."""
    import flexs_amd
    from flexs_amd import synth
    from flexs_amd.utils.population import PopulationEvaluator

    def med_us(fn, reps=200):
        for _ in range(20):
            fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e6

    out = {}
    ens = flexs_amd.Ensemble(build_members("ge", 90, AAS, 8, device))
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(16, 90, AAS, 11))
    out["dynappo_8xGE_L90_us"] = {str(k): med_us(lambda k=k: ens.get_fitness(seqs[:k])) for k in (4, 10)}
    del ens
    ens = flexs_amd.Ensemble(build_members("cnn", 237, AAS, 3, device))
    ev = PopulationEvaluator(ens, AAS, 237)
    rng = np.random.default_rng(5)
    cm = {}
    for P in (15, 40):
        x = rng.standard_normal((P, 237 * len(AAS)))
        cm[f"P={P}"] = med_us(lambda x=x: ev.evaluate(x), reps=100)
    one = synth.bytes_to_strings(synth.random_sequence_bytes(1, 237, AAS, 12))
    cm["N=1 get_fitness"] = med_us(lambda: ens.get_fitness(one), reps=100)
    out["cmaes_3xCNN_L237_us"] = cm
    out["what"] = ("explorer-size calls of configs[3] / configs[4], host in -> host out, median us per call: DyNA-PPO pattern "
                   "(8 x GE L=90, 4 / 10 sequences per Ensemble.get_fitness call) and CMA-ES pattern (3 x CNN L=237: decode + score "
                   "of a population of 15 / 40 in one fx_decode_score round trip; one-sequence get_fitness beside it)")
    return out


def explorer_round_block(device, torch):
    """SURVEY.md 8(f)-1/-2, the callers on either side of the path: one explorer round on configs[0]'s surrogate family --
    `Ensemble.train` of the 3-CNN ensemble on 1000 measured sequences (Adam / MSE / 20 epochs / batch 256: the hand-written
    HIP step of csrc/train_core.h, all members in one fx_train_fit call; the captured-PyTorch-graph path of round 2 is
    timed beside it), then one Adalead round (query budget 2000: several hundred model calls of 1-20
    sequences).  Wall times, second call of each (graphs captured, engine warm)."""
    import random

    import flexs_amd
    from flexs_amd import synth
    from flexs_amd.utils import rollouts

    ens = flexs_amd.Ensemble(build_members("cnn", L, ALPHABET, M, device))
    n = 1000
    seqs = synth.bytes_to_strings(synth.random_sequence_bytes(n, L, ALPHABET, 3))
    y = np.random.default_rng(0).random(n)
    out = {}
    ens.train(seqs, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); ens.train(seqs, y); torch.cuda.synchronize()
    out["train_3xCNN_n1000_ms"] = (time.perf_counter() - t0) * 1e3
    out["train_steps_per_member"] = 20 * ((n + 255) // 256)
    prev = os.environ.get("FLEXS_AMD_TRAIN")
    try:                                                   # round 2's path, same call: one captured PyTorch step per member
        os.environ["FLEXS_AMD_TRAIN"] = "graph"
        ens.train(seqs, y)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); ens.train(seqs, y); torch.cuda.synchronize()
        out["train_3xCNN_n1000_ms_pytorch_graph"] = (time.perf_counter() - t0) * 1e3
    except Exception as ex:  # noqa: BLE001 - a comparison figure only
        out["train_3xCNN_n1000_ms_pytorch_graph"] = f"failed: {type(ex).__name__}"
    finally:
        if prev is None:
            os.environ.pop("FLEXS_AMD_TRAIN", None)
        else:
            os.environ["FLEXS_AMD_TRAIN"] = prev
    # the fits BASELINE configs[3] / configs[4] retrain through every round (flexs/explorer.py:157-160): protein lengths, 20 letters
    try:
        for Lp, key in ((237, "train_3xCNN_L237_n500_ms"), (90, "train_3xCNN_L90_n500_ms")):
            pens = flexs_amd.Ensemble(build_members("cnn", Lp, AAS, 3, device))
            pseqs = synth.bytes_to_strings(synth.random_sequence_bytes(500, Lp, AAS, 3))
            py = np.random.default_rng(0).random(500)
            pens.train(pseqs, py); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); pens.train(pseqs, py); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            out[key] = min(ts) * 1e3
            del pens
        # forward + input-gradient + weight-gradient products of one row-step = 3 x 2 x dense MACs (6 485 108 at L = 237),
        # 500 rows x 20 epochs x 3 members per fit
        out["train_3xCNN_L237_frac_of_peak"] = 3 * 2.0 * synth.algorithmic_macs("cnn", 237, 20, H, F, K) * 500 * 20 * 3 / (out["train_3xCNN_L237_n500_ms"] * 1e-3) / 1e12 / PEAK_TF
    except Exception as ex:  # noqa: BLE001 - never at the cost of the line
        out["train_protein_error"] = f"{type(ex).__name__}: {ex}"[:200]
    for i in range(2):
        random.seed(1)
        c0 = ens.cost
        t0 = time.perf_counter()
        rollouts.adalead_round(ens, seqs, y, sequences_batch_size=100, model_queries_per_batch=2000, alphabet=ALPHABET)
        out["adalead_round_ms"] = (time.perf_counter() - t0) * 1e3
        out["adalead_model_queries"] = int(ens.cost - c0)
    out["what"] = ("one explorer round, 3 x CNN(32,100) L=8: Ensemble.train on 1000 measured sequences (fx_train_fit: hand-written "
                   "HIP forward+backward+Adam, 2 launches per mini-batch step for all members) + "
                   "flexs_amd.utils.rollouts.adalead_round (budget 2000 queries)")
    return out


MEMBER_PARALLEL_WORKLOADS = (("8xCNN L=8 A=4 N=1e5", "cnn", 8, "TGCA", 100_000, 400),
                             ("8xGE L=90 A=20 N=1e5", "ge", 90, AAS, 100_000, 800),
                             ("8xGE L=90 A=20 N=1e6", "ge", 90, AAS, 1_000_000, 100))


def member_parallel_block(world, rank, device, torch, dist, use_dist, steps_hint, solo_group=None):
    """north_star's split: an 8-member ensemble, members sharded over the ranks (contiguous blocks), every rank
    scores the SAME batch with its members, ONE all-gather of the stacked predictions, mean on every rank.
    Strong scaling: the batch is fixed, value = batch x steps / time.  Workloads: 8 x CNN L=8 (configs[1]'s
    surrogate, 8 members) and 8 x GlobalEpistasis L=90 A=20 (configs[3]) at 1e5 and 1e6 sequences.
    `speedup_vs_1gpu` divides by a one-GPU measurement OF THIS RUN: with one rank the block itself is that reference;
    with several, rank 0 first runs the same workload alone over a one-rank group (`solo_group`) while the others wait."""
    from flexs_amd import distributed as fd, synth

    out = {}
    for name, kind, Lx, alpha, n, steps in MEMBER_PARALLEL_WORKLOADS:
        mods = build_members(kind, Lx, alpha, 8, device)
        ref = None
        if world > 1 and solo_group is not None:
            if rank == 0:
                solo = fd.DistributedEnsemble(mods, mode="member", group=solo_group)
                with torch.cuda.stream(solo.stream):
                    d_solo = torch.from_numpy(synth.random_sequence_bytes(n, Lx, alpha, seed=7)).cuda()
                solo.stream.synchronize()
                el, _, _ = run_pipelined(solo, d_solo, n, steps, max(steps // 10, 5), torch, dist, False, want_events=False)
                ref = n * steps / el
                del solo, d_solo
            dist.barrier()
        ens = fd.DistributedEnsemble(mods, mode="member")
        ens.force_collective = use_dist
        with torch.cuda.stream(ens.stream):
            d_seq = torch.from_numpy(synth.random_sequence_bytes(n, Lx, alpha, seed=7)).cuda()   # same batch on every rank
        ens.stream.synchronize()
        elapsed, _, kern_ms = run_pipelined(ens, d_seq, n, steps, max(steps // 10, 5), torch, dist, use_dist)
        # correctness of the exchange: the gathered matrix must reproduce the local members' planes
        ens.launch(d_seq, n, 0, "matrix")
        mat = ens.finish(0)
        ens.launch(d_seq, n, 1, "mean")
        mean = ens.finish(1)
        torch.cuda.synchronize()
        ok = bool(torch.isfinite(mat).all()) and tuple(mat.shape) == (n, 8)
        if rank == 0:
            ok = ok and np.array_equal(np.mean(mat.cpu().numpy(), axis=1), mean.cpu().numpy())
        value = n * steps / elapsed
        if world == 1 and not use_dist:
            ref, ref_src = value, "this block (n_gpus = 1, no collective)"
        elif ref is not None:
            ref_src = "same run: rank 0 alone over a one-rank group, before the sharded measurement"
        else:
            ref_src = None
        out[name] = {"value": value, "unit": "sequences/s", "ms_per_step": elapsed / steps * 1e3,
                     "kernel_ms_this_rank": kern_ms, "steps": steps, "members": 8,
                     "members_per_rank": -(-8 // world), "gathered_bytes_per_rank": 4 * n * -(-8 // world) * world,
                     "one_gpu_reference": ref, "one_gpu_reference_source": ref_src,
                     "speedup_vs_1gpu": (value / ref) if ref else None,
                     "checked": ok}
        del ens, mods, d_seq
    out["what"] = ("8-member ensembles sharded member-parallel over the ranks (flexs/ensemble.py:54-59): fused kernel "
                   "for this rank's members + one RCCL all-gather of the stacked (N, 8) predictions + np.mean-order mean "
                   "on every rank; same batch on every rank (strong scaling), double-buffered so the gather of step k "
                   "overlaps step k+1; speedup_vs_1gpu = value / one_gpu_reference, the latter measured in THIS run "
                   "(one_gpu_reference_source); ideal = 8 / members_per_rank")
    return out

def prepared_block(timeout_s=90.0):
    """An A/B of kernel forms that were written after round 4's GPU budget was spent (csrc/OPTIONS.md `train_swizzle`: rotated LDS rows
    and staged conv kernels for GFP-length CNN fits; DEFAULT OFF, bit-identical on the CPU under the SIMT emulator) -- measured here
    because this run is the first time they meet a device.  In a CHILD process with a time limit: whatever happens to it, the
    contract line above is already measured and is printed; the child's answer (or what went wrong) goes into
    roofline.per_config.  Not part of `value`.  FLEXS_AMD_BENCH_PREPARED=0 skips it."""
    import subprocess

    if os.environ.get("FLEXS_AMD_BENCH_PREPARED", "1") == "0":
        return {"skipped": "FLEXS_AMD_BENCH_PREPARED=0"}
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "runs", "r5_train_swizzle_ab.py"), "--json"], cwd=ROOT,
                           capture_output=True, text=True, timeout=timeout_s)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": f"child exit {r.returncode}", "stderr_tail": r.stderr[-300:]}
        return json.loads(lines[-1])
    except subprocess.TimeoutExpired as ex:
        got = {"error": f"not finished within {timeout_s:.0f} s (child stopped); what it had reported until then is kept"}
        try:
            text = ex.stdout.decode() if isinstance(ex.stdout, bytes) else (ex.stdout or "")
            lines = [ln for ln in text.splitlines() if ln.startswith("{")]
            if lines:
                got.update(json.loads(lines[-1]))
        except Exception:  # noqa: BLE001
            pass
        return got
    except Exception as ex:  # noqa: BLE001 -- an experiment must not cost the record
        return {"error": f"{type(ex).__name__}: {ex}"}

