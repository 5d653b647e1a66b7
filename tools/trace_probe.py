#!/usr/bin/env python3
"""In-kernel timeline of one scoring launch (`make trace` build + engine option "trace", fx_debug_trace_read): where a launch's time goes --
LDS fill, first-tile latency, per-tile time, tail.  GPU box only.  Prints one JSON object per case.

    python tools/trace_probe.py            # the round-2 case list
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from flexs_amd import _native, synth  # noqa: E402
from flexs_amd.baselines.models.keras_model import Architecture  # noqa: E402

AAS = "ILVAGMFYWEDQNHCRKSTP"
eng = _native.Engine.get(0)
TICK_US = 0.01          # 100 MHz constant clock


def natives(kind, L, A, H, M, F=0, K=0):
    arch = Architecture(kind, L, A, H, num_filters=F, kernel_size=K)
    out = []
    for m in range(M):
        nm = _native.NativeModel(eng, {"cnn": 0, "mlp": 1, "ge": 2}[kind], L, A, F, H, K)
        nm.set_weights(synth.synthetic_weights(arch.shapes(), 1000 + m))
        out.append(nm)
    return out


def pct(x, q):
    return float(np.percentile(x, q)) if len(x) else None


def slowest_blocks(t, t0, k=4):
    """For the k workgroups that finish last (and the one that finishes first): tiles and finish time per SIMD."""
    done = np.where(t[:, :, 4] > 0, (t[:, :, 4] - t0) * TICK_US, 0.0)
    order = np.argsort(done.max(axis=1))
    order = [b for b in order if done[b].max() > 0]
    out = []
    for b in [order[0]] + order[-k:]:
        per = []
        for sd in range(4):
            w = np.flatnonzero(t[b, :, 7] == sd + 1)
            per.append({"tiles": [int(x) for x in t[b, w, 5]], "done": [round(float(x), 1) for x in done[b, w]]})
        out.append({"block": int(b), "tiles": int(t[b, :, 5].sum()), "simd": per})
    return out


def trace_case(label, kind, L, alpha, M, N, H=100, F=0, K=0, opts=None):
    ms_ = natives(kind, L, len(alpha), H, M, F, K)
    lut = _native.make_lut(alpha)
    d_in = torch.from_numpy(synth.random_sequence_bytes(N, L, alpha, 0)).cuda()
    stride = (N + 63) // 64 * 64
    d_pl = torch.empty((M, stride), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for k_, v_ in (opts or {}).items():
        eng.set_option(k_, v_)
    for _ in range(30):
        eng.score_planes_dev(ms_, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride)
    eng.sync()
    ev_us = eng.time_score_planes(ms_, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride, 300) / 300 * 1e3
    eng.set_option("trace", 1)
    res = []
    for _ in range(3):
        eng.score_planes_dev(ms_, d_in.data_ptr(), N, L, lut, d_pl.data_ptr(), stride)
        eng.sync()
        t = eng.trace_read().astype(np.int64)
        entered = t[:, :, 0] > 0
        t0 = t[:, :, 0][entered].min()
        worked = t[:, :, 5] > 0
        end_all = (t[:, :, 6][t[:, :, 6] > 0] - t0) * TICK_US
        fill = (t[:, :, 1] - t[:, :, 0])[t[:, :, 1] > 0] * TICK_US
        first_start = (t[:, :, 2][worked] - t0) * TICK_US
        first_dur = (t[:, :, 3] - t[:, :, 2])[worked] * TICK_US
        tiles = t[:, :, 5][worked]
        per_tile = ((t[:, :, 4] - t[:, :, 2])[worked] * TICK_US) / tiles
        last_done = (t[:, :, 4][worked] - t0) * TICK_US
        entry = (t[:, :, 0][entered] - t0) * TICK_US
        ph = [((t[:, :, k] - t[:, :, 2])[worked & (t[:, :, k] > 0)] * TICK_US) for k in (8, 9, 10)]
        simd_counts = [int(((t[:, :, 7] == v + 1) & entered).sum()) for v in range(4)]
        res.append({
            "span_us": float(end_all.max()), "exit_us_p10_p50_p90_max": [pct(end_all, 10), pct(end_all, 50), pct(end_all, 90), float(end_all.max())],
            "tile_start_all_us_p50_max": [pct((t[:, :, 2][t[:, :, 2] > 0] - t0) * TICK_US, 50), float(((t[:, :, 2][t[:, :, 2] > 0] - t0) * TICK_US).max())],
            "blocks": int(entered.any(axis=1).sum()), "waves_entered": int(entered.sum()),
            "waves_with_tiles": int(worked.sum()), "tiles_per_working_wave": [int(tiles.min()), float(tiles.mean()), int(tiles.max())],
            "entry_us_p50_max": [pct(entry, 50), float(entry.max())],
            "fill_us_p50_max": [pct(fill, 50), float(fill.max())],
            "first_tile_start_us_p50_max": [pct(first_start, 50), float(first_start.max())],
            "first_tile_dur_us_p10_p50_p90_max": [pct(first_dur, 10), pct(first_dur, 50), pct(first_dur, 90), float(first_dur.max())],
            "per_tile_us_p10_p50_p90": [pct(per_tile, 10), pct(per_tile, 50), pct(per_tile, 90)],
            "last_tile_done_us_p10_p50_p90_max": [pct(last_done, 10), pct(last_done, 50), pct(last_done, 90), float(last_done.max())],
            "first_tile_phase_ends_us_p50": [pct(x, 50) for x in ph], "waves_per_simd": simd_counts,
            "slowest_blocks": slowest_blocks(t, t0),
            "block0_waves_simd_tiles_firststart_firstend_lastend": [
                [int(t[0, w, 7]) - 1, int(t[0, w, 5]), round(float((t[0, w, 2] - t0) * TICK_US), 2) if t[0, w, 2] else None,
                 round(float((t[0, w, 3] - t0) * TICK_US), 2) if t[0, w, 3] else None,
                 round(float((t[0, w, 4] - t0) * TICK_US), 2) if t[0, w, 4] else None] for w in range(16) if t[0, w, 0]],
        })
    eng.set_option("trace", 0)
    for k_ in (opts or {}):
        eng.set_option(k_, {"cnn_big_units": 12, "cnn_seg": -1, "ge_bytetab": 1, "mlp_pair": 1, "wave_prio": 1, "stage_bytes": 1}.get(k_, 0))
    for k_ in (opts or {}):
        eng.set_option(k_, {"cnn_big_units": 12, "cnn_seg": -1, "ge_bytetab": 1, "mlp_pair": 1, "wave_prio": 1, "stage_bytes": 1, "dense_waves": 0, "stage_fill": 1, "cnn_quad": 1, "dma_fill": 1, "cnn_seg_multi": 1, "cnn_pair_seg4": 1}.get(k_, 0))
    out = {"what": label, "event_us_per_launch": ev_us, "trace": res[-1], "span_us_3runs": [r["span_us"] for r in res]}
    print(json.dumps(out), flush=True)
    return out


if __name__ == "__main__":
    rows = []
    if not os.environ.get("FLEXS_AMD_LIB"):
        raise SystemExit("the timeline needs the trace build: make -C flexs_amd/csrc trace; FLEXS_AMD_LIB=$PWD/flexs_amd/libflexs_amd_trace.so")
    if os.environ.get("FX_PHASES"):
        # phase build (make trace-phases): only the cases whose first-tile phases are in question
        for M, N in ((1, 100_000), (1, 4_000)):
            rows.append(trace_case(f"[phases] mlp L=14 M={M} N={N}", "mlp", 14, "UGCA", M, N))
        rows.append(trace_case("[phases] ge L=90 M=1 N=100000", "ge", 90, AAS, 1, 100_000))
        rows.append(trace_case("[phases] ge L=90 M=8 N=100000", "ge", 90, AAS, 8, 100_000))
        rows.append(trace_case("[phases] ge L=90 M=1 N=4000", "ge", 90, AAS, 1, 4_000))
        rows.append(trace_case("[phases] cnn L=8 M=1 N=10000", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5))
        rows.append(trace_case("[phases] cnn L=8 M=1 N=4000", "cnn", 8, "TGCA", 1, 4_000, F=32, K=5))
        rows.append(trace_case("[phases] cnn L=8 M=3 N=20", "cnn", 8, "TGCA", 3, 20, F=32, K=5))
        rows.append(trace_case("[phases] cnn L=14 M=3 N=20", "cnn", 14, "UGCA", 3, 20, F=32, K=5))
        json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe_phases.json"), "w"), indent=1)
        sys.exit(0)
    if os.environ.get("FX_SET") == "r6p":
        # round 6, phase build (make trace-phases): where a lockstep round of the slab form goes -- first layer / H x H layer 2 / layer 3
        rows.append(trace_case("[phases] mlp L=14 H=200 M=1 N=100000", "mlp", 14, "UGCA", 1, 100_000, H=200))
        rows.append(trace_case("[phases] mlp L=14 H=200 M=1 N=32768 (one full round)", "mlp", 14, "UGCA", 1, 32_768, H=200))
        rows.append(trace_case("[phases] ge L=90 H=200 M=1 N=100000", "ge", 90, AAS, 1, 100_000, H=200))
        rows.append(trace_case("[phases] mlp L=14 H=100 M=1 N=100000", "mlp", 14, "UGCA", 1, 100_000))
        json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe_r6p.json"), "w"), indent=1)
        sys.exit(0)
    if os.environ.get("FX_SET") == "r6":
        # round 6: where the launches furthest below the roofline spend their time
        rows.append(trace_case("mlp L=14 H=100 M=1 N=100000 (C3)", "mlp", 14, "UGCA", 1, 100_000))
        rows.append(trace_case("mlp L=14 H=100 M=1 N=1000000", "mlp", 14, "UGCA", 1, 1_000_000))
        rows.append(trace_case("mlp L=14 H=200 M=1 N=100000", "mlp", 14, "UGCA", 1, 100_000, H=200))
        rows.append(trace_case("cnn L=8 H=200 M=1 N=100000", "cnn", 8, "TGCA", 1, 100_000, H=200, F=32, K=5))
        rows.append(trace_case("ge L=90 M=1 N=100000", "ge", 90, AAS, 1, 100_000))
        rows.append(trace_case("ge L=90 M=8 N=100000 (C4)", "ge", 90, AAS, 8, 100_000))
        rows.append(trace_case("cnn L=8 M=1 N=10000 (C1)", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5))
        rows.append(trace_case("cnn L=8 M=3 N=100000 (C2 headline)", "cnn", 8, "TGCA", 3, 100_000, F=32, K=5))
        json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe_r6.json"), "w"), indent=1)
        sys.exit(0)
    if os.environ.get("FX_SET") == "seg":
        # small launches of long sequences: position-segmented forms
        for kind, L, alpha, M, N, kw in (("cnn", 100, "UGCA", 3, 20, dict(F=32, K=5)), ("cnn", 50, "UGCA", 3, 20, dict(F=32, K=5)),
                                         ("cnn", 14, "UGCA", 3, 20, dict(F=32, K=5)), ("cnn", 237, AAS, 3, 16, dict(F=32, K=5))):
            for multi in (1, 0):
                rows.append(trace_case(f"{kind} L={L} M={M} N={N} cnn_seg_multi={multi}", kind, L, alpha, M, N, opts={"cnn_seg_multi": multi}, **kw))
        json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe_seg.json"), "w"), indent=1)
        sys.exit(0)
    if os.environ.get("FX_SET") == "dma":
        # direct global -> LDS weight copies (engine option dma_fill) against the copies through registers
        for kind, L, alpha, M, N, kw in (("cnn", 8, "TGCA", 1, 4_000, dict(F=32, K=5)), ("cnn", 8, "TGCA", 1, 10_000, dict(F=32, K=5)),
                                         ("cnn", 8, "TGCA", 3, 10_000, dict(F=32, K=5)), ("cnn", 8, "TGCA", 1, 32_768, dict(F=32, K=5)),
                                         ("cnn", 14, "UGCA", 1, 10_000, dict(F=32, K=5)),
                                         ("mlp", 14, "UGCA", 1, 100_000, {}), ("mlp", 14, "UGCA", 1, 20_000, {}),
                                         ("ge", 90, AAS, 1, 100_000, {}), ("ge", 90, AAS, 8, 100_000, {})):
            for dma in (1, 0):
                rows.append(trace_case(f"{kind} L={L} M={M} N={N} dma_fill={dma}", kind, L, alpha, M, N, opts={"dma_fill": dma}, **kw))
        json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe_dma.json"), "w"), indent=1)
        sys.exit(0)
    for M, N in ((1, 4000), (1, 10_000), (1, 32_768), (3, 10_000), (3, 100_000)):
        rows.append(trace_case(f"cnn L=8 M={M} N={N}", "cnn", 8, "TGCA", M, N, F=32, K=5))
    rows.append(trace_case("cnn L=8 M=1 N=10000 big_units=1", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5, opts={"cnn_big_units": 1}))
    rows.append(trace_case("cnn L=8 M=1 N=10000 cnn_quad=0", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5, opts={"cnn_quad": 0}))
    rows.append(trace_case("cnn L=8 M=1 N=4000 cnn_quad=0", "cnn", 8, "TGCA", 1, 4_000, F=32, K=5, opts={"cnn_quad": 0}))
    rows.append(trace_case("cnn L=8 M=3 N=10000 cnn_quad=2", "cnn", 8, "TGCA", 3, 10_000, F=32, K=5, opts={"cnn_quad": 2}))
    rows.append(trace_case("cnn L=8 M=1 N=32768 cnn_quad=2", "cnn", 8, "TGCA", 1, 32_768, F=32, K=5, opts={"cnn_quad": 2}))
    rows.append(trace_case("cnn L=8 M=1 N=10000 stage_fill=0", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5, opts={"stage_fill": 0}))
    rows.append(trace_case("cnn L=8 M=1 N=4000 stage_fill=0", "cnn", 8, "TGCA", 1, 4_000, F=32, K=5, opts={"stage_fill": 0}))
    rows.append(trace_case("cnn L=8 M=3 N=10000 stage_fill=0", "cnn", 8, "TGCA", 3, 10_000, F=32, K=5, opts={"stage_fill": 0}))
    rows.append(trace_case("cnn L=50 M=3 N=100 (segmented)", "cnn", 50, "UGCA", 3, 100, F=32, K=5))
    rows.append(trace_case("cnn L=14 M=1 N=10000", "cnn", 14, "UGCA", 1, 10_000, F=32, K=5))
    rows.append(trace_case("cnn L=14 M=1 N=10000 stage_fill=0", "cnn", 14, "UGCA", 1, 10_000, F=32, K=5, opts={"stage_fill": 0}))
    rows.append(trace_case("cnn L=8 M=1 N=10000 variant=11", "cnn", 8, "TGCA", 1, 10_000, F=32, K=5, opts={"cnn_variant": 11}))
    rows.append(trace_case("cnn L=8 M=3 N=10000 variant=11", "cnn", 8, "TGCA", 3, 10_000, F=32, K=5, opts={"cnn_variant": 11}))
    rows.append(trace_case("cnn L=8 M=1 N=32768 variant=11", "cnn", 8, "TGCA", 1, 32_768, F=32, K=5, opts={"cnn_variant": 11}))
    for M, N in ((8, 100_000), (1, 100_000)):
        rows.append(trace_case(f"ge L=90 M={M} N={N}", "ge", 90, AAS, M, N))
        rows.append(trace_case(f"ge L=90 M={M} N={N} ge_bytetab=0", "ge", 90, AAS, M, N, opts={"ge_bytetab": 0}))
    rows.append(trace_case("mlp L=14 M=1 N=100000", "mlp", 14, "UGCA", 1, 100_000))
    rows.append(trace_case("mlp L=14 M=1 N=100000 mlp_pair=0", "mlp", 14, "UGCA", 1, 100_000, opts={"mlp_pair": 0}))
    for dw in (8, 16):
        for N in (50_000, 100_000, 200_000, 400_000, 1_000_000):
            rows.append(trace_case(f"mlp L=14 M=1 N={N} dense_waves={dw}", "mlp", 14, "UGCA", 1, N, opts={"dense_waves": dw}))
        for M, N in ((1, 100_000), (1, 400_000), (8, 100_000)):
            rows.append(trace_case(f"ge L=90 M={M} N={N} dense_waves={dw}", "ge", 90, AAS, M, N, opts={"dense_waves": dw}))
    rows.append(trace_case("mlp L=14 M=1 N=100000 stage_bytes=0", "mlp", 14, "UGCA", 1, 100_000, opts={"stage_bytes": 0}))
    rows.append(trace_case("ge L=90 M=8 N=100000 stage_bytes=0", "ge", 90, AAS, 8, 100_000, opts={"stage_bytes": 0}))
    rows.append(trace_case("ge L=90 M=1 N=100000 stage_bytes=0", "ge", 90, AAS, 1, 100_000, opts={"stage_bytes": 0}))
    rows.append(trace_case("ge L=90 M=8 N=100000 wave_prio=1", "ge", 90, AAS, 8, 100_000, opts={"wave_prio": 1}))
    rows.append(trace_case("ge L=90 M=1 N=100000 wave_prio=1", "ge", 90, AAS, 1, 100_000, opts={"wave_prio": 1}))
    rows.append(trace_case("cnn L=8 M=3 N=100000 wave_prio=1", "cnn", 8, "TGCA", 3, 100_000, F=32, K=5, opts={"wave_prio": 1}))
    rows.append(trace_case("cnn L=8 M=1 N=32768 wave_prio=1", "cnn", 8, "TGCA", 1, 32_768, F=32, K=5, opts={"wave_prio": 1}))
    rows.append(trace_case("cnn L=8 M=3 N=10000 wave_prio=1", "cnn", 8, "TGCA", 3, 10_000, F=32, K=5, opts={"wave_prio": 1}))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rows, open(os.path.join(ROOT, "gpurun_out", "trace_probe.json"), "w"), indent=1)
