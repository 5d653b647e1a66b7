"""Where the layer-parallel protein kernel spends its launch: the kernel leaves after stage k (cnn_lp_debug), launches timed from C."""
import sys; sys.path.insert(0, ".")
import numpy as np, torch
from flexs_amd import _native, synth
from flexs_amd.baselines import models as bm
eng = _native.Engine.get()
AAS = "ILVAGMFYWEDQNHCRKSTP"
for L, M, n in ((237, 1, 1), (237, 3, 40), (90, 1, 16)):
    members = [bm.CNN(L, 32, 100, AAS, seed=m) for m in range(M)]
    nat = [m.native() for m in members]
    d_in = torch.from_numpy(synth.random_sequence_bytes(n, L, AAS, 12)).cuda()
    d_pl = torch.empty((M, 64), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    out = []
    for stage in (0, 6, 5, 7, 4, 3, 2, 1):           # (the full kernel first: the early exits leave tickets / pool entries behind)
        eng.set_option("cnn_lp_debug", stage)
        eng.time_score_planes(nat, d_in.data_ptr(), n, L, members[0]._lut, d_pl.data_ptr(), 64, 20)
        ms = eng.time_score_planes(nat, d_in.data_ptr(), n, L, members[0]._lut, d_pl.data_ptr(), 64, 200)
        out.append((stage, ms / 200 * 1e3))
        # (stages that skip the barrier leave the counter behind the host's total: reset by re-creating nothing -- the counter only
        #  needs to be >= target, and skipped arrivals make it smaller; so run the stages that skip the barrier FIRST per engine)
    eng.set_option("cnn_lp_debug", 0)
    print(f"{M} x CNN L={L} N={n}: " + ", ".join(f"stage {s}: {t:.1f} us" for s, t in out), flush=True)
