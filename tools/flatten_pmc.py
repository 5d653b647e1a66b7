"""profiles/rN_pmc_bench.json in the form bench.py's pmc_block reads: the headline kernel's figures of tools/summarize_pmc.py's
per-kernel summary, flat, with the commit the passes were taken at.  usage: flatten_pmc.py <summary.json> <out.json> <commit>"""
import json, sys
src, dst, commit = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.load(open(src))
name = next(k for k in d if "k_score_cnn_mfma" in k)
k = d[name]
out = {"kernel": name, "hbm_bytes_per_launch": k.get("hbm_bytes"), "fetch_size_kib": k.get("FETCH_SIZE"), "write_size_kib": k.get("WRITE_SIZE"),
       "mfma_util": k.get("mfma_util"), "v_valu_per_mfma": k.get("v"), "mfma_instructions_per_launch": k.get("SQ_INSTS_MFMA"),
       "lds_bank_conflict_cycles": k.get("SQ_LDS_BANK_CONFLICT"), "lds_active_cycles": k.get("SQ_LDS_IDX_ACTIVE"),
       "kernel_us_profiled": (k.get("ns") or {}).get("sq", 0) / 1e3 or None, "commit": commit,
       "note": "rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ util set | SQ inst set, separate runs, no trace domains) over "
               "`python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras` (tools/archive/gpu_round4.sh), summarised by "
               "tools/summarize_pmc.py; hbm_bytes = FETCH_SIZE x 2 (gfx950 unit correction) + WRITE_SIZE, KiB -> bytes; "
               "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMD x CU-cycles)",
       "all_kernels": d}
json.dump(out, open(dst, "w"), indent=1)
print({a: out[a] for a in ("hbm_bytes_per_launch", "mfma_util", "v_valu_per_mfma", "commit")})
