"""`flexs_amd.baselines`: the surrogate models of the hot path, scored on the MI355X.

Only `models` lives here.  The reference's explorers (Adalead, CMA-ES, DyNA-PPO, CbAS, ...) are host control flow
that merely calls `model.get_fitness / train / cost`; they are not rebuilt and run unchanged against these models
(DESIGN.md section 7).  `flexs_amd.utils.population` offers the batched form of their decode-and-score step.
"""
import importlib

models = importlib.import_module("flexs_amd.baselines.models")
