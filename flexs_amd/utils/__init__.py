"""Host-side helpers around the engine:

* `sequence_utils`  -- alphabets and the string <-> one-hot codecs (device-backed batch forms included)
* `edit_distance`   -- `SeenSequences`: the DyNA-PPO environments' `sequence_density` on the K4 kernels
* `population`      -- decode-and-score a whole CMA-ES / DyNA-PPO population in one device round trip
"""
import importlib

edit_distance = importlib.import_module("flexs_amd.utils.edit_distance")
sequence_utils = importlib.import_module("flexs_amd.utils.sequence_utils")
