"""`NoisyAbstractModel` -- same contract as
flexs/baselines/models/noisy_abstract_model.py:9-101.

f_hat(x) = alpha^d f(x) + (1 - alpha^d) eps, d = edit distance to the nearest
cached sequence (first cache entry attaining it), eps ~ Exp(mean = f(neighbour)).

The O(Q*C) neighbour search runs on the GPU (bit-parallel Levenshtein, K4
fx_cache_min_dist) against a device-resident copy of the cache keys; the blend
is the K5 kernel (fx_nam_combine).  Everything that touches the global NumPy RNG
or the ground-truth landscape stays on the host in the reference's order, so
results and RNG state are bit-identical to the reference for the same seed.
"""
import numpy as np

import flexs_amd
from flexs_amd import _native
from flexs_amd.types import SEQUENCES_TYPE


class _CacheDict(dict):
    """`NoisyAbstractModel.cache`: a plain dict for every caller, which additionally counts removals -- the
    device-resident copy of the keys is append-only, so anything but an append (del / pop / popitem / clear) has to
    trigger a rebuild, wherever in the insertion order it happened."""

    removals = 0

    def __delitem__(self, key):
        super().__delitem__(key)
        self.removals += 1

    def pop(self, *args):
        n = len(self)
        out = super().pop(*args)
        self.removals += n != len(self)
        return out

    def popitem(self):
        out = super().popitem()
        self.removals += 1
        return out

    def clear(self):
        super().clear()
        self.removals += 1


_MT_FAST = None          # None = not checked yet; True / False = the raw-state shortcut of _rng_checkpoint is valid here


def _mt_state_layout_ok() -> bool:
    """Self-check of the shortcut below, once per process: the 2500 bytes at `state_address` must BE NumPy's MT19937 state
    as `np.random.get_state()` reports it (key[624] as uint32, then pos as int32), and writing them back must restore a
    stream that has moved on.  A NumPy that lays the struct out differently fails this and gets the slow, documented path
    -- never a silently corrupted random stream."""
    import ctypes

    try:
        bg = np.random.mtrand._rand._bit_generator
        if type(bg).__name__ != "MT19937":
            return False
        keep = np.random.get_state()                                  # the user's stream: put back at the end, whatever happens
        try:
            addr, size = bg.ctypes.state_address, 624 * 4 + 4
            name, key, pos = keep[0], keep[1], keep[2]
            raw = ctypes.string_at(addr, size)
            if name != "MT19937" or raw[:2496] != np.ascontiguousarray(key, dtype=np.uint32).tobytes() \
                    or int.from_bytes(raw[2496:2500], "little", signed=True) != int(pos):
                return False
            np.random.standard_exponential(700)                       # crosses a 624-word refill of the key
            moved = np.random.get_state()
            if np.array_equal(moved[1], key) and moved[2] == pos:
                return False
            ctypes.memmove(addr, raw, size)
            back = np.random.get_state()
            return bool(np.array_equal(back[1], key) and back[2] == pos)
        finally:
            np.random.set_state(keep)
    except Exception:                                                 # noqa: BLE001 (any surprise = no shortcut)
        return False


def _rng_checkpoint():
    """Returns restore(): puts NumPy's global RNG back where it is now.  `np.random.get_state()` costs ~40 us (it builds a
    tuple around a copy of the 624-word state) -- more than the device round trip it guards; for the stock MT19937 bit
    generator the 2500 bytes of its state struct (key[624] + pos; the legacy Gaussian cache is not touched by exponential
    draws) are copied directly instead -- after `_mt_state_layout_ok` has confirmed, once, that those bytes are the state."""
    import ctypes

    global _MT_FAST
    if _MT_FAST is None:
        _MT_FAST = _mt_state_layout_ok()
    bg = np.random.mtrand._rand._bit_generator
    if _MT_FAST and type(bg).__name__ == "MT19937":
        addr, size = bg.ctypes.state_address, 624 * 4 + 4
        saved = ctypes.string_at(addr, size)
        return lambda: ctypes.memmove(addr, saved, size)
    state = np.random.get_state()
    return lambda: np.random.set_state(state)


class NoisyAbstractModel(flexs_amd.Model):
    def __init__(self, landscape: flexs_amd.Landscape, signal_strength: float = 0.9, distance: str = "levenshtein",
                 device: int = None):
        super().__init__(f"NAMb_ss{signal_strength}")                 # noisy_abstract_model.py:36
        self.landscape = landscape
        self.ss = signal_strength
        self.cache = _CacheDict()
        if distance not in ("levenshtein", "hamming"):
            raise ValueError("distance must be 'levenshtein' (reference behaviour) or 'hamming'")
        self._mode = _native.FX_LEVENSHTEIN if distance == "levenshtein" else _native.FX_HAMMING
        self._device = device
        self._dev_cache = None            # NativeCache mirroring list(self.cache) in insertion order
        self._dev_keys = []               # python-side mirror of what was appended
        self._dev_token = None            # (id of the dict, its removal count) the device copy was built for
        self._pending = []                # keys THIS model put into `cache` since the last sync, in insertion order (see _note_new_keys)

    def __getstate__(self):
        state = self.__dict__.copy()           # copy / pickle: the device key store is rebuilt from `cache` on first use
        state["_dev_cache"] = None
        state["_dev_keys"] = []
        state["_dev_token"] = None
        state["_pending"] = []
        return state

    # ---------------------------------------------------------------- device cache sync
    @staticmethod
    def _rows(seqs, L: int) -> np.ndarray:
        """(n, L) byte rows; sequences shorter than the row are NUL-padded (ragged batches are legal for
        `editdistance.eval`, noisy_abstract_model.py:51, even though explorers keep one length)."""
        if all(len(s) == L for s in seqs):
            return _native.sequences_to_bytes(seqs, L=L)
        return _native.ragged_to_bytes(seqs, L)

    def _sync_device_cache(self, min_row: int, defer: bool = False):
        """Bring the device copy of `list(self.cache)` up to date; rows hold at least `min_row` bytes.  `defer`: the keys
        added since the last call are not appended here but returned as rows for the caller's next device call to append
        (fx_cache_nam_query takes them along); None when nothing is pending."""
        import itertools

        keys = self.cache.keys()
        n = len(self._dev_keys)
        stale = self._dev_cache is None or n > len(keys)
        if isinstance(self.cache, _CacheDict):
            # removals anywhere in the order are counted by the dict itself
            token = (id(self.cache), self.cache.removals)
            stale = stale or token != self._dev_token
        else:
            # the user replaced `cache` by a dict of their own: compare the whole mirrored prefix
            token = None
            stale = stale or any(str(a) != b for a, b in zip(keys, self._dev_keys))
        self._dev_token = token
        pending, self._pending = getattr(self, "_pending", []), []
        if not stale and token is not None and len(keys) == n + len(pending):
            # every key beyond the mirrored prefix was put there by this model (train / _fitness_function note what they add, in
            # order): no walk over the dict.  Skipping the first n keys of a dict is O(n) -- 12 us of a 49-us one-sequence call at
            # 10 000 cached sequences, growing with the cache (round 5, tools/runs/r5_nam_py_profile.py)
            fresh = pending
        else:
            fresh = [str(k) for k in itertools.islice(keys, 0 if stale else n, None)]
        need = max([min_row, 1] + [len(k) for k in fresh])
        if not stale and need > self._dev_cache.L:          # a longer sequence than any before: wider rows
            stale, fresh = True, [str(k) for k in keys]
        if stale:
            need = max([need] + [len(k) for k in fresh])
            self._dev_cache = self._new_device_cache(need)
            self._dev_keys = []
        if fresh:
            rows = self._rows(fresh, self._dev_cache.L)
            self._dev_keys.extend(fresh)
            if defer and not stale:
                return rows
            self._dev_cache.append(rows)
        return None

    def _new_device_cache(self, row_bytes: int):
        """Key store with `.L`, `.append(rows)` and `.min_dist(rows, mode)`; the multi-GPU model
        (flexs_amd.distributed.ShardedNoisyAbstractModel) substitutes a rank-sharded one."""
        return _native.NativeCache(_native.Engine.get(self._device), row_bytes)

    def _blend(self, signal, noise, dist, alpha_tab):
        """K5: alpha^d * signal + (1 - alpha^d) * noise in float64 on this rank's GPU (:93-94)."""
        return _native.Engine.get(self._device).nam_combine(signal, noise, dist, alpha_tab)

    def _get_min_distance(self, sequence):
        """noisy_abstract_model.py:42-60 for one query (kept for API parity)."""
        if len(self.cache) == 0:
            return 0, sequence
        d, nb = self._min_distances([sequence])
        return int(d[0]), nb[0]

    def _min_distances(self, sequences):
        self._sync_device_cache(max(len(s) for s in sequences))
        dist, arg = self._dev_cache.min_dist(self._rows(sequences, self._dev_cache.L), self._mode)
        return dist, [self._dev_keys[i] for i in arg]

    def _alpha_tab(self, L: int) -> np.ndarray:
        """ss ** d for d = 0..L (:93, Python float pow), kept while ss and L stay the same."""
        c = getattr(self, "_alpha_cache", None)
        if c is None or c[0] != (self.ss, L):
            c = self._alpha_cache = ((self.ss, L), np.array([self.ss ** d for d in range(L + 1)], np.float64))
        return c[1]

    def _fused_table_batch(self, new_seqs):
        """The uncached part of a batch in ONE device round trip (fx_cache_nam_query) when the landscape is a device table of
        packed k-mers (flexs_amd.landscapes.TFBinding) on this model's engine: neighbour search, both look-ups and the blend
        stay on the device; the RNG draws are made here, in query order, exactly as `np.random.exponential(scale=...)`
        makes them.  None = not applicable, or a case whose draws the reference makes differently (a negative neighbour
        value, a sequence missing from the table): the RNG state is put back and the general path below decides."""
        table_of = getattr(self.landscape, "_native_table", None)
        if table_of is None or not getattr(self.landscape, "batch_safe", False) or len(self.cache) == 0 or type(self) is not NoisyAbstractModel:
            return None
        L = getattr(self.landscape, "_L", None)
        if L is None or len(new_seqs) > 32768 or any(len(s) != L for s in new_seqs):
            return None
        try:
            table = table_of()
        except Exception:  # noqa: BLE001 - the general path reports what is wrong with the landscape
            return None
        if not table.bits or table.engine is not _native.Engine.get(self._device):
            return None
        pending = self._sync_device_cache(L, defer=True)              # the previous batch's sequences ride along with this call
        try:
            if self._dev_cache.L != L:
                raise ValueError("cache rows wider than the landscape's sequences")
            rows = _native.sequences_to_bytes(new_seqs, L=L)
        except ValueError:
            if pending is not None:
                self._dev_cache.append(pending)
            return None
        restore = _rng_checkpoint()
        E = np.random.standard_exponential(len(new_seqs))              # scale * E[i] is np.random.exponential(scale)[i]
        alpha_tab = self._alpha_tab(L)
        try:
            fit, dist, arg, flags = self._dev_cache.nam_query(table, rows, E, alpha_tab, self._mode, append=pending)
        except BaseException:
            self._dev_cache = None                                      # (whether the pending keys arrived is unknown: rebuild on next use)
            restore()
            raise
        if flags.any():
            restore()
            return None
        self.landscape.cost += 2 * len(new_seqs)                       # the two oracle queries per sequence (:86-87)
        return fit

    # ---------------------------------------------------------------- flexs.Model API
    def _note_new_keys(self, keys):
        """Call BEFORE `cache.update(zip(keys, ...))`: remembers, in order, the keys that update will append (first occurrences
        of keys not yet cached).  A cache somebody else also writes to is noticed by its length and walked as before."""
        cache = self.cache
        pending = getattr(self, "_pending", None)                      # (an instance restored from an older pickle has none)
        if pending is None or len(getattr(self, "_dev_keys", ())) + len(pending) != len(cache):
            self._pending = []                                         # (out of step already: the next sync walks the dict)
            return
        seen = set()
        for k in keys:
            k = k if type(k) is str else str(k)
            if k not in cache and k not in seen:
                seen.add(k)
                pending.append(k)

    def train(self, sequences: SEQUENCES_TYPE, labels: np.ndarray):
        # the reference's `self.cache.update(zip(sequences, labels))` takes any iterable, also a one-shot one (generator, map, zip):
        # materialise it once -- the note below must not be the pass that uses it up
        if not isinstance(sequences, (list, tuple, np.ndarray)):
            sequences = list(sequences)
        self._note_new_keys(sequences)
        self.cache.update(zip(sequences, labels))                      # :62-67

    def _fitness_function(self, sequences):
        # noisy_abstract_model.py:69-101.  Same decisions in the same order as the reference (which is cached is decided
        # for the whole batch before anything is added; a new sequence that occurs twice is computed twice and the second
        # value stays in the cache), on plain Python strings instead of a NumPy string array: 100 membership tests on
        # np.str_ scalars cost more than the distance kernel.
        seqs = [s if type(s) is str else str(s) for s in sequences]
        cache = self.cache
        fitnesses = np.empty(len(seqs))
        new_idx = []
        for i, s in enumerate(seqs):
            v = cache.get(s, cache)                                    # (the dict itself as the "absent" marker)
            if v is cache:
                new_idx.append(i)
            else:
                fitnesses[i] = v

        if new_idx:
            new_seqs = [seqs[i] for i in new_idx]
            new_fit = self._fused_table_batch(new_seqs)
            if new_fit is not None:
                fitnesses[new_idx] = new_fit
                self._note_new_keys(new_seqs)
                cache.update(zip(new_seqs, new_fit))                   # :99
                return fitnesses
            if len(cache) == 0:                                        # :44-45
                dist = np.zeros(len(new_seqs), np.int32)
                neighbours = new_seqs
            else:
                dist, neighbours = self._min_distances(new_seqs)
            signal = np.empty(len(new_seqs))
            noise = np.empty(len(new_seqs))
            done = False
            if getattr(self.landscape, "batch_safe", False):
                # Deterministic table landscape (e.g. flexs_amd.landscapes.TFBinding): query it in two
                # batches instead of 2*Q one-element calls.  Same values, same total landscape.cost
                # (+2 per query, :86-87); the RNG stream is unchanged because
                # np.random.exponential(scale=array) draws element by element in order.
                sig = self.landscape.get_fitness(new_seqs)
                nbf = self.landscape.get_fitness(neighbours)
                if (nbf >= 0).all():
                    signal[:] = sig
                    noise[:] = np.random.exponential(scale=nbf)
                    done = True
                else:                      # negative neighbour fitness consumes the RNG differently (:90-91)
                    self.landscape.cost -= 2 * len(new_seqs)
            if not done:
                for i, (seq, nb) in enumerate(zip(new_seqs, neighbours)):
                    # same call order as the reference loop (:86-91): two oracle queries, one RNG draw
                    signal[i] = self.landscape.get_fitness([seq]).item()
                    neighbor_fitness = self.landscape.get_fitness([nb]).item()
                    if neighbor_fitness >= 0:
                        noise[i] = np.random.exponential(scale=neighbor_fitness)
                    else:
                        noise[i] = np.random.choice(list(self.cache.values()))
            max_d = int(dist.max()) if len(dist) else 0
            alpha_tab = np.array([self.ss ** d for d in range(max_d + 1)], np.float64)   # :93, Python float pow
            new_fit = self._blend(signal, noise, dist, alpha_tab)
            fitnesses[new_idx] = new_fit
            self._note_new_keys(new_seqs)
            cache.update(zip(new_seqs, new_fit))                       # :99
        return fitnesses
