"""``NegativeSampling`` -- filtered uniform negative sampler, generated ON THE DEVICE and bit-exact with the
reference (mkb/sampling/negative_sampling.py:130-201) at a given seed.

Reference algorithm: one pool of ``2*size`` candidates per ``generate`` call from
``np.random.RandomState(seed).randint`` (:166); every batch row filters that same pool against its set of true
heads / tails with ``np.in1d(..., assume_unique=True, invert=True)`` (:153-156) and takes the survivors
cyclically up to ``size`` (:176-199).  Here the MT19937 stream, the masked-rejection draw, the three
``np.in1d`` branches and the per-row compaction all run in ``libmkb_hip.so`` (``mkb_sampler_generate``):
no ``.item()`` per row, no host loop, no H2D copy of the negatives.

Host side (this file) only builds the two filter dictionaries as CSR once (vectorised numpy instead of the
reference's python dict loop, :7-28) and owns the device handle.

The returned LongTensor ``[B, size]`` additionally carries ``._mkb_pool`` (pool ids, position map, per-position
multiplicities) so that ``model(sample, negative_sample, mode)`` can take the pooled scoring path.

Differences by design: an unseen ``(relation, tail)`` / ``(head, relation)`` raises ``KeyError`` like the
reference, but lazily (at the next ``check()``, ``Pipeline`` calls it when it reads the loss) because there is
no per-batch host sync; a row whose filter empties the pool raises ``RuntimeError`` where the reference spins
forever.
"""
import ctypes

import numpy as np
import torch

from .. import _hip

__all__ = ["NegativeSampling", "positive_triples"]


def positive_triples(triples):
    """Filter dictionaries ``true_head[(r,t)] -> heads``, ``true_tail[(h,r)] -> tails`` as numpy arrays
    (reference negative_sampling.py:7-28; only membership / len / min / max of each array matter)."""
    th, tt = _filter_csr(triples)
    return _csr_to_dict(*th), _csr_to_dict(*tt)


def _csr(a, b, v, stride):
    """Group values v by key a*stride+b -> (sorted unique keys, offsets, values sorted & unique per key)."""
    key = a * stride + b
    order = np.lexsort((v, key))
    key, v = key[order], v[order]
    if len(key):
        first = np.ones(len(key), dtype=bool)
        first[1:] = (key[1:] != key[:-1]) | (v[1:] != v[:-1])
        key, v = key[first], v[first]
    ukeys, start = np.unique(key, return_index=True)
    offsets = np.concatenate([start, [len(key)]]).astype(np.int64)
    return (np.ascontiguousarray(ukeys, dtype=np.int64), offsets, np.ascontiguousarray(v, dtype=np.int64), stride)


def _filter_csr(triples, n_entity=None, n_relation=None):
    t = np.asarray(triples, dtype=np.int64).reshape(-1, 3)
    n_entity = int(max(t[:, 0].max(), t[:, 2].max())) + 1 if n_entity is None else n_entity
    n_relation = int(t[:, 1].max()) + 1 if n_relation is None else n_relation
    head = _csr(t[:, 1], t[:, 2], t[:, 0], n_entity)      # key (r, t) -> heads
    tail = _csr(t[:, 0], t[:, 1], t[:, 2], n_relation)    # key (h, r) -> tails
    return head, tail


def _csr_to_dict(keys, offsets, values, stride):
    return {(int(k // stride), int(k % stride)): values[offsets[i]: offsets[i + 1]] for i, k in enumerate(keys)}


class PoolInfo:
    """Side outputs of one ``generate`` call consumed by the pooled scoring path."""

    enabled = False  # flipped on by mkb_amd.fused once the pooled kernels are loaded

    def __init__(self, pool, pos, cnt, size, mode_id, sample):
        self.pool, self.pos, self.cnt, self.size, self.mode_id = pool, pos, cnt, size, mode_id
        self.sample_ptr, self.batch = sample.data_ptr(), sample.shape[0]
        self.touched = None  # [2K + 2B] entity rows a step on this batch reads: pool | heads | tails

    @classmethod
    def discover(cls, negative_sample, sample, mode_id):
        """Shared-pool description of ARBITRARY negatives ``[B, K]`` (e.g. the reference's own CPU sampler's output moved
        to the device): if the batch draws on at most 2K distinct entities -- which is what mkb's sampler always
        produces, one pool per batch -- the pooled kernels apply.  Costs one ``torch.unique`` (a sort and a host sync);
        returns ``None`` when the negatives are too diverse."""
        B, K = negative_sample.shape
        uniq, inv = torch.unique(negative_sample, return_inverse=True)
        U = uniq.numel()
        if U > 2 * K:
            return None
        dev = negative_sample.device
        pool = torch.zeros(2 * K, dtype=torch.int64, device=dev)
        pool[:U] = uniq
        cnt = torch.zeros((B, 2 * K), dtype=torch.int32, device=dev)
        cnt.scatter_add_(1, inv, torch.ones_like(inv, dtype=torch.int32))
        return cls(pool, inv.to(torch.int32).contiguous(), cnt.to(torch.uint16), K, mode_id, sample)

    def usable_for(self, model, sample, mode_id):
        return (self.enabled and mode_id == self.mode_id and sample.shape[0] == self.batch
                and _hip.lib().mkb_pool_supported(model._tables(), self.batch, self.size)
                and sample.data_ptr() == self.sample_ptr and self.pool.device == model.entity_embedding.device)


class NegativeSampling:
    def __init__(self, size, train_triples, entities, relations, seed=42, rng="numpy"):
        """``rng="numpy"`` (default): the reference's ``np.random.RandomState(seed).randint`` stream, bit-exact negatives.
        ``rng="rocrand"``: opt-in, NOT the reference's numbers -- the pool comes from rocRAND's Philox4x32-10 inside the same
        draw kernel (counter based: no generator state on the device, every lane draws its own entry; the filter, the
        cyclic fill and every other semantic are unchanged)."""
        if rng not in ("numpy", "rocrand"):
            raise ValueError("rng must be 'numpy' (bit-exact with the reference) or 'rocrand'")
        self.rng = rng
        self.size = size
        self.n_entity = len(entities)
        self.n_relation = len(relations)
        self.seed = seed
        self._head_csr, self._tail_csr = _filter_csr(train_triples, self.n_entity, self.n_relation)
        self._handle = None
        self._device = None
        self._dicts = None

    # reference attributes (built lazily: the device path never needs the python dicts)
    @property
    def true_head(self):
        if self._dicts is None:
            self._dicts = (_csr_to_dict(*self._head_csr), _csr_to_dict(*self._tail_csr))
        return self._dicts[0]

    @property
    def true_tail(self):
        _ = self.true_head
        return self._dicts[1]

    def _ensure_handle(self, device):
        if self._handle is not None:
            if device != self._device:
                raise RuntimeError(f"sampler lives on {self._device}, sample is on {device}")
            return
        hk, ho, hv, _ = self._head_csr
        tk, to, tv, _ = self._tail_csr
        handle = ctypes.c_void_p()
        p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        with _hip.on_device(device):
            _hip.check(_hip.lib().mkb_sampler_create(ctypes.byref(handle), self.n_entity, self.n_relation, self.size,
                                                     int(self.seed) & 0xFFFFFFFF, p(hk), len(hk), p(ho), p(hv), p(tk),
                                                     len(tk), p(to), p(tv), _hip.stream_ptr()), "mkb_sampler_create")
        self._handle, self._device = handle, device
        if self.rng == "rocrand":
            _hip.check(_hip.lib().mkb_sampler_set_rng(handle, 1, int(self.seed) & 0xFFFFFFFFFFFFFFFF, 0), "mkb_sampler_set_rng")

    def generate(self, sample, mode):
        """-> LongTensor [B, size] on ``sample``'s device (reference: CPU tensor; ``.to(device)`` is then free)."""
        if mode not in ("head-batch", "tail-batch"):
            raise ValueError("mode must be 'head-batch' or 'tail-batch'")
        origin = sample.device
        if not sample.is_cuda:
            sample = sample.cuda()
        sample = _hip.contiguous(sample, torch.int64)
        dev = sample.device
        self._ensure_handle(dev)
        B, K = sample.shape[0], self.size
        if B == 0:
            self._empty_batch()
        neg = torch.empty((B, K), dtype=torch.int64, device=dev)
        pool = torch.empty(2 * K, dtype=torch.int64, device=dev)
        pos = torch.empty((B, K), dtype=torch.int32, device=dev)
        cnt = torch.empty((B, 2 * K), dtype=torch.uint16, device=dev)
        touched = torch.empty(2 * K + 2 * B, dtype=torch.int64, device=dev)  # pool | heads | tails (row-lazy Adam)
        mode_id = _hip.mode_id(mode)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_sampler_generate(self._handle, _hip.ptr(sample), B, mode_id, _hip.ptr(neg),
                                                       _hip.ptr(pool), _hip.ptr(pos), _hip.ptr(cnt), _hip.ptr(touched),
                                                       _hip.stream_ptr()), "mkb_sampler_generate")
        if origin != dev:
            self.check()
            return neg.to(origin)
        neg._mkb_pool = PoolInfo(pool, pos, cnt, K, mode_id, sample)
        neg._mkb_pool.touched = touched
        return neg

    def generate_with_catch_up(self, sample, mode, optimizer, param):
        """``generate(sample, mode)`` and ``optimizer.catch_up(param, rows this batch touches)`` as ONE launch that also
        draws the next pool (``mkb_adam_rows_catchup_generate``): the sampler costs no launch of its own.  ``optimizer``:
        a ``mkb_amd.optim.Adam(lazy_rows=True)`` holding ``param`` (the entity table) row-lazily.  Same negatives, pool
        and multiplicities, bit for bit, as ``generate``."""
        if mode not in ("head-batch", "tail-batch"):
            raise ValueError("mode must be 'head-batch' or 'tail-batch'")
        sample = _hip.contiguous(sample, torch.int64)
        _hip.require_device(param, sample)
        dev = sample.device
        self._ensure_handle(dev)
        B, K = sample.shape[0], self.size
        neg = torch.empty((B, K), dtype=torch.int64, device=dev)
        pool = torch.empty(2 * K, dtype=torch.int64, device=dev)
        pos = torch.empty((B, K), dtype=torch.int32, device=dev)
        cnt = torch.empty((B, 2 * K), dtype=torch.uint16, device=dev)
        touched = torch.empty(2 * K + 2 * B, dtype=torch.int64, device=dev)
        mode_id = _hip.mode_id(mode)
        optimizer.catch_up_generate(param, self._handle, sample, B, mode_id, neg, pool, pos, cnt, touched)
        neg._mkb_pool = PoolInfo(pool, pos, cnt, K, mode_id, sample)
        neg._mkb_pool.touched = touched
        return neg

    def generate_with_sharded_catch_up(self, sample, mode, optimizer, param, world, rank, local_ids):
        """``generate(sample, mode)`` for this rank's rows of a global batch and ``optimizer.catch_up_sharded(param, pool, world,
        rank, local_ids)`` for a ROW SHARD of the entity table (``mkb_amd.table_rows``) as one launch that also draws the next
        pool.  Same negatives, pool and multiplicities, bit for bit, as ``generate``."""
        if mode not in ("head-batch", "tail-batch"):
            raise ValueError("mode must be 'head-batch' or 'tail-batch'")
        sample = _hip.contiguous(sample, torch.int64)
        _hip.require_device(param, sample)
        dev = sample.device
        self._ensure_handle(dev)
        B, K = sample.shape[0], self.size
        neg = torch.empty((B, K), dtype=torch.int64, device=dev)
        pool = torch.empty(2 * K, dtype=torch.int64, device=dev)
        pos = torch.empty((B, K), dtype=torch.int32, device=dev)
        cnt = torch.empty((B, 2 * K), dtype=torch.uint16, device=dev)
        touched = torch.empty(2 * K + 2 * B, dtype=torch.int64, device=dev)  # (global ids: unused by the sharded caller)
        mode_id = _hip.mode_id(mode)
        optimizer.catch_up_sharded_generate(param, world, rank, local_ids, self._handle, sample, B, mode_id, neg, pool, pos, cnt, touched)
        neg._mkb_pool = PoolInfo(pool, pos, cnt, K, mode_id, sample)
        return neg

    def check(self):
        """Raise what the reference would have raised for the batches generated so far (synchronises)."""
        if self._handle is None:
            return
        with _hip.on_device(self._device):
            rc = _hip.lib().mkb_sampler_status(self._handle, _hip.stream_ptr())
        if rc == 0:
            return
        msg = _hip.lib().mkb_last_error().decode()
        if rc == _hip.ERR_KEY:
            raise KeyError(msg)
        raise RuntimeError(msg)

    def _empty_batch(self):
        """An empty batch, as the reference answers it (sampling/negative_sampling.py:166-201): the batch's pool IS drawn, then
        ``torch.stack`` of no rows raises RuntimeError.  The generator moves on by one pool here as well, so a caller that
        catches the error stays in step with the reference's stream."""
        if self.rng == "rocrand":
            _, (seed, draws) = self.get_state()
            self.set_state("rocrand", (seed, draws + 1))
        else:
            key, pos = self.get_state()
            rs = np.random.RandomState()
            rs.set_state(("MT19937", np.asarray(key, dtype=np.uint32), int(pos)))
            rs.randint(self.n_entity, size=2 * self.size)
            st = rs.get_state()
            self.set_state(st[1], int(st[2]))
        raise RuntimeError("stack expects a non-empty TensorList (an empty batch has no rows to sample negatives for)")

    # ---- RNG state (numpy MT19937 key + position), e.g. for checkpoint/resume or multi-GPU replication
    def get_state(self):
        if self.rng == "rocrand":  # counter based: (seed, pools drawn so far) is the whole state
            if self._handle is None:
                return "rocrand", (int(self.seed), 0)
            kind, seed, draws = ctypes.c_int(), ctypes.c_uint64(), ctypes.c_uint64()
            _hip.check(_hip.lib().mkb_sampler_get_rng(self._handle, ctypes.byref(kind), ctypes.byref(seed), ctypes.byref(draws)),
                       "mkb_sampler_get_rng")
            return "rocrand", (int(seed.value), int(draws.value))
        if self._handle is None:
            st = np.random.RandomState(self.seed).get_state()
            return st[1].astype(np.uint32), int(st[2])
        key = np.empty(624, dtype=np.uint32)
        pos = ctypes.c_int32()
        with _hip.on_device(self._device):
            _hip.check(_hip.lib().mkb_sampler_get_state(self._handle, key.ctypes.data_as(ctypes.c_void_p),
                                                        ctypes.byref(pos), _hip.stream_ptr()), "mkb_sampler_get_state")
        return key, pos.value

    def set_state(self, key, pos, device=None):
        if isinstance(key, str):  # ("rocrand", (seed, draws)) from get_state()
            if key != "rocrand" or self.rng != "rocrand":
                raise ValueError("generator state of another kind than this sampler's")
            if self._handle is None:
                self._ensure_handle(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
            _hip.check(_hip.lib().mkb_sampler_set_rng(self._handle, 1, int(pos[0]) & 0xFFFFFFFFFFFFFFFF, int(pos[1])), "mkb_sampler_set_rng")
            return
        key = np.ascontiguousarray(key, dtype=np.uint32)
        if self._handle is None:
            self._ensure_handle(torch.device("cuda", torch.cuda.current_device()) if device is None else device)
        with _hip.on_device(self._device):
            _hip.check(_hip.lib().mkb_sampler_set_state(self._handle, key.ctypes.data_as(ctypes.c_void_p), int(pos),
                                                        _hip.stream_ptr()), "mkb_sampler_set_state")

    def __del__(self):
        try:
            if self._handle is not None:
                _hip.lib().mkb_sampler_destroy(self._handle)
                self._handle = None
        except Exception:
            pass
