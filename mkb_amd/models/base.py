"""``BaseModel`` -- parameter tables + the dispatch from ``model(sample, negative_sample, mode)`` to the HIP
scoring kernels (reference: mkb/models/base.py:49-219).

Same constructor, parameter names (``entity_embedding``, ``relation_embedding``, ``gamma``,
``embedding_range``, ``modulus``), init order (entity table first, then relation table, base.py:86-100) and
call signature as the reference; ``forward`` is differentiable through a ``torch.autograd.Function`` whose
backward fills dense gradients exactly like the reference's ``index_select`` backward does.

Two kernel paths sit behind ``forward``:
  * general  -- arbitrary ``negative_sample`` ids (``mkb_score_fwd`` / ``mkb_score_bwd``);
  * pooled   -- ``negative_sample`` produced by ``mkb_amd.sampling.NegativeSampling`` on the device: all rows
                draw from ONE shared pool (negative_sampling.py:166), so each pool row is loaded once per row
                tile instead of once per (row, slot) (``mkb_pool_score_fwd`` / pooled backward).
"""
import math
import pickle

import torch
import torch.nn as nn

from .. import _gradshare, _hip, _links
from ..sampling.negative_sampling import PoolInfo
from ..utils.fmt import aligned_block

__all__ = ["BaseModel"]


# Negatives without a pool description are scanned for one (PoolInfo.discover) when they are big enough for the pooled
# kernels to pay for the scan; set AUTO_POOL = False to always take the general kernels.
AUTO_POOL = True
AUTO_POOL_MIN_SLOTS = 32768
# ``model(sample, negative_sample)`` range-checks user-supplied ids on the device (one small launch, no sync); the result
# surfaces as IndexError at ``model.check_ids()``.  The fused training step and the evaluation draw their ids from the
# dataset / the sampler and skip it.
VALIDATE_IDS = True


class _ScoreFn(torch.autograd.Function):
    """score = model.forward(...) on device; backward = dense d loss / d tables."""

    @staticmethod
    def forward(ctx, ent, rel, modulus, model, sample, cand, mode):
        B = sample.shape[0]
        K = 1 if cand is None else cand.shape[1]
        score = torch.empty((B, K), dtype=torch.float32, device=ent.device)
        tb = model._tables(ent, rel, modulus)
        with _hip.on_device(ent.device):
            _hip.check(_hip.lib().mkb_score_fwd(tb, _hip.ptr(sample), _hip.ptr(cand), B, K, mode, _hip.ptr(score),
                                                _hip.stream_ptr()), "mkb_score_fwd")
        ctx.model, ctx.mode = model, mode
        ctx.save_for_backward(ent, rel, modulus, sample, cand)
        return score

    @staticmethod
    def backward(ctx, dscore):
        ent, rel, modulus, sample, cand = ctx.saved_tensors
        model = ctx.model
        B = sample.shape[0]
        K = 1 if cand is None else cand.shape[1]
        dscore = _hip.contiguous(dscore, torch.float32)
        # (one dense buffer per table and backward pass: the positive and the negative score functions of a step share it)
        # (a row-lazy optimizer's table takes its rows straight into .grad: _gradshare.direct)
        g_ent = _gradshare.direct(model.entity_embedding, ent, lambda: _read_entity_ids(sample, cand)) if _few(cand) else None
        fresh_e = False
        if g_ent is None:
            g_ent, fresh_e = _gradshare.take(model.entity_embedding, ent)
        g_rel, fresh_r = _gradshare.take(model.relation_embedding, rel)
        g_mod = torch.zeros_like(modulus) if model.name == "pRotatE" else None
        tb = model._tables(ent, rel, modulus)
        gr = _hip.Grads(g_ent.data_ptr(), g_rel.data_ptr(), None if g_mod is None else g_mod.data_ptr())
        with _hip.on_device(ent.device):
            n_ws = _hip.lib().mkb_score_bwd_workspace_bytes(tb, B, K, ctx.mode)  # the glue owns every buffer (mkb_hip.h)
            ws = _hip.aligned_bytes(n_ws, ent.device) if n_ws > 0 else None
            _hip.check(_hip.lib().mkb_score_bwd(tb, gr, _hip.ptr(sample), _hip.ptr(cand), B, K, ctx.mode,
                                                _hip.ptr(dscore), _hip.ptr(ws), _hip.stream_ptr()), "mkb_score_bwd")
        return (g_ent if fresh_e else None), (g_rel if fresh_r else None), g_mod, None, None, None, None


# candidate lists up to this many ids are made current row by row in front of a forward pass (and their gradient rows recorded
# one by one behind the backward pass) when the table steps row-lazily; longer ones flush the table / take the dense gradient
DIRECT_MAX_IDS = 16384


def _few(cand):
    return cand is None or cand.numel() <= DIRECT_MAX_IDS


def _read_entity_ids(sample, cand=None):
    """int64 ids of the entity rows a score of (sample, cand) reads: heads, tails (and the candidates)."""
    ht = sample[:, 0::2].reshape(-1)
    return ht if cand is None else torch.cat([ht, cand.reshape(-1)])


class Base(nn.Module):
    """mkb/models/base.py:9-46."""

    @property
    def name(self):
        return self.__class__.__name__

    @property
    def _repr_title(self):
        return f"{self.name} model"

    @property
    def _repr_content(self):
        return {}

    def __repr__(self):
        return aligned_block(self._repr_title, self._repr_content)

    def save(self, path):
        if hasattr(self, "sync_parameters"):
            self.sync_parameters()
        with open(path, "wb") as handle:
            pickle.dump(self.cpu().eval(), handle, protocol=pickle.HIGHEST_PROTOCOL)


class BaseModel(Base):
    def __init__(self, entities, relations, hidden_dim, entity_dim, relation_dim, gamma):
        super().__init__()
        self.entities = {i: e for e, i in entities.items()}
        self.relations = {i: r for r, i in relations.items()}
        self.n_entity = len(entities)
        self.n_relation = len(relations)
        self.hidden_dim = hidden_dim
        self.entity_dim = entity_dim
        self.relation_dim = relation_dim

        self.gamma = nn.Parameter(torch.Tensor([gamma]), requires_grad=False)
        self.epsilon = 2
        self.embedding_range = nn.Parameter(
            torch.Tensor([(self.gamma.item() + self.epsilon) / self.hidden_dim]), requires_grad=False)

        self.entity_embedding = nn.Parameter(torch.zeros(self.n_entity, self.entity_dim))
        nn.init.uniform_(tensor=self.entity_embedding, a=-self.embedding_range.item(), b=self.embedding_range.item())
        self.relation_embedding = nn.Parameter(torch.zeros(self.n_relation, self.relation_dim))
        nn.init.uniform_(tensor=self.relation_embedding, a=-self.embedding_range.item(),
                         b=self.embedding_range.item())
        self._consts = None

    # ------------------------------------------------------------------ host-side constants
    def _constants(self):
        """(gamma, phase_div) as python floats; read once (each .item() is a device sync) and refreshed when
        the parameters are replaced (``_set_params``, ``.to()``, ``load_state_dict``)."""
        override = getattr(self, "_consts_override", None)
        if override is not None:  # dimension shard of a larger model (mkb_amd.parallel.shard_dims)
            return override
        if self._consts is None:
            gamma = self.gamma.item()
            phase_div = torch.tensor(self.embedding_range.item() / math.pi, dtype=torch.float32).item()
            self._consts = (gamma, phase_div)
        return self._consts

    def _apply(self, fn, *args, **kwargs):
        self._consts = None
        return super()._apply(fn, *args, **kwargs)

    def load_state_dict(self, *args, **kwargs):
        self._consts = None
        return super().load_state_dict(*args, **kwargs)

    def _tables(self, ent=None, rel=None, modulus=None):
        ent = self.entity_embedding if ent is None else ent
        rel = self.relation_embedding if rel is None else rel
        if modulus is None:
            modulus = getattr(self, "modulus", None)
        gamma, phase_div = self._constants()
        return _hip.Tables(_hip.MODEL_IDS[self.name], self.hidden_dim, self.n_entity, self.n_relation, self.entity_dim,
                           self.relation_dim, ent.data_ptr(), rel.data_ptr(),
                           None if modulus is None else modulus.data_ptr(), gamma, phase_div)

    def sync_parameters(self):
        """Bring the tables up to date when a row-lazy optimizer (mkb_amd.optim.Adam(lazy_rows=True)) is attached;
        no-op otherwise.  Called before any read of the tables outside the fused training step."""
        for p in (self.entity_embedding, self.relation_embedding):
            opt = _links.owner(p)
            if opt is not None:
                opt.flush(p)

    def _make_current(self, sample, entity_ids):
        """In front of a forward pass: the rows it reads are brought up to date when a row-lazy optimizer steps the tables
        (``entity_ids``: int64 ids, duplicates allowed; ``None`` = too many to list: the whole table is flushed).  One small
        launch per table with something pending -- the whole-table flush this replaces was 150 us per ``model(...)`` call at
        the headline shape."""
        for p, ids in ((self.entity_embedding, entity_ids), (self.relation_embedding, False)):
            opt = _links.owner(p)
            if opt is None:
                continue
            st = opt._state(p)
            st["fwd_n"] = st["n"]  # (the rows read at this step count are current from here on: _gradshare.direct)
            if st["n"] <= 0 or st.get("flushed") == st["n"]:
                continue  # nothing pending
            if ids is False:
                ids = sample[:, 1]
            if ids is None:
                opt.flush(p)
            else:
                opt.catch_up(p, ids)

    # ------------------------------------------------------------------ reference API
    @property
    def embeddings(self):
        self.sync_parameters()
        ent = self.entity_embedding.detach()
        rel = self.relation_embedding.detach()
        return {"entities": {self.entities[i]: ent[i] for i in range(self.n_entity)},
                "relations": {self.relations[i]: rel[i] for i in range(self.n_relation)}}

    @property
    def _repr_content(self):
        return {
            "Entities embeddings dim": f"{self.entity_dim}",
            "Relations embeddings dim": f"{self.relation_dim}",
            "Gamma": f"{self._constants()[0]}",
            "Number of entities": f"{self.n_entity}",
            "Number of relations": f"{self.n_relation}",
        }

    @staticmethod
    def format_sample(sample, negative_sample=None):
        """base.py:131-151."""
        if sample.dim() == 2:
            if negative_sample is None:
                return sample, (sample.size(0), 1)
            return sample, negative_sample.shape
        if sample.dim() == 3:
            return sample.reshape(sample.size(0) * sample.size(1), 3), (sample.size(0), sample.size(1))
        raise ValueError("sample must be [B,3] or [B,M,3]")

    def forward(self, sample, negative_sample=None, mode=None):
        _hip.require_device(self.entity_embedding, sample, negative_sample)
        sample, shape = self.format_sample(sample=sample, negative_sample=negative_sample)
        mode_id = _hip.mode_id(mode)
        sample = _hip.contiguous(sample, torch.int64)
        cand = None
        if sample.shape[0] == 0:
            # an empty batch, as the reference answers it (models/base.py:153-207 + the forward's view): positives -> an empty
            # [0, 1] score, negatives -> RuntimeError (its `view(B, K, -1)` of zero elements is ambiguous).  No launch.
            if mode_id != _hip.MODE_DEFAULT:
                raise RuntimeError("cannot reshape tensor of 0 elements into shape [0, %d, -1]: an empty batch has no negatives to score"
                                   % (negative_sample.shape[-1] if negative_sample is not None and negative_sample.dim() else 0))
            return (self.entity_embedding.sum() * 0.0).expand(0, 1).reshape(shape)  # (differentiable: backward adds nothing)
        if VALIDATE_IDS:
            self._launch_id_check(sample, negative_sample if mode_id != _hip.MODE_DEFAULT else None)
        lazy = _links.owner(self.entity_embedding) is not None or _links.owner(self.relation_embedding) is not None
        if mode_id != _hip.MODE_DEFAULT:
            pooled = getattr(negative_sample, "_mkb_pool", None)
            if (pooled is None and AUTO_POOL and negative_sample.dim() == 2 and negative_sample.shape[1] <= 512
                    and negative_sample.numel() >= AUTO_POOL_MIN_SLOTS and PoolInfo.enabled):
                # negatives from somewhere else (the reference's sampler, a checkpointed batch): look for the shared pool
                pooled = PoolInfo.discover(_hip.contiguous(negative_sample, torch.int64), sample, mode_id)
            if pooled is not None and pooled.usable_for(self, sample, mode_id):
                from ..fused import pooled_forward
                if lazy:
                    self._make_current(sample, pooled.touched if pooled.touched is not None
                                       else torch.cat([pooled.pool, _read_entity_ids(sample)]))
                return pooled_forward(self, sample, pooled, mode_id).view(shape)
            cand = _hip.contiguous(negative_sample, torch.int64)
        if lazy:
            self._make_current(sample, _read_entity_ids(sample, cand) if _few(cand) else None)
        modulus = getattr(self, "modulus", None)
        if modulus is None:  # autograd.Function needs a tensor slot; never read by the kernels
            modulus = self.gamma
        score = _ScoreFn.apply(self.entity_embedding, self.relation_embedding, modulus, self, sample, cand, mode_id)
        return score.view(shape)

    # ------------------------------------------------------------------ id validation
    def _launch_id_check(self, sample, negative_sample):
        """One small launch (``mkb_check_ids``) that flags ids outside the tables; nothing is synchronised here.  The
        reference's ``index_select`` raises IndexError on such ids; the kernels index the tables directly."""
        dev = self.entity_embedding.device
        flag = self.__dict__.get("_id_flag")
        if flag is None or flag.device != dev:
            flag = self.__dict__["_id_flag"] = torch.zeros(1, dtype=torch.int32, device=dev)
        cand = None
        if negative_sample is not None and getattr(negative_sample, "_mkb_pool", None) is None:  # (our sampler's output is trusted)
            cand = _hip.contiguous(negative_sample, torch.int64)
        with _hip.on_device(dev):
            _hip.check(_hip.lib().mkb_check_ids(_hip.ptr(sample), sample.shape[0], _hip.ptr(cand), 0 if cand is None else cand.numel(),
                                                self.n_entity, self.n_relation, _hip.ptr(flag), _hip.stream_ptr()), "mkb_check_ids")

    def check_ids(self):
        """Raise IndexError if any ``model(sample, negative_sample)`` call since the last check used an id outside the
        tables (synchronises; ``compose.Pipeline`` calls it once per epoch, like ``sampling.check()``)."""
        flag = self.__dict__.get("_id_flag")
        if flag is None:
            return
        bits = int(flag.item())
        if bits:
            flag.zero_()
            what = [name for bit, name in ((1, "entity id in sample"), (2, "relation id in sample"), (4, "candidate entity id")) if bits & bit]
            raise IndexError("index out of range in self: " + ", ".join(what)
                             + f" (tables hold {self.n_entity} entities, {self.n_relation} relations)")

    def _set_params(self, entities_embeddings, relations_embeddings, **kwargs):
        """base.py:209-215."""
        self.entity_embedding.data.copy_(entities_embeddings)
        self.relation_embedding.data.copy_(relations_embeddings)
        for parameter, weights in kwargs.items():
            self._parameters[parameter].data.copy_(weights)
        self._consts = None
        return self

    def distill(self, sample, negative_sample=None, mode=None):
        """base.py:217-219."""
        return self(sample=sample, negative_sample=negative_sample, mode=mode)
