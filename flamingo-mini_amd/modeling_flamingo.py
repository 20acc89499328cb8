"""FlamingoModel drop-in (reference: flamingo_mini/modeling_flamingo.py).

Same public surface — FlamingoBaseModel / FlamingoGPT2 / FlamingoOPT / FlamingoModel, forward() arguments, freeze_*,
parameters_trainable(), state_dict_trainable(), prepare_inputs_for_generation(), generate_captions(),
score_sequences() — and the same parameter names, so `state_dict_trainable()` checkpoints interchange.
The frozen CLIP and GPT-2 / OPT backbones stay stock Hugging Face modules on PyTorch-ROCm; the perceiver
resampler and the gated cross-attention blocks hooked into the LM layer loop run in libflamingo_fusion.
"""
from __future__ import annotations

import contextlib
import logging
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as TF
from transformers import PreTrainedModel
from transformers.modeling_outputs import CausalLMOutputWithPast

from . import functional as F
from .backbones import load_language_model, load_vision_encoder
from .configuration_flamingo import FlamingoConfig
from .gated_cross_attention import ModifiedLMBlock
from .perceiver_resampler import PerceiverResampler
from .utils import get_common_prefix_length


@contextlib.contextmanager
def suppress_model_loading_warnings(suppress: bool = True):
    logger = logging.getLogger("transformers.modeling_utils")
    level = logger.level
    if suppress:
        logger.setLevel(logging.CRITICAL)
    try:
        yield
    finally:
        logger.setLevel(level)


def _repeat_leading(t: torch.Tensor, m: int) -> torch.Tensor:
    """'n ... -> (n m) ...' (each row repeated m times, as beam search does with input_ids)."""
    return t.repeat_interleave(m, dim=0)


def _add_eoc_row(base_lm) -> None:
    """One extra embedding row for <EOC> (reference :315, :343: resize_token_embeddings(vocab_size + 1)).  from_pretrained builds the model
    on the meta device first (transformers >= 5); the mean / covariance initialisation of the new row cannot run there and is pointless
    (the checkpoint overwrites it), so it is only applied to real tensors."""
    real = base_lm.get_input_embeddings().weight.device.type != "meta"
    base_lm.resize_token_embeddings(base_lm.config.vocab_size + 1, mean_resizing=real)


class FlamingoBaseModel(ABC, PreTrainedModel):
    """CLIP -> PerceiverResampler -> LM whose layers were wrapped by ModifiedLMBlock (reference :43-306)."""

    config_class = FlamingoConfig
    _supports_sdpa = True

    def __init__(self, config: FlamingoConfig, suppress_warnings: bool = True):
        assert isinstance(config, FlamingoConfig)
        super().__init__(config)
        with suppress_model_loading_warnings(suppress_warnings):
            self.vision_encoder = load_vision_encoder(config)
        self.resampler = PerceiverResampler(
            dim=config.dim_visual, depth=config.resampler_depth, dim_head=config.resampler_dim_head,
            heads=config.resampler_heads, num_latents=config.resampler_num_latents,
            num_time_embeds=config.resampler_num_time_embeds, ff_mult=config.resampler_ff_mult, act=config.resampler_act)
        # Launch structure of the fusion path - plain attributes of the model (no environment variable), see set_launch_structure():
        # keys / values of all cross-attention layers in grouped launches ahead of the LM (functional.kv_project): 41.5 -> 40.2 ms per
        # step at the benchmark configuration; .hoist_kv = False restores the per-layer projection
        self.hoist_kv = True
        # 0: ONE projection call (one autograd node, one gradient bucket) for all layers; n > 0: one call per n consecutive layers, so that
        # under data parallelism the to_kv gradients of the upper layers are final - and their all-reduce starts - while backward is still
        # working on the lower ones (the data-parallel reducers set 4 = one grouped launch per call; a single bucket of all 36 to_kv
        # weights, 75 MB at flamingo-mini's size, would only become ready at the very end of backward)
        self.kv_project_group = 0

    def set_launch_structure(self, *, hoist_kv: Optional[bool] = None, kv_project_group: Optional[int] = None,
                             defer_wgrad: Optional[bool] = None, wgrad_group: "Optional[int] | str" = "keep") -> Dict[str, Any]:
        """How the fusion path batches its launches (results are identical for every setting; only the launch / gradient-bucket structure
        changes).  hoist_kv / kv_project_group: see __init__.  defer_wgrad / wgrad_group: the blocks' weight gradients run as launches
        grouped over `wgrad_group` consecutive layers (None = the library default of 12; data-parallel reducers use 4 so that a gradient
        bucket becomes final every four layers of backward).  Returns the PREVIOUS settings as a dict that can be passed back
        (`model.set_launch_structure(**previous)`) - data_parallel.GradientAllReducer.close() does exactly that."""
        blocks = [h.xattn_block for h in self.get_modified_layers()]
        prev = dict(hoist_kv=self.hoist_kv, kv_project_group=self.kv_project_group,
                    defer_wgrad=blocks[0].defer_wgrad if blocks else True, wgrad_group=blocks[0].wgrad_group if blocks else None)
        if hoist_kv is not None:
            self.hoist_kv = bool(hoist_kv)
        if kv_project_group is not None:
            self.kv_project_group = int(kv_project_group)
        for b in blocks:
            if defer_wgrad is not None:
                b.defer_wgrad = bool(defer_wgrad)
            if wgrad_group != "keep":
                b.wgrad_group = None if wgrad_group is None else int(wgrad_group)
        return prev

    def install_autograd_cuts(self, cuts, segment_layers: int = 4) -> None:
        """graphs.PiecewiseGraphedTrainStep: make the visual features and the hidden state in front of every `segment_layers`-th gated
        layer cut points of the autograd graph (`cuts.cut`), so that backward runs - and is captured - segment by segment.  None removes them."""
        self._autograd_cuts = cuts
        seg = max(1, int(segment_layers))
        if cuts is not None:
            # a K / V projection call must not serve layers of two segments (its backward node would be entered by both): one call per
            # segment, or per whole fraction of one
            if not (self.kv_project_group > 0 and seg % self.kv_project_group == 0):
                self._kv_group_before_cuts = (self.kv_project_group, seg)
                self.kv_project_group = seg
        elif hasattr(self, "_kv_group_before_cuts"):
            before, set_to = self._kv_group_before_cuts
            if self.kv_project_group == set_to:     # still what the cuts set (a reducer's close() may have restored its own value meanwhile)
                self.kv_project_group = before
            del self._kv_group_before_cuts
        for i, hook in enumerate(self.get_modified_layers()):
            hook.autograd_cut = cuts.cut if (cuts is not None and i > 0 and i % max(1, segment_layers) == 0) else None
        # a resampler that runs layer by layer (data parallelism: one gradient bucket per layer) makes every layer its own backward segment
        self.resampler.autograd_cut = cuts.cut if (cuts is not None and getattr(self.resampler, "layerwise", False)) else None

    def _init_weights(self, module):  # backbones initialise themselves; fusion modules use torch defaults like the reference
        pass

    def _init_layers(self, lm_layers: nn.ModuleList):
        """Wrap every `xattn_every`-th LM layer in place (reference :76-94)."""
        c = self.config
        for i in range(0, len(lm_layers), c.xattn_every):
            lm_layers[i] = ModifiedLMBlock(
                lm_layers[i], dim=c.dim, dim_visual=c.dim_visual, dim_head=c.xattn_dim_head, heads=c.xattn_heads,
                ff_mult=c.xattn_ff_mult, act=c.xattn_act, n_visual=c.resampler_num_latents)

    @abstractmethod
    def get_modified_layers(self) -> List[ModifiedLMBlock]:
        raise NotImplementedError

    # ---- freezing / trainable views (reference :100-138) ----
    def freeze_vm(self):
        self.vision_encoder.requires_grad_(False)

    def freeze_lm(self):
        """Freeze the LM except the (tied) token embedding and the gated xattn blocks."""
        self.lm.requires_grad_(False)
        self.lm.get_input_embeddings().weight.requires_grad = True
        for hook in self.get_modified_layers():
            hook.xattn_block.requires_grad_(True)

    def unfreeze_lm(self):
        self.lm.requires_grad_(True)

    def state_dict_trainable(self) -> Dict[str, torch.Tensor]:
        names = {n for n, p in self.named_parameters() if p.requires_grad}
        return {k: v for k, v in self.state_dict().items() if k in names}

    def parameters_trainable(self):
        return (p for p in self.parameters() if p.requires_grad)

    # ---- vision side (reference :140-181) ----
    def encode_resample_visuals(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """(N c h w) | (b N c h w) | (b N T c h w)  ->  (b N q d)."""
        if pixel_values.ndim == 4:
            b, N, T = 1, pixel_values.size(0), 1
        elif pixel_values.ndim == 5:
            b, N, T = pixel_values.size(0), pixel_values.size(1), 1
        elif pixel_values.ndim == 6:
            b, N, T = pixel_values.shape[:3]
        else:
            raise ValueError("pixel_values must have ndim 5 or 6!")
        frames = pixel_values.reshape(b * N * T, *pixel_values.shape[-3:])
        with torch.no_grad():
            feats = self.vision_encoder(frames).last_hidden_state                  # ((b N T), v, d), frozen
        feats = feats.reshape(b * N, T, feats.shape[-2], feats.shape[-1])          # frames go to the KV axis inside the resampler
        latents = self.resampler(feats)                                            # ((b N), q, d)
        return latents.reshape(b, N, latents.shape[-2], latents.shape[-1])

    # ---- forward (reference :183-306) ----
    def forward(self, input_ids=None, attention_mask=None, media_locations=None, pixel_values=None, visual_features=None,
                head_mask=None, inputs_embeds=None, use_cache: bool = False, past_key_values=None, return_dict: bool = True,
                labels=None, loss_reduction: str = "mean", text_time=None, **kwargs) -> CausalLMOutputWithPast:
        assert return_dict, "can only use return_dict=True at the moment!"
        assert (input_ids is None) != (inputs_embeds is None), "you must pass either input_ids or inputs_embeds!"
        ref = input_ids if input_ids is not None else inputs_embeds
        batch_size, seq_length = ref.shape[:2]
        device = ref.device
        xattn_past = None if past_key_values is None else past_key_values[0]
        lm_past = None if past_key_values is None else past_key_values[1]

        if visual_features is None:
            if xattn_past is None and pixel_values is not None:
                assert pixel_values.size(0) == batch_size, "pixel_values must have the same batch size as the textual input!"
                visual_features = self.encode_resample_visuals(pixel_values)
            else:  # cached K/V make the features irrelevant; only the shape is used
                visual_features = torch.zeros((batch_size, 1, self.config.resampler_num_latents, self.config.dim_visual),
                                              dtype=self.resampler.latents.dtype, device=device)
        if self.training and (getattr(self.lm, "gradient_checkpointing", False) or any(getattr(m, "gradient_checkpointing", False) for m in self.lm.modules())):
            # The conditioning of the hooks (and the K / V projected up front for every layer) lives for ONE call and is dropped when the LM
            # returns; a recomputation of LM blocks during backward would re-enter the hooks without it.
            raise NotImplementedError("activation checkpointing of the language model is not supported by the fused cross-attention hooks: "
                                      "their conditioning (visual features, hoisted K / V) is released when forward() returns. "
                                      "Disable gradient checkpointing (the frozen LM keeps few activations: only the trainable blocks save theirs).")
        if self.training and device.type == "cuda" and not torch.cuda.is_current_stream_capturing():
            # whatever optimizer drives an eager training loop: the error word of the fused cross-attention kernels' in-launch hand-offs is looked
            # at once per forward (non-blocking: the value the previous call copied to pinned memory) - raises SyncExchangeTimeout
            F.poll_sync_exchange("FlamingoBaseModel.forward")
        cuts = getattr(self, "_autograd_cuts", None)
        if cuts is not None and xattn_past is None:
            visual_features = cuts.cut(visual_features)      # (PiecewiseGraphedTrainStep: the resampler's backward becomes its own segment)
        if text_time is None:
            if media_locations is None:
                media_locations = torch.zeros((batch_size, seq_length), dtype=torch.int, device=device)
            text_time = F.text_time(media_locations)   # once per step, shared by every block (the reference recomputes it per layer)
        # (text_time given: a decode step of the static-cache path - generated tokens are never media tags, so the (b, 1) value of the
        # prompt's last token is valid for every later position and no per-step cumsum over a growing tensor is needed)
        hooks = self.get_modified_layers()
        hoisted = None
        if xattn_past is None and self.hoist_kv:
            # every layer's to_kv sees the same visual features: project for all layers in grouped launches (functional.kv_project)
            weights = [h.xattn_block.attn.to_kv.weight for h in hooks]
            cdt = F.autocast_compute_dtype(visual_features)
            if cdt is not None:              # torch.autocast over fp32 parameters: casts in the kernels' dtype (functional.autocast_compute_dtype)
                weights = F.autocast_params(weights, cdt)
            vf_cast = visual_features.to(weights[0].dtype)
            step = self.kv_project_group if self.kv_project_group > 0 else len(weights)
            hoisted = [kv for g in range(0, len(weights), step) for kv in F.kv_project(vf_cast, weights[g:g + step])]
        for i, hook in enumerate(hooks):
            hook.condition(visual_features, media_locations, None if xattn_past is None else xattn_past[i], text_time=text_time,
                           hoisted_kv=None if hoisted is None else hoisted[i])

        lm_kwargs = dict(input_ids=input_ids, attention_mask=attention_mask, inputs_embeds=inputs_embeds, use_cache=use_cache,
                         past_key_values=lm_past, return_dict=True, **kwargs)
        if head_mask is not None:
            lm_kwargs["head_mask"] = head_mask
        try:
            out = self.lm(**lm_kwargs)
        finally:
            # The conditioning is per call: dropping it here releases the hoisted K / V of every layer and - more importantly - the autograd
            # graph behind them.  The reference leaves it on the hooks until the next call; a graph kept alive that way pins the parameters'
            # AccumulateGrad nodes to the stream of THIS forward, which breaks a later capture of the step on another stream (graphs.py).
            for hook in hooks:
                hook.condition(None, None)
        logits = self.lm_head(out.last_hidden_state)

        new_xattn_past = [hook.kv_output for hook in hooks] if use_cache else None
        for hook in hooks:
            hook.kv_output = None

        loss = None
        if labels is not None:   # tokens < n predict n (reference :288-298)
            if logits.is_cuda and logits.dtype in (torch.float32, torch.bfloat16) and logits.ndim == 3 and seq_length > 1:
                loss = F.shifted_cross_entropy(logits, labels, reduction=loss_reduction)      # two fused passes over the logits
            else:   # host-side / exotic dtypes: the stock op (the loss is not part of the accelerated fusion path)
                flat = logits[..., :-1, :].contiguous().view(-1, logits.size(-1))
                if flat.dtype in (torch.bfloat16, torch.float16):
                    flat = flat.float()
                loss = TF.cross_entropy(flat, labels[..., 1:].contiguous().view(-1), reduction=loss_reduction)

        return CausalLMOutputWithPast(
            loss=loss, logits=logits,
            past_key_values=(tuple(new_xattn_past), out.past_key_values) if use_cache else None,
            hidden_states=getattr(out, "hidden_states", None), attentions=getattr(out, "attentions", None))


class FlamingoGPT2(FlamingoBaseModel):
    # GPT2LMHeadModel ties lm_head to the token embedding; declared here so that save_pretrained (safetensors: no aliased tensors) writes the
    # embedding once and from_pretrained re-ties it (transformers >= 5 refuses to save undeclared shared tensors)
    _tied_weights_keys = {"lm_head.weight": "lm.wte.weight"}

    def __init__(self, config: FlamingoConfig):
        assert config.lm.startswith("gpt")
        super().__init__(config)
        base_lm = load_language_model(config)
        assert config.dim == base_lm.config.n_embd, \
            f"specified {config.dim=} in FlamingoConfig, but {config.lm} has hidden size={base_lm.config.n_embd}"
        _add_eoc_row(base_lm)
        self.lm = base_lm.transformer
        self.lm_head = base_lm.lm_head
        self._init_layers(self.lm.h)
        self.post_init()

    def get_modified_layers(self):
        return [layer for layer in self.lm.h if isinstance(layer, ModifiedLMBlock)]


class FlamingoOPT(FlamingoBaseModel):
    _tied_weights_keys = {"lm_head.weight": "lm.decoder.embed_tokens.weight"}

    def __init__(self, config: FlamingoConfig):
        assert config.lm.startswith("facebook/opt")
        super().__init__(config)
        base_lm = load_language_model(config)
        assert config.dim == base_lm.config.hidden_size, \
            f"specified {config.dim=} in FlamingoConfig, but {config.lm} has hidden size={base_lm.config.hidden_size}"
        _add_eoc_row(base_lm)
        self.lm = base_lm.model
        self.lm_head = base_lm.lm_head
        self._init_layers(self.lm.decoder.layers)
        self.post_init()

    def get_modified_layers(self):
        return [layer for layer in self.lm.decoder.layers if isinstance(layer, ModifiedLMBlock)]


# Decode sessions hold HIP graphs and raw parameter addresses: they live in a weak side table, not in the model's __dict__, so
# copy.deepcopy(model) / pickling (EMA or evaluation copies) neither see nor try to copy them.  The table is keyed by the model and a
# session refers to its model only WEAKLY (a strong reference from the value would keep the key - and with it the model, its static KV
# cache and the captured graph - alive for ever): `del model` frees all of it.
import weakref
_DECODE_SESSIONS: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


class _DecodeSession:
    """Fixed-shape greedy decoding for one (batch, max_length) shape: the LM keeps a transformers StaticCache of max_length positions, the
    attention mask and the token buffer are preallocated, positions travel as a device tensor (`cache_position`), the cross-attention K / V
    of the prompt step are copied into persistent buffers, and the text_time of a generated token is the prompt's last value (generated
    tokens are never media tags).  A decode step then issues exactly the same work on the same addresses every time, so on the GPU it is
    captured once into a HIP graph and replayed (the eager step is host-bound: 72 layers, ~800 launches of a few microseconds each).
    If the capture raises (e.g. a host synchronisation inside the stock LM) the steps run eagerly.  Token-for-token equal to the
    growing-cache loop of FlamingoModel.generate (tests/test_model_plumbing.py)."""

    def __init__(self, model, b, max_length, device, ids_dtype, am_dtype, eos, pad, graph):
        from transformers.cache_utils import StaticCache
        self._model_ref = weakref.ref(model)
        self.b, self.max_length, self.eos, self.graph_wanted = b, max_length, eos, graph
        self.fill = pad if pad is not None else 0
        self.lm_family = "gpt2" if isinstance(model.flamingo, FlamingoGPT2) else "opt"
        self.cache = StaticCache(config=model.flamingo.lm.config, max_cache_len=max_length)
        self.ids_buf = torch.empty((b, max_length), dtype=ids_dtype, device=device)
        self.am_buf = torch.empty((b, max_length), dtype=am_dtype, device=device)
        self.finished = torch.zeros(b, dtype=torch.bool, device=device)
        self.n_new = torch.zeros((), dtype=torch.long, device=device)          # tokens appended while not every sequence had finished
        self.pos = torch.zeros((1,), dtype=torch.long, device=device)          # position of `tok` (the token the next step consumes)
        self.tok = torch.zeros((b, 1), dtype=ids_dtype, device=device)
        self.tt_step = torch.zeros((b, 1), dtype=torch.int32, device=device)
        self.xattn_past = None                                                 # persistent (k, v) per layer, filled by every prompt step
        self.replay = None
        self.capture_failed = False
        self.param_ptrs = self._param_ptrs()                                   # what a captured graph reads: checked before every reuse

    @property
    def model(self):
        m = self._model_ref()
        if m is None:
            raise RuntimeError("decode session used after its model was deleted")
        return m

    def _param_ptrs(self):
        return tuple(p.data_ptr() for p in self.model.parameters())

    def stale(self) -> bool:
        """Parameters were re-allocated since this session was built (model.to() / .half(), ShardedAdamW moving them into flat buffers):
        a captured graph would replay reads of freed memory."""
        return self.param_ptrs != self._param_ptrs()

    def _append(self, logits):             # choose, apply the eos bookkeeping of generate(), append at `pos`
        nxt = logits.float().argmax(-1)
        alive = ~self.finished.all()
        if self.eos is not None:
            nxt = torch.where(self.finished, torch.full_like(nxt, self.fill), nxt)
            self.finished.logical_or_(nxt == self.eos)
        self.tok.copy_(nxt[:, None])
        self.ids_buf.index_copy_(1, self.pos, self.tok)
        self.n_new.add_(alive.to(self.n_new.dtype))

    def _positions(self, L0=None):
        """How the LM family learns where a step's tokens sit without reading anything back to the host.  GPT-2 takes absolute cache
        positions.  OPT numbers the ATTENDED tokens (cumsum of the attention mask, padding skipped - modeling_opt's own rule), and derives
        that by slicing with the cache's length, which for a StaticCache is a device scalar (a host synchronisation: not capturable): the
        decode step hands it `position_ids` instead - the token at `pos` is the (number of ones in the mask so far)-th attended one."""
        if self.lm_family == "gpt2":
            return {"cache_position": self.pos if L0 is None else torch.arange(L0, device=self.pos.device)}
        if L0 is None:
            return {"position_ids": self.am_buf.sum(1, keepdim=True).long() - 1}
        return {}                       # the prompt step: an empty cache, OPT's own cumsum applies

    def _step(self):
        self.am_buf.index_fill_(1, self.pos, 1)
        o = self.model.flamingo(input_ids=self.tok, attention_mask=self.am_buf, use_cache=True, past_key_values=(self.xattn_past, self.cache),
                                text_time=self.tt_step, **self._positions())
        self.pos.add_(1)
        self._append(o.logits[:, -1])

    @torch.no_grad()
    def run(self, ids, ml, am, pixel_values, visual_features):
        L0 = ids.shape[1]
        dev = ids.device
        self.cache.reset()
        out = self.model.flamingo(input_ids=ids, attention_mask=am, media_locations=ml, use_cache=True, past_key_values=(None, self.cache),
                                  pixel_values=pixel_values, visual_features=visual_features, **self._positions(L0))
        fresh = out.past_key_values[0]
        if self.xattn_past is None:
            self.xattn_past = tuple((torch.empty_like(k, memory_format=torch.contiguous_format), torch.empty_like(v, memory_format=torch.contiguous_format))
                                    for k, v in fresh)
        for (kb, vb), (k, v) in zip(self.xattn_past, fresh):
            kb.copy_(k); vb.copy_(v)
        self.tt_step.copy_(F.text_time(ml)[:, -1:])
        self.ids_buf.fill_(self.fill)
        self.ids_buf[:, :L0] = ids
        self.am_buf.zero_()
        self.am_buf[:, :L0] = am
        self.finished.zero_()
        self.n_new.zero_()
        self.pos.fill_(L0)
        self._append(out.logits[:, -1])
        remaining = self.max_length - L0 - 1
        if self.graph_wanted and self.replay is None and not self.capture_failed and remaining > 1:
            self._step()                                                      # eager once: lazy initialisation, allocator pools
            remaining -= 1
            try:
                g = torch.cuda.CUDAGraph()
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    self._step()
                self.graph, self.replay = g, g.replay
            except Exception:                                                 # the steps run eagerly from here on (and in later calls)
                torch.cuda.synchronize()
                self.capture_failed = True
        run_step = self.replay or self._step
        i = 0
        while i < remaining:
            run_step()
            i += 1
            if self.eos is not None and i % 16 == 0 and bool(self.finished.all()):     # (a host sync: only every 16 tokens)
                break
        if self.eos is not None:
            return self.ids_buf[:, :L0 + int(self.n_new)].clone()
        return self.ids_buf.clone()


class FlamingoModel(PreTrainedModel):
    """LM-independent wrapper (reference :359-712)."""

    config_class = FlamingoConfig
    _LANGUAGE_MODEL_VERSIONS = {"gpt2": FlamingoGPT2, "facebook/opt": FlamingoOPT}
    # greedy decoding on the GPU (GPT-2- and, since round 5, OPT-backed models): fixed-shape decode steps (static_decode), replayed from a HIP graph (decode_graph);
    # plain attributes, settable per model - no environment variable
    static_decode = True
    decode_graph = True
    _keys_to_ignore_on_load_missing = [r"flamingo.vision_encoder"]

    def __init__(self, config: FlamingoConfig, model_class: Optional[type] = None):
        super().__init__(config)
        if model_class is None:
            model_class = self._find_flamingo_class(config.lm)
        self.flamingo: FlamingoBaseModel = model_class(config)
        if config.freeze_language_model:
            self.freeze_lm()
        if config.freeze_vision_model:
            self.freeze_vm()
        self.post_init()      # (transformers bookkeeping: the tied-weight table save_pretrained / from_pretrained use; _init_weights is a no-op)

    def _init_weights(self, module):
        pass

    @classmethod
    def is_lm_supported(cls, lm_id: str) -> bool:
        return any(lm_id.startswith(prefix) for prefix in cls._LANGUAGE_MODEL_VERSIONS)

    @classmethod
    def _find_flamingo_class(cls, language_model_id: str):
        for prefix, flamingo_class in cls._LANGUAGE_MODEL_VERSIONS.items():
            if language_model_id.startswith(prefix):
                return flamingo_class
        raise ValueError(f"unsupported language model {language_model_id}")

    def parameters_trainable(self):
        return self.flamingo.parameters_trainable()

    def install_autograd_cuts(self, cuts, segment_layers: int = 4) -> None:
        self.flamingo.install_autograd_cuts(cuts, segment_layers)

    def set_launch_structure(self, **kw):
        return self.flamingo.set_launch_structure(**kw)

    def freeze_vm(self):
        self.flamingo.freeze_vm()

    def freeze_lm(self):
        self.flamingo.freeze_lm()

    def unfreeze_lm(self):
        self.flamingo.unfreeze_lm()

    def state_dict_trainable(self):
        return self.flamingo.state_dict_trainable()

    def forward(self, input_ids=None, attention_mask=None, media_locations=None, pixel_values=None, visual_features=None,
                head_mask=None, inputs_embeds=None, use_cache: bool = False, past_key_values=None, return_dict: bool = True,
                labels=None, loss_reduction: str = "mean", **kwargs) -> CausalLMOutputWithPast:
        return self.flamingo(input_ids=input_ids, attention_mask=attention_mask, media_locations=media_locations,
                             pixel_values=pixel_values, visual_features=visual_features, head_mask=head_mask,
                             inputs_embeds=inputs_embeds, use_cache=use_cache, past_key_values=past_key_values,
                             return_dict=return_dict, labels=labels, loss_reduction=loss_reduction, **kwargs)

    # ---- generation helpers (reference :464-605) ----
    def prepare_inputs_for_generation(self, input_ids, media_locations=None, attention_mask=None, pixel_values=None,
                                      visual_features=None, past=None, past_key_values=None, **kwargs) -> Dict[str, Any]:
        """Replicate visuals / media_locations to the beam-expanded batch; with a cache only the last token is fed."""
        n_inputs = input_ids.shape[0]

        def expand(t):
            if t is None or t.shape[0] == n_inputs:
                return t
            assert n_inputs % t.shape[0] == 0
            return _repeat_leading(t, n_inputs // t.shape[0])

        cache = past_key_values if past_key_values is not None else past
        if cache is not None:
            input_ids = input_ids[:, -1:]
        return dict(input_ids=input_ids, past_key_values=cache, media_locations=expand(media_locations),
                    attention_mask=attention_mask, pixel_values=expand(pixel_values), visual_features=expand(visual_features), **kwargs)

    def _reorder_cache(self, past, beam_idx):
        xattn_past, lm_past = past
        pick = lambda t: t.index_select(0, beam_idx.to(t.device))
        xattn_new = tuple(tuple(pick(t) for t in layer) for layer in xattn_past)
        if hasattr(lm_past, "reorder_cache"):      # transformers >= 4.36 Cache objects
            lm_past.reorder_cache(beam_idx)
            lm_new = lm_past
        else:
            lm_new = tuple(tuple(pick(t) for t in layer) for layer in lm_past)
        return xattn_new, lm_new

    @torch.no_grad()
    def greedy_generate(self, input_ids, media_locations, attention_mask, pixel_values=None, visual_features=None,
                        max_length: int = 150, eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None):
        """Cached greedy decoding (the caption tokens/sec path); see generate()."""
        return self.generate(input_ids, media_locations=media_locations, attention_mask=attention_mask, pixel_values=pixel_values,
                             visual_features=visual_features, max_length=max_length, eos_token_id=eos_token_id, pad_token_id=pad_token_id)

    def _decode_step(self, step_ids, ml, am, past, pixel_values, visual_features):
        out = self.flamingo(input_ids=step_ids, attention_mask=am, media_locations=ml, use_cache=True, past_key_values=past,
                            pixel_values=pixel_values if past is None else None, visual_features=visual_features if past is None else None)
        return out.logits[:, -1].float(), out.past_key_values

    @staticmethod
    def _filter_logits(logits, temperature, top_k, top_p):
        """temperature / top-k / nucleus filtering as in transformers' logits warpers."""
        if temperature != 1.0:
            logits = logits / temperature
        if top_k and top_k > 0:
            kth = logits.topk(min(top_k, logits.shape[-1]), dim=-1).values[..., -1, None]
            logits = logits.masked_fill(logits < kth, float("-inf"))
        if top_p < 1.0:
            srt, idx = logits.sort(dim=-1, descending=True)
            cum = srt.softmax(-1).cumsum(-1)
            drop = cum - srt.softmax(-1) >= top_p               # keep the smallest prefix whose mass reaches top_p
            logits = logits.masked_fill(drop.scatter(-1, idx, drop), float("-inf"))
        return logits

    @torch.no_grad()
    def generate(self, inputs=None, media_locations=None, attention_mask=None, pixel_values=None, visual_features=None,
                 max_length: int = 150, num_beams: int = 1, do_sample: bool = False, temperature: float = 1.0, top_k: int = 0,
                 top_p: float = 1.0, eos_token_id: Optional[int] = None, pad_token_id: Optional[int] = None, bos_token_id: Optional[int] = None,
                 early_stopping: bool = True, length_penalty: float = 1.0, use_cache: bool = True, generator: Optional[torch.Generator] = None,
                 input_ids=None, static_decode: Optional[bool] = None, **unsupported):
        """Text generation with the cached cross-attention path: the first step runs CLIP + resampler and fills the xattn K/V and LM caches,
        every later step feeds one token (reference: HF `generate` through prepare_inputs_for_generation / _reorder_cache, :464-548).
        transformers >= 4.50 no longer gives PreTrainedModel a `generate`, so the decoding strategies the reference's callers use are
        implemented here: greedy, multinomial sampling (do_sample, temperature, top_k, top_p) and beam search (num_beams, length_penalty,
        early_stopping).  Unknown generation arguments raise instead of being ignored."""
        if unsupported:
            raise TypeError(f"generate(): unsupported generation arguments {sorted(unsupported)}")
        if not use_cache:
            raise ValueError("generate() always decodes with the cross-attention / LM caches (use_cache=True)")
        ids = inputs if inputs is not None else input_ids
        assert ids is not None and media_locations is not None, "generate() needs input ids and media_locations"
        am = attention_mask if attention_mask is not None else torch.ones_like(ids)
        ml = media_locations
        pad = pad_token_id if pad_token_id is not None else eos_token_id
        if num_beams > 1:
            if do_sample:
                raise ValueError("beam search with sampling is not implemented")
            return self._beam_search(ids, ml, am, pixel_values, visual_features, max_length, num_beams, eos_token_id, pad, early_stopping, length_penalty)
        if static_decode is None:       # greedy decoding on the GPU (GPT-2- and OPT-backed models): fixed-shape decode steps, replayed from a HIP graph
            static_decode = ids.is_cuda and self.static_decode
        if static_decode and not do_sample and isinstance(self.flamingo, (FlamingoGPT2, FlamingoOPT)) and ids.shape[1] + 1 < max_length:
            return self._static_greedy(ids, ml, am, pixel_values, visual_features, max_length, eos_token_id, pad)
        finished = torch.zeros(ids.shape[0], dtype=torch.bool, device=ids.device)
        past, step_ids = None, ids
        while ids.shape[1] < max_length:
            logits, past = self._decode_step(step_ids, ml, am, past, pixel_values, visual_features)
            if do_sample:
                probs = self._filter_logits(logits, temperature, top_k, top_p).softmax(-1)
                nxt = torch.multinomial(probs, 1, generator=generator)[:, 0]
            else:
                nxt = logits.argmax(-1)
            if eos_token_id is not None:
                nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
                finished = finished | (nxt == eos_token_id)
            ids = torch.cat([ids, nxt[:, None]], dim=1)
            ml = torch.cat([ml, torch.zeros_like(ml[:, :1])], dim=1)
            am = torch.cat([am, torch.ones_like(am[:, :1])], dim=1)
            step_ids = ids[:, -1:]
            if eos_token_id is not None and bool(finished.all()):
                break
        return ids

    def _static_greedy(self, ids, ml, am, pixel_values, visual_features, max_length, eos, pad, graph: Optional[bool] = None):
        """Greedy decoding with FIXED shapes (see _DecodeSession).  Sessions - preallocated caches and buffers plus, on the GPU, the captured
        HIP graph of one decode step - are kept per (batch, max_length, keys per sequence, eos / pad) and reused by later calls, so only the
        first caption batch of a shape pays for the capture.  a session whose model's parameters were re-allocated since (model.to(), ShardedAdamW) is rebuilt; reset_decode_sessions() drops them all."""
        b = ids.shape[0]
        if graph is None:
            graph = ids.is_cuda and not self.training and self.decode_graph
        sessions = _DECODE_SESSIONS.setdefault(self, {})
        n_media = int(ml.sum(-1).max()) if visual_features is None and pixel_values is None else \
            (visual_features.shape[1] if visual_features is not None else (pixel_values.shape[1] if pixel_values.ndim >= 5 else pixel_values.shape[0]))
        key = (b, max_length, n_media, str(ids.device), eos, pad, bool(graph))
        sess = sessions.get(key)
        if sess is not None and sess.stale():
            sessions.pop(key)
            sess = None
        if sess is None:
            if len(sessions) >= 4:                                            # a handful of shapes at most: each holds a KV cache and a graph
                sessions.pop(next(iter(sessions)))
            sess = sessions[key] = _DecodeSession(self, b, max_length, ids.device, ids.dtype, am.dtype, eos, pad, graph)
        return sess.run(ids, ml, am, pixel_values, visual_features)

    @property
    def _decode_sessions(self) -> dict:
        return _DECODE_SESSIONS.get(self, {})

    def reset_decode_sessions(self) -> None:
        _DECODE_SESSIONS.pop(self, None)

    def _beam_search(self, ids, ml, am, pixel_values, visual_features, max_length, nb, eos, pad, early_stopping, length_penalty):
        """Standard beam search over the cached decode path.  The prompt runs once per sequence; its caches are then replicated per beam
        (xattn K/V: repeat_interleave; LM cache: Cache.batch_repeat_interleave) and re-ordered every step (_reorder_cache)."""
        b, L0 = ids.shape
        dev = ids.device
        logits, past = self._decode_step(ids, ml, am, None, pixel_values, visual_features)
        logp = logits.log_softmax(-1)
        V = logp.shape[-1]
        xattn_past, lm_past = past
        xattn_past = tuple(tuple(_repeat_leading(t, nb) for t in kv) for kv in xattn_past)
        if hasattr(lm_past, "batch_repeat_interleave"):
            lm_past.batch_repeat_interleave(nb)
        else:
            lm_past = tuple(tuple(_repeat_leading(t, nb) for t in layer) for layer in lm_past)
        past = (xattn_past, lm_past)
        seqs = _repeat_leading(ids, nb)                                   # (b * nb, L)
        ml, am = _repeat_leading(ml, nb), _repeat_leading(am, nb)
        scores = torch.full((b, nb), float("-inf"), device=dev)
        scores[:, 0] = 0.0                                                # all beams of a sequence start identical: only one may branch
        logp = _repeat_leading(logp, nb)
        done_hyps = [[] for _ in range(b)]                                # (normalised score, token list) per sequence
        is_done = [False] * b
        while True:
            cand = (scores.reshape(-1, 1) + logp).reshape(b, nb * V)
            top_s, top_i = cand.topk(2 * nb, dim=-1)
            cur_len = seqs.shape[1] + 1
            next_scores = torch.full((b, nb), float("-inf"), device=dev)
            next_tok = torch.full((b, nb), pad if pad is not None else 0, dtype=torch.long, device=dev)
            next_src = torch.arange(nb, device=dev).repeat(b, 1)
            top_s_c, top_i_c = top_s.cpu(), top_i.cpu()
            for bi in range(b):
                if is_done[bi]:
                    continue
                k = 0
                for rank in range(2 * nb):
                    s_, i_ = float(top_s_c[bi, rank]), int(top_i_c[bi, rank])
                    beam, tok = i_ // V, i_ % V
                    if eos is not None and tok == eos:
                        if rank < nb:
                            done_hyps[bi].append((s_ / (cur_len ** length_penalty), seqs[bi * nb + beam].tolist() + [tok]))
                        continue
                    next_scores[bi, k], next_tok[bi, k], next_src[bi, k] = s_, tok, beam
                    k += 1
                    if k == nb:
                        break
                if len(done_hyps[bi]) >= nb:
                    done_hyps[bi] = sorted(done_hyps[bi], key=lambda t: -t[0])[:nb]
                    best_open = float(next_scores[bi].max()) / (cur_len ** length_penalty)
                    if early_stopping or done_hyps[bi][-1][0] >= best_open:
                        is_done[bi] = True
            beam_idx = (next_src + torch.arange(b, device=dev)[:, None] * nb).reshape(-1)
            seqs = torch.cat([seqs.index_select(0, beam_idx), next_tok.reshape(-1, 1)], dim=1)
            scores = next_scores
            if all(is_done) or seqs.shape[1] >= max_length:
                break
            past = self._reorder_cache(past, beam_idx)
            ml = torch.cat([ml, torch.zeros_like(ml[:, :1])], dim=1)
            am = torch.cat([am, torch.ones_like(am[:, :1])], dim=1)
            logits, past = self._decode_step(seqs[:, -1:], ml, am, past, None, None)
            logp = logits.log_softmax(-1)
        out = []
        for bi in range(b):
            hyps = list(done_hyps[bi])
            if not is_done[bi]:                                           # length limit: the open beams count as hypotheses too
                hyps += [(float(scores[bi, k]) / (seqs.shape[1] ** length_penalty), seqs[bi * nb + k].tolist()) for k in range(nb)
                         if float(scores[bi, k]) > float("-inf")]
            out.append(max(hyps, key=lambda t: t[0])[1])
        width = max(len(o) for o in out)
        fill = pad if pad is not None else 0
        return torch.tensor([o + [fill] * (width - len(o)) for o in out], dtype=torch.long, device=dev)

    @torch.no_grad()
    def generate_captions(self, processor, pixel_values=None, images=None, prompt: str = "<image>", max_length: int = 150,
                          num_beams: int = 1, device=None, **kwargs):
        """Caption a batch of images; the prompt is replicated for every image (reference :550-605).  `kwargs` are generation
        arguments (do_sample, temperature, top_k, top_p, length_penalty, ...) and go to generate(), which rejects unknown ones."""
        if device is None:
            device = self.device
        if images is not None:
            assert pixel_values is None, "you can only pass either images or visual features to generate_captions()!"
            if not isinstance(images, (list, tuple)):
                images = [images]
            pixel_values = processor(images=images, device=device)["pixel_values"]
        assert pixel_values is not None, "you must pass either images or visual features to generate_captions()!"
        batch_size = pixel_values.size(0)
        input_ids, media_locations, attention_mask = processor.encode_text(prompt, device)
        input_ids = input_ids[:1].expand(batch_size, -1).contiguous()
        media_locations = media_locations[:1].expand(batch_size, -1).contiguous()
        attention_mask = attention_mask[:1].expand(batch_size, -1).contiguous()
        lm_cfg = self.flamingo.lm.config
        if pixel_values.ndim == 4:
            pixel_values = pixel_values[:, None]       # (b c h w) -> one image per sequence
        out_ids = self.generate(inputs=input_ids, media_locations=media_locations, attention_mask=attention_mask, pixel_values=pixel_values,
                                num_beams=num_beams, early_stopping=True, use_cache=True, bos_token_id=lm_cfg.bos_token_id,
                                eos_token_id=lm_cfg.eos_token_id, pad_token_id=lm_cfg.eos_token_id, max_length=max_length, **kwargs)
        captions = processor.tokenizer.batch_decode(out_ids, skip_special_tokens=True)
        return [processor.remove_tags(t) for t in captions]

    @torch.no_grad()
    def score_sequences(self, input_ids, media_locations, attention_mask, pixel_values=None, visual_features=None,
                        k: int = 100000) -> torch.Tensor:
        """EXPERIMENTAL zero-shot scoring (reference :607-712): log-prob of each candidate sequence given the same
        visuals.  The shared prefix is run once with use_cache; its cross-attention K/V are reused for the candidates
        (the LM self-attention prefix is recomputed: HF Cache objects are not sliceable per candidate)."""
        assert visual_features is None or visual_features.ndim == 3
        n_choices = input_ids.size(0)
        n_reuse = get_common_prefix_length(input_ids)
        k = min(k, n_choices)
        out = self.flamingo(input_ids=input_ids[:1, :n_reuse], media_locations=media_locations[:1, :n_reuse],
                            attention_mask=attention_mask[:1, :n_reuse],
                            pixel_values=pixel_values.unsqueeze(0) if pixel_values is not None else None,
                            visual_features=visual_features.unsqueeze(0) if visual_features is not None else None, use_cache=True)
        next_tokens = input_ids[:, n_reuse]
        topk = out.logits[0, -1, :].index_select(0, next_tokens).topk(k).indices
        xattn_past = [tuple(t.expand(k, *t.shape[1:]) for t in kv) for kv in out.past_key_values[0]]
        out2 = self.flamingo(input_ids=input_ids[topk], media_locations=media_locations[topk], attention_mask=attention_mask[topk],
                             past_key_values=(xattn_past, None))
        logp = out2.logits[:, n_reuse - 1:-1].float().log_softmax(-1)
        tgt = input_ids[topk][:, n_reuse:]
        tok = logp.gather(-1, tgt[..., None])[..., 0]     # every position after the shared prefix counts, padded or not (reference :699-703: unmasked sum)
        scores = torch.full([n_choices], torch.finfo(torch.float).min, device=tok.device)
        scores[topk] = tok.sum(1)
        return scores.detach()
