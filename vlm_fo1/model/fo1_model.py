"""Engine-backed stand-in for the reference's `OmChatQwen25VLForCausalLM`
(vlm_fo1/model/language_model/omchat_qwen2_5_vl.py:28-572): exposes what the reference's callers touch —
`generate(**prepare_inputs(...))`, `config` attribute access, `get_vision_tower()/get_vision_tower_aux()`,
`eval()/to()` — and owns generation itself (greedy loop + KV cache in the engine) instead of leaning on HF
`GenerationMixin` internals, which drifted under transformers 5."""
from __future__ import annotations

import types
from typing import Dict, List, Optional

import torch

from vlm_fo1_amd.davit import DAVIT_LARGE
from vlm_fo1_amd.llm import LLMConfig
from vlm_fo1_amd.model import FO1Config, FO1Engine
from vlm_fo1_amd.vit import ViTConfig


class FO1HFConfig:
    """`config.json` with attribute access (`getattr(config, 'mm_*', default)` as the reference reads it:
    omchat_arch.py:11-31, builder.py:65-70, mm_utils.py:593)."""

    def __init__(self, d: dict, generation: Optional[dict] = None):
        self._d = dict(d)
        self._gen = dict(generation or {})
        for k, v in d.items():
            if k == "vision_config" and isinstance(v, dict):
                v = types.SimpleNamespace(**v)
            setattr(self, k, v)

    def to_dict(self):
        return dict(self._d)

    # ---- engine configuration derived from the checkpoint's config (asserted, never guessed) ----
    def engine_config(self) -> FO1Config:
        d = self._d
        txt = d.get("text_config", d)
        vis = d["vision_config"]
        rs = txt.get("rope_scaling") or d.get("rope_scaling") or {}
        heads = txt["num_attention_heads"]
        llm = LLMConfig(hidden_size=txt["hidden_size"], num_layers=txt["num_hidden_layers"], num_heads=heads,
                        num_kv_heads=txt["num_key_value_heads"], head_dim=txt["hidden_size"] // heads,
                        intermediate_size=txt["intermediate_size"], vocab_size=txt["vocab_size"],
                        rms_norm_eps=txt.get("rms_norm_eps", 1e-6), rope_theta=txt.get("rope_theta", 1e6),
                        mrope_section=tuple(rs.get("mrope_section", (16, 24, 24))),
                        max_seq=min(int(d.get("tokenizer_model_max_length") or 8192), 32768))
        vit = ViTConfig(depth=vis["depth"], hidden_size=vis["hidden_size"], num_heads=vis["num_heads"],
                        intermediate_size=vis["intermediate_size"], out_hidden_size=vis["out_hidden_size"],
                        patch_size=vis.get("patch_size", 14), spatial_merge_size=vis.get("spatial_merge_size", 2),
                        temporal_patch_size=vis.get("temporal_patch_size", 2), in_channels=vis.get("in_channels", vis.get("in_chans", 3)),
                        window_size=vis.get("window_size", 112), fullatt_block_indexes=tuple(vis.get("fullatt_block_indexes", (7, 15, 23, 31))))
        unsupported = []
        use_vt = bool(d.get("mm_use_vision_tower_region_feature", False))       # reference default: omchat_arch.py:23
        if not use_vt:
            # The reference itself cannot run this value: HFREModule.__call__ never binds `out_box_feat` without the vt branch and
            # raises UnboundLocalError (hybrid_finegrained_region_encoder.py:456/469; pinned by tests/test_oracle_hfre.py).  The
            # engine's aux-only route is the evident reading (aux block + aux-box embedding) — loadable, labelled as an extension.
            import warnings
            warnings.warn("mm_use_vision_tower_region_feature=False: the reference's HFRE raises UnboundLocalError for this configuration; "
                          "the engine runs its aux-only extension (aux region block + box embedding from the aux boxes)", stacklevel=2)
            if d.get("mm_use_vt_region_feature_only", False):
                unsupported.append("mm_use_vt_region_feature_only without mm_use_vision_tower_region_feature")
        if d.get("mm_region_feature_combination", "concat") not in ("concat", "concat_aux_pos"):
            unsupported.append(f"mm_region_feature_combination={d.get('mm_region_feature_combination')!r}")
        if d.get("mm_pos_embedding_strategy", "bbox_based") not in ("bbox_based", "feature_map_based", "hybrid"):
            unsupported.append(f"mm_pos_embedding_strategy={d.get('mm_pos_embedding_strategy')!r}")
        aux = str(d.get("mm_vision_tower_aux", "davit-large"))
        if "davit-large" not in aux:
            unsupported.append(f"mm_vision_tower_aux={aux!r}")
        if unsupported:
            raise NotImplementedError("checkpoint configuration not built for the MI355X engine: " + ", ".join(unsupported))
        return FO1Config(vit=vit, llm=llm, mm_projector_type=d.get("mm_projector_type", "linear"),
                         mm_projector_aux_type=d.get("mm_projector_aux_type", "linear"),
                         mm_use_simpleFPN_for_vt=bool(d.get("mm_use_simpleFPN_for_vt", False)),
                         mm_region_hidden_size=int(d["mm_region_hidden_size"]), mm_roi_output_size=int(d.get("mm_roi_output_size", 7)),
                         mm_apply_position_embedding=bool(d.get("mm_apply_position_embedding", True)),
                         mm_pos_embedding_strategy=d.get("mm_pos_embedding_strategy", "bbox_based"),
                         mm_apply_region_layer_norm=bool(d.get("mm_apply_region_layer_norm", False)),
                         mm_region_feature_combination=d.get("mm_region_feature_combination", "concat"),
                         mm_use_vt_region_feature_only=bool(d.get("mm_use_vt_region_feature_only", False)),
                         mm_use_vision_tower_region_feature=use_vt)

    def eos_ids(self) -> List[int]:
        e = self._gen.get("eos_token_id", self._d.get("eos_token_id"))
        if e is None:
            return []
        return list(e) if isinstance(e, (list, tuple)) else [int(e)]


class _TowerHandle:
    """What callers get from `get_vision_tower()` / `get_vision_tower_aux()`: loaded flag, config, processor."""

    def __init__(self, config, image_processor=None):
        self.is_loaded = True
        self.config = config
        self.image_processor = image_processor


class FO1ForCausalLM:
    def __init__(self, config: FO1HFConfig, weights: Dict[str, Dict[str, torch.Tensor]], device="cuda"):
        self.config = config
        self.device = torch.device(device)
        self.dtype = torch.bfloat16
        self.engine = FO1Engine(config.engine_config(), weights, device)
        self._vt = _TowerHandle(getattr(config, "vision_config", None))
        self._vt_aux = _TowerHandle(types.SimpleNamespace(**DAVIT_LARGE))
        self.use_graph = True   # replay a captured hipGraph per input-shape signature

    @classmethod
    def from_engine(cls, config: FO1HFConfig, engine: FO1Engine) -> "FO1ForCausalLM":
        """The drop-in model object around an engine that already exists (bench.py's `driver_level` block measures the eval drivers'
        own loop on the engine whose passes it has just timed, instead of loading a second copy of the weights)."""
        m = cls.__new__(cls)
        m.config, m.device, m.dtype, m.engine = config, torch.device(engine.dev), torch.bfloat16, engine
        m._vt = _TowerHandle(getattr(config, "vision_config", None))
        m._vt_aux = _TowerHandle(types.SimpleNamespace(**DAVIT_LARGE))
        m.use_graph = True
        return m

    def replica(self) -> "FO1ForCausalLM":
        """Same weights, private per-request state (KV cache, graphs, scratch): one per worker thread / HIP stream, so several
        requests can be in flight on one GPU (vlm_fo1_amd.sharded_eval.run_sharded with a list of workers)."""
        import copy
        r = copy.copy(self)
        r.engine = self.engine.replica()
        r.__dict__.pop("_worker_replicas", None)       # (sharded_eval.request_workers keeps the model's replicas on the model)
        return r

    # ---- nn.Module-ish surface the reference drivers call ----
    def eval(self):
        return self

    def to(self, *args, **kwargs):
        return self

    def get_model(self):
        return self

    def get_vision_tower(self):
        return self._vt

    def get_vision_tower_aux(self):
        return self._vt_aux

    # ---- generation ----
    def _request(self, inputs, images, images_aux, image_grid_thws, bbox_list):
        """One prepare_inputs(...) kwargs set -> the engine's request dict."""
        if inputs is None or inputs.dim() != 2 or inputs.shape[0] != 1:
            raise ValueError("generate: `inputs` must be a [1, L] id tensor (the reference drivers are batch-1)")
        if not images or image_grid_thws is None:
            raise ValueError("generate: the engine path needs one image (images / image_grid_thws)")
        dev = self.device
        grid = image_grid_thws[0].reshape(-1, 3)[0].tolist()
        if grid[0] != 1:
            raise NotImplementedError("video grids (t > 1) are outside the hot path")
        if not images_aux:
            raise ValueError("generate: images_aux is required (mm_use_region_index_token checkpoints)")
        boxes = None
        if bbox_list is not None and len(bbox_list) > 0 and bbox_list[0] is not None:
            boxes = bbox_list[0].to(device=dev, dtype=torch.float32)
        ids = getattr(inputs, "_fo1_ids", None)      # prepare_inputs keeps the host list next to the device tensor
        if ids is None or len(ids) != inputs.shape[1]:
            ids = inputs[0].tolist()
        self._check_model_max_length(ids, (grid[1] // 2) * (grid[2] // 2))
        return dict(ids=ids, pix=images[0].to(device=dev, dtype=torch.bfloat16), grid=(grid[1], grid[2]),
                    aux=images_aux[0].to(device=dev, dtype=torch.bfloat16), boxes=boxes)

    def _check_model_max_length(self, ids, n_image_tokens: int) -> None:
        """`tokenizer_model_max_length` semantics of the reference splice (omchat_qwen2_5_vl.py:374-378): the spliced embeddings and
        labels are cut to that length but `new_input_ids` is NOT, so with the default right padding the very next statement
        (`new_input_ids_padded[i, :cur_len] = cur_new_input_ids`, :411) fails with torch's size-mismatch RuntimeError — an over-long
        prompt never reaches the LLM.  Same here: same exception type, same sizes in the message.  (Left padding would run the
        reference on all-BOS ids with text-only positions; that variant is not built.)"""
        limit = getattr(self.config, "tokenizer_model_max_length", None)
        if limit is None:
            return
        from vlm_fo1_amd.llm import IMAGE_TOKEN_INDEX
        spliced = len(ids) + sum(n_image_tokens - 1 for t in ids if t == IMAGE_TOKEN_INDEX)      # a <regionfeat> sentinel becomes one row
        if spliced <= int(limit):
            return
        if getattr(self.config, "tokenizer_padding_side", "right") == "left":
            raise NotImplementedError("prompt longer than tokenizer_model_max_length with left padding is not built")
        raise RuntimeError(f"The expanded size of the tensor ({int(limit)}) must match the existing size ({spliced}) at non-singleton dimension 0.  "
                           f"Target sizes: [{int(limit)}].  Tensor sizes: [{spliced}]  (spliced prompt exceeds tokenizer_model_max_length; the "
                           "reference fails at omchat_qwen2_5_vl.py:411 in the same way)")

    def _device_stop_ids(self, stopping_criteria) -> Optional[List[int]]:
        """EOS ids + the ids of single-token stop keywords (mm_utils.KeywordsStoppingCriteria with `<|im_end|>`): the stop rule the
        batched decoder evaluates on the device.  None when a criterion cannot be expressed as 'last token in a set'."""
        ids = list(self.config.eos_ids())
        for c in (stopping_criteria or []):
            kws = getattr(c, "keyword_ids", None)
            if kws is None or any(int(k.numel()) != 1 for k in kws):
                return None
            ids += [int(k.reshape(-1)[0]) for k in kws]
        ids = sorted(set(ids))
        from vlm_fo1_amd.llm import BatchDecoder
        return ids if len(ids) <= BatchDecoder.MAX_STOP else None     # more ids than the device rule holds: host loop, nothing dropped

    @staticmethod
    def _fits_device_loop(max_new_tokens) -> bool:
        from vlm_fo1_amd.llm import BatchDecoder
        return int(max_new_tokens) <= BatchDecoder.IDS_CAP

    def _batch_plan(self, requests_kwargs: List[dict]):
        """Validation shared by generate_many / generate_many_async -> (engine requests, max_new_tokens, device stop ids), or None when
        the batch must take the one-by-one host loop (a stop criterion the device rule cannot express, a budget beyond its id buffer)."""
        k0 = requests_kwargs[0]
        if k0.get("do_sample") or (k0.get("temperature") not in (0, 0.0, None)):
            raise NotImplementedError("sampling is not built; every reference caller decodes greedily (temperature=0)")
        stop = self._device_stop_ids(k0.get("stopping_criteria"))
        if stop is None or not self._fits_device_loop(k0.get("max_new_tokens", 512)):
            return None
        for kw in requests_kwargs[1:]:       # one budget and one stop rule per packed batch: refuse a mixed batch rather than apply the first's
            if int(kw.get("max_new_tokens", 512)) != int(k0.get("max_new_tokens", 512)) or \
                    self._device_stop_ids(kw.get("stopping_criteria")) != stop or kw.get("do_sample") or \
                    (kw.get("temperature") not in (0, 0.0, None)):
                raise ValueError("generate_many: every request of a batch must share max_new_tokens, stopping criteria and greedy decoding")
        reqs = [self._request(kw.get("inputs"), kw.get("images"), kw.get("images_aux"), kw.get("image_grid_thws"), kw.get("bbox_list"))
                for kw in requests_kwargs]
        return reqs, int(k0.get("max_new_tokens", 512)), stop

    @staticmethod
    def _assemble(requests_kwargs, reqs, new) -> List[torch.LongTensor]:
        out = []
        for kw, req, ids in zip(requests_kwargs, reqs, new):
            inp = kw["inputs"]
            # [1, L_in + new] on the inputs' device, like HF generate — assembled on the host from the id lists both sides already hold
            # (one upload per request instead of an upload, a device concat and a read back)
            out.append(torch.tensor([list(req["ids"]) + list(ids)], dtype=inp.dtype).to(inp.device))
        return out

    @torch.no_grad()
    def generate_many(self, requests_kwargs: List[dict]) -> List[torch.LongTensor]:
        """Greedy generation for several prepare_inputs(...) kwargs sets at once: the images go through ONE packed prefill pass and
        the sequences decode together (weights streamed once per step, stop rule on the device).  Each result is [1, L_in + new]
        exactly as generate() returns it.  max_new_tokens / stopping criteria are taken from the first request (the eval drivers use
        the same for every item)."""
        if not requests_kwargs:
            return []
        plan = self._batch_plan(requests_kwargs)
        if plan is None:
            return [self.generate(**kw) for kw in requests_kwargs]
        reqs, max_new, stop = plan
        new = self.engine.generate_batch(reqs, max_new_tokens=max_new, stop_ids=stop, use_graph=self.use_graph)
        return self._assemble(requests_kwargs, reqs, new)

    @torch.no_grad()
    def generate_many_async(self, requests_kwargs: List[dict]):
        """generate_many in two halves for callers that keep several groups in flight (sharded_eval.run_sharded): the packed prefill
        runs now and its sequences join the engine's decode pool (FO1Engine.enable_decode_pool); the returned object's `result()` blocks
        until they have stopped and returns what generate_many returns.  Without a pool (or for a batch the device rule cannot take)
        the work is simply done now."""
        class _Ready:
            def __init__(self, value):
                self._value = value

            def result(self):
                return self._value

        if not requests_kwargs:
            return _Ready([])
        eng = self.engine
        plan = self._batch_plan(requests_kwargs) if getattr(eng, "_pool_svc", None) is not None else None
        if plan is None:
            return _Ready(self.generate_many(requests_kwargs))
        reqs, max_new, stop = plan
        handles = [eng.submit_batch(grp, max_new, stop, self.use_graph) for grp in eng.split_passes(reqs)]      # <= 32 requests and <= 64k ViT rows per pass
        model = self

        class _Pending:
            def result(self):
                new = [ids for h in handles for ids in h.result()]
                return model._assemble(requests_kwargs, reqs, new)

        return _Pending()

    @torch.no_grad()
    def generate(self, inputs=None, images=None, images_aux=None, image_grid_thws=None, bbox_list=None, do_sample=False,
                 temperature=0.0, max_new_tokens=512, streamer=None, top_p=1.0, use_cache=True, stopping_criteria=None,
                 pad_token_id=None, **unused) -> torch.LongTensor:
        """Greedy decoding of one prompt.  Returns [1, L_in + new] like HF generate (the reference slices
        `output_ids[0, inputs.shape[1]:]`, inference.py:47-48).  Without a streamer and with id-set stop criteria the whole loop
        runs on the device (BatchDecoder, no per-token host read); otherwise tokens are handed to the host one by one."""
        if do_sample or (temperature not in (0, 0.0, None)):
            raise NotImplementedError("sampling is not built; every reference caller decodes greedily (temperature=0)")
        req = self._request(inputs, images, images_aux, image_grid_thws, bbox_list)
        dev = self.device
        stop = self._device_stop_ids(stopping_criteria) if streamer is None else None
        if stop is not None and self._fits_device_loop(max_new_tokens):
            ids = self.engine.generate_batch([req], max_new_tokens=int(max_new_tokens), stop_ids=stop, use_graph=self.use_graph)[0]
            return torch.cat([inputs.to(dev), torch.tensor([ids], dtype=inputs.dtype, device=dev)], dim=1).to(inputs.device)
        eng = self.engine
        out = eng.prefill(req["ids"], req["pix"], req["grid"], req["aux"], req["boxes"], use_graph=self.use_graph)
        eng.llm.reserve(eng.llm.kv_len + int(max_new_tokens))
        tok = out["next_token"]
        eos = set(self.config.eos_ids())
        all_ids = inputs.to(dev)
        if streamer is not None:
            streamer.put(inputs.cpu())
        first = True
        if self.use_graph:
            eng.llm.sync_decode_state()
        n_max = int(max_new_tokens)
        for i in range(n_max):
            t = tok.to(torch.long).reshape(1, 1)
            all_ids = torch.cat([all_ids, t.to(all_ids.dtype)], dim=1)
            if streamer is not None:
                streamer.put(t.cpu())
            tid = int(t.item())
            stop_now = tid in eos
            if not stop_now and stopping_criteria:
                stop_now = any(bool(c(all_ids, None)) for c in stopping_criteria)
            if stop_now or i + 1 == n_max:          # no decode step after the last token
                break
            if self.use_graph:
                _, tok = eng.llm.decode_step_graph(tok if first else None)
                first = False
            else:
                _, _, tok = eng.llm.decode_step(tok)
        if streamer is not None:
            streamer.end()
        return all_ids.to(inputs.device)
