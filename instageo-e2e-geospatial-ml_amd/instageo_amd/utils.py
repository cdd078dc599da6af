"""Checkpoint plumbing of the Prithvi encoder (reference: ``instageo/model/utils.py:60-315``), host-side only.

Same function names and semantics as the reference so that its call sites (``model.py:221-251``) and tests read unchanged:
locating the state dict inside a checkpoint file, the patch-embedding projection key and its prefix, band selection for the
patch-embedding weight, and the MAE -> ViT key filter.  ``model`` arguments are duck-typed: anything with ``state_dict()`` and the
attributes ``pos_embed`` / ``temporal_encoding`` / ``location_encoding`` (see :func:`encoder_view`).
"""
from __future__ import annotations

import logging
from typing import Any, Dict, Iterable, Optional, Sequence, Tuple

import torch

log = logging.getLogger(__name__)
_PROJ_SUFFIXES = ("patch_embed.proj.weight", "patch_embed.projection.weight")


def patch_embed_weights_are_compatible(model_patch_embed: torch.Tensor, checkpoint_patch_embed: torch.Tensor) -> bool:
    """Equal rank and equal extents everywhere except the channel (band) axis 1 (utils.py:60-82)."""
    a, b = tuple(model_patch_embed.shape), tuple(checkpoint_patch_embed.shape)
    return len(a) == len(b) and a[:1] + a[2:] == b[:1] + b[2:]


def get_state_dict(state_dict: Dict[str, Any]) -> Dict[str, Any]:
    """The value under the first key that ends in ``state_dict`` (Lightning / MMSeg checkpoints), else the dict itself
    (utils.py:85-115)."""
    for k in state_dict.keys():
        if k.endswith("state_dict"):
            return state_dict[k]
    return state_dict


def get_common_prefix(keys: Iterable[str]) -> str:
    """Dot-joined name components shared by all keys but the last one, with a trailing dot (utils.py:118-145; the reference
    intersects the component SETS of ``keys[:-1]``, so for more than one shared component the order is that of a set)."""
    ks = list(keys)[:-1]
    shared = set.intersection(*[set(k.split(".")) for k in ks])
    return (".".join(shared) if len(shared) > 1 else shared.pop()) + "."


def get_proj_key(state_dict: Dict[str, Any], return_prefix: bool = False) -> Tuple[Optional[str], Optional[str]]:
    """First key naming the patch-embedding projection weight and, on request, whatever precedes that suffix (utils.py:148-178)."""
    proj_key = next((k for k in state_dict.keys() if k.endswith(_PROJ_SUFFIXES)), None)
    prefix = None
    if return_prefix and proj_key:
        for suf in _PROJ_SUFFIXES:
            if proj_key.endswith(suf):
                prefix = proj_key.replace(suf, "")
                break
    return proj_key, prefix


def remove_prefixes(state_dict: Dict[str, Any], prefix: str) -> Dict[str, Any]:
    """A new dict whose keys have every occurrence of ``prefix`` removed (utils.py:181-196)."""
    return {k.replace(prefix, ""): v for k, v in state_dict.items()}


def select_patch_embed_weights(state_dict: Dict[str, Any], model: Any, pretrained_bands: Sequence, model_bands: Sequence,
                               proj_key: Optional[str] = None) -> Dict[str, Any]:
    """Patch-embedding weight for the bands the model uses (utils.py:199-268): start from a Xavier-initialised weight of the
    model's shape and copy, band by band, the checkpoint's slice for every model band that was pretrained; bands the checkpoint
    lacks keep the random initialisation, incompatible patch shapes keep the model's weight altogether."""
    if not (isinstance(pretrained_bands, type(model_bands)) or isinstance(pretrained_bands, int) or isinstance(model_bands, int)):
        return state_dict
    state_dict = get_state_dict(state_dict)
    prefix = None
    if proj_key is None:
        proj_key, prefix = get_proj_key(state_dict, return_prefix=True)
    if proj_key is None or proj_key not in state_dict:
        raise Exception("Could not find key for patch embed weight in state_dict.")
    ckpt_w = state_dict[proj_key]
    own = model.state_dict()
    own_key, _ = get_proj_key(own)
    new_w = own[own_key or proj_key].clone()
    if patch_embed_weights_are_compatible(new_w, ckpt_w):
        torch.nn.init.xavier_uniform_(new_w.view(new_w.shape[0], -1))
        for index, band in enumerate(model_bands):
            if band in pretrained_bands:
                new_w[:, index] = ckpt_w[:, list(pretrained_bands).index(band)]
    else:
        log.warning("Incompatible shapes between patch embedding of model %s and of checkpoint %s", tuple(new_w.shape), tuple(ckpt_w.shape))
    state_dict[proj_key] = new_w
    if prefix:
        state_dict = remove_prefixes(state_dict, prefix)
    return state_dict


def checkpoint_filter_fn_vit(state_dict: Dict[str, Any], model: Any, pretrained_bands: Sequence, model_bands: Sequence) -> Dict[str, Any]:
    """Prithvi MAE checkpoint -> Prithvi ViT encoder keys (utils.py:271-315): ``_timm_module.`` dropped from old checkpoints, the
    model's own (fixed, frame-count dependent) ``pos_embed`` substituted, decoder / mask-token keys and unused temporal / location
    embeddings removed, the ``encoder.`` prefix stripped, then :func:`select_patch_embed_weights`."""
    clean: Dict[str, Any] = {}
    for k, v in state_dict.items():
        k = k.replace("_timm_module.", "")
        if "pos_embed" in k:
            v = model.pos_embed
        if "decoder" in k or "_dec" in k or k == "mask_token":
            continue
        if not model.temporal_encoding and "temporal_embed" in k:
            continue
        if not model.location_encoding and "location_embed" in k:
            continue
        clean[k.replace("encoder.", "") if k.startswith("encoder.") else k] = v
    return select_patch_embed_weights(clean, model, pretrained_bands, model_bands)


class encoder_view:
    """What :func:`checkpoint_filter_fn_vit` needs of a ``PrithviViT``, over the encoder part of a :class:`PrithviSeg`."""

    def __init__(self, seg_model: Any) -> None:
        self._m = seg_model
        variant = getattr(seg_model.cfg, "variant", "")
        self.temporal_encoding = self.location_encoding = str(variant).endswith("_tl")

    def state_dict(self) -> Dict[str, torch.Tensor]:
        pre = "prithvi_encoder."
        return {k[len(pre):]: v for k, v in self._m.state_dict().items() if k.startswith(pre)}

    @property
    def pos_embed(self) -> torch.Tensor:
        return self.state_dict()["pos_embed"]
