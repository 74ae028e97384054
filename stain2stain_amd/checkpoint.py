"""Checkpoint interchange with the reference.

The reference saves Lightning checkpoints (configs/callbacks/model_checkpoint.yaml:3-17) and its inference
scripts read ``checkpoint['state_dict']`` into the instantiated module (src/infer_simple_flowmatching.py:22,51).
Our modules register the same parameter / buffer names and OIHW fp32 master shapes as the reference's
(SharedEncoder / FlowMatchingDecoder / SegmentationDecoder), so a state_dict moves across unchanged; the
MFMA-packed bf16 copies (Wf / Wd) are derived data, rebuilt on first use after a load because the packing cache
is keyed on the master weight's version counter.  This file is the file-level glue:

  * ``read_checkpoint``      safe load (``weights_only=True`` -- nothing in the file is executed; the
                             reference's own ``weights_only=False`` is what its pickled hyper-parameters need,
                             we do not read those)
  * ``extract_state_dict``   Lightning layout or bare state_dict; strips ``_orig_mod.`` (torch.compile,
                             conditional_flow_matching.py:93-94 ``compile: true``) and an optional prefix
  * ``load_weights``         into any of our modules, strict by default, with a readable report on mismatch
  * ``save_checkpoint``      Lightning-shaped dict: ``state_dict`` + ``optimizer_states`` (torch.optim.Adam layout,
                             see CFMTrainer.optimizer_state_dict) + ``epoch`` / ``global_step``

No reference-trained checkpoint exists offline, so the tests pin the format with a checkpoint
written from the reference's own freshly initialised modules (tests/golden/make_golden.py).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

LIGHTNING_VERSION_TAG = "2.0.0"      # the reference pins lightning>=2.0.0 (requirements.txt)


def read_checkpoint(path: str, map_location="cpu") -> Dict:
    return torch.load(path, map_location=map_location, weights_only=True)


def extract_state_dict(ckpt: Dict, strip_prefix: Optional[str] = None) -> Dict[str, torch.Tensor]:
    sd = ckpt["state_dict"] if "state_dict" in ckpt and isinstance(ckpt["state_dict"], dict) else ckpt
    out = {}
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            raise TypeError(f"state_dict entry {k!r} is {type(v).__name__}, not a tensor")
        k = k.replace("_orig_mod.", "")
        if strip_prefix:
            if not k.startswith(strip_prefix):
                continue
            k = k[len(strip_prefix):]
        out[k] = v
    return out


def load_weights(module: torch.nn.Module, source, strip_prefix: Optional[str] = None, strict: bool = True):
    """``source``: path, Lightning checkpoint dict or state_dict.  Returns torch's (missing, unexpected) record."""
    ckpt = read_checkpoint(source) if isinstance(source, (str, bytes)) or hasattr(source, "__fspath__") else source
    sd = extract_state_dict(ckpt, strip_prefix)
    own = module.state_dict()
    bad = [f"{k}: checkpoint {tuple(sd[k].shape)} vs module {tuple(own[k].shape)}"
           for k in sd if k in own and tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError("stain2stain_amd: checkpoint does not fit this architecture:\n  " + "\n  ".join(bad))
    return module.load_state_dict(sd, strict=strict)


def save_checkpoint(path: str, module: torch.nn.Module, optimizer_state: Optional[Dict] = None, epoch: int = 0,
                    global_step: int = 0, extra: Optional[Dict] = None) -> Dict:
    """Writes what the reference's loaders read (``state_dict``), plus the optimiser state in torch.optim layout."""
    ckpt = {"epoch": int(epoch), "global_step": int(global_step), "pytorch-lightning_version": LIGHTNING_VERSION_TAG,
            "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
            "optimizer_states": [optimizer_state] if optimizer_state is not None else [], "lr_schedulers": []}
    if extra:
        ckpt.update(extra)
    torch.save(ckpt, path)
    return ckpt


def state_dict_keys(module: torch.nn.Module) -> Iterable[str]:
    return list(module.state_dict().keys())
