"""Checkpoint I/O with the reference's exact ``.pth`` layout.

* writer: the dict of ``src/train.py:303-316``
* reader: the resolution rules of ``app/model_utils.py:16-100`` and ``test/evaluate.py:83-166``
  (``model_type`` default 'unet', ``metadata_input_length`` default 4, legacy keys
  ``additional_embeddings`` / ``metadata_only_embeddings`` / ``'noemb'`` study names, and the
  ``model_state_dict`` | ``state_dict`` | bare state-dict variants).
Files are interchangeable with the reference in both directions (same keys, value types, tensor shapes and dtypes).
"""
from __future__ import annotations

import os
from typing import Any, Dict, Optional, Tuple

import torch

from .model import UrbanPredictor


def build_hyperparameters(cfg, model_type: str, temporal_embeddings: bool, metadata_embeddings: bool,
                          input_channels, target_channels) -> Dict[str, Any]:
    """``hyperparams`` dict of src/train.py:156-168, field for field: ``target_channels`` and ``input_channels`` are the
    comma-joined channel NAMES of ``CONFIG.dataset`` (strings), exactly as the reference writes them."""
    return {"learning_rate": cfg.learning_rate, "batch_size": cfg.batch_size, "weight_decay": cfg.weight_decay,
            "temporal_dim": cfg.temporal_dim, "meta_dim": cfg.meta_dim, "lstm_hidden": cfg.lstm_hidden,
            "model_type": model_type, "target_channels": ",".join(target_channels), "input_channels": ",".join(input_channels),
            "temporal_embeddings": temporal_embeddings, "metadata_embeddings": metadata_embeddings}


def save_checkpoint(path: str, model: torch.nn.Module, optimizer: Optional[torch.optim.Optimizer], *, epoch: int, step: int,
                    loss: float, hyperparameters: Dict[str, Any], model_type: str, study_name: str, trial_id: int,
                    metadata_input_length: int) -> Dict[str, Any]:
    """Write the best-validation checkpoint exactly as src/train.py:305-319 does."""
    checkpoint = {
        "epoch": epoch,
        "step": step,
        "model_state_dict": model.state_dict(),
        "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
        "loss": loss,
        "hyperparameters": hyperparameters,
        "model_type": model_type,
        "study_name": study_name,
        "trial_id": trial_id,
        "metadata_input_length": metadata_input_length,
    }
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    torch.save(checkpoint, path)
    return checkpoint


def resolve_embedding_flags(checkpoint: Dict[str, Any], study_name: str = "") -> Tuple[bool, bool]:
    """(temporal_embeddings, metadata_embeddings) as test/evaluate.py:92-113 / app/model_utils.py:41-64 resolve them."""
    hyper = checkpoint.get("hyperparameters", {}) if isinstance(checkpoint, dict) else {}
    if "temporal_embeddings" in hyper:
        return bool(hyper["temporal_embeddings"]), bool(hyper["metadata_embeddings"])
    default_emb = True
    if "noemb" in study_name or "noemb" in checkpoint.get("study_name", ""):
        default_emb = False
    if checkpoint.get("additional_embeddings", default_emb):
        return True, True
    if checkpoint.get("metadata_only_embeddings", False):
        return False, True
    return False, False


def model_kwargs_from_checkpoint(checkpoint: Dict[str, Any], spatial_channels: int = 23, seq_len: int = 10,
                                 out_channels: int = 2, study_name: str = "") -> Dict[str, Any]:
    """Constructor arguments as the reference's readers derive them (app/model_utils.py:66-88;
    test/evaluate.py:152-164 uses CONFIG for spatial_channels / seq_len)."""
    hyper = checkpoint.get("hyperparameters", {})
    t_emb, m_emb = resolve_embedding_flags(checkpoint, study_name)
    return dict(model_type=checkpoint.get("model_type", "unet"), spatial_channels=spatial_channels, seq_len=seq_len,
                temporal_dim=hyper.get("temporal_dim", 64), meta_features=checkpoint.get("metadata_input_length", 4),
                meta_dim=hyper.get("meta_dim", 64), lstm_dim=hyper.get("lstm_hidden", 96), out_channels=out_channels,
                temporal_embeddings=t_emb, metadata_embeddings=m_emb)


def load_model(model_path: str, device: str = "cuda", spatial_channels: int = 23, seq_len: int = 10, out_channels: int = 2,
               clean: bool = False) -> UrbanPredictor:
    """Mirror of app/model_utils.py:16-100 ``load_model``: checkpoint -> constructed, weight-loaded, eval-mode model.

    ``clean=True`` reproduces the reference's side effect of re-saving the file without the optimizer /
    bookkeeping keys (app/model_utils.py:23-36); it is off by default here.
    """
    checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
    if clean and isinstance(checkpoint, dict):
        dirty = False
        for key in ("optimizer_state_dict", "scheduler_state_dict", "optimizer", "loss", "epoch", "step"):
            if key in checkpoint:
                del checkpoint[key]
                dirty = True
        if dirty:
            torch.save(checkpoint, model_path)
    meta = checkpoint if isinstance(checkpoint, dict) and ("model_state_dict" in checkpoint or "state_dict" in checkpoint
                                                           or "hyperparameters" in checkpoint) else {}
    model = UrbanPredictor(**model_kwargs_from_checkpoint(meta, spatial_channels, seq_len, out_channels))
    if isinstance(checkpoint, dict) and "model_state_dict" in checkpoint:
        model.load_state_dict(checkpoint["model_state_dict"])
    elif isinstance(checkpoint, dict) and "state_dict" in checkpoint:
        model.load_state_dict(checkpoint["state_dict"])
    else:
        model.load_state_dict(checkpoint)
    model.to(device)
    model.eval()
    return model


def run_inference(model, input_tensor, meta_tensor, temp_series_tensor, device: str = "cuda"):
    """Mirror of app/model_utils.py:102-109 (argument order input, meta, temp_series; returns numpy)."""
    with torch.no_grad():
        out = model(input_tensor.to(device), temp_series_tensor.to(device), meta_tensor.to(device))
        return out.cpu().numpy()
