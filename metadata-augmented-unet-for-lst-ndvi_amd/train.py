"""Training driver with the reference's CLI surface (src/train.py:62-73) around the HIP hot path.

    python -m mau_amd.train --device gpu --model-type unet --no-temporal-embeddings --epochs 1 --steps-per-epoch 20

The reference wraps the step in Optuna trials, wandb logging and a dataset that is not shipped
(SURVEY D9); none of that is on the hot path.  This driver keeps: the flags, the study-name suffix
rule (:79-87), seeding (:104), model construction (:194-207), optimizer / loss selection
(:209-225), the inner step (:243-256), ``validate()`` (:20-60: eval-mode pass over a held-out split,
sample-weighted mean of the criterion's total), best-VALIDATION checkpointing in the exact ``.pth``
layout (:303-319) -- on synthetic batches with the loader's tuple layout (src/dataset.py:87-108).
Single process (the train step is one hipGraph replay, ``train_graph.GraphedTrainStep``), or data
parallel under ``torch.distributed.run`` (RCCL; eager step with overlapped collectives).
"""
from __future__ import annotations

import contextlib
import os
from typing import Optional

import torch
import typer

from . import compute_loss_l1_grad_ssim, compute_loss_mse, compute_loss_mse_gradient
from .checkpoint import build_hyperparameters, save_checkpoint
from .config import CONFIG
from .dist import GradSync, init_process_group_from_env
from .model import UrbanPredictor
from .optim import AdamW
from .train_graph import GraphedTrainStep

app = typer.Typer(add_completion=False)


def synthetic_batch(batch_size: int, device, gen: torch.Generator):
    """(inputs, metadatas, temp_series_padded, temp_series_lengths, t1_dates, t2_dates, targets), src/dataset.py:87-108."""
    ds = CONFIG.dataset
    e = ds.image_shape_edge
    mk = lambda *s: torch.randn(*s, generator=gen).to(device)
    n_meta = ds.nb_metadata_features
    return (mk(batch_size, ds.nb_input_channels, e, e), mk(batch_size, n_meta - 4 if n_meta >= 8 else n_meta),
            mk(batch_size, 24), torch.full((batch_size,), 24), mk(batch_size, 2), mk(batch_size, 2),
            mk(batch_size, len(ds.target_channels), e, e))


def validate(model: torch.nn.Module, loader, criterion, watchdog=None):
    """Loss on the validation set (src/train.py:20-60): eval mode, no_grad, every batch's ``total`` weighted by its sample
    count; a batch whose criterion raises ValueError is skipped; the model is put back into training mode.
    Returns (mean loss, {}) -- ``float('inf')`` when no batch counted.  (The reference additionally averages
    ``compute_all_loss``'s terms for wandb; logging is outside the hot path.)"""
    n_meta = CONFIG.dataset.nb_metadata_features
    model.eval()
    total, num = 0.0, 0
    with torch.no_grad():
        for inputs, metadata, temp_series, _lengths, t1, t2, targets in loader:
            metadata_full = torch.cat([metadata, t1, t2], dim=1) if n_meta >= 8 else metadata
            outputs = model(inputs, temp_series, metadata_full)
            try:
                batch_loss = criterion(outputs, targets)["total"]
                if batch_loss is not None:
                    total += batch_loss.item() * len(inputs)
                    num += len(inputs)
                    if watchdog is not None:
                        watchdog.kick()                                                     # (a read-back: the GPU was there)
            except ValueError as e:
                typer.echo(f"Skipping batch in validation due to error: {e}")
                continue
    model.train()
    if num == 0:
        return float("inf"), {}
    return total / num, {}


@app.command()
def main(device: str = "", wandblog: bool = False, n_trials: int = 1, force_study_name: bool = False,
         temporal_embeddings: bool = True, metadata_embeddings: bool = True, study_name: str = "urban-predictor",
         model_type: str = "unet++", jobid: str = "", epochs: Optional[int] = None, steps_per_epoch: int = 20,
         precision: str = "bf16", val_batches: int = 2, graph: bool = True):
    """The reference's CLI (src/train.py:62-73) + --epochs / --steps-per-epoch / --precision / --val-batches / --no-graph."""
    return run(device, wandblog, n_trials, force_study_name, temporal_embeddings, metadata_embeddings, study_name, model_type,
               jobid, epochs, steps_per_epoch, precision, val_batches, graph)["best"]


def run(device: str = "", wandblog: bool = False, n_trials: int = 1, force_study_name: bool = False,
        temporal_embeddings: bool = True, metadata_embeddings: bool = True, study_name: str = "urban-predictor",
        model_type: str = "unet++", jobid: str = "", epochs: Optional[int] = None, steps_per_epoch: int = 20,
        precision: str = "bf16", val_batches: int = 2, graph: bool = True, force_dist: bool = False):
    """Body of the CLI as a function; returns {'best' (validation loss), 'model', 'optimizer', 'checkpoint_path', 'history'}.
    ``force_dist``: take the data-parallel path (SyncBN + GradSync over the default process group) even with one rank -- the
    one-GPU rehearsal of the RCCL code path (tests/test_gpu_dist_rehearsal.py)."""
    assert model_type in ["unet", "unet++"], "model_type must be 'unet' or 'unet++'"          # src/train.py:78
    if not force_study_name:                                                                  # src/train.py:79-87
        study_name += "-emb" if temporal_embeddings and metadata_embeddings else "-tempemb" if temporal_embeddings \
            else "-metaemb" if metadata_embeddings else "-noemb"
    if device.lower() == "cpu":
        raise typer.BadParameter("this is the MI355X-native path: --device gpu (there is no CPU fallback)")
    rank, local, world = init_process_group_from_env()
    CONFIG.device = f"cuda:{local}"
    torch.cuda.set_device(local)
    torch.manual_seed(CONFIG.seed)                                                            # src/train.py:104
    cfg = CONFIG.training
    n_meta = CONFIG.dataset.nb_metadata_features
    model = UrbanPredictor(model_type=model_type, spatial_channels=CONFIG.dataset.nb_input_channels,
                           seq_len=CONFIG.dataset.temporal_length, temporal_dim=cfg.temporal_dim, meta_features=n_meta,
                           meta_dim=cfg.meta_dim, lstm_dim=cfg.lstm_hidden, out_channels=len(CONFIG.dataset.target_channels),
                           deep_supervision=False, temporal_embeddings=temporal_embeddings,
                           metadata_embeddings=metadata_embeddings).to(CONFIG.device)              # src/train.py:194-206
    model.set_precision(precision).train()
    if cfg.optimizer == "SGD":                                                                # src/train.py:209-216
        optimizer = torch.optim.SGD(model.parameters(), lr=cfg.learning_rate, momentum=cfg.momentum)
    elif cfg.optimizer == "Adam":
        optimizer = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    elif cfg.optimizer == "AdamW":      # torch.optim.AdamW's rule; the convolution weights' update + re-pack is one kernel (optim.py)
        optimizer = AdamW(model.parameters(), lr=cfg.learning_rate, weight_decay=cfg.weight_decay)
    else:
        raise NotImplementedError(f"Optimizer {cfg.optimizer} not implemented.")
    if cfg.loss == "mse":                                                                     # src/train.py:218-225
        criterion = compute_loss_mse
    elif cfg.loss == "mse-gradient":
        criterion = compute_loss_mse_gradient
    elif cfg.loss == "l1-gradient-ssim":
        criterion = compute_loss_l1_grad_ssim
    else:
        raise NotImplementedError(f"Loss {cfg.loss} not implemented.")
    sync = watchdog = None
    import torch.distributed as dist
    if world > 1 or (force_dist and dist.is_available() and dist.is_initialized()):
        # Collectives: RCCL called directly on our streams (dist.RcclComm) is the measured path of bench.py, where a supervisor
        # with a fallback stands behind it.  Here nothing does, and two communicators driven from two streams have not yet
        # run between two GPUs: between REAL ranks the training driver goes through ProcessGroupNCCL (its own watchdog, its own
        # timeout) unless MAU_RCCL_DIRECT=1 asks for the direct path -- which then runs under a watchdog of ours.
        direct = (os.environ["MAU_RCCL_DIRECT"] != "0") if "MAU_RCCL_DIRECT" in os.environ else (world == 1)
        model.set_sync_bn(dist.group.WORLD, direct=direct)
        sync = GradSync(model, direct=direct)
        if sync.comm is not None and world > 1:
            from .dist import CollectiveWatchdog
            watchdog = CollectiveWatchdog()
    hyper = build_hyperparameters(cfg, model_type, temporal_embeddings, metadata_embeddings,
                                  CONFIG.dataset.input_channels, CONFIG.dataset.target_channels)
    gen = torch.Generator().manual_seed(CONFIG.seed + rank)
    # held-out synthetic validation split (the reference's val_loader, src/train.py:183-192): drawn once, from its own generator
    vgen = torch.Generator().manual_seed(CONFIG.seed + 7919 + rank)
    val_loader = [synthetic_batch(cfg.batch_size, CONFIG.device, vgen) for _ in range(max(0, val_batches))]
    clip = 5.0 if cfg.gradient_clipping > 0 else 0.0                                            # src/train.py:253-254
    # one GPU: the step (forward, criterion, backward, clip, optimizer) is captured once and replayed (static shapes);
    # data parallel: eager, the RCCL collectives are launched from autograd hooks while backward still runs
    # (MAU_DP_GRAPH=1: the data-parallel step is captured too, collectives included -- train_graph.py)
    dp_graph = sync is not None and os.environ.get("MAU_DP_GRAPH", "0") == "1"
    gstep = GraphedTrainStep(model, optimizer, criterion, clip_grad_norm=clip, grad_sync=sync) if (graph and (sync is None or dp_graph)) else None
    best, step, ckpt_path, history = float("inf"), 0, None, []
    try:
        for epoch in range(epochs if epochs is not None else cfg.epochs):
            model.train()
            total, num = 0.0, 0
            for _ in range(steps_per_epoch):
                inputs, metadata, temp_series, _lengths, t1, t2, targets = synthetic_batch(cfg.batch_size, CONFIG.device, gen)
                metadata_full = torch.cat([metadata, t1, t2], dim=1) if n_meta >= 8 else metadata  # src/train.py:244
                if gstep is not None:
                    batch_loss = gstep(inputs, temp_series, metadata_full, targets)             # src/train.py:245-256 in one replay
                else:
                    outputs = model(inputs, temp_series, metadata_full)                         # src/train.py:245
                    batch_loss = criterion(outputs, targets).get("total", None)                 # src/train.py:247-249
                    if sync is not None:
                        sync.begin()
                    batch_loss.backward()
                    if sync is not None:
                        sync.finish()
                    if clip > 0:
                        torch.nn.utils.clip_grad_norm_(model.parameters(), clip)                # src/train.py:253-254
                    optimizer.step()
                    optimizer.zero_grad()
                total += batch_loss.detach().cpu().item() * len(inputs)                         # src/train.py:258-260
                if watchdog is not None:
                    watchdog.kick()                                                             # (the read-back above: the step HAS finished)
                num += len(inputs)
                step += 1
            epoch_loss = total / max(num, 1)
            val_loss, _ = validate(model, val_loader, criterion, watchdog)                      # src/train.py:286
            history.append((epoch_loss, val_loss))
            if rank == 0:
                typer.echo(f"Epoch {epoch + 1} | step {step} | Train Loss: {epoch_loss:.6f} | Val Loss: {val_loss:.6f}")
                if val_loss < best:                                                             # src/train.py:303-319
                    best = val_loss
                    name = f"{study_name}_trial_0_best_job{jobid}.pth"
                    ckpt_path = os.path.join(CONFIG.MODELS_DIR, name)
                    with (watchdog.paused() if watchdog is not None else contextlib.nullcontext()):   # slow storage is not a hung collective
                        save_checkpoint(ckpt_path, model, optimizer, epoch=epoch, step=step, loss=best,
                                        hyperparameters=hyper, model_type=model_type, study_name=study_name, trial_id=0,
                                        metadata_input_length=n_meta)
            if watchdog is not None:
                # the other ranks would otherwise sit in the next step's first collective while rank 0 writes: everybody waits HERE, clock
                # stopped, in a barrier of the process group (which has torch's own timeout behind it)
                with watchdog.paused():
                    dist.barrier()
    finally:
        if watchdog is not None:
            watchdog.close()          # also when the loop raises: a caller that handles the exception must not be killed later
    return {"best": best, "model": model, "optimizer": optimizer, "checkpoint_path": ckpt_path, "history": history}


if __name__ == "__main__":
    app()
