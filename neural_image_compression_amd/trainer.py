"""Training-loop counterpart of the reference's Trainer (Trainer.py:10-165) with data
parallelism added (the reference is single-device).

Reproduced contract: constructor signature (Trainer.py:11-13), step order zero_grad -> forward ->
rd_loss -> backward -> optimizer.step (:81-86), `_next_batch` wrap-around (:133-138), scheduler
modes 'plateau' / 'cosine' (:32-40, :94-99), interval defaults max_steps/200, /25, /200 (:27-29),
checkpoint dict keys model/optimizer/step/scheduler and `max_steps += step` on resume (:52-71),
scalar tags `losses/<key>` for every float of the loss dict (:140-143) and the three validation
tags (:162-164, including the reference's "validation_pnsr" spelling).
The reference's histogram logging (:167-217) ships whole latent tensors to the host every
`log_interval`; here `_log_histograms` / `_log_channel_activity` / `_log_entropy_params` emit the
same tags from device-side summaries (`data.tensor_stats` -> `lic_tensor_stats`: count, mean, std,
min, max and a 64-bin histogram; 6 doubles + 64 counters cross PCIe per tag).  Image / figure logging
(:219-345) is not reproduced.  Scalars go to a TensorBoard SummaryWriter when tensorboard is
installed, otherwise to <log_dir>/scalars.jsonl (summaries as JSON objects).
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import CosineAnnealingLR, ReduceLROnPlateau

from .parallel import GradientAllReducer, all_reduce_mean_scalars, broadcast_parameters


class _JsonlWriter:
    def __init__(self, log_dir, purge_step=0):
        os.makedirs(log_dir, exist_ok=True)
        self.f = open(os.path.join(log_dir, "scalars.jsonl"), "a")

    def add_scalar(self, tag, value, step):
        self.f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def add_summary(self, tag, summary, step):
        self.f.write(json.dumps({"tag": tag, "summary": summary, "step": int(step)}) + "\n")

    def close(self):
        self.f.close()


def _make_writer(log_dir, purge_step):
    try:
        from torch.utils.tensorboard import SummaryWriter  # noqa: WPS433
        return SummaryWriter(log_dir, purge_step=purge_step)
    except Exception:
        return _JsonlWriter(log_dir, purge_step)


class Trainer:
    def __init__(self, model, optimizer, train_loader, val_loader=None, rd_loss=None, lambda_val=0.005,
                 scheduler=None, max_steps=10000, resume=False, log_interval=None, img_interval=None,
                 val_interval=None, log_dir="runs/experiment", checkpoint_path="./checkpoints/checkpoint.pth",
                 device="cuda", distributed: Optional[bool] = None, writer=None):
        self.model = model.to(device)
        self.optimizer = optimizer
        self.train_loader = train_loader
        self.val_loader = val_loader
        if rd_loss is None:
            raise ValueError("You must provide a rate-distortion loss function (`rd_loss`)")
        self.rd_loss = rd_loss
        self.lambda_val = lambda_val
        self.device = device
        self.max_steps = max_steps
        self.step = 0
        self.train_iter = iter(train_loader)
        self.log_interval = log_interval if log_interval else int(self.max_steps / 200)
        self.log_statistics = True  # latent / likelihood summaries every log_interval (Trainer.py:88-92)
        self.img_interval = img_interval if img_interval else int(self.max_steps / 25)
        self.val_interval = val_interval if val_interval else int(self.max_steps / 200)
        if scheduler == 'plateau':
            self.scheduler = ReduceLROnPlateau(optimizer, mode='min', patience=100, factor=0.5)
            self.use_plateau = True
        elif scheduler == 'cosine':
            self.scheduler = CosineAnnealingLR(optimizer, T_max=max_steps, eta_min=1e-5)
            self.use_plateau = False
        else:
            self.scheduler = None
            self.use_plateau = False
        self.resume = resume
        self.checkpoint_path = checkpoint_path
        if self.resume and self.checkpoint_path is not None and os.path.exists(self.checkpoint_path):
            self.load_checkpoint()
        # data parallel: one process per GPU; gradients all-reduced (mean) once per step
        self.distributed = dist.is_initialized() if distributed is None else distributed
        self.rank = dist.get_rank() if self.distributed else 0
        self.reducer = None
        if self.distributed and dist.get_world_size() > 1:
            broadcast_parameters(self.model)
            if hasattr(self.model, "overlap_branches") and dist.get_backend() != "nccl":
                self.model.overlap_branches = False  # ranks time-sharing one GPU (gloo rehearsals) stall with it
            two = hasattr(self.model, "decoder") and getattr(self.model, "overlap_branches", False) and \
                next(self.model.parameters()).is_cuda
            self.reducer = GradientAllReducer(
                self.model.parameters(),
                stream_groups=[list(self.model.decoder.parameters())] if hasattr(self.model, "decoder") else None,
                group_streams=[self.model.side_stream()] if two else None)
        self.writer = writer if writer is not None else (_make_writer(log_dir, self.step) if self.rank == 0 else None)

    # -- checkpointing (Trainer.py:52-71) --------------------------------------------------------
    def save_checkpoint(self):
        if self.rank != 0:
            return
        checkpoint = {
            "model": self.model.state_dict(),
            "optimizer": self.optimizer.state_dict(),
            "step": self.step,
            "scheduler": self.scheduler.state_dict() if self.scheduler is not None else None,
        }
        d = os.path.dirname(self.checkpoint_path)
        if d:
            os.makedirs(d, exist_ok=True)
        torch.save(checkpoint, self.checkpoint_path)
        print(f"Checkpoint saved at step {self.step} -> {self.checkpoint_path}")

    def load_checkpoint(self):
        checkpoint = torch.load(self.checkpoint_path, map_location=self.device)
        self.model.load_state_dict(checkpoint["model"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        if self.scheduler is not None and checkpoint["scheduler"] is not None:
            self.scheduler.load_state_dict(checkpoint["scheduler"])
        self.step = checkpoint["step"]
        self.max_steps += self.step
        print(f"Checkpoint loaded -> Resuming from step {self.step}")

    # -- the step (Trainer.py:78-86) -------------------------------------------------------------
    def train_step(self, imgs):
        imgs = imgs.to(self.device)
        self.optimizer.zero_grad()
        model_out = self.model(imgs)
        results = self.rd_loss(model_out, imgs, self.lambda_val)
        results['loss'].backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.optimizer.step()
        return model_out, results

    def train(self):
        while self.step < self.max_steps:
            imgs = self._next_batch()
            model_out, results = self.train_step(imgs)
            self._log_scalars(results)
            if self.writer is not None and self.log_interval and self.step % self.log_interval == 0 and \
                    self.log_statistics and self._has_device_outputs(model_out):
                self._log_histograms(model_out)
                self._log_channel_activity(model_out, 'y')
                self._log_channel_activity(model_out, 'z')
                self._log_entropy_params(model_out)
            if self.val_loader is not None and self.val_interval and self.step % self.val_interval == 0:
                val_loss = self._validate()
                if self.use_plateau:
                    self.scheduler.step(val_loss)
            if self.scheduler is not None and not self.use_plateau:
                self.scheduler.step()
            if self.scheduler is not None and self.writer is not None:
                self.writer.add_scalar("train/learning_rate", self.optimizer.param_groups[0]['lr'], self.step)
            self.step += 1
        if self.writer is not None:
            self.writer.close()
        if self.checkpoint_path is not None:
            self.save_checkpoint()

    def _next_batch(self):
        try:
            return next(self.train_iter)
        except StopIteration:
            self.train_iter = iter(self.train_loader)
            return next(self.train_iter)

    def _log_scalars(self, results):
        if self.writer is None:
            return
        for k, v in results.items():
            if isinstance(v, (float, int)):
                self.writer.add_scalar(f"losses/{k}", v, self.step)

    # ---- Trainer.py:167-217 from device-side summaries --------------------------------------
    @staticmethod
    def _has_device_outputs(model_out):
        keys = ("y", "y_in", "z", "z_in", "logp_y", "logp_z", "p_y", "p_z")
        return all(k in model_out and torch.is_tensor(model_out[k]) and model_out[k].is_cuda for k in keys)

    def _summary(self, tag, t):
        from .data import tensor_stats
        st = tensor_stats(t)
        if hasattr(self.writer, "add_summary"):
            self.writer.add_summary(tag, st, self.step)
        else:  # a TensorBoard writer: the scalar moments
            for k in ("mean", "std", "min", "max"):
                self.writer.add_scalar(f"{tag}/{k}", st[k], self.step)
        return st

    def _log_histograms(self, model_out):
        ln2 = 0.6931471805599453
        for tag, key in (("latents/y", "y"), ("latents/y_hat", "y_in"), ("latents/z", "z"), ("latents/z_hat", "z_in"),
                         ("probability/logp_y", "logp_y"), ("probability/logp_z", "logp_z"),
                         ("probability/p_y", "p_y"), ("probability/p_z", "p_z")):
            st = self._summary(tag, model_out[key])
            if key in ("logp_y", "logp_z", "p_y", "p_z"):
                self.writer.add_scalar(f"probability/{key}_mean", st["mean"], self.step)
            if key in ("logp_y", "logp_z"):
                self.writer.add_scalar(f"entropy/entropy_{key[-1]}_mean", -st["mean"] / ln2, self.step)
        for n in ("y", "z"):
            logp = model_out["logp_" + n].detach()
            self._summary(f"entropy/{n}", -logp / ln2)
            self._summary(f"entropy/{n}_per_component", -logp.sum(dim=(2, 3)) / ln2)

    def _log_channel_activity(self, model_out, tensor_name='y'):
        logp = model_out['logp_' + tensor_name].detach()
        avg_bits_per_c = (-logp / 0.6931471805599453).mean(dim=(0, 2, 3))
        self.writer.add_scalar(f"activity/{tensor_name}_dead_channels_by_entropy",
                               float((avg_bits_per_c < 1e-4).float().sum()), self.step)

    def _log_entropy_params(self, model_out):
        if 'mu' in model_out and 'sigma' in model_out:
            self._summary("entropy_params/mu", model_out['mu'])
            self._summary("entropy_params/sigma", model_out['sigma'])
        if 'weights' in model_out:
            for k in ("weights", "mus", "sigmas"):
                self._summary(f"entropy_params/{k}", model_out[k])
            used = (model_out['weights'].detach() > 1e-4).float().sum(dim=1).mean()
            self.writer.add_scalar("entropy_params/used_components_mean", float(used), self.step)

    def _validate(self):
        self.model.eval()
        total_loss = bpp_loss = psnr_loss = 0.0
        with torch.no_grad():
            for imgs in self.val_loader:
                imgs = imgs.to(self.device)
                model_out = self.model(imgs, training=False)
                results = self.rd_loss(model_out, imgs, self.lambda_val)
                total_loss += float(results['loss'])
                bpp_loss += results['bpp_total']
                psnr_loss += results['psnr']
        self.model.train()
        n = len(self.val_loader)
        avg_loss, avg_bpp, avg_psnr = total_loss / n, bpp_loss / n, psnr_loss / n
        if self.distributed:
            # every rank validates its own shard of the images: average the three numbers over the ranks, so that
            # ReduceLROnPlateau sees the SAME value everywhere and the replicated parameters cannot drift apart
            avg_loss, avg_bpp, avg_psnr = all_reduce_mean_scalars((avg_loss, avg_bpp, avg_psnr), self.device)
        if self.writer is not None:
            self.writer.add_scalar("validation/validation_loss", avg_loss, self.step)
            self.writer.add_scalar("validation/validation_bpp", avg_bpp, self.step)
            self.writer.add_scalar("validation/validation_pnsr", avg_psnr, self.step)
        return avg_loss
