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


def _interval(explicit, max_steps, fraction):
    """the reference's default cadence: max_steps / fraction steps (Trainer.py:27-29)"""
    return explicit if explicit else int(max_steps / fraction)


def _build_scheduler(kind, optimizer, max_steps):
    """'plateau' -> ReduceLROnPlateau(min, patience 100, factor 0.5); 'cosine' -> CosineAnnealingLR(T_max =
    max_steps, eta_min 1e-5); anything else -> none (Trainer.py:32-40).  Returns (scheduler, steps_on_val_loss)."""
    if kind == "plateau":
        return ReduceLROnPlateau(optimizer, mode="min", patience=100, factor=0.5), True
    if kind == "cosine":
        return CosineAnnealingLR(optimizer, T_max=max_steps, eta_min=1e-5), False
    return None, False


class Trainer:
    def __init__(self, model, optimizer, train_loader, val_loader=None, rd_loss=None, lambda_val=0.005,
                 scheduler=None, max_steps=10000, resume=False, log_interval=None, img_interval=None,
                 val_interval=None, log_dir="runs/experiment", checkpoint_path="./checkpoints/checkpoint.pth",
                 device="cuda", distributed: Optional[bool] = None, writer=None, step_plan: bool = False):
        if rd_loss is None:
            raise ValueError("You must provide a rate-distortion loss function (`rd_loss`)")
        self.device = device
        self.model = model.to(device)
        self.optimizer, self.rd_loss, self.lambda_val = optimizer, rd_loss, lambda_val
        self.train_loader, self.val_loader = train_loader, val_loader
        self.train_iter = iter(train_loader)
        self.step, self.max_steps = 0, max_steps
        self.log_interval = _interval(log_interval, max_steps, 200)
        self.img_interval = _interval(img_interval, max_steps, 25)
        self.val_interval = _interval(val_interval, max_steps, 200)
        self.log_statistics = True  # latent / likelihood summaries every log_interval (Trainer.py:88-92)
        self.scheduler, self.use_plateau = _build_scheduler(scheduler, optimizer, max_steps)
        self.resume, self.checkpoint_path = resume, checkpoint_path
        if resume and checkpoint_path is not None and os.path.exists(checkpoint_path):
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
        # step_plan=True: forward + loss + backward are captured on the first batch and replayed by the library's
        # launch plan (plan.StepPlan: same kernels and streams, no Python between launches -- the host-paced bf16
        # configurations gain ~10 %).  One GPU, `rd_loss` must accept sync=False (loss.rd_loss does); a batch of
        # another shape (the last one of an epoch) takes the eager step.
        self.step_plan = bool(step_plan)
        self._plan = None
        if self.step_plan and self.reducer is not None:
            raise ValueError("step_plan replays one GPU's step; data-parallel training uses the eager step")

    # -- checkpointing: the reference's file format (Trainer.py:52-71) ---------------------------
    def _checkpoint_state(self):
        sched = None if self.scheduler is None else self.scheduler.state_dict()
        return {"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "step": self.step,
                "scheduler": sched}

    def save_checkpoint(self):
        if self.rank != 0:   # replicas are identical: one writer
            return
        folder = os.path.dirname(self.checkpoint_path)
        if folder:
            os.makedirs(folder, exist_ok=True)
        torch.save(self._checkpoint_state(), self.checkpoint_path)
        print(f"[trainer] step {self.step}: checkpoint written to {self.checkpoint_path}")

    def load_checkpoint(self):
        state = torch.load(self.checkpoint_path, map_location=self.device)
        self.model.load_state_dict(state["model"])
        self.optimizer.load_state_dict(state["optimizer"])
        if self.scheduler is not None and state["scheduler"] is not None:
            self.scheduler.load_state_dict(state["scheduler"])
        self.step = state["step"]
        self.max_steps += self.step   # `max_steps` counts the steps of THIS run (Trainer.py:70)
        print(f"[trainer] resumed from {self.checkpoint_path} at step {self.step}")

    # -- the step (Trainer.py:78-86) -------------------------------------------------------------
    def _planned_step(self, imgs):
        from .loss import _KEYS
        from .plan import StepPlan
        if self._plan is None:
            self._plan = StepPlan(self.model, self.rd_loss, self.lambda_val, imgs)
        if imgs.shape != self._plan.x.shape:
            return None
        model_out, res = self._plan.step(imgs)
        self.optimizer.step()
        results = dict(res)
        buf = results.pop('_buffer', None)
        if buf is not None:   # the plain numbers the eager rd_loss returns, from one device-to-host copy
            host = buf[:len(_KEYS)].tolist()
            for i, k in enumerate(_KEYS[1:], start=1):
                results[k] = host[i]
        return model_out, results

    def train_step(self, imgs):
        imgs = imgs.to(self.device)
        if self.step_plan and imgs.is_cuda:
            done = self._planned_step(imgs)
            if done is not None:
                return done
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
        """next training batch; the loader is restarted when it runs out (Trainer.py:133-138)"""
        batch = next(self.train_iter, None)
        if batch is None:
            self.train_iter = iter(self.train_loader)
            batch = next(self.train_iter)
        return batch

    def _log_scalars(self, results):
        """every plain number of the loss dict under losses/<key> (Trainer.py:140-143)"""
        if self.writer is None:
            return
        for key, value in results.items():
            if isinstance(value, (float, int)):
                self.writer.add_scalar("losses/" + key, value, self.step)

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
        """mean loss / bpp / PSNR over the validation loader in eval mode (Trainer.py:145-165); under data
        parallelism the three means are averaged over the ranks"""
        sums = [0.0, 0.0, 0.0]
        self.model.eval()
        with torch.no_grad():
            for imgs in self.val_loader:
                imgs = imgs.to(self.device)
                res = self.rd_loss(self.model(imgs, training=False), imgs, self.lambda_val)
                for i, v in enumerate((res["loss"], res["bpp_total"], res["psnr"])):
                    sums[i] += float(v)
        self.model.train()
        means = [v / len(self.val_loader) for v in sums]
        if self.distributed:
            # every rank validates its own shard of the images: average the three numbers over the ranks, so that
            # ReduceLROnPlateau sees the SAME value everywhere and the replicated parameters cannot drift apart
            means = all_reduce_mean_scalars(means, self.device)
        if self.writer is not None:
            # (tag spelling as in the reference, Trainer.py:162-164)
            for tag, v in zip(("validation_loss", "validation_bpp", "validation_pnsr"), means):
                self.writer.add_scalar("validation/" + tag, v, self.step)
        return means[0]
