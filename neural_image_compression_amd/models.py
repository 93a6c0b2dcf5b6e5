"""Model classes with the reference's names and forward contract (reference: Models.py:10-338).
`ScalableImageCoding` (Models.py:208-338) cannot execute upstream; it is provided with the
three one-line repairs that let it run, listed in its docstring."""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from . import functional as F_
from .components import (Decoder3x3, Decoder5x5, Encoder3x3, Encoder5x5, HyperDecoder3x3, HyperDecoder5x5,
                         HyperEncoder3x3, HyperEncoder5x5, LatentSpaceTransform)
from .entropy import (ContextModel, EntropyParameters, FactorizedEntropyBottleneck, GaussianConditional,
                      GaussianMixtureConditional)


class _HyperpriorContextModel(nn.Module):
    _stacks = None  # (Encoder, Decoder, HyperEncoder, HyperDecoder)

    def __init__(self, latent_channels: int = 192, K: int = 1):
        super().__init__()
        if not isinstance(latent_channels, int) or latent_channels < 1:
            raise ValueError(f"latent_channels must be int >= 1, got {latent_channels}")
        if not isinstance(K, int) or K < 1:
            raise ValueError(f"K must be int >= 1, got {K}")
        self.M = latent_channels
        self.K = K
        self.H = latent_channels
        self.distribution = 'Mean-Scale Gaussian' if K == 1 else 'Mixture of Gaussians'
        self.conditional = GaussianConditional() if K == 1 else GaussianMixtureConditional()
        Enc, Dec, HEnc, HDec = self._stacks
        self.encoder = Enc(latent_channels=self.M)
        self.decoder = Dec(latent_channels=self.M)
        self.hyper_encoder = HEnc(latent_channels=self.M)
        self.hyper_decoder = HDec(latent_channels=self.M)
        self.factorized_entropy_model = FactorizedEntropyBottleneck(self.M)
        self.context_model = ContextModel(latent_channels=self.M)
        self.entropy_parameters = EntropyParameters(latent_channels=self.M, hyper_latent_channels=self.H,
                                                    K=self.K)

    def set_precision(self, precision: str = "fp32", latent: str = None):
        """"fp32" (default, the reference's arithmetic) or "bf16": bf16 activation storage and bf16 MFMA with
        fp32 accumulation (BASELINE config 3; not in the reference) in the analysis / synthesis
        stacks and -- `latent`, default = `precision` -- in the hyper encoder / decoder, the context model and
        the entropy-parameter MLP.  The latents y, z, the raw entropy parameters, quantisation, every
        likelihood and the loss stay fp32; parameters and their gradients are always fp32."""
        latent = precision if latent is None else latent
        for v in (precision, latent):
            if v not in ("fp32", "bf16"):
                raise ValueError(f"precision must be 'fp32' or 'bf16', got {v!r}")
        self.encoder.precision = self.decoder.precision = precision
        for m in (self.hyper_encoder, self.hyper_decoder, self.context_model, self.entropy_parameters):
            m.precision = latent
        self.hyper_decoder.out_f32 = latent != "bf16"   # psi feeds the bf16 MLP directly
        return self

    def analysis_hyperprior(self, x: torch.Tensor, training: bool = True, noise=None, with_packed_params=False,
                            _fork=None):
        """Everything of `forward` except the synthesis transform (Models.py:49-97): the scope the
        north star quotes its roofline target on.  Returns the out-dict without 'x_hat'
        (`with_packed_params`: plus '_act', the activated entropy parameters as the one packed tensor
        the likelihood / coder-table kernels read).  `_fork(y_in)` is called as soon as y_in exists."""
        if x.shape[2] % 64 or x.shape[3] % 64:
            raise RuntimeError("H and W must be multiples of 64 (phi/psi shapes must agree, Models.py:73)")
        if self.use_step_prep and x.is_cuda:
            self.step_prep().run()   # one launch: every packed weight / GDN re-parametrisation of this step
        F_.AUX_STREAM = self.side_stream() if (x.is_cuda and self.overlap_branches) else None
        try:
            y = self.encoder(x)
        finally:
            F_.AUX_STREAM = None
        if x.is_cuda and self.overlap_branches:
            # backward: once dL/dy is complete the decoder's and the latent side's pending reductions start on the second
            # stream, beside the encoder's backward chain (functional.flush_point)
            y = F_.flush_point(y, self.side_stream())
        # bf16-storage consumers of y (hyper-encoder), y_in (decoder, context model) and z_in (hyper-decoder): their bf16
        # copies come out of the quantisation launches instead of one cast launch per consumer
        fuse = x.is_cuda and os.environ.get("LIC_QUANT_CASTS", "1") != "0"
        lat16 = fuse and self.hyper_encoder.precision == "bf16"
        dec16 = fuse and self.decoder.precision == "bf16"
        if training:
            if noise is None:
                # the reference draws rand_like(z) then rand_like(y) (Models.py:57-58); z's shape is known
                # before z is, and drawing both here lets the decoder branch start right after the encoder
                # (one generator launch for both; each tensor is laid out NHWC like the latent it perturbs, so the
                # quantisation kernel reads it in place -- an NCHW u_z cost a layout-copy launch per step)
                Bn, Mc, hy, wy = y.shape
                hz, wz = (hy + 3) // 4, (wy + 3) // 4
                nz, ny = Bn * hz * wz * Mc, Bn * hy * wy * Mc
                u = torch.rand(nz + ny, device=y.device, dtype=y.dtype)
                uz = u[:nz].view(Bn, hz, wz, Mc).permute(0, 3, 1, 2)
                uy = u[nz:].view(Bn, hy, wy, Mc).permute(0, 3, 1, 2)
            else:
                uz, uy = noise
            y_in = F_.quantize(y, uy, True, cast_in=lat16, cast_out=dec16 or lat16)
        else:
            y_in = F_.quantize(y, None, False, cast_in=lat16, cast_out=dec16 or lat16)
        if _fork is not None:
            _fork(y_in)
        z = self.hyper_encoder(y)
        z_in = F_.quantize(z, uz, True, cast_out=lat16) if training else F_.quantize(z, None, False, cast_out=lat16)
        pz_side = None
        if _fork is not None and x.is_cuda and self.overlap_branches and F_.PLAN_RECORDING and \
                os.environ.get("LIC_FACTORIZED_SIDE", "1") != "0":
            # the factorised likelihood of z_in depends on nothing else of the latent side: it goes behind the decoder on the
            # second stream (forward: into the stretch where that stream idles while this one finishes the entropy
            # parameters; backward: autograd runs its backward there too, off the longer of the two backward chains).
            # Recorded steps only (plan.StepPlan): +0.6 % replayed at config 3, -1 % for the eager fp32 step of config 2
            main_s, side_s = torch.cuda.current_stream(), self.side_stream()
            side_s.wait_stream(main_s)
            with torch.cuda.stream(side_s):
                pz_side = self.factorized_entropy_model.likelihood_and_log(z_in)
            z_in.record_stream(side_s)
            for t in pz_side:
                t.record_stream(main_s)   # (forward() joins the streams before anything reads them)
        both = (self.context_model.precision, self.hyper_decoder.precision)
        if x.is_cuda and (both == ("fp32", "fp32") or (both == ("bf16", "bf16") and not self.hyper_decoder.out_f32)):
            # Models.py:73 `torch.cat([phi, psi], dim=1)` without the copy: the context conv and the hyper
            # decoder's last conv write the two channel ranges of one NHWC buffer (phi first; fp32, or bf16 when
            # both producers hand bf16 features to the bf16 entropy-parameter MLP)
            Bn, _, hh, ww = y_in.shape
            c_phi = self.context_model.masked.out_channels
            c_psi = self.hyper_decoder.net[-1].out_channels
            comb = torch.empty((Bn, hh, ww, c_phi + c_psi), device=x.device,
                               dtype=torch.float32 if both[0] == "fp32" else torch.bfloat16)
            psi = self.hyper_decoder(z_in, out=comb[..., c_phi:])
            phi = self.context_model(y_in, out=comb[..., :c_phi])
            combined = F_.join_channels(phi, psi, comb)
        else:
            psi = self.hyper_decoder(z_in)
            phi = self.context_model(y_in)
            combined = torch.cat([phi, psi], dim=1)  # (bf16 features: a 6 MB layout copy, phi first)
        act = self.entropy_parameters.packed(combined)
        if self.K == 1:
            mu, sigma = self.entropy_parameters.split(act)
            params = {"mu": mu, "sigma": sigma}
        else:
            weights, mus, sigmas = self.entropy_parameters.split(act)
            params = {"weights": weights, "mus": mus, "sigmas": sigmas}
        p_z, logp_z = pz_side if pz_side is not None else self.factorized_entropy_model.likelihood_and_log(z_in)
        p_y, logp_y = self.conditional.packed_likelihood_and_log(y_in, act, self.K)
        out = {'y': y, 'y_in': y_in, 'z': z, 'z_in': z_in, 'p_z': p_z, 'logp_z': logp_z,
               'p_y': p_y, 'logp_y': logp_y, 'training': training}
        out.update(params)
        if with_packed_params:
            out['_act'] = act
        return out

    # The synthesis transform depends only on y_in; the hyper / context / likelihood branch only on
    # (y, y_in): independent until rd_loss.  With `overlap_branches` the decoder runs on a second HIP
    # stream beside the latent-side branch (many small launches that cannot fill 256 CUs by
    # themselves), in forward and -- autograd replays each op on its forward stream -- in backward.
    overlap_branches = True
    # all parameter-derived buffers refreshed by one launch per optimizer step (prep.StepPrep); LIC_STEP_PREP=0
    # restores the per-layer packing launches for an A/B
    use_step_prep = os.environ.get("LIC_STEP_PREP", "1") != "0"

    def __getstate__(self):
        # the step preparation holds weak references and device job tables of THIS object: a pickled / deep-copied
        # model builds its own on first use
        st = dict(self.__dict__)
        st.pop("_step_prep", None)
        st.pop("_side_stream", None)
        return st

    def step_prep(self):
        if getattr(self, "_step_prep", None) is None:
            from .prep import StepPrep
            self._step_prep = StepPrep(self)
        return self._step_prep

    def side_stream(self):
        """the second HIP stream of `overlap_branches` (created on first use)"""
        if getattr(self, "_side_stream", None) is None:
            # High priority: the decoder chain is the critical path of the overlapped region, the latent-side
            # launches on the main stream fill what it leaves (+0.6 % step, and the big decoder launches keep
            # ~5 % more of their stand-alone speed).  LIC_SIDE_PRIORITY=0 for the A/B.
            self._side_stream = torch.cuda.Stream(priority=int(os.environ.get("LIC_SIDE_PRIORITY", "-1")))
        return self._side_stream

    def forward(self, x: torch.Tensor, training: bool = True, noise=None):
        """`noise` (test hook, not in the reference): (u_z, u_y) uniform [0,1) tensors used instead
        of torch.rand_like, in the reference's draw order (z first, Models.py:57-58)."""
        if not (self.overlap_branches and x.is_cuda):
            out = self.analysis_hyperprior(x, training, noise)
            return {'x_hat': self.decoder(out['y_in']), **out}
        main = torch.cuda.current_stream()
        side = self.side_stream()
        box = {}

        def fork(y_in):
            side.wait_stream(main)
            with torch.cuda.stream(side):
                box['x_hat'] = self.decoder(y_in)
            y_in.record_stream(side)

        out = self.analysis_hyperprior(x, training, noise, _fork=fork)
        main.wait_stream(side)
        box['x_hat'].record_stream(main)
        return {'x_hat': box['x_hat'], **out}


class JointAutoregressiveHierarchical(_HyperpriorContextModel):
    """5x5 conv/GDN stacks (Models.py:10-106)."""
    _stacks = (Encoder5x5, Decoder5x5, HyperEncoder5x5, HyperDecoder5x5)


class HierarchicalMixtureResidual(_HyperpriorContextModel):
    """3x3 residual stacks (Models.py:109-205)."""
    _stacks = (Encoder3x3, Decoder3x3, HyperEncoder3x3, HyperDecoder3x3)


class ScalableImageCoding(nn.Module):
    """Models.py:208-338: the 5x5 model whose latent y is split into a base part y1 (`base_channels`) and an
    enhancement part y2, each with its own masked context model and entropy-parameter MLP over the shared hyper
    features, plus a `LatentSpaceTransform` of y1 (`F_tilde`, the features a frozen vision backbone is matched
    against by `vision_rd_loss`).  Same constructor, attributes, state-dict keys and out-dict keys as the reference.
    The reference's forward cannot run; the repairs made here, and nothing else:
      * `self.factorized_entropy_model(z_in, debug)` (Models.py:302) passes a positional argument the entropy
        model does not take -> called without it;
      * in mixture mode the second parameter dict overwrites the first (`params1 = {...weights2...}`,
        Models.py:298-299) -> it is `params2`;
      * `LatentSpaceTransform` keeps its channel count (components.LatentSpaceTransform)."""

    def __init__(self, latent_channels: int = 192, base_channels: int = 128, K: int = 1):
        super().__init__()
        if not isinstance(latent_channels, int) or latent_channels < 1:
            raise ValueError(f"latent_channels must be int >= 1, got {latent_channels}")
        if not isinstance(K, int) or K < 1:
            raise ValueError(f"K must be int >= 1, got {K}")
        if not isinstance(base_channels, int) or not 0 < base_channels < latent_channels:
            raise ValueError(f"base_channels must be an int in (0, latent_channels), got {base_channels}")
        self.M, self.M1, self.M2 = latent_channels, base_channels, latent_channels - base_channels
        self.H, self.K = latent_channels, K
        self.distribution = 'Mean-Scale Gaussian' if K == 1 else 'Mixture of Gaussians'
        self.conditional = GaussianConditional() if K == 1 else GaussianMixtureConditional()
        self.encoder = Encoder5x5(latent_channels=self.M)
        self.decoder = Decoder5x5(latent_channels=self.M)
        self.hyper_encoder = HyperEncoder5x5(latent_channels=self.M)
        self.hyper_decoder = HyperDecoder5x5(latent_channels=self.M)
        self.factorized_entropy_model = FactorizedEntropyBottleneck(self.M)
        self.context_model_1 = ContextModel(latent_channels=self.M1)
        self.context_model_2 = ContextModel(latent_channels=self.M2)
        self.entropy_parameters_1 = EntropyParameters(latent_channels=self.M1, hyper_latent_channels=self.H, K=self.K)
        self.entropy_parameters_2 = EntropyParameters(latent_channels=self.M2, hyper_latent_channels=self.H, K=self.K)
        self.LST = LatentSpaceTransform(latent_channels=self.M1, upsampling_factors=[2, 1, 1, 1])

    use_step_prep = _HyperpriorContextModel.use_step_prep
    step_prep = _HyperpriorContextModel.step_prep

    def forward(self, x: torch.Tensor, training: bool = True, debug=False, noise=None):
        """`noise` (test hook, not in the reference): (u_z, u_y) used instead of torch.rand_like, z first."""
        if x.shape[2] % 64 or x.shape[3] % 64:
            raise RuntimeError("H and W must be multiples of 64 (phi/psi shapes must agree, Models.py:283-284)")
        if self.use_step_prep and x.is_cuda:
            self.step_prep().run()
        y = self.encoder(x)
        z = self.hyper_encoder(y)
        if training:
            uz, uy = noise if noise is not None else (torch.rand_like(z), torch.rand_like(y))
            z_in, y_in = F_.quantize(z, uz, True), F_.quantize(y, uy, True)
        else:
            z_in, y_in = F_.quantize(z, None, False), F_.quantize(y, None, False)
        y1, y2 = torch.split(y_in, [self.M1, self.M2], dim=1)
        psi = self.hyper_decoder(z_in)
        out = {'y': y, 'y_in': y_in, 'y1': y1, 'y2': y2, 'z': z, 'z_in': z_in, 'training': training}
        params = {}
        for tag, yk, ctx, ep in (("1", y1, self.context_model_1, self.entropy_parameters_1),
                                 ("2", y2, self.context_model_2, self.entropy_parameters_2)):
            act = ep.packed(torch.cat([ctx(yk), psi], dim=1))
            names = ("mu", "sigma") if self.K == 1 else ("weights", "mus", "sigmas")
            params.update({n + tag: v for n, v in zip(names, ep.split(act))})
            out['p_y' + tag], out['logp_y' + tag] = self.conditional.packed_likelihood_and_log(yk, act, self.K)
        out['p_z'], out['logp_z'] = self.factorized_entropy_model.likelihood_and_log(z_in)
        out['x_hat'] = self.decoder(y_in)
        out['F_tilde'] = self.LST(y1)
        out.update(params)
        return out
