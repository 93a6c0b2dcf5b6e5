"""Step preparation: every parameter-derived buffer of a model refreshed by ONE kernel launch per
optimizer step (`lic_prep_run`, include/lic.h) instead of ~90 small launches spread through forward and
backward: the packed MFMA operand of every conv / convT weight for the forward and for the data gradient,
beta_eff and the two packed gamma_eff panels of every GDN, the column-matrix forms of the RGB stem / head
weights, and the in-place masking of the context model's weight (ContextModels.py:19).

`StepPrep(model).run()` is called at the top of the model's forward.  It compares the parameters' version
counters (bumped by optimizer.step()) with what the buffers were built from and launches only when something
changed, then publishes the buffers in `functional.PREPARED`, where the autograd Functions look them up by
(parameter, kind, version).  Anything not found there is packed on the fly exactly as before, so
stand-alone use of the functional layer does not depend on this module.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import List

import torch

from . import _lib as L
from . import functional as F_
from .layers import Conv2d, ConvTranspose2d, GDN, _pair

BF16 = torch.bfloat16
# sub-modules of the models that carry a `precision` attribute (models.set_precision)
_PRECISION_OWNERS = ("encoder", "decoder", "hyper_encoder", "hyper_decoder", "context_model", "entropy_parameters")


def _drop(keys):
    for k in keys:
        F_.PREPARED.pop(k, None)
    del keys[:]


class StepPrep:
    @property
    def model(self):
        return self._model()

    def __init__(self, model: torch.nn.Module):
        self._model = weakref.ref(model)   # (the model owns this object: no reference cycle)
        self._keys = []                    # registry keys published so far; dropped with the model
        weakref.finalize(model, _drop, self._keys)
        self._sig = None
        self._jobs_dev = None
        self._entries = []   # (param, kind, tensor)
        self._params: List[torch.nn.Parameter] = []
        self._versions = None
        self.launches = 0

    # ------------------------------------------------------------------------------------------
    def _signature(self):
        """what the job table was built for: device, precisions, parameter storage.  Walking `model.parameters()`
        costs ~0.2 ms of host time, so the parameter list is cached and re-read only when a cheap sentinel (count
        of registered parameters is not tracked by torch; the first and last storages are) moves."""
        m = self.model
        prec = tuple(getattr(getattr(m, n, None), "precision", "fp32") for n in _PRECISION_OWNERS)
        plist = getattr(self, "_plist", None)
        if plist is None:
            plist = self._plist = list(m.parameters())
        return (plist[0].device, prec, plist[0].data_ptr(), plist[-1].data_ptr(), len(plist))

    def _bf16_modules(self):
        ids = set()
        for n in _PRECISION_OWNERS:
            st = getattr(self.model, n, None)
            if st is not None and getattr(st, "precision", "fp32") == "bf16":
                ids.update(id(q) for q in st.modules())
        return ids

    def _build(self):
        lib = L.load()
        dev = next(self.model.parameters()).device
        bf16_ids = self._bf16_modules()
        jobs, entries, params = [], [], []

        def job(kind, src, dst, *, taps=1, K=1, N=1, s_tap=0, s_k=0, s_n=0, kdiv=0, s_kr=0, ndiv=0, s_nr=0, mask=None,
                transform=0, bound=0.0, pedestal=0.0):
            j = L.PrepJob()
            j.src, j.dst, j.mask = src.data_ptr(), (None if dst is None else dst.data_ptr()), \
                (None if mask is None else mask.data_ptr())
            j.s_tap, j.s_kq, j.s_kr, j.s_nq, j.s_nr = s_tap, s_k, s_kr, s_n, s_nr
            j.kind, j.taps, j.K, j.N, j.kdiv, j.ndiv = kind, taps, K, N, kdiv, ndiv
            j.transform, j.bound, j.pedestal = transform, bound, pedestal
            jobs.append(j)

        def pack(param, kind_name, half, kperm=False, **kw):
            taps, K, N = kw.get("taps", 1), kw["K"], kw["N"]
            if half:
                dst = torch.empty((lib.lic_packed_weight_bf16_elems(taps, K, N),), device=dev, dtype=BF16)
            else:
                dst = torch.empty((lib.lic_packed_weight_floats(taps, K, N),), device=dev, dtype=torch.float32)
            job((L.PREP_PACK_BF16_KPERM if kperm else L.PREP_PACK_BF16) if half else L.PREP_PACK_F32, param, dst, **kw)
            entries.append((param, ("bf16." if half else "f32.") + kind_name, dst))

        for m in self.model.modules():
            half = id(m) in bf16_ids
            if isinstance(m, GDN):
                Cc = m.beta.numel()
                ped = m.beta_reparam.pedestal_value
                beta_e = torch.empty((Cc,), device=dev, dtype=torch.float32)
                job(L.PREP_MAP, m.beta, beta_e, N=Cc, transform=1, bound=m.beta_reparam.bound_value, pedestal=ped)
                entries.append((m.beta, "f32.beta_e", beta_e))
                gkw = dict(K=Cc, N=Cc, transform=1, bound=m.gamma_reparam.bound_value, pedestal=ped)
                pack(m.gamma, "gdn_gT", half, s_k=1, s_n=Cc, **gkw)     # B operand [k = j][n = i] = gamma_eff[i][j]
                if half:   # the same operand in the K order of the one-launch conv+GDN kernel
                    pack(m.gamma, "gdn_gTp", True, kperm=True, s_k=1, s_n=Cc, **gkw)
                pack(m.gamma, "gdn_g", half, s_k=Cc, s_n=1, **gkw)      # backward: t . gamma_eff
                if half:   # ... in the K order of the one-sweep backward kernel (lic_gdn_bwd_bf16)
                    pack(m.gamma, "gdn_gp", True, kperm=True, s_k=Cc, s_n=1, **gkw)
                params += [m.beta, m.gamma]
            elif isinstance(m, (Conv2d, ConvTranspose2d)):
                w = m.weight
                if not w.is_contiguous() or m.groups != 1:
                    continue
                d0, d1, kh, kw_ = w.shape
                taps = kh * kw_
                tr = isinstance(m, ConvTranspose2d)
                cin, cout = (d0, d1) if tr else (d1, d0)
                s_ci, s_co = (d1 * taps, taps) if tr else (taps, d1 * taps)
                mask = getattr(m, "mask", None) if hasattr(m, "_tap_mask") else None
                if mask is not None:
                    job(L.PREP_MASK_INPLACE, w, None, N=w.numel(), mask=mask)
                    entries.append((w, "masked", w))
                if not tr and cin < 4:      # RGB stem: dense [taps*Cin][Cout] column matrix
                    pack(w, "stem", half, K=taps * cin, N=cout, kdiv=cin, s_k=1, s_kr=taps, s_n=cin * taps, mask=mask)
                    if half and mask is None and lib.lic_stem_gdn_bf16_supported(cin, cout, kh, kw_, _pair(m.stride),
                                                                                 _pair(m.padding)):
                        dst = torch.empty((lib.lic_stem_weight_bf16_elems(cout),), device=dev, dtype=BF16)
                        job(L.PREP_PACK_BF16_STEM, w, dst, N=cout)     # operand of the one-launch stem + GDN
                        entries.append((w, "bf16.stem16", dst))
                elif tr and cout < 4:       # RGB head: [Cin][taps*Cout] and its transpose for the data gradient
                    pack(w, "head", half, K=cin, N=taps * cout, s_k=cout * taps, ndiv=cout, s_n=1, s_nr=taps)
                    pack(w, "head_dx", half, K=taps * cout, N=cin, kdiv=cout, s_k=1, s_kr=taps, s_n=cout * taps)
                    if half and lib.lic_head_convt_bf16_supported(cin, cout, kh, kw_, _pair(m.stride), _pair(m.padding),
                                                                  _pair(m.output_padding)):
                        # the data gradient as a direct convolution of the image gradient (lic_stem_conv_bf16)
                        dst = torch.empty((lib.lic_stem_weight_bf16_elems(cin),), device=dev, dtype=BF16)
                        job(L.PREP_PACK_BF16_STEM, w, dst, N=cin)
                        entries.append((w, "bf16.head_dx16", dst))
                else:
                    pack(w, "fwd", half, taps=taps, K=cin, N=cout, s_tap=1, s_k=s_ci, s_n=s_co, mask=mask)
                    pack(w, "dgrad", half, taps=taps, K=cout, N=cin, s_tap=1, s_k=s_co, s_n=s_ci, mask=mask)
                params.append(w)
        if not jobs:
            self._jobs_dev = None
            return
        arr = (L.PrepJob * len(jobs))(*jobs)
        total = lib.lic_prep_plan(arr, len(jobs))
        if total <= 0:
            raise L.LicError(f"lic_prep_plan failed: {total}")
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self._jobs_dev = raw.to(dev)
        self._njobs, self._blocks = len(jobs), int(total)
        self._entries, self._params = entries, params
        self._versions = None
        self._ptrs = None

    # ------------------------------------------------------------------------------------------
    def run(self):
        """Refresh the derived buffers if any parameter changed since they were built (one launch)."""
        sig = self._signature()
        if sig != self._sig:
            self._plist = None          # storage moved (model.to(...), precision switch): re-read everything
            self._sig = self._signature()
            self._build()
        if self._jobs_dev is None:
            return
        # a parameter whose STORAGE was replaced (p.data = ..., a re-allocation) leaves dangling pointers in the device
        # job table: the table is rebuilt; a changed version counter only re-runs the launch
        ptrs = [p.data_ptr() for p in self._params]
        if ptrs != getattr(self, "_ptrs", None) and getattr(self, "_ptrs", None) is not None:
            self._plist = None
            self._sig = self._signature()
            self._build()
            if self._jobs_dev is None:
                return
            ptrs = [p.data_ptr() for p in self._params]
        self._ptrs = ptrs
        versions = [p._version for p in self._params]
        if versions == self._versions:
            return
        L.check(L.load().lic_prep_run(C.c_void_p(self._jobs_dev.data_ptr()), self._njobs, self._blocks, F_._stream()),
                "lic_prep_run")
        self.launches += 1
        self._versions = versions
        for p, kind, t in self._entries:
            key = (id(p), kind)
            if key not in F_.PREPARED:
                self._keys.append(key)
            F_.PREPARED[key] = (p._version, t, weakref.ref(p))

    def invalidate(self):
        """Force the next run() to re-derive every buffer.  Needed after writes that do not bump a parameter's
        version counter: `p.data.copy_()` / `p.data.mul_()` (EMA swaps, manual initialisation), raw-pointer kernels,
        collectives on `.data`.  (optimizer steps, `load_state_dict` and in-place ops on the Parameter itself bump the
        counter and need nothing.)"""
        self._versions = None

    def release(self):
        _drop(self._keys)
        self._entries, self._versions, self._sig = [], None, None
