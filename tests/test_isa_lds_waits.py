"""Build-time ISA check of liblic_hip.so: the LDS-DMA main loops read their MFMA fragments with
inline-asm `ds_read_b128` / `ds_read_b64_tr_b16` and place the `s_waitcnt lgkmcnt(N)` by hand, which hipcc's
own wait-count pass does not track.  This test disassembles every gfx950 code object of the built library and
checks, for every LDS read, that between the read and the first later instruction of the same straight-line
region that touches one of its destination registers there is an `s_waitcnt` that really covers it: LDS
returns in order, so `lgkmcnt(N)` completes this read iff at most N LGKM-counted instructions were issued
after it.  A toolchain bump that moves a fragment register between a read and its wait, or drops a wait,
fails here instead of silently corrupting convolution outputs.  (Reads whose consumer sits behind a branch --
the prefetch that wraps around a loop's back-edge -- are outside a purely textual check; the GPU parity
tests remain the gate for those.)  CPU only."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

_REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
_LGKM = re.compile(r"lgkmcnt\((\d+)\)")


def _regs(text):
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def _is_lgkm(mn):
    return mn.startswith("ds_") or mn.startswith("s_load") or mn.startswith("s_buffer_load") or \
        mn.startswith("s_scratch_load") or mn in ("s_memtime", "s_memrealtime", "s_sendmsg")


def _check_kernel(name, insns):
    """insns: list of (mnemonic, operand text).  Returns (reads checked, list of violations)."""
    bad, checked = [], 0
    for i, (mn, ops) in enumerate(insns):
        if not (mn.startswith("ds_read") or mn.startswith("ds_load")):
            continue
        dst = _regs(ops.split(",")[0])
        if not dst:
            continue
        younger, covered = 0, False
        for mn2, ops2 in insns[i + 1:]:
            if mn2.startswith("s_branch") or mn2.startswith("s_cbranch") or mn2 in ("s_endpgm", "s_setpc_b64",
                                                                                     "s_swappc_b64"):
                break  # consumer (if any) lies behind control flow: not decidable textually
            if mn2 == "s_waitcnt":
                m = _LGKM.search(ops2)
                if m is not None and int(m.group(1)) <= younger:
                    covered = True
            elif mn2.startswith("s_waitcnt_lgkmcnt"):
                covered = True  # explicit-operand form waits for zero
            if mn2 != "s_waitcnt" and _regs(ops2) & dst:
                checked += 1
                if not covered:
                    bad.append(f"{name}: `{mn} {ops}` (#{i}) reaches `{mn2} {ops2}` without a covering s_waitcnt "
                               f"lgkmcnt (<= {younger} needed)")
                break
            if _is_lgkm(mn2):
                younger += 1
    return checked, bad


def _disassemble(tmp_path):
    lib = os.path.join(ROOT, "neural_image_compression_amd", "liblic_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    work = tmp_path / "isa"
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    kernels = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], cwd=work, check=True, capture_output=True,
                             text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = kernels.setdefault(m.group(1), [])
                continue
            if cur is None or not line.startswith("\t"):
                continue
            body = line.split("//")[0].strip()
            if not body:
                continue
            parts = body.split(None, 1)
            cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
    return kernels


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs ROCm's llvm-objdump")
def test_every_lds_read_is_covered_by_a_wait_before_its_first_use(tmp_path):
    kernels = _disassemble(tmp_path)
    glds = [k for k in kernels if "igemm_kernel" in k and k.endswith("Lb1EEv11IgemmParams")]
    assert glds, "the LDS-DMA igemm variants were not found in the library"
    total, violations = 0, []
    counted_waits = 0
    for name, insns in kernels.items():
        n, bad = _check_kernel(name, insns)
        total += n
        violations += bad
        if name in glds:
            counted_waits += sum(1 for mn, ops in insns if mn == "s_waitcnt" and (_LGKM.search(ops) or [None, "0"])[1] != "0")
    assert not violations, "\n".join(violations[:10])
    assert total > 1000, f"only {total} LDS reads checked: the disassembly was not parsed as expected"
    # the hand-placed counted waits (lgkmcnt(TM+TN), not 0) are still what the main loops use
    assert counted_waits >= 6 * len(glds), (counted_waits, len(glds))


def test_the_checker_itself_flags_uncovered_reads():
    ok = [("ds_read_b128", "v[2:5], v66"), ("ds_read_b128", "v[6:9], v66 offset:1024"), ("s_waitcnt", "lgkmcnt(1)"),
          ("v_mfma_f32_32x32x2_f32", "a[0:15], v2, v10, a[0:15]"), ("s_waitcnt", "lgkmcnt(0)"),
          ("v_mfma_f32_32x32x2_f32", "a[0:15], v6, v10, a[0:15]")]
    assert _check_kernel("k", ok) == (2, [])
    too_early = [("ds_read_b128", "v[2:5], v66"), ("ds_read_b128", "v[6:9], v66 offset:1024"), ("s_waitcnt", "lgkmcnt(1)"),
                 ("v_mfma_f32_32x32x2_f32", "a[0:15], v6, v10, a[0:15]")]       # the YOUNGER read is still in flight
    n, bad = _check_kernel("k", too_early)
    assert n == 1 and len(bad) == 1 and "v[6:9]" in bad[0]        # (the first read has no consumer here)
    moved = [("ds_read_b128", "v[2:5], v66"), ("v_mov_b32_e32", "v20, v3"), ("s_waitcnt", "lgkmcnt(0)")]
    assert len(_check_kernel("k", moved)[1]) == 1                                # a fragment register copied before the wait
    behind_branch = [("ds_read_b128", "v[2:5], v66"), ("s_cbranch_scc1", "65000"), ("v_mov_b32_e32", "v20, v3")]
    assert _check_kernel("k", behind_branch) == (0, [])


# ---------------------------------------------------------------------------------------------------------------
# The hand-over rule of the 8-wave ping-pong tiles (igemm_bf16_kernel<256, TN, ..., 4, 8>): the two wave
# groups read LDS rows and weight-panel pieces that OTHER waves DMA'd, so a chunk may be read only one phase after
# the wait that retires it: every `s_barrier` behind which fragments are read must be preceded -- after the last
# LDS-DMA issue in front of it -- by an `s_waitcnt vmcnt(N)` with N <= (RING - 2) * NL (NL = DMA instructions per
# thread per chunk; RING = 4), i.e. every wave has waited for ITS pieces of the chunk before anybody passes the
# barrier.  The racy version of round 2 (e6ea9d4) waited AFTER that barrier, which orders only a wave's own pieces;
# it passed every parity test and produced NaNs in the two-stream training step.  This checker flags it (verified
# on that commit's kernel, tools/check_handover_on_commit.sh) and passes on the kernels in the tree.
# ---------------------------------------------------------------------------------------------------------------
_VMC = re.compile(r"vmcnt\((\d+)\)")


def _check_handover(name, insns, bound):
    """every fragment-read group (ds_read_b128 right behind an s_barrier) must have, between the last DMA issue in
    front of that barrier and the barrier, a wait vmcnt(<= bound).  Returns (groups checked, violations)."""
    bad, checked = [], 0
    for i, (mn, ops) in enumerate(insns):
        if mn != "ds_read_b128" or i == 0:
            continue
        # first read of a group: walk back over non-LDS instructions to the barrier it sits behind
        # (in textual order, through the branches of conditional waits: the racy version had some here)
        j = i - 1
        while j >= 0 and insns[j][0] not in ("s_barrier", "ds_read_b128", "global_load_lds_dwordx4") and \
                not insns[j][0].startswith("v_mfma"):
            j -= 1
        if j < 0 or insns[j][0] != "s_barrier":
            continue
        checked += 1
        k, ok, seen_dma = j - 1, False, False
        while k >= 0:
            m2, o2 = insns[k]
            if m2 == "global_load_lds_dwordx4":
                seen_dma = True
                break
            if m2 in ("s_endpgm", "s_setpc_b64"):
                break
            if m2 == "s_waitcnt":
                mm = _VMC.search(o2)
                if mm is not None and int(mm.group(1)) <= bound:
                    ok = True
            k -= 1
        if seen_dma and not ok:
            bad.append(f"{name}: fragment reads at #{i} follow a barrier with no vmcnt(<= {bound}) between the DMA issue "
                       f"at #{k} and the barrier at #{j}")
    return checked, bad


def _nl_bound(name):
    """(RING - 2) * NL of igemm_bf16_kernel<256, TN, ., ., 4, 8>: APASS = 2, BPASS = ceil(256 TN / 512)"""
    m = re.search(r"igemm_bf16_kernelILi256ELi(\d)E", name)
    tn = int(m.group(1))
    return 2 * (2 + (256 * tn + 511) // 512)


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs ROCm's llvm-objdump")
def test_eight_wave_tiles_wait_for_their_dma_before_the_barrier_that_publishes_it(tmp_path):
    kernels = _disassemble(tmp_path)
    eight = {k: v for k, v in kernels.items() if "igemm_bf16_kernelILi256E" in k and k.endswith("Li8EEv12IgemmHParams")}
    assert len(eight) >= 6, sorted(eight)
    total = 0
    for name, insns in eight.items():
        n, bad = _check_handover(name, insns, _nl_bound(name))
        assert not bad, "\n".join(bad[:5])
        total += n
    assert total >= 3 * len(eight), total   # the unrolled ring: several read groups per kernel


def test_the_handover_checker_flags_a_wait_behind_the_barrier():
    dma = ("global_load_lds_dwordx4", "v[0:1], off")
    good = [dma, ("s_waitcnt", "vmcnt(8) lgkmcnt(0)"), ("s_barrier", ""), ("v_mfma_f32_32x32x16_bf16", "a[0:15], v[0:3], v[4:7], a[0:15]"),
            ("s_barrier", ""), ("ds_read_b128", "v[0:3], v9"), ("ds_read_b128", "v[4:7], v9 offset:16")]
    assert _check_handover("k", good, 8) == (1, [])
    # the racy shape: the wait sits BEHIND the barrier (it orders only the wave's own pieces)
    racy = [dma, ("s_waitcnt", "lgkmcnt(0)"), ("s_barrier", ""), ("v_mfma_f32_32x32x16_bf16", "a[0:15], v[0:3], v[4:7], a[0:15]"),
            ("s_barrier", ""), ("s_cbranch_vccz", "12"), ("s_waitcnt", "vmcnt(8)"), ("ds_read_b128", "v[0:3], v9")]
    assert len(_check_handover("k", racy, 8)[1]) == 1
    shallow = [dma, ("s_waitcnt", "vmcnt(12)"), ("s_barrier", ""), ("ds_read_b128", "v[0:3], v9")]
    assert len(_check_handover("k", shallow, 8)[1]) == 1      # waits, but leaves too many pieces in flight
