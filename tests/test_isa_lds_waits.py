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
