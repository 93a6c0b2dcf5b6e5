"""Build-time ISA audit of halo_conv_bf16_kernel (lic_halo_bf16.h).  Its tap loop hides two kinds of loads from
hipcc -- the weight fragments (`global_load_dwordx4` into a ring of register sets, four taps ahead) and the A
fragments (`ds_read_b128`, one tap ahead) are inline asm, waited for by hand with counted `s_waitcnt vmcnt(N)
lgkmcnt(0)` -- so nothing in the compiler stops it from copying, spilling or re-using one of those registers
while the load is still in flight (cdna_hip_programming.md 5.7: "VGPR destination counts as written at ASMEND").
This test disassembles the built library and checks, on the unrolled two-chunk loop body (doubled, so that loads
whose consumer sits behind the back-edge are followed round):
  * every weight load reaches its first later reference only behind an `s_waitcnt vmcnt(N)` that really covers it
    (VMEM retires in order: N must not exceed the vector-memory operations issued after the load);
  * the body contains no scratch access and no AGPR <-> VGPR move (no spill code in the hot loop), and exactly the
    expected numbers of MFMAs / fragment reads / weight loads / halo DMA pieces;
  * outside the body, no instruction other than the loads themselves names a register that is still in flight when
    the body is left (the four youngest weight sets, the next tap's A fragments) before a full drain.
CPU only (needs ROCm's llvm-objdump)."""
import os
import re
import shutil
import subprocess

import pytest

from test_isa_lds_waits import OBJDUMP, ROOT, _regs

_VM = re.compile(r"vmcnt\((\d+)\)")
CTRL = ("s_branch", "s_cbranch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


def _is_vmem(mn):
    return mn.startswith(("global_", "buffer_", "scratch_", "flat_"))


def _kernels_with_addresses(tmp_path, pattern):
    """{kernel name: [(address, mnemonic, operands, branch target or None)]} of the gfx950 code objects"""
    lib = os.path.join(ROOT, "neural_image_compression_amd", "liblic_hip.so")
    if not os.path.exists(lib):
        import __graft_entry__ as g
        g.build()
    work = tmp_path / "isa"
    work.mkdir()
    shutil.copy(lib, work / "lib.so")
    subprocess.run([OBJDUMP, "--offloading", "lib.so"], cwd=work, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f], cwd=work, check=True, capture_output=True,
                             text=True).stdout
        cur, base = None, 0
        for line in txt.splitlines():
            m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
            if m:
                cur = out.setdefault(m.group(2), []) if pattern in m.group(2) else None
                base = int(m.group(1), 16)
                continue
            if cur is None or not line.startswith("\t"):
                continue
            code, _, comment = line.partition("//")
            parts = code.strip().split(None, 1)
            if not parts:
                continue
            am = re.match(r"\s*([0-9A-Fa-f]+):", comment)
            tm = re.search(r"<[^>]*\+0x([0-9a-f]+)>", comment)
            cur.append((int(am.group(1), 16) if am else -1, parts[0], parts[1] if len(parts) > 1 else "",
                        base + int(tm.group(1), 16) if tm and parts[0].startswith(("s_branch", "s_cbranch")) else None))
    return out


def _loops(insns):
    """[(start index, [(mnemonic, operands)])] of every innermost loop (from the target of a backward branch to that
    branch, no other control flow inside), most MFMAs first"""
    out = []
    index = {a: i for i, (a, _, _, _) in enumerate(insns)}
    for i, (addr, mn, ops, tgt) in enumerate(insns):
        if tgt is None or tgt > addr or tgt not in index:
            continue
        j = index[tgt]
        seg = insns[j:i]
        if any(m.startswith(CTRL) for _, m, _, _ in seg):
            continue
        out.append((j, [(m, o) for _, m, o, _ in seg]))
    out.sort(key=lambda b: -sum(m.startswith("v_mfma") for m, _ in b[1]))
    return out


def _body(insns):
    return _loops(insns)[0]


def _audit_loads(body):
    """weight loads of the doubled body: (checked, violations)"""
    two = body + body
    bad, checked = [], 0
    for i, (mn, ops) in enumerate(body):
        if mn != "global_load_dwordx4":
            continue
        dst = _regs(ops.split(",")[0])
        younger, covered = 0, False
        for mn2, ops2 in two[i + 1:i + 1 + len(body)]:
            if mn2 == "s_waitcnt":
                m = _VM.search(ops2)
                if m is not None and int(m.group(1)) <= younger:
                    covered = True
                continue
            if _regs(ops2) & dst:
                checked += 1
                if not covered:
                    bad.append(f"`{mn} {ops}` (#{i}) reaches `{mn2} {ops2}` with no covering vmcnt (<= {younger} needed)")
                break
            if _is_vmem(mn2):
                younger += 1
    return checked, bad


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs ROCm's llvm-objdump")
def test_halo_kernel_async_registers(tmp_path):
    kernels = _kernels_with_addresses(tmp_path, "halo_conv")
    names = [k for k in kernels if "halo_conv_bf16_kernel" in k]
    assert names, "halo_conv_bf16_kernel was not found in the library"
    # the transposed kernel: four phase loops (two-chunk bodies of 18 / 12 / 12 / 8 steps); each is audited for spill
    # code and for the coverage of its weight loads
    tnames = [k for k in kernels if "halo_convt_bf16_kernel" in k]
    assert tnames, "halo_convt_bf16_kernel was not found in the library"
    for name in tnames:
        loops = [b for _, b in _loops(kernels[name]) if sum(m.startswith("v_mfma") for m, _ in b) >= 100]
        assert sorted(sum(m.startswith("v_mfma") for m, _ in b) for b in loops) == [128, 192, 192, 288], name
        for body in loops:
            mn = [m for m, _ in body]
            steps = mn.count("v_mfma_f32_32x32x16_bf16") // 16
            assert mn.count("ds_read_b128") == 8 * steps and mn.count("global_load_dwordx4") == 4 * steps
            assert mn.count("buffer_load_dwordx4") == 12 and mn.count("s_barrier") == 2
            spill = [m for m in mn if m.startswith("scratch_") or m.startswith("v_accvgpr")]
            assert not spill, f"{name}: spill code inside a tap loop: {spill[:5]}"
            checked, bad = _audit_loads(body)
            assert not bad, "\n".join(bad[:10])
            assert checked == 4 * steps, checked
    for name in names:
        insns = kernels[name]
        start, body = _body(insns)
        mn = [m for m, _ in body]
        tw = 2 if "ILi2" in name else 3
        assert mn.count("v_mfma_f32_32x32x16_bf16") == 50 * 8 * tw, (name, mn.count("v_mfma_f32_32x32x16_bf16"))
        assert mn.count("ds_read_b128") == 50 * 8 and mn.count("global_load_dwordx4") == 50 * 2 * tw
        assert mn.count("buffer_load_dwordx4") == 40 and mn.count("s_barrier") == 2   # (the halo pieces: `... offen lds`)
        spill = [m for m in mn if m.startswith("scratch_") or m.startswith("v_accvgpr")]
        assert not spill, f"{name}: spill code inside the tap loop: {spill[:5]}"
        checked, bad = _audit_loads(body)
        assert not bad, "\n".join(bad[:10])
        assert checked == 50 * 2 * tw, checked
        # registers still in flight when the body is left: the last four taps' weight sets, the last tap's A reads
        loads = [ops for m, ops in body if m == "global_load_dwordx4"][-4 * 2 * tw:]
        reads = [ops for m, ops in body if m == "ds_read_b128"][-8:]
        live = set()
        for ops in loads + reads:
            live |= _regs(ops.split(",")[0])
        first_barrier = next(i for i, (_, m, _, _) in enumerate(insns) if m == "s_barrier")
        outside = insns[first_barrier:start] + insns[start + len(body):]
        touched = []
        for _, m, ops, _ in outside:
            if m in ("global_load_dwordx4", "ds_read_b128") or m == "s_waitcnt":
                continue
            if _regs(ops) & live:
                touched.append(f"{m} {ops}")
        assert not touched, f"{name}: registers with a load in flight are referenced outside the tap loop: {touched[:5]}"


def test_the_audit_flags_an_early_use():
    ok = [("global_load_dwordx4", "v[0:3], v[8:9], off"), ("global_load_dwordx4", "v[4:7], v[8:9], off offset:1024"),
          ("s_waitcnt", "vmcnt(1) lgkmcnt(0)"), ("v_mfma_f32_32x32x16_bf16", "a[0:15], v[0:3], a[16:19], a[0:15]"),
          ("s_waitcnt", "vmcnt(0)"), ("v_mfma_f32_32x32x16_bf16", "a[0:15], v[4:7], a[16:19], a[0:15]")]
    assert _audit_loads(ok) == (2, [])
    early = [ok[0], ok[1], ("s_waitcnt", "vmcnt(1) lgkmcnt(0)"), ok[5]]
    n, bad = _audit_loads(early)
    assert len(bad) == 1 and "v[4:7]" in bad[0]
    copied = [ok[0], ("v_mov_b32_e32", "v20, v1"), ("s_waitcnt", "vmcnt(0)")]
    assert len(_audit_loads(copied)[1]) == 1
