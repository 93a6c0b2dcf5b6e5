#!/bin/bash
# Runs the hand-over rule of tests/test_isa_lds_waits.py (_check_handover) on the 8-wave kernels of another commit's
# lic_gemm_bf16.hip (default e6ea9d4: the version whose wait sat behind the barrier and raced).  CPU only.
#   tools/check_handover_on_commit.sh [commit]
set -e
C=${1:-e6ea9d4}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
mkdir -p "$W/inc"
for f in lic_gemm_bf16.hip lic_common.h lic_patch.h; do git -C "$ROOT" show "$C:neural_image_compression_amd/csrc/$f" > "$W/$f"; done
git -C "$ROOT" show "$C:neural_image_compression_amd/csrc/lic_halo_bf16.h" > "$W/lic_halo_bf16.h" 2>/dev/null || true
git -C "$ROOT" show "$C:include/lic.h" > "$W/inc/lic.h"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I"$W/inc" -Wno-unused-result --cuda-device-only -S "$W/lic_gemm_bf16.hip" -o "$W/k.s" 2>/dev/null
python3 - "$W/k.s" "$ROOT" <<'PY'
import re, sys
sys.path.insert(0, sys.argv[2] + "/tests")
import test_isa_lds_waits as T
kernels, cur = {}, None
for line in open(sys.argv[1]):
    line = line.rstrip("\n")
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = kernels.setdefault(m.group(1), [])
        continue
    if line.startswith(".Lfunc_end"):
        cur = None
    if cur is None or not line.startswith("\t"):
        continue
    body = line.split(";")[0].strip()
    if not body or body.startswith("."):
        continue
    parts = body.split(None, 1)
    cur.append((parts[0], parts[1] if len(parts) > 1 else ""))
for name, insns in kernels.items():
    if "igemm_bf16_kernelILi256E" in name and name.endswith("Li8EEv12IgemmHParams"):
        n, bad = T._check_handover(name, insns, T._nl_bound(name))
        print(f"{name}: {n} fragment-read groups, {len(bad)} violations")
PY
rm -rf "$W"
