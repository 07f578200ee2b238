"""ISA guard (no GPU needed): the built library must not contain packed-fp32 VALU instructions with an `op_sel` modifier.
On gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` returns wrong lanes while another wave on the same SIMD executes f16 / bf16
MFMAs (profiles/r02_fault_rootcause.md, tools/micro/pkfma_beside_mfma.hip) -- the round-1 'co-residency fault'. The SLP
vectoriser produces that form; the Makefile builds with -fno-slp-vectorize and this test keeps it that way."""
import os
import re
import shutil
import struct
import subprocess

import pytest

from hcflow_amd import _lib

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def gfx950_code_objects(path):
    """Carve the gfx950 ELF images out of the clang offload bundles embedded in the host shared object."""
    data = open(path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = data.find(magic, pos)
        if i < 0:
            return out
        (nb,) = struct.unpack_from("<Q", data, i + 24)
        off = i + 32
        for _ in range(nb):
            eo, es, ts = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + ts].decode()
            off += ts
            if "gfx950" in triple and es > 0:
                out.append(data[i + eo:i + eo + es])
        pos = i + 24


@pytest.mark.skipif(not os.path.exists(OBJDUMP) or shutil.which("true") is None, reason="llvm-objdump not available")
def test_no_packed_fp32_with_op_sel(tmp_path):
    path = _lib.LIB_PATH
    assert os.path.exists(path), "build the library first (python -c 'import __graft_entry__ as g; g.build()')"
    objs = gfx950_code_objects(path)
    assert objs, "no gfx950 code object found in %s" % path
    bad, packed, mfma = [], 0, 0
    for n, blob in enumerate(objs):
        f = tmp_path / ("co_%d.elf" % n)
        f.write_bytes(blob)
        asm = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", str(f)], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
            if m:
                kernel = m.group(1)
            if "v_mfma" in line:
                mfma += 1
            if re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                packed += 1
                if "op_sel:" in line:
                    bad.append((kernel[:80], line.strip()[:120]))
    assert mfma > 1000, "the disassembly does not look like the conv library (%d MFMAs)" % mfma
    assert not bad, "packed fp32 with op_sel (wrong results beside MFMAs on gfx950): %r" % bad[:5]
