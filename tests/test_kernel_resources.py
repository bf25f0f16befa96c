"""What the compiler made of the trainer kernel, read from the built library's gfx950 code objects (no GPU needed).

The walk of k_adagrad_runs keeps a resident row pair in registers and counts its own memory waits; two ways of losing that
silently were met while it was written and are pinned here:
  * a construct that makes the compiler keep the run's scalars in an indexed private array puts 400 - 900 bytes of scratch under every
    lane and triples the epoch (seen with `lane == 0 ? ab : lane == 1 ? gab : hab` in close_run) -- no instance may use scratch
    beyond the handful of spill slots the launch-bounded instances have outside the walk;
  * the bench instance (fp32 fat rows, four floats per lane, one register chunk) must fit five waves per SIMD: 96 registers.
Skipped where the LLVM tools of the ROCm image are missing.
"""
import os
import re
import struct
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "graph-embeddings_amd", "lib", "libgeglove.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _kernel_metadata():
    """{kernel name: {field: int}} for every kernel of every gfx950 code object in libgeglove.so's .hip_fatbin."""
    objcopy, readelf = os.path.join(LLVM, "llvm-objcopy"), os.path.join(LLVM, "llvm-readelf")
    if not (os.path.exists(objcopy) and os.path.exists(readelf)):
        pytest.skip("llvm-objcopy / llvm-readelf not found under " + LLVM)
    if not os.path.exists(LIB):
        pytest.skip("libgeglove.so not built")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin.bin")
        subprocess.run([objcopy, "--dump-section", ".hip_fatbin=" + fat, LIB, os.path.join(tmp, "discard")], check=True)
        blob = open(fat, "rb").read()
        pos = 0
        while True:
            at = blob.find(MAGIC, pos)
            if at < 0:
                break
            (count,) = struct.unpack_from("<Q", blob, at + len(MAGIC))
            cur = at + len(MAGIC) + 8
            for _ in range(count):
                offset, size, tlen = struct.unpack_from("<QQQ", blob, cur)
                cur += 24
                triple = blob[cur:cur + tlen].decode()
                cur += tlen
                if "gfx950" not in triple or size == 0:
                    continue
                elf = os.path.join(tmp, "co.elf")
                open(elf, "wb").write(blob[at + offset:at + offset + size])
                notes = subprocess.run([readelf, "--notes", elf], check=True, capture_output=True, text=True).stdout
                for block in notes.split("    .name:")[1:]:
                    name = block.split()[0]
                    # a kernel's fields surround its .name line: take them from the text between the neighbouring .args / .name markers
                    start = notes.index("    .name:" + block[:len(name) + 12])
                    head = notes.rfind("- .", 0, start)
                    tail = notes.find("\n  - .", start)
                    text = notes[head:tail if tail > 0 else len(notes)]
                    out[name] = {k: int(v) for k, v in re.findall(r"\.(private_segment_fixed_size|vgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)", text)}
            pos = at + len(MAGIC)
    return out


def test_trainer_kernel_instances_keep_their_state_in_registers():
    meta = {k: v for k, v in _kernel_metadata().items() if "k_adagrad_runs" in k}
    assert len(meta) >= 80, "expected every (vector width, chunks, optimiser, bf16, fat) instance, found %d" % len(meta)
    for name, m in meta.items():
        assert m["private_segment_fixed_size"] <= 64, (name, m)          # spill slots of the launch-bounded instances: 20 - 28 bytes
    bench = [v for k, v in meta.items() if "ILi4ELi1ELi0ELb0ELb1E" in k]      # <VW 4, NCH 1, AdaGrad, fp32, fat>
    assert len(bench) == 1
    assert bench[0]["vgpr_count"] <= 96 and bench[0]["vgpr_spill_count"] == 0 and bench[0]["private_segment_fixed_size"] == 0, bench[0]
    # two LDS images per wave (row + accumulator row, 1 KB each) x 4 waves + the atomic strips: five workgroups fit a CU's 160 KB
    assert bench[0]["group_segment_fixed_size"] * 5 <= 160 * 1024, bench[0]
