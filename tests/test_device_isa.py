"""What the gfx950 assembly of the tile kernels must look like (no GPU needed: hipcc cross-compiles).

* resources: 0 bytes of scratch, at most 80 VGPRs (six waves per SIMD), LDS for two workgroups per CU -- the figures
  DESIGN.md 4.1 states; `make resource-usage` prints them, this test holds them.
* M0: JXLT_LDS_STORE_ROW writes M0 in inline assembly (`s_mov_b32 m0, base; ds_write_addtid_b32`) and cannot
  declare the clobber.  That is sound only while nothing else in the kernel reads or writes M0: every mention of m0
  must be such a move, every ds_write_addtid_b32 must have one a few instructions in front of it, and none of the
  instructions that read M0 implicitly (relative moves, LDS-direct loads, loads to LDS, GWS, messages) may occur.
  (That the hardware takes the whole base, above 64 KB too, is a GPU test: test_gpu_parity.py.)"""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "libjxl-tiny_amd"


@pytest.fixture(scope="module")
def encode_unit_asm():
    r = subprocess.run(["make", "-C", str(PKG), "csrc/jxlt_capi_encode.s"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return (PKG / "csrc" / "jxlt_capi_encode.s").read_text()


def _kernel_body(asm, name):
    """(mangled symbol, instructions without comments) of the kernel whose mangled name contains `name`."""
    m = re.search(r"^(_ZN8jxlt_dev\d+%s\w*):[^\n]*\n(.*?)^\s*\.amdhsa_kernel " % name, asm, re.S | re.M)
    assert m, name
    return m.group(1), [ln.split(";")[0].strip() for ln in m.group(2).splitlines()]


def _metadata(asm, symbol):
    """The kernel's entry of the code object's metadata (amdhsa.kernels) as {key: integer}."""
    for block in asm.split(".amdgpu_metadata", 1)[1].split("\n  - .agpr_count:")[1:]:
        if re.search(r"\.name:\s+%s\n" % re.escape(symbol), block):
            md = {k: int(v) for k, v in re.findall(r"^    \.(\w+):\s+(\d+)$", block, re.M)}
            md["agpr_count"] = int(block.split("\n", 1)[0])
            return md
    raise AssertionError(symbol)


def test_tile_kernel_resources(encode_unit_asm):
    symbol, _ = _kernel_body(encode_unit_asm, "tile12_kernelE")
    md = _metadata(encode_unit_asm, symbol)
    assert md["private_segment_fixed_size"] == 0 and md["vgpr_spill_count"] == 0 and md["sgpr_spill_count"] == 0, md
    assert md["vgpr_count"] <= 80 and md["agpr_count"] == 0, md  # six waves per SIMD
    assert md["group_segment_fixed_size"] <= 81920, md  # two workgroups per CU
    assert md["max_flat_workgroup_size"] == 768 and md["wavefront_size"] == 64, md
    # (the variant with computed roots runs for the rare tiles that file themselves: it may spill, it must fit beside)
    symbol, _ = _kernel_body(encode_unit_asm, "tile12_kernel_redoE")
    assert _metadata(encode_unit_asm, symbol)["group_segment_fixed_size"] <= 81920


@pytest.mark.parametrize("kernel", ["tile12_kernelE", "tile12_kernel_redoE", "tile12_kernel_debugE"])
def test_m0_belongs_to_the_row_stores_alone(encode_unit_asm, kernel):
    _, body = _kernel_body(encode_unit_asm, kernel)
    insts = [ln for ln in body if ln and not ln.startswith(".") and not ln.endswith(":")]
    implicit_readers = re.compile(r"^(v_movrel|s_movrel|v_interp|ds_gws|s_sendmsg|s_set_gpr_idx|ds_read_addtid|"
                                  r"buffer_load\w* .*\blds\b|global_load_lds|v_readlane_b32 .*m0|v_writelane_b32 .*m0)")
    assert not [i for i in insts if implicit_readers.match(i)]
    assert not [i for i in insts if "lds_direct" in i]
    mentions = [i for i in insts if re.search(r"\bm0\b", i)]
    assert mentions and all(re.match(r"^s_mov_b32 m0, (s\d+|0x[0-9a-f]+|\d+)$", i) for i in mentions), mentions[:5]
    stores = [k for k, i in enumerate(insts) if i.startswith("ds_write_addtid_b32")]
    assert stores
    for k in stores:
        # (the macro's move stands right in front of its eight stores)
        window = insts[max(0, k - 12):k]
        assert any(w.startswith("s_mov_b32 m0,") for w in window), (k, window)
        assert not any(w.startswith(("s_cbranch", "s_branch", "s_barrier")) for w in
                       window[max(j for j, w in enumerate(window) if w.startswith("s_mov_b32 m0,")):]), (k, window)
