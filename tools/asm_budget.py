#!/usr/bin/env python3
"""Per-phase instruction budget of tile12_kernel from its gfx950 assembly (no GPU needed).

Builds jxlt_capi_encode.hip with -DJXLT_ASM_MARKERS (an assembly comment at every phase boundary of the kernel), splits the
kernel's instruction stream at the markers and prices every VALU instruction with the issue costs measured by
tools/valu_issue_probe.hip / valu_issue_probe2.hip on the MI355X (profiles/r02_valu_issue_probe.txt,
profiles/r05_valu_issue_probe2.txt):

  cheap (2 cycles per wave64 instruction and SIMD)   v_add/sub/mul/fma/fmac/fmamk/fmaak_f32, v_mov_b32, v_add_u32 / sub_u32,
                                                     v_and / or / xor_b32 -- with VGPR, inline-constant or literal sources
  full  (4 cycles)                                   v_cndmask, v_cmp, v_min / max / med3, v_rndne / trunc / floor, v_cvt,
                                                     shifts, v_bfe, v_mul_lo / mul_hi / mad_*24, DPP forms,
                                                     ANY instruction with an SGPR source, three VGPR sources in one bank
  trans (8 cycles)                                   v_rcp, v_sqrt, v_rsq, v_log, v_exp

STATIC counts: loops (P1's band loop, the chroma-from-luma chain, the scan-order quantisation) are counted once; the
table therefore sits beside the DYNAMIC counts per phase that tools/phase_pmc.sh measures (SQ_INSTS_VALU per wave), and
the static class mix of a phase is what is applied to its dynamic count.

Usage: tools/asm_budget.py [--flags "<extra hipcc flags>"] [--kernel tile12_kernel] [--dump-phase NAME]
"""
import argparse
import collections
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HIPFLAGS = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "
            "-Wno-unused-function").split()

CHEAP = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmamk_f32", "v_fmaak_f32",
         "v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b64",
         "v_pk_mov_b32"}
# (priced by valu_issue_probe2: see profiles/r05_valu_issue_probe2.txt; until then assumed full-rate = 4 cycles)
TRANS = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_log_f32", "v_exp_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32"}


def strip_suffix(op):
    return re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)


def classify(op, operands, extra_cheap=()):
    """-> (class name, cycles)"""
    base = strip_suffix(op)
    is_dpp = op.endswith("_dpp") or "quad_perm" in operands or "row_sh" in operands or "row_mask" in operands
    if base in TRANS:
        return "trans", 8
    if base in ("v_readlane_b32", "v_readfirstlane_b32", "v_writelane_b32"):
        return "lane<->scalar", 4
    if base.startswith("v_permlane"):
        return "permlane", 4
    if is_dpp:
        return "dpp", 4
    if base.startswith("v_cndmask"):
        return "select", 4
    if base.startswith("v_cmp"):
        return "compare", 4
    if re.match(r"v_(min|max|med)3?_", base):
        return "min/max", 4
    if base in ("v_rndne_f32", "v_trunc_f32", "v_floor_f32", "v_ceil_f32", "v_fract_f32") or base.startswith("v_cvt_"):
        return "round/convert", 4
    # SGPR (or vcc / exec as data) source
    srcs = operands.split(",")[1:] if "," in operands else []
    sgpr_src = any(re.search(r"(^|[\s|\-])(s\d+|s\[\d+:\d+\]|vcc|exec)(\b|$)", s.strip()) for s in srcs)
    if base in CHEAP or base in extra_cheap:
        if sgpr_src:
            return "sgpr source", 4
        # three VGPR sources in one register bank (index mod 4)
        if base == "v_fma_f32":
            regs = re.findall(r"v(\d+)", ",".join(srcs))
            if len(regs) == 3 and len({int(r) % 4 for r in regs}) == 1 and len(set(regs)) > 1:
                return "same-bank fma", 4
        return "cheap", 2
    return "int/other (4)", 4


def build_asm(extra_flags):
    out = Path("/tmp/jxlt_asm_budget.s")
    cmd = ["/opt/rocm/bin/hipcc"] + HIPFLAGS + ["-DJXLT_ASM_MARKERS"] + extra_flags + [
        "-S", "--cuda-device-only", "-o", str(out), str(ROOT / "libjxl-tiny_amd/csrc/jxlt_capi_encode.hip")]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out.read_text()


def kernel_lines(asm, kernel):
    lines = asm.splitlines()
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^_ZN8jxlt_dev\d+%s(E|ENS)[^:]*:" % kernel, ln):
            start = i
            break
    if start is None:
        raise SystemExit("kernel %s not found" % kernel)
    for j in range(start, len(lines)):
        if lines[j].strip().startswith((".end_amdhsa_kernel", ".Lfunc_end", ".amdhsa_kernel", ".section")):
            return lines[start:j]
    return lines[start:]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", default="")
    ap.add_argument("--kernel", default="tile12_kernel")
    ap.add_argument("--dump-phase", default=None, help="print the instructions of this phase")
    ap.add_argument("--top", type=int, default=0, help="also list the N most frequent opcodes of every phase")
    ap.add_argument("--cheap", default="", help="comma-separated opcodes to price at 2 cycles in addition (probe results)")
    args = ap.parse_args()
    extra_cheap = set(x for x in args.cheap.split(",") if x)
    asm = build_asm(args.flags.split())
    lines = kernel_lines(asm, args.kernel)
    phase = "prologue"
    order = [phase]
    per = collections.OrderedDict()
    per[phase] = collections.Counter()
    cyc = collections.Counter()
    ops = collections.defaultdict(collections.Counter)
    other = collections.defaultdict(collections.Counter)
    for ln in lines:
        s = ln.strip()
        m = re.match(r";\s*JXLT_PHASE\s+(\S+)", s)
        if m:
            phase = m.group(1)
            if phase not in per:
                per[phase] = collections.Counter()
                order.append(phase)
            continue
        if not s or s.startswith(";") or s.startswith(".") or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        op = parts[0]
        operands = parts[1].split(";")[0] if len(parts) > 1 else ""
        if args.dump_phase == phase:
            print(s)
        if op.startswith("v_"):
            cls, c = classify(op, operands, extra_cheap)
            per[phase][cls] += 1
            per[phase]["VALU"] += 1
            cyc[phase] += c
            ops[phase][strip_suffix(op) + (" [%s]" % cls if cls not in ("cheap",) else "")] += 1
        elif op.startswith("s_"):
            per[phase]["SALU"] += 1
            other[phase][op] += 1
        elif op.startswith("ds_"):
            per[phase]["LDS"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            per[phase]["VMEM"] += 1
    classes = ["cheap", "select", "compare", "min/max", "round/convert", "dpp", "sgpr source", "same-bank fma",
               "int/other (4)", "lane<->scalar", "permlane", "trans"]
    hdr = "%-14s %6s %6s " % ("phase", "VALU", "cycles") + " ".join("%9s" % c[:9] for c in classes) + "   SALU   LDS  VMEM"
    print(hdr)
    tot = collections.Counter()
    tot_c = 0
    for ph in order:
        p = per[ph]
        if not p:
            continue
        print("%-14s %6d %6d " % (ph, p["VALU"], cyc[ph]) + " ".join("%9d" % p[c] for c in classes) +
              " %6d %5d %5d" % (p["SALU"], p["LDS"], p["VMEM"]))
        tot.update(p)
        tot_c += cyc[ph]
        if args.top:
            print("      " + ", ".join("%s x%d" % kv for kv in ops[ph].most_common(args.top)))
    print("%-14s %6d %6d " % ("total", tot["VALU"], tot_c) + " ".join("%9d" % tot[c] for c in classes) +
          " %6d %5d %5d" % (tot["SALU"], tot["LDS"], tot["VMEM"]))
    four = tot["VALU"] - tot["cheap"] - tot["trans"]
    print("4-cycle class: %d of %d static VALU instructions (%.1f %%), %.1f %% of the static issue cycles" %
          (four, tot["VALU"], 100.0 * four / max(1, tot["VALU"]), 100.0 * 4 * four / max(1, tot_c)))
    m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(asm.splitlines()))
    for name in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size"):
        mm = re.search(r"\.name:\s+_ZN8jxlt_dev\d+%sE.*?\.%s:\s+(\d+)" % (args.kernel, name), asm, re.S)
        if mm:
            print("%s: %s" % (name, mm.group(1)), end="   ")
    print()


if __name__ == "__main__":
    main()
