"""Static VALU opcode mix of the heavy kernels, priced with the measured per-class issue costs of tools/valu_ceiling.hip.

    python tools/valu_mix.py [gpurun_out/r06_valu_ceiling.json]  ->  JSON on stdout

Compiles homan_amd/csrc/*.hip to gfx950 assembly (device side only), histograms the VALU opcodes of k_raster_fwd and
k_bwd_sweep<true> (STATIC counts: every instruction of the kernel body once - a proxy for the dynamic mix, which the PMC counters
do not break down by opcode), maps every opcode onto one of the measured classes and prints the mix-weighted cycles per wave64
instruction at 1 / 2 / 4 / 8 waves per SIMD (from the wall-clock figures at the nominal 2.4 GHz: the in-kernel s_memtime counts
do not overlap fully across waves at high occupancy).  Opcodes without a measured class are listed and priced as `v_add_u32`."""
import collections
import glob
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
KERNELS = {"k_raster_fwd": "_Z12k_raster_fwd", "k_bwd_sweep<W32>": "_Z11k_bwd_sweepILb1EE", "k_bwd_lines<W32>": "_Z11k_bwd_linesILb1EE"}
# opcode prefix -> measured class of valu_ceiling.hip
CLASS = [("v_mov_b32", "v_mov_b32"), ("v_mov_b64", "v_mov_b32"), ("v_and_b32", "v_and_b32"), ("v_or_b32", "v_or_b32"),
         ("v_xor_b32", "v_xor_b32"), ("v_not_b32", "v_xor_b32"), ("v_lshlrev_b32", "v_lshlrev_b32"), ("v_lshrrev_b32", "v_lshlrev_b32"),
         ("v_ashrrev_i32", "v_lshlrev_b32"), ("v_lshlrev_b64", "v_lshlrev_b32"), ("v_lshrrev_b64", "v_lshlrev_b32"),
         ("v_sub_u32", "v_sub_u32"), ("v_subrev_u32", "v_sub_u32"), ("v_sub_co", "v_sub_u32"), ("v_add_co", "v_add_u32"),
         ("v_addc_co", "v_add_u32"), ("v_subb_co", "v_sub_u32"), ("v_max_i32", "v_max_i32"), ("v_min_i32", "v_max_i32"),
         ("v_max_u32", "v_max_i32"), ("v_min_u32", "v_max_i32"), ("v_bfe", "v_bfe_u32"), ("v_bitop3", "v_or3_b32"),
         ("v_bcnt", "v_bfe_u32"), ("v_mbcnt", "v_bfe_u32"), ("v_ffbl", "v_bfe_u32"), ("v_ffbh", "v_bfe_u32"), ("v_lshl_add_u64", "v_lshl_add_u32"),
         ("v_sub_f32", "v_sub_f32"), ("v_subrev_f32", "v_sub_f32"),
         ("v_pk_", "v_pk_fma_f32"), ("v_fma_f64", "v_fma_f64"), ("v_add_f64", "v_add_f64"), ("v_mul_f64", "v_fma_f64"),
         ("v_cvt_f64", "v_add_f64"), ("v_cvt_f32_f64", "v_add_f64"), ("v_ldexp_f64", "v_add_f64"), ("v_cmp_", "v_cmp_lt_f32+v_cndmask_b32"),
         ("v_cndmask", "v_cmp_lt_f32+v_cndmask_b32"), ("v_rcp", "v_rcp_f32"), ("v_rsq", "v_rcp_f32"), ("v_sqrt", "v_rcp_f32"),
         ("v_exp", "v_rcp_f32"), ("v_log", "v_rcp_f32"), ("v_mul_lo", "v_mul_lo_u32"), ("v_mul_hi", "v_mul_lo_u32"),
         ("v_mad_u64", "v_mul_lo_u32"), ("v_mad_i64", "v_mul_lo_u32"), ("v_mad_u32_u24", "v_mad_u32_u24"), ("v_mad_i32_i24", "v_mad_u32_u24"),
         ("v_mul_u32_u24", "v_mad_u32_u24"), ("v_mul_i32_i24", "v_mad_u32_u24"), ("v_alignbit", "v_alignbit_b32"),
         ("v_or3", "v_or3_b32"), ("v_and_or", "v_or3_b32"), ("v_lshl_or", "v_or3_b32"), ("v_lshl_add", "v_lshl_add_u32"),
         ("v_add_lshl", "v_lshl_add_u32"), ("v_add3", "v_or3_b32"), ("v_bfe", "v_or3_b32"), ("v_bfi", "v_or3_b32"), ("v_perm", "v_or3_b32"),
         ("v_fma_f32", "v_fma_f32"), ("v_fmac_f32", "v_fma_f32"), ("v_div_fmas", "v_fma_f32"), ("v_div_fixup", "v_fma_f32"),
         ("v_div_scale", "v_fma_f32"), ("v_mul_f32", "v_mul_f32"), ("v_add_f32", "v_add_f32"), ("v_sub_f32", "v_add_f32"),
         ("v_subrev_f32", "v_add_f32"), ("v_min_", "v_min_f32/v_max_f32"), ("v_max_", "v_min_f32/v_max_f32"), ("v_med3", "v_min_f32/v_max_f32"),
         ("v_cvt_", "v_cvt_f32_i32"), ("v_floor", "v_cvt_f32_i32"), ("v_ceil", "v_cvt_f32_i32"), ("v_trunc", "v_cvt_f32_i32"),
         ("v_rndne", "v_cvt_f32_i32"), ("v_fract", "v_cvt_f32_i32")]
NOT_VALU = ("v_readlane", "v_readfirstlane", "v_writelane", "v_accvgpr", "v_nop")


def asm_of(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-S",
                        "--cuda-device-only", "-I", os.path.join(ROOT, "homan_amd", "csrc"), "-o", out, src], check=True,
                       stderr=subprocess.DEVNULL)
        return open(out).read()


def kernel_bodies(text):
    bodies = {}
    cur = None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            bodies[cur] = []
            continue
        if cur and line.strip().startswith("s_endpgm"):
            cur = None
        elif cur:
            bodies[cur].append(line)
    return bodies


def classify(op):
    for pre, cls in CLASS:
        if op.startswith(pre):
            return cls
    return None


def main():
    ceil = json.load(open(sys.argv[1])) if len(sys.argv) > 1 else None
    out = {}
    for src in sorted(glob.glob(os.path.join(ROOT, "homan_amd", "csrc", "*.hip"))):
        bodies = kernel_bodies(asm_of(src))
        for name, sym in KERNELS.items():
            for k, lines in bodies.items():
                if not k.startswith(sym):
                    continue
                hist, dpp = collections.Counter(), 0
                for ln in lines:
                    t = ln.strip().split()
                    if not t or not t[0].startswith("v_") or t[0].startswith(NOT_VALU):
                        continue
                    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", t[0])
                    hist[op] += 1
                    dpp += t[0].endswith("_dpp") or "row_" in ln or "quad_perm" in ln
                total = sum(hist.values())
                rec = dict(valu_instructions_static=total, dpp=dpp, top=dict(hist.most_common(25)))
                if ceil:
                    unknown = collections.Counter()
                    for w in ("w1", "w2", "w4", "w8"):
                        acc = 0.0
                        for op, n in hist.items():
                            cls = classify(op)
                            if cls is None:
                                unknown[op] += n if w == "w1" else 0
                                cls = "v_add_u32"
                            acc += n * ceil["ops"][cls][w]["wall_cycles_per_instr_at_2.4GHz"]
                        rec["mix_cycles_per_instr_" + w] = round(acc / max(total, 1), 3)
                    rec["unclassified_priced_as_v_add_u32"] = dict(unknown.most_common(12))
                out[name] = rec
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
