"""Static issue-cost estimate of a kernel in the shipped code object, with the per-instruction costs measured by
tools/issue_probe.hip for a lone wave per SIMD (cycles; profiles/r03_issue_probe.txt, loop overhead taken out):
4 for an instruction with a 4-byte encoding (VOP1 / VOP2 / SOP*, s_nop 0, s_waitcnt), 5 for one with an 8-byte encoding
(VOP3, DPP, a 32-bit literal, DS, global), v_mfma_f64_16x16x4 64, v_rcp_f64 / v_rsq_f64 16, VOP2 v_cndmask_b32
(implicit VCC) 19, s_nop N 4 (N + 1), a taken branch ~16 more.
    python tools/isa_cost.py [substring of the kernel name]
Prints the total and the share of each mnemonic (every instruction counted once: a loop body is not multiplied)."""
import collections, re, sys, tempfile
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import codeobj as C

_VOP3_ONLY = ("v_fma_f64", "v_mul_f64", "v_add_f64", "v_max_f64", "v_min_f64", "v_mad_", "v_lshl_add", "v_lshl_or", "v_add3",
              "v_readlane", "v_writelane", "v_accvgpr", "v_and_or", "v_bfe", "v_perm", "v_alignbit", "v_mul_lo", "v_mul_hi",
              "v_cmp_class", "v_div_", "v_ldexp_f64", "v_trig", "v_mfma")
_LIT = re.compile(r"(?<![\w\[])(0x[0-9a-f]+|-?[0-9]{3,})(?![\w:\]])")

def cost(mn, ops):
    if mn.startswith("v_mfma_f64_16x16x4"): return 64.0
    if mn.startswith("v_rcp_f64") or mn.startswith("v_rsq_f64") or mn.startswith("v_sqrt_f64"): return 16.0
    if mn == "v_cndmask_b32_e32": return 19.0
    if mn == "s_nop": return 4.0 * (int(ops.strip(), 0) + 1)
    wide = (mn.endswith("_e64") or "_dpp" in mn or "row_newbcast" in ops or "quad_perm" in ops or "row_" in ops or "sdwa" in mn
            or mn.startswith(_VOP3_ONLY) or mn.startswith(("ds_", "global_", "flat_", "buffer_", "scratch_", "s_load"))
            or (mn.startswith(("s_mov_b32", "s_and_b32", "s_or_b32", "s_mul_i32", "s_add_u32", "s_add_i32", "s_cmp", "v_mov_b32", "v_and_b32",
                               "v_or_b32", "v_add_u32", "v_cmp")) and _LIT.search(ops) is not None))
    return 5.0 if wide else 4.0

def main():
    key = sys.argv[1] if len(sys.argv) > 1 else "fused_kernelINS0_5ShapeILi60ELi38ELi12ELi38EEEdLb0"
    out = tempfile.mkdtemp()
    for p in C.extract(outdir=out):
        for name, body in C.disassemble(p).items():
            if key not in name: continue
            tot = collections.Counter(); cnt = collections.Counter()
            for mn, ops in body:
                k = mn + ("(dpp)" if "row_newbcast" in ops or "quad_perm" in ops else "")
                tot[k] += cost(mn, ops); cnt[k] += 1
            total = sum(tot.values())
            print(f"{name}\n  {len(body)} instructions, {total:.0f} cycles if each is issued once")
            for k, v in tot.most_common(28):
                print(f"  {v:9.0f}  {100 * v / total:5.1f} %  {cnt[k]:5d}  {k}")

if __name__ == "__main__":
    main()
