"""Static issue-cost estimate of a kernel in the shipped code object, with the per-instruction costs measured by
tools/issue_probe.hip for a lone wave per SIMD (cycles): any instruction ~5.8 (VALU, SALU, s_waitcnt, LDS issue),
v_mfma_f64_16x16x4 64, v_rcp_f64 / v_rsq_f64 17.8, VOP2 v_cndmask_b32 (implicit VCC) 19, s_nop N 5.8 + 4 N.
    python tools/isa_cost.py [substring of the kernel name]
Prints the total, the share of each mnemonic, and -- for the loops it finds (backward branches) -- the cost per trip."""
import collections, sys, tempfile
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tests import codeobj as C

def cost(mn, ops):
    if mn.startswith("v_mfma_f64_16x16x4"): return 64.0
    if mn.startswith("v_rcp_f64") or mn.startswith("v_rsq_f64") or mn.startswith("v_sqrt_f64"): return 17.8
    if mn == "v_cndmask_b32_e32": return 19.0
    if mn == "s_nop": return 5.8 + 4.0 * int(ops.strip(), 0)
    return 5.8

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
