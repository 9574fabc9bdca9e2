// issue_probe.hip -- what one wave per SIMD pays, in shader-clock ticks, for the instruction mixes of the fused
// kernel (fccqp_fused_impl.h): DPP-fed and plain fp64 operations, 32-bit selects, moves between the two register
// files, fp64 MFMAs alone / chained / with VALU work in their shadow.  One wave per SIMD (1024 workgroups of 64),
// fixed registers, every sequence inside one asm statement.
//   hipcc --offload-arch=gfx950 -O3 -o issue_probe issue_probe.hip && ./issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define R2(a) "v[" #a ":" #a "+1]"
#define DPP(acc, m) "v_fmac_f64_dpp " R2(acc) ", v[120:121], " R2(m) " row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define FMAC(acc, m) "v_fmac_f64 " R2(acc) ", v[120:121], " R2(m) "\n\t"
#define FMA3(acc, m) "v_fma_f64 " R2(acc) ", v[120:121], " R2(m) ", v[124:125]\n\t"
#define FMAD(acc, m) "v_fma_f64 " R2(acc) ", v[120:121], " R2(m) ", " R2(acc) "\n\t"
#define MUL(acc, m) "v_mul_f64 " R2(acc) ", v[120:121], " R2(m) "\n\t"
#define ADD(acc, m) "v_add_f64 " R2(acc) ", v[120:121], " R2(m) "\n\t"
#define CNDS(acc, m) "v_cndmask_b32_e64 v" #acc ", v120, v" #m ", s[22:23]\n\t"
#define CNDV(acc, m) "v_cndmask_b32_e32 v" #acc ", v120, v" #m ", vcc\n\t"
#define MOV(acc, m) "v_mov_b32 v" #acc ", v" #m "\n\t"
#define MOV64(acc, m) "v_mov_b64 " R2(acc) ", " R2(m) "\n\t"
#define MOVDPP(acc, m) "v_mov_b64_dpp " R2(acc) ", " R2(m) " row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define ACCR(acc, m) "v_accvgpr_read_b32 v" #acc ", a" #m "\n\t"
#define ACCW(acc, m) "v_accvgpr_write_b32 a" #m ", v" #acc "\n\t"
#define RCP(acc, m) "v_rcp_f64 " R2(acc) ", " R2(m) "\n\t"
#define RDL(acc, m) "v_readlane_b32 s24, v" #m ", 3\n\t"
#define SMOV(acc, m) "s_mov_b32 s24, 0x" #m "\n\t"
#define CNDV2(acc, m) "v_cndmask_b32_e32 v" #acc ", v" #m ", v" #acc ", vcc\n\t"
#define CNDVE64(acc, m) "v_cndmask_b32_e64 v" #acc ", v120, v" #m ", vcc\n\t"
#define CNDS0(acc, m) "v_cndmask_b32_e64 v" #acc ", 0, v" #m ", s[22:23]\n\t"
#define CMPV_CND2(acc, m) "v_cmp_gt_f64_e32 vcc, " R2(acc) ", " R2(m) "\n\tv_cndmask_b32_e32 v108, 0, v" #m ", vcc\n\tv_cndmask_b32_e32 v109, 0, v" #m ", vcc\n\t"
#define CMPS_CND2(acc, m) "v_cmp_gt_f64_e64 s[24:25], " R2(acc) ", " R2(m) "\n\tv_cndmask_b32_e64 v108, 0, v" #m ", s[24:25]\n\tv_cndmask_b32_e64 v109, 0, v" #m ", s[24:25]\n\t"
#define NOP0(acc, m) "s_nop 0\n\t"
#define NOP1(acc, m) "s_nop 1\n\t"
#define NOP7(acc, m) "s_nop 7\n\t"
#define RSQ(acc, m) "v_rsq_f64 " R2(acc) ", " R2(m) "\n\t"
#define MAXF(acc, m) "v_max_f64 " R2(acc) ", v[120:121], " R2(m) "\n\t"
#define XOR(acc, m) "v_xor_b32 v" #acc ", v120, v" #m "\n\t"
#define MOVDPP32(acc, m) "v_mov_b32_dpp v" #acc ", v" #m " row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
#define DSR(acc, m) "ds_read_b64 " R2(acc) ", v126\n\t"
#define DSW(acc, m) "ds_write_b64 v126, " R2(m) "\n\t"
#define WAITL(acc, m) "s_waitcnt lgkmcnt(0)\n\t"
#define CMP(acc, m) "v_cmp_gt_f64_e64 s[24:25], " R2(acc) ", " R2(m) "\n\t"
// 16 instances on four accumulators (100, 102, 104, 106), sixteen distinct second operands
#define X16(OP) OP(100, 128) OP(102, 130) OP(104, 132) OP(106, 134) OP(100, 136) OP(102, 138) OP(104, 140) OP(106, 142) \
                OP(100, 144) OP(102, 146) OP(104, 148) OP(106, 150) OP(100, 152) OP(102, 154) OP(104, 156) OP(106, 158)
#define X8(OP) OP(100, 128) OP(102, 130) OP(104, 132) OP(106, 134) OP(100, 136) OP(102, 138) OP(104, 140) OP(106, 142)
// one accumulator: a dependent chain
#define C16(OP) OP(100, 128) OP(100, 130) OP(100, 132) OP(100, 134) OP(100, 136) OP(100, 138) OP(100, 140) OP(100, 142) \
                OP(100, 144) OP(100, 146) OP(100, 148) OP(100, 150) OP(100, 152) OP(100, 154) OP(100, 156) OP(100, 158)
// alternate two kinds
#define ALT16(A, B) A(100, 128) B(108, 130) A(102, 132) B(110, 134) A(104, 136) B(112, 138) A(106, 140) B(114, 142) \
                    A(100, 144) B(108, 146) A(102, 148) B(110, 150) A(104, 152) B(112, 154) A(106, 156) B(114, 158)
// groups of four of each
#define GRP16(A, B) A(100, 128) A(102, 132) A(104, 136) A(106, 140) B(108, 130) B(110, 134) B(112, 138) B(114, 142) \
                    A(100, 144) A(102, 148) A(104, 152) A(106, 156) B(108, 146) B(110, 150) B(112, 154) B(114, 158)
#define MFMA(d) "v_mfma_f64_16x16x4_f64 v[" #d ":" #d "+7], v[120:121], v[122:123], v[" #d ":" #d "+7]\n\t"
#define MFMA_A(d) "v_mfma_f64_16x16x4_f64 a[" #d ":" #d "+7], v[120:121], v[122:123], a[" #d ":" #d "+7]\n\t"
#define DPP4 DPP(100, 128) DPP(102, 130) DPP(104, 132) DPP(106, 134)
#define DPP12 DPP4 DPP4 DPP4
#define CLOB "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", \
             "v114", "v115", "s24", "s25", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", \
             "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", \
             "v185", "v186", "v187", "v188", "v189", "v190", "v191", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", \
             "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", \
             "a26", "a27", "a28", "a29", "a30", "a31"

#define KERNEL(NAME, SEQ)                                                                         \
  __global__ __launch_bounds__(64) void NAME(long long *cyc, int iters) {                         \
    asm volatile("v_mov_b32 v120, 0\n\tv_mov_b32 v121, 0x3ff00000\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0x3ff00000\n\t" \
                 "v_mov_b32 v126, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0x3ff00000\n\ts_mov_b64 s[22:23], 0x5555\n\t" ::: "v120", "v121", "v122", "v123", "v124", "v125", "v126", "s22", "s23"); \
    const long long t0 = clock64();                                                               \
    for (int it = 0; it < iters; ++it) asm volatile(SEQ ::: CLOB, "vcc");                         \
    const long long t1 = clock64();                                                               \
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                              \
  }

KERNEL(k_dpp, X16(DPP))
KERNEL(k_dpp_chain, C16(DPP))
KERNEL(k_fmac, X16(FMAC))
KERNEL(k_fmac_chain, C16(FMAC))
KERNEL(k_fma3, X16(FMA3))
KERNEL(k_fmad, X16(FMAD))
KERNEL(k_mul, X16(MUL))
KERNEL(k_add, X16(ADD))
KERNEL(k_cnds, X16(CNDS))
KERNEL(k_cndv, X16(CNDV))
KERNEL(k_mov, X16(MOV))
KERNEL(k_mov64, X16(MOV64))
KERNEL(k_movdpp, X16(MOVDPP))
KERNEL(k_accr, X16(ACCR))
KERNEL(k_accw, X16(ACCW))
KERNEL(k_rcp, X16(RCP))
KERNEL(k_rdl, X16(RDL))
KERNEL(k_smov, X16(SMOV))
KERNEL(k_cmp, X16(CMP))
KERNEL(k_alt_dpp_fmac, ALT16(DPP, FMAC))
KERNEL(k_grp_dpp_fmac, GRP16(DPP, FMAC))
KERNEL(k_alt_dpp_cnd, ALT16(DPP, CNDS))
KERNEL(k_alt_dpp_smov, ALT16(DPP, SMOV))
KERNEL(k_alt_fmac_cnd, ALT16(FMAC, CNDS))
KERNEL(k_alt_fma3_mul, ALT16(FMA3, MUL))
KERNEL(k_mfma_indep, MFMA(160) MFMA(168) MFMA(176) MFMA(184) MFMA(160) MFMA(168) MFMA(176) MFMA(184))
KERNEL(k_mfma_chain, MFMA(160) MFMA(160) MFMA(160) MFMA(160) MFMA(160) MFMA(160) MFMA(160) MFMA(160))
KERNEL(k_mfma_agpr, MFMA_A(0) MFMA_A(8) MFMA_A(16) MFMA_A(24) MFMA_A(0) MFMA_A(8) MFMA_A(16) MFMA_A(24))
KERNEL(k_mfma_dpp4, MFMA(160) DPP4 MFMA(168) DPP4 MFMA(176) DPP4 MFMA(184) DPP4 MFMA(160) DPP4 MFMA(168) DPP4 MFMA(176) DPP4 MFMA(184) DPP4)
KERNEL(k_mfma_dpp12, MFMA(160) DPP12 MFMA(168) DPP12 MFMA(176) DPP12 MFMA(184) DPP12 MFMA(160) DPP12 MFMA(168) DPP12 MFMA(176) DPP12 MFMA(184) DPP12)
KERNEL(k_dpp12x8, DPP12 DPP12 DPP12 DPP12 DPP12 DPP12 DPP12 DPP12)
#define CND4 CNDS(100, 128) CNDS(102, 130) CNDS(104, 132) CNDS(106, 134)
#define CND12 CND4 CND4 CND4
KERNEL(k_mfma_cnd12, MFMA(160) CND12 MFMA(168) CND12 MFMA(176) CND12 MFMA(184) CND12 MFMA(160) CND12 MFMA(168) CND12 MFMA(176) CND12 MFMA(184) CND12)

KERNEL(k_cndv2, X16(CNDV2))
KERNEL(k_cndve64, X16(CNDVE64))
KERNEL(k_cnds0, X16(CNDS0))
KERNEL(k_cmpv_cnd2, X8(CMPV_CND2))
KERNEL(k_cmps_cnd2, X8(CMPS_CND2))
KERNEL(k_nop0, X16(NOP0))
KERNEL(k_nop1, X16(NOP1))
KERNEL(k_nop7, X16(NOP7))
KERNEL(k_rsq, X16(RSQ))
KERNEL(k_fmav, "v_fma_f64 v[100:101], v[128:129], v[130:131], v[132:133]\n\t"
    "v_fma_f64 v[102:103], v[134:135], v[136:137], v[138:139]\n\t"
    "v_fma_f64 v[104:105], v[140:141], v[142:143], v[144:145]\n\t"
    "v_fma_f64 v[106:107], v[146:147], v[148:149], v[150:151]\n\t"
    "v_fma_f64 v[100:101], v[152:153], v[154:155], v[156:157]\n\t"
    "v_fma_f64 v[102:103], v[158:159], v[160:161], v[162:163]\n\t"
    "v_fma_f64 v[104:105], v[164:165], v[166:167], v[168:169]\n\t"
    "v_fma_f64 v[106:107], v[170:171], v[172:173], v[174:175]\n\t"
    )
KERNEL(k_max, X16(MAXF))
KERNEL(k_xor, X16(XOR))
KERNEL(k_movdpp32, X16(MOVDPP32))
KERNEL(k_dsr, X16(DSR) "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_dsw, X16(DSW) "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_dsr1, DSR(100, 128) "s_waitcnt lgkmcnt(0)\n\t")
KERNEL(k_waitl, X16(WAITL))
#define X64(OP) X16(OP) X16(OP) X16(OP) X16(OP)
KERNEL(k_dpp64, X64(DPP))
KERNEL(k_dpp128, X64(DPP) X64(DPP))
KERNEL(k_cnds64, X64(CNDS))
KERNEL(k_smov64, X64(SMOV))
KERNEL(k_nop64, X64(NOP0))
KERNEL(k_cndv64, X64(CNDV))
KERNEL(k_rcp64, X64(RCP))
typedef void (*kern_t)(long long *, int);
void run(const char *name, kern_t k, long long *cyc, int per_trip) {
  const int blocks = 1024, iters = 2000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, cyc, iters);
    (void)hipDeviceSynchronize();
  }
  static long long h[1024];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < blocks; ++i) s += h[i];
  printf("%-58s %7.2f ticks per trip  %6.2f per instruction (%d)\n", name, s / blocks / iters, s / blocks / iters / per_trip, per_trip);
  fflush(stdout);
}
int main() {
  long long *cyc;
  (void)hipMalloc(&cyc, 1024 * 8);
  run("v_fmac_f64_dpp x16, four accumulators", k_dpp, cyc, 16);
  run("v_fmac_f64_dpp x16, ONE accumulator (chain)", k_dpp_chain, cyc, 16);
  run("v_fmac_f64 x16", k_fmac, cyc, 16);
  run("v_fmac_f64 x16 chain", k_fmac_chain, cyc, 16);
  run("v_fma_f64 x16, three distinct sources", k_fma3, cyc, 16);
  run("v_fma_f64 x16, srcC = dst", k_fmad, cyc, 16);
  run("v_mul_f64 x16", k_mul, cyc, 16);
  run("v_add_f64 x16", k_add, cyc, 16);
  run("v_cndmask_b32_e64 (SGPR mask) x16", k_cnds, cyc, 16);
  run("v_cndmask_b32_e32 (vcc) x16", k_cndv, cyc, 16);
  run("v_mov_b32 x16", k_mov, cyc, 16);
  run("v_mov_b64 x16", k_mov64, cyc, 16);
  run("v_mov_b64_dpp x16", k_movdpp, cyc, 16);
  run("v_accvgpr_read_b32 x16", k_accr, cyc, 16);
  run("v_accvgpr_write_b32 x16", k_accw, cyc, 16);
  run("v_rcp_f64 x16", k_rcp, cyc, 16);
  run("v_readlane_b32 x16", k_rdl, cyc, 16);
  run("s_mov_b32 literal x16", k_smov, cyc, 16);
  run("v_cmp_gt_f64_e64 x16", k_cmp, cyc, 16);
  run("alternating dpp fmac / plain fmac (8 + 8)", k_alt_dpp_fmac, cyc, 16);
  run("groups of four dpp fmac / plain fmac (8 + 8)", k_grp_dpp_fmac, cyc, 16);
  run("alternating dpp fmac / cndmask (8 + 8)", k_alt_dpp_cnd, cyc, 16);
  run("alternating dpp fmac / s_mov (8 + 8)", k_alt_dpp_smov, cyc, 16);
  run("alternating plain fmac / cndmask (8 + 8)", k_alt_fmac_cnd, cyc, 16);
  run("alternating v_fma_f64 / v_mul_f64 (8 + 8)", k_alt_fma3_mul, cyc, 16);
  run("mfma f64 16x16x4 x8, four accumulators", k_mfma_indep, cyc, 8);
  run("mfma f64 16x16x4 x8, ONE accumulator (chain)", k_mfma_chain, cyc, 8);
  run("mfma f64 16x16x4 x8, accumulators in AGPRs", k_mfma_agpr, cyc, 8);
  run("8 x (mfma + 4 dpp fmac)", k_mfma_dpp4, cyc, 8);
  run("8 x (mfma + 12 dpp fmac)", k_mfma_dpp12, cyc, 8);
  run("8 x (12 dpp fmac) alone", k_dpp12x8, cyc, 8);
  run("8 x (mfma + 12 cndmask)", k_mfma_cnd12, cyc, 8);
  run("v_cndmask_b32_e32 vcc, distinct sources x16", k_cndv2, cyc, 16);
  run("v_cndmask_b32_e64 with vcc as the mask x16", k_cndve64, cyc, 16);
  run("v_cndmask_b32_e64 0, v, s[] x16", k_cnds0, cyc, 16);
  run("8 x (v_cmp_f64 -> vcc, 2 cndmask vcc)", k_cmpv_cnd2, cyc, 24);
  run("8 x (v_cmp_f64 -> s[], 2 cndmask s[])", k_cmps_cnd2, cyc, 24);
  run("s_nop 0 x16", k_nop0, cyc, 16);
  run("s_nop 1 x16", k_nop1, cyc, 16);
  run("s_nop 7 x16", k_nop7, cyc, 16);
  run("v_rsq_f64 x16", k_rsq, cyc, 16);
  run("v_fma_f64 x8, all sources varying", k_fmav, cyc, 8);
  run("v_max_f64 x16", k_max, cyc, 16);
  run("v_xor_b32 x16", k_xor, cyc, 16);
  run("v_mov_b32_dpp x16", k_movdpp32, cyc, 16);
  run("ds_read_b64 x16 + wait", k_dsr, cyc, 16);
  run("ds_write_b64 x16 + wait", k_dsw, cyc, 16);
  run("ds_read_b64 x1 + wait (round trip)", k_dsr1, cyc, 1);
  run("s_waitcnt lgkmcnt(0) (nothing pending) x16", k_waitl, cyc, 16);
  // the same instructions in longer loop bodies: (ticks per trip) = (instructions) x (cost) + (loop overhead)
  run("v_fmac_f64_dpp x64", k_dpp64, cyc, 64);
  run("v_fmac_f64_dpp x128", k_dpp128, cyc, 128);
  run("v_cndmask_b32_e64 x64", k_cnds64, cyc, 64);
  run("s_mov_b32 x64", k_smov64, cyc, 64);
  run("s_nop 0 x64", k_nop64, cyc, 64);
  run("v_cndmask_b32_e32 (vcc) x64", k_cndv64, cyc, 64);
  run("v_rcp_f64 x64", k_rcp64, cyc, 64);
  return 0;
}
