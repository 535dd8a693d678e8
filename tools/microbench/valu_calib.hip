// valu_calib.hip -- how fast does a gfx950 SIMD issue the instruction classes of the traversal step, and does a partly
// empty EXEC mask make a VALU instruction cheaper?  (the "4 cycles per wave64 instruction" model behind
// roofline.ceilings.valu gave fractions > 1; every other ceiling of the bench line is calibrated, this one was not.)
//
// Every kernel runs the same loop: ITER iterations of an UNROLL-long block of ONE instruction class (independent
// destinations, so neither the dependency check nor latency throttles a wave), `waves` waves per SIMD on every SIMD of the
// chip (256 CUs x 4 SIMDs), under a chosen EXEC mask.  Per class and mask it prints
//     cycles_per_inst   = SIMD cycles per wave-instruction      (s_memtime ticks of the slowest wave x 1 / instructions issued by the SIMD's waves)
//     inst_per_simd_clk = its reciprocal (the "valu_issue_peak" of calibration.json)
// and, under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES ...`, the counters can be set against the
// KNOWN instruction count of each launch (printed as `valu_insts`).
//
// usage: valu_calib [waves_per_simd=6] [iters=2000]          -> JSON lines on stdout
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                   \
    do                                                                                             \
    {                                                                                              \
        hipError_t e = (x);                                                                        \
        if (e != hipSuccess)                                                                       \
        {                                                                                          \
            std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                     \
            std::exit(1);                                                                          \
        }                                                                                          \
    } while (0)

constexpr int kUnroll = 64; // instructions per loop iteration (plus ~3 scalar loop instructions)

// one block of 64 instructions over 16 registers, each a read-modify-write chain of its own ("+v": with plain outputs the
// compiler gives every dead result the SAME register and pads the WAW hazard with s_nop): a register is read 16
// instructions after it was written, which no VALU latency reaches at 2+ waves per SIMD
#define REP16(OP)                                                                                  \
    OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define REP64(OP) REP16(OP) REP16(OP) REP16(OP) REP16(OP)

struct Out
{
    unsigned long long cycles; // s_memtime ticks of this wave's loop
    unsigned long long realtime; // s_memrealtime ticks (constant 100 MHz) of the same loop
    float              sink;
};

// The classes.  d = destination registers (float r[16] / pairs), a b c = loop-invariant sources.
enum Class : int
{
    kFma = 0,      // v_fma_f32
    kMul,          // v_mul_f32
    kAddU,         // v_add_u32
    kFmaMix,       // v_fma_mix_f32 (f16 plane x f32 + f32: the half-precision record's plane)
    kPerm,         // v_perm_b32
    kAlignbit,     // v_alignbit_b32
    kPkAdd,        // v_pk_add_f32
    kPkMul,        // v_pk_mul_f32
    kPkFma,        // v_pk_fma_f32
    kMin3,         // v_min3_f32
    kMax3,         // v_max3_f32
    kMaxF,         // v_max_f32
    kCmp,          // v_cmp_lt_f32 -> vcc
    kCmpSgpr,      // v_cmp_lt_f32 -> sgpr pair (VOP3)
    kCndmask,      // v_cndmask_b32
    kBfe,          // v_bfe_u32
    kDpp,          // v_mov_b32 with a DPP row_shr
    kFmaF64,       // v_fma_f64 (kSky)
    kRcp,          // v_rcp_f32 (transcendental pipe)
    kReadlane,     // v_readlane_b32 (VALU -> SGPR)
    kSMov,         // s_mov_b32 (scalar: for the SALU rate next to it)
    kDsRead64,     // ds_read_b64 (stack pop)
    kDsWrite64,    // ds_write_b64 (stack push)
    kCndmaskSgpr,  // v_cndmask_b32 (VOP3) with the mask in an SGPR pair
    kCmpCndmask,   // v_cmp_lt_f32 vcc + v_cndmask_b32 pairs (32 + 32)
    kBfi,          // v_bfi_b32 (select through a lane-mask VGPR: no VCC)
    kBfeI,         // v_bfe_i32 (sign-extending: a 0 / -1 lane mask from one bit)
    kMixNoSel,     // the step mix with its 12 v_cndmask replaced by v_fma_mix
    kMixBfi,       // the step mix with its 12 v_cndmask replaced by v_bfi_b32
    kMixSel24,     // the step mix with 24 v_cndmask (12 fewer v_fma_mix)
    kAddF,  // v_add_f32
    kSubF,  // v_sub_f32
    kMinF,  // v_min_f32
    kMovB,  // v_mov_b32
    kAndB,  // v_and_b32
    kOrB,  // v_or_b32
    kLshl,  // v_lshlrev_b32
    kLshr,  // v_lshrrev_b32
    kSubU,  // v_sub_u32
    kFmac,  // v_fmac_f32
    kMulS,  // v_mul_f32(sgpr)
    kAlignS,  // v_alignbit_b32(sgpr)
    kFmaMixS,  // v_fma_mix_f32(vvv,sgpr_free)
    kCvt,  // v_cvt_f32_u32
    kLshlAdd,  // v_lshl_add_u32
    kMulLo,  // v_mul_lo_u32
    kDivFix,  // v_div_fixup_f32
    kCmpClass,  // v_cmp_class_f32
    kStepMix,      // the half-precision quad step's multiset: 12 alignbit, 24 fma_mix, 4 max3, 4 min3, 8 cmp, 12 cndmask  (64 instructions)
    kNumClasses
};

static const char* kNames[kNumClasses] = {"v_fma_f32", "v_mul_f32", "v_add_u32", "v_fma_mix_f32", "v_perm_b32", "v_alignbit_b32", "v_pk_add_f32", "v_pk_mul_f32",
                                          "v_pk_fma_f32", "v_min3_f32", "v_max3_f32", "v_max_f32", "v_cmp_lt_f32(vcc)", "v_cmp_lt_f32(sgpr)", "v_cndmask_b32", "v_bfe_u32",
                                          "v_mov_b32_dpp", "v_fma_f64", "v_rcp_f32", "v_readlane_b32", "s_mov_b32", "ds_read_b64", "ds_write_b64", "v_cndmask_b32(sgpr)",
                                          "v_cmp+v_cndmask", "v_bfi_b32", "v_bfe_i32", "mix_no_cndmask", "mix_bfi_for_cndmask", "mix_24_cndmask", "v_add_f32", "v_sub_f32", "v_min_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_fmac_f32", "v_mul_f32(sgpr)", "v_alignbit_b32(sgpr)", "v_fma_mix_f32(vvv,sgpr_free)", "v_cvt_f32_u32", "v_lshl_add_u32", "v_mul_lo_u32", "v_div_fixup_f32", "v_cmp_class_f32", "half_quad_step_mix"};

template<int CLASS>
__global__ __launch_bounds__(256) void kIssue(int iters, unsigned long long execMask, Out* out)
{
    extern __shared__ uint2 lds[]; // sized by the host so that exactly `wavesPerSimd` workgroups fit a CU's 160 KB: every SIMD hosts the same number of waves
    float            r[16];
    double           dd[8];
    typedef float    f2 __attribute__((ext_vector_type(2)));
    f2               p[8];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = 1.0f + 0.001f * (threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < 8; ++i) dd[i] = 1.0 + 0.001 * (threadIdx.x + i), p[i] = f2{r[i], r[i + 8]};
    lds[threadIdx.x] = make_uint2(threadIdx.x, 0);
    float    a = 1.0000001f, b = 0.9999999f, c = 1e-9f;
    uint32_t ua = 0x3C003C00u, ub = 0x00050004u, uc = 5u;
    f2       pa = f2{a, b}, pb = f2{b, a};
    double   da = 1.0000001, db = 1e-12;
    uint32_t ldsAddr = threadIdx.x * 8u;
    const float    sb = __builtin_amdgcn_readfirstlane(0x3f7ffffe) == 0 ? 1.0f : __uint_as_float(__builtin_amdgcn_readfirstlane(0x3f7ffffe));
    const uint32_t sa = __builtin_amdgcn_readfirstlane(0x3c003c00);
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(ua), "+v"(ub), "+v"(uc), "+v"(pa), "+v"(pb), "+v"(da), "+v"(db), "+v"(ldsAddr));
    __syncthreads();
    unsigned long long savedExec;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1" : "=&s"(savedExec) : "s"(execMask) : "memory");
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it)
    {
#define R(i) r[(i)]
        if constexpr (CLASS == kFma)
        {
#define OP(i) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(R(i)) : "v"(b), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMul)
        {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(R(i)) : "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kAddU)
        {
#define OP(i) asm volatile("v_add_u32 %0, %1, %0" : "+v"(R(i)) : "v"(ub));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kFmaMix)
        {
#define OP(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(R(i)) : "v"(ua), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kPerm)
        {
#define OP(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(uc), "v"(ub));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kAlignbit)
        {
#define OP(i) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(R(i)) : "v"(uc));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kPkAdd)
        {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(pb));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kPkMul)
        {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[(i) & 7]) : "v"(pb));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kPkFma)
        {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[(i) & 7]) : "v"(pa), "v"(pb));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMin3)
        {
#define OP(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(b), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMax3)
        {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(b), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMaxF)
        {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(R(i)) : "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCmp)
        {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCmpSgpr)
        {
            unsigned long long m;
#define OP(i) asm volatile("v_cmp_lt_f32 %0, %1, %2" : "=s"(m) : "v"(a), "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCndmask)
        {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kBfe)
        {
#define OP(i) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kDpp)
        {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kFmaF64)
        {
#define OP(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(dd[(i) & 7]) : "v"(da), "v"(db));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kRcp)
        {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kReadlane)
        {
            uint32_t s;
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(ua));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kSMov)
        {
            uint32_t s;
#define OP(i) asm volatile("s_mov_b32 %0, 0x1234" : "=s"(s));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kDsRead64)
        {
#define OP(i) asm volatile("ds_read_b64 %0, %1" : "+v"(p[(i) & 7]) : "v"(ldsAddr) : "memory");
            REP64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        else if constexpr (CLASS == kDsWrite64)
        {
#define OP(i) asm volatile("ds_write_b64 %0, %1" : : "v"(ldsAddr), "v"(pa) : "memory");
            REP64(OP)
#undef OP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        else if constexpr (CLASS == kCndmaskSgpr)
        {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(R(i)) : "v"(b), "s"(execMask));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCmpCndmask)
        {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(R(i)) : "v"(a), "v"(b) : "vcc");
            REP16(OP) REP16(OP)
#undef OP
        }
        else if constexpr (CLASS == kBfi)
        {
#define OP(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(R(i)) : "v"(ua), "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kBfeI)
        {
#define OP(i) asm volatile("v_bfe_i32 %0, %0, 3, 29" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kAddF)
        {
#define OP(i) asm volatile("v_add_f32 %0, %1, %0" : "+v"(R(i)) : "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kSubF)
        {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(R(i)) : "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMinF)
        {
#define OP(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(R(i)) : "v"(a));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMovB)
        {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "=v"(R(i)) : "v"(R(((i) + 1) & 15)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kAndB)
        {
#define OP(i) asm volatile("v_and_b32 %0, %1, %0" : "+v"(R(i)) : "v"(ua));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kOrB)
        {
#define OP(i) asm volatile("v_or_b32 %0, %1, %0" : "+v"(R(i)) : "v"(ub));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kLshl)
        {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kLshr)
        {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kSubU)
        {
#define OP(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(R(i)) : "v"(uc));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kFmac)
        {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(R(i)) : "v"(b), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMulS)
        {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(R(i)) : "s"(sb));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kAlignS)
        {
#define OP(i) asm volatile("v_alignbit_b32 %0, %1, %1, %2" : "=v"(R(i)) : "s"(sa), "v"(R(((i) + 1) & 15)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kFmaMixS)
        {
#define OP(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(R(i)) : "v"(R(((i) + 1) & 15)), "v"(a), "v"(c));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCvt)
        {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(R(i)));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kLshlAdd)
        {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(R(i)) : "v"(uc));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kMulLo)
        {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(R(i)) : "v"(uc));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kDivFix)
        {
#define OP(i) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(a), "v"(b));
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kCmpClass)
        {
#define OP(i) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(a), "v"(uc) : "vcc");
            REP64(OP)
#undef OP
        }
        else if constexpr (CLASS == kStepMix || CLASS == kMixNoSel || CLASS == kMixBfi || CLASS == kMixSel24)
        {
            // 12 alignbit + 24 fma_mix + 4 max3 + 4 min3 + 8 cmp + 12 cndmask = 64: the box arithmetic, hit tests and part of the ordering network of one
            // half-precision quad step, interleaved roughly as the compiler emits them
#define AL(i) asm volatile("v_alignbit_b32 %0, %0, %0, %1" : "+v"(R(i)) : "v"(uc));
#define FM(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(R(i)) : "v"(ua), "v"(c));
#define FH(i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(R(i)) : "v"(ua), "v"(c));
#define MX(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(b), "v"(c));
#define MN(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(R(i)) : "v"(b), "v"(c));
#define CP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
#define CMS(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(R(i)) : "v"(b));
#define CMB(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(R(i)) : "v"(ua), "v"(b));
            if constexpr (CLASS == kMixSel24)
            {
                AL(0) AL(1) AL(2) CMS(3) CMS(4) CMS(5) CMS(6) CMS(7) CMS(8) MX(9) MN(10) CP(0) CP(0)
                AL(11) AL(12) AL(13) CMS(14) CMS(15) CMS(0) CMS(1) CMS(2) CMS(3) MX(4) MN(5) CP(0) CP(0)
            }
            else
            {
                AL(0) AL(1) AL(2) FM(3) FH(4) FM(5) FH(6) FM(7) FH(8) MX(9) MN(10) CP(0) CP(0)
                AL(11) AL(12) AL(13) FM(14) FH(15) FM(0) FH(1) FM(2) FH(3) MX(4) MN(5) CP(0) CP(0)
            }
            AL(6) AL(7) AL(8) FM(9) FH(10) FM(11) FH(12) FM(13) FH(14) MX(15) MN(0) CP(0) CP(0)
            AL(1) AL(2) AL(3) FM(4) FH(5) FM(6) FH(7) FM(8) FH(9) MX(10) MN(11) CP(0) CP(0)
            if constexpr (CLASS == kMixNoSel) { FM(12) FH(13) FM(14) FH(15) FM(0) FH(1) FM(2) FH(3) FM(4) FH(5) FM(6) FH(7) }
            else if constexpr (CLASS == kMixBfi) { CMB(12) CMB(13) CMB(14) CMB(15) CMB(0) CMB(1) CMB(2) CMB(3) CMB(4) CMB(5) CMB(6) CMB(7) }
            else { CMS(12) CMS(13) CMS(14) CMS(15) CMS(0) CMS(1) CMS(2) CMS(3) CMS(4) CMS(5) CMS(6) CMS(7) }
#undef AL
#undef FM
#undef FH
#undef MX
#undef MN
#undef CP
#undef CMS
#undef CMB
        }
#undef R
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_mov_b64 exec, %0" : : "s"(savedExec) : "memory");
    float sink = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sink += r[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) sink += static_cast<float>(dd[i]) + p[i].x + p[i].y;
    if ((threadIdx.x & 63) == 0)
    {
        Out o;
        o.cycles = t1 - t0;
        o.realtime = r1 - r0;
        o.sink = sink;
        out[(blockIdx.x * 256 + threadIdx.x) / 64] = o;
    }
}

struct Mask
{
    const char*        name;
    unsigned long long bits;
    int                lanes;
};

template<int CLASS>
void run(int wavesPerSimd, int iters, Out* dOut, std::vector<Out>& hOut, const std::vector<Mask>& masks)
{
    // `wavesPerSimd` waves on each of the 1024 SIMDs: workgroups of 4 waves (one per SIMD of a CU), wavesPerSimd workgroups per CU
    const int    blocks = 256 * wavesPerSimd;
    // W workgroups of (144 KB / W) fit a CU's 160 KB of LDS and a (W + 1)-th does not (W = 1: 64 KB, the largest a workgroup may ask for,
    // and 256 workgroups on 256 CUs: the dispatcher spreads them)
    const size_t ldsBytes = wavesPerSimd == 1 ? 65536u : std::max<size_t>((144u * 1024u / wavesPerSimd) / 1024u * 1024u, 8192u);
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&kIssue<CLASS>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(ldsBytes)));
    for (const Mask& m : masks)
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kIssue<CLASS>, dim3(blocks), dim3(256), ldsBytes, 0, 10, m.bits, dOut); // warm-up
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kIssue<CLASS>, dim3(blocks), dim3(256), ldsBytes, 0, iters, m.bits, dOut);
        CHECK(hipGetLastError());
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(hOut.data(), dOut, sizeof(Out) * blocks * 4, hipMemcpyDeviceToHost));
        unsigned long long maxCycles = 0, sumCycles = 0, sumReal = 0;
        for (int i = 0; i < blocks * 4; ++i) maxCycles = std::max(maxCycles, hOut[i].cycles), sumCycles += hOut[i].cycles, sumReal += hOut[i].realtime;
        const double mhz = sumReal ? static_cast<double>(sumCycles) / static_cast<double>(sumReal) * 100.0 : 0.0; // s_memtime ticks per 10 ns
        const double meanCycles = static_cast<double>(sumCycles) / (blocks * 4);
        const double instPerWave = static_cast<double>(iters) * kUnroll;
        // s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"); the wall time of the launch is
        // printed beside them (ns per SIMD-instruction x the clock = the same figure, if the clock is what rocm-smi says)
        const double cyclesPerInstTick = meanCycles / (instPerWave * wavesPerSimd);
        std::printf("{\"class\": \"%s\", \"exec\": \"%s\", \"active_lanes\": %d, \"waves_per_simd\": %d, \"insts_per_wave\": %.0f, \"valu_insts\": %.0f, "
                    "\"ms\": %.4f, \"wave_ticks_mean\": %.0f, \"wave_ticks_max\": %llu, \"ticks_per_simd_inst\": %.5f, \"ns_per_simd_inst\": %.4f, \"memtime_mhz\": %.1f, \"wave_ns_per_simd_inst\": %.4f}\n",
                    kNames[CLASS], m.name, m.lanes, wavesPerSimd, instPerWave, instPerWave * blocks * 4, ms, meanCycles, maxCycles, cyclesPerInstTick,
                    ms * 1e6 / (instPerWave * wavesPerSimd), mhz, static_cast<double>(sumReal) / (blocks * 4) * 10.0 / (instPerWave * wavesPerSimd));
        std::fflush(stdout);
        CHECK(hipEventDestroy(e0));
        CHECK(hipEventDestroy(e1));
    }
}

template<int... CS>
void runAll(int w, int it, Out* d, std::vector<Out>& h, const std::vector<Mask>& full, const std::vector<Mask>& all, std::integer_sequence<int, CS...>)
{
    (run<CS>(w, it, d, h, (CS == kFma || CS == kFmaMix || CS == kPkFma || CS == kStepMix || CS == kCndmask || CS == kFmaF64 || CS == kDsRead64 || CS == kBfi) ? all : full), ...);
}

int main(int argc, char** argv)
{
    const int wavesPerSimd = argc > 1 ? std::atoi(argv[1]) : 6;
    const int iters = argc > 2 ? std::atoi(argv[2]) : 2000;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"waves_per_simd\": %d, \"iters\": %d, \"unroll\": %d}\n", prop.gcnArchName, prop.multiProcessorCount,
                prop.clockRate, wavesPerSimd, iters, kUnroll);
    Out* dOut = nullptr;
    CHECK(hipMalloc(&dOut, sizeof(Out) * 256 * 8 * 4));
    std::vector<Out> hOut(256 * 8 * 4);
    const std::vector<Mask> full = {{"full", ~0ull, 64}};
    const std::vector<Mask> all = {{"full", ~0ull, 64},
                                   {"low32", 0x00000000FFFFFFFFull, 32},
                                   {"high32", 0xFFFFFFFF00000000ull, 32},
                                   {"low16", 0x000000000000FFFFull, 16},
                                   {"lanes16_47", 0x0000FFFFFFFF0000ull, 32},
                                   {"even_lanes", 0x5555555555555555ull, 32},
                                   {"every_4th", 0x1111111111111111ull, 16},
                                   {"one_lane", 0x0000000000000001ull, 1},
                                   {"rows_0_2", 0x0000FFFF0000FFFFull, 32}};
    runAll(wavesPerSimd, iters, dOut, hOut, full, all, std::make_integer_sequence<int, kNumClasses>{});
    CHECK(hipFree(dOut));
    return 0;
}
