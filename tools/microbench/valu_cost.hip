// Issue cost of single vector instructions on one SIMD of gfx950: one wave, a loop of 32 independent copies of the
// instruction, timed with the shader clock.   hipcc --offload-arch=gfx950 -O2 valu_cost.hip -o valu_cost && ./valu_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))

#define KERNEL(name, body, decl, clob)                                                         \
    __global__ void name(long long* out, int iters) {                                          \
        decl;                                                                                  \
        long long t0 = clock64();                                                              \
        for (int i = 0; i < iters; ++i) { asm volatile(REP32(body) ::: clob); }                \
        long long t1 = clock64();                                                              \
        if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*)out, (unsigned long long)(t1 - t0)); /* the slowest wave: the oldest wave is served first */                                       \
    }

// registers v[10:41] are scratch operands: declared clobbered
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","vcc","scc","s20","s21","s22","s23"

KERNEL(k_fma_f32, "v_fma_f32 v10, v12, v13, v14\n", , CLOB)
KERNEL(k_pk_fma_f32, "v_pk_fma_f32 v[10:11], v[12:13], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_pk_mul_f32, "v_pk_mul_f32 v[10:11], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_pk_add_f32, "v_pk_add_f32 v[10:11], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_fma_f64, "v_fma_f64 v[10:11], v[12:13], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_mul_f64, "v_mul_f64 v[10:11], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_add_f64, "v_add_f64 v[10:11], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_max_f64, "v_max_f64 v[10:11], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_rcp_f64, "v_rcp_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_rsq_f64, "v_rsq_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_sqrt_f64, "v_sqrt_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_div_scale_f64, "v_div_scale_f64 v[10:11], vcc, v[12:13], v[14:15], v[12:13]\n", , CLOB)
KERNEL(k_div_fmas_f64, "v_div_fmas_f64 v[10:11], v[12:13], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_div_fixup_f64, "v_div_fixup_f64 v[10:11], v[12:13], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_cmp_f64, "v_cmp_lt_f64 vcc, v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_cmp_f64_sgpr, "v_cmp_lt_f64 s[20:21], v[12:13], v[14:15]\n", , CLOB)
KERNEL(k_cndmask, "v_cndmask_b32 v10, v12, v13, vcc\n", , CLOB)
KERNEL(k_cvt_f32_f64, "v_cvt_f32_f64 v10, v[12:13]\n", , CLOB)
KERNEL(k_cvt_f64_f32, "v_cvt_f64_f32 v[10:11], v12\n", , CLOB)
KERNEL(k_cvt_f64_u32, "v_cvt_f64_u32 v[10:11], v12\n", , CLOB)
KERNEL(k_ldexp_f64, "v_ldexp_f64 v[10:11], v[12:13], v14\n", , CLOB)
KERNEL(k_frexp_mant_f64, "v_frexp_mant_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_mad_u64_u32, "v_mad_u64_u32 v[10:11], s[20:21], v12, v13, v[14:15]\n", , CLOB)
KERNEL(k_mul_lo_u32, "v_mul_lo_u32 v10, v12, v13\n", , CLOB)
KERNEL(k_mul_hi_u32, "v_mul_hi_u32 v10, v12, v13\n", , CLOB)
KERNEL(k_mul_u32_u24, "v_mul_u32_u24 v10, v12, v13\n", , CLOB)
KERNEL(k_mad_u32_u24, "v_mad_u32_u24 v10, v12, v13, v14\n", , CLOB)
KERNEL(k_add_u32, "v_add_u32 v10, v12, v13\n", , CLOB)
KERNEL(k_lshl_add_u32, "v_lshl_add_u32 v10, v12, 4, v13\n", , CLOB)
KERNEL(k_xor_b32, "v_xor_b32 v10, v12, v13\n", , CLOB)
KERNEL(k_alignbit, "v_alignbit_b32 v10, v12, v13, 7\n", , CLOB)
KERNEL(k_lshlrev_b64, "v_lshlrev_b64 v[10:11], 5, v[12:13]\n", , CLOB)
KERNEL(k_max3_f32, "v_max3_f32 v10, v12, v13, v14\n", , CLOB)
KERNEL(k_rcp_f32, "v_rcp_f32 v10, v12\n", , CLOB)
KERNEL(k_sqrt_f32, "v_sqrt_f32 v10, v12\n", , CLOB)
KERNEL(k_log_f32, "v_log_f32 v10, v12\n", , CLOB)
KERNEL(k_mov_b32, "v_mov_b32 v10, v12\n", , CLOB)
KERNEL(k_mov_b64, "v_mov_b64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_s_and_b64, "s_and_b64 s[20:21], s[22:23], exec\n", , CLOB)
KERNEL(k_s_nop, "s_nop 0\n", , CLOB)
KERNEL(k_readlane, "v_readlane_b32 s20, v12, 3\n", , CLOB)
KERNEL(k_trig_preop, "v_trig_preop_f64 v[10:11], v[12:13], v14\n", , CLOB)
KERNEL(k_fract_f64, "v_fract_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_floor_f64, "v_floor_f64 v[10:11], v[12:13]\n", , CLOB)
KERNEL(k_cvt_i32_f64, "v_cvt_i32_f64 v10, v[12:13]\n", , CLOB)
KERNEL(k_cndmask_sgpr, "v_cndmask_b32_e64 v10, v12, v13, s[22:23]\n", asm volatile("s_mov_b64 s[22:23], 0x5555" ::: "s22", "s23"), CLOB)
KERNEL(k_cndmask_vcc_init, "v_cndmask_b32 v10, v12, v13, vcc\n", asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc"), CLOB)
KERNEL(k_cndmask_8dst, "v_cndmask_b32_e64 v10, v12, v13, s[22:23]\nv_cndmask_b32_e64 v11, v12, v13, s[22:23]\nv_cndmask_b32_e64 v14, v12, v13, s[22:23]\nv_cndmask_b32_e64 v15, v12, v13, s[22:23]\n", asm volatile("s_mov_b64 s[22:23], 0x5555" ::: "s22", "s23"), CLOB)
KERNEL(k_add_u32_sgpr, "v_add_u32 v10, s22, v12\n", asm volatile("s_mov_b32 s22, 0x5555" ::: "s22"), CLOB)
KERNEL(k_fma_f32_4dst, "v_fma_f32 v10, v12, v13, v14\nv_fma_f32 v11, v12, v13, v14\nv_fma_f32 v15, v12, v13, v14\nv_fma_f32 v16, v12, v13, v14\n", , CLOB)
KERNEL(k_fma_f64_4dst, "v_fma_f64 v[10:11], v[12:13], v[14:15], v[16:17]\nv_fma_f64 v[18:19], v[12:13], v[14:15], v[16:17]\nv_fma_f64 v[20:21], v[12:13], v[14:15], v[16:17]\nv_fma_f64 v[22:23], v[12:13], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_min_f32, "v_min_f32 v10, v12, v13\n", , CLOB)
KERNEL(k_cmp_f32, "v_cmp_lt_f32 vcc, v12, v13\n", , CLOB)
KERNEL(k_bfe_u32, "v_bfe_u32 v10, v12, 2, 2\n", , CLOB)
KERNEL(k_and_b32, "v_and_b32 v10, 0xf0000000, v12\n", , CLOB)
// dependent chains (latency): each instruction reads the previous result
KERNEL(k_dep_fma_f64, "v_fma_f64 v[10:11], v[10:11], v[14:15], v[16:17]\n", , CLOB)
KERNEL(k_dep_fma_f32, "v_fma_f32 v10, v10, v13, v14\n", , CLOB)
KERNEL(k_dep_rcp_f64, "v_rcp_f64 v[10:11], v[10:11]\n", , CLOB)
KERNEL(k_dep_cmp_cnd, "v_cmp_lt_f64 vcc, v[12:13], v[14:15]\nv_cndmask_b32 v10, v12, v13, vcc\n", , CLOB)
KERNEL(k_dep_cmp_sand, "v_cmp_lt_f64 vcc, v[12:13], v[14:15]\ns_and_b64 s[20:21], vcc, exec\n", , CLOB)

struct Case { const char* name; void (*fn)(long long*, int); int per_rep; };
#define C(n) {#n, n, 1}
int main(int argc, char** argv) {
    int waves = argc > 1 ? atoi(argv[1]) : 1; // waves per SIMD-ish: threads per block = 64 * waves (one block)
    std::vector<Case> cases = {C(k_fma_f32), C(k_pk_fma_f32), C(k_pk_mul_f32), C(k_pk_add_f32), C(k_fma_f64), C(k_mul_f64), C(k_add_f64), C(k_max_f64), C(k_rcp_f64), C(k_rsq_f64),
        C(k_sqrt_f64), C(k_div_scale_f64), C(k_div_fmas_f64), C(k_div_fixup_f64), C(k_cmp_f64), C(k_cmp_f64_sgpr), C(k_cndmask), C(k_cvt_f32_f64), C(k_cvt_f64_f32), C(k_cvt_f64_u32),
        C(k_ldexp_f64), C(k_frexp_mant_f64), C(k_mad_u64_u32), C(k_mul_lo_u32), C(k_mul_hi_u32), C(k_mul_u32_u24), C(k_mad_u32_u24), C(k_add_u32), C(k_lshl_add_u32), C(k_xor_b32), C(k_alignbit),
        C(k_lshlrev_b64), C(k_max3_f32), C(k_rcp_f32), C(k_sqrt_f32), C(k_log_f32), C(k_mov_b32), C(k_mov_b64), C(k_s_and_b64), C(k_s_nop), C(k_readlane), C(k_trig_preop), C(k_fract_f64), C(k_floor_f64), C(k_cvt_i32_f64),
        C(k_cndmask_sgpr), C(k_cndmask_vcc_init), {"k_cndmask_8dst", k_cndmask_8dst, 4}, C(k_add_u32_sgpr), {"k_fma_f32_4dst", k_fma_f32_4dst, 4}, {"k_fma_f64_4dst", k_fma_f64_4dst, 4}, C(k_min_f32), C(k_cmp_f32), C(k_bfe_u32), C(k_and_b32),
        C(k_dep_fma_f64), C(k_dep_fma_f32), C(k_dep_rcp_f64), {"k_dep_cmp_cnd", k_dep_cmp_cnd, 2}, {"k_dep_cmp_sand", k_dep_cmp_sand, 2}};
    setvbuf(stdout, nullptr, _IONBF, 0);
    long long* d;
    if (hipMalloc(&d, 64 * sizeof(long long)) != hipSuccess) { printf("no device\n"); return 1; }
    const int iters = 2000;
    printf("one block of %d wave(s) on one CU; cycles per wave-instruction (shader clock; clock64 ticks at 100 MHz are rescaled if so)\n", waves);
    for (auto& c : cases) {
        long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            hipMemset(d, 0, sizeof(long long));
            hipLaunchKernelGGL(c.fn, dim3(1), dim3(64 * waves), 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("%-20s %8.2f ticks per instruction (total %lld)\n", c.name, (double)h / ((double)iters * 32.0 * c.per_rep), h);
    }
    return 0;
}
