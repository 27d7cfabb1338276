// Developer probe: sustained issue cost (cycles per instruction per SIMD) of the instruction kinds the traversal kernels are made of, with
// DISTINCT operands, at 1, 4 and 8 waves per SIMD.  Build and run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -w tools/probes/issue_probe.hip -o /tmp/issue_probe && /tmp/issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int K>
__global__ void rate(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f, e = a + 4.0f, f = a + 5.0f, g = a + 6.0f, h = a + 7.0f;
    double da = a, db = b, dc = c, dd = d;
    uint32_t u = threadIdx.x, v = u + 1;
    for (int it = 0; it < iters; it++) {
        if (K == 0) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %3\n\tv_fma_f32 %4, %5, %6, %7" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(d), "v"(f), "v"(g), "v"(h));) }
        if (K == 1) { REP16(asm volatile("v_add_f32 %0, %1, %2\n\tv_add_f32 %3, %4, %5" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(f), "v"(g));) }
        if (K == 2) { REP16(asm volatile("v_max3_f32 %0, %1, %2, %3\n\tv_min3_f32 %4, %5, %6, %7" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(d), "v"(f), "v"(g), "v"(h));) }
        if (K == 3) { REP16(asm volatile("v_fma_f32 %0, s20, %1, %2\n\tv_fma_f32 %3, s21, %4, %5" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(f), "v"(g) : "s20", "s21");) }
        if (K == 4) { REP16(asm volatile("v_cmp_le_f32 vcc, %0, %1\n\tv_cndmask_b32 %2, %3, %4, vcc" : : "v"(a), "v"(b), "v"(e), "v"(f), "v"(g) : "vcc");) }
        if (K == 5) { REP16(asm volatile("v_fma_f64 %0, %1, %2, %3\n\tv_fma_f64 %1, %0, %2, %3" : "+v"(da), "+v"(db) : "v"(dc), "v"(dd));) }
        if (K == 6) { REP16(asm volatile("v_add_f64 %0, %1, %2\n\tv_mul_f64 %1, %0, %2" : "+v"(da), "+v"(db) : "v"(dc));) }
        if (K == 7) { REP16(asm volatile("v_readlane_b32 s20, %0, 3\n\tv_writelane_b32 %1, s21, 5" : : "v"(u), "v"(v) : "s20", "s21");) }
        if (K == 8) { REP16(asm volatile("s_add_u32 s20, s20, 1\n\ts_and_b32 s21, s21, s20" ::: "s20", "s21", "scc");) }
        if (K == 9) { REP16(asm volatile("s_cmp_lg_u32 s20, 0\n\ts_cselect_b32 s21, s20, s21" ::: "s21", "scc");) }
        if (K == 10) { REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(da) : "v"(db), "v"(dc), "v"(dd)); asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "+v"(db) : "v"(da), "v"(dc), "v"(dd));) }
        if (K == 11) { REP16(asm volatile("v_fma_f32 %0, %1, %2, %3\n\ts_add_u32 s20, s20, 1" : "+v"(a) : "v"(b), "v"(c), "v"(d) : "s20", "scc");) }
        if (K == 12) { REP16(asm volatile("v_max_f32 %0, %1, %2\n\tv_min_f32 %3, %4, %5" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(f), "v"(g));) }
        if (K == 13) { REP16(asm volatile("v_and_b32 %0, %1, %2\n\tv_or_b32 %3, %4, %5" : "+v"(u), "+v"(v) : "v"(b), "v"(c), "v"(f), "v"(g));) }
        if (K == 14) { REP16(asm volatile("v_div_fixup_f64 %0, %1, %2, %3\n\tv_rcp_f64 %1, %0" : "+v"(da), "+v"(db) : "v"(dc), "v"(dd));) }
        if (K == 16) { REP16(asm volatile("v_pk_fma_f32 %0, s[20:21], %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_lo:[1,0,0]\n\tv_pk_fma_f32 %3, s[22:23], %1, %2 op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(da), "+v"(db) : "v"(dc), "v"(dd));) }
        if (K == 17) { REP16(asm volatile("v_max3_f32 %0, %1, %2, %3 clamp\n\tv_min3_f32 %4, %5, %6, %7" : "+v"(a), "+v"(e) : "v"(b), "v"(c), "v"(d), "v"(f), "v"(g), "v"(h));) }
        if (K == 18) { REP16(asm volatile("v_cmp_le_f32_e64 s[20:21], %1, %2\n\tv_addc_co_u32_e64 %0, vcc, %0, %0, s[20:21]" : "+v"(u) : "v"(b), "v"(c) : "s20", "s21", "vcc");) }
        if (K == 19) { REP16(asm volatile("v_max3_i32 %0, %1, %2, %3\n\tv_min3_i32 %4, %5, %6, %7" : "+v"(u), "+v"(v) : "v"(b), "v"(c), "v"(d), "v"(f), "v"(g), "v"(h));) }
        if (K == 15) { REP16(asm volatile("v_max_f64 %0, %1, %2\n\tv_min_f64 %1, %0, %2" : "+v"(da), "+v"(db) : "v"(dc));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + e + (float)(da + db) + u + v;
}
int main() {
    float* d_out; hipMalloc(&d_out, 1 << 24);
    const char* names[20] = {"v_fma_f32 x2 (3 VGPR srcs)", "v_add_f32 x2", "v_max3+v_min3", "v_fma_f32 x2 (1 SGPR src)", "v_cmp+v_cndmask", "v_fma_f64 x2", "v_add_f64+v_mul_f64", "v_readlane+v_writelane",
                             "s_add+s_and", "s_cmp+s_cselect", "v_pk_fma_f32 x2", "v_fma_f32 + s_add", "v_max_f32+v_min_f32", "v_and+v_or", "v_div_fixup_f64+v_rcp_f64", "v_max_f64+v_min_f64", "v_pk_fma_f32 x2 (SGPR pair, op_sel, neg)", "v_max3_f32 clamp + v_min3_f32", "v_cmp_e64(sgpr) + v_addc(sgpr carry)", "v_max3_i32+v_min3_i32"};
    const int iters = 4000;
    int waves_list[3] = {1, 5, 8};
    for (int wi = 0; wi < 3; wi++) {
        const int waves = waves_list[wi];
        for (int k = 0; k < 20; k++) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                const dim3 g(256 * 4 * waves), b(64);
                switch (k) {
                    case 0: rate<0><<<g, b>>>(d_out, iters); break; case 1: rate<1><<<g, b>>>(d_out, iters); break; case 2: rate<2><<<g, b>>>(d_out, iters); break;
                    case 3: rate<3><<<g, b>>>(d_out, iters); break; case 4: rate<4><<<g, b>>>(d_out, iters); break; case 5: rate<5><<<g, b>>>(d_out, iters); break;
                    case 6: rate<6><<<g, b>>>(d_out, iters); break; case 7: rate<7><<<g, b>>>(d_out, iters); break; case 8: rate<8><<<g, b>>>(d_out, iters); break;
                    case 9: rate<9><<<g, b>>>(d_out, iters); break; case 10: rate<10><<<g, b>>>(d_out, iters); break; case 11: rate<11><<<g, b>>>(d_out, iters); break;
                    case 12: rate<12><<<g, b>>>(d_out, iters); break; case 13: rate<13><<<g, b>>>(d_out, iters); break; case 14: rate<14><<<g, b>>>(d_out, iters); break;
                    case 15: rate<15><<<g, b>>>(d_out, iters); break; case 16: rate<16><<<g, b>>>(d_out, iters); break; case 17: rate<17><<<g, b>>>(d_out, iters); break; case 18: rate<18><<<g, b>>>(d_out, iters); break; case 19: rate<19><<<g, b>>>(d_out, iters); break;
                }
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double n_instr = (double)iters * 32.0 * waves;          // instructions per SIMD
            printf("waves/SIMD %d  %-28s %8.3f ms  %.2f ns per instruction per SIMD (x2.4 = %.2f cycles at 2.4 GHz)\n", waves, names[k], ms, ms * 1e6 / n_instr, ms * 1e6 / n_instr * 2.4);
        }
    }
    return 0;
}
