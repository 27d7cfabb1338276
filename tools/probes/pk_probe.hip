// Developer probe: (1) semantics of v_pk_fma_f32 with an SGPR-pair source and op_sel / neg_lo, (2) issue cost of v_pk_fma_f32 vs v_fma_f32 per wave
// at 1 and 5 waves per SIMD.  hipcc --offload-arch=gfx950 -O2 tools/probes/pk_probe.hip -o /tmp/pk_probe && /tmp/pk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void sem(const float* in, float* out) {
    // uniform pair {s0, s1} from memory (scalar load), per-lane pairs a = {a0,a1}, t = {t0,t1}
    const float s0 = in[0], s1 = in[1];
    u32x2 sp; sp.x = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, s0)); sp.y = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(uint32_t, s1));
    f32x2 a = {in[2] + threadIdx.x, in[3] + threadIdx.x}, t = {in[4], in[5]};
    f32x2 r0, r1, r2, r3;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r0) : "s"(sp), "v"(a), "v"(t));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0] neg_lo:[1,0,0]" : "=v"(r1) : "s"(sp), "v"(a), "v"(t));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,1] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r2) : "s"(sp), "v"(a), "v"(t));
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r3) : "s"(sp), "v"(a), "v"(t));
    float* o = out + threadIdx.x * 8;
    o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y;
}

template <int kKind>
__global__ void rate(float* out, int iters, unsigned long long* cyc) {
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; i++) v[i] = threadIdx.x * 0.001f + i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            if (kKind == 0) {        // two plain FMAs
                asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1" : "+v"(v[i]), "+v"(v[i + 1]));
            } else if (kKind == 1) { // one packed FMA on the same two values
                f32x2 p = {v[i], v[i + 1]};
                asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(p));
                v[i] = p.x; v[i + 1] = p.y;
            } else if (kKind == 2) { // two f64 FMAs? no: max3 + min3
                asm volatile("v_max3_f32 %0, %0, %1, %0\n\tv_min3_f32 %1, %1, %0, %1" : "+v"(v[i]), "+v"(v[i + 1]));
            } else {                 // two SALU ops
                asm volatile("s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1" ::: "s20", "s21");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
    float h_in[6] = {3.0f, 5.0f, 10.0f, 20.0f, 100.0f, 200.0f};
    float *d_in, *d_out; unsigned long long* d_cyc;
    hipMalloc(&d_in, sizeof h_in); hipMalloc(&d_out, 1 << 22); hipMalloc(&d_cyc, 8);
    hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice);
    sem<<<1, 64>>>(d_in, d_out);
    float o[16]; hipMemcpy(o, d_out, sizeof o, hipMemcpyDeviceToHost);
    printf("lane0: plain {s0*a0+t0, s1*a1+t1} = {%g, %g} expect {130, 300}\n", o[0], o[1]);
    printf("lane0: op_sel[1,0,0] hi[1,0,0] neg_lo -> {-s1*a0+t0, s1*a0+t0} = {%g, %g} expect {50, 150}\n", o[2], o[3]);
    printf("lane0: op_sel[0,1,1] hi[0,1,1] neg_lo -> {-s0*a1+t1, s0*a1+t1} = {%g, %g} expect {140, 260}\n", o[4], o[5]);
    printf("lane0: op_sel[0,0,0] hi[1,0,1] -> {s0*a0+t0, s1*a0+t1} = {%g, %g} expect {130, 250}\n", o[6], o[7]);
    const char* names[4] = {"2 x v_fma_f32", "1 x v_pk_fma_f32", "v_max3+v_min3", "2 x s_add_u32"};
    for (int waves = 1; waves <= 8; waves += (waves == 1 ? 3 : 4)) {
        for (int k = 0; k < 4; k++) {
            const int iters = 20000;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&] {
                if (k == 0) rate<0><<<256 * 4 * waves, 64>>>(d_out, iters, d_cyc);
                if (k == 1) rate<1><<<256 * 4 * waves, 64>>>(d_out, iters, d_cyc);
                if (k == 2) rate<2><<<256 * 4 * waves, 64>>>(d_out, iters, d_cyc);
                if (k == 3) rate<3><<<256 * 4 * waves, 64>>>(d_out, iters, d_cyc);
            };
            launch(); hipDeviceSynchronize();
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long cyc; hipMemcpy(&cyc, d_cyc, 8, hipMemcpyDeviceToHost);
            printf("waves/SIMD %d  %-18s  %.3f ms  wave0 memtime ticks per pair: %.2f\n", waves, names[k], ms, (double)cyc / (iters * 16.0));
        }
    }
    return 0;
}
