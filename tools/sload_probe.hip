// sload_probe.hip — what does a wave-uniform 64-byte record fetch through the scalar data cache cost on gfx950?
//
// The camera-ray traversal kernel (k_trace_w<0>) fetches the node record of a wave-uniform step with one
// s_load_dwordx16 and runs at ~0.6 of the VALU issue peak; an assembly loop with a third fewer instructions
// per step was not faster (DESIGN_HISTORY.md §9).  Is the scalar cache the floor?  Dependent chains of record fetches
// (next index = a dword of the record just fetched), 8 waves per SIMD on every CU, with
//   v0  s_load_dwordx16 only (latency / request rate of the scalar cache)
//   v1  s_load_dwordx16 + the 16 VALU of the two slab tests taken straight from SGPRs (what a step must do)
//   v2  as v1 with ~20 more VALU (what the compiled step does)
//   v3  the record fetched by the vector pipe instead (4 x global_load_dwordx4, same address in every lane)
//   v4  VALU only (16 per step, no fetch): the issue floor
// for tables of 16 KB (fits the scalar cache), 256 KB and 4 MB.
// build: hipcc --offload-arch=gfx950 -O3 -o sload_probe tools/sload_probe.hip ; run: ./sload_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr int kSteps = 4096;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) f32x4 *scalar_ptr;

__device__ __forceinline__ float vmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
__device__ __forceinline__ float vmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }

template <int MODE>
__global__ void __launch_bounds__(256, 8) k_probe(const float4 *__restrict__ tab, uint32_t mask, float *out) {
    __shared__ float2 lds[512];
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t wave_id = gid >> 6;
    uint32_t idx = __builtin_amdgcn_readfirstlane((wave_id * 7919u) & mask);
    const float ix = 1.0f + (float)(gid & 63u) * 1e-3f, iy = 1.5f, iz = 0.75f;
    float acc = 0.f;
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        if (MODE == 4) {
            float a = acc + 1.0f;
            const float t0 = vmax3(a * ix, a * iy, a * iz), t1 = vmin3((a + 1.f) * ix, a * iy, a * iz);
            const float t2 = vmax3(t0 * ix, t1 * iy, a * iz), t3 = vmin3(t0 * ix, t1 * iy, t2 * iz);
            acc = t3 * 1e-3f + t2 * 1e-4f;
            continue;
        }
        if (MODE >= 7) {  // v4 plus extra instructions of one kind: what does each cost beside a saturated VALU?
            float a = acc + 1.0f;
            const float t0 = vmax3(a * ix, a * iy, a * iz), t1 = vmin3((a + 1.f) * ix, a * iy, a * iz);
            const float t2 = vmax3(t0 * ix, t1 * iy, a * iz), t3 = vmin3(t0 * ix, t1 * iy, t2 * iz);
            acc = t3 * 1e-3f + t2 * 1e-4f;
            unsigned long long mm = 0;
            uint32_t x0 = idx, x1 = idx + 1, x2 = idx + 2, x3 = idx + 3;
            if (MODE == 7 || MODE == 8) {  // 8 / 16 independent 32-bit SALU
                asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                             "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                             : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3)::"scc");
                if (MODE == 8)
                    asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                                 "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                                 : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3)::"scc");
            } else if (MODE == 9) {  // 8 s_nop
                asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            } else if (MODE == 10) {  // 4 taken branches
                asm volatile("s_branch 1f\n1:\n\ts_branch 2f\n2:\n\ts_branch 3f\n3:\n\ts_branch 4f\n4:");
            } else if (MODE == 11) {  // 8 VALU compares
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %0\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %0\n\t"
                             "v_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %0\n\tv_cmp_lt_f32 vcc, %0, %1\n\tv_cmp_lt_f32 vcc, %1, %0"
                             ::"v"(t0), "v"(t1) : "vcc");
            } else if (MODE == 12) {  // 4 not-taken conditional branches
                asm volatile("s_cmp_eq_u32 %0, -1\n\ts_cbranch_scc1 1f\n\ts_cbranch_scc1 1f\n\ts_cbranch_scc1 1f\n\ts_cbranch_scc1 1f\n1:" ::"s"(x0) : "scc");
            } else if (MODE == 13) {  // 8 dependent SALU
                asm volatile("s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\t"
                             "s_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1\n\ts_add_u32 %0, %0, 1"
                             : "+s"(x0)::"scc");
            } else if (MODE == 15 || MODE == 16) {  // 1 / 2 ds_write_b64 (512 contiguous bytes per wave)
                lds[threadIdx.x] = make_float2(t0, t1);
                if (MODE == 16) lds[256 + threadIdx.x] = make_float2(t2, t3);
            } else if (MODE == 17) {  // 1 ds_write_b64 + 1 dependent ds_read_b64
                lds[threadIdx.x] = make_float2(t0, t1);
                const float2 r = lds[threadIdx.x ^ 1];
                acc += r.x;
            } else if (MODE == 18) {  // 2 exec-masked regions (s_and_saveexec / s_cbranch_execz / s_or exec), never skipped
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\ts_and_saveexec_b64 %2, vcc\n\ts_cbranch_execz 1f\n\tv_add_f32 %0, %0, %1\n1:\n\ts_or_b64 exec, exec, %2\n\t"
                             "v_cmp_lt_f32 vcc, %1, %0\n\ts_and_saveexec_b64 %2, vcc\n\ts_cbranch_execz 2f\n\tv_add_f32 %0, %0, %1\n2:\n\ts_or_b64 exec, exec, %2"
                             : "+v"(acc), "+v"(a), "=&s"(mm)::"vcc", "scc");
            } else if (MODE == 19) {  // 4 x (v_cmp -> vcc -> v_cndmask) dependent chain
                asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc\n\t"
                             "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_cmp_lt_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc"
                             : "+v"(acc), "+v"(a)::"vcc");
            } else if (MODE == 20) {  // 8 x (v_mul, s_add) interleaved
                asm volatile("v_mul_f32 %4, %4, %5\n\ts_add_u32 %0, %0, 1\n\tv_mul_f32 %4, %4, %5\n\ts_add_u32 %1, %1, 1\n\t"
                             "v_mul_f32 %4, %4, %5\n\ts_add_u32 %2, %2, 1\n\tv_mul_f32 %4, %4, %5\n\ts_add_u32 %3, %3, 1\n\t"
                             "v_mul_f32 %4, %4, %5\n\ts_add_u32 %0, %0, 1\n\tv_mul_f32 %4, %4, %5\n\ts_add_u32 %1, %1, 1\n\t"
                             "v_mul_f32 %4, %4, %5\n\ts_add_u32 %2, %2, 1\n\tv_mul_f32 %4, %4, %5\n\ts_add_u32 %3, %3, 1"
                             : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3), "+v"(acc) : "v"(ix) : "scc");
            } else if (MODE == 21) {  // the same 8 v_mul, then the same 8 s_add
                asm volatile("v_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\t"
                             "v_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\tv_mul_f32 %4, %4, %5\n\t"
                             "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1\n\t"
                             "s_add_u32 %0, %0, 1\n\ts_add_u32 %1, %1, 1\n\ts_add_u32 %2, %2, 1\n\ts_add_u32 %3, %3, 1"
                             : "+s"(x0), "+s"(x1), "+s"(x2), "+s"(x3), "+v"(acc) : "v"(ix) : "scc");
            } else if (MODE == 22) {  // 8 v_mul only
                asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\t"
                             "v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1"
                             : "+v"(acc) : "v"(ix));
            } else if (MODE == 23) {  // 8 v_pk_mul_f32 (two products each)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 pa = {acc, a}, pb = {ix, iy};
                asm volatile("v_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\t"
                             "v_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1\n\tv_pk_mul_f32 %0, %0, %1"
                             : "+v"(pa) : "v"(pb));
                acc = pa.x + pa.y;
            } else if (MODE == 24) {  // 8 independent v_pk_mul_f32
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 p0 = {acc, a}, p1 = {a, acc}, p2 = {t0, t1}, p3 = {t2, t3}, pb = {ix, iy};
                asm volatile("v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4\n\t"
                             "v_pk_mul_f32 %0, %0, %4\n\tv_pk_mul_f32 %1, %1, %4\n\tv_pk_mul_f32 %2, %2, %4\n\tv_pk_mul_f32 %3, %3, %4"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
                acc = (p0.x + p1.y) + (p2.x + p3.y);
            } else if (MODE == 14) {  // 8 64-bit mask ops
                unsigned long long m0 = idx, m1 = idx + 7;
                asm volatile("s_and_b64 %0, %0, %1\n\ts_or_b64 %1, %0, %1\n\ts_and_b64 %0, %0, %1\n\ts_or_b64 %1, %0, %1\n\t"
                             "s_and_b64 %0, %0, %1\n\ts_or_b64 %1, %0, %1\n\ts_and_b64 %0, %0, %1\n\ts_or_b64 %1, %0, %1"
                             : "+s"(m0), "+s"(m1)::"scc");
                x0 += (uint32_t)m0;
            }
            idx = (x0 + x1 + x2 + x3) & mask;
            continue;
        }
        if (MODE == 3) {
            const float4 *p = tab + (size_t)idx * 4;
            const float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
            const float tn0 = vmax3(q0.x * ix, q0.y * iy, q0.z * iz), tf0 = vmin3(q0.w * ix, q1.x * iy, q1.y * iz);
            const float tn1 = vmax3(q1.z * ix, q1.w * iy, q2.x * iz), tf1 = vmin3(q2.y * ix, q2.z * iy, q2.w * iz);
            acc += (tn0 <= tf0 ? tn0 : tf0) + (tn1 <= tf1 ? tn1 : tf1);
            idx = __builtin_amdgcn_readfirstlane(__float_as_uint(q3.x)) & mask;
            continue;
        }
        const scalar_ptr rec = (scalar_ptr)(uintptr_t)((const char *)tab + ((size_t)idx << 6));
        const f32x4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3];
        if (MODE == 6) {  // a second record per step: twice the bytes through the scalar cache
            const scalar_ptr rec2 = (scalar_ptr)(uintptr_t)((const char *)tab + ((size_t)(idx ^ 1u) << 6));
            const f32x4 u0 = rec2[0], u1 = rec2[1], u2 = rec2[2], u3 = rec2[3];
            acc += (u0.x + u1.y) + (u2.z + u3.w) + (u0.w + u1.x) + (u2.x + u3.y) + (u0.y + u0.z) + (u1.z + u1.w) + (u2.y + u2.w) + (u3.x + u3.z);
        }
        if (MODE >= 1) {
            const float tn0 = vmax3(r0.x * ix, r0.y * iy, r0.z * iz), tf0 = vmin3(r0.w * ix, r1.x * iy, r1.y * iz);
            const float tn1 = vmax3(r1.z * ix, r1.w * iy, r2.x * iz), tf1 = vmin3(r2.y * ix, r2.z * iy, r2.w * iz);
            float v = (tn0 <= tf0 ? tn0 : tf0) + (tn1 <= tf1 ? tn1 : tf1);
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 10; ++k) v = v * ix + tn1, v = fmaxf(v, tf0);
            }
            acc += v;
        } else {
            acc += r0.x;
        }
        idx = __float_as_uint(r3.x) & mask;
    }
    out[gid] = acc;
}

template <int MODE>
static void run(const char *name, const float4 *tab, uint32_t records, float *out, int blocks, int lds = 0) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), lds, 0, tab, records - 1u, out);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), lds, 0, tab, records - 1u, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    // blocks * 4 waves over 1024 SIMDs, kSteps each; SIMD-cycles per wave-step at 2.4 GHz
    const double wave_steps_per_simd = (double)blocks * 4.0 * kSteps / 1024.0;
    printf("%-58s %8.3f ms  %7.1f SIMD-cycles per wave-step\n", name, ms, ms * 1e-3 * 2.4e9 / wave_steps_per_simd);
}

int main() {
    const int blocks = 256 * 8;
    float *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    for (uint32_t records : {256u, 4096u, 65536u}) {
        std::vector<float> h((size_t)records * 16);
        // a random cyclic permutation in dword 12 of every record, small floats elsewhere
        std::vector<uint32_t> perm(records);
        for (uint32_t i = 0; i < records; ++i) perm[i] = i;
        uint64_t st = 88172645463325252ull;
        for (uint32_t i = records - 1; i > 0; --i) {
            st ^= st << 13, st ^= st >> 7, st ^= st << 17;
            std::swap(perm[i], perm[st % (i + 1)]);
        }
        for (uint32_t i = 0; i < records; ++i) {
            for (int k = 0; k < 16; ++k) h[(size_t)i * 16 + k] = 0.25f + 0.001f * (float)((i * 16 + k) % 977);
            const uint32_t nxt = perm[(i + 1) % records];  // not a permutation cycle guarantee; any spread will do
            memcpy(&h[(size_t)perm[i] * 16 + 12], &nxt, 4);
        }
        float4 *tab;
        CK(hipMalloc(&tab, h.size() * 4));
        CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        printf("---- table %u KB\n", records * 64 / 1024);
        run<0>("v0 s_load_dwordx16 chain only", tab, records, out, blocks);
        run<1>("v1 s_load_dwordx16 + 16 VALU slab tests from SGPRs", tab, records, out, blocks);
        run<2>("v2 as v1 + 20 VALU", tab, records, out, blocks);
        run<3>("v3 record through the vector pipe (uniform address) + 16", tab, records, out, blocks);
        if (records == 256u) {
            run<4>("v4 16 VALU, no fetch", tab, records, out, blocks);
            run<6>("v6 as v1 with a second record fetched per step", tab, records, out, blocks);
            run<7>("v7  v4 + 8 independent s_add_u32", tab, records, out, blocks);
            run<8>("v8  v4 + 16 independent s_add_u32", tab, records, out, blocks);
            run<13>("v13 v4 + 8 dependent s_add_u32", tab, records, out, blocks);
            run<14>("v14 v4 + 8 dependent 64-bit mask ops", tab, records, out, blocks);
            run<9>("v9  v4 + 8 s_nop", tab, records, out, blocks);
            run<10>("v10 v4 + 4 taken s_branch", tab, records, out, blocks);
            run<12>("v12 v4 + s_cmp + 4 not-taken s_cbranch", tab, records, out, blocks);
            run<11>("v11 v4 + 8 v_cmp", tab, records, out, blocks);
            run<22>("v22 v4 + 8 v_mul", tab, records, out, blocks);
            run<23>("v23 v4 + 8 dependent v_pk_mul_f32 (+1 add)", tab, records, out, blocks);
            run<24>("v24 v4 + 8 independent v_pk_mul_f32 (+3 add)", tab, records, out, blocks);
            run<20>("v20 v4 + 8 x (v_mul, s_add) interleaved", tab, records, out, blocks);
            run<21>("v21 v4 + 8 v_mul then 8 s_add", tab, records, out, blocks);
            run<15>("v15 v4 + 1 ds_write_b64", tab, records, out, blocks);
            run<16>("v16 v4 + 2 ds_write_b64", tab, records, out, blocks);
            run<17>("v17 v4 + ds_write_b64 + dependent ds_read_b64", tab, records, out, blocks);
            run<18>("v18 v4 + 2 exec-masked regions (2 VALU + 4 SALU + 2 cbranch)", tab, records, out, blocks);
            run<19>("v19 v4 + 4 x (v_cmp -> v_cndmask) chain", tab, records, out, blocks);
            for (int lds : {40 * 1024, 64 * 1024}) {
                const int w = lds == 40 * 1024 ? 4 : 2;
                char nm[96];
                snprintf(nm, sizeof nm, "v0 at %d waves per SIMD", w), run<0>(nm, tab, records, out, blocks, lds);
                snprintf(nm, sizeof nm, "v1 at %d waves per SIMD", w), run<1>(nm, tab, records, out, blocks, lds);
                snprintf(nm, sizeof nm, "v3 at %d waves per SIMD", w), run<3>(nm, tab, records, out, blocks, lds);
                snprintf(nm, sizeof nm, "v4 at %d waves per SIMD", w), run<4>(nm, tab, records, out, blocks, lds);
            }
        }
        CK(hipFree(tab));
    }
    return 0;
}
