// ta_probe.hip — what does a scattered 64-byte record gather cost on gfx950's vector-memory pipe?
//
// The bounce-ray traversal kernel (k_trace_w<1>) reads one 64-byte node record per lane and step as
// four 16-byte loads and sits at the texture pipe (DESIGN_HISTORY.md §5).  Two cost models fit round 1's
// counters equally well:
//   A  a wave-level dwordx4 load costs a fixed ~16 pipe cycles whatever lanes are active
//      -> only fewer wave-steps (higher lane occupancy) help
//   B  it costs one tag lookup per distinct 64/128-byte segment a quad touches
//      -> letting the 4 lanes of a quad fetch each other's records (64 contiguous bytes per quad and
//         load) cuts the lookups 4x
// This probe separates them: dependent chains of random record fetches from a 10 MB table
// (cache-resident like the scene), 6 waves per SIMD, with
//   v0  per-lane fetch, all lanes active            v1  per-lane fetch, 1 lane per quad active
//   v2  per-lane fetch, lanes 0..15 active          v3  quad-cooperative fetch + DPP transpose, all lanes
//   v4  quad-cooperative fetch, 1 lane per quad has a record to fetch
//   v5  per-lane fetch, half the lanes (even) active
// build: hipcc --offload-arch=gfx950 -O3 -o ta_probe tools/ta_probe.hip ; run: ./ta_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                \
            exit(1);                                                               \
        }                                                                          \
    } while (0)

constexpr uint32_t kRecords = 256 * 1024;  // x 64 B = 16 MB at most (power of two: the index update stays cheap)
__constant__ uint32_t c_mask = kRecords - 1u;   // records actually used - 1
__constant__ uint32_t c_percent = 38u;          // modes 8/9: share of lanes that walk a chain
constexpr int kSteps = 2048;

__device__ __forceinline__ uint32_t next_index(uint32_t idx, float acc) {
    const uint32_t h = idx * 2654435761u + __float_as_uint(acc);
    return (h >> 13) & c_mask;
}

// a stand-in for the two slab tests: ~40 VALU on the 14 payload dwords
__device__ __forceinline__ float consume(float4 a, float4 b, float4 c, float2 d, float x) {
    float s = x;
    s = fmaf(a.x, s, a.y), s = fmaf(a.z, s, a.w), s = fmaf(b.x, s, b.y), s = fmaf(b.z, s, b.w);
    s = fmaf(c.x, s, c.y), s = fmaf(c.z, s, c.w), s = fmaf(d.x, s, d.y);
    return s * 1e-3f;
}

template <int MODE>
__global__ void __launch_bounds__(256, 6) k_lane(const float4 *__restrict__ tab, float *out) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = true;
    if (MODE == 1) active = (lane & 3u) == 0;
    if (MODE == 2) active = lane < 16;
    if (MODE == 5) active = (lane & 1u) == 0;
    if (MODE == 8) active = ((gid * 2246822519u) >> 8) % 100u < c_percent;  // that share of the lanes, scattered
    uint32_t idx = (gid * 7919u) & c_mask;
    float acc = (float)gid * 1e-6f;
    if (active) {
#pragma unroll 1
        for (int s = 0; s < kSteps; ++s) {
            const float4 a = tab[idx * 4], b = tab[idx * 4 + 1], c = tab[idx * 4 + 2];
            const float2 d = ((const float2 *)tab)[idx * 8 + 6];
            acc = consume(a, b, c, d, acc);
            idx = next_index(idx, acc);
        }
    }
    out[gid] = acc + (float)idx;
}

__device__ __forceinline__ float xchg(float from, float keep, bool keep_mine, int ctrl_is_xor2) {
    const int t = ctrl_is_xor2 ? __builtin_amdgcn_mov_dpp(__float_as_int(from), 0x4E, 0xF, 0xF, true)
                               : __builtin_amdgcn_mov_dpp(__float_as_int(from), 0xB1, 0xF, 0xF, true);
    return keep_mine ? keep : __int_as_float(t);
}
__device__ __forceinline__ float4 xchg4(float4 from, float4 keep, bool keep_mine, int x2) {
    return make_float4(xchg(from.x, keep.x, keep_mine, x2), xchg(from.y, keep.y, keep_mine, x2),
                       xchg(from.z, keep.z, keep_mine, x2), xchg(from.w, keep.w, keep_mine, x2));
}


// One butterfly stage of the quad transpose on a register pair, one v_cndmask_b32_dpp per dword
// (D = vcc ? src1 : dpp(src0)):  po = keep_p ? p : partner's q ;  qo = keep_q ? q : partner's p
#define VMX_XCHG_PAIR(NAME, PERM)                                                                                     \
    __device__ __forceinline__ void NAME(const float4 &p, const float4 &q, float4 &po, float4 &qo,                    \
                                         unsigned long long keep_p, unsigned long long keep_q) {                     \
        asm volatile("s_nop 1\n\t"                                                                                   \
                     "s_mov_b64 vcc, %16\n\t"                                                                        \
                     "v_cndmask_b32_dpp %0, %12, %8, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %1, %13, %9, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %2, %14, %10, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "v_cndmask_b32_dpp %3, %15, %11, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "s_mov_b64 vcc, %17\n\t"                                                                        \
                     "v_cndmask_b32_dpp %4, %8, %12, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %5, %9, %13, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"          \
                     "v_cndmask_b32_dpp %6, %10, %14, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf\n\t"         \
                     "v_cndmask_b32_dpp %7, %11, %15, vcc quad_perm:" PERM " row_mask:0xf bank_mask:0xf"             \
                     : "=&v"(po.x), "=&v"(po.y), "=&v"(po.z), "=&v"(po.w), "=&v"(qo.x), "=&v"(qo.y), "=&v"(qo.z),    \
                       "=&v"(qo.w)                                                                                   \
                     : "v"(p.x), "v"(p.y), "v"(p.z), "v"(p.w), "v"(q.x), "v"(q.y), "v"(q.z), "v"(q.w), "s"(keep_p),  \
                       "s"(keep_q)                                                                                   \
                     : "vcc");                                                                                       \
    }
VMX_XCHG_PAIR(xchg_pair_1, "[1,0,3,2]")
VMX_XCHG_PAIR(xchg_pair_2, "[2,3,0,1]")

// quad-cooperative fetch: load i brings record of quad-lane i, lane j takes its 16-byte piece j
template <int MODE>
__global__ void __launch_bounds__(256, 6) k_quad(const float4 *__restrict__ tab, float *out) {
    const uint32_t lane = threadIdx.x & 63u, ql = lane & 3u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool wants = (MODE == 4 || MODE == 7) ? ql == 0 : true;
    if (MODE == 9) wants = ((gid * 2246822519u) >> 8) % 100u < c_percent;  // does this lane have a record of its own to fetch
    uint32_t idx = (gid * 7919u) & c_mask;
    float acc = (float)gid * 1e-6f;
    const bool even = (lane & 1u) == 0, low = (lane & 2u) == 0;
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        const uint32_t mine = wants ? idx : 0xFFFFFFFFu;
        float4 x[4];
#define FETCH(i)                                                                                        \
    {                                                                                                   \
        const uint32_t r = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, (i) * 0x55, 0xF, 0xF, true);   \
        x[i] = make_float4(0.f, 0.f, 0.f, 0.f);                                                         \
        if (r != 0xFFFFFFFFu) x[i] = tab[r * 4 + ql];                                                   \
    }
        FETCH(0) FETCH(1) FETCH(2) FETCH(3)
#undef FETCH
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        if (MODE >= 6) {  // asm form: one v_cndmask_b32_dpp per dword and stage
            xchg_pair_1(x[0], x[1], a0, a1, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
            xchg_pair_1(x[2], x[3], a2, a3, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
            xchg_pair_2(a0, a2, b0, b2, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);
            xchg_pair_2(a1, a3, b1, b3, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);
        } else {  // HIP form (hipcc emits v_mov_b32_dpp + v_cndmask_b32: two instructions per dword and stage)
            a0 = xchg4(x[1], x[0], even, 0), a1 = xchg4(x[0], x[1], !even, 0);
            a2 = xchg4(x[3], x[2], even, 0), a3 = xchg4(x[2], x[3], !even, 0);
            b0 = xchg4(a2, a0, low, 1), b2 = xchg4(a0, a2, !low, 1);
            b1 = xchg4(a3, a1, low, 1), b3 = xchg4(a1, a3, !low, 1);
        }
        if (wants) {
            acc = consume(b0, b1, b2, make_float2(b3.x, b3.y), acc);
            idx = next_index(idx, acc);
        }
    }
    out[gid] = acc + (float)idx;
}

// quad-cooperative fetch with the 4x4 transposition through LDS instead of registers (round 3 experiment): every lane
// writes its four 16-byte pieces into the 64-byte slots of their owners (4 KB of staging per wave) and reads its own
// record back: 4 ds_write_b128 + 4 ds_read_b128 instead of 32 v_cndmask_b32_dpp + 8 s_mov + 4 s_nop.
// SWZ: XOR the 16-byte piece position with bits of the owner's index so that the 16 quads do not all hit the same banks
template <int SWZ>
__global__ void __launch_bounds__(256, 6) k_quad_lds(const float4 *__restrict__ tab, float *out) {
    __shared__ float4 stage[4][64 * 4];  // per wave: 64 records x 4 pieces
    const uint32_t lane = threadIdx.x & 63u, ql = lane & 3u, wave = threadIdx.x >> 6;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool wants = ((gid * 2246822519u) >> 8) % 100u < c_percent;
    uint32_t idx = (gid * 7919u) & c_mask;
    float acc = (float)gid * 1e-6f;
    float4 *st = stage[wave];
    auto slot = [&](uint32_t owner, uint32_t piece) {
        const uint32_t p = SWZ == 0 ? piece : SWZ == 1 ? (piece ^ (owner & 3u)) : (piece ^ ((owner >> 2) & 3u));
        return owner * 4u + p;
    };
    uint32_t w[4], r[4];
    for (uint32_t i = 0; i < 4; ++i) w[i] = slot((lane & ~3u) + i, ql), r[i] = slot(lane, i);
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        const uint32_t mine = wants ? idx : 0u;  // a lane without a record of its own fetches record 0 (hot), as the kernel does
        float4 x[4];
#define FETCH(i)                                                                                        \
    {                                                                                                   \
        const uint32_t rr = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, (i) * 0x55, 0xF, 0xF, true);  \
        x[i] = tab[rr * 4 + ql];                                                                        \
    }
        FETCH(0) FETCH(1) FETCH(2) FETCH(3)
#undef FETCH
        st[w[0]] = x[0], st[w[1]] = x[1], st[w[2]] = x[2], st[w[3]] = x[3];
        __builtin_amdgcn_wave_barrier();
        const float4 b0 = st[r[0]], b1 = st[r[1]], b2 = st[r[2]], b3 = st[r[3]];
        __builtin_amdgcn_wave_barrier();
        if (wants) {
            acc = consume(b0, b1, b2, make_float2(b3.x, b3.y), acc);
            idx = next_index(idx, acc);
        }
    }
    out[gid] = acc + (float)idx;
}
// the register form with unconditional loads (as the kernel does), for a like-for-like comparison
__global__ void __launch_bounds__(256, 6) k_quad_uncond(const float4 *__restrict__ tab, float *out) {
    const uint32_t lane = threadIdx.x & 63u, ql = lane & 3u;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    bool wants = ((gid * 2246822519u) >> 8) % 100u < c_percent;
    uint32_t idx = (gid * 7919u) & c_mask;
    float acc = (float)gid * 1e-6f;
#pragma unroll 1
    for (int s = 0; s < kSteps; ++s) {
        const uint32_t mine = wants ? idx : 0u;
        float4 x[4];
#define FETCH(i)                                                                                        \
    {                                                                                                   \
        const uint32_t rr = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, (i) * 0x55, 0xF, 0xF, true);  \
        x[i] = tab[rr * 4 + ql];                                                                        \
    }
        FETCH(0) FETCH(1) FETCH(2) FETCH(3)
#undef FETCH
        float4 a0, a1, a2, a3, b0, b1, b2, b3;
        xchg_pair_1(x[0], x[1], a0, a1, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
        xchg_pair_1(x[2], x[3], a2, a3, 0x5555555555555555ull, 0xAAAAAAAAAAAAAAAAull);
        xchg_pair_2(a0, a2, b0, b2, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);
        xchg_pair_2(a1, a3, b1, b3, 0x3333333333333333ull, 0xCCCCCCCCCCCCCCCCull);
        if (wants) {
            acc = consume(b0, b1, b2, make_float2(b3.x, b3.y), acc);
            idx = next_index(idx, acc);
        }
    }
    out[gid] = acc + (float)idx;
}

template <class K>
double run(K kernel, const float4 *tab, float *out, int grid, const char *name, double lanes_frac) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, tab, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, tab, out);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double wave_steps = (double)grid * 4 * kSteps;
    const double cu_cycles_per_wave_step = ms * 1e-3 * 2.4e9 / (wave_steps / 256.0);
    printf("%-44s %8.3f ms  %7.1f CU-cycles per wave-step  %7.2f G records/s\n", name, ms, cu_cycles_per_wave_step,
           wave_steps * 64 * lanes_frac / (ms * 1e-3) / 1e9);
    return ms;
}

int main() {
    std::vector<float> h((size_t)kRecords * 16);
    uint32_t s = 12345;
    for (auto &v : h) {
        s = s * 1664525u + 1013904223u;
        v = (float)(s >> 8) * (1.0f / 16777216.0f);
    }
    float4 *tab;
    float *out;
    const int grid = 256 * 6;  // 6 blocks of 4 waves per CU = 6 waves per SIMD
    CK(hipMalloc(&tab, h.size() * 4));
    CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CK(hipMemcpy(tab, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (uint32_t recs : {4096u, 32768u, 262144u}) {
        for (uint32_t pct : {100u, 84u, 38u}) {
            const uint32_t mask = recs - 1u;
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_mask), &mask, 4));
            CK(hipMemcpyToSymbol(HIP_SYMBOL(c_percent), &pct, 4));
            printf("---- table %u KB, %u %% of the lanes\n", recs / 16, pct);
            run(k_lane<8>, tab, out, grid, "per-lane fetch (4 x 16 B per lane)", pct / 100.0);
            run(k_quad<9>, tab, out, grid, "quad fetch + asm cndmask_dpp transpose", pct / 100.0);
            run(k_quad_uncond, tab, out, grid, "  same, unconditional loads (as k_trace_w<1>)", pct / 100.0);
            run(k_quad_lds<0>, tab, out, grid, "quad fetch + LDS transposition, plain layout", pct / 100.0);
            run(k_quad_lds<1>, tab, out, grid, "quad fetch + LDS transposition, piece ^ owner", pct / 100.0);
            run(k_quad_lds<2>, tab, out, grid, "quad fetch + LDS transposition, piece ^ quad", pct / 100.0);
        }
    }
    {
        const uint32_t mask = kRecords - 1u, pct = 100u;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c_mask), &mask, 4));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c_percent), &pct, 4));
    }
    // self-check of the transposes: every variant walks the same chains
    {
        std::vector<float> r0((size_t)grid * 256), r1(r0.size());
        hipLaunchKernelGGL(k_lane<0>, dim3(grid), dim3(256), 0, 0, tab, out);
        CK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost));
        for (int v = 0; v < 6; ++v) {
            if (v == 0) hipLaunchKernelGGL(k_quad<3>, dim3(grid), dim3(256), 0, 0, tab, out);
            else if (v == 1) hipLaunchKernelGGL(k_quad<6>, dim3(grid), dim3(256), 0, 0, tab, out);
            else if (v == 2) hipLaunchKernelGGL(k_quad_uncond, dim3(grid), dim3(256), 0, 0, tab, out);
            else if (v == 3) hipLaunchKernelGGL(k_quad_lds<0>, dim3(grid), dim3(256), 0, 0, tab, out);
            else if (v == 4) hipLaunchKernelGGL(k_quad_lds<1>, dim3(grid), dim3(256), 0, 0, tab, out);
            else hipLaunchKernelGGL(k_quad_lds<2>, dim3(grid), dim3(256), 0, 0, tab, out);
            CK(hipMemcpy(r1.data(), out, r1.size() * 4, hipMemcpyDeviceToHost));
            size_t bad = 0;
            for (size_t i = 0; i < r0.size(); ++i) bad += r0[i] != r1[i];
            printf("check variant %d vs v0: %zu mismatches\n", v, bad);
        }
    }
    return 0;
}
