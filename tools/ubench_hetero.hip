// Development: MFMA-only waves beside vector-only waves on the same SIMD (the case MI355X_MICROARCH.md "Wave scheduling" describes), against
// waves that each run both kinds of instruction, with the clock read inside the kernel so that cycles and GHz are told apart (the chip
// lowers its clock under load: a variant that saves cycles can give the saving back as clock).
//
// A workgroup of 256 * W threads, one per CU: waves go to the SIMDs cyclically, so wave w and wave w + 4 share a SIMD.  Roles by wave:
//     waves 0..4*NM-1   MFMA role    4 v_mfma_f32_32x32x16_f16 per iteration (two accumulators, random operands)
//     the others        vector role  64 vector instructions per iteration (kinds below)
//     BOTH (NM < 0)     every wave runs 64 vector instructions with one MFMA after every 16 (ubench_overlap.hip's loop)
// Every wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around its loop; the host prints, per role, the median over waves
// of cycles per iteration, ns per iteration and the clock the wave saw.  The role under measurement runs `iters` iterations, the other role
// LONG iterations so that it is running for the whole of the measured role's loop.
//
//     hipcc --offload-arch=gfx950 -O3 -o ubench_hetero tools/ubench_hetero.hip && ./ubench_hetero
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp {
    unsigned long long cyc, rt;
    int role, iters;
};

// KIND: 0 v_fma_f32, 1 v_fma_mix_f32, 2 the as-coded epilogue's mix per 16: 7 v_fma_f32, 4 v_fma_mix_f32, 2 v_cvt_pk_f16_f32, 2 v_sub_f32, 1 v_exp_f32
template <int KIND>
__device__ __forceinline__ void vec16(float (&v)[32], float av, float bv, int base) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float &x = v[(base + i) & 31];
        const float y = v[(base + i + 5) & 31], z = v[(base + i + 11) & 31];
        int k = KIND;
        if (KIND == 2) k = i < 7 ? 0 : (i < 11 ? 1 : (i < 13 ? 3 : (i < 15 ? 4 : 5)));
        if (k == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(y), "v"(z));
        if (k == 1) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[0,1,0]" : "+v"(x) : "v"(av), "v"(bv));
        if (k == 3) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(av));
        if (k == 4) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(y));
        if (k == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    }
}

template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, Stamp *st, int nm, int iters_m, int iters_v, const float *rnd) {
    f32x16 acc[2];
    float v[32];
    f16x8 a, b;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        a[c] = (_Float16)rnd[(threadIdx.x * 8 + c) & 4095];
        b[c] = (_Float16)rnd[(threadIdx.x * 8 + c + 2048) & 4095];
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = rnd[(threadIdx.x + 64 * i) & 4095];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = rnd[(threadIdx.x + r) & 4095];
    float av = 1.0f + rnd[threadIdx.x & 4095] * 1e-3f, bv = rnd[(threadIdx.x + 7) & 4095] * 1e-3f;
    asm volatile("" : "+v"(av), "+v"(bv), "+v"(a), "+v"(b));
    const int wave = threadIdx.x >> 6;
    const int role = nm < 0 ? 2 : (wave < 4 * nm ? 0 : 1);
    const int iters = role == 0 ? iters_m : iters_v;
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (role == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i & 1]) : "v"(a), "v"(b));
        }
    } else if (role == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) vec16<KIND>(v, av, bv, 16 * q);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[q & 1]) : "v"(a), "v"(b));
                vec16<KIND>(v, av, bv, 16 * q);
            }
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += v[i];
    for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        Stamp t;
        t.cyc = c1 - c0;
        t.rt = r1 - r0;
        t.role = role;
        t.iters = iters;
        st[blockIdx.x * (blockDim.x >> 6) + wave] = t;
    }
}

static int g_blocks = 256;
struct Res {
    double cyc, ns, ghz;
};
template <int KIND>
static void run(float *out, Stamp *st, const float *rnd, int waves_per_simd, int nm, int iters_m, int iters_v, Res (&res)[3]) {
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL((k<KIND>), dim3(g_blocks), dim3(threads), 0, 0, out, st, nm, iters_m / 50 + 1, iters_v / 50 + 1, rnd);   // warm the clock state
    hipLaunchKernelGGL((k<KIND>), dim3(g_blocks), dim3(threads), 0, 0, out, st, nm, iters_m, iters_v, rnd);
    (void)hipDeviceSynchronize();
    std::vector<Stamp> h(g_blocks * 4 * waves_per_simd);
    (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
    for (int role = 0; role < 3; ++role) {
        std::vector<double> c, n;
        for (auto &t : h)
            if (t.role == role) {
                c.push_back((double)t.cyc / t.iters);
                n.push_back((double)t.rt * 10.0 / t.iters);
            }
        res[role] = {0, 0, 0};
        if (c.empty()) continue;
        std::sort(c.begin(), c.end());
        std::sort(n.begin(), n.end());
        res[role].cyc = c[c.size() / 2];
        res[role].ns = n[n.size() / 2];
        res[role].ghz = res[role].cyc / res[role].ns;
    }
}

template <int KIND>
static void table(float *out, Stamp *st, const float *rnd, const char *name) {
    const int IT = 20000, LONG = 8 * IT;
    Res r[3];
    printf("\n%s: per wave-iteration (vector role: 64 vector instructions; MFMA role: 4 MFMAs; both: 64 + 4), median over waves\n", name);
    printf("  %-52s %10s %9s %6s   %10s %9s %6s\n", "waves per SIMD", "vec cyc", "vec ns", "GHz", "mfma cyc", "mfma ns", "GHz");
    for (int kv = 1; kv <= 4; ++kv) {
        run<KIND>(out, st, rnd, kv, 0, IT, IT, r);
        printf("  %d vector-only                                        %10.1f %9.1f %6.2f\n", kv, r[1].cyc, r[1].ns, r[1].ghz);
    }
    for (int km = 1; km <= 2; ++km) {
        run<KIND>(out, st, rnd, km, km, IT, IT, r);
        printf("  %d MFMA-only                                          %10s %9s %6s   %10.1f %9.1f %6.2f\n", km, "", "", "", r[0].cyc, r[0].ns, r[0].ghz);
    }
    for (int kv = 1; kv <= 3; ++kv) {
        Res rv[3], rm[3];
        run<KIND>(out, st, rnd, 1 + kv, 1, LONG, IT, rv);        // vector role measured under a running MFMA wave
        run<KIND>(out, st, rnd, 1 + kv, 1, IT, LONG, rm);        // MFMA role measured beside running vector waves
        printf("  1 MFMA-only + %d vector-only (each beside the other)   %10.1f %9.1f %6.2f   %10.1f %9.1f %6.2f\n", kv, rv[1].cyc, rv[1].ns, rv[1].ghz,
               rm[0].cyc, rm[0].ns, rm[0].ghz);
    }
    for (int w = 1; w <= 4; w *= 2) {
        run<KIND>(out, st, rnd, w, -1, IT, IT, r);
        printf("  %d waves running both (1 MFMA per 16 vector)           %10.1f %9.1f %6.2f   (cycles for 64 vector + 4 MFMA)\n", w, r[2].cyc, r[2].ns, r[2].ghz);
    }
}

int main(int argc, char **argv) {
    if (argc > 1) g_blocks = atoi(argv[1]);
    float *out, *rnd, h[4096];
    Stamp *st;
    srand(1);
    for (int i = 0; i < 4096; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&st, 256 * 16 * sizeof(Stamp));
    (void)hipMalloc(&rnd, sizeof(h));
    (void)hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
    printf("%d workgroups (one per CU); throughput of a SIMD = (waves of the role) x (instructions per iteration) / (cycles per iteration)\n", g_blocks);
    table<0>(out, st, rnd, "v_fma_f32");
    table<1>(out, st, rnd, "v_fma_mix_f32");
    table<2>(out, st, rnd, "as-coded epilogue mix (7 fma, 4 fma_mix, 2 cvt_pk, 2 sub, 1 exp per 16)");
    return 0;
}
