// Minimal C++ client of the C ABI (include/scasml_hip.h): no Python, no torch -- the drop-in boundary a
// non-Python host would bind.  Solves MLP_full_history (solvers/MLP_full_history.py:64-196) at n = 2, M = 3 for
// B points read from stdin-free synthetic input and prints u (first column) one value per line.
//
//   hipcc -I include tests/abi_client.cpp -L scasml_gp_amd/lib -lscasml_hip -Wl,-rpath,$PWD/scasml_gp_amd/lib -o abi_client
//   ./abi_client <d> <B> <seed>
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "scasml_hip.h"

int main(int argc, char **argv) {
    const int d = argc > 1 ? atoi(argv[1]) : 10;
    const int B = argc > 2 ? atoi(argv[2]) : 8;
    const unsigned long long seed = argc > 3 ? strtoull(argv[3], nullptr, 10) : 0;
    if (scasml_abi_version() != SCASML_ABI_VERSION) return 2;

    // static schedule of the full-history recursion, n = 2, M = 3: MC_g = M^n, MC_f = M^(n-l), one random time per sample
    scasml_plan plan;
    memset(&plan, 0, sizeof plan);
    plan.variant = 1;
    plan.n = 2;
    const int M = 3;
    int sites[3] = {0, 0, 0};
    for (int n = 1; n <= 2; ++n) {
        int mg = 1;
        for (int i = 0; i < n; ++i) mg *= M;
        plan.mg[n] = mg;
        int s = mg;
        for (int l = 0; l < n; ++l) {
            int mc = 1;
            for (int i = 0; i < n - l; ++i) mc *= M;
            scasml_term &t = plan.term[n][l];
            t.q = 1;
            t.mc = mc;
            t.sites_l = sites[l];
            t.sites_lm1 = l ? sites[l - 1] : 0;
            s += mc * (1 + t.sites_l + t.sites_lm1);
        }
        sites[n] = s;
        plan.sites[n] = s;
    }
    const float sigma = 0.25f;
    scasml_problem prob = {d, SCASML_EQ_GRAD_DEPENDENT_NONLINEAR, 0.5f, -1.0f / d - sigma * sigma / 2, sigma, 1.0f};
    scasml_rng rng = {seed, 0u, 0u, 0, 1};

    std::vector<float> x((size_t)B * (d + 1));
    for (int b = 0; b < B; ++b) {                       // deterministic synthetic points in the reference's box
        for (int k = 0; k < d; ++k) x[(size_t)b * (d + 1) + k] = -0.5f + (float)((b * 131 + k * 17) % 97) / 96.0f;
        x[(size_t)b * (d + 1) + d] = 0.5f * (float)((b * 29) % 50) / 50.0f;
    }
    float *dx = nullptr, *dout = nullptr;
    if (hipMalloc(&dx, x.size() * sizeof(float)) != hipSuccess || hipMalloc(&dout, x.size() * sizeof(float)) != hipSuccess) return 3;
    hipMemcpy(dx, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice);
    const int rc = scasml_picard_tree(&prob, &plan, SCASML_MODE_MLP, dx, B, 0, rng, nullptr, nullptr, dout, nullptr, nullptr);
    if (rc != 0) {
        fprintf(stderr, "scasml_picard_tree: %d %s\n", rc, scasml_last_error());
        return 4;
    }
    std::vector<float> out(x.size());
    if (hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return 5;
    for (int b = 0; b < B; ++b) printf("%.9g\n", out[(size_t)b * (d + 1)]);
    hipFree(dx);
    hipFree(dout);
    return 0;
}
