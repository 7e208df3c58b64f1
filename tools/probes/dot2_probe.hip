// What does v_dot2c_f32_bf16 / v_dot2c_f32_f16 compute on gfx950?  (round 4: the AR decode step's products)
// hipcc --offload-arch=gfx950 -O2 tools/probes/dot2_probe.hip -o tools/probes/dot2_probe && tools/probes/dot2_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
typedef __attribute__((ext_vector_type(2))) _Float16 h2;

__global__ void k_bf16(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_fdot2_f32_bf16(*(const bf2*)&a[i], *(const bf2*)&b[i], c[i], false);
}
__global__ void k_f16(const uint32_t* a, const uint32_t* b, const float* c, float* o, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) o[i] = __builtin_amdgcn_fdot2(*(const h2*)&a[i], *(const h2*)&b[i], c[i], false);
}
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float h2f(uint16_t h) { _Float16 x; memcpy(&x, &h, 2); return (float)x; }
static uint16_t f2h(float f) { _Float16 x = (_Float16)f; uint16_t h; memcpy(&h, &x, 2); return h; }

int main() {
    const int n = 1 << 16;
    std::vector<uint32_t> a(n), b(n);
    std::vector<float> c(n), o(n);
    uint32_t *da, *db; float *dc, *dout;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, n * 4);
    for (int mode = 0; mode < 2; ++mode) {
        srand(1);
        auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
        for (int i = 0; i < n; ++i) {
            float a0 = rnd() * 3, a1 = rnd() * 3, b0 = rnd(), b1 = rnd();
            uint16_t qa0 = mode ? f2h(a0) : f2bf(a0), qa1 = mode ? f2h(a1) : f2bf(a1), qb0 = mode ? f2h(b0) : f2bf(b0), qb1 = mode ? f2h(b1) : f2bf(b1);
            a[i] = qa0 | ((uint32_t)qa1 << 16); b[i] = qb0 | ((uint32_t)qb1 << 16);
            c[i] = (i & 1) ? rnd() * 10 : 0.f;
        }
        hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
        if (mode == 0) hipLaunchKernelGGL(k_bf16, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
        else hipLaunchKernelGGL(k_f16, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
        hipMemcpy(o.data(), dout, n * 4, hipMemcpyDeviceToHost);
        double worst = 0; int n_exact_fma_chain = 0, n_exact_sum_first = 0, n_lo_hi_swapped = 0;
        for (int i = 0; i < n; ++i) {
            auto cv = [&](uint16_t h) { return mode ? h2f(h) : bf2f(h); };
            float a0 = cv(a[i] & 0xffff), a1 = cv(a[i] >> 16), b0 = cv(b[i] & 0xffff), b1 = cv(b[i] >> 16);
            double ref = (double)a0 * b0 + (double)a1 * b1 + c[i];
            double err = fabs(o[i] - ref) / (fabs(ref) + 1e-3);
            if (err > worst) worst = err;
            float chain = fmaf(a1, b1, fmaf(a0, b0, c[i]));                    // the old code's order for one pair
            float sumfirst = (float)((double)a0 * b0 + (double)a1 * b1) + c[i];
            n_exact_fma_chain += (o[i] == chain);
            n_exact_sum_first += (o[i] == sumfirst);
            n_lo_hi_swapped += (o[i] == (float)((double)a0 * b1 + (double)a1 * b0 + c[i]));
        }
        printf("%s: max rel err vs double %.3e; == fmaf chain (lo then hi) in %d / %d; == round(a0b0 + a1b1) + c in %d; looks lo/hi-crossed in %d\n",
               mode ? "v_dot2c_f32_f16" : "v_dot2c_f32_bf16", worst, n_exact_fma_chain, n, n_exact_sum_first, n_lo_hi_swapped);
        for (int i = 0; i < 3; ++i) printf("   sample a=%08x b=%08x c=%g -> %.9g\n", a[i], b[i], c[i], o[i]);
    }
    return 0;
}
