// fft_layout_bench.hip - rocFFT (through hipFFT) throughput of the opening conv's batched 3-D real transforms as a function
// of the data layout: planar volumes (the layout the product uses) against channel-innermost volumes (istride = C,
// idist = 1: what a channels-last unprojection result would hand over without a transpose).
//   hipcc --offload-arch=gfx950 -O3 tools/fft_layout_bench.hip -lhipfft -o gpurun_out/fft_layout_bench
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <cstdio>
#include <vector>

static float time_exec(hipfftHandle p, bool inverse, void *a, void *b, int reps, int execs, size_t in_step, size_t out_step)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&]() {
        for (int i = 0; i < execs; ++i) {
            if (inverse) hipfftExecC2R(p, (hipfftComplex *)((char *)a + i * in_step), (float *)((char *)b + i * out_step));
            else hipfftExecR2C(p, (float *)((char *)a + i * in_step), (hipfftComplex *)((char *)b + i * out_step));
        }
    };
    for (int i = 0; i < 3; ++i) run();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) run();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / reps;
}

int main()
{
    const int B = 4, C = 16, SX = 88, SY = 88, SZ = 28, SZc = SZ / 2 + 1;
    const size_t nreal = (size_t)B * C * SX * SY * SZ, ncplx = (size_t)B * C * SX * SY * SZc;
    float *re; hipfftComplex *cx;
    hipMalloc(&re, nreal * 4); hipMalloc(&cx, ncplx * 8);
    hipMemset(re, 0, nreal * 4); hipMemset(cx, 0, ncplx * 8);
    int n[3] = {SX, SY, SZ};
    {   // planar, one plan over all B*C volumes
        hipfftHandle f, i;
        hipfftPlanMany(&f, 3, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_R2C, B * C);
        hipfftPlanMany(&i, 3, n, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2R, B * C);
        printf("planar           fwd %.1f us   inv %.1f us\n", time_exec(f, false, re, cx, 50, 1, 0, 0), time_exec(i, true, cx, re, 50, 1, 0, 0));
    }
    {   // channel-innermost per sample: element (x,y,z,c) at ((x*SY+y)*SZ+z)*C + c; batch = C with dist 1; B execs
        hipfftHandle f, i;
        int ine[3] = {SX, SY, SZ}, one[3] = {SX, SY, SZc};
        hipfftResult r1 = hipfftPlanMany(&f, 3, n, ine, C, 1, one, C, 1, HIPFFT_R2C, C);
        hipfftResult r2 = hipfftPlanMany(&i, 3, n, one, C, 1, ine, C, 1, HIPFFT_C2R, C);
        if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) printf("channel-innermost plan failed %d %d\n", r1, r2);
        else printf("channel-inner x%d  fwd %.1f us   inv %.1f us\n", B,
                    time_exec(f, false, re, cx, 50, B, (size_t)C * SX * SY * SZ * 4, (size_t)C * SX * SY * SZc * 8),
                    time_exec(i, true, cx, re, 50, B, (size_t)C * SX * SY * SZc * 8, (size_t)C * SX * SY * SZ * 4));
    }
    {   // (b,c) innermost: element (x,y,z,b,c); one exec, batch = B*C with dist 1
        hipfftHandle f, i;
        int ine[3] = {SX, SY, SZ}, one[3] = {SX, SY, SZc};
        hipfftResult r1 = hipfftPlanMany(&f, 3, n, ine, B * C, 1, one, B * C, 1, HIPFFT_R2C, B * C);
        hipfftResult r2 = hipfftPlanMany(&i, 3, n, one, B * C, 1, ine, B * C, 1, HIPFFT_C2R, B * C);
        if (r1 != HIPFFT_SUCCESS || r2 != HIPFFT_SUCCESS) printf("batch-innermost plan failed %d %d\n", r1, r2);
        else printf("batch-innermost   fwd %.1f us   inv %.1f us\n", time_exec(f, false, re, cx, 50, 1, 0, 0), time_exec(i, true, cx, re, 50, 1, 0, 0));
    }
    {   // z transform done elsewhere: dense 2-D complex transforms over (x,y), one per (b,c,kz) / (b,o,kz)
        hipfftHandle f;
        int n2[2] = {SX, SY};
        hipfftComplex *big; hipMalloc(&big, (size_t)B * 32 * SZc * SX * SY * 8); hipMemset(big, 0, (size_t)B * 32 * SZc * SX * SY * 8);
        for (int batch : {B * C * SZc, B * 32 * SZc}) {
            hipfftPlanMany(&f, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2C, batch);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipfftExecC2C(f, big, big, HIPFFT_FORWARD);
            hipDeviceSynchronize(); hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) hipfftExecC2C(f, big, big, HIPFFT_FORWARD);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("2-D C2C 88x88 in place, batch %d: %.1f us\n", batch, ms * 1000.f / 50);
            hipfftDestroy(f);
        }
    }
    {   // which padded x,y length is cheapest for the dense 2-D plan (>= 86 needed)?
        hipfftComplex *big; hipMalloc(&big, (size_t)960 * 128 * 128 * 8); hipMemset(big, 0, (size_t)960 * 128 * 128 * 8);
        for (int L : {86, 88, 90, 96, 98, 100, 104, 108, 112, 120, 128}) {
            hipfftHandle f; int n2[2] = {L, L};
            if (hipfftPlanMany(&f, 2, n2, nullptr, 1, 0, nullptr, 1, 0, HIPFFT_C2C, 960) != HIPFFT_SUCCESS) { printf("plan %d failed\n", L); continue; }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int i = 0; i < 3; ++i) hipfftExecC2C(f, big, big, HIPFFT_FORWARD);
            hipDeviceSynchronize(); hipEventRecord(e0);
            for (int i = 0; i < 50; ++i) hipfftExecC2C(f, big, big, HIPFFT_FORWARD);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("2-D C2C %dx%d batch 960: %.1f us  (%.2f ns per point)\n", L, L, ms * 1000.f / 50, ms * 1e6f / 50 / (960.0f * L * L));
            hipfftDestroy(f);
        }
    }
    return 0;
}
