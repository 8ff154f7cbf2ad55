// issue_bench.hip - what one SIMD of gfx950 issues per cycle, by instruction class and by how many waves share it.
//
// The unprojection kernels' "speed of light" (profiles/r03_speed_of_light.md) rested on an INFERRED model: every
// instruction of any class costs its SIMD ~4 cycles and classes of different waves do not overlap.  This measures it:
// independent streams of one instruction class per wave (v_fma_f32, v_pk_fma_f32, v_mov, s_add_u32, s_nop, s_waitcnt,
// ds_read_b32, global_load_dwordx4 on one hot line), 1 / 2 / 4 / 8 waves per SIMD, and MIXED placements: waves that run
// only VALU next to waves that run only SALU on the same SIMD, and both classes interleaved inside one wave.
//
//   hipcc --offload-arch=gfx950 -O2 tools/issue_bench.hip -o /tmp/issue_bench && /tmp/issue_bench [--md]
//
// Timing is per wave (s_memtime = shader clock, bracketed by the kernel's wall clock for the MHz figure); a wave's SIMD
// comes from HW_ID, so "cycles per instruction per SIMD" = wave cycles / (instructions issued by ALL waves resident on
// that SIMD during the loop).  One workgroup per CU (96 KiB of LDS each) so residency is exactly what is asked for.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

enum Kind { K_FMA = 0, K_PKFMA, K_MOV, K_SALU, K_SNOP, K_WAITCNT, K_FMA_DEP, K_PKFMA_DEP, K_MIX_VS, K_MIX_VSN, K_LDS, K_VMEM, K_PKMUL, K_NKINDS };
static const char *kind_name[] = {"v_fma_f32 (8 independent)", "v_pk_fma_f32 (8 independent)", "v_mov_b32", "s_add_u32 (8 independent)",
                                  "s_nop 0", "s_waitcnt vmcnt(0) (nothing outstanding)", "v_fma_f32 (one dependent chain)",
                                  "v_pk_fma_f32 (one dependent chain)", "v_fma_f32 ; s_add_u32 alternating in ONE wave",
                                  "v_pk_fma_f32 ; s_nop 0 ; s_add_u32 in ONE wave", "ds_read_b32", "global_load_dwordx4 (one hot 1 KiB)",
                                  "v_pk_mul_f32 (8 independent)"};
// instructions per REP64 body, by kind
static int body_insts(int k) { return (k == K_MIX_VS) ? 128 : (k == K_MIX_VSN) ? 192 : 64; }

struct Stamp { unsigned long long t0, t1, w0, w1; unsigned hw, kind, xcc, pad; };

__device__ __forceinline__ unsigned long long wall() { return wall_clock64(); }

template <int KIND>
__device__ __forceinline__ void stream(int iters, float &sink, const float4 *hot, float *lds)
{
    float a0 = sink, a1 = sink + 1, a2 = sink + 2, a3 = sink + 3, a4 = sink + 4, a5 = sink + 5, a6 = sink + 6, a7 = sink + 7;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f; const v2 pm = {0.999f, 1.001f};
    float4 ld = make_float4(0, 0, 0, 0);
    const unsigned laddr = (unsigned)(threadIdx.x & 63) * 4u;
    (void)lds;
    for (int it = 0; it < iters; ++it) {
        if (KIND == K_FMA) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == K_PKFMA) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                              "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
        } else if (KIND == K_PKMUL) {
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
        } else if (KIND == K_MOV) {
            REP8(asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
                              "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));)
        } else if (KIND == K_SALU) {
            REP8(asm volatile("s_add_u32 s40, s40, 1\n s_add_u32 s41, s41, 1\n s_add_u32 s42, s42, 1\n s_add_u32 s43, s43, 1\n"
                              "s_add_u32 s44, s44, 1\n s_add_u32 s45, s45, 1\n s_add_u32 s46, s46, 1\n s_add_u32 s47, s47, 1\n"
                              ::: "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        } else if (KIND == K_SNOP) {
            REP64(asm volatile("s_nop 0\n");)
        } else if (KIND == K_WAITCNT) {
            REP64(asm volatile("s_waitcnt vmcnt(0)\n");)
        } else if (KIND == K_FMA_DEP) {
            REP64(asm volatile("v_fma_f32 %0, %0, %1, %0\n" : "+v"(a0) : "v"(m));)
        } else if (KIND == K_PKFMA_DEP) {
            REP64(asm volatile("v_pk_fma_f32 %0, %0, %1, %0\n" : "+v"(p0) : "v"(pm));)
        } else if (KIND == K_MIX_VS) {
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %0\n s_add_u32 s40, s40, 1\n v_fma_f32 %1, %1, %8, %1\n s_add_u32 s41, s41, 1\n"
                              "v_fma_f32 %2, %2, %8, %2\n s_add_u32 s42, s42, 1\n v_fma_f32 %3, %3, %8, %3\n s_add_u32 s43, s43, 1\n"
                              "v_fma_f32 %4, %4, %8, %4\n s_add_u32 s44, s44, 1\n v_fma_f32 %5, %5, %8, %5\n s_add_u32 s45, s45, 1\n"
                              "v_fma_f32 %6, %6, %8, %6\n s_add_u32 s46, s46, 1\n v_fma_f32 %7, %7, %8, %7\n s_add_u32 s47, s47, 1\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m)
                              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        } else if (KIND == K_MIX_VSN) {
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n s_nop 0\n s_add_u32 s40, s40, 1\n v_pk_fma_f32 %1, %1, %8, %1\n s_nop 0\n s_add_u32 s41, s41, 1\n"
                              "v_pk_fma_f32 %2, %2, %8, %2\n s_nop 0\n s_add_u32 s42, s42, 1\n v_pk_fma_f32 %3, %3, %8, %3\n s_nop 0\n s_add_u32 s43, s43, 1\n"
                              "v_pk_fma_f32 %4, %4, %8, %4\n s_nop 0\n s_add_u32 s44, s44, 1\n v_pk_fma_f32 %5, %5, %8, %5\n s_nop 0\n s_add_u32 s45, s45, 1\n"
                              "v_pk_fma_f32 %6, %6, %8, %6\n s_nop 0\n s_add_u32 s46, s46, 1\n v_pk_fma_f32 %7, %7, %8, %7\n s_nop 0\n s_add_u32 s47, s47, 1\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm)
                              : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "scc");)
        } else if (KIND == K_LDS) {
            REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n"
                              "ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(laddr) : "memory");)
        } else if (KIND == K_VMEM) {
            const float4 *p = hot + (threadIdx.x & 63);
            REP8(asm volatile("global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n"
                              "global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n global_load_dwordx4 %0, %1, off\n"
                              "s_waitcnt vmcnt(0)\n"
                              : "=&v"(ld) : "v"(p) : "memory");)
        }
    }
    sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + ld.x;
}

// waves whose (wave-in-workgroup / 4) is even run KA, the others KB (wave w sits on SIMD w % 4: checked against HW_ID)
template <int KA, int KB>
__global__ __launch_bounds__(1024) void bench_kernel(Stamp *st, int iters, float *out, const float4 *hot, int lds_floats)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (float)i;
    float sink = (float)threadIdx.x * 1e-3f;
    __syncthreads();
    const bool second = ((wave >> 2) & 1) != 0;
    const unsigned long long w0 = wall();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (!second) stream<KA>(iters, sink, hot, lds); else stream<KB>(iters, sink, hot, lds);
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall();
    if (lane == 0) {
        Stamp s; s.t0 = t0; s.t1 = t1; s.w0 = w0; s.w1 = w1; s.hw = __builtin_amdgcn_s_getreg(63492); s.kind = second ? KB : KA; s.xcc = __builtin_amdgcn_s_getreg(63508); s.pad = 0;
        st[blockIdx.x * (blockDim.x >> 6) + wave] = s;
    }
    if (sink == 123.456f) out[0] = sink + (float)lds_floats;
}

struct Row { std::string name; int wps; double cpi_simd; double cpi_wave; double mhz; std::string note; double resident; };
static std::vector<Row> rows;

template <int KA, int KB>
static void run(const char *label, int waves_per_simd, int iters = 200)
{
    // waves_per_simd 1/2/4: one workgroup of 256/512/1024 threads per CU (96 KiB LDS forbids a second one);
    // 8: two workgroups of 1024 threads per CU (64 KiB each)
    const int threads = waves_per_simd >= 4 ? 1024 : 256 * waves_per_simd;
    const int wg_per_cu = waves_per_simd == 8 ? 2 : 1;
    const int blocks = 256 * wg_per_cu;
    const size_t lds = waves_per_simd == 8 ? 64 * 1024 : 96 * 1024;
    const int nw = blocks * threads / 64;
    Stamp *st; hipMalloc(&st, sizeof(Stamp) * nw);
    float *out; hipMalloc(&out, 4);
    float4 *hot; hipMalloc(&hot, 4096); hipMemset(hot, 0, 4096);
    hipFuncSetAttribute((const void *)bench_kernel<KA, KB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((bench_kernel<KA, KB>), dim3(blocks), dim3(threads), lds, 0, st, rep == 0 ? 4 : iters, out, hot, (int)(lds / 4));
        hipDeviceSynchronize();
    }
    std::vector<Stamp> h(nw);
    hipMemcpy(h.data(), st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost);
    // per wave: cycles per own instruction; per SIMD: cycles / instructions of all waves on it (they run concurrently)
    std::vector<double> cw[2];
    double mhz = 0; int nm = 0;
    for (auto &s : h) {
        const double cyc = (double)(s.t1 - s.t0);
        const int k = (int)s.kind;
        const double n = (double)iters * body_insts(k);
        cw[k == KA ? 0 : 1].push_back(cyc / n);
        const double wallticks = (double)(s.w1 - s.w0);       // 100 MHz
        if (wallticks > 100) { mhz += cyc / (wallticks / 100.0); ++nm; }
    }
    // how many waves really shared a SIMD: per (XCC, SE, SH, CU, SIMD) the waves whose wall-clock interval covers the middle
    // of the first-listed wave's interval (two 1024-thread workgroups per CU are asked for at 8 waves / SIMD: did both fit?)
    double resident = 0; int nres = 0;
    {
        std::vector<std::pair<unsigned long long, size_t>> key;
        for (size_t i = 0; i < h.size(); ++i) {
            const unsigned hw = h[i].hw;    // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
            const unsigned long long k = ((unsigned long long)(h[i].xcc & 0xf) << 32) | ((hw >> 8) & 0xff) << 8 | ((hw >> 4) & 3);
            key.push_back({k, i});
        }
        std::sort(key.begin(), key.end());
        for (size_t a = 0; a < key.size();) {
            size_t b = a;
            while (b < key.size() && key[b].first == key[a].first) ++b;
            const Stamp &r = h[key[a].second];
            const unsigned long long mid = (r.w0 + r.w1) / 2;
            int c = 0;
            for (size_t j = a; j < b; ++j) { const Stamp &q = h[key[j].second]; if (q.w0 <= mid && q.w1 >= mid) ++c; }
            resident += c; ++nres;
            a = b;
        }
    }
    auto med = [](std::vector<double> &v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    const double ca = med(cw[0]), cb = med(cw[1]);
    Row r; r.name = label; r.wps = waves_per_simd; r.mhz = nm ? mhz / nm : 0; r.resident = nres ? resident / nres : 0;
    const double wres = r.resident > 0 ? r.resident : waves_per_simd;      // divide by what was really co-resident
    char note[256];
    if (KA == KB) {
        r.cpi_wave = ca; r.cpi_simd = ca / wres;
        snprintf(note, sizeof note, "%s", "");
    } else {
        // half the waves of a SIMD run KA, half KB, concurrently
        const double half = wres / 2;
        r.cpi_wave = ca; r.cpi_simd = 0;
        snprintf(note, sizeof note, "A-waves %.2f cyc/inst each (=> %.2f per SIMD for the A class), B-waves %.2f (=> %.2f)", ca, ca / half, cb, cb / half);
    }
    r.note = note;
    rows.push_back(r);
    printf("%-58s waves/SIMD asked %d resident %.2f  cyc/inst per wave %6.2f  per SIMD %6.2f  clock %.0f MHz  %s\n", label, waves_per_simd, r.resident, r.cpi_wave, r.cpi_simd, r.mhz, note);
    fflush(stdout);
    hipFree(st); hipFree(out); hipFree(hot);
}

template <int K>
static void sweep()
{
    for (int w : {1, 2, 4, 8}) run<K, K>(kind_name[K], w);
}

int main(int argc, char **argv)
{
    const bool md = argc > 1 && !strcmp(argv[1], "--md");
    sweep<K_FMA>(); sweep<K_PKFMA>(); sweep<K_PKMUL>(); sweep<K_MOV>(); sweep<K_FMA_DEP>(); sweep<K_PKFMA_DEP>();
    sweep<K_SALU>(); sweep<K_SNOP>(); sweep<K_WAITCNT>();
    sweep<K_MIX_VS>(); sweep<K_MIX_VSN>();
    sweep<K_LDS>(); sweep<K_VMEM>();
    // mixed placements: half of a SIMD's waves pure class A, the other half pure class B
    run<K_FMA, K_SALU>("A = v_fma_f32 waves, B = s_add_u32 waves", 2);
    run<K_FMA, K_SALU>("A = v_fma_f32 waves, B = s_add_u32 waves", 4);
    run<K_FMA, K_SALU>("A = v_fma_f32 waves, B = s_add_u32 waves", 8);
    run<K_PKFMA, K_SALU>("A = v_pk_fma_f32 waves, B = s_add_u32 waves", 4);
    run<K_PKFMA, K_SNOP>("A = v_pk_fma_f32 waves, B = s_nop waves", 4);
    run<K_FMA, K_SNOP>("A = v_fma_f32 waves, B = s_nop waves", 4);
    run<K_FMA, K_LDS>("A = v_fma_f32 waves, B = ds_read_b32 waves", 4);
    run<K_FMA, K_VMEM>("A = v_fma_f32 waves, B = global_load_dwordx4 waves", 4);
    run<K_FMA, K_PKFMA>("A = v_fma_f32 waves, B = v_pk_fma_f32 waves", 4);
    run<K_SALU, K_SNOP>("A = s_add_u32 waves, B = s_nop waves", 4);
    if (md) {
        printf("\n| stream | waves / SIMD asked | resident (measured) | cycles per instruction, one wave | cycles per instruction, per SIMD | clock MHz | note |\n|---|---:|---:|---:|---:|---:|---|\n");
        for (auto &r : rows) printf("| %s | %d | %.2f | %.2f | %s | %.0f | %s |\n", r.name.c_str(), r.wps, r.resident, r.cpi_wave,
                                    r.cpi_simd > 0 ? (std::to_string(r.cpi_simd).substr(0, 5)).c_str() : "-", r.mhz, r.note.c_str());
    }
    return 0;
}
