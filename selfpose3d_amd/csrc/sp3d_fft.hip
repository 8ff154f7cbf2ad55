// Batched 3-D real transforms for the frequency-domain 7x7x7 opening conv of V2V-Net (v2v_net.py:10-20 of the reference
// is the conv they replace; see v2v_net._front_fft).  Thin layer over hipFFT/rocFFT plans, kept behind the C ABI so the
// caller controls buffers: torch.fft clones its operand before every real transform on ROCm (the library may overwrite
// C2R inputs) - three ~55 MB copies per root-net step that the plan here does not need, because the padded input
// buffer is only read by the R2C plan and the spectrum is scratch that the C2R plan is allowed to destroy.
//
// Plans are cached per (device, direction, batch, SX, SY, SZ); the first call of a shape builds the plan (and its work
// buffer), so it must not happen inside a stream capture - later calls only enqueue kernels on `stream`.
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <tuple>

#include "../../include/sp3d.h"

namespace {
using Key = std::tuple<int, int, int, int, int, int>;
std::mutex g_mu;
std::map<Key, hipfftHandle> g_plans;

int get_plan(int inverse, int batch, int SX, int SY, int SZ, hipfftHandle *out)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    const Key k{dev, inverse, batch, SX, SY, SZ};
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(k);
    if (it != g_plans.end()) {
        *out = it->second;
        return SP3D_OK;
    }
    hipfftHandle p;
    int n[3] = {SX, SY, SZ};
    // dense default layouts: real (batch,SX,SY,SZ), complex (batch,SX,SY,SZ/2+1)
    if (hipfftPlanMany(&p, 3, n, nullptr, 1, 0, nullptr, 1, 0, inverse ? HIPFFT_C2R : HIPFFT_R2C, batch) != HIPFFT_SUCCESS)
        return SP3D_EFFT;
    g_plans.emplace(k, p);
    *out = p;
    return SP3D_OK;
}

int check_args(const void *a, const void *b, int batch, int SX, int SY, int SZ)
{
    if (batch <= 0 || SX <= 0 || SY <= 0 || SZ <= 0) return SP3D_EINVAL;
    if (!a || !b) return SP3D_ENULL;
    if ((int64_t)batch * SX * SY * SZ > (int64_t)1 << 40) return SP3D_ERANGE;
    return SP3D_OK;
}
} // namespace

extern "C" int sp3d_rfft3d(const float *in, float *out, int batch, int SX, int SY, int SZ, void *stream)
{
    int rc = check_args(in, out, batch, SX, SY, SZ);
    if (rc) return rc;
    hipfftHandle p;
    if ((rc = get_plan(0, batch, SX, SY, SZ, &p))) return rc;
    std::lock_guard<std::mutex> lock(g_mu);              // SetStream + Exec of one plan must not interleave
    if (hipfftSetStream(p, (hipStream_t)stream) != HIPFFT_SUCCESS) return SP3D_EFFT;
    if (hipfftExecR2C(p, const_cast<float *>(in), reinterpret_cast<hipfftComplex *>(out)) != HIPFFT_SUCCESS) return SP3D_EFFT;
    return SP3D_OK;
}

extern "C" int sp3d_irfft3d(float *in, float *out, int batch, int SX, int SY, int SZ, void *stream)
{
    int rc = check_args(in, out, batch, SX, SY, SZ);
    if (rc) return rc;
    hipfftHandle p;
    if ((rc = get_plan(1, batch, SX, SY, SZ, &p))) return rc;
    std::lock_guard<std::mutex> lock(g_mu);
    if (hipfftSetStream(p, (hipStream_t)stream) != HIPFFT_SUCCESS) return SP3D_EFFT;
    if (hipfftExecC2R(p, reinterpret_cast<hipfftComplex *>(in), out) != HIPFFT_SUCCESS) return SP3D_EFFT;
    return SP3D_OK;
}
