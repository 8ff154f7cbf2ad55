/*
 * sp3d.h - C ABI of libsp3d.so: the MI355X (gfx950) implementation of SelfPose3d's
 * multi-view heat-map -> voxel unprojection hot path.
 *
 * The reference has no FFI / operator registry: its seam is the Python module
 * `ProjectLayer` (/root/reference/lib/models/project_layer.py:15-106), constructed at
 * lib/models/cuboid_proposal_net.py:95, cuboid_proposal_net_soft.py:83,
 * pose_regression_net.py:37.  These entry points are what a binding for that seam (and for
 * the two small reductions either side of it) calls; `selfpose3d_amd/project_layer.py` is
 * the ctypes binding, INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer except `hm_views` (a HOST array of V
 *     device pointers) and `grid_size` (host) is a DEVICE pointer valid on `stream`'s device;
 *   - the library allocates nothing and never synchronises the host.  It keeps no state that selects behaviour or
 *     changes a result: every choice is a per-call argument.  Its only writable statics are three caches, each a pure
 *     function of (device, shape) - the hipFFT plan cache of the frequency-domain convolution (under a mutex), the
 *     per-device CU count, and the per-device "dynamic-LDS attribute already raised" flags (tests/test_host_cabi.py
 *     lists the library's writable symbols and fails on any other);
 *     work is enqueued on `stream` (a hipStream_t, NULL = default stream) and is
 *     HIP-graph capturable;
 *   - every output buffer is fully overwritten (no pre-zeroing needed) unless noted;
 *   - return 0 on success, a negative SP3D_E* for bad arguments, a positive hipError_t if a
 *     launch failed.  Nothing throws across this boundary.
 */
#ifndef SP3D_H
#define SP3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history: 3 = sp3d_unproject_bwd_packed[_det] take `int scatter` before `stream`, sp3d_set_bwd_scatter removed (round 5;
 * the number was bumped in round 6 - a caller built against version 2 must fail sp3d_abi_version(), not pass a stream as `scatter`) */
#define SP3D_ABI_VERSION 3
#define SP3D_MAX_VIEWS 16
#define SP3D_MAX_TOPK 32

/* packed per-(sample, view) camera record: fp32[SP3D_CAM_STRIDE], table shape (B, V, 64).
 * Built on the host by selfpose3d_amd/camera_pack.py::pack_cameras (one upload per call,
 * replacing lib/models/project_layer.py:64-75 + lib/utils/cameras.py:13-24 +
 * lib/utils/transforms.py:61-103 executed per (sample, view) in the reference).
 * Fields 0..29 are the camera; fields 32..62 are DERIVED from them (sp3d_camera_finish / camera_pack.finish: call it
 * again after editing a record by hand): what the packed-fp32 projection reads, laid out for it - operand pairs as
 * 8-byte aligned neighbours (a scalar-register pair feeds v_pk_*_f32 directly: 13 scalar moves per view and wave less),
 * the result of the per-view affine sanity test (12 scalar instructions less), all in two 64-byte scalar loads. */
#define SP3D_CAM_STRIDE 64
#define SP3D_CAM_R 0      /* [9]  world->camera rotation, row-major            cameras.py:14 */
#define SP3D_CAM_T 9      /* [3]  camera centre in world, mm  (Xc = R (X - T)) cameras.py:15 */
#define SP3D_CAM_F 12     /* [2]  fx, fy                                      cameras.py:16-18 */
#define SP3D_CAM_C 14     /* [2]  cx, cy                                      cameras.py:19-21 */
#define SP3D_CAM_K 16     /* [3]  radial k0,k1,k2                             cameras.py:22 */
#define SP3D_CAM_P 19     /* [2]  tangential p0,p1                            cameras.py:23 */
#define SP3D_CAM_A 21     /* [6]  2x3 crop affine, original image -> network input, row-major
                                  (fp32 cast of get_affine_transform, project_layer.py:69-72) */
#define SP3D_CAM_W0 27    /* original image width  = 2*center.x  (project_layer.py:68) */
#define SP3D_CAM_H0 28    /* original image height = 2*center.y */
#define SP3D_CAM_FLIP 29  /* 1.0 if flip_xcoords[b] else 0.0      (project_layer.py:82) */
/* derived block, floats 32..62: everything the packed projection reads, in the order it reads it */
#define SP3D_CAM_RXY 32   /* [6]  (R00,R10) (R01,R11) (R02,R12) - the x and y rows of R, column by column */
#define SP3D_CAM_TXY 38   /* [2]  T0, T1 */
#define SP3D_CAM_RZ 40    /* [3]  R20, R21, R22 */
#define SP3D_CAM_TZ 43    /*      T2 */
#define SP3D_CAM_K2 44    /* [3]  k0, k1, k2 */
#define SP3D_CAM_TAME 47  /*      1.0 if all six |A| <= 1e30 (finite and safe to multiply), else 0.0 */
#define SP3D_CAM_P2 48    /* [2]  p0, p1 */
#define SP3D_CAM_F2 50    /* [2]  fx, fy */
#define SP3D_CAM_C2 52    /* [2]  cx, cy */
#define SP3D_CAM_WH 54    /* [2]  W0, H0 */
#define SP3D_CAM_AXY 56   /* [6]  (A00,A10) (A01,A11) (A02,A12) - the affine, column by column */
#define SP3D_CAM_FLIP2 62 /*      flip */

enum {
    SP3D_OK = 0,
    SP3D_EINVAL = -1,        /* a dimension <= 0, V > SP3D_MAX_VIEWS, unknown layout ...  */
    SP3D_ENULL = -2,         /* a required pointer is NULL                                */
    SP3D_ERANGE = -3,        /* X*Y*Z, B*J*N or the launch grid overflows 32-bit limits   */
    SP3D_EUNSUPPORTED = -4,  /* combination not implemented (e.g. Jp not a multiple of 4) */
    SP3D_EFFT = -5           /* hipFFT refused to build or run a plan                      */
};

/* heat-map layouts accepted by the unprojection kernels (per-view pointers in both) */
enum {
    SP3D_LAYOUT_PLANAR = 0,  /* view c: (B, J, h, w) fp32 - the reference's layout (pose_resnet.py:203) */
    SP3D_LAYOUT_NHWC = 1,    /* view c: (B, h, w, Jp) fp32, Jp%4==0, channels >= J are ignored padding */
    /* OR-ed into hm_layout: write `cubes` channels-last, (B, X, Y, Z, J) with J%4==0 (NHWC input
     * only) - the layout MIOpen's 3D convolutions consume without an internal transpose. */
    SP3D_OUT_CHANNELS_LAST = 0x100,
    /* OR-ed into hm_layout (NHWC, Jp == 16): the packed heat-maps / the cubes are stored as bf16
     * (BASELINE configs[4] "mixed bf16": storage only - projection, interpolation and view fusion
     * stay fp32; cubes are rounded to nearest-even on the final store). */
    SP3D_HM_BF16 = 0x200,
    SP3D_OUT_BF16 = 0x400
};

int sp3d_abi_version(void);

/* Fill the derived fields (SP3D_CAM_RXY .. SP3D_CAM_FLIP2) of `records` camera records in HOST memory from their fields 0..29.
 * Pure host arithmetic (copies and one comparison), no device, no stream. */
int sp3d_camera_finish(float *table_host, int records);
const char *sp3d_error_string(int code);

/*
 * Re-tile V planar heat-maps (B,J,h,w) into ONE channels-last buffer packed[(V,B,h,w,Jp)],
 * channels J..Jp-1 zero-filled, so that one bilinear tap is a single Jp*4-byte read.
 * (Input side of ProjectLayer.get_voxel: the `heatmaps` list, project_layer.py:42-44.)
 */
int sp3d_pack_heatmaps(const float *const *hm_views, float *packed, int B, int V, int J, int Jp, int h, int w,
                       void *stream);
/* same with explicit storage types: in_bf16 / out_bf16 = 1 when the planar inputs / the packed output
 * hold bf16 instead of fp32 (bf16 needs Jp == 16) */
int sp3d_pack_heatmaps_ex(const void *const *hm_views, void *packed, int in_bf16, int out_bf16, int B, int V, int J,
                          int Jp, int h, int w, void *stream);

/*
 * ProjectLayer.get_voxel forward (project_layer.py:42-102; math: DESIGN.md §3).
 *   hm_views  HOST array of V device pointers, layout per `hm_layout` (Jp used for NHWC only)
 *   cam       (B,V,64) camera table            centers (B,3) grid centres, mm
 *   valid     (B) uint8; 0 => sample skipped: its cubes/grids rows are written as zeros
 *             (project_layer.py:48,51,54: `grid_center[i][3] >= 0`)
 *   cubes     (B,J,X,Y,Z) fp32, z fastest [(B,X,Y,Z,J) with SP3D_OUT_CHANNELS_LAST]
 *   grids     (B,X*Y*Z,3) fp32 or NULL (not wanted)
 *   grid_size HOST float[3] box edge lengths mm; W_in,H_in network input size (cfg IMAGE_SIZE)
 */
int sp3d_unproject_fwd(const float *const *hm_views, int hm_layout, int Jp, const float *cam, const float *centers,
                       const uint8_t *valid, float *cubes, float *grids, int B, int V, int J, int h, int w, int X,
                       int Y, int Z, const float *grid_size, int W_in, int H_in, void *stream);

/*
 * Same as sp3d_unproject_fwd for P output cubes that read from a batch of B <= P samples: cube p
 * samples the heat-maps / camera rows of sample `sample_of[p]` (device int32[P]; NULL = identity).
 * This is how the <= MAX_PEOPLE_NUM per-person fine grids of one batch are produced in ONE launch
 * instead of the reference's per-candidate loop (lib/models/multi_person_posenet.py:84-88 calling
 * pose_regression_net.py:46).  centers (P,3), valid (P), cubes (P,J,X,Y,Z), grids (P,N,3)|NULL.
 */
int sp3d_unproject_fwd_indexed(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                               const int32_t *sample_of, const float *centers, const uint8_t *valid, float *cubes,
                               float *grids, int P, int V, int J, int h, int w, int X, int Y, int Z,
                               const float *grid_size, int W_in, int H_in, void *stream);

/*
 * sp3d_unproject_fwd_indexed writing a PLANAR result into a larger caller-owned buffer: element (p, j, x, y, z) goes to
 * cubes[p*s[0] + j*s[1] + x*s[2] + y*s[3] + z], `out_strides` = HOST int64[4] in elements (z is contiguous, planes must
 * not overlap).  Everything outside the addressed elements is left untouched.  This is how the root cubes land
 * directly inside the zero-padded input of the frequency-domain opening convolution (the consumer of
 * cuboid_proposal_net.py:110's cubes) without a pad/copy pass.  NHWC input only, no grids, no channels-last.
 * 16-byte stores are used when every stride is a multiple of 4 and `cubes` is 16-byte aligned.
 */
int sp3d_unproject_fwd_strided(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                               const int32_t *sample_of, const float *centers, const uint8_t *valid, float *cubes,
                               const int64_t *out_strides, int P, int V, int J, int h, int w, int X, int Y, int Z,
                               const float *grid_size, int W_in, int H_in, void *stream);

/*
 * Gradient of get_voxel w.r.t. the heat-maps (autograd of project_layer.py:93-99;
 * cameras/grids never need gradients: proposals are detached, cuboid_proposal_net.py:57-59).
 *   hm_views      planar heat-maps of the forward pass (needed for the clamp mask)
 *   grad_cubes    (B,J,X,Y,Z)
 *   grad_hm_views HOST array of V device pointers (B,J,h,w); MUST be zero-filled by the
 *                 caller: the kernel accumulates with atomics (summation order is therefore
 *                 not deterministic; bounded by fp32 rounding).
 */
int sp3d_unproject_bwd(const float *const *hm_views, const float *cam, const float *centers, const uint8_t *valid,
                       const float *grad_cubes, float *const *grad_hm_views, int B, int V, int J, int h, int w,
                       int X, int Y, int Z, const float *grid_size, int W_in, int H_in, void *stream);

int sp3d_unproject_bwd_indexed(const float *const *hm_views, const float *cam, const int32_t *sample_of,
                               const float *centers, const uint8_t *valid, const float *grad_cubes,
                               float *const *grad_hm_views, int P, int V, int J, int h, int w, int X, int Y, int Z,
                               const float *grid_size, int W_in, int H_in, void *stream);

/*
 * Training pair with a line-coalesced scatter.  sp3d_unproject_fwd_train = sp3d_unproject_fwd_indexed
 * (NHWC fp32 input) that also writes pass_mask (P, X*Y*Z) uint16: bit j set where channel j's pre-clamp
 * value is inside [0,1] and the voxel is not NaN-zeroed - exactly where torch's clamp / index_put_
 * backward let the gradient through (project_layer.py:97-99).  sp3d_unproject_bwd_packed then needs no
 * heat-maps: it accumulates into grad_packed (V,B,h,w,Jp) fp32 channels-last, ZERO-FILLED by the caller,
 * so that one atomic instruction covers whole 64-byte pixels (15x the L2 atomic rate of planar scatter).
 */
int sp3d_unproject_fwd_train(const float *const *hm_views, int hm_layout, int Jp, const float *cam,
                             const int32_t *sample_of, const float *centers, const uint8_t *valid, float *cubes,
                             float *grids, uint16_t *pass_mask, int P, int V, int J, int h, int w, int X, int Y, int Z,
                             const float *grid_size, int W_in, int H_in, void *stream);
/*
 * Both scatters come in two kernels that give the same sums (the _det pair the same BITS): per tap (one memory atomic per
 * 2x2 tap and 64-byte pixel), and block merge (an 8x8x4 block of voxels first adds its taps in an LDS patch, 64-bit fixed
 * point, and every touched pixel leaves once - 3x faster where voxels lie closer than ~2 pixels, the 64^3 person cubes).
 * `scatter` picks per CALL (the library keeps no selector state); any other value: SP3D_EINVAL.
 */
enum {
    SP3D_SCATTER_AUTO = 0,    /* by voxel pitch: <= 50 mm on every axis -> merge, else per tap */
    SP3D_SCATTER_PER_TAP = 2,
    SP3D_SCATTER_MERGE = 3
};
int sp3d_unproject_bwd_packed(const float *cam, const int32_t *sample_of, const float *centers, const uint8_t *valid,
                              const float *grad_cubes, const uint16_t *pass_mask, float *grad_packed, int B, int P,
                              int V, int J, int Jp, int h, int w, int X, int Y, int Z, const float *grid_size,
                              int W_in, int H_in, int scatter, void *stream);

/*
 * DETERMINISTIC form of sp3d_unproject_bwd_packed (the reference's grid_sampler_2d_backward and the
 * fp32 atomics above both sum in hardware order).  Contributions are accumulated in 64-bit FIXED POINT with integer
 * atomics - integer addition is associative, so the result is bit-identical run to run and independent of scheduling.
 *   grad_fixed  (V,B,h,w,Jp) int64, zero-filled by the caller
 *   scale       DEVICE float: a power of two 2^k; a contribution v is added as round(v * scale).
 *               REQUIRED: |v * scale| < 2^50 for every contribution (the merge kernel rounds with the 1.5*2^52 mantissa
 *               trick, exact only below that; beyond it the two kernels stop agreeing).  The intended choice is k from
 *               max|grad_cubes| so that |v * scale| <= 2^40: 2^23 contributions per pixel then still fit in 63 bits.
 * sp3d_fixed_to_float(acc, out, scale, n) converts: out[i] = (float)(acc[i] / scale).
 */
int sp3d_unproject_bwd_packed_det(const float *cam, const int32_t *sample_of, const float *centers, const uint8_t *valid,
                                  const float *grad_cubes, const uint16_t *pass_mask, int64_t *grad_fixed,
                                  const float *scale, int B, int P, int V, int J, int Jp, int h, int w, int X, int Y,
                                  int Z, const float *grid_size, int W_in, int H_in, int scatter, void *stream);
int sp3d_fixed_to_float(const int64_t *acc, float *out, const float *scale, int64_t n, void *stream);

/*
 * core.proposal.nms + ProposalLayer.get_real_loc (lib/core/proposal.py:28-48,
 * lib/models/cuboid_proposal_net.py:42-52): 3x3x3 local-max mask, top-k over the flat volume
 * (ties: larger value, then LOWER flat index), unravel, index -> mm.
 *   root_cubes (B,X,Y,Z)   vals (B,k) fp32   idx (B,k,3) int64   locs (B,k,3) fp32 mm or NULL
 *   workspace  device scratch of sp3d_nms_topk_workspace_bytes(...) bytes
 */
int64_t sp3d_nms_topk_workspace_bytes(int B, int X, int Y, int Z, int k);
int sp3d_nms_topk(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                  const float *grid_center, float *vals, int64_t *idx, float *locs, void *workspace, void *stream);
/* the same plus ProposalLayer.forward in eval (cuboid_proposal_net.py:54-83, threshold rule :79-81) in the merge kernel:
 * grid_centers (B,k,5) = [x, y, z mm, (score > threshold) - 1, score]  (locs must be given; NULL = not wanted) */
int sp3d_nms_proposals(const float *root_cubes, int B, int X, int Y, int Z, int k, const float *grid_size,
                       const float *grid_center, float threshold, float *vals, int64_t *idx, float *locs,
                       float *grid_centers, void *workspace, void *stream);

/*
 * SoftArgmaxLayer.forward (lib/models/pose_regression_net.py:19-28):
 *   out[b,j,:] = sum_n softmax(beta * x[b,j,:])[n] * grids[b,n,:]
 *   x (Bv,J,N)   grids (Bv,N,3)   out (Bv,J,3)
 */
int sp3d_soft_argmax(const float *x, const float *grids, float *out, int Bv, int J, int64_t N, float beta,
                     void *stream);
/* same result with the voxel centres regenerated from (centers (Bv,3), grid_size, X,Y,Z) - the
 * values compute_grid (project_layer.py:22-40) would have produced - instead of read from `grids` */
int sp3d_soft_argmax_grid(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                          float *out, int Bv, int J, float beta, void *stream);
/* the training pair of the same layer (autograd of pose_regression_net.py:19-28 w.r.t. the V2V output; the voxel centres carry
 * no gradient): _train = sp3d_soft_argmax_grid that also keeps stats (Bv,J,2) = (max of beta x, sum of exp) per row, NULL
 * allowed; _bwd: grad_x[b,j,n] = beta p_n (g . grid_n - g . out[b,j]) in ONE elementwise pass (read x, write grad_x), given
 * out (Bv,J,3), stats and grad_out (Bv,J,3).  x, grad_x planar (Bv,J,X*Y*Z). */
int sp3d_soft_argmax_grid_train(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z, float *out,
                                float *stats, int Bv, int J, float beta, void *stream);
int sp3d_soft_argmax_grid_bwd(const float *x, const float *centers, const float *grid_size, int X, int Y, int Z,
                              const float *out, const float *stats, const float *grad_out, float *grad_x, int Bv, int J,
                              float beta, void *stream);

/*
 * Fused inference epilogue for the V2V conv stack (BatchNorm folded into the conv weights by the
 * caller): in place on y (batch, C, inner) [planar] or (batch, inner, C) [channels_last],
 *   mode 0: y += shift[c]   1: relu(y + shift[c])   2: relu(y + shift[c] + residual)
 *   mode 3: relu(y + shift[c]) + residual
 * replacing the separate bias / BatchNorm3d / ReLU / add kernels of lib/models/v2v_net.py:13-17,
 * 26-45, 60-69, 100-108.  Needs C % 4 == 0 (channels_last) or inner % 4 == 0 (planar).
 */
int sp3d_channel_shift_act(float *y, const float *shift, const float *residual, int mode, int64_t batch, int C,
                           int64_t inner, int channels_last, void *stream);

/*
 * First node of a HIP-graphed step: copy slot (*counter % R) of a PINNED HOST ring `ring[R][n]` (device-accessible
 * pointer) into the device buffer `dst[n]`, then ++*counter (`counter`: device uint32, zero before the first launch).
 * Lets every replay of one captured graph consume a fresh per-batch camera table (project_layer.py:64-75's per-call
 * host data) without a host->device copy command between graph launches.  The host writes slot t % R before launching
 * replay t and must not rewrite a slot before the replay that reads it has finished.
 */
int sp3d_fetch_ring(const float *ring, float *dst, uint32_t *counter, int R, int n, void *stream);

/* MaxPool3d(kernel 2, stride 2) of the V2V encoder (v2v_net.py:48-54) on channels-last activations:
 * x (B,X,Y,Z,C) -> y (B,X/2,Y/2,Z/2,C); X,Y,Z even, C % 4 == 0; NaN propagates like torch.max_pool3d. */
int sp3d_maxpool2x_cl(const float *x, float *y, int B, int X, int Y, int Z, int C, void *stream);

/* Tail of the frequency-domain 7x7x7 opening conv (replaces v2v_net.py:10-20's BatchNorm+ReLU after the conv): crop the
 * [0:X,0:Y,0:Z] corner of planar volumes src (B,C,SX,SY,SZ), add shift[c] (folded BN), optional ReLU, and write
 * channels-last dst (B,X,Y,Z,C) in one pass.  C % 4 == 0, Z % 4 == 0, SZ % 4 == 0, 16-byte aligned pointers. */
int sp3d_crop_shift_act_cl(const float *src, float *dst, const float *shift, int B, int C, int X, int Y, int Z, int SX, int SY,
                           int SZ, int relu, void *stream);

/* Batched 3-D real transforms (unnormalised, like rfftn / irfftn(norm="forward")) of the frequency-domain opening conv:
 *   sp3d_rfft3d : real in (batch,SX,SY,SZ) dense -> complex out (batch,SX,SY,SZ/2+1) interleaved (re,im); `in` is preserved
 *   sp3d_irfft3d: complex in (batch,SX,SY,SZ/2+1) -> real out (batch,SX,SY,SZ); `in` is SCRATCH (contents destroyed)
 * hipFFT plans cached per (device, direction, batch, SX,SY,SZ); the first call of a shape allocates (not capturable). */
int sp3d_rfft3d(const float *in, float *out, int batch, int SX, int SY, int SZ, void *stream);
int sp3d_irfft3d(float *in, float *out, int batch, int SX, int SY, int SZ, void *stream);

/* In-place batched 2-D complex transform of dense (batch,SX,SY) planes of interleaved (re,im) floats, unnormalised;
 * inverse != 0: e^{+i} kernel.  Cached hipFFT plan per (device, batch, SX, SY). */
int sp3d_cfft2d(float *data, int batch, int SX, int SY, int inverse, void *stream);
/* The same with the caller's zero-padding knowledge: only the first rows_in rows (index along SX) of every input plane are
 * non-zero and only the first rows_out rows of every result plane are needed (the others may be left unwritten / stale).
 * 88 x 88 planes run as ONE kernel with the plane in LDS (one HBM round trip instead of the plan's two strided passes);
 * other sizes use the hipFFT plan on whole planes. */
int sp3d_cfft2d_ex(float *data, int batch, int SX, int SY, int inverse, int rows_in, int rows_out, void *stream);

/* z passes of the ROOT GRID's opening conv (v2v_net.py:113-117 on the 80x80x20 grid) as direct DFTs, spectrum layout
 * (B, channels, SZ/2+1, SX, SY) so that the x,y passes are sp3d_cfft2d over batch = B*channels*(SZ/2+1) planes:
 *   sp3d_zdft_fwd_cl: x (B,X,Y,Z,C) channels-last real (the unprojection's channels-last result, C = 16 with zero pad
 *     channels) -> spec (B,Cout,SZ/2+1,SX,SY) complex = DFT_z of the rows zero-padded to SZ; rows with x >= X or y >= Y
 *     are written as zeros; only channels c < Cout are kept.
 *   sp3d_zdft_inv_cl: spec (B,O,SZ/2+1,SX,SY) complex (after the inverse x,y passes) -> y (B,X,Y,Z,O) channels-last real
 *     = act(shift[o] + unnormalised C2R_z(spec))[0:X,0:Y,0:Z]; relu != 0: ReLU.
 * Built for (Z,SZ) = (20,28), C = O = 16 (SP3D_EUNSUPPORTED otherwise: callers use sp3d_rfft3d / sp3d_irfft3d). */
int sp3d_zdft_fwd_cl(const float *x, float *spec, int B, int C, int Cout, int X, int Y, int Z, int SX, int SY, int SZ,
                     void *stream);
int sp3d_zdft_inv_cl(const float *spec, float *y, const float *shift, int B, int O, int X, int Y, int Z, int SX, int SY, int SZ,
                     int relu, void *stream);

/* Round 6: the root grid's unprojection FUSED with that z pass (replaces sp3d_unproject_fwd + sp3d_zdft_fwd_cl on the
 * inference path; reference: lib/models/project_layer.py:42-102 followed by the 7x7x7 opening conv of
 * lib/models/v2v_net.py:113-117).  A workgroup owns a 4 x 4 bundle of z columns (all Z = 20 voxels, all J channels), keeps
 * the fused values in LDS and stores their z-spectrum instead of the cubes:
 *   sp3d_unproject_fwd_zdft: same inputs as sp3d_unproject_fwd with hm_layout = SP3D_LAYOUT_NHWC, Jp = 16 ->
 *     spec (B, J, SZ/2+1, X/4, Y/4, 16) complex64: per (b, channel, kz) plane the 4 x 4 tiles in row-major tile order,
 *     inside a tile position 4 * (x % 4) + (y % 4) - every store is one whole 128-byte line; NO zero padding is stored.
 *     Values are bit-identical to sp3d_zdft_fwd_cl applied to sp3d_unproject_fwd's channels-last cubes.  Samples with
 *     valid == 0 give all-zero spectra.  Built for (Z,SZ) = (20,28), Jp = 16, X % 4 == Y % 4 == 0; spec 128-byte aligned.
 *   sp3d_cfft2d_88_tiled: the forward x,y pass reading that layout: tiled (batch, X/4, Y/4, 16) -> planes (batch,88,88)
 *     complex, = sp3d_cfft2d_ex(forward, rows_in = X) of the zero-padded planes, un-tiling while it loads. */
int sp3d_unproject_fwd_zdft(const float *const *hm_views, int Jp, const float *cam, const float *centers,
                            const uint8_t *valid, float *spec, int B, int V, int J, int h, int w, int X, int Y, int Z,
                            const float grid_size[3], int W_in, int H_in, int SZ, void *stream);
int sp3d_cfft2d_88_tiled(const float *tiled, float *planes, int batch, int X, int Y, void *stream);

/* Round 6: the forward channel contraction of the frequency-domain opening conv (v2v_net.py:113-117, 7x7x7 taps) with the
 * weight spectrum's transform along y done per bin instead of stored: Y[b,o,row,ky] = sum_c X[b,c,row,ky] * W^[o,c,row,ky],
 * W^ = G_3 + sum_{u=1..3} S_u cos(2 pi ky u / SY) + i D_u sin(2 pi ky u / SY).  T (rows, O, C, 14) fp32 holds
 * (G_3, S_1, D_1, S_2, D_2, S_3, D_3) as complex pairs - the taps transformed along the other two axes (rows = their bins),
 * sums and differences of the +-u taps; tw (SY, 3, 2) = (cos, sin).  12.6x fewer weight bytes than sp3d_freq_contract's
 * full spectrum (17.7 MB instead of 223 MB on the root grid).  C <= 16, 4 * SY <= 384. */
int sp3d_freq_contract_ty(const float *X, const float *T, const float *tw, float *Y, int B, int C, int O, int rows, int SY,
                          void *stream);

/*
 * Synthetic-root branch of the self-supervised root net (lib/models/cuboid_proposal_net_soft.py:151-241):
 *   sp3d_gaussian_target_3d   :168-203  target (B,X,Y,Z) = clip(max over the R roots of a 3-sigma-windowed 3D
 *                                       Gaussian); gx/gy/gz are the fp32 voxel-centre coordinates per axis
 *   sp3d_render_root_heatmaps :205-227  out (V,B,1,h,w) = clip(sum over roots of sigma-3 Gaussians at the roots'
 *                                       projections); cam is the (B,V,64) table whose affine is meta['trans'],
 *                                       stride = network-input px per heat-map px (the reference's 4.0)
 * roots (B,R,3) fp32 mm, R <= SP3D_MAX_TOPK.  The additive N(0,0.02) noise stays with the caller.
 */
int sp3d_gaussian_target_3d(const float *roots, int B, int R, const float *gx, const float *gy, const float *gz, int X,
                            int Y, int Z, float sigma, float *target, void *stream);
int sp3d_render_root_heatmaps(const float *roots, int B, int R, const float *cam, int V, int h, int w, float stride,
                              float *out, void *stream);
/*
 * Differentiable joint rendering of the self-supervised pose loss (lib/models/multi_person_posenet_ssv.py:409-465):
 * kps (N, P, J, 2) projected joints in heat-map pixels (N = views x samples), count (N) people per entry (NULL: P),
 * out (N, J, h, w) = clip(sum over people of sigma-Gaussians, 0, 1); _bwd returns d loss / d kps given d loss / d out.
 * P <= 16.
 */
int sp3d_render_joints_fwd(const float *kps, const int *count, int N, int P, int J, int h, int w, float sigma, float *out,
                           void *stream);
int sp3d_render_joints_bwd(const float *kps, const int *count, const float *grad_out, int N, int P, int J, int h, int w,
                           float sigma, float *grad_kps, void *stream);

/*
 * Channel contraction of a frequency-domain convolution (the 7x7x7 opening conv of V2VNet, lib/models/v2v_net.py:
 * 113-117, run as rFFT -> this -> irFFT in inference):  Y[b,o,f] = sum_c X[b,c,f] * W[o,c,f], complex64 stored as
 * interleaved (re,im) floats; X (B,C,F), W (O,C,F) (= conj FFT of the weights), Y (B,O,F), F = number of bins.
 */
int sp3d_freq_contract(const float *X, const float *W, float *Y, int B, int C, int O, int64_t F, void *stream);
/* general form, Y[i,j,f] = sum_k P[i,k,f] * Q[j,k,f] with optional conjugation of either operand; strides of the two
 * leading dimensions in complex elements (bins contiguous).  Also serves the two backward products of the layer:
 * grad input = sum_o Gy[b,o] W^[o,c], grad weight = sum_b conj(Gy[b,o]) X[b,c]  (autograd of v2v_net.py:113-117). */
int sp3d_freq_contract_ex(const float *P, const float *Q, float *Y, int I, int J, int K, int64_t F, int64_t sPi,
                          int64_t sPk, int64_t sQj, int64_t sQk, int conj_p, int conj_q, void *stream);

/*
 * Winograd F(2x2x2, 3x3x3) transforms for the low-resolution 3x3x3 convolutions of V2VNet in inference
 * (lib/models/v2v_net.py:23-45 at 1/4 resolution): channels-last activations only.
 *   sp3d_wino_input   x (B,X,Y,Z,C) -> V (64, T, C),  T = B*ceil(X/2)*ceil(Y/2)*ceil(Z/2)  (B^T d B per axis)
 *   [caller: M = bmm(V, U) with U (64, C, O) = G g G^T of the weights]
 *   sp3d_wino_output  M (64, T, O) -> y (B,X,Y,Z,O) = A^T m A, + shift[o] (+ residual) (+ ReLU): mode as in
 *                     sp3d_channel_shift_act (0 shift, 1 relu(shift), 2 relu(shift+res), 3 relu(shift)+res)
 */
int sp3d_wino_input(const float *x, float *V, int B, int X, int Y, int Z, int C, void *stream);
int sp3d_wino_output(const float *M, float *y, const float *shift, const float *residual, int mode, int B, int X, int Y,
                     int Z, int O, void *stream);
/* the same convolution in ONE launch for the full-resolution layers (C = 16 | 32 -> O = 32): transforms, the 64 products
 * (v_mfma_f32_32x32x2_f32) and the epilogue fused, U (64, C, 32) as above, x / y channels-last */
int sp3d_wino_fused(const float *x, const float *U, float *y, const float *shift, const float *residual, int mode, int B,
                    int X, int Y, int Z, int C, int O, void *stream);

/* sp3d_wino_fused with the products on the bf16 matrix pipe at fp32 accuracy: every fp32 operand is split exactly into
 * three bf16 pieces (hi+mid+lo) and the six significant piece products are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16 (dropped terms < 2^-24 relative).  U3: the pre-transformed weights of sp3d_wino_fused split on
 * the host into 24-byte records [mid(4ch) hi(4ch) lo(4ch)] of bf16, record index ((p*(C/8) + chunk)*2 + half)*32 + o,
 * channel = 8*chunk + 4*half + q (selfpose3d_amd/_lib.py: wino_weights_split).  Same arguments otherwise. */
int sp3d_wino_fused_split(const float *x, const void *U3, float *y, const float *shift, const float *residual, int mode, int B,
                          int X, int Y, int Z, int C, int O, void *stream);

/* The same scheme for the half-resolution layers (C = 32 | 64 -> O = 64): blocks of 4x4x1 tiles x 64 outputs on
 * v_mfma_f32_16x16x32_bf16, 16-channel chunks.  U3 records [mid(4ch) hi(4ch) lo(4ch)] at index
 * ((p*(C/16) + chunk)*4 + group)*64 + o, channel = 16*chunk + 4*group + q (_lib.wino_weights_split(U, 16)). */
int sp3d_wino_fused_split64(const float *x, const void *U3, float *y, const float *shift, const float *residual, int mode,
                            int B, int X, int Y, int Z, int C, int O, void *stream);

/* Direct 3x3x3 stride-1 'same' convolution (v2v_net.py:23-45 Res3DBlock convs) + the fused epilogue of
 * sp3d_channel_shift_act, as an implicit GEMM on v_mfma_f32_32x32x16_bf16 with exact three-piece bf16 splits of both
 * operands (fp32 accuracy, fp32 accumulation; no Winograd transforms).  x, y channels-last (B,X,Y,Z,C) / (B,X,Y,Z,O);
 * W3: 48-byte records of bf16 = the three B operands {hi,lo} {hi,hi} {mid,mid} (4 channels each) at index
 * ((tap*(C/8) + chunk)*2 + half)*O + o, tap = kz*9 + ky*3 + kx of w[o][c][kx][ky][kz] (_lib.conv_weights_split), 16-byte aligned.  (C,O) in {(16,32),(32,32)}. */
int sp3d_conv3_split(const float *x, const void *W3, float *y, const float *shift, const float *residual, int mode, int B,
                     int X, int Y, int Z, int C, int O, void *stream);

/*
 * Scatter + epilogue of ConvTranspose3d(kernel 2, stride 2) -> BatchNorm -> ReLU (+ skip) (lib/models/v2v_net.py:57-69,
 * 100-108) once the layer has been computed as one GEMM G (batch*X*Y*Z, 8*O) with column order (i,j,k,o):
 * out (batch,2X,2Y,2Z,O channels-last) = relu(G + shift[o]) + skip.
 */
int sp3d_upsample2x_scatter(const float *G, float *out, const float *shift, const float *skip, int64_t batch, int X, int Y,
                            int Z, int O, void *stream);

/* The same scatter fused with the network's 1x1x1 output conv (v2v_net.py:128-133) for the LAST up-sampling layer, whose
 * O = 32-channel result has no other consumer: head (batch,2X,2Y,2Z,J channels-last) = bout[j] + sum_o wout[j][o] *
 * (relu(G + shift[o]) + skip[o]).  wout (J,32) row-major, O must be 32. */
int sp3d_upsample2x_scatter_head(const float *G, float *head, const float *shift, const float *skip, const float *wout,
                                 const float *bout, int64_t batch, int X, int Y, int Z, int O, int J, void *stream);

/*
 * GROUPED training-mode batch normalisation on channels-last tensors (round 5): what lets the training pose net run all
 * candidate slots - and the backbone all cameras - as ONE batch and still be the reference's per-call BatchNorm
 * (lib/models/v2v_net.py:14,28,31,38,64 under the loop of lib/models/multi_person_posenet.py:84-88 and
 * multi_person_posenet_ssv.py:354-383; lib/models/pose_resnet.py BatchNorm2d under multi_person_posenet.py:44-47).
 *   x, y, dy, dx   (N, S, C), C contiguous (torch.channels_last / channels_last_3d); dtype SP3D_GBN_F32 | SP3D_GBN_F64;
 *                  C % 4 == 0 (F32) / C % 2 == 0 (F64), C <= 2048 (F32) / 1024 (F64)
 *   group_of       int32 (N): group of sample n, any assignment in [0, G)
 *   group_samples  int32 (G): number of samples in each group (0 allowed: statistics 0 / 1, never used)
 * per (group g, channel c), over the group's samples x S:  mean = E[x],  var = E[x^2] - mean^2 (float64 accumulation),
 *   y = (x - mean) / sqrt(var + eps) * weight + bias                          mode SP3D_GBN_PLAIN (0)
 *   y = max(0, that)                                                           mode SP3D_GBN_RELU (1)
 *   y = max(0, that + residual)   (the tail of a residual block in one pass)   mode SP3D_GBN_ADD_RELU (2), residual (N, S, C)
 * running_mean / running_var (NULL: none) receive the momentum updates of groups 0 .. G_update-1 IN THAT ORDER with the
 * unbiased variance - the sequence of updates the reference's loop makes (groups >= G_update, e.g. padding cubes, leave
 * them alone).  mean, invstd, scale, shift: (G, C) outputs, kept by the caller for the backward.
 * workspace: sp3d_gbn_workspace_bytes(G, C) bytes, ZERO-FILLED by the caller ONCE: the statistics merge with float64
 * atomics into SP3D_GBN_REPLICAS copies and the finalising kernel of each call zeroes them again, so forward and backward
 * calls of any number of layers with the same (G, C) may share one workspace on one stream.
 * backward: dx = d loss / d x given dy = d loss / d y (mode 1: dy counts only where y > 0, recomputed from x - y is not
 * needed; mode 2: where the forward's output `y` (N, S, C) is > 0, and grad_residual (N, S, C) receives that masked dy);
 * grad_weight, grad_bias (C) summed over all groups (NULL: skipped); k123: scratch of 3 * G * C elements.
 */
#define SP3D_GBN_F32 0
#define SP3D_GBN_F64 1
#define SP3D_GBN_REPLICAS 16
#define SP3D_GBN_PLAIN 0
#define SP3D_GBN_RELU 1
#define SP3D_GBN_ADD_RELU 2
int64_t sp3d_gbn_workspace_bytes(int G, int C);
int sp3d_gbn_forward(const void *x, void *y, int dtype, const int32_t *group_of, const int32_t *group_samples, int N,
                     int64_t S, int C, int G, int G_update, const void *weight, const void *bias, void *running_mean,
                     void *running_var, double eps, double momentum, int mode, const void *residual, void *mean,
                     void *invstd, void *scale, void *shift, double *workspace, void *stream);
int sp3d_gbn_backward(const void *x, const void *dy, void *dx, int dtype, const int32_t *group_of,
                      const int32_t *group_samples, int N, int64_t S, int C, int G, const void *weight, const void *mean,
                      const void *invstd, const void *scale, const void *shift, int mode, const void *y,
                      void *grad_residual, void *grad_weight, void *grad_bias, void *k123, double *workspace, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SP3D_H */
