/*
 * exaspim_affinity.h -- C ABI of the MI355X (gfx950) sliding-window 3D-UNet
 * affinity-inference path.
 *
 * The reference (AllenNeuralDynamics/aind-exaspim-neuron-segmentation) has no
 * FFI on this path: it is plain Python over torch. The entry points below are
 * what a binding for that path needs; each one names the reference code it
 * replaces (paths relative to src/aind_exaspim_neuron_segmentation/). The
 * reference-side ctypes stub a maintainer would add is in INTEGRATION.md.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch types.
 *  - Every function that can fail returns int: 0 = ok, negative = error
 *    (EXASPIM_E_*); exaspim_last_error() returns a thread-local message.
 *  - "dev" pointers are HIP device pointers owned by the caller (e.g.
 *    torch.Tensor.data_ptr()); the library never allocates or frees device
 *    memory and never synchronises the device. Kernels are enqueued on the
 *    caller's HIP stream ("stream", a hipStream_t passed as void*; NULL = the
 *    default stream).
 *  - Volumes are C-ordered (z, y, x). Activations handed across the ABI are
 *    NCDHW float32, like the reference's tensors; the blocked channels-last
 *    layout used between kernels (one plane of 32-byte voxel records per
 *    32-byte channel chunk) is internal to the workspace.
 */
#ifndef EXASPIM_AFFINITY_H
#define EXASPIM_AFFINITY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 3: exaspim_export_f16 added (entry points are only ever added within a major line);
 * 5: exaspim_unet_forward_absmax, exaspim_histogram_wide (EXASPIM_VOX_F64),
 *    exaspim_unet_set_options (replaces an environment switch) */
#define EXASPIM_ABI_VERSION 5

/* error codes */
#define EXASPIM_OK 0
#define EXASPIM_E_INVALID (-1)   /* bad argument (shape, dtype, NULL)        */
#define EXASPIM_E_HIP (-2)       /* HIP runtime error (message has details)  */
#define EXASPIM_E_WORKSPACE (-3) /* workspace too small                      */
#define EXASPIM_E_NODEVICE (-4)  /* no usable gfx950 device                  */

/* compute dtype of the network (activations and weights between kernels;
 * accumulation is always float32) */
#define EXASPIM_DT_F32 0  /* exact f32 MFMA (v_mfma_f32_32x32x2_f32)        */
#define EXASPIM_DT_BF16 1 /* bf16 storage, v_mfma_f32_32x32x16_bf16         */
#define EXASPIM_DT_F16 2  /* f16 storage,  v_mfma_f32_32x32x16_f16          */
/* OR-ed into "dtype" wherever a network is described: the Up blocks use
 * ConvTranspose3d(k=2, s=2) instead of trilinear upsampling, i.e. the
 * reference's UNet3D(trilinear=False) (unet3d.py:254-258; 136 state_dict
 * tensors, "upN.up.weight (Cin, Cin/2, 2, 2, 2)" and "upN.up.bias" precede each
 * block's DoubleConv in the canonical parameter order). */
#define EXASPIM_UP_CONVT 0x100

/* voxel dtype of an input volume */
#define EXASPIM_VOX_U8 0
#define EXASPIM_VOX_U16 1
#define EXASPIM_VOX_I16 2
#define EXASPIM_VOX_F32 3
#define EXASPIM_VOX_F64 4 /* float64, and wider integers converted on the host: what float32 cannot carry */

typedef struct exaspim_unet exaspim_unet; /* opaque engine handle */

/* 3-D block of a (possibly larger, possibly sharded) volume. "dims" is the
 * shape of the local array, "origin" the global coordinate of its first
 * voxel, "global" the shape of the whole volume. Single-device callers pass
 * origin = 0 and global = dims. */
typedef struct exaspim_block {
    int32_t dims[3];
    int32_t origin[3];
    int32_t global[3];
} exaspim_block;

/* sliding-window geometry: predict()'s patch_shape / overlap / trim
 * (inference.py:36-38) */
typedef struct exaspim_window {
    int32_t patch[3];
    int32_t overlap[3];
    int32_t trim;
} exaspim_window;

int exaspim_abi_version(void);
const char* exaspim_last_error(void);

/* ---- model: replaces load_model() / UNet3D (inference.py:400-424,
 *      machine_learning/unet3d.py:16-336). Every "dtype" argument is one of
 *      EXASPIM_DT_*, optionally OR-ed with EXASPIM_UP_CONVT to select the
 *      UNet3D(trilinear=False) variant whose Up blocks use
 *      ConvTranspose3d(k=2, s=2) (unet3d.py:254-258) ---------------------- */

/* Number of float32 values in the canonical parameter vector for a UNet3D
 * with level widths channels[0..4] (unet3d.py:56) and "out_channels" head
 * outputs: the state_dict tensors in state_dict order, "num_batches_tracked"
 * skipped, i.e. per conv: weight(Cout,Cin,3,3,3), bias, bn.weight, bn.bias,
 * bn.running_mean, bn.running_var; with EXASPIM_UP_CONVT each Up block is
 * preceded by up.weight(Cin,Cin/2,2,2,2), up.bias; finally outc.conv.weight,
 * outc.conv.bias. Only the EXASPIM_UP_CONVT bit of "dtype" matters here. */
size_t exaspim_unet_param_count(const int32_t channels[5], int32_t out_channels,
                                int32_t dtype);

/* Size of the packed device image of the weights for a compute dtype. */
size_t exaspim_unet_packed_bytes(const int32_t channels[5], int32_t out_channels,
                                 int32_t dtype);

/* Host-only: folds eval-mode BatchNorm (eps 1e-5, unet3d.py:144,147) into the
 * preceding convolution (in float64), converts to "dtype" and lays the result
 * out in MFMA fragment order. "packed_host" receives packed_bytes bytes which
 * the caller uploads to the device unchanged. */
int exaspim_unet_pack_weights(const int32_t channels[5], int32_t out_channels,
                              int32_t dtype, const float* params, size_t n_params,
                              void* packed_host, size_t packed_bytes);

/* Binds a packed weight image that already lives on device "device". The
 * image must outlive the handle. */
int exaspim_unet_create(const int32_t channels[5], int32_t out_channels,
                        int32_t dtype, int32_t device, const void* packed_dev,
                        size_t packed_bytes, exaspim_unet** out);
void exaspim_unet_destroy(exaspim_unet* h);

/* Scratch bytes exaspim_unet_forward needs for a batch of n patches of
 * d x h x w voxels (each a multiple of 16, the constraint the reference's
 * Up.forward imposes, unet3d.py:281-288). */
size_t exaspim_unet_workspace_bytes(const exaspim_unet* h, int32_t n, int32_t d,
                                    int32_t hgt, int32_t w);

/* UNet3D.forward (unet3d.py:77-105): x_dev is float32 (n,1,d,h,w); out_dev
 * receives float32 (n,out_channels,d,h,w) logits, or sigmoid(logits) when
 * apply_sigmoid != 0 (inference.py:158). */
int exaspim_unet_forward(exaspim_unet* h, const float* x_dev, float* out_dev,
                         int32_t n, int32_t d, int32_t hgt, int32_t w,
                         int32_t apply_sigmoid, void* workspace_dev,
                         size_t workspace_bytes, void* stream);

/* The same forward pass for a caller that discards the outputs within "trim"
 * voxels of every patch face, as _predict_batch does (inference.py:161-162:
 * outputs[..., trim:-trim, trim:-trim, trim:-trim]). Those voxels of out_dev are
 * left untouched, and the last two convolutions skip the work that only they
 * would have needed; every other voxel is bit-identical to exaspim_unet_forward.
 * trim = 0, or a trim that would leave nothing, is the full forward pass. */
int exaspim_unet_forward_trimmed(exaspim_unet* h, const float* x_dev, float* out_dev,
                                 int32_t n, int32_t d, int32_t hgt, int32_t w,
                                 int32_t apply_sigmoid, int32_t trim, void* workspace_dev,
                                 size_t workspace_bytes, void* stream);

/* The same forward pass from a batch that exaspim_gather_patches_as has already written in
 * the first convolution's operand layout (exaspim_unet_input_layout: EXASPIM_IN_PADDED_F32
 * for a float32 engine, EXASPIM_IN_PADDED_SPLIT_F16 / _BF16 for the 16-bit ones):
 * x_prepared_dev is (n, d + 2, hgt + 2, w + 2) 4-byte words. _get_batch_inputs feeding
 * model(inputs) (inference.py:155-157) without the float32 patch tensor in between: one
 * launch and an 8-byte-per-voxel round trip less per batch; out_dev gets the same bits. */
int exaspim_unet_input_layout(const exaspim_unet* h);
int exaspim_unet_forward_prepared(exaspim_unet* h, const void* x_prepared_dev, float* out_dev,
                                  int32_t n, int32_t d, int32_t hgt, int32_t w,
                                  int32_t apply_sigmoid, int32_t trim, void* workspace_dev,
                                  size_t workspace_bytes, void* stream);

/* Range probe for the 16-bit storage modes. The reference loads ANY trained state_dict
 * (inference.py:400-424) and runs it in float32; IEEE-half storage holds |v| <= 65504 (stores
 * saturate there) with 11 significant bits. This is exaspim_unet_forward (nothing fused away,
 * nothing trimmed) that also raises absmax_dev[0] (inc.0), absmax_dev[1 + i] (i-th MFMA
 * convolution, the order of exaspim_unet_timing_begin's mask) and, with EXASPIM_UP_CONVT,
 * absmax_dev[18 + j] (up(j+1).up) to the largest |activation| the layer stored, as float32
 * (caller-zeroed float[EXASPIM_ABSMAX_SLOTS]; a NaN is reported as NaN). Run on a float32
 * engine it gives the true ranges of a checkpoint on real patches; on an f16 engine a value of
 * 65504 means a store saturated. UNet3D(compute_dtype="auto") decides with it on the first batch. */
#define EXASPIM_ABSMAX_SLOTS 22
int exaspim_unet_forward_absmax(exaspim_unet* h, const float* x_dev, float* out_dev,
                                int32_t n, int32_t d, int32_t hgt, int32_t w,
                                int32_t apply_sigmoid, float* absmax_dev, void* workspace_dev,
                                size_t workspace_bytes, void* stream);

/* Per-handle switches between bit-identical execution plans (tests and measurements; default 0):
 * SEPARATE_POOL runs every MaxPool3d(2) (unet3d.py:195) as its own launch instead of in the epilogue
 * of the convolution in front of it; SEPARATE_DEEP_POOLS does so for all but the first one. */
#define EXASPIM_OPT_SEPARATE_POOL 1u
#define EXASPIM_OPT_SEPARATE_DEEP_POOLS 2u
#define EXASPIM_OPT_PLAIN_UPSAMPLE 4u /* trilinear x2 on the un-pipelined kernel (same bits) */
#define EXASPIM_OPT_FIRST_PER_GROUP 8u /* inc.0 of the 16-bit modes group by group instead of on row strips (same bits) */
#define EXASPIM_OPT_UPSAMPLE_PER_THREAD 16u /* trimmed level-0 upsampling: per-thread pipeline instead of shared source rows (same bits) */
int exaspim_unet_set_options(exaspim_unet* h, uint32_t options);

/* Measurement hooks (bench.py's roofline leg). timing_begin arms HIP-event
 * timing, on the launch stream, of the MFMA convolutions whose bit is set in
 * conv_mask (bit i = i-th 3x3x3 conv after inc.0 in state_dict order: inc.3,
 * down1.0, down1.3, ... up4.0 = bit 15, up4.3 = bit 16); at most 16384 launches
 * are recorded. timing_read waits for the recorded events and returns, per
 * conv, the summed milliseconds and the number of launches, then disarms. */
int exaspim_unet_timing_begin(exaspim_unet* h, uint32_t conv_mask);
int exaspim_unet_timing_read(exaspim_unet* h, double ms_sum[17], int32_t count[17]);

/* ---- pre-processing: replaces np.minimum + img_util.normalize +
 *      _get_batch_inputs (inference.py:79-80,166-192;
 *      utils/img_util.py:362-379,405-428,504-533) ------------------------- */

/* Adds the 65536-bin histogram of min(voxel, clip) (clip applied iff
 * has_clip) over "n" voxels into hist_dev (uint64[65536], caller-zeroed).
 * 8/16-bit integer voxels are binned by value (I16: value + 32768); with a
 * fractional clip (np.minimum promotes the image to float64) every voxel above
 * it lands in bin ceil(clip), which then stands for the clip value itself. F32
 * voxels are binned by an order-preserving 32-bit key: pass 0 bins the key's
 * high 16 bits; pass 1 bins the low 16 bits of keys whose high half equals
 * "prefix"; voxels above a clip that float32 cannot hold (a float64 image, which
 * travels as float32) get the key of the float32 just above the clip, which
 * again stands for the clip value itself. Order statistics, and from them numpy's linear-interpolated
 * percentiles (img_util.py:526), follow exactly from these counts. */
int exaspim_histogram(const void* vol_dev, int32_t vox_dtype, size_t n,
                      double clip, int32_t has_clip, int32_t pass,
                      uint32_t prefix, uint64_t* hist_dev, void* stream);

/* The same for EXASPIM_VOX_F64 volumes (the reference takes ANY numeric array and works in
 * float64, inference.py:79-80, img_util.py:526-531): voxels are binned by an order-preserving
 * 64-bit key, 16 bits per pass -- pass p (0..3) bins key bits [48 - 16p, 64 - 16p) of the voxels
 * whose key bits above that field equal "prefix" (0 for pass 0). The clip is a float64 itself,
 * so min(voxel, clip) is exact. Four passes per order statistic give it exactly. */
int exaspim_histogram_wide(const void* vol_dev, int32_t vox_dtype, size_t n,
                           double clip, int32_t has_clip, int32_t pass,
                           uint64_t prefix, uint64_t* hist_dev, void* stream);

/* Builds a batch of network inputs: for patch i with global start
 * starts_dev[3*i..3*i+2] (int32 z,y,x), out[i] (float32 patch[0] x patch[1] x
 * patch[2]) = float32(clip01((min(v, clip) - mn) / denom)) evaluated in
 * float64, with v taken from the volume block and the part of the patch that
 * sticks out of the GLOBAL volume filled by numpy 'reflect' padding of the
 * in-volume part (img_util.py:378-379). denom = mx - mn + 1e-8. */
/* Layouts exaspim_gather_patches_as can write a batch in: the float32 patches above, or a
 * copy with a one-voxel zero border, (n, patch[0] + 2, patch[1] + 2, patch[2] + 2) 4-byte
 * words, holding the same float32 values or every value v split for the 16-bit matrix pipe,
 * bits(hi) | bits(lo) << 16 with hi = half(v), lo = half(v - float(hi)) (IEEE half or
 * bfloat16) -- what the engine's own padding pass makes of the float32 patch. */
#define EXASPIM_IN_F32 0
#define EXASPIM_IN_PADDED_F32 1
#define EXASPIM_IN_PADDED_SPLIT_F16 2
#define EXASPIM_IN_PADDED_SPLIT_BF16 3
int exaspim_gather_patches_as(const void* vol_dev, int32_t vox_dtype,
                              const exaspim_block* blk, const int32_t* starts_dev,
                              int32_t n, const int32_t patch[3], double clip,
                              int32_t has_clip, double mn, double denom, int32_t layout,
                              void* out_dev, void* stream);
int exaspim_gather_patches(const void* vol_dev, int32_t vox_dtype,
                           const exaspim_block* blk, const int32_t* starts_dev,
                           int32_t n, const int32_t patch[3], double clip,
                           int32_t has_clip, double mn, double denom,
                           float* out_dev, void* stream);

/* ---- post-processing: replaces the stitch loop and the final divide
 *      (inference.py:99-116,120-125) -------------------------------------- */

/* accum[c, s:e] += pred[i, c, trim:trim+(e-s)] for every patch i of the batch
 * in batch order, s = start + trim, e = min(s + patch - 2*trim, global dim),
 * restricted to the accumulator block. pred_dev is float32
 * (n, channels, patch) (the sigmoid output of exaspim_unet_forward);
 * accum_dev is float32 (channels, blk->dims). Deterministic: each voxel is
 * summed by one thread in patch order, so repeated runs are bit-identical. */
int exaspim_stitch_accumulate(const float* pred_dev, const int32_t* starts_dev,
                              int32_t n, int32_t channels,
                              const exaspim_window* win, float* accum_dev,
                              const exaspim_block* blk, void* stream);

/* accum[c, v] /= (number of patches whose trimmed output covers v) where that
 * number is non-zero (it is a product of three per-axis counts fixed by the
 * geometry alone), capped at 2048 where the reference's float16 weights stop
 * counting (inference.py:92); uncovered voxels keep 0. */
int exaspim_stitch_finalize(float* accum_dev, int32_t channels,
                            const exaspim_window* win, const exaspim_block* blk,
                            void* stream);

/* dst[i] = (IEEE half) src[i], round to nearest even, for i < n: the reduced-precision
 * export of a finalised result. The consumer of predict()'s output,
 * affinities_to_segmentation, starts with affinities.astype(np.float32)
 * (inference.py:223), so it takes a float16 array as it is; values are in [0, 1], the
 * rounding error is at most 2.4e-4. src_dev float32, dst_dev 16-bit, both 16-byte
 * aligned. */
int exaspim_export_f16(const float* src_dev, void* dst_dev, size_t n, void* stream);

/* ---- synthetic input for benchmarks and tests --------------------------- */

/* vol[z,y,x] = splitmix64(seed + global linear index) % 2000 as uint16. */
int exaspim_synth_volume_u16(uint16_t* vol_dev, const exaspim_block* blk,
                             uint64_t seed, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EXASPIM_AFFINITY_H */
