/* mi355_unet3d.h -- C ABI of libmi355unet3d.so (gfx950 / MI355X only).
 *
 * The reference (ellisdg/3DUnetCNN) has no native layer: its hot path is torch ATen ops called from
 *   unet3d/models/pytorch/classification/myronenko.py:17-21,47-58   (GroupNorm -> ReLU -> Conv3d, residual add)
 *   unet3d/models/pytorch/classification/resnet.py:12-22            (conv3x3x3 / conv1x1x1, bias=False)
 *   unet3d/models/pytorch/segmentation/unet.py:27-44                (upsample -> pad -> cat)
 *   unet3d/models/pytorch/classification/decoder.py:99-106          (ConvTranspose3d k3 s2 p1 | 1x1x1 + trilinear x2)
 *   unet3d/train/training_utils.py:59-72,101-112                    (fwd -> criterion -> backward -> optimizer.step)
 * Each entry point below names the ATen call(s) it replaces. SURVEY.md section 8(b) is the contract.
 *
 * Conventions
 *  - All activations are NDHWC ("channels last 3d") views, fp32 by default (mi355_act.dtype): element (n,z,y,x,c) lives at
 *    p[(((n*D+z)*H+y)*W+x)*ld + c], ld >= C, ld % 4 == 0, p 16-byte aligned. ld > C is how the
 *    skip-concat is expressed: producers write straight into a channel slice of the concat buffer.
 *  - Ownership: the caller owns every buffer including workspaces; the library never allocates
 *    device memory and keeps no global mutable state.
 *  - Async: every call only enqueues work on `stream` (a hipStream_t passed as void*); no call
 *    synchronises. Re-entrant; one thread per device or one process per device are both fine.
 *  - Errors: 0 on success, negative mi355 status otherwise (no C++ exceptions cross the boundary).
 */
#ifndef MI355_UNET3D_H
#define MI355_UNET3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355_STATUS_OK 0
#define MI355_STATUS_EINVAL (-1)       /* bad shape / alignment / null pointer */
#define MI355_STATUS_EUNSUPPORTED (-2) /* combination not implemented */
#define MI355_STATUS_ELAUNCH (-3)      /* hipGetLastError() != hipSuccess after launch */
#define MI355_STATUS_EWORKSPACE (-4)   /* workspace too small */

/* Storage type of an activation tensor. fp32 is the reference's own; a 16-bit type is what its AutocastUNet (segmentation/unet.py:53-58)
 * keeps conv outputs in under torch autocast -- fp16 in the reference's amp mode (train/train.py:33-37), bf16 in BASELINE configs[2]:
 * HipAutocastUNet(activation_storage="bf16" | "fp16") stores every activation and activation gradient between the input volume and the
 * logits that way (statistics, weights, weight gradients, logits and the loss stay fp32). A 16-bit tensor goes with the operand precision
 * of its own type (MI355_ACT_BF16 with MI355_PREC_BF16, MI355_ACT_F16 with MI355_PREC_F16: a plain input of the 16-bit convolution kernels
 * IS the matrix operand as stored). A 16-bit view
 * has the same (n, d, h, w, c, ld) meaning in ELEMENTS; p must be 8-byte aligned (16-byte for the 16-bit convolution kernels, which also
 * want ld % 8 == 0). Where an entry point takes a raw tensor pointer beside the views (mi355_conv_desc.residual, mi355_gn_bwd_fuse.gx:
 * the type of y; the `addend` of mi355_gn_act_bwd: the type of dx) the pointer has the type named there. Every entry point checks the
 * types it is given: MI355_STATUS_EUNSUPPORTED for a combination it has no kernel for (the Winograd fp32 kernels, mixed-type operands of
 * one call other than those listed at the entry point), never a silent reinterpretation. */
#define MI355_ACT_F32 0
#define MI355_ACT_BF16 1
#define MI355_ACT_F16 2 /* IEEE fp16 storage: the tensors of the reference's own amp mode (fp16 autocast, train/train.py:33-37) */

/* NDHWC activation view (fp32 unless dtype says otherwise; a caller that zero-initialises the struct gets fp32). */
typedef struct mi355_act {
  void* p;
  int32_t n, d, h, w, c, ld;
  int32_t dtype; /* MI355_ACT_* */
} mi355_act;

/* Input-side fusion of the conv kernels. */
#define MI355_IN_PLAIN 0       /* conv reads x as is */
#define MI355_IN_AFFINE_ACT 1  /* conv reads act(scale[n,c]*x + shift[n,c]) : fused GroupNorm/InstanceNorm apply + (Leaky)ReLU */
#define MI355_IN_ZERO_INSERT 2 /* conv reads the x2 zero-inserted x : ConvTranspose3d(k3,s2,p1) fwd and stride-2 dgrad */
#define MI355_IN_S2D 3         /* kd == 1 only: x is a FINE tensor (2d,2h,2w,c); the conv sees its space-to-depth view
                                  (d,h,w,8c), logical channel p*c + k with p = 4a+2b+e  <->  x[2z+a,2y+b,2x+e,k].
                                  dgrad of ConvTranspose3d(k2,s2) (MONAI DynUNet up block). x->d/h/w are the FINE extents. */

/* Arithmetic of the 3x3x3 stride-1 conv kernels (forward, dgrad and wgrad). Accumulation is fp32 in every mode; the modes differ in how
 * the products are formed on the matrix pipe. The tensors a call reads and writes are fp32 unless their views say MI355_ACT_BF16 (goes
 * with MI355_PREC_BF16 only) or MI355_ACT_F16 (with MI355_PREC_F16 only): the operands are then the stored values; outputs are rounded
 * once, on store. */
#define MI355_PREC_F32 0     /* v_mfma_f32_32x32x2_f32: exact fp32 products (bitwise a k-ordered fmaf chain). 157 TFLOP/s peak. */
#define MI355_PREC_BF16X3 1  /* each operand split into 2 bf16 planes (hi + lo), 3 bf16-MFMA products hi*hi + hi*lo + lo*hi:
                                product error <= ~2^-16 relative ("3xBF16 fp32 emulation"). 2.5 PFLOP/s / 3 peak. */
#define MI355_PREC_BF16X6 2  /* 3 planes, the 6 products of order <= 2: ~2^-23, fp32-class. 2.5 PFLOP/s / 6 peak. */
#define MI355_PREC_BF16 3    /* operands rounded to bf16, one product: autocast-style mixed precision. 2.5 PFLOP/s peak. */
#define MI355_PREC_F16 4     /* operands rounded to IEEE fp16 (round to nearest even), one v_mfma_f32_32x32x16_f16 product, fp32 accumulate:
                                the arithmetic of torch.cuda.amp.autocast (fp16) around the reference's AutocastUNet
                                (models/pytorch/segmentation/unet.py:53-58); fp32 tensors only. 2.5 PFLOP/s peak. */

/* Output-side layout of the conv kernels. */
#define MI355_OUT_PLAIN 0
#define MI355_OUT_D2S 1        /* kd == 1 only: y is a FINE tensor (2d,2h,2w,c); logical output channel p*c + k of coarse voxel
                                  (z,y,x) is stored at y[2z+a,2y+b,2x+e,k]: ConvTranspose3d(k2,s2,bias=False) forward as one
                                  GEMM with 8c logical output channels. For mi355_conv3d_wgrad the same flag says dy is such a
                                  fine tensor (wgrad of the transposed conv). y->d/h/w are the FINE extents. */

typedef struct mi355_conv_desc {
  int32_t kd;        /* cubic kernel extent: 1 or 3 */
  int32_t stride;    /* 1 or 2 (applies to IN_PLAIN / IN_AFFINE_ACT) */
  int32_t pad;       /* front zero padding per axis (kd/2 for the reference's convs) */
  int32_t in_mode;   /* MI355_IN_* */
  float act_slope;   /* IN_AFFINE_ACT: 0 -> ReLU (myronenko.py:13), 0.01 -> LeakyReLU (DynUNet). Must lie in [0, 1] (so must every entry of
                        in_slope): the kernels evaluate act(u) as max(u, slope * u) */
  const float* in_scale; /* [n][cin]  IN_AFFINE_ACT */
  const float* in_shift; /* [n][cin]  IN_AFFINE_ACT */
  const float* bias;     /* [cout] or NULL (ConvTranspose3d / DynUNet output block keep a bias) */
  const void* residual;  /* NULL or NDHWC tensor with the logical output shape AND y's storage type, added in the epilogue (myronenko.py:56) */
  int32_t residual_ld;
  const float* out_chscale; /* NULL or [n][cout]: per-(n,channel) scale after the residual add = Dropout3d mask/(1-p) (myronenko.py:78-79) */
  /* output window: logical output voxel (z,y,x) is stored at (z+off_z, y+off_y, x+off_x) of y when that
     lies inside y's (d,h,w); this is F.pad with positive (zero fill, caller pre-zeroes) or negative (crop)
     amounts, unet.py:34-40. */
  int32_t off_z, off_y, off_x;
  int32_t out_d, out_h, out_w; /* logical output extent (before the window shift) */
  const float* in_slope; /* NULL or [cin]: per-input-channel negative slope overriding act_slope (IN_AFFINE_ACT). A concat buffer
                            whose first channels are a raw tensor (slope 1, scale 1, shift 0 = identity) and whose last channels
                            are a normalised+LeakyReLU'd skip (MONAI UnetUpBlock: cat((up, skip), 1) -> conv) needs this. */
  int32_t out_mode;      /* MI355_OUT_* */
  int32_t precision;     /* MI355_PREC_*: arithmetic of the 3x3x3 stride-1 convolutions (everything else is always F32) */
  int32_t wformat;       /* MI355_W_*: what the `wp` argument of mi355_conv3d_fwd points at */
  /* ---- norm statistics fused into the epilogue (myronenko.py:17-21: the conv's output is the next block's GroupNorm input) ---- */
  float* moments_out;    /* NULL or [n][B][cout][3]: per-(sample, spatial tile, channel) partial moments (count, sum, M2 about
                            sum/count) of the values this call stores, B = mi355_conv3d_stats_blocks(x, y, desc). mi355_gn_finalize
                            turns them into the next conv's scale/shift: the statistics pass over the tensor disappears. */
  const struct mi355_gn_bwd_fuse* gn_bwd; /* NULL or: this call is a dgrad whose output dA is the gradient wrt act(GroupNorm(gx)); the
                            epilogue also emits the partial sums of mi355_gn_act_bwd (see mi355_gn_bwd_fuse) */
} mi355_conv_desc;

/* Backward-side fusion: y = dA (gradient wrt the activated, normalised tensor). With u = scale*gx + shift, du = dA*act'(u),
 * xhat = (gx - mean)*rstd the epilogue writes partials_out[n][B][c][2] = (sum du, sum du*xhat) over each spatial tile -- the
 * first of the two passes of mi355_gn_act_bwd, with dA still in registers. gx has the logical shape of y. */
typedef struct mi355_gn_bwd_fuse {
  const void* gx; int32_t gx_ld;          /* the normalised tensor, in the storage type of y (its gradient) */
  const float* scale; const float* shift; /* [n][c] (mi355_gn_stats / mi355_gn_finalize) */
  const float* mean_rstd;                 /* [n][groups][2] */
  int32_t groups; float act_slope;
  float* partials_out;                    /* [n][B][c][2], B = mi355_conv3d_stats_blocks */
} mi355_gn_bwd_fuse;

#define MI355_W_PACKED 0   /* a pack made by mi355_pack_conv_weight / mi355_pack_conv_weight_bf16 (see mi355_conv3d_uses_bf16) */
#define MI355_W_OIDHW4 1   /* the UNPACKED Conv3d weight [cout][4][3][3][3] of a 4-input-channel 3x3x3 stride-1 pad-1 conv
                              (x->c == 4, IN_PLAIN / IN_AFFINE_ACT, OUT_PLAIN): the network's first layer runs on a dedicated
                              exact-fp32 kernel whose GEMM K index is the fused (tap, ci) pair (csrc/conv3d_c4.hip), whatever
                              `precision` says. mi355_conv3d_wgrad takes that path by itself whenever x->c == 4. */
#define MI355_W_PACKED_F32_NARROW 2 /* the fp32 pack of mi355_pack_conv_weight, consumed by the narrow-output kernel whatever
                              `precision` says: 3x3x3 stride-1 pad-1 conv with y->c <= 4 output channels, IN_PLAIN, OUT_PLAIN, no
                              bias / residual / channel scale / window (the dgrad of the 4-channel first layer, which feeds the
                              gamma/beta gradients of the network's first norm, myronenko.py:17-21). One thread per output voxel
                              on the vector ALU (exact fp32 FMA chains) instead of a 32-wide MFMA N tile that would be 7/8
                              padding. With MI355_W_PACKED and MI355_PREC_F32 such a call takes this kernel by itself. */

/* ---- weight packing -------------------------------------------------------------------------- */
/* Packed layout consumed by mi355_conv3d_fwd: wp[tap][cinP/4][coutP][4], cinP = roundup(cin,8),
 * coutP = roundup(cout,32), zero filled. mode 0: forward pack of a Conv3d weight (OIDHW, resnet.py:12-22).
 * mode 1: dgrad pack of the same weight (taps flipped, roles of ci/co swapped) so that dgrad is run by
 * mi355_conv3d_fwd on dy. mode 2: forward pack of a ConvTranspose3d weight (IODHW, decoder.py:99-102);
 * mode 3: dgrad pack of a ConvTranspose3d weight (= plain stride-2 correlation of dy). */
size_t mi355_packed_weight_elems(int32_t cout, int32_t cin, int32_t kd, int32_t mode);
int mi355_pack_conv_weight(const float* w, float* wp, int32_t cout, int32_t cin, int32_t kd, int32_t mode, void* stream);

/* All fp32 / Winograd packs of a training step in ONE launch (the reference re-reads nn.Conv3d.weight on every call,
 * unet3d/models/pytorch/classification/resnet.py:12-17; here the kernel-layout copies are refreshed once per optimizer step, and ~70
 * launches of 5-12 us are 0.6 ms of launch latency per step). `tasks` is a DEVICE array: a caller whose weight and pack buffers keep
 * their addresses (a training loop) builds it once. kind MI355_PACK_F32: out = mi355_pack_conv_weight(w, cout, cin, kd, mode);
 * MI355_PACK_WINO: out = mi355_wino_pack_weight(w, cout, cin, mode) (kd ignored). The launch is cut into chunks of MI355_PACK_CHUNK work
 * items -- F32: one output element (mi355_packed_weight_elems of them); WINO: one (dz, ci, co) triple, 16 outputs from 9 weights
 * (mi355_wino_weight_elems / 16 of them). `first_chunk` of a task = the chunks (ceil(items / MI355_PACK_CHUNK)) of all tasks before
 * it, `total_chunks` = the sum over all tasks. */
#define MI355_PACK_F32 0
#define MI355_PACK_WINO 1
#define MI355_PACK_WINO3 2    /* out = mi355_wino3d_pack_weight(w, cout, cin, mode) (kd ignored); work items = roundup(cin, 4) * roundup(cout, 32): one
                                 (ci, co) pair, 64 outputs from 27 weights */
#define MI355_PACK_LP 16      /* kind = MI355_PACK_LP + precision (MI355_PREC_BF16X3 .. MI355_PREC_F16): out = mi355_pack_conv_weight_bf16(w, ..., precision),
                                 modes 0 / 1, kd 3; work items = kd^3 * roundup(cin, 16) * roundup(cout, 32) */
#define MI355_PACK_CHUNK 1024
typedef struct {
  const float* w; float* out;
  int32_t cout, cin, kd, mode, kind, first_chunk;
} mi355_pack_task;
int mi355_pack_weights_batch(const mi355_pack_task* tasks, int32_t ntasks, int32_t total_chunks, void* stream);

/* Packed layouts of the bf16 paths: [tap][cinP/8][plane][coutP][8] bf16, cinP = roundup(cin,16), planes per `precision`.
 * Same `mode` / role conventions as mi355_pack_conv_weight (modes 0 and 1, kd == 3). mi355_conv3d_uses_bf16 tells the caller
 * which pack a given conv call consumes (1: the bf16 pack of desc->precision, 0: the fp32 pack). */
size_t mi355_packed_weight_bytes_bf16(int32_t cout, int32_t cin, int32_t kd, int32_t precision);
int mi355_pack_conv_weight_bf16(const float* w, void* wp, int32_t cout, int32_t cin, int32_t kd, int32_t mode, int32_t precision, void* stream);
int mi355_conv3d_uses_bf16(const mi355_conv_desc* desc);

/* ---- convolution ------------------------------------------------------------------------------ */
/* Replaces torch.nn.Conv3d.forward (F.conv3d) for k in {1,3}, stride in {1,2}, bias-free or biased, with the
 * GroupNorm-apply + ReLU prologue and the residual / dropout epilogue fused; also runs dgrad (packed mode 1/3)
 * and ConvTranspose3d forward (packed mode 2 + MI355_IN_ZERO_INSERT). fp32 in, fp32 accumulate on
 * v_mfma_f32_32x32x2_f32 (exact fp32). */
int mi355_conv3d_fwd(const mi355_act* x, const float* wp, const mi355_act* y, const mi355_conv_desc* desc, void* stream);

/* Number of spatial tiles per sample (B) that mi355_conv3d_fwd will write partial statistics for (desc->moments_out /
 * desc->gn_bwd->partials_out), or 0 when this call cannot fuse them (windowed / depth-to-space outputs, the narrow kernel):
 * the caller then runs mi355_gn_stats / the unfused mi355_gn_act_bwd instead. A non-NULL desc->gn_bwd (its fields are not read) asks
 * for the norm-backward sums specifically: a kernel may carry the moments epilogue but not that one (the 16-bit plane-ring kernel on
 * 33..64 input channels). */
int32_t mi355_conv3d_stats_blocks(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* desc);

/* Writes the (demangled) name of the kernel instantiation mi355_conv3d_fwd launches for this problem, e.g.
 * "conv3d_mfma<3, 1, 4, 8, 8, 16, 4, 4, 1, 2, 1, 1>", so HIP-event timings can be matched to a rocprofv3 trace. */
int mi355_conv3d_fwd_config(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* desc, char* out, size_t n);

/* Weight gradient dw[co][ci][tap] (OIDHW, written not accumulated) =
 *   sum_{n,q} dy[n,q,co] * in(x)[n, stride*q + tap - pad, ci]           (autograd of F.conv3d wrt weight)
 * with the same input-side fusion as the forward (desc->in_mode PLAIN or AFFINE_ACT). Two-pass, deterministic:
 * per-workgroup partial slabs in `ws` then a reduction. ws_bytes from mi355_conv3d_wgrad_workspace. */
size_t mi355_conv3d_wgrad_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* desc);
int mi355_conv3d_wgrad(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* desc,
                       void* ws, size_t ws_bytes, void* stream);

/* ---- GroupNorm / InstanceNorm ----------------------------------------------------------------- */
/* Statistics of torch.nn.GroupNorm(G, C, eps, affine) (myronenko.py:23-31) / InstanceNorm3d (G == C):
 * writes mean_rstd[n][g][2] and the folded per-(n,c) scale = gamma*rstd, shift = beta - mean*gamma*rstd
 * that the next conv applies in its prologue. ws: >= mi355_gn_workspace(x) bytes. */
size_t mi355_gn_workspace(const mi355_act* x);
int mi355_gn_stats(const mi355_act* x, int32_t groups, float eps, const float* gamma, const float* beta,
                   float* mean_rstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream);

/* Partial moments of x in the format of mi355_conv_desc.moments_out: out[n][B][c][3], B = mi355_gn_moments_blocks(x). One streaming
 * read; for tensors no conv epilogue produced (the network input, the trilinear-upsampled half of a concat buffer). */
int32_t mi355_gn_moments_blocks(const mi355_act* x);
int mi355_gn_moments(const mi355_act* x, float* out, void* stream);
/* Folds epilogue records in[n][blocks][c][k] (k = 3: moment records, k = 2: the norm-backward sums) into out[n][out_blocks][c][k],
 * each output record the in-order merge of a contiguous run of input records (same format, deterministic): a 128^3 layer leaves 8192
 * records per channel, too many for the one-workgroup-per-(sample, group) finalisation kernels to walk. out_blocks <= blocks. */
int mi355_gn_records_reduce(const float* in, int32_t n, int32_t blocks, int32_t c, int32_t k, float* out, int32_t out_blocks, void* stream);
/* Statistics from partial moments: the channels [0, c_a) of the normalised tensor come from part_a[n][blocks_a][c_a][3] and, when
 * part_b != NULL, the channels [c_a, c_a + c_b) from part_b[n][blocks_b][c_b][3] (a concat buffer whose halves were written by two
 * producers, unet.py:42). Combined in double in a fixed order (deterministic). Outputs as mi355_gn_stats. */
int mi355_gn_finalize(const float* part_a, int32_t blocks_a, int32_t c_a, const float* part_b, int32_t blocks_b, int32_t c_b,
                      int32_t n, int32_t groups, float eps, const float* gamma, const float* beta,
                      float* mean_rstd, float* scale, float* shift, void* stream);

/* Backward of act(GroupNorm(x)) given dA = dLoss/d(act output):
 *   du = dA * act'(u), dgamma[c] = sum du*xhat, dbeta[c] = sum du,
 *   dx = rstd*(gamma*du - mean_g(gamma*du) - xhat*mean_g(gamma*du*xhat)) (+ addend if not NULL)
 * dgamma/dbeta are written (not accumulated). dx may alias dA. */
int mi355_gn_act_bwd(const mi355_act* x, const mi355_act* dA, const mi355_act* dx, const void* addend, int32_t addend_ld,
                     int32_t groups, float act_slope, const float* gamma, const float* mean_rstd,
                     const float* scale, const float* shift, float* dgamma, float* dbeta,
                     void* ws, size_t ws_bytes, void* stream);
/* The same with the first pass already done by the dgrad conv that produced dA (mi355_gn_bwd_fuse): partials[n][blocks][c][2]. */
int mi355_gn_act_bwd_fused(const mi355_act* x, const mi355_act* dA, const mi355_act* dx, const void* addend, int32_t addend_ld,
                           int32_t groups, float act_slope, const float* gamma, const float* mean_rstd,
                           const float* scale, const float* shift, float* dgamma, float* dbeta,
                           const float* partials, int32_t blocks, void* ws, size_t ws_bytes, void* stream);

/* ---- upsample + pad + concat ------------------------------------------------------------------ */
/* F.interpolate(scale_factor=2, mode="trilinear", align_corners=False) (decoder.py:105-106) followed by
 * F.pad to the skip size (unet.py:34-40) and the "upsampled channels first" half of torch.cat (unet.py:42):
 * writes channels [0, lo.c) of `cat` (cat->c == lo->c, cat->ld = total channels); voxels of cat outside the
 * shifted window are zero filled. off_* = diff//2 per axis (python floor division, may be -1 = crop). */
int mi355_upsample2x_fwd(const mi355_act* lo, const mi355_act* cat, int32_t off_z, int32_t off_y, int32_t off_x, void* stream);
/* Transposed operation: dlo (written) from the channel slice dcat. */
int mi355_upsample2x_bwd(const mi355_act* dcat, const mi355_act* dlo, int32_t off_z, int32_t off_y, int32_t off_x, void* stream);

/* ---- layout at the module boundary ------------------------------------------------------------ */
/* NCDHW contiguous (the reference's layout, training_utils.py:109) <-> NDHWC view. */
int mi355_ncdhw_to_ndhwc(const float* src, const mi355_act* dst, void* stream);
int mi355_ndhwc_to_ncdhw(const mi355_act* src, float* dst, void* stream);
/* y = a + b (NDHWC views with possibly different ld); y may alias a. */
int mi355_add(const mi355_act* a, const mi355_act* b, const mi355_act* y, void* stream);
/* y[n,v,c] = x[n,v,c] * chscale[n][c]  (Dropout3d backward / standalone forward). y may alias x. */
int mi355_chscale(const mi355_act* x, const float* chscale, const mi355_act* y, void* stream);
/* y = x in y's storage type (fp32 <-> bf16, round to nearest even; same types: a strided copy). Same logical shape. Replaces
 * tensor.to(dtype): the 16-bit copy of the network input the first block's 1x1x1 shortcut reads under activation_storage="bf16", and the
 * bridge around an entry point that has no kernel for a storage type. */
int mi355_cast(const mi355_act* x, const mi355_act* y, void* stream);

/* ---- 1x1x1 projection to a few classes -------------------------------------------------------- */
/* final_convolution (variational.py:59-60; unet.py:50) and DynUNet's output block: Conv3d(cin -> cout<=8, k=1),
 * reading NDHWC and writing the logits directly in the reference's NCDHW layout. w is OIDHW = [cout][cin].
 * in_scale/in_shift ([n][cin], both or neither NULL) + act_slope: the same fused norm-apply + (Leaky)ReLU prologue as the
 * conv kernels (DynUNet's output block reads InstanceNorm3d -> LeakyReLU of the last up block). */
int mi355_proj_fwd(const mi355_act* x, const float* in_scale, const float* in_shift, float act_slope, const float* w,
                   const float* bias, float* logits_ncdhw, int32_t cout, void* stream);
/* backward: dx (NDHWC, written; gradient wrt the ACTIVATED input when a prologue is given), dw[cout][cin] and dbias[cout]
 * (written). ws >= mi355_proj_workspace bytes. */
size_t mi355_proj_workspace(const mi355_act* x, int32_t cout);
int mi355_proj_bwd(const mi355_act* x, const float* in_scale, const float* in_shift, float act_slope, const float* w,
                   const float* dlogits_ncdhw, const mi355_act* dx, float* dw, float* dbias, int32_t cout,
                   void* ws, size_t ws_bytes, void* stream);

/* ---- sliding-window inference ------------------------------------------------------------------ */
/* Accumulation step of MONAI's SlidingWindowInferer, which the reference builds from config["inference"]
 * (unet3d/scripts/script_utils.py:290-293) and calls as inferer(images, model) at unet3d/train/training_utils.py:106-107 and
 * unet3d/predict/volumetric.py:147-148: for one window with origin (z0,y0,x0)
 *   out[c, z0+z, y0+y, x0+x] += w[z,y,x] * pred[c,z,y,x] ;  count[z0+z, y0+y, x0+x] += w[z,y,x]
 * pred: NCDHW window prediction [c][rd][rh][rw]; w: importance map [rd][rh][rw] (constant or Gaussian); out: [c][D][H][W];
 * count: [D][H][W]. Windows are accumulated one launch at a time on the stream (deterministic, no atomics). */
int mi355_sw_accumulate(const float* pred, const float* importance, float* out, float* count, int32_t c,
                        int32_t rd, int32_t rh, int32_t rw, int32_t D, int32_t H, int32_t W,
                        int32_t z0, int32_t y0, int32_t x0, void* stream);
/* Batched form (one launch per window BATCH instead of per window): starts = device int32 [nw][4] = (sample, z0, y0, x0).
 * mi355_sw_gather: windows[w][c][rd][rh][rw] = volume[sample_w][c][z0_w + z][y0_w + y][x0_w + x] (volume: [n][c][D][H][W]) -- the
 * `sw_batch_size` network inputs of MONAI's loop in one pass. mi355_sw_accumulate_batch: folds the nw predictions pred[w][c][...]
 * into out [n][c][D][H][W] / count [n][D][H][W]: one thread per output voxel walks the windows in order, so overlapping windows of a
 * batch are race-free and the sums are bitwise those of nw sequential mi355_sw_accumulate calls. */
int mi355_sw_gather(const float* volume, int32_t n, int32_t c, int32_t D, int32_t H, int32_t W, const int32_t* starts, int32_t nw,
                    int32_t rd, int32_t rh, int32_t rw, float* windows, void* stream);
int mi355_sw_accumulate_batch(const float* pred, const float* importance, float* out, float* count, int32_t n, int32_t c,
                              int32_t rd, int32_t rh, int32_t rw, int32_t D, int32_t H, int32_t W, const int32_t* starts, int32_t nw,
                              void* stream);
/* out[c][v] /= count[v] (final normalisation by the accumulated importance). */
int mi355_sw_normalize(float* out, const float* count, int32_t c, int64_t voxels, void* stream);

/* ---- steps either side of the network, on the device ---------------------- */
/* After the network: activation of unet3d/predict/volumetric.py:151-156 (activation 0 none | 1 sigmoid | 2 softmax over c) and the
 * label-map decode of unet3d/utils/one_hot.py:44-118 in one pass over the logits [c][voxels] (one sample, NCDHW):
 * hierarchy != 0 -> convert_one_hot_to_label_map_using_hierarchy (:101-118); else threshold mask (any > thr, or sum > thr when
 * sum_then_threshold) + argmax (:68-92). probs ([c][voxels]) and label_map (int16 [voxels]) are optional outputs; labels: device
 * int16[c]. c <= 16. */
int mi355_postprocess(const float* logits, int32_t c, int64_t voxels, int32_t activation, float threshold, const int16_t* labels,
                      int32_t hierarchy, int32_t sum_then_threshold, float* probs, int16_t* label_map, void* stream);
/* Before the network: compile_one_hot_encoding (unet3d/utils/one_hot.py:7-37, used by transforms/one_hot.py:7-16): channel g of
 * out (uint8 [c][voxels]) is 1 where round(label_map) is close (atol 1e-8, rtol 1e-5) to any of the values
 * label_values[group_offsets[g] .. group_offsets[g+1]) (device arrays). */
int mi355_one_hot(const float* label_map, int64_t voxels, const float* label_values, const int32_t* group_offsets, int32_t c,
                  uint8_t* out, void* stream);
/* MONAI NormalizeIntensityD(channel_wise=True, nonzero=False) (unet3d/datasets/segmentation.py:77-86; brats2020_config.json:140-144):
 * y[c][v] = (x[c][v] - mean_c) / std_c with the population std, std 0 -> 1. x, y: [c][voxels] (one sample). */
size_t mi355_zscore_workspace(int32_t c);
int mi355_zscore(const float* x, float* y, int32_t c, int64_t voxels, void* ws, size_t ws_bytes, void* stream);
/* Resampling of one channel-first sample through an affine voxel map:
 *   dst[c][z][y][x] = interp(src[c], M * (z, y, x, 1)),  M = 3x4 row-major HOST array, rows = source (z, y, x) coordinate.
 * Replaces MONAI ResizeD(spatial_size, mode=("trilinear","nearest")) = F.interpolate(size=..., align_corners=False)
 * (unet3d/datasets/segmentation.py:63-68: M = diag(in/out) with offset 0.5*in/out - 0.5 for trilinear, offset 0 + FLOOR for
 * torch's "nearest") and the ResampleToMatch of the predictions back onto the source grid (unet3d/predict/volumetric.py:135-136,
 * 168-170: M = inv(A_src) * A_dst in voxel coordinates). mode: MI355_RESAMPLE_*. padding 0: border (coordinates clamped to the
 * volume, what F.interpolate does), 1: zeros outside. */
#define MI355_RESAMPLE_TRILINEAR 0
#define MI355_RESAMPLE_NEAREST 1        /* round half to even (grid_sample nearest) */
#define MI355_RESAMPLE_NEAREST_FLOOR 2  /* floor (F.interpolate nearest) */
int mi355_resample_affine(const float* src, float* dst, int32_t c, int32_t sd, int32_t sh, int32_t sw, int32_t dd, int32_t dh,
                          int32_t dw, const float* m, int32_t mode, int32_t padding, void* stream);

/* ---- first-layer backward in one pass over dy (csrc/conv3d_c4_bwd.hip, round 6) ----------------------------------------------------------
 * The 4 -> 32 channel 3x3x3 conv of BASELINE.json's "128^3 x 4ch" line (unet3d/models/pytorch/classification/myronenko.py:17-21 via
 * resnet.py:12-17) sits behind the network's first GroupNorm + ReLU (myronenko.py:9-15), and the network input needs no gradient: the
 * backward of the pair is dW of the conv and (dgamma, dbeta) of the norm. mi355_conv3d_c4_bwd produces dW and the norm-backward records
 * (gn_fuse.h format, [n][mi355_conv3d_c4_bwd_blocks(x)][4][2]) reading dy ONCE and never writing the data gradient -- it replaces
 * mi355_conv3d_wgrad + mi355_conv3d_fwd(dgrad pack) + mi355_gn_act_bwd for this layer (ATen: conv backward + native_group_norm_backward);
 * mi355_gn_bwd_params turns the records into dgamma / dbeta. x: the fp32 network input (4 channels); dy: 32 channels, fp32 / bf16 / fp16
 * storage (exact fp32 arithmetic on the stored values in every precision mode, as the calls it replaces); desc: the
 * forward conv's descriptor (norm prologue: in_mode, in_scale, in_shift, act_slope / in_slope); wp_dgrad: mi355_pack_conv_weight(mode 1);
 * mean_rstd / groups: mi355_gn_stats of x. MI355_EUNSUPPORTED for any other shape (use the three calls). */
int mi355_conv3d_c4_bwd_supported(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* desc);
int32_t mi355_conv3d_c4_bwd_blocks(const mi355_act* x);
size_t mi355_conv3d_c4_bwd_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* desc);
int mi355_conv3d_c4_bwd(const mi355_act* x, const mi355_act* dy, const float* wp_dgrad, float* dw, const mi355_conv_desc* desc,
                        const float* mean_rstd, int32_t groups, float* partials_out, void* ws, size_t ws_bytes, void* stream);
/* dgamma / dbeta of a norm from partial records alone (ws: mi355_gn_workspace(x) bytes) */
int mi355_gn_bwd_params(const mi355_act* x, int32_t groups, const float* gamma, const float* mean_rstd, float* dgamma, float* dbeta,
                        const float* partials, int32_t blocks, void* ws, size_t ws_bytes, void* stream);

/* ---- Winograd F(2x2, 3x3) x direct-z form of the 3x3x3 stride-1 convolution (csrc/conv3d_wino.hip) --------------------------------
 * The product path of the eligible fp32 forward / dgrad convolutions (round 3; replaces the ATen / cuDNN call behind nn.Conv3d of
 * unet3d/models/pytorch/classification/resnet.py:12-17). Same operation as mi355_conv3d_fwd for kd 3 / stride 1 / pad 1, plain or norm-prologue input, plain un-windowed output (bias, residual, out_chscale and the fused statistics
 * honoured), fp32, 12 instead of 27 multiplications per output and (ci, co). Weights: mi355_wino_pack_weight (mode 0 forward,
 * mode 1 dgrad of Conv3d, as mi355_pack_conv_weight) into mi355_wino_weight_elems(cout, cin) floats. */
size_t mi355_wino_weight_elems(int32_t cout, int32_t cin);
int mi355_wino_pack_weight(const float* w, float* up, int32_t cout, int32_t cin, int32_t mode, void* stream);
int mi355_conv3d_wino_fwd(const mi355_act* x, const float* up, const mi355_act* y, const mi355_conv_desc* desc, void* stream);
/* 1 when mi355_conv3d_wino_fwd accepts this call (shape, mode, strides, alignment of x / y / the fused operands), else 0: a caller that
 * routes by size thresholds asks here and falls back to mi355_conv3d_fwd instead of failing. */
int mi355_conv3d_wino_supported(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* desc);
/* records per sample its fused-statistics epilogues (desc->moments_out / desc->gn_bwd, formats of gn_fuse.h) write: 2 x 8 x 16 voxel tiles */
int32_t mi355_conv3d_wino_stats_blocks(const mi355_act* y);
/* ---- Winograd F(2x2x2, 3x3x3): the same convolution with the transform along z as well (csrc/conv3d_wino3d.hip, round 6) -----------------
 * 8 instead of 12 (27) multiplications per output and (ci, co); the contract, the eligible calls (mi355_conv3d_wino3d_supported: those of
 * mi355_conv3d_wino_supported with at most 1024 input channels) and the statistics records (mi355_conv3d_wino_stats_blocks: one per
 * 2 x 8 x 16 voxel tile) are those of mi355_conv3d_wino_fwd; the weights are packed by mi355_wino3d_pack_weight (64 transform points,
 * input channels padded to 4) into mi355_wino3d_weight_elems(cout, cin) floats. Replaces the same ATen / cuDNN call
 * (unet3d/models/pytorch/classification/resnet.py:12-17). */
size_t mi355_wino3d_weight_elems(int32_t cout, int32_t cin);
int mi355_wino3d_pack_weight(const float* w, float* up, int32_t cout, int32_t cin, int32_t mode, void* stream);
int mi355_conv3d_wino3d_fwd(const mi355_act* x, const float* up, const mi355_act* y, const mi355_conv_desc* desc, void* stream);
int mi355_conv3d_wino3d_supported(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* desc);
/* weight gradient in the same domain, F(3x3, 2x2) x direct z (csrc/conv3d_wgrad_wino.hip; replaces the ATen weight-gradient call behind
 * nn.Conv3d.backward for resnet.py:12-17): contract of mi355_conv3d_wgrad for kd 3 / stride 1 / pad 1. A z-marching plane ring: all
 * three dz per workgroup, every plane staged and transformed once. MI355_EUNSUPPORTED for channel counts that are not multiples
 * of 4 (use mi355_conv3d_wgrad). */
size_t mi355_conv3d_wgrad_wino_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* desc);
int mi355_conv3d_wgrad_wino(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* desc, void* ws, size_t ws_bytes,
                            void* stream);

/* ---- Dice loss -------------------------------------------------------------------------------- */
/* monai.losses.DiceLoss as configured by examples/brats2020/brats2020_config.json:112-116 (sigmoid=True,
 * include_background=True, smooth_nr=smooth_dr=1e-5, reduction="mean"), plus `batch` and `squared_pred`.
 * logits fp32 NCDHW, target uint8 or fp32 NCDHW (target_is_u8). Writes loss[0] and, if dlogits != NULL,
 * dloss/dlogits (scaled by grad_scale). stats: [n*c*3] floats scratch kept for inspection (I, sum p, sum y).
 * variant MI355_DICE_GENERALIZED: monai.losses.GeneralizedDiceLoss (w_type "square"; the loss doc/Configuration.md:41 configures);
 * include_background == 0 leaves channel 0 out (both variants). */
#define MI355_DICE_PLAIN 0
#define MI355_DICE_GENERALIZED 1
size_t mi355_dice_workspace(int32_t n, int32_t c, int64_t voxels);
int mi355_dice_fwd_bwd(const float* logits, const void* target, int32_t target_is_u8, int32_t n, int32_t c, int64_t voxels,
                       int32_t sigmoid, int32_t batch, int32_t squared_pred, int32_t variant, int32_t include_background,
                       float smooth_nr, float smooth_dr, float* loss, float* dlogits, float grad_scale, void* ws, size_t ws_bytes,
                       void* stream);

/* The options of monai.losses.DiceLoss beyond the shipped configuration (the reference builds the loss from the config's kwargs:
 * unet3d/scripts/script_utils.py:61-77): channel softmax, label-map targets (to_onehot_y), jaccard, per-class weight, reduction.
 * Forward and backward are separate calls: with reduction "none" the upstream gradient is one value per term.
 *   activation: applied to the logits before the sums; softmax runs over ALL channels, then include_background == 0 drops channel 0.
 *   target_kind: F32 / U8 = same shape as the logits; LABELS = int32 [n][voxels] class indices (y_c = label == c).
 *   class_weight: device pointer, one factor per COUNTED class (c, or c-1 without background), or NULL.
 *   loss: 1 value (mean / sum) or one per term (none: [n][counted classes], or [counted classes] with batch).
 *   ws: mi355_dice_workspace(n, c, voxels) bytes, written by forward and read by backward (the caller keeps it).
 *   upstream: device pointer to d(result)/d(loss value[s]) (n_upstream = number of loss values) or NULL for 1. */
#define MI355_DICE_ACT_NONE 0
#define MI355_DICE_ACT_SIGMOID 1
#define MI355_DICE_ACT_SOFTMAX 2
#define MI355_DICE_TARGET_F32 0
#define MI355_DICE_TARGET_U8 1
#define MI355_DICE_TARGET_LABELS 2
#define MI355_DICE_REDUCE_MEAN 0
#define MI355_DICE_REDUCE_SUM 1
#define MI355_DICE_REDUCE_NONE 2
typedef struct mi355_dice_opts {
  int32_t activation, target_kind, batch, squared_pred, include_background, jaccard, reduction;
  float smooth_nr, smooth_dr;
  const float* class_weight;
} mi355_dice_opts;
int mi355_dice_ex_forward(const mi355_dice_opts* opts, const float* logits, const void* target, int32_t n, int32_t c, int64_t voxels,
                          float* loss, void* ws, size_t ws_bytes, void* stream);
int mi355_dice_ex_backward(const mi355_dice_opts* opts, const float* logits, const void* target, int32_t n, int32_t c, int64_t voxels,
                           const float* upstream, int32_t n_upstream, float* dlogits, const void* ws, void* stream);

/* ---- cross-entropy --------------------------------------------------------------------------- */
/* The cross-entropy leg of the reference's loss look-up (scripts/script_utils.py:61-77: torch.nn.BCEWithLogitsLoss /
 * CrossEntropyLoss via the torch fallback, monai.losses.DiceCELoss), fused into one pass over the logits: loss value and
 * d(loss)/d(logits) together.
 *   mode MI355_CE_SOFTMAX: torch.nn.CrossEntropyLoss(reduction="mean") with PROBABILITY targets (what DiceCELoss passes for a
 *        one-hot target): loss = -1/(n*voxels) * sum_{n,v} sum_c y_c * log_softmax(z)_c.
 *   mode MI355_CE_BCE:     torch.nn.BCEWithLogitsLoss(reduction="mean"): mean over n*c*voxels of softplus(z) - y*z.
 * logits fp32 NCDHW, target uint8 or fp32 NCDHW. loss[0] = weight * CE (or += when accumulate_loss != 0, e.g. after
 * mi355_dice_fwd_bwd); dlogits (optional) receives weight * grad_scale * dCE/dlogits, ADDED to its contents when
 * accumulate_grad != 0. c <= 16. */
#define MI355_CE_SOFTMAX 0
#define MI355_CE_BCE 1
size_t mi355_ce_workspace(int64_t voxels);
int mi355_ce_fwd_bwd(const float* logits, const void* target, int32_t target_is_u8, int32_t n, int32_t c, int64_t voxels, int32_t mode,
                     float weight, float* loss, int32_t accumulate_loss, float* dlogits, int32_t accumulate_grad, float grad_scale,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- optimizer -------------------------------------------------------------------------------- */
/* torch.optim.Adam(lr, betas, eps, weight_decay=0, amsgrad=False).step() (script_utils.py:80-81) over one flat
 * parameter buffer; grad is multiplied by grad_scale first (1/world_size after the RCCL sum). step is 1-based.
 * Hyper-parameters are doubles, as torch's Python scalars are: the fp32 factors the kernel multiplies by are float(1 - beta),
 * float(lr / (1 - beta1^step)), ... rounded ONCE from the double expression, exactly what torch hands its kernels
 * (float(1 - 0.999) and 1.f - float(0.999) differ by 1.3e-5 relative; tests/test_launch_audit.py found it). */
int mi355_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                    double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step, float grad_scale, void* stream);

/* Library / build identification: returns e.g. "mi355_unet3d gfx950 <version>". */
const char* mi355_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MI355_UNET3D_H */
