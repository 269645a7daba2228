/*
 * qdiff_hip.h — C ABI of libqdiff_hip.so, the MI355X (gfx950) integer engine that sits under
 * the q-diffusion `qdiff` Python API.
 *
 * The reference (Xiuyu-Li/q-diffusion) has no FFI layer: its hot path is a chain of ATen calls
 * issued from qdiff/quant_layer.py, qdiff/adaptive_rounding.py and qdiff/quant_block.py.  Each
 * entry point below replaces one of those implicit op sequences and cites
 * the reference lines whose arithmetic it reproduces.
 *
 * Conventions
 *   - plain C: raw device pointers, ints, POD structs; no torch types.  `stream` is a hipStream_t
 *     passed as void*.  All kernels are enqueued on `stream` and return without synchronising
 *     (safe inside hipGraph capture: no allocation, no sync, no host read-back).
 *   - every function returns 0 on success, non-zero on error; qd_last_error() returns a
 *     thread-local message.  Nothing falls back to the host: a bad argument is an error.
 *   - the library is stateless (no global mutable state besides the thread-local error text).
 *   - quantisation parameters that the reference keeps as tensors / nn.Parameters (delta,
 *     zero_point) are read by the kernels from DEVICE memory, so the host never has to .item() them.
 *
 * Integer conventions
 *   activation code  q  = clamp(rint(x/delta) + zp, qmin, qmax)        (quant_layer.py:82-87)
 *   stored byte      a' = q - off,  off = 128 for unsigned 8-bit grids, 0 for signed grids
 *   "true zero"      z' = zp - off  (the byte that dequantises to 0.0; used for conv padding)
 *   weight code      W  = clamp(floor(w/dw)+(alpha>=0)+zw, 0, 2^b-1)   (adaptive_rounding.py:49-59)
 *   stored weight    w' = W (int4 path: the raw nibble, unpacked in-kernel) or W - 128 (int8 path); the
 *                    per-channel remainder zw resp. zw-128 is restored in the epilogue through activation
 *                    row sums (qd_conv_seg.zw).
 */
#ifndef QDIFF_HIP_H
#define QDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QD_ABI_VERSION 20

/* element types of floating-point tensors crossing the ABI */
enum { QD_F32 = 0, QD_F16 = 1, QD_BF16 = 2 };
/* epilogues of qd_conv2d_i8 */
enum { QD_EPI_LINEAR = 0, QD_EPI_GEGLU_I8 = 1, QD_EPI_HEADS_I8 = 2, QD_EPI_HEADS_T_I8 = 3 };

int         qd_abi_version(void);
const char* qd_last_error(void);
/* 1 if a gfx950 device is visible to this process, else 0 (never throws). */
int         qd_device_ok(void);

/* Box calibration (ABI v20; measurement aid of bench.py, not on the product path): `blocks` workgroups of 256 threads run `iters`
 * iterations of kind 0: 8 dense v_mfma_i32_32x32x32_i8 per wave (four independent accumulators; 8 * 65536 integer ops each), or
 * kind 1: 32 v_exp_f32 per wave (four independent registers).  ticks[block] = elapsed shader-clock ticks of the block's first wave;
 * the caller times the launch (HIP events): 512 blocks = two waves per SIMD, 1024 = four.  sink: one int the kernel never writes. */
int qd_box_probe(int kind, int blocks, int iters, long long* ticks, int* sink, void* stream);

/* ------------------------------------------------------------------------------------------
 * Quantiser parameters.  Every entry point below that takes `qparams` / `oq_params` expects a device float[4]
 *     {delta, zero_point, rinv, fast} produced by qd_make_qparams from the reference's per-tensor delta / zero_point
 *     (quant_layer.py:66-89; both read from DEVICE memory, no host round trip).  rinv is the correctly rounded
 *     reciprocal of delta; fast != 0 certifies — by an exhaustive check over all 2^23 mantissas — that
 *     y = x*rinv; e = fma(-y, delta, x); q = fma(e, rinv, y) equals the IEEE quotient x / delta bit for bit for this
 *     delta, so the kernels may use it instead of the division that torch.round(x / delta) semantics require;
 *     otherwise they divide.  Codes are bit-identical either way.
 * ------------------------------------------------------------------------------------------ */
int qd_make_qparams(const float* delta, const float* zero_point, float* out4, void* stream);

/* ------------------------------------------------------------------------------------------
 * K1  activation quantiser.   Replaces UniformAffineQuantizer.forward on an *input activation*
 *     (qdiff/quant_layer.py:66-89) and the two-quantizer + torch.cat split path of
 *     QuantModule.forward (quant_layer.py:256-264).
 *
 *     x is a logical [B][C][S] tensor addressed by element strides (sb, sc, ss): NCHW-contiguous
 *     is (C*S, S, 1), channels_last / token-major is (S*C, 1, C).  Channels [c0, c0+clen) are
 *     quantised with qparams (qd_make_qparams) and written as
 *     int8 to out[(b*S+s)*ldo + oc0 + (c-c0)]; channels up to clen_pad are filled with the
 *     "true zero" byte so that padded K lanes contribute nothing.
 *     qmin/qmax/off describe the integer grid (see header comment).
 * ------------------------------------------------------------------------------------------ */
int qd_quantize_act(const void* x, int x_dtype, int64_t B, int64_t C, int64_t S,
                    int64_t sb, int64_t sc, int64_t ss,
                    int c0, int clen, int clen_pad,
                    const float* qparams, int qmin, int qmax, int off,
                    int8_t* out, int64_t ldo, int oc0, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2  weight packer.   Replaces the per-forward weight fake-quant of AdaRoundQuantizer.forward
 *     (qdiff/adaptive_rounding.py:49-61, branch learned_hard_sigmoid / soft_targets=False) and of
 *     UniformAffineQuantizer.forward on weights (quant_layer.py:82-88; round-to-nearest) by a
 *     one-time pack.  w is the fp32 weight viewed as [Cout][Cin_total][taps] (PyTorch OIHW with
 *     HW flattened); the channel slice [c0, c0+clen) is quantised with per-out-channel delta/zp
 *     (fp32 [Cout]) and, when alpha != NULL (same layout as the slice, [Cout][clen][taps]),
 *     AdaRound hard rounding.  Output row n, tap t: bytes [kofs, kofs+clen_pad) of
 *     wq[(n*taps+t)*ldk ...] hold W-128 (bits==8 → mode 8) or W-zw (mode 0: direct s8) ;
 *     mode 4 packs W as nibbles (two per byte, layout in DESIGN.md §4.2) at byte offset kofs/2.
 *     wsum[n] += sum of the stored s8 values (mode 8/0) or of (W-zw) (mode 4) over the slice.
 *     codes (optional, may be NULL): int32 [Cout][clen][taps] raw W codes for tests.
 * ------------------------------------------------------------------------------------------ */
int qd_pack_weights(const float* w, const float* alpha, const float* delta, const float* zp,
                    int Cout, int Cin_total, int taps, int c0, int clen, int clen_pad,
                    int n_levels, int mode, uint8_t* wq, int64_t ldk, int kofs,
                    int32_t* wsum, int32_t* codes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3/K4  integer convolution / GEMM.   Replaces fwd_func(input, weight, bias, **fwd_kwargs) of
 *     QuantModule.forward (qdiff/quant_layer.py:276 → F.conv2d / F.conv1d(k=1) / F.linear on
 *     dequantised fp32 operands) by an implicit GEMM on the integer codes:
 *        M = B*Ho*Wo, N = Cout, K = taps * sum(seg.clen)
 *        out[m][n] = sum_seg scale_s[n] * I_s[m][n] + bias[n] + rowbias[b(m)][n] + residual[m][n]
 *        I_s = acc_s - zc_s[n] - zw_s[n] * (Asum_s[m] - kz_s)
 *     acc_s is the exact int32 MFMA accumulator of stored bytes; zc/zw/kz restore the zero points
 *     (DESIGN.md §3).  Two segments implement the split-shortcut of quant_layer.py:257-269.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t        c0;       /* first channel of the segment inside an x row (bytes)              */
    int32_t        clen;     /* channel count, multiple of 16 (padded)                            */
    int32_t        kofs;     /* reserved (row-major weight layouts of ABI <= 9); ignored                */
    int32_t        kstep0;   /* index of the segment's first 64-wide K-step in the tiled weight array    */
    const int8_t*  wzp;     /* reserved (ABI <= 9); ignored                                              */
    const float*   scale;    /* [Cout]  delta_x * delta_w[n]                                       */
    const int32_t* zc;       /* [Cout]  z' * Wsum[n]            or NULL (symmetric activations)    */
    const int32_t* zw;       /* [Cout]  zw[n]-128 (int8 tiles) | zw[n] (int4 tiles)                          */
    const int32_t* zfill;    /* [2] {z', K_seg*z'} device scalars or NULL (= 0)                    */
    const int8_t*  fill16;   /* 16 bytes of z' — the source of out-of-image taps for the LDS-DMA loader —
                                or NULL (= zeros)                                                  */
} qd_conv_seg;

typedef struct {
    const int8_t*  x;        /* [B][H][W][ldx] stored activation bytes                            */
    const uint8_t* w;        /* MFMA-tile-ordered weights of qd_pack_weights_t4 (wbits=4) / _t8 (wbits=8) */
    void*          out;      /* [M][ldo]                                                           */
    const float*   bias;     /* [Cout] or NULL                                                     */
    const float*   rowbias;  /* [B][ld_rowbias] per-sample per-channel add (timestep emb) or NULL  */
    const void*    residual; /* [M][ldr] same dtype as out, or NULL                                */
    int64_t        ldx, ldk, ldo, ldr, ld_rowbias;
    int32_t        B, H, W, Ho, Wo, Cout;
    int32_t        kh, kw, stride, pad_t, pad_l;
    int32_t        wbits;    /* 8 or 4 (16: qd_conv2d_bf16)                                        */
    int32_t        out_dtype;/* QD_F32 / QD_F16                                                    */
    int32_t        nseg;     /* 1 or 2                                                             */
    int32_t        w_tiled;  /* must be 1 (the row-major layouts of ABI <= 9 are gone)                  */
    int32_t        epilogue; /* QD_EPI_LINEAR, or QD_EPI_GEGLU_I8 (w_tiled only): the weight rows were packed
                                interleaved per 32 (value tile, gate tile, value tile, ...); the epilogue computes
                                value*gelu(gate) (ldm/modules/attention.py:42-44) and writes it QUANTISED with
                                oq_* (the act quantiser of the following Linear) as int8 rows out[M][ldo]        */
    qd_conv_seg    seg[2];
    const float*   oq_params;/* QD_EPI_GEGLU_I8: {delta, zero_point} of the output quantiser (device)             */
    int32_t        oq_min, oq_max, oq_off, _pad2;
    void*          splitk_ws;      /* optional scratch of >= qd_conv2d_i8_splitk_ws_bytes(d) bytes (16-B aligned); when
                                      given, layers too small to fill the chip are contracted split-K: int32 partials
                                      per K range (exact), then one pass that sums them and applies the epilogue.
                                      NULL = never split.  Results do not depend on it (integer partial sums).       */
    int64_t        splitk_ws_bytes;
    /* QD_EPI_HEADS_I8 / QD_EPI_HEADS_T_I8 (w_tiled, one segment): the Linear is a q/k/v projection of an
     * attention block (qdiff/quant_block.py:190-221).  Rows are m = b*hd_T + t, columns n = h*hd_d + dd.
     * y = (I*scale + bias) * oq_prescale is quantised with oq_* (the block's act_quantizer_q/k/v) and
     * written as int8 in the operand layout of qd_attn_i8, exactly what qd_quantize_heads would produce:
     *   HEADS_I8  : out[(b*H+h)][hd_Tpad][hd_dpad]               (transpose=0 layout)
     *   HEADS_T_I8: out[(b*H+h)][hd_dpad][hd_Tpad], key-permuted (transpose=1 layout); hd_sum[(b*H+h)][hd_dpad]
     *               += column sums (caller zeroes hd_sum first).
     * HEADS_I8 also accepts an fp32 `residual` (added before the quantiser): with hd_H = 1, hd_d = Cout it is
     * "Linear + residual -> int8 rows of the next Linear" (transformer FF output feeding SpatialTransformer.proj_out).
 * Pad bytes are not written: out must be zero-initialised once (it can then be reused).
     * hd_T must be a multiple of 128 (a tile of rows never straddles two samples).  Weights: int4 tiles of any width,
     * or int8 tiles with Cout > 64 (the 1x1 q / k / v convolutions of the pixel-space AttnBlock, quant_block.py:354-386). */
    int32_t        hd_H, hd_d, hd_T, hd_Tpad, hd_dpad;
    float          oq_prescale;
    int32_t*       hd_sum;
    /* optional (w_tiled, QD_EPI_LINEAR, fp32 out, Ho*Wo % 128 == 0): the kernel also writes the first level of the
     * GroupNorm statistics of its output, gn_part[b][Ho*Wo/128][Cout][2] = {sum, sum of squares} over each 128-row
     * chunk, which qd_groupnorm_silu_quant accepts as part_in (one pass over the tensor less).  Disables split-K.
     * gn_ld: channels per chunk row of the statistics buffer (0 = Cout).  A value > Cout lets two producers write the
     * statistics of the two halves of a skip concatenation (openaimodel.py:776) into column ranges of one buffer, the
     * same way `out` + `ldo` place their fp32 rows into column ranges of one activation buffer: the concatenation
     * then costs no copy at all. */
    float*         gn_part;
    int64_t        gn_ld;
    /* non-zero: x holds the HALF-resolution map [B][H/2][W/2][ldx] and the convolution runs on its nearest-neighbour 2x
     * up-sampling (Upsample: openaimodel.py:105-120, ddim diffusion.py:36-52) — H, W are the up-sampled sizes; the
     * replication happens in the im2col source address, no up-sampled tensor exists.  Needs stride 1, kh*kw > 1, even H, W. */
    int32_t        upsample2x;
    int32_t        _pad3;
} qd_conv_desc;

int qd_conv2d_i8(const qd_conv_desc* d, void* stream);

/* qd_conv_config (ABI v20): kgroups 1 (default; initial value QD_KGROUPS) = a launch with at most one output tile per CU and at least
 * eight K-steps runs 512-thread blocks whose two groups of four waves contract alternate K-steps and add their int32
 * accumulators in LDS before the (unchanged) epilogue — two waves per SIMD where the four-wave block leaves one; 0 = always the
 * four-wave block; -1 = leave unchanged.  The integers, and therefore the bytes written, are the same either way. */
void qd_conv_config(int kgroups);

/* Grouped launch (ABI v20): n = 1..3 descriptors of ONE problem shape with head-layout epilogues (QD_EPI_HEADS_I8 / _T_I8) — the
 * q / k / v projections of an attention block, which the reference evaluates as three Linears on the rows of one LayerNorm
 * (qdiff/quant_block.py:193-199, ldm attention.py:174-182; the DDIM AttnBlock's q / k / v convolutions diffusion.py:142-144; the
 * three row subsets of the LDM AttentionBlock's qkv conv1d, openaimodel.py:311-326) — as ONE launch whose grid's second
 * dimension selects the descriptor: three 256-block launches (one wave per SIMD each) become one that fills the chip, two
 * prologue ramps and two dependent-launch boundaries disappear.  Members that do not qualify (different shapes, another
 * epilogue, a tile shape the group kernel is not built for) run as the n single launches of qd_conv2d_i8: the bytes written are
 * the same either way. */
int qd_conv2d_i8_group(const qd_conv_desc* const* descs, int n, void* stream);

/* Scratch bytes qd_conv2d_i8 would use for a split-K contraction of this descriptor (shape fields only are
 * read); 0 when the layer is launched unsplit. */
int64_t qd_conv2d_i8_splitk_ws_bytes(const qd_conv_desc* d);

/* K2b  tile-ordered int4 packer for the LDS-DMA contraction kernel (csrc/igemm_dma.hip).  Same code
 *     formula as qd_pack_weights (adaptive_rounding.py:49-59).  Output wt[kstep][ntile][1024 B]:
 *     kstep enumerates (tap, 64-channel step) of the slice starting at kstep0, ntile = n/32, and each
 *     1-KB block is [ksub(2)][half(2)][n%32][8 B] = the 16 nibbles of row n for
 *     K = ksub*32 + half*16 + 0..15 (nibble order inside the 8 bytes as in qd_pack_weights mode 4).
 *     The stored operand is the RAW code W (0..15); its zero point is restored in the epilogue
 *     through qd_conv_seg.zw[n] = zw[n] and activation row sums.  wsum[n] += sum W over the slice. */
int qd_pack_weights_t4(const float* w, const float* alpha, const float* delta, const float* zp,
                       int Cout, int Cin_total, int taps, int c0, int clen, int clen_pad, int n_levels,
                       uint8_t* wt, int kstep0, int ntiles, int32_t* wsum, void* stream);

/* K2c  the same tile order for 8-bit weights: wt[kstep][ntile][2 KB], each block [ksub(2)][half(2)][n%32][16 B] =
 *     the 16 stored bytes W-128 of row n for K = ksub*32 + half*16 + 0..15 (qd_conv_desc.wbits = 8, w_tiled = 1,
 *     qd_conv_seg.zw[n] = zw[n]-128).  wsum[n] += sum (W-128) over the slice. */
int qd_pack_weights_t8(const float* w, const float* alpha, const float* delta, const float* zp,
                       int Cout, int Cin_total, int taps, int c0, int clen, int clen_pad, int n_levels,
                       uint8_t* wt, int kstep0, int ntiles, int32_t* wsum, void* stream);

/* Test hook: same contraction, but writes the raw int32 I_s[m][n] of segment 0 (after zero-point
 * restoration) to iout[M][Cout].  Used for the bit-exact accumulator tests (oracle tier T0). */
int qd_conv2d_i8_acc(const qd_conv_desc* d, int32_t* iout, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6  timestep-embedding MLPs.   Replaces `time_embed` = Linear -> SiLU -> Linear (ldm openaimodel.py:758-759; ddim
 *     diffusion.py:318-320) and the per-ResBlock `emb_layers` / `temb_proj` = SiLU -> Linear on the shared embedding
 *     (openaimodel.py:225-232 with qdiff/quant_block.py:98-107; ddim diffusion.py:127), every Linear a QuantModule
 *     (quant_layer.py:248-279) on M = batch rows: one launch evaluates n_layers Linears that share the input x [B][K]
 *     (fp32, row stride ldx): out[b][layer.out_off + n] = Linear_layer(act_quant_layer(silu?(x)))[b][n].
 *     layers: DEVICE array of n_layers records, 64 bytes each:
 *         { const uint8_t* w (tile-ordered weights, first K-step of the only segment); const float* scale;
 *           const int32_t* zc; const int32_t* zw; const float* bias; const float* qparams (float[4]);
 *           const int32_t* zfill; int32_t Cout; int32_t out_off; }          (semantics of qd_conv_seg / qd_conv_desc)
 *     blocks: DEVICE int32 [n_blocks][2] = {layer index, first output channel}: one workgroup per 64 output channels.
 *     qmin/qmax/off: the (common) activation grid.  B*K must fit LDS (<= ~44 KB): callers split taller batches.
 *     Results equal qd_quantize_act + qd_conv2d_i8 on the same SiLU values bit for bit (same integers, same float order).
 * ------------------------------------------------------------------------------------------ */
int qd_temb_mlp(const float* x, int64_t ldx, int B, int K, int apply_silu, const void* layers, int n_layers,
                const int32_t* blocks, int n_blocks, int wbits, int qmin, int qmax, int off, float* out, int64_t ldo,
                void* stream);

/* ------------------------------------------------------------------------------------------
 * K5  GroupNorm -> SiLU -> quantise.   Replaces nn.GroupNorm(32,C) / GroupNorm32 → x*sigmoid(x)
 *     → act_quantizer of the following QuantModule (ddim/models/diffusion.py:121-123,127-130;
 *     ldm/modules/diffusionmodules/openaimodel.py:201-205,225-232, util.py:214-216;
 *     qdiff/quant_layer.py:82-88).  x is channels-last [B][S][C] (ldx = row stride, elements).
 *     ws: workspace of qd_groupnorm_ws_bytes(B,C,S) bytes.  apply_silu=0 gives GroupNorm→quant
 *     (attention blocks: openaimodel.py:324, attention.py:280-281, ddim diffusion.py:175).
 *     Two-quantizer outputs are not needed here (split only affects 1x1 skips).
 *     If yout != NULL the fp32 normalised (+SiLU) tensor is also written ([B*S][ldy]).
 *     part_in != NULL: the first statistics level was already produced by the kernel that wrote x
 *     (qd_conv_desc.gn_part: [B][nchunk_in][part_ld >= C][2] sums / sums of squares per row chunk; part_ld = 0
 *     means C); the statistics pass over x is skipped.
 *     raw != NULL (and raw->out != NULL): see qd_raw_quant above.
 * ------------------------------------------------------------------------------------------ */
/* Optional second output of qd_groupnorm_silu_quant: the RAW (un-normalised) input quantised for another consumer of
 * the same tensor, i.e. the 1x1 skip connection of a residual block, which reads the very tensor `in_layers` normalises
 * (qdiff/quant_block.py:108-111; openaimodel.py:266-278), with up to two channel segments that carry their own
 * activation quantisers (the split shortcut, quant_layer.py:257-269).  Codes are those of qd_quantize_act, written to
 * out[row][oc0 + (c - c0)] for c in [c0, c0 + clen); c0 / clen / oc0 are multiples of 16.                               */
typedef struct {
    int32_t      c0, clen, oc0;
    int32_t      qmin, qmax, off;
    const float* qparams;
} qd_raw_seg;
typedef struct {
    int8_t*    out;       /* [B*S][ldo] int8 rows of the consumer (NULL: no raw output) */
    int64_t    ldo;
    int32_t    nseg;      /* 1 or 2 */
    int32_t    _pad;
    qd_raw_seg seg[2];
} qd_raw_quant;

int64_t qd_groupnorm_ws_bytes(int64_t B, int64_t C, int64_t S);
int qd_groupnorm_silu_quant(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx,
                            int groups, float eps, const float* gamma, const float* beta,
                            int apply_silu,
                            const float* qparams, int qmin, int qmax, int off,
                            int8_t* out, int64_t ldo, float* yout, int64_t ldy,
                            void* ws, const float* part_in, int nchunk_in, int64_t part_ld, const qd_raw_quant* raw,
                            void* stream);
/* The same for the second norm of a `use_scale_shift_norm` residual block (qdiff/quant_block.py:99-103, openaimodel.py:
 * 266-272: `h = out_norm(h) * (1 + scale) + shift`, then SiLU -> Dropout -> conv): mod holds one fp32 row per sample,
 * [B][mod_ld >= 2 C] = scale | shift, the two halves of the block's embedding projection (th.chunk(emb_out, 2, dim=1)).
 * The modulation is folded into the per-(sample, channel) affine of the normalisation; no raw second output.            */
int qd_groupnorm_mod_silu_quant(const void* x, int x_dtype, int64_t B, int64_t S, int C, int64_t ldx,
                                int groups, float eps, const float* gamma, const float* beta,
                                int apply_silu,
                                const float* qparams, int qmin, int qmax, int off,
                                int8_t* out, int64_t ldo, float* yout, int64_t ldy,
                                void* ws, const float* part_in, int nchunk_in, int64_t part_ld,
                                const float* mod, int64_t mod_ld, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9a LayerNorm -> quantise (up to 3 consumers).  Replaces nn.LayerNorm (attention.py:229-231)
 *     followed by the act_quantizers of to_q/to_k/to_v (or the GEGLU proj) QuantModules.
 *     x: [M][C] rows (ldx elements).  nout in 1..3; out[i] int8 [M][ldo].
 * ------------------------------------------------------------------------------------------ */
int qd_layernorm_quant(const void* x, int x_dtype, int64_t M, int C, int64_t ldx, float eps,
                       const float* gamma, const float* beta, int nout,
                       const float* const* qparams, const int* qmin, const int* qmax, const int* off,
                       int8_t* const* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9b GEGLU -> quantise.  Replaces `x, gate = proj(x).chunk(2,-1); x*F.gelu(gate)`
 *     (ldm/modules/attention.py:42-44) + the act_quantizer of the FF output Linear.
 *     h: [M][2F] (ldh), out int8 [M][ldo].
 * ------------------------------------------------------------------------------------------ */
int qd_geglu_quant(const void* h, int h_dtype, int64_t M, int F, int64_t ldh,
                   const float* qparams, int qmin, int qmax, int off,
                   int8_t* out, int64_t ldo, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7/K8 quantised attention.   Replaces the q/k/v/probability quantisers + the two einsum/bmm
 *     contractions + fp32 softmax of cross_attn_forward (qdiff/quant_block.py:190-221),
 *     QuantQKMatMul/QuantSMVMatMul (:123-134,152-157 with openaimodel.py:384-406) and
 *     QuantAttnBlock.forward (:354-386).
 *
 *     Step 1  qd_quantize_heads: x logical [B][T][H][d] (element strides sb, st, sh, sd),
 *             y = x*prescale quantised like K1 and written head-major:
 *               transpose=0:  out[(b*H+h)][Tpad][dpad]  (+ rsum[(b*H+h)][Tpad] = row sums)
 *               transpose=1:  out[(b*H+h)][dpad][Tpad] with the key index permuted inside each
 *                             32-key tile   (+ rsum[(b*H+h)][dpad] = column sums)
 *             Padding rows/cols are zero bytes; buffers must be Tpad = ceil32(T), dpad = ceil32(d).
 *     Step 2  qd_attn_i8: for every (b,h): S = (q-zq)(k-zk)^T * cs ; P = softmax_j(S);
 *             u = clamp(rint(P/dw)+zpw, wmin, wmax);  O = sum_j (u-zpw)(v-zv) * dw*dv.
 *             O is written as out[b][t][h*d + c] (merged heads, ldo = row stride, fp32).
 *     prm: device float[16] = {cs (=dq*dk*scale), zq', zk', dw, zpw, dv_dw (=dw*dv), zv', ...}
 *          (layout in DESIGN.md §4.4); built once on device by the host, never read back.
 *     out8 != NULL: O is not written as fp32 but quantised with oq_* = the act quantiser of the Linear that
 *           consumes it (to_out[0], quant_block.py:221 -> quant_layer.py:256) and stored as its int8 input rows
 *           out8[b*T+t][ldo8] (K1 semantics; H*d must already be that Linear's padded input width).
 *     q_asym: 0 when zq' == 0 (symmetric q quantiser), else 1: the kernel then restores the per-key term
 *           -zq' * sum_d k'[j][d].  qsum is ignored (may be NULL): the per-query terms -zk'*qsum_i + d*zq'*zk' are
 *           constant along a softmax row and cancel exactly.
 *     kterm (ABI 18; the slot that used to be `ksum`): NULL, or the table qd_attn_keyterm wrote for THIS k operand and
 *           THIS prm.  Shapes with qd_attn_uses_keyterm(d, S, q_asym) == 1 (d < 64, d % 32 != 0, S >= 512: SD's 4096-token level)
 *           then seed the score accumulators from the table instead of issuing constant-operand MFMAs (2 of the 4
 *           score MFMAs of a 32x32 tile carried no data); results are bit-identical with and without it.  Other shapes
 *           ignore it.
 *     qd_attn_keyterm: kterm[bh][j] = 0x4B400000 - zq' * sum_{c < dpad} k[bh][j][c]  (int32 [BH][Spad], dpad 32, 64 or 96; pad
 *           bytes of k are zero).  One pass over the K operand; recompute whenever k or prm[1] changes — a cross-attention
 *           whose context is constant over a sampling run computes it once (quant_block.py:193-195 recomputes k per step).
 *     qd_attn_config: process-wide launcher knobs, -1 = leave unchanged.  pipe_mode 0 = register-fed kernel everywhere,
 *           2 = LDS-staged kernel where it pays (default), 3 = LDS-staged wherever eligible; xcd 0/1 = XCD-aware block
 *           order; ktab 0 = ignore kterm (constant-operand MFMAs, A/B runs); lean 0 = attn_kernel for every head dim, 1 = lean /
 *           LDS-staged kernels for d < 64 (default), 3 = also d = 80 on the lean kernel (measured slower).  Initial values:
 *           QD_ATTN_PIPE / QD_ATTN_XCD / QD_ATTN_KTAB / QD_ATTN_LEAN, read once.
 *     ws / ws_bytes (ABI v20): scratch of the LDS-staged path, which runs as THREE launches since round 6 — per-query softmax
 *           statistics first (sweep 1 at up to 6 waves per SIMD), then the probability / P.V sweep for blocks that need only the
 *           lo operand bytes of the 16-bit codes (4 waves per SIMD) and for the others (3): 16 bytes per padded query + 4 per
 *           128-query block.  qd_attn_ws_bytes(BH, T, S, d) is the size for a shape (0: the shape runs on a one-kernel path
 *           and ws may be NULL); contents are undefined afterwards, one buffer may serve every call of a stream.
 * ------------------------------------------------------------------------------------------ */
int qd_quantize_heads(const void* x, int x_dtype, int B, int T, int H, int d,
                      int64_t sb, int64_t st, int64_t sh, int64_t sd, float prescale,
                      const float* qparams, int qmin, int qmax, int off, int transpose,
                      int8_t* out, int32_t* rsum, int Tpad, int dpad, void* stream);

int qd_attn_uses_keyterm(int d, int S, int q_asym);
int qd_attn_keyterm(const int8_t* k, int BH, int Spad, int dpad, const float* prm, int32_t* kterm, void* stream);
void qd_attn_config(int pipe_mode, int xcd, int ktab, int lean);
int64_t qd_attn_ws_bytes(int BH, int T, int S, int d);
int qd_attn_i8(const int8_t* q, const int8_t* k, const int8_t* vt,
               const int32_t* qsum, const int32_t* kterm, const int32_t* vsum,
               int BH, int H, int T, int S, int d, int Tpad, int Spad, int dpad,
               const float* prm, int wbits, int wmin, int wmax, int q_asym,
               float* out, int64_t ldo,
               int8_t* out8, int64_t ldo8, const float* oq_params, int oq_min, int oq_max, int oq_off,
               void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K7s/K8s  the two attention contractions as standalone batched integer GEMMs, for callers that use
 *     QuantQKMatMul / QuantSMVMatMul on their own (qdiff/quant_block.py:114-160 with the softmax applied by the
 *     caller in between, ldm openaimodel.py:384-406).  Operands are the same qd_quantize_heads layouts.
 *     qd_bmm_qk_i8: out[bh][t][s] = cs * sum_d (q'-zq')(k'-zk')   prm = {cs, zq', zk'} (device floats);
 *                   out rows have stride ldo, heads stride bstride (elements).
 *     qd_bmm_pv_i8: w[bh][t][s] fp32 probabilities (row stride ldw, head stride wbstride) are quantised with
 *                   u = clamp(rint(w/dw)+zpw, wmin, wmax) and contracted with v:
 *                   out[bh][c][t] = dw*dv * sum_s (u-zpw)(v'-zv')   ("bct": row stride ldo, head stride obstride);
 *                   vt / vsum = transpose=1 output of qd_quantize_heads; prm[3..6] = {dw, zpw, dw*dv, zv'}.
 * ------------------------------------------------------------------------------------------ */
int qd_bmm_qk_i8(const int8_t* q, const int8_t* k, int BH, int T, int S, int d, int Tpad, int Spad, int dpad,
                 const float* prm, float* out, int64_t ldo, int64_t bstride, void* stream);
int qd_bmm_pv_i8(const float* w, int64_t ldw, int64_t wbstride, const int8_t* vt, const int32_t* vsum,
                 int BH, int T, int S, int d, int Spad, int dpad, const float* prm, int wbits, int wmin, int wmax,
                 float* out, int64_t ldo, int64_t obstride, void* stream);

/* ------------------------------------------------------------------------------------------
 * Calibration: fused fake-quantisation of an activation tensor, forward and backward, for the
 *     step-size phase of block / layer reconstruction (qdiff/block_recon.py:72-110 trains `delta` by autograd through
 *     UniformAffineQuantizer.forward, quant_layer.py:82-88, with the straight-through rounding of :16-20).
 *     y      = (clamp(rint(x / delta) + zp, qmin, qmax) - zp) * delta
 *     gx     = d(loss)/dx       (the gradient passes where the un-clamped code lies in [qmin, qmax])
 *     gdelta_part[b] = block b's share of d(loss)/d(delta); the caller sums qd_fakequant_blocks(n) partials.
 *     delta / zero_point are device scalars (the reference keeps them as tensors / Parameters).  fp32 only.
 * ------------------------------------------------------------------------------------------ */
int64_t qd_fakequant_blocks(int64_t n);
int qd_fakequant_fwd(const float* x, int64_t n, const float* delta, const float* zero_point, int qmin, int qmax, float* y,
                     void* stream);
int qd_fakequant_bwd(const float* x, const float* gy, int64_t n, const float* delta, const float* zero_point, int qmin, int qmax,
                     float* gx, float* gdelta_part, void* stream);

/* ------------------------------------------------------------------------------------------
 * First-stage decoder: the convolutions of the KL-f8 / VQ-f4 `Decoder`
 *     (ldm/modules/diffusionmodules/model.py:465-572 `Decoder.forward`: conv_in, ResnetBlock :80-141, AttnBlock :144-196
 *     q/k/v/proj_out, Upsample :48-63 nearest-2x + conv, conv_out; called from ldm/models/autoencoder.py:330-333 `decode`
 *     and ldm/models/diffusion/ddpm.py `decode_first_stage`) as bf16 x bf16 -> fp32 implicit GEMMs on
 *     v_mfma_f32_32x32x16_bf16 — the loader, ring and epilogue of qd_conv2d_i8 in its floating-point mode.
 *
 *     qd_conv2d_bf16: same descriptor as qd_conv2d_i8 with
 *         x          bf16 channels-last rows [B*H*W][ldx]; ldx, seg[0].c0, seg[0].clen count bf16 ELEMENTS (multiples of 8)
 *         w          tile-ordered bf16 of qd_pack_weights_bf16 (wbits = 16, w_tiled = 1)
 *         nseg = 1, epilogue = QD_EPI_LINEAR; seg[0].scale / zc / zw / zfill / fill16 unused (out-of-image taps read 0)
 *         out        out[m][n] = sum_k x w + bias[n] (+ residual[m][n]); out_dtype = QD_F32 or QD_BF16, the residual has
 *                    the type of the output
 *         gn_part, upsample2x, kh/kw/stride/pad as for qd_conv2d_i8; rowbias, split-K, oq_* / hd_* unused.
 *     qd_pack_weights_bf16: fp32 [Cout][Cin][taps] (OIHW) -> bf16 (round to nearest even) in the order the kernel reads:
 *         per (tap, 32-channel K-step, 32-output-channel tile) 2 KB as [k-half (16 ch)][lane-half (8 ch)][n % 32][8 bf16];
 *         channels Cin .. clen_pad-1 (clen_pad % 8 == 0 = the descriptor's clen) are zero.  wt: qd_pack_weights_bf16_bytes.
 *     qd_groupnorm_silu_bf16: GroupNorm (+ SiLU) of fp32 rows into bf16 rows (model.py:38-45 `Normalize` / `nonlinearity`
 *         in front of every convolution); ws / part_in / nchunk_in / part_ld as for qd_groupnorm_silu_quant.
 *
 *     fp16 operands (ABI 18): the reference decodes under fp16 autocast (scripts/txt2img.py:231-236), i.e. with IEEE halves
 *     (11 significant bits) where bf16 has 8.  Same kernels, same bytes per K-step, v_mfma_f32_32x32x16_f16 at the same rate:
 *         qd_pack_weights_h16(..., wbits, ...)     wbits 16 = bf16 (what qd_pack_weights_bf16 does), 17 = fp16
 *         qd_conv2d_bf16 with desc.wbits = 17      x / w are halves; out_dtype QD_F32 or QD_F16 (residual of that type)
 *         qd_groupnorm_silu_h16(..., out_dtype, ...)  QD_BF16 or QD_F16 rows
 * ------------------------------------------------------------------------------------------ */
int qd_conv2d_bf16(const qd_conv_desc* d, void* stream);
int64_t qd_pack_weights_bf16_bytes(int Cout, int taps, int clen_pad);
int qd_pack_weights_bf16(const float* w, int Cout, int Cin, int taps, int clen_pad, uint8_t* wt, void* stream);
int qd_groupnorm_silu_bf16(const float* x, int64_t B, int64_t S, int C, int64_t ldx, int groups, float eps,
                           const float* gamma, const float* beta, int apply_silu, void* out, int64_t ldo, void* ws,
                           const float* part_in, int nchunk_in, int64_t part_ld, void* stream);
int qd_pack_weights_h16(const float* w, int Cout, int Cin, int taps, int clen_pad, int wbits, uint8_t* wt, void* stream);
int qd_groupnorm_silu_h16(const float* x, int64_t B, int64_t S, int C, int64_t ldx, int groups, float eps,
                          const float* gamma, const float* beta, int apply_silu, int out_dtype, void* out, int64_t ldo, void* ws,
                          const float* part_in, int nchunk_in, int64_t part_ld, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QDIFF_HIP_H */
