/*
 * gcpnet_hip.h -- C ABI of libgcpnet_hip.so: MI355X (gfx950) kernels for GCPNet's geometry-complete
 * message-passing hot path.
 *
 * The reference (BioinfoMachineLearning/GCPNet) has no native boundary for this path: it is a Python nn.Module
 * API.  The drop-in boundary is therefore the Python package `gcpnet_amd`, which mirrors
 * the reference classes; every arithmetic step those classes perform is one of the entry points below.  Each
 * entry point cites the reference code it replaces (paths relative to /root/reference/src/models).
 *
 * Conventions
 *   - plain pointers and sizes only; all float tensors are contiguous fp32 row-major device memory,
 *     index arrays are int32 device memory;
 *   - scalar features are [rows, dim]; vector features are [rows, channels, 3]; frames are [rows, 3, 3]
 *     with frame row a in {x_diff, x_cross, x_vertical} (components/__init__.py:268);
 *   - no ownership transfer; kernels are launched on the caller's `stream` (a hipStream_t) and return
 *     immediately; return value 0 = launched, > 0 = hipError_t, < 0 = invalid argument (GCPNET_E_*);
 *   - nothing here allocates device memory.
 */
#ifndef GCPNET_HIP_H
#define GCPNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCPNET_ABI_VERSION 4

#define GCPNET_E_BADARG (-1)
#define GCPNET_E_UNSUPPORTED (-2)

/* activation codes: models/__init__.py:42-57 (get_nonlinearity) */
enum { GCP_ACT_NONE = 0, GCP_ACT_RELU = 1, GCP_ACT_LEAKYRELU = 2, GCP_ACT_SELU = 3, GCP_ACT_SILU = 4, GCP_ACT_SIGMOID = 5 };

/* vector-gating mode of a GCP2 block: components/gcpnet.py:369-389 */
enum { GCP_VMODE_NONE = 0, GCP_VMODE_SCALAR_GATE = 1 /* vector_gate */, GCP_VMODE_SELF_GATE = 2 /* sigma(norm) */ };

#define GCP_MAX_SEG 3

/* A row-wise concatenation of up to 3 sources, each optionally gathered by an index array:
 * column block k of row r is ptr[k][(idx[k] ? idx[k][r] : r) * dim[k] ...].  This is how
 * GCPMessagePassing.message builds [h_row | e | h_col] and [chi_row | xi | chi_col]
 * (components/gcpnet.py:907-917) without materialising the concatenation. */
typedef struct {
    int n;
    const float* ptr[GCP_MAX_SEG];
    const int32_t* idx[GCP_MAX_SEG];
    int dim[GCP_MAX_SEG]; /* scalars: floats per row; vectors: channels per row */
} gcp_concat_t;

/* Weights of one GCP2 block (components/gcpnet.py:298-324), reference layouts, plus packed MFMA operand images
 * produced by gcpnet_pack_gcp2_weights(). */
typedef struct {
    int si, vi, so, vo, hidden; /* scalar in/out, vector in/out channels, hidden vector channels H */
    int use_frames;             /* !ablate_frame_updates: 9 frame scalars appended to the merged input */
    const float* w_down;        /* vector_down.weight        [H, vi]            (NULL when vi == 0) */
    const float* w_frames;      /* vector_down_frames.weight [3, vi]            (NULL unless use_frames) */
    const float* w_up;          /* vector_up.weight          [vo, H]            (NULL when vo == 0) */
    const float* w_scalar;      /* scalar_out.weight         [so, si + H + 9]                       */
    const float* b_scalar;      /* scalar_out.bias           [so]                                   */
    const float* w_gate;        /* vector_out_scale.weight   [vo, so]           (NULL unless scalar gate) */
    const float* b_gate;        /* vector_out_scale.bias     [vo]                                   */
    const float* pack;          /* packed images, layout private to the library; size from gcpnet_gcp2_pack_floats() */
} gcp2_weights_t;

typedef struct {
    int act_s, act_v;     /* nonlinearities (scalar, vector) */
    float slope;          /* leaky-relu slope */
    int vmode;            /* GCP_VMODE_* */
    int vector_residual;  /* components/gcpnet.py:341-342,365-366 */
    int e3;               /* enable_e3_equivariance: |.| on the x_cross projections (components/__init__.py:305-309) */
    int fused_residual;   /* backward only: the block was applied as x + GCP(x) (ResGCP, components/gcpnet.py:921-924),
                             so d(x) = d(out) + GCP^T d(out); the add is done in the kernel epilogue */
} gcp2_opts_t;

/* ---- weight packing ------------------------------------------------------------------------------------ */
/* Number of floats gcpnet_pack_gcp2_weights() writes for these dims. */
int64_t gcpnet_gcp2_pack_floats(int si, int vi, int so, int vo, int hidden, int use_frames);
/* Packs scalar_out / vector_out_scale into MFMA operand order (forward, backward-data, gate, gate-backward). */
int gcpnet_pack_gcp2_weights(const gcp2_weights_t* w, float* pack_out, void* stream);
/* the same for n <= GCP_MAX_CHAIN blocks of ONE shape (the blocks of a ResGCP chain) in one launch */
int gcpnet_pack_gcp2_weights_multi(int n, const gcp2_weights_t* w, float* const* pack_out, void* stream);

/* ---- GCP2 forward: replaces GCP2.forward (components/gcpnet.py:394-468) on `rows` rows --------------------
 * s_in/v_in: concatenated inputs; frames: per-row frames [rows,3,3] (per-edge frames for edge rows; the
 * out-edge frame mean for node rows, which is what scalarize(node_inputs=True) reduces to:
 * components/__init__.py:286,314-323).  res_s/res_v (optional) are added to the outputs (ResGCP,
 * components/gcpnet.py:921-924).  s_pre [rows,so] and gate [rows,vo] (sigmoid of the vector gate) are saved
 * for the backward when non-NULL.
 * s_add (optional): scalar inputs whose share of scalar_out was computed beforehand at their source rows
 * ("project, then gather": for a message GCP over [h_row | e | h_col] the two node terms are node-level GEMMs, 16x
 * fewer rows than edges): s_add->ptr[k] is a [n_src, so] table, s_add->idx[k] the gather (NULL = row r), dim[k] = so;
 * s_pre = scalar_out([s_in | norms | frame scalars]) + sum_k table_k[idx_k[r]].
 * gcpnet_gcp2_forward_lds_bytes: LDS a wave-tile of these dims needs (the kernel returns GCPNET_E_UNSUPPORTED above 160 KB:
 * the 32 x (si + H + 9) merged tile is the large term; a caller can shrink si by pre-projecting columns into s_add). */
int64_t gcpnet_gcp2_forward_lds_bytes(int si, int vi, int so, int vo, int hidden, int use_frames);
/* v_add (optional): the same for the vector inputs: vector_down and vector_down_frames are linear too, so the shares of
 * gathered sources (chi[row], chi[col]) are computed per source row; v_add->ptr[k] is a [n_src, 3, HF'] table
 * ([vector_down ; vector_down_frames] W_k chi, xyz-major, HF' = H + 3 rounded up to 4, zero padded, dim[k] = HF'),
 * added to [vh | vf] of the gathering row.  Needs H + 3 <= 32 and vo <= 32. */
int gcpnet_gcp2_forward(int rows, const gcp_concat_t* s_in, const gcp_concat_t* v_in, const float* frames,
                        const gcp2_weights_t* w, const gcp2_opts_t* opts, const gcp_concat_t* s_add,
                        const gcp_concat_t* v_add, const float* res_s, const float* res_v, float* s_out, float* v_out,
                        float* s_pre, float* gate, void* stream);

/* ---- chain of residual GCP2 blocks: x_k = x_{k-1} + GCP_k(x_{k-1}), k = 1..n (ResGCP, components/gcpnet.py:921-924).
 * One launch; the (s, V) state of a 32-row tile stays on chip between the blocks.  All blocks share the dims
 * (si == so <= 128, vi == vo), frames and gating mode; each item carries its own weights, activations and outputs
 * (s_out/v_out = x_k, s_pre/gate saved for the backward when non-NULL). */
/* Tile-blocked layout ("tb") of a [rows, W] fp32 matrix that only the chain kernels and gcpnet_tn_gemm exchange (s_pre, ds_pre, the
 * intermediate scalar states of a chain): rows in tiles of 32, columns padded to Wp = 32 ceil(W / 32); element (r, c) sits at float
 *     (r / 32) * 32 Wp + (((c / 32) * 4 + (c % 32) / 8) * 64 + ((c / 4) % 2) * 32 + r % 32) * 4 + c % 4,
 * i.e. register quad q of accumulator tile t of lane (r % 32, half) of the 32-row wave-tile is one 16-byte piece and the 64 pieces
 * of a (t, q) are 1 KB contiguous: a wave's store / load instruction of such a tensor moves eight full 128-byte lines instead of
 * touching 32 rows (row-major, each lane pair owns 32 bytes of a row), and needs no LDS transposition.  A tb buffer holds
 * gcpnet_tb_floats(rows, W) floats (whole tiles); rows past the end hold unspecified finite values. */
int64_t gcpnet_tb_floats(int rows, int width);
/* Sign mask of s_pre (ABI 4).  The backward of a block with piecewise-linear activations (None / relu / leakyrelu: every shipped
 * configuration) needs of s_pre only WHERE it is positive -- act'(s_pre) of components/gcpnet.py:441,465 and of the gate's input,
 * :345-347 -- i.e. one bit per element instead of four bytes.  s_sign [ceil(rows / 32)][Wp / 64][64 lanes] uint32, Wp = 32 ceil(W / 32)
 * (even number of 32-wide tiles): bit (16 t + r) % 32 of word (16 t + r) / 32 of lane (r_ % 32, half) of a tile is s_pre > 0 for
 * register r of accumulator tile t of that lane (the element the tile-blocked layout above puts at piece (t, r / 4), float r % 4).
 * The register-resident chain forward writes it when s_sign != NULL (beside s_pre, which the weight-gradient GEMM still reads);
 * gcpnet_gcp2_chain_backward reads it INSTEAD of s_pre when every item carries one: 16 KB less HBM traffic per 32-row tile and
 * block at width 128, and no 64-register request in flight through the block's vector stage.
 * gcpnet_tb_sign_words(rows, width): uint32 words of such a buffer (0 when the width has an odd number of tiles). */
int64_t gcpnet_tb_sign_words(int rows, int width);
typedef struct {
    gcp2_weights_t w;
    gcp2_opts_t o;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
    int s_out_tb;  /* s_out is written tile-blocked (register-resident kernel only: E_UNSUPPORTED elsewhere) */
    int s_pre_tb;  /* s_pre is written tile-blocked */
    uint32_t* s_sign;  /* optional sign mask of s_pre (see above; register-resident kernel, piecewise-linear activations only) */
} gcp2_chain_item_t;
#define GCP_MAX_CHAIN 8
int gcpnet_gcp2_chain_forward(int rows, const float* s0, const float* v0, const float* frames, int n,
                              const gcp2_chain_item_t* items, void* stream);
/* 1 if a chain of residual blocks (si == so, vi == vo) of this shape runs in gcpnet_gcp2_chain_forward's register-resident
 * kernel (state in MFMA accumulator-layout registers for the whole chain), 0 if it would take the LDS-resident fallback. */
int gcpnet_gcp2_chain_forward_registers_ok(int si, int vi, int so, int vo, int hidden, int use_frames);


/* ---- the first message GCP after project-then-gather, alone (n == 0) or fused in front of the chain it feeds:
 * scalar input e_in [rows, w.si] and vector input xi_in [rows, w.vi, 3] are plain (un-gathered) tensors, the gathered
 * sources enter through s_add / v_add as in gcpnet_gcp2_forward; the block is not residual and its outputs (s_out, v_out;
 * s_pre / gate when non-NULL) are written in any case.  Returns GCPNET_E_UNSUPPORTED outside the register-resident kernel's
 * shapes (so in {64, 128}, w.si <= 64, w.vi <= 20, vo <= 32, frames in use); the caller then uses gcpnet_gcp2_forward. */
typedef struct {
    const float* e_in;
    const float* xi_in;
    gcp_concat_t s_add, v_add;
    gcp2_weights_t w;
    gcp2_opts_t o;
    float* s_out;
    float* v_out;
    float* s_pre;
    float* gate;
} gcp2_head_t;
int gcpnet_gcp2_headchain_forward(int rows, const gcp2_head_t* head, const float* frames, int n,
                                  const gcp2_chain_item_t* items, void* stream);

/* ---- multi-wave workgroup kernels (gcp_wg_*.hip): one WORKGROUP of 4 or 8 wavefronts per 32-row tile, output columns
 * split across the waves, the tile's merged input [s | norms | frame scalars] once in LDS.  They cover any output width (so a
 * multiple of 4) with a plain (un-gathered) input: the first message GCP after project-then-gather, the residual message
 * chain (ResGCP, components/gcpnet.py:921-924) and the node-row GCPs (embeddings, feed-forward, position update;
 * components/gcpnet.py:394-468 on N rows), at every shipped hidden size including (256, 32).
 * Weights go through their own packed image (gcpnet_wg_pack into gcp2_weights_t.pack; `gated` = the block has a
 * vector_out_scale Linear). */
int64_t gcpnet_wg_pack_floats(int si, int vi, int so, int vo, int hidden, int use_frames, int gated);
int gcpnet_wg_pack(const gcp2_weights_t* w, int gated, float* pack_out, void* stream);
/* The same from a VIEW of the stored scalar_out weight: logical W'[r][c] (r < so, c < K = si + H + 9) is W[r * ld + col(c)], or
 * W[col(c) * ld + r] when `trans`, col(c) running through `nseg` <= 3 column ranges (start[k], len[k]) whose lengths add up to
 * K.  Packs column slices (project-then-gather: the kernel keeps [e | norms | frame scalars] of scalar_out's columns; a node-level
 * projection uses one source's columns) and transposed weights (input gradient of a Linear) without materialising them.
 * With vi == 0, vo == 0 a block is a plain Linear + activation: gcpnet_wg_forward then replaces nn.Linear on [rows, si]. */
int gcpnet_wg_pack_view(const gcp2_weights_t* w, int gated, const float* W, int ld, int trans, int nseg, const int* start,
                        const int* len, float* pack_out, void* stream);
/* Several images in ONE launch (a training step re-packs every block's weights after the optimizer has updated them: 34 images per
 * NMS model step): job j = the arguments of gcpnet_wg_pack_view (W = w.w_scalar, ld = K, one segment (0, K) for a plain pack). */
typedef struct {
    gcp2_weights_t w;
    int gated;
    const float* W;
    int ld, trans, nseg;
    int start[3], len[3];
    float* out;
} gcp_wg_pack_job_t;
int gcpnet_wg_pack_multi(int n, const gcp_wg_pack_job_t* jobs, void* stream);

typedef struct {
    gcp2_weights_t w;   /* dims of THIS block, reference-layout weights, w.pack = image of gcpnet_wg_pack */
    gcp2_opts_t o;
    float* s_out;       /* [rows, so]     block output (the chain state after this block); may be NULL when nobody reads it */
    float* v_out;       /* [rows, vo, 3]  */
    float* s_pre;       /* [rows, so]     saved for the backward (NULL in inference) */
    float* gate;        /* [rows, vo]     sigmoid of the vector gate, saved for the backward */
    int residual;       /* out = in + GCP(in) (needs si == so, vi == vo) */
    int s_out_tb, s_pre_tb;  /* the tensor is written tile-blocked (so a multiple of 32; see gcp2_chain_item_t) */
} gcp_wg_block_t;

/* Forward of n blocks on `rows` rows in one launch: block 0 reads s_in [rows, w.si] / v_in [rows, w.vi, 3] (plus the
 * pre-projected gathered tables s_add / v_add as in gcpnet_gcp2_forward, may be NULL); blocks 1.. run on the previous block's
 * output, which stays on chip (their si == so, vi == vo).  All blocks share so, vo, frames, gating mode and options.
 * Returns GCPNET_E_UNSUPPORTED for shapes outside the kernel (vi == 0, so not a multiple of 4, tiles above 160 KB of LDS,
 * chains wider than 512 scalars); the caller then uses gcpnet_gcp2_forward. */
#define GCP_WG_MAX_BLOCKS 9
int gcpnet_wg_forward(int rows, const float* s_in, const float* v_in, const float* frames, const gcp_concat_t* s_add,
                      const gcp_concat_t* v_add, int n, const gcp_wg_block_t* blocks, void* stream);

/* Backward of ONE block in the workgroup form (the adjoint of one gcp_wg_block_t; a chain is walked block by block, last to
 * first, the state gradient travelling through d_s_in / d_v_in).  Two modes, chosen by gcpnet_wg_backward_plan():
 *   fused   (edge-row blocks whose K + 1 <= 160 / 288 columns fit the accumulators): the weight gradients of scalar_out and
 *           vector_out_scale are accumulated on chip over all tiles of a persistent workgroup and leave as per-workgroup partial
 *           sums dw_part [grid, so, K + 1] (bias gradient in the last column) and dwg_part [grid, vo, so + 1], to be summed by
 *           gcpnet_wg_reduce; needs s_in;
 *   plain   ds_pre / ext / dgate are written per row for gcpnet_tn_gemm, exactly as gcpnet_gcp2_backward does.
 * In both modes wsm_part [grid, vo*H + (H+3)*vi] receives per-workgroup partial sums of d vector_up [vo, H], d vector_down [H, vi]
 * and d vector_down_frames [3, vi] (in this order).  ds_pre / dvhf are optional outputs in fused mode (the first message GCP:
 * gradients of the gathered addend tables). */
typedef struct {
    int nw, kt, fused, split;
    int grid;      /* workgroups = leading dimension of the *_part buffers */
    int kw;        /* K + 1 */
    int n_small;   /* width of wsm_part */
    int ext_w;     /* row width of `ext`   ((H + 9) rounded up to 4) */
    int dgate_w;   /* row width of `dgate` (vo rounded up to 4) */
} gcp_wg_bwd_plan_t;
int gcpnet_wg_backward_plan(int rows, const gcp2_weights_t* w, const gcp2_opts_t* o, int want_fused, gcp_wg_bwd_plan_t* plan);

typedef struct {
    gcp2_weights_t w;     /* w.pack = image of gcpnet_wg_pack */
    gcp2_opts_t o;
    int residual;
    const float* s_in;    /* [rows, si]  (fused mode only) */
    const float* v_in;    /* [rows, vi, 3] */
    const float* frames;
    const gcp_concat_t* v_add;  /* as in the forward (NULL = none) */
    const float* s_pre;
    const float* gate;
    const float* d_s_out;
    const float* d_v_out;
    float* d_s_in;
    float* d_v_in;
    float* ds_pre;        /* optional [rows, so] */
    float* dvhf;          /* optional [rows, 3, HF'] */
    float* ext;           /* plain mode [rows, ext_w] */
    float* dgate;         /* plain mode [rows, dgate_w] */
    float* dw_part;       /* fused mode */
    float* dwg_part;      /* fused mode, scalar-gated blocks */
    float* wsm_part;
    int tb;               /* tile-blocked tensors (the layout of gcp2_chain_item_t; so resp. si a multiple of 32): bit 0 s_pre, bit 1 d_s_out,
                             bit 2 d_s_in, bit 3 ds_pre -- what only these kernels and gcpnet_tn_gemm exchange inside one backward pass */
} gcp_wg_bwd_args_t;
int gcpnet_wg_backward(int rows, const gcp_wg_bwd_args_t* args, void* stream);
/* out_w[r, c] = sum_g parts[g, r, c] for c < CW, out_b[r] = sum_g parts[g, r, CW] when C == CW + 1; fixed summation order. */
int gcpnet_wg_reduce(const float* parts, int n_parts, int R, int C, int CW, float* out_w, float* out_b, void* stream);
/* The same for up to GCP_WG_REDUCE_MAX_JOBS partial buffers in ONE launch (the three of a fused backward: dw_part, dwg_part,
 * wsm_part). */
typedef struct {
    const float* parts;
    int n_parts, R, C, CW;
    float* out_w;
    float* out_b;  /* may be NULL */
} gcp_wg_reduce_job_t;
#define GCP_WG_REDUCE_MAX_JOBS 4
int gcpnet_wg_reduce_multi(int n_jobs, const gcp_wg_reduce_job_t* jobs, void* stream);

/* ---- GCP2 backward (data path) ----------------------------------------------------------------------------
 * Given d(s_out), d(v_out) and the saved s_pre/gate, writes d(s_in) [rows, si] and d(v_in) [rows, vi, 3] in the
 * concatenated layout, plus what the weight gradients need:
 *   - per-row operands of the big weight-gradient GEMMs (gcpnet_tn_gemm), row-major:
 *       ds_pre [rows, so], dgate [rows, vo'], ext [rows, (H+9)'] (= [|vh| norms | frame scalars]),
 *     where x' = x rounded up to a multiple of 4 floats (16-byte DMA pieces; the padding is written as zeros);
 *   - w_part [gcpnet_gcp2_bwd_tiles(rows), vo*H + (H+3)*vi] (optional, may be NULL): per 32-row tile, that tile's
 *     share of d vector_up.weight [vo, H], d vector_down.weight [H, vi] and d vector_down_frames.weight [3, vi], in
 *     this order; sum over tiles with gcpnet_reduce_partials. */
typedef struct {
    float* ds_pre;
    float* dgate;
    float* ext;
    float* w_part;
    float* dvhf;   /* optional [rows, 3, HF']: d[vh | vf] per row = the gradient of the v_add tables' gathered rows */
} gcp2_bwd_scratch_t;
int gcpnet_gcp2_bwd_tiles(int rows);

int gcpnet_gcp2_backward(int rows, const gcp_concat_t* s_in, const gcp_concat_t* v_in, const float* frames,
                         const gcp2_weights_t* w, const gcp2_opts_t* opts, const gcp_concat_t* v_add /* as in the forward */,
                         const float* s_pre, const float* gate,
                         const float* d_s_out, const float* d_v_out, float* d_s_in, float* d_v_in,
                         const gcp2_bwd_scratch_t* scratch, void* stream);

/* ---- backward of a chain of residual GCP2 blocks (the adjoint of gcpnet_gcp2_chain_forward), one launch:
 * d(s), d(V) of a 32-row tile stay on chip between the blocks; per block k (items[k], forward order) the kernel reads
 * the block's input vectors v_in = V_{k-1} [rows, vi, 3] and the saved s_pre / gate, and fills items[k].sc exactly as
 * gcpnet_gcp2_backward does.  d_s_out / d_v_out are the gradients of the chain's output, d_s_in / d_v_in those of its
 * input.  Returns GCPNET_E_UNSUPPORTED for shapes outside the kernel (si == so in {32, 64, 128}, vi == vo <= 20 and a
 * multiple of 4, 16-byte aligned tensors); the caller then chains gcpnet_gcp2_backward with fused_residual. */
typedef struct {
    gcp2_weights_t w;
    gcp2_opts_t o;
    const float* v_in;
    const float* s_pre;
    const float* gate;
    gcp2_bwd_scratch_t sc;
    int tb;  /* s_pre is read and sc.ds_pre written in the tile-blocked layout (see gcp2_chain_item_t) */
    const uint32_t* s_sign;  /* optional sign mask of s_pre (gcp2_chain_item_t.s_sign): read instead of s_pre when EVERY item has one.
                              * s_pre may then be NULL (a forward that stored only the mask): the call returns GCPNET_E_BADARG, and
                              * launches nothing, if it cannot run the sign-mask kernel for the chain (an item without a mask, an
                              * activation that is not piecewise linear, or the fp32-MFMA A/B form forced by gcpnet_debug_set_fp32_mfma) */
} gcp2_chain_bwd_item_t;
/* 1 if gcpnet_gcp2_chain_backward takes a chain of residual blocks of this shape (callers that save tile-blocked tensors in the
 * forward ask first: there is no other consumer of that layout) */
int gcpnet_gcp2_chain_backward_ok(int si, int vi, int so, int vo, int hidden, int use_frames);
int gcpnet_gcp2_chain_backward(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                               const float* d_s_out, const float* d_v_out, float* d_s_in, float* d_v_in, void* stream);
/* The same when the chain's output went into a segment sum / mean (the aggregation, components/gcpnet.py:939-947; the reference
 * gets the adjoint -- d(message)[r] = d(aggregate)[col[r]] / count -- from autograd through torch_scatter): the incoming gradient of
 * row r is out_scale[j] * d_*_tab[j] with j = out_idx[r] (d_s_tab [n_seg, so], d_v_tab [n_seg, vo, 3]; out_scale NULL = 1), read
 * from the segment-level tables where the kernel needs it instead of from a materialised [rows, .] copy. */
int gcpnet_gcp2_chain_backward_gathered(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                        const float* d_s_tab, const float* d_v_tab, const int32_t* out_idx, const float* out_scale,
                                        float* d_s_in, float* d_v_in, void* stream);
/* Both forms with the TAIL SPLIT (ABI 4).  A tile's chain (the loop of components/gcpnet.py:921-924, walked backwards) is one wave's
 * serial job and a CU holds eight of them: when the tile count leaves a last, partly filled round of waves, the launch cuts some
 * tiles' chains in two -- the first halves are dispatched ahead of everything, the second halves last -- so that the pieces fill
 * the idle slots (same arithmetic, same results; the hand-over of d(s), d(V) between the two workgroups of a tile goes through
 * d_s_in / d_v_in and one flag word per tile, agent-scope release / acquire, correct under any dispatch order).
 * gcpnet_gcp2_chain_backward_flags: how many flag words a launch of this shape would use (0: it would not split).
 * `flags`: n_flags >= that many uint32 of device memory no other launch touches meanwhile (zeroed by the call, on the stream);
 * out_idx NULL = the plain form (d_s_out / d_v_out per-row tensors). */
int gcpnet_gcp2_chain_backward_flags(int rows, int n, int si, int vi, int so, int vo, int hidden, int use_frames);
/* test hook: n_split >= 0 cuts the first min(n_split, tiles) tiles at block k_split whatever the size (rev != 0: reversed workgroup
 * order, so that second halves run first and take tiles over); n_split < 0 = back to the planned split */
void gcpnet_debug_force_chain_split(int n_split, int k_split, int rev);
int gcpnet_gcp2_chain_backward_split(int rows, const float* frames, int n, const gcp2_chain_bwd_item_t* items,
                                     const float* d_s_out, const float* d_v_out, const int32_t* out_idx, const float* out_scale,
                                     float* d_s_in, float* d_v_in, uint32_t* flags, int n_flags, void* stream);

/* ---- weight-gradient GEMM: out[m, n] (+)= sum_r A[r, m] * B[r, n] -------------------------------------------
 * A and B are row-wise concatenations (gcp_concat_t with per-segment leading dimension), optionally passed
 * through an activation, optionally extended by a column of ones (bias gradients).  Used for
 * d scalar_out.weight = ds_pre^T [s | norms | frame scalars] and d vector_out_scale.weight (the small vector weights
 * get their gradients from the backward kernels' per-tile partial sums). */
#define GCP_TN_MAX_SEG 4
typedef struct {
    int n;
    const float* ptr[GCP_TN_MAX_SEG];
    const int32_t* idx[GCP_TN_MAX_SEG];
    int dim[GCP_TN_MAX_SEG];
    int ld[GCP_TN_MAX_SEG];
    int act;       /* activation applied on load */
    float slope;
    int ones;      /* append a column of ones */
    int tb[GCP_TN_MAX_SEG];  /* 0: rows of ld floats; 1: the segment is tile-blocked (see gcp2_chain_item_t; its width is dim, no gather) */
} gcp_operand_t;

typedef struct {
    int rows;          /* reduction length */
    gcp_operand_t a;   /* [rows, M]  (M = sum dims + ones) */
    gcp_operand_t b;   /* [rows, N] */
    float* out;        /* out[m * out_sm + n * out_sn] = result[m, n] for m < out_m, n < out_n (the rest is padding) */
    int64_t out_sm, out_sn;
    int out_m, out_n;
    float* out2;       /* optional: out2[m] = result[m, out2_n] for m < out_m (the bias gradient: the ones column) */
    int out2_n;
    float* partial;    /* scratch [splits, M, N] */
    int splits;
    /* optional second destination for the rows m >= m_split of the result (m_split > 0; ABI 4): out_b[(m - m_split) * out_b_sm + n] for
     * m < out_m, n < out_n and out2_b[m - m_split] for the ones column -- two weight gradients that share their second operand in ONE
     * product (d scalar_out.weight and, below it, G of gcp2_wgrad_job_t.gate_lin) */
    int m_split;
    float* out_b;
    int64_t out_b_sm;
    float* out2_b;
} gcp_tn_problem_t;

#define GCP_TN_MAX_PROBLEMS 8
int gcpnet_tn_gemm(int n_problems, const gcp_tn_problem_t* problems, void* stream);
/* the row-split count the library recommends for a [rows, M]^T [rows, N] problem (always even, >= 2; by `rows` only at present).
 * `splits` of a problem is the caller's to choose (1 .. 4096; an odd count is served by the earlier kernels): a problem needs
 * splits * M * N floats of scratch in `partial` */
int gcpnet_tn_splits(int rows, int M, int N);

/* Column sums out[width] = sum_p parts[p, width] in a fixed order (deterministic); tmp holds
 * gcpnet_reduce_partials_groups(n_parts) * width floats.  Used for gcp2_bwd_scratch_t.w_part.  Up to
 * GCP_REDUCE_MAX_JOBS independent sums per call (two launches in total). */
typedef struct {
    const float* parts;
    int n_parts, width;
    float* tmp;
    float* out;
} gcp_reduce_job_t;
#define GCP_REDUCE_MAX_JOBS 8
int gcpnet_reduce_partials(int n_jobs, const gcp_reduce_job_t* jobs, void* stream);
int gcpnet_reduce_partials_groups(int n_parts);

/* ---- the weight gradients of n GCP2 blocks from their backward kernels' scratch, in ONE call -----------------------------------
 * (autograd through GCP2.forward, components/gcpnet.py:394-468, gives per block d scalar_out.weight / bias, d vector_down[_frames],
 * d vector_up and d vector_out_scale.weight / bias.)  For each job the call builds
 *     d scalar_out.weight | bias        = ds_pre^T [scalar input segments | norms, frame scalars (ext) | 1]
 *     d vector_out_scale.weight | bias  = dgate^T [act_v(s_pre) | 1]                                   (gated blocks)
 *       or, with gate_lin (act_v is the identity, every shipped configuration: the gate Linear reads s_pre itself, gcpnet.py:345-347):
 *       G = dgate^T [scalar input segments | ext | 1] -- the SAME second operand as the first product -- and then
 *       d vector_out_scale.weight = G[:, :K] scalar_out.weight^T + G[:, K] (x) scalar_out.bias   (s_pre = [s | ext] W^T + b is linear in
 *       what the block's input already holds), so that s_pre is never read -- and, with the chain kernels' sign masks, never stored
 *     d [vector_up | vector_down | vector_down_frames] = column sums of the per-tile partial sums w_part
 * as gcp_tn_problem_t / gcp_reduce_job_t records and launches them GCP_TN_MAX_PROBLEMS / GCP_REDUCE_MAX_JOBS at a time -- what the
 * host mirror otherwise assembles block by block in its own language (40 small records and 24 scratch allocations per 7-block chain).
 * `workspace`: gcpnet_gcp2_weight_grads_workspace(n, jobs) floats (the GEMMs' per-split partial sums, the reductions' group sums);
 * it must stay allocated until the launches have run. */
typedef struct {
    int rows;
    int so, vo, vi, hidden, use_frames;  /* the block's dimensions, as in gcp2_weights_t */
    int gated;                           /* vector_out_scale in use (d_w_gate / d_b_gate wanted) */
    int act_v;
    float slope;
    gcp_operand_t s_in;   /* the block's scalar input as the GEMM reads it: n, ptr, idx, dim, ld, tb of its segments (act / ones unused) */
    const float* ds_pre;  /* [rows, so], or tile-blocked */
    int ds_pre_tb;
    const float* s_pre;   /* [rows, so], or tile-blocked (gated blocks) */
    int s_pre_tb;
    const float* ext;     /* [rows, (hidden + 9 use_frames)'], NULL when vi == 0 */
    const float* dgate;   /* [rows, vo'] (gated blocks) */
    const float* w_part;  /* [n_parts, w_width] per-tile partial sums of the small vector weights, NULL when vi == 0 */
    int n_parts, w_width;
    float* d_w_scalar;    /* [so, si + hidden + 9 use_frames] */
    float* d_b_scalar;    /* [so] */
    float* d_w_small;     /* [w_width] */
    float* d_w_gate;      /* [vo, so] */
    float* d_b_gate;      /* [vo] */
    int gate_lin;            /* the gate gradients from the block's inputs (see above); needs act_v == GCP_ACT_NONE and the two weights below */
    const float* w_scalar;   /* scalar_out.weight [so, si + hidden + 9 use_frames] (gate_lin) */
    const float* b_scalar;   /* scalar_out.bias [so] (gate_lin) */
} gcp2_wgrad_job_t;
int64_t gcpnet_gcp2_weight_grads_workspace(int n, const gcp2_wgrad_job_t* jobs);
int gcpnet_gcp2_weight_grads(int n, const gcp2_wgrad_job_t* jobs, float* workspace, void* stream);

/* `to` waits for everything enqueued on `from` so far (hipEventRecord + hipStreamWaitEvent on an event of a library-owned ring): the
 * fork of the weight-gradient stream off the caller's stream, and its join back, without the host language's stream / event objects. */
int gcpnet_stream_wait_stream(void* to, void* from);

/* ---- segment reductions: torch_scatter.scatter(reduce=sum|mean) over sorted segments ------------------------
 * out[s, 0:D] = reduce_{p in [seg_ptr[s], seg_ptr[s+1])} x[(perm ? perm[p] : p) * ldx + 0:D]
 * (components/gcpnet.py:939-947 aggregate; components/__init__.py:195-198 centroids; :314-323 node scalarize). */
int gcpnet_segment_reduce(int n_seg, const int32_t* seg_ptr, const int32_t* perm, const float* x, int64_t ldx, int D,
                          int mean, float* out, int64_t ldo, int accumulate, void* stream);
/* out[r, 0:D] = x[idx[r] * ldx + 0:D] * (scale ? scale[idx[r]] : 1)  (backward of the above; gathers) */
int gcpnet_gather_rows(int rows, const int32_t* idx, const float* x, int64_t ldx, int D, const float* scale,
                       float* out, int64_t ldo, void* stream);

/* ---- frames: localize (components/__init__.py:221-269, unmasked) ---------------------------------------------*/
int gcpnet_localize(int n_edges, const int32_t* row, const int32_t* col, const float* x, int norm_x_diff,
                    float* frames, void* stream);

/* ---- GCPLayerNorm (components/__init__.py:138-167) fused with the residual add in front of it
 * (components/gcpnet.py:1220-1226,1242-1246): (s, v) = norm((s_a + s_b), (v_a + v_b)); s_b/v_b may be NULL. */
int gcpnet_layernorm_forward(int rows, int sdim, int vdim, const float* s_a, const float* s_b, const float* v_a,
                             const float* v_b, const float* gamma, const float* beta, float* s_out, float* v_out,
                             float* stats /* [rows,3]: mean, rstd, vnorm */, float* s_sum, float* v_sum, void* stream);
/* d_gamma_beta [2 * sdim] receives d gamma followed by d beta (summed over rows in a fixed order: per-block partial sums in
 * `scratch`, gcpnet_layernorm_bwd_scratch_floats(rows, sdim) floats, reduced by gcpnet_reduce_partials). */
int gcpnet_layernorm_backward(int rows, int sdim, int vdim, const float* s_sum, const float* v_sum,
                              const float* stats, const float* gamma, const float* d_s_out, const float* d_v_out,
                              float* d_s, float* d_v, float* d_gamma_beta, float* scratch, void* stream);
int64_t gcpnet_layernorm_bwd_scratch_floats(int rows, int sdim);

/* ---- small elementwise pieces ---------------------------------------------------------------------------------
 * y = a + alpha * b, clamped to [lo, hi] when clamp != 0 (position update, components/gcpnet.py:1156-1158,1258) */
int gcpnet_axpy_clamp(int64_t n, const float* a, const float* b, float alpha, int clamp, float lo, float hi, float* y,
                      void* stream);

/* ---- inter-node force term of the position update (components/gcpnet.py:1143-1153, `ablate_x_force_update: false`):
 * per edge z = act(A[row] + B[col]) with A = phi_force_i(h), B = phi_force_j(h) [N, s] (computed per node by the caller),
 * coef = W3 z (W3 = phi_force_ij.1.weight [3, s]), force[e] = coef[0] x_diff + coef[1] x_cross + coef[2] x_vertical (rows of
 * frames[e]); the scatter-mean over `col` that follows is gcpnet_segment_reduce.  Backward: d_pre [E, s] (to be summed per
 * row -> dA and per col -> dB) and per-block shares of d W3 in part [gcpnet_edge_force_bwd_blocks(E), 3 s]
 * (gcpnet_reduce_partials). */
int gcpnet_edge_force_forward(int64_t E, int s, const float* A, const float* B, const int32_t* row, const int32_t* col,
                              const float* W3, const float* frames, int act, float slope, float* force, void* stream);
int gcpnet_edge_force_backward(int64_t E, int s, const float* A, const float* B, const int32_t* row, const int32_t* col,
                               const float* W3, const float* frames, int act, float slope, const float* d_force,
                               float* d_pre, float* part, void* stream);
int gcpnet_edge_force_bwd_blocks(int64_t E);

/* ---- learnable scalar message gate (components/gcpnet.py:892-896,932-934, `use_scalar_message_attention`, GCPInteractions2):
 * att[r] = sigmoid(<x[r, :], w> + b[0]) with w = scalar_message_attention.0.weight [1, s]; out[r, :] = x[r, :] att[r].
 * s a multiple of 4, <= 1024; x, out, w 16-byte aligned.  Backward: d_x [rows, s] and per-block shares of (d_w [s], d_b in
 * column s) in part [gcpnet_row_gate_bwd_blocks(rows), s + 4] (gcpnet_reduce_partials). */
int gcpnet_row_gate_forward(int64_t rows, int s, const float* x, const float* w, const float* b, float* out, float* att,
                            void* stream);
int gcpnet_row_gate_backward(int64_t rows, int s, const float* x, const float* w, const float* att, const float* d_out,
                             float* d_x, float* part, void* stream);
int gcpnet_row_gate_bwd_blocks(int64_t rows);

/* out[r, j] = sum_k in[r, k] W[k, j], rows x K times a tiny row-major W [K, J] (K * J <= 4096): the per-source-row side of
 * the vector projections of gcpnet_gcp2_forward's v_add tables ([vector_down ; vector_down_frames] applied at the source rows). */
int gcpnet_rows_matmul_small(int64_t rows, int K, int J, const float* in, int64_t ld_in, const float* W, float* out,
                             int64_t ld_out, void* stream);

/* ---- activation between separately launched pieces: y = act(x), or with `grad` != NULL y = grad * act'(x)
 * (models/__init__.py:42-57) */
int gcpnet_activation(int64_t n, const float* x, const float* grad, int act, float slope, float* y, void* stream);

/* ---- frame gate of GCP2 (`frame_gate: true`, components/gcpnet.py:369-384; vectorize, components/__init__.py:329-378):
 * g [rows, ldg >= 9] = vector_out_scale_frames(act_v(s_pre)), frames [rows, 3, 3] (edge frames, or the mean out-edge frames for node
 * rows: vectorize is linear in the frame), w_up_frames = vector_up_frames.weight [vo, 3], vu [rows, vo, 3] = vector_up output:
 *   out[r, o, :] = vu[r, o, :] * act_v(safe_norm(sum_c w_up_frames[o, c] * sum_a g[r, 3c + a] frames[r, a, :])).
 * Backward: d_vu, d_g [rows, ldg] and per-wave shares of d w_up_frames in part [gcpnet_frame_gate_bwd_parts(rows), vo * 3]
 * (gcpnet_reduce_partials). */
int gcpnet_frame_gate_forward(int64_t rows, int vo, const float* g, int ldg, const float* frames, const float* w_up_frames,
                              const float* vu, int act, float slope, float* out, void* stream);
int gcpnet_frame_gate_backward(int64_t rows, int vo, const float* g, int ldg, const float* frames, const float* w_up_frames,
                               const float* vu, int act, float slope, const float* d_out, float* d_vu, float* d_g, float* part,
                               void* stream);
int gcpnet_frame_gate_bwd_parts(int64_t rows);

/* ---- scalarize on node rows with enable_e3_equivariance (replaces components/__init__.py:283-321 for node_inputs=True with the
 * E(3) |.|, the one case the mean out-edge frame cannot express): out[n, 3 k + a] = mean over the out-edges e of n (CSR by source
 * node: seg_ptr [n_nodes + 1], perm = edge ids in segment order or NULL) of f(frames[e, a, :] . vf[n, :, k]), f = |.| for a == 1
 * when e3.  vf: [n_nodes, 3 (xyz), ldk], channel k < 3 innermost.  Forward: d_out = d_vf = NULL, writes out [n_nodes, 9].
 * Backward: d_out [n_nodes, 9] given, writes d_vf (same layout as vf; entries k >= 3 untouched); frames are constants. */
int gcpnet_node_scalarize(int n_nodes, const int32_t* seg_ptr, const int32_t* perm, const float* vf, int ldk, const float* frames,
                          int e3, float* out, const float* d_out, float* d_vf, void* stream);

/* ---- dropout (components/__init__.py:97-135: nn.Dropout on the scalars, VectorDropout on whole 3-vectors), train mode:
 * y[g * group + j] = keep(g) ? x[g * group + j] / keep_prob : 0 with keep(g) = uniform(seed, g) < keep_prob, a counter-based
 * hash: the backward is the same call on the gradient with the same seed (no mask is stored).  group = 1 or 3. */
int gcpnet_dropout(int64_t n_groups, int group, const float* x, float keep_prob, uint64_t seed, float* y, void* stream);

/* ---- host-glue helpers of the small-graph regime (a step of the NMS model is ~600 launches of a few microseconds each) -------------
 * gcpnet_copy2d_multi: n strided 2-D copies dst[r, c] = src[r, c] (strides in floats; src NULL = zero fill) in one launch -- the
 * column slices, zero-padded and transposed forms of the small vector weights that project-then-gather wants, and the pieces of an
 * assembled weight gradient (ATen: one cat / pad / clone / copy_ launch each). */
typedef struct {
    const float* src;
    float* dst;
    int rows, cols;
    int64_t src_rs, src_cs, dst_rs, dst_cs;
} gcp_copy2d_job_t;
#define GCP_COPY2D_MAX_JOBS 48
int gcpnet_copy2d_multi(int n, const gcp_copy2d_job_t* jobs, void* stream);
/* gb = alpha g where lo <= alpha b <= hi (everywhere without `clamp`), else 0: the adjoint of gcpnet_axpy_clamp w.r.t. b
 * (components/gcpnet.py:1156-1158: x + clamp(weight * update, -100, 100)) */
int gcpnet_axpy_clamp_backward(int64_t n, const float* g, const float* b, float alpha, int clamp, float lo, float hi, float* gb,
                               void* stream);

/* ---- fused Adam over many small parameter tensors (the reference's optimizer: torch.optim.Adam, configs/model/gcpnet_*.yaml;
 * amsgrad = False): one launch per GCP_ADAM_MAX_TENSORS tensors instead of ~10 launches per tensor.  `step` counts from 1. */
typedef struct {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t n;
} gcp_adam_tensor_t;
#define GCP_ADAM_MAX_TENSORS 96
int gcpnet_adam_step(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                     int step, void* stream);
/* The same with the step count in device memory (*step_dev = steps taken so far; the call uses *step_dev + 1 for the bias corrections
 * and advances it behind the update): nothing of the call depends on a host value that changes from step to step, so a training step
 * that ends in it can be captured into a hipGraph and replayed (torch.optim.Adam(capturable=True) semantics). */
int gcpnet_adam_step_dev(int n, const gcp_adam_tensor_t* tensors, float lr, float beta1, float beta2, float eps, float weight_decay,
                         int64_t* step_dev, void* stream);

/* ---- input side ---------------------------------------------------------------------------------
 * NMS featuriser, src/datamodules/components/nms_dataset.py:23-61 + helper.py:16-59:
 *   e_out [E, n_attr + n_rbf] = [edge_attr | RBF(|x_row - x_col|; mu = linspace(0, d_max, n_rbf), sigma = d_max / n_rbf)],
 *   xi_out [E, 1, 3] = unit(x_row - x_col) (0 for coincident points: nan_to_num);
 *   h_out [N, 1] = |vel|, chi_out [N, 3, 3] = [vel, unit(x[i+1] - x[i]), unit(x[i-1] - x[i])] with the orientations taken along
 *   the node order inside each graph (`batch` = graph id per node, NULL = one graph; zero at a graph's ends). */
int gcpnet_nms_edge_features(int64_t E, const float* x, const int32_t* row, const int32_t* col, const float* edge_attr, int n_attr,
                             float d_max, int n_rbf, float* e_out, float* xi_out, void* stream);
int gcpnet_nms_node_features(int64_t N, const float* vel, const float* x, const int32_t* batch, float* h_out, float* chi_out,
                             void* stream);
/* ATOM3D / LBA node vectors, src/datamodules/components/atom3d_dataset.py:65-84 (`_node_features`) -> helper.py:52-59
 * (`_orientations`): chi_out [N, 2, 3] = [unit(x[i+1] - x[i]), unit(x[i-1] - x[i])] along the node order inside each graph (`batch`
 * = graph id per node, NULL = one graph; zero rows at a graph's two ends, 0 for coincident points: nan_to_num).  The LBA edge
 * features (`_edge_features`, atom3d_dataset.py:42-62: 16 RBFs of the edge length + the unit difference) are
 * gcpnet_nms_edge_features with n_attr = 0. */
int gcpnet_orientations(int64_t N, const float* x, const int32_t* batch, float* chi_out, void* stream);
/* Radius graph (atom3d_dataset.py:110-112 recipe): for every node the `max_neighbors` (<= 64) nearest other nodes of its graph
 * within `radius`, ascending by distance (ties: lower node id), nbr [N, max_neighbors] (-1 padded) and count [N], indexed by node
 * id.  The caller provides the cell list: nodes sorted by global cell = graph * (nx*ny*nz) + (cz*ny + cy)*nx + cx with cell edge
 * >= radius (x_sorted [N,3], order = node id per sorted position, cell_of, cell_start [n_graphs*nx*ny*nz + 1]).  Edge (row =
 * neighbour, col = node) lists built from it are col-sorted by construction. */
int gcpnet_radius_graph(int N, const float* x_sorted, const int32_t* order, const int32_t* cell_of, const int32_t* cell_start, int nx,
                        int ny, int nz, double radius /* double: r * r is formed as torch_cluster forms it, (float)(double r * double r) */, int max_neighbors, int32_t* nbr, int32_t* count, void* stream);
/* The same cell list with torch_cluster 1.6.0's neighbour selection (`radius_graph(x, r, batch, loop=False, max_num_neighbors)`,
 * the call of atom3d_dataset.py:110-112): candidates are met in ascending node id (its CUDA kernel walks a graph's nodes in index
 * order), distance test strict (d^2 < r^2), the walk stops at max_neighbors + 1 hits WITH the node itself among the candidates, the
 * self loop is removed afterwards -- so a node keeps the lowest ids, not the nearest, and up to max_neighbors + 1 of them when its
 * own id is not among the first max_neighbors + 1 in range.  nbr [N, max_neighbors + 1] (-1 padded), ascending by id. */
int gcpnet_radius_graph_first(int N, const float* x_sorted, const int32_t* order, const int32_t* cell_of, const int32_t* cell_start,
                              int nx, int ny, int nz, double radius /* double: r * r is formed as torch_cluster forms it, (float)(double r * double r) */, int max_neighbors, int32_t* nbr, int32_t* count, void* stream);

/* ---- profiling hook: when `buf` (device memory, n_tiles * 8 uint64) is non-NULL, each 32-row wave-tile of the GCP2
 * forward / backward kernels writes s_memtime stamps at its phase boundaries; NULL switches it off. */
int gcpnet_debug_set_phase_timing(void* buf, int64_t n_tiles);

/* ---- arithmetic switch (tests / A-B measurements).  Several kernels compute their large fp32 products -- W^T ds_pre in
 * gcpnet_gcp2_chain_backward and in gcpnet_wg_backward (plain mode, one K tile per wave), scalar_out over the state and the
 * gate Linear in gcpnet_gcp2_chain_forward, scalar_out in gcpnet_wg_forward (8-wave shapes) -- on the bf16 matrix pipe with
 * both operands split into three bf16 terms and six products kept (fp32 accumulation; the result is exact to fp32 round-off:
 * csrc/gcp_bf16x3.h).  on != 0 selects the v_mfma_f32_32x32x2_f32 form of the same products instead; the environment
 * variables GCPNET_CHAIN_BWD_FP32_MFMA, GCPNET_CHAIN_FWD_FP32_MFMA, GCPNET_WG_BWD_FP32_MFMA and GCPNET_WG_FWD_B6=0 set the
 * initial state per kernel (GCPNET_WG_FWD_B6=all: the bf16 form for the 4-wave shapes of gcpnet_wg_forward too).  Returns
 * the previous setting (-1: never set); on < 0 returns to that state (restoring a saved `previous`). */
int gcpnet_debug_set_fp32_mfma(int on);

/* 1 when the library was built with -DGCP_DEBUG_KNOBS: only such a build honours the measurement knobs that change RESULTS
 * (GCPNET_DEBUG_SKIP_TN, GCPNET_TN_DEBUG).  The shipped build returns 0; bench.py refuses to produce a line with a 1. */
int gcpnet_debug_knobs_compiled(void);

/* Resident workgroups per CU of the pipelined weight-gradient GEMM kernels (0: the 128 x 160 form, 1: the 256 x 288 form) as
 * hipOccupancyMaxActiveBlocksPerMultiprocessor reports them; negative: a HIP error code (tools/tn_occupancy.py). */
int gcpnet_debug_tn_occupancy(int wide);

int gcpnet_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GCPNET_HIP_H */
