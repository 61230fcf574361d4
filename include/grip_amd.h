/* grip_amd.h -- C ABI of the MI355X-native CLIP prompt-tuning + pseudolabel engine.
 *
 * The reference (BatsResearch/menghini-neurips23-code) is pure Python and has no FFI; its
 * "operator API" for the hot path is a set of Python classes over the third-party `clip`
 * package.  Each entry point below replaces the arithmetic behind one of them; the Python
 * host layer in menghini-neurips23-code_amd/{clip,models,utils} keeps the reference's class
 * and function signatures and binds these symbols with ctypes.
 *
 * Conventions: extern "C"; every function returns 0 on success and a non-zero grip_status
 * otherwise (grip_last_error() gives the message); no C++ exception crosses the boundary; the
 * caller owns every buffer; all device work is enqueued on the caller's hipStream_t (passed as
 * void*); handles are not thread-safe (one per process / GPU); no hidden device allocation after
 * grip_tower_create.  Device pointers are plain pointers into HBM (torch tensors' data_ptr()).
 *
 * Dtypes: GEMM operands and the residual stream f16 (MFMA v_mfma_f32_16x16x32_f16, f32 accumulate, adds into the
 * stream in f32); LayerNorm statistics, softmax, head, embeddings and gradients f32.  A tower created with
 * dims.precision = 1 computes everything in f32 instead (exact comparison mode, inference only); dims.precision = 2 is the f32 tower with its
 * block GEMMs and attention on the f16 matrix pipes as hi / lo operand pairs (three products, f32 accumulate: ~22 mantissa bits; inference only).
 */
#ifndef GRIP_AMD_H
#define GRIP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GRIP_OK = 0,
    GRIP_ERR_ARG = 1,        /* bad dimension / null pointer / unsupported shape */
    GRIP_ERR_HIP = 2,        /* a HIP runtime call or kernel launch failed */
    GRIP_ERR_WORKSPACE = 3,  /* workspace too small */
    GRIP_ERR_STATE = 4       /* backward without a matching training-mode forward, tower not finalized */
} grip_status;

const char* grip_last_error(void);
/* ABI version of this header; the host layer refuses a library that reports another one. */
int grip_abi_version(void);
#define GRIP_ABI_VERSION 8

/* ------------------------------------------------------------------------------------------
 * Tower description.  kind 0 = vision transformer (clip_model.visual, wrapped by
 * CustomVisionTransformer, models/clip_encoders.py:105-194), kind 1 = text transformer
 * (clip_model.{token_embedding, positional_embedding, transformer, ln_final, text_projection},
 * wrapped by CustomTextEncoder, models/clip_encoders.py:25-90).  Head dim is width / heads = 64. */
typedef struct {
    int32_t kind;        /* 0 vision, 1 text */
    int32_t width;       /* d: 768 (ViT-B), 1024 (ViT-L); text 512 / 768 */
    int32_t layers;
    int32_t heads;       /* width / 64 */
    int32_t embed_dim;   /* E: 512 / 768 */
    int32_t seq0;        /* tokens without prompt: grid^2 + 1 (vision) or context length 77 (text) */
    int32_t patch;       /* vision: patch size; text: 0 */
    int32_t resolution;  /* vision: input resolution; text: 0 */
    int32_t vocab;       /* text: vocabulary size; vision: 0 */
    int32_t max_prefix;  /* largest number of prompt tokens any call will use (sizes nothing but checks) */
    int32_t precision;   /* 0 = f16 GEMM operands + f16 residual stream (what clip.load gives the reference on a GPU);
                            1 = exact comparison mode: f32 weights, activations, residual stream and attention
                            (v_mfma_f32_16x16x4_f32), i.e. the arithmetic of the reference's CPU path (clip.load(..., "cpu")
                            keeps fp32).  Inference only: train != 0 is GRIP_ERR_ARG on a precision-1 tower.
                            2 = split f16 (the middle tier of the screen-and-refine pseudolabel pass): blobs, activations, residual stream,
                            LayerNorm, softmax and every interface as precision 1; the four GEMMs of a block and the attention products
                            carry each operand as f16 hi + f16 lo (x = hi + lo) and form a b from three v_mfma_f32_16x16x32_f16 with f32
                            accumulation (a_hi b_hi + a_hi b_lo + a_lo b_hi): embeddings within ~1e-6 relative of the precision-1 tower's
                            at ~2.5x its throughput.  grip_tower_finalize derives the split weight copies (slots "...#S").  Inference
                            only; width must be a multiple of 256. */
} grip_dims;

/* Weight layout.  A tower's frozen weights live in two caller-owned device blobs: one f16 (GEMM
 * operands) and one f32 (biases, LayerNorm affine, embeddings).  grip_layout_slot enumerates the
 * slots: `name` is the OpenAI state_dict key relative to the tower (e.g.
 * "transformer.resblocks.3.attn.in_proj_weight", "conv1.weight", "proj") for primary slots the
 * host fills, or ends in "#T" for derived slots (transposed copies used by the backward pass,
 * written by grip_tower_finalize).  Returns GRIP_OK, or GRIP_ERR_ARG when `slot` is past the end. */
typedef struct {
    char name[96];
    int32_t dtype;      /* 0 = GEMM-operand blob (f16; f32 elements when dims.precision != 0), 1 = f32 blob */
    int32_t derived;    /* 1 = written by grip_tower_finalize */
    int64_t offset;     /* element offset inside its blob (16-byte aligned) */
    int64_t rows;       /* logical shape rows x cols, row-major */
    int64_t cols;
    int64_t ld;         /* leading dimension in elements (>= cols; conv1 rows are zero-padded to a multiple of 64) */
} grip_slot;

int grip_layout_slot(const grip_dims* dims, int slot, grip_slot* out);
int grip_layout_size(const grip_dims* dims, int64_t* n_f16, int64_t* n_f32);

typedef struct grip_tower grip_tower;

/* Replaces clip.load's module construction for one tower.  Blobs must outlive the handle.  `f16_blob` is the GEMM-operand
 * blob of n_f16 ELEMENTS: f16, or f32 when dims.precision != 0. */
int grip_tower_create(const grip_dims* dims, void* f16_blob, void* f32_blob, grip_tower** out);
/* Builds derived weights (transposed copies) on `stream`; call after the primary slots are filled
 * and again whenever they change. */
int grip_tower_finalize(grip_tower* t, void* stream);
int grip_tower_destroy(grip_tower* t);

/* Workspace bytes for a forward over `batch` units (images, or class prompts for the text tower)
 * with `n_prefix` prompt tokens.  seq_len: text tower only, see grip_text_forward (0 = full context; pass 0 for the
 * vision tower).  train != 0 keeps the activations backward needs. */
int grip_workspace_bytes(const grip_tower* t, int batch, int n_prefix, int seq_len, int train, size_t* bytes);

/* ------------------------------------------------------------------------------------------
 * CustomVisionTransformer.forward(x, image_prefix) / clip_model.encode_image(x)
 * (models/clip_encoders.py:123-194; :93-102 with n_prefix = 0).
 *   images     [batch, 3, R, R] f32 (images_f16 = 0) or f16 (images_f16 = 1), NCHW
 *   prefix     [n_prefix, width] f32, or NULL when n_prefix == 0; inserted between CLS and the
 *              patches after the positional embedding, shared by the whole batch
 *   out_emb    [batch, embed_dim] f32 (un-normalised, as the reference returns it)
 *   flags      GRIP_FWD_TRAIN (= the `train` argument of earlier ABIs: 1 keeps the activations backward needs) and / or
 *              GRIP_FWD_NO_POS_EMB (forward(..., pos_emb=False), :141: CLS and patches without the positional embedding)
 *   generation NULL, or receives the number of this train-mode forward (0 without GRIP_FWD_TRAIN).  Several train-mode forwards may
 *              be outstanding, each on its own workspace; a second one on the SAME workspace overwrites the first one's
 *              saved activations, and a backward that presents the first one's number then fails with GRIP_ERR_STATE.
 */
int grip_vit_forward(grip_tower* t, const void* images, int images_f16, const float* prefix, int n_prefix,
                     int batch, float* out_emb, void* workspace, size_t workspace_bytes, int flags, uint64_t* generation, void* stream);

/* Input-gradient chain of the frozen ViT down to the prompt slice (autograd of the above w.r.t.
 * image_prefix only; no weight gradients exist).  Must follow a train-mode forward on the same
 * workspace, exactly once per forward; generation = the number that forward returned (0 = do not check).  grad_emb [batch, embed_dim] f32 -> grad_prefix [n_prefix, width] f32 (summed over batch). */
int grip_vit_backward_prefix(grip_tower* t, const float* grad_emb, const float* prefix, float* grad_prefix,
                             void* workspace, size_t workspace_bytes, uint64_t generation, void* stream);

/* CustomTextEncoder.forward(class_embeddings, classes) / clip_model.encode_text(tokens)
 * (models/clip_encoders.py:43-90; :13-22 with n_prefix = 0).  Tokenisation stays on the host.
 *   token_ids  [n_class, seq0] int32 device; eot_index [n_class] int32 device (= token_ids.argmax(-1))
 *   prefix     [prefix_classes, n_prefix, width] f32; prefix_classes is 1 (broadcast, CoOp) or n_class
 *   seq_len    0 or seq0: encode all seq0 positions as the reference does.  0 < seq_len < seq0: encode only the first
 *              seq_len positions; must be > max(eot_index).  The text transformer is causal and only the EOT row is
 *              read, so positions after the last EOT cannot influence any output: the result is the same, the work
 *              shrinks by seq0 / seq_len (CoOp prompts are ~20 of 77 tokens).
 *   out_emb    [n_class, embed_dim] f32
 *   flags      GRIP_FWD_TRAIN: keep the activations grip_text_backward_prefix needs (the `train` argument of ABI <= 3).
 *              GRIP_FWD_SHARED_PREFIX: the caller vouches that token_ids[c][0 .. n_prefix] is the same for every class
 *              (SOT + the context placeholders of CustomTextEncoder, :63-74) and eot_index[c] > n_prefix.  With one shared
 *              context (prefix_classes == 1) those 1 + n_prefix positions then carry identical activations for every class
 *              in every layer -- same inputs, same positions, causal mask -- so the engine encodes them ONCE and keeps
 *              only the class-specific positions per class (1 + n_prefix + n_class * (seq_len - 1 - n_prefix) rows
 *              instead of n_class * seq_len: 425 instead of 2 142 for 102 classes x 21 positions, 16 context tokens); the
 *              class positions attend to the shared keys and their own, and the backward adds every class's share of
 *              the shared keys' gradient in class order.  Embeddings and prompt gradient are the same function of the
 *              inputs; ignored (plain layout) when prefix_classes != 1, n_prefix == 0 or dims.precision != 0.
 *              GRIP_FWD_NO_POS_EMB: CustomTextEncoder.forward(enable_pos_emb=False), x = token / prompt embeddings only.
 */
#define GRIP_FWD_TRAIN 1
#define GRIP_FWD_SHARED_PREFIX 2
#define GRIP_FWD_NO_POS_EMB 4 /* models/clip_encoders.py:70-74, enable_pos_emb=False: the positional embedding is not added */
#define GRIP_FWD_STREAM_HILO 8 /* grip_vit_forward, inference on an f16 tower (ABI 8): the residual stream is carried as a COMPENSATED pair of f16 numbers
                                  (hi + lo, ~22 mantissa bits; every add into it in f32 as before), so the 2 x layers roundings of the stream no longer
                                  accumulate -- measured: 2.5 - 3x less direction error against the f32 tower for ~2 bytes more traffic per stream element and
                                  residual GEMM.  GEMM operands stay f16 (the hi part).  The SCREEN of the pseudolabel pass (grip_amd.pseudolabels) runs in
                                  this mode; train-mode forwards and every other caller keep the plain f16 stream the reference's GPU path has. */
int grip_text_forward(grip_tower* t, const int32_t* token_ids, const int32_t* eot_index, const float* prefix,
                      int n_prefix, int prefix_classes, int n_class, int seq_len, float* out_emb,
                      void* workspace, size_t workspace_bytes, int flags, uint64_t* generation, void* stream);

/* grad_emb [n_class, embed_dim] -> grad_prefix [prefix_classes, n_prefix, width] (summed over classes
 * when prefix_classes == 1). */
int grip_text_backward_prefix(grip_tower* t, const float* grad_emb, float* grad_prefix,
                              void* workspace, size_t workspace_bytes, uint64_t generation, void* stream);

/* ------------------------------------------------------------------------------------------
 * Cosine x logit-scale head + softmax + argmax: the block inlined 45 times in the reference, e.g.
 * methods/semi_supervised_learning/textual_prompt.py:98-109, and utils/clip_pseudolabels.py:35-41.
 *   img_emb [n, e] f32, txt_emb [c, e] f32 (both un-normalised), scale = logit_scale.exp()
 *   logits [n, c] f32; probs [n, c] f32 (softmax) or NULL; argmax_logits / argmax_probs [n] int32 or NULL
 *   txt_norm_scratch [c, e] f32 device scratch
 */
int grip_cosine_head(const float* img_emb, const float* txt_emb, float scale, int n, int c, int e,
                     float* logits, float* probs, int32_t* argmax_logits, int32_t* argmax_probs,
                     float* txt_norm_scratch, void* stream);

/* Backward of logits = scale * normalize(img) @ normalize(txt).T :
 * grad_logits [n, c] -> grad_img [n, e] and/or grad_txt [c, e] (either may be NULL). */
int grip_cosine_head_backward(const float* img_emb, const float* txt_emb, float scale, int n, int c, int e,
                              const float* grad_logits, float* grad_img, float* grad_txt, void* stream);

/* Mean cross-entropy over the rows with row_weight != 0, each weighted: the three FPL losses
 * (methods/semi_supervised_learning/textual_fpl.py:123-165, methods/transductive_zsl/textual_fpl.py:117-147,
 * methods/unsupervised_learning/visual_fpl.py:107-122) are sums of two such masked means; the host
 * passes per-row weights w_i = gamma_group / |group|.  loss [1] f32, grad_logits [n, c] f32. */
int grip_weighted_ce(const float* logits, const int32_t* labels, const float* row_weight, int n, int c,
                     float* loss, float* grad_logits, void* stream);

/* ------------------------------------------------------------------------------------------
 * UPTModel's prompt mixer (models/prompts_models.py:99-119 construct it, :129-146 run it): proj_coop_pre / proj_vpt_pre (Linear
 * to `dim`), clip.model.Transformer(width = dim, layers = 1, heads = 1) over the sequence cat((coop, vpt), dim 0) = [2, P, dim]
 * (sequence length 2, batch P), the .to(float16) round trip of :138-145, proj_coop_post / proj_vpt_post.  The only trainable
 * weights on the path: grip_upt_mixer_backward returns the gradient of EVERY tensor of the struct (float32 values; the
 * reference's float16 branch, multimodal_prompt.py:46, is the same kernels with fp16 rounding points: half_linears).
 * One struct describes the tensors (forward: values; backward's `grads`: buffers of the same shapes that receive the gradients):
 *   coop [P, text_width], vpt [P, vision_width]                 coop_embeddings / vpt_embeddings (P = n_prompt tokens each)
 *   coop_pre_w [dim, text_width], coop_pre_b [dim]              proj_coop_pre;   vpt_pre_* likewise with vision_width
 *   ln1_g, ln1_b [dim]; in_w [3 dim, dim], in_b [3 dim]; out_w [dim, dim], out_b [dim]      resblocks.0.ln_1 / attn.in_proj_* / attn.out_proj
 *   ln2_g, ln2_b [dim]; fc_w [4 dim, dim], fc_b [4 dim]; proj_w [dim, 4 dim], proj_b [dim]  resblocks.0.ln_2 / mlp.c_fc / mlp.c_proj
 *   coop_post_w [text_width, dim], coop_post_b [text_width]     proj_coop_post;  vpt_post_* likewise
 * forward:  coop_out [P, text_width], vpt_out [P, vision_width] (what CustomTextEncoder / CustomImageEncoder receive as prompts).
 * backward: d_coop_out / d_vpt_out = the prompt gradients grip_text_backward_prefix / grip_vit_backward_prefix returned; must follow
 * a forward on the same workspace (it holds the saved activations).  The gradient entering the fp16 tensor is rounded to fp16 as
 * autograd does.  All device pointers f32; work is enqueued on `stream`; no atomics (bit-reproducible). */
typedef struct {
    int32_t n_prompt, text_width, vision_width, dim;
    int32_t half_linears;   /* 1 = the reference's float16 branch (multimodal_prompt.py:46 on a GPU): the four projection Linears and the two prompt embeddings are
                               fp16 tensors around the fp32 block.  The host passes their values as f32 (exactly representable); the outputs of the pre / post
                               projections are rounded to fp16, and in the backward so are the gradient entering proj_*_pre (it crosses the fp16 -> fp32 cast
                               in front of the block) and the prompt gradients.  Parameter gradients come back f32; the host casts them to the parameters' dtype. */
    int32_t reserved_;
    float *coop, *vpt;
    float *coop_pre_w, *coop_pre_b, *vpt_pre_w, *vpt_pre_b;
    float *ln1_g, *ln1_b, *in_w, *in_b, *out_w, *out_b, *ln2_g, *ln2_b, *fc_w, *fc_b, *proj_w, *proj_b;
    float *coop_post_w, *coop_post_b, *vpt_post_w, *vpt_post_b;
} grip_upt_mixer;
int grip_upt_mixer_workspace(int n_prompt, int text_width, int vision_width, int dim, size_t* bytes);
int grip_upt_mixer_forward(const grip_upt_mixer* m, float* coop_out, float* vpt_out, void* workspace, size_t workspace_bytes, void* stream);
int grip_upt_mixer_backward(const grip_upt_mixer* m, const float* d_coop_out, const float* d_vpt_out, const grip_upt_mixer* grads,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * CLIP preprocessing of one decoded image: the `_transform` the reference applies per item on the host
 * (data/dataset.py:64-79 via clip.load's preprocess): Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize.
 * Bit-exact with Pillow's 8-bit bicubic resample; only the cropped rows / columns are produced.
 *   img [H, W, 3] u8 device; out [3, n_px, n_px] f32 device; tmp [H * n_px * 3] u8 device scratch
 *   hcoef [W_out, hksize] / hbounds [W_out, 2] and vcoef [H_out, vksize] / vbounds [H_out, 2]: int32 device tables of
 *   Pillow's precompute_coeffs + normalize_coeffs_8bpc (22-bit fixed point; bounds = first input index, count);
 *   hksize = vksize = 0 (tables NULL): the image already has the resized size, crop + normalise only.
 *   mean3 / std3: HOST pointers to 3 floats. */
int grip_preprocess_image(const uint8_t* img, int H, int W,
                          const int32_t* hcoef, const int32_t* hbounds, int hksize, int W_out,
                          const int32_t* vcoef, const int32_t* vbounds, int vksize, int H_out,
                          int crop_left, int crop_top, int n_px, const float* mean3, const float* std3,
                          uint8_t* tmp, float* out, void* stream);

/* The same for a whole batch of decoded images of different sizes in ONE launch pair (the host decodes JPEGs on a thread pool and
 * uploads the batch as one packed buffer).  `items_device` is a DEVICE array of n_items descriptors; every pointer inside is a
 * device pointer; hksize = vksize = 0 marks an image that already has the resized size.  max_H = the largest H of the batch.
 * mean3 / std3: HOST pointers to 3 floats. */
typedef struct {
    const uint8_t* img;        /* [H, W, 3] u8 */
    int32_t H, W;
    const int32_t* hcoef;      /* [W_out, hksize] */
    const int32_t* hbounds;    /* [W_out, 2] */
    int32_t hksize, W_out;
    const int32_t* vcoef;      /* [H_out, vksize] */
    const int32_t* vbounds;    /* [H_out, 2] */
    int32_t vksize, H_out;
    int32_t crop_left, crop_top;
    uint8_t* tmp;              /* [H, n_px, 3] u8 scratch */
    float* out;                /* [3, n_px, n_px] f32 */
} grip_preprocess_item;
int grip_preprocess_batch(const grip_preprocess_item* items_device, int n_items, int max_H, int n_px,
                          const float* mean3, const float* std3, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sequential per-class leaderboard: utils/clip_pseudolabels.py:49-112 and the nine
 * assign_pseudo_labels (e.g. methods/transductive_zsl/multimodal_fpl.py:194-285).  Host function,
 * exact: order-dependent, one pass in dataset order.
 *   probs [n, c] f32 host; pred [n] int32 host (arg-max the caller took: of probs in
 *   clip_pseudolabels.py:39, of logits in assign_pseudo_labels)
 *   path_rank [n] int64 host: rank of image i's path string among all paths (ties in the score are
 *   broken by the path, descending; equal strings get equal ranks)
 *   out_img / out_class: capacity c * min(k, n); *out_count = pairs emitted, boards concatenated
 *   in class order, each in its final list order.
 */
int grip_leaderboard_scan(const float* probs, const int32_t* pred, const int64_t* path_rank,
                          int64_t n, int c, int64_t k, int32_t* out_img, int32_t* out_class, int64_t* out_count);

/* The same scan over probabilities that are only known to a per-row relative accuracy (screen and refine): rows encoded by the
 * f16 towers carry rel_eps[i] > 0, rows re-encoded by an exact (precision = 1) tower carry 0.  Every order the reference's scan
 * depends on (the arg-max, utils/clip_pseudolabels.py:39; `board[-1].score < score`, :75 / :95; the sorted inserts, :79-82 /
 * :97-100) is decided on the intervals [p (1 - eps), p (1 + eps)].  ambiguous[i] is set to 1 for every row with rel_eps[i] > 0 that
 * took part in a comparison the intervals could not decide (an undecidable arg-max only counts when one of the candidate classes
 * could admit the image: when all of them certainly reject it the image spills to every class whichever of them is the true
 * arg-max).  *n_ambiguous == 0  =>  the lists written are the lists grip_leaderboard_scan returns on the TRUE probabilities,
 * provided every |true_i[j] - p_i[j]| <= rel_eps[i] * p_i[j] + abs_eps (abs_eps: absolute slack of the un-refined rows, for softmax outputs in
 * the denormal range, which have no relative accuracy; rows with rel_eps[i] == 0 are final and carry none).  A row may also carry the bound of an
 * intermediate tier (0 < rel_eps[i] << the screen's): any mix of per-row bounds is valid.  Otherwise the caller re-encodes the marked rows exactly, sets their
 * rel_eps to 0 and calls again; the marked set grows strictly, so the loop ends (host side: grip_amd.pseudolabels.refine_scan).
 *   k == 10000000 (the reference's "label everything" branch, :27-44): out_img / out_class need capacity n and every row
 *   whose arg-max is undecidable is marked.  Otherwise capacity c * min(k, n) as above.
 *   ambiguous [n] uint8 host (overwritten).
 *   bound_form (ABI 8): 0 = the RELATIVE form above.  1 = LOG-ODDS: rel_eps[i] is delta_i and the proviso reads "the ODDS p / (1 - p) of every
 *   entry of row i are within a factor e^{+-delta_i} of the true ones": the interval of an entry is [p / (p + (1 - p) e^delta),
 *   p / (p + (1 - p) e^-delta)] (with 2^-20 of slack for the f32 evaluation of p and 1 - p, and abs_eps).  It is what an error of a cheaper tower's
 *   embedding direction produces -- the LOGITS move by scale * <de, t_c> whatever the probabilities are, and the softmax
 *   (utils/clip_pseudolabels.py:38) turns a spread d of those errors over the classes into a factor within e^{+-d} on every entry's odds.  For
 *   p << 1 it is the relative form with eps = e^delta - 1; for the p ~ 0.9+ entries that sit on the board thresholds of a peaked pool it is
 *   (1 - p) times tighter.  delta < 700.
 *   threads (ABI 8): worker threads of the scan's pre-filter; 0 = $GRIP_SCAN_THREADS, else the CPUs the process may use divided by
 *   $LOCAL_WORLD_SIZE (at most 16).  The lists and marks do not depend on it. */
int grip_leaderboard_scan_bounded(const float* probs, const int32_t* pred, const int64_t* path_rank, const float* rel_eps, float abs_eps,
                                  int bound_form, int threads, int64_t n, int c, int64_t k, int32_t* out_img, int32_t* out_class,
                                  int64_t* out_count, uint8_t* ambiguous, int64_t* n_ambiguous);

/* ------------------------------------------------------------------------------------------
 * clip.tokenize's byte-level BPE (host function; the reference tokenises on every CustomTextEncoder.forward,
 * models/clip_encoders.py:60, and in utils/clip_pseudolabels.py:25).
 *   grip_bpe_create        merges = the text of the merges table, one "a b" pair per line in rank order (the lines of
 *                          bpe_simple_vocab_16e6.txt after its header, in the byte-to-unicode alphabet); ids follow openai/CLIP:
 *                          256 bytes, 256 end-of-word bytes, the merges, <|startoftext|>, <|endoftext|>
 *   grip_bpe_encode_word   one pre-token (raw UTF-8 bytes of one match of the CLIP pre-tokenisation pattern) -> ids
 *   grip_bpe_encode_ascii  a cleaned, lower-cased ASCII text: pre-tokenisation (the CLIP pattern restricted to ASCII) + BPE */
typedef struct grip_bpe grip_bpe;
int grip_bpe_create(const char* merges, size_t n_bytes, grip_bpe** out);
int grip_bpe_destroy(grip_bpe* t);
int grip_bpe_special_ids(const grip_bpe* t, int32_t* sot, int32_t* eot, int32_t* vocab_size);
int grip_bpe_encode_word(grip_bpe* t, const uint8_t* word, int n, int32_t* ids, int cap, int* n_out);
int grip_bpe_encode_ascii(grip_bpe* t, const char* text, int n, int32_t* ids, int cap, int* n_out);

/* ------------------------------------------------------------------------------------------
 * Launch width of the persistent kernels (ABI 7).  The pool-encode GEMMs and the pipelined attention size their grids to the WHOLE chip (one
 * workgroup per CU, each walking many tiles).  A host that runs an encode on a CU-masked stream (hipExtStreamCreateWithCUMask) next to other work --
 * the frozen image tower's look-ahead encode of a textual-prompt epoch beside the latency-bound prompt steps, textual_prompt.py:95-103 -- tells the
 * library how many CUs that stream owns, so that the persistent grids fit them (a 256-workgroup grid on 192 CUs would run in two rounds).
 * n_cus = 0 restores the full width.  Process-wide host state read at launch time: set it around the launches it is meant for.  Results do not
 * depend on it (tile -> workgroup assignment never changes an element's arithmetic). */
int grip_set_cu_budget(int n_cus);

/* ------------------------------------------------------------------------------------------
 * Multi-GPU exchange (one process per GPU, RCCL over the xGMI mesh).  The unlabeled pool shards contiguously over the
 * ranks and the per-rank embeddings are all-gathered once per pass; this replaces the accelerator.gather
 * sites of the reference (e.g. methods/semi_supervised_learning/textual_prompt.py:146-147, 285-286) and, for the trainable
 * prompts, DDP's gradient all-reduce behind accelerator.backward (textual_prompt.py:131).
 *   grip_comm_unique_id   rank 0 fills 128 bytes (an ncclUniqueId) and hands them to the other ranks through any side
 *                         channel (torch.distributed store, a file, MPI)
 *   grip_comm_init_rank   collective over the n_ranks processes; the calling thread's current HIP device is the rank's GPU
 *   grip_allgather_embeddings   local [rows_per_rank, e] f32 device -> global [n_ranks * rows_per_rank, e] f32 device, rank-major
 *                         (the caller pads the last shard to rows_per_rank rows and drops the padding afterwards)
 *   grip_allreduce_mean   in place: every rank ends with the mean over ranks of grads[0..n)
 * Both collectives are enqueued on `stream`.  RCCL is loaded at run time ($GRIP_RCCL_LIBRARY, else librccl.so). */
typedef struct grip_comm grip_comm;
int grip_comm_unique_id(uint8_t* id128);
int grip_comm_init_rank(const uint8_t* id128, int n_ranks, int rank, grip_comm** out);
int grip_comm_destroy(grip_comm* comm);
int grip_allgather_embeddings(grip_comm* comm, const float* local, float* global, int64_t rows_per_rank, int e, void* stream);
int grip_allreduce_mean(grip_comm* comm, float* grads, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIP_AMD_H */
