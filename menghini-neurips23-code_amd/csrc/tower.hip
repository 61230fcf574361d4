// Tower handle, weight layout and the forward/backward launch sequences behind the C ABI
// (include/grip_amd.h).  A tower is the frozen CLIP ViT (kind 0) or text transformer (kind 1);
// the launch sequence follows CustomVisionTransformer.forward (models/clip_encoders.py:123-194)
// and CustomTextEncoder.forward (:43-90) of the reference, with the per-block arithmetic of the
// published openai/CLIP ResidualAttentionBlock: x += out_proj(MHSA(ln_1 x)); x += c_proj(QuickGELU(c_fc(ln_2 x))).
//
// HBM layout: residual stream x (resid_t = f16, accumulated in f32 inside the GEMM epilogues) [B*S, d] row-major (token-major, so every GEMM sees one
// [M, K] x [N, K]^T problem over all images of the chunk); GEMM operands (LayerNorm output, packed
// qkv [M, 3d], attention output, MLP hidden [M, 4d]) f16.  Row counts are padded to the 128-row GEMM
// tile in the workspace; padding rows are never stored to and never read back.
#include <stdarg.h>
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "common.h"

// ---------------------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
void grip_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* grip_last_error(void) { return g_err; }
extern "C" int grip_abi_version(void) { return GRIP_ABI_VERSION; }

static int g_cu_budget = 0;
int grip_cu_budget() { return g_cu_budget; }
extern "C" int grip_set_cu_budget(int n_cus) {
    GRIP_REQUIRE(n_cus >= 0 && (n_cus == 0 || n_cus >= 8), "set_cu_budget: %d (0 = all, else at least 8)", n_cus);
    g_cu_budget = n_cus;
    return GRIP_OK;
}

// ---------------------------------------------------------------------------------------------- layout
struct LayerW {
    // f16 (element offsets into the f16 blob)
    int64_t in_w, out_w, fc_w, proj_w;          // [3d,d] [d,d] [4d,d] [d,4d]
    int64_t in_wT, out_wT, fc_wT, proj_wT;      // derived transposes: [d,3d] [d,d] [d,4d] [4d,d]
    int64_t in_wG, fc_wG;                       // derived: LayerNorm-folded operands f16(gamma o W) of the QKV and c_fc GEMMs
    int64_t in_wS = -1, out_wS = -1, fc_wS = -1, proj_wS = -1;   // derived, precision 2 only: the four weights in the split layout of gemm_split.hip
    // f32
    int64_t ln1_g, ln1_b, in_b, out_b, ln2_g, ln2_b, fc_b, proj_b;
    int64_t in_cs, in_bb, fc_cs, fc_bb;         // derived: colsum(W') and W beta + b of the two folded GEMMs
};
struct Layout {
    std::vector<grip_slot> slots;
    std::vector<LayerW> layer;
    int64_t n16 = 0, n32 = 0;
    int kpad = 0;
    // vision
    int64_t conv_w = -1, cls = -1, pos = -1, lnpre_g = -1, lnpre_b = -1, lnpost_g = -1, lnpost_b = -1, proj = -1, projT = -1;
    // text
    int64_t tok = -1;
};

static int64_t add_slot(Layout& L, const std::string& name, int dtype, int derived, int64_t rows, int64_t cols, int64_t ld = 0) {
    grip_slot s;
    memset(&s, 0, sizeof(s));
    snprintf(s.name, sizeof(s.name), "%s", name.c_str());
    s.dtype = dtype;
    s.derived = derived;
    s.rows = rows;
    s.cols = cols;
    s.ld = ld ? ld : cols;
    int64_t& n = dtype == 0 ? L.n16 : L.n32;
    n = round_up64(n, 64);  // 128-byte (f16) / 256-byte (f32) aligned slots
    s.offset = n;
    n += rows * s.ld;
    L.slots.push_back(s);
    return s.offset;
}

static int build_layout(const grip_dims& D, Layout& L) {
    GRIP_REQUIRE(D.kind == 0 || D.kind == 1, "dims.kind must be 0 (vision) or 1 (text)");
    GRIP_REQUIRE(D.width > 0 && D.width % 128 == 0 && D.heads * 64 == D.width, "width must be a multiple of 128 with head dim 64 (width=%d heads=%d)", D.width, D.heads);
    GRIP_REQUIRE(D.embed_dim > 0 && D.embed_dim % 128 == 0, "embed_dim must be a multiple of 128");
    GRIP_REQUIRE(D.layers > 0 && D.seq0 > 0, "layers / seq0 must be positive");
    const int64_t d = D.width;
    if (D.kind == 0) {
        GRIP_REQUIRE(D.patch > 0 && D.resolution % D.patch == 0, "vision: resolution %% patch != 0");
        const int g = D.resolution / D.patch;
        GRIP_REQUIRE(D.seq0 == g * g + 1, "vision: seq0 must be grid^2 + 1");
        L.kpad = (int)round_up64(3 * D.patch * D.patch, 64);
        L.conv_w = add_slot(L, "conv1.weight", 0, 0, d, 3 * D.patch * D.patch, L.kpad);
        L.cls = add_slot(L, "class_embedding", 1, 0, 1, d);
        L.pos = add_slot(L, "positional_embedding", 1, 0, D.seq0, d);
        L.lnpre_g = add_slot(L, "ln_pre.weight", 1, 0, 1, d);
        L.lnpre_b = add_slot(L, "ln_pre.bias", 1, 0, 1, d);
    } else {
        GRIP_REQUIRE(D.vocab > 0, "text: vocab must be positive");
        L.tok = add_slot(L, "token_embedding.weight", 1, 0, D.vocab, d);
        L.pos = add_slot(L, "positional_embedding", 1, 0, D.seq0, d);
    }
    L.layer.resize((size_t)D.layers);
    for (int i = 0; i < D.layers; ++i) {
        LayerW& w = L.layer[(size_t)i];
        const std::string p = "transformer.resblocks." + std::to_string(i) + ".";
        w.ln1_g = add_slot(L, p + "ln_1.weight", 1, 0, 1, d);
        w.ln1_b = add_slot(L, p + "ln_1.bias", 1, 0, 1, d);
        w.in_w = add_slot(L, p + "attn.in_proj_weight", 0, 0, 3 * d, d);
        w.in_b = add_slot(L, p + "attn.in_proj_bias", 1, 0, 1, 3 * d);
        w.out_w = add_slot(L, p + "attn.out_proj.weight", 0, 0, d, d);
        w.out_b = add_slot(L, p + "attn.out_proj.bias", 1, 0, 1, d);
        w.ln2_g = add_slot(L, p + "ln_2.weight", 1, 0, 1, d);
        w.ln2_b = add_slot(L, p + "ln_2.bias", 1, 0, 1, d);
        w.fc_w = add_slot(L, p + "mlp.c_fc.weight", 0, 0, 4 * d, d);
        w.fc_b = add_slot(L, p + "mlp.c_fc.bias", 1, 0, 1, 4 * d);
        w.proj_w = add_slot(L, p + "mlp.c_proj.weight", 0, 0, d, 4 * d);
        w.proj_b = add_slot(L, p + "mlp.c_proj.bias", 1, 0, 1, d);
        w.in_wT = add_slot(L, p + "attn.in_proj_weight#T", 0, 1, d, 3 * d);
        w.out_wT = add_slot(L, p + "attn.out_proj.weight#T", 0, 1, d, d);
        w.fc_wT = add_slot(L, p + "mlp.c_fc.weight#T", 0, 1, d, 4 * d);
        w.proj_wT = add_slot(L, p + "mlp.c_proj.weight#T", 0, 1, 4 * d, d);
        w.in_wG = add_slot(L, p + "attn.in_proj_weight#G", 0, 1, 3 * d, d);
        w.fc_wG = add_slot(L, p + "mlp.c_fc.weight#G", 0, 1, 4 * d, d);
        w.in_cs = add_slot(L, p + "attn.in_proj#colsum", 1, 1, 1, 3 * d);
        w.in_bb = add_slot(L, p + "attn.in_proj#bias", 1, 1, 1, 3 * d);
        w.fc_cs = add_slot(L, p + "mlp.c_fc#colsum", 1, 1, 1, 4 * d);
        w.fc_bb = add_slot(L, p + "mlp.c_fc#bias", 1, 1, 1, 4 * d);
        if (D.precision == 2) {     // [N, K/32, 32 hi | 32 lo'] f16: as many bytes as the f32 matrix (the operand blob has f32 elements)
            w.in_wS = add_slot(L, p + "attn.in_proj_weight#S", 0, 1, 3 * d, d);
            w.out_wS = add_slot(L, p + "attn.out_proj.weight#S", 0, 1, d, d);
            w.fc_wS = add_slot(L, p + "mlp.c_fc.weight#S", 0, 1, 4 * d, d);
            w.proj_wS = add_slot(L, p + "mlp.c_proj.weight#S", 0, 1, d, 4 * d);
        }
    }
    const char* lnf = D.kind == 0 ? "ln_post" : "ln_final";
    L.lnpost_g = add_slot(L, std::string(lnf) + ".weight", 1, 0, 1, d);
    L.lnpost_b = add_slot(L, std::string(lnf) + ".bias", 1, 0, 1, d);
    const char* pj = D.kind == 0 ? "proj" : "text_projection";
    L.proj = add_slot(L, pj, 0, 0, d, D.embed_dim);
    L.projT = add_slot(L, std::string(pj) + "#T", 0, 1, D.embed_dim, d);
    L.n16 = round_up64(L.n16, 64);
    L.n32 = round_up64(L.n32, 64);
    return GRIP_OK;
}

extern "C" int grip_layout_slot(const grip_dims* dims, int slot, grip_slot* out) {
    GRIP_REQUIRE(dims && out, "layout_slot: null pointer");
    try {
        Layout L;
        int rc = build_layout(*dims, L);
        if (rc) return rc;
        GRIP_REQUIRE(slot >= 0 && slot < (int)L.slots.size(), "layout_slot: slot %d past the end (%d)", slot, (int)L.slots.size());
        *out = L.slots[(size_t)slot];
        return GRIP_OK;
    } catch (...) { grip_set_error("layout_slot: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_layout_size(const grip_dims* dims, int64_t* n_f16, int64_t* n_f32) {
    GRIP_REQUIRE(dims && n_f16 && n_f32, "layout_size: null pointer");
    try {
        Layout L;
        int rc = build_layout(*dims, L);
        if (rc) return rc;
        *n_f16 = L.n16;
        *n_f32 = L.n32;
        return GRIP_OK;
    } catch (...) { grip_set_error("layout_size: exception"); return GRIP_ERR_ARG; }
}

// ---------------------------------------------------------------------------------------------- handle
// Activation buffers are declared half_t* but hold f32 elements in exact mode (element size `es` below); the forward passes
// them on as untyped pointers + the f32 flag, and the backward (f16 towers only) uses them as declared.
struct Workspace {  // carve of the caller's buffer for one (batch, n_prefix, train) problem
    int batch = 0, P = 0, train = 0, S = 0, M = 0;
    int64_t Mp = 0;
    resid_t* x = nullptr;      // inference residual stream
    half_t* xn = nullptr;
    half_t* qkv = nullptr;
    half_t* att = nullptr;
    half_t* h = nullptr;
    half_t* cls16 = nullptr;   // [round_up(batch,128), d]
    half_t* patches = nullptr; // alias of h
    float* patch_out = nullptr;// alias of qkv
    half_t* row_x = nullptr;   // compact last-block buffers (f16 towers): [Bp, d] stream rows, [Bp, d] attention rows,
    half_t* row_att = nullptr; // [Bp, d] LayerNorm output, [Bp, 4d] MLP hidden
    half_t* row_xn = nullptr;
    half_t* row_h = nullptr;
    // ... and, when a TRAIN-mode forward runs its last block for the read rows only (rows_last): what that block saves for its backward
    // and the backward's own compact scratch
    int rows_last = 0;
    int ks_row = 1;            // split factor of the compact fc input-gradient GEMM
    half_t* row_hpre = nullptr;   // [Bp, 4d] pre-activations
    resid_t* row_xmid = nullptr;  // [Bp, d] stream rows after the attention branch
    resid_t* row_xout = nullptr;  // [Bp, d] stream rows after the block (what ln_post / ln_final reads)
    float* drow = nullptr;        // [Bp, d] f32 gradient of those stream rows
    half_t* drow_h = nullptr;     // ... its f16 copy (GEMM operand)
    half_t* row_dh = nullptr;     // [Bp, 4d]
    float* row_dln = nullptr;     // [ks_row][Bp, d]
    half_t* row_datt = nullptr;   // [Bp, d]
    float* stat_part = nullptr;// [d/64, M, 2] partial row sums emitted by the residual GEMM epilogues (f16 towers)
    float* rowstat = nullptr;  // [Mp, 2] (mean, rstd) of the residual stream's rows, consumed by the LayerNorm-folded GEMMs
    half_t* row_x_lo = nullptr;   // ... and of the last block's compact read rows
    half_t* x_lo = nullptr;    // inference, f16 towers: the lo parts of the stream when the forward runs with GRIP_FWD_STREAM_HILO (GemmArgs::resid_lo)
    int hilo = 0;              // ... and whether this forward does
    // train-mode saves, one per layer (x_in has layers+1 entries)
    std::vector<resid_t*> x_in, x_mid;
    std::vector<half_t*> qkv_l, att_l, hpre_l;
    // backward scratch
    float* dx = nullptr;
    float* dln = nullptr;      // [max(ks_fc, ks_in)][Mp, d] f32 split-K partials of the EPI_F32 input-gradient GEMMs
    int ks_fc = 1, ks_in = 1;  // their split factors (gemm_pick_ksplit)
    // shared-prefix layout of the text tower (common.h, seq_row): Ps = 1 + P shared leading rows, then S - Ps rows per class
    int Ps = 0;
    int rs = 0;                // rows between consecutive sequences for the (sequence, position) -> row gathers: S, or S - Ps
    float* kv_part = nullptr;  // [batch, Ps, 2, d] f32: per-class shares of the shared keys' dK / dV (train)
    // cooperative split-K of the train-mode c_proj GEMM (text tower; GemmArgs::coop_scratch): factor, partial tiles, (tile, wave) tickets
    int coop_ks = 1;
    float* coop_scratch = nullptr;
    int* coop_cnt = nullptr;
    size_t coop_cnt_bytes = 0;
    half_t* dxh = nullptr;
    half_t* dh = nullptr;
    half_t* dqkv = nullptr;
    half_t* datt = nullptr;
    float* dcls = nullptr;     // [round_up(batch,128), d] f32
    half_t* gemb16 = nullptr;  // [round_up(batch,128), E] f16
    float* scale = nullptr;    // [2] loss scale and its inverse
    size_t bytes = 0;
};

struct grip_tower {
    grip_dims D;
    Layout L;
    half_t* w16;      // GEMM-operand blob: f16, or f32 when f32 != 0 (exact mode) -- always addressed through wop()
    float* w32;
    int f32 = 0;      // dims.precision != 0: activations, residual stream, attention and the operand blob's primary slots are f32 (inference only)
    int w_exact = 0;  // precision 2: no block weight has a non-zero lo part (set by grip_tower_finalize; GemmArgs::w_exact)
    int split = 0;    // dims.precision == 2: the four GEMMs of a block run on the split-f16 kernel (gemm_split.hip); their A operands are written in
                      // its layout by the producers (LayerNorm, f32 attention, GELU epilogue), their weights by grip_tower_finalize (#S slots)
    uint64_t generation = 0;   // counts train-mode forwards (see `pending`)
    const void* wop(int64_t elem_off) const { return (const char*)w16 + elem_off * (f32 ? 4 : 2); }
    void* wop(int64_t elem_off) { return (char*)w16 + elem_off * (f32 ? 4 : 2); }
    bool finalized = false;
    // State of the train-mode forwards whose activations are still waiting for their backward, keyed by workspace: several
    // forwards may be outstanding at once (model(aug_1) and model(aug_2) before one loss.backward()), each on its own
    // workspace.  A second train-mode forward on the SAME workspace overwrites the first one's activations; its generation
    // number replaces the first one's, so a backward that presents the old generation fails with GRIP_ERR_STATE instead
    // of returning gradients of the wrong forward.
    struct TrainState {
        Workspace w;
        const int32_t* eot = nullptr;
        int prefix_classes = 0;
        uint64_t generation = 0;
        bool consumed = false;     // its backward has run (the backward works in place on the saved activations: one per forward)
    };
    std::map<void*, TrainState> pending;
};

// GRIP_LAST_BLOCK_FULL=1: the last block runs for every row, as the reference's does (A/B; the tests hold the two paths equal)
static bool last_block_full() {
    static const bool full = getenv("GRIP_LAST_BLOCK_FULL") && atoi(getenv("GRIP_LAST_BLOCK_FULL")) != 0;
    return full;
}

// Train-mode forwards of the text tower run LayerNorm-folded GEMMs (run_blocks).
static bool train_fold(const grip_tower* t) {
    // GRIP_TRAIN_FOLD (developer A/B): 0 = off, 1 = text tower only, 2 = both towers (default)
    static const int mode = getenv("GRIP_TRAIN_FOLD") ? atoi(getenv("GRIP_TRAIN_FOLD")) : 2;
    return !t->f32 && (mode >= 2 || (mode == 1 && t->D.kind == 1));
}

static int carve(const grip_tower* t, int batch, int P, int train, char* base, Workspace& w, int seq_len = 0, int shared = 0) {
    const grip_dims& D = t->D;
    GRIP_REQUIRE(batch > 0 && P >= 0 && P <= D.max_prefix, "batch must be positive and 0 <= n_prefix <= max_prefix (batch=%d n_prefix=%d max=%d)", batch, P, D.max_prefix);
    const int64_t d = D.width;
    const size_t es = t->f32 ? 4 : 2;      // activation / operand element size
    GRIP_REQUIRE(!(t->f32 && train), "exact (f32) towers are inference-only: no train-mode workspace");
    w.batch = batch; w.P = P; w.train = train;
    GRIP_REQUIRE(seq_len >= 0 && seq_len <= D.seq0 && (D.kind == 1 || seq_len == 0), "seq_len %d out of range", seq_len);
    w.S = D.kind == 0 ? D.seq0 + P : (seq_len ? seq_len : D.seq0);
    GRIP_REQUIRE(D.kind == 0 || P < w.S - 1, "text: n_prefix %d does not fit the sequence length %d", P, w.S);
    GRIP_REQUIRE(w.S <= 608, "sequence length %d exceeds the fused-attention limit 608", w.S);
    GRIP_REQUIRE(!shared || (D.kind == 1 && P > 0 && !t->f32), "the shared-prefix layout is the f16 text tower's, with a context (n_prefix > 0)");
    w.Ps = shared ? P + 1 : 0;
    w.rs = w.S - w.Ps;
    w.M = w.Ps + batch * (int64_t)w.rs;
    w.Mp = round_up64(w.M, 256);
    const int64_t Bp = round_up64(batch, 256);
    size_t off = 0;
    auto take = [&](size_t nbytes) { char* p = base ? base + off : nullptr; off += (nbytes + 255) / 256 * 256; return (void*)p; };
    if (!train) w.x = (resid_t*)take(w.Mp * d * es);
    if (!train && !t->f32) w.x_lo = (half_t*)take(w.Mp * d * 2);      // lo parts of a compensated stream (GRIP_FWD_STREAM_HILO)
    w.xn = (half_t*)take(w.Mp * d * es);
    if (!train) { w.qkv = (half_t*)take(w.Mp * 3 * d * es); w.att = (half_t*)take(w.Mp * d * es); }
    w.h = (half_t*)take(w.Mp * 4 * d * es);
    w.cls16 = (half_t*)take(Bp * d * es);
    w.rows_last = train && !t->f32 && !shared && !last_block_full();
    if (!train || w.rows_last) {       // compact [batch, .] buffers of a last block that runs for the read rows only (f32 / split towers: 4-byte elements)
        w.row_x = (half_t*)take(Bp * d * es);
        if (!train && !t->f32) w.row_x_lo = (half_t*)take(Bp * d * 2);
        w.row_att = (half_t*)take(Bp * d * es);
        w.row_xn = (half_t*)take(Bp * d * es);
        w.row_h = (half_t*)take(Bp * 4 * d * es);
    }
    if (w.rows_last) {
        w.row_hpre = (half_t*)take(Bp * 4 * d * 2);
        w.row_xmid = (resid_t*)take(Bp * d * sizeof(resid_t));
        w.row_xout = (resid_t*)take(Bp * d * sizeof(resid_t));
        w.drow = (float*)take(Bp * d * 4);
        w.drow_h = (half_t*)take(Bp * d * 2);
        w.row_dh = (half_t*)take(Bp * 4 * d * 2);
        w.ks_row = gemm_pick_ksplit(batch, d, 4 * d);
        w.row_dln = (float*)take(Bp * d * 4 * (size_t)w.ks_row);
        w.row_datt = (half_t*)take(Bp * d * 2);
    }
    if (!t->f32) {
        w.stat_part = (float*)take(w.Mp * (d / 64) * 2 * sizeof(float));
        w.rowstat = (float*)take(w.Mp * 2 * sizeof(float));
    }
    if (D.kind == 0) {
        const int64_t G2 = D.seq0 - 1;
        const int64_t prow = round_up64(batch * G2, 256);
        w.patches = prow * t->L.kpad <= w.Mp * 4 * d ? w.h : (half_t*)take(prow * t->L.kpad * es);   // alias of h when it fits
        w.patch_out = (float*)take(batch * G2 * d * 4);
    }
    if (train) {
        const int Lc = D.layers;
        w.x_in.assign((size_t)Lc + 1, nullptr); w.x_mid.assign((size_t)Lc, nullptr);
        w.qkv_l.assign((size_t)Lc, nullptr); w.att_l.assign((size_t)Lc, nullptr); w.hpre_l.assign((size_t)Lc, nullptr);
        for (int i = 0; i <= Lc; ++i) w.x_in[(size_t)i] = (resid_t*)take(w.Mp * d * sizeof(resid_t));
        for (int i = 0; i < Lc; ++i) {
            w.x_mid[(size_t)i] = (resid_t*)take(w.Mp * d * sizeof(resid_t));
            w.qkv_l[(size_t)i] = (half_t*)take(w.Mp * 3 * d * 2);
            w.att_l[(size_t)i] = (half_t*)take(w.Mp * d * 2);
            w.hpre_l[(size_t)i] = (half_t*)take(w.Mp * 4 * d * 2);
        }
        w.dx = (float*)take(w.Mp * d * 4);
        // split-K partial buffers of the two EPI_F32 input-gradient GEMMs of a block (summed by ln_bwd_add)
        w.ks_fc = gemm_pick_ksplit((int)w.M, d, 4 * d);
        w.ks_in = gemm_pick_ksplit((int)w.M, d, 3 * d);
        w.dln = (float*)take(w.Mp * d * 4 * (size_t)(w.ks_fc > w.ks_in ? w.ks_fc : w.ks_in));
        w.dxh = (half_t*)take(w.Mp * d * 2);
        w.dh = (half_t*)take(w.Mp * 4 * d * 2);
        w.dqkv = (half_t*)take(w.Mp * 3 * d * 2);
        w.datt = (half_t*)take(w.Mp * d * 2);
        w.dcls = (float*)take(Bp * d * 4);
        w.gemb16 = (half_t*)take(Bp * D.embed_dim * 2);
        w.scale = (float*)take(256);
        if (w.Ps) w.kv_part = (float*)take((size_t)batch * w.Ps * 2 * d * 4);
        if (D.kind == 1 && !t->f32) {      // text tower, f16
            w.coop_ks = gemm_pick_coop_split((int)w.M, d, 4 * d);
            if (w.coop_ks > 1) {
                const size_t tiles = (size_t)((w.M + 63) / 64) * (size_t)(d / 128);
                w.coop_scratch = (float*)take(tiles * (size_t)w.coop_ks * 4 * 2048 * sizeof(float));
                w.coop_cnt_bytes = tiles * 4 * sizeof(int);
                w.coop_cnt = (int*)take(w.coop_cnt_bytes);
            }
        }
    }
    w.bytes = off;
    return GRIP_OK;
}

extern "C" int grip_tower_create(const grip_dims* dims, void* f16_blob, void* f32_blob, grip_tower** out) {
    GRIP_REQUIRE(dims && f16_blob && f32_blob && out, "tower_create: null pointer");
    try {
        grip_tower* t = new grip_tower();
        t->D = *dims;
        if (dims->precision < 0 || dims->precision > 2) { delete t; GRIP_REQUIRE(false, "dims.precision must be 0 (f16), 1 (f32 exact) or 2 (split f16)"); }
        if (dims->precision == 2 && dims->width % 256 != 0) { delete t; GRIP_REQUIRE(false, "precision 2 (split f16) needs width %% 256 == 0 (256 x 256 GEMM tiles); width = %d", dims->width); }
        t->f32 = dims->precision != 0;
        t->split = dims->precision == 2;
        int rc = build_layout(*dims, t->L);
        if (rc) { delete t; return rc; }
        t->w16 = (half_t*)f16_blob;
        t->w32 = (float*)f32_blob;
        *out = t;
        return GRIP_OK;
    } catch (...) { grip_set_error("tower_create: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_tower_destroy(grip_tower* t) {
    delete t;
    return GRIP_OK;
}

extern "C" int grip_tower_finalize(grip_tower* t, void* stream) {
    GRIP_REQUIRE(t, "tower_finalize: null handle");
    hipStream_t s = (hipStream_t)stream;
    const int d = t->D.width;
    int rc;
    const int f = t->f32;
    if (!f)       // the per-layer transposes feed the dgrad GEMMs only: exact towers have no backward
        for (const LayerW& w : t->L.layer) {
            if ((rc = launch_transpose(t->wop(w.in_w), t->wop(w.in_wT), f, 3 * d, d, d, s))) return rc;
            if ((rc = launch_transpose(t->wop(w.out_w), t->wop(w.out_wT), f, d, d, d, s))) return rc;
            if ((rc = launch_transpose(t->wop(w.fc_w), t->wop(w.fc_wT), f, 4 * d, d, d, s))) return rc;
            if ((rc = launch_transpose(t->wop(w.proj_w), t->wop(w.proj_wT), f, d, 4 * d, 4 * d, s))) return rc;
        }
    if (!f)       // LayerNorm-folded operands of the QKV and c_fc GEMMs (EPI_LNFOLD_*): W' = f16(gamma o W), colsum(W'), W beta + b
        for (const LayerW& w : t->L.layer) {
            float* F = t->w32;
            if ((rc = launch_ln_fold_weights(t->w16 + w.in_w, F + w.ln1_g, F + w.ln1_b, F + w.in_b, t->w16 + w.in_wG, F + w.in_cs, F + w.in_bb, 3 * d, d, s))) return rc;
            if ((rc = launch_ln_fold_weights(t->w16 + w.fc_w, F + w.ln2_g, F + w.ln2_b, F + w.fc_b, t->w16 + w.fc_wG, F + w.fc_cs, F + w.fc_bb, 4 * d, d, s))) return rc;
        }
    if (t->split) { // split-layout copies of the block weights (the f32 originals stay: patch embedding and the final projection use them)
        int* flag = nullptr;        // [0] raised by the split kernel when a scaled weight leaves the f16 range, [1] when a weight is not an f16 number
        GRIP_CHECK_HIP(hipMalloc((void**)&flag, 2 * sizeof(int)));      // (finalize is a one-off: a sync here is fine)
        hipError_t e = hipMemsetAsync(flag, 0, 2 * sizeof(int), s);
        rc = e == hipSuccess ? GRIP_OK : GRIP_ERR_HIP;
        for (const LayerW& w : t->L.layer) {
            if (rc) break;
            if ((rc = launch_split_rows((const float*)t->wop(w.in_w), t->wop(w.in_wS), 3 * d, d, d, s, 1, flag, flag + 1))) break;
            if ((rc = launch_split_rows((const float*)t->wop(w.out_w), t->wop(w.out_wS), d, d, d, s, 1, flag, flag + 1))) break;
            if ((rc = launch_split_rows((const float*)t->wop(w.fc_w), t->wop(w.fc_wS), 4 * d, d, d, s, 1, flag, flag + 1))) break;
            if ((rc = launch_split_rows((const float*)t->wop(w.proj_w), t->wop(w.proj_wS), d, 4 * d, 4 * d, s, 1, flag, flag + 1))) break;
        }
        int flags[2] = {0, 0};
        if (!rc && (hipMemcpyAsync(flags, flag, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = GRIP_ERR_HIP;
        const int over = flags[0];
        t->w_exact = flags[1] == 0;         // fp16 checkpoints: no block weight needs its lo part (GemmArgs::w_exact)
        (void)hipFree(flag);
        if (rc == GRIP_ERR_HIP) { grip_set_error("tower_finalize: HIP error while splitting the weights"); return rc; }
        if (rc) return rc;
        GRIP_REQUIRE(!over, "tower_finalize: a block weight times %g leaves the f16 range (or is not finite): split-f16 towers (precision 2) need |w| < %g",
                     (double)gemm_split_weight_scale(), 65504.0 / (double)gemm_split_weight_scale());
    }
    if ((rc = launch_transpose(t->wop(t->L.proj), t->wop(t->L.projT), f, d, t->D.embed_dim, t->D.embed_dim, s))) return rc;
    t->finalized = true;
    return GRIP_OK;
}

extern "C" int grip_workspace_bytes(const grip_tower* t, int batch, int n_prefix, int seq_len, int train, size_t* bytes) {
    GRIP_REQUIRE(t && bytes, "workspace_bytes: null pointer");
    try {
        Workspace w;
        int rc = carve(t, batch, n_prefix, train, nullptr, w, seq_len);
        if (rc) return rc;
        *bytes = w.bytes;
        if (t->D.kind == 1 && n_prefix > 0 && !t->f32) {     // either row layout of a text forward fits (GRIP_FWD_SHARED_PREFIX)
            Workspace ws;
            rc = carve(t, batch, n_prefix, train, nullptr, ws, seq_len, 1);
            if (rc) return rc;
            if (ws.bytes > *bytes) *bytes = ws.bytes;
        }
        return GRIP_OK;
    } catch (...) { grip_set_error("workspace_bytes: exception"); return GRIP_ERR_ARG; }
}

// ---------------------------------------------------------------------------------------------- forward
#define RUN(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// One tower pass over the residual stream.  f16 towers never run a stand-alone LayerNorm inside the blocks: the statistics of
// every row travel with the stream -- the assembly / embedding kernel writes (mean, rstd) of x0, each residual GEMM epilogue
// emits the partial sums of the rows it stores and a 104-byte-per-row kernel finalises them -- and ln_1 / ln_2 are folded
// into the QKV / c_fc GEMMs, whose A operand is the raw stream (EPI_LNFOLD_*: rstd (x W'^T - mean colsum(W')) + (W beta + b)).
// That removes two full read + write passes over the stream per block (8.8 % of the pool encode's GPU time in round 1).
// Exact (f32) towers keep the literal LayerNorm -> GEMM sequence.
static int run_blocks(grip_tower* t, Workspace& w, resid_t* x0, int causal, const int32_t* read_rows, hipStream_t s, resid_t** x_final, bool* compact) {
    const int d = t->D.width, H = t->D.heads, f = t->f32;
    const int gf = t->split ? 2 : f;      // GemmArgs.f32 of the four block GEMMs, and the output layout of what feeds them
    const float* F = t->w32;
    resid_t* x = x0;
    const int parts = d / 64;
    // Train-mode forwards (the prompt-tuning batches: a few thousand rows) keep the literal LayerNorm -> GEMM sequence: there the
    // stream is L2-resident and a LayerNorm launch costs ~5 us, while the statistics-carrying epilogues make the 20-us GEMMs of
    // those steps 5 % slower (r02: VPT step 3.8 -> 4.0 ms with the fold).  The switch is the MODE, never the batch size: every
    // inference call computes a row the same way whatever chunk it arrives in (sharded / re-chunked encodes stay bit-identical).
    // r04: train-mode forwards fold as well, without a finalising launch: the GEMM that consumes a stream adds the producer's partial row sums itself
    // (GemmArgs::stat_in; gemm.hip row_stat, fetched per ROW before the K loop).  For the text tower's few hundred rows every kernel is a launch, not a byte
    // count (24 LayerNorm launches of 4.7 us in a 1.5-ms CoOp step); for the image tower's 3 408 rows the folded epilogues cost 1 - 4 us per GEMM against the
    // 5.5-us LayerNorm they replace (VPT step 2.98 -> 2.95 ms, UPT 3.29 -> 3.24 ms, 22 launches fewer; r02's form -- statistics loaded per row group in the
    // epilogue plus a finalising launch -- had been 5 % slower).  The backward is unchanged: it differentiates LayerNorm from the saved stream rows.
    // GRIP_TRAIN_FOLD = 0 / 1 / 2: off / text tower only / both (developer A/B).
    const bool fold = !f && (!w.train || train_fold(t));
    const bool parts_in = fold && w.train;     // consumers read stat_part directly
    // Last block at inference: only ONE row per sequence of the final stream is ever read (CLS: ln_post(x[:, 0]),
    // models/clip_encoders.py:189; EOT: :86-89), and past the block's attention rows do not mix.  So the block computes K and V for
    // every row but its attention output, out-proj, LayerNorm and MLP for that row alone (M = batch instead of batch x S):
    // 2.2 of the block's 2.9 GFLOP per ViT-B/16 image are never issued, the embedding is unchanged.  GRIP_LAST_BLOCK_FULL=1
    // computes the whole block as the reference does (A/B; the tests hold the two paths equal).
    const bool rows_only = fold && !w.train && !last_block_full() && !w.Ps;
    // Compensated stream (GRIP_FWD_STREAM_HILO; inference, folded f16 towers): the two residual epilogues of a block read hi + lo and write both; everything
    // else (the GEMMs' A operand, the last block's gathered rows, ln_post) reads the hi part, i.e. the stream as an f16 tower has always seen it.
    const bool hilo = w.hilo && fold && !w.train && w.x_lo;
    *compact = false;
    for (int l = 0; l < t->D.layers; ++l) {
        const LayerW& lw = t->L.layer[(size_t)l];
        const bool last = l + 1 == t->D.layers;
        if (last && rows_only) {
            // K and V of every row (columns d .. 3d of the packed projection); Q only for the rows that are read
            GemmArgs a{};
            a.A = x; a.W = t->w16 + lw.in_wG + (int64_t)d * d; a.M = w.M; a.m_pad = w.Mp; a.N = 2 * d; a.K = d; a.bias = F + lw.in_bb + d; a.colsum = F + lw.in_cs + d;
            a.rowstat = w.rowstat; a.out = w.qkv + d; a.ldc = 3 * d;
            RUN(launch_gemm(EPI_LNFOLD_F16, a, s));
            RUN(launch_gather_rows(x, read_rows, w.S, w.row_x, w.batch, d, s));
            if (hilo) RUN(launch_gather_rows(w.x_lo, read_rows, w.S, w.row_x_lo, w.batch, d, s));      // the read rows keep their compensation through the block's two adds
            RUN(launch_layernorm_f16(w.row_x, F + lw.ln1_g, F + lw.ln1_b, w.row_xn, 0, w.batch, d, s));
            a = GemmArgs{};
            a.A = w.row_xn; a.W = t->w16 + lw.in_w; a.M = w.batch; a.m_pad = round_up64(w.batch, 256); a.N = d; a.K = d; a.bias = F + lw.in_b; a.out = w.row_h; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_F16, a, s));
            RUN(launch_attention_row(w.qkv, w.row_h, read_rows, w.row_att, w.batch, w.S, H, causal, s));
            const int64_t Bp = round_up64(w.batch, 256);
            a = GemmArgs{};
            a.A = w.row_att; a.W = t->w16 + lw.out_w; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = d; a.bias = F + lw.out_b; a.resid = w.row_x; a.out = w.row_x; a.ldc = d;
            if (hilo) { a.resid_lo = w.row_x_lo; a.stat_part = w.stat_part; }      // (the compensated form lives in the statistics-carrying epilogue; nobody reads these sums)
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            RUN(launch_layernorm_f16(w.row_x, F + lw.ln2_g, F + lw.ln2_b, w.row_xn, 0, w.batch, d, s));
            a = GemmArgs{};
            a.A = w.row_xn; a.W = t->w16 + lw.fc_w; a.M = w.batch; a.m_pad = Bp; a.N = 4 * d; a.K = d; a.bias = F + lw.fc_b; a.out = w.row_h; a.ldc = 4 * d;
            RUN(launch_gemm(EPI_BIAS_GELU_F16, a, s));
            a = GemmArgs{};
            a.A = w.row_h; a.W = t->w16 + lw.proj_w; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = 4 * d; a.bias = F + lw.proj_b; a.resid = w.row_x; a.out = w.row_x; a.ldc = d;
            if (hilo) { a.resid_lo = w.row_x_lo; a.stat_part = w.stat_part; }
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            x = w.row_x;
            *compact = true;
            break;
        }
        if (last && f && !w.train && !last_block_full() && !w.Ps) {
            // The same for the f32 / split-f16 towers (r04: the refinement tiers and the exact mode computed the whole 12th block -- 6 % of their FLOPs -- for
            // one row per image): ln_1 and the K / V projection for every row, then the read rows alone: Q, a one-row f32 attention, out-proj, ln_2, MLP.
            const int64_t Bp = round_up64(w.batch, 256);
            const size_t wb = (size_t)d * d * 4;                      // bytes of d weight rows (f32 elements, or the split layout: the same row pitch)
            const char* in_w = (const char*)t->wop(t->split ? lw.in_wS : lw.in_w);
            float* qkv32 = (float*)(void*)w.qkv;
            GemmArgs a{};
            RUN(launch_layernorm_f16(x, F + lw.ln1_g, F + lw.ln1_b, w.xn, gf, w.M, d, s));
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.xn; a.W = in_w + wb; a.M = w.M; a.m_pad = w.Mp; a.N = 2 * d; a.K = d; a.bias = F + lw.in_b + d; a.out = qkv32 + d; a.ldc = 3 * d;
            RUN(launch_gemm(EPI_BIAS_F16, a, s));
            RUN(launch_gather_rows4(x, read_rows, w.S, w.row_x, w.batch, d, s));
            RUN(launch_gather_rows4(w.xn, read_rows, w.S, w.row_xn, w.batch, d, s));
            a = GemmArgs{};
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.row_xn; a.W = in_w; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = d; a.bias = F + lw.in_b; a.out = w.row_h; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_F16, a, s));
            RUN(launch_attention_row_f32(qkv32, (const float*)(const void*)w.row_h, read_rows, (float*)(void*)w.row_att, w.batch, w.S, H, causal, s, t->split));
            a = GemmArgs{};
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.row_att; a.W = t->wop(t->split ? lw.out_wS : lw.out_w); a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = d; a.bias = F + lw.out_b;
            a.resid = w.row_x; a.out = w.row_x; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            RUN(launch_layernorm_f16(w.row_x, F + lw.ln2_g, F + lw.ln2_b, w.row_xn, gf, w.batch, d, s));
            a = GemmArgs{};
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.row_xn; a.W = t->wop(t->split ? lw.fc_wS : lw.fc_w); a.M = w.batch; a.m_pad = Bp; a.N = 4 * d; a.K = d; a.bias = F + lw.fc_b; a.out = w.row_h; a.ldc = 4 * d;
            RUN(launch_gemm(EPI_BIAS_GELU_F16, a, s));
            a = GemmArgs{};
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.row_h; a.W = t->wop(t->split ? lw.proj_wS : lw.proj_w); a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = 4 * d; a.bias = F + lw.proj_b;
            a.resid = w.row_x; a.out = w.row_x; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            x = w.row_x;
            *compact = true;
            break;
        }
        if (last && w.rows_last) {
            // The same in TRAIN mode (prompt steps; plain row layout): the block's K, V and -- one GEMM, saved for the backward -- Q of every
            // row, then attention output, out-proj, ln_2 and the MLP for the read rows alone, each saving what its backward needs in compact
            // [batch, .] buffers.  The backward (run_blocks_backward) mirrors it: 0.14 ms of a 3.15-ms VPT step.
            const int64_t Bp = round_up64(w.batch, 256);
            half_t* qkv = w.qkv_l[(size_t)l];
            GemmArgs a{};
            RUN(launch_layernorm_f16(x, F + lw.ln1_g, F + lw.ln1_b, w.xn, 0, w.M, d, s));
            a.rot_rows = 1;
            a.A = w.xn; a.W = t->w16 + lw.in_w; a.M = w.M; a.m_pad = w.Mp; a.N = 3 * d; a.K = d; a.bias = F + lw.in_b; a.out = qkv; a.ldc = 3 * d;
            RUN(launch_gemm(EPI_BIAS_F16, a, s));
            RUN(launch_attention_row(qkv, nullptr, read_rows, w.row_att, w.batch, w.S, H, causal, s, /*train=*/1));
            RUN(launch_gather_rows(x, read_rows, w.S, w.row_x, w.batch, d, s));
            a = GemmArgs{};
            a.A = w.row_att; a.W = t->w16 + lw.out_w; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = d; a.bias = F + lw.out_b; a.resid = w.row_x; a.out = w.row_xmid; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            RUN(launch_layernorm_f16(w.row_xmid, F + lw.ln2_g, F + lw.ln2_b, w.row_xn, 0, w.batch, d, s));
            a = GemmArgs{};
            a.A = w.row_xn; a.W = t->w16 + lw.fc_w; a.M = w.batch; a.m_pad = Bp; a.N = 4 * d; a.K = d; a.bias = F + lw.fc_b; a.out = w.row_h; a.out2 = w.row_hpre; a.ldc = 4 * d;
            RUN(launch_gemm(EPI_BIAS_GELU_F16, a, s));
            a = GemmArgs{};
            a.A = w.row_h; a.W = t->w16 + lw.proj_w; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = 4 * d; a.bias = F + lw.proj_b; a.resid = w.row_xmid; a.out = w.row_xout; a.ldc = d;
            RUN(launch_gemm(EPI_BIAS_RESID, a, s));
            x = w.row_xout;
            *compact = true;
            break;
        }
        half_t* qkv = w.train ? w.qkv_l[(size_t)l] : w.qkv;
        half_t* att = w.train ? w.att_l[(size_t)l] : w.att;
        resid_t* x_mid = w.train ? w.x_mid[(size_t)l] : x;
        resid_t* x_out = w.train ? w.x_in[(size_t)l + 1] : x;
        GemmArgs a{};
        a.rot_rows = w.train;     // train-mode GEMMs may stagger their K walks by tile row (common.h)
        if (!fold) {
            RUN(launch_layernorm_f16(x, F + lw.ln1_g, F + lw.ln1_b, w.xn, gf, w.M, d, s));
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.xn; a.W = t->wop(t->split ? lw.in_wS : lw.in_w); a.M = w.M; a.m_pad = w.Mp; a.N = 3 * d; a.K = d; a.bias = F + lw.in_b; a.out = qkv; a.ldc = 3 * d;
            RUN(launch_gemm(EPI_BIAS_F16, a, s));
            if (t->split && attention_split_supported(w.S)) RUN(launch_attention_fwd_split((const float*)(const void*)qkv, att, w.batch, w.S, H, causal, s));
            else if (f) RUN(launch_attention_fwd_f32((const float*)(const void*)qkv, (float*)(void*)att, w.batch, w.S, H, causal, s, t->split));
            else RUN(launch_attention_fwd(qkv, att, w.batch, w.S, H, causal, s, w.Ps));
        } else {
            a.A = x; a.W = t->w16 + lw.in_wG; a.M = w.M; a.m_pad = w.Mp; a.N = 3 * d; a.K = d; a.bias = F + lw.in_bb; a.colsum = F + lw.in_cs; a.rowstat = w.rowstat;
            if (parts_in && l > 0) { a.stat_in = w.stat_part; a.stat_parts = parts; }      // (block 0: the embedding kernel wrote rowstat)
            a.out = qkv; a.ldc = 3 * d;
            RUN(launch_gemm(EPI_LNFOLD_F16, a, s));
            RUN(launch_attention_fwd(qkv, att, w.batch, w.S, H, causal, s, w.Ps));
        }
        a = GemmArgs{};
        a.rot_rows = w.train;
        a.f32 = gf; a.w_exact = t->w_exact; a.A = att; a.W = t->wop(t->split ? lw.out_wS : lw.out_w); a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = d; a.bias = F + lw.out_b; a.resid = x; a.out = x_mid; a.ldc = d;
        a.stat_part = fold ? w.stat_part : nullptr;
        a.resid_lo = hilo ? w.x_lo : nullptr;
        RUN(launch_gemm(EPI_BIAS_RESID, a, s));
        a = GemmArgs{};
        a.rot_rows = w.train;
        if (!fold) {
            RUN(launch_layernorm_f16(x_mid, F + lw.ln2_g, F + lw.ln2_b, w.xn, gf, w.M, d, s));
            a.f32 = gf; a.w_exact = t->w_exact; a.A = w.xn; a.W = t->wop(t->split ? lw.fc_wS : lw.fc_w); a.bias = F + lw.fc_b;
        } else {
            if (parts_in) { a.stat_in = w.stat_part; a.stat_parts = parts; }
            else RUN(launch_ln_stats_finalize(w.stat_part, parts, w.rowstat, w.M, d, s));
            a.A = x_mid; a.W = t->w16 + lw.fc_wG; a.bias = F + lw.fc_bb; a.colsum = F + lw.fc_cs; a.rowstat = w.rowstat;
        }
        a.M = w.M; a.m_pad = w.Mp; a.N = 4 * d; a.K = d; a.out = w.h; a.ldc = 4 * d;
        a.out2 = w.train ? w.hpre_l[(size_t)l] : nullptr;
        RUN(launch_gemm(fold ? EPI_LNFOLD_GELU_F16 : EPI_BIAS_GELU_F16, a, s));
        a = GemmArgs{};
        a.rot_rows = w.train;
        a.f32 = gf; a.w_exact = t->w_exact; a.A = w.h; a.W = t->wop(t->split ? lw.proj_wS : lw.proj_w); a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = 4 * d; a.bias = F + lw.proj_b; a.resid = x_mid; a.out = x_out; a.ldc = d;
        a.stat_part = (!fold || (last && !hilo)) ? nullptr : w.stat_part;     // the final LayerNorm (CLS / EOT rows only) reads the stream itself
        a.resid_lo = hilo ? w.x_lo : nullptr;      // (the compensated form lives in the statistics-carrying epilogue: the last block's unread sums are its price)
        if (w.train && w.coop_ks > 1) { a.ksplit = w.coop_ks; a.coop_scratch = w.coop_scratch; a.coop_counter = w.coop_cnt; }
        RUN(launch_gemm(EPI_BIAS_RESID, a, s));
        if (fold && !last && !parts_in) RUN(launch_ln_stats_finalize(w.stat_part, parts, w.rowstat, w.M, d, s));
        x = x_out;
    }
    *x_final = x;
    return GRIP_OK;
}

static int check_ws(grip_tower* t, int batch, int P, int train, void* ws, size_t ws_bytes, Workspace& w, int seq_len = 0, int shared = 0) {
    GRIP_REQUIRE(t && ws, "null tower / workspace");
    if (!t->finalized) { grip_set_error("tower not finalized: call grip_tower_finalize after filling the weight blobs"); return GRIP_ERR_STATE; }
    RUN(carve(t, batch, P, train, (char*)ws, w, seq_len, shared));
    if (w.bytes > ws_bytes) { grip_set_error("workspace too small: need %zu bytes, got %zu", w.bytes, ws_bytes); return GRIP_ERR_WORKSPACE; }
    GRIP_REQUIRE(((uintptr_t)ws & 255) == 0, "workspace must be 256-byte aligned");
    return GRIP_OK;
}

// Book-keeping of a finished forward: a train-mode one registers its state under the workspace (replacing whatever
// forward used that workspace before) and hands out a fresh generation number; an inference one invalidates it.
static void note_forward(grip_tower* t, void* workspace, int train, const Workspace& w, const int32_t* eot, int prefix_classes, uint64_t* generation) {
    if (train) {
        // Consumed entries only serve a clearer error message ("already back-propagated"); workspaces that were freed since (pools dropped with a
        // key change, graph pins released per GRIP iteration) would otherwise stay in the map for good, and a recycled address could hit a stale one.
        if (t->pending.size() > 32)
            for (auto it = t->pending.begin(); it != t->pending.end();)
                it = (it->second.consumed && it->first != workspace) ? t->pending.erase(it) : std::next(it);
        grip_tower::TrainState& st = t->pending[workspace];
        st.w = w; st.eot = eot; st.prefix_classes = prefix_classes; st.generation = ++t->generation; st.consumed = false;
        if (generation) *generation = st.generation;
    } else {
        t->pending.erase(workspace);
        if (generation) *generation = 0;
    }
}

extern "C" int grip_vit_forward(grip_tower* t, const void* images, int images_f16, const float* prefix, int n_prefix,
                                int batch, float* out_emb, void* workspace, size_t workspace_bytes, int flags, uint64_t* generation, void* stream) {
    try {
        GRIP_REQUIRE(t && t->D.kind == 0, "vit_forward: not a vision tower");
        GRIP_REQUIRE((flags & ~(GRIP_FWD_TRAIN | GRIP_FWD_NO_POS_EMB | GRIP_FWD_STREAM_HILO)) == 0, "vit_forward: unknown flag bits 0x%x", flags);
        const int train = flags & GRIP_FWD_TRAIN;
        GRIP_REQUIRE(!(flags & GRIP_FWD_STREAM_HILO) || (!train && !t->f32), "vit_forward: GRIP_FWD_STREAM_HILO is an inference mode of the f16 towers");
        GRIP_REQUIRE(images && out_emb && (n_prefix == 0 || prefix), "vit_forward: null pointer");
        Workspace w;
        RUN(check_ws(t, batch, n_prefix, train, workspace, workspace_bytes, w));
        hipStream_t s = (hipStream_t)stream;
        const grip_dims& D = t->D;
        const int d = D.width, G2 = D.seq0 - 1, f = t->f32;
        const float* F = t->w32;
        RUN(launch_im2col(images, images_f16, w.patches, f, batch, D.resolution, D.patch, t->L.kpad, s));
        GemmArgs a{};
        a.f32 = f; a.A = w.patches; a.W = t->wop(t->L.conv_w); a.M = batch * G2; a.m_pad = round_up64((int64_t)batch * G2, 256); a.N = d; a.K = t->L.kpad; a.out = w.patch_out; a.ldc = d;
        RUN(launch_gemm(EPI_F32, a, s));
        resid_t* x0 = train ? w.x_in[0] : w.x;
        w.hilo = (flags & GRIP_FWD_STREAM_HILO) ? 1 : 0;
        RUN(launch_vit_assemble_ln(w.patch_out, F + t->L.cls, (flags & GRIP_FWD_NO_POS_EMB) ? nullptr : F + t->L.pos, prefix, n_prefix, F + t->L.lnpre_g, F + t->L.lnpre_b, x0, f, (train && !train_fold(t)) ? nullptr : w.rowstat, batch, G2, d, s,
                                   w.hilo ? w.x_lo : nullptr));
        resid_t* xf = nullptr;
        bool compact = false;
        RUN(run_blocks(t, w, x0, /*causal=*/0, nullptr, s, &xf, &compact));
        RUN(launch_gather_ln_f16(xf, nullptr, compact ? 1 : w.S, F + t->L.lnpost_g, F + t->L.lnpost_b, w.cls16, f, batch, d, s));
        a = GemmArgs{};
        a.f32 = f; a.A = w.cls16; a.W = t->wop(t->L.projT); a.M = batch; a.N = D.embed_dim; a.K = d; a.out = out_emb; a.ldc = D.embed_dim;
        RUN(launch_gemm(EPI_F32, a, s));
        note_forward(t, workspace, train, w, nullptr, 0, generation);
        return GRIP_OK;
    } catch (...) { grip_set_error("vit_forward: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_text_forward(grip_tower* t, const int32_t* token_ids, const int32_t* eot_index, const float* prefix,
                                 int n_prefix, int prefix_classes, int n_class, int seq_len, float* out_emb,
                                 void* workspace, size_t workspace_bytes, int flags, uint64_t* generation, void* stream) {
    try {
        GRIP_REQUIRE(t && t->D.kind == 1, "text_forward: not a text tower");
        GRIP_REQUIRE(token_ids && eot_index && out_emb && (n_prefix == 0 || prefix), "text_forward: null pointer");
        GRIP_REQUIRE(n_prefix == 0 || prefix_classes == 1 || prefix_classes == n_class, "text_forward: prefix_classes must be 1 or n_class");
        GRIP_REQUIRE((flags & ~(GRIP_FWD_TRAIN | GRIP_FWD_SHARED_PREFIX | GRIP_FWD_NO_POS_EMB)) == 0, "text_forward: unknown flag bits 0x%x", flags);
        const int train = flags & GRIP_FWD_TRAIN;
        // the caller vouches for identical tokens at positions 0 .. n_prefix in every class; exact (f32) towers keep the plain layout
        const int shared = (flags & GRIP_FWD_SHARED_PREFIX) && n_prefix > 0 && prefix_classes == 1 && !t->f32;
        Workspace w;
        RUN(check_ws(t, n_class, n_prefix, train, workspace, workspace_bytes, w, seq_len, shared));
        hipStream_t s = (hipStream_t)stream;
        const grip_dims& D = t->D;
        const int d = D.width, f = t->f32;
        const float* F = t->w32;
        resid_t* x0 = train ? w.x_in[0] : w.x;
        if (train && w.coop_cnt) GRIP_CHECK_HIP(hipMemsetAsync(w.coop_cnt, 0, w.coop_cnt_bytes, s));     // the tickets start at zero (and return to it)
        RUN(launch_text_embed(token_ids, D.seq0, F + t->L.tok, (flags & GRIP_FWD_NO_POS_EMB) ? nullptr : F + t->L.pos, prefix, n_prefix, prefix_classes, x0, f, (train && !train_fold(t)) ? nullptr : w.rowstat, n_class, w.S, d, D.vocab, s, w.Ps));
        resid_t* xf = nullptr;
        bool compact = false;
        RUN(run_blocks(t, w, x0, /*causal=*/1, eot_index, s, &xf, &compact));
        RUN(launch_gather_ln_f16(xf, compact ? nullptr : eot_index, compact ? 1 : w.rs, F + t->L.lnpost_g, F + t->L.lnpost_b, w.cls16, f, n_class, d, s));
        GemmArgs a{};
        a.f32 = f; a.A = w.cls16; a.W = t->wop(t->L.projT); a.M = n_class; a.N = D.embed_dim; a.K = d; a.out = out_emb; a.ldc = D.embed_dim;
        RUN(launch_gemm(EPI_F32, a, s));
        note_forward(t, workspace, train, w, eot_index, prefix_classes, generation);
        return GRIP_OK;
    } catch (...) { grip_set_error("text_forward: exception"); return GRIP_ERR_ARG; }
}

// ---------------------------------------------------------------------------------------------- test hooks
// Kernel-level entry points for the unit parity tests (tests/test_gpu_kernels.py).  Not part of the
// drop-in ABI (declared in include/grip_amd_debug.h, not in grip_amd.h); they launch exactly the kernels the towers use.
extern "C" int grip_debug_gemm(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid,
                               const void* aux, void* out, void* out2, float scalar, int m_pad, int variant, void* stream) {
    GemmArgs a{};
    a.variant = variant & 0xff;
    a.rot_rows = (variant >> 8) & 1;                  // bit 8: the train-mode launches' row-dependent K rotation
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.m_pad = m_pad; a.bias = bias; a.resid = resid;
    a.aux = aux; a.out = out; a.out2 = out2; a.ldc = N; a.scalar = scalar;
    if (a.variant == 7) { a.f32 = 1; a.variant = 0; } // f32 operands: the exact-mode kernel (gemm_f32.hip)
    return launch_gemm(epi, a, (hipStream_t)stream);
}
// Split-K EPI_F32 product: out holds `ksplit` partial [M, N] buffers `split_stride` floats apart (ksplit = 0: the launcher's
// own choice, returned through *ksplit_used); their sum in index order is what ln_bwd_add consumes.
extern "C" int grip_debug_gemm_splitk(const void* A, const void* W, int M, int N, int K, float* out, int ksplit, int64_t split_stride, int* ksplit_used,
                                      int m_pad, int variant, void* stream) {
    GemmArgs a{};
    a.variant = variant & 0xff;
    a.rot_rows = (variant >> 8) & 1;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.m_pad = m_pad; a.out = out; a.ldc = N;
    a.ksplit = ksplit ? ksplit : gemm_pick_ksplit(M, N, K);
    a.split_stride = split_stride;
    if (ksplit_used) *ksplit_used = a.ksplit;
    return launch_gemm(EPI_F32, a, (hipStream_t)stream);
}
extern "C" int grip_debug_gemm_ln(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid, void* out, void* out2,
                                  float* stat_part, const float* rowstat, const float* colsum, int m_pad, int variant, void* stream) {
    GemmArgs a{};
    a.variant = variant;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.m_pad = m_pad; a.bias = bias; a.resid = resid; a.out = out; a.out2 = out2; a.ldc = N;
    a.stat_part = stat_part; a.rowstat = rowstat; a.colsum = colsum;
    return launch_gemm(epi, a, (hipStream_t)stream);
}
// The two prompt-step forms of r04 (csrc/gemm.hip).  epi 7 / 8 with stat_in != NULL: the consumer adds the producer's stat_parts partial pairs itself
// (loader-wave kernels) or the launcher finalises them into `rowstat` (must be writable) first.  epi 3 with ksplit > 1: cooperative split-K, partial
// tiles through coop_scratch (>= tiles * ksplit * 32 KiB), tickets in coop_counter (tiles * 4 ints, zero on entry, zero again on exit).
extern "C" int grip_debug_gemm_train(int epi, const void* A, const void* W, int M, int N, int K, const float* bias, const void* resid, void* out, void* out2,
                                     float* stat_part, float* rowstat, const float* colsum, const float* stat_in, int stat_parts, int ksplit,
                                     float* coop_scratch, int* coop_counter, int m_pad, void* stream) {
    GemmArgs a{};
    a.rot_rows = 1;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.m_pad = m_pad; a.bias = bias; a.resid = resid; a.out = out; a.out2 = out2; a.ldc = N;
    a.stat_part = stat_part; a.rowstat = rowstat; a.colsum = colsum; a.stat_in = stat_in; a.stat_parts = stat_parts;
    a.ksplit = ksplit; a.coop_scratch = coop_scratch; a.coop_counter = coop_counter;
    return launch_gemm(epi, a, (hipStream_t)stream);
}
extern "C" int grip_debug_coop_split(int M, int N, int K) { return gemm_pick_coop_split(M, N, K); }
extern "C" int grip_debug_ln_fold(const void* W, const float* gamma, const float* beta, const float* bias, void* Wg, float* colsum, float* bias_out,
                                  int N, int K, const float* stat_part, int parts, float* rowstat, int M, int d, void* stream) {
    int rc = launch_ln_fold_weights((const half_t*)W, gamma, beta, bias, (half_t*)Wg, colsum, bias_out, N, K, (hipStream_t)stream);
    if (rc || !stat_part) return rc;
    return launch_ln_stats_finalize(stat_part, parts, rowstat, M, d, (hipStream_t)stream);
}
// Split-f16 GEMM (gemm_split.hip) on f32 inputs: A [m_pad, K] and W [N, K] are first rewritten in the split layout into the caller's scratch
// buffers a_split / w_split (4 bytes per element each), then out = epi(A W^T) -- f32, or the split layout for epi 2 (out: 4 bytes per element).
extern "C" int grip_debug_gemm_split(int epi, const float* A, const float* W, int M, int N, int K, const float* bias, const float* resid, void* out,
                                     void* a_split, void* w_split, int m_pad, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    RUN(launch_split_rows(A, a_split, m_pad, K, K, s));
    int* flag = nullptr;        // as grip_tower_finalize: is every element of W an f16 number?  (then the two-product kernel runs: GemmArgs::w_exact)
    GRIP_CHECK_HIP(hipMalloc((void**)&flag, sizeof(int)));
    int inexact = 1;
    int rc = hipMemsetAsync(flag, 0, sizeof(int), s) == hipSuccess ? launch_split_rows(W, w_split, N, K, K, s, 1, nullptr, flag) : GRIP_ERR_HIP;
    if (!rc && (hipMemcpyAsync(&inexact, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) rc = GRIP_ERR_HIP;
    (void)hipFree(flag);
    if (rc) return rc;
    GemmArgs a{};
    a.w_exact = !inexact;
    a.f32 = 2; a.A = a_split; a.W = w_split; a.M = M; a.N = N; a.K = K; a.m_pad = m_pad; a.bias = bias; a.resid = resid; a.out = out; a.ldc = N;
    return launch_gemm(epi, a, s);
}
int gemm_split_last_wlo();
extern "C" int grip_debug_split_last_wlo(void) { return gemm_split_last_wlo(); }
// f32 rows -> the split layout (out: 4 bytes per element), e.g. to read a split-layout result back on the host side of a test
extern "C" int grip_debug_split_rows(const float* x, void* out, int64_t rows, int K, void* stream) {
    return launch_split_rows(x, out, rows, K, K, (hipStream_t)stream);
}
// Attention of a precision-2 tower: qkv f32 -> out in the split layout (4 bytes per element); mfma != 0: attention_split.hip (S <= 320), else the
// f32 vector-ALU kernel with split output
extern "C" int grip_debug_attention_split(const void* qkv, void* out, int B, int S, int H, int causal, int mfma, void* stream) {
    if (mfma) return launch_attention_fwd_split((const float*)qkv, out, B, S, H, causal, (hipStream_t)stream);
    return launch_attention_fwd_f32((const float*)qkv, (float*)out, B, S, H, causal, (hipStream_t)stream, 1);
}
extern "C" int grip_debug_attention_exact(const void* qkv, void* out, int B, int S, int H, int causal, void* stream) {
    return launch_attention_fwd_f32((const float*)qkv, (float*)out, B, S, H, causal, (hipStream_t)stream);
}
extern "C" int grip_debug_attention(const void* qkv, void* out, int B, int S, int H, int causal, void* stream) {
    return launch_attention_fwd((const half_t*)qkv, (half_t*)out, B, S, H, causal, (hipStream_t)stream);
}
extern "C" int grip_debug_layernorm(const float* x, const float* gamma, const float* beta, void* out, int M, int d, void* stream) {
    return launch_layernorm_f16_from_f32(x, gamma, beta, (half_t*)out, M, d, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------- backward
// Input-gradient chain of one tower down to its prompt slice.  Enters with dx / dxh holding the
// (loss-scaled) gradient w.r.t. the final residual stream, leaves with dx = gradient w.r.t. x0.
static int run_blocks_backward(grip_tower* t, Workspace& w, int causal, const int32_t* read_rows, hipStream_t s) {
    const int d = t->D.width, H = t->D.heads;
    const half_t* W = t->w16;
    const float* F = t->w32;
    const int64_t part = (int64_t)w.Mp * d;       // floats between split-K partial buffers in w.dln
    for (int l = t->D.layers - 1; l >= 0; --l) {
        const LayerW& lw = t->L.layer[(size_t)l];
        if (l + 1 == t->D.layers && w.rows_last) {
            // the last block ran for the read rows only (run_blocks): enters with drow / drow_h = the gradient of those stream rows
            const int64_t Bp = round_up64(w.batch, 256);
            const int64_t rpart = Bp * d;
            GemmArgs a{};
            a.A = w.drow_h; a.W = W + lw.proj_wT; a.M = w.batch; a.m_pad = Bp; a.N = 4 * d; a.K = d; a.aux = w.row_hpre; a.out = w.row_dh; a.ldc = 4 * d;
            RUN(launch_gemm(EPI_GELUGRAD_F16, a, s));
            a = GemmArgs{};
            a.A = w.row_dh; a.W = W + lw.fc_wT; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = 4 * d; a.out = w.row_dln; a.ldc = d;
            a.ksplit = w.ks_row; a.split_stride = rpart;
            RUN(launch_gemm(EPI_F32, a, s));
            RUN(launch_ln_bwd_add(w.row_xmid, w.row_dln, w.ks_row, rpart, F + lw.ln2_g, w.drow, w.drow_h, w.batch, d, s));
            a = GemmArgs{};
            a.A = w.drow_h; a.W = W + lw.out_wT; a.M = w.batch; a.m_pad = Bp; a.N = d; a.K = d; a.out = w.row_datt; a.ldc = d;
            RUN(launch_gemm(EPI_F16, a, s));
            // d(q, k, v) of every row from the read rows' d(attention output): rank-one in (q, dO) per head
            RUN(launch_attention_row_bwd(w.qkv_l[(size_t)l], w.row_att, w.row_datt, read_rows, w.dqkv, w.batch, w.S, H, causal, s));
            a = GemmArgs{};
            a.rot_rows = 1;
            a.A = w.dqkv; a.W = W + lw.in_wT; a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = 3 * d; a.out = w.dln; a.ldc = d;
            a.ksplit = w.ks_in; a.split_stride = part;
            RUN(launch_gemm(EPI_F32, a, s));
            // the stream gradient entering the block: drow at the read rows, zero elsewhere (no fill, no scatter)
            RUN(launch_ln_bwd_init(w.x_in[(size_t)l], w.dln, w.ks_in, part, F + lw.ln1_g, w.drow, read_rows, w.S, w.dx, w.dxh, w.M, d, s));
            continue;
        }
        GemmArgs a{};
        a.rot_rows = 1;           // (backward GEMMs: train mode by definition)
        // d(pre-activation) = (dx @ W_proj) * quickgelu'(h_pre)
        a.A = w.dxh; a.W = W + lw.proj_wT; a.M = w.M; a.m_pad = w.Mp; a.N = 4 * d; a.K = d; a.aux = w.hpre_l[(size_t)l]; a.out = w.dh; a.ldc = 4 * d;
        RUN(launch_gemm(EPI_GELUGRAD_F16, a, s));
        a = GemmArgs{};
        a.rot_rows = 1;
        a.A = w.dh; a.W = W + lw.fc_wT; a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = 4 * d; a.out = w.dln; a.ldc = d;
        a.ksplit = w.ks_fc; a.split_stride = part;
        RUN(launch_gemm(EPI_F32, a, s));
        RUN(launch_ln_bwd_add(w.x_mid[(size_t)l], w.dln, w.ks_fc, part, F + lw.ln2_g, w.dx, w.dxh, w.M, d, s));
        a = GemmArgs{};
        a.rot_rows = 1;
        a.A = w.dxh; a.W = W + lw.out_wT; a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = d; a.out = w.datt; a.ldc = d;
        RUN(launch_gemm(EPI_F16, a, s));
        RUN(launch_attention_bwd(w.qkv_l[(size_t)l], w.att_l[(size_t)l], w.datt, w.dqkv, w.batch, w.S, H, causal, s, w.Ps, w.kv_part));
        a = GemmArgs{};
        a.rot_rows = 1;
        a.A = w.dqkv; a.W = W + lw.in_wT; a.M = w.M; a.m_pad = w.Mp; a.N = d; a.K = 3 * d; a.out = w.dln; a.ldc = d;
        a.ksplit = w.ks_in; a.split_stride = part;
        RUN(launch_gemm(EPI_F32, a, s));
        RUN(launch_ln_bwd_add(w.x_in[(size_t)l], w.dln, w.ks_in, part, F + lw.ln1_g, w.dx, w.dxh, w.M, d, s));
    }
    return GRIP_OK;
}

static int backward_head_of_tower(grip_tower* t, Workspace& w, const float* grad_emb, const int32_t* index, hipStream_t s) {
    const grip_dims& D = t->D;
    const int d = D.width;
    // scale + f16 cast of dL/d(emb); d(ln output) = g @ proj^T  (W = proj [d, E] as stored: N = d, K = E)
    RUN(launch_grad_scale_cast(grad_emb, w.gemb16, w.scale, w.batch * D.embed_dim, s));
    GemmArgs a{};
    a.A = w.gemb16; a.W = t->w16 + t->L.proj; a.M = w.batch; a.N = d; a.K = D.embed_dim; a.out = w.dcls; a.ldc = d;
    RUN(launch_gemm(EPI_F32, a, s));
    if (w.rows_last)      // compact: the final stream exists for the read rows only (run_blocks)
        return launch_ln_bwd_scatter(w.row_xout, w.dcls, nullptr, 1, t->w32 + t->L.lnpost_g, w.drow, w.drow_h, w.batch, d, s);
    // row of (sequence b, position index[b]): b * rs + index[b] in either layout (shared-prefix: Ps + b*(S-Ps) + index - Ps); every other row of the
    // stream gradient starts at zero (written by the same kernel)
    RUN(launch_ln_bwd_scatter_fill(w.x_in[(size_t)D.layers], w.dcls, index, w.rs, w.Ps, t->w32 + t->L.lnpost_g, w.dx, w.dxh, w.batch, (int)w.M, d, s));
    return GRIP_OK;
}

// Finds the pending train-mode forward of `workspace`.  generation != 0 must equal the number that forward handed out: a later
// train-mode forward on the same workspace has overwritten the activations, and the gradients would silently be those of
// the wrong forward.  The entry is marked consumed (one backward per forward) and stays until the workspace's next forward.
static int check_bwd(grip_tower* t, void* workspace, size_t workspace_bytes, uint64_t generation, grip_tower::TrainState& st) {
    GRIP_REQUIRE(t && workspace, "backward: null pointer");
    auto it = t->pending.find(workspace);
    if (it == t->pending.end() || !it->second.w.train) {
        grip_set_error("backward without a matching train-mode forward on this workspace");
        return GRIP_ERR_STATE;
    }
    if (generation != 0 && it->second.generation != generation) {
        grip_set_error("backward: the activations of forward #%llu were overwritten by forward #%llu on the same workspace "
                       "(two train-mode forwards before a backward need two workspaces)",
                       (unsigned long long)generation, (unsigned long long)it->second.generation);
        return GRIP_ERR_STATE;
    }
    if (it->second.consumed) {      // kept until the workspace's next forward so that this is not reported as "no matching forward"
        grip_set_error("backward: forward #%llu has already been back-propagated (one backward per forward: the backward works in place on the "
                       "saved activations, so retain_graph / a second backward needs a second forward)", (unsigned long long)it->second.generation);
        return GRIP_ERR_STATE;
    }
    if (it->second.w.bytes > workspace_bytes) { grip_set_error("backward: workspace too small"); return GRIP_ERR_WORKSPACE; }
    st = it->second;
    it->second.consumed = true;
    return GRIP_OK;
}

extern "C" int grip_vit_backward_prefix(grip_tower* t, const float* grad_emb, const float* prefix, float* grad_prefix,
                                        void* workspace, size_t workspace_bytes, uint64_t generation, void* stream) {
    try {
        GRIP_REQUIRE(t && t->D.kind == 0 && grad_emb && prefix && grad_prefix, "vit_backward_prefix: bad arguments");
        grip_tower::TrainState st;
        RUN(check_bwd(t, workspace, workspace_bytes, generation, st));
        Workspace& w = st.w;
        GRIP_REQUIRE(w.P > 0, "vit_backward_prefix: forward had no prompt tokens");
        hipStream_t s = (hipStream_t)stream;
        RUN(backward_head_of_tower(t, w, grad_emb, nullptr, s));
        RUN(run_blocks_backward(t, w, 0, nullptr, s));
        RUN(launch_vit_prefix_grad(w.dx, prefix, t->w32 + t->L.lnpre_g, w.scale, grad_prefix, w.batch, w.S, w.P, t->D.width, s));
        return GRIP_OK;
    } catch (...) { grip_set_error("vit_backward_prefix: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_text_backward_prefix(grip_tower* t, const float* grad_emb, float* grad_prefix,
                                         void* workspace, size_t workspace_bytes, uint64_t generation, void* stream) {
    try {
        GRIP_REQUIRE(t && t->D.kind == 1 && grad_emb && grad_prefix, "text_backward_prefix: bad arguments");
        grip_tower::TrainState st;
        RUN(check_bwd(t, workspace, workspace_bytes, generation, st));
        Workspace& w = st.w;
        GRIP_REQUIRE(w.P > 0, "text_backward_prefix: forward had no prompt tokens");
        hipStream_t s = (hipStream_t)stream;
        RUN(backward_head_of_tower(t, w, grad_emb, st.eot, s));
        RUN(run_blocks_backward(t, w, 1, st.eot, s));
        if (w.Ps)   // shared-prefix layout: every class's share already met in the shared rows (rows 1 .. P)
            RUN(launch_text_prefix_grad(w.dx, w.scale, grad_prefix, 1, w.S, w.P, 1, t->D.width, s));
        else
            RUN(launch_text_prefix_grad(w.dx, w.scale, grad_prefix, w.batch, w.S, w.P, st.prefix_classes, t->D.width, s));
        return GRIP_OK;
    } catch (...) { grip_set_error("text_backward_prefix: exception"); return GRIP_ERR_ARG; }
}

extern "C" int grip_debug_attention_bwd(const void* qkv, const void* o, const void* d_out, void* dqkv, int B, int S, int H, int causal, void* stream) {
    return launch_attention_bwd((const half_t*)qkv, (const half_t*)o, (const half_t*)d_out, (half_t*)dqkv, B, S, H, causal, (hipStream_t)stream);
}
