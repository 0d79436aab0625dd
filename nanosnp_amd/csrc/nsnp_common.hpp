// nsnp_common.hpp -- shared definitions of the HIP side (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "nanosnp.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int   i32x4 __attribute__((ext_vector_type(4)));

#define NSNP_CDIV(a, b) (((a) + (b) - 1) / (b))

// ---- model constants of the PileupModel (PileupModel/config/ont_pileup.yaml:6-20) -----------
constexpr int PW = NSNP_PILEUP_WINDOW;     // 33 positions
constexpr int PC = NSNP_PILEUP_CHANNELS;   // 18 channels
constexpr int PH = 64;                     // LSTM hidden size
constexpr int PG = 4 * PH;                 // 256 gate rows
constexpr int PCENTER = 16;                // ForwardLayer slices position 16 (model.py:68)
constexpr int PSTEPS1 = PCENTER + 1;       // layer-1 steps each direction actually needs

// ---- packed-weight image layout ---------------------------------------------------------------
// One "A image" serves v_mfma_f32_16x16x4_f32 with the weight matrix as the A operand:
//   img[tile][j4][lane][e]  (f32)   = Wp(row = 16*tile + (lane & 15), k = 4*(4*j4 + e) + (lane >> 4))
// so one 16-byte read per lane feeds 4 consecutive K-steps of one 16-row tile.
// Gate rows are permuted so that a lane's 4 accumulator registers of tile i are the (i,f,g,o)
// pre-activations of hidden unit 4*i + (lane >> 4); see pack.cpp for the row/column maps.

struct PileupWeightsDev {
    // per direction d (0 fwd, 1 reverse)
    float* l0_whh[2];   // [16][4][64][4]  64 KB   recurrent, K = 64
    float* l0_wih[2];   // [16][1][64][4]  16 KB   input channels 0..15
    float* l0_wlast[2]; // [16][64]         4 KB   K-step 4: channels 16,17, bias, zero (register operand)
    float* l1_wih[2];   // [16][8][64][4] 128 KB   K = 128 in H0 storage order
    float* l1_bias[2];  // [16][64][4]      4 KB   b_ih + b_hh in accumulator layout, gate rows scaled like the weights
    float* l1_bias_raw[2];  // the same, unscaled (layer-1 projection kernel of the f16x3 two-kernel path)
    float* l1_whh[2];   // [16][4][64][4]  64 KB
    float* proj_w;      // [8][8][64][4]   64 KB   K = 128 in H1c storage order
    float* proj_b;      // [8][64][4]
    float* dense_w;     // [16][8][64][4] 128 KB   K = 128 in proj accumulator order
    float* dense_b;     // [16][64][4]
    float* head_w;      // [2][16][64][4]  32 KB   rows 0..20 genotype, 21..23 zygosity, rest 0
    float* head_b;      // [2][64][4]
    float* arena;       // one allocation holding all of the above
    size_t arena_bytes;
    bool   loaded;
};

// fp16 hi/lo images of the same weights for the f16x3 path (pileup_forward_f16x3.hip)
struct PileupWeightsF16 {
    void* l0_whh[2]; void* l0_wih_hi[2]; void* l0_wih_lo[2]; void* l1_wih[2]; void* l1_whh[2];
    void* l1f_hi[2]; void* l1f_lo[2]; float* l1f_bias;      // fused projection + recurrence kernel
    void* l0_wih_rs[2];
    void* l1_wih_rs[2]; void* l1_whh_rs[2];                  // register-stationary layer-1 kernel
    void* l0_whh_rs[2];                                      // register-stationary layer-0 kernel (K order of its exchange rows)
    void* proj_w; void* dense_w; void* head_w;
    void* arena; size_t arena_bytes; bool loaded;
};

// bf16 three-plane images of the same weights for the bf16x3 path (pileup_forward_bf16x3.hip): [tile][K block][plane][lane] of 8 bf16
struct PileupWeightsB3 {
    void* l0_whh[2]; void* l0_wih[2]; void* l1_wih[2]; void* l1_whh[2];
    void* proj_w; void* dense_w; void* head_w;
    float* proj_b; float* dense_b; float* head_b;             // fp32, natural row order
    void* arena; size_t arena_bytes; bool loaded;
};

struct HapWeightsDev;   // hap_forward.hip
struct CatWeightsDev;   // cat_forward.hip

// optional per-kernel timing with HIP events on the launch stream (nsnp_ctx_enable_timing)
// NSNP_K_HAPLSTM / NSNP_K_CATCONV / NSNP_K_CAT bracket a whole chain of launches of one pass with ONE event pair (the fused
// LSTM step launches of a HaplotypeModel pass; the conv GEMM + pool launches of a CatModel pass; a whole CatModel pass)
enum { NSNP_K_L0 = 0, NSNP_K_PROJ1, NSNP_K_L1, NSNP_K_HEAD, NSNP_K_ENCODE, NSNP_K_HAPFEAT, NSNP_K_HAPLSTM, NSNP_K_CATCONV, NSNP_K_CAT,
       NSNP_K_COUNT };
struct KernelTimer {
    std::vector<hipEvent_t> start[NSNP_K_COUNT], stop[NSNP_K_COUNT];
    size_t used[NSNP_K_COUNT];
    bool enabled;
};

// argmax / max outputs of a fused forward + post-processing call (nsnp_pileup_forward_windows_calls); all null otherwise
struct PostOut { uint8_t* gt_arg; uint8_t* zy_arg; float* gt_max; float* zy_max; };

struct nsnp_ctx {
    int device;
    int n_cu;
    hipError_t last_err;
    bool attr_set;
    bool attr_set_f16;
    bool attr_set_b3;
    int hap_b3x;        // bf16x3 LSTM steps of the tile GEMM: 1 = 256 x 256 workgroup tiles where they fill the chip (default), 0 = always 128 x 128
    int cat_conv_pix2;  // k_cat_conv: 1 = blocks of <= 64 output channels on 256-pixel workgroups (default), 0 = 128 pixels everywhere
    int cat_conv_lds;   // legacy CatModel convolutions: 1 = the LDS-staged kernel k_cat_conv (default; all three arithmetics), 0 = the gathering CONV mode of k_hap_gemm
    int cat_precision;  // legacy CatModel forward: 0 = exact fp32 MFMA (default), 1 = f16x3 split (opt-in)
    int hap_precision;  // HaplotypeModel forward: 0 = exact fp32 MFMA (default), 1 = f16x3 split (opt-in)
    int precision;      // PileupModel forward: 0 = exact fp32 MFMA (default), 1 = f16x3 split (opt-in), 2 = bf16x3 (three bf16 terms per operand, six bf16 MFMAs per product)
    int fused_waves;    // 0 = automatic, else 4 / 8 / 12 waves per workgroup of the fused kernel
    int l0_rs;          // f16x3: 1 = register-stationary layer-0 kernel (default), 0 = LDS-image kernel
    int l1_rs_groups;   // 0 = automatic, else 2 / 4 groups of 16 sites per workgroup of that kernel
    int l1_rs;          // f16x3 fused layer 1: 1 = register-stationary kernel, 0 = LDS-image / ring kernel
    int l0_rs_groups;   // 0 = automatic, else 1 / 2 / 4 groups of 16 sites per workgroup
    int fused_l1;       // f16x3: 1 = fused projection + layer-1 recurrence kernel (default), 0 = two kernels
    int proj1_tiles;    // 16-row tiles per wave of the layer-1 projection kernel (persistent grid sizing)
    int force_wpb;      // 0 = automatic; else waves per recurrence workgroup (tuning / tests)
    int l0_wx_lds;      // fp32 layer 0 (16-site workgroups): input-part weight fragments in LDS, 128 VGPRs, four workgroups per SIMD set
    int rs_prio;        // f16x3 register-stationary kernels: bit 0 = static s_setprio 1 for waves 4-7 of the layer-1 kernel, bit 1 = for odd layer-0 workgroups
    int head_rs;        // fp32 heads: 1 = output tiles split over the 8 waves of a workgroup, weights in VGPRs (default); 0 = one wave per 16 sites
    int l1_stagger;     // fp32 register-stationary layer 1: waves 4-7 run a group's next-step input part ahead of its cell (default 0)
    // workspace (sized by nsnp_ctx_reserve)
    int64_t chunk_sites;
    float*  ws_h0;      // [chunk][33][128] (fp32 / f16x3) or [chunk][33][192] (bf16x3: three bf16 planes)
    int     ws_h0_floats;   // floats per site and step ws_h0 is allocated for (128 or 192)
    float*  ws_xp1;     // [2][chunk*17][256]
    float*  ws_h1c;     // [chunk][128]
    PileupWeightsDev pw;
    PileupWeightsF16 pw16;
    PileupWeightsB3 pwb3;
    HapWeightsDev* hw;
    void*  hap_ws; size_t hap_ws_bytes; int64_t hap_ws_chunk;     // (the pass size the workspace was made for)
    int64_t hap_chunk;  // sites per pass of the HaplotypeModel forward (option "hap_pass_sites", default 16384)
    CatWeightsDev* cw;
    void*  cat_ws; size_t cat_ws_bytes;
    int64_t* sel_tmp; size_t sel_tmp_bytes;   // select_sites scratch
    void* tok_ws; size_t tok_ws_bytes;        // mpileup tokeniser scratch: 20 bytes per 8 KB tile of text (mpileup_tokenise.hip)
    int tok_fused;      // mpileup tokeniser: 0 = three launches (default), 1 = one launch, chained scan (opt-in: its tiles spin on their predecessors)
    // column encode: AF threshold + smallest-passing-count table of the last min_af (pileup_encode.hip)
    bool af_cached; uint64_t af_bits, af_t; int af_k, af_mode; uint32_t af_table_words[128];
    bool af2_cached; uint64_t af2_bits, af2_t; int af2_k, af2_mode; uint32_t af2_table_words[128];     // the indel threshold when it differs
    KernelTimer* timer;
    void* comm; int comm_rank, comm_world;    // optional RCCL communicator of nsnp_comm_init / nsnp_comm_attach (nsnp_comm.hip)
    bool comm_borrowed;                        // attached by the caller: never destroyed here
};

// records an event pair around one kernel launch when timing is enabled
struct ScopedKernelTimer {
    nsnp_ctx* ctx; int k; hipStream_t s; hipEvent_t stop_ev; bool on;
    ScopedKernelTimer(nsnp_ctx* c, int kernel, hipStream_t stream);
    void stop();                      // records the stop event now (the destructor does nothing afterwards)
    ~ScopedKernelTimer();
};

#define NSNP_HIP(ctx, call)                                                        \
    do {                                                                           \
        hipError_t e_ = (call);                                                    \
        if (e_ != hipSuccess) { (ctx)->last_err = e_; return NSNP_EHIP; }          \
    } while (0)

// host-side packers (pack.cpp compiled as part of the .hip translation unit set)
typedef float (*nsnp_wfun)(const void* user, int row, int k);
void nsnp_pack_image(float* img, int n_tiles, int n_j4, nsnp_wfun f, const void* user);

int nsnp_ctx_need_xp1(nsnp_ctx* ctx);   // legacy unfused layer-1 path only (synchronous allocation on first use)

// kernels' launchers (pileup_forward.hip, pileup_forward_f16x3.hip, pileup_forward_bf16x3.hip).  `post` (may be null): argmax / max
// outputs of a fused forward + post-processing call; a launcher whose heads kernel writes them returns with *post_written = true,
// otherwise the caller runs nsnp_pileup_postprocess behind it.
int nsnp_pileup_forward_impl(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx,
                             int64_t N, float* gt, float* zy, const PostOut* post, bool* post_written, hipStream_t s);
int nsnp_pileup_pack_weights(nsnp_ctx* ctx, const float* const* w);
int nsnp_pileup_forward_f16x3(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx,
                              int64_t N, float* gt, float* zy, hipStream_t s);
int nsnp_pileup_pack_weights_f16(nsnp_ctx* ctx, const float* const* w);
int nsnp_pileup_forward_bf16x3(nsnp_ctx* ctx, const int32_t* x, const int64_t* center_idx, int64_t N, float* gt, float* zy,
                               const PostOut* post, hipStream_t s);
int nsnp_pileup_pack_weights_bf16(nsnp_ctx* ctx, const float* const* w);
