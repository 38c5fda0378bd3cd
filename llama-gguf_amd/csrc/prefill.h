// prefill.h — batched prompt processing on the f16 matrix cores (prefill.hip; internal).
#pragma once

#include "common.h"

namespace lgh {

constexpr int kPfTokens = 128;                    // tokens per pass = rows of an XH activation matrix
constexpr uint32_t kPfSlabBytes = kPfTokens * 512;   // one 256-element slab of XH: 128 tokens x 256 f16

constexpr int kPfSsqChunks = 8;                   // per-token partial sums of squares (one per 2048 columns): hidden <= 16384

inline size_t xh_bytes(uint32_t k) { return (size_t)(k / 256) * kPfSlabBytes; }

struct PfSeg {
  const uint8_t* w;      // tile16 weights
  uint32_t ntiles;       // 16-row tiles
  uint32_t col0;         // first column in the partial-sum rows
  uint32_t rg_begin;     // first row group (workgroup x index) of this matrix
  int fmt;
};

struct PfGemm {
  PfSeg seg[3];
  int nseg;
  uint32_t nblk;         // 256-element blocks of k
  uint32_t S;            // k-splits (gridDim.y)
  uint32_t ncols;        // floats per partial-sum row
  const uint8_t* xh;     // input activations [128][k] f16, XH layout
  float* part;           // [S][part_rows][ncols]
  uint32_t part_rows;    // rows per split: 128, or kPfMoeRows when the experts of a MoE layer share one buffer
  const int* row_base;   // optional (MoE): device word, first row of this expert's batch in the buffer
  uint32_t m_tiles;      // token tiles that hold real tokens
  const int* m_count;    // optional (MoE): device word with the number of real rows; overrides m_tiles, 0 rows = nothing to do
};

bool pf_supported_type(int dev_type);
size_t pf_part_bytes(const uint32_t* n_rows, int nw, uint32_t k, uint32_t part_rows = kPfTokens);
hipError_t pf_gemm_launch(const DevWeight* const* W, int nw, const uint8_t* xh, float* part, size_t part_bytes, uint32_t m_tokens,
                          uint32_t* S_out, uint32_t* ncols_out, hipStream_t st, uint32_t expert = 0, const int* m_count = nullptr,
                          uint32_t part_rows = kPfTokens, const int* row_base = nullptr);
hipError_t pf_row_epi_launch(const float* part, uint32_t S, uint32_t ncols, uint32_t col0, const float* bias, float* hidden, uint32_t H,
                             const float* nw, uint8_t* xh, float* ssq, uint32_t m_tokens, hipStream_t st, const float* moe_w = nullptr);
hipError_t pf_qkv_epi_launch(const float* part, uint32_t S, uint32_t ncols, uint32_t QD, uint32_t KD, uint32_t head_dim, const float* bq,
                             const float* bk, const float* bv, const float* rope_cs, uint32_t pos0, uint32_t max_seq, float* qbuf,
                             float* kcache, float* vcache, const float* ssq, uint32_t H, float eps, int neox, uint32_t m_tokens, hipStream_t st);
hipError_t pf_swiglu_launch(const float* part, uint32_t S, uint32_t F, uint8_t* xh, const float* ssq, uint32_t H, float eps, uint32_t m_tokens,
                            hipStream_t st, const int* row_tok = nullptr, const int* m_count = nullptr);
// ---- MoE layers (moe.rs:321-413): the block's (token, slot) pairs grouped by expert.  Expert e's rows are
// base[e] .. base[e] + counts[e] - 1 of ONE row space shared by all experts (bases padded to 16: at most kPfMoeRows rows),
// so that the SwiGLU of all experts and the final combine are one launch each.
//   lists[e][i]   token | slot << 8 of the expert's i-th row (token order)
//   rowmap[r]     e | i << 8 of row r, -1 for padding          tokmap[t * top_k + s]   row of (token, slot)
constexpr int kPfMaxExperts = 64;
constexpr int kPfMaxTopK = 8;
constexpr int kPfMoeRows = kPfTokens * 2 + 128;   // 2 experts per token is what the buffers are sized for (checked on the host)
hipError_t pf_moe_group_launch(const int* sel, uint32_t m_tokens, uint32_t top_k, uint32_t n_experts, int* counts, int* bases, int* lists,
                               int* rowmap, int* tokmap, hipStream_t st);
// rows of every expert's batch gathered in one launch: expert e's XH at xh_out + e * xh_bytes(K)
hipError_t pf_moe_gather_launch(const uint8_t* xh, uint32_t K, const int* lists, const int* counts, uint8_t* xh_out, uint32_t n_experts,
                                hipStream_t st);
// silu(gate) * up of every expert's rows in one launch: partial sums [S][kPfMoeRows][2F] -> expert e's XH at xh_out + e * xh_bytes(F)
hipError_t pf_moe_swiglu_launch(const float* part, uint32_t S, uint32_t F, uint8_t* xh_out, const int* rowmap, const int* lists,
                                const float* ssq, uint32_t H, float eps, hipStream_t st);
// h += sum_s w[t][s] * (down-projection row of (t, s)), next XH and sums of squares: pf_row_epi with the rows looked up in tokmap
hipError_t pf_moe_combine_launch(const float* part, uint32_t S, const int* tokmap, const float* moe_w, uint32_t top_k, float* hidden, uint32_t H,
                                 const float* nw, uint8_t* xh, float* ssq, uint32_t m_tokens, hipStream_t st);
hipError_t pf_to_xh_launch(const float* x, uint32_t K, uint8_t* xh, uint32_t m_tokens, hipStream_t st);
// dequant.hip: rows tokens[0..m) of the embedding table -> dst[m][hidden]
hipError_t embed_batch_launch(int src_type, const uint8_t* table, const int* tokens, float* dst, uint32_t hidden, uint32_t m_tokens,
                              hipStream_t st);
// attention.hip: causal attention of a block of m tokens at positions pos0 .. pos0+m-1 (their K/V rows already cached)
hipError_t attn_prefill_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv,
                               uint32_t head_dim, uint32_t max_seq, float scale, uint32_t pos0, uint32_t m_tokens, uint8_t* xh_out,
                               hipStream_t st);
// byte offset of element k of token t in an XH matrix
__host__ __device__ inline size_t xh_offset(uint32_t t, uint32_t k) {
  const uint32_t ch = k >> 3, j = k & 7, pj = (j == 1 || j == 5) ? j + 1 : (j == 2 || j == 6) ? j - 1 : j;
  return (size_t)(ch >> 5) * kPfSlabBytes + t * 512 + (((ch & 31) ^ (t & 15)) << 4) + pj * 2;
}

}  // namespace lgh
