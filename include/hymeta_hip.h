/*
 * hymeta_hip.h -- C ABI of libhymeta_hip.so: the Hymeta hybrid-attention hot
 * path as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference (FAndromedA/vllm-hymeta) has no FFI of its own: its boundary
 * for this path is a set of Python callables that wrap Triton / flash-attn /
 * vLLM device code.  Each entry point below replaces one of those callables
 * and cites it (path:line under the reference tree).  The Python package
 * vllm_hymeta binds these symbols with ctypes and keeps the reference's
 * function names and argument meaning above them.
 *
 * Conventions
 *  - Every pointer is a DEVICE pointer owned by the caller; nothing is
 *    allocated, freed or retained by the library.
 *  - Strides are in ELEMENTS of the tensor's dtype; the innermost (head_dim)
 *    dimension must be contiguous.
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream).  All
 *    work is enqueued on it and nothing synchronises with the host, so every
 *    call may be captured into a HIP graph.
 *  - Return value: HYMETA_OK (0) or a negative error code; no exception ever
 *    crosses the boundary.  hymeta_strerror() names a code.
 *  - Activations, KV cache and (by default) the recurrent state are bf16;
 *    accumulation is fp32.
 *
 * Versioning
 *  - hymeta_abi_version() returns the HYMETA_ABI_VERSION the library was built
 *    with; a binding compares it with the header it was written against before
 *    the first call (version 2: every argument struct starts with struct_size;
 *    version 3: hymeta_gla_decode_args ends with qkv_x_scale / qkv_w_scale).
 *  - Every argument struct starts with `uint32_t struct_size`, which the caller
 *    sets to sizeof(the struct) of ITS declaration.  An entry point returns
 *    HYMETA_ERR_INVALID_ARG, before it reads any other field, when the value
 *    is not the size the library was built with: a binding that is a field
 *    short (or long) gets an error code, never a kernel that reads past the
 *    end of its struct.
 */
#ifndef HYMETA_HIP_H
#define HYMETA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HYMETA_ABI_VERSION 3

typedef void* hymeta_stream_t;

enum hymeta_status {
  HYMETA_OK = 0,
  HYMETA_ERR_INVALID_ARG = -1,  /* null pointer, negative size, bad stride */
  HYMETA_ERR_UNSUPPORTED = -2,  /* shape / dtype / mode the kernels do not cover */
  HYMETA_ERR_WORKSPACE = -3,    /* workspace missing or too small */
  HYMETA_ERR_LAUNCH = -4,       /* hipGetLastError() != hipSuccess after launch */
  HYMETA_ERR_SCRATCH_LIMIT = -5 /* an allocation-free entry point was asked for more than its fixed scratch holds */
};

enum hymeta_dtype { HYMETA_BF16 = 0, HYMETA_F32 = 1, HYMETA_F16 = 2, HYMETA_I8 = 3 };

/* Fused gate prologue applied to the raw projections before the recurrence. */
enum hymeta_gate_mode {
  /* q, k, g are used as given (the reference kernel-level contract). */
  HYMETA_GATE_NONE = 0,
  /* Hymeta-7B: q = relu(q), k = relu(k), g = logsigmoid(g) / gate_normalizer
   * (vllm_hymeta/model_for_7B/gla_attention.py:251-262). */
  HYMETA_GATE_7B = 1,
  /* Hymeta-70B "first" layers: q = relu(q), k~ = sigmoid(float(k)),
   * k = min(k~, gate_clamp), g = log(1 - k), k -> bf16; the g input is ignored
   * (vllm_hymeta/models/modeling_hymeta.py:611-619). */
  HYMETA_GATE_70B_CLAMP = 2,
  /* Hymeta-70B other layers: k~ = sigmoid(float(k)), gl = 1 - (1 - lb) * k~,
   * k = 1 - gl, g = log(gl), k -> bf16, with lb = lower_bound[head, :]
   * (vllm_hymeta/models/modeling_hymeta.py:620-629). */
  HYMETA_GATE_70B_LOWER_BOUND = 3
};

const char* hymeta_strerror(int code);
int hymeta_abi_version(void);
/* Host helper with no counterpart in the reference: *id = the identifier of the graph capture `stream` is
 * recording into (unique per capture), 0 when the stream is not capturing.  The Python front end uses it to
 * give every captured graph its own clearing of the split-K ticket counters (see hymeta_gptq_gemm). */
int hymeta_stream_capture_id(hymeta_stream_t stream, uint64_t* id);

/* ------------------------------------------------------------------------
 * GLA single-token decode step with slot-indexed in-place state.
 * Replaces my_fused_recurrent_fwd_kernel + MyFusedRecurrentFunction.forward
 * (vllm_hymeta/models/my_fused_recurrent.py:8-119; 7B variant
 * vllm_hymeta/model_for_7B/my_fused_recurrent.py:9-119), reached through
 * my_fused_recurrent_gla (vllm_hymeta/models/my_fused_recurrent_gla.py:8-23).
 *
 * Per (sequence b, q-head h), kv-head j = h / (num_heads / num_kv_heads):
 *   S  = state[slot_idx[b], h]                       (Dk x Dv, row-major)
 *   S' = bf16(k_j * v_j^T) + exp(g_j)[:,None] * S    (fp32)
 *   out[b,h,:] = sum_dk bf16(bf16(scale) * q[b,h,dk]) * S'[dk,:]   (fp32 sum)
 *   state[slot_idx[b], h] = S'  (rounded to state_dtype)
 * The two bf16 roundings of the query side are the reference kernel's: its
 * `load(q) * scale` (my_fused_recurrent.py:59) is a tensor-times-scalar product,
 * which Triton computes in the tensor's dtype with the scalar cast to it.
 * slot_idx[b] == -1 skips sequence b (state and out untouched).
 * Optional fused prologue (gate_mode) and epilogue (per-head RMS norm over Dv
 * with weight out_norm_weight[Dv] and out_norm_eps, the g_norm of
 * vllm_hymeta/model_for_7B/gla_attention.py:233).
 *
 * One state per key/value head (state_heads == num_kv_heads).  The reference
 * expands k, v and g to the query heads (repeat_kv,
 * vllm_hymeta/model_for_7B/gla_attention.py:246-282) and keeps one state per
 * query head; where the gate does not depend on the query head (every gate mode
 * but HYMETA_GATE_70B_LOWER_BOUND) the states of the query heads of one group
 * receive the same update at every token and are equal for ever.  With
 * state_heads == num_kv_heads the tensor is [slots, num_kv_heads, Dk, Dv], the
 * group's state is read and written once per step and contracted with every
 * query of the group; out is bit-identical to the per-query-head layout.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_gla_decode_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* q;            /* bf16 [batch, num_heads,    head_dim] */
  const void* k;            /* bf16 [batch, num_kv_heads, head_dim] */
  const void* v;            /* bf16 [batch, num_kv_heads, head_dim] */
  const void* g;            /* g_dtype [batch, num_kv_heads, head_dim]; may be NULL for HYMETA_GATE_70B_* */
  void* out;                /* bf16 [batch, num_heads, head_dim] */
  void* state;              /* state_dtype [slots, state_heads, head_dim, head_dim] */
  const int32_t* slot_idx;  /* [batch] */
  const float* lower_bound;     /* fp32 [num_heads, head_dim], HYMETA_GATE_70B_LOWER_BOUND only */
  const void* out_norm_weight;  /* bf16 [head_dim] or NULL (no epilogue norm) */
  int32_t batch;
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;         /* 128 */
  int64_t q_stride_b, q_stride_h;
  int64_t k_stride_b, k_stride_h;
  int64_t v_stride_b, v_stride_h;
  int64_t g_stride_b, g_stride_h;
  int64_t out_stride_b, out_stride_h;
  int64_t state_stride_slot, state_stride_h;  /* the (Dk, Dv) block itself is contiguous */
  int32_t g_dtype;          /* enum hymeta_dtype */
  int32_t state_dtype;      /* enum hymeta_dtype */
  int32_t gate_mode;        /* enum hymeta_gate_mode */
  float scale;              /* <= 0 selects head_dim^-0.5 */
  float gate_normalizer;    /* HYMETA_GATE_7B, 16 in the reference */
  float gate_clamp;         /* HYMETA_GATE_70B_CLAMP, 0.95 in the reference */
  float out_norm_eps;
  int32_t state_heads;      /* heads the state tensor holds: 0 or num_heads = one state per query head (the
                             * reference's layout); num_kv_heads = one state per key/value head, see above */
  /* The gate in its low-rank form (7B: gk_proj = Linear(hidden, 16) -> Linear(16, Hkv * D, bias),
   * vllm_hymeta/model_for_7B/gla_attention.py:157-166): with g_w1 given, g is the FIRST linear's output, bf16
   * [batch, 16] rows g_stride_b apart (g_stride_h unused), and the kernel applies the second linear itself:
   * g_w1 bf16 [num_kv_heads * head_dim, 16] contiguous, g_bias bf16 [num_kv_heads * head_dim] or NULL, result rounded
   * to bf16 as the linear's output is.  NULL g_w1: g is the finished gate, as above. */
  const void* g_w1;
  const void* g_bias;
  /* q, k, v (and the low-rank g) taken straight from the partial sums of the fused qkv projection that was cut into K
   * slices (see hymeta_gemm_partials below: fp32 slabs [qkv_splits][batch][qkv_n], bias bf16 [qkv_n] or NULL): with
   * qkv_slabs given the q / k / v / g pointers and strides are ignored, head h of q is columns q_col + h * head_dim ...,
   * kv-head j of k / v columns k_col / v_col + j * head_dim ..., the 16 rank columns start at g_col (g_w1 required for
   * the 7B gate); every value is the slices' sum in slice order plus bias, rounded to bf16 -- what the projection's
   * reduce launch would have written. */
  const void* qkv_slabs;
  const void* qkv_bias;
  int32_t qkv_splits, qkv_n;
  int32_t q_col, k_col, v_col, g_col;
  /* Both linears of the low-rank gate inside the kernel (a checkpoint whose qkv projection is quantised cannot carry
   * the gate's first linear as extra rows): with g_w0 bf16 [16, g_hidden] contiguous given (g_w1 required), g is the
   * LAYER INPUT, bf16 [batch, g_hidden] rows g_stride_b apart -- also beside qkv_slabs, whose g_col is then unused.
   * State per key/value head only. */
  const void* g_w0;
  int32_t g_hidden;
  /* qkv_slabs of a W8A8 projection: int32 slabs, finished as hymeta_w8a8_gemm finishes them (x_scale fp32 [batch],
   * w_scale fp32 [qkv_n], 16-byte aligned; both or neither).  State per key/value head only. */
  const float* qkv_x_scale;
  const float* qkv_w_scale;
} hymeta_gla_decode_args;

int hymeta_gla_decode(const hymeta_gla_decode_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * GLA chunked prefill over ALL prefill sequences of a step in one launch.
 * Replaces the per-sequence Python loop around fla.ops.gla.fused_chunk_gla
 * (vllm_hymeta/models/modeling_hymeta.py:470-529 for 70B;
 * vllm_hymeta/model_for_7B/gla_attention.py:179-221 for 7B), including the
 * state read (:488), the first-chunk choice of initial state (:501-504), the
 * write-back of the final state (:528) and, through context_lens, the slot
 * zeroing of vllm_hymeta/models/modeling_hymeta.py:1202-1227.
 *
 * Sequence i owns tokens cu_seqlens[i] .. cu_seqlens[i+1]-1 and state slot
 * slot_idx[i] (-1 skips it).  Its initial state is
 *   context_lens == NULL or context_lens[i] != 0 : state[slot]
 *   context_lens[i] == 0, meta_state != NULL     : meta_state[h]   (70B)
 *   context_lens[i] == 0, meta_state == NULL     : zeros           (7B)
 * Tokens: S_t = diag(exp g_t) S_{t-1} + k_t^T v_t ; o_t = (scale q_t) S_t.
 * The final state is stored to state[slot] rounded to state_dtype.
 *
 * When few (sequence, head) pairs carry long sequences the call can cut each
 * sequence into num_segments runs that execute in parallel (three launches:
 * per-segment state contributions, a scan over them, the outputs); this needs
 * a device workspace of hymeta_gla_prefill_workspace_bytes().  Results equal
 * the single-run form up to fp32 summation order of the state.
 *   num_segments == 1                      : single run, workspace unused
 *   num_segments == 0, workspace != NULL   : hymeta_gla_prefill_segments() decides
 *   num_segments == 0, workspace == NULL   : single run
 *   num_segments  > 1                      : as given (clamped to the chunk count)
 *
 * With one state per key/value head (state_heads == num_kv_heads < num_heads,
 * at most 8 query heads per group) a workgroup walks one (sequence, kv-head)
 * and serves every query head of the group: the key side of a chunk (gate,
 * decay scan, state update) is computed once instead of once per query head.
 * Segment counts and the workspace are then sized over state_heads.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_gla_prefill_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* q;            /* bf16 [total_tokens, num_heads,    head_dim] */
  const void* k;            /* bf16 [total_tokens, num_kv_heads, head_dim] */
  const void* v;            /* bf16 [total_tokens, num_kv_heads, head_dim] */
  const void* g;            /* g_dtype [total_tokens, num_kv_heads, head_dim] */
  void* out;                /* bf16 [total_tokens, num_heads, head_dim] */
  void* state;              /* state_dtype [slots, state_heads, head_dim, head_dim] */
  const int32_t* slot_idx;      /* [num_seqs] */
  const int32_t* cu_seqlens;    /* [num_seqs + 1] */
  const int32_t* context_lens;  /* [num_seqs] or NULL */
  const void* meta_state;       /* state_dtype [state_heads, head_dim, head_dim] or NULL */
  const float* lower_bound;     /* as in hymeta_gla_decode_args */
  const void* out_norm_weight;  /* as in hymeta_gla_decode_args */
  int32_t num_seqs;
  int32_t total_tokens;
  int32_t max_seqlen;       /* upper bound on any sequence length in this call */
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;         /* 128 */
  int64_t q_stride_t, q_stride_h;
  int64_t k_stride_t, k_stride_h;
  int64_t v_stride_t, v_stride_h;
  int64_t g_stride_t, g_stride_h;
  int64_t out_stride_t, out_stride_h;
  int64_t state_stride_slot, state_stride_h;
  int64_t meta_state_stride_h;
  int32_t g_dtype;
  int32_t state_dtype;
  int32_t gate_mode;
  float scale;
  float gate_normalizer;
  float gate_clamp;
  float out_norm_eps;
  void* workspace;          /* device scratch, 16-byte aligned, or NULL */
  size_t workspace_bytes;
  int32_t num_segments;     /* see above */
  int32_t state_heads;      /* as in hymeta_gla_decode_args; state and meta_state hold that many heads */
} hymeta_gla_prefill_args;

/* Segment count the library would choose for this shape (1 = single run); num_heads = heads that run as
 * separate workgroups: the query heads, or state_heads when a group shares its state. */
int hymeta_gla_prefill_segments(int32_t num_seqs, int32_t num_heads, int32_t max_seqlen);
/* Workspace size for a call with that many segments and that state layout (state_heads 0 = num_heads);
 * 0 when num_segments <= 1. */
size_t hymeta_gla_prefill_workspace_bytes(int32_t num_seqs, int32_t num_heads, int32_t state_heads, int32_t num_segments);
int hymeta_gla_prefill_varlen(const hymeta_gla_prefill_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * Paged KV write.  Replaces torch.ops._C_cache_ops.reshape_and_cache_flash as
 * called at vllm_hymeta/attention/meta_attention.py:790-799 (kv_cache_dtype
 * "auto": a byte copy).  Token t goes to
 *   cache[slot / block_size, slot % block_size, :, :],  slot = slot_mapping[t];
 * a negative slot (PAD_SLOT_ID) writes nothing.  Bit-exact.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_reshape_and_cache_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* key;          /* [num_tokens, num_kv_heads, head_dim] */
  const void* value;
  void* key_cache;          /* [num_blocks, block_size, num_kv_heads, head_dim] */
  void* value_cache;
  const int64_t* slot_mapping;  /* [num_tokens] */
  int32_t num_tokens;
  int32_t num_kv_heads;
  int32_t head_dim;
  int32_t block_size;
  int32_t elem_bytes;       /* 2 for bf16 / fp16 */
  int64_t num_blocks;       /* slots >= num_blocks * block_size are skipped as out of range */
  int64_t key_stride_t, key_stride_h;
  int64_t value_stride_t, value_stride_h;
  int64_t cache_stride_block, cache_stride_pos, cache_stride_h;  /* same for both caches */
} hymeta_reshape_and_cache_args;

int hymeta_reshape_and_cache(const hymeta_reshape_and_cache_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * Single-query paged decode attention: sliding window over the paged real
 * keys plus dense attention over the always-visible meta keys, one softmax.
 * Replaces FlashAttentionWithMetaToken.forward
 * (vllm_hymeta/attention/fa_kernel_metatoken.py:124-194), i.e. the sequence
 * flash_attn fwd_kvcache (:157-168) + _flash_attn_forward over the meta keys
 * (:174-188) + _update_out_and_lse (:110-122), as dispatched from
 * vllm_hymeta/attention/meta_attention.py:954-981.  With num_meta == 0 it is
 * the stock paged sliding-window decode of the 7B SWA layer
 * (vllm_hymeta/model_for_7B/modeling_gla_swa.py:193-218).
 *
 * Sequence b has cache_seqlens[b] keys, the decoded token's own key included
 * (written beforehand by hymeta_reshape_and_cache); key position p lives at
 * cache[block_tables[b, p / block_size], p % block_size].  The query sits at
 * position L-1 and sees positions max(0, L-1-window_left) .. L-1
 * (window_left < 0: all), plus every meta key.
 *
 * Split-KV: the key range of a sequence is cut into at most `num_splits`
 * parts handled by different workgroups; parts are merged by a second kernel
 * through `workspace`.  hymeta_meta_swa_paged_decode_workspace_bytes() gives
 * the size needed for (batch, num_heads, head_dim, num_splits).
 * num_splits <= 0 lets the library choose from batch, num_kv_heads and
 * max_seqlen_k; pass the same value to the size query.
 *
 * Fused KV write (decode-only steps): with new_key != NULL the call also does
 * the work of the hymeta_reshape_and_cache that would precede it
 * (meta_attention.py:790-799): row b of new_key / new_value is stored at cache
 * slot slot_mapping[b] (negative: skipped), which must be the slot of position
 * cache_seqlens[b] - 1, before sequence b is attended.  One launch instead of
 * two per layer and step.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_paged_decode_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* q;            /* bf16 [batch, num_heads, head_dim] */
  void* out;                /* bf16 [batch, num_heads, head_dim] */
  const void* key_cache;    /* bf16 [num_blocks, block_size, num_kv_heads, head_dim] */
  const void* value_cache;
  const int32_t* block_tables;   /* [batch, max_blocks_per_seq] */
  const int32_t* cache_seqlens;  /* [batch] */
  const void* meta_key;     /* bf16 [num_meta, num_kv_heads, head_dim] or NULL */
  const void* meta_value;
  void* workspace;
  size_t workspace_bytes;
  int32_t batch;
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;         /* 128 */
  int32_t block_size;       /* multiple of 16 */
  int32_t max_blocks_per_seq;
  int32_t num_meta;         /* multiple of 32 (128 in the reference) or 0 */
  int32_t window_left;      /* flash-attn left window; < 0 = unbounded */
  int32_t max_seqlen_k;     /* upper bound on cache_seqlens; <= 0: max_blocks_per_seq * block_size */
  int32_t num_splits;
  int64_t q_stride_b, q_stride_h;
  int64_t out_stride_b, out_stride_h;
  int64_t cache_stride_block, cache_stride_pos, cache_stride_h;
  int64_t block_table_stride;
  int64_t meta_stride_t, meta_stride_h;
  float scale;              /* <= 0 selects head_dim^-0.5 */
  const void* new_key;      /* bf16 [batch, num_kv_heads, head_dim] or NULL */
  const void* new_value;
  const int64_t* slot_mapping;   /* [batch], required with new_key */
  int64_t new_key_stride_b, new_key_stride_h;
  int64_t new_value_stride_b, new_value_stride_h;
} hymeta_paged_decode_args;

size_t hymeta_meta_swa_paged_decode_workspace_bytes(int32_t batch, int32_t num_heads,
                                                    int32_t head_dim, int32_t num_splits);
int hymeta_meta_swa_paged_decode_splits(int32_t batch, int32_t num_kv_heads,
                                        int32_t max_seqlen_k, int32_t window_left,
                                        int32_t num_meta);
int hymeta_meta_swa_paged_decode(const hymeta_paged_decode_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * The same paged decode for callers that cannot allocate or carry a workspace: code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.
 * args->workspace / workspace_bytes are ignored; the split-KV partials live in a file-scope device
 * array of HYMETA_STATIC_SCRATCH_BYTES inside the library (one copy per GPU the library runs on, held
 * until the process ends).  A call whose partials need more -- batch * num_heads * num_splits *
 * (head_dim + 2) * 4 bytes with more than one split -- returns HYMETA_ERR_SCRATCH_LIMIT and launches
 * nothing; unsplit calls need no scratch and always fit.  Everything else (arguments, kernels, split
 * choice, results bit for bit) is hymeta_meta_swa_paged_decode.  The entry point calls no hipMalloc /
 * hipFree or other allocating routine; its first call on a device looks the array's address up
 * (hipGetSymbolAddress), so make that one call outside stream capture.  All calls on one GPU share
 * the array: calls that overlap on one GPU (two streams) race; order them on one stream.  Nothing in
 * the array has to start at zero and nothing is carried from call to call.
 * ---------------------------------------------------------------------- */
#define HYMETA_STATIC_SCRATCH_BYTES (16u << 20)
int hymeta_meta_swa_paged_decode_static(const hymeta_paged_decode_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * Varlen prefill attention: causal sliding window over the real keys plus
 * dense attention over the meta keys, one softmax; optional meta-query rows.
 * Replaces FlashAttentionVarlenWithMetaToken.forward
 * (vllm_hymeta/attention/fa_kernel_metatoken.py:198-285): the sequence
 * _flash_attn_varlen_forward (:235-253) + _flash_attn_forward over the meta
 * keys (:261-275) + _update_out_and_lse (:277) + the concat that puts the
 * meta-query rows first (:283); dispatched from
 * vllm_hymeta/attention/meta_attention.py:866-884 (no cache) and :900-919 /
 * :933-953 (through the block table).  Semantics as the dense-mask
 * naive_metatoken_flash_attn (:377-410).
 *
 * Sequence i: queries cu_seqlens_q[i] .. cu_seqlens_q[i+1]-1; keys either
 *  - un-paged: rows cu_seqlens_k[i] .. cu_seqlens_k[i+1]-1 of key/value, or
 *  - paged (block_tables != NULL): seq_lens_k[i] positions read through
 *    block_tables[i] from key_cache/value_cache (chunk already written).
 * Query r of Lq sits at position Lk - Lq + r and sees positions
 * [pos - window_left, pos] (window_left < 0: [0, pos]) and every meta key.
 * Output rows: if meta_query != NULL, rows 0..num_meta-1 are the meta
 * queries (they see the meta keys only), followed by the real rows.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_swa_prefill_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* q;            /* bf16 [total_q, num_heads, head_dim] */
  const void* key;          /* bf16 [total_k, num_kv_heads, head_dim] (un-paged) or NULL */
  const void* value;
  const void* key_cache;    /* paged form, as in hymeta_paged_decode_args, or NULL */
  const void* value_cache;
  const int32_t* block_tables;  /* [num_seqs, max_blocks_per_seq] or NULL */
  const int32_t* cu_seqlens_q;  /* [num_seqs + 1] */
  const int32_t* cu_seqlens_k;  /* [num_seqs + 1], un-paged form */
  const int32_t* seq_lens_k;    /* [num_seqs], paged form */
  const void* meta_query;   /* bf16 [num_meta, num_heads, head_dim] or NULL */
  const void* meta_key;     /* bf16 [num_meta, num_kv_heads, head_dim] or NULL */
  const void* meta_value;
  void* out;                /* bf16 [(num_meta if meta_query) + total_q, num_heads, head_dim] */
  int32_t num_seqs;
  int32_t total_q;
  int32_t max_seqlen_q;     /* upper bound on any query length in this call */
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;         /* 128 */
  int32_t num_meta;         /* multiple of 32 or 0 */
  int32_t window_left;
  int32_t block_size;
  int32_t max_blocks_per_seq;
  int64_t q_stride_t, q_stride_h;
  int64_t k_stride_t, k_stride_h;
  int64_t v_stride_t, v_stride_h;
  int64_t out_stride_t, out_stride_h;
  int64_t cache_stride_block, cache_stride_pos, cache_stride_h;
  int64_t block_table_stride;
  int64_t meta_q_stride_t, meta_q_stride_h;
  int64_t meta_stride_t, meta_stride_h;
  float scale;
  /* Key slices (a short chunk of a long sequence leaves most compute units without a workgroup:
   * the visible keys of a query block are then cut into num_splits slices whose partial results a
   * second launch merges).  num_splits <= 0: the library chooses -- 1 unless `workspace` holds
   * hymeta_meta_swa_prefill_workspace_bytes(out rows, num_heads, 0) bytes; 1: never; 2..8: as
   * given (HYMETA_ERR_WORKSPACE without the room).  max_seqlen_k: upper bound on any key length,
   * 0 = unknown (then layers without a window are never sliced). */
  void* workspace;
  size_t workspace_bytes;
  int32_t num_splits;
  int32_t max_seqlen_k;
} hymeta_swa_prefill_args;

/* The library's choice of key slices; meta_rows = num_meta when meta_query is given, else 0. */
int hymeta_meta_swa_prefill_splits(int32_t total_q, int32_t num_heads, int32_t num_meta, int32_t meta_rows,
                                   int32_t window_left, int32_t max_seqlen_k);
/* out_rows = total_q (+ num_meta with meta_query); num_splits <= 0: room for any choice. */
size_t hymeta_meta_swa_prefill_workspace_bytes(int32_t out_rows, int32_t num_heads, int32_t num_splits);
int hymeta_meta_swa_prefill(const hymeta_swa_prefill_args* args, hymeta_stream_t stream);


/* ------------------------------------------------------------------------
 * NeoX (rotate-half) rotary embedding, in place on q and k.
 * Replaces vllm._custom_ops.rotary_embedding as called from
 * HymetaRotaryEmbedding.forward (vllm_hymeta/models/modeling_hymeta.py:220-234)
 * and the stock get_rope of the 7B SWA layer
 * (vllm_hymeta/model_for_7B/modeling_gla_swa.py:89-90,204-218).
 * For token t at position p = positions[t] + position_offset and pair
 * (i, i + rot_dim/2):  x' = x cos - y sin ;  y' = y cos + x sin, with
 * cos = cache[p, i], sin = cache[p, rot_dim/2 + i]; every product and the
 * sum are rounded to bf16, as the bf16 instantiation of the vLLM op does.
 * position_offset carries the +num_meta_tokens shift of the 70B model
 * (vllm_hymeta/models/modeling_hymeta.py:1314-1323).
 * ---------------------------------------------------------------------- */
typedef struct hymeta_rope_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  void* q;                   /* bf16 [num_tokens, num_heads,    head_dim] */
  void* k;                   /* bf16 [num_tokens, num_kv_heads, head_dim] or NULL */
  const int64_t* positions;  /* [num_tokens] */
  const void* cos_sin_cache; /* bf16 [max_position, rot_dim] */
  int32_t num_tokens;
  int32_t num_heads;
  int32_t num_kv_heads;
  int32_t head_dim;
  int32_t rot_dim;           /* == head_dim for the Hymeta models */
  int32_t max_position;      /* rows of the cache; positions beyond it are an error of the caller */
  int64_t position_offset;
  int64_t q_stride_t, q_stride_h;
  int64_t k_stride_t, k_stride_h;
} hymeta_rope_args;

int hymeta_rope_neox(const hymeta_rope_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * 70B branch combine: y = bf16( (RMSNorm1(o_attn) + RMSNorm2(o_lin)) / 2 ).
 * Replaces HymetaRMSNormTP._forward x2 + the average
 * (vllm_hymeta/models/modeling_hymeta.py:151-173, 909-919); eager form
 * hf_model/modeling_hymeta.py:127-128.  RMS statistics are over the FULL
 * hidden size; under tensor parallelism each rank holds hidden/tp columns.
 *
 *  hymeta_rms_sumsq2      : stats[t, 0|1] = mean over the LOCAL columns of
 *                           o_attn[t]^2 | o_lin[t]^2  (fp32).  The caller
 *                           all-reduces(sum) stats and divides by tp
 *                           (:158-160) -- one [T,2] message instead of the
 *                           reference's two [T,1] ones.
 *  hymeta_hybrid_combine  : applies both norms with the given statistics
 *                           (stats == NULL: computes them locally, tp = 1)
 *                           and averages.  Per element, as the reference:
 *                           xn = bf16(x * rsqrt(var + eps)); y = bf16((xn1*w1 + xn2*w2) / 2).
 * ---------------------------------------------------------------------- */
typedef struct hymeta_combine_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* o_attn;    /* bf16 [num_tokens, hidden_local] */
  const void* o_lin;     /* bf16 [num_tokens, hidden_local] */
  const float* weight1;  /* fp32 [hidden_local] (norm1, softmax branch) */
  const float* weight2;  /* fp32 [hidden_local] (norm2, linear branch) */
  float* stats;          /* fp32 [num_tokens, 2]: out of sumsq2, in (nullable) of combine */
  void* out;             /* bf16 [num_tokens, hidden_local] */
  int32_t num_tokens;
  int32_t hidden_local;  /* multiple of 8 */
  int64_t attn_stride_t, lin_stride_t, out_stride_t;
  float eps;
} hymeta_combine_args;

int hymeta_rms_sumsq2(const hymeta_combine_args* args, hymeta_stream_t stream);
int hymeta_hybrid_combine(const hymeta_combine_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * Block-level RMS norm with the residual add folded in.
 * Replaces vLLM's RMSNorm as the decoder blocks call it:
 *   attn_norm(x)                 vllm_hymeta/models/modeling_hymeta.py:1024,
 *                                model_for_7B/modeling_gla_swa.py:294
 *   mlp_norm(x, residual)        modeling_hymeta.py:1039, modeling_gla_swa.py:302
 *   final norm                   modeling_hymeta.py:1191, modeling_gla_swa.py:497
 * Arithmetic of the device op those calls dispatch to (vllm==0.10.0
 * rms_norm / fused_add_rms_norm, third party):
 *   residual != NULL: z = bf16(x + residual); residual <- z   (in place)
 *   residual == NULL: z = x
 *   out = bf16( bf16(z * rsqrt(mean(z^2) + eps)) * weight ),  mean and rsqrt in fp32.
 * out may alias x.  hidden is a multiple of 8.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_add_rmsnorm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;         /* bf16 [num_tokens, hidden] */
  void* residual;        /* bf16 [num_tokens, hidden], updated in place, or NULL */
  const void* weight;    /* bf16 [hidden] */
  void* out;             /* bf16 [num_tokens, hidden] */
  int32_t num_tokens;
  int32_t hidden;
  int64_t x_stride_t, residual_stride_t, out_stride_t;
  float eps;
} hymeta_add_rmsnorm_args;

int hymeta_add_rmsnorm(const hymeta_add_rmsnorm_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * The rank-16 gate projection of the 7B GLA layer in one launch.
 * Replaces the two un-quantised linears of gk_proj
 * (vllm_hymeta/model_for_7B/gla_attention.py:157-176: hidden -> 16 without bias,
 * 16 -> num_kv_heads * head_dim with bias; hf_7B_model/gla_attention.py:108-112)
 * for decode-sized batches, where each of the two library GEMMs is a launch of its own:
 *   t   = bf16( x w0^T )            [num_tokens, 16], fp32 sums
 *   out = bf16( t w1^T + bias )     [num_tokens, out_features], fp32 sums
 * rank must be 16, hidden a multiple of 8.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_lowrank_gate_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;         /* bf16 [num_tokens, hidden] */
  const void* w0;        /* bf16 [rank, hidden], contiguous */
  const void* w1;        /* bf16 [out_features, rank], contiguous */
  const void* bias;      /* bf16 [out_features] or NULL */
  void* out;             /* bf16 [num_tokens, out_features] */
  int32_t num_tokens;
  int32_t hidden;
  int32_t rank;          /* 16 */
  int32_t out_features;
  int64_t x_stride_t, out_stride_t;
} hymeta_lowrank_gate_args;

int hymeta_lowrank_gate(const hymeta_lowrank_gate_args* args, hymeta_stream_t stream);


/* ------------------------------------------------------------------------
 * Fused dequant-GEMM for GPTQ checkpoints: out = x W (+ bias), W dequantised on
 * the fly from the gptqmodel / AutoGPTQ tensors.  Replaces the vLLM quantised
 * linear method behind every *ParallelLinear(quant_config=...) of the Hymeta
 * layers (vllm_hymeta/models/modeling_hymeta.py:249-262,422-430,724-732,858-861;
 * vllm_hymeta/model_for_7B/gla_attention.py:147-176, modeling_gla_swa.py:108-121,
 * 177-192) for checkpoints produced by hf_7B_model/quant_7B.py:32-36 (8-bit,
 * group 64) and the module lists of hf_7B_model/gla_swa_gptq.py:18-24 /
 * hf_model/hymeta_gptq.py:22-35.  The arithmetic itself lives in vLLM 0.10.0
 * (third-party): parity unpinned.
 *
 * Tensor format (GPTQ v1): pack = 32 / bits;
 *   qweight int32 [K / pack, N]   : row r holds k = r*pack .. r*pack+pack-1 of each column,
 *                                   value j in bits [bits*j, bits*(j+1));
 *   qzeros  int32 [K / group, N / pack] : zero of column n in bits [bits*(n%pack), ...) of entry n/pack,
 *                                   stored minus zero_offset (1 in the v1 format);
 *   scales  [K / group, N]  fp16 or bf16;
 *   w[k, n] = scales[k/group, n] * (q[k, n] - (qzeros[k/group, n] + zero_offset)).
 * Rows are in natural order (g_idx[k] = k / group; act-order checkpoints are not supported).
 * Split-K: partial tiles meet in `workspace` (ticket counters + fp32 slabs); the last workgroup
 * to reach an output tile sums its slabs and writes it.  split_k <= 0 lets the library choose
 * the decomposition (K slices for few output tiles; for more 64 x 128 tiles than compute units,
 * whole tiles round by round with the remainder cut in K); its workspace,
 * hymeta_gptq_gemm_workspace_bytes(M, N, 0), is a bound that holds for every K.  Without a
 * workspace the library runs unsplit.  split_k > 1 forces that many slices and needs
 * hymeta_gptq_gemm_workspace_bytes(M, N, split_k) bytes.
 * Up to 64 rows with split_k <= 0, 4-bit weights of group size 128 or 64 (K % 256 == 0) and 8-bit weights of group size
 * 64 (K % 128 == 0) with N % 128 == 0 run on a weight-streaming kernel (one workgroup per compute unit, x shared through
 * LDS, q kept exact in bf16, scale and zero applied to each group's fp32 partial sums); the widest matrices and other
 * formats on the tiled kernels with the same arithmetic.
 * From 129 to 256 rows (8-bit: from 65; group size 64 or 128, split_k <= 0) the 129..256-row streaming kernel runs instead,
 * on the shapes it cuts into at least 128 workgroups: it reads the packed weights ONCE for all rows and expands each
 * word to bf16(scale * (q - zero)) -- the values hymeta_gptq_dequant writes -- on its way into LDS, so its result
 * equals hymeta_bf16_gemm on the expanded matrix bit for bit (the kernels below 129 rows keep q exact and apply scale
 * and zero to fp32 partial sums: results of the two regimes agree to bf16 rounding of the weights, not bit for bit).
 *
 * Workspace contract of the three split-K GEMMs (hymeta_gptq_gemm, hymeta_bf16_gemm,
 * hymeta_w8a8_gemm): the first HYMETA_GEMM_TICKET_BYTES bytes are the ticket counters.  Zero them
 * ONCE when the workspace is allocated (hipMemset); every call leaves them zero, so no call spends
 * a memset on them.  The rest is scratch with no content carried between calls.  One workspace can
 * serve all three entry points, one call at a time (calls that share a workspace must be ordered
 * on one stream).
 * ---------------------------------------------------------------------- */
#define HYMETA_GEMM_TICKET_BYTES 16384
typedef struct hymeta_gptq_gemm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;          /* bf16 [M, K] */
  const int32_t* qweight;
  const int32_t* qzeros;
  const void* scales;
  const void* bias;       /* bf16 [N] or NULL */
  void* out;              /* bf16 [M, N] */
  float* workspace;       /* hymeta_gptq_gemm_workspace_bytes() bytes, or NULL (then unsplit, or an error if split_k > 1) */
  size_t workspace_bytes;
  int32_t M, K, N;
  int32_t bits;           /* 4 or 8 */
  int32_t group_size;     /* 32, 64 or 128 */
  int32_t scales_dtype;   /* HYMETA_F16 or HYMETA_BF16 */
  int32_t zero_offset;    /* 1 for the v1 format, 0 for v2 */
  int32_t split_k;        /* <= 0: library chooses (workspace sized with split_k = 0) */
  int64_t x_stride_m, out_stride_m;
} hymeta_gptq_gemm_args;

int hymeta_gptq_gemm_split_k(int32_t M, int32_t K, int32_t N, int32_t group_size);
size_t hymeta_gptq_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);
int hymeta_gptq_gemm(const hymeta_gptq_gemm_args* args, hymeta_stream_t stream);

/* Whole-matrix form of the same dequantisation: out bf16 [K, N] = w[k, n] above, rounded to
 * bf16.  For prefill-sized M the host side expands the weights once per call and runs a plain
 * library bf16 GEMM on them (the choice vLLM's GPTQ linear method makes above its own row
 * threshold for the layers cited above). */
int hymeta_gptq_dequant(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out,
                        int32_t K, int32_t N, int32_t bits, int32_t group_size, int32_t scales_dtype,
                        int32_t zero_offset, hymeta_stream_t stream);

/* The same expansion written transposed: out bf16 [N, K] = w[k, n], the layout of a torch Linear weight (so that
 * y = F.linear(x, out)), which the library bf16 GEMM runs faster at prefill-sized M than the [K, N] one.  Needs
 * N % 64 == 0 and K % (32 * (32 / bits)) == 0 (K % 256 for 4-bit, K % 128 for 8-bit); anything else is
 * HYMETA_ERR_UNSUPPORTED and the caller uses hymeta_gptq_dequant. */
int hymeta_gptq_dequant_nk(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out,
                           int32_t K, int32_t N, int32_t bits, int32_t group_size, int32_t scales_dtype,
                           int32_t zero_offset, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * bf16 linear for decode-sized batches (M <= 128; 65 to 128 rows need K % 64 == 0, K >= 256 and split_k == 0;
 * 129 to 256 rows: the shapes for which hymeta_bf16_gemm_takes says so, with a workspace and split_k == 0):
 * out = bf16(x W^T + bias), fp32 accumulation,
 * one rounding; W [N, K] as torch stores a Linear weight.  Replaces the un-quantised linear
 * method (F.linear through vLLM's ColumnParallelLinear / RowParallelLinear, third party) at the
 * call sites vllm_hymeta/models/modeling_hymeta.py:249-262,422-430,724-732,858-861 and
 * vllm_hymeta/model_for_7B/gla_attention.py:118-160 when the weights are not quantised: at these
 * row counts the weights are the traffic and a library GEMM reads them at 2-3 TB/s.
 * Needs K % 32 == 0, N % 64 == 0; larger M is HYMETA_ERR_UNSUPPORTED (use a library GEMM).
 * Split-K meets in fp32 slabs summed in slice order (deterministic): split_k <= 0 lets the
 * library choose when a workspace is given (size it with hymeta_bf16_gemm_workspace_bytes(M, N, 0),
 * a bound that does not depend on K) and means 1 without one.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_bf16_gemm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;          /* bf16 [M, K] */
  const void* weight;     /* bf16 [N, K] */
  const void* bias;       /* bf16 [N] or NULL */
  void* out;              /* bf16 [M, N] */
  void* workspace;        /* [ticket counters, see hymeta_gptq_gemm][fp32 slabs], or NULL */
  size_t workspace_bytes;
  int32_t M, K, N;
  int32_t split_k;
  int64_t x_stride_m, w_stride_n, out_stride_m;
} hymeta_bf16_gemm_args;

int hymeta_bf16_gemm_split_k(int32_t M, int32_t K, int32_t N);
size_t hymeta_bf16_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);
int hymeta_bf16_gemm(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream);
/* 1 when the entry point named by `form` -- 0: hymeta_bf16_gemm (split_k 0, workspace of
 * hymeta_bf16_gemm_workspace_bytes(M, N, 0)); 1: hymeta_bf16_gemm_silu_mul (N counts the gate and the up rows);
 * 2: hymeta_bf16_gemm_f32out -- takes the shape, 0 when it would answer
 * HYMETA_ERR_UNSUPPORTED and the caller runs a library GEMM: every covered shape up to 128 rows; from 129 to 256
 * rows what the 129..256-row streaming kernel cuts into at least 128 workgroups (tiles of 112 to 160 columns; K
 * slices of at least 512 bytes for narrow matrices).  No counterpart in the reference (its linears are vLLM's):
 * a question of the host-side dispatch. */
int hymeta_bf16_gemm_takes(int32_t M, int32_t K, int32_t N, int32_t form);

/* Logits of a bf16 lm_head: out fp32 [M, N] (out_stride_m in fp32 elements, a multiple of 4) = x weight^T (+ bias)
 * with fp32 accumulation and no rounding of the result; workspace and split_k unused.  M <= 128 (to 256 where
 * hymeta_bf16_gemm_takes(M, K, N, 2) says so: wide matrices), K a multiple of 64 and at least 256, N a multiple of 64;
 * other shapes HYMETA_ERR_UNSUPPORTED.
 * Replaces lm_head.float() applied to hidden_states.float() in compute_logits
 * (vllm_hymeta/model_for_7B/modeling_gla_swa.py:548,579-583, vllm_hymeta/models/modeling_hymeta.py:1429,1462-1466):
 * a checkpoint's bf16 lm_head widened to fp32 holds the same values, so the product of the bf16 operands accumulated
 * in fp32 is that fp32 GEMM up to the order of the sums, at half the weight bytes. */
int hymeta_bf16_gemm_f32out(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream);

/* Greedy sampling of a decode step: out[m] = index of the largest of logits[m, 0 .. N) (fp32 rows stride_m elements
 * apart; the lowest index among equals), in launches that a captured decode step can hold.  N and stride_m
 * multiples of 4.  workspace (hymeta_argmax_rows_workspace_bytes(M) bytes, or NULL) lets a wide row be scanned by 32
 * workgroups with a second small launch picking among them; without it one workgroup scans a row.
 * Stands in for the argmax of vLLM's sampler on greedy requests (the reference hands its logits to that sampler:
 * vllm_hymeta/model_for_7B/modeling_gla_swa.py:579-590). */
size_t hymeta_argmax_rows_workspace_bytes(int32_t M);
int hymeta_argmax_rows(const float* logits, int64_t stride_m, int32_t M, int32_t N, int64_t* out, void* workspace,
                       size_t workspace_bytes, hymeta_stream_t stream);

/* Multi-step decoding: sequence i < num_queries moves on by one token in place -- input_tokens[i] =
 * sampled_token_ids[i] (either may be NULL), input_positions[i] = seq_lens[i] (may be NULL), seq_lens[i] += 1,
 * slot_mapping[i] = block_tables[i][pos / block_size] * block_size + pos % block_size; the graph padding
 * num_queries <= i < num_seqs gets slot -1.  int64 tokens, positions and slots, int32 lengths and block tables
 * (rows block_table_stride entries apart).  One launch for the device update of
 * MetaAttentionMetadata.advance_step (vllm_hymeta/attention/meta_attention.py:338-409, which calls vLLM's
 * advance_step_flashattn, third party). */
int hymeta_advance_step(int32_t num_seqs, int32_t num_queries, int32_t block_size, int64_t* input_tokens,
                        const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens, int64_t* slot_mapping,
                        const int32_t* block_tables, int64_t block_table_stride, hymeta_stream_t stream);

/* The gate/up projection of a gated MLP with its activation in the GEMM's epilogue, for decode-sized batches:
 * weight = [gate rows | up rows] (N rows in all, the fused gate_up_proj), out bf16 [M, N / 2] =
 * silu(x gate^T) * (x up^T), both factors rounded to bf16 before the product -- the arithmetic of
 * hymeta_bf16_gemm followed by hymeta_silu_and_mul (the same bits whenever that call sums K in one slice),
 * in one launch and without the [M, N] intermediate.
 * Replaces MergedColumnParallelLinear + SiluAndMul in the MLP forward
 * (vllm_hymeta/model_for_7B/modeling_gla_swa.py:108-121, vllm_hymeta/models/modeling_hymeta.py:262-271).
 * bias must be NULL, the workspace fields are ignored.  Takes M <= 128, K % 64 == 0, N % 128 == 0, and from 129 to
 * 256 rows the shapes for which hymeta_bf16_gemm_takes(M, K, N, 1) says so (matrices wide enough for 128 whole tiles);
 * other shapes are HYMETA_ERR_UNSUPPORTED: run the two calls. */
int hymeta_bf16_gemm_silu_mul(const hymeta_bf16_gemm_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * W8A8 int8 linear: per-output-channel int8 weights, dynamic per-token int8
 * activations (compressed-tensors "W8A8" as written by hf_model/quant_w8a8.py:121-132
 * with llmcompressor; executed in the reference by vLLM's scaled_int8_quant +
 * scaled_mm -- third-party arithmetic, parity unpinned):
 *   s_x[m] = max|x[m,:]| / 127 ; xq = clamp(rint(x / s_x), -128, 127)
 *   out[m, n] = bf16( (sum_k xq[m,k] wq[n,k]) * s_x[m] * w_scale[n] + bias[n] )
 * xq / x_scale are caller-owned scratch ([M, K] int8, [M] fp32).
 * Split-K as in hymeta_gptq_gemm (exact int32 slabs): split_k <= 0 lets the
 * library choose when a workspace is given, and means 1 without one.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_w8a8_gemm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;           /* bf16 [M, K]; NULL: xq / x_scale below already hold the quantised rows (inputs) */
  const int8_t* qweight;   /* int8 [N, K] */
  const float* w_scale;    /* fp32 [N] */
  const void* bias;        /* bf16 [N] or NULL */
  void* out;               /* bf16 [M, N] */
  int8_t* xq;              /* scratch int8 [M, K] */
  float* x_scale;          /* scratch fp32 [M] */
  int32_t M, K, N;
  int64_t x_stride_m, out_stride_m;
  void* workspace;         /* hymeta_w8a8_gemm_workspace_bytes() bytes, or NULL */
  size_t workspace_bytes;
  int32_t split_k;
} hymeta_w8a8_gemm_args;

int hymeta_w8a8_gemm_split_k(int32_t M, int32_t K, int32_t N);
size_t hymeta_w8a8_gemm_workspace_bytes(int32_t M, int32_t N, int32_t split_k);

int hymeta_w8a8_gemm(const hymeta_w8a8_gemm_args* args, hymeta_stream_t stream);

/* W8A8 form of hymeta_bf16_gemm_silu_mul (same layout, same rule: equals hymeta_w8a8_gemm followed by
 * hymeta_silu_and_mul bit for bit).  bias must be NULL; K % 128 == 0, N % 128 == 0; any M (up to 256 rows a
 * streaming kernel where it covers the shape, the tiled kernel otherwise and above). */
int hymeta_w8a8_gemm_silu_mul(const hymeta_w8a8_gemm_args* args, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * K-sliced linears whose slice sum is left to the consumer.
 * A decode-sized linear with few output columns (o_proj / out_proj / down_proj) is cut along K so that every
 * compute unit streams weights; its slices' partial sums then have to be added.  The *_deferred calls stop
 * before that sum and describe the partial sums; hymeta_add_rmsnorm_partials -- the fused residual add + RMS
 * norm that follows those projections in the decoder block (model_for_7B/modeling_gla_swa.py:294-302,
 * models/modeling_hymeta.py:1024-1039) -- adds them as it reads its input, which saves the reduce launch and
 * one pass over the output.  Values are those of the plain calls: slices added in slice order, bias (and the
 * W8A8 scales) applied, the sum rounded to bf16 before the residual add.
 * The partial sums live in the GEMM call's workspace: consume them before the next call that uses it.
 * ---------------------------------------------------------------------- */
typedef struct hymeta_gemm_partials {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* slabs;     /* [splits][M][N] fp32 (bf16 linear) or int32 (W8A8) */
  int32_t splits;        /* 0: the call wrote `out` itself (no K slices): nothing is deferred */
  int32_t is_int32;
  const float* x_scale;  /* W8A8: fp32 [M] activation scales (the args' x_scale buffer) */
  const float* w_scale;  /* W8A8: fp32 [N] */
  const void* bias;      /* bf16 [N] or NULL */
  int32_t M, N;
} hymeta_gemm_partials;

int hymeta_bf16_gemm_deferred(const hymeta_bf16_gemm_args* args, hymeta_gemm_partials* partials, hymeta_stream_t stream);
int hymeta_w8a8_gemm_deferred(const hymeta_w8a8_gemm_args* args, hymeta_gemm_partials* partials, hymeta_stream_t stream);
int hymeta_gptq_gemm_deferred(const hymeta_gptq_gemm_args* args, hymeta_gemm_partials* partials, hymeta_stream_t stream);
/* out bf16 [M, N] = the linear's result (what the plain call would have written). */
int hymeta_gemm_partials_reduce(const hymeta_gemm_partials* partials, void* out, int64_t out_stride_m, hymeta_stream_t stream);
/* hymeta_add_rmsnorm with x = the linear's result taken from its partial sums (args->x is ignored; args->residual
 * must be given; num_tokens == M, hidden == N). */
int hymeta_add_rmsnorm_partials(const hymeta_add_rmsnorm_args* args, const hymeta_gemm_partials* partials, hymeta_stream_t stream);
/* hymeta_add_rmsnorm (partials NULL or splits 0) or hymeta_add_rmsnorm_partials that ALSO writes the dynamic per-token
 * int8 quantisation of its bf16 output -- xq int8 [num_tokens, hidden] contiguous, x_scale fp32 [num_tokens], the bits
 * hymeta_w8a8_quantize gives on that output -- for the W8A8 linear that consumes the norm (qkv_proj, gate_up_proj):
 * called with x == NULL and these two buffers, hymeta_w8a8_gemm / _silu_mul / _deferred skip their quantise launch. */
/* hymeta_rope_neox with the fused qkv projection taken from its partial sums: writes the projection's whole bf16 row
 * (args->q = the [num_tokens, N] output buffer, q heads first, then k heads, head_dim apart; args->k NULL or q + H * D;
 * v and any further columns follow and are written unrotated) -- the linear's reduce launch and the RoPE launch
 * (model_for_7B/modeling_gla_swa.py:196-197) in one.  num_kv_heads = the k heads to rotate. */
int hymeta_rope_neox_partials(const hymeta_rope_args* args, const hymeta_gemm_partials* partials, hymeta_stream_t stream);
int hymeta_add_rmsnorm_int8(const hymeta_add_rmsnorm_args* args, const hymeta_gemm_partials* partials, int8_t* xq,
                            float* x_scale, hymeta_stream_t stream);


/* The two ends of the same linear for prefill-sized M, where the host side puts a library int8
 * GEMM (exact int32 sums) between them instead of the fused kernel, the choice vLLM's scaled_mm
 * makes for the layers cited above:
 *   hymeta_w8a8_quantize : x bf16 [M, K] -> xq int8 [M, K], x_scale fp32 [M]   (same arithmetic as above)
 *   hymeta_w8a8_scale_out: acc int32 [M, N] -> out bf16 = acc * x_scale[m] * w_scale[n] + bias[n] */
int hymeta_w8a8_quantize(const void* x, int8_t* xq, float* x_scale, int32_t M, int32_t K, int64_t x_stride_m,
                         hymeta_stream_t stream);
int hymeta_w8a8_scale_out(const int32_t* acc, const float* x_scale, const float* w_scale, const void* bias, void* out,
                          int32_t M, int32_t N, int64_t acc_stride_m, int64_t out_stride_m, hymeta_stream_t stream);

/* ------------------------------------------------------------------------
 * Sparse mixture-of-experts block (70B, 16 experts, top-1, shared expert on the
 * side): replaces HymetaMoE.forward -> vLLM FusedMoE(renormalize=False,
 * reduce_results=True) (vllm_hymeta/models/modeling_hymeta.py:273-364) and the
 * shared-expert sum of HybridBlock.forward (:1044-1068).  Eager oracle:
 * hf_model/moe.py:34-101 (which renormalises the top-k weights: `renormalize`).
 * The expert arithmetic lives in vLLM 0.10.0 (third-party): parity unpinned.
 *
 * Call order per layer (all stream-ordered, no host read of the expert counts):
 *   hymeta_moe_route     logits [T, E] -> topk_ids / topk_weights [T, k], the (token, choice)
 *                        pairs counting-sorted by expert: expert_offsets [E + 1], sorted_pair
 *                        [T k] (pair index per sorted slot), pair_pos [T k] (slot of a pair, -1
 *                        when its expert is outside [local_expert_begin, local_expert_end):
 *                        expert parallelism)
 *   hymeta_moe_gemm      h13[slot] = x[sorted_pair[slot] / k] W13[e]^T      (a_rows = sorted_pair)
 *   hymeta_silu_and_mul  h[slot] = silu(h13[slot, :I]) * h13[slot, I:]      (also SiluAndMul of the dense MLP, :263-270)
 *   hymeta_moe_gemm      y[slot] = bf16(w[pair] * (h[slot] W2[e]^T))        (row_scale = topk_weights)
 *   hymeta_moe_combine   out[t] = sum_k y[pair_pos[t, k]] (+ shared[t])
 * followed, under tensor / expert parallelism, by the caller's all-reduce.
 * Weight stacks are K-contiguous: W13 [E, 2 I, H], W2 [E, H, I] (vLLM's w13 / w2).
 * ---------------------------------------------------------------------- */
typedef struct hymeta_moe_route_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* router_logits;   /* logits_dtype [num_tokens, num_experts] */
  int32_t* topk_ids;           /* out [num_tokens, top_k] */
  float* topk_weights;         /* out [num_tokens, top_k] */
  int32_t* counts;             /* scratch [num_experts] */
  int32_t* cursors;            /* scratch [num_experts] */
  int32_t* expert_offsets;     /* out [num_experts + 1] */
  int32_t* sorted_pair;        /* out [num_tokens * top_k] */
  int32_t* pair_pos;           /* out [num_tokens * top_k] */
  int32_t num_tokens;
  int32_t num_experts;         /* <= 64 */
  int32_t top_k;               /* 1 or 2 */
  int32_t renormalize;         /* 0: softmax probabilities as they are (the vLLM path) */
  int32_t logits_dtype;        /* HYMETA_BF16 or HYMETA_F32 */
  int32_t local_expert_begin, local_expert_end;
  int64_t logits_stride_t;
} hymeta_moe_route_args;

int hymeta_moe_route(const hymeta_moe_route_args* args, hymeta_stream_t stream);

/* Decode-sized batches (num_tokens <= HYMETA_MOE_ROUTER_MAX_TOKENS): the router's linear layer and
 * hymeta_moe_route in one launch -- logits = x gate_weight^T (fp32 sums of the bf16 products, rounded to bf16 as the
 * reference's bf16 ReplicatedLinear gate rounds them: vllm_hymeta/models/modeling_hymeta.py:312-324), then the same
 * softmax / top-k / offsets / scatter.  ticket: one device word, zero before the first call; every call leaves it zero
 * (calls that share a ticket must not overlap).  logits_out: optional bf16 [num_tokens, num_experts], the logits that
 * were routed.  hidden and x_stride_t multiples of 8; larger batches: HYMETA_ERR_UNSUPPORTED (run the linear layer and
 * hymeta_moe_route). */
#define HYMETA_MOE_ROUTER_MAX_TOKENS 256
typedef struct hymeta_moe_router_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;               /* bf16 [num_tokens, hidden] */
  const void* gate_weight;     /* bf16 [num_experts, hidden] contiguous */
  void* logits_out;            /* bf16 [num_tokens, num_experts] or NULL */
  int32_t* topk_ids;           /* out [num_tokens, top_k] */
  float* topk_weights;         /* out [num_tokens, top_k] */
  int32_t* expert_offsets;     /* out [num_experts + 1] */
  int32_t* sorted_pair;        /* out [num_tokens * top_k] */
  int32_t* pair_pos;           /* out [num_tokens * top_k] */
  uint32_t* ticket;
  int32_t num_tokens, num_experts, top_k, renormalize, hidden;
  int32_t local_expert_begin, local_expert_end;
  int64_t x_stride_t;
  /* The router of a W8A8 checkpoint (a recipe over every Linear quantises it too): operand_dtype = HYMETA_I8, x = the
   * tokens' int8 rows with x_scale fp32 [num_tokens] (hymeta_w8a8_quantize), gate_weight int8 [num_experts, hidden],
   * gate_scale fp32 [num_experts]; the logits are those of hymeta_w8a8_gemm on these operands, bit for bit.  hidden and
   * x_stride_t multiples of 16.  HYMETA_BF16 (0): the fields are ignored. */
  int32_t operand_dtype;
  const float* x_scale;
  const float* gate_scale;
} hymeta_moe_router_args;

int hymeta_moe_router(const hymeta_moe_router_args* args, hymeta_stream_t stream);

typedef struct hymeta_moe_gemm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;               /* bf16 rows of K */
  const void* weights;         /* bf16 [experts, N, K] */
  void* out;                   /* bf16 [slots, N] */
  const int32_t* expert_offsets;   /* [experts + 1], device */
  const int32_t* a_rows;       /* source row of slot s = a_rows[s] / a_rows_div, or NULL: s */
  const int32_t* scale_rows;   /* index into row_scale per slot, or NULL: s */
  const float* row_scale;      /* or NULL */
  int32_t a_rows_div;
  int32_t max_rows;            /* upper bound on the slots in use (num_tokens * top_k) */
  int32_t K, N;
  int32_t local_expert_begin, local_expert_end;
  int64_t x_stride_m, out_stride_m, w_stride_e;
  /* The combine inside the down projection of a decode-sized batch (hymeta_moe_gemm_fuses(...) == 1; otherwise leave
   * these NULL or get HYMETA_ERR_UNSUPPORTED):
   *  combine_tokens [slots]: row `slot` is written to out[combine_tokens[slot]] instead of out[slot] -- with one choice
   *                 per token and every expert local (sorted_pair is then the token of a slot) this is hymeta_moe_combine
   *  combine_shared bf16 [num_tokens, N] or NULL: the shared expert's rows, added to the rounded row as
   *                 hymeta_moe_combine adds them
   * (Folding the activation into this GEMM's operand loads was built and measured too: 7 % slower than the separate
   * launch -- the sigmoid sits between a load and its use, and its registers cost the kernel a wave per SIMD.) */
  const int32_t* combine_tokens;
  const void* combine_shared;
  int64_t combine_shared_stride_m;
  /* W8A8 experts (a compressed-tensors "W8A8" checkpoint as hf_model/quant_w8a8.py:121-132 writes it quantises the
   * experts' Linear layers too; executed in the reference by vLLM's int8 fused_moe -- third-party arithmetic, parity
   * unpinned): operand_dtype = HYMETA_I8, x = int8 rows (hymeta_w8a8_quantize of the tokens for the first GEMM,
   * hymeta_silu_and_mul_int8 for the second), x_scale fp32 per SOURCE row (indexed like x), weights int8 [experts, N, K],
   * w_scale fp32 [experts, N].  out = bf16( (sum_k xq wq) * x_scale[row] * w_scale[e, n] (* row_scale) ): exact int32
   * sums, scales applied once.  K % 64 == 0; strides in elements, multiples of 16.  HYMETA_BF16 (0): the fields are
   * ignored. */
  int32_t operand_dtype;
  const float* x_scale;
  const float* w_scale;
} hymeta_moe_gemm_args;

int hymeta_moe_gemm(const hymeta_moe_gemm_args* args, hymeta_stream_t stream);
/* 1 when hymeta_moe_gemm runs a problem of these sizes in the kernel that has the fused combine (K in bf16 elements:
 * pass K / 2 for int8 operands, the choice goes by the bytes of a row). */
int hymeta_moe_gemm_fuses(int32_t max_rows, int32_t experts, int32_t K, int32_t N);

/* GPTQ experts (hf_model/hymeta_gptq.py:30-32 lists the routed experts' projections among the quantised modules; the
 * reference runs them through vLLM's quantised fused_moe, third party: parity unpinned).  The tensors of one projection
 * stacked over the experts, each expert's as hymeta_gptq_gemm takes them: qweight int32 [experts, K / pack, N], qzeros
 * int32 [experts, K / group, N / pack], scales fp16 | bf16 [experts, K / group, N], contiguous; pointers address expert 0
 * (a rank that holds experts [lo, hi) only passes its base minus lo experts, as for hymeta_moe_gemm).  Rows, a_rows,
 * row_scale, offsets: as hymeta_moe_gemm.  The kernel walks an expert's rows in tiles of 16 or 32 and re-reads the
 * expert's weights per tile: meant for decode-sized batches; for prefill-sized ones hymeta_moe_gptq_expand + the bf16
 * hymeta_moe_gemm.  bits 4 | 8, group 32 | 64 | 128, K % group == 0, N % 64 == 0. */
typedef struct hymeta_moe_gptq_gemm_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* x;               /* bf16 rows of K */
  const int32_t* qweight;
  const int32_t* qzeros;
  const void* scales;
  void* out;                   /* bf16 [slots, N] */
  const int32_t* expert_offsets;   /* [experts + 1], device */
  const int32_t* a_rows;       /* source row of slot s = a_rows[s] / a_rows_div, or NULL: s */
  const int32_t* scale_rows;   /* index into row_scale per slot, or NULL: s */
  const float* row_scale;      /* or NULL */
  int32_t a_rows_div;
  int32_t max_rows;
  int32_t K, N;
  int32_t bits, group_size, scales_dtype, zero_offset;
  int32_t local_expert_begin, local_expert_end;
  int64_t x_stride_m, out_stride_m;
} hymeta_moe_gptq_gemm_args;

int hymeta_moe_gptq_gemm(const hymeta_moe_gptq_gemm_args* args, hymeta_stream_t stream);

/* `experts` stacked GPTQ matrices (layouts above, starting at the pointers given) -> bf16 W^T [experts, N, K], the
 * layout of the bf16 expert stacks: one launch.  K % (16 * (32 / bits)) == 0. */
int hymeta_moe_gptq_expand(const int32_t* qweight, const int32_t* qzeros, const void* scales, void* out, int32_t experts,
                           int32_t K, int32_t N, int32_t bits, int32_t group_size, int32_t scales_dtype, int32_t zero_offset,
                           hymeta_stream_t stream);

/* num_rows_dev: optional device count of the rows in use (e.g. expert_offsets + E); NULL: num_rows. */
int hymeta_silu_and_mul(const void* in, void* out, const int32_t* num_rows_dev, int32_t num_rows,
                        int32_t inter, int64_t in_stride, int64_t out_stride, hymeta_stream_t stream);

/* The same activation followed by the row's dynamic int8 quantisation (the input of a W8A8 down projection): the bf16
 * values hymeta_silu_and_mul would write are quantised as hymeta_w8a8_quantize quantises them (s = max|h| / 127,
 * q = clamp(rint(h / s))) without passing through memory: out_q int8 [num_rows, inter] contiguous, out_scale fp32
 * [num_rows].  inter % 8 == 0 and inter <= 24576; one workgroup per row. */
int hymeta_silu_and_mul_int8(const void* in, int8_t* out_q, float* out_scale, const int32_t* num_rows_dev, int32_t num_rows,
                             int32_t inter, int64_t in_stride, hymeta_stream_t stream);

typedef struct hymeta_moe_combine_args {
  uint32_t struct_size;     /* sizeof(this struct) as the caller compiled it; see "Versioning" above */
  const void* y;               /* bf16 [slots, hidden] */
  const void* shared;          /* bf16 [num_tokens, hidden] or NULL */
  void* out;                   /* bf16 [num_tokens, hidden] */
  const int32_t* pair_pos;     /* [num_tokens * top_k] */
  const float* topk_weights;   /* unused (kept for callers that weight at combine time) */
  int32_t num_tokens, hidden, top_k;
  int64_t y_stride_m, shared_stride_m, out_stride_m;
} hymeta_moe_combine_args;

int hymeta_moe_combine(const hymeta_moe_combine_args* args, hymeta_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * One-shot all-reduce of the small messages of a tensor-parallel decode step over peer-mapped device
 * memory (xGMI between the GPUs of a node): the row-parallel out_proj / down_proj outputs, bf16
 * [<= 256, hidden], and the [T, 2] fp32 branch variances.  Replaces, for these sizes, vLLM's
 * tensor_model_parallel_all_reduce as the reference calls it (vllm_hymeta/models/modeling_hymeta.py:
 * 158-160 for the variances, 858-861 and the RowParallelLinear layers for the projections); larger
 * messages stay with RCCL.  Every rank writes its message into a slot of every rank's exchange buffer,
 * raises a flag there, waits for the flags of all ranks in its own buffer and adds the slots in rank
 * order in fp32, rounding once: bit-identical results on all ranks.
 *
 *   hymeta_p2p_create            allocates this rank's exchange buffer (uncached device memory on the current device,
 *                                room for messages of max_message_bytes) and, when ipc_handle is not NULL, writes its
 *                                HYMETA_P2P_HANDLE_BYTES-byte HIP IPC handle there; world in {1, 2, 4, 8}
 *   hymeta_p2p_connect           all_handles = the world handles in rank order (the caller gathers them, e.g. with an
 *                                all-gather of its process group); peer_devices = the ranks' device ordinals as THIS
 *                                process numbers them, or NULL.  Checks peer access, maps the peers' buffers and writes a
 *                                token into each with a runtime copy; anything that does not work is an error code here
 *                                (HYMETA_ERR_UNSUPPORTED), never a fault in a kernel later
 *   hymeta_p2p_verify            after every rank has connected (a barrier of the caller's): HYMETA_OK when the token of
 *                                every mapped peer has arrived in this rank's buffer
 *   hymeta_p2p_connect_pointers  the same for ranks that live in one process: buffers[q] = hymeta_p2p_local_buffer of
 *                                rank q's communicator
 *   hymeta_p2p_all_reduce        in place: data_bf16[0 .. n_bf16) and stats[0 .. n_stats) become their sums over the
 *                                ranks, one exchange for both (either may be empty); the fp32 sums are multiplied by
 *                                stats_scale (1 for a plain sum; 1 / world for the mean of the ranks' mean squares, which
 *                                is what the branch norms need).  n_bf16 a multiple of 8, data 16-byte
 *                                aligned; (n_stats * 4 rounded up to 16) + n_bf16 * 2 <= max_message_bytes, else
 *                                HYMETA_ERR_SCRATCH_LIMIT.  One kernel launch, no host value that changes per call (the
 *                                epoch lives in device memory), so a captured launch replays.  All ranks must make the
 *                                same calls with the same sizes in the same order, and a rank's calls on one communicator
 *                                must be ordered (one stream or one graph): calls that overlap on a rank race.
 *   hymeta_p2p_set_wait_limit    bound of a wait for a peer in ticks of the 100 MHz wall clock (default 2 s).  A block
 *                                that gives up counts a timeout and goes on with undefined output: every launch ends.
 *   hymeta_p2p_timeouts          timeouts so far (reads device memory: synchronises; not for a captured region)
 * -------------------------------------------------------------------------------------------- */
#define HYMETA_P2P_HANDLE_BYTES 64
typedef struct hymeta_p2p_comm hymeta_p2p_comm;

int hymeta_p2p_create(int rank, int world, size_t max_message_bytes, hymeta_p2p_comm** comm, void* ipc_handle);
void* hymeta_p2p_local_buffer(const hymeta_p2p_comm* comm);
int hymeta_p2p_connect(hymeta_p2p_comm* comm, const void* all_handles, const int32_t* peer_devices);
int hymeta_p2p_verify(hymeta_p2p_comm* comm);
int hymeta_p2p_connect_pointers(hymeta_p2p_comm* comm, void* const* buffers);
int hymeta_p2p_set_wait_limit(hymeta_p2p_comm* comm, uint64_t ticks_100mhz);
int hymeta_p2p_all_reduce(hymeta_p2p_comm* comm, void* data_bf16, int64_t n_bf16, float* stats, int64_t n_stats,
                          float stats_scale, hymeta_stream_t stream);
/* The same exchange for a K-sliced linear's result that is still in its partial sums (hymeta_*_gemm_deferred): the
 * push adds the slices while it reads them -- the bits hymeta_gemm_partials_reduce would have written -- and the sum over
 * the ranks goes to out_bf16 [M, N] contiguous: the row-parallel projection's reduce launch disappears into its
 * all-reduce.  M * N * 2 <= max_message_bytes, N a multiple of 8. */
int hymeta_p2p_all_reduce_partials(hymeta_p2p_comm* comm, const hymeta_gemm_partials* partials, void* out_bf16,
                                   hymeta_stream_t stream);
int hymeta_p2p_timeouts(hymeta_p2p_comm* comm, uint32_t* count);
int hymeta_p2p_destroy(hymeta_p2p_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* HYMETA_HIP_H */
