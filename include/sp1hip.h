/* include/sp1hip.h — C ABI of libsp1hip.so, the MI355X (gfx950) backend for SP1's core-shard
 * commit/open hot path.
 *
 * This is the drop-in boundary: plain `extern "C"` functions, raw device pointers, sizes, no C++ or
 * torch types. It follows the conventions of the reference's own GPU FFI crate `sp1-gpu-sys`
 * (/root/reference/sp1-gpu/crates/sys/src/runtime.rs:L3-L172): the caller owns every buffer,
 * every compute call is asynchronous on the given stream, field elements are u32 words in
 * Montgomery form (R = 2^32) exactly as `KoalaBear` is laid out in Rust memory, extension elements
 * are 4 consecutive words. A Rust crate wraps these the way `sp1-gpu-cudart` wraps `sp1-gpu-sys`
 * (see INTEGRATION.md).
 *
 * Device layouts:
 *   base tensor   [height x width]  COLUMN-major: word (row r, col c) at  c*height + r
 *   ext vector    [len]             4 base columns (SoA): coordinate k of element i at k*len + i
 *   digest        8 words, array-of-structs
 *   Merkle tree   leaf-first layers back to back: 2^h leaf digests, 2^(h-1) parents, ..., root;
 *                 (2^(h+1) - 1) digests in total
 *
 * Every function returns SP1HIP_SUCCESS (0) or a negative sp1hip_status; the message for the
 * calling thread's last failure is sp1hip_last_error(). Nothing throws across this ABI.
 */
#ifndef SP1HIP_H
#define SP1HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SP1HIP_SUCCESS = 0,
    SP1HIP_ERROR_INVALID_ARGUMENT = -1,
    SP1HIP_ERROR_OUT_OF_MEMORY = -2,   /* cf. CUDA_OUT_OF_MEMORY,       runtime.rs:L3-L15 */
    SP1HIP_ERROR_NOT_READY = -3,       /* cf. CUDA_ERROR_NOT_READY_SLOP, runtime.rs:L3-L15 */
    SP1HIP_ERROR_RUNTIME = -4,         /* any other HIP failure */
    SP1HIP_ERROR_NO_DEVICE = -5,
    SP1HIP_ERROR_BUFFER_TOO_SMALL = -6
} sp1hip_status;

typedef void* sp1hip_stream_t; /* hipStream_t; NULL = the default stream */
typedef void* sp1hip_event_t;  /* hipEvent_t */

/* Extension-field element passed by value (4 Montgomery words). */
typedef struct { uint32_t c[4]; } sp1hip_ext_t;

/* One committed tensor of a `Message<Tensor>`: column-major device data, `width` columns. */
typedef struct {
    const uint32_t* d_data;
    uint32_t width;
} sp1hip_tensor_t;

const char* sp1hip_last_error(void);
const char* sp1hip_version(void);

/* ---------------------------------------------------------------- runtime
 * Replaces sp1-gpu-sys `cuda_malloc[_async]`, `cuda_free[_async]`, `cuda_malloc_host`,
 * `cuda_mem_copy_*`, `cuda_stream_*`, `cuda_event_*`, `cuda_mem_get_info`
 * (/root/reference/sp1-gpu/crates/sys/src/runtime.rs:L16-L172). */
int sp1hip_device_count(int* count);
int sp1hip_set_device(int device);
int sp1hip_get_device(int* device);
int sp1hip_mem_info(size_t* free_bytes, size_t* total_bytes);
/* The library keeps its working buffers (codewords, trees, fold layers) in per-stream free lists so that
 * steady-state proving never calls the driver; this returns every cached block to the driver. */
int sp1hip_mem_trim(size_t* released_bytes);
/* A caller stream that has been passed to this library owns helper streams (the commit's encode stream, the zerocheck's fork
 * streams), their events and the working buffers cached for it and for them. Call this BEFORE destroying such a stream (the
 * prover pool does it for its own slots): it waits for the stream, destroys the helpers and hands every block cached for them
 * and for the stream back to the driver. Without it the helpers outlive the stream, and a later stream that reuses the handle
 * value would inherit them. */
int sp1hip_stream_release(sp1hip_stream_t stream);
/* Host threads (the caller included) the library uses for the arithmetic between device hand-overs: SP1HIP_HOST_THREADS,
 * else min(8, CPUs / (2 x LOCAL_WORLD_SIZE)) with CPUs honouring the cgroup quota. Decided once per process. */
int sp1hip_host_threads(void);
int sp1hip_malloc(void** d_ptr, size_t bytes);
int sp1hip_free(void* d_ptr);
int sp1hip_malloc_async(void** d_ptr, size_t bytes, sp1hip_stream_t stream);
int sp1hip_free_async(void* d_ptr, sp1hip_stream_t stream);
int sp1hip_malloc_host(void** h_ptr, size_t bytes);
int sp1hip_free_host(void* h_ptr);
int sp1hip_memcpy_h2d_async(void* d_dst, const void* h_src, size_t bytes, sp1hip_stream_t stream);
int sp1hip_memcpy_d2h_async(void* h_dst, const void* d_src, size_t bytes, sp1hip_stream_t stream);
int sp1hip_memcpy_d2d_async(void* d_dst, const void* d_src, size_t bytes, sp1hip_stream_t stream);
int sp1hip_memset_async(void* d_dst, int value, size_t bytes, sp1hip_stream_t stream);
int sp1hip_stream_create(sp1hip_stream_t* stream);
int sp1hip_stream_destroy(sp1hip_stream_t stream);     /* releases what the library keeps for the stream first (sp1hip_stream_release) */
int sp1hip_stream_synchronize(sp1hip_stream_t stream);
int sp1hip_stream_query(sp1hip_stream_t stream); /* SP1HIP_ERROR_NOT_READY while work is pending */
int sp1hip_event_create(sp1hip_event_t* event);
int sp1hip_event_destroy(sp1hip_event_t event);
int sp1hip_event_record(sp1hip_event_t event, sp1hip_stream_t stream);
int sp1hip_event_synchronize(sp1hip_event_t event);
int sp1hip_event_elapsed_ms(float* ms, sp1hip_event_t start, sp1hip_event_t stop);
int sp1hip_stream_wait_event(sp1hip_stream_t stream, sp1hip_event_t event);

/* ---------------------------------------------------------------- kernel timers (measurement aid)
 * When enabled, the library brackets its hot kernels with HIP events on the caller's stream
 * (the equivalent of the reference's NVTX ranges + `SP1_GPU_*_TIMING` switches,
 * /root/reference/sp1-gpu/crates/tracing/src/tracer.rs). `sp1hip_timers_read` synchronises the recorded
 * events and returns, for `name` ("leaf_hash", "compress", "ntt_pass", ...), the number of launches and
 * their summed duration since the last `sp1hip_timers_reset`. */
int sp1hip_timers_enable(int on);
int sp1hip_timers_reset(void);
int sp1hip_timers_read(const char* name, uint64_t* launches, double* total_ms);

/* ---------------------------------------------------------------- layout
 * Host traces are row-major `[rows][cols]` (`RowMajorMatrix`, slop Tensor); the device wants
 * column-major. Replaces sp1-gpu's transpose kernels (/root/reference/sp1-gpu/crates/sys/lib/transpose). */
int sp1hip_transpose_to_col_major(uint32_t* d_out, const uint32_t* d_in_row_major, size_t rows, size_t cols,
                                  sp1hip_stream_t stream);
int sp1hip_transpose_to_row_major(uint32_t* d_out, const uint32_t* d_in_col_major, size_t rows, size_t cols,
                                  sp1hip_stream_t stream);
/* Host traces -> device tables in one call (staging half). Replaces the per-chip
 * "copy host trace to device" + `DeviceTensor::transpose` of `device_main_tracegen`
 * (/root/reference/sp1-gpu/crates/jagged_tracegen/src/lib.rs:L819-L835). Table i is `rows x cols` Montgomery
 * words, row-major, at h_data (pinned memory — `sp1hip_malloc_host` / `sp1hip_host_register` — for an
 * asynchronous copy at full PCIe rate; pageable memory works but is copied through the runtime's bounce
 * buffers); d_out[i] receives it column-major (`cols` columns of `rows` words). Chunked copies on a side
 * stream overlap the transposes on `stream`; returns after enqueueing. The host buffers must stay untouched
 * until `stream` has passed this call. */
typedef struct {
    const uint32_t* h_data;
    uint64_t rows;
    uint32_t cols;
} sp1hip_host_table_t;
int sp1hip_stage_tables(const sp1hip_host_table_t* tables, int n_tables, uint32_t* const* d_out, sp1hip_stream_t stream);
/* hipHostRegister / hipHostUnregister on caller-owned memory (the reference's `cuda_host_register`,
 * /root/reference/sp1-gpu/crates/sys/src/runtime.rs:L75-L172). */
int sp1hip_host_register(void* h_ptr, size_t bytes);
int sp1hip_host_unregister(void* h_ptr);
/* canonical <-> Montgomery, in place */
int sp1hip_to_monty(uint32_t* d_data, size_t n, sp1hip_stream_t stream);
int sp1hip_from_monty(uint32_t* d_data, size_t n, sp1hip_stream_t stream);

/* ---------------------------------------------------------------- Reed–Solomon encode (a4)
 * Replaces `Dft::dft(src, log_blowup, DftOrdering::BitReversed, dim = 0)` as called by
 * `CpuDftEncoder::encode_batch` (/root/reference/slop/crates/basefold-prover/src/encoder.rs:L22-L38,
 * /root/reference/slop/crates/dft/src/lib.rs:L17-L71) and sp1-gpu-sys `batch_coset_dft`
 * (/root/reference/sp1-gpu/crates/sys/src/dft.rs:L12-L59).
 * d_in: [2^lg_n x n_cols] column-major coefficients; d_out: [2^(lg_n+lg_blowup) x n_cols]
 * column-major, row j of a column holds f(w_N^{bitrev(j)}). d_out must not alias d_in. */
int sp1hip_rs_encode_batch(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols,
                           sp1hip_stream_t stream);

/* ---------------------------------------------------------------- Poseidon2 Merkle TCS (a2, a3, a6)
 * Replaces `TensorCsProver::commit_tensors` / `prove_openings_at_indices` and
 * `ComputeTcsOpenings::compute_openings_at_indices`
 * (/root/reference/slop/crates/merkle-tree/src/tcs.rs:L15-L47, p3sync.rs:L40-L238).
 * d_tree: (2^(lg_height+1) - 1) * 8 words. d_root_and_commit: 16 words — the Merkle root followed by
 * the commitment compress(root, hash([lg_height, total_width])). */
int sp1hip_merkle_commit(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, uint32_t* d_tree,
                         uint32_t* d_root_and_commit, sp1hip_stream_t stream);
/* d_indices: n_idx row indices. d_values: [n_idx][total_width] row-major (tensor order);
 * d_paths: [n_idx][lg_height][8]. Either output may be NULL. */
int sp1hip_merkle_open(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, const uint32_t* d_tree,
                       const uint32_t* d_indices, size_t n_idx, uint32_t* d_values, uint32_t* d_paths,
                       sp1hip_stream_t stream);
/* Batched permutations / hashes for testing and for small host-side uses:
 * d_states [n][16] permuted in place. */
int sp1hip_poseidon2_permute(uint32_t* d_states, size_t n, sp1hip_stream_t stream);
/* The same permutation in its all-integer formulation (modular-add linear layers, unsigned Montgomery S-box). The
 * production kernels use the exact-fp64 linear layer + signed S-box of sp1_amd/csrc/poseidon2.hpp; this entry point
 * exists so that the two independent formulations can be compared on hundreds of millions of states. */
int sp1hip_poseidon2_permute_integer_form(uint32_t* d_states, size_t n, sp1hip_stream_t stream);
/* The Fiat-Shamir transcript's permutation, on the HOST (no device needed): h_states [n][16] canonical Montgomery
 * words, permuted in place. The transcript is a duplex sponge (`DuplexChallenger<F, Perm, 16, 8>` through
 * `IopCtx::Challenger`, /root/reference/slop/crates/challenger/src/lib.rs:L25-L87): every permutation waits for the
 * one before, and the GPU waits for the transcript, so its latency is proof time. form 0 = what the transcript runs
 * (AVX-512 on CPUs that have it: sp1_amd/csrc/p2_host.cpp; the scalar integer form otherwise), 1 = scalar integer
 * form, 2 = scalar fp64 form; all three return the same words. sp1hip_host_permutation_is_vectorised: 1 if form 0 is
 * the AVX-512 path on this CPU. */
int sp1hip_poseidon2_permute_host(uint32_t* h_states, size_t n, int form);
int sp1hip_host_permutation_is_vectorised(void);
/* Host-only test hooks for the vectorised interaction-variable rounds of LogUp-GKR (sp1_amd/csrc/gkr_host.cpp; the rounds of
 * `InteractionLayer`, /root/reference/crates/hypercube/src/logup_gkr/logup_poly.rs:L240-L316, on coefficient planes): tab = 4 tables
 * (n0, d0, n1, d1) x 4 planes of `stride` words (multiple of 16, >= 2 * pairs rounded up to 8, + 16), eq = 4 planes of eq_stride
 * words. sums: out24 = [x0 | y0 | xh | yh | e0 | es] over the first real_pairs (even, odd) pairs; fold: out[k] = t[2k] +
 * alpha (t[2k+1] - t[2k]). Return 0, or -1 when the CPU has no AVX-512 (the library then runs the scalar rounds) or an
 * argument is malformed. */
int sp1hip_gkr_host_simd_available(void);
int sp1hip_gkr_host_round_sums(const uint32_t* tab, size_t stride, const uint32_t* eq, size_t eq_stride, size_t real_pairs, uint32_t* out24);
int sp1hip_gkr_host_round_fold(const uint32_t* tab, uint32_t* out, size_t stride, size_t real_pairs, const uint32_t* alpha4);

/* ---------------------------------------------------------------- the commit path over BabyBear (BASELINE config 2, "both fields")
 * The second `IopCtx` of the reference (/root/reference/slop/crates/baby-bear/src/baby_bear_poseidon2.rs:L11-L55): p = 2^31 - 2^27 + 1,
 * Montgomery words (R = 2^32), Poseidon2 width 16 with x^7, 8 + 13 rounds, same sponge / compression / commit_tensors and the same
 * layouts as the KoalaBear entry points above. PARITY NOTE: the internal diffusion matrix is not restated in the reference tree
 * (oracle/bb_commit.hpp). sp1hip_bb_commit_mles: d_codewords[k] receives the codeword of mle k (2^(lg_n + lg_blowup) x width,
 * column-major), d_tree the (2^(lg + 1) - 1) digests leaf-first; synchronises the stream to return the commitment. */
int sp1hip_bb_rs_encode_batch(uint32_t* d_out, const uint32_t* d_in, int lg_n, int lg_blowup, size_t n_cols, sp1hip_stream_t stream);
int sp1hip_bb_merkle_commit(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, uint32_t* d_tree,
                            uint32_t* d_root_and_commit, sp1hip_stream_t stream);
int sp1hip_bb_commit_mles(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t* const* d_codewords,
                          uint32_t* d_tree, uint32_t h_commit[8], sp1hip_stream_t stream);
int sp1hip_bb_poseidon2_permute(uint32_t* d_states, size_t n, sp1hip_stream_t stream);

/* ---------------------------------------------------------------- BaseFold kernels (a13, a14)
 * batch: out[r] = sum_c coeff[c] * col_c[r] over all columns of all tensors (message order)
 *   (`FriCpuProver::batch`, /root/reference/slop/crates/basefold-prover/src/fri.rs:L31-L80).
 *   d_coeffs: [total_cols] ext, array-of-structs (4 words each). d_out: ext vector [2^lg_height]. */
int sp1hip_basefold_batch(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height,
                          const uint32_t* d_coeffs, uint32_t* d_out, sp1hip_stream_t stream);
/* codeword fold (`p3_fri::fold_even_odd` as called at fri.rs:L118): ext vector 2^lg_n -> 2^(lg_n-1) */
int sp1hip_fold_even_odd(const uint32_t* d_codeword, int lg_n, sp1hip_ext_t beta, uint32_t* d_out,
                         sp1hip_stream_t stream);
/* `Mle::fold` (/root/reference/slop/crates/multilinear/src/fold.rs:L12-L26): out[i] = m[2i] + beta m[2i+1] */
int sp1hip_fold_mle(const uint32_t* d_mle, int lg_n, sp1hip_ext_t beta, uint32_t* d_out, sp1hip_stream_t stream);
/* eq(point, .) table (`partial_lagrange`, /root/reference/slop/crates/multilinear/src/lagrange.rs:L19-L45);
 * h_point: dim ext elements on the host; d_out: ext vector [2^dim]. */
int sp1hip_partial_lagrange(const sp1hip_ext_t* h_point, int dim, uint32_t* d_out, sp1hip_stream_t stream);
/* `eval_mle_at_point` for every column (/root/reference/slop/crates/multilinear/src/eval.rs:L9-L21):
 * d_evals[c] = sum_r d_eq[r] * col_c[r]; d_evals: [total_cols] ext array-of-structs. */
int sp1hip_mle_eval_columns(const sp1hip_tensor_t* tensors, int n_tensors, int lg_height, const uint32_t* d_eq,
                            uint32_t* d_evals, sp1hip_stream_t stream);
/* `Mle::fixed_at_zero` for an ext mle (/root/reference/slop/crates/multilinear/src/restrict.rs:L75-L87):
 * d_out[4] = sum_i d_eq[i] * d_mle[2 i]; d_mle ext vector [2^lg_n], d_eq ext vector [2^(lg_n-1)]. */
int sp1hip_ext_fixed_at_zero(const uint32_t* d_mle, int lg_n, const uint32_t* d_eq, uint32_t* d_out,
                             sp1hip_stream_t stream);

/* `mle_fix_last_variable` for a whole table (/root/reference/slop/crates/multilinear/src/restrict.rs:L10-L72; the
 * zerocheck prover's per-round table update, /root/reference/crates/hypercube/src/prover/zerocheck/mod.rs:L156-L171):
 * out[i][c] = x + alpha (y - x), x = row 2i, y = row 2i + 1, or the column's padding value when 2i + 1 == rows.
 * d_in: column-major, `rows x width` base words (in_is_ext == 0) or the extension layout below (in_is_ext != 0).
 * d_out: extension table of ceil(rows / 2) rows in the layout the sumcheck kernels use: word q of column c, row i at
 * d_out[(4 c + q) * out_rows + i]. d_padding: `width` base words (or 4 width words, (c, q) order, for an extension
 * input) in DEVICE memory, or NULL for zero padding. */
int sp1hip_fix_last_variable(const uint32_t* d_in, uint64_t rows, uint32_t width, int in_is_ext, sp1hip_ext_t alpha,
                             const uint32_t* d_padding, uint32_t* d_out, sp1hip_stream_t stream);

/* ---------------------------------------------------------------- transcript (a17)
 * `DuplexChallenger<KoalaBear, KoalaPerm, 16, 8>` (/root/reference/slop/crates/challenger/src/lib.rs:L25-L87).
 * Host object; `grind` runs the witness search on the GPU and returns the SMALLEST valid witness. */
typedef struct sp1hip_challenger_s sp1hip_challenger_t;
int sp1hip_challenger_new(sp1hip_challenger_t** out);
int sp1hip_challenger_clone(const sp1hip_challenger_t* ch, sp1hip_challenger_t** out);
void sp1hip_challenger_free(sp1hip_challenger_t* ch);
int sp1hip_challenger_observe(sp1hip_challenger_t* ch, const uint32_t* felts, size_t n);
int sp1hip_challenger_sample(sp1hip_challenger_t* ch, uint32_t* out);
int sp1hip_challenger_sample_ext(sp1hip_challenger_t* ch, sp1hip_ext_t* out);
int sp1hip_challenger_sample_bits(sp1hip_challenger_t* ch, int bits, uint32_t* out);
int sp1hip_challenger_check_witness(sp1hip_challenger_t* ch, int bits, uint32_t witness, int* ok);
int sp1hip_challenger_grind(sp1hip_challenger_t* ch, int bits, uint32_t* witness, sp1hip_stream_t stream);
/* Proof-of-work witnesses to USE instead of searching, in the order the proof grinds (a shard proof: LogUp-GKR 12 bits,
 * BaseFold batching 5 bits, BaseFold queries `proof_of_work_bits`); canonical words, at most 4. The reference's rayon
 * `find_any` (and its CUDA grind) return ANY valid witness, this library's search the smallest: a caller that replays a
 * Rust-made proof injects that proof's witnesses and gets the same bytes. An injected witness the transcript rejects
 * is an error (SP1HIP_ERROR_INVALID_ARGUMENT). Consumed by grind; cleared with n = 0. */
int sp1hip_challenger_inject_pow_witnesses(sp1hip_challenger_t* ch, const uint32_t* witnesses, int n);
/* 34 words: sponge state[16], n_in, in[8], n_out, out[8] */
int sp1hip_challenger_state(const sp1hip_challenger_t* ch, uint32_t* out34);

/* ---------------------------------------------------------------- BaseFold prover (a15, a16)
 * `BasefoldProver::commit_mles` (/root/reference/slop/crates/basefold-prover/src/prover.rs:L78-L99):
 * RS-encode every mle of one commitment round and Merkle-commit the codewords. The returned
 * handle owns the device codewords + tree (`BasefoldProverData`); the input mles stay caller-owned
 * and must outlive the handle (they are read again by `sp1hip_basefold_prove`). */
typedef struct sp1hip_basefold_data_s sp1hip_basefold_data_t;
int sp1hip_commit_mles(const sp1hip_tensor_t* mles, int n_mles, int lg_n, int lg_blowup, uint32_t h_commit[8],
                       sp1hip_basefold_data_t** out, sp1hip_stream_t stream);
/* Prover data is scratch of the stream it was committed on: freeing it orders the reuse of its blocks behind THAT stream's
 * work. A handle that was also opened on another stream (a proving key shared by provers on their own streams) waits for
 * the device when it is freed; do not free a handle while a prove call that uses it is still running on another thread. */
void sp1hip_basefold_data_free(sp1hip_basefold_data_t* data);
/* accessors for parity tests */
int sp1hip_basefold_data_codeword(const sp1hip_basefold_data_t* data, int mle_index, const uint32_t** d_codeword,
                                  uint32_t* width, int* lg_height);
int sp1hip_basefold_data_tree(const sp1hip_basefold_data_t* data, const uint32_t** d_tree, int* lg_height);

typedef struct {
    int log_blowup;          /* core: 2  (/root/reference/crates/primitives/src/fri_params.rs:L5-L15) */
    int num_queries;         /* core: 124 */
    int proof_of_work_bits;  /* core: 16 */
} sp1hip_fri_config_t;

/* `BasefoldProver::prove_trusted_mle_evaluations` (prover.rs:L102-L243). `rounds[r]` are the handles of
 * the commitment rounds in order; h_claims holds one ext per column flattened round -> mle -> column.
 * Writes the bincode encoding of `BasefoldProof` (/root/reference/slop/crates/basefold/src/verifier.rs:L94-L116)
 * into h_proof (capacity *proof_len on entry, size on return; SP1HIP_ERROR_BUFFER_TOO_SMALL sets the needed
 * size and leaves the challenger untouched). */
int sp1hip_basefold_prove(const sp1hip_ext_t* h_point, int dim, sp1hip_basefold_data_t* const* rounds, int n_rounds,
                          const sp1hip_ext_t* h_claims, size_t n_claims, sp1hip_fri_config_t config,
                          sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                          sp1hip_stream_t stream);
size_t sp1hip_basefold_proof_size(int dim, const uint32_t* round_widths, int n_rounds, sp1hip_fri_config_t config);

/* ---------------------------------------------------------------- stacked + jagged commit (a5, a7, a8)
 * One chip table: column-major [rows x cols] device words (rows = real rows, no padding). */
typedef struct {
    const uint32_t* d_data;
    uint64_t rows;
    uint32_t cols;
} sp1hip_table_t;
typedef struct sp1hip_stacked_data_s sp1hip_stacked_data_t;

/* `StackedPcsProver::commit_multilinears` (/root/reference/slop/crates/stacked/src/prover.rs:L59-L94) with
 * `interleave_multilinears_with_fixed_rate` (/root/reference/slop/crates/stacked/src/fixed_rate.rs:L6-L47):
 * dense column-major concatenation of the tables, zero-padded to a multiple of 2^log_stacking_height
 * (at least one column), cut into batches of `batch_size` stacked columns, BaseFold-committed.
 * The tables are copied; the returned handle owns the dense buffer and the BaseFold data. */
int sp1hip_stacked_commit(const sp1hip_table_t* tables, int n_tables, int log_stacking_height, int batch_size,
                          int lg_blowup, uint32_t h_commit[8], uint64_t* num_added_vals, sp1hip_stacked_data_t** out,
                          sp1hip_stream_t stream);
void sp1hip_stacked_data_free(sp1hip_stacked_data_t* data);
int sp1hip_stacked_data_info(const sp1hip_stacked_data_t* data, sp1hip_basefold_data_t** basefold, int* n_batches,
                             const uint32_t** d_dense, uint64_t* padded_area);
int sp1hip_stacked_batch(const sp1hip_stacked_data_t* data, int k, sp1hip_tensor_t* batch);
/* `JaggedProver::commit_multilinears` (/root/reference/slop/crates/jagged/src/prover.rs:L106-L160), the call made by
 * `ShardProver::commit_traces` (/root/reference/crates/hypercube/src/prover/shard.rs:L462-L468): tables with zero
 * rows are counted but not committed; the result is compress(stacked_commit, hash([n + 2, rows.., cols..]))
 * with the two padding tables appended. */
int sp1hip_jagged_commit(const sp1hip_table_t* tables, int n_tables, int max_log_row_count, int log_stacking_height,
                         int batch_size, int lg_blowup, uint32_t h_commit[8], sp1hip_stacked_data_t** out,
                         sp1hip_stream_t stream);

/* ---------------------------------------------------------------- jagged PCS evaluation proof
 * `JaggedProver::prove_trusted_evaluations` (/root/reference/slop/crates/jagged/src/prover.rs:L162-L328): the jagged
 * sumcheck over the dense vector (`HadamardProduct`, hadamard.rs:L52-L146), the jagged-eval sumcheck
 * (jagged_eval/sumcheck_eval.rs:L185-L243) and the stacked / BaseFold opening of the dense commitments
 * (stacked/src/prover.rs:L107-L152) — everything between the zerocheck point and the end of the shard proof's
 * `evaluation_proof`. rounds[r]: handles returned by sp1hip_jagged_commit, in commitment order (SP1:
 * preprocessed, main); h_z_row: the zerocheck point (max_log_row_count ext); h_claims: the evaluations at z_row
 * of every column of every table of every round (round -> table -> column; no padding columns),
 * claims_per_round[r] of them for round r. Writes bincode(JaggedPcsProof)
 * (/root/reference/slop/crates/jagged/src/verifier.rs:L17-L26) into h_proof (capacity *proof_len on entry, size on
 * return; SP1HIP_ERROR_BUFFER_TOO_SMALL sets the needed size). The challenger is advanced only on success. */
int sp1hip_jagged_prove(const sp1hip_ext_t* h_z_row, int max_log_row_count, sp1hip_stacked_data_t* const* rounds,
                        int n_rounds, const sp1hip_ext_t* h_claims, const size_t* claims_per_round,
                        sp1hip_fri_config_t config, sp1hip_challenger_t* challenger, uint8_t* h_proof,
                        size_t* proof_len, sp1hip_stream_t stream);

/* ---------------------------------------------------------------- LogUp-GKR
 * One chip of the shard for the lookup argument. `interactions`: HOST words describing its sends (first) and
 * receives, the data form of `Interaction { values: Vec<VirtualPairCol>, multiplicity, kind }`
 * (/root/reference/crates/hypercube/src/lookup/interaction.rs:L11-L24):
 *   [n_interactions, then per interaction: is_send, kind, n_values, vcol(multiplicity), vcol(value_0), ...]
 *   vcol = [n_terms, constant (canonical), then n_terms x (is_main, column, weight (canonical))]
 * Traces: column-major device tensors with `real_rows` rows. Chips must be passed in name order (BTreeSet<Chip>). */
typedef struct {
    const char* name;
    const uint32_t* interactions;
    uint32_t n_words;
    uint32_t main_width, prep_width;
    const uint32_t* d_main;
    const uint32_t* d_prep;
    uint64_t real_rows;
} sp1hip_gkr_chip_t;

/* `GkrProverImpl::prove_logup_gkr` (/root/reference/crates/hypercube/src/logup_gkr/prover.rs:L70-L215): 12-bit grind,
 * alpha / beta challenges, the fraction circuit over every (row, interaction) (execution.rs:L112-L382), one degree-3
 * sumcheck per circuit layer (cpu.rs:L146-L226, logup_poly.rs:L70-L553) and the trace-column openings at the
 * final point. Writes bincode(LogupGkrProof) (/root/reference/crates/hypercube/src/logup_gkr/proof.rs:L32-L62): its
 * `logup_evaluations` (point + per-chip openings) are the `h_zeta` / `h_openings` inputs of
 * sp1hip_zerocheck_prove. The challenger is advanced only on success. */
int sp1hip_logup_gkr_prove(const sp1hip_gkr_chip_t* chips, int n_chips, int max_log_row_count,
                           sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                           sp1hip_stream_t stream);

/* ---------------------------------------------------------------- zerocheck (a9-a12)
 * One chip of the shard. `program` is a HOST array of n_instr [op, a, b] triples in SSA form
 * (instruction k defines value k): 0 LOAD_MAIN col, 1 LOAD_PREP col, 2 CONST canonical, 3 PUBLIC idx,
 * 4 ADD, 5 SUB, 6 MUL, 7 NEG a, 8 ASSERT_ZERO a — the data form of `Air::eval` over
 * `ConstraintSumcheckFolder` (/root/reference/crates/hypercube/src/folder.rs:L276-L323); single-row constraints.
 * Optional pseudo-instruction 16 HINT kind col (defines no value, asserts nothing): kind 1 = "the next 163 ASSERT_ZEROs are
 * the Poseidon2 permutation sub-AIR — eval_external_round r = 0..7, then eval_internal_rounds,
 * /root/reference/crates/hypercube/src/operations/poseidon2/air.rs:L66-L144 — over main columns [col, col + 179)"; the
 * prover then evaluates those constraints with a fused kernel instead of interpreting them (the hint is verified against the
 * program before it is used; SP1HIP_ZC_MACRO=0 ignores hints; the proof bytes do not depend on it).
 * Traces: column-major device tensors with `real_rows` rows (padding rows are implicit zeros). */
typedef struct {
    const uint32_t* program;
    uint32_t n_instr;
    uint32_t main_width, prep_width, num_constraints;
    const uint32_t* d_main;
    const uint32_t* d_prep;
    uint64_t real_rows;
} sp1hip_zc_chip_t;

/* `ShardProver::zerocheck` (/root/reference/crates/hypercube/src/prover/shard.rs:L474-L646): one sumcheck over all
 * chips (RLC with lambda sampled from the transcript), 2^max_log_row_count rows per chip. h_zeta: the
 * GKR point (max_log_row_count ext); h_openings: per chip, in order, the main then preprocessed column
 * evaluations at zeta; alpha / gkr_batch: the two challenges sampled by the caller before the call.
 * Output: bincode `PartialSumcheckProof<EF>` (/root/reference/slop/crates/sumcheck/src/proof.rs:L10-L14) followed by
 * u64 n_chips and, per chip, Vec<EF> = preprocessed then main column evaluations at the sumcheck point.
 * The challenger ends in the state after the openings have been observed (shard.rs:L609-L640). */
int sp1hip_zerocheck_prove(const sp1hip_zc_chip_t* chips, int n_chips, int max_log_row_count,
                           const sp1hip_ext_t* h_zeta, const sp1hip_ext_t* h_openings, sp1hip_ext_t alpha,
                           sp1hip_ext_t gkr_batch, const uint32_t* h_publics, int n_publics,
                           sp1hip_challenger_t* challenger, uint8_t* h_proof, size_t* proof_len,
                           sp1hip_stream_t stream);

/* Host-only check of the constraint-program compiler (no GPU needed): plans `program` exactly as sp1hip_zerocheck_prove
 * does (immediates folded, instruction order chosen, registers allocated, fused multiply-adds, forwarded operands; the
 * chunked / undivided / finely cut forms) and interprets the chosen form on ONE row (Montgomery words). form: 0 = the whole
 * allocated program, 1 = chunks, 2 = undivided, 3 = fine. out_values[k] = value of constraint k at the row; out_stats
 * (optional) = {instruction words, pieces, registers}. Fails if a constraint is not evaluated exactly once. */
int sp1hip_zerocheck_plan_eval(const uint32_t* program, uint32_t n_instr, uint32_t main_width, uint32_t prep_width,
                               const uint32_t* main_row, const uint32_t* prep_row, const uint32_t* publics,
                               uint32_t n_publics, int form, uint32_t* out_values, uint32_t n_constraints,
                               uint32_t* out_stats);
/* Host-only check of the polynomial-identity pieces (hint kind 7, sp1_amd/csrc/zc_poly.hpp) without a GPU: plans `program`, builds
 * every identity's device table for the batching challenge `alpha` exactly as sp1hip_zerocheck_prove does, and evaluates the
 * tables on ONE main row the way the kernels do (affine forms, then products). out_collapsed = the sum over the identities of
 * their batched value; out_direct = the same sum taken constraint by constraint, sum_k alpha^(n - 1 - k) C_k(row) over the
 * constraints the identities cover (n = num constraints of the program); *n_identities = how many hints the planner accepted.
 * Equal for every row and alpha, or the pieces are wrong. */
int sp1hip_zerocheck_poly_check(const uint32_t* program, uint32_t n_instr, uint32_t main_width, uint32_t prep_width,
                                const uint32_t* main_row, sp1hip_ext_t alpha, sp1hip_ext_t* out_collapsed, sp1hip_ext_t* out_direct,
                                uint32_t* n_identities);
/* Host-only: the bilinear extension of a column's row quad (r00, r01, r10, r11 = rows 4q .. 4q + 3, Montgomery words) at node
 * `node` (0 .. 11) of the bivariate grid the fused first two zerocheck rounds evaluate — the same function the kernels call
 * (zc_biv_interp: one 36-bit accumulation and its reduction); for tests of its edge cases without a GPU. */
int sp1hip_zerocheck_biv_interp_host(uint32_t r00, uint32_t r01, uint32_t r10, uint32_t r11, uint32_t node, uint32_t* out);

/* ---------------------------------------------------------------- one whole shard proof
 * A chip of the shard with everything the stages need: constraint program (zerocheck, see sp1hip_zc_chip_t),
 * interaction program (LogUp-GKR, see sp1hip_gkr_chip_t) and its device traces. Chips in name order. */
typedef struct {
    const char* name;
    const uint32_t* program;        /* [n_instr][3] constraint program (host) */
    uint32_t n_instr, num_constraints;
    const uint32_t* interactions;   /* interaction program words (host) */
    uint32_t n_words;
    uint32_t main_width, prep_width;
    const uint32_t* d_main;
    const uint32_t* d_prep;
    uint64_t real_rows;
} sp1hip_shard_chip_t;

typedef struct {
    int max_log_row_count;          /* core: 22 */
    int log_stacking_height;        /* core: 21 */
    int batch_size;                 /* stacked columns per BaseFold batch */
    sp1hip_fri_config_t fri;
} sp1hip_shard_params_t;

/* `ShardProver::prove_shard_with_data` (/root/reference/crates/hypercube/src/prover/shard.rs:L650-L792) — the body of
 * `AirProver::prove_shard_with_pk`, the seam a proving backend plugs into. `preprocessed`: the proving key's
 * preprocessed commitment round (sp1hip_jagged_commit over the preprocessed traces of the chips that have one, in
 * chip order, done once at setup). The challenger must already have absorbed the verifying key
 * (`vk.observe_into`). Commits the main traces, runs LogUp-GKR, zerocheck and the jagged evaluation proof in the
 * reference's transcript order and writes bincode(ShardProof)
 * (/root/reference/crates/hypercube/src/verifier/proof.rs:L47-L94). Size protocol and transcript commit-on-success as
 * for the stage entry points. */
int sp1hip_prove_shard(const sp1hip_shard_chip_t* chips, int n_chips, const uint32_t* h_publics, int n_publics,
                       sp1hip_stacked_data_t* preprocessed, sp1hip_shard_params_t params, sp1hip_challenger_t* challenger,
                       uint8_t* h_proof, size_t* proof_len, sp1hip_stream_t stream);

/* ---------------------------------------------------------------- device trace generation (recursion machine)
 * `CudaTracegenAir::generate_trace_device` of the recursion chips
 * (/root/reference/sp1-gpu/crates/tracegen/src/recursion/{alu_base,alu_ext,select,poseidon2_wide,prefix_sum_checks}.rs, mod.rs):
 * d_events = the execution record's event array copied to the device as is (Montgomery words in the `#[repr(C)]` layout
 * of /root/reference/crates/recursion/executor/src/lib.rs: BaseAluIo 3 words, ExtAluIo<Block> 12, SelectIo 5, MemEvent 4,
 * PrefixSumChecksEvent 20, Poseidon2Event 32 = input[16] | output[16]); d_trace = the chip's main trace, column-major
 * [width][height] (widths 3, 12, 5, 8, 15, 179), rows past the events are the reference's padding rows (zeros; for
 * Poseidon2 the permutation trace of the zero state). height >= the number of event rows (MemoryVar: 2 events per row). */
int sp1hip_tracegen_recursion_base_alu(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                       sp1hip_stream_t stream);
int sp1hip_tracegen_recursion_ext_alu(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                      sp1hip_stream_t stream);
int sp1hip_tracegen_recursion_select(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                     sp1hip_stream_t stream);
int sp1hip_tracegen_recursion_memory_var(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                         sp1hip_stream_t stream);
int sp1hip_tracegen_recursion_prefix_sum_checks(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                                sp1hip_stream_t stream);
int sp1hip_tracegen_recursion_poseidon2_wide(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                             sp1hip_stream_t stream);

/* Device trace generation for the RISC-V machine's Global chip (`CudaTracegenAir for GlobalChip`,
 * /root/reference/sp1-gpu/crates/sys/lib/tracegen/riscv/global.cu:L1-L252; CPU definition
 * /root/reference/crates/core/machine/src/global/mod.rs:L131-L260): events = `GlobalInteractionEvent { message: [u32; 8],
 * is_receive: bool, kind: u8 }` as 9 u32 words each (word 8 = is_receive | kind << 8, the `#[repr(C)]` image); writes the
 * column-major [241][height] table: message, limbs, the Poseidon2 permutation's 179 columns, the lifted curve point, the running
 * digest sum (a scan over the septic curve's group law) and the reference's padding rows. Synchronises the stream before it
 * returns (it reports an event without a curve point as an error). */
int sp1hip_tracegen_riscv_global(uint32_t* d_trace, uint64_t height, const uint32_t* d_events, uint64_t n_events,
                                 sp1hip_stream_t stream);

/* Device trace generation for the RISC-V instruction chips of a core shard (round 6): Add, Addi, Sub, Addw, Subw, Mul, ShiftRight,
 * Branch — `generate_trace_into` / `event_to_row` of /root/reference/crates/core/machine/src/alu/{add_sub,addw,subw,mul,sr}/ and
 * control_flow/branch/, the R / I / ALU register adapters and CPUState they share (adapter/{state,register}, memory/consistency/trace.rs).
 * The reference fills these tables on the host and copies them (sp1-gpu/crates/jagged_tracegen/src/lib.rs:L819-L835); here a row
 * is made from an 88-byte event on the device. An event is the instruction's `AluEvent` / `BranchEvent` with its register access
 * records (core/executor/src/events/{alu,branch,memory}.rs), flattened:
 *   ops  = opcode | op_a << 8 | op_b << 16 | op_c << 24 | imm_b << 32 | imm_c << 33   (register numbers; Opcode as in opcode.rs)
 *   a / b / c = the operand values (a: the value written; c: the immediate when imm_c), a_prev = the value op_a held before
 *   (for a branch: the value of rs1), *_pts = the timestamp of the register's previous access, aux = next_pc (branches).
 * Writes the column-major [width][height] table in Montgomery words, rows >= n_events as the chip's padding rows. Clock carries
 * across 2^24 (MemoryBump / StateBump rows) stay with the caller: they are a handful of rows per shard. */
typedef struct { uint64_t pc, clk, ops, a, b, c, a_prev, a_pts, b_pts, c_pts, aux; } sp1hip_rv64_alu_event_t;
typedef enum { SP1HIP_RV64_CHIP_ADD = 0, SP1HIP_RV64_CHIP_ADDI = 1, SP1HIP_RV64_CHIP_SUB = 2, SP1HIP_RV64_CHIP_ADDW = 3, SP1HIP_RV64_CHIP_SUBW = 4,
               SP1HIP_RV64_CHIP_MUL = 5, SP1HIP_RV64_CHIP_SHIFT_RIGHT = 6, SP1HIP_RV64_CHIP_BRANCH = 7 } sp1hip_rv64_chip;
int sp1hip_tracegen_riscv_alu_width(int chip);       /* columns of the chip's table; -1 for an unknown chip */
int sp1hip_tracegen_riscv_alu(int chip, uint32_t* d_table, uint32_t height, const sp1hip_rv64_alu_event_t* d_events, uint32_t n_events,
                              sp1hip_stream_t stream);

/* ---------------------------------------------------------------- guest execution (host code, no device work)
 * An rv64im executor for SP1 guest ELFs: what `MinimalExecutor` + `TracingVM` do for the core prover
 * (/root/reference/crates/core/executor/src/minimal.rs, tracing.rs:L57-L147, vm.rs:L139-L431; the controller's shard loop is
 * /root/reference/crates/prover/src/worker/controller/core.rs:L231). It runs the program one shard of at most `max_cycles`
 * instructions at a time and hands back the events the chips' trace generation starts from:
 *   events        [n_events][SP1HIP_RV64_EVENT_WORDS] u64 per executed instruction: pc, clk, opcode (the reference's `Opcode`
 *                 discriminant), op_a, op_b, op_c, flags (1 imm_b | 2 imm_c | 4 memory access), a, b, c, previous value of op_a,
 *                 previous timestamps of the op_a / op_b / op_c register accesses, memory address, previous timestamp and value of
 *                 the memory word, its new value, next pc, one spare word
 *   local memory  [n_local][5] u64 per address touched in the shard (registers are addresses 0..31): address, timestamp and
 *                 value before the shard's first access, timestamp and value after its last (`MemoryLocalEvent`,
 *                 events/memory.rs)
 *   keccak        [n_keccak][SP1HIP_RV64_KECCAK_WORDS] u64 per KECCAK_PERMUTE call: clk, state pointer, 25 x (previous timestamp,
 *                 word read), the 25 words written
 *   poseidon2     [n_poseidon2][SP1HIP_RV64_POSEIDON2_WORDS] u64 per POSEIDON2 call: clk, pointer, 8 x (previous timestamp, word
 *                 read), the 8 words written (at clk)
 *   sha_extend    [n][SP1HIP_RV64_SHA_EXTEND_WORDS]: clk, w_ptr, 48 steps x (4 x (previous timestamp, word read) for w[i-15], w[i-2],
 *                 w[i-16], w[i-7]; previous timestamp and value of w[i]; w[i] written), then 64 x (timestamp, value) before and after
 *   sha_compress  [n][SP1HIP_RV64_SHA_COMPRESS_WORDS]: clk, w_ptr, h_ptr, 8 x (previous timestamp, h word), 64 x (previous timestamp, w
 *                 word), the 8 h words written
 *   uint256       [n][SP1HIP_RV64_UINT256_WORDS]: clk, x_ptr, y_ptr, 4 x (previous timestamp, x word), 8 x (previous timestamp, y /
 *                 modulus word), the 4 x words written (x * y mod modulus; a zero modulus means 2^256)
 * After the program halts, sp1hip_rv64_global_memory lists every address the run touched as (address, initial value, final value,
 * final timestamp): the `MemoryGlobalInit` / `MemoryGlobalFinalize` events. Supervisor mode only; system calls: HALT, WRITE,
 * ENTER / EXIT_UNCONSTRAINED, COMMIT, COMMIT_DEFERRED_PROOFS, HINT_LEN, HINT_READ, KECCAK_PERMUTE, POSEIDON2, SHA_EXTEND, SHA_COMPRESS, UINT256_MUL, SECP256K1_ADD, SECP256K1_DOUBLE, and the families of sp1hip_rv64_precompile_events (any other stops the run with
 * SP1HIP_ERROR_RUNTIME and a message naming it). The pointers stay valid until the next call on the same handle. */
#define SP1HIP_RV64_EVENT_WORDS 20
#define SP1HIP_RV64_KECCAK_WORDS 77
#define SP1HIP_RV64_POSEIDON2_WORDS 26
#define SP1HIP_RV64_SHA_EXTEND_WORDS 786
#define SP1HIP_RV64_SHA_COMPRESS_WORDS 155
#define SP1HIP_RV64_UINT256_WORDS 31
#define SP1HIP_RV64_SECP_ADD_WORDS 43
#define SP1HIP_RV64_SECP_DOUBLE_WORDS 26
/* The other field / curve precompiles, one family per chip that proves them (sp1hip_rv64_precompile_events). Every event starts
 * [clk, first argument, second argument, system-call code]; then
 *   two-operand calls (curve additions, ED_ADD, Fp / Fp2 operations): n x (previous timestamp, x word), n x (previous timestamp,
 *     y word), the n x words written at clk + 1 (y is read at clk) — n = 8 / 12 words for a secp256r1 / bn254 / ed25519 or a
 *     bls12-381 point or Fp2 element, 4 / 6 for an Fp element;
 *   doublings: n x (previous timestamp, word), the n words written (at clk);
 *   ED_DECOMPRESS (second argument = the sign bit): 4 x (previous timestamp, x word before), 4 x (previous timestamp, y word read
 *     at pointer + 32), the 4 x words written at clk + 1;
 *   UINT256_ADD_CARRY / UINT256_MUL_CARRY: the pointers c, d, e (registers x12, x13, x14), those registers' previous timestamps,
 *     4 x (previous timestamp, word before) for each of a, b, c (read at clk, clk + 1, clk + 2), d, e (rewritten at clk + 3,
 *     clk + 4), the 4 d and 4 e words written (low and high half of a + b + c or a * b + c).
 * Operands must be reduced field elements; the affine formulas have no special cases (equal x in an addition, y = 0 in a
 * doubling and points off the curve whose denominators vanish stop the run with SP1HIP_ERROR_RUNTIME). */
#define SP1HIP_RV64_FAMILY_SECP256R1_ADD 0
#define SP1HIP_RV64_FAMILY_SECP256R1_DOUBLE 1
#define SP1HIP_RV64_FAMILY_BN254_ADD 2
#define SP1HIP_RV64_FAMILY_BN254_DOUBLE 3
#define SP1HIP_RV64_FAMILY_BLS12381_ADD 4
#define SP1HIP_RV64_FAMILY_BLS12381_DOUBLE 5
#define SP1HIP_RV64_FAMILY_BN254_FP 6
#define SP1HIP_RV64_FAMILY_BLS12381_FP 7
#define SP1HIP_RV64_FAMILY_BN254_FP2_ADDSUB 8
#define SP1HIP_RV64_FAMILY_BLS12381_FP2_ADDSUB 9
#define SP1HIP_RV64_FAMILY_BN254_FP2_MUL 10
#define SP1HIP_RV64_FAMILY_BLS12381_FP2_MUL 11
#define SP1HIP_RV64_FAMILY_ED_ADD 12
#define SP1HIP_RV64_FAMILY_ED_DECOMPRESS 13
#define SP1HIP_RV64_FAMILY_UINT256_OPS 14
#define SP1HIP_RV64_FAMILIES 15
typedef void* sp1hip_rv64_vm_t;
typedef struct {
    uint64_t shard;                      /* index of this shard in the run */
    uint64_t n_cycles;                   /* instructions executed in this shard */
    uint64_t n_events, n_local, n_keccak, n_poseidon2, n_sha_extend, n_sha_compress, n_uint256, n_secp256k1_add, n_secp256k1_double; /* n_events = n_cycles, or 0 when recording is off */
    uint64_t pc_start, next_pc;          /* `PublicValues::pc_start / next_pc` (HALT_PC = 1 after HALT) */
    uint64_t clk_start, clk_end;         /* `initial_timestamp / last_timestamp` */
    uint32_t halted, exit_code;
    uint32_t commit_syscall, commit_deferred_syscall;
    uint32_t committed_value_digest[8];  /* as set by COMMIT so far */
    uint32_t deferred_proofs_digest[8];
    uint64_t estimated_area, estimated_max_height;   /* the shard-cutting estimator's totals (0 without sp1hip_rv64_set_shard_limits) */
} sp1hip_rv64_shard_info_t;

/* Shard cutting by trace area, as the reference's executor does it (`ShapeChecker`, crates/core/executor/src/vm/shapes.rs:L27-L245;
 * thresholds `ShardingThreshold`, opts.rs:L12-L14, less the HALT allowance of splicing.rs:L413-L428): after every instruction
 * the estimate grows by the cost (columns) of the instruction's chip, by memory_local_cost + 2 global_cost per address first
 * touched in the shard, by syscall_core_cost + global_cost per call sent to a precompile shard, by 32 memory_bump_cost +
 * state_bump_cost when the clock's high limb moves; a shard ends when the estimate reaches element_threshold or a table
 * height_threshold rows — never between a COMMIT and the HALT. opcode_cost / opcode_chip are indexed by the opcode numbers of
 * sp1hip_rv64_events (chip: any numbering < 64 that groups the opcodes of one table); ALU / load instructions into x0 go to the
 * AluX0 / LoadX0 tables. fixed_area = the preprocessed tables (Program, Byte, Range). The caller owns the cost model: this
 * library does not know the chips. */
typedef struct {
    uint64_t element_threshold, height_threshold, fixed_area;
    uint64_t opcode_cost[64];
    uint32_t opcode_chip[64];
    uint64_t alu_x0_cost, load_x0_cost, memory_local_cost, global_cost, syscall_core_cost, memory_bump_cost, state_bump_cost;
} sp1hip_rv64_shard_limits_t;
int sp1hip_rv64_create(const uint8_t* elf, uint64_t elf_len, sp1hip_rv64_vm_t* out);
void sp1hip_rv64_destroy(sp1hip_rv64_vm_t vm);
/* One entry of the input stream (`SP1Stdin::write_slice`): what the guest's next HINT_LEN / HINT_READ pair consumes. */
int sp1hip_rv64_write_stdin(sp1hip_rv64_vm_t vm, const uint8_t* data, uint64_t len);
int sp1hip_rv64_run_shard(sp1hip_rv64_vm_t vm, uint64_t max_cycles, sp1hip_rv64_shard_info_t* info);
/* on = 0: the following shards run without keeping their instruction events (a rank that proves shard r of an execution runs
 * shards 0..r-1 this way); memory state, local memory events and public values are kept as always. */
int sp1hip_rv64_set_recording(sp1hip_rv64_vm_t vm, int on);
/* limits != NULL: sp1hip_rv64_run_shard also ends a shard where the estimator above says so (max_cycles still bounds it);
 * NULL: cycle counts only (the default). */
int sp1hip_rv64_set_shard_limits(sp1hip_rv64_vm_t vm, const sp1hip_rv64_shard_limits_t* limits);
const uint64_t* sp1hip_rv64_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_local_memory(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_keccak_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_poseidon2_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_sha_extend_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_sha_compress_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_uint256_events(sp1hip_rv64_vm_t vm);
const uint64_t* sp1hip_rv64_secp256k1_add_events(sp1hip_rv64_vm_t vm);     /* [n][43]: clk, p_ptr, q_ptr, 8 x (ts, p word), 8 x (ts, q word), 8 p words written */
const uint64_t* sp1hip_rv64_secp256k1_double_events(sp1hip_rv64_vm_t vm);  /* [n][26]: clk, p_ptr, 8 x (ts, p word), 8 p words written */
/* The shard's events of one SP1HIP_RV64_FAMILY_*: [n_events][words_per_event] u64, layouts above. */
int sp1hip_rv64_precompile_events(sp1hip_rv64_vm_t vm, uint32_t family, uint64_t* n_events, uint64_t* words_per_event, const uint64_t** data);
/* The transpiled program (`Program::instructions`): [n][6] u64 = opcode, op_a, op_b, op_c, imm_b, imm_c; instruction i sits at
 * pc_base + 4 i. */
int sp1hip_rv64_program(sp1hip_rv64_vm_t vm, uint64_t* pc_base, uint64_t* n_instructions, const uint64_t** table);
int sp1hip_rv64_global_memory(sp1hip_rv64_vm_t vm, uint64_t* n, const uint64_t** table);
/* The program's memory image (`Program::memory_image`, /root/reference/crates/core/executor/src/disassembler/elf.rs:L320-L353):
 * [n][2] u64 = (address, value) of every 8-byte word of the ELF's PT_LOAD segments, file bytes and zero fill alike, by ascending
 * address. The reference initialises these words through the verifying key (`initial_global_cumulative_sum`,
 * core/executor/src/program.rs:L170-L199: no MemoryGlobalInit row) and finalises every one of them, touched or not
 * (prover/src/worker/controller/global.rs:L145-L300). */
int sp1hip_rv64_memory_image(sp1hip_rv64_vm_t vm, uint64_t* n, const uint64_t** table);
/* which = 0: the bytes written to the public-values descriptor; 1: stdout / stderr. */
int sp1hip_rv64_output(sp1hip_rv64_vm_t vm, int which, const uint8_t** data, uint64_t* len);

/* ---------------------------------------------------------------- the AirProver slot: setup / proving key
 * `MachineVerifyingKey` (/root/reference/crates/hypercube/src/verifier/config.rs:L71-L81), Montgomery words. The
 * septic digest is x[7] then y[7]. */
typedef struct {
    uint32_t pc_start[3];
    uint32_t initial_global_cumulative_sum[14];
    uint32_t preprocessed_commit[8];
    uint32_t enable_untrusted_programs;
} sp1hip_vk_t;
typedef struct sp1hip_pk_s sp1hip_pk_t;

/* `ShardProver::setup_from_preprocessed_data_and_traces` (/root/reference/crates/hypercube/src/prover/shard.rs:L406-L429),
 * the body of `AirProver::setup_from_vk` once the preprocessed traces exist: commits them (one jagged round) and returns
 * the proving key = that commitment round (`PreprocessedData`) + the verifying key it defines. `preprocessed_tables`:
 * the column-major device tables of the chips that have preprocessed columns, in chip (name) order; they stay
 * caller-owned and must outlive the key. */
int sp1hip_setup(const sp1hip_table_t* preprocessed_tables, int n_tables, const uint32_t pc_start[3],
                 const uint32_t initial_global_cumulative_sum[14], uint32_t enable_untrusted_programs,
                 sp1hip_shard_params_t params, sp1hip_pk_t** out, sp1hip_stream_t stream);
void sp1hip_pk_free(sp1hip_pk_t* pk);
int sp1hip_pk_vk(const sp1hip_pk_t* pk, sp1hip_vk_t* out);
/* `MachineVerifyingKey::observe_into` (config.rs:L97-L112): commit, pc_start, septic x / y, untrusted flag, 6 zeros. */
int sp1hip_vk_observe_into(const sp1hip_vk_t* vk, sp1hip_challenger_t* challenger);
/* `AirProver::prove_shard_with_pk` (shard.rs:L321-L345) from the generated traces on: a default challenger absorbs the
 * verifying key, then `sp1hip_prove_shard`. `pow_witnesses` (canonical words, may be NULL / 0): see
 * sp1hip_challenger_inject_pow_witnesses. */
int sp1hip_prove_shard_with_pk(const sp1hip_pk_t* pk, const sp1hip_shard_chip_t* chips, int n_chips, const uint32_t* h_publics,
                               int n_publics, const uint32_t* pow_witnesses, int n_pow_witnesses, uint8_t* h_proof, size_t* proof_len,
                               sp1hip_stream_t stream);

/* ---------------------------------------------------------------- prover pool: N shard proofs in flight per GPU
 * The library-side counterpart of the reference's `ProverSemaphore`
 * (/root/reference/crates/hypercube/src/prover/permits.rs:L36-L66; its GPU worker builder takes ONE permit,
 * /root/reference/sp1-gpu/crates/prover_components/src/builder.rs:L107) plus the pinned `trace_buffers` queue of its shard
 * prover: `n_slots` prover slots (a host thread + a stream each) prove shards concurrently and fill each other's
 * transcript hand-over gaps, while one stager thread uploads the host traces of the next shards
 * (`sp1hip_stage_tables`). Tickets are served in submission order. */
typedef struct sp1hip_pool_s sp1hip_pool_t;
typedef uint64_t sp1hip_ticket_t;

/* One chip of a shard for the pool: sp1hip_shard_chip_t with the main trace EITHER on the host (`h_main`: row-major
 * [real_rows][main_width] Montgomery words — pinned memory for a full-rate asynchronous copy — which the pool stages into
 * a column-major device table it owns) OR already resident (`d_main`, column-major). `d_prep`: the chip's preprocessed
 * device table (the one the proving key was set up from), or NULL. Everything the struct points to stays caller-owned and
 * must remain valid and unchanged until the ticket has been waited for. */
typedef struct {
    const char* name;
    const uint32_t* program;
    uint32_t n_instr, num_constraints;
    const uint32_t* interactions;
    uint32_t n_words;
    uint32_t main_width, prep_width;
    const uint32_t* h_main;
    const uint32_t* d_main;
    const uint32_t* d_prep;
    uint64_t real_rows;
} sp1hip_pool_chip_t;

/* Where a ticket spent its time (milliseconds) and which slot proved it. */
typedef struct {
    double staging_ms;  /* submit -> host traces enqueued for upload (includes waiting for a staging buffer) */
    double queued_ms;   /* staged -> a prover slot took it */
    double proving_ms;  /* sp1hip_prove_shard_with_pk on the slot's stream */
    int slot;
} sp1hip_pool_times_t;

int sp1hip_pool_create(int device, int n_slots, sp1hip_pool_t** out);
/* Finishes every submitted shard, then stops the threads and destroys the streams (the slots' helper streams, their events and
   every arena block cached for them included). Like any destructor of this ABI it must not run concurrently with another
   call on the same pool: a thread blocked in sp1hip_pool_wait holds a pointer into it. */
void sp1hip_pool_destroy(sp1hip_pool_t* pool);
/* Queue one shard: `AirProver::prove_shard_with_pk` (/root/reference/crates/hypercube/src/prover/shard.rs:L321-L345) from
 * the generated traces on. Returns at once. */
int sp1hip_pool_submit(sp1hip_pool_t* pool, const sp1hip_pk_t* pk, const sp1hip_pool_chip_t* chips, int n_chips,
                       const uint32_t* h_publics, int n_publics, sp1hip_ticket_t* ticket);
/* Block until the ticket's proof exists and copy bincode(ShardProof) out (size protocol as sp1hip_prove_shard:
 * SP1HIP_ERROR_BUFFER_TOO_SMALL sets *proof_len and leaves the ticket claimable). A failed proof returns its status and
 * message here. `times` may be NULL. A ticket can be collected once. */
int sp1hip_pool_wait(sp1hip_pool_t* pool, sp1hip_ticket_t ticket, uint8_t* h_proof, size_t* proof_len, sp1hip_pool_times_t* times);
/* The same without blocking: SP1HIP_ERROR_NOT_READY while the shard is in flight. */
int sp1hip_pool_try_wait(sp1hip_pool_t* pool, sp1hip_ticket_t ticket, uint8_t* h_proof, size_t* proof_len, sp1hip_pool_times_t* times);

#ifdef __cplusplus
}
#endif
#endif /* SP1HIP_H */
