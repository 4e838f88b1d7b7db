/*
 * pygsd_hip.h -- C-ABI of libpygsd_hip.so: the MI355X (gfx950) implementation of the sparse
 * message-passing hot path of SherylHYX/pytorch_geometric_signed_directed.
 *
 * The reference has NO native layer and NO FFI: every conv layer calls
 * torch_geometric.nn.conv.MessagePassing.propagate (gather -> scale -> scatter-reduce) from
 * Python.  Each entry point below names the reference call site(s) it replaces
 * (paths relative to torch_geometric_signed_directed/).  Plain pointers and sizes only:
 * device pointers are raw HIP device addresses, `stream` is a hipStream_t passed as void*.
 * No torch types cross this boundary.  The host-side binding a reference maintainer would add
 * is the ctypes stub shown in INTEGRATION.md.
 *
 * Conventions
 *   - every function returns 0 on success, nonzero on failure; pygsd_last_error() then holds a
 *     thread-local message (the Python host raises RuntimeError with it).
 *   - all work is enqueued asynchronously on `stream`; nothing synchronises the device except
 *     pygsd_prof_collect.
 *   - CSR indices are int32 (n_rows < 2^31, nnz < 2^31 per shard); feature matrices are
 *     row-major with a row stride `ld*` given in ELEMENTS (so column slices are addressable).
 *   - the library is stateless apart from the thread-local error string and the optional
 *     kernel-timing recorder (pygsd_prof_*).
 */
#ifndef PYGSD_HIP_H
#define PYGSD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PYGSD_ABI_VERSION 17

/* ABI version of the loaded library (== PYGSD_ABI_VERSION it was built with). */
int pygsd_version(void);
/* Message of the last failing call on this thread ("" if none). */
const char* pygsd_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * SpMM / segment-reduce over a CSR grouped by OUTPUT row.
 *
 *   Y[r, :] = alpha * scale_r * sum_{e = rowptr[r] .. rowptr[r+1]-1} val[e] * X[col[e], :]
 *             + beta * Z[r, :]
 *   scale_r = 1                               (mean == 0)
 *           = 1 / max(rowptr[r+1]-rowptr[r],1) (mean != 0)
 *   val == NULL means all ones; Z == NULL means no beta term; ldz == 0 broadcasts ONE row of Z to every output
 *   row (a bias vector added in the epilogue: DiGCNConv.update, nn/directed/DiGCNConv.py:90-93).
 *
 * Replaces MessagePassing.propagate(edge_index, x=..., norm=|edge_weight=...) with
 * message() = norm.view(-1,1) * x_j and aggr in {add, mean}:
 *   nn/directed/MagNetConv.py:196-240 (message :251), nn/general/MSConv.py:193-221 (:233),
 *   nn/directed/DiGCNConv.py:86 (:88), nn/directed/DGCNConv.py:95 (:99),
 *   nn/general/conv_base.py:111 (:116), nn/signed/SGCNConv.py:101-119 (:128, aggr='mean' :73).
 * alpha/beta/Z fuse the Chebyshev recurrence T_k = 2 S T_{k-1} - T_{k-2}
 *   (MagNetConv.py:216,222,228,234).
 * The same entry point computes the backward dX = S^T dY when handed the CSR grouped by source.
 * Deterministic: no atomics; the summation order inside a row is fixed by the CSR order.
 * nnz_hint: total number of CSR entries if the caller knows it (0 = unknown).  Tuning only: rows with
 * >= 48 entries on average (>= 28 in the dual-operator kernel) run a variant with deeper gather
 * pipelining, sparser ones (and unknown) a low-register variant with twice the wavefront occupancy --
 * bit-identical results; rows with fewer than ~4-10 entries on average at widths <= 128 (signed SBM parts)
 * run a rows-per-wavefront variant in which every lane group owns its own row and sums it sequentially in
 * CSR order (the reference's scatter order): equal to the other variants to fp32 rounding, deterministic.
 * long_rows (may be NULL): hub rows.  One wavefront owns one output row, so a row with 10^5..10^6 entries
 * (power-law graphs; the reference's scatter has no such cliff) would serialise the launch.  The caller
 * lists the rows with MORE than PYGSD_LONG_ROW entries; the row-per-wavefront kernel skips them and a
 * second pair of kernels reduces them in 4096-entry segments, adding the segment partials in segment
 * order (deterministic, no atomics; the summation ORDER of a hub row differs from the single-wavefront
 * order, i.e. results agree to fp32 rounding, not bitwise, with long_rows = NULL).  Rows listed must
 * really exceed PYGSD_LONG_ROW entries (others would be computed twice, harmlessly); rows above the
 * threshold that are NOT listed are left unwritten.
 * ------------------------------------------------------------------------------------------- */
#define PYGSD_LONG_ROW 4096
typedef struct pygsd_long_rows {
    const int32_t* rows;      /* device: ids of the rows with > PYGSD_LONG_ROW entries                 */
    int32_t n_rows;           /* how many                                                             */
    int32_t max_entries;      /* entries of the longest of them                                       */
    void* workspace;          /* device scratch, pygsd_spmm_long_rows_workspace() bytes               */
    int64_t workspace_bytes;
} pygsd_long_rows;

/* K = 1 magnetic layer forward with 64 input and 64 output features (the north-star shape of MagNetConv.py:189-247), the dense
 * stage in the dual SpMM's epilogue:  Ta = S_a X_a, Tb = S_b X_b (both written: the backward's operands), and
 *   out_r = (Xa - Xb) W[0] + (Ta - Tb) W[1] + bias ,  out_i = (Xa + Xb) W[0] + (Ta + Tb) W[1] + bias
 * with W [2][64][64] row-major and bias [64] or NULL -- what pygsd_spmm2_csr_f32 followed by pygsd_magnetic_dense_fwd_f32
 * compute, without re-reading T from HBM.  Rows longer than PYGSD_LONG_ROW are not split here: callers with hub rows keep the
 * two-call form.  Optional path (dense.set_fused_k1 / PYGSD_FUSE_K1), see DESIGN.md for the measurement. */
int pygsd_spmm2_k1_dense_f32(const int32_t* rowptr, const int32_t* col, const float* val_a, const float* val_b, const float* Xa,
                             const float* Xb, int64_t ldx, float* Ta, float* Tb, int64_t ldt, const float* W, const float* bias,
                             float* out_r, float* out_i, int64_t ldo, int32_t n_rows, int64_t nnz_hint, void* stream);
int pygsd_spmm_long_rows_workspace(int32_t n_long, int32_t max_entries, int32_t n_feat, int32_t dual,
                                   int64_t* bytes);

int pygsd_spmm_csr_f32(const int32_t* rowptr, const int32_t* col, const float* val,
                       const float* X, int64_t ldx,
                       float* Y, int64_t ldy,
                       const float* Z, int64_t ldz,
                       int32_t n_rows, int32_t n_feat,
                       float alpha, float beta, int32_t mean, int64_t nnz_hint,
                       const pygsd_long_rows* long_rows, void* stream);

/* bf16-storage variant (BASELINE config "DiGCN_Inception_Block ... bf16"): X, Y, Z are bf16 row-major
 * (n_feat and the row strides multiples of 8, 16-byte aligned), edge values and accumulation fp32,
 * the result rounded to nearest-even bf16.  The reference has no reduced-precision path; parity is
 * defined against the fp32 oracle evaluated on the bf16-rounded inputs. */
int pygsd_spmm_csr_bf16(const int32_t* rowptr, const int32_t* col, const float* val,
                        const void* X, int64_t ldx,
                        void* Y, int64_t ldy,
                        const void* Z, int64_t ldz,
                        int32_t n_rows, int32_t n_feat,
                        float alpha, float beta, int32_t mean,
                        void* stream);
/* Same gather (bf16 X, fp32 values), but Y and Z are FLOAT arrays (ldy / ldz in floats): the partial products of a
 * column-phased product (parallel.PropagateEngine) accumulate in fp32 through beta * Z with Z = Y, and the caller
 * rounds the finished rows to bf16 once instead of once per phase.  No reference counterpart. */
int pygsd_spmm_csr_bf16_acc_f32(const int32_t* rowptr, const int32_t* col, const float* val, const void* X,
                                int64_t ldx, float* Y, int64_t ldy, const float* Z, int64_t ldz,
                                int32_t n_rows, int32_t n_feat, float alpha, float beta, void* stream);

/* Two operators sharing ONE sparsity pattern, two inputs, two outputs, one traversal:
 *   Ya = alpha * sum val_a[e] * Xa[col[e]] + beta * Za ;  Yb likewise with val_b / Xb / Zb.
 * This is the complex Hermitian (magnetic) Laplacian product of MagNetConv / MSConv: the real
 * and imaginary operators (MagNetConv.py:196-203; duplicates :204-211) walk the same pattern.
 * Xa/Xb share ldx, Ya/Yb share ldy, Za/Zb share ldz. */
int pygsd_spmm2_csr_f32(const int32_t* rowptr, const int32_t* col,
                        const float* val_a, const float* val_b,
                        const float* Xa, const float* Xb, int64_t ldx,
                        float* Ya, float* Yb, int64_t ldy,
                        const float* Za, const float* Zb, int64_t ldz,
                        int32_t n_rows, int32_t n_feat,
                        float alpha, float beta, int64_t nnz_hint,
                        const pygsd_long_rows* long_rows, void* stream);

/* Per-edge gradient of the edge values (SDDMM): out[e] = < A[ia[e], :], B[ib[e], :] >.
 * Backward of message() w.r.t. `norm` (needed by trainable_q, MagNetConv.py:58-59,141-142, and by
 * any edge_weight that requires grad). */
int pygsd_sddmm_coo_f32(const int32_t* ia, const int32_t* ib, int64_t nnz,
                        const float* A, int64_t lda, const float* B, int64_t ldb,
                        int32_t n_feat, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention aggregate of SDGNN / SiGAT: what torch_geometric.nn.GATConv computes for the reference at
 * nn/signed/SDGNN.py:35-41,57-64 and nn/signed/SiGAT.py:59-64 (heads handled one at a time by the host).
 *   pygsd_gat_alpha_csr_f32    : alpha[e] = softmax over CSR row r of leaky_relu(a_src[col[e]] + a_dst[r])
 *                                (max-shifted exp, denominator + 1e-16), CSR order.  The weighted sum
 *                                out = sum alpha * h[col] is then pygsd_spmm_csr_f32 with val = alpha.
 *   pygsd_gat_alpha_bwd_csr_f32: ds[e] = alpha[e] * (d_e - sum_k alpha[k] d_k) * lrelu'(s_e) with d_e = <g_r, h_col[e]>
 *                                and the sum over row r -- the backward of torch_geometric.utils.softmax in its
 *                                own form, so that a one-entry row gives exactly 0 and row sums cancel as the
 *                                reference's do (round 6; before: <g_r, out_r> for the sum, equal in exact
 *                                arithmetic only.  `out` is still read for rows handled by the long_rows path);
 *                                the gradient w.r.t. the pre-activation score s_e = a_src[col] + a_dst[r];
 *                                written, together with alpha, in COO order through perm (so that
 *                                d a_src / d a_dst are row sums over the two CSR orientations).
 * long_rows (may be NULL; every entry point of this section and the segment / SNEA ones below take it): hub
 * rows.  These kernels give one wavefront (16 lanes for the segment sum) to a row, which serialises a launch on a
 * row with 10^5..10^6 entries (power-law graphs: SDGNN / SiGAT motif lists, nn/signed/SDGNN.py:198-254; the
 * reference's scatter has no such cliff).  The caller lists the rows with MORE than PYGSD_LONG_ROW entries; the
 * row kernels skip them and a segment-parallel path handles them: per 4096-entry segment the softmax statistics
 * (max, sum exp) resp. the backward's row dot, folded per row IN SEGMENT ORDER, then the per-entry outputs and
 * per-segment partial row sums, again added in segment order -- no atomics, run-to-run deterministic; results
 * of hub rows agree with the single-wavefront order to fp32 rounding, all other rows bitwise.  Workspace:
 * pygsd_segment_long_rows_workspace() bytes in long_rows->workspace.
 * ------------------------------------------------------------------------------------------- */
int pygsd_segment_long_rows_workspace(int32_t n_long, int32_t max_entries, int64_t* bytes);
int pygsd_gat_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const float* a_src, const float* a_dst,
                            int32_t n_rows, float negative_slope, float* alpha,
                            const pygsd_long_rows* long_rows, void* stream);
int pygsd_gat_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const int32_t* perm,
                                const float* a_src, const float* a_dst, float negative_slope,
                                const float* alpha, const float* h, int64_t ldh, const float* g, int64_t ldg,
                                const float* out, int64_t ldo, int32_t n_rows, int32_t n_feat,
                                float* ds_coo, float* alpha_coo, const pygsd_long_rows* long_rows, void* stream);

/* Vectorised form of the above for n_feat % 4 == 0, n_feat <= 256, 16-byte aligned rows: float4 gathers,
 * ds written in CSR (by-target) order and its per-row sum da_dst = gradient of a_dst produced in the same
 * pass.  The by-source sums (gradient of a_src) and the backward aggregate then read ds / alpha through the
 * by-source -> by-target slot map of the pattern (pygsd_segment_sum_f32 / pygsd_gather_f32). */
int pygsd_gat_alpha_bwd_csr_v2_f32(const int32_t* rowptr, const int32_t* col, const float* a_src,
                                   const float* a_dst, float negative_slope, const float* alpha,
                                   const float* h, int64_t ldh, const float* g, int64_t ldg,
                                   const float* out, int64_t ldo, int32_t n_rows, int32_t n_feat,
                                   float* ds_csr, float* da_dst, const pygsd_long_rows* long_rows, void* stream);

/* out[r] = sum over CSR row r of w[perm[slot]] (perm == NULL: w[slot]).  Unlike pygsd_csr_row_sum_f32 (one
 * sequential sum per row, the reference's scatter order, used for degrees) this one splits a row over 16 lanes:
 * fixed but different summation order, for gradient reductions. */
int pygsd_segment_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* w, int32_t n_rows,
                          float* out, const pygsd_long_rows* long_rows, void* stream);

/* Segment softmax over per-entry logits given in CSR order (one segment = one CSR row), max-shifted with
 * the + 1e-16 denominator of torch_geometric.utils.softmax: the attention of SNEAConv
 * (nn/signed/SNEAConv.py:135-146: alpha = softmax(tanh(alpha_func([x_j, x_i])), index)), whose logits mix
 * two feature sets per edge type and are therefore formed by the caller.
 * Backward: dlogits = alpha * (dalpha - sum_segment alpha * dalpha). */
int pygsd_segment_softmax_csr_f32(const int32_t* rowptr, const float* logits, int32_t n_rows, float* alpha,
                                  const pygsd_long_rows* long_rows, void* stream);
int pygsd_segment_softmax_bwd_csr_f32(const int32_t* rowptr, const float* alpha, const float* dalpha,
                                      int32_t n_rows, float* dlogits, const pygsd_long_rows* long_rows, void* stream);

/* SNEAConv's attention, fused (nn/signed/SNEAConv.py:70-146).  Slot e of target row i has source col[e] and
 * edge_type[e] in {0: positive edge or self loop, 1: negative edge} (edge_type == NULL: all 0, s1/d1/share1 and
 * the other type-1 arrays may be NULL).  pre_e = s_t[col[e]] + d_t[i] + bias[0] (bias: device scalar, NULL = 0), alpha = softmax over the row of
 * tanh(pre) (max-shifted, + 1e-16).  Because the reference's message is the TARGET row times alpha, the
 * aggregate is out_i = x1_i * share0[i] + x2_i * share1[i] with share_t[i] = sum of alpha over the row's type-t
 * slots; the caller forms s_t = <x_t, a_src>, d_t = <x_t, a_dst> and that final product.
 * Backward: from dshare_t it returns dpre per type in CSR order (zero where the slot has the other type;
 * summing them by source gives ds_t) and dd_t[i] = per-row sums (= the gradient of d_t and, summed, of bias). */
int pygsd_snea_alpha_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                             const float* s0, const float* s1, const float* d0, const float* d1,
                             const float* bias, int32_t n_rows, float* alpha, float* share0, float* share1,
                             const pygsd_long_rows* long_rows, void* stream);
int pygsd_snea_alpha_bwd_csr_f32(const int32_t* rowptr, const int32_t* col, const uint8_t* edge_type,
                                 const float* s0, const float* s1, const float* d0, const float* d1,
                                 const float* bias, const float* alpha, const float* dshare0,
                                 const float* dshare1, int32_t n_rows,
                                 float* dpre0, float* dpre1, float* dd0, float* dd1,
                                 const pygsd_long_rows* long_rows, void* stream);

/* ---------------------------------------------------------------------------------------------
 * COO -> CSR (operator build).  Groups the nnz entries by seg[e] (stable: entries of one group
 * keep their COO order, which is the order torch's scatter_add_ sums them in the reference) and
 * emits rowptr[n_seg+1], col[e'] = (int32) other[perm[e']], perm[e'] = original entry id.
 * Replaces the implicit grouping done by scatter(..., index=edge_index[i]) inside
 * MessagePassing.propagate.  seg/other are the int64 rows of the caller's edge_index.
 * Workspace: call pygsd_csr_from_coo_workspace first; pass a device buffer of that many bytes.
 * ------------------------------------------------------------------------------------------- */
int pygsd_csr_from_coo_workspace(int64_t nnz, int32_t n_seg, size_t* bytes);
int pygsd_csr_from_coo(const int64_t* seg, const int64_t* other, int64_t nnz, int32_t n_seg,
                       int32_t* rowptr, int32_t* col, int32_t* perm,
                       void* workspace, size_t workspace_bytes, void* stream);

/* out[i] = src[perm[i]]  (re-order edge values into CSR order). */
int pygsd_gather_f32(const float* src, const int32_t* perm, int64_t n, float* out, void* stream);

/* Stable sort of (key, payload = entry id) pairs by a 64-bit key using only bits
 * [0, key_bits): the sort inside torch_geometric.utils.coalesce
 * (utils/directed/get_magnetic_Laplacian.py:60, utils/general/get_magnetic_signed_Laplacian.py:64). */
int pygsd_sort_keys_u64_workspace(int64_t n, size_t* bytes);
int pygsd_sort_keys_u64(const uint64_t* keys_in, uint64_t* keys_out, int32_t* perm_out,
                        int64_t n, int32_t key_bits,
                        void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * (Signed) magnetic Laplacian build on the device -- utils/directed/get_magnetic_Laplacian.py:47-85,
 * utils/general/get_magnetic_signed_Laplacian.py:47-90 (called from MagNetConv.__norm__ :100-103 /
 * MSConv.__norm__ :99-102 on EVERY forward unless cached=True).
 *   pygsd_maglap_sort  : drop self loops, symmetrise (cat [row,col],[col,row]), stable sort by
 *                        (row, col), mark duplicate runs; writes the number E_s of distinct entries to
 *                        the DEVICE int64 *d_num_unique (the caller reads it back to size the outputs --
 *                        the same host round-trip coalesce's boolean mask costs the reference).
 *   pygsd_maglap_merge : coalesce(add): per distinct (row, col) in sorted order A_s = sum(w)/2,
 *                        Theta_arg = sum(+-w), (|w| sums); deg = row sums of A_s (unsigned), of the |w|
 *                        sums (signed, absolute_degree) or of |A_s| (signed, not absolute_degree).
 *                        w == NULL means all ones.  Outputs: out_row/out_col int64[E_s], a_sym, theta
 *                        float[E_s], deg float[n], off_ptr int32[n+1] (first sorted entry of each row).
 *   pygsd_maglap_values: off-diagonal values of L: sym != 0: -(D^-1/2 A_s D^-1/2 (.) exp(i 2 pi q Theta_arg))
 *                        (diagonal is 1); sym == 0: -(A_s (.) exp(...)) (diagonal is deg).
 *                        mir_real / mir_imag (optional, both or neither): the value of the MIRRORED entry
 *                        (col, row) -- the operator is Hermitian, so it is the same magnitude multiplied in
 *                        the mirrored entry's own row/col order with the conjugate phase.
 *   pygsd_maglap_assemble_csr: compute layout of the scaled operator S = 2 L / lambda_max - I: ONE int32
 *                        CSR (rowptr[n+1], col[E_s+n]) over the symmetric pattern incl. the diagonal
 *                        (columns ascending), shared by both orientations, with vb_* = S[row, col]
 *                        (backward / by-source product) and vf_* = S[col, row] (forward / by-target
 *                        product); v = (2 x) / lambda_max with +inf -> 0 (MagNetConv.py:106-107,115-116),
 *                        diagonal = (2 diag) / lambda_max + diag_shift (the folded -1 self loops, :108-112).
 *                        No further sorts or gathers.
 * The same workspace (pygsd_maglap_workspace bytes) must be passed, untouched, to sort and merge.
 * ------------------------------------------------------------------------------------------- */
int pygsd_maglap_workspace(int64_t n_edges, size_t* bytes);
int pygsd_maglap_sort(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n,
                      void* workspace, size_t workspace_bytes, int64_t* d_num_unique, void* stream);
int pygsd_maglap_merge(const float* w, int64_t n_edges, int32_t n, int32_t is_signed, int32_t absolute_degree,
                       int64_t num_unique, void* workspace, size_t workspace_bytes,
                       int64_t* out_row, int64_t* out_col, float* a_sym, float* theta, float* deg,
                       int32_t* off_ptr, void* stream);
int pygsd_maglap_assemble_csr(const int64_t* out_row, const int64_t* out_col, const float* off_real,
                              const float* off_imag, const float* mir_real, const float* mir_imag,
                              const float* diag, const int32_t* off_ptr, int64_t num_unique, int32_t n,
                              float lambda_max, float diag_shift, int32_t* rowptr, int32_t* col,
                              float* vb_real, float* vb_imag, float* vf_real, float* vf_imag, void* stream);
int pygsd_maglap_values(const int64_t* out_row, const int64_t* out_col, const float* a_sym,
                        const float* theta, const float* deg, int64_t num_unique, double q, int32_t sym,
                        float* off_real, float* off_imag, float* mir_real, float* mir_imag, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused build of the scaled (signed) magnetic operator (csrc/magop.hip) -- the same result as
 * pygsd_maglap_sort -> _merge -> _values -> _assemble_csr in two stages around ONE host read, for the
 * case the layers run on every uncached forward (MagNetConv.py:157-181 / MSConv.py:150-180: fixed q,
 * no gradient w.r.t. edge_weight).  Edge list in, compute layout out; no int64 COO intermediates.
 *   pygsd_magop_stage1: range-check the node ids, emit both orientations as u64 keys, radix-sort on the
 *                       ROW bits only (stable), order / coalesce every row inside one wavefront (rows of
 *                       65..4096 symmetrised entries: one block, LDS), sum duplicates in sorted order and
 *                       the row degree sequentially in column order.  Writes rowptr int32[n+1] (final CSR
 *                       row pointer incl. the diagonal entry of every row), deg float[n] and the DEVICE
 *                       int64 d_info[4] = { E_s, #rows with more than 4096 symmetrised entries (if > 0 the
 *                       result is incomplete: use the pygsd_maglap_* pipeline), 1 if a node id was outside
 *                       [0, n), one such id }.  w == NULL means all ones.  sym != 0 also prepares deg^-1/2.
 *   pygsd_magop_stage2: S = 2 L / lambda_max + diag_shift I into col int32[E_s+n] and the four value arrays
 *                       float[E_s+n] (each 16-byte aligned): vb_* = S[row, col], vf_* = S[col, row] (see
 *                       pygsd_maglap_assemble_csr), columns ascending.  E_s need not be known to the host to
 *                       launch it: arrays of the upper bound 2 n_edges + n serve (rowptr[n] = E_s + n), so the
 *                       read of d_info can overlap this stage.
 * Both stages take the same, untouched workspace of pygsd_magop_workspace(n_edges, n, w != NULL) bytes.
 * ------------------------------------------------------------------------------------------- */
int pygsd_magop_workspace(int64_t n_edges, int32_t n, int32_t weighted, size_t* bytes);
int pygsd_magop_stage1(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                       int32_t is_signed, int32_t absolute_degree, int32_t sym, void* workspace,
                       size_t workspace_bytes, int32_t* rowptr, float* deg, int64_t* d_info, void* stream);
int pygsd_magop_stage2(int64_t n_edges, int32_t n, int32_t weighted, double q, int32_t sym, float lambda_max,
                       float diag_shift, void* workspace, size_t workspace_bytes, const int32_t* rowptr,
                       const float* deg, int32_t* col, float* vb_real, float* vb_imag, float* vf_real,
                       float* vf_imag, void* stream);
/* Round 5: for weighted graphs the bucket plan takes (see pygsd_magop_unit below: <= 2^25 nodes, <= 3000 buckets ...)
 * pygsd_magop_stage1 no longer sorts the stream globally: it is split once into buckets of consecutive rows that fit a workgroup's
 * LDS, the weights travelling in a second 4-byte stream; a workgroup per bucket groups it by row in LDS and writes the row-grouped
 * (key, weight) streams the sort used to produce -- minus the order inside a row -- and the rows are merged by the same kernels
 * as behind the sort -- same records, same results bit for bit WHERE THE ORDER OF ARRIVAL CANNOT SHOW: a neighbour's run of one or
 * two entries (a + b = b + a in fp32; the row degree is summed over the column-sorted distinct entries either way).  A run of three
 * or more entries of one neighbour (an edge listed three times, a reciprocal pair with a duplicate) or an over-full half bucket
 * are counted in d_info[1] (as are, in either form, rows of more than 4096 symmetrised entries); the caller then repeats the first stage with
 * pygsd_magop_stage1_sorted -- the radix sort on the row bits in front of the merge, rows of up to 4096 entries, duplicates summed
 * in (direction, list position) order like the reference -- same arguments, same workspace, and runs pygsd_magop_stage2 again.
 * PYGSD_WEIGHTED_BUILD_FORM=sort in the environment makes pygsd_magop_stage1 itself take the sorted form. */
int pygsd_magop_stage1_sorted(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                              int32_t is_signed, int32_t absolute_degree, int32_t sym, void* workspace,
                              size_t workspace_bytes, int32_t* rowptr, float* deg, int64_t* d_info, void* stream);
/* The UNWEIGHTED build in one call (round 4): edge list -> final CSR + the four value arrays.  With unit weights a node's
 * degree is half its number of symmetrised entries (known from the row bounds of the sorted stream), multiplicities and phase
 * arguments are small integers: the wavefront that orders a row parks ONE 8-byte record per distinct neighbour, and after the
 * scan of the distinct counts the same mapping writes values and diagonals into the final slots -- half the intermediate
 * traffic of pygsd_magop_stage1 + _stage2, bit-identical results (get_magnetic_Laplacian.py:47-85 with edge_weight = None,
 * as MagNetConv.py:157-181 calls it on every uncached forward).  Outputs as the two stages (rowptr [n + 1], deg [n], col / v*
 * allocated for 2 n_edges + n slots, 16-byte aligned); d_info[0] = E_s, d_info[1] != 0: a node has more than 512 symmetrised
 * entries (or, in the bucket form below, 512 consecutive rows hold more than 32 768) and the outputs are invalid -- the caller
 * then takes the two-stage pipeline; d_info[2], [3] as stage 1.  workspace from pygsd_magop_workspace(n_edges, n, 0).
 * Two forms with bit-identical outputs: by default the stream is never sorted globally -- it is split once into buckets of up to
 * 1024 consecutive rows that fit a workgroup's LDS and ordered there (graphs of <= 2^25 nodes, <= 3000 buckets, an average
 * bucket of <= 24 576 entries); other graphs, or PYGSD_UNIT_BUILD_FORM=sort in the environment, take a radix sort on the row
 * bits first.
 * phase: 0 = the whole build; 1 = everything up to the row pointer -- d_info is final when this part has run, so the caller can
 * queue its device -> host read of d_info here and then call again with phase = 2 (same arguments, untouched workspace) for the
 * kernel that writes the slots: the host then learns the sizes while 40 % of the build is still running, instead of after it. */
int pygsd_magop_unit(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n, int32_t sym, double q, float lambda_max,
                     float diag_shift, void* workspace, size_t workspace_bytes, int32_t* rowptr, float* deg, int32_t* ccol,
                     float* vb_real, float* vb_imag, float* vf_real, float* vf_imag, int64_t* d_info, int32_t phase, void* stream);
/* The same one-call build for weights that are all +1 or -1 (round 5) -- the signed graphs of MSConv / MSGNN
 * (get_magnetic_signed_Laplacian.py:52-90 as MSConv.py:78-119 calls it on every uncached forward: SDSBM / SSBM signs, BASELINE
 * config 4) and edge lists that carry explicit unit weights.  Sums of +-1 are small integers, exact in every order of summation,
 * so the unordered LDS placement of the bucket form serves here as it does for unit weights, with ONE extra bit per stream entry
 * (the sign, below the direction bit) and the merged record also carrying the run's number of negative entries:
 * A_s = (entries - 2 negative) / 2, Theta_arg = sum of +w (forward) / -w (reversed), bit-identical to the two stages and to the
 * pygsd_maglap_* pipeline.  -1 is admitted where the degree counts |w| (is_signed && absolute_degree: deg = entries / 2, as with
 * unit weights); elsewhere only +1.  Not taken -- d_info[1] != 0, outputs invalid, the caller takes pygsd_magop_stage1 / _stage2
 * -- when a weight is anything else (validated on the device by the first kernel: no host read of the weights), when -1 meets
 * another degree convention, and for the graphs the bucket form does not take (see above; here <= 2^24 nodes).  Arguments,
 * workspace (pygsd_magop_workspace(n_edges, n, 0)), outputs, d_info and `phase` as pygsd_magop_unit. */
int pygsd_magop_unit_signed(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n, int32_t is_signed,
                            int32_t absolute_degree, int32_t sym, double q, float lambda_max, float diag_shift, void* workspace,
                            size_t workspace_bytes, int32_t* rowptr, float* deg, int32_t* ccol, float* vb_real, float* vb_imag,
                            float* vf_real, float* vf_imag, int64_t* d_info, int32_t phase, void* stream);

/* ---------------------------------------------------------------------------------------------
 * add_remaining_self_loops + degree normalisation: torch_geometric's gcn_norm as called at
 * nn/directed/DGCNConv.py:75 and conv_norm_rw of nn/general/conv_base.py:12-31.
 *   pygsd_self_loops_scan : flag non-loop edges, exclusive scan, remember the LAST listed loop of each
 *                           node (last_loop int32[n], -1 = none); *d_num_kept (device) = #non-loops.
 *   pygsd_self_loops_emit : out = non-loop edges in order, then n loops (v, v) whose weight is the
 *                           node's listed loop weight if it had one, else fill_value.  out_w == NULL
 *                           skips the weights; w == NULL means all ones.
 *   pygsd_csr_row_sum_f32 : deg[r] = sum of w[perm[slot]] over CSR row r (sequential, COO order); perm == NULL:
 *                           w is already in CSR order.
 *   pygsd_degree_scale_f32: mode 0: deg^-1/2[row] * w * deg^-1/2[col]; mode 1: deg^-1[row] * w
 *                           (inf -> 0 as masked_fill_ does).
 * ------------------------------------------------------------------------------------------- */
int pygsd_self_loops_workspace(int64_t n_edges, size_t* bytes);
int pygsd_self_loops_scan(const int64_t* row, const int64_t* col, int64_t n_edges, int32_t n,
                          void* workspace, size_t workspace_bytes, int32_t* last_loop,
                          int64_t* d_num_kept, void* stream);
int pygsd_self_loops_emit(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges, int32_t n,
                          float fill_value, int64_t num_kept, void* workspace, size_t workspace_bytes,
                          const int32_t* last_loop, int64_t* out_row, int64_t* out_col, float* out_w,
                          void* stream);
int pygsd_csr_row_sum_f32(const int32_t* rowptr, const int32_t* perm, const float* w, int32_t n_rows,
                          float* out, void* stream);
int pygsd_degree_scale_f32(const int64_t* row, const int64_t* col, const float* w, const float* deg,
                           int64_t n_edges, int32_t mode, float* out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Element-wise epilogues of the path.
 * complex ReLU: mask = (real >= 0); out_real = mask*real; out_imag = mask*imag
 *   (nn/directed/complex_relu.py:21-22).  In-place allowed (out == in). */
int pygsd_complex_relu_f32(const float* real, const float* imag, float* out_real, float* out_imag,
                           int64_t n, void* stream);
/* backward of the above: g_real_in = mask*g_real, g_imag_in = mask*g_imag (mask from `real`). */
int pygsd_complex_relu_bwd_f32(const float* real, const float* g_real, const float* g_imag,
                               float* gi_real, float* gi_imag, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused dense stage of MagNetConv / MSConv on the MFMA matrix cores (exact fp32 MFMA).
 *   forward : out_real = sum_k (A_k - B_k) W_k + bias ;  out_imag = sum_k (A_k + B_k) W_k + bias
 *   backward: P = g_real + g_imag, M = g_imag - g_real;
 *             dA_k = P W_k^T, dB_k = M W_k^T, dW_k = A_k^T P + B_k^T M, dbias = colsum(P)
 * A_k / B_k are the k-th Chebyshev terms of the real / imaginary chain ([n_rows, f_in], contiguous,
 * 16-byte aligned); W is [k1, f_in, f_out]; a / b / da / db are HOST arrays of k1 device pointers.
 * Replaces the four matmuls per Chebyshev order, the out_real = rr - ii / out_imag = ir + ri
 * combination and the in-place bias adds of nn/directed/MagNetConv.py:189-192,198-211,217-247
 * (nn/general/MSConv.py:185-230), and their autograd backward.
 * ldg (backward): row stride of g_real / g_imag in elements; 0 = ONE row of f_out values broadcast to every node
 * (the upstream gradient of a loss that sums the outputs over the nodes: nothing [N, f_out] is materialised).
 * pygsd_magnetic_dense_supported: 1 if (f_in, f_out, k1) is covered by the fused kernels
 * (multiples of 16, f_out in {16,32,48,64,128}, f_in < 64 or a multiple of 64, k1 <= 4).
 * Arithmetic (pygsd_dense_f32_form): two forms.  SPLIT, the default where a shape has it -- the forward at f_in = 64 / 128 with
 * f_out a multiple of 64, the backward at f_out = 64 / 128 with f_in a multiple of 64 (every MagNetConv / MSConv layer of hidden
 * width 64 or 128): fp32 operands as three bf16 pieces each, the six largest partial products per product on the bf16 matrix
 * pipe, fp32 accumulation (csrc/tall.hip).  Closer to the float64 result than an fp32 fmaf chain (forward: a fifth of the chain's
 * error relative to the sum of |terms| -- the partial products of a 32-feature block are summed apart and added to the running
 * sum once; backward dA / dB: a third of it), not bitwise any fp32 summation order.  EXACT -- an fmaf chain per output on
 * v_mfma_f32_16x16x4_f32 -- for every other shape and on request.
 * Non-finite values, both forms: +-inf, NaN and magnitudes up to FLT_MAX come out where the reference's sequence of fp32
 * torch.matmul's puts them (rr = sum A_k W_k, ii = sum B_k W_k, then rr - ii and rr + ii; their autograd).  The pieces of the
 * split form carry finite values below the largest bf16 (3.39e38) only; a 16-row tile (a weight-gradient element) whose sums
 * come out non-finite is computed again from the fp32 operands by fmaf chains in the reference's order of operations
 * (csrc/dense.hip: exact_rows, reduce_dw_checked_kernel; tests/test_gpu_nonfinite.py).
 * ------------------------------------------------------------------------------------------- */
int pygsd_magnetic_dense_supported(int32_t f_in, int32_t f_out, int32_t k1);
/* form: 0 = split where the shape allows (default; PYGSD_DENSE_F32=exact at load selects 1), 1 = exact everywhere, anything else
 * = query only.  Returns the form in force before the call.  Process-wide. */
int pygsd_dense_f32_form(int32_t form);
int pygsd_magnetic_dense_fwd_f32(const float* const* a, const float* const* b, int32_t k1,
                                 const float* w, const float* bias,
                                 float* out_real, float* out_imag,
                                 int32_t n_rows, int32_t f_in, int32_t f_out, void* stream);
int pygsd_magnetic_dense_bwd_workspace(int32_t n_rows, int32_t f_in, int32_t f_out, int32_t k1,
                                       size_t* bytes);
int pygsd_magnetic_dense_bwd_f32(const float* const* a, const float* const* b, int32_t k1,
                                 const float* w, const float* g_real, const float* g_imag, int64_t ldg,
                                 float* const* da, float* const* db, float* dw, float* dbias,
                                 int32_t n_rows, int32_t f_in, int32_t f_out,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Piece layouts (round 5; no reference counterpart: the reference is single-device).  The sharded propagate moves feature
 * rows in 16-float PIECES: a rank's send buffers hold, per inbound phase, one slot per destination rank with that rank's column
 * slice of the rows of the phase; its receive buffer of the return exchange holds, per return chunk, one slot per source rank with
 * that rank's column slice of the products.  `pygsd_piece_layout` describes where element (row t, column c) of an [n_rows, F]
 * operand lives in such a set of buffers, so that the dense kernels can WRITE the last gradient term straight into the send
 * buffers (no packing pass) and READ the last Chebyshev term straight out of the receive buffer (no merge pass):
 *     blk = t / blk_rows,  u = t % blk_rows,  r: lo[r] <= u < lo[r + 1],  j = c / slot_floats,
 *     slot = (blk + replica) * slots_per_blk + j        (replica = 0 for reads; 0 .. replicas - 1 for stores)
 *     element offset from the operand pointer = base[r] + (slot * rows[r] + (u - lo[r])) * row_stride + c % slot_floats
 * Constraints: 1 <= n_chunks <= 4, at most 8 blocks, slot_floats = 16 * 2^k, 16-byte aligned bases / strides.
 *   send buffers of phase c = [p_r][p_c][phase rows][groups * fw]: blk_rows = rows of the range (one block), lo = phase bounds,
 *     rows[c] = rows of phase c, slots_per_blk = p_c, replicas = p_r, row_stride = groups * fw, slot_floats = fw, base[c] = the
 *     phase's buffer; the operand pointer of feature group g is the first buffer + g * fw floats.
 *   receive buffer of the return = per chunk r [p_r][p_c][chunk rows][groups * fw]: blk_rows = rows of a row block, lo = chunk
 *     bounds, slots_per_blk = p_c, replicas = 1.
 * pygsd_magnetic_dense_fwd_pieces_f32 / _bwd_pieces_f32: the fused dense stage with the LAST term's operands a[k1 - 1] / b[k1 - 1]
 * read through `last_in` and (backward) the last term's gradients da[k1 - 1] / db[k1 - 1] stored through `last_out`; NULL = plain
 * row-major as in pygsd_magnetic_dense_{fwd,bwd}_f32, whose results they reproduce bit for bit (f_in = 64 or 128 only).
 * pygsd_gather_pieces_f32: outs[g][t, :] = (zs ? zs[g][t, :] : 0) + the row read through `layout` from srcs[g] -- the merge of a
 * returned product, with the addend of the adjoint's last step (gX = dT_0 + S^T dT_1) folded in; n_groups <= 4, width a
 * multiple of 16 floats, ldz / ldo row strides in floats (multiples of 4).
 * ------------------------------------------------------------------------------------------- */
typedef struct pygsd_piece_layout {
    int64_t base[4];
    int32_t lo[5];
    int32_t rows[4];
    int32_t n_chunks;
    int32_t blk_rows;
    int32_t slots_per_blk;
    int32_t row_stride;
    int32_t slot_floats;
    int32_t replicas;
} pygsd_piece_layout;
int pygsd_magnetic_dense_fwd_pieces_f32(const float* const* a, const float* const* b, int32_t k1, const float* w,
                                        const float* bias, float* out_real, float* out_imag, int32_t n_rows, int32_t f_in,
                                        int32_t f_out, const pygsd_piece_layout* last_in, void* stream);
int pygsd_magnetic_dense_bwd_pieces_f32(const float* const* a, const float* const* b, int32_t k1, const float* w,
                                        const float* g_real, const float* g_imag, int64_t ldg, float* const* da,
                                        float* const* db, float* dw, float* dbias, int32_t n_rows, int32_t f_in, int32_t f_out,
                                        void* workspace, size_t workspace_bytes, const pygsd_piece_layout* last_in,
                                        const pygsd_piece_layout* last_out, void* stream);
int pygsd_gather_pieces_f32(const float* const* srcs, const pygsd_piece_layout* layout, const float* const* zs, int64_t ldz,
                            float* const* outs, int64_t ldo, int32_t n_groups, int32_t n_rows, int32_t width, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Node-id validation.  minmax[0] = min(minmax[0], min ids), minmax[1] = max(minmax[1], max ids) (the caller
 * initialises minmax to {INT64_MAX, INT64_MIN}; several lists may be folded into one pair).  The host raises
 * IndexError when an id falls outside [0, num_nodes) -- where the reference's index_select / scatter_add_
 * raise (nn/directed/MagNetConv.py:196-240 via MessagePassing.propagate) -- instead of letting the SpMM gather
 * out of bounds.
 * ------------------------------------------------------------------------------------------- */
int pygsd_id_range_i64(const int64_t* ids, int64_t n, int64_t* minmax, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Streaming float4 copy dst[0..n) = src[0..n) (n a multiple of 4, 16-byte aligned pointers, non-temporal
 * loads and stores).  Measurement only: bench.py times it on >= 1 GiB in the same run as the yardstick of
 * ACHIEVABLE HBM bandwidth (`roofline.achievable_peak`).  No reference counterpart.
 * ------------------------------------------------------------------------------------------- */
int pygsd_stream_copy_f32(const float* src, float* dst, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Send-buffer packing of the sharded propagate (no reference counterpart: the reference is single-device).
 * xs[g] (g < groups <= 4): local feature group g, [n_rows, row_bytes] with row stride ld_bytes.  out, viewed as
 * [phases][p_r][p_c][n_rows / phases][groups][row_bytes / p_c] bytes: for every column phase c the equal-split
 * all-to-all input whose chunk d = i * p_c + j carries column slice j of rows [c, c + 1) * n_rows / phases, the
 * groups side by side (p_r = p_c = 1: the all-gather input of the row layout).  Byte-wise, 16-byte units: fp32
 * and bf16 alike; row_bytes / p_c and ld_bytes must be multiples of 16.
 * ------------------------------------------------------------------------------------------- */
int pygsd_pack_slices(const void* const* xs, int32_t groups, int32_t n_rows, int32_t row_bytes, int64_t ld_bytes,
                      int32_t p_r, int32_t p_c, int32_t phases, void* out, void* stream);
/* out = sum_j weights[j] * xs[j], every operand read once: the hop accumulations feat += w[h] * cur of SIMPA / DIMPA
 * (nn/signed/SIMPA.py:77-93, nn/directed/DIMPA.py:52-57) as one pass instead of one per hop.  xs: HOST array of k (<= 8)
 * device pointers to contiguous [n_rows, n_cols] matrices; weights: HOST array of k floats (they travel by value);
 * n_cols % 4 == 0, 16-byte aligned.  out has the row stride ldo (elements, a multiple of 4): it may be a column block of a
 * wider matrix -- SIMPA's cat([feat_p, feat_n], dim=1) (SIMPA.py:95) written in place. */
int pygsd_weighted_sum_f32(const float* const* xs, const float* weights, int32_t k, int64_t n_rows, int32_t n_cols,
                           float* out, int64_t ldo, void* stream);
/* out[j] = <g, xs[j]>, j < k (<= 8): the gradients of SIMPA's hop weights (autograd's reduction of g * cur_h for
 * SIMPA.py:77-93), g read ONCE for all k products.  g: [n_rows, n_cols] with row stride ldg (a column block of the
 * upstream gradient of the concatenated output); xs: HOST array of device pointers to contiguous [n_rows, n_cols]
 * matrices; out: k floats on the device; workspace >= 32 KiB (64 KiB lets k > 4 use the full grid), 8-byte aligned.  Products and sums in float64 in a fixed
 * order, rounded to fp32 once (deterministic; the correctly rounded dot product of the fp32 operands). */
int pygsd_dots_f32(const float* g, int64_t ldg, const float* const* xs, int32_t k, int64_t n_rows, int32_t n_cols,
                   float* out, void* workspace, size_t workspace_bytes, void* stream);
/* ---------------------------------------------------------------------------------------------
 * Tall-skinny linear maps of the non-magnetic layers on the matrix cores (csrc/tall.hip).
 *
 *   Y[n_rows, f_out] = [X_0 | X_1 | ... | X_{n_seg-1}] W (+ bias)
 *
 * One pass over tall operands (n_rows ~ 10^5..10^7): every segment row is read once and every output row written once.
 * Replaces the Linear / weight products around the aggregations -- x W of DiGCNConv (nn/directed/DiGCNConv.py:66), the
 * Linear of the inception block (nn/directed/DiGCN_Inception_Block.py:44-46), the Linear over [aggregated | own] features
 * of SGCNConv (nn/signed/SGCNConv.py:121-126) -- and, with the upstream gradient and the back-propagated aggregates as the
 * segments and w_transposed != 0, their input gradients [g | dP] W^T in one product instead of one GEMM per block plus
 * accumulation passes.
 * dtype: 0 = fp32, 1 = bf16 storage with fp32 accumulation (v_mfma_f32_16x16x32_bf16), for X, W, bias and Y alike.
 * fp32 products take one of two forms (pygsd_tall_f32_form): SPLIT, the default wherever K, f_out and every input segment
 * are multiples of 32 columns and K * f_out <= 24576 -- each fp32 value as the sum of three bf16 values, the six largest of
 * the nine partial products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (the partial products of a 32-column block summed
 * apart, added to the running sum once): within 0.9e-7 * sum |x| |w| of the float64 product where an fp32 fmaf chain is within
 * 3.5e-7 (measured, profiles/r5t_tall_forms.json), 2.7x fewer matrix cycles, but
 * not bitwise any fp32 summation order; EXACT -- an fmaf chain per
 * output on v_mfma_f32_16x16x4_f32 -- for every other shape and on request.  Non-finite values: the three pieces carry finite
 * values below the largest bf16 (3.39e38) only, so a 16-row tile (pygsd_tall_gram: an element of the result) whose sums come
 * out non-finite -- an operand held +-inf, NaN or a magnitude beyond that, or the sum itself overflowed -- is computed again
 * from the fp32 operands by fmaf chains: +-inf and NaN stand where torch.matmul puts them, in either form
 * (csrc/tall.hip: any_not_finite / exact_tile_store, csrc/gram.hip: gram_finish_kernel; tests/test_gpu_nonfinite.py).
 * xs / ldx / widths: HOST arrays over the n_seg (<= 4) column segments -- device pointer
 * (16-byte aligned), row stride in elements (a multiple of 16 bytes) and width (a multiple of 32 columns for bf16, 16 for
 * fp32).  W[k][n] (k over the concatenated segment columns) sits at w[k * ldw + n], or at w[n * ldw + k] when w_transposed.
 * The f_out = sum of out_widths output columns are written to n_out (<= 8) column segments ys / ldy / out_widths of the same
 * form (one [n_rows, f_out] matrix, or one matrix per consumer so that each is gathered from contiguous rows).
 * bias: f_out elements (over the concatenated output columns) or NULL.  Shapes: pygsd_tall_linear_supported(dtype, K = sum of widths, f_out) -- K, f_out <= 256 in
 * the multiples above with K * f_out <= 32768 (bf16) / 16384 (fp32) (W lives in 64 KB of LDS); the host falls back to
 * library GEMMs otherwise.
 * ------------------------------------------------------------------------------------------- */
int pygsd_tall_linear_supported(int32_t dtype, int32_t k_total, int32_t f_out);
/* form: 0 = split where the shape allows (default; PYGSD_TALL_F32=exact at load selects 1), 1 = exact everywhere, anything
 * else = query only.  Returns the form in force before the call.  Process-wide; not meant to be flipped while launches are
 * being issued from other threads. */
int pygsd_tall_f32_form(int32_t form);
int pygsd_tall_linear(const void* const* xs, const int64_t* ldx, const int32_t* widths, int32_t n_seg, const void* w,
                      int64_t ldw, int32_t w_transposed, const void* bias, void* const* ys, const int64_t* ldy,
                      const int32_t* out_widths, int32_t n_out, int64_t n_rows, int32_t dtype, void* stream);
/* out[c] = sum_r x[r * ldx + c], c < f, accumulated in fp32 in a fixed order (deterministic): the bias gradients of the
 * layers above (the reference's autograd reduces dY over the nodes for DiGCNConv.py:90-93 / torch.nn.Linear biases).
 * dtype as above; f a multiple of 4 (fp32) / 8 (bf16), at most 1024 / 2048; workspace from pygsd_column_sums_workspace. */
int pygsd_column_sums_workspace(int64_t n_rows, int32_t f, int32_t dtype, size_t* bytes);
int pygsd_column_sums(const void* x, int64_t ldx, int64_t n_rows, int32_t f, int32_t dtype, float* out, void* workspace,
                      size_t workspace_bytes, void* stream);
/* Weight gradients of the tall linear maps above (csrc/gram.hip):
 *
 *   out[K, F] (fp32, row-major, ld = F) = [X_0 | X_1 | ...]^T [G_0 | G_1 | ...]        reduction over the n_rows rows
 *
 * -- what autograd's `mm` backward computes for the reference's x W products (nn/directed/DiGCNConv.py:66,
 * nn/directed/DiGCN_Inception_Block.py:44-46, nn/signed/SGCNConv.py:121-126): dW = x^T dY.  Every operand row is read once
 * per column chunk of the other side (coalesced 16-byte row loads), transposed through a wavefront-private LDS image
 * (bf16: ds_read_b64_tr_b16) and multiplied on the matrix cores (fp32: exact v_mfma_f32_16x16x4_f32; bf16:
 * v_mfma_f32_16x16x32_bf16 with fp32 accumulation); per-block partials are added in a fixed order (deterministic).
 * dtype, xs / ldx / x_widths and gs / ldg / g_widths as for pygsd_tall_linear (HOST arrays over up to 8 segments a side;
 * widths multiples of 16 columns; at most 16 column chunks a side -- X in chunks of <= 64 columns, G of <= 128).  The result is fp32 for both dtypes.  workspace from pygsd_tall_gram_workspace. */
int pygsd_tall_gram_workspace(int64_t n_rows, int32_t k_total, int32_t f_total, int32_t dtype, size_t* bytes);
int pygsd_tall_gram(const void* const* xs, const int64_t* ldx, const int32_t* x_widths, int32_t n_x, const void* const* gs,
                    const int64_t* ldg, const int32_t* g_widths, int32_t n_g, int64_t n_rows, int32_t dtype, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);
/* C[m, n] (+)= A[m, k] B[k, n] (+ bias[n]) in exact fp32 for ANY shapes and strides (csrc/gemm.hip): the catch-all behind
 * the MFMA kernels above for the dense products they do not tile -- odd widths, reductions deeper than 256, transposed
 * views: torch.mm / torch.matmul of the reference's layers at such shapes (e.g. the 2879-wide first MagNetConv of
 * examples/magnet_node.py through nn/directed/MagNetConv.py:217-247, the 10-class Conv1d head of MagNet_node_classification)
 * and their autograd gradients.  A[i][j] sits at a[i * sa_m + j * sa_k], B[i][j] at b[i * sb_k + j * sb_n] (element
 * strides: a transpose is a stride swap), C row-major with row stride ldc.  accumulate != 0: C += ...  A reduction that is
 * long against the output is split into k-ranges added in a fixed order (workspace from pygsd_gemm_f32_workspace; 0 bytes
 * otherwise).  Deterministic. */
int pygsd_gemm_f32_workspace(int64_t m, int64_t n, int64_t k, size_t* bytes);
int pygsd_gemm_f32(const float* a, int64_t sa_m, int64_t sa_k, const float* b, int64_t sb_k, int64_t sb_n, const float* bias,
                   float* c, int64_t ldc, int64_t m, int64_t n, int64_t k, int32_t accumulate, void* workspace,
                   size_t workspace_bytes, void* stream);
/* The same product for bf16 operands (ABI v16): A, B and bias hold bf16, every product is exact in fp32, the sums are fp32
 * fmaf chains, and the result -- plus an optional fp32 addend Z[m, n] at row stride ldz (NULL: none) -- is stored in fp32
 * (c_is_f32 != 0) or rounded to bf16 ONCE (c_is_f32 == 0).  The catch-all behind pygsd_tall_linear / pygsd_tall_gram for bf16
 * widths their MFMA tiles do not take (a 20-, 48- or 320-wide bf16 DiGCNConv: torch.matmul(x, self.weight) of
 * nn/directed/DiGCNConv.py:66 and its gradients), so that no bf16 product of the path reaches hipBLASLt either.  Several
 * column segments accumulate through Z in fp32 and are rounded by the last call.  Workspace: pygsd_gemm_f32_workspace. */
int pygsd_gemm_bf16(const void* a, int64_t sa_m, int64_t sa_k, const void* b, int64_t sb_k, int64_t sb_n, const void* bias,
                    void* c, int64_t ldc, int32_t c_is_f32, const float* z, int64_t ldz, int64_t m, int64_t n, int64_t k,
                    void* workspace, size_t workspace_bytes, void* stream);

/* A 64-bit fingerprint of `bytes` bytes at `data` (device memory, 4-byte aligned) into *out (device): the sum of a mixed
 * function of every 4-byte word and its position -- order-independent across blocks (deterministic), position-dependent.
 * The host side (memo.py) compares it with the fingerprint taken when an operator was memoised: the reference re-derives its
 * operators from the CURRENT contents of edge_index / edge_weight on every uncached forward (nn/directed/MagNetConv.py:157-181,
 * nn/directed/DGCNConv.py:76-97, nn/general/conv_base.py:86-114), so a memo hit must also survive writes that bypass torch's
 * version counter (`t.data[...] = ...`).  No reference counterpart (the reference has no memo). */
int pygsd_fingerprint_u64(const void* data, size_t bytes, uint64_t* out, void* stream);

/* Keeps `stream` busy for `microseconds` (one idle lane polling the constant-rate wall clock).  Measurement
 * only: the single-GPU rehearsal of the sharded propagate (tools/emulate_sharded.py) uses it as the wire time
 * of an xGMI exchange on its communication stream.  No reference counterpart. */
int pygsd_spin_us(double microseconds, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel-timing recorder (measurement only; used by bench.py for the roofline object).
 * When enabled, every kernel launch of this library is bracketed by a pair of hipEvents on the
 * launch stream.  pygsd_prof_collect synchronises those events and returns, for kernel class
 * `kernel_id` (PYGSD_K_*), the number of launches and their summed duration in milliseconds.
 * ------------------------------------------------------------------------------------------- */
enum {
    PYGSD_K_SPMM = 0,
    PYGSD_K_SPMM2 = 1,
    PYGSD_K_SDDMM = 2,
    PYGSD_K_BUILD = 3,
    PYGSD_K_ELEMENTWISE = 4,
    PYGSD_K_DENSE = 5,      /* fused dense forward */
    PYGSD_K_DENSE_BWD = 6,  /* fused dense backward (+ partial reduction) */
    PYGSD_K_COUNT = 7
};
int pygsd_prof_enable(int32_t on);
int pygsd_prof_reset(void);
int pygsd_prof_collect(int32_t kernel_id, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* PYGSD_HIP_H */
