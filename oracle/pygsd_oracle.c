/*
 * ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Scalar C restatement of the primitives of the sparse message-passing path of
 * SherylHYX/pytorch_geometric_signed_directed.  Single-threaded, literal (one edge at a time, in
 * COO order), float32 arithmetic in the same order the reference's ATen CPU kernels use.
 * Built by oracle/Makefile into oracle/libpygsd_oracle.so; only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product library (libpygsd_hip.so) never links
 * or calls it.
 *
 * Parity pinning: checked in tests/test_oracle_c.py against the golden vectors recorded from the
 * reference's own Python (tests/golden/ fixtures, oracle/gen_golden.py) and against oracle/ref_layers.py.
 *
 * Reference call sites restated (paths under torch_geometric_signed_directed/):
 *   oracle_propagate_f32         MessagePassing.propagate with message = norm.view(-1,1) * x_j
 *                                (nn/directed/MagNetConv.py:196-240,251; nn/directed/DiGCNConv.py:86-88;
 *                                 nn/directed/DGCNConv.py:95-99; nn/general/conv_base.py:111-116;
 *                                 nn/signed/SGCNConv.py:101-128 with aggr='mean')
 *   oracle_magnetic_laplacian    utils/directed/get_magnetic_Laplacian.py:47-85 and
 *                                utils/general/get_magnetic_signed_Laplacian.py:47-90
 *   oracle_complex_relu_f32      nn/directed/complex_relu.py:21-22
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* out[dst[e], :] += w[e] * x[src[e], :]  (w == NULL -> 1); mean != 0 divides by max(count, 1). */
int oracle_propagate_f32(const int64_t* src, const int64_t* dst, const float* w, int64_t n_edges,
                         const float* x, int64_t n_feat, float* out, int64_t n_out, int mean)
{
    memset(out, 0, sizeof(float) * (size_t)(n_out * n_feat));
    float* cnt = mean ? (float*)calloc((size_t)(n_out > 0 ? n_out : 1), sizeof(float)) : NULL;
    for (int64_t e = 0; e < n_edges; ++e) {
        const float* xr = x + src[e] * n_feat;
        float* o = out + dst[e] * n_feat;
        if (w) {
            const float we = w[e];
            for (int64_t f = 0; f < n_feat; ++f) {
                const float msg = we * xr[f]; /* message() rounds the product ...      */
                o[f] = o[f] + msg;            /* ... then scatter_add_ rounds the sum  */
            }
        } else {
            for (int64_t f = 0; f < n_feat; ++f) o[f] = o[f] + xr[f];
        }
        if (cnt) cnt[dst[e]] += 1.0f;
    }
    if (cnt) {
        for (int64_t r = 0; r < n_out; ++r) {
            const float c = cnt[r] < 1.0f ? 1.0f : cnt[r];
            for (int64_t f = 0; f < n_feat; ++f) out[r * n_feat + f] /= c;
        }
        free(cnt);
    }
    return 0;
}

void oracle_complex_relu_f32(const float* re, const float* im, int64_t n, float* ore, float* oim)
{
    for (int64_t i = 0; i < n; ++i) {
        const float m = re[i] >= 0.0f ? 1.0f : 0.0f;
        ore[i] = m * re[i];
        oim[i] = m * im[i];
    }
}

typedef struct {
    int64_t key;
    int64_t ord; /* position before sorting: makes qsort stable */
    float w, th, ab;
} entry_t;

static int cmp_entry(const void* a, const void* b)
{
    const entry_t* x = (const entry_t*)a;
    const entry_t* y = (const entry_t*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->ord < y->ord ? -1 : (x->ord > y->ord ? 1 : 0);
}

/*
 * (Signed) magnetic Laplacian, COO.  Outputs must hold up to 2*n_edges + n entries; *out_nnz gets
 * E_s + n: the coalesced symmetrised off-diagonals sorted by (row, col), then n self loops.
 * sym != 0: L = I - D^-1/2 A_s D^-1/2 (.) exp(i Theta); sym == 0: L = D - A_s (.) exp(i Theta).
 */
int oracle_magnetic_laplacian(const int64_t* row, const int64_t* col, const float* w, int64_t n_edges,
                              int64_t n, double q, int sym, int is_signed, int absolute_degree,
                              int64_t* out_row, int64_t* out_col, float* out_re, float* out_im,
                              int64_t* out_nnz)
{
    entry_t* ent = (entry_t*)malloc(sizeof(entry_t) * (size_t)(2 * n_edges + 1));
    float* deg = (float*)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    if (!ent || !deg) return 1;
    int64_t m = 0;
    for (int64_t e = 0; e < n_edges; ++e) m += row[e] != col[e];
    /* cat([row, col]), cat([col, row]) over the loop-free edges */
    int64_t k = 0;
    for (int64_t e = 0; e < n_edges; ++e) {
        if (row[e] == col[e]) continue;
        const float we = w ? w[e] : 1.0f;
        ent[k] = (entry_t){row[e] * n + col[e], k, we, we, fabsf(we)};
        ent[m + k] = (entry_t){col[e] * n + row[e], m + k, we, -we, fabsf(we)};
        ++k;
    }
    qsort(ent, (size_t)(2 * m), sizeof(entry_t), cmp_entry);
    /* coalesce(add): merge runs of equal keys, summing in sorted order */
    int64_t es = 0;
    float *a_sym = out_re, *theta = out_im; /* reuse the outputs as scratch */
    float* a_abs = (float*)malloc(sizeof(float) * (size_t)(2 * m + 1));
    if (!a_abs) return 1;
    for (int64_t i = 0; i < 2 * m;) {
        int64_t j = i;
        float s = 0.0f, t = 0.0f, a = 0.0f;
        while (j < 2 * m && ent[j].key == ent[i].key) {
            s = s + ent[j].w;
            t = t + ent[j].th;
            a = a + ent[j].ab;
            ++j;
        }
        out_row[es] = ent[i].key / n;
        out_col[es] = ent[i].key % n;
        a_sym[es] = s / 2.0f;
        theta[es] = t;
        a_abs[es] = a / 2.0f;
        ++es;
        i = j;
    }
    for (int64_t i = 0; i < es; ++i) {
        float d = a_sym[i];
        if (is_signed) d = absolute_degree ? a_abs[i] : fabsf(a_sym[i]);
        deg[out_row[i]] = deg[out_row[i]] + d;
    }
    /* torch casts the Python complex 1j*2*pi*q to complex64 before multiplying the fp32 tensor */
    const float two_pi_q = (float)(2.0 * M_PI * q);
    for (int64_t i = 0; i < es; ++i) {
        const float ph = two_pi_q * theta[i];
        const float c = cosf(ph), s = sinf(ph);
        float mag;
        if (sym) {
            const float dr = deg[out_row[i]], dc = deg[out_col[i]];
            const float ir = dr == 0.0f ? 0.0f : powf(dr, -0.5f);
            const float ic = dc == 0.0f ? 0.0f : powf(dc, -0.5f);
            mag = ir * a_sym[i] * ic;
        } else {
            mag = a_sym[i];
        }
        out_re[i] = -(mag * c);
        out_im[i] = -(mag * s);
    }
    for (int64_t v = 0; v < n; ++v) {
        out_row[es + v] = v;
        out_col[es + v] = v;
        out_re[es + v] = sym ? 1.0f : deg[v];
        out_im[es + v] = 0.0f;
    }
    *out_nnz = es + n;
    free(ent);
    free(deg);
    free(a_abs);
    return 0;
}
