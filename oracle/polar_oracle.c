/* oracle/polar_oracle.c — TEST INFRASTRUCTURE ONLY (CPU restatement, the parity oracle).
 *
 * A plain-C99 restatement of the reference's polar SC/SCL path
 * (/root/reference/PolarC/PolarCode.{h,cpp}); every function cites the reference
 * lines it follows.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (polar_amd/) never does.
 *
 * PINNING: this oracle is checked (tests/test_oracle_vs_ref.py, here in the build
 * container) against the UNMODIFIED reference compiled into oracle/_ref/, and
 * (tests/test_oracle_golden.py, everywhere) against the golden vectors that the
 * reference produced (tests/golden/, generator script committed next to them),
 * including the reference's own deterministic main.cpp BLER table.
 *
 * Own structure (not the reference's): flat arenas allocated once per handle,
 * no per-decode new/delete, and the Tal-Vardy copy-on-write of
 * getArrayPointer_* (PolarCode.cpp:305-373) replaced by "write your own slot,
 * read through a per-layer slot pointer" — legal because every write to a
 * (layer, column) array is a complete overwrite performed by ALL active paths
 * in the same step, so no path ever needs the old content of the slot it owns.
 * Arithmetic, operation order, tie-breaks and path-index (LIFO stack) semantics
 * are the reference's.
 *
 * Third-party arithmetic the reference depends on and this file restates or calls:
 *   - libstdc++ 11.4 std::sort (introsort; tie order matters at PolarCode.cpp:38-40)
 *     -> restated in orc_introsort() below, pinned against _ref tables.
 *   - libstdc++ std::minstd_rand0 + std::normal_distribution<double> (Marsaglia
 *     polar) (PolarCode.cpp:688-689) -> restated in orc_normal().
 *   - glibc rand()/exp()/log()/pow()/sqrt(): called directly (same glibc).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle_synth.h"   /* the oracle's own statement of the workload definition (not the product header) */

typedef struct {
    int n, N, K, crc;
    double eps;
    uint8_t *frozen;   /* [N] 1 = frozen (PolarCode.h:45) */
    uint16_t *order;   /* [N] _channel_order_descending (PolarCode.h:46) */
    uint16_t *bitrev;  /* [N] (PolarCode.h:48) */
    uint8_t *crcm;     /* [crc][K] (PolarCode.h:47) */
    /* decoder arenas, sized for Lmax */
    int Lmax;
    double *llr0;      /* [N] layer 0, shared by all paths (never rewritten) */
    double *llr;       /* layers 1..n: layer lam at llr + Lmax*llr_off[lam], [slot][2^(n-lam)] */
    double *prob;      /* probability domain: layers 0..n, [slot][2*2^(n-lam)] */
    size_t *llr_off;   /* [n+1] element offset of layer lam within one slot-major block */
    uint8_t *cl, *cr;  /* column 0 / column 1 of C_lam: same offsets as llr */
    uint8_t *info;     /* [Lmax][N] */
    double *pm;        /* [Lmax] */
    uint8_t *active;   /* [Lmax] */
    uint16_t *pL;      /* [(n+1)][Lmax] slot pointer for LLR/P layer */
    uint16_t *pC;      /* [(n+1)][Lmax] slot pointer for column 0 of C layer */
    uint16_t *stack;   /* inactive path indices, LIFO */
    int sp;
} orc_t;

/* ---------- create_bit_rev_order: PolarCode.cpp:647-656 ---------- */
static void orc_bitrev(orc_t *c) {
    for (int i = 0; i < c->N; ++i) {
        unsigned t = (unsigned)i;
        unsigned r = (t & 1u) << (c->n - 1);
        for (int j = c->n - 1; j; --j) {
            t >>= 1;
            r += (t & 1u) << (j - 1);
        }
        c->bitrev[i] = (uint16_t)r;
    }
}

/* ---------- libstdc++ std::sort restated (bits/stl_algo.h, GCC 11.4) ----------
 * comparator: PolarCode.cpp:40  z[bitrev[i1]] < z[bitrev[i2]]  */
typedef struct { const double *z; const uint16_t *br; } orc_cmp_t;
static int orc_less(const orc_cmp_t *k, uint16_t a, uint16_t b) { return k->z[k->br[a]] < k->z[k->br[b]]; }
static void orc_swap16(uint16_t *a, uint16_t *b) { uint16_t t = *a; *a = *b; *b = t; }

static void orc_move_median_to_first(uint16_t *res, uint16_t *a, uint16_t *b, uint16_t *c, const orc_cmp_t *k) {
    if (orc_less(k, *a, *b)) {
        if (orc_less(k, *b, *c)) orc_swap16(res, b);
        else if (orc_less(k, *a, *c)) orc_swap16(res, c);
        else orc_swap16(res, a);
    } else if (orc_less(k, *a, *c)) orc_swap16(res, a);
    else if (orc_less(k, *b, *c)) orc_swap16(res, c);
    else orc_swap16(res, b);
}
static uint16_t *orc_unguarded_partition(uint16_t *first, uint16_t *last, uint16_t *pivot, const orc_cmp_t *k) {
    for (;;) {
        while (orc_less(k, *first, *pivot)) ++first;
        --last;
        while (orc_less(k, *pivot, *last)) --last;
        if (!(first < last)) return first;
        orc_swap16(first, last);
        ++first;
    }
}
/* heap fallback (std::__partial_sort(first,last,last)) — std::__adjust_heap / __push_heap */
static void orc_adjust_heap(uint16_t *first, long hole, long len, uint16_t val, const orc_cmp_t *k) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (orc_less(k, first[child], first[child - 1])) child--;
        first[hole] = first[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        first[hole] = first[child - 1];
        hole = child - 1;
    }
    long parent = (hole - 1) / 2;
    while (hole > top && orc_less(k, first[parent], val)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = val;
}
static void orc_heapsort(uint16_t *first, uint16_t *last, const orc_cmp_t *k) {
    long len = last - first;
    if (len >= 2) {
        for (long parent = (len - 2) / 2;; --parent) {
            orc_adjust_heap(first, parent, len, first[parent], k);
            if (parent == 0) break;
        }
    }
    while (last - first > 1) {
        --last;
        uint16_t v = *last;
        *last = *first;
        orc_adjust_heap(first, 0, last - first, v, k);
    }
}
static void orc_introsort_loop(uint16_t *first, uint16_t *last, long depth, const orc_cmp_t *k) {
    while (last - first > 16) {
        if (depth == 0) { orc_heapsort(first, last, k); return; }
        --depth;
        uint16_t *mid = first + (last - first) / 2;
        orc_move_median_to_first(first, first + 1, mid, last - 1, k);
        uint16_t *cut = orc_unguarded_partition(first + 1, last, first, k);
        orc_introsort_loop(cut, last, depth, k);
        last = cut;
    }
}
static void orc_unguarded_linear_insert(uint16_t *last, const orc_cmp_t *k) {
    uint16_t val = *last;
    uint16_t *next = last - 1;
    while (orc_less(k, val, *next)) { *last = *next; last = next; --next; }
    *last = val;
}
static void orc_insertion_sort(uint16_t *first, uint16_t *last, const orc_cmp_t *k) {
    if (first == last) return;
    for (uint16_t *i = first + 1; i != last; ++i) {
        if (orc_less(k, *i, *first)) {
            uint16_t val = *i;
            memmove(first + 1, first, (size_t)(i - first) * sizeof(uint16_t));
            *first = val;
        } else orc_unguarded_linear_insert(i, k);
    }
}
static void orc_introsort(uint16_t *first, uint16_t *last, const orc_cmp_t *k) {
    if (first == last) return;
    long len = last - first, lg = 0;
    while ((1L << (lg + 1)) <= len) ++lg; /* std::__lg */
    orc_introsort_loop(first, last, 2 * lg, k);
    if (last - first > 16) {
        orc_insertion_sort(first, first + 16, k);
        for (uint16_t *i = first + 16; i != last; ++i) orc_unguarded_linear_insert(i, k);
    } else orc_insertion_sort(first, last, k);
}

/* ---------- initialize_frozen_bits: PolarCode.cpp:17-58 ---------- */
static void orc_construct(orc_t *c) {
    double *z = (double *)malloc(sizeof(double) * (size_t)c->N);
    for (int i = 0; i < c->N; ++i) z[i] = c->eps;
    for (int it = 0; it < c->n; ++it) {                 /* :23-33 */
        int inc = 1 << it;
        for (int j = 0; j < inc; ++j)
            for (int i = 0; i < c->N; i += 2 * inc) {
                double c1 = z[i + j], c2 = z[i + j + inc];
                z[i + j] = c1 + c2 - c1 * c2;
                z[i + j + inc] = c1 * c2;
            }
    }
    for (int i = 0; i < c->N; ++i) c->order[i] = (uint16_t)i;   /* :35-37 */
    orc_cmp_t k = { z, c->bitrev };
    orc_introsort(c->order, c->order + c->N, &k);               /* :38-40 */
    int eff = c->K + c->crc;                                    /* :42-49 */
    for (int i = 0; i < eff; ++i) c->frozen[c->order[i]] = 0;
    for (int i = eff; i < c->N; ++i) c->frozen[c->order[i]] = 1;
    for (int b = 0; b < c->crc; ++b)                            /* :51-56 (global rand()) */
        for (int j = 0; j < c->K; ++j) c->crcm[(size_t)b * c->K + j] = (uint8_t)(rand() % 2);
    free(z);
}

static void orc_alloc_decoder(orc_t *c, int Lmax) {
    int n = c->n, N = c->N;
    c->Lmax = Lmax;
    c->llr_off = (size_t *)calloc((size_t)n + 2, sizeof(size_t));
    size_t off = 0;
    for (int lam = 0; lam <= n; ++lam) { c->llr_off[lam] = off; off += (size_t)1 << (n - lam); }
    c->llr_off[n + 1] = off; /* = 2N-1 */
    c->llr0 = (double *)calloc((size_t)N, sizeof(double));
    c->llr = (double *)calloc((size_t)Lmax * off, sizeof(double));
    c->prob = (double *)calloc((size_t)Lmax * off * 2, sizeof(double));
    c->cl = (uint8_t *)calloc((size_t)Lmax * off, 1);
    c->cr = (uint8_t *)calloc((size_t)Lmax * off, 1);
    c->info = (uint8_t *)calloc((size_t)Lmax * N, 1);
    c->pm = (double *)calloc((size_t)Lmax, sizeof(double));
    c->active = (uint8_t *)calloc((size_t)Lmax, 1);
    c->pL = (uint16_t *)calloc((size_t)(n + 1) * Lmax, sizeof(uint16_t));
    c->pC = (uint16_t *)calloc((size_t)(n + 1) * Lmax, sizeof(uint16_t));
    c->stack = (uint16_t *)calloc((size_t)Lmax, sizeof(uint16_t));
}

/* PolarCode::PolarCode PolarCode.h:19-28 */
void *orc_create(int n, int K, double eps, int crc) {
    orc_t *c = (orc_t *)calloc(1, sizeof(orc_t));
    c->n = n; c->N = 1 << n; c->K = K; c->crc = crc; c->eps = eps;
    c->frozen = (uint8_t *)calloc((size_t)c->N, 1);
    c->order = (uint16_t *)calloc((size_t)c->N, sizeof(uint16_t));
    c->bitrev = (uint16_t *)calloc((size_t)c->N, sizeof(uint16_t));
    c->crcm = (uint8_t *)calloc((size_t)(crc > 0 ? crc : 1) * K, 1);
    orc_bitrev(c);
    orc_construct(c);
    orc_alloc_decoder(c, 128);
    return c;
}
void orc_destroy(void *h) {
    orc_t *c = (orc_t *)h;
    free(c->frozen); free(c->order); free(c->bitrev); free(c->crcm);
    free(c->llr_off); free(c->llr0); free(c->llr); free(c->prob); free(c->cl); free(c->cr);
    free(c->info); free(c->pm); free(c->active); free(c->pL); free(c->pC); free(c->stack);
    free(c);
}
void orc_get_frozen(void *h, uint8_t *o) { orc_t *c = h; memcpy(o, c->frozen, (size_t)c->N); }
void orc_get_order(void *h, uint16_t *o) { orc_t *c = h; memcpy(o, c->order, 2 * (size_t)c->N); }
void orc_get_bitrev(void *h, uint16_t *o) { orc_t *c = h; memcpy(o, c->bitrev, 2 * (size_t)c->N); }
void orc_get_crc_matrix(void *h, uint8_t *o) { orc_t *c = h; memcpy(o, c->crcm, (size_t)c->crc * c->K); }
void orc_set_crc_matrix(void *h, const uint8_t *m) { orc_t *c = h; memcpy(c->crcm, m, (size_t)c->crc * c->K); }
void orc_set_tables(void *h, const uint8_t *fz, const uint16_t *ord) {
    orc_t *c = h; memcpy(c->frozen, fz, (size_t)c->N); memcpy(c->order, ord, 2 * (size_t)c->N);
}

/* ---------- encode: PolarCode.cpp:60-91 ---------- */
void orc_encode(void *h, const uint8_t *info, uint8_t *coded) {
    orc_t *c = (orc_t *)h;
    int N = c->N, K = c->K;
    uint8_t *u = (uint8_t *)calloc((size_t)N, 1);
    for (int i = 0; i < K; ++i) u[c->order[i]] = info[i];              /* :65-67 */
    for (int i = K; i < K + c->crc; ++i) {                             /* :68-74 */
        uint8_t bit = 0;
        for (int j = 0; j < K; ++j) bit = (uint8_t)((bit + c->crcm[(size_t)(i - K) * K + j] * info[j]) % 2);
        u[c->order[i]] = bit;
    }
    for (int it = 0; it < c->n; ++it) {                                /* :76-83 */
        int inc = 1 << it;
        for (int j = 0; j < inc; ++j)
            for (int i = 0; i < N; i += 2 * inc) u[i + j] = (uint8_t)((u[i + j] + u[i + j + inc]) % 2);
    }
    for (int i = 0; i < N; ++i) coded[i] = u[c->bitrev[i]];            /* :85-87 */
    free(u);
}

/* ---------- crc_check: PolarCode.cpp:93-108 ---------- */
static int orc_crc_check(const orc_t *c, const uint8_t *u) {
    for (int i = c->K; i < c->K + c->crc; ++i) {
        uint8_t bit = 0;
        for (int j = 0; j < c->K; ++j)
            bit = (uint8_t)((bit + c->crcm[(size_t)(i - c->K) * c->K + j] * u[c->order[j]]) % 2);
        if (bit != u[c->order[i]]) return 0;
    }
    return 1;
}

/* slot-major accessors */
static double *LLR(orc_t *c, int lam, int slot) { return c->llr + (size_t)c->Lmax * c->llr_off[lam] + ((size_t)slot << (c->n - lam)); }
static double *PRB(orc_t *c, int lam, int slot) { return c->prob + 2 * ((size_t)c->Lmax * c->llr_off[lam] + ((size_t)slot << (c->n - lam))); }
static uint8_t *CL(orc_t *c, int lam, int slot) { return c->cl + (size_t)c->Lmax * c->llr_off[lam] + ((size_t)slot << (c->n - lam)); }
static uint8_t *CR(orc_t *c, int lam, int slot) { return c->cr + (size_t)c->Lmax * c->llr_off[lam] + ((size_t)slot << (c->n - lam)); }

/* initializeDataStructures + assignInitialPath: PolarCode.cpp:195-272.
 * The inactive-path stack is filled 0..L-1, so the first pop yields L-1. */
static int orc_init(orc_t *c, int L) {
    c->sp = 0;
    for (int l = 0; l < L; ++l) { c->active[l] = 0; c->stack[c->sp++] = (uint16_t)l; c->pm[l] = 0.0; }
    memset(c->info, 0, (size_t)L * c->N);
    int l0 = c->stack[--c->sp];
    c->active[l0] = 1;
    for (int lam = 0; lam <= c->n; ++lam) { c->pL[lam * c->Lmax + l0] = (uint16_t)l0; c->pC[lam * c->Lmax + l0] = (uint16_t)l0; }
    return l0;
}
/* clonePath: PolarCode.cpp:274-288 */
static int orc_clone(orc_t *c, int l) {
    int lp = c->stack[--c->sp];
    c->active[lp] = 1;
    c->pm[lp] = c->pm[l];
    for (int lam = 0; lam <= c->n; ++lam) {
        c->pL[lam * c->Lmax + lp] = c->pL[lam * c->Lmax + l];
        c->pC[lam * c->Lmax + lp] = c->pC[lam * c->Lmax + l];
    }
    return lp;
}
/* killPath: PolarCode.cpp:290-303 */
static void orc_kill(orc_t *c, int l) {
    c->active[l] = 0;
    c->stack[c->sp++] = (uint16_t)l;
    c->pm[l] = 0.0;
}

/* f-node, exact + min-sum branches: PolarCode.cpp:437-446 */
static double orc_f(double a, double b) {
    double fa = fabs(a), fb = fabs(b);
    double mx = (fa < fb) ? fb : fa;
    if (40 > mx) return log((exp(a + b) + 1) / (exp(a) + exp(b)));
    return (double)((a < 0) ? -1 : (a > 0)) * ((b < 0) ? -1 : (b > 0)) * ((fb < fa) ? fb : fa);
}

/* recursivelyCalcLLR(n, phi): PolarCode.cpp:422-455, unrolled bottom-up:
 * layers lam_top..n are recomputed, g at lam_top (phi_lam odd), f below. */
static void orc_calc_llr(orc_t *c, int L, int phi) {
    int n = c->n, lam_top;
    if (phi == 0) lam_top = 1;
    else { int tz = 0; while (((phi >> tz) & 1) == 0) ++tz; lam_top = n - tz; }
    for (int lam = lam_top; lam <= n; ++lam) {
        int S = 1 << (n - lam);
        int odd = (phi >> (n - lam)) & 1;
        for (int l = 0; l < L; ++l) {
            if (!c->active[l]) continue;
            const double *in = (lam == 1) ? c->llr0 : LLR(c, lam - 1, c->pL[(lam - 1) * c->Lmax + l]);
            double *out = LLR(c, lam, l);
            if (!odd) {
                for (int b = 0; b < S; ++b) out[b] = orc_f(in[2 * b], in[2 * b + 1]);
            } else {
                const uint8_t *cl = CL(c, lam, c->pC[lam * c->Lmax + l]);
                for (int b = 0; b < S; ++b) out[b] = (1 - 2 * cl[b]) * in[2 * b] + in[2 * b + 1];   /* :449-450 */
            }
            c->pL[lam * c->Lmax + l] = (uint16_t)l;
        }
    }
}

/* recursivelyCalcP(n, phi): PolarCode.cpp:375-420 (probability domain, with the
 * cross-path max-normalisation sigma per layer). Layout p[2*beta + {0,1}]. */
static void orc_calc_p(orc_t *c, int L, int phi) {
    int n = c->n, lam_top;
    if (phi == 0) lam_top = 1;
    else { int tz = 0; while (((phi >> tz) & 1) == 0) ++tz; lam_top = n - tz; }
    for (int lam = lam_top; lam <= n; ++lam) {
        int S = 1 << (n - lam);
        int odd = (phi >> (n - lam)) & 1;
        double sigma = 0.0;
        for (int l = 0; l < L; ++l) {
            if (!c->active[l]) continue;
            const double *q = PRB(c, lam - 1, c->pL[(lam - 1) * c->Lmax + l]);
            double *p = PRB(c, lam, l);
            const uint8_t *cl = CL(c, lam, c->pC[lam * c->Lmax + l]);
            for (int b = 0; b < S; ++b) {
                if (!odd) {
                    p[2 * b] = 0.5f * (q[2 * (2 * b)] * q[2 * (2 * b + 1)] + q[2 * (2 * b) + 1] * q[2 * (2 * b + 1) + 1]);
                    p[2 * b + 1] = 0.5f * (q[2 * (2 * b) + 1] * q[2 * (2 * b + 1)] + q[2 * (2 * b)] * q[2 * (2 * b + 1) + 1]);
                } else {
                    uint8_t u = cl[b];
                    p[2 * b] = 0.5f * q[2 * (2 * b) + (u % 2)] * q[2 * (2 * b + 1)];
                    p[2 * b + 1] = 0.5f * q[2 * (2 * b) + ((u + 1) % 2)] * q[2 * (2 * b + 1) + 1];
                }
                sigma = (sigma < p[2 * b]) ? p[2 * b] : sigma;
                sigma = (sigma < p[2 * b + 1]) ? p[2 * b + 1] : sigma;
            }
            c->pL[lam * c->Lmax + l] = (uint16_t)l;
        }
        for (int l = 0; l < L; ++l) {
            if (sigma == 0) break;
            if (!c->active[l]) continue;
            double *p = PRB(c, lam, l);
            for (int b = 0; b < S; ++b) { p[2 * b] = p[2 * b] / sigma; p[2 * b + 1] = p[2 * b + 1] / sigma; }
        }
    }
}

/* recursivelyUpdateC(n, phi) for odd phi: PolarCode.cpp:457-473 */
static void orc_update_c(orc_t *c, int L, int phi) {
    int n = c->n;
    int lam = n, ph = phi;
    for (;;) {
        int psi = ph >> 1;
        int S = 1 << (n - lam);
        for (int l = 0; l < L; ++l) {
            if (!c->active[l]) continue;
            const uint8_t *cl = CL(c, lam, c->pC[lam * c->Lmax + l]);
            const uint8_t *cr = CR(c, lam, l);
            uint8_t *dst = (psi % 2) ? CR(c, lam - 1, l) : CL(c, lam - 1, l);
            for (int b = 0; b < S; ++b) {
                dst[2 * b] = (uint8_t)((cl[b] + cr[b]) % 2);
                dst[2 * b + 1] = cr[b];
            }
            if (!(psi % 2)) c->pC[(lam - 1) * c->Lmax + l] = (uint16_t)l;
        }
        if ((psi % 2) == 1 && lam - 1 >= 1) { lam = lam - 1; ph = psi; } else break;
    }
}

static void orc_set_bit(orc_t *c, int l, int phi, uint8_t u) {
    if (phi % 2) CR(c, c->n, l)[0] = u;
    else { CL(c, c->n, l)[0] = u; c->pC[c->n * c->Lmax + l] = (uint16_t)l; }
}

static int orc_cmp_desc(const void *a, const void *b) {
    double x = *(const double *)a, y = *(const double *)b;
    return (x < y) - (x > y);
}

/* continuePaths_FrozenBit: PolarCode.cpp:475-487 */
static void orc_frozen(orc_t *c, int L, int phi, int llr_mode) {
    for (int l = 0; l < L; ++l) {
        if (!c->active[l]) continue;
        orc_set_bit(c, l, phi, 0);
        if (llr_mode) {
            double v = LLR(c, c->n, c->pL[c->n * c->Lmax + l])[0];
            c->pm[l] += log(1 + exp(-v));
        }
        c->info[(size_t)l * c->N + phi] = 0;
    }
}

/* continuePaths_UnfrozenBit: PolarCode.cpp:489-607 */
static void orc_unfrozen(orc_t *c, int L, int phi, int llr_mode) {
    double pf[256], pr[256];
    uint8_t cont[256];
    int na = 0, np = 0;
    for (int l = 0; l < L; ++l) {
        if (!c->active[l]) { pf[2 * l] = NAN; pf[2 * l + 1] = NAN; continue; }          /* :498-501 */
        if (llr_mode) {
            double v = LLR(c, c->n, c->pL[c->n * c->Lmax + l])[0];
            pf[2 * l] = -(c->pm[l] + log(1 + exp(-v)));                                    /* :505 */
            pf[2 * l + 1] = -(c->pm[l] + log(1 + exp(v)));                                 /* :506 */
        } else {
            const double *p = PRB(c, c->n, c->pL[c->n * c->Lmax + l]);
            pf[2 * l] = p[0]; pf[2 * l + 1] = p[1];                                        /* :510-511 */
        }
        pr[np++] = pf[2 * l]; pr[np++] = pf[2 * l + 1];
        na++;
    }
    int rho = L;
    if (2 * na < L) rho = 2 * na;                                                          /* :521-523 */
    for (int i = 0; i < 2 * L; ++i) cont[i] = 0;
    qsort(pr, (size_t)np, sizeof(double), orc_cmp_desc);                                   /* :528 (only the rho-th VALUE is used) */
    double thr = pr[rho - 1];                                                              /* :530 */
    int cnt = 0;
    for (int i = 0; i < 2 * L; ++i) {                                                      /* :533-541 */
        if (pf[i] > thr) { cont[i] = 1; cnt++; }
        if (cnt == rho) break;
    }
    if (cnt < rho) {                                                                       /* :543-553 */
        for (int i = 0; i < 2 * L; ++i) {
            if (pf[i] == thr) { cont[i] = 1; cnt++; }
            if (cnt == rho) break;
        }
    }
    for (int l = 0; l < L; ++l) {                                                          /* :555-560 */
        if (!c->active[l]) continue;
        if (cont[2 * l] == 0 && cont[2 * l + 1] == 0) orc_kill(c, l);
    }
    for (int l = 0; l < L; ++l) {                                                          /* :562-605 */
        if (cont[2 * l] == 0 && cont[2 * l + 1] == 0) continue;
        double v = 0.0;
        if (llr_mode) v = LLR(c, c->n, c->pL[c->n * c->Lmax + l])[0];
        if (cont[2 * l] == 1 && cont[2 * l + 1] == 1) {
            int lp = orc_clone(c, l);
            orc_set_bit(c, l, phi, 0);
            orc_set_bit(c, lp, phi, 1);
            memcpy(c->info + (size_t)lp * c->N, c->info + (size_t)l * c->N, (size_t)phi);   /* :574 */
            c->info[(size_t)l * c->N + phi] = 0;
            c->info[(size_t)lp * c->N + phi] = 1;
            if (llr_mode) {
                c->pm[l] += log(1 + exp(-v));                                              /* :580 */
                c->pm[lp] += log(1 + exp(v));                                              /* :582 */
            }
        } else if (cont[2 * l] == 1) {
            orc_set_bit(c, l, phi, 0);
            c->info[(size_t)l * c->N + phi] = 0;
            if (llr_mode) c->pm[l] += log(1 + exp(-v));                                    /* :593 */
        } else {
            orc_set_bit(c, l, phi, 1);
            c->info[(size_t)l * c->N + phi] = 1;
            if (llr_mode) c->pm[l] += log(1 + exp(v));                                     /* :601 */
        }
    }
}

/* findMostProbablePath: PolarCode.cpp:609-644 */
static int orc_best(orc_t *c, int L, int check_crc, int llr_mode) {
    int lp = 0, any = 0;
    double p1 = 0, pl = 1.7976931348623157e308;
    for (int l = 0; l < L; ++l) {
        if (!c->active[l]) continue;
        if (check_crc && !orc_crc_check(c, c->info + (size_t)l * c->N)) continue;
        any = 1;
        if (llr_mode) {
            if (c->pm[l] < pl) { pl = c->pm[l]; lp = l; }
        } else {
            /* :632-637  p_m[c_m[1]]: column 1 of C_n holds the last decided bit */
            uint8_t u = CR(c, c->n, l)[0];
            const double *p = PRB(c, c->n, c->pL[c->n * c->Lmax + l]);
            if (p1 < p[u]) { lp = l; p1 = p[u]; }
        }
    }
    if (any) return lp;
    return orc_best(c, L, 0, llr_mode);
}

/* decode_scl: PolarCode.cpp:150-190 */
static void orc_decode(orc_t *c, int L, int llr_mode, uint8_t *out, double *pm_out) {
    for (int phi = 0; phi < c->N; ++phi) {
        if (llr_mode) orc_calc_llr(c, L, phi); else orc_calc_p(c, L, phi);
        if (c->frozen[phi]) orc_frozen(c, L, phi, llr_mode); else orc_unfrozen(c, L, phi, llr_mode);
        if (phi % 2) orc_update_c(c, L, phi);
    }
    int l = orc_best(c, L, c->crc > 0, llr_mode);
    for (int b = 0; b < c->K; ++b) out[b] = c->info[(size_t)l * c->N + c->order[b]];     /* :172-174 */
    if (pm_out) *pm_out = c->pm[l];
}

/* decode_scl_llr: PolarCode.cpp:130-148 */
int orc_decode_scl_llr(void *h, const double *llr, int L, uint8_t *out, double *pm_out) {
    orc_t *c = (orc_t *)h;
    if (L < 1 || L > c->Lmax) return -1;
    orc_init(c, L);
    memcpy(c->llr0, llr, sizeof(double) * (size_t)c->N);
    orc_decode(c, L, 1, out, pm_out);
    return 0;
}
int orc_decode_scl_llr_batch(void *h, const double *llr, long B, int L, uint8_t *out) {
    orc_t *c = (orc_t *)h;
    for (long b = 0; b < B; ++b)
        if (orc_decode_scl_llr(h, llr + b * (long)c->N, L, out + b * (long)c->K, NULL)) return -1;
    return 0;
}
/* decode_scl_p1: PolarCode.cpp:110-128 */
int orc_decode_scl_p1(void *h, const double *p1, const double *p0, int L, uint8_t *out) {
    orc_t *c = (orc_t *)h;
    if (L < 1 || L > c->Lmax) return -1;
    int l0 = orc_init(c, L);
    double *q = PRB(c, 0, l0);
    for (int b = 0; b < c->N; ++b) { q[2 * b] = p0[b]; q[2 * b + 1] = p1[b]; }
    orc_decode(c, L, 0, out, NULL);
    return 0;
}

/* ---------- PolarM decode_sc_p1 -> polar_decode / cnop / vnop: PolarCode.m:290-295, 870-895 ----------
 * Recursive SC in the probability domain on p1 = P(bit = 1), in doubles like MATLAB.
 * PolarM's frozen_bits and info_bits are indexed in the same decoding order phi as PolarC's
 * (PolarCode.m:76-83 apply bit_reversed_order to the channel vector before sorting), and the
 * recursion pairs adjacent inputs y(1:2:end), y(2:2:end) exactly like PolarC's layers, so the
 * PolarC tables are used as they are.  Returns u (decisions) and x (re-encoded partial sums);
 * a leaf with y == 0.5 yields 0.5 (sign(0) = 0), as in MATLAB. */
static void orc_m_decode(const double *y, const uint8_t *fz, int len, double *u, double *x) {
    if (len == 1) {                                   /* :872-878 */
        if (fz[0]) { x[0] = 0; }
        else { double t = 1 - 2 * y[0]; double sg = (double)((t > 0) - (t < 0)); x[0] = (1 - sg) / 2; }
        u[0] = x[0];
        return;
    }
    int h = len / 2;
    double *t1 = (double *)malloc(sizeof(double) * (size_t)h * 3);
    double *x1 = t1, *x2 = t1 + h, *tmp = t1 + 2 * h;
    /* u1est = cnop(y(1:2:end), y(2:2:end)); cnop(w1,w2) = w1.*(1-w2) + w2.*(1-w1)  (:880, :889-891) */
    for (int i = 0; i < h; ++i) { double w1 = y[2 * i], w2 = y[2 * i + 1]; tmp[i] = w1 * (1 - w2) + w2 * (1 - w1); }
    orc_m_decode(tmp, fz, h, u, x1);                  /* f(1:N/2) */
    /* u2est = vnop(cnop(u1hardprev, y(1:2:end)), y(2:2:end)); vnop = w1.*w2 ./ (w1.*w2 + (1-w1).*(1-w2)) (:882, :893-895) */
    for (int i = 0; i < h; ++i) {
        double a = x1[i], w = y[2 * i];
        double w1 = a * (1 - w) + w * (1 - a);
        double w2 = y[2 * i + 1];
        tmp[i] = w1 * w2 / (w1 * w2 + (1 - w1) * (1 - w2));
    }
    orc_m_decode(tmp, fz + h, h, u + h, x2);          /* f(N/2+1:end) */
    /* x = reshape([cnop(u1hardprev,u2hardprev); u2hardprev],1,[])  (:885) */
    for (int i = 0; i < h; ++i) { x[2 * i] = x1[i] * (1 - x2[i]) + x2[i] * (1 - x1[i]); x[2 * i + 1] = x2[i]; }
    free(t1);
}
/* decode_sc_p1(p1): PolarCode.m:290-295; out[K] doubles (MATLAB returns doubles, 0.5 possible). */
int orc_decode_sc_p1(void *h, const double *p1, double *out) {
    orc_t *c = (orc_t *)h;
    int N = c->N;
    double *u = (double *)malloc(sizeof(double) * (size_t)N * 2);
    double *x = u + N;
    orc_m_decode(p1, c->frozen, N, u, x);
    for (int b = 0; b < c->K; ++b) out[b] = u[c->order[b]];
    free(u);
    return 0;
}

/* ---------- libstdc++ minstd_rand0 + normal_distribution<double> restated ---------- */
typedef struct { uint32_t x; int saved_ok; double saved; } orc_rng_t;
static uint32_t orc_minstd(orc_rng_t *r) { r->x = (uint32_t)(((uint64_t)r->x * 16807u) % 2147483647u); return r->x; }
/* std::generate_canonical<double,53>(minstd_rand0): two draws, range R = 2147483646 */
static double orc_canonical(orc_rng_t *r) {
    const long double R = 2147483646.0L;
    double sum = 0.0, tmp = 1.0;
    for (int k = 2; k != 0; --k) {
        sum += (double)(orc_minstd(r) - 1u) * tmp;
        tmp = (double)((long double)tmp * R);
    }
    double ret = sum / tmp;
    if (ret >= 1.0) ret = nextafter(1.0, 0.0);
    return ret;
}
static double orc_normal(orc_rng_t *r) {
    if (r->saved_ok) { r->saved_ok = 0; return r->saved * 1.0 + 0.0; }
    double x, y, r2;
    do {
        x = 2.0 * orc_canonical(r) - 1.0;
        y = 2.0 * orc_canonical(r) - 1.0;
        r2 = x * x + y * y;
    } while (r2 > 1.0 || r2 == 0.0);
    double mult = sqrt(-2 * log(r2) / r2);
    r->saved = x * mult;
    r->saved_ok = 1;
    return (y * mult) * 1.0 + 0.0;
}

/* ---------- get_bler_quick, faithful (serial RNGs): PolarCode.cpp:658-785 ----------
 * bler_out[n_L][n_e]. max_err/max_runs are 100/1000 in the reference (:661-662). */
void orc_get_bler_quick_ref(void *h, const double *ebno, int n_e, const uint8_t *Ls, int n_L,
                            int max_runs, int max_err, double *bler_out) {
    orc_t *c = (orc_t *)h;
    int N = c->N, K = c->K;
    double *num_err = (double *)calloc((size_t)n_L * n_e, sizeof(double));
    double *num_run = (double *)calloc((size_t)n_L * n_e, sizeof(double));
    uint8_t *coded = (uint8_t *)malloc((size_t)N), *info = (uint8_t *)calloc((size_t)K, 1), *dec = (uint8_t *)malloc((size_t)K);
    double *bpsk = (double *)malloc(sizeof(double) * N), *rx = (double *)malloc(sizeof(double) * N);
    double *noise = (double *)malloc(sizeof(double) * N), *llr = (double *)malloc(sizeof(double) * N);
    uint8_t *prev = (uint8_t *)malloc((size_t)n_e);
    double N_0 = 1.0;
    orc_rng_t g = { 1u, 0, 0.0 };                                                /* default_random_engine, seed 1 */
    for (int run = 0; run < max_runs; ++run) {
        if ((run % 100) == 0) for (int i = 0; i < K; ++i) info[i] = (uint8_t)(rand() % 2);     /* :703-707 */
        for (int i = 0; i < N; ++i) noise[i] = orc_normal(&g);                                  /* :708-710 */
        orc_encode(c, info, coded);                                                             /* :712 */
        for (int i = 0; i < N; ++i) bpsk[i] = 2.0f * ((double)coded[i]) - 1.0f;                 /* :715 */
        for (int li = 0; li < n_L; ++li) {
            memset(prev, 0, (size_t)n_e);
            for (int ie = 0; ie < n_e; ++ie) {
                if (num_err[li * n_e + ie] > max_err) continue;                                 /* :725 */
                num_run[li * n_e + ie]++;                                                       /* :728 */
                int run_sim = 1;
                for (int j = 0; j < ie; ++j) if (prev[j]) run_sim = 0;                          /* :732-738 */
                if (!run_sim) continue;
                double s = pow(10.0f, ebno[ie] / 20) * sqrt(((double)K) / ((double)N));         /* :744-745 */
                for (int i = 0; i < N; ++i) rx[i] = s * bpsk[i] + sqrt(N_0 / 2) * noise[i];     /* :747 */
                for (int i = 0; i < N; ++i) llr[i] = -4 * rx[i] * s / N_0;                      /* :752 */
                orc_decode_scl_llr(c, llr, Ls[li], dec, NULL);                                  /* :756 */
                int err = 0;
                for (int i = 0; i < K; ++i) if (info[i] != dec[i]) { err = 1; break; }
                if (err) num_err[li * n_e + ie]++; else prev[ie] = 1;                           /* :766-769 */
            }
        }
    }
    for (int i = 0; i < n_L * n_e; ++i) bler_out[i] = num_err[i] / num_run[i];                  /* :777-781 */
    free(num_err); free(num_run); free(coded); free(info); free(dec); free(bpsk); free(rx);
    free(noise); free(llr); free(prev);
}

/* ---------- synthetic (counter-based) workload: definition in include/polar_synth.h, restated in oracle_synth.h ---------- */
/* info bits of trial `trial` (block = trial/100 mirrors the every-100-runs refresh) */
void orc_synth_info(void *h, uint64_t seed, uint64_t trial, uint8_t *info) {
    orc_t *c = (orc_t *)h;
    uint64_t block = trial / 100;
    for (int i = 0; i < c->K; ++i) info[i] = (uint8_t)osy_bit(seed, block, OSY_INFO, i);
}
/* LLRs of trial `trial` at amplitude s: info -> encode -> BPSK -> +noise -> llr */
void orc_synth_llr(void *h, uint64_t seed, uint64_t trial, double s, double *llr, uint8_t *info_out) {
    orc_t *c = (orc_t *)h;
    uint8_t *info = (uint8_t *)malloc((size_t)c->K), *coded = (uint8_t *)malloc((size_t)c->N);
    orc_synth_info(h, seed, trial, info);
    orc_encode(h, info, coded);
    for (int p = 0; p < c->N / 2; ++p) {
        double z0, z1;
        osy_noise_pair(seed, trial, (uint32_t)p, &z0, &z1);
        llr[2 * p] = osy_bpsk_llr(s, coded[2 * p], z0);
        llr[2 * p + 1] = osy_bpsk_llr(s, coded[2 * p + 1], z1);
    }
    if (info_out) memcpy(info_out, info, (size_t)c->K);
    free(info); free(coded);
}
void orc_synth_llr_batch(void *h, uint64_t seed, uint64_t trial0, long B, double s, double *llr, uint8_t *info_out) {
    orc_t *c = (orc_t *)h;
    for (long b = 0; b < B; ++b)
        orc_synth_llr(h, seed, trial0 + (uint64_t)b, s, llr + b * (long)c->N, info_out ? info_out + b * (long)c->K : NULL);
}
/* ---------- 16/8/4-ASK Gray BICM front end: Constellation.m:84-93 (modulate), :123-144 (demap),
 * sweep conventions main_MC_CC_Comparison.m:88-96: sigma = sqrt(1/2)*10^(-snr_db/20), n0 = sigma^2,
 * y = sym + noise*sigma, fresh info every run (:50). ---------- */
void orc_synth_bicm_llr(void *h, int cid, uint64_t seed, uint64_t trial, double snr_db, double *llr, uint8_t *info_out) {
    orc_t *c = (orc_t *)h;
    const int nb = osy_nbits(cid);
    uint8_t *info = (uint8_t *)malloc((size_t)c->K), *coded = (uint8_t *)malloc((size_t)c->N);
    for (int i = 0; i < c->K; ++i) info[i] = (uint8_t)osy_bit(seed, trial, OSY_INFO, i);   /* block = trial: fresh info bits every run */
    orc_encode(h, info, coded);
    const double norm = osy_norm(cid);
    const double sigma = sqrt(1.0 / 2) * pow(10.0, -snr_db / 20);
    const double n0 = sigma * sigma;
    const int nsym = c->N / nb;
    for (int i = 0; i < c->N; ++i) llr[i] = 0.0;    /* positions beyond nsym*nb: p1 = 0.5 <=> llr = 0 (:94) */
    for (int i = 0; i < nsym; ++i) {
        int sym = 0;
        for (int j = 0; j < nb; ++j) sym += (1 << j) * coded[i * nb + j];
        double x = osy_point(cid, sym) / norm;
        double y = x + osy_symbol_noise(seed, trial, (uint32_t)i) * sigma;
        osy_demap(cid, norm, y, n0, llr + (size_t)i * nb, NULL);
    }
    if (info_out) memcpy(info_out, info, (size_t)c->K);
    free(info); free(coded);
}
void orc_synth_bicm_llr_batch(void *h, int cid, uint64_t seed, uint64_t trial0, long B, double snr_db, double *llr, uint8_t *info_out) {
    orc_t *c = (orc_t *)h;
    for (long b = 0; b < B; ++b)
        orc_synth_bicm_llr(h, cid, seed, trial0 + (uint64_t)b, snr_db, llr + b * (long)c->N, info_out ? info_out + b * (long)c->K : NULL);
}

double orc_snr_sqrt_linear(void *h, double ebno_db) {
    orc_t *c = (orc_t *)h;
    return pow(10.0f, ebno_db / 20) * sqrt(((double)c->K) / ((double)c->N));
}

/* Batched Monte-Carlo on the synthetic workload — the CPU mirror of the GPU engine's
 * semantics (DESIGN.md "Monte-Carlo"): trials {t0 + i*stride : i < T}; per trial one noise vector shared
 * by all (L, Eb/N0) (PolarCode.cpp:708-710); ascending Eb/N0 with the "decoded at a lower
 * Eb/N0 => counted as run, not simulated" hack (:728-742).  Early stop (:725) is evaluated
 * by the CALLER between batches (batch-granular), so this routine takes an `enabled` mask.
 * err/run are uint64 [n_L][n_e] accumulators. */
static void orc_mc_batch_impl(void *h, int cid, uint64_t seed, uint64_t t0, long T, long stride, const double *ebno, int n_e,
                  const uint8_t *Ls, int n_L, const uint8_t *enabled, uint64_t *err, uint64_t *run) {
    orc_t *c = (orc_t *)h;
    int N = c->N, K = c->K;
    double *llr = (double *)malloc(sizeof(double) * N);
    uint8_t *info = (uint8_t *)malloc((size_t)K), *dec = (uint8_t *)malloc((size_t)K);
    uint8_t *prev = (uint8_t *)malloc((size_t)n_e);
    for (long t = 0; t < T; ++t) {
        for (int li = 0; li < n_L; ++li) {
            memset(prev, 0, (size_t)n_e);
            for (int ie = 0; ie < n_e; ++ie) {
                if (!enabled[li * n_e + ie]) continue;
                run[li * n_e + ie]++;
                int run_sim = 1;
                for (int j = 0; j < ie; ++j) if (prev[j]) run_sim = 0;
                if (!run_sim) continue;
                if (cid == 0) {
                    double s = orc_snr_sqrt_linear(h, ebno[ie]);
                    orc_synth_llr(h, seed, t0 + (uint64_t)t * (uint64_t)stride, s, llr, info);
                } else {
                    orc_synth_bicm_llr(h, cid, seed, t0 + (uint64_t)t * (uint64_t)stride, ebno[ie], llr, info);
                }
                orc_decode_scl_llr(c, llr, Ls[li], dec, NULL);
                int e = 0;
                for (int i = 0; i < K; ++i) if (info[i] != dec[i]) { e = 1; break; }
                if (e) err[li * n_e + ie]++; else prev[ie] = 1;
            }
        }
    }
    free(llr); free(info); free(dec); free(prev);
}
void orc_mc_batch(void *h, uint64_t seed, uint64_t t0, long T, long stride, const double *ebno, int n_e,
                  const uint8_t *Ls, int n_L, const uint8_t *enabled, uint64_t *err, uint64_t *run) {
    orc_mc_batch_impl(h, 0, seed, t0, T, stride, ebno, n_e, Ls, n_L, enabled, err, run);
}
/* same on the ASK/BICM channel: the sweep axis is the SNR in dB (main_MC_CC_Comparison.m:42,88) */
void orc_mc_batch_bicm(void *h, int cid, uint64_t seed, uint64_t t0, long T, long stride, const double *snr_db, int n_s,
                       const uint8_t *Ls, int n_L, const uint8_t *enabled, uint64_t *err, uint64_t *run) {
    orc_mc_batch_impl(h, cid, seed, t0, T, stride, snr_db, n_s, Ls, n_L, enabled, err, run);
}

/* ---------- Monte-Carlo code construction: PolarM PolarCode.m:143-196 `monte_carlo` (receiver
 * 'bicm') with `polar_encode` (:855-867) and the genie-aided `polar_decode_monte` (:897-914),
 * restated as the same RECURSIONS the MATLAB text uses (the device kernel is iterative and keeps
 * layers in bit-reversed order, so this is an independent formulation). MATLAB cannot run in the
 * build image: "parity unpinned" by the reference for this part; the random inputs come from the
 * counter-based generator of include/polar_synth.h instead of MATLAB's rand/randn. ---------- */
static void mc_encode_rec(const uint8_t *u, int N, uint8_t *x) {             /* :855-867 */
    if (N == 1) { x[0] = u[0]; return; }
    uint8_t *a = (uint8_t *)malloc((size_t)N);
    uint8_t *b = a + N / 2;
    for (int k = 0; k < N / 2; ++k) { a[k] = (uint8_t)((u[2 * k] + u[2 * k + 1]) % 2); b[k] = u[2 * k + 1]; }
    mc_encode_rec(a, N / 2, x);
    mc_encode_rec(b, N / 2, x + N / 2);
    free(a);
}
static double mc_cnop(double w1, double w2) { return w1 * (1 - w2) + w2 * (1 - w1); }            /* :889-891 */
static double mc_vnop(double w1, double w2) { return w1 * w2 / (w1 * w2 + (1 - w1) * (1 - w2)); } /* :893-895 */
static void mc_decode_monte_rec(const double *y, const uint8_t *info, int N, double *x, uint8_t *ber) {   /* :897-914 */
    if (N == 1) {
        if ((y[0] > 0.5 && info[0] == 1) || (y[0] <= 0.5 && info[0] == 0)) ber[0] = 0; else ber[0] = 1;
        x[0] = (double)info[0];
        return;
    }
    double *est = (double *)malloc(sizeof(double) * (size_t)N * 2);
    double *x1 = est + N / 2, *x2 = x1 + N / 2;
    for (int k = 0; k < N / 2; ++k) est[k] = mc_cnop(y[2 * k], y[2 * k + 1]);
    mc_decode_monte_rec(est, info, N / 2, x1, ber);
    for (int k = 0; k < N / 2; ++k) est[k] = mc_vnop(mc_cnop(x1[k], y[2 * k]), y[2 * k + 1]);
    mc_decode_monte_rec(est, info + N / 2, N / 2, x2, ber + N / 2);
    for (int k = 0; k < N / 2; ++k) { x[2 * k] = mc_cnop(x1[k], x2[k]); x[2 * k + 1] = x2[k]; }
    free(est);
}
/* one run: fills p1[N] (may be NULL) and adds the per-position error flags to num_err[N] */
void orc_mc_construction_run(int n, int cid, double design_snr_db, uint64_t seed, uint64_t trial, double *p1_out, uint64_t *num_err) {
    const int N = 1 << n, nb = osy_nbits(cid), nsym = N / nb;
    uint8_t *info = (uint8_t *)malloc((size_t)N * 3), *coded = info + N, *ber = coded + N;
    double *p1 = (double *)malloc(sizeof(double) * (size_t)N * 2), *x = p1 + N;
    for (int i = 0; i < N; ++i) info[i] = (uint8_t)osy_bit(seed, trial, OSY_MCINFO, i);
    mc_encode_rec(info, N, coded);
    const double norm = osy_norm(cid);
    const double sigma = sqrt(1.0 / 2) * pow(10.0, -design_snr_db / 20);      /* :170 */
    const double n0 = sigma * sigma;
    for (int i = 0; i < N; ++i) p1[i] = 0.5;                                   /* :176 */
    for (int i = 0; i < nsym; ++i) {
        int sym = 0;
        for (int j = 0; j < nb; ++j) sym += (1 << j) * coded[i * nb + j];
        double xs = osy_point(cid, sym) / norm;
        double y = xs + sigma * osy_symbol_noise(seed, trial, (uint32_t)i);   /* :171-172 */
        osy_demap(cid, norm, y, n0, NULL, p1 + (size_t)i * nb);               /* :178 */
    }
    mc_decode_monte_rec(p1, info, N, x, ber);
    for (int i = 0; i < N; ++i) num_err[i] += ber[i];
    if (p1_out) memcpy(p1_out, p1, sizeof(double) * (size_t)N);
    free(info); free(p1);
}
void orc_mc_construction(int n, int cid, double design_snr_db, uint64_t seed, uint64_t trial0, long num_runs, uint64_t *num_err) {
    for (long t = 0; t < num_runs; ++t) orc_mc_construction_run(n, cid, design_snr_db, seed, trial0 + (uint64_t)t, NULL, num_err);
}
