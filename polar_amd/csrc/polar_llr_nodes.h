// polar_llr_nodes.h — LLR-domain node arithmetic of the list kernels (the reference's own operation order with table-driven
// exp / log1p): f-node, g-node, the path-metric terms. Included by polar_kernels.hip only (one translation unit, three builds).
#pragma once
namespace {

// ---- fp64 transcendentals for the f-node and the path metric --------------------------------
// The reference evaluates log((e^(a+b)+1)/(e^a+e^b)) and log(1+e^x) with libm. Bit-identity with
// glibc's exp/log is not reachable on a GPU (ocml differs in the last ulp as well); what parity
// needs is that every DECISION (sign of a leaf LLR, order of path metrics) is the reference's,
// i.e. an absolute accuracy far below any decision margin. These routines keep ~1e-16 absolute
// accuracy (the rounding level of the reference's own 1+e^x) at ~1/3 of the instruction count
// of the libm-style sequence, using two small LDS tables (no division):
//   exp(-x) = T[k&63] * 2^(k>>6) * p5(s),  -x = k*ln2/64 + s  (k <= 0, T[f] = 2^(f/64))
//   log(m)  = LC[j] + log1p((m - c_j)/c_j),   c_j = 1 + j/128 = m rounded to 7 mantissa bits
// and the identity  f(a,b) = sgn(a)sgn(b)min(|a|,|b|) + h(|a+b|) - h(|a-b|),  h(x) = log1p(e^-x).
// Structural exactness is preserved: h(x) == 0 exactly for x >= 36.74 (where the reference's
// 1+e^-x rounds to 1), f(0,b) == 0 exactly, f is symmetric, log(1+e^x) -> +inf for x > 709.78.
__device__ __forceinline__ double h_fn(double x, const Tabs &tb) {       // log1p(e^-x), x >= 0
    return log_1p2(1.0 + exp_neg(x, tb), tb);
}
// h(x) - h(y): the two evaluations of an f-node written in lockstep, so that their table reads are
// issued together (2 LDS round trips per f-node instead of 4) and the two dependent fp64 chains
// overlap. Same operations and rounding as h_fn(x) - h_fn(y).
__device__ __forceinline__ double h_diff(double x, double y, const Tabs &tb) {
    const double kx = __builtin_rint(x * -92.332482616893657), ky = __builtin_rint(y * -92.332482616893657);
    const int ix = (int)kx, iy = (int)ky;
    const double tx = tb.T[ix & 63], ty = tb.T[iy & 63];
    double sx = __builtin_fma(kx, -0.010830424696223417, -x), sy = __builtin_fma(ky, -0.010830424696223417, -y);
    sx = __builtin_fma(kx, -2.5728046223276688e-14, sx); sy = __builtin_fma(ky, -2.5728046223276688e-14, sy);
    double px = sx * (1.0 / 120.0) + 1.0 / 24.0, py = sy * (1.0 / 120.0) + 1.0 / 24.0;
    px = __builtin_fma(px, sx, 1.0 / 6.0); py = __builtin_fma(py, sy, 1.0 / 6.0);
    px = __builtin_fma(px, sx, 0.5); py = __builtin_fma(py, sy, 0.5);
    px = __builtin_fma(px, sx, 1.0); py = __builtin_fma(py, sy, 1.0);
    px = __builtin_fma(px, sx, 1.0); py = __builtin_fma(py, sy, 1.0);
    const double mx = 1.0 + __builtin_ldexp(tx * px, ix >> 6), my = 1.0 + __builtin_ldexp(ty * py, iy >> 6);
    double cx, cy;
    const int nx = log_slot(mx, cx), ny = log_slot(my, cy);
    const double rcx = tb.RC[nx], rcy = tb.RC[ny], lcx = tb.LC[nx], lcy = tb.LC[ny];
    const double qx = (mx - cx) * rcx, qy = (my - cy) * rcy;
    double ux = qx * (-1.0 / 6.0) + 0.2, uy = qy * (-1.0 / 6.0) + 0.2;
    ux = __builtin_fma(ux, qx, -0.25); uy = __builtin_fma(uy, qy, -0.25);
    ux = __builtin_fma(ux, qx, 1.0 / 3.0); uy = __builtin_fma(uy, qy, 1.0 / 3.0);
    ux = __builtin_fma(ux, qx, -0.5); uy = __builtin_fma(uy, qy, -0.5);
    ux = __builtin_fma(ux, qx, 1.0); uy = __builtin_fma(uy, qy, 1.0);
    return __builtin_fma(qx, ux, lcx) - __builtin_fma(qy, uy, lcy);
}
__device__ __attribute__((noinline)) double f_literal(double a, double b) {
    return log((exp(a + b) + 1) / (exp(a) + exp(b)));
}
__device__ __attribute__((noinline)) double softplus_literal(double x) { return log(1 + exp(x)); }
// f-node (check node), exact + min-sum branches: PolarCode.cpp:437-446
__device__ __forceinline__ double f_node(double a, double b, const Tabs &tb) {
    const double fa = fabs(a), fb = fabs(b);
    const double mx = __builtin_fmax(fa, fb);      // (v_max_f64 / v_min_f64 with |.| source modifiers)
    const double mn = __builtin_fmin(fa, fb);
    // sgn(a)*sgn(b)*min(|a|,|b|) (PolarCode.cpp:443-445, sgn(0) = 0): the magnitude with the XOR of the
    // two sign bits, done on the high word instead of int->double conversions and multiplies; it is
    // also the leading term of the exact expression below
    const int sx = (__double2hiint(a) ^ __double2hiint(b)) & (int)0x80000000;
    const double ms = __hiloint2double(__double2hiint(mn) | sx, __double2loint(mn));
    if (40 > mx) {
        // |f| <= min(|a|,|b|): when that is within a few orders of the rounding noise (1e-16) the
        // reference's result IS its rounding noise (e.g. exactly 0 once e^a, e^b round to 1), so the
        // literal expression is evaluated for those (physically never occurring) elements.
        if (POLAR_UNLIKELY2(mn < 9.5367431640625e-07)) return f_literal(a, b);
        return ms + h_diff(fabs(a + b), fabs(a - b), tb);
    }
    return (mn == 0.0) ? 0.0 : ms;     // min-sum branch
}
// Two f-nodes at once: same results as f_node() twice, but ONE wave-uniform branch around the two exact
// evaluations, so that their four h() chains sit in one basic block and overlap (a per-node divergent
// branch serialises the nodes). Used where the nodes are otherwise strictly serial (rate-0 blocks); in
// the unrolled layer loops it was measured slower (register pressure: -1.4 % fused loop, -25 % LDS visits). Lanes that do not need the exact value compute it on whatever they
// hold (finite garbage at worst: the table index is masked) and discard it.
__device__ __forceinline__ void f_node2(double a0, double b0, double a1, double b1, const Tabs &tb, double &r0, double &r1) {
    const double fa0 = fabs(a0), fb0 = fabs(b0), fa1 = fabs(a1), fb1 = fabs(b1);
    const double mx0 = __builtin_fmax(fa0, fb0), mn0 = __builtin_fmin(fa0, fb0);
    const double mx1 = __builtin_fmax(fa1, fb1), mn1 = __builtin_fmin(fa1, fb1);
    const int s0 = (__double2hiint(a0) ^ __double2hiint(b0)) & (int)0x80000000;
    const int s1 = (__double2hiint(a1) ^ __double2hiint(b1)) & (int)0x80000000;
    const double m0 = __hiloint2double(__double2hiint(mn0) | s0, __double2loint(mn0));     // sgn*sgn*min (min-sum value)
    const double m1 = __hiloint2double(__double2hiint(mn1) | s1, __double2loint(mn1));
    r0 = (mn0 == 0.0) ? 0.0 : m0;
    r1 = (mn1 == 0.0) ? 0.0 : m1;
    const bool e0 = 40 > mx0, e1 = 40 > mx1;
    if (wave_any(e0 || e1)) {
        const double x0 = m0 + h_diff(fabs(a0 + b0), fabs(a0 - b0), tb);
        const double x1 = m1 + h_diff(fabs(a1 + b1), fabs(a1 - b1), tb);
        const bool t0 = e0 && mn0 < 9.5367431640625e-07, t1 = e1 && mn1 < 9.5367431640625e-07;
        if (e0) r0 = x0;
        if (e1) r1 = x1;
        if (POLAR_UNLIKELY2(wave_any(t0 || t1))) {                    // noise regime (see f_node)
            if (t0) r0 = f_literal(a0, b0);
            if (t1) r1 = f_literal(a1, b1);
        }
    }
}
// g-node: PolarCode.cpp:449-450  (1 - 2u)*a + b
__device__ __forceinline__ double g_node(double a, double b, unsigned u) {
    // (1 - 2u) is +1 or -1 and the product with it is exact: flip the sign bit of a, then add
    const double sa = __hiloint2double(__double2hiint(a) ^ (int)(u << 31), __double2loint(a));
    return sa + b;
}
// The path-metric terms log(1 + exp(-+llr)) of PolarCode.cpp:483,505-506 for a = |llr| >= 0, with ONE
// h evaluation: log(1+e^-a) = h(a) (exactly 0 for a >= 36.74, where the reference's 1+e^-a rounds to
// 1), log(1+e^a) = a + h(a) (+inf beyond the fp64 exp overflow point 709.78, as the reference).
// `skip` (wave-uniform: every lane has a >= 37) avoids the transcendental altogether.
__device__ __forceinline__ void softplus_pair(double a, bool skip, const Tabs &tb, double &sneg, double &spos) {
    double hx = 0.0;
    if (!skip) {
        if (POLAR_UNLIKELY2(a < 9.5367431640625e-07)) {            // noise regime: literal expressions (see f_node)
            sneg = softplus_literal(-a);
            spos = softplus_literal(a);
            return;
        }
        // per-lane saturation: 1 + e^-a rounds to 1 for a >= 37, and an infinite (or > 1e78) leaf LLR must not
        // reach the range reduction of exp_neg (inf * c - inf = NaN would poison the metric)
        hx = (a >= 37.0) ? 0.0 : h_fn(__builtin_fmin(a, 37.0), tb);
    }
    sneg = hx;
    spos = (a > 709.782712893384) ? __builtin_inf() : a + hx;
}


// leaf terms for the path metric. LLR-domain kernel: `leaf` is the LLR; E-domain: stored form.
//   neg  = (llr < 0);  al = |llr|;  sneg = log(1+e^-|llr|);  spos = log(1+e^|llr|)
// actw: the wave mask of `active` (kept by the caller: a ballot of a compound bool costs a round trip through a VGPR)
template <bool ED>
__device__ __forceinline__ void leaf_terms(double leaf, bool active, u64 actw, const Tabs &tb, bool &neg, double &al, double &sneg, double &spos) {
    if (!ED) {
        al = fabs(leaf);
        neg = leaf < 0;
        const bool skip = wave_all(!active || al >= 37.0);
        sneg = 0.0; spos = 0.0;
        if (active) softplus_pair(al, skip, tb, sneg, spos);
    } else {
        const double m = fabs(leaf);
        const bool isl = m > 1.0;
        neg = (__double2hiint(leaf) < 0) && m != 1.0;
        al = m;
        const u64 m_e = actw & __builtin_amdgcn_fcmp(m, 1.0, 13);           // ULE: active lanes holding an E-form value
        if (POLAR_LIKELY2(m_e != 0)) {
            const double l = -ed_log(__builtin_fmin(__builtin_fmax(m, ED_EMIN), 1.0), tb);
            if (!isl) al = l;
        }
        const double onep = 1.0 + m;                 // == 1 exactly from E <= 2^-53 on, as the reference's 1 + e^-|x|
        sneg = 0.0;
        if (POLAR_LIKELY2((m_e & __builtin_amdgcn_fcmp(onep, 1.0, 14)) != 0)) {      // UNE
            const double h = log_1p2(__builtin_fmin(onep, 2.0), tb);
            if (!isl) sneg = h;
        }
        spos = (al > 709.782712893384) ? __builtin_inf() : al + sneg;
    }
}
// The same terms for the one-codeword-per-wave kernels (exp-domain), K leaves at once (round 6): a lone wave pays an LDS round trip
// for the table look-up of every logarithm, and the wave-uniform skips above put each of the 2 K logarithms into a basic block of
// its own — 2 K round trips one after the other. Here all look-ups sit in ONE block (their reads issue together, one wait); the
// values are those of leaf_terms<true> bit for bit: a lane the skips would have left alone computes log(1) = 0 (E <= 2^-53:
// 1 + E == 1 -> q = 0 -> +0) or is overridden by its L-form select.
template <int K>
__device__ __forceinline__ void leaf_terms_ed_block(const double (&leaf)[K], u64 actw, const Tabs &tb, bool (&neg)[K], double (&al)[K], double (&sneg)[K], double (&spos)[K]) {
    double m[K];
    u64 any_e = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        m[i] = fabs(leaf[i]);
        neg[i] = (__double2hiint(leaf[i]) < 0) && m[i] != 1.0;
        al[i] = m[i];
        sneg[i] = 0.0;
        any_e |= actw & __builtin_amdgcn_fcmp(m[i], 1.0, 13);           // ULE: active lanes holding an E-form value
    }
    if (POLAR_LIKELY2(any_e != 0)) {
        // ed_log / log_1p2 (polar_edom.h) split into "table slot" and "polynomial": the 4 K table values are read first — the empty asm
        // needs all of them, so the reads are issued back to back and waited for once — then the polynomials run
        double l[K], h[K], qa[K], qb[K], ea[K], rca[K], lca[K], rcb[K], lcb[K];
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const double x = __builtin_fmin(__builtin_fmax(m[i], ED_EMIN), 1.0);        // ed_log(x): exponent, mantissa in [1, 2)
            const int hi = __double2hiint(x);
            ea[i] = (double)((hi >> 20) - 1023);
            const double ma = __hiloint2double((hi & 0x000FFFFF) | 0x3FF00000, __double2loint(x));
            const double mb = __builtin_fmin(1.0 + m[i], 2.0);                          // log_1p2(1 + E)
            double ca, cb;
            const int ja = log_slot(ma, ca), jb = log_slot(mb, cb);
            rca[i] = tb.RC[ja]; lca[i] = tb.LC[ja]; rcb[i] = tb.RC[jb]; lcb[i] = tb.LC[jb];
            qa[i] = ma - ca; qb[i] = mb - cb;
        }
#pragma unroll
        for (int i = 0; i < K; ++i) asm volatile("" : "+v"(rca[i]), "+v"(lca[i]), "+v"(rcb[i]), "+v"(lcb[i]));
        auto poly = [](double q, double lc) {
            double p = q * (-1.0 / 6.0) + 0.2;
            p = __builtin_fma(p, q, -0.25);
            p = __builtin_fma(p, q, 1.0 / 3.0);
            p = __builtin_fma(p, q, -0.5);
            p = __builtin_fma(p, q, 1.0);
            return __builtin_fma(q, p, lc);
        };
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const double la = poly(qa[i] * rca[i], lca[i]);
            l[i] = -__builtin_fma(ea[i], 6.93147180369123816490e-01, __builtin_fma(ea[i], 1.90821492927058770002e-10, la));
            h[i] = poly(qb[i] * rcb[i], lcb[i]);
        }
#pragma unroll
        for (int i = 0; i < K; ++i)
            if (!(m[i] > 1.0)) { al[i] = l[i]; sneg[i] = h[i]; }
    }
#pragma unroll
    for (int i = 0; i < K; ++i) spos[i] = (al[i] > 709.782712893384) ? __builtin_inf() : al[i] + sneg[i];
}
}  // namespace
