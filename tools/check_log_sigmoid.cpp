// Host check of the log-sigmoid arithmetic of sage_kernels.h (log_sigmoid): the same operations in the same order,
// with v_rcp_f64 replaced by a float reciprocal (a worse start than the hardware gives), against libm.
// g++ -O2 -ffp-contract=off -o /tmp/check_ls tools/check_log_sigmoid.cpp && /tmp/check_ls
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
static inline double rcp_approx(double d) { return (double)(1.0f / (float)d); }   // stand-in for v_rcp_f64's limited accuracy
static inline double log_sigmoid_fast(double x) {
    double t = -fabs(x);
    t = t < -750.0 ? -750.0 : t;
    const double n = rint(t * 1.4426950408889634);
    double r = fma(n, -6.93147180369123816490e-01, t);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double u = ldexp(p, (int)n);
    const double z = 1.0 + u;
    const double e = u - (z - 1.0);
    const bool big = z > 1.4142135623730951;
    const double m = big ? 0.5 * z : z;
    const double f = m - 1.0, d = m + 1.0;
    double y = rcp_approx(d);
    y = y * fma(-d, y, 2.0);
    y = y * fma(-d, y, 2.0);
    double s = f * y;
    s = fma(fma(-s, d, f), y, s);
    const double s2 = s * s;
    double q = 1.0 / 23.0;
    q = fma(q, s2, 1.0 / 21.0);
    q = fma(q, s2, 1.0 / 19.0);
    q = fma(q, s2, 1.0 / 17.0);
    q = fma(q, s2, 1.0 / 15.0);
    q = fma(q, s2, 1.0 / 13.0);
    q = fma(q, s2, 1.0 / 11.0);
    q = fma(q, s2, 1.0 / 9.0);
    q = fma(q, s2, 1.0 / 7.0);
    q = fma(q, s2, 1.0 / 5.0);
    q = fma(q, s2, 1.0 / 3.0);
    // log(m) = 2s + 2s*s2*q
    double lg = fma(2.0 * s * s2, q, 2.0 * s);
    lg = lg + (big ? 6.93147180559945286227e-01 : 0.0);
    lg = lg + (big ? 0.5 * e : e) ;   // d log(z)/dz * e, z ~ 1 (or e/z ~ e/2 for the halved case is within 1e-16 either way)
    return fmin(x, 0.0) - lg;
}
int main() {
    double worst = 0, worst_x = 0; long nbad = 0;
    srand(1);
    for (long i = 0; i < 20000000; ++i) {
        double x;
        const int kind = i % 4;
        const double r = (rand() + 0.5) / (RAND_MAX + 1.0);
        if (kind == 0) x = (r - 0.5) * 2.0;            // small dots
        else if (kind == 1) x = (r - 0.5) * 80.0;
        else if (kind == 2) x = (r - 0.5) * 1600.0;
        else x = (r - 0.5) * 1e-6;
        const double ref = fmin(x, 0.0) - log1p(exp(-fabs(x)));
        const double got = log_sigmoid_fast(x);
        const double err = fabs(got - ref) / fmax(fabs(ref), 1e-300);
        if (err > worst) { worst = err; worst_x = x; }
        if (err > 1e-14) ++nbad;
    }
    printf("worst rel err %.3e at x=%.17g, >1e-14: %ld\n", worst, worst_x, nbad);
    double sp[] = {0.0, -0.0, 1e-300, -1e-300, 745.0, -745.0, 800.0, -800.0, INFINITY, -INFINITY, NAN, 36.7, -36.7, 0.881373587, -0.881373587};
    for (double x : sp) printf("x=%g fast=%.17g ref=%.17g\n", x, log_sigmoid_fast(x), fmin(x, 0.0) - log1p(exp(-fabs(x))));
}
