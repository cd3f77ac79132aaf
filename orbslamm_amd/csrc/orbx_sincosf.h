// sinf / cosf of a binary32 angle in [0, 2*pi], as glibc >= 2.28 computes them (sysdeps/ieee754/flt-32/s_sinf.c,
// s_cosf.c, sincosf.h, s_sincosf_data.c -- the ARM optimized-routines algorithm): the argument widened to binary64,
// quadrant n = round(x * 2/pi) through a 2^24-scaled float->int conversion, r = x - n * (pi/2) with ONE double
// constant, then a degree-7 odd / degree-8 even polynomial in binary64, rounded ONCE to binary32.
//
// Why this is here: the reference writes `(float)cos(angle)`, `(float)sin(angle)` with a float `angle` under
// `using namespace std;` (ORBextractor.cc:65,112-113), i.e. std::cos(float) = cosf.  That step is therefore
// libm-version dependent in the reference itself; the CPU checker calls the host libm, the kernel evaluates this
// sequence.  The file is plain C so that tests/cpp/sincos_check.c runs THIS VERY TEXT on the host against libm's
// cosf/sinf over every binary32 angle in [0, 2*pi] (1 086 918 621 values): identical, both when every operation is
// rounded separately (-ffp-contract=off, how the kernel is built) and when the compiler fuses them (-mfma
// -ffp-contract=fast, how glibc's x86-64 *_fma ifunc variants are built).  Valid for 0 <= y <= 2*pi only: the
// |y| < 2^-12 shortcuts of the library (sin y = y, cos y = 1) are what the polynomial rounds to anyway, larger
// arguments (> 120) take another reduction that a keypoint angle never needs.
#ifndef ORBX_SINCOSF_H
#define ORBX_SINCOSF_H
#ifndef ORBX_HD
#define ORBX_HD static inline
#endif
ORBX_HD void orbx_sincosf_0_2pi(float y, float* sn, float* cs)
{
    const double x = (double)y;
    // 2/pi * 2^24, truncated toward zero, + 2^23, arithmetic shift: the quadrant 0..4
    const int n = ((int)(x * 0x1.45F306DC9C883p+23) + 0x800000) >> 24;
    const double r = x - (double)n * 0x1.921FB54442D18p0;
    const double r2 = r * r;
    const double sg = ((n + 1) & 2) ? -1.0 : 1.0;  // sign[n & 3] = {1, -1, -1, 1}
    const double q = (n & 2) ? -1.0 : 1.0;         // second coefficient table: the cosine polynomial negated
    // odd polynomial on r * sign
    const double xs = r * sg;
    const double x3 = xs * r2;
    const double s1 = 0x1.1107605230bc4p-7 + r2 * -0x1.994eb3774cf24p-13;
    const double x7 = x3 * r2;
    const double s = xs + x3 * -0x1.555545995a603p-3;
    const float podd = (float)(s + x7 * s1);
    // even polynomial
    const double x4 = r2 * r2;
    const double c2 = q * -0x1.6c087e89a359dp-10 + r2 * (q * 0x1.99343027bf8c3p-16);
    const double c1 = q * 0x1p0 + r2 * (q * -0x1.ffffffd0c621cp-2);
    const double x6 = x4 * r2;
    const double c = c1 + x4 * (q * 0x1.55553e1068f19p-5);
    const float peven = (float)(c + x6 * c2);
    // sinf evaluates sinf_poly(.., n), cosf sinf_poly(.., n ^ 1): odd polynomial when the low bit is clear
    *sn = (n & 1) ? peven : podd;
    *cs = (n & 1) ? podd : peven;
}
#endif
