/* Test infrastructure: runs the kernel's sinf/cosf sequence (orbslamm_amd/csrc/orbx_sincosf.h, the
 * same text the HIP kernel compiles) on the host against libm's cosf / sinf -- what the reference's
 * `(float)cos(angle)` with a float argument under `using namespace std` means (ORBextractor.cc:65,112-113).
 * usage: sincos_check [step]   step 1 = every binary32 argument in [0, 2pi] (1 086 918 621 values, ~12 s): bad=0.
 * build: gcc -O2 -ffp-contract=off -o sincos_check sincos_check.c -lm     (the kernel's arithmetic)
 *        gcc -O2 -mfma -ffp-contract=fast ...                            (glibc's *_fma variants): also bad=0
 * Also counts how often (float)cos((double)angle) -- round 1's reading of the source -- differs from cosf. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include "../../orbslamm_amd/csrc/orbx_sincosf.h"
int main(int argc, char** argv)
{
    const float hi = 6.2831860f;
    uint32_t hb, step = 1;
    memcpy(&hb, &hi, 4);
    if (argc > 1) step = (uint32_t)atoi(argv[1]);
    long bad = 0, n = 0, dbl = 0;
    for (uint32_t b = 0; b <= hb; b += step) {
        float f, s, c;
        memcpy(&f, &b, 4);
        orbx_sincosf_0_2pi(f, &s, &c);
        const float gs = sinf(f), gc = cosf(f);
        const float ds = (float)sin((double)f), dc = (float)cos((double)f);
        n++;
        if (memcmp(&ds, &gs, 4) || memcmp(&dc, &gc, 4)) dbl++;
        if (memcmp(&s, &gs, 4) || memcmp(&c, &gc, 4)) {
            if (bad < 10) printf("x=%.9g mine %.9g %.9g libm %.9g %.9g\n", f, s, c, gs, gc);
            bad++;
        }
    }
    printf("n=%ld bad=%ld double_then_cast_differs=%ld\n", n, bad, dbl);
    return 0;
}
