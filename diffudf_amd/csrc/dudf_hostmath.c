/* Host build of dudf_math.h so the sin/cos used by the HIP sweeps can be checked on CPU
 * (tests/test_host_math.py).  Not part of the product path. */
#include "dudf_math.h"
void dudf_host_sincos(const float* x, float* s, float* c, long n) {
    for (long i = 0; i < n; ++i) dudf_sincos(x[i], &s[i], &c[i]);
}
