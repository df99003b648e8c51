// tools/libm_check.cpp -- pv_libm.h against the host libm, bit for bit.
//   g++ -O2 -ffp-contract=off -std=c++17 -I planeverb_amd/csrc tools/libm_check.cpp -o /tmp/libm_check
//   /tmp/libm_check [stride]      stride 1 = every float (about 80 s), default 97
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "pv_libm.h"

int main(int argc, char** argv) {
    const unsigned stride = argc > 1 ? (unsigned)std::atoi(argv[1]) : 97u;
    long n = 0, badLog = 0, badPow = 0;
    for (unsigned long long u = 0; u <= 0x7f800000ull; u += stride) {
        const float x = pva::pvFloatBits((uint32_t)u);
        if (u >= 1 && u < 0x7f800000ull && pva::pvBitsF(pva::pvLog10f(x)) != pva::pvBitsF(std::log10(x))) ++badLog;
        if (u >= 1 && u < 0x7f800000ull && pva::pvBitsF(pva::pvLog10fNonNeg(x)) != pva::pvBitsF(std::log10(x))) ++badLog;
        // the fast form's domain: positive normal finite floats (pv_rt60.hip's lane-per-cell kernel)
        if (u >= 0x00800000ull && u < 0x7f800000ull &&
            (!pva::pvIsNormalPositive(x) || pva::pvBitsF(pva::pvLog10fNormal(x)) != pva::pvBitsF(std::log10(x))))
            ++badLog;
        if (u < 0x00800000ull && pva::pvIsNormalPositive(x)) ++badLog;
        if (pva::pvBitsF(pva::pvPowf(x, 0.8f)) != pva::pvBitsF(std::pow(x, 0.8f))) ++badPow;
        ++n;
    }
    // the special values the analysis can feed log10f
    const float specials[] = {0.f, -0.f, -1.f, INFINITY, NAN};
    for (float s : specials) {
        const float a = pva::pvLog10f(s), b = std::log10(s);
        if (!((a != a && b != b) || pva::pvBitsF(a) == pva::pvBitsF(b))) ++badLog;
    }
    if (pva::pvIsNormalPositive(INFINITY) || pva::pvIsNormalPositive(NAN) || pva::pvIsNormalPositive(-1.f) ||
        pva::pvIsNormalPositive(0.f) || !pva::pvIsNormalPositive(1.f) || !pva::pvIsNormalPositive(1.17549435e-38f))
        ++badLog;
    {  // the staged batch form = the scalar form (same operations): a sample of normal floats through both
        for (unsigned long long u = 0x00800000ull; u + 8 * 4099 < 0x7f800000ull; u += 7919ull * stride) {
            float in[8], o[8];
            for (int n = 0; n < 8; ++n) in[n] = pva::pvFloatBits((uint32_t)(u + 4099ull * n));
            pva::pvLog10fNormalBatch(in, o, pva::PvLogTabConst{});
            for (int n = 0; n < 8; ++n)
                if (pva::pvBitsF(o[n]) != pva::pvBitsF(std::log10(in[n]))) ++badLog;
        }
    }
    const float nonneg[] = {0.f, INFINITY, NAN, 1.f, 1.17549435e-38f, 1e-45f};  // the branch-free form's domain
    for (float s : nonneg) {
        const float a = pva::pvLog10fNonNeg(s), b = std::log10(s);
        if (!((a != a && b != b) || pva::pvBitsF(a) == pva::pvBitsF(b))) ++badLog;
    }
    std::printf("{\"values\": %ld, \"stride\": %u, \"log10f_mismatches\": %ld, \"powf_mismatches\": %ld}\n", n, stride,
                badLog, badPow);
    return (badLog || badPow) ? 1 : 0;
}
