// oracle/ref_dsp_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// the reverb-bus split FindGainA / FindGainB / FindGainC lives in an anonymous namespace of
// PlaneverbDSP/src/PvDSPContext.cpp:165-228, so the only way to call the reference's own compiled code is to pull
// that UNMODIFIED file into this translation unit (compiled where it lies under /root/reference, never copied) and
// export three C wrappers behind it.  Built by `make -C oracle ref` into oracle/_ref/libpvrefdsp.so together with
// the three DSP sources the file's classes need to link.
#include <cmath>
namespace std {
using ::atan2f;  // MSVC exposes std::atan2f, libstdc++ does not (Context::SubmitSource, PvDSPContext.cpp:299-310 --
}                // not on the FindGain path): the C library's own atan2f made visible under that name, nothing more
#include <PvDSPContext.cpp>

extern "C" {
float pvrefdsp_find_gain_a(float rt60, float wet) { return PlaneverbDSP::FindGainA(rt60, wet); }
float pvrefdsp_find_gain_b(float rt60, float wet) { return PlaneverbDSP::FindGainB(rt60, wet); }
float pvrefdsp_find_gain_c(float rt60, float wet) { return PlaneverbDSP::FindGainC(rt60, wet); }
}
