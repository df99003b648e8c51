// tests/host/sandbox_probe.cpp -- TEST INFRASTRUCTURE: a caller of the reference's C++ API written against the
// reference's OWN header (ProjectPlaneverb/include/Planeverb.h) and nothing else, doing what PlaneverbSandbox does
// (main.cpp:14-21 config + Init; Editor.cpp:245-281 LoadGeometry -> AddGeometry; Editor.cpp:36-37 listener / emitter;
// AudioCore.cpp:95 GetOutput; Editor.cpp:45,457 GetImpulseResponse).  It is LINKED against
// bindings/PlaneverbAmdBinding.cpp + libplaneverb_amd.so instead of ProjectPlaneverb.lib: that the link succeeds with
// all 12 namespace functions resolved, and that the numbers it prints equal the reference's, is the drop-in claim of
// INTEGRATION.md section 2.  Built here (where /root/reference exists) by `make -C oracle ref` into
// oracle/_ref/sandbox_probe; run by tests/test_gpu_live.py on the GPU box.
//
//   sandbox_probe scene.pv lx lz ex ez [irx irz]...     prints one JSON object
#include <Planeverb.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <thread>
#include <vector>

static unsigned bits(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    Planeverb::PlaneverbConfig config;  // main.cpp:14-19
    config.gridResolution = Planeverb::pv_LowResolution;
    config.gridBoundaryType = Planeverb::pv_AbsorbingBoundary;
    config.gridSizeInMeters = Planeverb::vec2(25.f, 25.f);
    config.tempFileDirectory = ".";
    config.maxThreadUsage = 1;
    try {
        Planeverb::Init(&config);
    } catch (Planeverb::PlaneverbErrorCode e) {
        std::fprintf(stderr, "Init threw %d\n", (int)e);
        return 3;
    }
    Planeverb::SetListenerPosition(Planeverb::vec3((float)std::atof(argv[2]), 0.f, (float)std::atof(argv[3])));
    {  // Editor.cpp:245-281
        std::ifstream f(argv[1]);
        size_t n = 0;
        f >> n;
        for (size_t i = 0; i < n; ++i) {
            size_t id;
            Planeverb::AABB a;
            f >> id >> a.position.x >> a.position.y >> a.width >> a.height >> a.absorption;
            Planeverb::AddGeometry(&a);
        }
    }
    const Planeverb::EmissionID e = Planeverb::Emit(Planeverb::vec3((float)std::atof(argv[4]), 0.f, (float)std::atof(argv[5])));
    // the reference API has no "iteration finished" call: the Sandbox just polls every frame.  One second is hundreds
    // of iterations of the 71^2 grid; geometry queued above is live from the second one on (PvContext.cpp:80-90).
    std::this_thread::sleep_for(std::chrono::milliseconds(1000));
    const Planeverb::PlaneverbOutput o = Planeverb::GetOutput(e);
    std::printf("{\"emitter_id\": %u, \"output_bits\": [%u, %u, %u, %u, %u, %u, %u, %u], \"irs\": [", (unsigned)e,
                bits(o.occlusion), bits(o.wetGain), bits(o.rt60), bits(o.lowpass), bits(o.direction.x),
                bits(o.direction.y), bits(o.sourceDirectivity.x), bits(o.sourceDirectivity.y));
    for (int a = 6; a + 1 < argc; a += 2) {
        const auto ir = Planeverb::GetImpulseResponse(Planeverb::vec3((float)std::atof(argv[a]), 0.f, (float)std::atof(argv[a + 1])));
        std::printf("%s{\"n\": %u, \"cells_hex\": \"", a > 6 ? ", " : "", ir.second);
        const unsigned char* p = reinterpret_cast<const unsigned char*>(ir.first);
        for (size_t i = 0; i < (size_t)ir.second * sizeof(Planeverb::Cell); ++i) std::printf("%02x", p[i]);
        std::printf("\"}");
    }
    // an emitter off the grid, ChangeSettings, Update / Remove / EndEmission: the rest of the 12 functions
    Planeverb::UpdateEmission(e, Planeverb::vec3(40.f, 0.f, 5.f));
    const float offGrid = Planeverb::GetOutput(e).occlusion;
    Planeverb::AABB moved;
    moved.position = Planeverb::vec2(3.f, 3.f);
    moved.width = 1.f;
    moved.height = 1.f;
    moved.absorption = 0.5f;
    const Planeverb::PlaneObjectID g = Planeverb::AddGeometry(&moved);
    moved.position.x = 4.f;
    Planeverb::UpdateGeometry(g, &moved);
    Planeverb::RemoveGeometry(g);
    Planeverb::EndEmission(e);
    config.gridSizeInMeters = Planeverb::vec2(10.f, 10.f);
    Planeverb::ChangeSettings(&config);
    Planeverb::Exit();
    std::printf("], \"off_grid_occlusion\": %g, \"geometry_id\": %u}\n", offGrid, (unsigned)g);
    return 0;
}
