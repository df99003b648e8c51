// pv_core.h -- host-side grid arithmetic, scene rasteriser and .pv I/O for the MI355X Planeverb solver.
//
// Everything here is index / parameter arithmetic that the reference does on the host in float32 and that must
// be replicated literally: true division where the reference divides, multiplication by the
// reciprocal where it multiplies.  This file is compiled WITHOUT fast-math and with -ffp-contract=off.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pva {

// reference constants: ProjectPlaneverb/include/PvTypes.h:83-101
constexpr float kC = 343.21f;
constexpr float kAudibleThreshold = 0.00000316f;
constexpr float kDryDirectionLen = 0.005f;
constexpr float kDryGainLen = 0.01f;
constexpr float kWetGainLen = 0.080f;
constexpr float kSqrt2 = 1.4142136f;
constexpr float kPointsPerWavelength = 3.5f;
constexpr float kSchroederOffset = 0.01f;
constexpr float kDistanceGainThreshold = 0.891251f;
constexpr float kDelayCloseThreshold = 5.f;
constexpr float kInvalidDryGain = -1.f;
constexpr int kLowResolution = 275;

struct GridSpec {
    int res = 0;
    float sizeX = 0, sizeY = 0;  // metres
    float dx = 0, dt = 0;
    unsigned fs = 0;
    int T = 0;                   // response length
    float gsx = 0, gsy = 0;      // the reference's float m_gridSize (Grid.cpp:48-49)
    int gx = 0, gy = 0;          // (int)m_gridSize
    int NX = 0, NY = 0;          // cell array = (gx+1) x (gy+1), index x*NY + y (FDTD.cpp:99)
    float courant = 0;           // FDTD.cpp:90
    int nDir = 0, nDry = 0, nWet = 0, nCut = 0;  // Analyzer.cpp:170-171,237,286
    int nFree = 0;               // FreeGrid.cpp:99
};

// Grid.cpp:390-396, Grid.cpp:46-55
GridSpec makeGridSpec(float sizeX, float sizeY, int res);
// same parameters but explicit cell counts (used for the free-field window, FreeGrid.cpp:71-94)
GridSpec makeGridSpecCells(int gx, int gy, int res);
// Grid.cpp:12-27
std::vector<float> gaussianPulse(const GridSpec& g);
// does the host's expf reproduce the reference's pulse table?  (checked once per process, a warning on stderr if not)
bool pulseMatchesReferenceLibm();
void warnIfPulseDiffers();
// FDTD.cpp:97-98 : (int)((pos + offset) / dx)
void listenerCell(const GridSpec& g, float lx, float lz, int* cx, int* cy);
// Analyzer.cpp:200-201 : (int)(pos * (1/dx))
void listenerCellRecip(const GridSpec& g, float lx, float lz, int* cx, int* cy);
// Analyzer.cpp:106-116 ; returns false when the reference returns nullptr (with >= instead of >)
bool resultCell(const GridSpec& g, float ex, float ez, int* cx, int* cy);

struct Box {
    float x, y, w, h, R;  // PvMathTypes.h:31-49 : centre, full extents, absorption parameter
};

// beta / R planes with the reference's rasteriser semantics (Grid.cpp:84-108,136-144,229-296)
class MaterialPlane {
public:
    void init(const GridSpec& g);
    void add(const Box& b);     // Grid::AddAABB
    void remove(const Box& b);  // Grid::RemoveAABB (clears overlaps of other boxes too: SURVEY Q4)
    const std::vector<uint8_t>& beta() const { return beta_; }
    const std::vector<float>& R() const { return R_; }
    // Cell::by (PvTypes.h:113): never read by the solver, but part of the AoS Cell GetImpulseResponse hands out
    const std::vector<uint8_t>& by() const { return by_; }
    // dirty row range [lo, hi) since the last clearDirty(); empty when lo >= hi
    int dirtyLo() const { return dirtyLo_; }
    int dirtyHi() const { return dirtyHi_; }
    void clearDirty();
    void markAllDirty();

private:
    void bounds(const Box& b, int* sx, int* sy, int* ex, int* ey) const;
    GridSpec g_;
    std::vector<uint8_t> beta_, by_;
    std::vector<float> R_;
    int dirtyLo_ = 0, dirtyHi_ = 0;
};

// .pv scene files: PlaneverbSandbox/src/Editor/Editor.cpp:219-281
bool loadPv(const std::string& path, std::vector<Box>* out, std::string* err);
bool savePv(const std::string& path, const std::vector<std::pair<int, Box>>& boxes, std::string* err);

// PlaneverbDSP/src/PvDSPContext.cpp:165-228
void reverbBusGains(float rt60, float wet, float* a, float* b, float* c);

// Row-streaming air segments (pv_seg.h, PVA_OPT_STREAM_ROWS): cover every tile with air[ti * nty + tj] != 0 by exactly
// one segment.  Tile rows are cut into maximal runs of air tiles and those into chunks of <= wmax tile columns; identical
// chunks of consecutive tile rows form a rectangle; rectangles are cut into pieces of about equal height -- ROW-granular,
// not tile-granular -- of about (total rows / target) rows, at most 7 * rxi (a segment may touch 8 tile rows).  Sorted by
// (first row, tile column).  Pure index arithmetic (no reference counterpart).
struct SegRect {
    int row0, nrows, tj0, w;  // array rows [row0, row0 + nrows) x tile columns [tj0, tj0 + w)
};
std::vector<SegRect> planSegments(const uint8_t* air, int ntx, int nty, int rxi, int wmax, int target);

}  // namespace pva
