// pv_launch.h -- host-callable launchers for the kernels in pv_kernels.hip
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "pv_device.h"

namespace pva {

// supported (K steps per launch, interior rows per tile) instantiations of the fused stencil
bool stepConfigSupported(int K, int rxi);
bool mergedConfigOk(int K, int rxi);
// stacked tiles (W waves share one tall tile): always one merged launch; their blocks load
// stepConfigExtraRows() rows beyond rxi + 2K at the bottom, which the guard band must cover
bool stepConfigStacked(int K, int rxi);
int stepConfigExtraRows(int K, int rxi);
// which: bit 0 = air-tile kernel, bit 1 = general-tile kernel (both write disjoint tiles of the same planes);
// 4 = both in ONE merged launch (one block per general tile first, then the air tiles)
// The general kernel goes to stream2 when given (the caller orders the two streams with events).
void launchStep(int K, int rxi, const StepArgs& a, hipStream_t stream, int which = 3, hipStream_t stream2 = nullptr);
// which = 4 | kStepGeneralPacked: the merged launch whose general-tile arm is the packed one also at K = 12 (scenes with many wall tiles)
constexpr int kStepGeneralPacked = 32;
// persistent patch kernel (pv_patch.h): the AIR tiles of one sweep by `blocks` resident workgroups (one per CU, multiple
// of 8); the general tiles of the sweep are launched with launchStep(..., which = 16)
bool patchConfigOk(int K, int rxi);
void launchStepPatch(int K, int rxi, const StepArgs& a, int blocks, hipStream_t stream);
// sparse-emitter mode with the forward sums inside the stencil (pv_stream.h): per K-step launch the classify pass, the merged
// launch with the per-launch tile classes, and the open half tiles
bool openConfigOk(int K, int rxi);
bool unpackedAirOk();
void launchStreamClassify(const ClassifyArgs& c, hipStream_t stream);
void launchStreamIdle(const ClassifyArgs& c, int* idleHost, hipStream_t stream);
void launchStepOpen(int K, int rxi, const StepArgs& a, const OpenArgs& o, hipStream_t stream);
// row-streaming air segments (pv_seg.h): columns per lane of the configuration's segment kernel (0 = it has none), the
// tile columns a segment can span, and the launch (general tiles + a.numSeg segments in one grid)
int segConfigColumns(int K, int rxi);
int segConfigMaxTileColumns(int K, int rxi);
void launchStepSeg(int K, int rxi, const StepArgs& a, hipStream_t stream);
// batched merged launch: ba.n runs (blockIdx.y) of identically configured solvers in one grid; only for
// batchConfigOk() configurations
bool batchConfigOk(int K, int rxi);
bool edgeConfigOk(int K, int rxi);  // the batched kernel of this configuration has the edge-tile arm (tile class 2)
void launchBatch(int K, int rxi, const BatchArgs& ba, hipStream_t stream);
// tile classes: 0 air, 1 general (also appended to `list`), 2 edge tile (only when allowEdge and the configuration
// has the mirror-pair air tile)
void launchTileClass(int K, int rxi, const FaceCoef* coef, uint8_t* tileClass, int* list, int* count,
                     const Geometry& g, hipStream_t stream, bool allowEdge);
// dead tiles (all-wall interior): dead[tile] = 1, *count += number of them
void launchTileDead(const FaceCoef* coef, uint8_t* dead, int* count, const Geometry& g, int K, hipStream_t stream);
// resident kernel (pv_resident.hip): one launch per run; a.ntiles workgroups that must all be co-resident
bool residentConfigOk(int K, int rxi);
int residentExtraRows(int K, int rxi);   // rows its blocks load beyond rxi + 2K at the bottom
int residentMaxBlocks(int K, int rxi, int device);
void launchResident(int K, int rxi, const ResidentArgs& a, hipStream_t stream);
// do the two (idle) streams share a hardware queue?  stamps = 4 words of pinned host memory (pv_probe.hip)
bool streamsShareQueue(hipStream_t a, hipStream_t b, unsigned long long* stamps);
// shader clock of the moment (pv_probe.hip): MHz by a timed s_sleep, or 0
float clockProbeMHz(int device, float* byMemtime);
// the device's own streaming bandwidth in GB/s: {copy 16 B per lane, copy 4 B per lane, read only, write only} (pv_probe.hip)
bool bandwidthProbeGBs(int device, float out[4]);
// error flag, {cells of non-zero tiles, cells with an onset}, resident claim counter (or NULL) -> 4 ints of pinned host memory
void launchRunStatus(int* err, int* counts, const unsigned* claims, int* outHost, hipStream_t stream);
#ifdef PV_RESIDENT_TRACE
void residentDumpTrace();  // development builds: the phase stamps of the last launch to stderr
#endif
// cells = NX*NY must satisfy smallGridFits()
bool smallGridFits(int NX, int NY);
void launchSmallGrid(const SmallArgs& a, hipStream_t stream);
void launchZero(float* p, long long n, hipStream_t stream);
void launchBeginRun(const BeginArgs& a, hipStream_t stream);
void launchCoefs(const float* mat, FaceCoef* coef, const Geometry& g, hipStream_t stream);
void launchLaneSelfTest(float* out128, hipStream_t stream);
void launchAnalysis(const AnalyzeArgs& a, hipStream_t stream);
// the three phases of launchAnalysis separately (slab groups run the middle one per slab, the others on the whole map)
void launchFarCells(const AnalyzeArgs& a, hipStream_t stream);
void launchAnalysisFar(const AnalyzeArgs& a, hipStream_t stream);  // launchAnalysis' first pass (the lazy far frame, or every far cell)
// no-onset cells of the window take the six persistent result planes (occlusion, wet gain, decay time, lowpass, source
// direction x / y) from another solver's maps (pv_rt60.hip; Solver::run's carryFrom)
void launchCarryResults(const AnalyzeArgs& a, const float* srcOut, hipStream_t stream);
void launchAnalysisCells(const AnalyzeArgs& a, hipStream_t stream);  // = the three below, one after the other
void launchOnset(const AnalyzeArgs& a, hipStream_t stream);   // onsets of the window's cells into the delay map
void launchEncode(const AnalyzeArgs& a, hipStream_t stream);  // dry gain, source direction, low-pass (reads the onsets)
void launchRt60(const AnalyzeArgs& a, hipStream_t stream);    // wet gain, decay time (reads the onsets)
void launchAnalysisDirection(const AnalyzeArgs& a, hipStream_t stream);
// the whole analysis of a grid whose history window is the grid, in one launch (pv_fused.hip); fusedAnalysisOk: its phase
// counters fit; launchRunFinish: a run's output queries + status words in one launch (last kernel of a run)
bool fusedAnalysisBuilt();  // false in the product build: the arm lives in the experimental build only
bool fusedAnalysisOk(const AnalyzeArgs& a);
void launchAnalysisFused(const FusedArgs& f, hipStream_t stream);
// (zeroWords / nZero: words to clear behind everything else -- the resident kernel's flags, for the next run; with them the error flag)
void launchRunFinish(const float* res, long long n, const long long* cellsHost, int nq, float* outHost, const FarInfo& far,
                     int* err, int* counts, const unsigned* claims, int* statusHost, unsigned* zeroWords, int nZero,
                     unsigned long long* stamp, hipStream_t stream);
// wet gain + decay time (pv_rt60.hip): sixteen / four lanes per cell in one launch, the lane-per-cell form in a second one where
// AnalyzeArgs::rt60Tile announces it; the form is chosen on the device
void launchRt60Forms(const AnalyzeArgs& a, hipStream_t stream);
// slab halos: src[i] -> dst[i] for up to six blocks of n floats (n % 4 == 0, 16-byte aligned); dst[i] = NULL skips a block
void launchHaloPush(const float* const src[6], float* const dst[6], long long n, const HaloHandoff& hand, hipStream_t stream);
void launchHistRow(const AnalyzeArgs& a, int X, float* outTxPitch, hipStream_t stream);
void launchCopyBlock(const float* src, long long sstride, int spitch, int sr0, int sc0, float* dst, long long dstride,
                     int dpitch, int dr0, int dc0, int nr, int nc, int nplanes, const int* srcPlanesDev,
                     const int* dstPlanesDev, hipStream_t stream, const unsigned* abortWord = nullptr);  // plane maps: NULL = identity
// (far: where the last run's far cells begin -- their listener direction is computed, not read: pv_device.h FarInfo)
void launchGatherQueries(const float* res, long long n, const long long* cellsHost, int nq, float* outHost, const FarInfo& far,
                         hipStream_t stream);  // nq <= 64
void launchGatherOutput(const float* res, long long n, long long cell, float* out8Host, const FarInfo& far, hipStream_t stream);
// direction planes of the far cells, for whole-map read-backs; delay plane = "no onset" everywhere (solver creation)
void launchFarDirections(float* res, long long n, const FarInfo& far, hipStream_t stream);
void launchFillDelay(float* delay, long long n, hipStream_t stream);
void launchPackResults(const float* res, long long n, float* res8, hipStream_t stream);
void launchPackWindow(const float* res, long long n, int gy, int r0, int c0, int nr, int nc, float* out8, const FarInfo& far,
                      hipStream_t stream);
void launchStreamAccum(const AnalyzeArgs& a, const uint8_t* hasEmitter, uint8_t* tileOpen, int ntiles,
                       hipStream_t stream);
void launchStreamFinalize(const AnalyzeArgs& a, hipStream_t stream);
void launchEfree(const float* hist, long long plane, long long cellOff, int n, float r, float* out,
                 hipStream_t stream);
void launchIr(const AnalyzeArgs& a, int X, int Y, float* out3T, hipStream_t stream);
void launchUnpad(const float* padded, float* dense, const Geometry& g, hipStream_t stream);
void launchPad(const float* dense, float* padded, const Geometry& g, hipStream_t stream);
void launchHistPlane(const AnalyzeArgs& a, int t, float* dense, int NX, int NY, int histRows,
                     hipStream_t stream);

}  // namespace pva
