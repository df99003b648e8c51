// PlaneverbAmdBinding.cpp -- link with -lplaneverb_amd instead of ProjectPlaneverb.lib
#include <Planeverb.h>
#include <vector>
#include "planeverb_amd.h"
namespace Planeverb {
void Init(const PlaneverbConfig* c) {
    PlaneverbInit(c->gridSizeInMeters.x, c->gridSizeInMeters.y, c->gridResolution, (int)c->gridBoundaryType,
                  const_cast<char*>(c->tempFileDirectory), (int)c->maxThreadUsage, /*pv_GPU*/ 1);
    if (!PlaneverbIsRunning()) throw pv_InvalidConfig;              // PvContext.cpp:101-107
}
void Exit() { PlaneverbExit(); }
void ChangeSettings(const PlaneverbConfig* c) { Exit(); Init(c); }  // PvContext.cpp:46-50
EmissionID Emit(const vec3& p) { return (EmissionID)PlaneverbEmit(p.x, p.y, p.z); }
void UpdateEmission(EmissionID id, const vec3& p) { PlaneverbUpdateEmission((int)id, p.x, p.y, p.z); }
void EndEmission(EmissionID id) { PlaneverbEndEmission((int)id); }
PlaneverbOutput GetOutput(EmissionID id) {
    ::PlaneverbOutput o = PlaneverbGetOutput((int)id);
    PlaneverbOutput r;
    r.occlusion = o.occlusion; r.wetGain = o.wetGain; r.rt60 = o.rt60; r.lowpass = o.lowpass;
    r.direction = vec2(o.directionX, o.directionY);
    r.sourceDirectivity = vec2(o.sourceDirectionX, o.sourceDirectionY);
    return r;
}
PlaneObjectID AddGeometry(const AABB* t) {
    return (PlaneObjectID)PlaneverbAddGeometry(t->position.x, t->position.y, t->width, t->height, t->absorption);
}
void UpdateGeometry(PlaneObjectID id, const AABB* t) {
    PlaneverbUpdateGeometry((int)id, t->position.x, t->position.y, t->width, t->height, t->absorption);
}
void RemoveGeometry(PlaneObjectID id) { PlaneverbRemoveGeometry((int)id); }
void SetListenerPosition(const vec3& p) { PlaneverbSetListenerPosition(p.x, p.y, p.z); }
// Planeverb.h:47, FDTD.cpp:60-79.  The reference returns a pointer into its IR cube; here the AoS Cells are
// materialised on demand (the GPU keeps a pressure history and re-derives vx, vy) into a buffer this binding owns:
// valid until the calling thread's next GetImpulseResponse, like upstream's until the next iteration overwrites it.
std::pair<const Cell*, unsigned> GetImpulseResponse(const vec3& p) {
    static_assert(sizeof(Cell) == sizeof(PlaneverbCell), "PvTypes.h:106-121 vs planeverb_amd.h");
    static thread_local std::vector<Cell> buf;
    const int T = PlaneverbGetImpulseResponse(p.x, p.y, p.z, nullptr, 0);
    if (T <= 0) return std::make_pair((const Cell*)nullptr, 0u);
    buf.resize((size_t)T);
    if (PlaneverbGetImpulseResponse(p.x, p.y, p.z, reinterpret_cast<PlaneverbCell*>(buf.data()), T) != T)
        return std::make_pair((const Cell*)nullptr, 0u);
    return std::make_pair((const Cell*)buf.data(), (unsigned)T);
}
}  // namespace Planeverb
