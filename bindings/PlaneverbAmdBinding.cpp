// PlaneverbAmdBinding.cpp -- link with -lplaneverb_amd instead of ProjectPlaneverb.lib
#include <Planeverb.h>
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
}  // namespace Planeverb
